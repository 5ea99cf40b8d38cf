// Links libwfpt.so. WFPT_LIB_DIR names the directory that holds it (default: the in-tree build output,
// ../wavefront_path_tracer_amd, produced by `python -m wavefront_path_tracer_amd._build` with hipcc --offload-arch=gfx950).
// libwfpt.so itself links the HIP runtime; RCCL is opened at run time by the gather entry points only.
use std::env;
use std::path::PathBuf;

fn main() {
    let dir = env::var("WFPT_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("..").join("wavefront_path_tracer_amd")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=wfpt");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=WFPT_LIB_DIR");
    println!("cargo:rerun-if-changed=../include/wfpt.h");
}
