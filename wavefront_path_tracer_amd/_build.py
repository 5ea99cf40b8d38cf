"""In-tree build of libwfpt.so (HIP kernels + C ABI) for gfx950 with hipcc.

hipcc cross-compiles without a GPU. The .so is git-ignored but travels to the GPU box with the tree.
"""
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libwfpt.so")
SOURCES = ["wfpt_kernels.hip", "wfpt_api.hip", "wfpt_bvh_build.hip", "wfpt_host.cpp"]
HEADERS = [os.path.join(CSRC, "wfpt_kernels.h"), os.path.join(CSRC, "wfpt_bvh4.h"), os.path.join(CSRC, "wfpt_device_math.h"),
           os.path.join(ROOT, "include", "wfpt.h")]
# -ffp-contract=off: results must be bit-identical to the oracle, the only fused ops are explicit fmaf.
# Correctly rounded fp32 divide/sqrt is hipcc's default and is requested explicitly anyway.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math", "-Wall", "-Wno-unused-function"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libwfpt.so cannot be built (and there is no CPU fallback)")
    return exe


def stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile every HIP source into wavefront_path_tracer_amd/libwfpt.so; returns the path."""
    if not force and not stale():
        return LIB_PATH
    out = os.environ.get("WFPT_LIB_OUT", LIB_PATH)  # tuning builds: WFPT_EXTRA_FLAGS="-D..." WFPT_LIB_OUT=...
    cmd = [hipcc()] + FLAGS + os.environ.get("WFPT_EXTRA_FLAGS", "").split()
    cmd += ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-o", out]
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout)
    if verbose and res.stdout.strip():
        print(res.stdout)
    return out


if __name__ == "__main__":
    print(build(force=True, verbose=True))
