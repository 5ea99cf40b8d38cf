"""The C-ABI shared library: loads, exports every symbol include/wfpt.h declares, struct layouts match the
reference's #[repr(C)] structs, and device entry points fail LOUDLY without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest


def test_library_exports_every_declared_symbol(wf):
    declared = wf.abi_symbols()
    assert len(declared) >= 45
    L = wf.lib()
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, missing
    for name in ("wfpt_create", "wfpt_kernel_run", "wfpt_kernel_timing_us", "wfpt_render_sample", "wfpt_read_accumulated",
                 "wfpt_build_bvh", "wfpt_set_counters", "wfpt_read_counters", "wfpt_swap_ray_queues"):
        assert name in declared


def test_symbols_are_plain_c(wf):
    """nm shows unmangled `T wfpt_*` entries: the boundary is extern "C"."""
    out = subprocess.run(["nm", "-D", "--defined-only", wf._build.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (wfpt_[a-z0-9_]+)$", out, flags=re.M))
    assert set(wf.abi_symbols()) <= exported


def test_header_struct_sizes(wf, tmp_path):
    """Compile include/wfpt.h as C and check sizeof/offsetof against the reference layouts."""
    src = tmp_path / "layout.c"
    src.write_text(r'''
#include <stdio.h>
#include <stddef.h>
#include "wfpt.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu ", sizeof(wfpt_sphere), sizeof(wfpt_material), sizeof(wfpt_bvh_node),
         sizeof(wfpt_gpu_camera), sizeof(wfpt_frame_buffer), sizeof(wfpt_ray), sizeof(wfpt_hit_payload));
  printf("%zu %zu %zu %zu %zu %zu\n", offsetof(wfpt_sphere, radius), offsetof(wfpt_material, fuzz),
         offsetof(wfpt_bvh_node, aabb_max), offsetof(wfpt_gpu_camera, pitch), offsetof(wfpt_ray, inv_direction),
         offsetof(wfpt_ray, pixel_idx));
  return 0;
}''')
    exe = tmp_path / "layout"
    inc = os.path.join(wf._build.ROOT, "include")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", inc, str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()
    assert [int(x) for x in out] == [32, 32, 32, 32, 16, 48, 16, 16, 16, 16, 16, 32, 44]
    assert wf.SPHERE.itemsize == 32 and wf.RAY.itemsize == 48 and wf.HIT.itemsize == 16


def test_stage_names(wf):
    L = wf.lib()
    for name, idx in wf.STAGES.items():
        assert L.wfpt_stage_from_name(name.encode()) == idx
        assert L.wfpt_stage_name(idx).decode() == name
    # kernel.rs:32 loads gpu_wavefront_pt/shaders/{name}.wgsl; these five are the reference's shaders
    assert [L.wfpt_stage_name(i).decode() for i in range(5)] == ["generate_rays", "extend", "shade", "miss_kernel", "accumulate"]
    assert L.wfpt_stage_from_name(b"display") == -1 and L.wfpt_stage_from_name(None) == -1
    assert b"gfx950" in L.wfpt_build_info()


def test_invalid_arguments_are_reported_not_crashed(wf):
    L = wf.lib()
    assert L.wfpt_create(None, None, 0, None, 0, None, 0, None, None, None) is None
    assert b"null" in L.wfpt_last_error(None)
    for fn in (L.wfpt_render_sample, L.wfpt_synchronize, L.wfpt_reset_image, L.wfpt_swap_ray_queues):
        assert fn(None) == -1  # WFPT_ERR_INVALID_ARGUMENT
    assert L.wfpt_kernel_run(None, 1, 1, 1) == -1
    assert L.wfpt_kernel_timing_us(None, 1) == 0.0
    L.wfpt_destroy(None)  # no-op


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="only meaningful on a box without a GPU")
def test_no_gpu_means_loud_failure(wf):
    """The product path has no CPU fallback: creating a context without a device is an error, not a detour."""
    assert wf.device_count() == 0
    with pytest.raises(wf.WfptError) as e:
        wf.shirley_path_tracer(64, 64)
    assert "no HIP device" in str(e.value)
    x = np.ones(4, "<f4")
    with pytest.raises(wf.WfptError):
        wf.selftest_math(0, x)


def test_product_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under the package, include/, the examples or tools/ may mention it (only tests/,
    __graft_entry__.smoke() and bench.py's cpu_baseline leg do)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for base in ("wavefront_path_tracer_amd", "include", "tools", "examples", "wfpt-sys"):
        for dirpath, _, files in os.walk(os.path.join(root, base)):
            for f in files:
                if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp", ".c", ".sh", ".rs")):
                    text = open(os.path.join(dirpath, f)).read()
                    assert "orc_" not in text and "wfpt_oracle" not in text and "from oracle" not in text \
                        and "import oracle" not in text, os.path.join(dirpath, f)


def test_wfpt_sys_crate_matches_the_header(wf):
    """wfpt-sys/ (the Rust -sys crate a maintainer of the reference would depend on; uncompiled here, the image has no Rust):
    src/lib.rs is what tools/gen_wfpt_sys.py generates from include/wfpt.h today, its extern "C" list is exactly the
    header's function list, and every name is exported by the built library."""
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    committed = open(os.path.join(root, "wfpt-sys", "src", "lib.rs")).read()
    fresh = subprocess.run([sys.executable, os.path.join(root, "tools", "gen_wfpt_sys.py"), "--stdout"], check=True,
                           stdout=subprocess.PIPE, text=True).stdout
    assert committed == fresh, "wfpt-sys/src/lib.rs is stale: run python tools/gen_wfpt_sys.py"
    rust_fns = re.findall(r"pub fn (wfpt_[a-z0-9_]+)\(", committed)
    assert sorted(rust_fns) == wf.abi_symbols() and len(set(rust_fns)) == len(rust_fns)
    L = wf.lib()
    for name in rust_fns:
        assert hasattr(L, name), f"libwfpt.so does not export {name}"
    # struct sizes as Rust would lay them out (#[repr(C)], 4-byte scalars only) against the header's static asserts
    sizes = {}
    for name, body in re.findall(r"pub struct (wfpt_\w+) \{(.*?)\n\}", committed, flags=re.S):
        n = 0
        for m in re.finditer(r"pub \w+: (?:\[(?:f32|u32|i32); (\d+)\]|(f32|u32|i32)),", body):
            n += 4 * int(m.group(1)) if m.group(1) else 4
        sizes[name] = n
    assert sizes["wfpt_sphere"] == 32 and sizes["wfpt_material"] == 32 and sizes["wfpt_bvh_node"] == 32
    assert sizes["wfpt_gpu_camera"] == 32 and sizes["wfpt_frame_buffer"] == 16 and sizes["wfpt_ray"] == 48
    assert sizes["wfpt_hit_payload"] == 16 and sizes["wfpt_triangle"] == 48 and sizes["wfpt_params"] == 44
    cargo = open(os.path.join(root, "wfpt-sys", "Cargo.toml")).read()
    assert 'links = "wfpt"' in cargo and os.path.exists(os.path.join(root, "wfpt-sys", "build.rs"))
