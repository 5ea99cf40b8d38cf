"""What the compiler makes of the hot kernels, checked without a GPU (hipcc cross-compiles): the fused bounce kernels, the stage API's extend and
the refill traversal must keep 8 waves per SIMD (<= 64 vector, <= 80 scalar registers: four 512-thread workgroups per CU beside 39.7 KB of LDS each),
the middle launches must not touch scratch, and the LDS walk's inner-node loop must be the hand-written one (descend_asm: DESIGN.md section 4, round 5)
with its instruction budget -- the launch is bound by instruction issue, vector AND scalar, so a regression here is a regression of the headline."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def device_asm(tmp_path_factory):
    from wavefront_path_tracer_amd import _build
    out = tmp_path_factory.mktemp("isa") / "wfpt_kernels.s"
    flags = [f for f in _build.FLAGS if f not in ("-shared", "-fPIC")]
    cmd = [_build.hipcc()] + flags + ["--offload-device-only", "-S", "-I" + os.path.join(ROOT, "include"), "-I" + _build.CSRC, "-o", str(out),
                                      os.path.join(_build.CSRC, "wfpt_kernels.hip")]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert res.returncode == 0, res.stdout[-3000:]
    return open(out).read()


def metadata(asm, mangled_part):
    """The AMDGPU metadata entry (.vgpr_count, .sgpr_count, .private_segment_fixed_size ...) of the kernel whose mangled name holds `mangled_part`."""
    m = re.search(r"\.name:\s+(\S*" + re.escape(mangled_part) + r"\S*)\n(.*?)\.wavefront_size:", asm, re.S)
    assert m, mangled_part
    return {k: int(v) for k, v in re.findall(r"\.(\w+):\s+(\d+)\n", m.group(2))}


def body(asm, mangled_part):
    m = re.search(r"^(_Z\S*" + re.escape(mangled_part) + r"\S*):.*?^\.Lfunc_end", asm, re.S | re.M)
    assert m, mangled_part
    return m.group(0)


KERNELS = {  # mangled fragment: (what it is, scratch bytes allowed)
    "bounce_kernelILi1EjLi0ELb1ELb0E": ("bounce_kernel<middle>, spheres in LDS, default walk", 0),
    # (the first launch keeps the magic numbers of generate_rays' tile-index division in scratch: loop-invariant, re-loaded once per work item)
    "bounce_kernelILi0EjLi0ELb1ELb0E": ("bounce_kernel<first>", 24),
    "bounce_binned_kernelILi1EjLi0ELb0ELi4E": ("bounce_binned_kernel<middle>", 0),
    "extend_kernelILb0EjLi0ELb1ELb0E": ("extend_kernel, spheres in LDS, default walk", 0),
    "refill_kernelILi0ELi1ELb1E": ("refill_kernel<first>, triangles, rays from the dense array", 16),
    "refill_kernelILi1ELi1ELb1E": ("refill_kernel<middle>, triangles, rays from the dense array", 16),
}


@pytest.mark.parametrize("frag", sorted(KERNELS))
def test_hot_kernels_keep_eight_waves_per_simd(device_asm, frag):
    what, scratch = KERNELS[frag]
    md = metadata(device_asm, frag)
    assert md["vgpr_count"] <= 64, (what, md)
    assert md["sgpr_count"] <= 80, (what, md)
    assert md["private_segment_fixed_size"] <= scratch, (what, md)
    assert md.get("agpr_count", 0) == 0, (what, md)


@pytest.mark.parametrize("frag", ["bounce_kernelILi1EjLi0ELb1ELb0E", "bounce_kernelILi0EjLi0ELb1ELb0E", "bounce_binned_kernelILi1EjLi0ELb0ELi4E",
                                  "extend_kernelILb0EjLi0ELb1ELb0E"])
def test_inner_loop_is_the_hand_written_one(device_asm, frag):
    text = body(device_asm, frag)
    m = re.search(r"^\.Lwfpt_loop\d+:\n(.*?)^\.Lwfpt_end\d+:", text, re.S | re.M)
    assert m, "the LDS walk of this kernel does not contain descend_asm's loop"
    ops = re.findall(r"^\t([a-z_0-9]+)", m.group(1), re.M)
    valu = [o for o in ops if o.startswith("v_")]
    salu = [o for o in ops if o.startswith("s_") and not o.startswith(("s_cbranch", "s_branch", "s_waitcnt", "s_nop"))]
    lds = [o for o in ops if o.startswith("ds_")]
    # the whole loop, pop path and parent-table climb included (a descending visit runs 36 vector + 11 scalar of them)
    assert len(valu) <= 52 and len(salu) <= 24 and len(lds) == 7, (len(valu), len(salu), len(lds))
    assert sum(1 for o in valu if o == "v_fma_f32") == 18 and "v_max3_f32" in valu and "v_min3_f32" in valu
