"""The part of parity that can be pinned to something the reference itself holds: its TEXT.

tests/golden/reference_literals.json is written by tests/golden/extract_reference_literals.py from the reference's
sources (numeric literals and struct field orders only, with file:line). These tests assert that the oracle
(oracle/orc_math.h, oracle/wfpt_oracle.c), the device code (csrc/wfpt_device_math.h, csrc/wfpt_kernels.hip), the host
restatement (csrc/wfpt_host.cpp), the boundary header (include/wfpt.h) and the host mirrors carry exactly those values.
Floating-point results of WGSL built-ins remain "parity unpinned" (DESIGN.md section 2); this shrinks the unpinned
surface to them. No GPU needed; /root/reference is only touched by the staleness check, which is skipped without it.
"""
import inspect
import json
import os
import re

import numpy as np
import struct
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "reference_literals.json")
CSRC = os.path.join(ROOT, "wavefront_path_tracer_amd", "csrc")


@pytest.fixture(scope="module")
def ref():
    with open(GOLD) as f:
        return json.load(f)


def text(*parts):
    with open(os.path.join(ROOT, *parts)) as f:
        return f.read()


def f32(x):
    return struct.unpack("<f", struct.pack("<f", x))[0]


def c_float_literals(src):
    """Every C float literal of `src` as an f32 value."""
    return {f32(float(m)) for m in re.findall(r"(?<![\w.])(\d+\.\d*(?:[eE][+-]?\d+)?|\d+[eE][+-]?\d+)f\b", src)}


def function_body(src, name):
    m = re.search(r"\b" + re.escape(name) + r"\s*\([^)]*\)\s*(?://[^\n]*)?\s*\{", src)
    assert m, f"function {name} not found"
    depth, i = 1, m.end()
    while depth:
        depth += {"{": 1, "}": -1}.get(src[i], 0)
        i += 1
    return src[m.end():i - 1]


def both(entries):
    """shade.wgsl and generate_rays.wgsl carry identical copies of the RNG: the two extractions must agree."""
    values = [e["value"] for e in entries]
    assert len(set(map(str, values))) == 1, f"the reference's two copies differ: {entries}"
    return values[0]


def test_committed_json_is_current(tmp_path):
    """With the reference present (build container), re-extract into a scratch file and compare with the committed one."""
    if not os.path.isdir("/root/reference/gpu_wavefront_pt/shaders"):
        pytest.skip("/root/reference is not present on this box; the committed extraction is used as is")
    fresh = str(tmp_path / "fresh.json")
    subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "extract_reference_literals.py"), "/root/reference", fresh],
                   check=True, stdout=subprocess.DEVNULL)
    with open(fresh) as f:
        assert json.load(f) == json.load(open(GOLD)), "reference_literals.json is stale: re-run the extractor and commit"


@pytest.mark.parametrize("path,prefix", [(("oracle", "orc_math.h"), "orc_"), (("wavefront_path_tracer_amd", "csrc", "wfpt_device_math.h"), "")])
def test_rng_constants(ref, path, prefix):
    """PCG-RXS-M-XS-32, the as-written advance, jenkins and the 2^-32 scale (shade.wgsl:218-266 == generate_rays.wgsl:133-181)."""
    src = text(*path)
    r = ref["rng"]
    nxt = function_body(src, prefix + "rng_next_int")
    mult, plus = both(r["lcg_mult"]), both(r["lcg_plus"])
    assert re.search(rf"\* {mult}u \+ {plus}u", nxt), "LCG step"
    assert re.search(rf">> {both(r['rxs_shift_base'])}u?\) \+ {both(r['rxs_shift_add'])}u", nxt), "RXS shift"
    assert f"* {both(r['mcg_mult'])}u" in nxt, "M multiplier"
    assert re.search(rf"\(word >> {both(r['xs_shift'])}u?\) \^ word", nxt), "XS shift"
    adv = function_body(src, prefix + "advance")
    assert f"cur_mult = {both(r['advance_cur_mult'])}u" in adv and f"cur_plus = {both(r['advance_cur_plus'])}u" in adv
    assert "delta == 1" in adv, "the reference's advance accumulates only when delta == 1 (generate_rays.wgsl:162)"
    jen = function_body(src, prefix + "jenkins_hash")
    steps = [(m[0], m[1], int(m[2])) for m in re.findall(r"x (\+=|\^=) x (<<|>>) (\d+)", jen)]
    want = [[(s["op"], s["dir"], s["shift"]) for s in copy] for copy in r["jenkins_steps"]]
    assert want[0] == want[1] and steps == want[0], f"jenkins steps {steps} vs reference {want[0]}"
    scale = both(r["u32_to_float_scale"])
    assert f32(scale) == 2.0 ** -32 and f32(scale) in c_float_literals(src), "u32 -> f32 scale"


def test_kernel_literals(ref):
    """Window, sentinels, sampler exponent, pi and the sky colours in the device kernels and in the oracle."""
    dev, orc = text("wavefront_path_tracer_amd", "csrc", "wfpt_kernels.hip"), text("oracle", "wfpt_oracle.c")
    ex, sh = ref["extend"], ref["shade"]
    assert ex["no_hit"]["value"] == ex["no_hit_test"]["value"] == ex["box_miss"]["value"] == 1e30
    assert ex["use_bvh"]["value"] == "true"
    for name, src in (("device", dev), ("oracle", orc)):
        lits = c_float_literals(src)
        for what, v in (("no-hit sentinel", ex["no_hit"]["value"]), ("t_min", ex["t_min"]["value"]),
                        ("unit-sphere exponent", sh["unit_sphere_pow_exponent"]["value"]),
                        ("degenerate direction", sh["degenerate_direction_length"]["value"]), ("pi", sh["pi"]["value"])):
            assert f32(v) in lits, f"{name}: {what} {v} missing"
        for c in ref["miss_kernel"]["sky_white"]["value"] + ref["miss_kernel"]["sky_blue"]["value"]:
            assert f32(c) in lits, f"{name}: sky component {c} missing"
    assert ref["generate_rays"]["pi"]["value"] == sh["pi"]["value"]
    # where exactly: the hit window and the sampler
    assert re.search(r"t > 0\.001f && t < nearest", function_body(dev, "hit_prim"))
    assert "0.33333f" in function_body(dev, "rng_next_in_unit_sphere")
    assert re.search(r"kPi = 3\.1415927f", dev) and re.search(r"ORC_PI 3\.1415927f", orc)
    # the oracle keeps the reference's box-miss value (extend.wgsl:181); the device deliberately reports a LARGER one, so that a
    # pair of missed boxes is not walked down while `nearest` still holds the no-hit sentinel of the same value
    assert re.search(r"return 1e30f;", function_body(orc, "hit_bvh_node"))
    # WFPT_FLAG_EXACT_TRAVERSAL (template EXACT) restores the reference's value on the device too
    assert re.search(r"\? \(EXACT \? 1e30f : kBoxMiss\) : tmin", function_body(dev, "hit_bvh_node"))
    m = re.search(r"constexpr float kBoxMiss = ([0-9.e+]+)f;", dev)
    assert m and float(m.group(1)) > ex["box_miss"]["value"] and np.isfinite(np.float32(m.group(1)))


def test_builder_and_loop_constants(ref):
    bins = ref["bvh"]["bins"]["value"]
    assert re.search(rf"#define ORC_BINS {bins}\b", text("oracle", "wfpt_oracle.c"))
    assert re.search(rf"kReferenceBins = {bins};", text("wavefront_path_tracer_amd", "csrc", "wfpt_host.cpp"))
    lp = ref["loop"]
    sys.path.insert(0, ROOT)
    import wavefront_path_tracer_amd as W
    assert (W.SPP, W.SPF) == (lp["spp"]["value"], lp["spf"]["value"])
    sig = inspect.signature(W.PathTracer.__init__).parameters
    assert sig["max_wavefronts"].default == lp["max_wavefronts"]["value"]
    assert sig["miss_floor"].default == lp["miss_floor"]["value"]
    hpp = text("wavefront_path_tracer_amd", "host", "wfpt.hpp")
    assert re.search(rf"SPP = {lp['spp']['value']};", hpp) and re.search(rf"SPF = {lp['spf']['value']};", hpp)
    assert re.search(rf"max_wavefronts = {lp['max_wavefronts']['value']};", hpp)
    assert re.search(rf"miss_floor = {lp['miss_floor']['value']};", hpp)
    # the traversal must cope with at least the reference's stack depth (extend.wgsl:38); the oracle's explicit stack is larger
    m = re.search(r"#define ORC_MAX_STACK (\d+)", text("oracle", "wfpt_oracle.c"))
    assert int(m.group(1)) >= ref["extend"]["stack_size"]["value"]
    # stage names are the shader basenames (kernel.rs:32)
    for name in ref["stage_names"]:
        assert name in W.STAGES, f"stage {name!r} missing from the stage API"


C_TYPE = {"Vec4": ("float", 4), "Vec3": ("float", 3), "f32": ("float", 1), "u32": ("uint32_t", 1),
          "vec4f": ("float", 4), "vec3f": ("float", 3)}
RENAMED = {"camera_position": "position", "pos": "position",  # GPUCamera: camera_controller.rs:164 / generate_rays.wgsl:14
           "aabbMin": "aabb_min", "leftFirst": "left_first", "aabbMax": "aabb_max", "primCount": "prim_count",
           "mat_idx": "material_idx", "mat_type": "material_type", "refract_idx": "refract_index",
           "invDirection": "inv_direction", "defocusRadius": "defocus_radius", "focusDistance": "focus_distance"}


def header_struct(name):
    hdr = re.sub(r"/\*.*?\*/", "", text("include", "wfpt.h"), flags=re.S)
    m = re.search(r"typedef struct " + name + r" \{(.*?)\} " + name + ";", hdr, flags=re.S)
    assert m, f"{name} not in include/wfpt.h"
    out = []
    for ctype, field, dim in re.findall(r"(float|uint32_t)\s+(\w+)(?:\[(\d+)\])?;", m.group(1)):
        out.append((field, ctype, int(dim) if dim else 1))
    return out


@pytest.mark.parametrize("ref_name,c_name", [("Sphere", "wfpt_sphere"), ("Material", "wfpt_material"), ("BVHNode", "wfpt_bvh_node"),
                                              ("GPUFrameBuffer", "wfpt_frame_buffer"), ("GPUCamera", "wfpt_gpu_camera"),
                                              ("wgsl_Ray", "wfpt_ray"), ("wgsl_HitPayload", "wfpt_hit_payload"),
                                              ("wgsl_BVHNode", "wfpt_bvh_node"), ("wgsl_FrameBuffer", "wfpt_frame_buffer"),
                                              ("wgsl_CameraData", "wfpt_gpu_camera")])
def test_struct_field_order(ref, ref_name, c_name):
    """Field order and types of include/wfpt.h's PODs against the reference's #[repr(C)] structs and WGSL structs."""
    want = ref["structs"][ref_name]["fields"]
    got = header_struct(c_name)
    assert len(got) == len(want), f"{c_name}: {got} vs {ref_name} {want}"
    for (field, ctype, dim), (rname, rtype) in zip(got, want):
        assert (ctype, dim) == C_TYPE[rtype], f"{c_name}.{field}: {ctype}[{dim}] vs {ref_name}.{rname}: {rtype}"
        # Sphere.material_type is mat_type in WGSL; HitPayload.mat_type keeps its WGSL name in wfpt_hit_payload
        assert field in (rname, RENAMED.get(rname, rname)), f"{c_name}.{field} vs {ref_name}.{rname}"


def test_wgsl_prefix_structs(ref):
    """The WGSL Sphere / Material structs omit the Rust structs' trailing pad word: a prefix, same order."""
    for wg, rs in (("wgsl_Sphere", "Sphere"), ("wgsl_Material", "Material")):
        w, r = ref["structs"][wg]["fields"], ref["structs"][rs]["fields"]
        assert len(w) == len(r) - 1 and r[-1][0] == "_buffer"
        for (wn, wt), (rn, rt) in zip(w, r):
            assert C_TYPE[wt] == C_TYPE[rt] and RENAMED.get(wn, wn) == rn
