#!/usr/bin/env python3
"""Reads the reference's SOURCE TEXT (build container only: /root/reference does not travel to the GPU box) and writes
tests/golden/reference_literals.json: the numeric literals and the struct field orders the hot path depends on, each
with the file:line it was found at. tests/test_reference_literals.py then asserts that the oracle, the device code and
include/wfpt.h carry the same values, so the part of "parity" that CAN be pinned to something the reference holds --
its text -- is pinned by a committed, re-runnable extraction instead of by the builder's reading.

What this does NOT pin (stays "parity unpinned"): the floating-point results of the WGSL built-ins (sqrt, sin, cos,
pow, normalize ...), whose precision is backend-defined, and the scene's unseeded thread_rng.

Only values (numbers, identifiers) are extracted -- data, not source text.

    python tests/golden/extract_reference_literals.py [/root/reference [out.json]]
"""
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_literals.json")


def lines(rel):
    with open(os.path.join(REF, rel)) as f:
        return f.read().split("\n")


def find(rel, pattern, group=1, conv=str, nth=0, all_=False):
    """First (or nth, or every) match of `pattern` in file `rel`: (converted value, 'rel:line')."""
    rx = re.compile(pattern)
    hits = []
    for i, ln in enumerate(lines(rel), 1):
        m = rx.search(ln)
        if m:
            hits.append((conv(m.group(group)), f"{rel}:{i}"))
    if not hits:
        raise SystemExit(f"{rel}: pattern {pattern!r} not found -- the reference changed?")
    return hits if all_ else hits[nth]


def lit(value_at):
    v, at = value_at
    return {"value": v, "at": at}


def struct_fields(rel, header_rx, field_rx):
    """Field names (in order) of the struct whose header line matches header_rx; stops at the closing brace."""
    ls = lines(rel)
    rx_h, rx_f = re.compile(header_rx), re.compile(field_rx)
    for i, ln in enumerate(ls):
        if rx_h.search(ln):
            fields = []
            for j in range(i + 1, len(ls)):
                if ls[j].strip().startswith("}"):
                    return {"fields": fields, "at": f"{rel}:{i + 1}-{j + 1}"}
                m = rx_f.search(ls[j])
                if m:
                    fields.append([m.group(1), m.group(2).strip().rstrip(",")])
    raise SystemExit(f"{rel}: struct {header_rx!r} not found")


def jenkins_steps(rel):
    """The (operator, direction, shift) steps of jenkins_hash, in source order."""
    rx = re.compile(r"^\s*x (\+=|\^=) x (<<|>>) (\d+)u;")
    steps = []
    for i, ln in enumerate(lines(rel), 1):
        m = rx.search(ln)
        if m:
            steps.append({"op": m.group(1), "dir": m.group(2), "shift": int(m.group(3)), "at": f"{rel}:{i}"})
    if len(steps) != 5:
        raise SystemExit(f"{rel}: expected the 5 steps of jenkins_hash, found {len(steps)}")
    return steps


RUST_FIELD = r"^\s*(?:pub\s+)?([A-Za-z_][A-Za-z0-9_]*)\s*:\s*([A-Za-z0-9_<>\[\]; ]+)"
WGSL_FIELD = r"^\s*([A-Za-z_][A-Za-z0-9_]*)\s*:\s*([A-Za-z0-9_<>]+)"
SH = "gpu_wavefront_pt/shaders/"

out = {
    "_generated_by": "tests/golden/extract_reference_literals.py (values and identifiers only, no source text)",
    "rng": {
        # PCG-RXS-M-XS-32 (shade.wgsl and generate_rays.wgsl carry identical copies; both are read and must agree)
        "lcg_mult": [lit(find(SH + s, r"\*state \* (\d+)u \+ (\d+)u", 1, int)) for s in ("shade.wgsl", "generate_rays.wgsl")],
        "lcg_plus": [lit(find(SH + s, r"\*state \* (\d+)u \+ (\d+)u", 2, int)) for s in ("shade.wgsl", "generate_rays.wgsl")],
        "rxs_shift_base": [lit(find(SH + s, r"new_state >> \(\(new_state >> (\d+)u\) \+ (\d+)u\)", 1, int)) for s in ("shade.wgsl", "generate_rays.wgsl")],
        "rxs_shift_add": [lit(find(SH + s, r"new_state >> \(\(new_state >> (\d+)u\) \+ (\d+)u\)", 2, int)) for s in ("shade.wgsl", "generate_rays.wgsl")],
        "mcg_mult": [lit(find(SH + s, r"\^ new_state\) \* (\d+)u", 1, int)) for s in ("shade.wgsl", "generate_rays.wgsl")],
        "xs_shift": [lit(find(SH + s, r"\(word >> (\d+)u\) \^ word", 1, int)) for s in ("shade.wgsl", "generate_rays.wgsl")],
        "advance_cur_mult": [lit(find(SH + s, r"var cur_mult = (\d+)u", 1, int)) for s in ("shade.wgsl", "generate_rays.wgsl")],
        "advance_cur_plus": [lit(find(SH + s, r"var cur_plus = (\d+)u", 1, int)) for s in ("shade.wgsl", "generate_rays.wgsl")],
        "u32_to_float_scale": [lit(find(SH + s, r"f32\(x\) \* ([0-9.e+-]+)f", 1, float)) for s in ("shade.wgsl", "generate_rays.wgsl")],
        # jenkins one-at-a-time variant: the five (operator, direction, shift) steps in order
        "jenkins_steps": [jenkins_steps(SH + s) for s in ("shade.wgsl", "generate_rays.wgsl")],
    },
    "shade": {
        "pi": lit(find(SH + "shade.wgsl", r"const PI = ([0-9.]+)f", 1, float)),
        "unit_sphere_pow_exponent": lit(find(SH + "shade.wgsl", r"pow\(rng_next_float\(state\), ([0-9.]+)f\)", 1, float)),
        "degenerate_direction_length": lit(find(SH + "shade.wgsl", r"length\(extension_direction\) < ([0-9.]+)", 1, float)),
    },
    "generate_rays": {
        "pi": lit(find(SH + "generate_rays.wgsl", r"const PI = ([0-9.]+)f", 1, float)),
    },
    "extend": {
        "stack_size": lit(find(SH + "extend.wgsl", r"const STACKSIZE:u32 = (\d+)", 1, int)),
        "no_hit": lit(find(SH + "extend.wgsl", r"var nearest_hit: f32 = ([0-9.e+]+)", 1, float)),
        "no_hit_test": lit(find(SH + "extend.wgsl", r"if nearest_hit < ([0-9.e+]+)", 1, float)),
        "box_miss": lit(find(SH + "extend.wgsl", r"return ([0-9.e+]+);", 1, float)),
        "t_min": lit(find(SH + "extend.wgsl", r"hit\(ray, node\.leftFirst \+ idx, ([0-9.]+), nearest_hit", 1, float)),
        "use_bvh": lit(find(SH + "extend.wgsl", r"const USE_BVH = (true|false)", 1, str)),
    },
    "miss_kernel": {
        "sky_white": lit(find(SH + "miss_kernel.wgsl", r"\(1\.0 - a\) \* vec3f\(([0-9., ]+)\) \+ a \* vec3f\(([0-9., ]+)\)", 1,
                              lambda s: [float(x) for x in s.split(",")])),
        "sky_blue": lit(find(SH + "miss_kernel.wgsl", r"\(1\.0 - a\) \* vec3f\(([0-9., ]+)\) \+ a \* vec3f\(([0-9., ]+)\)", 2,
                             lambda s: [float(x) for x in s.split(",")])),
    },
    "bvh": {"bins": lit(find("wavefront_common/src/bvh.rs", r"const BINS: usize = (\d+)", 1, int))},
    "loop": {
        "max_wavefronts": lit(find("gpu_wavefront_pt/src/path_tracer.rs", r"while wavefront < (\d+)", 1, int)),
        "miss_floor": lit(find("gpu_wavefront_pt/src/path_tracer.rs", r"if num_misses < (\d+)", 1, int)),
        "spp": lit(find("wavefront_common/src/parameters.rs", r"pub const SPP: u32 = (\d+)", 1, int)),
        "spf": lit(find("wavefront_common/src/parameters.rs", r"pub const SPF: u32 = (\d+)", 1, int)),
    },
    "structs": {
        "Sphere": struct_fields("wavefront_common/src/sphere.rs", r"pub struct Sphere", RUST_FIELD),
        "Material": struct_fields("wavefront_common/src/material.rs", r"pub struct Material", RUST_FIELD),
        "BVHNode": struct_fields("wavefront_common/src/bvh.rs", r"pub struct BVHNode", RUST_FIELD),
        "GPUFrameBuffer": struct_fields("wavefront_common/src/gpu_structs.rs", r"pub struct GPUFrameBuffer", RUST_FIELD),
        "GPUCamera": struct_fields("wavefront_common/src/camera_controller.rs", r"pub struct GPUCamera", RUST_FIELD),
        "wgsl_Ray": struct_fields(SH + "extend.wgsl", r"^struct Ray", WGSL_FIELD),
        "wgsl_HitPayload": struct_fields(SH + "extend.wgsl", r"^struct HitPayload", WGSL_FIELD),
        "wgsl_BVHNode": struct_fields(SH + "extend.wgsl", r"^struct BVHNode", WGSL_FIELD),
        "wgsl_Sphere": struct_fields(SH + "extend.wgsl", r"^struct Sphere", WGSL_FIELD),
        "wgsl_Material": struct_fields(SH + "shade.wgsl", r"^struct Material", WGSL_FIELD),
        "wgsl_CameraData": struct_fields(SH + "generate_rays.wgsl", r"^struct CameraData", WGSL_FIELD),
        "wgsl_FrameBuffer": struct_fields(SH + "generate_rays.wgsl", r"^struct FrameBuffer", WGSL_FIELD),
    },
    "stage_names": sorted(os.path.splitext(n)[0] for n in os.listdir(os.path.join(REF, SH)) if n.endswith(".wgsl")),
}

with open(OUT, "w") as fo:
    json.dump(out, fo, indent=1, sort_keys=True)
    fo.write("\n")
print(f"wrote {OUT}")
