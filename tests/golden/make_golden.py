#!/usr/bin/env python3
"""Generates the golden fixtures in this directory with the ORACLE (oracle/wfpt_oracle.c).

The reference ships no fixtures, images or assertions and cannot be built here (Rust + WGSL, no toolchain),
so these vectors are outputs of the build's own CPU restatement: they pin the oracle against drift and let
the GPU tests check full-size renders without re-running the CPU. "Parity unpinned" by the reference itself.

    python tests/golden/make_golden.py        # rewrites *.npz next to this file (about a minute)
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import oracle as O  # noqa: E402
from helpers import inputs_for, make_mesh_oracle, make_oracle, mesh_inputs  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def downsample(acc, w, h, f):
    img = acc.reshape(h, w, 3)
    hh, ww = (h // f) * f, (w // f) * f
    return img[:hh, :ww].reshape(hh // f, f, ww // f, f, 3).mean(axis=(1, 3)).astype("<f4")


def render_case(kind, w, h, spp, bounces, rng_mode, ds):
    o = make_oracle(O, inputs_for(O, kind, w, h), w, h, max_wavefronts=bounces, rng_mode=rng_mode)
    tables = []
    for _ in range(spp):
        o.render_sample()
        tables.append(o.bounce_table())
    acc = o.accumulated()
    out = dict(width=w, height=h, spp=spp, bounces=bounces, rng_mode=rng_mode,
               tables=np.stack([np.pad(t, ((0, bounces - len(t)), (0, 0))) for t in tables]),
               totals=o.totals(), acc_sha256=sha(acc), acc_small=downsample(acc, w, h, ds), downsample=ds)
    o.close()
    return out, acc


def main():
    # (1) the deterministic 5-sphere scene (scene.rs:12-46): full image + stage dumps of sample 1
    w = h = 64
    n = w * h
    for mode in (0, 1):
        o = make_oracle(O, inputs_for(O, "simple", w, h), w, h, rng_mode=mode, max_wavefronts=4)
        o.set_frame(1, 0)
        o.reset_image()
        o.set_counters([0, 0, n])
        o.generate_rays(w // 8, h // 8, False)
        rays = o.rays(n)
        o.extend(*O.workgroup_size_64(n))
        c = o.counters()
        hits, misses = o.hits(int(c[1])), o.misses(int(c[0]))
        c[2] = 0
        o.set_counters(c)
        o.shade(*O.workgroup_size_64(int(c[1])))
        ext = o.extension_rays(int(c[1]))
        o.miss(*O.workgroup_size_64(int(c[0])))
        image = o.image()
        o.close()
        case, acc = render_case("simple", w, h, 4, 4, mode, 4)
        np.savez_compressed(os.path.join(HERE, f"simple_64x64_mode{mode}.npz"), rays=rays, hits=hits, misses=misses,
                            ext=ext, image_after_first_wavefront=image, acc=acc, **case)
    simple_wide()
    # (2) the seeded Shirley scene itself
    sp, mt = O.scene_book_one_final(1)
    sp2, nodes = O.build_bvh(sp)
    np.savez_compressed(os.path.join(HERE, "shirley_seed1_scene.npz"), spheres_generated=sp, spheres_bvh_order=sp2,
                        materials=mt, nodes=nodes)
    # (3) Shirley renders: BASELINE config 1 (400x225) + its strict twin (400x224), and full HD
    for (w, h, spp, bounces, ds) in ((400, 224, 4, 4, 8), (400, 225, 4, 4, 8), (1920, 1080, 2, 8, 16)):
        for mode in (0, 1):
            case, _ = render_case("shirley", w, h, spp, bounces, mode, ds)
            np.savez_compressed(os.path.join(HERE, f"shirley_{w}x{h}_mode{mode}.npz"), **case)
            print(w, h, mode, case["acc_sha256"][:16], case["totals"])


def simple_wide():
    # (1b) the same 5-sphere scene at 128x72 (SURVEY 8c fixture list): image + per-bounce tables
    for mode in (0, 1):
        case, acc = render_case("simple", 128, 72, 4, 4, mode, 4)
        np.savez_compressed(os.path.join(HERE, f"simple_128x72_mode{mode}.npz"), acc=acc, **case)


def mesh_golden():
    # (4) build extension: triangle soup, 5000 triangles with 8x edges, both RNG modes
    w, h, spp, bounces = 200, 120, 2, 6
    for mode in (0, 1):
        o = make_mesh_oracle(O, mesh_inputs(O, w, h, 5000, 8.0), w, h, max_wavefronts=bounces, rng_mode=mode)
        acc = o.render(spp)
        np.savez_compressed(os.path.join(HERE, f"mesh5000_200x120_mode{mode}.npz"), width=w, height=h, spp=spp, bounces=bounces,
                            table=o.bounce_table(), totals=o.totals(), acc_sha256=sha(acc), acc=acc)
        o.close()


def config5_golden():
    # (5) BASELINE config 5 at its stated size: 1 000 000-triangle soup (seed 1, 32-bin SAH), 1920x1080, 8 bounces, 1 spp,
    # both RNG modes: per-bounce table + image hash + 16x down-sampled image (the full image would be 25 MB)
    w, h, spp, bounces, ds, n_tri = 1920, 1080, 1, 8, 16, 1000000
    inputs = mesh_inputs(O, w, h, n_tri)
    for mode in (0, 1):
        o = make_mesh_oracle(O, inputs, w, h, max_wavefronts=bounces, rng_mode=mode)
        o.render_sample()
        acc, table = o.accumulated(), o.bounce_table()
        np.savez_compressed(os.path.join(HERE, f"mesh1m_1920x1080_mode{mode}.npz"), width=w, height=h, spp=spp, bounces=bounces,
                            n_triangles=n_tri, n_bins=32, rng_mode=mode, table=table, totals=o.totals(), acc_sha256=sha(acc),
                            acc_small=downsample(acc, w, h, ds), downsample=ds, n_nodes=len(inputs[2]))
        print("config 5", mode, sha(acc)[:16], o.totals(), len(inputs[2]))
        o.close()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "config5":
        config5_golden()
    else:
        mesh_golden()
        main()
        config5_golden()
