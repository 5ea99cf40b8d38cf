"""Gate of the direction key of the class-binned loop (round 5; an offline design tool like tests/model_binning.py, CPU only).

tests/model_binning.py scores the EXACT direction of the extension ray as a binning key ("post-shade" keys). The kernel has to produce the
key when the hit is WRITTEN, before shade has run for it. In WFPT_RNG_PIXEL shade's RNG is keyed by (pixel, frame) alone, so the unit
vector `rb` every scatter of a path draws is the same at every bounce: the launch that shades hit b-1 holds the rb that hit b will use.
This file scores what the kernel can compute from (hit point, primitive, incoming direction, rb) with approximate arithmetic:
    lambertian: ext = n + rb            (|ext|^2 = 2 + 2 n.rb)        metal: the mirror direction, fuzz ignored        dielectric: a class of its own
against the exact key, with rb quantised to 8 bits per component, and for a first launch that has no rb (primitive classes only).

Usage: python tests/model_dirkey.py [width height [bounces]]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402
from tests.helpers import inputs_for, make_oracle  # noqa: E402
from tests.model_binning import binned_order, score  # noqa: E402
from tests.model_schedule import rounds_of  # noqa: E402


def bins(y, k):
    return np.clip(((y + 1.0) * (k / 2.0)).astype(np.int64), 0, k - 1)


def main():
    w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (960, 544)
    bounces = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    O.build()
    inputs = inputs_for(O, "shirley", w, h)
    spheres = inputs[0]
    ground = int(np.argmax(spheres["radius"]))
    o = make_oracle(O, inputs, w, h, max_wavefronts=8)
    n = w * h
    o.set_frame(1, 0)
    o.reset_image()
    o.set_counters([0, 0, n])
    o.generate_rays(w // 8, h // 8, False)
    n_rays = n
    prev = None
    for b in range(bounces):
        segs, nl = rounds_of(o, n_rays)
        rays = o.rays(n_rays).copy()
        if prev is not None:
            prim, mat, din = prev
            org, d = rays["origin"][:, :3].astype(np.float64), rays["direction"][:, :3].astype(np.float64)
            c, r = spheres["center"][prim][:, :3].astype(np.float64), spheres["radius"][prim].astype(np.float64)
            nrm = (org - c) / r[:, None]
            is_ground = (prim == ground).astype(np.int64)
            dn = d / np.linalg.norm(d, axis=1, keepdims=True)
            din_n = din / np.linalg.norm(din, axis=1, keepdims=True)
            # what the writing launch can know: rb of the path (lambertian: ext - n exactly; elsewhere drawn here as a stand-in with the same law)
            rng = np.random.default_rng(5)
            rb = d - nrm
            other = mat != 0
            v = rng.normal(size=(int(other.sum()), 3))
            rb[other] = v / np.linalg.norm(v, axis=1, keepdims=True)
            rbq = np.round(rb * 127.0) / 127.0
            mirror = din_n - 2.0 * np.sum(din_n * nrm, axis=1, keepdims=True) * nrm

            def approx_y(rbv, only_y=False):
                ndot = nrm[:, 1] * rbv[:, 1] if only_y else np.sum(nrm * rbv, axis=1)
                y_l = (nrm[:, 1] + rbv[:, 1]) / np.sqrt(np.maximum(2.0 + 2.0 * ndot, 1e-6))
                return np.where(mat == 0, y_l, np.where(mat == 1, mirror[:, 1], din_n[:, 1]))
            base = score(segs, nl, mat)
            print(f"bounce {b}: {n_rays} rays", flush=True)
            keys = {
                "built in round 4: ground, lamb, metal, diel (4)": np.where(is_ground == 1, 0, mat + 1),
                "exact: ground|other x dir.y 8 bins (16)": is_ground * 8 + bins(dn[:, 1], 8),
                "exact: ground|other x dir.y 4 bins (8)": is_ground * 4 + bins(dn[:, 1], 4),
                "approx (n + rb | mirror | straight): x 8 bins (16)": is_ground * 8 + bins(approx_y(rb), 8),
                "approx, rb in 8 bits per component: x 8 bins (16)": is_ground * 8 + bins(approx_y(rbq), 8),
                "approx, rb in 8 bits: x 4 bins (8)": is_ground * 4 + bins(approx_y(rbq), 4),
                "approx, only rb.y carried: x 8 bins (16)": is_ground * 8 + bins(approx_y(rb, True), 8),
                "approx, only rb.y carried: x 4 bins (8)": is_ground * 4 + bins(approx_y(rb, True), 4),
                "approx, diel in a class of its own: ground x 6 | other x 6 | diel... (13)": np.where(mat == 2, 12, is_ground * 6 + bins(approx_y(rbq), 6)),
                "approx x 8 bins, ground|lamb+metal|diel (17)": np.where(mat == 2, 16, is_ground * 8 + bins(approx_y(rbq), 8)),
            }
            tot0 = base["trace"] + base["shade"]
            for name, key in keys.items():
                s = score(segs, nl, mat, binned_order(np.asarray(key)))
                print(f"    {name:72s} trace x{s['trace'] / base['trace']:.3f}  item max/mean {s['mom']:.2f}  shade x{s['shade'] / base['shade']:.3f}  "
                      f"total x{(s['trace'] + s['shade']) / tot0:.3f}", flush=True)
        ext = O.workgroup_size_64(n_rays)
        o.extend(*ext)
        c = o.counters()
        misses, hits = int(c[0]), int(c[1])
        hq = o.hits(hits)
        prev = (hq["sphere_idx"].astype(np.int64), hq["mat_type"].astype(np.int64), rays["direction"][hq["ray_idx"]][:, :3].astype(np.float64))
        c[2] = 0
        o.set_counters(c)
        o.shade(*O.workgroup_size_64(hits))
        o.miss(*O.workgroup_size_64(misses))
        n_rays = int(o.counters()[2])
        o.swap_ray_queues()
        o.set_counters([0, 0, n_rays, 0])
    o.close()


if __name__ == "__main__":
    main()
