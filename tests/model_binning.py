"""Schedule model of GLOBAL binning of the hit queue, evaluated offline on the oracle's queues
(build container, CPU only; a design tool like tests/model_schedule.py, not part of the product).

Question: if the hits of wavefront b are written to K queues by a class key, so that a work item of the next launch is 512 hits
of ONE class, how many wave-level traversal rounds (and how much shade divergence) does the next launch save against the hit
order it has today?  tests/model_schedule.py answered it for sorts INSIDE a 512 / 1024-ray item; this file scores keys over the
WHOLE wavefront, and only keys that exist when the hit is written (the hit's primitive, its material, the hit point and normal,
the incoming direction) unless marked "post-shade" (the extension ray's own direction: known at write time only if shade runs
before the write, which WFPT_RNG_PIXEL allows and WFPT_RNG_DISPATCH does not).

Per key and bounce: traversal wave cost relative to hit order (C_VISIT / C_LEAF as model_schedule), wave-level rounds relative
to hit order, E[max over an item's 8 waves] / mean (what the item's barrier waits for), shade cost relative to hit order
(a wave runs every material branch one of its lanes needs), and the total (traversal + shade) relative to hit order.

Usage: python tests/model_binning.py [width height [bounces]]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402
from tests.helpers import inputs_for, make_oracle  # noqa: E402
from tests.model_schedule import C_LEAF, C_VISIT, lane_work, pad, rounds_of, wave_cost  # noqa: E402

# shade, wave instructions (from the kernel's ISA, DESIGN section 4: ~400 per hit with every branch taken):
# common part (record + RNG key + normal + 1/d + stores), the unit-sphere sampler of lambertian and metal, metal's reflect,
# the dielectric branch (normalize, schlick's pow, refract)
S_COMMON, S_SAMPLER, S_METAL, S_DIEL = 140.0, 150.0, 15.0, 120.0


def shade_cost(mat):
    """mat (waves, 64) material type per lane (-1 = no hit in this lane): wave instructions per wave."""
    live = mat >= 0
    any_l = live.any(axis=1)
    smp = ((mat == 0) | (mat == 1)).any(axis=1)
    met = (mat == 1).any(axis=1)
    die = (mat == 2).any(axis=1)
    return any_l * S_COMMON + smp * S_SAMPLER + met * S_METAL + die * S_DIEL


def score(segs, nl, mat, order=None):
    if order is not None:
        segs, nl, mat = segs[order], nl[order], mat[order]
    s, l = pad(segs, 512).reshape(-1, 64, 16), pad(nl, 512).reshape(-1, 64)
    m = pad(mat + 1, 512).reshape(-1, 64) - 1  # padding lanes: -1
    wc = wave_cost(s, l)
    k = np.arange(16)
    rounds = s.max(axis=1).sum(axis=1) + (k[None, None, :] < l[:, :, None]).any(axis=1).sum(axis=1)
    items = wc.reshape(-1, 8)
    live_items = items.sum(axis=1) > 0
    mom = (items[live_items].max(axis=1) / np.maximum(items[live_items].mean(axis=1), 1e-9)).mean()
    return dict(trace=wc.sum(), rounds=rounds.sum(), mom=mom, shade=shade_cost(m).sum())


def binned_order(key):
    return np.argsort(key, kind="stable")


def quantile_bins(x, k):
    edges = np.quantile(x, np.linspace(0, 1, k + 1)[1:-1])
    return np.searchsorted(edges, x)


def main():
    w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (960, 544)
    bounces = int(sys.argv[3]) if len(sys.argv) > 3 else 5
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
    prev = None  # per ray of this wavefront: (primitive it leaves, its material, normal.y at the origin)
    for b in range(bounces):
        segs, nl = rounds_of(o, n_rays)
        rays = o.rays(n_rays)
        if prev is not None:
            prim, mat = prev
            org, d = rays["origin"][:, :3], rays["direction"][:, :3]
            c, r = spheres["center"][prim][:, :3], spheres["radius"][prim]
            ny = (org[:, 1] - c[:, 1]) / r
            is_ground = (prim == ground).astype(np.int64)
            dn = d / np.linalg.norm(d, axis=1, keepdims=True)
            dist = np.linalg.norm(org - np.array([0.0, 0.0, 0.0]), axis=1)
            base = score(segs, nl, mat)
            work = lane_work(segs, nl).sum()
            print(f"bounce {b}: {n_rays} rays, lane util in hit order {work / 64 / base['trace']:.2f}, item max/mean {base['mom']:.2f}, "
                  f"ground share {is_ground.mean():.2f}, materials {np.bincount(mat, minlength=3) / len(mat)}", flush=True)
            keys = {
                "material (3)": mat,
                "ground|other (2)": is_ground,
                "ground, lamb, metal, diel (4)": np.where(is_ground == 1, 0, mat + 1),
                "(4) x normal.y>0.5 (<=8)": np.where(is_ground == 1, 0, mat + 1) * 2 + (ny > 0.5),
                "(4) x normal.y 4 bins": np.where(is_ground == 1, 0, (mat + 1) * 4 + np.clip(((ny + 1.0) * 2).astype(np.int64), 0, 3)),
                "(4) x far from the scene centre (8)": np.where(is_ground == 1, 0, mat + 1) * 2 + (dist > 8.0),
                "ground x dist 4 bins | other x mat": np.where(is_ground == 1, quantile_bins(dist, 4), 4 + mat),
                "post-shade: (4) x dir.y sign (8)": np.where(is_ground == 1, 0, mat + 1) * 2 + (dn[:, 1] > 0),
                "post-shade: ground|other x dir.y 8 bins (16)": is_ground * 8 + np.clip(((dn[:, 1] + 1.0) * 4).astype(np.int64), 0, 7),
                "post-shade: (4) x dir.y 4 bins (16)": np.where(is_ground == 1, 0, mat + 1) * 4 + np.clip(((dn[:, 1] + 1.0) * 2).astype(np.int64), 0, 3),
                "(4), metal x exact dir.y>0 (5)": np.where(is_ground == 1, 0, np.where(mat == 1, 4 + (dn[:, 1] > 0), mat + 1)),
                "(4), metal+diel x exact dir.y 4 bins": np.where(is_ground == 1, 0, np.where(mat >= 1, 4 + (mat - 1) * 4 + np.clip(((dn[:, 1] + 1.0) * 2).astype(np.int64), 0, 3), 1)),
                "post-shade: all x dir.y 8 bins": np.clip(((dn[:, 1] + 1.0) * 4).astype(np.int64), 0, 7),
                "post-shade: ground|other x dir.y 4 bins (8)": is_ground * 4 + np.clip(((dn[:, 1] + 1.0) * 2).astype(np.int64), 0, 3),
                "post-shade: ground x dir.y 6 quantiles | other (7)": np.where(is_ground == 1, quantile_bins(dn[:, 1], 6), 6),
                "post-shade: ground x dir.y 4q | lamb x 2 | metal x 2 | diel (9)": np.where(is_ground == 1, quantile_bins(dn[:, 1], 4), np.where(mat == 2, 8, 4 + 2 * np.minimum(mat, 1) + (dn[:, 1] > 0.3))),
                "upper bound: true cost, 16 bins": quantile_bins(lane_work(segs, nl), 16),
            }
            tot0 = base["trace"] + base["shade"]
            for name, key in keys.items():
                s = score(segs, nl, mat, binned_order(np.asarray(key)))
                print(f"    {name:48s} trace x{s['trace'] / base['trace']:.3f}  rounds x{s['rounds'] / base['rounds']:.3f}  "
                      f"item max/mean {s['mom']:.2f}  shade x{s['shade'] / base['shade']:.3f}  total x{(s['trace'] + s['shade']) / tot0:.3f}",
                      flush=True)
        ext = O.workgroup_size_64(n_rays)
        o.extend(*ext)
        c = o.counters()
        misses, hits = int(c[0]), int(c[1])
        hq = o.hits(hits)
        prev = (hq["sphere_idx"].astype(np.int64), hq["mat_type"].astype(np.int64))
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
