"""Wave64 schedule models of the LDS-resident traversal, evaluated offline on the oracle's ray queues (build container,
CPU only; a design tool, not part of the product). For every wavefront of one sample of the Shirley scene the oracle
reports, per ray, the inner visits between consecutive leaf visits (orc_ray_rounds); the models below price a
while-while schedule in wave instructions:

  base    : 64 consecutive rays per wave, a round costs max-over-lanes(inner visits) * C_VISIT + C_LEAF
  sort K  : the rays of one 512-ray item are sorted by a key before they are dealt to the 8 waves
  refill  : an item is P rays; a wave whose idle lanes reach T takes new rays from the item's pool between rounds

Usage: python tests/model_schedule.py [width height]
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402
from tests.helpers import inputs_for, make_oracle  # noqa: E402

C_VISIT, C_LEAF, C_SETUP, C_REFILL = 69.0, 90.0, 300.0, 45.0


def rounds_of(o, n):
    segs = np.zeros((n, 16), np.uint8)
    nl = np.zeros(n, np.uint8)
    o.L.orc_ray_rounds(o.h, n, segs.ctypes.data_as(C.c_void_p), nl.ctypes.data_as(C.c_void_p))
    return segs.astype(np.int32), nl.astype(np.int32)


def wave_cost(segs, nl):
    """segs (waves, 64, 16), nl (waves, 64): while-while cost per wave."""
    k = np.arange(16)
    has_leaf = (k[None, None, :] < nl[:, :, None])
    inner = segs.max(axis=1)                    # (waves, 16)
    leaf = has_leaf.any(axis=1)                 # (waves, 16)
    return inner.sum(axis=1) * C_VISIT + leaf.sum(axis=1) * C_LEAF


def lane_work(segs, nl):
    return segs.sum(axis=1) * C_VISIT + nl * C_LEAF


def pad(a, m):
    n = a.shape[0]
    p = (-n) % m
    if p:
        a = np.concatenate([a, np.zeros((p,) + a.shape[1:], a.dtype)])
    return a


def model_base(segs, nl):
    s, l = pad(segs, 64).reshape(-1, 64, 16), pad(nl, 64).reshape(-1, 64)
    return wave_cost(s, l).sum()


def model_sort(segs, nl, key, item=512):
    s, l, k = pad(segs, item), pad(nl, item), pad(key, item)
    k = k.reshape(-1, item)
    order = np.argsort(k, axis=1, kind="stable") + (np.arange(k.shape[0]) * item)[:, None]
    order = order.reshape(-1)
    return model_base(s[order], l[order])


def model_refill(segs, nl, P, T, max_items=400, rng=None):
    """Event model of one workgroup (8 waves) per item of P rays; returns (sum of wave instructions, rays)."""
    n = segs.shape[0]
    items = np.arange(0, n - P + 1, P)
    if len(items) > max_items:
        items = rng.choice(items, max_items, replace=False)
    total, rays = 0.0, 0
    for i0 in items:
        S, L = segs[i0:i0 + P], nl[i0:i0 + P]
        cursor = 0
        rays += P
        # each wave: arrays of ray id per lane (-1 idle), round index per lane
        waves = []
        for w in range(8):
            take = min(64, P - cursor)
            ids = np.full(64, -1)
            ids[:take] = np.arange(cursor, cursor + take)
            cursor += take
            waves.append([ids, np.zeros(64, np.int32), 0.0])
        # waves advance in time order (the pool is shared): simple discrete-event loop
        done = [False] * 8
        while not all(done):
            w = min((i for i in range(8) if not done[i]), key=lambda i: waves[i][2])
            ids, rnd, t = waves[w]
            act = ids >= 0
            if not act.any():
                done[w] = True
                continue
            a = np.where(act)[0]
            seg = S[ids[a], np.minimum(rnd[a], 15)]
            leaf = rnd[a] < L[ids[a]]
            t += seg.max() * C_VISIT + (C_LEAF if leaf.any() else 0.0)
            rnd[a] += 1
            # a ray ends after the visits that follow its last leaf
            fin = rnd[a] > L[ids[a]]
            ids[a[fin]] = -1
            idle = int((ids < 0).sum())
            if cursor < P and idle >= T:
                take = min(idle, P - cursor)
                free = np.where(ids < 0)[0][:take]
                ids[free] = np.arange(cursor, cursor + take)
                rnd[free] = 0
                cursor += take
                t += C_REFILL
            waves[w][2] = t
        total += sum(wv[2] for wv in waves)
    return total, rays


def main():
    w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (960, 544)
    O.build()
    o = make_oracle(O, inputs_for(O, "shirley", w, h), w, h, max_wavefronts=8)
    n = w * h
    o.set_frame(1, 0)
    o.reset_image()
    o.set_counters([0, 0, n])
    o.generate_rays(w // 8, h // 8, False)
    ext = O.workgroup_size_64(n)
    n_rays = n
    rng = np.random.default_rng(1)
    for b in range(6):
        segs, nl = rounds_of(o, n_rays)
        rays = o.rays(n_rays)
        work = lane_work(segs, nl).sum()
        base = model_base(segs, nl)
        waves = (n_rays + 63) // 64
        line = [f"bounce {b}: {n_rays} rays, lane work {work / n_rays:.0f}/ray, base wave cost {base / waves:.0f} (util {work / 64 / base:.2f})"]
        dy = np.array(rays["direction"][:, 1], np.float32) if "direction" in rays.dtype.names else None
        if dy is not None:
            c = model_sort(segs, nl, dy)
            line.append(f"sort dy {c / waves:.0f} ({work / 64 / c:.2f})")
            c = model_sort(segs, nl, np.abs(dy))
            line.append(f"sort |dy| {c / waves:.0f} ({work / 64 / c:.2f})")
        c = model_sort(segs, nl, lane_work(segs, nl))
        line.append(f"sort true cost {c / waves:.0f} ({work / 64 / c:.2f})")
        c = model_sort(segs, nl, nl.astype(np.float32))
        line.append(f"sort n_leaves {c / waves:.0f} ({work / 64 / c:.2f})")
        print("; ".join(line), flush=True)
        for P in (512, 1024, 2048, 4096):
            out = []
            for T in (8, 16, 24, 32, 48):
                tot, r = model_refill(segs, nl, P, T, max_items=60, rng=rng)
                out.append(f"T={T}: {tot / (r / 64):.0f} ({work / n_rays * r / 64 / tot:.2f})")
            print(f"    refill P={P}: " + "; ".join(out), flush=True)
        o.extend(*ext)
        c = o.counters()
        misses, hits = int(c[0]), int(c[1])
        c[2] = 0
        o.set_counters(c)
        o.shade(*O.workgroup_size_64(hits))
        o.miss(*O.workgroup_size_64(misses))
        n_rays = int(o.counters()[2])
        o.swap_ray_queues()
        ext = O.workgroup_size_64(n_rays)
        o.set_counters([0, 0, n_rays, 0])


if __name__ == "__main__":
    main()
