"""Schedule model (design tool, CPU only, like tests/model_schedule.py): would the LDS-resident walk do better with the refill traversal's
"if-if" schedule -- every iteration each lane does ONE step, an inner visit or (when at least T lanes wait at a leaf, or no lane has a
node to visit) a leaf -- than with its "while-while" rounds, in which every lane waits at its leaf until the slowest descent is over?
Per wave of 64 consecutive rays of the oracle's queues: cost = inner iterations x C_VISIT + leaf rounds x C_LEAF.

Result (640x360, 400 waves per bounce, relative to while-while at bounces 0 / 1 / 2 / 3):
  tests/model_schedule.py's constants (C_VISIT 69, C_LEAF 90)          T = 8:  1.04 / 0.95 / 0.89 / 0.86
  the kernel's own costs (a visit ~127 pipe cycles, a leaf round ~350:   T = 8:  1.09 / 1.02 / 1.00 / 0.99
    sphere test with sqrt and two IEEE divisions, near-tie watch, pop)   T = 24: 1.05 / 1.00 / 0.96 / 0.95
The sphere scene's leaves are 2.75 x a visit, so gathering lanes for them is what pays, and while-while gathers them all: not built.

Usage: python tests/model_leaf_threshold.py [c_visit c_leaf]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O  # noqa: E402
from tests.helpers import inputs_for, make_oracle  # noqa: E402
import tests.model_schedule as MS  # noqa: E402
from tests.model_schedule import pad, rounds_of  # noqa: E402

if __name__ == "__main__" and len(sys.argv) > 2:
    MS.C_VISIT, MS.C_LEAF = float(sys.argv[1]), float(sys.argv[2])
C_VISIT, C_LEAF, wave_cost = MS.C_VISIT, MS.C_LEAF, MS.wave_cost

def sim_ifif(segs, nl, T):
    """one wave: segs (64,16) visits before each leaf, nl (64,) leaf count; after the last leaf a ray may have trailing visits segs[nl]."""
    n = segs.shape[0]
    seg_i = np.zeros(n, int)            # current segment index
    rem = segs[np.arange(n), 0].copy()  # visits left in the current segment
    done = np.zeros(n, bool)
    # a ray with nl leaves has segments 0..nl (the last one = visits after the last leaf, then it ends)
    cost = 0.0
    it_v = it_l = 0
    while True:
        at_leaf = (~done) & (rem == 0) & (seg_i < nl)
        fin = (~done) & (rem == 0) & (seg_i >= nl)
        done |= fin
        inner = (~done) & (rem > 0)
        if not inner.any() and not at_leaf.any():
            break
        if inner.any():
            rem[inner] -= 1
            cost += C_VISIT; it_v += 1
        at_leaf = (~done) & (rem == 0) & (seg_i < nl)
        inner2 = (~done) & (rem > 0)
        if at_leaf.any() and (at_leaf.sum() >= T or not inner2.any()):
            cost += C_LEAF; it_l += 1
            seg_i[at_leaf] += 1
            idx = np.where(at_leaf)[0]
            rem[idx] = segs[idx, np.minimum(seg_i[idx], 15)]
    return cost, it_v, it_l

def main():
    w, h = 640, 360
    O.build()
    o = make_oracle(O, inputs_for(O, "shirley", w, h), w, h, max_wavefronts=8)
    n = w * h
    o.set_frame(1, 0); o.reset_image(); o.set_counters([0, 0, n]); o.generate_rays(w // 8, h // 8, False)
    n_rays = n
    rng = np.random.default_rng(1)
    for b in range(4):
        segs, nl = rounds_of(o, n_rays)
        s, l = pad(segs, 64).reshape(-1, 64, 16), pad(nl, 64).reshape(-1, 64)
        pick = rng.choice(len(s), min(400, len(s)), replace=False)
        base = wave_cost(s[pick], l[pick]).sum()
        out = [f"bounce {b}: while-while {base / len(pick):.0f}"]
        for T in (1, 4, 8, 16, 24, 32):
            tot = 0.0
            for i in pick:
                c, _, _ = sim_ifif(s[i], l[i], T)
                tot += c
            out.append(f"T={T}: x{tot / base:.3f}")
        print("; ".join(out), flush=True)
        ext = O.workgroup_size_64(n_rays)
        o.extend(*ext)
        c = o.counters(); misses, hits = int(c[0]), int(c[1]); c[2] = 0; o.set_counters(c)
        o.shade(*O.workgroup_size_64(hits)); o.miss(*O.workgroup_size_64(misses))
        n_rays = int(o.counters()[2]); o.swap_ray_queues(); o.set_counters([0, 0, n_rays, 0])

if __name__ == "__main__":
    main()
