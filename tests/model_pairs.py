"""Schedule model of "two rays per lane" in the fused bounce launch (round 5; build container, CPU only; a design tool like
tests/model_schedule.py, not part of the product).

Today a wave of bounce_kernel<middle> walks 64 rays (hits h .. h + 63 of one 512-hit work item) in a while-while loop: a round of inner
visits lasts as long as its longest lane, and a lane whose ray is done idles until the wave's last ray is. Question: if a workgroup took TWO
consecutive work items per pass -- every lane shades hit h and hit h + 512, walks the first ray and, when that walk ends, goes on with the
second in the SAME loop (nothing crosses lanes, the queue order is untouched) -- how many wave-level rounds are left? K = 1 is today's
schedule; K = 2, 3, 4 rays per lane.

Per ray the oracle reports the inner visits between consecutive leaf visits (orc_ray_rounds); a wave's pass costs
sum over rounds of max-over-lanes(inner visits) * C_VISIT + (some lane at a leaf) * C_LEAF, in vector instructions of the shipped kernel
(35 per inner visit of the hand-written loop, ~100 per leaf round).

A second question, same data: an "if-if" loop (every iteration ONE inner visit for the lanes that sit on an inner node; the leaf code runs
when T lanes wait at a leaf or no lane can descend; T = 64 is today's while-while loop) -- does serving leaves earlier shorten the wave's walk?

Results (960x544, wavefronts 1-5; profiles/r05_rejected_experiments.txt): chaining COSTS 10-13 % (rays that start at different times no longer
descend together: the longest lane of every round stays as long, and there are more rounds), the if-if loop gains at most 3-4 % at T = 16-24
before its own bookkeeping. The while-while loop is within a few percent of the best any schedule can do for a FIXED assignment of rays to
lanes; what is left is the assignment (the class-binned queue of the pixel-keyed mode).

Usage: python tests/model_pairs.py [width height [bounces]]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402
from tests.helpers import inputs_for, make_oracle  # noqa: E402
from tests.model_schedule import pad, rounds_of  # noqa: E402

C_VISIT, C_LEAF = 35.0, 100.0


def chained_cost(segs, nl, K):
    """segs (n, 16), nl (n,): rays in queue order; K consecutive 512-ray items per pass. Returns (vector instructions, inner rounds, leaf rounds)."""
    item = 512 * K
    s, l = pad(segs, item), pad(nl, item)
    n_pass = s.shape[0] // item
    s = s.reshape(n_pass, K, 8, 64, 16)   # pass, ray of the lane, wave, lane, round
    l = l.reshape(n_pass, K, 8, 64)
    R = 17 * K                             # a ray has at most 16 leaf visits: 17 segments with the trailing one
    seq = np.zeros((n_pass, 8, 64, R), np.int32)
    leaf = np.zeros((n_pass, 8, 64, R), bool)
    pos = np.zeros((n_pass, 8, 64), np.int64)
    k16 = np.arange(16)
    for k in range(K):
        sk, lk = s[:, k], np.minimum(l[:, k], 15)
        live = sk.sum(axis=-1) + lk > 0 # (padding rays do nothing)
        for r in range(17):             # segment r of this ray goes to slot pos + r while r <= nl
            use = (r <= lk) & live
            idx = pos + r
            val = sk[..., r] if r < 16 else np.zeros_like(pos)
            np.put_along_axis(seq, idx[..., None], np.where(use, val, np.take_along_axis(seq, idx[..., None], -1)[..., 0])[..., None], -1)
            lf = use & (r < lk)
            np.put_along_axis(leaf, idx[..., None], (lf | np.take_along_axis(leaf, idx[..., None], -1)[..., 0])[..., None], -1)
        pos = pos + np.where(live, lk + 1, 0)
    inner = seq.max(axis=2)              # (pass, wave, R): the longest lane of every round
    leaf_r = leaf.any(axis=2)
    return inner.sum() * C_VISIT + leaf_r.sum() * C_LEAF, inner.sum(), leaf_r.sum()


def ifif_cost(segs, nl, T, n_waves=400, seed=1):
    """An if-if loop with leaf threshold T on a sample of waves (64 consecutive rays each): (inner-visit iterations, leaf rounds, rays)."""
    s, l = pad(segs, 64).reshape(-1, 64, 16), np.minimum(pad(nl, 64).reshape(-1, 64), 15)
    pick = np.random.default_rng(seed).choice(s.shape[0], min(n_waves, s.shape[0]), replace=False)
    tot_v = tot_l = rays = 0
    for wv in pick:
        S, L = s[wv], l[wv]
        live = (S.sum(1) + L) > 0
        rays += int(live.sum())
        seg, rem, done = np.zeros(64, int), S[:, 0].copy(), ~live

        def settle():  # a lane with no inner visit left in segment k: at a leaf if k < L, else its walk is over
            at = (~done) & (rem == 0)
            fin = at & (seg >= L)
            done[fin] = True
            return at & ~fin
        waiting = settle()
        while not done.all():
            desc = (~done) & (rem > 0)
            if desc.any():
                tot_v += 1
                rem[desc] -= 1
            waiting = settle()
            if waiting.sum() >= T or (waiting.any() and not ((~done) & (rem > 0)).any()):
                tot_l += 1
                seg[waiting] += 1
                rem[waiting] = S[waiting, np.minimum(seg[waiting], 15)]
                waiting = settle()
    return tot_v, tot_l, rays


def main():
    w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (960, 544)
    bounces = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    O.build()
    o = make_oracle(O, inputs_for(O, "shirley", w, h), w, h, max_wavefronts=8)
    n = w * h
    o.set_frame(1, 0)
    o.reset_image()
    o.set_counters([0, 0, n])
    o.generate_rays(w // 8, h // 8, False)
    ext = O.workgroup_size_64(n)
    n_rays = n
    tot = {K: 0.0 for K in (1, 2, 3, 4)}
    for b in range(bounces):
        segs, nl = rounds_of(o, n_rays)
        if b > 0:  # wavefront 0 is the first launch (generate_rays + extend): coherent primary rays, not a middle launch
            line = [f"wavefront {b}: {n_rays} rays, {segs.sum() / n_rays:.1f} inner visits and {nl.mean():.2f} leaf visits per ray"]
            base = None
            for K in (1, 2, 3, 4):
                c, ir, lr = chained_cost(segs, nl, K)
                base = base or c
                tot[K] += c
                line.append(f"K={K}: {ir * 64 / n_rays:.1f} inner + {lr * 64 / n_rays:.1f} leaf rounds per 64 rays, lanes {segs.sum() / ir:.1f}, walk {c / base:.3f}")
            print("; ".join(line), flush=True)
            if b <= 3:
                out, ref = [], None
                for T in (64, 32, 24, 16, 8, 1):
                    v, lf, r = ifif_cost(segs, nl, T)
                    c = v * C_VISIT + lf * C_LEAF
                    ref = ref or c
                    out.append(f"T={T}: {v * 64 / r:.1f} + {lf * 64 / r:.1f} rounds, walk {c / ref:.3f}")
                print("    if-if loop, leaf threshold T (sample of 400 waves): " + "; ".join(out), flush=True)
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
    print("all middle wavefronts, walk instructions relative to K = 1: " + ", ".join(f"K={K}: {tot[K] / tot[1]:.3f}" for K in tot))


if __name__ == "__main__":
    main()
