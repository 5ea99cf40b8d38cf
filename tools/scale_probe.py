"""What one rank of an N-GPU run of bench.py does, measured on ONE GPU: a context holding the bands of rank `r` of N
(pixel-keyed RNG, 64 samples in flight), timed over K frames of 64 samples per pixel exactly as bench.py's step is (reset,
render); prints ms per frame, the per-rank ceiling of the strong-scaling curve (before the gather and rank imbalance: the
slowest rank bounds the job, so every rank of N is measured when --all-ranks is given) and the per-stage times.

    python tools/scale_probe.py [--all-ranks] [--frames K]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import wavefront_path_tracer_amd as W

ap = argparse.ArgumentParser()
ap.add_argument("--all-ranks", action="store_true")
ap.add_argument("--frames", type=int, default=8)
ap.add_argument("--spp", type=int, default=64)
ap.add_argument("--scene", choices=["shirley", "mesh"], default="shirley")
ap.add_argument("--batch", type=int, default=0, help="samples in flight per launch (0 = --spp, at most 128): BASELINE config 3 is --spp 256 --batch 128")
ap.add_argument("--flags", type=int, default=0, help="WFPT_FLAG_* bits for every context (e.g. 128 = no class binning)")
args = ap.parse_args()
w, h = 1920, 1080
batch = args.batch or min(args.spp, 128)


def header():
    """Provenance of what is printed: command, commit the library was built from, device, date (profiles/ files carry it)."""
    import datetime
    from wavefront_path_tracer_amd import _build
    info = _build.build_info()
    try:
        import torch
        dev = torch.cuda.get_device_name(0)
    except Exception:
        dev = "?"
    print(f"# command: python tools/scale_probe.py {' '.join(sys.argv[1:])}   commit {info.get('git_head')} (dirty at build: {info.get('git_dirty')}, "
          f"sources match build: {info.get('sources_match_build')})   device: {dev}   {datetime.datetime.utcnow().strftime('%Y-%m-%dT%H:%M:%SZ')}", flush=True)


header()


def run(world, rank):
    make = (lambda *a, **k: W.mesh_path_tracer(a[0], a[1], 1000000, **k)) if args.scene == "mesh" else W.shirley_path_tracer
    pt = make(w, h, max_wavefronts=8, rng_mode=W.RNG_PIXEL, tile_rank=rank, tile_world=world, batch=batch, flags=args.flags)
    for _ in range(2):
        pt.reset_progress()
        pt.render(args.spp)
    pt.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.frames):
        pt.reset_progress()
        pt.render(args.spp)
    pt.synchronize()
    el = (time.perf_counter() - t0) / args.frames * 1e3
    ms = np.zeros(W.STAGE_COUNT)
    n = np.zeros(W.STAGE_COUNT, np.int64)
    for _ in range(2):
        pt.reset_progress()
        m, l = pt.render_timed(args.spp)
        ms += m
        n += l
    stages = {k: round(float(ms[v]) / 2, 3) for k, v in W.STAGES.items() if n[v]}
    pt.close()
    return el, stages


base, st = run(1, 0)
print(f"N=1: {base:.3f} ms per {args.spp}-spp frame   stages (ms per frame, event-timed) {st}", flush=True)
for world in (2, 4, 8):
    ranks = range(world) if args.all_ranks else (0,)
    worst = 0.0
    for r in ranks:
        ms, st = run(world, r)
        worst = max(worst, ms)
        print(f"N={world} rank {r}: {ms:.3f} ms per frame of its slab   stages {st}", flush=True)
    print(f"N={world}: slowest measured rank {worst:.3f} ms -> per-rank ceiling {base / worst:.2f}x (efficiency {base / worst / world:.2f})", flush=True)
