"""Times the device BVH builder (csrc/wfpt_bvh_build.hip) beside the host builder on the same input.

    python tools/bench_bvh.py [--triangles N | --spheres] [--bins B] [--repeat K] [--no-host]

Prints one JSON line. `device_ms` is the span between the first and the last build kernel (wfpt.h), `wall_ms`
also contains the host<->device copies of primitives and nodes; `host_s` is wfpt_build_bvh(_triangles) on one core.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wavefront_path_tracer_amd as W  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--triangles", type=int, default=1000000)
    ap.add_argument("--spheres", action="store_true", help="the seeded Shirley scene with the reference's 4096 bins")
    ap.add_argument("--bins", type=int, default=32)
    ap.add_argument("--repeat", type=int, default=5)
    ap.add_argument("--no-host", action="store_true")
    args = ap.parse_args()

    if args.spheres:
        prims = W.Scene.book_one_final(1).spheres
        build = lambda p, device: W.BVHTree(len(p)).build_bvh_tree(p, device=device)  # noqa: E731
        what, bins = f"book_one_final seed 1 ({len(prims)} spheres)", 4096
    else:
        prims = W.Scene.random_mesh(args.triangles, seed=1).triangles
        build = lambda p, device: W.BVHTree(len(p)).build_bvh_tree_triangles(p, args.bins, device=device)  # noqa: E731
        what, bins = f"random mesh seed 1 ({len(prims)} triangles)", args.bins

    def run(device):
        tree = W.BVHTree(len(prims))
        p = prims.copy()
        t0 = time.perf_counter()
        if args.spheres:
            tree.build_bvh_tree(p, device=device)
        else:
            tree.build_bvh_tree_triangles(p, args.bins, device=device)
        return tree, p, time.perf_counter() - t0

    run(0)  # warm-up: module load, first allocations
    dev_ms, wall_ms = [], []
    for _ in range(args.repeat):
        tree, p_dev, wall = run(0)
        dev_ms.append(tree.device_ms)
        wall_ms.append(wall * 1e3)
    out = {"workload": what, "n_bins": bins, "nodes": int(len(tree.nodes)), "repeat": args.repeat,
           "device_ms": round(min(dev_ms), 3), "device_ms_mean": round(sum(dev_ms) / len(dev_ms), 3),
           "wall_ms": round(min(wall_ms), 3)}
    if not args.no_host:
        host_tree, p_host, host_s = run(None)
        out["host_s"] = round(host_s, 3)
        out["identical"] = bool(host_tree.nodes.tobytes() == tree.nodes.tobytes() and p_host.tobytes() == p_dev.tobytes())
        out["speedup_device_vs_host"] = round(host_s * 1e3 / min(dev_ms), 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
