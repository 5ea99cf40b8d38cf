#!/usr/bin/env python3
"""bench.py -- headline benchmark of the wavefront kernel chain on MI355X.

Workload (BASELINE.json configs[1], the configuration the metric is quoted on): seeded Shirley random-spheres scene,
1920x1080, 64 samples per pixel, 8 bounces. One STEP is one whole FRAME of that configuration:

    reset accumulation -> 64 samples per pixel (frames 1..64 of the reference's RenderProgress), each sample
    generate_rays -> 8 x (extend, scan, shade, miss_kernel) -> accumulate, all 64 in flight in one pass of the
    device-resident loop (path_tracer.rs:291-295 is the reference's spp loop) -> [N > 1: one gather of the frame]

so `--steps K` times K such frames whatever K is (the samples per frame are `--spp`, default 64, never K).
Inputs (scene, BVH, camera) are resident in HBM before the timed region.

  python bench.py --gpus 1 --steps 20 --warmup 5
  python bench.py --gpus N ...            # no launcher needed: starts its N ranks itself (torch.distributed.run, as a child)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU; the frame is sharded by 8-row pixel bands (band k -> rank k % N), no collective inside the
bounce loop, one RCCL gather of the accumulated slabs to rank 0 per frame (scaling = "strong": the frame is fixed,
per-GPU work shrinks). Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

# OpenMP's DEFAULT team (torch's and the oracle's) stays at the pool's documented CPU share, 16 per GPU; the CPU baseline sets its own
# thread counts from the cgroup quota and the affinity mask (cpu_baseline, oracle.cpu_share). Must be set before torch (which loads an
# OpenMP runtime) is imported.
os.environ.setdefault("OMP_NUM_THREADS", str(max(1, min(len(os.sched_getaffinity(0)), 16))))
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md "Chip-level parameters"); the device-derived figure is reported beside it

# algorithmic bytes per unit of work of each stage (minimal SoA chain, scene traffic excluded)
B_EXTEND_RAY, B_EXTEND_HIT, B_EXTEND_MISS = 24.0, 12.0, 4.0  # extend: ray read; hit / miss written
B_SHADE_HIT = 92.0                                           # shade: hit + ray gather + throughput RMW + extension ray
B_MISS = 36.0                                                # miss_kernel: index + (dir.y, pixel) + throughput RMW
B_GENERATE_PIXEL = 28.0 + 12.0                               # generate_rays: ray written + image reset folded in


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10, help="timed frames (one step = one frame of --spp samples per pixel)")
    ap.add_argument("--warmup", type=int, default=2, help="untimed frames before the timed ones")
    ap.add_argument("--spp", type=int, default=64, help="samples per pixel of one frame (the metric's configuration: 64)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--bounces", type=int, default=8)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--scene", choices=["shirley", "mesh"], default="shirley",
                    help="shirley: BASELINE configs 1-4 (the reference's scene); mesh: config 5's random triangle soup")
    ap.add_argument("--triangles", type=int, default=1000000)
    ap.add_argument("--rng-mode", choices=["auto", "dispatch", "pixel"], default="auto",
                    help="auto: dispatch (reference-faithful) on 1 GPU, pixel (shard-invariant) on N > 1")
    ap.add_argument("--split-shade", action="store_true", help="BASELINE config 4: per-material shade stages")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-lds-scene", action="store_true", help="experiment: traverse the scene from HBM / L2 although it fits LDS")
    ap.add_argument("--no-binning", action="store_true",
                    help="WFPT_FLAG_NO_BINNING: the hit queue stays in thread order (a work item = 512 consecutive hits); the default since round 5, kept for old command lines")
    ap.add_argument("--binning", action="store_true",
                    help="WFPT_FLAG_BINNING (pixel-keyed RNG only): the class-binned loop whatever the size of the slab")
    ap.add_argument("--no-refill", action="store_true", help="mesh scene: fused bounce kernel (lanes keep their ray) instead of dynamic lane refill")
    ap.add_argument("--binary-bvh", action="store_true", help="mesh scene: walk the binary tree instead of the four-wide collapse")
    ap.add_argument("--unfused", action="store_true", help="run the stage kernels one by one (extend, scan, shade, miss_kernel per wavefront)")
    ap.add_argument("--exact-traversal", action="store_true",
                    help="WFPT_FLAG_EXACT_TRAVERSAL: the reference's slab arithmetic and 1e30 box-miss value, operation for operation")
    ap.add_argument("--batch", type=int, default=0, help="samples kept in flight per launch (0 = all --spp samples of a frame, at most 128)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0,
                    help="CPU-baseline sample: whole samples per pixel of the same frame until this much time is spent")
    ap.add_argument("--cpu-threads", type=int, default=0,
                    help="CPU-baseline OpenMP threads (0 = min(cgroup CPU quota, affinity mask); without a quota: 16 and the whole mask, both reported)")
    ap.add_argument("--no-stage-times", action="store_true")
    ap.add_argument("--no-pixel-anchor", action="store_true",
                    help="N = 1, --rng-mode auto: skip the pixel-keyed repeat of the timed frames (`value_pixel_mode`, the anchor of scaling sweeps)")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                    help="nccl = the frame is gathered by RCCL over xGMI through the C ABI (one GPU per rank). gloo: rehearsal "
                         "only -- ranks may share one GPU, the gather goes through host memory")
    ap.add_argument("--force-rccl", action="store_true",
                    help="N = 1: run the RCCL branch anyway (unique id -> comm_init -> gather -> gathered) with a one-rank communicator")
    ap.add_argument("--dump", default=None, help="write the tone-mapped frame (PPM) here (rank 0)")
    return ap.parse_args(argv)


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD (torch.distributed.run), relay its
    output and exit code. Nothing in this process has touched the GPU or imported torch (a process that has initialised
    the GPU must never be replaced by exec on this pool; a child is always safe)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


def cpu_baseline(args, rng_mode):
    """The oracle (kind "port": the build's own CPU restatement of the reference's chain; the reference's
    cpu_wavefront_pt has no source) timed on this box's host cores, OpenMP, on a bounded sample of the SAME
    workload: same scene/seed/camera/size/bounces, the frame's first samples per pixel until --cpu-seconds are spent
    (Mrays/s is spp-invariant). BASELINE.md section 2 asks for ALL the cores the job has: the thread count is
    min(cgroup CPU quota, affinity mask) (oracle.cpu_share); where no quota is set the frame is timed twice, at 16 threads (the
    pool's documented CPU share per GPU) and at the whole affinity mask, both legs are reported and `value` is the LARGER one.
    Returns (json object, samples rendered by the first leg, their accumulated image)."""
    from oracle import oracle as O
    share = O.cpu_share()
    if args.cpu_threads:
        thread_legs = [max(1, args.cpu_threads)]
    elif share["quota_cores"] is not None:
        thread_legs = [share["granted"]]
    else:
        thread_legs = sorted({min(16, share["affinity"]), share["affinity"]})

    def make(**kw):
        kw = dict(dict(seed=args.seed, max_wavefronts=args.bounces, rng_mode=rng_mode), **kw)
        if args.scene == "mesh":
            return O.mesh_oracle(args.width, args.height, args.triangles, **kw)
        return O.shirley_oracle(args.width, args.height, **kw)

    legs, image, spp0 = [], None, 0
    for threads in thread_legs:
        o = make()
        O.set_num_threads(threads)
        t0 = time.perf_counter()
        spp = 0
        while True:
            o.render_sample()
            spp += 1
            el = time.perf_counter() - t0
            if el >= args.cpu_seconds or spp >= args.spp:
                break
        rays = int(o.totals()[0])
        legs.append({"threads": O.lib().orc_num_threads(), "value": round(rays / el / 1e6, 4), "unit": "Mrays/s",
                     "sample": f"the first {spp} of the frame's {args.spp} samples per pixel, {rays} rays, {el:.1f} s"})
        if image is None:
            image, spp0 = o.accumulated().copy(), spp
        o.close()
    best = max(legs, key=lambda l: l["value"])
    out = {"value": best["value"], "unit": "Mrays/s", "cores": best["threads"], "kind": "port",
           "host_cores": share["host_cpus"], "cpu_share": share, "legs": legs,
           "sample": (f"{best['sample']} ({args.width}x{args.height}, {args.bounces} bounces), OpenMP x{best['threads']}; the job's CPU share: "
                      f"{share['affinity']} CPUs in the affinity mask of {share['host_cpus']} on the box, cgroup quota "
                      f"{share['quota_cores'] if share['quota_cores'] is not None else 'none'} ({share['quota_source']})"
                      + ("" if len(legs) == 1 else f"; timed at {' and '.join(str(l['threads']) for l in legs)} threads, the larger figure is `value`"))}
    # single-thread figure: the serial twin of the oracle on one sample per pixel of the same frame
    o1 = make(serial=True)
    t0 = time.perf_counter()
    o1.render_sample()
    el1 = time.perf_counter() - t0
    out["single_thread"] = {"value": round(int(o1.totals()[0]) / el1 / 1e6, 4), "unit": "Mrays/s", "cores": 1,
                            "sample": f"1 sample per pixel, {int(o1.totals()[0])} rays, {el1:.1f} s"}
    out["build"] = "gcc -O3 -ffp-contract=off (oracle/Makefile)"
    out["march"] = "x86-64-v3 (not BASELINE.md's -march=native: the .so is built in the build container and travels to the GPU box)"
    o1.close()
    return out, spp0, image


def baseline_metric():
    """The headline metric string of BASELINE.json (committed next to this file), verbatim."""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        return "Mrays/s (extend+shade) at 1920×1080, 64 spp, 8 bounces; 1/2/4/8 GPU"


def provenance(W, gpu_index):
    """Where this line comes from: the command, the commit the library was built from (stamped at build time: the GPU box has no
    .git), whether the sources still match that build, the device and the date."""
    import datetime
    from wavefront_path_tracer_amd import _build
    info = _build.build_info()
    try:
        import torch
        name = torch.cuda.get_device_name(gpu_index) or ""
        arch = getattr(torch.cuda.get_device_properties(gpu_index), "gcnArchName", "")
        hw = W.device_info(gpu_index)
        dev = {"name": f"{name or 'AMD GPU'} ({arch}, {hw['compute_units']} CUs, {hw['total_memory_bytes'] >> 30} GiB)"}
    except Exception:
        dev = {"name": None}
    return {"command": "python " + " ".join([os.path.basename(sys.argv[0])] + sys.argv[1:]),
            "git_head": info.get("git_head"), "git_dirty_at_build": info.get("git_dirty"), "library_built_utc": info.get("built_utc"),
            "source_sha256": info.get("source_sha256"), "sources_match_build": info.get("sources_match_build"),
            "device": dev.get("name"), "date_utc": datetime.datetime.utcnow().strftime("%Y-%m-%dT%H:%M:%SZ")}


def workload_key(args, world=1):
    """What has to be equal for a committed profile to be a profile OF THIS RUN: the scene and its size, the frame, the samples per
    pixel and in flight, the bounces, the seed, the RNG mode and the number of ranks the frame is cut into."""
    fused = not (args.split_shade or args.unfused)
    batch = args.batch or min(args.spp, 128 if fused else 64)
    mode = args.rng_mode if args.rng_mode != "auto" else ("dispatch" if world == 1 else "pixel")
    return {"scene": args.scene, "triangles": args.triangles if args.scene == "mesh" else None, "width": args.width, "height": args.height,
            "spp": args.spp, "bounces": args.bounces, "seed": args.seed, "rng_mode": mode, "samples_in_flight": min(batch, args.spp), "ranks": world}


def profile_workload(pmc):
    """The workload a committed profile was taken on: the key it carries (profiles written since round 5), else re-derived from the bench
    command it records (the command line goes through this file's own parser); None if neither can be had."""
    if isinstance(pmc.get("workload_key"), dict):
        return pmc["workload_key"]
    cmd = (pmc.get("bench_command") or "").split()
    if "bench.py" not in [os.path.basename(c) for c in cmd]:
        return None
    argv = cmd[[os.path.basename(c) for c in cmd].index("bench.py") + 1:]
    try:
        a = parse(argv)
    except SystemExit:
        return None
    return workload_key(a, max(a.gpus, 1))


def load_pmc(scene, variant, key=None):
    """Committed rocprofv3 --pmc summary of THIS workload and loop variant (profiles/r*_pmc_<scene>_<variant>.json,
    written by profiles/summarize_rocprof.py from the same bench command), newest round first; None if there is none. With `key`
    (workload_key of the run at hand) a profile of any other workload -- another frame size, spp, scene size, RNG mode ... -- is no match."""
    import glob
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_{scene}_{variant}.json")), reverse=True):
        try:
            d = json.load(open(p))
        except Exception:
            continue
        if key is not None and profile_workload(d) != key:
            continue
        d["source"] = os.path.relpath(p, ROOT)
        return d
    return None


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))  # before torch or the GPU are touched

    import numpy as np
    import torch
    import wavefront_path_tracer_amd as W
    from wavefront_path_tracer_amd import tiles

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    if not torch.cuda.is_available() or W.device_count() < 1:
        sys.exit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    rehearsal = args.dist_backend == "gloo"
    n_dev = torch.cuda.device_count()
    # one GPU per rank. With fewer GPUs than ranks the ranks wrap around: RCCL then refuses the communicator with its own
    # "duplicate GPU" error and the run FAILS (only the gloo rehearsal may share a GPU)
    gpu_index = local_rank % n_dev
    torch.cuda.set_device(gpu_index)
    dist = None
    if world > 1:
        # torch.distributed is the rendezvous and the control plane only (barrier, max of the timings, the 128-byte RCCL
        # id): gloo on CPU tensors. The data plane -- the one gather of the frame -- is RCCL over xGMI behind the C ABI
        # (wfpt_comm_init / wfpt_gather_accumulated), so a host without PyTorch shards exactly the same way.
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    coll_dev = torch.device("cpu")  # where the control-plane tensors live

    mode_name = args.rng_mode if args.rng_mode != "auto" else ("dispatch" if world == 1 else "pixel")
    rng_mode = W.RNG_DISPATCH if mode_name == "dispatch" else W.RNG_PIXEL
    flags = ((W.FLAG_SPLIT_SHADE if args.split_shade else 0) | (W.FLAG_NO_GRAPH if args.no_graph else 0) |
             (W.FLAG_UNFUSED if args.unfused else 0) | (W.FLAG_BINARY_BVH if args.binary_bvh else 0) |
             (W.FLAG_NO_REFILL if args.no_refill else 0) | (W.FLAG_NO_LDS_SCENE if args.no_lds_scene else 0) |
             (W.FLAG_EXACT_TRAVERSAL if args.exact_traversal else 0) | (W.FLAG_NO_BINNING if args.no_binning else 0) | (W.FLAG_BINNING if args.binning else 0))
    # samples in flight per launch = the whole frame's samples (64): the late wavefronts are small, and a launch of few work
    # items per workgroup ends on a long tail (32 / 64 / 128 in flight: 18.8 / 19.4 / 19.6 Grays/s on a 128-spp job)
    fused = not (args.split_shade or args.unfused)
    batch = args.batch or min(args.spp, 128 if fused else 64)
    kw = dict(seed=args.seed, max_wavefronts=args.bounces, rng_mode=rng_mode, flags=flags, tile_rank=rank,
              tile_world=world, device=gpu_index, batch=batch)
    if args.scene == "mesh":
        pt = W.mesh_path_tracer(args.width, args.height, args.triangles, **kw)
        scene_name = (f"random triangle soup (BASELINE config 5: {args.triangles} triangles, seed {args.seed}; build extension, "
                      "the reference has no triangle code)")
    else:
        pt = W.shirley_path_tracer(args.width, args.height, **kw)
        scene_name = f"Shirley random-spheres (scene.rs:48-107, seed {args.seed})"

    loop_kind = pt.loop_kind  # the loop this context enqueues, as the library decided it (flags, RNG mode, slab size, scene)

    def sync():
        pt.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    def fail(msg):
        print(f"[bench rank {rank}] {msg}", file=sys.stderr, flush=True)
        if dist is not None:
            dist.destroy_process_group()
        sys.exit(3)

    use_rccl = (world > 1 and not rehearsal) or (world == 1 and args.force_rccl)
    watchdog = None
    if use_rccl and world > 1:
        # a communicator or a first send / receive that never completes would otherwise sit until the caller's own time limit:
        # say where it stopped and end the rank (the warm-up frames disarm this)
        import threading

        def stuck():
            print(f"[bench rank {rank}] the RCCL communicator / first gather did not complete within 300 s; giving up", file=sys.stderr, flush=True)
            os._exit(4)
        watchdog = threading.Timer(300.0, stuck)
        watchdog.daemon = True
        watchdog.start()
    gather_path = "none" if not (world > 1 or use_rccl) else (
        "gloo through host memory (REHEARSAL)" if not use_rccl else "RCCL send/recv to rank 0 behind the C ABI")
    if use_rccl:
        # The RCCL bootstrap needs a socket interface; on one node the loopback always works (the data itself goes over xGMI)
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        err = None
        try:
            uid = [W.comm_unique_id() if rank == 0 else None]
        except W.WfptError as e:
            uid, err = [None], str(e)
        if dist is not None:
            dist.broadcast_object_list(uid, src=0)
        if uid[0] is not None:
            try:
                pt.comm_init(uid[0], rank, world)  # collective: ncclCommInitRank on every rank's own GPU
            except W.WfptError as e:
                err = str(e)
        # every rank must take the same path: if the communicator could not be built on any rank, every rank stops
        bad = 0 if (err is None and uid[0] is not None) else 1
        if dist is not None:
            flag_t = torch.tensor([bad], dtype=torch.int32)
            dist.all_reduce(flag_t, op=dist.ReduceOp.MAX)
            bad = int(flag_t.item())
        if bad:
            fail(f"the RCCL communicator could not be built ({err or 'failed on another rank'}); {world} ranks on {n_dev} GPU(s). "
                 "One GPU per rank is required (use --dist-backend gloo only to rehearse the sharding on one GPU)")

    def gather():
        if use_rccl:
            pt.gather_accumulated()  # peers -> rank 0 over xGMI, de-interleaved on rank 0's GPU; asynchronous on the context's stream
            return None
        if world == 1:
            return None
        # rehearsal: ranks share a GPU (which RCCL refuses): slabs go through host memory (gloo)
        return tiles.gather_slabs(pt.accumulated(), rank, world, args.width, args.height)

    def render_frame(spp):
        """One step: a whole frame of `spp` samples per pixel, frames 1..spp of RenderProgress, then the job's only collective."""
        pt.reset_progress()
        pt.render(spp)
        return gather()

    for _ in range(max(args.warmup, 1)):  # also primes (untimed) the captured launch shapes and warms the communicator
        render_frame(args.spp)
    sync()
    if watchdog is not None:
        watchdog.cancel()
    rays0 = pt.totals().copy()
    t0 = time.perf_counter()
    frame = None
    for _ in range(args.steps):        # EXACTLY K steps
        frame = render_frame(args.spp)
    pt.synchronize()
    own_elapsed = time.perf_counter() - t0  # this rank's own K steps, its leg of the gather included, before it waits for the others
    sync()
    elapsed = time.perf_counter() - t0
    rays = pt.totals() - rays0     # [rays traced by extend, hits, misses] on this rank
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        r = torch.tensor(rays.astype(np.int64), device=coll_dev)
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        rays_total = r.cpu().numpy().astype(np.uint64)
    else:
        rays_total = rays
    if use_rccl:
        if rank == 0:
            frame = pt.gathered()  # outside the timed region: the frame crosses PCIe only for the dump / the check
    elif world == 1:
        frame = pt.accumulated()
    # ---- where an N > 1 figure comes from ("scaling factor, gather time"), all OUTSIDE the timed region:
    #   rank_ms_per_step         every rank's own wall time per step of the timed loop (its renders + its leg of the gather)
    #   rank_render_ms_per_step  the same K frames once more WITHOUT the gather, every rank timing itself: what the slab costs
    #   gather_ms                one gather timed alone, every rank idle when it starts (hipEvent pair on the context's stream behind the
    #                            C ABI, wfpt_gather_accumulated_timed; the gloo rehearsal: wall time of the host-memory gather)
    #   value_per_rank_ceiling   the job's rays / the slowest rank's render time: N x the slowest rank, the curve's ceiling before the gather
    scaling_detail = None
    if world > 1 or use_rccl:
        def per_rank(x):
            if dist is None:
                return [float(x)]
            xs = [None] * world
            dist.all_gather_object(xs, float(x))
            return xs
        rank_ms = per_rank(own_elapsed / args.steps * 1e3)
        sync()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            pt.reset_progress()
            pt.render(args.spp)
        pt.synchronize()
        render_ms = per_rank((time.perf_counter() - t1) / args.steps * 1e3)
        sync()
        if use_rccl:
            g_ms = pt.gather_accumulated_timed()
        else:
            t2 = time.perf_counter()
            frame = gather()
            g_ms = (time.perf_counter() - t2) * 1e3
        gather_ms = per_rank(g_ms)
        sync()
        scaling_detail = {"rank_ms_per_step": [round(x, 4) for x in rank_ms], "rank_render_ms_per_step": [round(x, 4) for x in render_ms],
                          "gather_ms": round(max(gather_ms), 4), "gather_ms_per_rank": [round(x, 4) for x in gather_ms],
                          "gather_ms_note": ("one gather timed alone after a barrier, every rank idle at its start; hipEvent pair around wfpt_gather_accumulated "
                                             "on each rank's stream (rank 0: the receives + the de-interleave; peers: their send), outside the timed region"
                                             if use_rccl else "REHEARSAL: wall time of the gloo gather through host memory, outside the timed region")}
    gather_check = None
    if world > 1:
        # The gather's send / receive legs have no multi-GPU test on the one-GPU boxes this is developed on, so every N > 1 run
        # checks them itself (untimed): each rank's own slab, band by band, must be what rank 0 holds at those bands' rows.
        import zlib
        mine = np.ascontiguousarray(pt.accumulated(), "<f4").reshape(-1, args.width, 3)
        crc = 0
        for local, band in enumerate(tiles.bands_of(rank, world, args.height)):
            n_rows = min(8, args.height - band * 8)
            crc = zlib.crc32(mine[local * 8:local * 8 + n_rows].tobytes(), crc)
        crcs = [None] * world
        dist.all_gather_object(crcs, crc)
        bad_ranks = []
        if rank == 0:
            full = frame.cpu().numpy() if hasattr(frame, "cpu") else np.asarray(frame)
            full = np.ascontiguousarray(full, "<f4").reshape(args.height, args.width, 3)
            for r in range(world):
                c = 0
                for band in tiles.bands_of(r, world, args.height):
                    c = zlib.crc32(full[band * 8:min(band * 8 + 8, args.height)].tobytes(), c)
                if c != crcs[r]:
                    bad_ranks.append(r)
        flag = [bad_ranks]
        dist.broadcast_object_list(flag, src=0)
        if flag[0]:
            fail(f"the gathered frame differs from the slabs of rank(s) {flag[0]} ({gather_path})")
        gather_check = f"every rank's slab equals its bands of the gathered frame (crc32 per rank, {world} ranks)"
    if world == 1 and args.force_rccl:  # the one-rank communicator's assembled frame must be the context's own image
        if not np.array_equal(np.ascontiguousarray(frame).view(np.uint32), np.ascontiguousarray(pt.accumulated()).view(np.uint32)):
            fail("RCCL branch: the gathered frame differs from the accumulated image")

    # ---- the anchor of a scaling sweep. N > 1 runs are pixel-keyed (band sharding must not change the image), N = 1 runs
    # dispatch-keyed (reference-faithful) by default, and the pixel-keyed chain traces ~6 % more rays per frame: a 1 -> N ratio
    # should compare like with like, so the N = 1 line also carries the pixel-keyed figure of the same K frames.
    pixel_anchor = None
    if world == 1 and mode_name == "dispatch" and args.rng_mode == "auto" and not args.no_pixel_anchor:
        kw2 = dict(kw, rng_mode=W.RNG_PIXEL)
        pt2 = (W.mesh_path_tracer(args.width, args.height, args.triangles, **kw2) if args.scene == "mesh"
               else W.shirley_path_tracer(args.width, args.height, **kw2))
        for _ in range(max(min(args.warmup, 2), 1)):
            pt2.reset_progress(); pt2.render(args.spp)
        pt2.synchronize()
        r2 = pt2.totals().copy()
        t2 = time.perf_counter()
        for _ in range(args.steps):
            pt2.reset_progress(); pt2.render(args.spp)
        pt2.synchronize()
        e2 = time.perf_counter() - t2
        pixel_anchor = {"value": round(float((pt2.totals() - r2)[0]) / e2 / 1e6, 3),
                        "note": ("`value` is dispatch-keyed (shade.wgsl:72's RNG key, the reference's); N > 1 runs are pixel-keyed. `value_pixel_mode` is the "
                                 f"same {args.steps} frames pixel-keyed on this GPU ({round(e2 / args.steps * 1e3, 4)} ms per frame, "
                                 f"{int((pt2.totals() - r2)[0])} rays): the N = 1 anchor a 1 -> N scaling ratio should use "
                                 "(or run every N with --rng-mode pixel)")}
        pt2.close()

    # ---- per-stage times and the roofline of the dominant kernel: the same K frames again with hipEvent pairs around
    # every launch on the context's stream (a second pass, so the events do not perturb `value`)
    stage = None
    if not args.no_stage_times:
        ms = np.zeros(W.STAGE_COUNT, np.float64)
        launches = np.zeros(W.STAGE_COUNT, np.int64)
        r0, w0 = pt.totals().copy(), pt.wavefront_totals().astype(np.float64)
        for _ in range(args.steps):
            pt.reset_progress()
            m, l = pt.render_timed(args.spp)  # same batching as pt.render
            ms += m
            launches += l
        rt = pt.totals() - r0
        wt = pt.wavefront_totals().astype(np.float64) - w0  # rows (rays traced, hits, misses) per wavefront over these K frames
        shade_ms = float(sum(ms[W.STAGES[k]] for k in ("shade", "shade_lambertian", "shade_metal", "shade_dielectric")))
        refill = fused and args.scene == "mesh" and not args.no_refill
        other = None
        if fused:
            # Two candidates for "the dominant kernel", both priced with SURVEY 8(d)'s per-unit figures times the units they process:
            #   middle: the bounce launches of wavefronts 1 .. max-1 = shade(b-1) + extend(b) [+ miss_kernel(b-1)];
            #   first : the launch of wavefront 0 = generate_rays + extend(0).
            # The one with the larger share of the frame is `roofline`; the other one is priced beside it (`roofline.other_launch`).
            shaded, applied = wt[:-1, 1].sum(), wt[:-1, 2].sum()        # hits / misses of wavefronts 0 .. max-2
            rays_k, hits_out, miss_out = wt[1:, 0].sum(), wt[1:, 1].sum(), wt[1:, 2].sum()
            rays_0, hits_0, miss_0 = wt[0, 0], wt[0, 1], wt[0, 2]
            pixels = float(args.width) * args.height * args.spp * args.steps / world
            if refill:  # shade, miss_kernel, generate_rays and the compaction are launches of their own there: these launches are extend alone
                mid = ("refill_kernel<middle> (four-wide extend of one wavefront, dynamic lane refill; its rays come shaded from shade_rays_kernel)",
                       B_EXTEND_RAY * rays_k + B_EXTEND_HIT * hits_out + B_EXTEND_MISS * miss_out,
                       64.0 * rays_k)  # one 32-byte ray in, one dense 32-byte result out, per ray
                first = ("refill_kernel<first> (four-wide extend of wavefront 0, dynamic lane refill; its primary rays come from generate_dense_kernel)",
                         B_EXTEND_RAY * rays_0 + B_EXTEND_HIT * hits_0 + B_EXTEND_MISS * miss_0, 64.0 * rays_0)
            else:
                kn = "bounce_binned_kernel" if loop_kind == "fused_binned" else "bounce_kernel"
                mid = (kn + "<middle> (shade + extend + miss_kernel of one wavefront)",
                       B_SHADE_HIT * shaded + B_EXTEND_RAY * rays_k + B_EXTEND_HIT * hits_out + B_EXTEND_MISS * miss_out + B_MISS * applied,
                       # what the fused design itself has to move: record in (32), throughput RMW (2 x 16, padded pixels), record out
                       # (32) / miss out (8), applied miss (8 + 32) -- no extension-ray queue, no hit-queue gather
                       64.0 * shaded + 32.0 * hits_out + 8.0 * miss_out + 40.0 * applied)
                first = (kn + "<first> (generate_rays + extend of wavefront 0)",
                         B_GENERATE_PIXEL * pixels + B_EXTEND_RAY * rays_0 + B_EXTEND_HIT * hits_0 + B_EXTEND_MISS * miss_0,
                         16.0 * pixels + 32.0 * hits_0 + 8.0 * miss_0)  # image reset (16, padded pixel), record / miss out: the ray never leaves registers
            first_dominant = ms[W.STAGES["bounce_first"]] > ms[W.STAGES["bounce"]]
            (kname, k_bytes, k_own), kstage = (first, "bounce_first") if first_dominant else (mid, "bounce")
            o_name, o_bytes, o_own = mid if first_dominant else first
            o_stage = "bounce" if first_dominant else "bounce_first"
            other = {"kernel": o_name, "ms": float(ms[W.STAGES[o_stage]]), "n": int(launches[W.STAGES[o_stage]]), "bytes": float(o_bytes), "own": float(o_own)}
            chain_ms = float(sum(ms[W.STAGES[k]] for k in ("bounce_first", "bounce", "bounce_last", "scan", "compact", "miss_kernel"))) + (shade_ms if refill else 0.0)
        else:
            kname, kstage = "extend_kernel", "extend"
            k_bytes = B_EXTEND_RAY * float(rt[0]) + B_EXTEND_HIT * float(rt[1]) + B_EXTEND_MISS * float(rt[2])
            k_own = k_bytes + 8.0 * float(rt[2])  # + (dir.y, pixel) handed to miss_kernel through the miss queue
            chain_ms = float(ms[W.STAGES["extend"]] + ms[W.STAGES["scan"]] + ms[W.STAGES["miss_kernel"]]) + shade_ms
        stage = {"ms": {k: round(float(ms[v]), 4) for k, v in W.STAGES.items() if launches[v]},
                 "launches": {k: int(launches[v]) for k, v in W.STAGES.items() if launches[v]},
                 "kname": kname, "k_ms": float(ms[W.STAGES[kstage]]), "k_n": int(launches[W.STAGES[kstage]]),
                 "k_bytes": float(k_bytes), "k_own": float(k_own), "rays": int(rt[0]), "kstage": kstage if fused else "extend", "other": other,
                 "wavefront_rays": [int(x) for x in wt[:, 0] if x > 0],  # rays traced per wavefront, summed over these K frames
                 "extend_shade_mrays_s": float(rt[0]) / (chain_ms * 1e-3) / 1e6 if chain_ms > 0 else 0.0}

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    out = {
        "metric": baseline_metric(),  # BASELINE.json's metric, verbatim; config.workload says what one step rendered
        "value": round(float(rays_total[0]) / elapsed / 1e6, 3),
        "unit": "Mrays/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,  # the reference publishes no number for this metric
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"{scene_name}, {args.width}x{args.height}, {args.spp} spp, {args.bounces} bounces",
                   "step": (f"one frame: accumulation reset, {args.spp} samples per pixel (each generate_rays -> {args.bounces} x (extend, scan, "
                            "shade, miss_kernel) -> accumulate)" + (", one gather of the frame to rank 0" if world > 1 else "")),
                   "rng_mode": mode_name, "shade": "per-material" if args.split_shade else "unified",
                   "loop": {"stages": "stage kernels one by one", "fused": "fused bounce launches, hit queue in thread order",
                            "fused_binned": "fused bounce launches, hit queue binned by cost class (work item = 512 hits of one class)",
                            "refill": "four-wide traversal with dynamic lane refill (generate / shade / compact as launches of their own)"}[loop_kind],
                   "loop_kind": loop_kind,
                   "traversal": (("LDS-resident binary BVH" + (", reference slab arithmetic (exact-traversal)" if args.exact_traversal else ""))
                                 if args.scene == "shirley" and not args.no_lds_scene else
                                 ("binary BVH from HBM" if (args.binary_bvh or args.scene == "shirley") else
                                  "four-wide collapsed BVH (64-byte quantised nodes, top of the tree staged in LDS) from HBM")),
                   "launch": "direct" if args.no_graph else "hipGraph",
                   "samples_in_flight": sorted({min(batch, args.spp), args.spp % batch} - {0}, reverse=True),
                   "parallelism": "single GPU" if world == 1 else
                   f"pixel bands of 8 rows over {world} ranks + 1 gather per frame ({gather_path})",
                   "gather": gather_path,
                   "gather_check": gather_check,
                   "gather_verified": ("in this run (gather_check): every N > 1 run checks its own gather; before the first multi-GPU run the ncclSend / ncclRecv "
                                       "legs had never executed on hardware (one-GPU development boxes)" if (world > 1 and use_rccl) else
                                       "n/a here; the ncclSend / ncclRecv legs of the N > 1 gather have never run on hardware (one-GPU development boxes): "
                                       "the first multi-GPU run verifies them itself (gather_check)"),
                   "rays_traced": int(rays_total[0])},
    }
    if scaling_detail is not None:
        slowest = max(scaling_detail["rank_render_ms_per_step"])
        scaling_detail["value_per_rank_ceiling"] = round(float(rays_total[0]) / args.steps / (slowest * 1e-3) / 1e6, 3) if slowest > 0 else None
        scaling_detail["value_per_rank_ceiling_note"] = ("rays of one step / the slowest rank's render time without the gather (= N x the slowest rank): what "
                                                          "`value` would be with a free gather; value / this = the gather's and the barrier's share")
        out.update(scaling_detail)
    out["provenance"] = provenance(W, gpu_index)
    out["workload_key"] = workload_key(args, world)
    if pixel_anchor is not None:
        out["value_pixel_mode"] = pixel_anchor["value"]
        out["config"]["rng_mode_anchor"] = pixel_anchor["note"]
    info = W.device_info(gpu_index)
    # double data rate: 2 transfers per memory clock; what hipDeviceProp_t reports for this board, next to the spec figure
    peak_device = 2.0 * info["memory_clock_khz"] * 1e3 * info["memory_bus_width_bits"] / 8.0 / 1e9
    if stage is not None:
        per_launch_bytes = stage["k_bytes"] / max(stage["k_n"], 1)
        avg_s = stage["k_ms"] * 1e-3 / max(stage["k_n"], 1)
        achieved = per_launch_bytes / avg_s / 1e9 if avg_s > 0 else 0.0
        variant = "split" if args.split_shade else ("unfused" if args.unfused else "fused")
        if args.binary_bvh:
            variant += "_binary"
        if args.no_refill:
            variant += "_norefill"
        if args.no_lds_scene:
            variant += "_nolds"
        if args.exact_traversal:
            variant += "_exact"
        binned = loop_kind == "fused_binned"  # what the context really enqueues (wfpt_loop_kind_of), not what the flags suggest
        if binned:
            variant += "_binned"
        # PMC counters cannot be collected inside a plain run: `traffic` and `secondary` come from the committed rocprofv3
        # profile of THIS command -- same scene and size, frame, samples per pixel and in flight, bounces, RNG mode, ranks and loop
        # variant (workload_key) -- and are labelled as such; a profile of any other workload is no match and the fields stay null
        pmc = load_pmc(args.scene, variant, workload_key(args, world))
        per_kernel = ((pmc or {}).get("launches") or {})
        pk = per_kernel.get(stage["kstage"], {})
        fabric = pk.get("fabric_bytes_per_launch")
        lds_scene = args.scene == "shirley" and not args.no_lds_scene
        out["roofline"] = {"kernel": stage["kname"],
                           "bound": ("valu" if lds_scene else "l1"),
                           "bound_note": ("bound by wave64 VALU issue (LDS-resident BVH, no HBM traffic for the scene); achieved / peak / frac are the "
                                          "HBM figures BASELINE asks for, secondary holds the VALU figures (DESIGN.md section 4)"
                                          if lds_scene else
                                          "the L1 -> register (TA) path of the per-lane node fetches with wave64 VALU issue near the ceiling of the four-box "
                                          "visit's instruction mix beside it and the L2-miss (fabric) traffic behind (`secondary`, `traffic`); achieved / "
                                          "peak / frac are the HBM figures BASELINE asks for, of this launch's own (extend) bytes (DESIGN.md section 8)"),
                           "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                           "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                           "traffic": round(fabric, 1) if fabric else None,
                           "traffic_source": (f"from_profile: {pmc['source']} (rocprofv3 --pmc of this workload -- same scene and size, frame, samples per pixel and "
                                              f"in flight, bounces, RNG mode and loop; the profile's own command: `{pmc.get('bench_command')}`, commit {pmc.get('git_head')})"
                                              if fabric else None),
                           "traffic_note": ("FETCH_SIZE x the factor calibrated for this kernel's access shape + WRITE_SIZE: requests on the fabric side of L2 "
                                            "(MI355X_MICROARCH.md: Infinity-Cache hits are counted, so this is fabric traffic, an upper bound on HBM traffic); "
                                            + (pk.get("calibration_note") or "")) if fabric else None,
                           "traffic_per_algorithmic_byte": round(fabric / per_launch_bytes, 3) if fabric and per_launch_bytes else None,
                           "peak_device": round(peak_device, 1),
                           "peak_device_source": f"hipDeviceProp_t: 2 x {info['memory_clock_khz']} kHz x {info['memory_bus_width_bits']} bit / 8",
                           "algorithmic_bytes_per_launch": round(per_launch_bytes, 1),
                           "fused_design_bytes_per_launch": round(stage["k_own"] / max(stage["k_n"], 1), 1),
                           "avg_launch_us": round(avg_s * 1e6, 3), "launches": stage["k_n"]}
        if args.scene == "mesh":  # the scene itself is the compulsory read of an HBM-resident scene: nodes + triangles
            scene_bytes = 32.0 * 2.0 * args.triangles + 48.0 * args.triangles  # ~2n reference nodes of 32 B + n triangles of 48 B (112 MB at 1 M)
            out["roofline"]["compulsory_scene_bytes"] = round(scene_bytes, 1)
            if fabric:
                out["roofline"]["traffic_per_compulsory_scene_byte"] = round(fabric / scene_bytes, 2)
        if stage["other"]:
            o = stage["other"]
            o_avg = o["ms"] * 1e-3 / max(o["n"], 1)
            o_bytes = o["bytes"] / max(o["n"], 1)
            ok = per_kernel.get("bounce_first" if stage["kstage"] == "bounce" else "bounce", {})
            out["roofline"]["other_launch"] = {"kernel": o["kernel"], "launches": o["n"], "avg_launch_us": round(o_avg * 1e6, 3),
                                               "algorithmic_bytes_per_launch": round(o_bytes, 1),
                                               "achieved": round(o_bytes / o_avg / 1e9, 2) if o_avg > 0 else 0.0,
                                               "frac": round(o_bytes / o_avg / 1e9 / HBM_PEAK_GBS, 5) if o_avg > 0 else 0.0,
                                               "traffic": round(ok["fabric_bytes_per_launch"], 1) if ok.get("fabric_bytes_per_launch") else None,
                                               "secondary": ok.get("secondary")}
        if pk.get("secondary"):
            sec = dict(pk["secondary"])
            sec["source"] = f"from_profile: {pmc['source']} ({pmc.get('samples_in_flight')} samples in flight)"
            out["roofline"]["secondary"] = sec
        out["wavefront_rays"] = stage["wavefront_rays"]
        out["stage_ms"] = stage["ms"]
        out["stage_launches"] = stage["launches"]
        out["extend_shade_mrays_s"] = round(stage["extend_shade_mrays_s"], 3)
    if world == 1 and not args.no_cpu_baseline:
        cb, n_cpu, cpu_image = cpu_baseline(args, rng_mode)
        # the oracle is the checker: one UNTIMED frame of the samples it got through, rendered again by the GPU with the
        # timed run's batching, must be its image bit for bit
        pt.reset_progress()
        pt.render(n_cpu)
        same = np.array_equal(np.ascontiguousarray(cpu_image).view(np.uint32), np.ascontiguousarray(pt.accumulated()).view(np.uint32))
        cb["gpu_image_vs_oracle"] = ("bit-identical" if same else "DIFFERENT") + f" ({n_cpu} spp frame)"
        cb["gpu_over_cpu"] = round(out["value"] / cb["value"], 1) if cb["value"] else None  # against the LARGER CPU figure
        out["cpu_baseline"] = cb
    if args.dump and frame is not None:
        if hasattr(frame, "cpu"):
            frame = frame.cpu().numpy().reshape(-1, 3)
        rgb = W.tonemap_rgb8(frame, args.spp)
        with open(args.dump, "wb") as f:
            f.write(b"P6\n%d %d\n255\n" % (args.width, args.height))
            f.write(rgb.tobytes())
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
