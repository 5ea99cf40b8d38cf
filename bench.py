#!/usr/bin/env python3
"""bench.py -- headline benchmark of the wavefront kernel chain on MI355X.

Workload (BASELINE.json configs[1]): seeded Shirley random-spheres scene, 1920x1080, 8 bounces; one STEP is
one sample per pixel of the whole frame: generate_rays -> 8 x (extend, scan, shade, miss_kernel) -> accumulate.
Default K = 64 steps = the 64 spp the metric is quoted on. Inputs (scene, BVH, camera) are resident in HBM
before the timed region; the timed region holds K steps (and, for N > 1, the final RCCL gather).

  python bench.py --gpus 1 --steps 64 --warmup 4
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU; the frame is sharded by 8-row pixel bands (band k -> rank k % N), no collective
inside the bounce loop, one gather of the accumulated slabs to rank 0 at the end (scaling = "strong": the
frame is fixed, per-GPU work shrinks). Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

# The CPU baseline runs an OpenMP oracle; a GPU box shows every host CPU but the job owns 16 per GPU. Must be set
# before torch (which loads an OpenMP runtime) is imported.
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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=4)
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
    ap.add_argument("--no-refill", action="store_true", help="mesh scene: fused bounce kernel (lanes keep their ray) instead of dynamic lane refill")
    ap.add_argument("--binary-bvh", action="store_true", help="mesh scene: walk the binary tree instead of the four-wide collapse")
    ap.add_argument("--unfused", action="store_true", help="run the stage kernels one by one (extend, scan, shade, miss_kernel per wavefront)")
    ap.add_argument("--batch", type=int, default=0, help="samples kept in flight per launch (0 = library default, 16)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0,
                    help="CPU-baseline sample: whole samples per pixel of the same workload until this much time is spent")
    ap.add_argument("--no-stage-times", action="store_true")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                    help="nccl = the frame is gathered by RCCL over xGMI through the C ABI (one GPU per rank). gloo: rehearsal "
                         "only -- ranks may share one GPU, the gather goes through host memory")
    ap.add_argument("--dump", default=None, help="write the tone-mapped frame (PPM) here (rank 0)")
    return ap.parse_args()


def cpu_baseline(args, rng_mode, gpu_frame=None):
    """The oracle (kind "port": the build's own CPU restatement of the reference's chain; the reference's
    cpu_wavefront_pt has no source) timed on this box's host cores, OpenMP, on a bounded sample of the SAME
    workload: same scene/seed/camera/size/bounces, fewer samples per pixel (Mrays/s is spp-invariant)."""
    from oracle import oracle as O
    if args.scene == "mesh":
        o = O.mesh_oracle(args.width, args.height, args.triangles, seed=args.seed, max_wavefronts=args.bounces, rng_mode=rng_mode)
    else:
        o = O.shirley_oracle(args.width, args.height, seed=args.seed, max_wavefronts=args.bounces, rng_mode=rng_mode)
    t0 = time.perf_counter()
    spp = 0
    while True:
        o.render_sample()
        spp += 1
        el = time.perf_counter() - t0
        if el >= args.cpu_seconds or spp >= max(args.steps, 4):
            break
    rays = int(o.totals()[0])
    cores = O.lib().orc_num_threads()
    out = {"value": round(rays / el / 1e6, 4), "unit": "Mrays/s", "cores": cores, "kind": "port",
           "sample": f"{spp} of the workload's {args.steps} samples per pixel ({args.width}x{args.height}, {args.bounces} bounces, "
                     f"{rays} rays, {el:.1f} s, OpenMP x{cores})"}
    # the oracle is the checker: when it got through all K samples in its time budget, its accumulated image is the
    # image the timed GPU steps must have produced, bit for bit (outside the timed region, costs one comparison)
    if gpu_frame is not None and spp == args.steps:
        import numpy as np
        same = np.array_equal(np.ascontiguousarray(o.accumulated()).view(np.uint32), np.ascontiguousarray(gpu_frame).view(np.uint32))
        out["gpu_image_vs_oracle"] = "bit-identical" if same else "DIFFERENT"
    o.close()
    # single-thread figure: the serial twin of the oracle on one sample per pixel of the same frame
    mk = O.mesh_oracle if args.scene == "mesh" else O.shirley_oracle
    kw = dict(seed=args.seed, max_wavefronts=args.bounces, rng_mode=rng_mode, serial=True)
    o1 = mk(args.width, args.height, args.triangles, **kw) if args.scene == "mesh" else mk(args.width, args.height, **kw)
    t0 = time.perf_counter()
    o1.render_sample()
    el1 = time.perf_counter() - t0
    out["single_thread"] = {"value": round(int(o1.totals()[0]) / el1 / 1e6, 4), "unit": "Mrays/s", "cores": 1,
                            "sample": f"1 sample per pixel, {int(o1.totals()[0])} rays, {el1:.1f} s"}
    out["build"] = "gcc -O3 -march=x86-64-v3 -ffp-contract=off (oracle/Makefile)"
    o1.close()
    return out


def baseline_metric():
    """The headline metric string of BASELINE.json (committed next to this file), verbatim."""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        return "Mrays/s (extend+shade) at 1920\u00d71080, 64 spp, 8 bounces; 1/2/4/8 GPU"


def load_pmc(scene, variant):
    """Committed rocprofv3 --pmc summary of THIS workload and loop variant (profiles/r*_pmc_<scene>_<variant>.json,
    written by profiles/summarize_rocprof.py from the same bench command), newest round first; None if there is none."""
    import glob
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_{scene}_{variant}.json")), reverse=True):
        try:
            d = json.load(open(p))
            d["source"] = os.path.relpath(p, ROOT)
            return d
        except Exception:
            continue
    return None


def main():
    args = parse()
    import numpy as np
    import torch
    import wavefront_path_tracer_amd as W
    from wavefront_path_tracer_amd import tiles

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N > 1 must be launched with torch.distributed.run (one process per GPU)")
        args.gpus = world
    if not torch.cuda.is_available() or W.device_count() < 1:
        sys.exit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    rehearsal = args.dist_backend == "gloo"
    gpu_index = local_rank % torch.cuda.device_count() if rehearsal else local_rank
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
             (W.FLAG_NO_REFILL if args.no_refill else 0) | (W.FLAG_NO_LDS_SCENE if args.no_lds_scene else 0))
    # samples in flight per launch: 64 at N=1 (32 / 64 / 128: 18.8 / 19.4 / 19.6 Grays/s on a 128-spp job: the late wavefronts are
    # small, and a launch of few work items per workgroup ends on a long tail); each rank of N holds 1/N of the pixels, so it
    # scales with N (up to the library's 128) to keep launches as large
    batch = args.batch or min(128, 64 * world)
    kw = dict(seed=args.seed, max_wavefronts=args.bounces, rng_mode=rng_mode, flags=flags, tile_rank=rank,
              tile_world=world, device=gpu_index, batch=batch)
    if args.scene == "mesh":
        pt = W.mesh_path_tracer(args.width, args.height, args.triangles, **kw)
        scene_name = (f"random triangle soup (BASELINE config 5: {args.triangles} triangles, seed {args.seed}; build extension, "
                      "the reference has no triangle code)")
    else:
        pt = W.shirley_path_tracer(args.width, args.height, **kw)
        scene_name = f"Shirley random-spheres (scene.rs:48-107, seed {args.seed})" 

    def sync():
        pt.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    gather_path = "none" if world == 1 else ("gloo through host memory (REHEARSAL)" if rehearsal else "RCCL send/recv to rank 0 behind the C ABI")
    if world > 1 and not rehearsal:
        # The RCCL bootstrap needs a socket interface; on one node the loopback always works (the data itself goes over xGMI)
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        err = None
        try:
            uid = [W.comm_unique_id() if rank == 0 else None]
        except W.WfptError as e:
            uid, err = [None], str(e)
        dist.broadcast_object_list(uid, src=0)
        if uid[0] is not None:
            try:
                pt.comm_init(uid[0], rank, world)  # collective: ncclCommInitRank on every rank's own GPU
            except W.WfptError as e:
                err = str(e)
        # every rank must take the same path: if the communicator could not be built anywhere, say so LOUDLY in the JSON line
        # and move the slabs through host memory instead, so that the run still measures the sharded render
        flags_t = torch.tensor([0 if (err is None and uid[0] is not None) else 1], dtype=torch.int32)
        dist.all_reduce(flags_t, op=dist.ReduceOp.MAX)
        if int(flags_t.item()):
            rehearsal_gather = True
            gather_path = f"FALLBACK: gloo through host memory, the RCCL communicator failed ({err or 'on another rank'})"
            print(f"[bench rank {rank}] {gather_path}", file=sys.stderr, flush=True)
        else:
            rehearsal_gather = False
    else:
        rehearsal_gather = rehearsal

    def gather():
        if world == 1:
            return None
        if rehearsal_gather:  # ranks share a GPU (which RCCL refuses) or RCCL is unusable: slabs go through host memory (gloo)
            return tiles.gather_slabs(pt.accumulated(), rank, world, args.width, args.height)
        pt.gather_accumulated()  # peers -> rank 0 over xGMI, de-interleaved on rank 0's GPU; asynchronous on the context's stream
        return None

    pt.render(args.warmup)
    # prime (untimed) the captured launch shapes the timed K steps will replay: full batches + the remainder
    for nb in {min(batch, args.steps), args.steps % batch}:
        if nb:
            pt.render(nb)
    if world > 1:
        gather()  # warm the communicator too
    pt.reset_progress()  # the timed K steps are frames 1..K: the frames the oracle renders in the cpu_baseline leg
    sync()
    rays0 = pt.totals().copy()
    t0 = time.perf_counter()
    pt.render(args.steps)          # EXACTLY K steps
    frame = gather()               # the job's only collective
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
    if world == 1:
        frame = pt.accumulated()
    elif not rehearsal_gather and rank == 0:
        frame = pt.gathered()  # outside the timed region: the frame crosses PCIe only for the dump / the check

    # ---- per-stage times and the roofline of the dominant kernel: same K steps again with hipEvent pairs around
    # every launch on the context's stream (a second pass, so the events do not perturb `value`)
    fused = not (args.split_shade or args.unfused)
    stage = None
    if not args.no_stage_times:
        ms = np.zeros(W.STAGE_COUNT, np.float64)
        launches = np.zeros(W.STAGE_COUNT, np.int64)
        r0, w0 = pt.totals().copy(), pt.wavefront_totals().astype(np.float64)
        m, l = pt.render_timed(args.steps)  # same batching as pt.render
        ms += m
        launches += l
        rt = pt.totals() - r0
        wt = pt.wavefront_totals().astype(np.float64) - w0  # rows (rays traced, hits, misses) per wavefront over these K steps
        shade_ms = float(sum(ms[W.STAGES[k]] for k in ("shade", "shade_lambertian", "shade_metal", "shade_dielectric")))
        if fused:
            # dominant kernel: the middle bounce launches = shade(b-1) + extend(b) + miss_kernel(b-1), b = 1 .. max-1.
            # Algorithmic bytes = SURVEY 8(d)'s per-unit figures times the units those launches process.
            kname, kstage = "bounce_kernel<middle> (shade + extend + miss_kernel of one wavefront)", "bounce"
            shaded, applied = wt[:-1, 1].sum(), wt[:-1, 2].sum()        # hits / misses of wavefronts 0 .. max-2
            rays, hits_out, miss_out = wt[1:, 0].sum(), wt[1:, 1].sum(), wt[1:, 2].sum()
            k_bytes = (B_SHADE_HIT * shaded + B_EXTEND_RAY * rays + B_EXTEND_HIT * hits_out + B_EXTEND_MISS * miss_out + B_MISS * applied)
            # what the fused design itself has to move: record in (32), throughput RMW (24), record out (32) / miss out (8),
            # applied miss (8 + 24) -- no extension-ray queue, no hit-queue gather
            k_own = 56.0 * shaded + 32.0 * hits_out + 8.0 * miss_out + 32.0 * applied
            chain_ms = float(sum(ms[W.STAGES[k]] for k in ("bounce_first", "bounce", "bounce_last", "scan")))
        else:
            kname, kstage = "extend_kernel", "extend"
            k_bytes = B_EXTEND_RAY * float(rt[0]) + B_EXTEND_HIT * float(rt[1]) + B_EXTEND_MISS * float(rt[2])
            k_own = k_bytes + 8.0 * float(rt[2])  # + (dir.y, pixel) handed to miss_kernel through the miss queue
            chain_ms = float(ms[W.STAGES["extend"]] + ms[W.STAGES["scan"]]) + shade_ms
        stage = {"ms": {k: round(float(ms[v]), 4) for k, v in W.STAGES.items() if launches[v]},
                 "launches": {k: int(launches[v]) for k, v in W.STAGES.items() if launches[v]},
                 "kname": kname, "k_ms": float(ms[W.STAGES[kstage]]), "k_n": int(launches[W.STAGES[kstage]]),
                 "k_bytes": float(k_bytes), "k_own": float(k_own), "rays": int(rt[0]),
                 "extend_shade_mrays_s": float(rt[0]) / (chain_ms * 1e-3) / 1e6 if chain_ms > 0 else 0.0}

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    out = {
        "metric": baseline_metric(),  # BASELINE.json's metric, verbatim; config.workload says what THIS run rendered (K = spp)
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
        "config": {"workload": f"{scene_name}, {args.width}x{args.height}, {args.steps} spp, {args.bounces} bounces",
                   "step": "one sample per pixel: generate_rays -> bounces x (extend, scan, shade, miss_kernel) -> accumulate",
                   "rng_mode": mode_name, "shade": "per-material" if args.split_shade else "unified",
                   "loop": "fused bounce launches" if fused else "stage kernels one by one",
                   "traversal": ("LDS-resident binary BVH" if args.scene == "shirley" else
                                 ("binary BVH from HBM" if args.binary_bvh else "four-wide collapsed BVH (128-byte nodes) from HBM")),
                   "launch": "direct" if args.no_graph else "hipGraph",
                   "samples_in_flight": sorted({min(batch, args.steps), args.steps % batch} - {0}, reverse=True),
                   "parallelism": "single GPU" if world == 1 else
                   f"pixel bands of 8 rows over {world} ranks + 1 gather ({gather_path})",
                   "rays_traced": int(rays_total[0])},
    }
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
        pmc = load_pmc(args.scene, variant)
        # PMC traffic is a property of (scene, loop variant): the profile stores HBM bytes per algorithmic byte of the
        # same kernel on the same workload, scaled here by this run's algorithmic bytes per launch; null without a profile
        ratio = (pmc or {}).get("hbm_bytes_per_algorithmic_byte")
        out["roofline"] = {"kernel": stage["kname"], "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                           "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                           "traffic": round(ratio * per_launch_bytes, 1) if ratio else None,
                           "traffic_source": (pmc or {}).get("source"),
                           "peak_device": round(peak_device, 1),
                           "peak_device_source": f"hipDeviceProp_t: 2 x {info['memory_clock_khz']} kHz x {info['memory_bus_width_bits']} bit / 8",
                           "algorithmic_bytes_per_launch": round(per_launch_bytes, 1),
                           "fused_design_bytes_per_launch": round(stage["k_own"] / max(stage["k_n"], 1), 1),
                           "avg_launch_us": round(avg_s * 1e6, 3), "launches": stage["k_n"],
                           "note": ("traversal is VALU-issue bound (LDS-resident BVH), not HBM-bound; see DESIGN.md section 4"
                                    if args.scene == "shirley" else
                                    "BVH read from HBM / Infinity Cache through L2; bound by random-line throughput behind the L1 (DESIGN.md section 8)")}
        if pmc and "secondary" in pmc:
            out["roofline"]["secondary"] = pmc["secondary"]  # VALU occupancy, active lanes: measured under rocprofv3 --pmc
        out["stage_ms"] = stage["ms"]
        out["stage_launches"] = stage["launches"]
        out["extend_shade_mrays_s"] = round(stage["extend_shade_mrays_s"], 3)
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args, rng_mode, frame)
    if args.dump and frame is not None:
        if hasattr(frame, "cpu"):
            frame = frame.cpu().numpy().reshape(-1, 3)
        rgb = W.tonemap_rgb8(frame, args.steps)
        with open(args.dump, "wb") as f:
            f.write(b"P6\n%d %d\n255\n" % (args.width, args.height))
            f.write(rgb.tobytes())
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
