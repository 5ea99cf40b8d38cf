#!/usr/bin/env python3
"""Condenses the rocprofv3 passes of tools/profile.sh (gpurun_out/prof_<TAG>_<pass>/...) into the small summaries
committed under profiles/.

    python profiles/summarize_rocprof.py gpurun_out TAG ROUND SCENE VARIANT [n_pixels]
      e.g.   ... gpurun_out p1 r02 shirley fused

Writes
  profiles/<ROUND>_kernel_stats_<SCENE>_<VARIANT>.csv   copy of rocprofv3 --kernel-trace --stats
  profiles/<ROUND>_pmc_<SCENE>_<VARIANT>.json           per-kernel mean counter values per launch + the summary of the
                                                         dominant kernel that bench.py reads for roofline.traffic
  profiles/<ROUND>_bench_line_<SCENE>_<VARIANT>.json    the bench line printed by the same command (stats pass)

Fabric bytes: FETCH_SIZE / WRITE_SIZE are in KiB and count the L2's memory-side requests (Infinity-Cache hits included: fabric
traffic, an upper bound on HBM traffic). On gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md "HBM");
other shapes are uncalibrated, so the read side is calibrated per access shape: streaming kernels on `accumulate_kernel` of the same
profile, whose byte count is known exactly (samples x 16 B/pixel of throughput images plus 12 B/pixel of `accumulated`, 16 B per
lane); the refill traversal's per-lane gathers of 64-byte nodes on tools/microbench_node_fetch.hip (profiles/r*_fetch_calibration.json).
Raw and corrected figures are both reported, per kernel (`launches`: the middle AND the first bounce launch). Every file written
carries the bench command, the commit and source digest the library was built from, the device and the date (from the bench line's
`provenance`).
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

NAMES = [("bounce_kernel<0", "bounce_first"), ("bounce_kernel<1", "bounce"), ("bounce_kernel<2", "bounce_last"),
         ("bounce_binned_kernel<0", "bounce_first"), ("bounce_binned_kernel<1", "bounce"), ("bounce_binned_kernel<2", "bounce_last"),
         ("scan_binned_kernel", "scan"), ("plan_kernel", "plan"),
         ("refill_kernel<0", "bounce_first"), ("refill_kernel<1", "bounce"), ("compact_kernel", "compact"),
         ("extend_kernel", "extend"), ("shade_kernel", "shade"), ("shade_rays_kernel", "shade_rays"), ("miss_kernel", "miss_kernel"), ("scan_kernel", "scan"),
         ("generate_rays_kernel", "generate_rays"), ("generate_dense_kernel", "generate_rays"), ("accumulate_kernel", "accumulate")]


def kname(full):
    for needle, name in NAMES:
        if needle in full:
            return name
    return None


def pmc(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        k = kname(r["Kernel_Name"])
        if k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def newest(pattern):
    """gpurun merges a pass's files into what an earlier session with the same tag left behind: keep the latest run's only."""
    files = sorted(glob.glob(pattern), key=os.path.getmtime)
    return files[-1:]


def load_gather_factor(here):
    """FETCH_SIZE calibration for per-lane GATHERS of 64-byte records as 4 x 16 B (the refill traversal's node fetch), measured by
    tools/microbench_node_fetch.hip calib under rocprofv3 --pmc FETCH_SIZE: newest profiles/r*_fetch_calibration.json."""
    for p in sorted(glob.glob(os.path.join(here, "r*_fetch_calibration.json")), reverse=True):
        try:
            d = json.load(open(p))
            # bytes that cross the fabric per FETCH_SIZE byte: every L2 miss is one 128-byte line request tallied as 64 bytes, whatever part
            # of the line the lanes asked for (gather128 reads whole lines: its factor IS that ratio); a 64-byte node uses
            # gather64 / gather128 = 0.48 of what its miss moves
            return float(d["gather_128B_record"]["bytes_per_FETCH_SIZE_byte"]), os.path.basename(p)
        except Exception:
            continue
    return None, None


def main():
    src, tag, rnd, scene, variant = sys.argv[1:6]
    n_pixels = int(sys.argv[6]) if len(sys.argv) > 6 else 1920 * 1080
    here = os.path.dirname(os.path.abspath(__file__))
    base = os.path.join(src, f"prof_{tag}_")
    stats = newest(base + "stats/*/*_kernel_stats.csv")
    if stats:
        shutil.copy(stats[0], os.path.join(here, f"{rnd}_kernel_stats_{scene}_{variant}.csv"))
    line = json.load(open(base + "stats.json"))
    json.dump(line, open(os.path.join(here, f"{rnd}_bench_line_{scene}_{variant}.json"), "w"), indent=1)
    out = {}
    for d in ("fetch", "write", "sq", "sq2", "sq3", "sq4", "sq5", "sq6", "sqc", "tcc", "tcp", "ta", "ta2"):
        for p in newest(base + d + "/*/*_counter_collection.csv"):
            for k, counters in pmc(p).items():
                for c, v in counters.items():
                    out.setdefault(k, {})[c] = {"launches": len(v), "mean": sum(v) / len(v), "max": max(v), "sum": sum(v)}
    prov = line.get("provenance") or {}
    steps = line["steps"]
    batches = line["config"]["samples_in_flight"]
    acc = out.get("accumulate", {})
    cal_stream = None
    if "FETCH_SIZE" in acc and len(batches) == 1:
        true_read = (batches[0] * 16.0 + 12.0) * n_pixels  # every accumulate launch of this command carries the same batch
        cal_stream = true_read / (acc["FETCH_SIZE"]["mean"] * 1024.0)
    cal_gather, cal_gather_src = load_gather_factor(here)
    dominant_stage = {"bounce_kernel<middle>": "bounce", "refill_kernel<middle>": "bounce", "bounce_kernel<first>": "bounce_first",
                      "refill_kernel<first>": "bounce_first"}.get(line["roofline"]["kernel"].split(" ")[0], "extend")
    summary = {"kernel": dominant_stage,
               "bench_command": prov.get("command") or ("python3 bench.py " + " ".join(sys.argv[7:]) if len(sys.argv) > 7 else "python3 bench.py (arguments not recorded)"),
               "git_head": prov.get("git_head"), "git_dirty_at_build": prov.get("git_dirty_at_build"), "source_sha256": prov.get("source_sha256"),
               "device": prov.get("device"), "date_utc": prov.get("date_utc"),
               "samples_in_flight": batches[0] if len(batches) == 1 else batches,
               "workload_key": line.get("workload_key"),  # bench.py matches a run to a profile by this (scene and size, frame, spp, bounces, RNG mode, ranks)
               "fetch_calibration": {"streaming_16B_per_lane_on_accumulate": cal_stream, "gather_64B_records": cal_gather,
                                     "gather_source": cal_gather_src},
               "launches": {}}
    # rays per launch of each candidate kernel (for per-ray figures): wavefront 0 / wavefronts 1.., summed over the timed frames
    wr = line.get("wavefront_rays") or []
    frames = max(steps, 1)
    rays_per_launch = {"bounce_first": (wr[0] / frames) if wr else None,
                       "bounce": (sum(wr[1:]) / frames / max(len(wr) - 1, 1)) if len(wr) > 1 else None}
    needles = {"bounce": ("bounce_kernel<1", "refill_kernel<1", "bounce_binned_kernel<1"),
               "bounce_first": ("bounce_kernel<0", "refill_kernel<0", "bounce_binned_kernel<0"), "extend": ("extend_kernel",)}
    gather_shape = scene == "mesh" and "nolds" not in variant and "norefill" not in variant  # the refill traversal: per-lane 4 x 16 B node fetches
    for dom in ("bounce", "bounce_first", "extend"):
        k = out.get(dom)
        if not k:
            continue
        e = {}
        if "FETCH_SIZE" in k and "WRITE_SIZE" in k:
            fetch_kib, write_kib = k["FETCH_SIZE"]["mean"], k["WRITE_SIZE"]["mean"]
            use_gather = gather_shape and dom in ("bounce", "bounce_first") and cal_gather
            cal = cal_gather if use_gather else (cal_stream or 1.0)
            e.update({"launches_profiled": k["FETCH_SIZE"]["launches"], "FETCH_SIZE_KiB_mean": fetch_kib, "WRITE_SIZE_KiB_mean": write_kib,
                      "fetch_calibration_used": cal,
                      "calibration_note": ("read side x the factor measured for per-lane gathers (" + str(cal_gather_src) + ": one 128-byte line request, tallied as 64 bytes, "
                                           "per L2 miss; a 64-byte node is 0.48 of the line its miss moves)"
                                           if use_gather else "read side x the factor measured on accumulate_kernel of this very profile (16 B per lane, streaming)"),
                      "fabric_bytes_per_launch_raw": (fetch_kib + write_kib) * 1024.0,
                      "fabric_bytes_per_launch": (cal * fetch_kib + write_kib) * 1024.0,
                      "fabric_read_bytes_per_launch": cal * fetch_kib * 1024.0, "fabric_write_bytes_per_launch": write_kib * 1024.0})
        avg_ns = None
        if stats:
            tot = cnt = 0.0
            for r in csv.DictReader(open(stats[0])):
                if any(n in r["Name"] for n in needles[dom]):
                    tot += float(r["TotalDurationNs"]); cnt += float(r["Calls"])
            avg_ns = tot / cnt if cnt else None
            e["avg_launch_us_stats_pass"] = avg_ns / 1e3 if avg_ns else None
        if "SQ_ACTIVE_INST_VALU" in k and "GRBM_GUI_ACTIVE" in k:
            lanes = k["SQ_THREAD_CYCLES_VALU"]["mean"] / k["SQ_ACTIVE_INST_VALU"]["mean"]
            sec = {"lanes_per_valu_instruction": round(lanes, 2), "valu_insts_per_launch": k["SQ_INSTS_VALU"]["mean"],
                   "salu_insts_per_launch": k["SQ_INSTS_SALU"]["mean"], "lds_insts_per_launch": k["SQ_INSTS_LDS"]["mean"]}
            if avg_ns:
                # VALU issue. tools/microbench_valu.hip (profiles/r03_microbench_valu.txt), chip-wide wall-clock rates: a SIMD sustains
                # ~0.95-1.0 wave64 instructions per ns of the FULL-RATE class (v_fma / v_mul / v_add / v_and: ~2.3 cycles each at the
                # clock the load leaves) and ~0.58 per ns of the HALF-RATE class (v_min / v_max / v_min3 / v_max3 / v_cndmask / v_cmp /
                # v_lshl_add / v_cvt_f32_ubyte: ~4.1 cycles), 0.30 per ns of v_rcp / v_sqrt. The traversal's inner visit is about half
                # and half (profiles/r03_isa_inner_visit.txt), ~2.9 cycles per instruction: its issue ceiling is ~0.80 per ns per SIMD.
                rate = k["SQ_INSTS_VALU"]["mean"] / 1024.0 / avg_ns  # wave instructions per ns per SIMD
                sec["valu_issue_per_ns_per_simd"] = round(rate, 4)
                sec["valu_issue_ceiling_per_ns_per_simd"] = {"full_rate_ops": 0.95, "half_rate_ops": 0.58, "inner_visit_mix": 0.80}
                sec["valu_issue_frac_of_visit_mix_ceiling"] = round(rate / 0.80, 4)
                sec["shader_clock_ghz"] = round(k["GRBM_GUI_ACTIVE"]["mean"] / 8.0 / avg_ns, 3)
            if "SQ_WAIT_ANY" in k and "SQ_WAVE_CYCLES" in k:
                sec["wait_any_frac"] = round(k["SQ_WAIT_ANY"]["mean"] / k["SQ_WAVE_CYCLES"]["mean"], 4)
                sec["wait_inst_any_frac"] = round(k["SQ_WAIT_INST_ANY"]["mean"] / k["SQ_WAVE_CYCLES"]["mean"], 4)
            if "TA_BUSY_avr" in k:
                # GRBM_GUI_ACTIVE is summed over the 8 XCDs, TA_BUSY_avr / _max are per-TA figures: the L1 address path of config 5
                cyc = k["GRBM_GUI_ACTIVE"]["mean"] / 8.0
                sec["ta_busy_frac"] = {"avr": round(k["TA_BUSY_avr"]["mean"] / cyc, 3), "max": round(k["TA_BUSY_max"]["mean"] / cyc, 3)}
                if "TA_ADDR_STALLED_BY_TC_CYCLES_sum" in k:
                    sec["ta_busy_frac"]["addr_stalled_by_tc"] = round(k["TA_ADDR_STALLED_BY_TC_CYCLES_sum"]["mean"] / 256.0 / cyc, 3)
            if "TCC_HIT_sum" in k:
                sec["l2_hit_rate"] = round(k["TCC_HIT_sum"]["mean"] / max(k["TCC_HIT_sum"]["mean"] + k["TCC_MISS_sum"]["mean"], 1.0), 4)
                req = k.get("TCP_TCC_READ_REQ_sum", {}).get("mean")
                sec["l2_read_requests_per_launch"] = req
                if req and rays_per_launch.get(dom):
                    sec["l2_read_requests_per_ray"] = round(req / rays_per_launch[dom], 2)
            e["secondary"] = sec
        summary["launches"][dom] = e
    summary["note"] = (f"means over every launch of each kernel in `{summary['bench_command']}` ({batches} samples in flight). FETCH_SIZE / WRITE_SIZE count requests "
                       "on the fabric side of L2 (MI355X_MICROARCH.md: Infinity-Cache hits included), so *fabric* bytes, an upper bound on HBM bytes; gfx950 tallies "
                       "a 128-byte read request as 64 bytes, hence the read side is multiplied by a factor calibrated on a known byte count of the same access "
                       "shape. bench.py quotes launches[<dominant>].fabric_bytes_per_launch as roofline.traffic when its own run has the same samples in flight.")
    summary["kernels"] = out
    json.dump(summary, open(os.path.join(here, f"{rnd}_pmc_{scene}_{variant}.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps({kk: vv for kk, vv in summary.items() if kk != "kernels"}, indent=1))
    for name in sorted(out):
        print(name, {c: round(v["mean"], 1) for c, v in sorted(out[name].items())})


if __name__ == "__main__":
    main()
