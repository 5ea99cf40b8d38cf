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

HBM bytes: FETCH_SIZE / WRITE_SIZE are in KiB. On gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x
(MI355X_MICROARCH.md "HBM"); other widths are uncalibrated, so the read side is calibrated on `accumulate_kernel`, whose
byte count is known exactly (it reads samples x 16 B/pixel of throughput images plus 12 B/pixel of `accumulated`, 16 B per
lane); both raw and corrected figures are reported. WRITE_SIZE is exact for 16-byte-per-lane stores.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

NAMES = [("bounce_kernel<0", "bounce_first"), ("bounce_kernel<1", "bounce"), ("bounce_kernel<2", "bounce_last"),
         ("refill_kernel<0", "bounce_first"), ("refill_kernel<1", "bounce"), ("compact_kernel", "compact"),
         ("extend_kernel", "extend"), ("shade_kernel", "shade"), ("shade_rays_kernel", "shade_rays"), ("miss_kernel", "miss_kernel"), ("scan_kernel", "scan"),
         ("generate_rays_kernel", "generate_rays"), ("accumulate_kernel", "accumulate")]


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
    for d in ("fetch", "write", "sq", "sq2", "tcc", "tcp", "ta", "ta2"):
        for p in newest(base + d + "/*/*_counter_collection.csv"):
            for k, counters in pmc(p).items():
                for c, v in counters.items():
                    out.setdefault(k, {})[c] = {"launches": len(v), "mean": sum(v) / len(v), "max": max(v), "sum": sum(v)}
    dom = "bounce" if "bounce" in out else "extend"
    k, acc = out.get(dom, {}), out.get("accumulate", {})
    summary = {"kernel": dom, "bench_command": "python3 bench.py " + " ".join(sys.argv[7:]) if len(sys.argv) > 7 else None}
    steps = line["steps"]
    batches = line["config"]["samples_in_flight"]
    summary["samples_in_flight"] = batches[0] if len(batches) == 1 else batches
    if "FETCH_SIZE" in k and "WRITE_SIZE" in k:
        fetch_kib, write_kib = k["FETCH_SIZE"]["mean"], k["WRITE_SIZE"]["mean"]
        cal = None
        if "FETCH_SIZE" in acc and len(batches) == 1:
            true_read = (batches[0] * 16.0 + 12.0) * n_pixels  # every accumulate launch of this command carries the same batch
            cal = true_read / (acc["FETCH_SIZE"]["mean"] * 1024.0)
        hbm = ((cal or 1.0) * fetch_kib + write_kib) * 1024.0
        alg = line["roofline"]["algorithmic_bytes_per_launch"]
        summary.update({
            "launches_profiled": k["FETCH_SIZE"]["launches"],
            "FETCH_SIZE_KiB_mean": fetch_kib, "WRITE_SIZE_KiB_mean": write_kib,
            "fetch_calibration_on_accumulate": cal,
            "hbm_bytes_per_launch_raw": (fetch_kib + write_kib) * 1024.0,
            "hbm_bytes_per_launch": hbm,
            "algorithmic_bytes_per_launch": alg,
            "fused_design_bytes_per_launch": line["roofline"].get("fused_design_bytes_per_launch"),
            "hbm_bytes_per_algorithmic_byte": hbm / alg,
            "note": f"means over every launch of the dominant kernel in `bench.py --steps {steps} --warmup {line['warmup']}` "
                    f"({batches} samples in flight); FETCH_SIZE x calibration (gfx950 reports half the bytes of wide reads; "
                    "calibrated on accumulate, whose bytes are known exactly); WRITE_SIZE exact. bench.py quotes "
                    "hbm_bytes_per_launch as roofline.traffic when its own run has the same samples in flight."})
    # average duration of the dominant kernel in the UNPROFILED-counter pass (rocprofv3 --kernel-trace --stats)
    avg_ns = None
    if stats:
        needle = {"bounce": ("bounce_kernel<1", "refill_kernel<1"), "extend": ("extend_kernel",)}[dom]
        tot = cnt = 0.0
        for r in csv.DictReader(open(stats[0])):
            if any(n in r["Name"] for n in needle):
                tot += float(r["TotalDurationNs"]); cnt += float(r["Calls"])
        avg_ns = tot / cnt if cnt else None
        summary["avg_launch_us_stats_pass"] = avg_ns / 1e3 if avg_ns else None
    if "SQ_ACTIVE_INST_VALU" in k and "GRBM_GUI_ACTIVE" in k:
        lanes = k["SQ_THREAD_CYCLES_VALU"]["mean"] / k["SQ_ACTIVE_INST_VALU"]["mean"]
        summary["secondary"] = {"lanes_per_valu_instruction": round(lanes, 2),
                                "valu_insts_per_launch": k["SQ_INSTS_VALU"]["mean"], "salu_insts_per_launch": k["SQ_INSTS_SALU"]["mean"],
                                "lds_insts_per_launch": k["SQ_INSTS_LDS"]["mean"]}
        if avg_ns:
            # VALU issue. tools/microbench_valu.hip (profiles/r03_microbench_valu.txt), chip-wide wall-clock rates: a SIMD sustains
            # ~0.95-1.0 wave64 instructions per ns of the FULL-RATE class (v_fma / v_mul / v_add / v_and: ~2.3 cycles each at the
            # clock the load leaves) and ~0.58 per ns of the HALF-RATE class (v_min / v_max / v_min3 / v_max3 / v_cndmask / v_cmp /
            # v_lshl_add / v_cvt_f32_ubyte: ~4.1 cycles), 0.30 per ns of v_rcp / v_sqrt. The traversal's inner visit is about half
            # and half (profiles/r03_isa_inner_visit.txt), ~2.9 cycles per instruction: its issue ceiling is ~0.80 per ns per SIMD.
            rate = k["SQ_INSTS_VALU"]["mean"] / 1024.0 / avg_ns  # wave instructions per ns per SIMD
            summary["secondary"]["valu_issue_per_ns_per_simd"] = round(rate, 4)
            summary["secondary"]["valu_issue_ceiling_per_ns_per_simd"] = {"full_rate_ops": 0.95, "half_rate_ops": 0.58, "inner_visit_mix": 0.80}
            summary["secondary"]["valu_issue_frac_of_visit_mix_ceiling"] = round(rate / 0.80, 4)
            if "GRBM_GUI_ACTIVE" in k:
                summary["secondary"]["shader_clock_ghz"] = round(k["GRBM_GUI_ACTIVE"]["mean"] / 8.0 / avg_ns, 3)
        if "SQ_WAIT_ANY" in k and "SQ_WAVE_CYCLES" in k:
            summary["secondary"]["wait_any_frac"] = round(k["SQ_WAIT_ANY"]["mean"] / k["SQ_WAVE_CYCLES"]["mean"], 4)
            summary["secondary"]["wait_inst_any_frac"] = round(k["SQ_WAIT_INST_ANY"]["mean"] / k["SQ_WAVE_CYCLES"]["mean"], 4)
    if "TA_BUSY_avr" in k and "GRBM_GUI_ACTIVE" in k:
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs, TA_BUSY_avr / _max are per-TA figures: the L1 address path of config 5
        cyc = k["GRBM_GUI_ACTIVE"]["mean"] / 8.0
        summary["ta_busy_frac"] = {"avr": round(k["TA_BUSY_avr"]["mean"] / cyc, 3), "max": round(k["TA_BUSY_max"]["mean"] / cyc, 3)}
        if "TA_ADDR_STALLED_BY_TC_CYCLES_sum" in k:
            summary["ta_busy_frac"]["addr_stalled_by_tc"] = round(k["TA_ADDR_STALLED_BY_TC_CYCLES_sum"]["mean"] / 256.0 / cyc, 3)
    if "TCC_HIT_sum" in k:
        summary["l2_hit_rate"] = k["TCC_HIT_sum"]["mean"] / max(k["TCC_HIT_sum"]["mean"] + k["TCC_MISS_sum"]["mean"], 1.0)
        summary["l2_read_requests_per_launch"] = k.get("TCP_TCC_READ_REQ_sum", {}).get("mean")
    summary["kernels"] = out
    json.dump(summary, open(os.path.join(here, f"{rnd}_pmc_{scene}_{variant}.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps({kk: vv for kk, vv in summary.items() if kk != "kernels"}, indent=1))
    for name in sorted(out):
        print(name, {c: round(v["mean"], 1) for c, v in sorted(out[name].items())})


if __name__ == "__main__":
    main()
