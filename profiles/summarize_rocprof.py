#!/usr/bin/env python3
"""Condenses rocprofv3 CSV output (gpurun_out/prof_*/<host>/*_{kernel_stats,counter_collection}.csv) into the
small summaries committed under profiles/.

    python profiles/summarize_rocprof.py gpurun_out r01 [n_pixels [samples per accumulate launch, comma separated]]

Expects gpurun_out/prof_{stats,fetch,write,sq,sq2}/<host>/..., as written by
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- python3 bench.py --steps 64 --warmup 32 --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_fetch -- python3 bench.py ...   (one pass per counter group)

Writes profiles/<tag>_kernel_stats.csv (copy of rocprofv3 --kernel-trace --stats), profiles/<tag>_pmc.json
(per-kernel mean counter values per launch) and profiles/<tag>_pmc_extend.json (HBM bytes per extend launch,
read by bench.py for roofline.traffic).

FETCH_SIZE / WRITE_SIZE are in KiB. On gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x
(MI355X_MICROARCH.md "HBM"); other widths are uncalibrated, so the summary calibrates the read side on the
accumulate kernel, whose byte count is known exactly (reads 2 x 12 B/pixel, writes 12 B/pixel, 16 B per lane),
and reports both the raw and the corrected figure.
"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

KERNELS = ["extend_kernel", "shade_kernel", "miss_kernel", "scan_kernel", "generate_rays_kernel", "accumulate_kernel"]


def kname(full):
    for k in KERNELS:
        if k in full:
            return k
    return None


def pmc(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        k = kname(r["Kernel_Name"])
        if k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def main():
    src, tag = sys.argv[1], sys.argv[2]
    n_pixels = int(sys.argv[3]) if len(sys.argv) > 3 else 1920 * 1080
    # samples per accumulate launch, in launch order (an accumulate launch of b samples reads (b + 1) * 12 B/pixel).
    # Default = what `bench.py --steps 64 --warmup 32` does with its 32 samples in flight: warm-up 32, prime 32,
    # 2 x 32 timed, 2 x 32 timed again.
    acc_batches = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [32] * 6
    here = os.path.dirname(os.path.abspath(__file__))
    stats = glob.glob(os.path.join(src, "prof_stats", "*", "*_kernel_stats.csv"))
    if stats:
        shutil.copy(stats[0], os.path.join(here, f"{tag}_kernel_stats.csv"))
    out = {}
    for d in ("prof_fetch", "prof_write", "prof_sq", "prof_sq2"):
        for p in glob.glob(os.path.join(src, d, "*", "*_counter_collection.csv")):
            for k, counters in pmc(p).items():
                for c, v in counters.items():
                    out.setdefault(k, {})[c] = {"launches": len(v), "mean": sum(v) / len(v), "max": max(v), "sum": sum(v)}
    json.dump(out, open(os.path.join(here, f"{tag}_pmc.json"), "w"), indent=1, sort_keys=True)
    ext, acc = out.get("extend_kernel", {}), out.get("accumulate_kernel", {})
    if "FETCH_SIZE" in ext and "WRITE_SIZE" in ext:
        fetch_kib, write_kib = ext["FETCH_SIZE"]["mean"], ext["WRITE_SIZE"]["mean"]
        cal = None
        if "FETCH_SIZE" in acc:
            true_read = sum((b + 1) * 12.0 * n_pixels for b in acc_batches) / len(acc_batches)
            cal = true_read / (acc["FETCH_SIZE"]["mean"] * 1024.0)  # true read bytes / reported
        summary = {
            "kernel": "extend_kernel", "launches": ext["FETCH_SIZE"]["launches"],
            "FETCH_SIZE_KiB_mean": fetch_kib, "WRITE_SIZE_KiB_mean": write_kib,
            "fetch_calibration_on_accumulate": cal,
            "accumulate_WRITE_SIZE_KiB_mean": acc.get("WRITE_SIZE", {}).get("mean"),
            "note": "means over every extend launch of `python3 bench.py --steps 64 --warmup %d --no-cpu-baseline` "
                    "(every launch carries %s samples); FETCH_SIZE x fetch_calibration (gfx950 reports half the read bytes; "
                    % (acc_batches[0], "/".join(str(b) for b in sorted(set(acc_batches)))) +
                    "calibrated on accumulate, whose bytes are known exactly); WRITE_SIZE is exact",
            "hbm_bytes_per_launch_raw": (fetch_kib + write_kib) * 1024.0,
            "hbm_bytes_per_launch": ((cal or 1.0) * fetch_kib + write_kib) * 1024.0,
        }
        json.dump(summary, open(os.path.join(here, f"{tag}_pmc_extend.json"), "w"), indent=1)
        print(json.dumps(summary, indent=1))
    for k in KERNELS:
        if k in out:
            print(k, {c: round(v["mean"], 1) for c, v in sorted(out[k].items())})


if __name__ == "__main__":
    main()
