#!/usr/bin/env python3
"""tools/issue_budget.py PMC_JSON [MICROBENCH_TXT]: the issue budget of the fused bounce launches from a
committed profile summary (profiles/r05_pmc_shirley_fused.json, written by profiles/summarize_rocprof.py from tools/profile.sh's passes).

Two views of the same launch, for the middle and the first launch side by side:
  (1) where the WAVES' cycles go -- the SQ's own partition: executing an instruction (by unit) | ready but not issued (another wave of the SIMD
      had the slot: issue contention) | waiting at s_waitcnt / s_barrier; the three add up to SQ_WAVE_CYCLES;
  (2) how busy the two issue-limited UNITS are -- the SIMD's vector ALU (instructions of each class x the class's measured cost,
      tools/microbench_valu.hip) and the CU's scalar unit (one instruction per cycle for the CU's four SIMDs).
"""
import json
import sys

# measured cost per wave64 instruction on one SIMD, in ns (profiles/r03_microbench_valu.txt, r05_microbench_salu.txt: chip-wide rates at 8 waves per SIMD)
NS_FULL, NS_HALF, NS_QUARTER, NS_SALU = 1.0 / 0.93, 1.0 / 0.588, 1.0 / 0.295, 1.0 / 0.553


def main():
    d = json.load(open(sys.argv[1]))
    print(f"# {sys.argv[1]}: {d.get('bench_command')}   commit {d.get('git_head')}   {d.get('device')}   {d.get('date_utc')}")
    for name, key in (("middle launches (bounce_kernel<middle>, mean of 7 per frame)", "bounce"), ("first launch (bounce_kernel<first>)", "bounce_first")):
        k = {c: v["mean"] for c, v in d["kernels"][key].items()}
        dur_us = d["launches"][key]["avg_launch_us_stats_pass"]
        wc = k["SQ_WAVE_CYCLES"]
        print(f"\n{name}: {dur_us:.1f} us")
        print("  (1) wave cycles (SQ_WAVE_CYCLES = 100 %)")
        rows = [("executing: vector ALU", k["SQ_ACTIVE_INST_VALU"]), ("executing: scalar ALU / scalar memory", k["SQ_ACTIVE_INST_SCA"]),
                ("executing: LDS", k["SQ_ACTIVE_INST_LDS"]), ("executing: vector memory", k.get("SQ_ACTIVE_INST_FLAT", 0) + k.get("SQ_ACTIVE_INST_VMEM", 0)),
                ("executing: branch / waitcnt / barrier / nop", k["SQ_ACTIVE_INST_MISC"]),
                ("ready, not issued (the SIMD issued another wave)", k["SQ_WAIT_INST_ANY"]),
                ("waiting at s_waitcnt or s_barrier (LDS data, memory, the workgroup's slowest wave)", k["SQ_WAIT_ANY"])]
        tot = 0.0
        for label, v in rows:
            tot += v / wc
            print(f"      {label:86s} {100 * v / wc:5.1f} %")
        print(f"      {'sum':86s} {100 * tot:5.1f} %")
        n_simd = 1024.0
        ns = dur_us * 1e3
        valu = k["SQ_INSTS_VALU"]
        print("  (2) issue-limited units")
        if "SQ_INSTS_VALU_FMA_F32" in k:
            fma, add, mul, trans, i32, i64, cvt = (k.get("SQ_INSTS_VALU_" + c, 0.0) for c in ("FMA_F32", "ADD_F32", "MUL_F32", "TRANS_F32", "INT32", "INT64", "CVT"))
            # full rate: fma / add / mul / the integer add-and-logic half of INT32; half rate: min / max / compare / select / shifts / conversions / the
            # rest; quarter rate: transcendental. The counters do not split INT32 or name min / max / cmp / cndmask: whatever the seven classes leave
            # of SQ_INSTS_VALU is counted half rate, INT32 half and half.
            other = max(valu - (fma + add + mul + trans + i32 + i64 + cvt), 0.0)
            full = fma + add + mul + 0.5 * i32
            half = other + cvt + i64 + 0.5 * i32
            busy = (full * NS_FULL + half * NS_HALF + trans * NS_QUARTER) / n_simd / ns
            print(f"      vector instructions per launch {valu / 1e9:.3f} G: fma {fma / valu:.2f} add {add / valu:.2f} mul {mul / valu:.2f} int32 {i32 / valu:.2f} cvt {cvt / valu:.2f} "
                  f"transcendental {trans / valu:.3f} other (min / max / cmp / select / lane ops) {other / valu:.2f}")
            print(f"      vector ALU busy (classes x measured costs)                         {100 * busy:5.1f} % of the launch")
        rate = valu / n_simd / ns
        print(f"      vector issue {rate:.3f} per ns per SIMD = {100 * rate / 0.753:.0f} % of what the inner visit's VALU mix alone sustains (0.753), "
              f"{100 * rate / 0.611:.0f} % of what it sustained beside round 4's scalar load (0.611)")
        salu = k["SQ_INSTS_SALU"]
        print(f"      scalar instructions per launch {salu / 1e9:.3f} G = {salu / valu:.2f} per vector instruction; scalar unit busy {100 * salu * NS_SALU / n_simd / ns:5.1f} % "
              f"(one unit per CU: 0.553 per ns per SIMD)")
        print(f"      lanes per vector instruction {k['SQ_THREAD_CYCLES_VALU'] / k['SQ_ACTIVE_INST_VALU']:.1f} of 64; LDS bank conflicts {100 * k['SQ_LDS_BANK_CONFLICT'] / max(k['SQ_LDS_IDX_ACTIVE'], 1):.0f} % of the LDS-active cycles; "
              f"instruction-cache misses {k.get('SQC_ICACHE_MISSES', 0):.0f} of {k.get('SQC_ICACHE_REQ', 0) / 1e6:.0f} M fetches")


if __name__ == "__main__":
    main()
