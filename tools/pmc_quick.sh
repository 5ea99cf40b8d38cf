#!/bin/bash
# tools/pmc_quick.sh TAG "COUNTERS..." [bench args]: one rocprofv3 --pmc pass of bench.py and a per-kernel mean of each counter
tag=$1; counters=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
out=gpurun_out/pmcq_$tag
rm -rf $out
timeout -k 10 180 rocprofv3 --kernel-trace --pmc $counters --output-format csv -d $out -- python3 bench.py --no-cpu-baseline "$@" > $out.json 2> $out.err || { echo FAILED; tail -5 $out.err; exit 1; }
python3 - $out <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void wfpt::(anonymous namespace)::", "").replace("wfpt::(anonymous namespace)::", "").split("(")[0]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k in sorted(acc, key=lambda k: -sum(dur[k])):
    if sum(dur[k]) < 1e5: continue
    print(f"{k}: {len(dur[k])} launches, mean {sum(dur[k]) / len(dur[k]) / 1e3:.1f} us")
    for c, v in sorted(acc[k].items()):
        print(f"    {c:>40}: mean {sum(v) / len(v):.4g}")
PY
find $out -name "*kernel_trace.csv" -size +20M -delete 2>/dev/null
