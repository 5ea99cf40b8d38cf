#!/bin/bash
# tools/calibrate_fetch.sh ROUND  (on the GPU box, from the repo root; build/node_fetch is cross-compiled in the build container:
#   hipcc --offload-arch=gfx950 -O3 -o build/node_fetch tools/microbench_node_fetch.hip)
# rocprofv3 --pmc FETCH_SIZE over `build/node_fetch calib`, whose kernels read KNOWN byte counts in three access shapes, and the
# factor (known bytes per FETCH_SIZE byte) of each shape -> gpurun_out/<ROUND>_fetch_calibration.json (copy it to profiles/).
rnd=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
out=gpurun_out/fetch_calib
rm -rf $out
timeout -k 10 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out -- ./build/node_fetch calib > $out.json 2> $out.err || { echo FAILED; tail -5 $out.err; exit 1; }
python3 - $out $out.json $rnd <<'PY'
import csv, glob, json, sys, datetime
known = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
fetch = {}
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            name = "gather64" if "gather_known<4>" in r["Kernel_Name"] else "gather128" if "gather_known<8>" in r["Kernel_Name"] else \
                   "stream16" if "stream_known" in r["Kernel_Name"] else None
            if name:
                fetch[name] = float(r["Counter_Value"]) * 1024.0  # KiB
res = {"tool": "tools/microbench_node_fetch.hip calib under rocprofv3 --pmc FETCH_SIZE (tools/calibrate_fetch.sh)",
       "date_utc": datetime.datetime.utcnow().strftime("%Y-%m-%dT%H:%M:%SZ"), "table_bytes": known["table_bytes"], "records_read": known["records_read"]}
for name, key in (("gather64", "gather_64B_record"), ("gather128", "gather_128B_record"), ("stream16", "stream_16B_per_lane")):
    if name in fetch:
        res[key] = {"known_bytes": known[name + "_bytes"], "FETCH_SIZE_bytes": fetch[name], "bytes_per_FETCH_SIZE_byte": known[name + "_bytes"] / fetch[name]}
res["reading"] = ("known_bytes = the bytes the lanes asked for (records x record size; a 1 GiB table, 32 x the L2s, so practically every record is an L2 miss). "
                  "FETCH_SIZE is TCC_EA0_RDREQ x 64 B: if gather64 and gather128 report the same FETCH_SIZE per record, a missing 64-byte record costs one "
                  "128-byte line request tallied as 64 bytes, i.e. the fabric moves 2 x what gather64 asked for.")
json.dump(res, open(f"gpurun_out/{sys.argv[3]}_fetch_calibration.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
