#!/bin/bash
# tools/ab.sh TAG [bench args...] -- runs bench.py once per library variant (default + build/libwfpt_*.so) on the GPU box
# and prints value / ms_per_step / stage_ms of each. Lines land in gpurun_out/ab_TAG_<variant>.json.
tag=$1; shift
for lib in wavefront_path_tracer_amd/libwfpt.so build/libwfpt_*.so; do
  [ -f "$lib" ] || continue
  name=$(basename $lib .so | sed 's/libwfpt_\?//'); [ -z "$name" ] && name=default
  WFPT_LIB=$PWD/$lib timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline "$@" > gpurun_out/ab_${tag}_${name}.json 2> gpurun_out/ab_${tag}_${name}.err || { echo "$name FAILED"; tail -3 gpurun_out/ab_${tag}_${name}.err; continue; }
  python - "$name" gpurun_out/ab_${tag}_${name}.json <<'PY'
import json,sys
d=json.load(open(sys.argv[2]))
frac = d.get('roofline', {}).get('frac')
print(f"{sys.argv[1]:>14}: {d['value']:9.1f} Mrays/s  {d['ms_per_step']:.4f} ms/step  frac {frac if frac is None else round(frac, 4)}  {d.get('stage_ms')}")
PY
done
