#!/bin/bash
# tools/evidence.sh ROUND  (on the GPU box, from the repo root): every bench line and rocprofv3 pass that DESIGN.md quotes for a round, in one go.
# Lines land in gpurun_out/<ROUND>_line_<name>.json; profiles in gpurun_out/prof_<tag>_*; condense the latter with profiles/summarize_rocprof.py
# and copy the lines to profiles/<ROUND>_bench_line_<name>.json.
rnd=${1:-r05}
line() { name=$1; shift
  timeout -k 10 600 python bench.py "$@" > gpurun_out/${rnd}_line_$name.json 2> gpurun_out/${rnd}_line_$name.err || { echo "$name FAILED"; tail -3 gpurun_out/${rnd}_line_$name.err; return 1; }
  python - gpurun_out/${rnd}_line_$name.json $name <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
cb = d.get("cpu_baseline", {})
print(f"{sys.argv[2]:>16}: {d['value']:10.1f} Mrays/s  {d['ms_per_step']:9.4f} ms/step  frac {d['roofline']['frac']:.4f}  {cb.get('gpu_image_vs_oracle', '')}")
PY
}
line default --gpus 1 --steps 20 --warmup 5 &&
line c1 --width 400 --height 225 --spp 4 --bounces 4 --steps 200 --warmup 20 --cpu-seconds 5 &&
line spp256 --spp 256 --steps 5 --warmup 2 --cpu-seconds 5 &&
line 4k --width 3840 --height 2160 --steps 5 --warmup 2 --cpu-seconds 10 &&
line split --split-shade --steps 10 --warmup 2 --cpu-seconds 5 &&
line unfused --unfused --steps 10 --warmup 2 --cpu-seconds 5 &&
line exact --exact-traversal --steps 10 --warmup 2 --cpu-seconds 5 &&
line pixel --rng-mode pixel --steps 10 --warmup 2 --cpu-seconds 5 &&
line pixel_binned --rng-mode pixel --binning --steps 10 --warmup 2 --no-cpu-baseline &&
line mesh --scene mesh --steps 4 --warmup 1 --cpu-seconds 10 &&
bash tools/profile.sh ${rnd}s && bash tools/profile.sh ${rnd}sp --rng-mode pixel --binning && bash tools/profile_mesh.sh ${rnd}m &&
python tools/scale_probe.py --all-ranks --frames 4 > gpurun_out/${rnd}_scale_probe.txt 2>&1 &&
python tools/scale_probe.py --all-ranks --frames 3 --spp 256 --batch 128 > gpurun_out/${rnd}_scale_probe_c3.txt 2>&1 &&
python tools/scale_probe.py --scene mesh --frames 2 > gpurun_out/${rnd}_scale_probe_mesh.txt 2>&1 &&
if [ -f build/libwfpt_stamps.so ]; then
  WFPT_LIB=$PWD/build/libwfpt_stamps.so python tools/stamps_probe.py > gpurun_out/${rnd}_phase_stamps.txt 2>&1
  WFPT_LIB=$PWD/build/libwfpt_stamps.so python tools/stamps_probe_mesh.py 16 > gpurun_out/${rnd}_phase_stamps_mesh.txt 2>&1
fi
tail -3 gpurun_out/${rnd}_scale_probe*.txt
