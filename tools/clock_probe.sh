#!/bin/bash
# tools/clock_probe.sh: shader clock and package power (rocm-smi) once a second while bench.py renders ~10 s of the headline workload
python bench.py --steps 32768 --warmup 64 --no-cpu-baseline --no-stage-times > gpurun_out/clk_bench.json 2>/dev/null &
BP=$!
for i in $(seq 1 60); do
  kill -0 $BP 2>/dev/null || break
  echo "t=$i $(rocm-smi --showclocks --showpower 2>/dev/null | grep -i 'sclk\|Package Power' | sed 's/.*: //' | tr '\n' ' ')"
  sleep 1
done
wait $BP
python -c "import json; d=json.load(open('gpurun_out/clk_bench.json')); print(d['value'], d['ms_per_step'])"
