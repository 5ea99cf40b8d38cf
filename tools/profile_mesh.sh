#!/bin/bash
# tools/profile_mesh.sh TAG [bench args]: the PMC passes that matter for the HBM-resident scene (config 5), 8 samples
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
out=gpurun_out
args="--scene mesh --steps 1 --warmup 1 --no-cpu-baseline --no-pixel-anchor $*"   # frames of 64 samples per pixel, all 64 in flight
run() { name=$1; shift
  rm -rf $out/prof_${tag}_$name
  timeout -k 10 240 rocprofv3 "$@" --output-format csv -d $out/prof_${tag}_$name -- python3 bench.py $args > $out/prof_${tag}_$name.json 2> $out/prof_${tag}_$name.err \
    || { echo "pass $name FAILED"; tail -5 $out/prof_${tag}_$name.err; return 1; }
  echo "pass $name ok: $(python3 -c "import json,sys; d=json.load(open('$out/prof_${tag}_$name.json')); print(d['value'], d['roofline']['avg_launch_us'])")"
}
run stats --kernel-trace --stats &&
run fetch --kernel-trace --pmc FETCH_SIZE &&
run write --kernel-trace --pmc WRITE_SIZE &&
run sq --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE &&
run sq2 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS &&
run tcc --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum &&
run tcp --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum &&
run ta --kernel-trace --pmc GRBM_GUI_ACTIVE TA_BUSY_avr TA_BUSY_max &&
run ta2 --kernel-trace --pmc TD_TD_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
find $out/prof_${tag}_* -name "*kernel_trace.csv" -size +20M -delete 2>/dev/null
