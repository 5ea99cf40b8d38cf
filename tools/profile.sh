#!/bin/bash
# tools/profile.sh TAG [bench.py args...]   (run on the GPU box, from the repo root)
# rocprofv3 passes of ONE bench command, each into gpurun_out/prof_TAG_<pass>/ : kernel trace + stats, then the PMC
# groups in passes of their own (FETCH_SIZE and WRITE_SIZE do not fit one pass; --pmc is never combined with tracing
# domains other than --kernel-trace). profiles/summarize_rocprof.py condenses them into profiles/.
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
out=gpurun_out
args="--steps 2 --warmup 1 --no-cpu-baseline --no-pixel-anchor $*"   # 3 + 2 frames of 64 samples per pixel, all 64 in flight
run() { # name, rocprof options...
  name=$1; shift
  rm -rf $out/prof_${tag}_$name
  timeout -k 10 240 rocprofv3 "$@" --output-format csv -d $out/prof_${tag}_$name -- python3 bench.py $args > $out/prof_${tag}_$name.json 2> $out/prof_${tag}_$name.err \
    || { echo "pass $name FAILED"; tail -5 $out/prof_${tag}_$name.err; return 1; }
  echo "pass $name ok: $(python3 -c "import json,sys; d=json.load(open('$out/prof_${tag}_$name.json')); print(d['value'], d['roofline']['avg_launch_us'])")"
}
run stats --kernel-trace --stats &&
run fetch --kernel-trace --pmc FETCH_SIZE &&
run write --kernel-trace --pmc WRITE_SIZE &&
run sq --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE &&
run sq2 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH &&
run tcc --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum
# round 5: the issue budget of the bounce launches: what the waves' cycles go to besides VALU issue. Each pass may fail on
# a counter the profiler does not know on this box without stopping the others.
run sq3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU
run sq4 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES
run sq5 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_IFETCH SQ_IFETCH_LEVEL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_CYCLES
run sq6 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT
run sqc --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_STALL
# keep what travels back small: the per-dispatch traces are large
find $out/prof_${tag}_* -name "*kernel_trace.csv" -size +20M -delete 2>/dev/null
du -sh $out/prof_${tag}_* 2>/dev/null | tail -8
