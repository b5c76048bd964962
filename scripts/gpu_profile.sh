#!/bin/bash
# One GPU-box session: smoke, bench, rocprofv3 kernel-trace stats + PMC passes of the same bench command.
# Usage: scripts/gpu_profile.sh <tag> [extra bench args]
TAG=${1:-r01}; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
BENCH="python $PWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-single --no-config4-leg $*"     # (only the main loop's launches under the profiler)
echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tee $OUT/smoke_$TAG.log | tail -3
echo "== bench"; timeout 600 python bench.py --steps 3 --warmup 1 "$@" 2>&1 | tee $OUT/bench_$TAG.log | tail -2
cd /tmp
echo "== rocprof stats"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_${TAG}_stats -o stats --output-format csv -- $BENCH > $OUT/prof_${TAG}_stats.log 2>&1
echo "== pmc passes"
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout 600 rocprofv3 --kernel-trace --pmc $C -d $OUT/prof_${TAG}_pmc_$N -o pmc --output-format csv -- $BENCH > $OUT/prof_${TAG}_pmc_$N.log 2>&1
  echo "pmc $C rc=$?"
done
cd $OUT && find . -name "*.csv" | head -40
