#!/bin/bash
# ONE parametrised GPU-box session (replaces the per-session scripts of rounds 1-4):  scripts/gpu_session.sh <tag> <stage> [<stage> ...]
# Every stage runs under its own `timeout` and logs to gpurun_out/<tag>_<stage>.log.  Stages:
#   tests[:<pytest -k expression>]   the -m gpu suite (or a subset)                  sparse    wrnn_sparse_kernel bring-up: stage-level + parity tests
#   asbench                          the bench's RAW / MoL legs as benchmarked       flips     RAW class-index flip rate vs the C oracle (scripts/gpu_raw_flips.py)
#   probe:<gpu_perf_probe.py args>   loop-kernel timing sweep ('@' for ' ' in args)   smoke     __graft_entry__.smoke()
#   bench[:<bench.py args>]          bench.py --steps 3 --warmup 1                    profile   scripts/gpu_profile.sh <tag> (smoke + bench + rocprofv3 stats + PMC)
#   place                            placement read-out of the duo kernel             sprof[:args] / cprof[:args]  phase clocks of wrnn_sparse_kernel / wrnn_chain_kernel
TAG=${1:-x}; shift
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
F='^Trainable\|amdgpu.ids'
for ST in "$@"; do
  NAME=${ST%%:*}; ARG=""; [ "$ST" != "$NAME" ] && ARG=${ST#*:}
  LOG=gpurun_out/${TAG}_${NAME}.log
  echo "== $ST"
  case $NAME in
    tests)   if [ -n "$ARG" ]; then timeout 2400 python -m pytest tests -m gpu -q --maxfail=8 -s -k "$ARG" 2>&1 | grep -v "$F" | tee $LOG | tail -30
             else timeout 2400 python -m pytest tests -m gpu -q --maxfail=8 -s 2>&1 | grep -v "$F" | tee $LOG | tail -40; fi ;;
    sparse)  timeout 420 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -s -k "block_sparse" 2>&1 | grep -v "$F" | tee $LOG | tail -30 ;;
    asbench) timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -s -k "as_benchmarked or config5" 2>&1 | grep -v "$F" | tee $LOG | tail -20 ;;
    flips)   timeout 900 python scripts/gpu_raw_flips.py --json gpurun_out/${TAG}_raw_flips.json ${ARG//@/ } 2>&1 | grep -v "$F" | tee $LOG | tail -20 ;;
    probe)   timeout 600 python scripts/gpu_perf_probe.py --out gpurun_out/${TAG}_probe.json ${ARG//@/ } 2>&1 | grep -v "$F" | cut -c1-260 | tee -a $LOG | tail -40 ;;
    smoke)   timeout 300 python __graft_entry__.py smoke 2>&1 | grep -v "$F" | tee $LOG | tail -3 ;;
    bench)   timeout 900 python bench.py --steps 3 --warmup 1 ${ARG//@/ } 2>&1 | grep -v "$F" | tee $LOG | tail -1 | cut -c1-1500 ;;
    profile) bash scripts/gpu_profile.sh $TAG ${ARG//@/ } ;;
    sprof)   timeout 240 python scripts/gpu_sparse_profile.py --out gpurun_out/${TAG}_sparse_phase_clocks.json ${ARG//@/ } 2>&1 | grep -v "$F" | tee -a $LOG | tail -8 ;;
    cprof)   timeout 240 python scripts/gpu_chain_profile.py --out gpurun_out/${TAG}_chain_phase_clocks.json ${ARG//@/ } 2>&1 | grep -v "$F" | tee -a $LOG | tail -8 ;;
    place)   timeout 120 python scripts/gpu_duo_placement.py 2>&1 | grep -v "$F" | tee $LOG | tail -12 ;;
    *)       echo "unknown stage $ST" ;;
  esac
done
