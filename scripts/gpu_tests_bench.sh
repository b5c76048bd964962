#!/bin/bash
# GPU-box session without the profiler: every -m gpu test, smoke, bench.
TAG=${1:-x}
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== tests"; timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tee gpurun_out/parity_$TAG.log | tail -5
echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tee gpurun_out/smoke_$TAG.log | tail -2
echo "== bench"; timeout 600 python bench.py --steps 3 --warmup 1 2>&1 | tee gpurun_out/bench_$TAG.log | tail -1 | cut -c1-400
