#!/bin/bash
# One GPU-box session: self tests, parity tests, full-size properties, timing sweep.  Logs under gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-x}
echo "== selftests"; timeout 300 python -m pytest tests -m gpu -q -x -s -k "selftests" 2>&1 | tee gpurun_out/selftest_$TAG.log | tail -6
echo "== parity";    timeout 1500 python -m pytest tests -m gpu -q -k "not full_size and not selftests" 2>&1 | tee gpurun_out/parity_$TAG.log | tail -30
echo "== fullsize";  timeout 600 python -m pytest tests -m gpu -q -s -k "full_size" 2>&1 | tee gpurun_out/fullsize_$TAG.log | tail -12
echo "== probe";     timeout 900 python scripts/gpu_perf_probe.py --out gpurun_out/probe_$TAG.json 2>&1 | tee gpurun_out/probe_$TAG.log | tail -60
