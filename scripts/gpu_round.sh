#!/bin/bash
# One GPU-box session: self tests, parity tests, full-size properties, bench.  Logs under gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== rocm-smi" ; rocm-smi --showproductname 2>/dev/null | head -8
echo "== selftests"; timeout 600 python -m pytest tests -m gpu -q -x -s -k "selftests" 2>&1 | tee gpurun_out/selftest.log | tail -15
echo "== parity";    timeout 1500 python -m pytest tests -m gpu -q -k "not full_size and not selftests" 2>&1 | tee gpurun_out/parity.log | tail -40
echo "== fullsize";  timeout 900 python -m pytest tests -m gpu -q -s -k "full_size" 2>&1 | tee gpurun_out/fullsize.log | tail -25
echo "== smoke";     timeout 300 python __graft_entry__.py smoke 2>&1 | tee gpurun_out/smoke.log | tail -5
echo "== bench";     timeout 900 python bench.py --steps 3 --warmup 1 2>&1 | tee gpurun_out/bench.log | tail -5
