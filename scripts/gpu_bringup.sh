#!/bin/bash
# Bring-up session of the loop kernel on the GPU box: stage-level diagnostic first, then the parity tests that exercise it,
# smoke, a timing sweep.  Usage: scripts/gpu_bringup.sh <tag> [probe args]
TAG=${1:-x}; shift
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== exchange / step-range diagnostics"
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "selftests or exchange_layers or step_ranges or workspace_does" 2>&1 | tail -40 | tee gpurun_out/bringup_diag_$TAG.log
echo "== parity (loop kernel)"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --maxfail=6 -k "loop and not stream and not exchange_layers and not step_ranges and not workspace_does and not full_size and not config4 and not sparse" 2>&1 | tail -30 | tee gpurun_out/bringup_parity_$TAG.log
echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2 | tee gpurun_out/smoke_$TAG.log
echo "== probe"; timeout 600 python scripts/gpu_perf_probe.py --T 1500 --out gpurun_out/probe_$TAG.json "$@" 2>&1 | tail -40
