#!/bin/bash
# Full GPU-box session: every -m gpu test, then smoke + bench + rocprofv3 stats/PMC profile of the bench command.
TAG=${1:-x}
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== tests"; timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tee gpurun_out/parity_$TAG.log | tail -6
bash scripts/gpu_profile.sh $TAG
