#!/bin/bash
# Full GPU-box session: every -m gpu test, then smoke + bench + rocprofv3 stats/PMC profile of the bench command.
TAG=${1:-x}
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== tests"; timeout 2400 python -m pytest tests -m gpu -q --maxfail=8 -s 2>&1 | grep -v "^Trainable\|amdgpu.ids" | tee gpurun_out/parity_$TAG.log | tail -40
bash scripts/gpu_profile.sh $TAG
