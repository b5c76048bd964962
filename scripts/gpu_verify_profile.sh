#!/bin/bash
# One GPU call: the loop-kernel parity subset (golden, exchange layers, step ranges, ragged / many-segment geometries, full size);
# only if it is green, smoke + bench + rocprofv3 stats + PMC passes (scripts/gpu_profile.sh).  usage: gpu_verify_profile.sh <tag>
TAG=${1:-rXX}
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q --maxfail=4 --durations=8 \
  -k "exchange_layers or step_ranges or (golden and loop and not stream and not g1 and not g3) or (many_segments and loop) or more_segments or (teacher and loop) or config1 or full_size or segment_table or bench_geometry or corpus_slice" \
  2>&1 | grep -v "^Trainable\|amdgpu.ids" | tee gpurun_out/parity_subset_$TAG.log | tail -25
grep -q " passed" gpurun_out/parity_subset_$TAG.log && ! grep -qE "[0-9]+ (failed|error)" gpurun_out/parity_subset_$TAG.log || { echo "PARITY SUBSET NOT GREEN: no profile"; exit 1; }
shift
bash scripts/gpu_profile.sh $TAG "$@"
