#!/bin/bash
# Round 4, session o (evidence only, final sources): the bench at 512 segments; phase clocks of the duo kernel at depth 1 / 4 / 8.
TAG=r04o
mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=$PWD/gpurun_out
echo "== bench 512 segments"; timeout 400 python bench.py --utterances 32 --no-single --no-cpu-baseline 2>&1 | tail -1 > $OUT/${TAG}_bench_512seg.json; cut -c1-400 $OUT/${TAG}_bench_512seg.json
for D in 1 4 8; do
  B=$((D * 64))
  echo "== phase clocks depth $D"; timeout 200 python scripts/gpu_duo_profile.py --depth $D --B $B --T 1200 --out $OUT/${TAG}_duo_phase_clocks_depth$D.json 2>&1 | grep -v "^Trainable\|amdgpu.ids" | tail -12
done
