#!/bin/bash
# Round 4, session c: A/B builds of on-chain latency trims of wrnn_duo_kernel; then the whole -m gpu suite on the current sources.
TAG=r04c
mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=gpurun_out
for V in base nosleep fc3early hfirst combo; do
  SO=""; [ $V != base ] && SO="--so gpurun_ab/libwrnn_$V.so"
  echo "== probe $V"
  timeout 300 python scripts/gpu_perf_probe.py --T 1500 --B 12,128,256,512 --variants d1,d2,d4,d8 $SO --out $OUT/${TAG}_probe_$V.json 2>&1 | grep -v "^Trainable\|amdgpu.ids" | cut -c1-115
done | tee $OUT/${TAG}_probe.log
echo "== local vs write-through at depth 1, 2"; timeout 300 python scripts/gpu_perf_probe.py --T 1500 --B 12,128 --variants d1,d1wt,d2,d2pf --out $OUT/${TAG}_probe_wt.json 2>&1 | grep -v "^Trainable\|amdgpu.ids" | cut -c1-115 | tee $OUT/${TAG}_probe_wt.log
echo "== all gpu tests"; timeout 2400 python -m pytest tests -m gpu -q --maxfail=10 2>&1 | grep -v "^Trainable\|amdgpu.ids" | tail -25 | tee $OUT/${TAG}_gpu_tests.txt
