#!/bin/bash
# Round 4, session a: the lean wrnn_duo_kernel -- parity of the duo variants, placement, timing against the round-3 library.
TAG=r04a
mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=gpurun_out
echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2 | tee $OUT/${TAG}_smoke.log
echo "== duo parity"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "duo or selftest" 2>&1 | grep -v "^Trainable\|amdgpu.ids" | tail -15 | tee $OUT/${TAG}_parity.log
echo "== placement"; timeout 200 python scripts/gpu_duo_placement.py 2>&1 | tail -12 | tee $OUT/${TAG}_placement.log
echo "== probe new"; timeout 600 python scripts/gpu_perf_probe.py --T 1000 --B 64,128,192,256,512 --variants g1,d1,g2,d2,d2lf,d3,d3lf,d4,d4lf,d4wt,d4lfwt,d6,d6pf,d8,d8pf,d8wt --out $OUT/${TAG}_probe_new.json 2>&1 | grep -v "^Trainable\|amdgpu.ids" | cut -c1-260 | tee $OUT/${TAG}_probe_new.log
echo "== probe old (round-3 library)"; timeout 300 python scripts/gpu_perf_probe.py --T 1000 --B 256,512 --variants d4,d8 --so gpurun_ab/libwavernn_r03.so --out $OUT/${TAG}_probe_r03.json 2>&1 | grep -v "^Trainable\|amdgpu.ids" | cut -c1-260 | tee $OUT/${TAG}_probe_r03.log
echo "== phase clocks"; for D in 4 8; do timeout 200 python scripts/gpu_duo_profile.py --depth $D --B $((D*64)) --T 600 --out $OUT/${TAG}_duo_phase_clocks_depth$D.json 2>&1 | grep -v "^Trainable\|amdgpu.ids" | tail -9; done | tee $OUT/${TAG}_phase.log
