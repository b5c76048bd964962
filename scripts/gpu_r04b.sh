#!/bin/bash
# Round 4, session b: placement modes (a layer per XCD / a slot's chain per XCD) x stage order of the lean wrnn_duo_kernel; Tacotron decoder
# kernels against the reference-made golden; bench line with the new planner.
TAG=r04b
mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=gpurun_out
echo "== duo parity + continuation check"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "duo or continuation or step_ranges" 2>&1 | grep -v "^Trainable\|amdgpu.ids" | tail -8 | tee $OUT/${TAG}_parity.log
echo "== config 3 (reference golden)"; timeout 600 python -m pytest tests/test_gpu_config3.py -m gpu -q -x -s -k "decoder_kernel or bigru" 2>&1 | grep -v "^Trainable\|amdgpu.ids" | cut -c1-400 | tail -12 | tee $OUT/${TAG}_config3.log
echo "== probe placements"; timeout 900 python scripts/gpu_perf_probe.py --T 1000 --B 12,64,128,192,256,512 --variants g1,d1p0,d1p1,d1p0lf,d1p1lf,d1p1pf,d2p0,d2p1,d2p0pf,d2p1pf,d3p0,d3p1,d4p0,d4p1,d4p0pf,d4p1pf,d6p0,d6p1,d8p0,d8p1,d8p0pf,d8p1pf --out $OUT/${TAG}_probe.json 2>&1 | grep -v "^Trainable\|amdgpu.ids" | cut -c1-120 | tee $OUT/${TAG}_probe.log
echo "== phase clocks d1 d4"; for D in 1 4; do timeout 200 python scripts/gpu_duo_profile.py --depth $D --B $((D*64)) --T 600 --out $OUT/${TAG}_duo_phase_clocks_depth$D.json 2>&1 | grep -v "^Trainable\|amdgpu.ids" | tail -9; done | tee $OUT/${TAG}_phase.log
echo "== bench"; timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/${TAG}_bench.json | cut -c1-600
