export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "duo and (many_segments or step_ranges or more_segments)" 2>&1 | grep -v "^Trainable\|amdgpu.ids" | tail -4
timeout 200 python scripts/gpu_perf_probe.py --T 1000 --B 256,512 --variants g4,d4,g8,d8,nola8 --out gpurun_out/r03e_probe.json 2>&1 | grep -v "^Trainable\|amdgpu.ids" | cut -c1-140
timeout 120 python scripts/gpu_duo_profile.py --depth 8 --B 512 --out gpurun_out/r03e_duo_phase_clocks.json 2>&1 | grep -v "^Trainable\|amdgpu.ids"
