export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "duo and (many_segments or step_ranges)" 2>&1 | grep -v "^Trainable\|amdgpu.ids" | tail -3
timeout 200 python scripts/gpu_perf_probe.py --T 1000 --B 256,512 --variants g4,d4,d8 --out gpurun_out/r03g_probe.json 2>&1 | grep -v "^Trainable\|amdgpu.ids" | cut -c1-140
timeout 120 python scripts/gpu_duo_profile.py --depth 8 --B 512 --out gpurun_out/r03g_duo_phase_clocks.json 2>&1 | grep -v "^Trainable\|amdgpu.ids"
timeout 120 python scripts/gpu_duo_profile.py --depth 4 --B 256 --out gpurun_out/r03g_duo_phase_clocks_d4.json 2>&1 | grep -v "^Trainable\|amdgpu.ids"
