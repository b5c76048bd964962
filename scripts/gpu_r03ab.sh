export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 200 python scripts/gpu_duo_profile.py --depth 4 --B 256 --T 600 --out gpurun_out/r03ab_duo_phase_clocks_depth4.json 2>&1 | grep -v "^Trainable\|amdgpu.ids" | cut -c1-220
timeout 200 python scripts/gpu_duo_profile.py --depth 8 --B 512 --T 600 --out gpurun_out/r03ab_duo_phase_clocks_depth8.json 2>&1 | grep -v "^Trainable\|amdgpu.ids" | cut -c1-220
timeout 200 python scripts/gpu_phase_profile.py --B 16,128 --T 1000 --out gpurun_out/r03ab_loop_phase_clocks_depth1_2.json 2>&1 | grep -v "^Trainable\|amdgpu.ids" | tail -5 | cut -c1-300
