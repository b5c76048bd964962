export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 300 python scripts/gpu_taco_profile.py 2>&1 | grep -v "^Trainable\|amdgpu.ids" | tee gpurun_out/r03s_taco_profile.json | head -60
