export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 300 python scripts/gpu_taco_parts.py 2>&1 | grep -v "^Trainable\|amdgpu.ids" | tee gpurun_out/r03t_taco_parts.json | head -40
