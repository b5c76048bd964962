export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 500 python -m pytest tests/test_gpu_config3.py -q -x -s 2>&1 | grep -v "^Trainable\|amdgpu.ids" | grep -i "decoder kernel\|config 3\|passed\|failed\|error\|assert" | cut -c1-330 | head -20
timeout 300 python scripts/gpu_taco_profile.py 2>&1 | grep -v "^Trainable\|amdgpu.ids" | tee gpurun_out/r03ad_taco_profile.json | head -8
