export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 500 python -m pytest tests/test_gpu_config3.py -q -x -s 2>&1 | grep -v "^Trainable\|amdgpu.ids" | grep -i "decoder kernel\|config 3\|passed\|failed\|error\|assert" | cut -c1-330 | head -20
timeout 300 python scripts/gpu_taco_profile.py 2>&1 | grep -v "^Trainable\|amdgpu.ids" | tee gpurun_out/r03ac_taco_profile.json | head -12
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | grep -v "^Trainable\|amdgpu.ids" | tail -3
bash scripts/gpu_profile.sh r03zz 2>&1 | grep -v "^\./prof" | tail -14 | cut -c1-400
