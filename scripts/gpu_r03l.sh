export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "duo" 2>&1 | grep -v "^Trainable\|amdgpu.ids" | tail -3
timeout 300 python -m pytest tests/test_gpu_fullsize.py -q -x -k "every_depth and duo" 2>&1 | grep -v "^Trainable\|amdgpu.ids" | tail -3
timeout 200 python scripts/gpu_perf_probe.py --T 1000 --B 256,512 --variants d4,d8,d4 --out gpurun_out/r03l_probe.json 2>&1 | grep -v "^Trainable\|amdgpu.ids" | cut -c1-120
