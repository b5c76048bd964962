export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "duo" 2>&1 | grep -v "^Trainable\|amdgpu.ids" | tail -5
timeout 300 python scripts/gpu_perf_probe.py --T 1000 --B 192,256,512 --variants d3,d3o,d4,d4o,d6,d6o,d8,d8o --out gpurun_out/r03w_probe_split_fc.json 2>&1 | grep -v "^Trainable\|amdgpu.ids" | cut -c1-120
timeout 500 python -m pytest tests/test_gpu_fullsize.py -q -x -k "every_depth and duo" 2>&1 | grep -v "^Trainable\|amdgpu.ids" | tail -3
