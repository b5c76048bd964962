export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "RAW or raw" 2>&1 | grep -v "^Trainable\|amdgpu.ids" | tail -4
timeout 400 python -m pytest tests/test_gpu_fullsize.py -q -x -k "every_depth and RAW" 2>&1 | grep -v "^Trainable\|amdgpu.ids" | tail -3
timeout 200 python scripts/gpu_perf_probe.py --mode RAW --T 600 --B 128,256,512 --variants g2,g2ns,g4,g4ns,g8,g8ns --out gpurun_out/r03n_probe_raw_solo.json 2>&1 | grep -v "^Trainable\|amdgpu.ids" | cut -c1-140
