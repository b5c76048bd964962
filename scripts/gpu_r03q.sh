export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 500 python -m pytest tests/test_gpu_config3.py -q -x -s 2>&1 | grep -v "^Trainable\|amdgpu.ids" | grep -i "decoder kernel\|config 3\|passed\|failed\|error\|assert" | cut -c1-400 | head -40
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | grep -v "^Trainable\|amdgpu.ids" | tail -5
timeout 200 python scripts/gpu_perf_probe.py --T 1000 --B 16,256,512 --variants g1,g4,d4,d8 --out gpurun_out/r03q_probe_fragpart.json 2>&1 | grep -v "^Trainable\|amdgpu.ids" | cut -c1-120
bash scripts/gpu_pmc_one.sh r03q --algo duo --depth 4 --B 256 --T 600 2>&1 | grep -i "rc=\|BANK\|IDX_ACTIVE\|MFMA_BUSY\|WAVE_CYCLES" | head -20
