export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "duo and (many_segments or more_segments or step_ranges)" 2>&1 | grep -v "^Trainable\|amdgpu.ids" | tail -3
timeout 200 python scripts/gpu_perf_probe.py --T 1000 --B 256,512 --variants d4l4,d4l1,d4l2,d4l3,d8l7,d8l2,d8l3,d8l4 --out gpurun_out/r03m_probe_lag.json 2>&1 | grep -v "^Trainable\|amdgpu.ids" | cut -c1-120
