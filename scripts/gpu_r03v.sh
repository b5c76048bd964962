export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 300 python scripts/gpu_perf_probe.py --T 1000 --B 16,128,192,256 --variants g1,g1lf,g1x,g1lfx,g2,g2lf,g3,g3pf,g4,g4pf --out gpurun_out/r03v_probe_pubfirst.json 2>&1 | grep -v "^Trainable\|amdgpu.ids" | cut -c1-120
timeout 200 python scripts/gpu_perf_probe.py --mode RAW --T 600 --B 16,128,256 --variants g1,g1lf,g2,g2lf,g4,g4pf --out gpurun_out/r03v_probe_pubfirst_raw.json 2>&1 | grep -v "^Trainable\|amdgpu.ids" | cut -c1-120
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | grep -v "^Trainable\|amdgpu.ids" | tail -5
