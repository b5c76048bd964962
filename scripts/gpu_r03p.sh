export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 200 python scripts/gpu_perf_probe.py --T 1000 --B 128,192,256 --variants g2,d2,g3,d3,g4,d4 --out gpurun_out/r03p_probe_min_depth.json 2>&1 | grep -v "^Trainable\|amdgpu.ids" | cut -c1-120
