export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "(golden or many_segments or fused or teacher or selftest or more_segments or step_ranges) and not duo" 2>&1 | grep -v "^Trainable\|amdgpu.ids" | tail -6
timeout 200 python scripts/gpu_perf_probe.py --T 1000 --B 12,128,256 --variants g1,g1nf,g2,g2nf,g4,g4nf --out gpurun_out/r03i_probe_fused_mol.json 2>&1 | grep -v "^Trainable\|amdgpu.ids" | cut -c1-140
timeout 200 python scripts/gpu_perf_probe.py --mode RAW --T 600 --B 12,128,256 --variants g1,g1nf,g2,g2nf,g2na,g4,g4nf,g4na,g4nfna --out gpurun_out/r03i_probe_fused_raw.json 2>&1 | grep -v "^Trainable\|amdgpu.ids" | cut -c1-140
