export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 200 python scripts/gpu_perf_probe.py --T 1000 --B 512 --variants d8,d8p,d8i,d8h,d8pi,d6,d8 --out gpurun_out/r03d_probe_prio.json 2>&1 | grep -v "^Trainable\|amdgpu.ids" | cut -c1-140
bash scripts/gpu_pmc_one.sh r03d --algo duo --depth 8 --B 512 --T 600 2>&1 | tail -40
