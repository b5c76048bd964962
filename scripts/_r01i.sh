mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== probe"; timeout 600 python scripts/gpu_perf_probe.py --variants p2,p2nl8,p3 --B 128,180 --out gpurun_out/probe_r01i.json 2>&1 | grep variant | cut -c1-200
echo "== phases"; timeout 300 python scripts/gpu_phase_profile.py --cases 3:180,2:128 --out gpurun_out/phases_r01i.json > /dev/null 2>&1
