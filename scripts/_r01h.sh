mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pipe tests"; timeout 900 python -m pytest tests -m gpu -q -k "pipe" 2>&1 | tee gpurun_out/parity_r01h.log | tail -4
echo "== probe"; timeout 600 python scripts/gpu_perf_probe.py --variants u8,p2,p2nl8,p3,p3nl8,auto --B 128,180,256 --out gpurun_out/probe_r01h.json 2>&1 | grep variant | cut -c1-200
echo "== phases"; timeout 300 python scripts/gpu_phase_profile.py --cases 3:180,2:128 --out gpurun_out/phases_r01h.json 2>&1 | grep -v "^  \"" | tail -40
