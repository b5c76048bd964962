mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pipe tests"; timeout 900 python -m pytest tests -m gpu -q -k "pipe" 2>&1 | tee gpurun_out/parity_r01g.log | tail -4
echo "== probe"; timeout 600 python scripts/gpu_perf_probe.py --variants u8,p2,p2nl16,p3,p3nl16 --B 128,180,256,360 --out gpurun_out/probe_r01g.json 2>&1 | grep variant | cut -c1-200
echo "== phases"; timeout 300 python scripts/gpu_phase_profile.py --cases 3:180 --out gpurun_out/phases_r01g.json 2>&1 | tail -32
