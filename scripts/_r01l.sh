mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== tests"; timeout 1200 python -m pytest tests -m gpu -q -k "pipe or block_sparse or config1 or corpus_equals" 2>&1 | tee gpurun_out/parity_r01l.log | tail -6
echo "== probe"; timeout 600 python scripts/gpu_perf_probe.py --variants p2,p3 --B 128,180 --out gpurun_out/probe_r01l.json 2>&1 | grep variant | cut -c1-200
echo "== phases"; timeout 300 python scripts/gpu_phase_profile.py --cases 3:180,2:128 --out gpurun_out/phases_r01l.json > /dev/null 2>&1
