mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pipe tests"; timeout 900 python -m pytest tests -m gpu -q -k "pipe" 2>&1 | tee gpurun_out/parity_r01f.log | tail -5
echo "== phases"; timeout 300 python scripts/gpu_phase_profile.py --cases 2:128,3:180 --out gpurun_out/phases_r01f.json 2>&1 | grep -v "^ \"S\|^ \"" | tail -60
