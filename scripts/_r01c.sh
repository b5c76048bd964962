mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== tests"; timeout 900 python -m pytest tests -m gpu -q -k "many_segments or segment_table or fails_loudly" 2>&1 | tee gpurun_out/parity_r01c.log | tail -8
bash scripts/gpu_profile.sh r01c
