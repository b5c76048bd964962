mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== config4"; timeout 600 python -m pytest tests -m gpu -q -s -k "config4" 2>&1 | tee gpurun_out/config4_r01n.log | tail -6
