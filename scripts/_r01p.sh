mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== tests"; timeout 900 python -m pytest tests -m gpu -q -k "hoisted or golden or many_segments" 2>&1 | tee gpurun_out/parity_r01p.log | tail -8
echo "== bench"; timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
