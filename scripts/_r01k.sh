mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pre tests"; timeout 900 python -m pytest tests -m gpu -q -k "pre_loop or end_to_end" 2>&1 | tee gpurun_out/parity_r01k.log | tail -25
echo "== bench"; timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400
