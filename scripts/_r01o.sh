mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== tests"; timeout 900 python -m pytest tests -m gpu -q -k "post_loop or end_to_end or corpus_equals or config1 or fails_loudly" 2>&1 | tee gpurun_out/parity_r01o.log | tail -12
echo "== bench"; timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
