export TMPDIR=/tmp; mkdir -p gpurun_out
echo "== tests"; timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 -s 2>&1 | grep -v "^Trainable\|amdgpu.ids" | tee gpurun_out/r03h_gpu_tests.txt | tail -30
echo "== bench"; timeout 400 python bench.py --steps 5 --warmup 2 2>&1 | tee gpurun_out/r03h_bench.log | tail -1 > gpurun_out/r03h_bench_line.json; cut -c1-1500 gpurun_out/r03h_bench_line.json
