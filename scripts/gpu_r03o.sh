export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -s -k "non_shipped or selftest or end_to_end" 2>&1 | grep -v "^Trainable\|amdgpu.ids" | tail -8
