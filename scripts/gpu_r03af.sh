export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q --maxfail=8 -s 2>&1 | grep -v "^Trainable\|amdgpu.ids" | tee gpurun_out/parity_r03zz.log | tail -12 | cut -c1-250
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
