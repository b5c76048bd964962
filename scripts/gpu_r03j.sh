export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "duo and (many_segments or step_ranges or more_segments or golden)" 2>&1 | grep -v "^Trainable\|amdgpu.ids" | tail -3
for so in libwavernn_amd.so libwavernn_amd_loadsfirst.so; do echo "== $so"; timeout 200 python scripts/gpu_perf_probe.py --so wavernn_amd/csrc/$so --T 1000 --B 256,512 --variants d4,d8,d4 --out gpurun_out/r03j_probe_$so.json 2>&1 | grep -v "^Trainable\|amdgpu.ids" | cut -c1-120; done
