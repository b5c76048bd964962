#!/bin/bash
# Round 4, session p: hh workgroups run the last slot's gh stage at the top of the next step (DUO_GH_SHIFT): duo parity, timing, then the
# smoke / bench / rocprofv3 session on these sources.
TAG=r04p
mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=$PWD/gpurun_out
echo "== probes"; timeout 200 python scripts/gpu_perf_probe.py --T 2000 --B 128,192,256,512 --variants d2pf,d3pf,d4pf,d8pf --out $OUT/${TAG}_probe.json 2>&1 | grep -v "^Trainable\|amdgpu.ids" | cut -c1-115 | tee $OUT/${TAG}_probe.log
timeout 200 python scripts/gpu_perf_probe.py --mode RAW --T 600 --B 256 --variants d4pf --out $OUT/${TAG}_probe_raw.json 2>&1 | grep -v "^Trainable\|amdgpu.ids" | cut -c1-115 | tee -a $OUT/${TAG}_probe.log
echo "== tests"; timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -k "duo or golden or continu or step_range or slab or depth or mel_rows or full_size or bench_workload or corpus or end_to_end" 2>&1 | grep -v "^Trainable\|amdgpu.ids" | tail -12 | tee $OUT/${TAG}_tests.log
bash scripts/gpu_profile.sh $TAG
