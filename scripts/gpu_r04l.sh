#!/bin/bash
# Round 4, session l: cI formed two steps ahead without sentinel / re-arm (layer 4): parity of the duo kernel, timing, fabric traffic.
TAG=r04l
mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=$PWD/gpurun_out; ROOT=$PWD
echo "== tests"; timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -k "duo or golden or continu or step_range or slab or depth or mel_rows or full_size or bench_workload or corpus or end_to_end" 2>&1 | grep -v "^Trainable\|amdgpu.ids" | tail -12 | tee $OUT/${TAG}_tests.log
echo "== probes"; timeout 300 python scripts/gpu_perf_probe.py --T 2000 --B 12,128,256,512 --variants d1pf,d2pf,d4pf,d8pf --out $OUT/${TAG}_probe.json 2>&1 | grep -v "^Trainable\|amdgpu.ids" | cut -c1-115 | tee $OUT/${TAG}_probe.log
echo "== traffic"
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc_${TAG}_$C -o pmc --output-format csv -- python $ROOT/scripts/gpu_probe_one.py --algo duo --depth 4 --B 256 --T 1600 --reps 2 > $OUT/pmc_${TAG}_$C.log 2>&1
  echo "pmc $C rc=$?"
  python - <<PY
import csv, glob, collections
for f in glob.glob("$OUT/pmc_${TAG}_$C/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:40]
        if 'duo' in k:
            acc[(k, r['Counter_Name'])] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
    for (k, c), v in sorted(acc.items()): print(f'{k:42s} {c:14s} {v / n[(k, c)]:16.1f} per dispatch ({n[(k, c)]} dispatches)')
PY
done 2>&1 | grep -v Segm | tee $OUT/${TAG}_traffic.log
