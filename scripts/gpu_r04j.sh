#!/bin/bash
# Round 4, session j: (1) fabric traffic with / without the one-entry pad behind every layer ring (L2 set spread), (2) the last up-sampling
# stage formed inside the loop (wrnn_options.mel_stage): its tests + everything that runs through WaveRNN.generate / generate_corpus,
# (3) timing probes.
TAG=r04j
mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=$PWD/gpurun_out; ROOT=$PWD
echo "== probes"; timeout 300 python scripts/gpu_perf_probe.py --T 2000 --B 12,128,256,512 --variants d1pf,d2pf,d4pf,d8pf --out $OUT/${TAG}_probe_pad.json 2>&1 | grep -v "^Trainable\|amdgpu.ids" | cut -c1-115 | tee $OUT/${TAG}_probe_pad.log
timeout 300 python scripts/gpu_perf_probe.py --so gpurun_ab/libwrnn_nopad.so --T 2000 --B 12,128,256,512 --variants d1pf,d2pf,d4pf,d8pf --out $OUT/${TAG}_probe_nopad.json 2>&1 | grep -v "^Trainable\|amdgpu.ids" | cut -c1-115 | tee $OUT/${TAG}_probe_nopad.log
echo "== traffic"
cd /tmp
for V in pad nopad; do
  SO=""; [ $V = nopad ] && SO="--so $ROOT/gpurun_ab/libwrnn_nopad.so"
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc_${TAG}_${V}_$C -o pmc --output-format csv -- python $ROOT/scripts/gpu_probe_one.py --algo duo --depth 4 --B 256 --T 1600 --reps 2 $SO > $OUT/pmc_${TAG}_${V}_$C.log 2>&1
    echo "pmc $V $C rc=$?"
    python - <<PY
import csv, glob, collections
for f in glob.glob("$OUT/pmc_${TAG}_${V}_$C/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:40]
        if 'duo' in k:
            acc[(k, r['Counter_Name'])] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
    for (k, c), v in sorted(acc.items()): print(f'$V {k:42s} {c:14s} {v / n[(k, c)]:16.1f} per dispatch ({n[(k, c)]} dispatches)')
PY
  done
done 2>&1 | tee $OUT/${TAG}_traffic.log
cd $ROOT
echo "== tests"; timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_config3.py tests/test_gpu_dist.py -m gpu -q -x -k "mel_rows or end_to_end or corpus or full_size or pre_loop or config3 or dist or bench_workload" 2>&1 | grep -v "^Trainable\|amdgpu.ids" | tail -15 | tee $OUT/${TAG}_tests.log
