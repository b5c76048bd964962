export TMPDIR=/tmp; mkdir -p gpurun_out
OUT=$PWD/gpurun_out
cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQC_INST[A-Z_]*\|SQ_INSTS_SALU\|SQ_WAIT_INST_ANY" | sort -u | tr '\n' ' '; echo
CMD="python $GRAFT_REPO_ROOT/scripts/gpu_probe_one.py --algo duo --depth 4 --B 256 --T 600"
for C in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-30)
  timeout 200 rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc_r03ae_$N -o pmc --output-format csv -- $CMD > $OUT/pmc_r03ae_$N.log 2>&1
  echo "pmc $C rc=$?"
  python - <<PY
import csv, glob, collections
for f in glob.glob("$OUT/pmc_r03ae_$N/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:40]
        if 'duo' in k or 'loop_kernel' in k:
            acc[(k, r['Counter_Name'])] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
    for (k, c), v in sorted(acc.items()): print(f'{k:42s} {c:32s} {v / n[(k, c)]:16.1f} per dispatch ({n[(k, c)]} dispatches)')
PY
done
