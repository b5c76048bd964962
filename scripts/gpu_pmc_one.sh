#!/bin/bash
# rocprofv3 PMC passes of ONE loop-kernel configuration: scripts/gpu_pmc_one.sh <tag> <probe args...>
TAG=$1; shift
mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=$PWD/gpurun_out; CMD="python $PWD/scripts/gpu_probe_one.py $*"
cd /tmp
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD" "TA_BUSY_avr TA_BUSY_max TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-30)
  timeout 200 rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc_${TAG}_$N -o pmc --output-format csv -- $CMD > $OUT/pmc_${TAG}_$N.log 2>&1
  echo "pmc $C rc=$?"
  python - <<PY
import csv, glob, collections
for f in glob.glob("$OUT/pmc_${TAG}_$N/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:40]
        if 'duo' in k or 'loop_kernel' in k:
            acc[(k, r['Counter_Name'])] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
    for (k, c), v in sorted(acc.items()): print(f'{k:42s} {c:32s} {v / n[(k, c)]:16.1f} per dispatch ({n[(k, c)]} dispatches)')
PY
done
