#!/bin/bash
# Compiler view of every kernel in libwavernn_amd.so (hipcc -Rpass-analysis=kernel-resource-usage; runs without a GPU):
# VGPR / AGPR / SGPR counts, spills, scratch, occupancy.  Usage: scripts/kernel_resources.sh > profiles/<tag>_kernel_resources.txt
cd "$(dirname "$0")/../wavernn_amd/csrc"
for f in wrnn_duo.hip wrnn_chain.hip wrnn_loop.hip wrnn_generic.hip wrnn_taco.hip wrnn_sparse.hip wrnn_stream.hip wrnn_cond.hip wrnn_pre.hip wrnn_post.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $f -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
    grep -E "Function Name|SGPRs:|VGPRs:|AGPRs:|ScratchSize|Occupancy|Spill" | sed -e 's/^[^:]*:[0-9]*:[0-9]*: remark: *//' -e 's/ \[-Rpass.*$//' |
    awk '/Function Name/ {if (line) print line; cmd="c++filt " $3; cmd | getline nm; close(cmd); line=nm " |"; next} {line=line " " $0 ";"} END {print line}'
done
