#!/bin/bash
# Incremental build for experiments: one object per source under ${WRNN_OBJ:-/tmp/wrnn_obj}, recompiled when the source (or a header) is newer;
# same flags as build.sh.  usage: build_inc.sh [-DFLAG ...]   (WRNN_SO_OUT=<file> for an A/B library)
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OBJ=${WRNN_OBJ:-/tmp/wrnn_obj}; mkdir -p $OBJ
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value $*"
echo "$FLAGS" > $OBJ/.flags.new
if ! cmp -s $OBJ/.flags.new $OBJ/.flags 2>/dev/null; then rm -f $OBJ/*.o; mv $OBJ/.flags.new $OBJ/.flags; fi
NEWEST_H=$(ls -t *.h ../../include/*.h | head -1)
pids=()
for f in wrnn_abi wrnn_cond wrnn_stream wrnn_generic wrnn_loop wrnn_duo wrnn_octo wrnn_chain wrnn_sparse wrnn_pre wrnn_post wrnn_taco wrnn_selftest; do
  if [ ! -f $OBJ/$f.o ] || [ $f.hip -nt $OBJ/$f.o ] || [ $NEWEST_H -nt $OBJ/$f.o ]; then
    $HIPCC $FLAGS -c $f.hip -o $OBJ/$f.o & pids+=($!)
    if [ ${#pids[@]} -ge 6 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
  fi
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC $OBJ/*.o -o ${WRNN_SO_OUT:-libwavernn_amd.so}
