#!/bin/bash
# Builds libwavernn_amd.so (HIP kernels + C ABI) for gfx950, in-tree.  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value "$@" \
  wrnn_abi.hip wrnn_cond.hip wrnn_stream.hip wrnn_generic.hip wrnn_loop.hip wrnn_duo.hip wrnn_octo.hip wrnn_chain.hip wrnn_sparse.hip wrnn_pre.hip wrnn_post.hip wrnn_taco.hip wrnn_selftest.hip -o ${WRNN_SO_OUT:-libwavernn_amd.so}
