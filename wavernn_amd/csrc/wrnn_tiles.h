// wrnn_tiles.h -- the cross-wave partial-tile exchange shared by the persistent loop kernels (wrnn_loop.hip, wrnn_duo.hip, wrnn_chain.hip,
// wrnn_sparse.hip): a wave's accumulator tiles go through LDS in the accumulators' own fragment order.
#pragma once
#include "wrnn_device.h"

namespace wrnn {

constexpr int LDC = 520;            // LDS row stride (floats) of the RAW samplers' [segment][class] logit rows

// partial tile of wave w, slot s, in the accumulator's own fragment order: part[(w*NSLOT + s)*256 + lane*4 + r] = row 4*(lane>>4) + r,
// segment lane & 15.  ONE conflict-free ds_write_b128 per tile; a reader wave whose lane l takes (row 4q + (l & 3), segment
// (l >> 2) & 15) -- the (pu, pj) mapping of the loop / duo kernels, the one publish4's quads need -- reads 64 consecutive words
// (round 2's row-major [16][16] tile made both sides 2-way bank-conflicted: 43 % conflict cycles in profiles/r03k_summary.md).
template <int NSLOT>
__device__ __forceinline__ void put_partial(float *part, int w, int s, int lane, f32x4 acc)
{
    *reinterpret_cast<f32x4 *>(part + (w * NSLOT + s) * 256 + lane * 4) = acc;
}

// fragment row ri (0 .. 16*RT-1) of slot block `base` (0 critical / RT off-path), segment j: sum over the 4 waves
template <int NSLOT>
__device__ __forceinline__ float get_partial(const float *part, int base, int ri, int j)
{
    const int o = (base + (ri >> 4)) * 256 + (((ri & 15) >> 2) * 16 + j) * 4 + (ri & 3);
    float s = part[o];
#pragma unroll
    for (int w = 1; w < NW; ++w) s += part[w * NSLOT * 256 + o];
    return s;
}

}  // namespace wrnn
