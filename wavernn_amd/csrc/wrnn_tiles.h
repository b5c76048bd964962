// wrnn_tiles.h -- building blocks shared by the clustered persistent loop kernels (wrnn_cluster.hip, wrnn_pipe.hip):
// LDS tile stride, MFMA tile with the A operand in LDS, cross-wave partial-tile reduction, granule sweeps.
#pragma once
#include "wrnn_device.h"

namespace wrnn {

constexpr int LDC = 520;            // LDS row stride (floats) of activation / fc3 tiles in this kernel

// A operand from LDS (fc3): lane reads W[fi][kbase + 16r + 4(l>>4) .. +3] exactly like the activation operand
__device__ __forceinline__ f32x4 mfma_tile_lds(const float *w_lane, const float *act_lane)
{
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < AF / 4; r += 2) {
        const float4 a0 = *reinterpret_cast<const float4 *>(w_lane + 16 * r);
        const float4 a1 = *reinterpret_cast<const float4 *>(w_lane + 16 * (r + 1));
        const float4 b0 = *reinterpret_cast<const float4 *>(act_lane + 16 * r);
        const float4 b1 = *reinterpret_cast<const float4 *>(act_lane + 16 * (r + 1));
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b0.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b1.x, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b0.y, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b1.y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b0.z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b1.z, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b0.w, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b1.w, acc1, 0, 0, 0);
    }
    return acc0 + acc1;
}

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

// the row-major form (part[(w*NSLOT + s)*256 + i*16 + j]) for readers that map a 16-lane row to one unit (wrnn_sparse.hip)
template <int NSLOT>
__device__ __forceinline__ void put_partial_rm(float *part, int w, int s, int lane, f32x4 acc)
{
    float *p = part + (w * NSLOT + s) * 256 + ((lane >> 4) * 4) * 16 + (lane & 15);
    p[0] = acc[0]; p[16] = acc[1]; p[32] = acc[2]; p[48] = acc[3];
}
template <int NSLOT>
__device__ __forceinline__ float get_partial_rm(const float *part, int base, int ri, int j)
{
    const int o = (base + (ri >> 4)) * 256 + (ri & 15) * 16 + j;
    float s = part[o];
#pragma unroll
    for (int w = 1; w < NW; ++w) s += part[w * NSLOT * 256 + o];
    return s;
}

// Sweep one layer of this cluster's granules (byte offset `soff` into the buffer resource) until every tag
// matches; thread (r = tid>>4, c = tid&15) owns row r (segment), column pairs own_col(i, c).  Writes the values
// to dst[r][..] and, if ADD, adds them onto acc[r][..] (the residual adds of fatchord_version.py:212,216).
// NL = loads in flight per thread: 16 = the whole row slice at once (64 registers); 8 = two dependent half
// sweeps per pass (32 registers; used where the register file is full of weights, U = 8).
template <bool ADD, int NL>
__device__ __forceinline__ bool sweep_layer(__amdgpu_buffer_rsrc_t rs, int soff, unsigned tag, int nb, int tid,
                                            float *dst, float *acc, unsigned *status)
{
    const int r = tid >> 4, c = tid & 15;
    if (r >= nb) return true;
    const int voff = r * (H * 8) + c * 16;
    unsigned spins = 0;
    if (NL == 16) {
        u32x4 x[16];
        for (;;) {
#pragma unroll
            for (int i = 0; i < 16; ++i) x[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + i * 256, soff, 16 /* sc1 */);
            bool ok = true;
#pragma unroll
            for (int i = 0; i < 16; ++i) ok &= (x[i].y == tag) & (x[i].w == tag);
            if (ok) break;
            ++spins;
            if ((spins & 255u) == 0u) {
                if (spins > SPIN_LIMIT || ld_agent32(status) != 0u) return false;
            }
            __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float2 v = make_float2(__uint_as_float(x[i].x), __uint_as_float(x[i].z));
            *reinterpret_cast<float2 *>(dst + r * LDC + own_col(i, c)) = v;
            if (ADD) {
                float2 s = *reinterpret_cast<float2 *>(acc + r * LDC + own_col(i, c));
                s.x += v.x; s.y += v.y;
                *reinterpret_cast<float2 *>(acc + r * LDC + own_col(i, c)) = s;
            }
        }
    } else {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            u32x4 x[8];
            for (;;) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    x[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + (8 * half + i) * 256, soff, 16 /* sc1 */);
                bool ok = true;
#pragma unroll
                for (int i = 0; i < 8; ++i) ok &= (x[i].y == tag) & (x[i].w == tag);
                if (ok) break;
                ++spins;
                if ((spins & 255u) == 0u) {
                    if (spins > SPIN_LIMIT || ld_agent32(status) != 0u) return false;
                }
                __builtin_amdgcn_s_sleep(1);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int col = own_col(8 * half + i, c);
                const float2 v = make_float2(__uint_as_float(x[i].x), __uint_as_float(x[i].z));
                *reinterpret_cast<float2 *>(dst + r * LDC + col) = v;
                if (ADD) {
                    float2 s = *reinterpret_cast<float2 *>(acc + r * LDC + col);
                    s.x += v.x; s.y += v.y;
                    *reinterpret_cast<float2 *>(acc + r * LDC + col) = s;
                }
            }
        }
    }
    return true;
}


// Two row tiles that share the activation operand (the two 16-row tiles of one GRU matrix): B fragments are read
// from LDS once, four independent accumulator chains keep the matrix pipe busy.  Per tile the accumulation order is
// exactly mfma_tile's (even r -> chain 0, odd r -> chain 1, then chain 0 + chain 1), so results are bit-identical.
__device__ __forceinline__ void mfma_tile2(const float (&a0)[AF], const float (&a1)[AF], const float *act_lane,
                                           f32x4 &o0, f32x4 &o1)
{
    // all 8 B fragments up front (32 registers, free during the MFMA phase): 64 MFMAs then issue back to back
    float4 b[AF / 4];
#pragma unroll
    for (int r = 0; r < AF / 4; ++r) b[r] = *reinterpret_cast<const float4 *>(act_lane + 16 * r);
    __builtin_amdgcn_sched_barrier(0);      // keep the 8 ds_read_b128 ahead of the MFMA stream (hipcc sinks them otherwise)
    f32x4 c00 = {0.f, 0.f, 0.f, 0.f}, c01 = {0.f, 0.f, 0.f, 0.f}, c10 = {0.f, 0.f, 0.f, 0.f}, c11 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < AF / 4; r += 2) {
        const float4 b0 = b[r], b1 = b[r + 1];
        c00 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[4 * r + 0], b0.x, c00, 0, 0, 0);
        c10 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[4 * r + 0], b0.x, c10, 0, 0, 0);
        c01 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[4 * r + 4], b1.x, c01, 0, 0, 0);
        c11 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[4 * r + 4], b1.x, c11, 0, 0, 0);
        c00 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[4 * r + 1], b0.y, c00, 0, 0, 0);
        c10 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[4 * r + 1], b0.y, c10, 0, 0, 0);
        c01 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[4 * r + 5], b1.y, c01, 0, 0, 0);
        c11 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[4 * r + 5], b1.y, c11, 0, 0, 0);
        c00 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[4 * r + 2], b0.z, c00, 0, 0, 0);
        c10 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[4 * r + 2], b0.z, c10, 0, 0, 0);
        c01 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[4 * r + 6], b1.z, c01, 0, 0, 0);
        c11 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[4 * r + 6], b1.z, c11, 0, 0, 0);
        c00 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[4 * r + 3], b0.w, c00, 0, 0, 0);
        c10 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[4 * r + 3], b0.w, c10, 0, 0, 0);
        c01 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[4 * r + 7], b1.w, c01, 0, 0, 0);
        c11 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[4 * r + 7], b1.w, c11, 0, 0, 0);
    }
    o0 = c00 + c01;
    o1 = c10 + c11;
}

// mfma_tile with the 8 B fragments loaded up front (same accumulation order, bit-identical result)
__device__ __forceinline__ f32x4 mfma_tile_pre(const float (&a)[AF], const float *act_lane)
{
    float4 b[AF / 4];
#pragma unroll
    for (int r = 0; r < AF / 4; ++r) b[r] = *reinterpret_cast<const float4 *>(act_lane + 16 * r);
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < AF / 4; r += 2) {
        const float4 b0 = b[r], b1 = b[r + 1];
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * r + 0], b0.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * r + 4], b1.x, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * r + 1], b0.y, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * r + 5], b1.y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * r + 2], b0.z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * r + 6], b1.z, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * r + 3], b0.w, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * r + 7], b1.w, acc1, 0, 0, 0);
    }
    return acc0 + acc1;
}

// Streaming sweep: like sweep_layer<ADD, 8> but ONE pass keeps 8 loads in flight continuously -- slot i is consumed
// (tag check + LDS write) and immediately reloaded with piece i+8 -- instead of two dependent half sweeps.  Values are
// written to dst unconditionally and the pass repeats until every tag matched; the residual add runs afterwards from
// dst (the thread re-reads its own LDS writes, in order).
template <bool ADD>
__device__ __forceinline__ bool sweep_stream(__amdgpu_buffer_rsrc_t rs, int soff, unsigned tag, int nb, int tid,
                                             float *dst, float *acc, unsigned *status)
{
    const int r = tid >> 4, c = tid & 15;
    if (r >= nb) return true;
    const int voff = r * (H * 8) + c * 16;
    unsigned spins = 0;
    for (;;) {
        u32x4 x[8];
        bool ok = true;
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + i * 256, soff, 16 /* sc1 */);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            ok &= (x[i].y == tag) & (x[i].w == tag);
            *reinterpret_cast<float2 *>(dst + r * LDC + own_col(i, c)) = make_float2(__uint_as_float(x[i].x), __uint_as_float(x[i].z));
            x[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + (8 + i) * 256, soff, 16 /* sc1 */);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            ok &= (x[i].y == tag) & (x[i].w == tag);
            *reinterpret_cast<float2 *>(dst + r * LDC + own_col(8 + i, c)) = make_float2(__uint_as_float(x[i].x), __uint_as_float(x[i].z));
        }
        if (ok) break;
        ++spins;
        if ((spins & 255u) == 0u) {
            if (spins > SPIN_LIMIT || ld_agent32(status) != 0u) return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    if (ADD) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float2 v = *reinterpret_cast<const float2 *>(dst + r * LDC + own_col(i, c));
            float2 s = *reinterpret_cast<float2 *>(acc + r * LDC + own_col(i, c));
            s.x += v.x; s.y += v.y;
            *reinterpret_cast<float2 *>(acc + r * LDC + own_col(i, c)) = s;
        }
    }
    return true;
}


// Half-row sweep: the 8 pieces [8*half, 8*half+8) of every row (k in [256*half, 256*half+256)), one pass of 8 loads.
__device__ __forceinline__ bool sweep_half(__amdgpu_buffer_rsrc_t rs, int soff, unsigned tag, int nb, int tid, float *dst,
                                           int half, unsigned *status)
{
    const int r = tid >> 4, c = tid & 15;
    if (r >= nb) return true;
    const int voff = r * (H * 8) + c * 16 + half * 8 * 256;
    unsigned spins = 0;
    u32x4 x[8];
    for (;;) {
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + i * 256, soff, 16 /* sc1 */);
        bool ok = true;
#pragma unroll
        for (int i = 0; i < 8; ++i) ok &= (x[i].y == tag) & (x[i].w == tag);
        if (ok) break;
        ++spins;
        if ((spins & 255u) == 0u) {
            if (spins > SPIN_LIMIT || ld_agent32(status) != 0u) return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
        *reinterpret_cast<float2 *>(dst + r * LDC + own_col(8 * half + i, c)) = make_float2(__uint_as_float(x[i].x), __uint_as_float(x[i].z));
    return true;
}

}  // namespace wrnn
