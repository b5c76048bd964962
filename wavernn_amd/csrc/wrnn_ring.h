// wrnn_ring.h -- device building blocks shared by the persistent loop kernels that exchange activations through the tag-free
// sentinel ring in MFMA-fragment order (wrnn_loop.hip: one workgroup per CU, two roles; wrnn_duo.hip: two workgroups per CU, four
// roles): fragment-order addressing, the issue / finish halves of a polled layer, publish / re-arm stores, register-operand MFMA
// tiles, the conditioning-slab loads.  See wrnn_loop.hip for the protocol.
#pragma once
#include "wrnn_tiles.h"

namespace wrnn {

constexpr int LU = 16;                       // hidden units per workgroup
constexpr int LNJ = H / LU;                  // workgroups per role per cluster (32)
constexpr int LNWGC = 2 * LNJ;               // workgroups per cluster (64)
constexpr int XT = SEG * H;                  // floats of one layer in fragment order (8192)
constexpr unsigned SENT = 0xFFFFFFFFu;       // "not written yet"

// LDS carve (floats).  Per group: GH[3][256], HOWN[256], XS[16], SP[32 ints] (segment table), FR[2][16 ints] (conditioning frame
// of every segment at step t in FR[t & 1]; the other half is filled for step t+1 during step t)
//   XO[256]: the owned 16 units x 16 segments of the residual input of this workgroup's GRU (role A: xi, role B: x1), in publish order
constexpr int LGRP = 3 * 256 + 256 + 16 + 32 + 32 + 256;
constexpr int O_HOWN = 768, O_XS = 1024, O_SP = 1040, O_FR = 1072, O_XO = 1104;
// workgroup barrier that hands over LDS only: waits for this wave's LDS traffic (lgkmcnt), not -- as __syncthreads() does through
// its fences -- for every global load and store it has in flight (vmcnt)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// fragment-order offset (floats) of (wave w, k-block r, lane): 4 consecutive k of one segment
__device__ __forceinline__ int frag_off(int w, int r, int lane) { return ((w * 8 + r) * 64 + lane) * 4; }

// This wave's 8 B fragments of one exchanged layer (byte offset soff in the exchange buffer), as two halves so a stage can put
// work between them: issue() fires the 8 buffer_load_dwordx4 (sc1); finish() checks that no word is still the sentinel and,
// only if one is, falls into the polling loop (re-load, bounded spin).  Lanes of segments >= nb are not waited for (their
// columns are garbage, used by nothing).  Wave-uniform result.
__device__ __forceinline__ void issue(__amdgpu_buffer_rsrc_t rs, int soff, int w, int lane, u32x4 (&x)[8])
{
    const int voff = frag_off(w, 0, lane) * 4;
#pragma unroll
    for (int r = 0; r < 8; ++r) x[r] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + r * 1024, soff, 16 /* sc1 */);
}
// issue() without a branch: with on == false the eight loads go to an out-of-range offset (they return zeros, nobody reads them)
__device__ __forceinline__ void issue_sel(__amdgpu_buffer_rsrc_t rs, int soff, int w, int lane, u32x4 (&x)[8], bool on)
{
    const int voff = frag_off(w, 0, lane) * 4 + (on ? 0 : 0x7FFF0000);
#pragma unroll
    for (int r = 0; r < 8; ++r) x[r] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + r * 1024, soff, 16 /* sc1 */);
}
// non-blocking form: true (and b filled) if the fragments already in x carry no sentinel
__device__ __forceinline__ bool try_finish(int lane, int nb, const u32x4 (&x)[8], float (&b)[32])
{
    const bool live = (lane & 15) < nb;
    unsigned m = 0u;
#pragma unroll
    for (int r = 0; r < 8; ++r) m = max(max(m, max(x[r].x, x[r].y)), max(x[r].z, x[r].w));
    if (!__all(m != SENT || !live)) return false;
    // columns of absent segments (ragged last group) keep whatever the buffer holds -- the sentinel, a NaN: an MFMA column, the
    // partial sums and the pointwise math are all per segment, nothing of an absent segment is ever published or written out
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        b[4 * r + 0] = __uint_as_float(x[r].x);
        b[4 * r + 1] = __uint_as_float(x[r].y);
        b[4 * r + 2] = __uint_as_float(x[r].z);
        b[4 * r + 3] = __uint_as_float(x[r].w);
    }
    return true;
}
__device__ __forceinline__ bool finish(__amdgpu_buffer_rsrc_t rs, int soff, int w, int lane, int nb, u32x4 (&x)[8], float (&b)[32],
                                       unsigned *status, unsigned &spins)
{
    const int voff = frag_off(w, 0, lane) * 4;
    const bool live = (lane & 15) < nb;
    spins = 0;
    for (;;) {
        // the sentinel is the largest unsigned value: ONE compare of the running maximum of the 32 words (a chain of v_max3_u32
        // in the vector unit) instead of 32 compare / scalar-and pairs, which serialise on the VALU -> SALU hand-off
        unsigned m = 0u;
#pragma unroll
        for (int r = 0; r < 8; ++r) m = max(max(m, max(x[r].x, x[r].y)), max(x[r].z, x[r].w));
        if (__all(m != SENT || !live)) break;
        ++spins;
        if ((spins & 255u) == 0u) {
            if (spins > SPIN_LIMIT || ld_agent32(status) != 0u) return false;
        }
        __builtin_amdgcn_s_sleep(1);
#pragma unroll
        for (int r = 0; r < 8; ++r) x[r] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + r * 1024, soff, 16 /* sc1 */);
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        b[4 * r + 0] = __uint_as_float(x[r].x);
        b[4 * r + 1] = __uint_as_float(x[r].y);
        b[4 * r + 2] = __uint_as_float(x[r].z);
        b[4 * r + 3] = __uint_as_float(x[r].w);
    }
    return true;
}

// One value per thread (unit u = 4 (tid >> 6) + (tid & 3), segment j = (tid >> 2) & 15) -> the workgroup's 1 KB block of a
// layer: the four units of a quad are gathered with DPP and stored by the quad's first lane as ONE 16-byte sc1 store.
// `soff` = byte offset of the block in the exchange buffer; `on` = this quad publishes (segment live, rows owned).
__device__ __forceinline__ void publish4(__amdgpu_buffer_rsrc_t rs, int soff /* bytes: layer + block */, int tid, float v, bool on)
{
    const int iv = __builtin_bit_cast(int, v);
    const int v0 = __builtin_amdgcn_update_dpp(0, iv, 0x00, 0xF, 0xF, true);     // quad_perm [0,0,0,0]
    const int v1 = __builtin_amdgcn_update_dpp(0, iv, 0x55, 0xF, 0xF, true);     // [1,1,1,1]
    const int v2 = __builtin_amdgcn_update_dpp(0, iv, 0xAA, 0xF, 0xF, true);     // [2,2,2,2]
    const int v3 = __builtin_amdgcn_update_dpp(0, iv, 0xFF, 0xF, 0xF, true);     // [3,3,3,3]
    if (on && (tid & 3) == 0) {
        const u32x4 q = {(unsigned)v0, (unsigned)v1, (unsigned)v2, (unsigned)v3};
        __builtin_amdgcn_raw_buffer_store_b128(q, rs, (tid & ~3) * 4, soff, 16 /* sc1 */);
    }
}

// publish4 without a branch: quads that do not publish store to an out-of-range buffer offset (the buffer unit drops it), so the
// store can sit inside one scheduling region with the MFMA tiles of the next stage (fused stages)
__device__ __forceinline__ void publish4_nb(__amdgpu_buffer_rsrc_t rs, int soff, int tid, float v, bool on)
{
    const int iv = __builtin_bit_cast(int, v);
    const int v0 = __builtin_amdgcn_update_dpp(0, iv, 0x00, 0xF, 0xF, true);
    const int v1 = __builtin_amdgcn_update_dpp(0, iv, 0x55, 0xF, 0xF, true);
    const int v2 = __builtin_amdgcn_update_dpp(0, iv, 0xAA, 0xF, 0xF, true);
    const int v3 = __builtin_amdgcn_update_dpp(0, iv, 0xFF, 0xF, 0xF, true);
    const u32x4 q = {(unsigned)v0, (unsigned)v1, (unsigned)v2, (unsigned)v3};
    const int voff = (on && (tid & 3) == 0) ? (tid & ~3) * 4 : 0x7FFFFFF0;      // past num_records: dropped
    __builtin_amdgcn_raw_buffer_store_b128(q, rs, voff, soff, 16 /* sc1 */);
}

// 1 MFMA : NV VALU, N times (the fused stages' interleave: the wave issues in order, so the pointwise instructions have to sit
// BETWEEN the MFMAs -- up to 7 fit in the 32 cycles one 16x16x4 f32 MFMA occupies the pipe)
template <int N, int NV>
__device__ __forceinline__ void interleave_mfma_valu()
{
#pragma unroll
    for (int q = 0; q < N; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
    }
}

// Re-arm (fill with the sentinel) THIS WAVE's quarter (64 floats = 16 lanes x 16 bytes) of the workgroup's block in up to
// four layers of a ring slot: lanes 0-15 layer la, 16-31 lb, 32-47 lc, 48-63 ld (< 0: none).  Each wave re-arms exactly the
// words it later publishes, so its own program order + the vmcnt(0) drain at the end of the next step order re-arm before data.
__device__ __forceinline__ void rearm(__amdgpu_buffer_rsrc_t rs, int soff_slot0 /* bytes: layer 0 of the slot + block + quarter */,
                                      int lane, int la, int lb, int lc, int ld)
{
    const int which = lane >> 4;
    const int layer = which == 0 ? la : (which == 1 ? lb : (which == 2 ? lc : ld));
    if (layer >= 0) {
        const u32x4 q = {SENT, SENT, SENT, SENT};
        __builtin_amdgcn_raw_buffer_store_b128(q, rs, layer * (XRING * XT * 4) + (lane & 15) * 16, soff_slot0, 16 /* sc1 */);
    }
}

// MFMA tiles with the B fragments already in registers; per tile the accumulation order is mfma_tile's (even r -> chain 0,
// odd r -> chain 1, chain 0 + chain 1), so results are bit-identical to the LDS-operand forms of wrnn_tiles.h.
__device__ __forceinline__ void mfma3(const float (&a0)[AF], const float (&a1)[AF], const float (&a2)[AF], const float (&b)[32],
                                      f32x4 &o0, f32x4 &o1, f32x4 &o2)
{
    f32x4 c00 = {0.f, 0.f, 0.f, 0.f}, c01 = c00, c10 = c00, c11 = c00, c20 = c00, c21 = c00;
#pragma unroll
    for (int r = 0; r < 8; r += 2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            c00 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[4 * r + e], b[4 * r + e], c00, 0, 0, 0);
            c10 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[4 * r + e], b[4 * r + e], c10, 0, 0, 0);
            c20 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[4 * r + e], b[4 * r + e], c20, 0, 0, 0);
            c01 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[4 * r + 4 + e], b[4 * r + 4 + e], c01, 0, 0, 0);
            c11 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[4 * r + 4 + e], b[4 * r + 4 + e], c11, 0, 0, 0);
            c21 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[4 * r + 4 + e], b[4 * r + 4 + e], c21, 0, 0, 0);
        }
    }
    o0 = c00 + c01;
    o1 = c10 + c11;
    o2 = c20 + c21;
}
__device__ __forceinline__ f32x4 mfma1(const float (&a)[AF], const float (&b)[32])
{
    f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0;
#pragma unroll
    for (int r = 0; r < 8; r += 2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * r + e], b[4 * r + e], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * r + 4 + e], b[4 * r + 4 + e], c1, 0, 0, 0);
        }
    }
    return c0 + c1;
}

// one fc3 tile with the A operand in LDS (fragment order, this wave's slice at `a_lane`), B in registers; mfma_tile's order
__device__ __forceinline__ f32x4 mfma1_lds(const float *a_lane /* F3 tile + frag_off(w, 0, lane) */, const float (&b)[32])
{
    f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0;
#pragma unroll
    for (int r = 0; r < 8; r += 2) {
        const float4 a0 = *reinterpret_cast<const float4 *>(a_lane + 256 * r);
        const float4 a1 = *reinterpret_cast<const float4 *>(a_lane + 256 * (r + 1));
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b[4 * r + 0], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b[4 * r + 4], c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b[4 * r + 1], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b[4 * r + 5], c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b[4 * r + 2], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b[4 * r + 6], c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b[4 * r + 3], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b[4 * r + 7], c1, 0, 0, 0);
    }
    return c0 + c1;
}

// this wave's 8 conditioning fragments cI(t) of one group (plain loads: written by the previous kernel on the stream)
__device__ __forceinline__ void load_cI(const float *cI_grp, int w, int lane, float4 (&c)[8])
{
    const float4 *cp = reinterpret_cast<const float4 *>(cI_grp + frag_off(w, 0, lane));
#pragma unroll
    for (int r = 0; r < 8; ++r) c[r] = cp[r * 64];
}

// xi(t) = W_I[:,0] * x_{t-1} + cI(t)   (fatchord_version.py:208-209 with the conditioning part hoisted), this wave's fragments
__device__ __forceinline__ void make_xi(const float4 (&c)[8], const float *WI0, float xs, int w, int lane, float (&b)[32])
{
    const int k0 = KCH * w + 4 * (lane >> 4);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const float4 wv = *reinterpret_cast<const float4 *>(WI0 + k0 + 16 * r);
        b[4 * r + 0] = fmaf(wv.x, xs, c[r].x);
        b[4 * r + 1] = fmaf(wv.y, xs, c[r].y);
        b[4 * r + 2] = fmaf(wv.z, xs, c[r].z);
        b[4 * r + 3] = fmaf(wv.w, xs, c[r].w);
    }
}

// ---- in-loop conditioning: cI(t) = b_I + W_I[:, 1:] . [m_t ; a1_t] (fatchord_version.py:203-209 without the x_{t-1}
// column) for the 16 rows of unit block J and the 16 segments of a group, as ONE wave's 28 MFMAs (K = 112, no split, no LDS, no barrier)
// straight from the up-sampled mel row and the frame's aux row.  K is ordered so that lane (row / segment fi, k-quad kq) owns the 28
// CONSECUTIVE inputs [28 kq, 28 kq + 28): seven 16-byte loads per lane.  The accumulator layout (rows 4 (lane >> 4) + i, segment
// lane & 15) IS the consumers' fragment order of the block: one 16-byte store per lane at block + 16 lane.
constexpr int CK = KCOND / 4;                // inputs per lane (28)
struct CondTile {
    float w[CK];                             // A fragments: W_I[16 J + fi][1 + 28 kq + kk]
    float bias[4];                           // b_I[16 J + 4 (lane >> 4) + i]
};
__device__ __forceinline__ void cond_tile_init(CondTile &c, const float *I_cT /* [KCOND][H] */, const float *I_b, int J, int lane)
{
    const int fi = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int kk = 0; kk < CK; ++kk) c.w[kk] = I_cT[(size_t)(CK * kq + kk) * H + LU * J + fi];
#pragma unroll
    for (int i = 0; i < 4; ++i) c.bias[i] = I_b[LU * J + 4 * kq + i];
}
// mel_row = &mels_up[p][0] (80 floats), aux_row = &aux[frame][0] (a1 = its first 32 floats) of THIS lane's segment; valid == false (the
// fold's zero padding, an absent segment): zero conditioning -> b_I.  In two halves -- the lane's 28 inputs, the 28 MFMAs -- so that a
// wave that forms TWO row blocks for the same segments (wrnn_sparse.hip) loads the inputs once.
__device__ __forceinline__ void cond_inputs(const float *mel_row, const float *aux_row, bool valid, int lane, float4 (&v)[7])
{
    const int kq = lane >> 4;
    // inputs 28 kq + 4 j .. + 3:  kq 0, 1: mel;  kq 2: mel 56..79 (j < 6), aux 0..3 (j = 6);  kq 3: aux 4..31
    const float *lo = (kq == 3) ? aux_row + 4 : mel_row + CK * kq;
    const float *hi = (kq == 2) ? aux_row : lo + 24;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid) v[j] = *reinterpret_cast<const float4 *>(j < 6 ? lo + 4 * j : hi);
    }
}
__device__ __forceinline__ f32x4 cond_mfma(const CondTile &c, const float4 (&v)[7])
{
    f32x4 a0 = {c.bias[0], c.bias[1], c.bias[2], c.bias[3]}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(c.w[4 * j + 0], v[j].x, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(c.w[4 * j + 1], v[j].y, a1, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(c.w[4 * j + 2], v[j].z, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(c.w[4 * j + 3], v[j].w, a1, 0, 0, 0);
    }
    return a0 + a1;
}
__device__ __forceinline__ f32x4 cond_tile(const CondTile &c, const float *mel_row, const float *aux_row, bool valid, int lane)
{
    float4 v[7];
    cond_inputs(mel_row, aux_row, valid, lane, v);
    return cond_mfma(c, v);
}

// The same inputs with the mel row formed from the three rows of the last up-sampling stage's input that its 2 s + 1 taps reach
// (s = LAST_SCALE): rows r0 (= row j / s - 1), r0 + MEL, r0 + 2 MEL; co = the three tap sums of this lane's phase j % s.
__device__ __forceinline__ void cond_inputs_rows(const float *r0, const float *co, const float *aux_row, bool valid, int lane, float4 (&v)[7])
{
    const int kq = lane >> 4;
#pragma unroll
    for (int j = 0; j < 7; ++j) v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) {
        if (kq == 3) {
#pragma unroll
            for (int j = 0; j < 7; ++j) v[j] = *reinterpret_cast<const float4 *>(aux_row + 4 + 4 * j);
        } else {
            const float c0 = co[0], c1 = co[1], c2 = co[2];
            const float *lo = r0 + CK * kq;
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                if (j == 6 && kq == 2) { v[j] = *reinterpret_cast<const float4 *>(aux_row); continue; }
                const float4 a = *reinterpret_cast<const float4 *>(lo + 4 * j);
                const float4 b = *reinterpret_cast<const float4 *>(lo + MEL + 4 * j);
                const float4 d = *reinterpret_cast<const float4 *>(lo + 2 * MEL + 4 * j);
                v[j] = make_float4(fmaf(c2, d.x, fmaf(c1, b.x, c0 * a.x)), fmaf(c2, d.y, fmaf(c1, b.y, c0 * a.y)),
                                   fmaf(c2, d.z, fmaf(c1, b.z, c0 * a.z)), fmaf(c2, d.w, fmaf(c1, b.w, c0 * a.w)));
            }
        }
    }
}
__device__ __forceinline__ f32x4 cond_tile_rows(const CondTile &c, const float *r0, const float *co, const float *aux_row, bool valid, int lane)
{
    float4 v[7];
    cond_inputs_rows(r0, co, aux_row, valid, lane, v);
    return cond_mfma(c, v);
}

// GRU pointwise math with the hardware transcendentals (v_exp_f32 / v_rcp_f32, ~1 ulp each) instead of the library's expf, tanhf
// and IEEE divisions: ~25 VALU instead of ~120 on the critical back half of every gate stage.  Same algebra as gru_update
// (wrnn_device.h); absolute error ~1e-7 per value, the size of the fp32 rounding already present (MoL tolerance 1e-5: tests).
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }
__device__ __forceinline__ float gru_update_fast(float gi_r, float gi_z, float gi_n, float gh_r, float gh_z, float gh_n, float h)
{
    const float r = fast_sigmoid(gh_r + gi_r);
    const float z = fast_sigmoid(gh_z + gi_z);
    const float n = fast_tanh(gi_n + gh_n * r);
    return (h - n) * z + n;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Exchange-buffer geometry and wave-level helpers shared by the two-workgroups-per-CU kernels (wrnn_duo.hip: dense, 4 clusters x <= 8
// slots; wrnn_sparse.hip: block-sparse GRUs, 16 clusters x 1 slot): a REGION of the buffer belongs to one (slot, cluster) pair of the
// dense kernel / one cluster of the sparse kernel and holds DNX layers x DRING ring entries (by step) of XT floats each.
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr int DNX = 17;                      // layers: 0 h1  1 h2  2 y1  3 y2  4 cI  5 x1  6 x2  7 x_t  8-11 gh1  12-15 gh2 ([32 unit blocks][256 threads][r, z, n, tag])  16 RAW logits
constexpr int DRING = 4;                     // ring entries per layer (by step)
constexpr int DAHEAD_IH = 2;                 // re-arm distance (steps) of the sentinel layers (wrnn_duo.hip: those an ih workgroup publishes; header there)
constexpr int DAHEAD_HH = 3;                 // ... of the layers an hh workgroup publishes
constexpr int DGHRING = 2;                   // ring entries of the tagged gh words (no sentinel, no re-arm: two suffice)
constexpr int DLAYER_ENTRIES = DRING;         // (a pad entry behind every layer ring was measured: no change in traffic or step time, profiles/r04j_traffic.log)
constexpr size_t DXBUF_FLOATS = (size_t)LMAXG * MAXCL * DNX * DLAYER_ENTRIES * XT;      // [slot][cluster][layer][ring (+ pad)][XT]
constexpr int XTB = XT * 4;                  // bytes of one layer entry (32 KB)
constexpr int DLAYERB = DLAYER_ENTRIES * XTB; // bytes of one layer's ring
constexpr int DSLOTB = DNX * DLAYERB;        // bytes of one region
static_assert((size_t)LMAXG * MAXCL * DSLOTB < 0x7FFFFFFFull, "32-bit buffer offsets");

// one fc3 tile with the A fragments already loaded (fragment order), B in registers; mfma_tile's order
__device__ __forceinline__ f32x4 mfma1_frag(const float4 (&av)[8], const float (&b)[32])
{
    f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0;
#pragma unroll
    for (int r = 0; r < 8; r += 2) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r].x, b[4 * r + 0], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r + 1].x, b[4 * r + 4], c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r].y, b[4 * r + 1], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r + 1].y, b[4 * r + 5], c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r].z, b[4 * r + 2], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r + 1].z, b[4 * r + 6], c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r].w, b[4 * r + 3], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r + 1].w, b[4 * r + 7], c1, 0, 0, 0);
    }
    return c0 + c1;
}

// the fragments of a wave as MFMA B operands
__device__ __forceinline__ void frag_to_b(const u32x4 (&x)[8], float (&b)[32])
{
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        b[4 * r + 0] = __uint_as_float(x[r].x);
        b[4 * r + 1] = __uint_as_float(x[r].y);
        b[4 * r + 2] = __uint_as_float(x[r].z);
        b[4 * r + 3] = __uint_as_float(x[r].w);
    }
}
// no word of the live segments' fragments is still the sentinel (wave-uniform; one compare of the running unsigned maximum)
__device__ __forceinline__ bool frag_there(const u32x4 (&x)[8], bool live)
{
    unsigned m = 0u;
#pragma unroll
    for (int r = 0; r < 8; ++r) m = max(max(m, max(x[r].x, x[r].y)), max(x[r].z, x[r].w));
    return __all(m != SENT || !live);
}

// A bounded wait that does not fail the launch's control flow: `there()` (wave-uniform) is re-evaluated after every `reload()`; when
// the spin limit expires or another workgroup has raised the abort flag the wave marks itself dead -- it skips every later wait and
// runs on with whatever the buffers hold (its barrier sequence is unchanged, nothing hangs, wrnn_status() reports the failure).
template <class There, class Reload>
__device__ __forceinline__ void wait_for(There there, Reload reload, unsigned *status, bool &dead, unsigned code, int step)
{
    unsigned spins = 0;
    while (!dead && !there()) {
        if ((++spins & 255u) == 0u) {
            if (ld_agent32(status) != 0u) { dead = true; break; }
            if (spins > SPIN_LIMIT) { report_failure(status, code, blockIdx.x, step, threadIdx.x); dead = true; break; }
        }
        __builtin_amdgcn_s_sleep(1);
        reload();
    }
}

// the workgroup's 1 KB block of a layer: the four units of a quad gathered with DPP, ONE 16-byte store by the quad's first lane;
// `local`: producer and consumers share an XCD (seen at run time) -> a plain write-back store that stays in its L2, else write-through
__device__ __forceinline__ void publish4l(__amdgpu_buffer_rsrc_t rs, int soff, int tid, float v, bool on, bool local)
{
    const int iv = __builtin_bit_cast(int, v);
    const int v0 = __builtin_amdgcn_update_dpp(0, iv, 0x00, 0xF, 0xF, true);
    const int v1 = __builtin_amdgcn_update_dpp(0, iv, 0x55, 0xF, 0xF, true);
    const int v2 = __builtin_amdgcn_update_dpp(0, iv, 0xAA, 0xF, 0xF, true);
    const int v3 = __builtin_amdgcn_update_dpp(0, iv, 0xFF, 0xF, 0xF, true);
    if (on && (tid & 3) == 0) {
        const u32x4 q = {(unsigned)v0, (unsigned)v1, (unsigned)v2, (unsigned)v3};
        if (local) __builtin_amdgcn_raw_buffer_store_b128(q, rs, (tid & ~3) * 4, soff, 0);
        else __builtin_amdgcn_raw_buffer_store_b128(q, rs, (tid & ~3) * 4, soff, 16 /* sc1 */);
    }
}

// row of the slab's per-segment aux tables for position p of a segment whose conditioning ends at lim: its frame (Stretch2d: constant
// over a hop) relative to the segment's first frame of the slab (tbase = segment * rows per segment - that frame); the fold's zero pad
// -> the zero row
__device__ __forceinline__ int table_row(int p, int lim, int tbase, unsigned magic, int shift, int hop, int zrow)
{
    const int q = magic ? (int)(__umulhi((unsigned)p, magic) >> shift) : p / hop;
    return p < lim ? tbase + q : zrow;
}

// three gate tiles, ONE accumulator chain per tile: consecutive MFMAs of a chain are 3 issue slots (96 cycles) apart, more than the
// 40-cycle dependent latency; per output the k terms are summed in ascending order (the oracle's single chain)
__device__ __forceinline__ void mfma3s(const float (&a0)[AF], const float (&a1)[AF], const float (&a2)[AF], const float (&b)[32],
                                       f32x4 &o0, f32x4 &o1, f32x4 &o2)
{
    f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[k], b[k], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[k], b[k], c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[k], b[k], c2, 0, 0, 0);
    }
    o0 = c0; o1 = c1; o2 = c2;
}

// ---- the same tiles with the WEIGHT operand read from the accumulation registers (round 6) -----------------------------------------------
// A one-workgroup-per-CU kernel (wrnn_chain.hip, wrnn_sparse.hip: 512 registers per lane) holds more weight registers than the 256 architectural
// VGPRs.  Left to itself the allocator keeps the surplus in AGPRs AS SPILL SLOTS: every stage began with ~230 v_accvgpr_read / _write copies of the
// tiles it was about to multiply -- on the slot's chain (2,000 such copies in either kernel).  gfx950's MFMA reads srcA from either half of the
// register file, so the tiles named here are CONSTRAINED to AGPRs ("a") and used in place; same products, same order: bit-identical results.
// The hazard recogniser does not see through inline assembly: a block ends with 24 wait states before its accumulators are read (an 8-pass
// XDL write needs <= 19); consecutive MFMAs on one accumulator are >= 2 issue slots apart as in the builtin forms.
__device__ __forceinline__ void mfma_ag(f32x4 &c, float a, float b) { asm("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c) : "a"(a), "v"(b)); }
// (the first product of a chain takes the constant 0 as srcC, as the builtin form does: a zeroed register read by the very next instruction is a
// VALU-write -> MFMA-read the hazard recogniser cannot pad inside inline assembly)
__device__ __forceinline__ f32x4 mfma_ag0(float a, float b)
{
    f32x4 c;
    asm("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=&v"(c) : "a"(a), "v"(b));
    return c;
}
__device__ __forceinline__ void mfma3s_ag(const float (&a0)[AF], const float (&a1)[AF], const float (&a2)[AF], const float (&b)[32], f32x4 &o0, f32x4 &o1, f32x4 &o2)
{
    f32x4 c0 = mfma_ag0(a0[0], b[0]), c1 = mfma_ag0(a1[0], b[0]), c2 = mfma_ag0(a2[0], b[0]);
#pragma unroll
    for (int k = 1; k < 32; ++k) {
        mfma_ag(c0, a0[k], b[k]);
        mfma_ag(c1, a1[k], b[k]);
        mfma_ag(c2, a2[k], b[k]);
    }
    asm volatile("s_nop 15\n\ts_nop 7" : "+v"(c0), "+v"(c1), "+v"(c2));
    o0 = c0; o1 = c1; o2 = c2;
}
__device__ __forceinline__ f32x4 mfma1_ag(const float (&a)[AF], const float (&b)[32])
{
    f32x4 c0 = mfma_ag0(a[0], b[0]), c1 = mfma_ag0(a[4], b[4]);
#pragma unroll
    for (int r = 0; r < 8; r += 2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (r + e > 0) {
                mfma_ag(c0, a[4 * r + e], b[4 * r + e]);
                mfma_ag(c1, a[4 * r + 4 + e], b[4 * r + 4 + e]);
            }
        }
    }
    asm volatile("s_nop 15\n\ts_nop 7" : "+v"(c0), "+v"(c1));
    return c0 + c1;
}

}  // namespace wrnn
