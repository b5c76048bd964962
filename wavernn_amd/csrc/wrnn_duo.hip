// wrnn_duo.hip -- the TWO-WORKGROUPS-PER-CU form of the persistent WaveRNN loop kernel (MOL) for MI355X (gfx950 / CDNA4).
//
// Same path, same arithmetic and the same tag-free sentinel exchange in MFMA-fragment order as wrnn_loop.hip (reference
// models/fatchord_version.py:201-241); what changes is WHO holds WHAT, for one reason (profiles/r02r_summary.md): a
// wrnn_loop_kernel workgroup keeps 224 weight registers per lane, so it is alone on its CU with ONE wave per SIMD, and a
// single in-order wave cannot overlap its 256 MFMAs per group-step (8.2 k cycles) with the ~14 k cycles of loads, partial
// sums, pointwise math, publishes and barrier waits around them -- the matrix pipe idles 63 % of the time.  Here every
// role of wrnn_loop.hip is cut in two along the line between its critical and its off-critical half:
//
//     role 0  A-ih : rnn1 W_ih (3 gate tiles) + fc1 (1 tile)   = 128 weight registers   phases P0 (gates -> h1, x1), P2 (fc1 -> y1)
//     role 1  B-ih : rnn2 W_ih (3 gate tiles) + fc2 (1 tile)   = 128                    phases P0 (gates -> h2, x2), P2 (fc2 -> y2)
//     role 2  A-hh : rnn1 W_hh (3 gate tiles)                  =  96                    phase  P1 (gh1(t+1) = W_hh1 . h1(t) + b_hh)
//     role 3  B-hh : rnn2 W_hh (3 gate tiles)                  =  96                    phase  P1 (gh2(t+1)) [+ fc3 and sampling, below]
//
// so a workgroup fits 256 registers per lane and TWO workgroups share a CU: two waves per SIMD, one running MFMAs while the
// other issues its VALU / LDS / memory work (MI355X guide: the matrix and vector pipes of a SIMD run concurrently for
// different waves).  A cluster is 128 workgroups on the same 64 CUs; the issued MFMA count per CU and group-step drops from
// 1024 to ~900 (fc3 below) and, more to the point, can now overlap everything else.
//
// What the cut costs: gh(t+1) = W_hh . h(t) + b_hh now crosses workgroups -- three more exchange layers per GRU
// ([gate][16 units x 16 segments] = 3 KB per workgroup and group-step, read by ONE workgroup, a full step after it was
// written: off the critical path).  fc3 + sampling (MOL: 30 x 512, 0.4 % of the FLOPs) used to run redundantly in all 32 workgroups of
// the sampling role (64 of its 256 MFMAs per group-step); here ONE hh workgroup per slot runs it (slot i: role 2 + (i & 1),
// unit block J = i >> 1), reading the fc3 fragments from L2, and hands x_t to the 32 A-ih workgroups through the 16-word
// exchange layer wrnn_loop.hip already uses for role-B-sampled slots.  That is one more hop on a slot's chain (this kernel
// is for >= 2 groups in flight per cluster, where the busy time of a workgroup bounds a step, not the latency of a
// slot; wrnn_loop_kernel stays the kernel for one group per cluster).
//
// Ring discipline (conservative form of wrnn_loop.hip's): 8 ring entries per layer; at step t a wave re-arms its own words of
// entry (t + 4) % 8 -- data of step t - 4, which every consumer left behind long ago -- and drains its stores at step t + 1;
// the entry is written again at step t + 3 (gh(t+4), published one step early) or t + 4.  Any poll of that entry belongs to a
// consumer's step >= t + 4, which exists only if every workgroup of the cluster has published something of its step t + 2
// or later, i.e. has passed the drain of step t + 1: the re-arm is visible before the poll.
#include <type_traits>

#include "wrnn_ring.h"

namespace wrnn {

constexpr int DNX = 16;                      // layers: 0 h1  1 h2  2 y1  3 y2  4 -  5 x1  6 x2  7 x_t  8-11 gh1  12-15 gh2 ([32 unit blocks][256 threads][r, z, n, -]: 4 layers' worth)
constexpr int DRING = 8;
constexpr int DAHEAD = 4;                    // re-arm distance (steps)
constexpr int DNWGC = 4 * LNJ;               // workgroups per cluster (128)
constexpr size_t DXBUF_FLOATS = (size_t)LMAXG * MAXCL * DNX * DRING * XT;      // [slot][cluster][layer][ring][XT]
constexpr int DLOGS = 36;                    // as LOGS of wrnn_loop.hip
constexpr int DPART = 2 * NW * 3 * 256;      // two ping-pong sets of [wave][slot 0..2][16][16]
// Build-time variants of the ih roles (measured on hardware, profiles/r03e_*: the look-ahead bought nothing -- an ih workgroup waits
// for data that is not published yet, not for load latency -- and the 32 registers it takes had to come from somewhere):
//   DUO_IH_XAHEAD 1 = one-stage look-ahead of the ih roles' operand loads (needs DUO_FC_LDS 1 or DUO_FC_GLB 1 to fit 256 registers)
//   DUO_FC_LDS    1 = the fc1 / fc2 tile in LDS (A-fragment order) instead of 32 registers
#ifndef DUO_IH_XAHEAD
#define DUO_IH_XAHEAD 0
#endif
#ifndef DUO_FC_LDS
#define DUO_FC_LDS 0
#endif
//   DUO_FC_GLB    1 = the ih roles read their fc1 / fc2 tile from L2 every fc stage (LoopArgs.fc12f, fragment order, 32 KB per workgroup,
//                 issued before the operand poll) instead of holding it in 32 registers -- which then carry the look-ahead.
//                 Measured (profiles/r03x_probe_fc_glb_ih_lookahead.json): alone 3 % slower (0 spills); with DUO_IH_XAHEAD the combined
//                 kernel still spills 46-55 VGPRs and runs 50 % slower.  Off.
#ifndef DUO_FC_GLB
#define DUO_FC_GLB 0
#endif
//   DUO_FAST_PW   1 = hardware exp / rcp in the GRU pointwise math (gru_update_fast), 0 = the library forms (gru_update, as wrnn_loop.hip)
//   DUO_PUBLISH_FIRST 1 = a stage starts with the previous stage's back half (barrier, pointwise, PUBLISH) and only then issues its own
//                 loads: the publication leaves ~1 k cycles earlier per hop, which is what bounds a step when the slots' chains are
//                 not hidden (<= 4 groups in flight); 0 = loads first, as wrnn_loop.hip (better latency hiding when busy-bound)
#ifndef DUO_PUBLISH_FIRST
#define DUO_PUBLISH_FIRST 1
#endif
//   DUO_FC_LAG    the fc stage of slot k - lag follows the gate stage of slot k (see duo_ih's main loop)
#ifndef DUO_FC_LAG
#define DUO_FC_LAG 8       // (measured, profiles/r03m_probe_lag.json: the plain order wins at depth 4 and 8; lag 1-2 cost 9-17 %)
#endif
#ifndef DUO_FAST_PW
#define DUO_FAST_PW 1
#endif
//   DUO_SPLIT_FC  1 = the fc1 / fc2 stages of the odd slots run on the hh workgroups (load balance of the two workgroups of a CU).
//                 Measured (profiles/r03w_probe_split_fc.json): 9-14 % SLOWER at depth 3-8 -- the hh workgroup then has no registers
//                 for its one-stage look-ahead (which alone is worth 5-13 %), and its gh stages are less off the chain than the
//                 phase clocks suggested.  Off.
#ifndef DUO_SPLIT_FC
#define DUO_SPLIT_FC 0
#endif
//   DUO_HH_XAHEAD 1 = the hh workgroups load the next stage's operand fragments one stage ahead.  With the fc tile next to W_hh (128
//                 weight registers, DUO_SPLIT_FC) the 32 look-ahead registers spill (71 VGPRs to scratch), as they did in duo_ih: off.
#ifndef DUO_HH_XAHEAD
#define DUO_HH_XAHEAD (!DUO_SPLIT_FC)
#endif
#ifndef DUO_MFMA3
#define DUO_MFMA3 mfma3
#endif
// per-group LDS state of an ih workgroup (floats): HOWN[256] (h of the owned unit x segment), XS[16] (x_{t-1}; layer 1), SP[32 ints]
// (segment table), FR[2][16 ints] (conditioning frame of every segment at step t in FR[t & 1]), XO[256] (the owned units' slice of
// the GRU input, for the residual sum).  The saved state of a slot in global memory is [GH 768 = gh(t1)][these DGRP floats] = LGRP.
constexpr int DGRP = 256 + 16 + 32 + 32 + 256;
constexpr int D_HOWN = 0, D_XS = 256, D_SP = 272, D_FR = 304, D_XO = 336;
static_assert(768 + DGRP == LGRP, "saved state layout");

struct DuoLds {
    int off_part, off_log, off_wi0, off_fc, off_misc, off_prof, total;
};
__host__ __device__ inline DuoLds duo_lds(int G)
{
    DuoLds l;
    int o = G * DGRP;
    l.off_part = o; o += DPART;
    l.off_log = o;  o += SEG * DLOGS;
    l.off_wi0 = o;  o += H;
    l.off_fc = o;   o += DUO_FC_LDS ? XT : 0;    // ih: the 16 owned rows of fc1 / fc2 in A-fragment order (the 4th weight tile lives in LDS so
                                             // that the registers it would take hold the one-stage look-ahead of the operand loads)
    l.off_misc = o; o += 16 + 2 * LMAXG;     // [0] failure flag; [16 + 2 i], [17 + 2 i]: first segment / segment count of slot i
    l.off_prof = o; o += 2 * 16;             // [16] u64 phase clocks (profiling builds)
    l.total = o;
    return l;
}

// re-arm THIS WAVE's quarter (16 lanes x 16 bytes) of the workgroup's 1 KB block in up to four layers of one ring entry
__device__ __forceinline__ void duo_rearm(__amdgpu_buffer_rsrc_t rs, int soff_entry0 /* bytes: layer 0 of the (slot, entry) + block + quarter */,
                                          int lane, int la, int lb, int lc, int ld)
{
    const int which = lane >> 4;
    const int layer = which == 0 ? la : (which == 1 ? lb : (which == 2 ? lc : ld));
    if (layer >= 0) {
        const u32x4 q = {SENT, SENT, SENT, SENT};
        __builtin_amdgcn_raw_buffer_store_b128(q, rs, layer * (DRING * XT * 4) + (lane & 15) * 16, soff_entry0, 16 /* sc1 */);
    }
}

// one tile with the A fragments already loaded (fragment order), B in registers; mfma_tile's order
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

// one fc3 tile with the A fragments read from global memory (L2-resident, fragment order), B in registers; mfma_tile's order
__device__ __forceinline__ f32x4 mfma1_glb(const float *a_lane /* tile + frag_off(w, 0, lane) */, const float (&b)[32])
{
    float4 av[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) av[r] = *reinterpret_cast<const float4 *>(a_lane + 256 * r);
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

// three gate tiles, ONE accumulator chain per tile: consecutive MFMAs of a chain are 3 issue slots (96 cycles) apart, more than the
// 40-cycle dependent latency, and 12 accumulator registers instead of mfma3's 24 leave room for the look-ahead operands.  Per
// output the k terms are summed in ascending order (mfma3 sums even and odd k-blocks separately): closer to the oracle's single chain.
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

#define DXL(i, layer, ring) ((((((i) * MAXCL + cl) * DNX + (layer)) * DRING) + (ring)) * XT)
#define DPARTOF(q) (PART + (q) * (NW * 3 * 256))

// (fast_sigmoid / fast_tanh / gru_update_fast: wrnn_ring.h)

// bounded poll of this thread's 16-byte gh word {r, z, n, tag}: ONE 16-byte sc1 store by ONE lane, so the word is its own flag (MI355X
// guide, hand-off form R2) -- tag = consuming step + 1, no sentinel, hence no re-arm store and no ordering rule for these layers
// (the sentinel fill 0xFFFFFFFF and the tags of earlier laps never match).  `live` threads only.
__device__ __forceinline__ bool poll_gh(__amdgpu_buffer_rsrc_t rs, int voff, int soff, bool live, unsigned tag, u32x4 &g, unsigned *status)
{
    unsigned spins = 0;
    while (__any(live && g.w != tag)) {
        if ((++spins & 255u) == 0u) {
            if (spins > SPIN_LIMIT || ld_agent32(status) != 0u) return false;
        }
        __builtin_amdgcn_s_sleep(1);
        g = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 16 /* sc1 */);
    }
    return true;
}

// bounded poll of up to three 4-byte words of this thread (sentinel = not written); `live` threads only.  Wave-uniform result.
__device__ __forceinline__ bool poll3(__amdgpu_buffer_rsrc_t rs, int voff, int s0, int s1, int s2, bool live, unsigned &g0, unsigned &g1,
                                      unsigned &g2, unsigned *status)
{
    unsigned spins = 0;
    while (__any(live && (g0 == SENT || g1 == SENT || g2 == SENT))) {
        if ((++spins & 255u) == 0u) {
            if (spins > SPIN_LIMIT || ld_agent32(status) != 0u) return false;
        }
        __builtin_amdgcn_s_sleep(1);
        g0 = __builtin_amdgcn_raw_buffer_load_b32(rs, voff, s0, 16 /* sc1 */);
        g1 = __builtin_amdgcn_raw_buffer_load_b32(rs, voff, s1, 16 /* sc1 */);
        g2 = __builtin_amdgcn_raw_buffer_load_b32(rs, voff, s2, 16 /* sc1 */);
    }
    return true;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// ih workgroup: LA = layer 1 (rnn1 W_ih + fc1; needs x_{t-1} for xi) or layer 2 (rnn2 W_ih + fc2).  Owns the GRU state of its 16
// units (h, the gate pointwise math) and publishes h / the residual sum / relu(fc).
// PROF (thread 0, shader clocks per segment of a stage, [ph0: 0-7, ph2: 8-15]): 0 front issue, 1 barrier wait, 2 back half, 3 operand
// wait / poll, 4 hygiene + look-ahead issue + operand build, 5 MFMA tiles + partial writes, 6 stages, 7 stages that had to poll
// ---------------------------------------------------------------------------------------------------------------------------------
template <bool LA, bool PROF>
__device__ __forceinline__ void duo_ih(const LoopArgs &a, float *smem, int cl, int J, int ncl)
{
    const int G = a.G;
    const DuoLds L = duo_lds(G);
    float *PART = smem + L.off_part, *WI0 = smem + L.off_wi0, *FC = smem + L.off_fc;
    int *FAIL = reinterpret_cast<int *>(smem + L.off_misc);
    int *GEO = FAIL + 16;
    u64 *PROFL = reinterpret_cast<u64 *>(smem + L.off_prof);
    u64 plast = 0;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
#define PH(k)                                                                  \
    do {                                                                       \
        if (PROF && tid == 0) {                                                \
            const u64 now_ = __builtin_amdgcn_s_memtime();                     \
            PROFL[k] += now_ - plast;                                          \
            plast = now_;                                                      \
        }                                                                      \
    } while (0)
    const int fi = lane & 15, kq = lane >> 4;
    const int kbase_lane = KCH * w + 4 * kq;
    const int pu = 4 * w + (tid & 3), pj = (tid >> 2) & 15;   // pointwise role: (owned unit, segment)
    const int prow = LU * J + pu;
    const int T0 = a.t0, T1 = a.t1;
    const int NR = a.Btot, NGR = a.NG;
    constexpr int L_H = LA ? 0 : 1, L_XR = LA ? 5 : 6, L_Y = LA ? 2 : 3, L_GH = LA ? 8 : 12;       // layers this role publishes / reads gh from
    constexpr int L_P0 = LA ? -1 : 5, L_P2 = LA ? 6 : 2;                                           // layers its stages poll

    float A_ih[3][AF];
#pragma unroll
    for (int g = 0; g < 3; ++g) load_afrag(A_ih[g], LA ? a.w_ih1 : a.w_ih2, LA ? H : H + AUX, g * H + LU * J + fi, true, kbase_lane);
#if !DUO_FC_LDS && !DUO_FC_GLB
    float A_fc[AF];
    load_afrag(A_fc, LA ? a.fc1_w : a.fc2_w, H + AUX, LU * J + fi, true, kbase_lane);
#endif
    const float *bhh = LA ? a.b_hh1 : a.b_hh2;

    for (int q = tid; q < L.total; q += NT) smem[q] = 0.f;
    __syncthreads();
    if constexpr (LA) {
        WI0[2 * tid] = a.I_w0[2 * tid];
        WI0[2 * tid + 1] = a.I_w0[2 * tid + 1];
    }
#if DUO_FC_LDS
    {   // fc1 / fc2 rows [16 J, 16 J + 16) -> LDS in A-fragment order: FC[wave][r][lane (row fi, k-quad kq)][4] = W[16 J + fi][128 wave + 16 r + 4 kq ..]
        const float *fcw = LA ? a.fc1_w : a.fc2_w;
        for (int q = tid; q < XT / 4; q += NT) {
            const int l6 = q & 63, r = (q >> 6) & 7, wv = (q >> 9) & 3;
            reinterpret_cast<float4 *>(FC)[q] =
                *reinterpret_cast<const float4 *>(fcw + (size_t)(LU * J + (l6 & 15)) * (H + AUX) + KCH * wv + 16 * r + 4 * (l6 >> 4));
        }
    }
#endif
    int nact = 0;
    for (int i = 0; i < G; ++i)
        if (cl + ncl * i < NGR) nact = i + 1;
    u64 nbpack = 0;                                     // segment count of slot i in byte i (a scalar register pair: no LDS round trip per stage)
    // saved state: the slot layout of wrnn_loop.hip ([cluster][2 J + layer][slot][LGRP]): [0, 768) = gh(t1) of the finished launch
    // (the next launch's first step reads it straight from there), then the DGRP floats of the LDS state
    const size_t state_wg = ((size_t)(cl * LNWGC + 2 * J + (LA ? 0 : 1)) * G) * LGRP;
    for (int i = 0; i < nact; ++i) {
        float *GP = smem + i * DGRP;
        const int g = cl + ncl * i;
        const int b0 = (int)(((long)g * NR) / NGR), nb = (int)(((long)(g + 1) * NR) / NGR) - b0;
        if (tid == 0) { GEO[2 * i] = a.rb0 + b0; GEO[2 * i + 1] = nb; }
        nbpack |= (u64)(unsigned)nb << (8 * i);
        if (a.resume) {
            const float4 *src = reinterpret_cast<const float4 *>(a.state + state_wg + (size_t)i * LGRP + 768);
            for (int q = tid; q < DGRP / 4; q += NT) reinterpret_cast<float4 *>(GP)[q] = src[q];
        } else {
            GP[D_HOWN + tid] = 0.f;                     // fatchord_version.py:194-196: h1 = h2 = 0, x = 0
            if (tid < SEG) {
                GP[D_XS + tid] = 0.f;
                int *SP = reinterpret_cast<int *>(GP + D_SP);
                const int sc = a.rb0 + b0 + (tid < nb ? tid : nb - 1);
                SP[tid] = a.seg_pos[sc];
                SP[SEG + tid] = a.seg_lim[sc];
                const int p0 = SP[tid] + T0;
                reinterpret_cast<int *>(GP + D_FR)[SEG * (T0 & 1) + tid] = (p0 < SP[SEG + tid]) ? (p0 / a.hop) : a.NF;
            }
        }
    }
    __syncthreads();

    const __amdgpu_buffer_rsrc_t xrs = make_rsrc(a.xbuf, (unsigned)(DXBUF_FLOATS * 4));
    const int ghoff = (256 * (J & 7) + tid) * 16;       // byte offset of this thread's {r, z, n, -} gh word in layer L_GH + (J >> 3): eight unit blocks per layer

    float touch = 0.f;
    int pp = 0;
    bool ok = true;
    unsigned fcode = 0u;
    int t = T0;

    const bool prio_mfma = (a.tuning & 8) != 0;          // A/B: raise the wave's priority around its MFMA tiles
    if (a.tuning & 16) __builtin_amdgcn_s_setprio(1);    // A/B: static priority for the ih workgroups (the longer instruction stream)
    const bool lookahead = DUO_IH_XAHEAD && (a.tuning & 1) == 0;          // A/B: bit 0 = no one-stage look-ahead of the operand loads
    // fc1 / fc2 stages of the ODD slots run on the hh workgroup of the same layer and unit block (which holds the same fc tile): the
    // ih workgroup's instruction stream is what bounds a busy step (gates + fc = 16 k cycles per group-step against the hh
    // workgroup's 9 k; profiles/r03g_duo_phase_clocks_depth8.json).  tuning bit 7 = all fc stages here, as before.
    const bool split_fc = DUO_SPLIT_FC && (a.tuning & 128) == 0 && !DUO_IH_XAHEAD;
    const int last_fc = split_fc ? ((nact - 1) & ~1) : nact - 1;           // the last fc stage this workgroup runs in a step
    enum { BK_NONE = 0, BK_GATES, BK_RELU };
    // x: the operand fragments of the NEXT stage, loaded one stage ahead (before this stage's MFMA tiles): xa = 1 an exchanged layer
    // (may still hold sentinels: checked at the stage's start), xa = 2 the conditioning slab cI (layer 1's gate stages: plain data)
    u32x4 x[8];
    int xa = 0;
    int bk = BK_NONE, bi = 0, bpp = 0, bt = 0;
    float bc0 = 0.f, bc1 = 0.f, bc2 = 0.f;
    u32x4 bg = {0u, 0u, 0u, 0u};                        // gh word of the pending GATES half (loaded in its stage's front)
    unsigned xtw = 0u;
    int cur = 0;                                        // PROF: 0 gate stage, 8 fc stage

    auto poll_xt = [&](int i, int ring, int nb, unsigned &v) -> bool {
        unsigned spins = 0;
        while (__any(fi < nb && v == SENT)) {
            if ((++spins & 255u) == 0u) {
                if (spins > SPIN_LIMIT || ld_agent32(a.status) != 0u) return false;
            }
            __builtin_amdgcn_s_sleep(1);
            v = __builtin_amdgcn_raw_buffer_load_b32(xrs, fi * 4, DXL(i, 7, ring) * 4, 16 /* sc1 */);
        }
        return true;
    };

    auto run_back = [&]() -> bool {
        if (bk == BK_NONE) return true;
        float *GP = smem + bi * DGRP;
        const float *PB = DPARTOF(bpp);
        const int nb = (int)((nbpack >> (8 * bi)) & 255u);
        const int bring = bt % DRING;
        if (!ok) FAIL[0] = 1;
        lds_barrier();
        if (FAIL[0] != 0) return false;
        PH(cur + 1);
        if (bk == BK_GATES) {                           // GRU cell pointwise (ATen gru_cell) -> publish h and the residual sum
            const float gir = get_partial<3>(PB, 0, pu, pj) + bc0;
            const float giz = get_partial<3>(PB, 1, pu, pj) + bc1;
            const float gin = get_partial<3>(PB, 2, pu, pj) + bc2;
            float ghr, ghz, ghn;
            if (bt > T0) {                              // gh(t) from the hh workgroup of the same unit block (published during step t - 1)
                const bool got = poll_gh(xrs, ghoff, DXL(bi, L_GH + (J >> 3), bring) * 4, pj < nb, (unsigned)bt + 1u, bg, a.status);
                if (!got) { ok = false; if (fcode == 0u) fcode = 0x500u | (LA ? 0u : 8u) | 6u; }
                ghr = __uint_as_float(bg.x); ghz = __uint_as_float(bg.y); ghn = __uint_as_float(bg.z);
            } else if (a.resume) {                      // first step of a continuing launch: gh(t0), saved by the launch that ended there
                const float *sg = a.state + state_wg + (size_t)bi * LGRP;
                ghr = sg[tid]; ghz = sg[256 + tid]; ghn = sg[512 + tid];
            } else {                                    // t = 0: gh = W_hh . 0 + b_hh = b_hh
                ghr = bhh[prow]; ghz = bhh[H + prow]; ghn = bhh[2 * H + prow];
            }
            const float hprev = GP[D_HOWN + tid];
            const float hn = DUO_FAST_PW ? gru_update_fast(gir, giz, gin, ghr, ghz, ghn, hprev) : gru_update(gir, giz, gin, ghr, ghz, ghn, hprev);
            GP[D_HOWN + tid] = hn;
            publish4(xrs, (DXL(bi, L_H, bring) + 256 * J) * 4, tid, hn, pj < nb);
            publish4(xrs, (DXL(bi, L_XR, bring) + 256 * J) * 4, tid, GP[D_XO + tid] + hn, pj < nb);      // x1 = xi + h1 (:212) / x2 = x1 + h2 (:216)
            if (tid < SEG) {   // conditioning frame of every segment at the NEXT step, into the other half of FR (first read a step from here)
                const int *SP = reinterpret_cast<const int *>(GP + D_SP);
                const int p1 = SP[tid] + bt + 1;
                reinterpret_cast<int *>(GP + D_FR)[SEG * ((bt + 1) & 1) + tid] = (p1 < SP[SEG + tid]) ? (p1 / a.hop) : a.NF;
            }
        } else {                                        // fc1 / fc2 + relu -> publish y1 / y2
            publish4(xrs, (DXL(bi, L_Y, bring) + 256 * J) * 4, tid, fmaxf(get_partial<3>(PB, 0, pu, pj) + bc0, 0.f), pj < nb);
        }
        bk = BK_NONE;
        PH(cur + 2);
        return true;
    };

    int ring = 0, tc = 0;
    // A STAGE (phase ph = 0 gates / 2 fc, slot i), one-stage software pipeline as in wrnn_loop.hip: consume the loads issued a
    // stage ago, run the previous stage's back half, issue the next stage's loads, build the operands, run the MFMA tiles.
    auto stage = [&](auto PHC, int i) -> bool {
        constexpr int ph = decltype(PHC)::value;
        float *GP = smem + i * DGRP;
        const int g = cl + ncl * i;
        const int nb = (int)((nbpack >> (8 * i)) & 255u);
        float4 c[8];
        float v0 = 0.f, v1 = 0.f, v2 = 0.f;
        u32x4 gw = {0u, 0u, 0u, 0u};
        constexpr bool polled = !(LA && ph == 0);
        constexpr int xl = ph == 0 ? L_P0 : L_P2;
        float b[32];
        bool ready = false;
        cur = ph == 0 ? 0 : 8;
        if (PROF && tid == 0 && plast == 0) plast = __builtin_amdgcn_s_memtime();
#if DUO_PUBLISH_FIRST
        // ---------------- the previous stage's back half FIRST: its publication is what the next hop of its slot's chain waits for ----------------
        if (!run_back()) return false;
#endif
        if (polled) {
            if (DUO_IH_XAHEAD && xa == 1) ready = try_finish(lane, nb, x, b);
            else issue(xrs, DXL(i, xl, ring) * 4, w, lane, x);
        } else {
            if (DUO_IH_XAHEAD && xa == 2) {
#pragma unroll
                for (int r = 0; r < 8; ++r) c[r] = make_float4(__uint_as_float(x[r].x), __uint_as_float(x[r].y), __uint_as_float(x[r].z), __uint_as_float(x[r].w));
            } else load_cI(a.cIf + ((size_t)tc * NGR + g) * XT, w, lane, c);
        }
#if DUO_FC_GLB
        float4 av[8];                                   // the fc tile's A fragments of this wave, in flight under the operand wait
        if (ph == 2) {
            const float4 *fp = reinterpret_cast<const float4 *>(a.fc12f + (size_t)((LA ? 0 : LNJ) + J) * XT + frag_off(w, 0, lane));
#pragma unroll
            for (int r = 0; r < 8; ++r) av[r] = fp[r * 64];
        }
#endif
        if (ph == 0) {
            if constexpr (LA) {
                v0 = a.b_ih1[prow]; v1 = a.b_ih1[H + prow]; v2 = a.b_ih1[2 * H + prow];      // layer 1: b_ih1 (layer 2's b_ih2 is inside c2f)
                if (t > T0)                             // x_{t-1}, sampled by an hh workgroup (first step of a launch: from the state)
                    xtw = __builtin_amdgcn_raw_buffer_load_b32(xrs, fi * 4, DXL(i, 7, (t + DRING - 1) % DRING) * 4, 16 /* sc1 */);
            } else {
                const int fr = reinterpret_cast<const int *>(GP + D_FR)[SEG * (t & 1) + pj];
                v0 = a.c2f[(size_t)fr * 3 * H + prow];
                v1 = a.c2f[(size_t)fr * 3 * H + H + prow];
                v2 = a.c2f[(size_t)fr * 3 * H + 2 * H + prow];
            }
            if (t > T0)                                 // gh(t) of this slot: consumed by this stage's back half, one stage from now
                gw = __builtin_amdgcn_raw_buffer_load_b128(xrs, ghoff, DXL(i, L_GH + (J >> 3), ring) * 4, 16 /* sc1 */);
        } else {
            const int fr = reinterpret_cast<const int *>(GP + D_FR)[SEG * (t & 1) + pj];
            v0 = (LA ? a.c3f : a.c4f)[(size_t)fr * H + prow];
        }
        PH(cur + 0);
#if !DUO_PUBLISH_FIRST
        // ---------------- the previous stage's back half ----------------
        if (!run_back()) return false;
#endif
        // ---------------- operands -> MFMA tiles -> this wave's partial tiles ----------------
        if (polled) {
            unsigned spins = 0;
            if (!ready) {
                ok = ok && finish(xrs, DXL(i, xl, ring) * 4, w, lane, nb, x, b, a.status, spins);
                if (!ok && fcode == 0u) fcode = 0x500u | (LA ? 0u : 8u) | (unsigned)ph;
            }
            if (PROF && tid == 0) { PROFL[cur + 6] += 1; PROFL[cur + 7] += !ready; }
        }
        PH(cur + 3);
        if (ph == 2 && i == last_fc) {
            // ring hygiene, once per step, after the last layer this workgroup polls in the step has arrived (see the header)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const int ringn = (t + DAHEAD) % DRING;
#pragma unroll 1
            for (int i2 = 0; i2 < nact; ++i2) duo_rearm(xrs, (DXL(i2, 0, ringn) + 256 * J + 64 * w) * 4, lane, L_H, L_Y, L_XR, -1);
        }
        if (LA && ph == 0) {
            float xs = GP[D_XS + fi];
            if (t > T0) {
                ok = ok && poll_xt(i, (t + DRING - 1) % DRING, nb, xtw);
                if (!ok && fcode == 0u) fcode = 0x500u | 0x20u;
                xs = (fi < nb) ? __uint_as_float(xtw) : 0.f;
            }
            make_xi(c, WI0, xs, w, lane, b);            // xi(t) (:208-209)
        }
        if (ph == 0 && w == (J >> 3)) {
            // the owned units' slice of this GRU's input (layer 1: xi, layer 2: x1) -> LDS in publish order, for the residual sum
#pragma unroll
            for (int r = 0; r < 8; ++r)
                if (r == (J & 7)) *reinterpret_cast<float4 *>(GP + D_XO + 4 * lane) = make_float4(b[4 * r], b[4 * r + 1], b[4 * r + 2], b[4 * r + 3]);
        }
        {   // the next stage's operand fragments, one stage ahead: an exchanged layer (not across a step boundary: nothing of the
            // next step is published yet) or, for layer 1's gate stages, the conditioning slab (plain data: also across the boundary)
            int nph = ph, ni = i + 1;
            if (ni >= nact) { nph = ph + 2; ni = 0; }
            xa = 0;
            if (lookahead) {
                if (nph <= 2) {
                    if (LA && nph == 0) {
                        const u32x4 *cp = reinterpret_cast<const u32x4 *>(a.cIf + ((size_t)tc * NGR + cl + ncl * ni) * XT + frag_off(w, 0, lane));
#pragma unroll
                        for (int r = 0; r < 8; ++r) x[r] = cp[r * 64];
                        xa = 2;
                    } else {
                        issue(xrs, DXL(ni, nph == 0 ? L_P0 : L_P2, ring) * 4, w, lane, x);
                        xa = 1;
                    }
                } else if (LA && t + 1 < T1) {
                    const u32x4 *cp = reinterpret_cast<const u32x4 *>(a.cIf + ((size_t)(tc + 1) * NGR + cl) * XT + frag_off(w, 0, lane));
#pragma unroll
                    for (int r = 0; r < 8; ++r) x[r] = cp[r * 64];
                    xa = 2;
                }
            }
        }
        PH(cur + 4);
        float *PW = DPARTOF(pp);
        if (prio_mfma) __builtin_amdgcn_s_setprio(1);
        if (ph == 0) {
            f32x4 o0, o1, o2;
            DUO_MFMA3(A_ih[0], A_ih[1], A_ih[2], b, o0, o1, o2);
            put_partial<3>(PW, w, 0, lane, o0);
            put_partial<3>(PW, w, 1, lane, o1);
            put_partial<3>(PW, w, 2, lane, o2);
            bk = BK_GATES;
        } else {
#if DUO_FC_LDS
            put_partial<3>(PW, w, 0, lane, mfma1_lds(FC + frag_off(w, 0, lane), b));
#elif DUO_FC_GLB
            put_partial<3>(PW, w, 0, lane, mfma1_frag(av, b));
#else
            put_partial<3>(PW, w, 0, lane, mfma1(A_fc, b));
#endif
            bk = BK_RELU;
        }
        if (prio_mfma) __builtin_amdgcn_s_setprio(0);
        if (LA && ph == 0 && t + 1 < T1) {     // pull the next step's conditioning block of this group (32 KB) into this XCD's L2
            asm volatile("" ::"v"(touch));
            touch = a.cIf[((size_t)(tc + 1) * NGR + g) * XT + 32 * tid];
        }
        PH(cur + 5);
        bi = i; bpp = pp; bt = t; bc0 = v0; bc1 = v1; bc2 = v2; bg = gw;
        pp ^= 1;
        return true;
    };

    // Order of a step's stages: the gate stage of slot k, then the fc stage of slot k - lag.  lag = nact is wrnn_loop.hip's order (all
    // gate stages, then all fc stages); a small lag lets a slot's fc stage run as soon as its operand (two hops behind the slot's
    // gate stage) can be there instead of behind the gate stages of every other slot -- the slots' chains, which bound a step
    // with <= 4 groups in flight, get shorter.  wrnn_options.tuning bits 9-11: lag (0 = the default)
    int lag = (a.tuning >> 9) & 7;
    if (lag == 0) lag = DUO_FC_LAG;
    if (lag > nact || DUO_IH_XAHEAD) lag = nact;        // (the ih look-ahead variant assumes the plain order)
    for (; t < T1; ++t) {
        ring = t % DRING;
        tc = t - a.cI_t0;
#pragma unroll 1
        for (int k = 0; k < nact + lag; ++k) {
            if (k < nact) {
                if (!stage(std::integral_constant<int, 0>{}, k)) goto bail;
            }
            if (k >= lag && !(split_fc && ((k - lag) & 1))) {
                if (!stage(std::integral_constant<int, 2>{}, k - lag)) goto bail;
            }
        }
    }
    if (!run_back()) goto bail;
    // ---- what the next launch of this round needs from the ring: gh(T1) of every slot (published during step T1 - 1) -> the saved
    //      state in global memory, and, for layer 1, x_{T1-1} -> XS
#pragma unroll 1
    for (int i = 0; i < nact; ++i) {
        float *GP = smem + i * DGRP;
        const int nb = (int)((nbpack >> (8 * i)) & 255u);
        const int r1 = T1 % DRING;
        u32x4 gq = __builtin_amdgcn_raw_buffer_load_b128(xrs, ghoff, DXL(i, L_GH + (J >> 3), r1) * 4, 16 /* sc1 */);
        ok = ok && poll_gh(xrs, ghoff, DXL(i, L_GH + (J >> 3), r1) * 4, pj < nb, (unsigned)T1 + 1u, gq, a.status);
        float *sg = a.state + state_wg + (size_t)i * LGRP;
        sg[tid] = __uint_as_float(gq.x); sg[256 + tid] = __uint_as_float(gq.y); sg[512 + tid] = __uint_as_float(gq.z);
        if constexpr (LA) {
            unsigned v = __builtin_amdgcn_raw_buffer_load_b32(xrs, fi * 4, DXL(i, 7, (T1 + DRING - 1) % DRING) * 4, 16 /* sc1 */);
            ok = ok && poll_xt(i, (T1 + DRING - 1) % DRING, nb, v);
            GP[D_XS + fi] = (fi < nb) ? __uint_as_float(v) : 0.f;
        }
    }
    if (!ok) { if (fcode == 0u) fcode = 0x500u | 0x21u; FAIL[0] = 1; }
    __syncthreads();
    if (FAIL[0] != 0) goto bail;
    asm volatile("" ::"v"(touch));
    for (int i = 0; i < nact; ++i) {
        const float4 *GP = reinterpret_cast<const float4 *>(smem + i * DGRP);
        float4 *dst = reinterpret_cast<float4 *>(a.state + state_wg + (size_t)i * LGRP + 768);
        for (int q = tid; q < DGRP / 4; q += NT) dst[q] = GP[q];
    }
    if (PROF && tid == 0 && a.prof) {
        for (int k = 0; k < 16; ++k) a.prof[(size_t)(blockIdx.x & 255) * 32 + k] += PROFL[k];
    }
    return;
bail:
    if (fcode != 0u) report_failure(a.status, fcode, blockIdx.x, t, tid);
#undef PH
}

// ---------------------------------------------------------------------------------------------------------------------------------
// hh workgroup: gh(t+1) = W_hh . h(t) + b_hh of its 16 units for every slot (off the critical path), and -- for at most one slot --
// fc3 + mixture-of-logistics sampling (utils/distribution.py:87-123).  Keeps no state between launches.
// ---------------------------------------------------------------------------------------------------------------------------------
// PROF: as duo_ih, [gh stages: 0-7, the sampling stage: 8-15]
template <bool LA, bool PROF>
__device__ __forceinline__ void duo_hh(const LoopArgs &a, float *smem, int cl, int J, int ncl)
{
    const int G = a.G;
    const DuoLds L = duo_lds(G);
    float *PART = smem + L.off_part, *LOG = smem + L.off_log;
    int *FAIL = reinterpret_cast<int *>(smem + L.off_misc);
    int *GEO = FAIL + 16;
    u64 *PROFL = reinterpret_cast<u64 *>(smem + L.off_prof);
    u64 plast = 0;
    int cur = 0;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
#define PH(k)                                                                  \
    do {                                                                       \
        if (PROF && tid == 0) {                                                \
            const u64 now_ = __builtin_amdgcn_s_memtime();                     \
            PROFL[k] += now_ - plast;                                          \
            plast = now_;                                                      \
        }                                                                      \
    } while (0)
    const int fi = lane & 15, kq = lane >> 4;
    const int kbase_lane = KCH * w + 4 * kq;
    const int pu = 4 * w + (tid & 3), pj = (tid >> 2) & 15;
    const int prow = LU * J + pu;
    const int T0 = a.t0, T1 = a.t1, C = a.C;
    const int NR = a.Btot, Nall = a.Nall, NGR = a.NG;
    constexpr int L_H = LA ? 0 : 1, L_GH = LA ? 8 : 12, L_Y = LA ? 2 : 3, L_P2 = LA ? 6 : 2;

    float A_hh[3][AF];
#pragma unroll
    for (int g = 0; g < 3; ++g) load_afrag(A_hh[g], LA ? a.w_hh1 : a.w_hh2, H, g * H + LU * J + fi, true, kbase_lane);
#if DUO_SPLIT_FC
    float A_fc[AF];                                      // fc1 / fc2 rows of the unit block: the fc stages of the odd slots (see duo_ih)
    load_afrag(A_fc, LA ? a.fc1_w : a.fc2_w, H + AUX, LU * J + fi, true, kbase_lane);
#endif
    const float *bhh = LA ? a.b_hh1 : a.b_hh2;
    const float bh_r = bhh[prow], bh_z = bhh[H + prow], bh_n = bhh[2 * H + prow];
    const float b3a = a.fc3_b[pu];                                              // logit rows pu and 16 + pu
    const float b3b = (16 + pu < 30) ? a.fc3_b[16 + pu] : 0.f;

    for (int q = tid; q < L.total; q += NT) smem[q] = 0.f;
    __syncthreads();
    int nact = 0;
    for (int i = 0; i < G; ++i)
        if (cl + ncl * i < NGR) nact = i + 1;
    u64 nbpack = 0;                                     // segment count of slot i in byte i
    for (int i = 0; i < nact; ++i) {
        const int g = cl + ncl * i;
        const int b0 = (int)(((long)g * NR) / NGR), nb = (int)(((long)(g + 1) * NR) / NGR) - b0;
        if (tid == 0) { GEO[2 * i] = a.rb0 + b0; GEO[2 * i + 1] = nb; }
        nbpack |= (u64)(unsigned)nb << (8 * i);
        if (tid < SEG) {                                // segment table of the slot (the conditioning frame of its fc stages)
            int *SP = reinterpret_cast<int *>(smem + i * DGRP + D_SP);
            const int sc = a.rb0 + b0 + (tid < nb ? tid : nb - 1);
            SP[tid] = a.seg_pos[sc];
            SP[SEG + tid] = a.seg_lim[sc];
        }
    }
    __syncthreads();
    const bool split_fc = DUO_SPLIT_FC && (a.tuning & 128) == 0 && !DUO_IH_XAHEAD;
    // the slot this workgroup samples: slot s <-> hh role (s & 1 ? layer 2 : layer 1), unit block s >> 1
    const int my_slot = (J < LMAXG / 2) ? 2 * J + (LA ? 0 : 1) : -1;
    const bool sampler = my_slot >= 0 && my_slot < nact;

    const __amdgpu_buffer_rsrc_t xrs = make_rsrc(a.xbuf, (unsigned)(DXBUF_FLOATS * 4));
    int pp = 0;
    bool ok = true;
    unsigned fcode = 0u;
    int t = T0;
    const bool prio_mfma = (a.tuning & 8) != 0;
    if (a.tuning & 32) __builtin_amdgcn_s_setprio(1);    // A/B: static priority for the hh workgroups
    enum { BK_NONE = 0, BK_GH, BK_SAMPLE, BK_RELU };
    u32x4 x[8];
    bool xahead = false;
    int bk = BK_NONE, bi = 0, bpp = 0, bt = 0;
    float bc0 = 0.f, bc1 = 0.f;

    auto run_back = [&]() -> bool {
        if (bk == BK_NONE) return true;
        const float *PB = DPARTOF(bpp);
        const int nb = (int)((nbpack >> (8 * bi)) & 255u);
        if (!ok) FAIL[0] = 1;
        lds_barrier();
        if (FAIL[0] != 0) return false;
        PH(cur + 1);
        if (bk == BK_GH) {                              // gh(t+1) of the owned (unit, segment) -> ring entry (t + 1): consumed by the ih workgroup J at step t + 1
            const float g0 = get_partial<3>(PB, 0, pu, pj) + bh_r, g1 = get_partial<3>(PB, 1, pu, pj) + bh_z, g2 = get_partial<3>(PB, 2, pu, pj) + bh_n;
            const int r1 = (bt + 1) % DRING;
            if (pj < nb) {                              // one 16-byte word {r, z, n, tag} per (unit, segment): no quad gather, one store, one load at the reader
                const u32x4 q = {__float_as_uint(g0), __float_as_uint(g1), __float_as_uint(g2), (unsigned)bt + 2u};      // tag: consuming step (bt + 1) + 1
                __builtin_amdgcn_raw_buffer_store_b128(q, xrs, (256 * (J & 7) + tid) * 16, DXL(bi, L_GH + (J >> 3), r1) * 4, 16 /* sc1 */);
            }
        } else if (bk == BK_RELU) {                     // fc1 / fc2 + relu of an odd slot -> publish y1 / y2 (as duo_ih)
            publish4(xrs, (DXL(bi, L_Y, bt % DRING) + 256 * J) * 4, tid, fmaxf(get_partial<3>(PB, 0, pu, pj) + bc0, 0.f), pj < nb);
        } else {                                        // fc3 logits -> sample x_t (utils/distribution.py:102-121)
            const int b0 = GEO[2 * bi];
            {   // 30 logit rows x 16 segments: thread (rows pu and 16 + pu, segment pj) -- the partial tiles' conflict-free reader mapping
                const int row = pu, sj = pj;
                const float lg = get_partial<3>(PB, 0, row, sj) + b3a;
                const float lg2 = get_partial<3>(PB, 1, row, sj) + b3b;
                LOG[sj * DLOGS + row] = lg;
                if (a.dbg_logits && sj < nb) a.dbg_logits[((size_t)bt * Nall + b0 + sj) * C + row] = lg;
                if (row < 14) {
                    LOG[sj * DLOGS + 16 + row] = lg2;
                    if (a.dbg_logits && sj < nb) a.dbg_logits[((size_t)bt * Nall + b0 + sj) * C + 16 + row] = lg2;
                }
            }
            lds_barrier();
            {   // 16-lane row = one segment (su), lane sm = mixture; bc0 / bc1 = this thread's pre-transformed noise
                const int su = tid >> 4, sm = tid & 15;
                float best = (sm < 10) ? mol_gumbel_pre(LOG[su * DLOGS + sm], bc0) : -INFINITY;
                int bidx = sm;
                argmax_row16(best, bidx);
                if (sm == 0 && su < nb) {
                    float xv = mol_sample_pre(LOG[su * DLOGS + 10 + bidx], LOG[su * DLOGS + 20 + bidx], bc1);
                    a.out[(size_t)(b0 + su) * a.T + bt] = xv;
                    if (a.force_x) xv = a.force_x[(size_t)(b0 + su) * a.T + bt];
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(xv), xrs, su * 4, DXL(bi, 7, bt % DRING) * 4, 16 /* sc1 */);
                }
            }
        }
        bk = BK_NONE;
        PH(cur + 2);
        return true;
    };

    int ring = 0;
    // kind 1: gh stage of slot i (polls h(t)); kind 2: fc stage of an odd slot (polls x2(t) / y1(t)); kind 3: sampling stage of my_slot (polls y2(t))
    auto stage = [&](auto KC, int i) -> bool {
        constexpr int kind = decltype(KC)::value;
        const int nb = (int)((nbpack >> (8 * i)) & 255u);
        float v0 = 0.f, v1 = 0.f;
        float b[32];
        bool ready = false;
        cur = kind == 1 ? 0 : 8;
        if (PROF && tid == 0 && plast == 0) plast = __builtin_amdgcn_s_memtime();
#if DUO_PUBLISH_FIRST
        if (!run_back()) return false;
#endif
        constexpr int xl = kind == 1 ? L_H : (kind == 2 ? L_P2 : 3);
        if (DUO_HH_XAHEAD && xahead) ready = try_finish(lane, nb, x, b);
        else issue(xrs, DXL(i, xl, ring) * 4, w, lane, x);
        if (kind == 2) {                                // conditioning frame of (segment pj, step t): aux columns + bias of fc1 / fc2 as per-frame tables
            const int *SP = reinterpret_cast<const int *>(smem + i * DGRP + D_SP);
            const int p0 = SP[pj] + t;
            const int fr = (p0 < SP[SEG + pj]) ? (p0 / a.hop) : a.NF;
            v0 = (LA ? a.c3f : a.c4f)[(size_t)fr * H + prow];
        }
        if (kind == 3) {
            // this step's sampling noise, pre-transformed (wrnn_noise_mol_kernel): thread (segment tid >> 4, mixture tid & 15)
            const int b0 = GEO[2 * i];
            const int su = tid >> 4, sm = tid & 15;
            const float *nrow = a.noise_pre + (size_t)(t - a.noise_t0) * 11 * Nall;
            const int suc = su < nb ? su : nb - 1;
            v0 = nrow[(size_t)(b0 + suc) * 10 + (sm < 10 ? sm : 9)];
            v1 = nrow[(size_t)10 * Nall + b0 + suc];
        }
        PH(cur + 0);
#if !DUO_PUBLISH_FIRST
        if (!run_back()) return false;
#endif
        {
            unsigned spins = 0;
            if (!ready) {
                ok = ok && finish(xrs, DXL(i, xl, ring) * 4, w, lane, nb, x, b, a.status, spins);
                if (!ok && fcode == 0u) fcode = 0x600u | (LA ? 0u : 8u) | (unsigned)kind;
            }
            if (PROF && tid == 0) { PROFL[cur + 6] += 1; PROFL[cur + 7] += !ready; }
        }
        PH(cur + 3);
        if (sampler && kind == 3) {
            // ring hygiene (see the header): drain, then re-arm this wave's words of entry (t + 4) % 8: the x_t words of the slot it
            // samples (the gh words carry a step tag instead of relying on a sentinel: nothing to re-arm)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const int ringn = (t + DAHEAD) % DRING;
            if (lane == 48) {                // the 4 x_t words this wave publishes (segments 4 w ..)
                const u32x4 q = {SENT, SENT, SENT, SENT};
                __builtin_amdgcn_raw_buffer_store_b128(q, xrs, 16 * w, DXL(my_slot, 7, ringn) * 4, 16 /* sc1 */);
            }
        }
        {   // the next stage's polled layer, one stage ahead (not across a step boundary)
            xahead = false;
            if (DUO_HH_XAHEAD && (a.tuning & 1) == 0 && kind != 3) {
                int nk = 0, ni = 0;                     // the stage that follows in the step: gh of every slot, fc of the odd ones, sampling
                if (kind == 1 && i + 1 < nact) { nk = 1; ni = i + 1; }
                else if (kind == 1 && split_fc && nact >= 2) { nk = 2; ni = 1; }
                else if (kind == 2 && i + 2 < nact) { nk = 2; ni = i + 2; }
                else if (sampler) { nk = 3; ni = my_slot; }
                if (nk != 0) {
                    xahead = true;
                    issue(xrs, DXL(ni, nk == 1 ? L_H : (nk == 2 ? L_P2 : 3), ring) * 4, w, lane, x);
                }
            }
        }
        PH(cur + 4);
        float *PW = DPARTOF(pp);
        if (prio_mfma) __builtin_amdgcn_s_setprio(1);
        if (kind == 1) {
            f32x4 o0, o1, o2;
            mfma3(A_hh[0], A_hh[1], A_hh[2], b, o0, o1, o2);
            put_partial<3>(PW, w, 0, lane, o0);
            put_partial<3>(PW, w, 1, lane, o1);
            put_partial<3>(PW, w, 2, lane, o2);
            bk = BK_GH;
        } else if (kind == 2) {
#if DUO_SPLIT_FC
            put_partial<3>(PW, w, 0, lane, mfma1(A_fc, b));
#endif
            bk = BK_RELU;
        } else {
            put_partial<3>(PW, w, 0, lane, mfma1_glb(a.fc3f + frag_off(w, 0, lane), b));
            put_partial<3>(PW, w, 1, lane, mfma1_glb(a.fc3f + XT + frag_off(w, 0, lane), b));
            bk = BK_SAMPLE;
        }
        if (prio_mfma) __builtin_amdgcn_s_setprio(0);
        PH(cur + 5);
        bi = i; bpp = pp; bt = t; bc0 = v0; bc1 = v1;
        pp ^= 1;
        return true;
    };

    for (; t < T1; ++t) {
        ring = t % DRING;
#pragma unroll 1
        for (int i = 0; i < nact; ++i)
            if (!stage(std::integral_constant<int, 1>{}, i)) goto bail;
        if (split_fc) {
#pragma unroll 1
            for (int i = 1; i < nact; i += 2)
                if (!stage(std::integral_constant<int, 2>{}, i)) goto bail;
        }
        if (sampler) {
            if (!stage(std::integral_constant<int, 3>{}, my_slot)) goto bail;
        }
    }
    if (!run_back()) goto bail;
    if (PROF && tid == 0 && a.prof) {
        for (int k = 0; k < 16; ++k) a.prof[(size_t)(blockIdx.x & 255) * 32 + 16 + k] += PROFL[k];
    }
    return;
bail:
    if (fcode != 0u) report_failure(a.status, fcode, blockIdx.x, t, tid);
#undef PH
}
#undef DXL
#undef DPARTOF

// Grid = clusters x 128 workgroups of 256 threads (two per CU), cooperative launch.  Workgroup wg of a cluster: role wg & 3, unit
// block wg >> 2.  Whole XCDs per cluster (speed only: nothing depends on the placement).
template <bool PROF>
__global__ __launch_bounds__(NT, 2) void wrnn_duo_kernel(const LoopArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int cl, wg;
    const int ncl = gridDim.x / DNWGC;
    {
        const int b = blockIdx.x, nblk = gridDim.x;
        if (nblk % 8 == 0 && ncl >= 1 && 8 % ncl == 0) {
            const int xpc = 8 / ncl, per_xcd = nblk / 8;
            const int xcd = b % 8;
            cl = xcd / xpc;
            wg = (xcd % xpc) * per_xcd + b / 8;
        } else {
            cl = b / DNWGC;
            wg = b % DNWGC;
        }
    }
    // role / unit block of workgroup wg.  Speed only: blocks are observed to be dealt round-robin over the CUs of an XCD, i.e. (with
    // 64 blocks per XCD on 32 CUs) local blocks q and q + 32 share a CU.  The pairing below puts an ih workgroup (128 MFMAs per
    // group-step and wave) next to the hh workgroup of the same layer and unit block (96): every SIMD then carries 224 MFMAs per
    // group-step, and one wave's tiles can run under the other's pointwise / load / barrier time.  (wg & 3 as the role would
    // pair two workgroups of the SAME role on every CU.)
    int role, J;
    {
        const int half = DNWGC / 2;                     // 64: the workgroups of one XCD when a cluster spans two
        const int q = wg % half, xh = wg / half;
        role = ((q >> 5) << 1) | (q & 1);               // second half of an XCD's blocks: hh; odd: layer 2
        J = xh * (LNJ / 2) + ((q & 31) >> 1);
    }
    if (!PROF && a.prof && threadIdx.x == 0 && blockIdx.x < 2048) {   // placement read-out (test / profiling hook): HW_ID and XCC_ID of the block
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        a.prof[blockIdx.x] = ((u64)xcc << 32) | hw | ((u64)(unsigned)(role | (J << 2) | (cl << 8)) << 40);
    }
    if (role == 0) duo_ih<true, PROF>(a, smem, cl, J, ncl);
    else if (role == 1) duo_ih<false, PROF>(a, smem, cl, J, ncl);
    else if (role == 2) duo_hh<true, PROF>(a, smem, cl, J, ncl);
    else duo_hh<false, PROF>(a, smem, cl, J, ncl);
}

size_t duo_lds_bytes(int G) { return (size_t)duo_lds(G).total * sizeof(float); }
size_t duo_xbuf_bytes(int G) { return (size_t)G * MAXCL * DNX * DRING * XT * sizeof(float); }     // the slots a launch with depth G touches: a prefix
size_t duo_xbuf_bytes_max() { return DXBUF_FLOATS * sizeof(float); }
int duo_max_depth() { return LMAXG; }

// clusters this device can host at two workgroups per CU (64 CUs per cluster, as wrnn_loop_kernel)
int duo_clusters(int n_cus)
{
    int ncl = n_cus / LNWGC;
    if (ncl > MAXCL) ncl = MAXCL;
    while (ncl > 1 && (8 % ncl) != 0) --ncl;
    return ncl;
}

hipError_t launch_duo(const LoopArgs &args, int ncl, hipStream_t stream)
{
    if (ncl < 1 || args.G < 1 || args.G > LMAXG || !args.fc3f) return hipErrorInvalidValue;
    const size_t lds = duo_lds_bytes(args.G);
    // phase clocks (wrnn_options.phase_clocks) unless tuning bit 6 asks for the placement read-out through the same buffer
    const void *fn = (args.prof && !(args.tuning & 64)) ? (const void *)wrnn_duo_kernel<true> : (const void *)wrnn_duo_kernel<false>;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    LoopArgs a = args;
    void *params[] = {(void *)&a};
    return hipLaunchCooperativeKernel(fn, dim3(ncl * DNWGC), dim3(NT), params, (unsigned)lds, stream);
}

}  // namespace wrnn
