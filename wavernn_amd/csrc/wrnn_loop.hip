// wrnn_loop.hip -- the ROLE-SPLIT pipelined persistent WaveRNN loop kernel (MOL and RAW) for MI355X (gfx950 / CDNA4).
//
// Replaces the `for i in range(seq_len)` loop of fatchord/WaveRNN `WaveRNN.generate()` (reference
// models/fatchord_version.py:201-241) for one ROUND of folded segments (<= clusters x G groups of <= 16) over a range of
// steps [t0, t1).  Second generation of the clustered design (wrnn_cluster.hip / wrnn_pipe.hip); what changed and why
// (profiles/r01t_summary.md: MFMA busy 25.7 %, 30 % of the issued MFMA rows were padding, sweeps 30 % of a group-step):
//
//   * ROLES instead of padding.  A cluster is still 64 workgroups holding one fp32 copy of the loop weights in registers,
//     but workgroup wg = 2 J + role owns SIXTEEN hidden units [16 J, 16 J + 16) of ONE recurrent layer:
//         role A: rnn1 (W_ih 3 gate tiles + W_hh 3 gate tiles) + the same 16 rows of fc1,
//         role B: rnn2 (3 + 3 tiles)                            + the same 16 rows of fc2,
//     so every v_mfma_f32_16x16x4_f32 tile is 16 live rows (gate g of the 16 owned units; 7 tiles per workgroup against
//     10 two-thirds-full ones before; 224 weight registers, which fit the AGPR half of the register file -- 9 tiles did not
//     and spilled the B operands).  fc3 lives in LDS in A-fragment order: MOL -- every workgroup holds all 30 rows (2 tiles)
//     and samples redundantly, no 5th exchange; RAW -- workgroup wg owns logit rows [8 wg, 8 wg + 8) and the 512 logits are
//     a 5th exchange.
//   * TAG-FREE exchange in MFMA-FRAGMENT ORDER.  A layer (h1, h2, y1, y2, RAW logits) of a group is 16 segments x 512 f32 =
//     32 KB stored as [wave w][k-block r][lane][4]: exactly the B fragments wave w feeds its MFMAs, so a consumer wave
//     loads its operands with 8 buffer_load_dwordx4 (sc1) straight into registers -- no LDS staging, no sweep barrier, half
//     the bytes of the 8-byte {tag, value} granules.  Arrival is detected by value: every slot is pre-filled with a
//     sentinel NaN pattern (0xFFFFFFFF, which no finite activation takes) and a consumer re-reads until none of its 32
//     words is the sentinel.  Slots form a ring of 3 by step; a producer re-arms its own part of slot (t+1) % 3 at the top
//     of step t, by which time every consumer is provably done with step t-2, and drains its stores (s_waitcnt vmcnt(0))
//     before it publishes anything of step t, so the re-arm is visible before any poll of step t+1 can start.
//     A producer's 16 units x 16 segments are one contiguous 1 KB block of the layer: 64 lanes x 16-byte sc1 stores.
//   * no activation tile in LDS at all: the residual sums x1 = xi + h1, x2 = x1 + h2 (:212,:216) are re-formed in registers
//     from the conditioning slab and the ring slots (still valid: re-armed two steps later), each wave touching only its own
//     K chunk, so the only workgroup barrier of a stage is the one between writing and summing the four waves' partial
//     tiles; partial buffers ping-pong.  5 barriers per group-step (was 14); 4.3 KB of LDS per group in flight.
//   * the hoisted I-layer conditioning cI arrives in fragment order and in SLABS of a few hundred steps (wrnn_cond.hip):
//     the host loops slabs x rounds, the kernel saves / restores its per-group state, the workspace no longer scales with T.
// Arithmetic and summation order per output are those of wrnn_cluster.hip / wrnn_pipe.hip (K split over 4 waves, two
// accumulator chains per tile, partials added in wave order), so results are bit-identical to those kernels.
#include "wrnn_tiles.h"

namespace wrnn {

constexpr int LU = 16;                       // hidden units per workgroup
constexpr int LNJ = H / LU;                  // workgroups per role per cluster (32)
constexpr int LNWGC = 2 * LNJ;               // workgroups per cluster (64)
constexpr int XT = SEG * H;                  // floats of one layer in fragment order (8192)
constexpr unsigned SENT = 0xFFFFFFFFu;       // "not written yet"

// LDS carve (floats).  Per group: GH[3][256], HOWN[256], XS[16], SP[32 ints]
constexpr int LGRP = 3 * 256 + 256 + 16 + 32;
constexpr int O_HOWN = 768, O_XS = 1024, O_SP = 1040;
constexpr int LPART = 2 * NW * 3 * 256;      // two ping-pong sets of [wave][slot 0..2][16][16]
struct LoopLds {
    int off_part, off_log, off_wi0, off_f3, off_lgt, off_misc, total;
};
__host__ __device__ inline LoopLds loop_lds(int mode, int G)
{
    LoopLds l;
    int o = G * LGRP;
    l.off_part = o; o += LPART;
    l.off_log = o;  o += SEG * 32;
    l.off_wi0 = o;  o += H;
    l.off_f3 = o;   o += (mode == 1 ? 2 : 1) * XT;         // fc3 in A-fragment order [tile][wave][r][lane][4]
    l.off_lgt = o;  o += (mode == 0) ? SEG * LDC : 0;      // RAW: the gathered logits as [segment][class] rows
    l.off_misc = o; o += 16;                               // [0] failure flag
    l.total = o;
    return l;
}

// fragment-order offset (floats) of (wave w, k-block r, lane): 4 consecutive k of one segment
__device__ __forceinline__ int frag_off(int w, int r, int lane) { return ((w * 8 + r) * 64 + lane) * 4; }

// Load this wave's 8 B fragments of one exchanged layer (byte offset soff in the exchange buffer) until no word is the
// sentinel.  Lanes of segments >= nb are not waited for and read as zero.  Wave-uniform result; bounded spin.
__device__ __forceinline__ bool consume(__amdgpu_buffer_rsrc_t rs, int soff, int w, int lane, int nb, float (&b)[32], unsigned *status)
{
    const int voff = frag_off(w, 0, lane) * 4;
    const bool live = (lane & 15) < nb;
    unsigned spins = 0;
    u32x4 x[8];
    for (;;) {
#pragma unroll
        for (int r = 0; r < 8; ++r) x[r] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + r * 1024, soff, 16 /* sc1 */);
        bool ok = true;
#pragma unroll
        for (int r = 0; r < 8; ++r) ok &= (x[r].x != SENT) & (x[r].y != SENT) & (x[r].z != SENT) & (x[r].w != SENT);
        if (__all(ok || !live)) break;
        ++spins;
        if ((spins & 255u) == 0u) {
            if (spins > SPIN_LIMIT || ld_agent32(status) != 0u) return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        b[4 * r + 0] = live ? __uint_as_float(x[r].x) : 0.f;
        b[4 * r + 1] = live ? __uint_as_float(x[r].y) : 0.f;
        b[4 * r + 2] = live ? __uint_as_float(x[r].z) : 0.f;
        b[4 * r + 3] = live ? __uint_as_float(x[r].w) : 0.f;
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
    // every earlier store of this wave (the re-arm of this block's next ring slot) has left before the data goes out
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (on && (tid & 3) == 0) {
        const u32x4 q = {(unsigned)v0, (unsigned)v1, (unsigned)v2, (unsigned)v3};
        __builtin_amdgcn_raw_buffer_store_b128(q, rs, (tid & ~3) * 4, soff, 16 /* sc1 */);
    }
}

// Re-arm (fill with the sentinel) THIS WAVE's quarter (64 floats = 16 lanes x 16 bytes) of the workgroup's block in up to
// three layers of a ring slot: lanes 0-15 layer la, 16-31 layer lb, 32-47 layer lc (lc < 0: none).  Each wave re-arms exactly
// the words it later publishes, so its own program order + the vmcnt(0) drain in publish4 order re-arm before data.
__device__ __forceinline__ void rearm(__amdgpu_buffer_rsrc_t rs, int soff_slot0 /* bytes: layer 0 of the slot + block + quarter */,
                                      int lane, int la, int lb, int lc)
{
    const int which = lane >> 4;
    const int layer = which == 0 ? la : (which == 1 ? lb : lc);
    if (which < 3 && layer >= 0) {
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

// this wave's 8 fragments of a layer that is known to be complete (it was consumed earlier in the step): no check; sc1, so
// never served from this CU's L1 (which may still hold the slot's lines of three steps ago)
__device__ __forceinline__ void reload(__amdgpu_buffer_rsrc_t rs, int soff, int w, int lane, u32x4 (&x)[8])
{
    const int voff = frag_off(w, 0, lane) * 4;
#pragma unroll
    for (int r = 0; r < 8; ++r) x[r] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + r * 1024, soff, 16 /* sc1 */);
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

// The whole life of one workgroup in one role (compile-time, so the two roles are disjoint code with separate register
// allocations).  MODE: 0 RAW (C == 512), 1 MOL (C == 30).
template <int MODE, bool roleA>
__device__ __forceinline__ void loop_role(const LoopArgs &a, float *smem, int cl, int wg, int ncl)
{
    constexpr bool MOL = MODE == 1;
    const int G = a.G;
    const LoopLds L = loop_lds(MODE, G);
    float *PART = smem + L.off_part, *LOG = smem + L.off_log, *WI0 = smem + L.off_wi0, *F3 = smem + L.off_f3, *LGT = smem + L.off_lgt;
    int *FAIL = reinterpret_cast<int *>(smem + L.off_misc);

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int J = wg >> 1;                              // owned units [16 J, 16 J + 16)
    const int fi = lane & 15, kq = lane >> 4;
    const int kbase_lane = KCH * w + 4 * kq;
    const int pu = 4 * w + (tid & 3), pj = (tid >> 2) & 15;   // pointwise role: (owned unit, segment)
    const int prow = LU * J + pu;                       // hidden index of the pointwise role
    const int T0 = a.t0, T1 = a.t1, C = a.C;
    const int NR = a.Btot, Nall = a.Nall;               // segments of this round / of the whole call (row stride of noise, logits)
    const int NGR = a.NG;                               // groups of this round
    const bool leader = wg == 0;
    const int f3half = wg & 1;                          // RAW: logit rows [8 wg, 8 wg + 8) = units 8 f3half .. + 8 of block J

    // ---- one-time: weight slice -> register-resident MFMA A fragments (7 tiles = 224 registers) ------------------------
    float A_ih[3][AF], A_hh[3][AF], A_fc[AF];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        const int grow = g * H + LU * J + fi;
        load_afrag(A_ih[g], roleA ? a.w_ih1 : a.w_ih2, roleA ? H : H + AUX, grow, true, kbase_lane);
        load_afrag(A_hh[g], roleA ? a.w_hh1 : a.w_hh2, H, grow, true, kbase_lane);
    }
    load_afrag(A_fc, roleA ? a.fc1_w : a.fc2_w, H + AUX, LU * J + fi, true, kbase_lane);
    // per-thread constants of the pointwise role
    float bi_r = 0.f, bi_z = 0.f, bi_n = 0.f;           // role A: b_ih1 (role B's b_ih2 is inside c2f)
    if constexpr (roleA) { bi_r = a.b_ih1[prow]; bi_z = a.b_ih1[H + prow]; bi_n = a.b_ih1[2 * H + prow]; }
    const float *bhh = roleA ? a.b_hh1 : a.b_hh2;
    const float bh_r = bhh[prow], bh_z = bhh[H + prow], bh_n = bhh[2 * H + prow];
    const float b3a = MOL ? a.fc3_b[tid >> 4] : a.fc3_b[prow];                       // MOL: logit row tid >> 4; RAW: row of the pointwise role
    const float b3b = (MOL && 16 + (tid >> 4) < 30) ? a.fc3_b[16 + (tid >> 4)] : 0.f;

    for (int q = tid; q < L.total; q += NT) smem[q] = 0.f;
    __syncthreads();
    WI0[2 * tid] = a.I_w0[2 * tid];
    WI0[2 * tid + 1] = a.I_w0[2 * tid + 1];
    // fc3 -> LDS in A-fragment order: F3[tile s][wave][r][lane (row fi, k-quad kq)][4] = fc3_w[row(s, fi)][128 wave + 16 r + 4 kq ..]
    for (int q = tid; q < (MOL ? 2 : 1) * (XT / 4); q += NT) {
        const int l6 = q & 63, r = (q >> 6) & 7, wv = (q >> 9) & 3, sidx = q >> 11;
        const int rfi = l6 & 15, rkq = l6 >> 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        int row = -1;
        if (MOL) { if (16 * sidx + rfi < 30) row = 16 * sidx + rfi; }
        else { if ((rfi >> 3) == f3half) row = LU * J + rfi; }
        if (row >= 0) v = *reinterpret_cast<const float4 *>(a.fc3_w + (size_t)row * H + KCH * wv + 16 * r + 4 * rkq);
        reinterpret_cast<float4 *>(F3)[q] = v;
    }

    // ---- groups of this round run by this cluster: slot i <-> group cl + ncl * i ---------------------------------------
    int nact = 0;
    for (int i = 0; i < G; ++i)
        if (cl + ncl * i < NGR) nact = i + 1;
    const size_t state_wg = ((size_t)(cl * LNWGC + wg) * G) * LGRP;
    for (int i = 0; i < nact; ++i) {
        float *GP = smem + i * LGRP;
        const int g = cl + ncl * i;
        const int b0 = (int)(((long)g * NR) / NGR), nb = (int)(((long)(g + 1) * NR) / NGR) - b0;
        if (a.resume) {
            const float4 *src = reinterpret_cast<const float4 *>(a.state + state_wg + (size_t)i * LGRP);
            for (int q = tid; q < LGRP / 4; q += NT) reinterpret_cast<float4 *>(GP)[q] = src[q];
        } else {
            // fatchord_version.py:194-196: h1 = h2 = 0, x = 0  =>  gh = W_hh . 0 + b_hh = b_hh
            GP[tid] = bh_r; GP[256 + tid] = bh_z; GP[512 + tid] = bh_n;
            GP[O_HOWN + tid] = 0.f;
            if (tid < SEG) {
                GP[O_XS + tid] = 0.f;
                int *SP = reinterpret_cast<int *>(GP + O_SP);
                const int sc = a.rb0 + b0 + (tid < nb ? tid : nb - 1);
                SP[tid] = a.seg_pos[sc];
                SP[SEG + tid] = a.seg_lim[sc];
            }
        }
    }
    __syncthreads();

    const __amdgpu_buffer_rsrc_t xrs = make_rsrc(a.xbuf, (unsigned)(XBUF_FLOATS * 4));
    // exchange layout: [cluster][slot][layer 0..4 = h1 h2 y1 y2 lg][ring 0..2][XT floats]
    const int xcl = cl * LMAXG;
    float touch = 0.f;
    int pp = 0;                                          // partial-tile ping-pong
    bool ok = true;
    unsigned fcode = 0u;                                 // phase whose poll gave up (reported on the exit path)
    int t = T0;

#define XLAYER(i, layer, ring) ((((xcl + (i)) * NXLAYER + (layer)) * XRING + (ring)) * XT)
#define STAGE_BARRIER()                                        \
    do {                                                       \
        if (!ok) FAIL[0] = 1;                                  \
        __syncthreads();                                       \
        if (FAIL[0] != 0) goto bail;                           \
    } while (0)
#define PARTP (PART + pp * (NW * 3 * 256))
#define GROUP_NB(g) ((int)(((long)((g) + 1) * NR) / NGR) - (int)(((long)(g) * NR) / NGR))

    for (; t < T1; ++t) {
        const int ring = t % XRING, ringn = (t + 1) % XRING;
        const int tc = t - a.cI_t0;                     // row of the conditioning slab

        // ---- re-arm this wave's words of the NEXT step's ring slot (every consumer is done with step t-2) ---------------
#pragma unroll 1
        for (int i = 0; i < nact; ++i)
            rearm(xrs, (XLAYER(i, 0, ringn) + 256 * J + 64 * w) * 4, lane, roleA ? 0 : 1, roleA ? 2 : 3,
                  (!MOL && (w >> 1) == f3half) ? 4 : -1);
        __syncthreads();                                 // x_{t-1} of the last group sampled in the previous step is visible

        // =========================== P1 (role A): rnn1 gates (fatchord_version.py:208-210) =======================
        if constexpr (roleA) {
#pragma unroll 1
            for (int i = 0; i < nact; ++i) {
                float *GP = smem + i * LGRP;
                const int g = cl + ncl * i;
                const int nb = GROUP_NB(g);
                float b[32];
                {
                    float4 c[8];
                    load_cI(a.cIf + ((size_t)tc * NGR + g) * XT, w, lane, c);
                    make_xi(c, WI0, GP[O_XS + fi], w, lane, b);
                }
                f32x4 o0, o1, o2;
                mfma3(A_ih[0], A_ih[1], A_ih[2], b, o0, o1, o2);
                put_partial<3>(PARTP, w, 0, lane, o0);
                put_partial<3>(PARTP, w, 1, lane, o1);
                put_partial<3>(PARTP, w, 2, lane, o2);
                STAGE_BARRIER();
                {
                    const float gir = get_partial<3>(PARTP, 0, pu, pj) + bi_r;
                    const float giz = get_partial<3>(PARTP, 1, pu, pj) + bi_z;
                    const float gin = get_partial<3>(PARTP, 2, pu, pj) + bi_n;
                    const float hn = gru_update(gir, giz, gin, GP[tid], GP[256 + tid], GP[512 + tid], GP[O_HOWN + tid]);
                    GP[O_HOWN + tid] = hn;
                    publish4(xrs, (XLAYER(i, 0, ring) + 256 * J) * 4, tid, hn, pj < nb);
                }
                pp ^= 1;
            }
        }

        // =========================== P2: h1 arrives.  A: gh1(t+1) = W_hh1 . h1.  B: rnn2 gates on x1 = xi + h1 (:212-214) =====
#pragma unroll 1
        for (int i = 0; i < nact; ++i) {
            float *GP = smem + i * LGRP;
            const int g = cl + ncl * i;
            const int nb = GROUP_NB(g);
            float b[32];
            if constexpr (roleA) {
                ok = ok && consume(xrs, XLAYER(i, 0, ring) * 4, w, lane, nb, b, a.status);
                if (!ok && fcode == 0u) fcode = 0x400u | 1u;
                f32x4 o0, o1, o2;
                mfma3(A_hh[0], A_hh[1], A_hh[2], b, o0, o1, o2);
                put_partial<3>(PARTP, w, 0, lane, o0);
                put_partial<3>(PARTP, w, 1, lane, o1);
                put_partial<3>(PARTP, w, 2, lane, o2);
                STAGE_BARRIER();
                GP[tid] = get_partial<3>(PARTP, 0, pu, pj) + bh_r;
                GP[256 + tid] = get_partial<3>(PARTP, 1, pu, pj) + bh_z;
                GP[512 + tid] = get_partial<3>(PARTP, 2, pu, pj) + bh_n;
            } else {
                // aux columns of rnn2 + b_ih2: per-frame table (unconditional loads, issued before the poll)
                const int *SP = reinterpret_cast<const int *>(GP + O_SP);
                const int p = SP[pj] + t;
                const int f2 = (p < SP[SEG + pj]) ? (p / a.hop) : a.NF;
                const float c2r = a.c2f[(size_t)f2 * 3 * H + prow];
                const float c2z = a.c2f[(size_t)f2 * 3 * H + H + prow];
                const float c2n = a.c2f[(size_t)f2 * 3 * H + 2 * H + prow];
                float4 c[8];
                load_cI(a.cIf + ((size_t)tc * NGR + g) * XT, w, lane, c);
                ok = ok && consume(xrs, XLAYER(i, 0, ring) * 4, w, lane, nb, b, a.status);
                if (!ok && fcode == 0u) fcode = 0x400u | 2u;
                {   // x1 = xi + h1 (:212)
                    float xi[32];
                    make_xi(c, WI0, GP[O_XS + fi], w, lane, xi);
#pragma unroll
                    for (int q = 0; q < 32; ++q) b[q] = xi[q] + b[q];
                }
                f32x4 o0, o1, o2;
                mfma3(A_ih[0], A_ih[1], A_ih[2], b, o0, o1, o2);
                put_partial<3>(PARTP, w, 0, lane, o0);
                put_partial<3>(PARTP, w, 1, lane, o1);
                put_partial<3>(PARTP, w, 2, lane, o2);
                STAGE_BARRIER();
                {
                    const float gir = get_partial<3>(PARTP, 0, pu, pj) + c2r;
                    const float giz = get_partial<3>(PARTP, 1, pu, pj) + c2z;
                    const float gin = get_partial<3>(PARTP, 2, pu, pj) + c2n;
                    const float hn = gru_update(gir, giz, gin, GP[tid], GP[256 + tid], GP[512 + tid], GP[O_HOWN + tid]);
                    GP[O_HOWN + tid] = hn;
                    publish4(xrs, (XLAYER(i, 1, ring) + 256 * J) * 4, tid, hn, pj < nb);
                }
            }
            pp ^= 1;
        }

        // =========================== P3: h2 arrives.  A: fc1 + relu on x2 = x1 + h2 (:216-218).  B: gh2(t+1) = W_hh2 . h2 =====
#pragma unroll 1
        for (int i = 0; i < nact; ++i) {
            float *GP = smem + i * LGRP;
            const int g = cl + ncl * i;
            const int nb = GROUP_NB(g);
            float b[32];
            if constexpr (roleA) {
                const int *SP = reinterpret_cast<const int *>(GP + O_SP);
                const int p = SP[pj] + t;
                const int f3 = (p < SP[SEG + pj]) ? (p / a.hop) : a.NF;
                const float c3v = a.c3f[(size_t)f3 * H + prow];
                // x1 is re-formed from the conditioning slab and the h1 slot (both complete; issued ahead of the h2 poll)
                float4 c[8];
                u32x4 h1[8];
                load_cI(a.cIf + ((size_t)tc * NGR + g) * XT, w, lane, c);
                reload(xrs, XLAYER(i, 0, ring) * 4, w, lane, h1);
                ok = ok && consume(xrs, XLAYER(i, 1, ring) * 4, w, lane, nb, b, a.status);
                if (!ok && fcode == 0u) fcode = 0x400u | 3u;
                {
                    float xi[32];
                    make_xi(c, WI0, GP[O_XS + fi], w, lane, xi);
                    const bool live = fi < nb;
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        const float h10 = live ? __uint_as_float(h1[r].x) : 0.f, h11 = live ? __uint_as_float(h1[r].y) : 0.f;
                        const float h12 = live ? __uint_as_float(h1[r].z) : 0.f, h13 = live ? __uint_as_float(h1[r].w) : 0.f;
                        b[4 * r + 0] = (xi[4 * r + 0] + h10) + b[4 * r + 0];      // (xi + h1) + h2, the reference's order
                        b[4 * r + 1] = (xi[4 * r + 1] + h11) + b[4 * r + 1];
                        b[4 * r + 2] = (xi[4 * r + 2] + h12) + b[4 * r + 2];
                        b[4 * r + 3] = (xi[4 * r + 3] + h13) + b[4 * r + 3];
                    }
                }
                put_partial<3>(PARTP, w, 0, lane, mfma1(A_fc, b));
                STAGE_BARRIER();
                publish4(xrs, (XLAYER(i, 2, ring) + 256 * J) * 4, tid, fmaxf(get_partial<3>(PARTP, 0, pu, pj) + c3v, 0.f), pj < nb);
            } else {
                ok = ok && consume(xrs, XLAYER(i, 1, ring) * 4, w, lane, nb, b, a.status);
                if (!ok && fcode == 0u) fcode = 0x400u | 4u;
                f32x4 o0, o1, o2;
                mfma3(A_hh[0], A_hh[1], A_hh[2], b, o0, o1, o2);
                put_partial<3>(PARTP, w, 0, lane, o0);
                put_partial<3>(PARTP, w, 1, lane, o1);
                put_partial<3>(PARTP, w, 2, lane, o2);
                STAGE_BARRIER();
                GP[tid] = get_partial<3>(PARTP, 0, pu, pj) + bh_r;
                GP[256 + tid] = get_partial<3>(PARTP, 1, pu, pj) + bh_z;
                GP[512 + tid] = get_partial<3>(PARTP, 2, pu, pj) + bh_n;
            }
            pp ^= 1;
        }

        // =========================== P4 (role B): y1 arrives; fc2 + relu (:220-221) ==============================
        if constexpr (!roleA) {
#pragma unroll 1
            for (int i = 0; i < nact; ++i) {
                float *GP = smem + i * LGRP;
                const int g = cl + ncl * i;
                const int nb = GROUP_NB(g);
                const int *SP = reinterpret_cast<const int *>(GP + O_SP);
                const int p = SP[pj] + t;
                const int f4 = (p < SP[SEG + pj]) ? (p / a.hop) : a.NF;
                const float c4v = a.c4f[(size_t)f4 * H + prow];
                float b[32];
                ok = ok && consume(xrs, XLAYER(i, 2, ring) * 4, w, lane, nb, b, a.status);
                if (!ok && fcode == 0u) fcode = 0x400u | 5u;
                put_partial<3>(PARTP, w, 0, lane, mfma1(A_fc, b));
                STAGE_BARRIER();
                publish4(xrs, (XLAYER(i, 3, ring) + 256 * J) * 4, tid, fmaxf(get_partial<3>(PARTP, 0, pu, pj) + c4v, 0.f), pj < nb);
                pp ^= 1;
            }
        }

        // =========================== P5 (both roles): y2 arrives; fc3 (:223) + sampling (:225-237) ===================
#pragma unroll 1
        for (int i = 0; i < nact; ++i) {
            float *GP = smem + i * LGRP;
            float *XS = GP + O_XS;
            const int g = cl + ncl * i;
            const int nb = GROUP_NB(g);
            const int b0 = a.rb0 + (int)(((long)g * NR) / NGR);           // first segment of the group in the call's segment table
            const size_t tn = (size_t)(t - a.noise_t0);
            float b[32];
            if constexpr (MOL) {
                // this step's sampling noise, pre-transformed (wrnn_noise_mol_kernel): thread (segment tid >> 4, mixture tid & 15)
                const int su = tid >> 4, sm = tid & 15;
                const float *nrow = a.noise_pre + tn * 11 * Nall;
                const int suc = su < nb ? su : nb - 1;
                const float nz0 = nrow[(size_t)(b0 + suc) * 10 + (sm < 10 ? sm : 9)];
                const float nz1 = nrow[(size_t)10 * Nall + b0 + suc];
                ok = ok && consume(xrs, XLAYER(i, 3, ring) * 4, w, lane, nb, b, a.status);
                if (!ok && fcode == 0u) fcode = 0x400u | 6u;
                put_partial<3>(PARTP, w, 0, lane, mfma1_lds(F3 + frag_off(w, 0, lane), b));
                put_partial<3>(PARTP, w, 1, lane, mfma1_lds(F3 + XT + frag_off(w, 0, lane), b));
                STAGE_BARRIER();
                {   // 30 logit rows x 16 segments: thread (row tid >> 4 and 16 + row, segment tid & 15)
                    const int row = tid >> 4, sj = tid & 15;
                    const float lg = get_partial<3>(PARTP, 0, row, sj) + b3a;
                    LOG[sj * 32 + row] = lg;
                    if (a.dbg_logits && leader && sj < nb) a.dbg_logits[((size_t)t * Nall + b0 + sj) * C + row] = lg;
                    if (row < 14) {
                        const float lg2 = get_partial<3>(PARTP, 1, row, sj) + b3b;
                        LOG[sj * 32 + 16 + row] = lg2;
                        if (a.dbg_logits && leader && sj < nb) a.dbg_logits[((size_t)t * Nall + b0 + sj) * C + 16 + row] = lg2;
                    }
                }
                __syncthreads();
                {   // utils/distribution.py:102-121: 16-lane row = one segment (su), lane sm = mixture
                    float best = (sm < 10) ? mol_gumbel_pre(LOG[su * 32 + sm], nz0) : -INFINITY;
                    int bidx = sm;
                    argmax_row16(best, bidx);
                    if (sm == 0 && su < nb) {
                        float x = mol_sample_pre(LOG[su * 32 + 10 + bidx], LOG[su * 32 + 20 + bidx], nz1);
                        if (leader) a.out[(size_t)(b0 + su) * a.T + t] = x;
                        if (a.force_x) x = a.force_x[(size_t)(b0 + su) * a.T + t];
                        XS[su] = x;
                    }
                }
            } else {
                ok = ok && consume(xrs, XLAYER(i, 3, ring) * 4, w, lane, nb, b, a.status);
                if (!ok && fcode == 0u) fcode = 0x400u | 6u;
                put_partial<3>(PARTP, w, 0, lane, mfma1_lds(F3 + frag_off(w, 0, lane), b));
                STAGE_BARRIER();
                publish4(xrs, (XLAYER(i, 4, ring) + 256 * J) * 4, tid, get_partial<3>(PARTP, 0, pu, pj) + b3a, (pj < nb) && ((pu >> 3) == f3half));
                // the 512 logits of every segment -> LGT [segment][class]
                ok = ok && consume(xrs, XLAYER(i, 4, ring) * 4, w, lane, nb, b, a.status);
                if (!ok && fcode == 0u) fcode = 0x400u | 7u;
                {
                    float *lp = LGT + fi * LDC + kbase_lane;
#pragma unroll
                    for (int r = 0; r < 8; ++r) *reinterpret_cast<float4 *>(lp + 16 * r) = make_float4(b[4 * r], b[4 * r + 1], b[4 * r + 2], b[4 * r + 3]);
                }
                STAGE_BARRIER();
                // fatchord_version.py:232-237: softmax -> Categorical (renormalise) -> argmax(p / q); one wave per 4 segments
#pragma unroll 1
                for (int s = 0; s < 4; ++s) {
                    const int sj = 4 * w + s;
                    if (sj < nb) {                                           // wave-uniform
                        float lg[8], qn[8];
                        float mx = -INFINITY;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            qn[e] = a.noise[(tn * Nall + b0 + sj) * C + lane + 64 * e];
                            lg[e] = LGT[sj * LDC + lane + 64 * e];
                            if (a.dbg_logits && leader) a.dbg_logits[((size_t)t * Nall + b0 + sj) * C + lane + 64 * e] = lg[e];
                            mx = fmaxf(mx, lg[e]);
                        }
#pragma unroll
                        for (int m = 32; m >= 1; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m, 64));
                        float sum = 0.f;
#pragma unroll
                        for (int e = 0; e < 8; ++e) { lg[e] = expf(lg[e] - mx); sum += lg[e]; }
#pragma unroll
                        for (int m = 32; m >= 1; m >>= 1) sum += __shfl_xor(sum, m, 64);
                        float sum2 = 0.f;
#pragma unroll
                        for (int e = 0; e < 8; ++e) { lg[e] = lg[e] / sum; sum2 += lg[e]; }
#pragma unroll
                        for (int m = 32; m >= 1; m >>= 1) sum2 += __shfl_xor(sum2, m, 64);
                        float best = -INFINITY;
                        int bidx = 0;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float rr = (lg[e] / sum2) / qn[e];
                            if (rr > best) { best = rr; bidx = lane + 64 * e; }
                        }
#pragma unroll
                        for (int m = 32; m >= 1; m >>= 1) {
                            const float ob = __shfl_xor(best, m, 64);
                            const int oi = __shfl_xor(bidx, m, 64);
                            if (ob > best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
                        }
                        if (lane == 0) {
                            float x = 2.f * (float)bidx / ((float)C - 1.f) - 1.f;
                            if (leader) a.out[(size_t)(b0 + sj) * a.T + t] = x;
                            if (a.force_x) x = a.force_x[(size_t)(b0 + sj) * a.T + t];
                            XS[sj] = x;
                        }
                    }
                }
                __syncthreads();                         // LGT is read by every wave before the next group overwrites it
            }
            pp ^= 1;
            if (t + 1 < T1) {   // pull the next step's conditioning block of this group (32 KB) into this XCD's L2
                asm volatile("" ::"v"(touch));
                touch = a.cIf[((size_t)(tc + 1) * NGR + g) * XT + 32 * tid];
            }
        }
    }
    asm volatile("" ::"v"(touch));
    // ---- save the per-group state for the next slab of steps ------------------------------------------------------
    __syncthreads();
    for (int i = 0; i < nact; ++i) {
        const float4 *GP = reinterpret_cast<const float4 *>(smem + i * LGRP);
        float4 *dst = reinterpret_cast<float4 *>(a.state + state_wg + (size_t)i * LGRP);
        for (int q = tid; q < LGRP / 4; q += NT) dst[q] = GP[q];
    }
    return;
bail:   // a bounded spin expired (or another workgroup raised the abort flag): record the first failure, leave
    if (fcode != 0u) report_failure(a.status, fcode, blockIdx.x, t, tid);
#undef XLAYER
#undef STAGE_BARRIER
#undef PARTP
#undef GROUP_NB
}

// Grid = clusters x 64 workgroups of 256 threads, cooperative launch.  Workgroup wg of a cluster: role A if wg is even.
template <int MODE>
__global__ __launch_bounds__(NT, 1) void wrnn_loop_kernel(const LoopArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int cl, wg;                                         // cluster, workgroup-in-cluster: whole XCDs per cluster (speed only)
    const int ncl = gridDim.x / LNWGC;
    {
        const int b = blockIdx.x, nblk = gridDim.x;
        if (nblk % 8 == 0 && ncl >= 1 && 8 % ncl == 0) {
            const int xpc = 8 / ncl, per_xcd = nblk / 8;
            const int xcd = b % 8;
            cl = xcd / xpc;
            wg = (xcd % xpc) * per_xcd + b / 8;
        } else {
            cl = b / LNWGC;
            wg = b % LNWGC;
        }
    }
    if ((wg & 1) == 0) loop_role<MODE, true>(a, smem, cl, wg, ncl);
    else loop_role<MODE, false>(a, smem, cl, wg, ncl);
}

size_t loop_lds_bytes(int mode, int G) { return (size_t)loop_lds(mode, G).total * sizeof(float); }

// deepest pipeline: the exchange buffer has LMAXG slots per cluster (LDS is not the limit: 4.3 KB per group)
int loop_max_depth(int mode)
{
    int g = LMAXG;
    while (g > 1 && loop_lds_bytes(mode, g) > 160 * 1024) --g;
    return g;
}

size_t loop_state_floats(int G) { return (size_t)MAXCL * LNWGC * G * LGRP; }

int loop_clusters(int n_cus)
{
    int ncl = n_cus / LNWGC;
    if (ncl > MAXCL) ncl = MAXCL;
    while (ncl > 1 && (8 % ncl) != 0) --ncl;
    return ncl;
}

// One cooperative launch: clusters x 64 workgroups; args.G groups in flight per cluster, steps [args.t0, args.t1).
hipError_t launch_loop(const LoopArgs &args, int ncl, int mode, hipStream_t stream)
{
    if (ncl < 1 || args.G < 1 || args.G > loop_max_depth(mode)) return hipErrorInvalidValue;
    const size_t lds = loop_lds_bytes(mode, args.G);
    const void *fn = mode == 1 ? (const void *)wrnn_loop_kernel<1> : (const void *)wrnn_loop_kernel<0>;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    LoopArgs a = args;
    void *params[] = {(void *)&a};
    return hipLaunchCooperativeKernel(fn, dim3(ncl * LNWGC), dim3(NT), params, (unsigned)lds, stream);
}

}  // namespace wrnn
