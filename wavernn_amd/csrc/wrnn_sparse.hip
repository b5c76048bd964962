// wrnn_sparse.hip -- persistent WaveRNN loop kernel for BLOCK-SPARSE GRU weights (MOL) on MI355X (gfx950 / CDNA4); round 5 rebuild.
//
// BASELINE config 5: the four GRU matrices pruned per gate to ~5 % density in 16x1 blocks (wavernn_amd/prune.py: the rule of the
// reference's "Pruning - Scratchpad" notebook, JSON :40-186, applied to block magnitudes); the loop it runs is the reference's
// models/fatchord_version.py:201-241.  With 95 % of the gate weights gone a step is no longer matrix work (2.7 us of f32 MFMA for 256
// segments on the whole chip) but the LATENCY of its five dependent stages: x_{t-1} -> h1 -> h2 -> fc1 -> fc2 -> sample.  So the chip
// runs SIXTEEN independent chains side by side instead of sixteen groups through four deep pipelines (wrnn_duo.hip):
//
//   * 16 clusters of 16 CUs (half an XCD: every exchanged layer stays inside one L2), ONE workgroup of 4 waves per CU (one wave per
//     SIMD: 512 registers per lane -- every weight of the workgroup is register-resident, nothing spills), ONE group of <= 16 segments per
//     cluster: the step IS the chain, no software pipeline, every stage is straight-line code in one instruction stream per wave.
//   * a workgroup owns 64 hidden units of ONE GRU -- CUs 0-7 of a cluster: rnn1, CUs 8-15: rnn2 -- with BOTH halves of the cell: its rows
//     of W_ih (on the chain) and of W_hh (gh(t+1) = W_hh . h(t) + b_hh, needed a step later: it stays in the wave's registers, no exchange).
//     rnn1 workgroups multiply the I-layer conditioning cI through W_ih ahead of time: when x_{t-1} arrives only the x u1 term and the cell
//     are left; rnn2's workgroup 0 runs fc3 + the mixture-of-logistics sampling (utils/distribution.py:87-123; both fc3 tiles in LDS); its
//     other seven workgroups, which would wait ~3.5 us per step for x1, form cI (from the mel / the x25 signal and the frame's aux row, as
//     wrnn_duo.hip: 32 row blocks over 28 waves, four of them two blocks from one set of inputs).
//     EVERY workgroup owns 32 rows of fc1 and 32 rows of fc2 (registers): the two dense layers are the duo kernel's fc stage (K split
//     over the 4 waves, partial tiles through LDS, one barrier) spread over all 64 waves of the cluster.
//   * gate stage, no K split: wave w of a workgroup owns the 16-row block 4 ub + w of all three gates over the whole COMPACTED K
//     (NBP <= 48 / 64 surviving columns, the pack's sp_vals / sp_cols): NBP / 4 MFMAs per gate, A = the packed block values, B = the
//     activations GATHERED from the layer: the layers the gate stages read (h1 h2 x1 cI) are k-major -- element (k, n) at float 16 k + n, so
//     the 16 lanes of a k-quad read 64 contiguous bytes (in the fc stages' fragment order they would touch 4 x as many) --: one 4-byte sc1
//     load per lane and MFMA, every word its own arrival flag (the sentinel).  The accumulators ARE the gate pre-activations
//     of 4 consecutive units x 1 segment per lane: no partial tiles, no LDS, no barrier; the GRU cell runs in the accumulator registers
//     and the lane's four new h values go out as four 4-byte stores (k-major h), its residual sums likewise (x1) or as ONE 16-byte
//     store in the fc stages' fragment order (x2).
//   * exchange: the duo kernel's buffer geometry and ring rules (wrnn_ring.h; wrnn_duo.hip "Ring discipline"): sentinel layers h1 h2 x1
//     x2 y1 y2 with 4 ring entries, re-armed TWO steps ahead by the wave that publishes the words, after its last poll of the step (y1, in
//     the fc2 stage) and drained at the top of its next step; cI without a sentinel inside a launch (cI(t + 2) is formed by rnn2
//     workgroups at the end of their step t, drained at the top of their step t + 1 -- before x2(t + 1) goes out -- and gathered by rnn1
//     workgroups at the end of step t + 1, behind y1(t + 1), which needed every x2(t + 1); the first two steps of a launch are polled); x_t as tagged
//     8-byte words {x, step + 1} in two entries (no re-arm at all).  The skew argument is simpler than the duo kernel's: every
//     workgroup polls x2(t) and y1(t) of EVERY workgroup in every step, so nobody is ever more than one stage ahead of anybody.
//     tests/test_sparse_exchange_model.py runs these rules as a discrete-event model under adversarial timing.
//   * the conditioning is the duo kernel's: cI and (wrnn_options.mel_stage) the last up-sampling stage formed inside the loop, per-segment
//     aux tables per slab of <= 1,651 steps, a launch per slab, 16 floats of state per (unit, segment) between launches: the workspace
//     depends on neither T nor the corpus.
// Skipping exact zeros changes no partial sum; the summation ORDER differs from the dense kernels (surviving columns ascending, four
// at a time), so parity is to the MoL tolerance (tests/test_gpu_parity.py, tests/test_gpu_fullsize.py).
#include <type_traits>

#include "wrnn_ring.h"

namespace wrnn {

constexpr int SPCLUSTERS = 16;               // clusters of 16 CUs per chip
constexpr int SPWG = 16;                     // workgroups per cluster (one per CU)
constexpr int SPPART = 2 * NW * 2 * 256;     // two ping-pong sets of [wave][tile 0..1][lane][4]
constexpr int SPSTATE_WG = NT * 16 + SEG;    // saved state of a workgroup: per thread {h[4], gh_r[4], gh_z[4], gh_n[4]}, then x_{t1-1} (rnn1)
constexpr int SPSTATE_CL = SPWG * SPSTATE_WG; // ... of a cluster
static_assert(SPCLUSTERS <= LMAXG * MAXCL, "one exchange-buffer region per cluster");
static_assert(SPCLUSTERS * SPWG <= XCC_WORDS, "placement table");

struct SpLds {
    int off_seg, off_part, off_log, off_misc, off_prof, off_ct1, off_f3, total;
};
__host__ __device__ inline SpLds sp_lds()
{
    SpLds l;
    int o = 0;
    l.off_seg = o;  o += 64;                 // ints: 16 positions | 16 limits | 16 table-row bases of this slab | 16 mel offsets
    l.off_part = o; o += SPPART;
    l.off_log = o;  o += 32;                 // the sampling workgroup: fc3.bias
    l.off_misc = o; o += 64;                 // placement table of the cluster (ints)
    l.off_prof = o; o += 64;                 // [32] u64 phase clocks (profiling builds)
    o = (o + 3) & ~3;
    l.off_ct1 = o;  o += 5 * 64 * (CK + 4);   // rnn2's cI-forming workgroups: the I-layer tiles [wave 0..3 | wave 0's SECOND block][kk | bias][lane]
    l.off_f3 = o;   o += 2 * XT;             // the sampling workgroup: fc3 (30 x 512 = two 16-row tiles) in A-fragment order
    l.total = o;
    return l;
}

// The 16-row block of the three gates of one pruned matrix that a wave owns: MPW = NBP / 4 MFMAs per gate.  MFMA i of gate g contracts
// the surviving blocks 4 i .. 4 i + 3 of the block row: lane (fi = lane & 15, kq = lane >> 4) holds the value of row fi in block 4 i + kq
// (A operand) and gathers the activation of that block's COLUMN for segment fi (B operand) from byte offset off[g][i] of a layer entry.
template <int MPW>
struct GateTiles {
    float a[3][MPW];
    int off[3][MPW];
};
template <int MPW>
__device__ __forceinline__ void gate_tiles_init(GateTiles<MPW> &gt, const float *sp_vals, const int *sp_cols, int m, int rb, int lane)
{
    const int fi = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int i = 0; i < MPW; ++i) {
            const size_t r = ((size_t)(m * 32 + rb) * 3 + g) * (4 * MPW) + 4 * i + kq;
            gt.a[g][i] = sp_vals[r * 16 + fi];
            const int c = sp_cols[r];                       // (padding blocks: column 0 with zero values)
            gt.off[g][i] = c * 64 + fi * 4;                 // k-major layer: element (column c, segment fi)
        }
}
template <int MPW>
__device__ __forceinline__ void gather_issue(__amdgpu_buffer_rsrc_t rs, int soff, const GateTiles<MPW> &gt, unsigned (&v)[3][MPW])
{
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int i = 0; i < MPW; ++i) v[g][i] = __builtin_amdgcn_raw_buffer_load_b32(rs, gt.off[g][i], soff, 16 /* sc1 */);
}
// no gathered word of a live segment's lane is still the sentinel (wave-uniform)
template <int MPW>
__device__ __forceinline__ bool gather_there(const unsigned (&v)[3][MPW], unsigned extra, bool live)
{
    unsigned m = extra;
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int i = 0; i < MPW; ++i) m = max(m, v[g][i]);
    return __all(m != SENT || !live);
}
// (round 6: the tile values are CONSTRAINED to AGPRs and read in place -- wrnn_ring.h, mfma_ag: the allocator used the AGPRs as spill slots and copied the
// tiles of every stage into VGPRs first, 2,045 v_accvgpr moves in this kernel, most of them on the cluster's chain; same products, same order)
template <int MPW>
__device__ __forceinline__ void gate_mfma(const GateTiles<MPW> &gt, const unsigned (&v)[3][MPW], f32x4 &o0, f32x4 &o1, f32x4 &o2)
{
    f32x4 c0 = mfma_ag0(gt.a[0][0], __uint_as_float(v[0][0])), c1 = mfma_ag0(gt.a[1][0], __uint_as_float(v[1][0])), c2 = mfma_ag0(gt.a[2][0], __uint_as_float(v[2][0]));
#pragma unroll
    for (int i = 1; i < MPW; ++i) {                          // three independent chains interleaved (96 cycles between dependent MFMAs)
        mfma_ag(c0, gt.a[0][i], __uint_as_float(v[0][i]));
        mfma_ag(c1, gt.a[1][i], __uint_as_float(v[1][i]));
        mfma_ag(c2, gt.a[2][i], __uint_as_float(v[2][i]));
    }
    asm volatile("s_nop 15\n\ts_nop 7" : "+v"(c0), "+v"(c1), "+v"(c2));
    o0 = c0; o1 = c1; o2 = c2;
}

// Round 6 -- BLOCK-SPARSE fc1 / fc2 (FCS; the reference's pruning recipe prunes the Linear layers with the GRUs: "Pruning - Scratchpad.ipynb" :199-204).
// ONE 16-row block of fc1 (waves 0-1 of a workgroup) or of fc2 (waves 2-3) per wave over the compacted K, packed and gathered exactly like a gate
// tile (the pack's sp_fc_vals / sp_fc_cols): MPW MFMAs, the accumulators ARE rows 4 kq .. 4 kq + 3 of the block for segment fi -- no K split, no
// partial tiles, no LDS, no barrier, and 2 x (12 + 12) registers per lane instead of the 128 of the four dense tiles.
template <int MPW>
struct FcTile {
    float a[MPW];
    int off[MPW];
};
template <int MPW>
__device__ __forceinline__ void fc_tile_init(FcTile<MPW> &ft, const float *vals, const int *cols, int m, int rb, int lane)
{
    const int fi = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int i = 0; i < MPW; ++i) {
        const size_t r = ((size_t)m * 32 + rb) * (4 * MPW) + 4 * i + kq;
        ft.a[i] = vals[r * 16 + fi];
        ft.off[i] = cols[r] * 64 + fi * 4;                  // k-major layer: element (column, segment fi)
    }
}

__device__ __forceinline__ unsigned max4(const u32x4 &q) { return max(max(q.x, q.y), max(q.z, q.w)); }
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#define SPX(k)                                                                 \
    do {                                                                       \
        if (PROF && tid == 0) {                                                \
            const u64 now_ = __builtin_amdgcn_s_memtime();                     \
            PROFL[k] += now_ - plast;                                          \
            plast = now_;                                                      \
        }                                                                      \
    } while (0)

// two fc row tiles that share the activation operand (B fragments in registers); per tile mfma1's order (two chains by k-block parity)
__device__ __forceinline__ void mfma2(const float (&a0)[AF], const float (&a1)[AF], const float (&b)[32], f32x4 &o0, f32x4 &o1)
{
    f32x4 c00 = mfma_ag0(a0[0], b[0]), c10 = mfma_ag0(a1[0], b[0]), c01 = mfma_ag0(a0[4], b[4]), c11 = mfma_ag0(a1[4], b[4]);
#pragma unroll
    for (int r = 0; r < 8; r += 2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (r + e > 0) {
                mfma_ag(c00, a0[4 * r + e], b[4 * r + e]);
                mfma_ag(c10, a1[4 * r + e], b[4 * r + e]);
                mfma_ag(c01, a0[4 * r + 4 + e], b[4 * r + 4 + e]);
                mfma_ag(c11, a1[4 * r + 4 + e], b[4 * r + 4 + e]);
            }
        }
    }
    asm volatile("s_nop 15\n\ts_nop 7" : "+v"(c00), "+v"(c10), "+v"(c01), "+v"(c11));
    o0 = c00 + c01;
    o1 = c10 + c11;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// One workgroup of a cluster: 64 units (unit block ub) of rnn1 (LA) or rnn2.  rg = the cluster's region of the exchange buffer, gid =
// its group of the round, wgi = index in the cluster (fc rows [32 wgi, 32 wgi + 32) of fc1 and of fc2).
// ---------------------------------------------------------------------------------------------------------------------------------
// PROF (wrnn_options.phase_clocks; thread 0 of every workgroup, shader clocks per segment of a step, in program order): rnn1: 0 drain + wait for
// x_{t-1}, 1 cell + publish, 2 wait for h1(t), 3 gh tiles, 4 wait x2, 5 fc1, 6 wait y1, 7 fc2, 8 wait cI(t+1), 9 W_ih . cI tiles;
// rnn2: 0 drain + wait for x1(t), 1 gate tiles + cell + publish, 2 wait x2, 3 fc1, 6 wait y1, 7 fc2, 8 wait y2, 9 fc3 + sampling, 4 wait h2 (there), 5 gh
// tiles, 10 cI(t+2) formed; 15 = steps
template <int NBP, bool FCS, bool LA, bool PROF>
__device__ __forceinline__ void sp_role(const LoopArgs &a, float *smem, const int rg, const int gid, const int ub, const int wgi, const bool loc)
{
    constexpr int MPW = NBP / 4;
    const SpLds L = sp_lds();
    float *PART = smem + L.off_part, *fc3b = smem + L.off_log, *F3 = smem + L.off_f3;
    int *SEGT = reinterpret_cast<int *>(smem + L.off_seg);
    u64 *PROFL = reinterpret_cast<u64 *>(smem + L.off_prof);
    u64 plast = 0;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fi = lane & 15, kq = lane >> 4;           // MFMA coordinates: A row / B segment, k-quad; D: rows 4 kq + e, segment fi
    const int rb = 4 * ub + w;                          // this wave's 16-unit block of the layer (gate stages)
    const int u0 = LU * rb + 4 * kq;                    // ... this lane's four units u0 .. u0 + 3 (of segment fi)
    const int kbase_lane = KCH * w + 4 * kq;            // fc stages: K split over the waves
    const int pu = 4 * w + (tid & 3), pj = (tid >> 2) & 15;      // fc pointwise role: rows 32 wgi + pu and 32 wgi + 16 + pu, segment pj
    const int T0 = a.t0, T1 = a.t1;
    // (everything the stage lambdas need of the launch arguments as local values: see wrnn_duo.hip)
    unsigned *const status = a.status;
    const int hop = a.hop, resume = a.resume, Tall = a.T, Nall = a.Nall, noise_t0 = a.noise_t0, C = a.C;
    const unsigned magic = a.hop_magic;
    const int mshift = a.hop_shift;
    const int zrow = a.Nall * a.tab_fps;
    float *const outp = a.out, *const dbgl = a.dbg_logits;
    const float *const forcex = a.force_x, *const noise_pre = a.noise_pre;
    const float *const mels_up = a.mels_up, *const aux_fr = a.aux_fr, *const mel_coef = a.mel_coef;
    const int mel_stage = a.mel_stage;
    const int NR = a.Btot, NGR = a.NG;
    const int b0 = (int)(((long)gid * NR) / NGR), nb = (int)(((long)(gid + 1) * NR) / NGR) - b0;
    const int b0g = a.rb0 + b0;                         // first segment of the group in the call
    constexpr int L_H = LA ? 0 : 1, L_XR = LA ? 5 : 6, L_IN = LA ? 4 : 5;
    float *const state_wg = a.state + (size_t)gid * SPSTATE_CL + (size_t)wgi * SPSTATE_WG;
    const bool sampler = !LA && ub == 0;

    // ---- weights: the wave's gate tiles of W_ih and W_hh, the workgroup's two fc1 and two fc2 tiles
    GateTiles<MPW> gi, gh;
    gate_tiles_init(gi, a.sp_vals, a.sp_cols, LA ? 0 : 2, rb, lane);
    gate_tiles_init(gh, a.sp_vals, a.sp_cols, LA ? 1 : 3, rb, lane);
    float A_fc1[2][AF], A_fc2[2][AF];
    FcTile<MPW> ft;                                     // FCS: this wave's fc tile -- waves 0-1: row block 2 wgi + w of fc1, waves 2-3: row block 2 wgi + w - 2 of fc2
    if constexpr (FCS) fc_tile_init(ft, a.sp_fc_vals, a.sp_fc_cols, w >> 1, 2 * wgi + (w & 1), lane);
    else {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            load_afrag(A_fc1[q], a.fc1_w, H + AUX, 2 * LU * wgi + LU * q + fi, true, kbase_lane);
            load_afrag(A_fc2[q], a.fc2_w, H + AUX, 2 * LU * wgi + LU * q + fi, true, kbase_lane);
        }
    }
    // FCS: fc3 FOLDED into the fc2 tiles (as wrnn_chain.hip, round 6): a wave that owns an fc2 row block multiplies ITS 16 rows of y2 through fc3's columns of
    // those rows -- two 16-row tiles x 4 MFMAs; K-slot kq of MFMA e <-> row 4 kq + e, i.e. the B operand IS the lane's accumulator element e: no data moves --
    // and publishes two partial logit tiles (layers 3 and 16) instead of y2; the sampling workgroup adds the 32 waves' tiles instead of running 64 MFMAs per wave.
    const bool fold3 = FCS && (a.tuning & 16) == 0;     // (A/B: tuning bit 4 = y2 published, dense fc3 on the sampling workgroup)
    float A3[2][4];
    if constexpr (FCS) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) A3[q][e] = (16 * q + fi < C) ? a.fc3_w[(size_t)(16 * q + fi) * H + LU * (2 * wgi + (w & 1)) + 4 * kq + e] : 0.f;
    }
    for (int q = tid; q < L.off_f3; q += NT) smem[q] = 0.f;
    if (sampler) {                                      // fc3 -> LDS (fragment order as in the pack)
        for (int q = tid; q < 2 * XT / 4; q += NT) reinterpret_cast<float4 *>(F3)[q] = reinterpret_cast<const float4 *>(a.fc3f)[q];
    }
    // constants of the gate pointwise role (units u0 + e): rnn1: b_ih1, u1 = W_ih1 . w0 (the x_{t-1} term), w0 (rnn2's b_ih2 is inside c2f); b_hh
    float cb[3][4], ux[3][4], w0o[4], bh[3][4];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            cb[g][e] = LA ? a.b_ih1[g * H + u0 + e] : 0.f;
            ux[g][e] = LA ? a.u1[g * H + u0 + e] : 0.f;
            bh[g][e] = (LA ? a.b_hh1 : a.b_hh2)[g * H + u0 + e];
        }
#pragma unroll
    for (int e = 0; e < 4; ++e) w0o[e] = LA ? a.I_w0[u0 + e] : 0.f;
    const float b3a = sampler ? a.fc3_b[pu] : 0.f, b3b = (sampler && 16 + pu < 30) ? a.fc3_b[16 + pu] : 0.f;
    // The I-layer conditioning cI (512 rows = 32 blocks of 16) is formed by rnn2's SEVEN non-sampling workgroups, which otherwise wait ~3.5 us
    // per step for x1 (profiles/r05d_sparse_phase_clocks.json; on rnn1's workgroups it made them late for x_t): wave (ub >= 1, w) forms block
    // 4 (ub - 1) + w, and wave 0 of ub = 1 .. 4 a second block 27 + ub from the same inputs (28 more MFMAs, no more loads).
    const bool cond_wg = !LA && ub >= 1;
    const bool cond2 = cond_wg && ub <= 4 && w == 0;
    const int cblk0 = cond_wg ? 4 * (ub - 1) + w : 0, cblk1 = cond2 ? 27 + ub : 0;
    // (the I-layer tiles live in LDS and are read when a block is formed, off the chain: with them in registers every stage of the rnn2
    // workgroups -- fc1 / fc2 are on the chain -- ran ~0.3 us slower, profiles/r05e_sparse_phase_clocks.json)
    float *CT0 = smem + L.off_ct1 + w * 64 * (CK + 4), *CT1 = smem + L.off_ct1 + 4 * 64 * (CK + 4);
    __syncthreads();
    if (sampler && tid >= 32 && tid < 64) fc3b[tid - 32] = tid - 32 < 30 ? a.fc3_b[tid - 32] : 0.f;
    if (cond_wg) {
        CondTile c1;
        cond_tile_init(c1, a.I_cT, a.I_b, cblk0, lane);
#pragma unroll
        for (int kk = 0; kk < CK; ++kk) CT0[kk * 64 + lane] = c1.w[kk];
#pragma unroll
        for (int i = 0; i < 4; ++i) CT0[(CK + i) * 64 + lane] = c1.bias[i];
        if (cond2) {
            cond_tile_init(c1, a.I_cT, a.I_b, cblk1, lane);
#pragma unroll
            for (int kk = 0; kk < CK; ++kk) CT1[kk * 64 + lane] = c1.w[kk];
#pragma unroll
            for (int i = 0; i < 4; ++i) CT1[(CK + i) * 64 + lane] = c1.bias[i];
        }
    }
    if (tid < SEG) {
        const int sc = b0g + (tid < nb ? tid : nb - 1);
        const int pos = a.seg_pos[sc];
        SEGT[tid] = pos;
        SEGT[SEG + tid] = a.seg_lim[sc];
        SEGT[2 * SEG + tid] = sc * a.tab_fps - (pos + a.tab_t0) / a.hop;
        SEGT[3 * SEG + tid] = a.mel_stage ? a.seg_moff[sc] : 0;
    }
    __syncthreads();

    const __amdgpu_buffer_rsrc_t xrs = make_rsrc(a.xbuf, (unsigned)(DXBUF_FLOATS * 4));
    const __amdgpu_buffer_rsrc_t crs = make_rsrc(a.c2f, 0x7FFFF000u);          // rnn2: per-frame table of its aux columns + b_ih2
    const __amdgpu_buffer_rsrc_t f1rs = make_rsrc(a.c3f, 0x7FFFF000u), f2rs = make_rsrc(a.c4f, 0x7FFFF000u);
    const int cbase = rg * DSLOTB;
    const int voff_frag = frag_off(w, 0, lane) * 4;      // fc stages: this lane's first fragment of a layer (bytes)
    const int voff_blk = rb * 1024 + lane * 16;          // gate stages: this lane's quarter-row of the wave's 1 KB block of a k-major layer (re-arm stores)
    const int voff_own = u0 * 64 + fi * 4;               // ... the word of (unit u0, segment fi) in a k-major layer; units u0 + e: + 64 e
    const bool live = fi < nb;                           // this lane's segment exists (gate stages, fragment polls)

    bool dead = false;
    int pp = 0;
    int t = T0;
    float h[4] = {0.f, 0.f, 0.f, 0.f};                   // the GRU state of the lane's four units (fatchord_version.py:194-195: zeros)
    float ghr[4], ghz[4], ghn[4];                        // gh(t) = W_hh . h(t - 1) + b_hh of those units: formed during step t - 1, kept here
    f32x4 gacc[3];                                       // rnn1: W_ih1 . cI(t), formed at the end of step t - 1
    u32x4 own = {0u, 0u, 0u, 0u};                        // the lane's own words of the gates' input layer (cI / x1) = the residual input
    gacc[0] = gacc[1] = gacc[2] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (resume) {
        const float4 hv = *reinterpret_cast<const float4 *>(state_wg + tid * 16), r4 = *reinterpret_cast<const float4 *>(state_wg + tid * 16 + 4),
                     z4 = *reinterpret_cast<const float4 *>(state_wg + tid * 16 + 8), n4 = *reinterpret_cast<const float4 *>(state_wg + tid * 16 + 12);
        h[0] = hv.x; h[1] = hv.y; h[2] = hv.z; h[3] = hv.w;
        ghr[0] = r4.x; ghr[1] = r4.y; ghr[2] = r4.z; ghr[3] = r4.w; ghz[0] = z4.x; ghz[1] = z4.y; ghz[2] = z4.z; ghz[3] = z4.w;
        ghn[0] = n4.x; ghn[1] = n4.y; ghn[2] = n4.z; ghn[3] = n4.w;
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) { ghr[e] = bh[0][e]; ghz[e] = bh[1][e]; ghn[e] = bh[2][e]; }      // W_hh . 0 + b_hh
    }

    // the lane's four units of a k-major layer (h1 h2 x1 cI: the layers the gate stages GATHER from -- [k][16 segments], 64 contiguous bytes
    // per k: a gathering wave's load touches 4 x 64 B instead of 16 x 64 B in fragment order): four 4-byte stores, 64 bytes apart
    auto store4 = [&](const u32x4 &q, int voff_layer, int soff) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (loc) __builtin_amdgcn_raw_buffer_store_b32(q[e], xrs, voff_layer + voff_own + 64 * e, soff, 0);
            else __builtin_amdgcn_raw_buffer_store_b32(q[e], xrs, voff_layer + voff_own + 64 * e, soff, 16 /* sc1 */);
        }
    };
    auto load_own = [&](int soff) {
        u32x4 q;
#pragma unroll
        for (int e = 0; e < 4; ++e) q[e] = __builtin_amdgcn_raw_buffer_load_b32(xrs, voff_own + 64 * e, soff, 16 /* sc1 */);
        return q;
    };
    auto store16 = [&](const u32x4 &q, int voff, int soff) {
        if (loc) __builtin_amdgcn_raw_buffer_store_b128(q, xrs, voff, soff, 0);       // the whole cluster was seen on one XCD: a plain store stays in its L2
        else __builtin_amdgcn_raw_buffer_store_b128(q, xrs, voff, soff, 16 /* sc1 */);
    };
    // ring hygiene (header): after the last poll of step t this wave resets its OWN words of entry (t + 2) % 4 in the sentinel layers it
    // publishes: its quarters of the workgroup's two y1 and two y2 blocks (one 16-lane group each), its block of h and of the residual sum
    auto rearm = [&]() {
        const int so = cbase + ((t + DAHEAD_IH) & (DRING - 1)) * XTB;
        const u32x4 q = {SENT, SENT, SENT, SENT};
        if constexpr (FCS) {
            store16(q, (w < 2 ? 2 : 3) * DLAYERB + (2 * wgi + (w & 1)) * 1024 + lane * 16, so);      // the wave's OWN 1 KB block of y1 (k-major) / y2 (fold3: tile-0 partial logits)
            if (fold3 && w >= 2) store16(q, 16 * DLAYERB + (2 * wgi + (w & 1)) * 1024 + lane * 16, so);       // ... and the tile-1 partial logits
        }
        else store16(q, (kq < 2 ? 2 : 3) * DLAYERB + (2 * wgi + (kq & 1)) * 1024 + w * 256 + fi * 16, so);
        store16(q, L_H * DLAYERB + voff_blk, so);
        store16(q, L_XR * DLAYERB + voff_blk, so);
    };

    // ---------------- fc stage: relu(fc1([x2, a3])) -> y1 (which = 1) / relu(fc2([y1, a4])) -> y2 (which = 2), the workgroup's 32 rows
    auto fc = [&](auto WC) {
        constexpr int which = decltype(WC)::value;
        constexpr int LI = which == 1 ? 6 : 2, LO = which == 1 ? 2 : 3;
        const int sb = cbase + (t & (DRING - 1)) * XTB;
        if constexpr (FCS) {
            // the gathered form: only the waves that own a tile of this layer run it (wave-uniform); its input layer (x2 / y1) is k-major
            if ((w < 2) == (which == 1)) {
                unsigned v[MPW];
#pragma unroll
                for (int i = 0; i < MPW; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b32(xrs, ft.off[i], sb + LI * DLAYERB, 16 /* sc1 */);
                const int fr = table_row(SEGT[fi] + t, SEGT[SEG + fi], SEGT[2 * SEG + fi], magic, mshift, hop, zrow);
                const int r0 = 2 * LU * wgi + LU * (w & 1) + 4 * kq;                   // the lane's rows r0 .. r0 + 3 (of segment fi)
                const u32x4 cv = __builtin_amdgcn_raw_buffer_load_b128(which == 1 ? f1rs : f2rs, (fr * H + r0) * 4, 0, 0);
                auto there = [&] {
                    unsigned m = 0u;
#pragma unroll
                    for (int i = 0; i < MPW; ++i) m = max(m, v[i]);
                    return __all(m != SENT || !live);
                };
                if (__builtin_expect(!there(), 0))
                    wait_for(there,
                             [&] {
#pragma unroll
                                 for (int i = 0; i < MPW; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b32(xrs, ft.off[i], sb + LI * DLAYERB, 16 /* sc1 */);
                             },
                             status, dead, 0x700u | (unsigned)which, t);
                SPX(LA ? (which == 1 ? 4 : 6) : (which == 1 ? 2 : 6));
                f32x4 c0 = mfma_ag0(ft.a[0], __uint_as_float(v[0])), c1 = mfma_ag0(ft.a[1], __uint_as_float(v[1])), c2 = mfma_ag0(ft.a[2], __uint_as_float(v[2]));
#pragma unroll
                for (int i = 3; i < MPW; ++i) {              // three chains by i % 3, added at the end
                    if (i % 3 == 0) mfma_ag(c0, ft.a[i], __uint_as_float(v[i]));
                    else if (i % 3 == 1) mfma_ag(c1, ft.a[i], __uint_as_float(v[i]));
                    else mfma_ag(c2, ft.a[i], __uint_as_float(v[i]));
                }
                asm volatile("s_nop 15\n\ts_nop 7" : "+v"(c0), "+v"(c1), "+v"(c2));
                const f32x4 acc = c0 + c1 + c2;
                u32x4 q;
#pragma unroll
                for (int e = 0; e < 4; ++e) q[e] = __float_as_uint(fmaxf(acc[e] + __uint_as_float(cv[e]), 0.f));      // relu(fc([x, a]) + b): the aux columns and the bias sit in the per-frame table
                u32x4 pl[2] = {q, q};
                if constexpr (which == 2) {
                    if (fold3) {                            // (whole-wave MFMAs: outside the per-lane `live` branch; a column of an absent segment is never stored)
#pragma unroll
                        for (int tl = 0; tl < 2; ++tl) {
                            f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                            for (int e = 0; e < 4; ++e) c = __builtin_amdgcn_mfma_f32_16x16x4f32(A3[tl][e], __uint_as_float(q[e]), c, 0, 0, 0);
                            pl[tl] = u32x4{__float_as_uint(c[0]), __float_as_uint(c[1]), __float_as_uint(c[2]), __float_as_uint(c[3])};
                        }
                    }
                }
                if (live) {
                    if constexpr (which == 1) {             // y1: gathered by the fc2 tiles -> k-major
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (loc) __builtin_amdgcn_raw_buffer_store_b32(q[e], xrs, LO * DLAYERB + (r0 + e) * 64 + fi * 4, sb, 0);
                            else __builtin_amdgcn_raw_buffer_store_b32(q[e], xrs, LO * DLAYERB + (r0 + e) * 64 + fi * 4, sb, 16 /* sc1 */);
                        }
                    } else if (fold3) {                     // partial logits of this wave's 16 y2 rows: tile 0 -> layer 3, tile 1 -> layer 16 (block = the fc2 row block)
                        store16(pl[0], 3 * DLAYERB + (2 * wgi + (w & 1)) * 1024 + lane * 16, sb);
                        store16(pl[1], 16 * DLAYERB + (2 * wgi + (w & 1)) * 1024 + lane * 16, sb);
                    } else store16(q, LO * DLAYERB + (2 * wgi + (w & 1)) * 1024 + lane * 16, sb);      // y2: read by the dense fc3 of the sampling workgroup -> fragment order
                }
                rearm();                                // (behind this wave's last sentinel poll of the step; tests/test_sparse_exchange_model.py, SparseFcSim)
            }
            SPX(LA ? (which == 1 ? 5 : 7) : (which == 1 ? 3 : 7));
        } else {
        u32x4 x[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) x[r] = __builtin_amdgcn_raw_buffer_load_b128(xrs, voff_frag + r * 1024, sb + LI * DLAYERB, 16 /* sc1 */);
        const int fr = table_row(SEGT[pj] + t, SEGT[SEG + pj], SEGT[2 * SEG + pj], magic, mshift, hop, zrow);
        const int vo = (fr * H + 2 * LU * wgi + pu) * 4;
        const float cv0 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(which == 1 ? f1rs : f2rs, vo, 0, 0));
        const float cv1 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(which == 1 ? f1rs : f2rs, vo, LU * 4, 0));
        if (__builtin_expect(!frag_there(x, live), 0))
            wait_for([&] { return frag_there(x, live); },
                     [&] {
#pragma unroll
                         for (int r = 0; r < 8; ++r) x[r] = __builtin_amdgcn_raw_buffer_load_b128(xrs, voff_frag + r * 1024, sb + LI * DLAYERB, 16 /* sc1 */);
                     },
                     status, dead, 0x700u | (unsigned)which, t);
        SPX(LA ? (which == 1 ? 4 : 6) : (which == 1 ? 2 : 6));
        float b[32];
        frag_to_b(x, b);
        float *PW = PART + pp * (NW * 2 * 256);
        f32x4 o0, o1;
        if constexpr (which == 1) mfma2(A_fc1[0], A_fc1[1], b, o0, o1);
        else mfma2(A_fc2[0], A_fc2[1], b, o0, o1);
        put_partial<2>(PW, w, 0, lane, o0);
        put_partial<2>(PW, w, 1, lane, o1);
        lds_barrier();
        publish4l(xrs, sb + LO * DLAYERB + (2 * wgi) * 1024, tid, fmaxf(get_partial<2>(PW, 0, pu, pj) + cv0, 0.f), pj < nb, loc);
        publish4l(xrs, sb + LO * DLAYERB + (2 * wgi + 1) * 1024, tid, fmaxf(get_partial<2>(PW, 1, pu, pj) + cv1, 0.f), pj < nb, loc);
        if constexpr (which == 2) rearm();              // (behind the last poll of the step -- y1(t) above -- and behind the publication: off the chain)
        pp ^= 1;
        SPX(LA ? (which == 1 ? 5 : 7) : (which == 1 ? 3 : 7));
        }
    };

    // ---------------- gate stages ----------------
    // the GRU cell of the lane's four units (ATen gru_cell; hardware exp / rcp as the duo kernel's MoL path) and the publication of
    // the residual sum (on the chain: first) and of h
    auto cell_publish = [&](const float (&gir)[4], const float (&giz)[4], const float (&gin)[4], const float (&xin)[4]) {
        u32x4 qx, qh;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            h[e] = gru_update_fast(gir[e], giz[e], gin[e], ghr[e], ghz[e], ghn[e], h[e]);
            qx[e] = __float_as_uint(xin[e] + h[e]);       // x1 = xi + h1 (:212) / x2 = x1 + h2 (:216)
            qh[e] = __float_as_uint(h[e]);
        }
        if (live) {
            const int sb = cbase + (t & (DRING - 1)) * XTB;
            if constexpr (LA || FCS) store4(qx, L_XR * DLAYERB, sb);        // x1: gathered by rnn2's gate stage (k-major); FCS: x2 is gathered too (fc1 tiles)
            else store16(qx, L_XR * DLAYERB + voff_blk, sb);               // x2: read by the fc1 stages (fragment order: units u0 .. u0 + 3 of segment fi = one word)
            store4(qh, L_H * DLAYERB, sb);
        }
    };

    // rnn1, front half: W_ih1 . cI(ts) of the wave's rows + the lane's own cI words (the residual input xi - w0 x), as soon as cI(ts) is there.
    // In two parts, so that the forming of cI(ts + 1) runs under the gather's latency: issue | ... | finish
    unsigned fv[3][MPW];
    auto front_issue = [&](int ts) {
        const int so = cbase + L_IN * DLAYERB + (ts & (DRING - 1)) * XTB;
        gather_issue(xrs, so, gi, fv);
        own = load_own(so);
    };
    auto front_finish = [&](int ts) {
        const int so = cbase + L_IN * DLAYERB + (ts & (DRING - 1)) * XTB;
        if (__builtin_expect(!gather_there(fv, max4(own), live), 0))
            wait_for([&] { return gather_there(fv, max4(own), live); }, [&] { gather_issue(xrs, so, gi, fv); own = load_own(so); }, status, dead, 0x720u, ts);
        SPX(8);
        gate_mfma(gi, fv, gacc[0], gacc[1], gacc[2]);
        SPX(9);
    };
    // rnn1, back half (the chain: sampling -> here): x_{t-1} arrives as a tagged word {x, tag = t}
    auto back_a = [&]() {
        float xv = 0.f;
        if (t > T0) {
            const int sx = cbase + 7 * DLAYERB + ((t - 1) & 1) * XTB;
            u32x2 xq = __builtin_amdgcn_raw_buffer_load_b64(xrs, fi * 8, sx, 16 /* sc1 */);
            if (__builtin_expect(__any(live && xq.y != (unsigned)t), 0))
                wait_for([&] { return !__any(live && xq.y != (unsigned)t); }, [&] { xq = __builtin_amdgcn_raw_buffer_load_b64(xrs, fi * 8, sx, 16 /* sc1 */); },
                         status, dead, 0x730u, t);
            xv = __uint_as_float(xq.x);
        } else if (resume) xv = state_wg[NT * 16 + fi];
        SPX(0);
        float gir[4], giz[4], gin[4], xin[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            gir[e] = gacc[0][e] + fmaf(xv, ux[0][e], cb[0][e]);
            giz[e] = gacc[1][e] + fmaf(xv, ux[1][e], cb[1][e]);
            gin[e] = gacc[2][e] + fmaf(xv, ux[2][e], cb[2][e]);
            xin[e] = fmaf(w0o[e], xv, __uint_as_float(own[e]));      // xi of the lane's units (:208-209)
        }
        cell_publish(gir, giz, gin, xin);
        SPX(1);
    };
    // rnn2: the whole gate stage is on the chain (x1 -> here)
    auto gates_b = [&]() {
        const int so = cbase + L_IN * DLAYERB + (t & (DRING - 1)) * XTB;
        unsigned v[3][MPW];
        gather_issue(xrs, so, gi, v);
        own = load_own(so);
        // FCS: every wave also waits for the tagged word x_{t-1} (it exists long before x1(t) does: nothing on the chain) -- a wave whose gathers are
        // empty (a fully pruned block row) would otherwise run free of the ring; the skew argument: tests/test_sparse_exchange_model.py, SparseFcSim
        u32x2 xq = {0u, (unsigned)t};
        if (FCS && t > T0) xq = __builtin_amdgcn_raw_buffer_load_b64(xrs, fi * 8, cbase + 7 * DLAYERB + ((t - 1) & 1) * XTB, 16 /* sc1 */);
        const int fr = table_row(SEGT[fi] + t, SEGT[SEG + fi], SEGT[2 * SEG + fi], magic, mshift, hop, zrow);
        u32x4 c2[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) c2[q] = __builtin_amdgcn_raw_buffer_load_b128(crs, (fr * 3 * H + q * H + u0) * 4, 0, 0);
        if (__builtin_expect(!gather_there(v, max4(own), live), 0))
            wait_for([&] { return gather_there(v, max4(own), live); }, [&] { gather_issue(xrs, so, gi, v); own = load_own(so); }, status, dead, 0x728u, t);
        if constexpr (FCS) {
            if (__builtin_expect(__any(live && xq.y != (unsigned)t), 0))
                wait_for([&] { return !__any(live && xq.y != (unsigned)t); },
                         [&] { xq = __builtin_amdgcn_raw_buffer_load_b64(xrs, fi * 8, cbase + 7 * DLAYERB + ((t - 1) & 1) * XTB, 16 /* sc1 */); }, status, dead, 0x729u, t);
        }
        SPX(0);
        f32x4 o0, o1, o2;
        gate_mfma(gi, v, o0, o1, o2);
        float gir[4], giz[4], gin[4], xin[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            gir[e] = o0[e] + __uint_as_float(c2[0][e]);
            giz[e] = o1[e] + __uint_as_float(c2[1][e]);
            gin[e] = o2[e] + __uint_as_float(c2[2][e]);
            xin[e] = __uint_as_float(own[e]);
        }
        cell_publish(gir, giz, gin, xin);
        SPX(1);
    };
    // gh(t + 1) = W_hh . h(t) + b_hh of the wave's rows (h(t) of the whole layer gathered from the ring): stays in this lane's registers
    auto gh_stage = [&]() {
        const int so = cbase + L_H * DLAYERB + (t & (DRING - 1)) * XTB;
        unsigned v[3][MPW];
        gather_issue(xrs, so, gh, v);
        if (__builtin_expect(!gather_there(v, 0u, live), 0))
            wait_for([&] { return gather_there(v, 0u, live); }, [&] { gather_issue(xrs, so, gh, v); }, status, dead, 0x740u | (LA ? 0u : 8u), t);
        SPX(LA ? 2 : 4);
        f32x4 o0, o1, o2;
        gate_mfma(gh, v, o0, o1, o2);
#pragma unroll
        for (int e = 0; e < 4; ++e) { ghr[e] = o0[e] + bh[0][e]; ghz[e] = o1[e] + bh[1][e]; ghn[e] = o2[e] + bh[2][e]; }
        SPX(LA ? 3 : 5);
    };
    // rnn2's non-sampling workgroups: cI(tt) = b_I + W_I[:, 1:] . [m ; a1] of the wave's row block(s) (fatchord_version.py:203-209 without the
    // x_{t-1} column), formed from the mel (or, wrnn_options.mel_stage, from the x25 signal: the last up-sampling stage too) and the frame's
    // aux row -- wrnn_ring.h -- into the k-major layer 4
    auto cond_store = [&](const f32x4 &v, int blk, int soff) {
        const int vo = (LU * blk + 4 * kq) * 64 + fi * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (loc) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[e]), xrs, vo + 64 * e, soff, 0);
            else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[e]), xrs, vo + 64 * e, soff, 16 /* sc1 */);
        }
    };
    auto cond_step = [&](int tt) {
        const int p = SEGT[fi] + tt;
        const bool valid = live && p < SEGT[SEG + fi];
        const int fr = magic ? (int)(__umulhi((unsigned)p, magic) >> mshift) : p / hop;
        float4 v[7];
        if (mel_stage) {
            const int j = p + SEGT[3 * SEG + fi];
            const int row = j / LAST_SCALE;
            cond_inputs_rows(mels_up + (size_t)(row - 1) * MEL, mel_coef + 3 * (j - row * LAST_SCALE), aux_fr + (size_t)fr * (4 * AUX), valid, lane, v);
        } else cond_inputs(mels_up + (size_t)p * MEL, aux_fr + (size_t)fr * (4 * AUX), valid, lane, v);
        const int so = cbase + 4 * DLAYERB + (tt & (DRING - 1)) * XTB;
        CondTile c1;
#pragma unroll
        for (int kk = 0; kk < CK; ++kk) c1.w[kk] = CT0[kk * 64 + lane];
#pragma unroll
        for (int i = 0; i < 4; ++i) c1.bias[i] = CT0[(CK + i) * 64 + lane];
        cond_store(cond_mfma(c1, v), cblk0, so);
        if (cond2) {
#pragma unroll
            for (int kk = 0; kk < CK; ++kk) c1.w[kk] = CT1[kk * 64 + lane];
#pragma unroll
            for (int i = 0; i < 4; ++i) c1.bias[i] = CT1[(CK + i) * 64 + lane];
            cond_store(cond_mfma(c1, v), cblk1, so);
        }
        SPX(10);
    };
    // rnn2's workgroup 0: fc3 (30 x 512: two 16-row tiles in A-fragment order, in LDS) + the mixture-of-logistics sampling of step t
    auto sample = [&]() {
        const int sb = cbase + (t & (DRING - 1)) * XTB;
        // x[r] = this wave's fragments of y2 -- or (fold3) the tile-0 partial logits of the fc2 row blocks 8 w + r (rows 4 kq + e of segment fi: the
        // accumulators' layout, the same addresses); x1[r]: their tile-1 partial logits (layer 16)
        u32x4 x[8], x1[8];
        auto ask = [&] {
#pragma unroll
            for (int r = 0; r < 8; ++r) x[r] = __builtin_amdgcn_raw_buffer_load_b128(xrs, voff_frag + r * 1024, sb + 3 * DLAYERB, 16 /* sc1 */);
            if (fold3) {
#pragma unroll
                for (int r = 0; r < 8; ++r) x1[r] = __builtin_amdgcn_raw_buffer_load_b128(xrs, voff_frag + r * 1024, sb + 16 * DLAYERB, 16 /* sc1 */);
            } else {
#pragma unroll
                for (int r = 0; r < 8; ++r) x1[r] = u32x4{0u, 0u, 0u, 0u};
            }
        };
        ask();
        const int su = tid >> 4, sm = tid & 15;         // sampling role: 16-lane row = segment su, lane sm = mixture
        const float *nrow = noise_pre + (size_t)(t - noise_t0) * 11 * Nall;
        const int suc = su < nb ? su : nb - 1;
        const float nz0 = nrow[(size_t)(b0g + suc) * 10 + (sm < 10 ? sm : 9)];
        const float nz1 = nrow[(size_t)10 * Nall + b0g + suc];
        if (__builtin_expect(!(frag_there(x, live) && frag_there(x1, live)), 0))
            wait_for([&] { return frag_there(x, live) && frag_there(x1, live); }, ask, status, dead, 0x750u, t);
        SPX(8);
        float *PW = PART + pp * (NW * 2 * 256);
        if (fold3) {
            f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                s0 += f32x4{__uint_as_float(x[r].x), __uint_as_float(x[r].y), __uint_as_float(x[r].z), __uint_as_float(x[r].w)};
                s1 += f32x4{__uint_as_float(x1[r].x), __uint_as_float(x1[r].y), __uint_as_float(x1[r].z), __uint_as_float(x1[r].w)};
            }
            put_partial<2>(PW, w, 0, lane, s0);
            put_partial<2>(PW, w, 1, lane, s1);
        } else {
            float b[32];
            frag_to_b(x, b);
            put_partial<2>(PW, w, 0, lane, mfma1_lds(F3 + frag_off(w, 0, lane), b));
            put_partial<2>(PW, w, 1, lane, mfma1_lds(F3 + XT + frag_off(w, 0, lane), b));
        }
        lds_barrier();
        if (dbgl && pj < nb) {                          // test hook: the 30 logits of every segment (thread: rows pu and 16 + pu, segment pj)
            dbgl[((size_t)t * Nall + b0g + pj) * C + pu] = get_partial<2>(PW, 0, pu, pj) + b3a;
            if (pu < 14) dbgl[((size_t)t * Nall + b0g + pj) * C + 16 + pu] = get_partial<2>(PW, 1, pu, pj) + b3b;
        }
        {   // utils/distribution.py:102-121: lane sm < 10 of a 16-lane row sums the partial tiles of ITS mixture logit (row sm of segment su) straight
            // from LDS -- no second pass through LDS, no second barrier --, Gumbel-max over the row, then lane 0 fetches mean and log-scale of the winner
            float best = (sm < 10) ? mol_gumbel_pre(get_partial<2>(PW, 0, sm < 10 ? sm : 0, su) + fc3b[sm < 10 ? sm : 0], nz0) : -INFINITY;
            int bidx = sm;
            argmax_row16(best, bidx);
            if (sm == 0 && su < nb) {
                const float mean = get_partial<2>(PW, (10 + bidx) >> 4, (10 + bidx) & 15, su) + fc3b[10 + bidx];
                const float ls = get_partial<2>(PW, (20 + bidx) >> 4, (20 + bidx) & 15, su) + fc3b[20 + bidx];
                float xv = mol_sample_pre(mean, ls, nz1);
                outp[(size_t)(b0g + su) * Tall + t] = xv;
                if (forcex) xv = forcex[(size_t)(b0g + su) * Tall + t];
                const u32x2 q = {__float_as_uint(xv), (unsigned)t + 1u};          // one 8-byte word {x_t, tag}: its own flag, two entries, no re-arm
                __builtin_amdgcn_raw_buffer_store_b64(q, xrs, su * 8, cbase + 7 * DLAYERB + (t & 1) * XTB, 16 /* sc1 */);
            }
        }
        pp ^= 1;
        SPX(9);
    };

    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    if (PROF && tid == 0) plast = __builtin_amdgcn_s_memtime();
    if constexpr (LA) {
        front_issue(T0);
        front_finish(T0);
    } else if (cond_wg) {                               // the two (FCS: three) steps a launch starts with; every later cI is formed CLEAD steps ahead
        cond_step(T0);
        if (T0 + 1 < T1) cond_step(T0 + 1);
        if (FCS && T0 + 2 < T1) cond_step(T0 + 2);
    }
    // FCS: cI THREE steps ahead -- its readers (end of step t + 2) have seen x_{t+1}, i.e. every wave of the cluster has passed the top of step t + 1 =
    // the forming wave's drain; two ahead leaned on "y1(t + 1) needed every x2(t + 1)", which a gathered fc stage does not give (SparseFcSim)
    constexpr int CLEAD = FCS ? 3 : 2;
    for (; t < T1; ++t) {
        if (PROF && tid == 0) PROFL[15] += 1;
        // ring hygiene: last step's re-arm stores (and, rnn1, the cI formed at its end) are out before anything of this step is published
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (FCS) lds_barrier();               // the four waves of a workgroup start a step together (they no longer meet in a dense fc stage): SparseFcSim
        if constexpr (LA) {
            // Off the chain, placed where this workgroup waits anyway (profiles/r05b .. r05d_sparse_phase_clocks.json): gh(t + 1) while x1 -> rnn2
            // -> x2 is under way, W_ih . cI(t + 1) under the sampling of step t.
            back_a();
            gh_stage();                                 // (needs h1(t) of every rnn1 workgroup: one hop behind the publication above)
            fc(I1{});
            fc(I2{});
            if (t + 1 < T1) { front_issue(t + 1); front_finish(t + 1); }
        } else {
            // gh(t + 1) is needed at the cell of step t + 1: behind fc2 (and, in the sampling workgroup, behind the sampling) it sits in the
            // wait for x1(t + 1); between fc1 and fc2 its gather outlasted y1's hop and held up y2 (r05b phase clocks).  The non-sampling
            // workgroups then form cI(t + 2): it overwrites cI(t - 2), gathered by every rnn1 workgroup at the end of its step t - 3; it is
            // drained at the top of step t + 1, before x2(t + 1) goes out, and gathered by workgroups that have polled y1(t + 1), which needed
            // x2(t + 1) of every rnn2 workgroup (header).
            gates_b();
            fc(I1{});
            fc(I2{});
            if (sampler) sample();
            gh_stage();
            if (cond_wg && t + CLEAD < T1) cond_step(t + CLEAD);
        }
    }
    if (PROF && tid == 0 && a.prof) {
        for (int k = 0; k < 16; ++k) a.prof[(size_t)(blockIdx.x & 255) * 32 + k] += PROFL[k];
    }
    // ---- what the next launch of this round needs: h and gh(T1) of every (unit, segment), rnn1: x_{T1-1}; the workgroups that form cI leave the sentinel in the cI
    //      entries of steps T1 and T1 + 1 (the next launch polls its first two steps)
    *reinterpret_cast<float4 *>(state_wg + tid * 16) = make_float4(h[0], h[1], h[2], h[3]);
    *reinterpret_cast<float4 *>(state_wg + tid * 16 + 4) = make_float4(ghr[0], ghr[1], ghr[2], ghr[3]);
    *reinterpret_cast<float4 *>(state_wg + tid * 16 + 8) = make_float4(ghz[0], ghz[1], ghz[2], ghz[3]);
    *reinterpret_cast<float4 *>(state_wg + tid * 16 + 12) = make_float4(ghn[0], ghn[1], ghn[2], ghn[3]);
    if constexpr (LA) {
        const int sx = cbase + 7 * DLAYERB + ((T1 - 1) & 1) * XTB;
        u32x2 xq = __builtin_amdgcn_raw_buffer_load_b64(xrs, fi * 8, sx, 16 /* sc1 */);
        wait_for([&] { return !__any(live && xq.y != (unsigned)T1); }, [&] { xq = __builtin_amdgcn_raw_buffer_load_b64(xrs, fi * 8, sx, 16 /* sc1 */); },
                 status, dead, 0x761u, T1);
        if (tid < SEG) state_wg[NT * 16 + tid] = live ? __uint_as_float(xq.x) : 0.f;
    }
    if (cond_wg) {                                      // (a block of a k-major layer is 1 KB: one 16-byte store per lane)
        const u32x4 q = {SENT, SENT, SENT, SENT};
#pragma unroll
        for (int e = 0; e < CLEAD; ++e) {
            store16(q, cblk0 * 1024 + lane * 16, cbase + 4 * DLAYERB + ((T1 + e) & (DRING - 1)) * XTB);
            if (cond2) store16(q, cblk1 * 1024 + lane * 16, cbase + 4 * DLAYERB + ((T1 + e) & (DRING - 1)) * XTB);
        }
    }
}

// Grid = 16 clusters x 16 workgroups of 256 threads (one per CU), cooperative launch.  Placement (speed only, verified at run time): block
// b is observed to run on XCD b % 8 and the blocks of an XCD to be dealt round-robin over its 32 CUs.  XCD x hosts clusters x (its CUs
// 0-15) and 8 + x (CUs 16-31); CU c of a cluster: c / 8 = rnn1 | rnn2, unit block c % 8.  Group g of a round runs on cluster g: the first
// eight groups take one cluster on every XCD.
template <int NBP, bool FCS, bool PROF>
__global__ __launch_bounds__(NT, 1) void wrnn_sparse_kernel(const LoopArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    check_kind(a);
    const int b = blockIdx.x;
    const int xcd = b % 8, q = b / 8;
    const int cl = xcd + 8 * (q >> 4);
    const int cu = q & 15;
    if (cl >= a.NG) return;                             // a cluster without a group of this round
    // ---- placement handshake (as wrnn_duo.hip): a cluster seen on ONE XCD exchanges everything through that XCD's L2 with plain stores
    bool loc = false;
    {
        int *TAB = reinterpret_cast<int *>(smem) + sp_lds().off_misc;
        const int tid = threadIdx.x;
        unsigned *tab = a.xcc_tab + cl * SPWG;
        if (tid == 0) {
            const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xFu;          // HW_REG_XCC_ID
            __hip_atomic_store(tab + cu, xcc + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!PROF && a.prof && (a.tuning & 64)) {                                                 // placement read-out (test / profiling hook)
                const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
                a.prof[blockIdx.x] = ((u64)xcc << 32) | hw | ((u64)(unsigned)(cu | (cl << 8)) << 40);
            }
        }
        unsigned v = 1u;
        if (tid < SPWG) {
            unsigned spins = 0;
            v = ld_agent32(tab + tid);
            while (v == 0u && ++spins < 200000u) {
                __builtin_amdgcn_s_sleep(2);
                v = ld_agent32(tab + tid);
            }
            TAB[tid] = (int)v;
        }
        __syncthreads();
        const int ok = (tid < SPWG) ? (v != 0u && (int)v == TAB[0]) : 1;
        loc = __syncthreads_and(ok) != 0;
        if (a.tuning & 256) loc = false;                // A/B: everything written through
        __syncthreads();
    }
    if (cu < 8) sp_role<NBP, FCS, true, PROF>(a, smem, cl, cl, cu, cu, loc);
    else sp_role<NBP, FCS, false, PROF>(a, smem, cl, cl, cu - 8, cu, loc);
}

// clusters of 16 CUs: the kernel's block -> role map is written for the whole 256-CU chip
int sparse_clusters(int n_cus) { return n_cus >= SPCLUSTERS * SPWG ? SPCLUSTERS : 0; }
size_t sparse_state_floats() { return (size_t)SPCLUSTERS * SPSTATE_CL; }
size_t sparse_xbuf_bytes() { return (size_t)SPCLUSTERS * DSLOTB; }      // the regions a launch touches: a prefix of the duo kernel's buffer
size_t sparse_lds_bytes() { return (size_t)sp_lds().total * sizeof(float); }

hipError_t launch_sparse(const LoopArgs &args, int nbp, hipStream_t stream)
{
    if ((nbp != 48 && nbp != 64) || !args.fc3f || !args.u1 || !args.xcc_tab || !args.sp_vals || args.NG < 1 || args.NG > SPCLUSTERS) return hipErrorInvalidValue;
    const size_t lds = sparse_lds_bytes();
    const bool prof = args.prof && !(args.tuning & 64);              // phase clocks (wrnn_options.phase_clocks)
    const bool fcs = args.sp_fc_vals != nullptr;                     // the pack's Linear layers are block-sparse too: gathered fc stages
    const void *fn = fcs ? (nbp == 48 ? (prof ? (const void *)wrnn_sparse_kernel<48, true, true> : (const void *)wrnn_sparse_kernel<48, true, false>)
                                      : (prof ? (const void *)wrnn_sparse_kernel<64, true, true> : (const void *)wrnn_sparse_kernel<64, true, false>))
                         : (nbp == 48 ? (prof ? (const void *)wrnn_sparse_kernel<48, false, true> : (const void *)wrnn_sparse_kernel<48, false, false>)
                                      : (prof ? (const void *)wrnn_sparse_kernel<64, false, true> : (const void *)wrnn_sparse_kernel<64, false, false>));
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    LoopArgs a = args;
    void *params[] = {(void *)&a};
    return hipLaunchCooperativeKernel(fn, dim3(SPCLUSTERS * SPWG), dim3(NT), params, (unsigned)lds, stream);
}

}  // namespace wrnn
