// wrnn_pipe.hip -- the PIPELINED clustered persistent WaveRNN loop kernel (MOL) for MI355X (gfx950 / CDNA4).
//
// Same arithmetic, weight split and exchange protocol as wrnn_cluster.hip at U = 8 (4 clusters of 64 workgroups,
// each cluster holds one fp32 copy of the loop weights in registers; reference models/fatchord_version.py:201-241),
// but every cluster advances G INDEPENDENT groups of segments at once and interleaves them stage by stage:
//
//      for t:  S1(g0) S1(g1) S1(g2)  S2(g0) S2(g1) S2(g2)  S3(g0) ...  S6(g2)
//
// A stage of one group ends with publishing that group's granules; by the time the workgroup comes back to the
// same group (G-1 other units later) its all-gather has landed, so the ~3-4 us exchange latency that bounds the
// single-group kernels is hidden behind MFMA work on the other groups.
//
// What had to move to make G groups fit one CU:
//   * LDS holds one activation tile per group (x = xi, then xi+h1, xi+h1+h2, y1, y2) plus ONE transient tile for
//     the freshly gathered h (used at once for the residual add and the W_hh.h product of the NEXT step, then free
//     for the next group).  G = 3 groups of <= 15 segments + the transient tile = 125 KB.
//   * the 30-row MOL fc3 no longer fits LDS (62 KB replicated): it is DISTRIBUTED -- workgroup wg < 60 of the cluster
//     computes one K-half (wg / 30) of logit row wg % 30 for all segments of the group (256-long VALU dot products over
//     the gathered half of y2) and the 30 x 2 partial logits per segment make a 5th, tiny all-gather (480 B per segment;
//     the consumer adds the two halves).  Sampling stays replicated (argmax as a 4-step DPP row all-reduce).
//   * cI(t+1) is fetched after the last poll of each group's step (vector loads return in order) from rows touched
//     into L2 one step earlier; the MOL noise arrives pre-transformed (wrnn_noise_mol_kernel); no per-group prefetch
//     registers exist.
//   * sweeps stream (8 loads in flight, slot i reloaded with piece i+8 as soon as it is consumed) or take the whole row
//     slice at once (NL = 16, default); the two row tiles of a GRU matrix share one set of B fragments (mfma_tile2).
// Register file: 10 weight tiles x 32 = 320 registers per lane, as in the U = 8 cluster kernel.
#include "wrnn_tiles.h"

namespace wrnn {

constexpr int PU = 8;                 // hidden units per workgroup
constexpr int PNWGC = H / PU;         // workgroups per cluster (64)
constexpr int PGR = 3 * PU;           // GRU gate rows per workgroup (24)
constexpr int PRT = 2;                // 16-row MFMA tiles per GRU matrix
constexpr int PNSLOT = 4;             // partial-tile slots per wave: 0,1 critical; 2,3 hidden-to-hidden
static_assert((PGR * SEG) % (NT / 2) == 0, "gh reduce split over waves 2,3");

template <int G>
struct PipeCfg {
    static constexpr int R = (G >= 3) ? 15 : 16;                     // segment rows per group tile
    static constexpr int TILE = R * LDC;
    static constexpr int GRP = 2 * PGR * SEG + 2 * PU * SEG + 3 * SEG; // per-group small state: GH1 GH2 HOWN1 HOWN2 XS POS LIM
    static constexpr int OFF_HS = 0;                                   // [R][LDC] transient gathered h
    static constexpr int OFF_ACT = OFF_HS + TILE;                      // [G][R][LDC]
    static constexpr int OFF_PART = OFF_ACT + G * TILE;                // [NW][PNSLOT][16][16]
    static constexpr int OFF_GRP = OFF_PART + NW * PNSLOT * 256;       // [G][GRP]
    static constexpr int OFF_LOG = OFF_GRP + G * GRP;                  // [SEG][32] logits of the group being sampled
    static constexpr int OFF_WI0 = OFF_LOG + SEG * 32;                 // [H]  I.weight[:,0]
    static constexpr int OFF_W3R = OFF_WI0 + H;                        // [H]  fc3 row owned by this workgroup (wg < 30)
    static constexpr int OFF_SCR = OFF_W3R + H;                        // [16 k-chunks][SEG] fc3 partial dot products
    static constexpr int OFF_BI1 = OFF_SCR + 16 * SEG;                 // [PGR] b_ih1, then b_hh1, b_hh2 of the owned rows
    static constexpr int OFF_BH1 = OFF_BI1 + PGR;
    static constexpr int OFF_BH2 = OFF_BH1 + PGR;
    static constexpr int OFF_GEO = OFF_BH2 + PGR;                      // [2*MAXG] ints: b0, nb of every slot
    static constexpr int OFF_PROF = ((OFF_GEO + 2 * MAXG + 3) / 4) * 4;   // [NPROF] u64 phase clocks (PROF builds)
    static constexpr int LDS_FLOATS = OFF_PROF + 2 * NPROF;
    static_assert(G >= 1 && G <= MAXG, "G in 1..MAXG");
    static_assert(LDS_FLOATS * 4 <= 160 * 1024, "LDS budget");
    static_assert(TILE % 4 == 0 && OFF_PART % 4 == 0 && OFF_GRP % 4 == 0 && OFF_WI0 % 4 == 0, "alignment");
};

// Phase clock (PROF builds only): thread 0 adds the shader cycles since the previous mark to phase k.
#define PH(k)                                                                  \
    do {                                                                       \
        if (PROF && tid == 0) {                                                \
            const u64 now_ = __builtin_amdgcn_s_memtime();                     \
            PROFL[k] += now_ - plast;                                          \
            plast = now_;                                                      \
        }                                                                      \
    } while (0)

// NL = 8 selects the streaming sweep, NL = 16 the whole-row sweep
template <bool ADD, int NL>
__device__ __forceinline__ bool pipe_sweep(__amdgpu_buffer_rsrc_t rs, int soff, unsigned tag, int nb, int tid, float *dst,
                                           float *acc, unsigned *status)
{
    if (NL == 8) return sweep_stream<ADD>(rs, soff, tag, nb, tid, dst, acc, status);
    return sweep_layer<ADD, 16>(rs, soff, tag, nb, tid, dst, acc, status);
}

// conditioning frame of segment j of a group at step t, from the group's LDS copy of the segment table
__device__ __forceinline__ int group_frame(const float *GP, int j, int t, int hop, int NF)
{
    const int *SP = reinterpret_cast<const int *>(GP + 2 * PGR * SEG + 2 * PU * SEG + SEG);
    const int p = SP[j] + t;
    return (p < SP[SEG + j]) ? (p / hop) : NF;
}

// G: groups in flight per cluster.  NL: sweep loads in flight per thread (16 or 8).  PROF: per-phase clocks.
template <int G, int NL, bool PROF>
__global__ __launch_bounds__(NT, 1) void wrnn_pipe_kernel(const LoopArgs a)
{
    using K = PipeCfg<G>;
    constexpr int R = K::R;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *HS = smem + K::OFF_HS, *PART = smem + K::OFF_PART, *LOG = smem + K::OFF_LOG, *WI0 = smem + K::OFF_WI0;
    float *W3R = smem + K::OFF_W3R, *SCR = smem + K::OFF_SCR;
    float *BI1 = smem + K::OFF_BI1, *BH1 = smem + K::OFF_BH1, *BH2 = smem + K::OFF_BH2;
    int *GEO = reinterpret_cast<int *>(smem + K::OFF_GEO);
    u64 *PROFL = reinterpret_cast<u64 *>(smem + K::OFF_PROF);
    u64 plast = 0;
    float touch = 0.f;                                  // in-flight L2 touch of a future cI block (S6)

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // cluster / workgroup-in-cluster: whole XCDs per cluster (block b runs on XCD b % 8; speed only)
    int cl, wg;
    const int ncl = gridDim.x / PNWGC;
    {
        const int b = blockIdx.x, nblk = gridDim.x;
        if (nblk % 8 == 0 && ncl >= 1 && 8 % ncl == 0) {
            const int xpc = 8 / ncl, per_xcd = nblk / 8;
            const int xcd = b % 8;
            cl = xcd / xpc;
            wg = (xcd % xpc) * per_xcd + b / 8;
        } else {
            cl = b / PNWGC;
            wg = b % PNWGC;
        }
    }
    const int fi = lane & 15, kq = lane >> 4;
    const int kbase_lane = KCH * w + 4 * kq;
    const int Btot = a.Btot, T = a.T, C = a.C, NG = a.NG;
    const int er = tid >> 4, ec = tid & 15;             // elementwise / sweep role: row er, column pairs own_col(i, ec)
    const int pu = tid >> 4, pj = tid & 15;             // pointwise role: (owned unit pu, segment pj) for tid < 16*PU
    const bool pw_thread = tid < 16 * PU;
    const int prow = PU * wg + (pu % PU);
    const bool fc3_wg = wg < 60;                        // this workgroup owns one K-half of logit row wg % 30
    const int f3row = wg % 30, f3half = (wg / 30) & 1;

    // ---- one-time: weight slice -> register-resident MFMA A fragments (row ri = gate*PU + u, tile ri/16) ------
    float A_ih1[PRT][AF], A_hh1[PRT][AF], A_ih2[PRT][AF], A_hh2[PRT][AF], A_fc1[AF], A_fc2[AF];
#pragma unroll
    for (int rt = 0; rt < PRT; ++rt) {
        const int ri = 16 * rt + fi;
        const bool vg = ri < PGR;
        const int grow = (ri / PU) * H + PU * wg + (ri % PU);
        load_afrag(A_ih1[rt], a.w_ih1, H, grow, vg, kbase_lane);
        load_afrag(A_hh1[rt], a.w_hh1, H, grow, vg, kbase_lane);
        load_afrag(A_ih2[rt], a.w_ih2, H + AUX, grow, vg, kbase_lane);
        load_afrag(A_hh2[rt], a.w_hh2, H, grow, vg, kbase_lane);
    }
    {
        const bool vf = fi < PU;
        const int frow = PU * wg + fi;
        load_afrag(A_fc1, a.fc1_w, H + AUX, frow, vf, kbase_lane);
        load_afrag(A_fc2, a.fc2_w, H + AUX, frow, vf, kbase_lane);
    }
    for (int q = tid; q < K::LDS_FLOATS; q += NT) smem[q] = 0.f;       // no NaN bit patterns in never-written rows
    __syncthreads();
    WI0[2 * tid] = a.I_w0[2 * tid];
    WI0[2 * tid + 1] = a.I_w0[2 * tid + 1];
    if (fc3_wg) { W3R[2 * tid] = a.fc3_w[(size_t)f3row * H + 2 * tid]; W3R[2 * tid + 1] = a.fc3_w[(size_t)f3row * H + 2 * tid + 1]; }
    const float b3 = (fc3_wg && f3half == 0) ? a.fc3_b[f3row] : 0.f;       // the bias rides on the k < 256 partial
    if (tid < PGR) {
        const int grow = (tid / PU) * H + PU * wg + (tid % PU);
        BI1[tid] = a.b_ih1[grow];
        BH1[tid] = a.b_hh1[grow];
        BH2[tid] = a.b_hh2[grow];
    }

    const __amdgpu_buffer_rsrc_t grs = make_rsrc(a.gran, GRAN_WORDS * 8);
    constexpr int LAYER_BYTES = SEG * H * 8;
    constexpr int SLOT_BYTES = NGRAN * LAYER_BYTES;
    const int soff_cl = cl * MAXG * SLOT_BYTES;

    if (PROF) plast = __builtin_amdgcn_s_memtime();
    unsigned tagbase = 0u;
    for (int round = 0;; ++round, tagbase += (unsigned)T) {
        const int gfirst = cl + ncl * (round * G);
        if (gfirst >= NG) break;
        int nact = 0;                                                    // active slots of this round (wave-uniform)
#pragma unroll
        for (int i = 0; i < G; ++i)
            if (gfirst + ncl * i < NG) nact = i + 1;
        __syncthreads();                                                 // previous round's LDS reads are done
        if (tid < G) {
            const int g = gfirst + ncl * tid;
            int b0 = 0, nb = 0;
            if (g < NG) {
                b0 = (int)(((long)g * Btot) / NG);
                nb = (int)(((long)(g + 1) * Btot) / NG) - b0;
            }
            GEO[2 * tid] = b0;
            GEO[2 * tid + 1] = nb;
        }
        // ---- state init (fatchord_version.py:194-196: h1 = h2 = 0, x = 0): gh = W_hh . 0 + b_hh = b_hh ----
        __syncthreads();
#pragma unroll 1
        for (int i = 0; i < nact; ++i) {
            float *GP = smem + K::OFF_GRP + i * K::GRP;
            float *ACT = smem + K::OFF_ACT + i * K::TILE;
            const int b0 = GEO[2 * i], nb = GEO[2 * i + 1];
            for (int q = tid; q < PGR * SEG; q += NT) { GP[q] = BH1[q >> 4]; GP[PGR * SEG + q] = BH2[q >> 4]; }
            if (tid < 2 * PU * SEG) GP[2 * PGR * SEG + tid] = 0.f;                  // HOWN1 + HOWN2
            if (tid < SEG) {
                GP[2 * PGR * SEG + 2 * PU * SEG + tid] = 0.f;                        // XS
                int *SP = reinterpret_cast<int *>(GP + 2 * PGR * SEG + 2 * PU * SEG + SEG);
                const int sc = b0 + (tid < nb ? tid : nb - 1);                       // segment table of the group -> LDS
                SP[tid] = a.seg_pos[sc];
                SP[SEG + tid] = a.seg_lim[sc];
            }
            if (er < R) {                                                            // xi(0) = cI(0)  (x_{-1} = 0)
                const int erc = er < nb ? er : nb - 1;
                const float *crow = a.cI + ((size_t)0 * Btot + b0 + erc) * H + 2 * ec;
#pragma unroll
                for (int c = 0; c < 16; ++c)
                    *reinterpret_cast<float2 *>(ACT + er * LDC + own_col(c, ec)) = *reinterpret_cast<const float2 *>(crow + 32 * c);
            }
            {   // warm L2 with the conditioning block of step 1
                asm volatile("" ::"v"(touch));
                const int line = (tid < 16 * nb) ? tid : 0;
                touch = a.cI[((size_t)(T > 1 ? 1 : 0) * Btot + b0) * H + 32 * line];
            }
        }
        __syncthreads();

        for (int t = 0; t < T; ++t) {
            const unsigned tag = tagbase + (unsigned)t + 1u;

            // =========================== S1: GRU1 (fatchord_version.py:210) ===========================
#pragma unroll 1
            for (int i = 0; i < nact; ++i) {
                const int nb = GEO[2 * i + 1];
                float *GP = smem + K::OFF_GRP + i * K::GRP;
                float *ACT = smem + K::OFF_ACT + i * K::TILE;
                float *GH1 = GP, *HOWN1 = GP + 2 * PGR * SEG;
                u64 *G1 = a.gran + (size_t)(cl * MAXG + i) * NGRAN * SEG * H + 0 * SEG * H;
                __syncthreads();                                         // PART free; ACT written by S6 visible
                PH(0);
                {
                    f32x4 o0, o1;
                    mfma_tile2(A_ih1[0], A_ih1[1], ACT + fi * LDC + kbase_lane, o0, o1);
                    put_partial<PNSLOT>(PART, w, 0, lane, o0);
                    put_partial<PNSLOT>(PART, w, 1, lane, o1);
                }
                PH(1);
                __syncthreads();
                PH(2);
                if (pw_thread && pj < nb) {
                    const float gir = get_partial<PNSLOT>(PART, 0, 0 * PU + pu, pj) + BI1[0 * PU + pu];
                    const float giz = get_partial<PNSLOT>(PART, 0, 1 * PU + pu, pj) + BI1[1 * PU + pu];
                    const float gin = get_partial<PNSLOT>(PART, 0, 2 * PU + pu, pj) + BI1[2 * PU + pu];
                    const float hn = gru_update(gir, giz, gin, GH1[(0 * PU + pu) * SEG + pj], GH1[(1 * PU + pu) * SEG + pj],
                                                GH1[(2 * PU + pu) * SEG + pj], HOWN1[pu * SEG + pj]);
                    HOWN1[pu * SEG + pj] = hn;
                    publish(G1, tag, pj, prow, hn);
                }
                PH(3);
            }

            // =========================== S2: GRU2 (:212-214) ==========================================
#pragma unroll 1
            for (int i = 0; i < nact; ++i) {
                const int nb = GEO[2 * i + 1];
                float *GP = smem + K::OFF_GRP + i * K::GRP;
                float *ACT = smem + K::OFF_ACT + i * K::TILE;
                float *GH1 = GP, *GH2 = GP + PGR * SEG, *HOWN2 = GP + 2 * PGR * SEG + PU * SEG;
                u64 *G2 = a.gran + (size_t)(cl * MAXG + i) * NGRAN * SEG * H + 1 * SEG * H;
                const bool is_pw = pw_thread && pj < nb;
                // unconditional (clamped) loads: a load inside a divergent branch is waited for at the branch end
                const int f2 = group_frame(GP, pj, t, a.hop, a.NF);
                const float c2r = a.c2f[(size_t)f2 * 3 * H + prow];
                const float c2z = a.c2f[(size_t)f2 * 3 * H + H + prow];
                const float c2n = a.c2f[(size_t)f2 * 3 * H + 2 * H + prow];
                // h1(t) -> HS ; ACT = xi + h1 (:212)
                PH(8);
                bool ok = pipe_sweep<true, NL>(grs, soff_cl + i * SLOT_BYTES + 0 * LAYER_BYTES, tag, nb, tid, HS, ACT, a.status);
                if (!ok) report_failure(a.status, 0x300u | 1u, blockIdx.x, t, tid);
                PH(4);
                if (__syncthreads_or(!ok)) return;
                PH(5);
                {
                    f32x4 o0, o1;
                    mfma_tile2(A_ih2[0], A_ih2[1], ACT + fi * LDC + kbase_lane, o0, o1);
                    put_partial<PNSLOT>(PART, w, 0, lane, o0);
                    put_partial<PNSLOT>(PART, w, 1, lane, o1);
                    mfma_tile2(A_hh1[0], A_hh1[1], HS + fi * LDC + kbase_lane, o0, o1);   // gh1(t+1) = W_hh1 . h1(t)
                    put_partial<PNSLOT>(PART, w, PRT + 0, lane, o0);
                    put_partial<PNSLOT>(PART, w, PRT + 1, lane, o1);
                }
                PH(6);
                __syncthreads();
                PH(7);
                if (is_pw) {
                    const float gir = get_partial<PNSLOT>(PART, 0, 0 * PU + pu, pj) + c2r;
                    const float giz = get_partial<PNSLOT>(PART, 0, 1 * PU + pu, pj) + c2z;
                    const float gin = get_partial<PNSLOT>(PART, 0, 2 * PU + pu, pj) + c2n;
                    const float hn = gru_update(gir, giz, gin, GH2[(0 * PU + pu) * SEG + pj], GH2[(1 * PU + pu) * SEG + pj],
                                                GH2[(2 * PU + pu) * SEG + pj], HOWN2[pu * SEG + pj]);
                    HOWN2[pu * SEG + pj] = hn;
                    publish(G2, tag, pj, prow, hn);
                }
                if (tid >= NT / 2) {                                     // waves 2,3 (idle during the pointwise math of waves 0,1)
#pragma unroll
                    for (int q0 = 0; q0 < (PGR * SEG) / (NT / 2); ++q0) {
                        const int q = (tid - NT / 2) + (NT / 2) * q0;
                        GH1[q] = get_partial<PNSLOT>(PART, PRT, q >> 4, q & 15) + BH1[q >> 4];
                    }
                }
                PH(8);
            }

            // =========================== S3: fc1 + relu (:216-218) ====================================
#pragma unroll 1
            for (int i = 0; i < nact; ++i) {
                const int nb = GEO[2 * i + 1];
                float *GP = smem + K::OFF_GRP + i * K::GRP;
                float *ACT = smem + K::OFF_ACT + i * K::TILE;
                float *GH2 = GP + PGR * SEG;
                u64 *G3 = a.gran + (size_t)(cl * MAXG + i) * NGRAN * SEG * H + 2 * SEG * H;
                const bool is_pw = pw_thread && pj < nb;
                const float c3v = a.c3f[(size_t)group_frame(GP, pj, t, a.hop, a.NF) * H + prow];
                // h2(t) -> HS ; ACT = x1 + h2 (:216)
                PH(8);
                bool ok = pipe_sweep<true, NL>(grs, soff_cl + i * SLOT_BYTES + 1 * LAYER_BYTES, tag, nb, tid, HS, ACT, a.status);
                if (!ok) report_failure(a.status, 0x300u | 2u, blockIdx.x, t, tid);
                PH(4);
                if (__syncthreads_or(!ok)) return;
                PH(5);
                put_partial<PNSLOT>(PART, w, 0, lane, mfma_tile_pre(A_fc1, ACT + fi * LDC + kbase_lane));
                {
                    f32x4 o0, o1;
                    mfma_tile2(A_hh2[0], A_hh2[1], HS + fi * LDC + kbase_lane, o0, o1);   // gh2(t+1) = W_hh2 . h2(t)
                    put_partial<PNSLOT>(PART, w, PRT + 0, lane, o0);
                    put_partial<PNSLOT>(PART, w, PRT + 1, lane, o1);
                }
                PH(6);
                __syncthreads();
                PH(7);
                if (is_pw) publish(G3, tag, pj, prow, fmaxf(get_partial<PNSLOT>(PART, 0, pu, pj) + c3v, 0.f));
                if (tid >= NT / 2) {                                     // waves 2,3 (idle during the pointwise math of waves 0,1)
#pragma unroll
                    for (int q0 = 0; q0 < (PGR * SEG) / (NT / 2); ++q0) {
                        const int q = (tid - NT / 2) + (NT / 2) * q0;
                        GH2[q] = get_partial<PNSLOT>(PART, PRT, q >> 4, q & 15) + BH2[q >> 4];
                    }
                }
                PH(8);
            }

            // =========================== S4: fc2 + relu (:220-221) ====================================
#pragma unroll 1
            for (int i = 0; i < nact; ++i) {
                const int nb = GEO[2 * i + 1];
                float *GP = smem + K::OFF_GRP + i * K::GRP;
                float *ACT = smem + K::OFF_ACT + i * K::TILE;
                u64 *G4 = a.gran + (size_t)(cl * MAXG + i) * NGRAN * SEG * H + 3 * SEG * H;
                const bool is_pw = pw_thread && pj < nb;
                const float c4v = a.c4f[(size_t)group_frame(GP, pj, t, a.hop, a.NF) * H + prow];
                PH(8);
                bool ok = pipe_sweep<false, NL>(grs, soff_cl + i * SLOT_BYTES + 2 * LAYER_BYTES, tag, nb, tid, ACT, nullptr, a.status);   // ACT <- y1
                if (!ok) report_failure(a.status, 0x300u | 3u, blockIdx.x, t, tid);
                PH(4);
                if (__syncthreads_or(!ok)) return;
                PH(5);
                put_partial<PNSLOT>(PART, w, 0, lane, mfma_tile_pre(A_fc2, ACT + fi * LDC + kbase_lane));
                PH(6);
                __syncthreads();
                PH(7);
                if (is_pw) publish(G4, tag, pj, prow, fmaxf(get_partial<PNSLOT>(PART, 0, pu, pj) + c4v, 0.f));
                PH(8);
            }

            // =========================== S5: fc3 (:223): workgroup wg < 60 = K-half wg/30 of logit row wg%30 ==========
            if (fc3_wg) {                                               // workgroup-uniform
#pragma unroll 1
                for (int i = 0; i < nact; ++i) {
                    const int nb = GEO[2 * i + 1];
                    float *ACT = smem + K::OFF_ACT + i * K::TILE;
                    u64 *G5 = a.gran + (size_t)(cl * MAXG + i) * NGRAN * SEG * H + 4 * SEG * H;
                    bool ok = sweep_half(grs, soff_cl + i * SLOT_BYTES + 3 * LAYER_BYTES, tag, nb, tid, ACT, f3half, a.status);   // ACT <- half of y2
                    if (!ok) report_failure(a.status, 0x300u | 4u, blockIdx.x, t, tid);
                    PH(4);
                    if (__syncthreads_or(!ok)) return;
                    PH(5);
                    {   // thread (segment pj, k-chunk pu): 16 terms of this half of logit[f3row][pj]
                        const int k0 = 256 * f3half + 16 * pu;
                        const float *xr = ACT + (pj < R ? pj : R - 1) * LDC + k0;
                        const float *wr = W3R + k0;
                        float s = 0.f;
#pragma unroll
                        for (int k = 0; k < 16; k += 4) {
                            const float4 x4 = *reinterpret_cast<const float4 *>(xr + k);
                            const float4 w4 = *reinterpret_cast<const float4 *>(wr + k);
                            s = fmaf(w4.x, x4.x, s); s = fmaf(w4.y, x4.y, s); s = fmaf(w4.z, x4.z, s); s = fmaf(w4.w, x4.w, s);
                        }
                        SCR[pu * SEG + pj] = s;
                    }
                    __syncthreads();
                    if (tid < nb) {
                        float s = SCR[tid];
#pragma unroll
                        for (int kc = 1; kc < 16; ++kc) s += SCR[kc * SEG + tid];
                        publish(G5, tag, tid, 2 * f3row + f3half, s + b3);   // granule pair (2 row, 2 row + 1) = the two halves
                    }
                    PH(9);
                }
            }

            // =========================== S6: sampling (utils/distribution.py:102-121) + xi(t+1) ========
#pragma unroll 1
            for (int i = 0; i < nact; ++i) {
                float *GP = smem + K::OFF_GRP + i * K::GRP;
                float *ACT = smem + K::OFF_ACT + i * K::TILE;
                float *XS = GP + 2 * PGR * SEG + 2 * PU * SEG;
                const int nb = GEO[2 * i + 1], b0 = GEO[2 * i];
                // this step's sampling noise: clamped, unconditional loads (rows >= nb are never sampled)
                const float *nrow = a.noise_pre + (size_t)t * 11 * Btot;       // log(-log u1) / log u2 - log(1-u2)
                const int puc = pu < nb ? pu : nb - 1;
                const float nz0 = nrow[(b0 + puc) * 10 + (pj < 10 ? pj : 9)];
                const float nz1 = nrow[10 * Btot + b0 + puc];
                PH(10);
                // gather the 30 logits of every segment: granule pair c = (k < 256 partial + bias, k >= 256 partial) of row c;
                // thread (segment er, ec) reads rows ec and ec + 16
                {
                    bool ok = true;
                    if (er < nb) {
                        const int soff = soff_cl + i * SLOT_BYTES + 4 * LAYER_BYTES;
#pragma unroll
                        for (int h2 = 0; h2 < 2; ++h2) {
                            const int c = ec + 16 * h2;
                            if (c < 30 && ok) {
                                const int voff = er * (H * 8) + c * 16;
                                unsigned spins = 0;
                                u32x4 x;
                                for (;;) {
                                    x = __builtin_amdgcn_raw_buffer_load_b128(grs, voff, soff, 16 /* sc1 */);
                                    if (x.y == tag && x.w == tag) break;
                                    ++spins;
                                    if ((spins & 255u) == 0u) {
                                        if (spins > SPIN_LIMIT || ld_agent32(a.status) != 0u) { ok = false; break; }
                                    }
                                    __builtin_amdgcn_s_sleep(1);
                                }
                                const float lg = __uint_as_float(x.x) + __uint_as_float(x.z);
                                LOG[er * 32 + c] = lg;
                                if (a.dbg_logits && wg == 0 && ok) a.dbg_logits[((size_t)t * Btot + b0 + er) * C + c] = lg;
                            }
                        }
                    }
                    if (!ok) report_failure(a.status, 0x300u | 5u, blockIdx.x, t, tid);
                    PH(11);
                    if (__syncthreads_or(!ok)) return;
                    PH(12);
                }
                // cI of the next step (row er, owned columns).  Issued AFTER the last poll of this group's step: vector
                // loads return in order, so an earlier issue would put the (cold) conditioning rows in front of every
                // poll.  The rows are L2-warm: the block was touched one step ago (below); the sampling math hides the rest.
                float2 cn[16];
                {
                    const int tn = (t + 1 < T) ? t + 1 : t;
                    const int erc = er < nb ? er : nb - 1;
                    const float *crow = a.cI + ((size_t)tn * Btot + b0 + erc) * H + 2 * ec;
#pragma unroll
                    for (int c = 0; c < 16; ++c) cn[c] = *reinterpret_cast<const float2 *>(crow + 32 * c);
                    // touch the block of step t+2 (nb rows x 2 KB = one 128-B line per thread) so it is in this XCD's L2
                    // by the time it is needed; the previous touch is retired first (issued one unit ago)
                    asm volatile("" ::"v"(touch));
                    const int tt = (t + 2 < T) ? t + 2 : T - 1;
                    const int line = (tid < 16 * nb) ? tid : 0;
                    touch = a.cI[((size_t)tt * Btot + b0) * H + 32 * line];
                }
                {   // 16-lane group = one segment (pu), lane pj = mixture
                    float best = (pj < 10) ? mol_gumbel_pre(LOG[pu * 32 + pj], nz0) : -INFINITY;
                    int bidx = pj;
                    argmax_row16(best, bidx);
                    if (pj == 0 && pu < nb) {
                        float x = mol_sample_pre(LOG[pu * 32 + 10 + bidx], LOG[pu * 32 + 20 + bidx], nz1);
                        if (wg == 0) a.out[(size_t)(b0 + pu) * T + t] = x;
                        if (a.force_x) x = a.force_x[(size_t)(b0 + pu) * T + t];
                        XS[pu] = x;
                    }
                }
                PH(13);
                __syncthreads();
                PH(14);
                if (er < R) {                                           // xi(t+1) = W_I[:,0] * x_t + cI(t+1)  (:208-209)
                    const float xs = XS[er];
#pragma unroll
                    for (int c = 0; c < 16; ++c) {
                        const float2 wv = *reinterpret_cast<const float2 *>(WI0 + own_col(c, ec));
                        *reinterpret_cast<float2 *>(ACT + er * LDC + own_col(c, ec)) = make_float2(fmaf(wv.x, xs, cn[c].x), fmaf(wv.y, xs, cn[c].y));
                    }
                }
                PH(15);
            }
        }
    }
    asm volatile("" ::"v"(touch));
    if (PROF && tid == 0 && a.prof && blockIdx.x < MAXWG) {
#pragma unroll
        for (int k = 0; k < NPROF; ++k) a.prof[(size_t)blockIdx.x * NPROF + k] = PROFL[k];
    }
}

template <int G, int NL, bool PROF>
static hipError_t launch_pipe_t(const LoopArgs &args, int ncl, hipStream_t stream)
{
    using K = PipeCfg<G>;
    const size_t lds = (size_t)K::LDS_FLOATS * sizeof(float);
    hipError_t e = hipFuncSetAttribute((const void *)wrnn_pipe_kernel<G, NL, PROF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    LoopArgs a = args;
    void *params[] = {(void *)&a};
    return hipLaunchCooperativeKernel((const void *)wrnn_pipe_kernel<G, NL, PROF>, dim3(ncl * PNWGC), dim3(NT), params, (unsigned)lds, stream);
}

// segment rows per group of the G-deep pipeline (15 at G = 3: three tiles + the transient tile must fit 160 KiB)
int pipe_rows(int G) { return G >= 3 ? 15 : 16; }

// clusters of 64 workgroups on an n_cus-CU device
int pipe_clusters(int n_cus)
{
    int ncl = n_cus / PNWGC;
    if (ncl > MAXCL) ncl = MAXCL;
    while (ncl > 1 && (8 % ncl) != 0) --ncl;
    return ncl;
}

// args.prof != nullptr selects the phase-clock build (NL = 8 only)
hipError_t launch_pipe(const LoopArgs &args, int G, int ncl, int nl, hipStream_t stream)
{
    if (ncl < 1) return hipErrorInvalidValue;
    if (nl != 16) nl = 8;
    if (args.prof) {
        if (G == 1) return launch_pipe_t<1, 8, true>(args, ncl, stream);
        if (G == 2) return launch_pipe_t<2, 8, true>(args, ncl, stream);
        if (G == 3) return launch_pipe_t<3, 8, true>(args, ncl, stream);
        return hipErrorInvalidValue;
    }
    if (G == 1) return nl == 16 ? launch_pipe_t<1, 16, false>(args, ncl, stream) : launch_pipe_t<1, 8, false>(args, ncl, stream);
    if (G == 2) return nl == 16 ? launch_pipe_t<2, 16, false>(args, ncl, stream) : launch_pipe_t<2, 8, false>(args, ncl, stream);
    if (G == 3) return nl == 16 ? launch_pipe_t<3, 16, false>(args, ncl, stream) : launch_pipe_t<3, 8, false>(args, ncl, stream);
    return hipErrorInvalidValue;
}

#undef PH

}  // namespace wrnn
