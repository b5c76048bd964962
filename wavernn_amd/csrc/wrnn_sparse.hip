// wrnn_sparse.hip -- pipelined clustered persistent WaveRNN loop kernel for BLOCK-SPARSE GRU weights (MOL), MI355X.
//
// BASELINE config 5: the four GRU matrices pruned per gate to ~5 % density in 16x1 blocks (wavernn_amd/prune.py; the
// reference's "Pruning - Scratchpad" rule applied to block magnitudes).  Same loop, exchange protocol and stage pipeline as
// wrnn_pipe.hip (reference models/fatchord_version.py:201-241), but:
//   * a workgroup owns 16 consecutive hidden units, so the 16 rows of one MFMA tile ARE one block row of a gate: the tile
//     multiplies only the block row's surviving columns (~26 of 512, padded to NBP).  A fragments = the packed block
//     values, B fragments = GATHERED activations act[segment][col[k]] (ds_read_b32 with per-lane column indices).  The GRU
//     MFMA work drops from 128 to NBP/4 = 12-16 MFMAs per block row; fc1/fc2 stay dense (one 16-row tile each).
//   * the whole weight set of a workgroup is ~160 registers, so a full copy of the model fits 32 workgroups: the chip runs
//     EIGHT independent clusters, one per XCD (block b -> XCD b % 8), each with G = 2 groups of 16 segments in flight --
//     256 segments per round -- and every all-gather stays inside one XCD's L2.
//   * fc3: workgroup r < 30 of the cluster computes logit row r (VALU dot over the gathered y2), 5th tiny all-gather.
// Skipping exact zeros does not change any partial sum; the summation ORDER differs from the dense kernels (surviving
// columns in ascending order, split in four contiguous runs over the waves), so parity is to the MoL tolerance.
#include "wrnn_tiles.h"

namespace wrnn {

constexpr int SU = 16;                // hidden units per workgroup = one 16x1 block row per gate
constexpr int SNWGC = H / SU;         // workgroups per cluster (32)
constexpr int SGR = 3 * SU;           // GRU gate rows per workgroup (48 = 3 block rows)
constexpr int SNSLOT = 6;             // partial-tile slots per wave: 0-2 input-to-hidden gates, 3-5 hidden-to-hidden gates

template <int G>
struct SparseCfg {
    static constexpr int TILE = SEG * LDC;
    static constexpr int GRP = 2 * SGR * SEG + 2 * SU * SEG + 3 * SEG;   // GH1 GH2 HOWN1 HOWN2 XS POS LIM
    static constexpr int OFF_HS = 0;
    static constexpr int OFF_ACT = OFF_HS + TILE;
    static constexpr int OFF_PART = OFF_ACT + G * TILE;                  // [NW][SNSLOT][16][16]
    static constexpr int OFF_GRP = OFF_PART + NW * SNSLOT * 256;
    static constexpr int OFF_LOG = OFF_GRP + G * GRP;
    static constexpr int OFF_WI0 = OFF_LOG + SEG * 32;
    static constexpr int OFF_W3R = OFF_WI0 + H;
    static constexpr int OFF_SCR = OFF_W3R + H;
    static constexpr int OFF_BI1 = OFF_SCR + 16 * SEG;
    static constexpr int OFF_BH1 = OFF_BI1 + SGR;
    static constexpr int OFF_BH2 = OFF_BH1 + SGR;
    static constexpr int OFF_GEO = OFF_BH2 + SGR;
    static constexpr int LDS_FLOATS = ((OFF_GEO + 2 * SPG + 3) / 4) * 4;
    static_assert(G >= 1 && G <= SPG, "G in 1..SPG");
    static_assert(LDS_FLOATS * 4 <= 160 * 1024, "LDS budget");
    static_assert(OFF_PART % 4 == 0 && OFF_GRP % 4 == 0 && OFF_WI0 % 4 == 0 && GRP % 4 == 0, "alignment");
};

// one block row: MPW dependent MFMAs of this wave's run of surviving blocks; B = act[segment fi][col]
template <int MPW>
__device__ __forceinline__ void sp_tiles3(const float (&A)[3][MPW], const int (&C)[3][MPW], const float *act_row,
                                          f32x4 &o0, f32x4 &o1, f32x4 &o2)
{
    float b[3][MPW];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int i = 0; i < MPW; ++i) b[g][i] = act_row[C[g][i]];
    f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f}, c2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < MPW; ++i) {                                    // three independent chains interleaved
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[0][i], b[0][i], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[1][i], b[1][i], c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[2][i], b[2][i], c2, 0, 0, 0);
    }
    o0 = c0; o1 = c1; o2 = c2;
}

__device__ __forceinline__ int sp_frame(const float *GP, int j, int t, int hop, int NF)
{
    const int *SP = reinterpret_cast<const int *>(GP + 2 * SGR * SEG + 2 * SU * SEG + SEG);
    const int p = SP[j] + t;
    return (p < SP[SEG + j]) ? (p / hop) : NF;
}

// G: groups in flight per cluster.  NBP: padded surviving blocks per block row (48 or 64).
template <int G, int NBP>
__global__ __launch_bounds__(NT, 1) void wrnn_sparse_kernel(const LoopArgs a)
{
    using K = SparseCfg<G>;
    constexpr int MPW = NBP / 16;                      // MFMAs per wave per block row
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *HS = smem + K::OFF_HS, *PART = smem + K::OFF_PART, *LOG = smem + K::OFF_LOG, *WI0 = smem + K::OFF_WI0;
    float *W3R = smem + K::OFF_W3R, *SCR = smem + K::OFF_SCR;
    float *BI1 = smem + K::OFF_BI1, *BH1 = smem + K::OFF_BH1, *BH2 = smem + K::OFF_BH2;
    int *GEO = reinterpret_cast<int *>(smem + K::OFF_GEO);
    float touch = 0.f;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int ncl = gridDim.x / SNWGC;
    int cl, wg;
    if (ncl == 8 && gridDim.x % 8 == 0) { cl = blockIdx.x % 8; wg = blockIdx.x / 8; }      // one cluster per XCD
    else { cl = blockIdx.x / SNWGC; wg = blockIdx.x % SNWGC; }
    const int fi = lane & 15, kq = lane >> 4;
    const int kbase_lane = KCH * w + 4 * kq;
    const int Btot = a.Btot, T = a.T, C = a.C, NG = a.NG;
    const int er = tid >> 4, ec = tid & 15;
    const int pu = tid >> 4, pj = tid & 15;             // pointwise role: (owned unit pu, segment pj), all 256 threads
    const int prow = SU * wg + pu;
    const bool fc3_wg = wg < 30;                        // this workgroup owns logit row `wg`

    // ---- one-time: packed block rows -> registers.  Matrix m, gate g, this wave's MFMA i covers blocks 4*(MPW*w+i)+kq.
    float A_ih1[3][MPW], A_hh1[3][MPW], A_ih2[3][MPW], A_hh2[3][MPW], A_fc1[AF], A_fc2[AF];
    int C_ih1[3][MPW], C_hh1[3][MPW], C_ih2[3][MPW], C_hh2[3][MPW];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int i = 0; i < MPW; ++i) {
            const int blk = 4 * (MPW * w + i) + kq;
            const size_t r0 = ((size_t)(0 * SNWGC + wg) * 3 + g) * NBP + blk, r1 = ((size_t)(1 * SNWGC + wg) * 3 + g) * NBP + blk;
            const size_t r2 = ((size_t)(2 * SNWGC + wg) * 3 + g) * NBP + blk, r3 = ((size_t)(3 * SNWGC + wg) * 3 + g) * NBP + blk;
            A_ih1[g][i] = a.sp_vals[r0 * 16 + fi]; C_ih1[g][i] = a.sp_cols[r0];
            A_hh1[g][i] = a.sp_vals[r1 * 16 + fi]; C_hh1[g][i] = a.sp_cols[r1];
            A_ih2[g][i] = a.sp_vals[r2 * 16 + fi]; C_ih2[g][i] = a.sp_cols[r2];
            A_hh2[g][i] = a.sp_vals[r3 * 16 + fi]; C_hh2[g][i] = a.sp_cols[r3];
        }
    load_afrag(A_fc1, a.fc1_w, H + AUX, SU * wg + fi, true, kbase_lane);
    load_afrag(A_fc2, a.fc2_w, H + AUX, SU * wg + fi, true, kbase_lane);
    for (int q = tid; q < K::LDS_FLOATS; q += NT) smem[q] = 0.f;
    __syncthreads();
    WI0[2 * tid] = a.I_w0[2 * tid];
    WI0[2 * tid + 1] = a.I_w0[2 * tid + 1];
    if (fc3_wg) { W3R[2 * tid] = a.fc3_w[(size_t)wg * H + 2 * tid]; W3R[2 * tid + 1] = a.fc3_w[(size_t)wg * H + 2 * tid + 1]; }
    const float b3 = fc3_wg ? a.fc3_b[wg] : 0.f;
    if (tid < SGR) {
        const int grow = (tid / SU) * H + SU * wg + (tid % SU);
        BI1[tid] = a.b_ih1[grow];
        BH1[tid] = a.b_hh1[grow];
        BH2[tid] = a.b_hh2[grow];
    }

    const __amdgpu_buffer_rsrc_t grs = make_rsrc(a.gran, GRAN_WORDS * 8);
    constexpr int LAYER_BYTES = SEG * H * 8;
    constexpr int SLOT_BYTES = NGRAN * LAYER_BYTES;
    const int soff_cl = cl * SPG * SLOT_BYTES;

    unsigned tagbase = 0u;
    for (int round = 0;; ++round, tagbase += (unsigned)T) {
        const int gfirst = cl + ncl * (round * G);
        if (gfirst >= NG) break;
        int nact = 0;
#pragma unroll
        for (int i = 0; i < G; ++i)
            if (gfirst + ncl * i < NG) nact = i + 1;
        __syncthreads();
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
        __syncthreads();
        // ---- state init (fatchord_version.py:194-196): h = 0, x = 0, gh = b_hh ------------------------------------
#pragma unroll 1
        for (int i = 0; i < nact; ++i) {
            float *GP = smem + K::OFF_GRP + i * K::GRP;
            float *ACT = smem + K::OFF_ACT + i * K::TILE;
            const int b0 = GEO[2 * i], nb = GEO[2 * i + 1];
            for (int q = tid; q < SGR * SEG; q += NT) { GP[q] = BH1[q >> 4]; GP[SGR * SEG + q] = BH2[q >> 4]; }
            for (int q = tid; q < 2 * SU * SEG; q += NT) GP[2 * SGR * SEG + q] = 0.f;           // HOWN1 + HOWN2
            if (tid < SEG) {
                GP[2 * SGR * SEG + 2 * SU * SEG + tid] = 0.f;                                    // XS
                int *SP = reinterpret_cast<int *>(GP + 2 * SGR * SEG + 2 * SU * SEG + SEG);
                const int sc = b0 + (tid < nb ? tid : nb - 1);
                SP[tid] = a.seg_pos[sc];
                SP[SEG + tid] = a.seg_lim[sc];
            }
            {
                const int erc = er < nb ? er : nb - 1;
                const float *crow = a.cI + ((size_t)0 * Btot + b0 + erc) * H + 2 * ec;
#pragma unroll
                for (int c = 0; c < 16; ++c)
                    *reinterpret_cast<float2 *>(ACT + er * LDC + own_col(c, ec)) = *reinterpret_cast<const float2 *>(crow + 32 * c);
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
                float *GH1 = GP, *HOWN1 = GP + 2 * SGR * SEG;
                u64 *G1 = a.gran + (size_t)(cl * SPG + i) * NGRAN * SEG * H + 0 * SEG * H;
                __syncthreads();
                {
                    f32x4 o0, o1, o2;
                    sp_tiles3<MPW>(A_ih1, C_ih1, ACT + fi * LDC, o0, o1, o2);
                    put_partial_rm<SNSLOT>(PART, w, 0, lane, o0);
                    put_partial_rm<SNSLOT>(PART, w, 1, lane, o1);
                    put_partial_rm<SNSLOT>(PART, w, 2, lane, o2);
                }
                __syncthreads();
                if (pj < nb) {
                    const float gir = get_partial_rm<SNSLOT>(PART, 0, 0 * SU + pu, pj) + BI1[0 * SU + pu];
                    const float giz = get_partial_rm<SNSLOT>(PART, 0, 1 * SU + pu, pj) + BI1[1 * SU + pu];
                    const float gin = get_partial_rm<SNSLOT>(PART, 0, 2 * SU + pu, pj) + BI1[2 * SU + pu];
                    const float hn = gru_update(gir, giz, gin, GH1[(0 * SU + pu) * SEG + pj], GH1[(1 * SU + pu) * SEG + pj],
                                                GH1[(2 * SU + pu) * SEG + pj], HOWN1[pu * SEG + pj]);
                    HOWN1[pu * SEG + pj] = hn;
                    publish(G1, tag, pj, prow, hn);
                }
            }

            // =========================== S2: GRU2 (:212-214) ==========================================
#pragma unroll 1
            for (int i = 0; i < nact; ++i) {
                const int nb = GEO[2 * i + 1];
                float *GP = smem + K::OFF_GRP + i * K::GRP;
                float *ACT = smem + K::OFF_ACT + i * K::TILE;
                float *GH1 = GP, *GH2 = GP + SGR * SEG, *HOWN2 = GP + 2 * SGR * SEG + SU * SEG;
                u64 *G2 = a.gran + (size_t)(cl * SPG + i) * NGRAN * SEG * H + 1 * SEG * H;
                const int f2 = sp_frame(GP, pj, t, a.hop, a.NF);
                const float c2r = a.c2f[(size_t)f2 * 3 * H + prow];
                const float c2z = a.c2f[(size_t)f2 * 3 * H + H + prow];
                const float c2n = a.c2f[(size_t)f2 * 3 * H + 2 * H + prow];
                bool ok = sweep_layer<true, 16>(grs, soff_cl + i * SLOT_BYTES + 0 * LAYER_BYTES, tag, nb, tid, HS, ACT, a.status);   // h1 -> HS; ACT = xi + h1
                if (!ok) report_failure(a.status, 0x400u | 1u, blockIdx.x, t, tid);
                if (__syncthreads_or(!ok)) return;
                {
                    f32x4 o0, o1, o2;
                    sp_tiles3<MPW>(A_ih2, C_ih2, ACT + fi * LDC, o0, o1, o2);
                    put_partial_rm<SNSLOT>(PART, w, 0, lane, o0);
                    put_partial_rm<SNSLOT>(PART, w, 1, lane, o1);
                    put_partial_rm<SNSLOT>(PART, w, 2, lane, o2);
                    sp_tiles3<MPW>(A_hh1, C_hh1, HS + fi * LDC, o0, o1, o2);           // gh1(t+1) = W_hh1 . h1(t)
                    put_partial_rm<SNSLOT>(PART, w, 3, lane, o0);
                    put_partial_rm<SNSLOT>(PART, w, 4, lane, o1);
                    put_partial_rm<SNSLOT>(PART, w, 5, lane, o2);
                }
                __syncthreads();
                if (pj < nb) {
                    const float gir = get_partial_rm<SNSLOT>(PART, 0, 0 * SU + pu, pj) + c2r;
                    const float giz = get_partial_rm<SNSLOT>(PART, 0, 1 * SU + pu, pj) + c2z;
                    const float gin = get_partial_rm<SNSLOT>(PART, 0, 2 * SU + pu, pj) + c2n;
                    const float hn = gru_update(gir, giz, gin, GH2[(0 * SU + pu) * SEG + pj], GH2[(1 * SU + pu) * SEG + pj],
                                                GH2[(2 * SU + pu) * SEG + pj], HOWN2[pu * SEG + pj]);
                    HOWN2[pu * SEG + pj] = hn;
                    publish(G2, tag, pj, prow, hn);
                }
#pragma unroll
                for (int q0 = 0; q0 < (SGR * SEG) / NT; ++q0) {
                    const int q = tid + NT * q0;
                    GH1[q] = get_partial_rm<SNSLOT>(PART, 3, q >> 4, q & 15) + BH1[q >> 4];
                }
            }

            // =========================== S3: fc1 + relu (:216-218) ====================================
#pragma unroll 1
            for (int i = 0; i < nact; ++i) {
                const int nb = GEO[2 * i + 1];
                float *GP = smem + K::OFF_GRP + i * K::GRP;
                float *ACT = smem + K::OFF_ACT + i * K::TILE;
                float *GH2 = GP + SGR * SEG;
                u64 *G3 = a.gran + (size_t)(cl * SPG + i) * NGRAN * SEG * H + 2 * SEG * H;
                const float c3v = a.c3f[(size_t)sp_frame(GP, pj, t, a.hop, a.NF) * H + prow];
                bool ok = sweep_layer<true, 16>(grs, soff_cl + i * SLOT_BYTES + 1 * LAYER_BYTES, tag, nb, tid, HS, ACT, a.status);   // h2 -> HS; ACT = x1 + h2
                if (!ok) report_failure(a.status, 0x400u | 2u, blockIdx.x, t, tid);
                if (__syncthreads_or(!ok)) return;
                put_partial_rm<SNSLOT>(PART, w, 0, lane, mfma_tile_pre(A_fc1, ACT + fi * LDC + kbase_lane));
                {
                    f32x4 o0, o1, o2;
                    sp_tiles3<MPW>(A_hh2, C_hh2, HS + fi * LDC, o0, o1, o2);           // gh2(t+1) = W_hh2 . h2(t)
                    put_partial_rm<SNSLOT>(PART, w, 3, lane, o0);
                    put_partial_rm<SNSLOT>(PART, w, 4, lane, o1);
                    put_partial_rm<SNSLOT>(PART, w, 5, lane, o2);
                }
                __syncthreads();
                if (pj < nb) publish(G3, tag, pj, prow, fmaxf(get_partial_rm<SNSLOT>(PART, 0, pu, pj) + c3v, 0.f));
#pragma unroll
                for (int q0 = 0; q0 < (SGR * SEG) / NT; ++q0) {
                    const int q = tid + NT * q0;
                    GH2[q] = get_partial_rm<SNSLOT>(PART, 3, q >> 4, q & 15) + BH2[q >> 4];
                }
            }

            // =========================== S4: fc2 + relu (:220-221) ====================================
#pragma unroll 1
            for (int i = 0; i < nact; ++i) {
                const int nb = GEO[2 * i + 1];
                float *GP = smem + K::OFF_GRP + i * K::GRP;
                float *ACT = smem + K::OFF_ACT + i * K::TILE;
                u64 *G4 = a.gran + (size_t)(cl * SPG + i) * NGRAN * SEG * H + 3 * SEG * H;
                const float c4v = a.c4f[(size_t)sp_frame(GP, pj, t, a.hop, a.NF) * H + prow];
                bool ok = sweep_layer<false, 16>(grs, soff_cl + i * SLOT_BYTES + 2 * LAYER_BYTES, tag, nb, tid, ACT, nullptr, a.status);   // ACT <- y1
                if (!ok) report_failure(a.status, 0x400u | 3u, blockIdx.x, t, tid);
                if (__syncthreads_or(!ok)) return;
                put_partial_rm<SNSLOT>(PART, w, 0, lane, mfma_tile_pre(A_fc2, ACT + fi * LDC + kbase_lane));
                __syncthreads();
                if (pj < nb) publish(G4, tag, pj, prow, fmaxf(get_partial_rm<SNSLOT>(PART, 0, pu, pj) + c4v, 0.f));
            }

            // =========================== S5: fc3, one logit row per workgroup (:223) ==================
            if (fc3_wg) {
#pragma unroll 1
                for (int i = 0; i < nact; ++i) {
                    const int nb = GEO[2 * i + 1];
                    float *ACT = smem + K::OFF_ACT + i * K::TILE;
                    u64 *G5 = a.gran + (size_t)(cl * SPG + i) * NGRAN * SEG * H + 4 * SEG * H;
                    bool ok = sweep_layer<false, 16>(grs, soff_cl + i * SLOT_BYTES + 3 * LAYER_BYTES, tag, nb, tid, ACT, nullptr, a.status);   // ACT <- y2
                    if (!ok) report_failure(a.status, 0x400u | 4u, blockIdx.x, t, tid);
                    if (__syncthreads_or(!ok)) return;
                    {   // thread (segment pj, k-chunk pu): 32 terms of logit[wg][pj]
                        const float *xr = ACT + pj * LDC + 32 * pu;
                        const float *wr = W3R + 32 * pu;
                        float s = 0.f;
#pragma unroll
                        for (int k = 0; k < 32; k += 4) {
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
                        publish(G5, tag, tid, wg, s + b3);
                    }
                }
            }

            // =========================== S6: sampling (utils/distribution.py:102-121) + xi(t+1) ========
#pragma unroll 1
            for (int i = 0; i < nact; ++i) {
                float *GP = smem + K::OFF_GRP + i * K::GRP;
                float *ACT = smem + K::OFF_ACT + i * K::TILE;
                float *XS = GP + 2 * SGR * SEG + 2 * SU * SEG;
                const int nb = GEO[2 * i + 1], b0 = GEO[2 * i];
                const float *nrow = a.noise_pre + (size_t)t * 11 * Btot;       // log(-log u1) / log u2 - log(1-u2)
                const int puc = pu < nb ? pu : nb - 1;
                const float nz0 = nrow[(b0 + puc) * 10 + (pj < 10 ? pj : 9)];
                const float nz1 = nrow[10 * Btot + b0 + puc];
                {   // gather the 30 logits of every segment: thread (segment er, c = ec < 15) reads logits 2c, 2c+1
                    bool ok = true;
                    if (er < nb && ec < 15) {
                        const int voff = er * (H * 8) + ec * 16;
                        const int soff = soff_cl + i * SLOT_BYTES + 4 * LAYER_BYTES;
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
                        const float l0 = __uint_as_float(x.x), l1 = __uint_as_float(x.z);
                        LOG[er * 32 + 2 * ec] = l0;
                        LOG[er * 32 + 2 * ec + 1] = l1;
                        if (a.dbg_logits && wg == 0 && ok) {
                            a.dbg_logits[((size_t)t * Btot + b0 + er) * C + 2 * ec] = l0;
                            a.dbg_logits[((size_t)t * Btot + b0 + er) * C + 2 * ec + 1] = l1;
                        }
                    }
                    if (!ok) report_failure(a.status, 0x400u | 5u, blockIdx.x, t, tid);
                    if (__syncthreads_or(!ok)) return;
                }
                float2 cn[16];
                {   // cI(t+1) after the last poll of the group's step (vector loads return in order); rows touched a step ago
                    const int tn = (t + 1 < T) ? t + 1 : t;
                    const int erc = er < nb ? er : nb - 1;
                    const float *crow = a.cI + ((size_t)tn * Btot + b0 + erc) * H + 2 * ec;
#pragma unroll
                    for (int c = 0; c < 16; ++c) cn[c] = *reinterpret_cast<const float2 *>(crow + 32 * c);
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
                __syncthreads();
                {   // xi(t+1) = W_I[:,0] * x_t + cI(t+1)  (:208-209)
                    const float xs = XS[er];
#pragma unroll
                    for (int c = 0; c < 16; ++c) {
                        const float2 wv = *reinterpret_cast<const float2 *>(WI0 + own_col(c, ec));
                        *reinterpret_cast<float2 *>(ACT + er * LDC + own_col(c, ec)) = make_float2(fmaf(wv.x, xs, cn[c].x), fmaf(wv.y, xs, cn[c].y));
                    }
                }
            }
        }
    }
    asm volatile("" ::"v"(touch));
}

template <int G, int NBP>
static hipError_t launch_sparse_t(const LoopArgs &args, int ncl, hipStream_t stream)
{
    using K = SparseCfg<G>;
    const size_t lds = (size_t)K::LDS_FLOATS * sizeof(float);
    hipError_t e = hipFuncSetAttribute((const void *)wrnn_sparse_kernel<G, NBP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    LoopArgs a = args;
    void *params[] = {(void *)&a};
    return hipLaunchCooperativeKernel((const void *)wrnn_sparse_kernel<G, NBP>, dim3(ncl * SNWGC), dim3(NT), params, (unsigned)lds, stream);
}

// clusters of 32 workgroups on an n_cus-CU device (8 = one per XCD on MI355X)
int sparse_clusters(int n_cus)
{
    int ncl = n_cus / SNWGC;
    if (ncl > SPCL) ncl = SPCL;
    return ncl;
}

hipError_t launch_sparse(const LoopArgs &args, int G, int ncl, int nbp, hipStream_t stream)
{
    if (ncl < 1 || (nbp != 48 && nbp != 64)) return hipErrorInvalidValue;
    if (G == 1) return nbp == 48 ? launch_sparse_t<1, 48>(args, ncl, stream) : launch_sparse_t<1, 64>(args, ncl, stream);
    if (G == 2) return nbp == 48 ? launch_sparse_t<2, 48>(args, ncl, stream) : launch_sparse_t<2, 64>(args, ncl, stream);
    return hipErrorInvalidValue;
}

}  // namespace wrnn
