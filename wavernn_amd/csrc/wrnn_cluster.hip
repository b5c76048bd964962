// wrnn_cluster.hip -- the clustered persistent WaveRNN loop kernel for MI355X (gfx950 / CDNA4).
//
// Replaces the `for i in range(seq_len)` loop of fatchord/WaveRNN `WaveRNN.generate()`
// (reference models/fatchord_version.py:201-241) for ANY number of folded segments in ONE launch.
//
// Design (DESIGN.md section "K-loop"):
//   * the chip is split into NCL = U/2 independent CLUSTERS of NWGC = 512/U workgroups (one workgroup per CU):
//     U = 8 -> 4 clusters of 64 CUs (2 XCDs each), U = 4 -> 2 clusters of 128 CUs, U = 2 -> the whole chip.
//     Every cluster holds ONE complete fp32 copy of the loop weights on-chip: workgroup g of a cluster owns
//     hidden units [U*g, U*g+U) of both GRUs and the same rows of fc1/fc2 (RAW: and of fc3) as MFMA A-fragments
//     that stay in VGPR/AGPRs for the whole kernel (10 tiles x 32 registers at U = 8); the 30-row MOL fc3 is
//     replicated per workgroup in LDS.
//   * the B segments are cut into NG = ceil(B/16) groups of <= 16 segments (the MFMA N dimension).  Cluster c
//     runs groups c, c+NCL, ... one after the other, each for all T steps; clusters never talk to each other.
//   * per step a cluster does 4 (RAW: 5) all-gathers of a 512-wide activation vector per segment, as 8-byte
//     {step tag, f32} granules: one relaxed agent-scope (sc1) store per value, sc1 polling sweeps, no fences,
//     no flags -- the data is its own flag (MI355X guide, Guideline 16 form R2).  Smaller clusters mean fewer
//     readers per granule and a full 16-row MFMA tile per workgroup (U = 8: 24 of 32 GRU tile rows are live,
//     against 6 of 16 in the chip-wide split).
//   * v_mfma_f32_16x16x4_f32 (exact f32, an fmaf chain): rows = gate rows of the owned units, cols = the 16
//     segments of the group.  4 waves (one per SIMD, 512-register budget each) split K = 512; partial tiles are
//     summed through LDS in a fixed order.  Activations live in LDS as [segment][k] rows of stride 520 floats
//     (conflict-free ds_read_b128 for the (row, k-quad) lane mapping of the MFMA operands).
//   * the sampling tail (fc3 + mixture-of-logistics / softmax sampling) is computed redundantly by every
//     workgroup of the cluster from identical inputs, so x_t needs no further exchange; wavefront shuffles do
//     the argmax / softmax reductions.
//   * conditioning products that do not depend on the recurrence were hoisted (wrnn_cond.hip); GRU
//     hidden-to-hidden products (W_hh.h) depend only on the previous step and run while granules are in flight.
#include "wrnn_tiles.h"

namespace wrnn {

template <int U>
struct ClusterCfg {
    static constexpr int NWGC = H / U;                 // workgroups per cluster
    static constexpr int NCL = U / 2;                  // clusters per 256-CU chip
    static constexpr int GR = 3 * U;                   // GRU gate rows per workgroup
    static constexpr int RT = (GR + 15) / 16;          // 16-row MFMA tiles per GRU matrix
    static constexpr int NSLOT = 2 * RT;               // partial-tile slots per wave: [0,RT) critical, [RT,2RT) off-path
    static constexpr int GHI = (GR * SEG + NT - 1) / NT;   // gh-reduce items per thread
    // LDS carve (floats)
    static constexpr int OFF_ACT = 0;
    static constexpr int OFF_HS = OFF_ACT + SEG * LDC;
    static constexpr int OFF_W3 = OFF_HS + SEG * LDC;              // [32][LDC] (MOL: 30 rows used; RAW: U rows used)
    static constexpr int OFF_PART = OFF_W3 + 32 * LDC;             // [NW][NSLOT][16][16]
    static constexpr int OFF_GH1 = OFF_PART + NW * NSLOT * 256;    // [GR][SEG]
    static constexpr int OFF_GH2 = OFF_GH1 + GR * SEG;
    static constexpr int OFF_HOWN1 = OFF_GH2 + GR * SEG;           // [U][SEG]
    static constexpr int OFF_HOWN2 = OFF_HOWN1 + U * SEG;
    static constexpr int OFF_LOG = OFF_HOWN2 + U * SEG;            // [SEG][32]
    static constexpr int OFF_XS = OFF_LOG + SEG * 32;              // [SEG]
    static constexpr int OFF_WI0 = OFF_XS + SEG;                   // [H]  I.weight[:,0]
    static constexpr int OFF_BI1 = OFF_WI0 + H;                    // [GR] b_ih1 of the owned gate rows
    static constexpr int OFF_BH1 = OFF_BI1 + GR;                   // [GR] b_hh1
    static constexpr int OFF_BH2 = OFF_BH1 + GR;                   // [GR] b_hh2
    static constexpr int OFF_B3 = OFF_BH2 + GR;                    // [32] fc3 bias (MOL: rows 0..29; RAW: owned rows)
    static constexpr int LDS_FLOATS = OFF_B3 + 32;
    static_assert(U == 2 || U == 4 || U == 8, "U in {2,4,8}");
    static_assert(NSLOT >= 2, "MOL fc3 needs two partial slots");
    static_assert(LDS_FLOATS * 4 <= 160 * 1024, "LDS budget");
    static_assert(OFF_HS % 4 == 0 && OFF_W3 % 4 == 0 && OFF_PART % 4 == 0 && OFF_WI0 % 2 == 0, "alignment");
};

// U: hidden units per workgroup (2, 4 or 8).  MODE: 0 RAW (C == 512), 1 MOL (C == 30).  NL: sweep loads in flight.
template <int U, int MODE, int NL>
__global__ __launch_bounds__(NT, 1) void wrnn_cluster_kernel(const LoopArgs a)
{
    using K = ClusterCfg<U>;
    constexpr int RT = K::RT, NSLOT = K::NSLOT, GR = K::GR;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *ACT = smem + K::OFF_ACT, *HS = smem + K::OFF_HS, *W3 = smem + K::OFF_W3, *PART = smem + K::OFF_PART;
    float *GH1 = smem + K::OFF_GH1, *GH2 = smem + K::OFF_GH2, *HOWN1 = smem + K::OFF_HOWN1, *HOWN2 = smem + K::OFF_HOWN2;
    float *LOG = smem + K::OFF_LOG, *XS = smem + K::OFF_XS, *WI0 = smem + K::OFF_WI0;
    float *BI1 = smem + K::OFF_BI1, *BH1 = smem + K::OFF_BH1, *BH2 = smem + K::OFF_BH2, *B3 = smem + K::OFF_B3;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // ---- cluster / workgroup-in-cluster.  Dispatch order puts block b on XCD b % 8 (observed, used for speed
    //      only): a cluster is built from whole XCDs so its granule traffic stays inside 8/NCL L2s.
    int cl, wg;
    {
        const int b = blockIdx.x, nblk = gridDim.x;
        const int ncl = nblk / K::NWGC;                    // clusters in this launch (1 .. NCL)
        if (nblk % 8 == 0 && ncl >= 1 && 8 % ncl == 0) {
            const int xpc = 8 / ncl, per_xcd = nblk / 8;   // XCDs per cluster, blocks per XCD
            const int xcd = b % 8;
            cl = xcd / xpc;
            wg = (xcd % xpc) * per_xcd + b / 8;
        } else {
            cl = b / K::NWGC;
            wg = b % K::NWGC;
        }
    }
    const int ncl = gridDim.x / K::NWGC;
    const int fi = lane & 15, kq = lane >> 4;          // MFMA fragment row / k-quad of this lane
    const int kbase_lane = KCH * w + 4 * kq;
    const int Btot = a.Btot, T = a.T, C = a.C, NG = a.NG;
    // elementwise / sweep role: row er (segment) and the 16 column pairs own_col(i, ec), i = 0..15
    const int er = tid >> 4, ec = tid & 15;

    // ---- one-time: gather the weight slice into register-resident MFMA A fragments --------------------
    // GRU fragment row ri = gate*U + u (gate 0..2 = r,z,n; u = owned unit), tile ri/16; fc tiles: row = u.
    float A_ih1[RT][AF], A_hh1[RT][AF], A_ih2[RT][AF], A_hh2[RT][AF], A_fc1[AF], A_fc2[AF];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int ri = 16 * rt + fi;
        const bool vg = ri < GR;
        const int grow = (ri / U) * H + U * wg + (ri % U);
        load_afrag(A_ih1[rt], a.w_ih1, H, grow, vg, kbase_lane);
        load_afrag(A_hh1[rt], a.w_hh1, H, grow, vg, kbase_lane);
        load_afrag(A_ih2[rt], a.w_ih2, H + AUX, grow, vg, kbase_lane);
        load_afrag(A_hh2[rt], a.w_hh2, H, grow, vg, kbase_lane);
    }
    {
        const bool vf = fi < U;
        const int frow = U * wg + fi;
        load_afrag(A_fc1, a.fc1_w, H + AUX, frow, vf, kbase_lane);
        load_afrag(A_fc2, a.fc2_w, H + AUX, frow, vf, kbase_lane);
    }
    // fc3 -> LDS tile rows: MOL rows 0..29 (replicated), RAW rows U*wg .. U*wg+U-1 (distributed); rest zero
    for (int q = tid; q < 32 * (H / 4); q += NT) {
        const int row = q / (H / 4), k4 = (q % (H / 4)) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (MODE == 1) { if (row < 30) v = *reinterpret_cast<const float4 *>(a.fc3_w + (size_t)row * H + k4); }
        else { if (row < U) v = *reinterpret_cast<const float4 *>(a.fc3_w + (size_t)(U * wg + row) * H + k4); }
        *reinterpret_cast<float4 *>(W3 + row * LDC + k4) = v;
    }
    WI0[2 * tid] = a.I_w0[2 * tid];                                 // I.weight[:,0]: tap of x_{t-1}
    WI0[2 * tid + 1] = a.I_w0[2 * tid + 1];

    // per-thread constants
    const int pu = tid >> 4, pj = tid & 15;             // pointwise role: (owned unit, segment) for tid < 16U
    const bool pw_thread = tid < 16 * U;
    const int prow = U * wg + (pu % U);                 // hidden index this pointwise thread owns
    // biases of the owned rows live in LDS (read once per step; keeps ~10 registers free)
    if (tid < GR) {
        const int grow = (tid / U) * H + U * wg + (tid % U);
        BI1[tid] = a.b_ih1[grow];
        BH1[tid] = a.b_hh1[grow];
        BH2[tid] = a.b_hh2[grow];
    }
    if (tid < 32) {
        float v = 0.f;
        if (MODE == 1) { if (tid < 30) v = a.fc3_b[tid]; }
        else { if (tid < U) v = a.fc3_b[U * wg + tid]; }
        B3[tid] = v;
    }

    u64 *Gc = a.gran + (size_t)cl * NGRAN * SEG * H;    // this cluster's granule layers
    u64 *G1 = Gc + 0 * SEG * H, *G2 = Gc + 1 * SEG * H, *G3 = Gc + 2 * SEG * H, *G4 = Gc + 3 * SEG * H, *G5 = Gc + 4 * SEG * H;
    const __amdgpu_buffer_rsrc_t grs = make_rsrc(a.gran, MAXCL * NGRAN * SEG * H * 8);
    const int soff0 = cl * NGRAN * SEG * H * 8;
    constexpr int LAYER_BYTES = SEG * H * 8;

    unsigned tagbase = 0u;
    for (int g = cl; g < NG; g += ncl, tagbase += (unsigned)T) {
        const int b0 = (int)(((long)g * Btot) / NG);
        const int nb = (int)(((long)(g + 1) * Btot) / NG) - b0;
        const int erc = er < nb ? er : nb - 1;          // clamped row for conditioning reads of unused rows
        const bool is_pw = pw_thread && (pj < nb);

        // ---- state init (fatchord_version.py:194-196: h1 = h2 = 0, x = 0) ------------------------------
        __syncthreads();                                                 // previous group's LDS reads are done
        for (int q = tid; q < SEG * LDC; q += NT) HS[q] = 0.f;
        if (tid < 2 * U * SEG) HOWN1[tid] = 0.f;                         // HOWN1 + HOWN2 (contiguous)
        if (tid < SEG) XS[tid] = 0.f;
        float2 cn[16];                                                   // cI of the NEXT step, row er, owned columns
        {
            const float *crow = a.cI + ((size_t)0 * Btot + b0 + erc) * H + 2 * ec;
#pragma unroll
            for (int i = 0; i < 16; ++i) cn[i] = *reinterpret_cast<const float2 *>(crow + 32 * i);
        }
        __syncthreads();
        {                                                                // xi(0) = cI(0)  (x_{-1} = 0)
            const float xs = XS[er];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float2 wv = *reinterpret_cast<const float2 *>(WI0 + own_col(i, ec));
                *reinterpret_cast<float2 *>(ACT + er * LDC + own_col(i, ec)) = make_float2(fmaf(wv.x, xs, cn[i].x), fmaf(wv.y, xs, cn[i].y));
            }
        }
        // gh1 of step 0 = W_hh1 . 0 + b_hh1 (run the generic path so every step is identical)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
            put_partial<NSLOT>(PART, w, RT + rt, lane, mfma_tile(A_hh1[rt], HS + fi * LDC + kbase_lane));
        __syncthreads();
#pragma unroll
        for (int i = 0; i < K::GHI; ++i) {
            const int q = tid + NT * i;
            if (q < GR * SEG) GH1[q] = get_partial<NSLOT>(PART, RT, q >> 4, q & 15) + BH1[q >> 4];
        }
        __syncthreads();

        for (int t = 0; t < T; ++t) {
            const unsigned tag = tagbase + (unsigned)t + 1u;
            float c2r = 0, c2z = 0, c2n = 0, c3v = 0, c4v = 0;
            if (is_pw) {
                const int f = cond_frame(a.seg_pos, a.seg_lim, b0 + pj, t, a.hop, a.NF);
                c2r = a.c2f[(size_t)f * 3 * H + prow];
                c2z = a.c2f[(size_t)f * 3 * H + H + prow];
                c2n = a.c2f[(size_t)f * 3 * H + 2 * H + prow];
                c3v = a.c3f[(size_t)f * H + prow];
                c4v = a.c4f[(size_t)f * H + prow];
            }
            float nz0 = 0.5f, nz1 = 0.5f;                     // MOL: u1 (mixture pj of segment pu), u2 (pj == 0)
            if (MODE == 1) {
                if (pu < nb) {
                    const float *nrow = a.noise + (size_t)t * 11 * Btot;
                    if (pj < 10) nz0 = nrow[(b0 + pu) * 10 + pj];
                    if (pj == 0) nz1 = nrow[10 * Btot + b0 + pu];
                }
            }

            // =========================== S1: GRU1 (fatchord_version.py:210) ===========================
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
                put_partial<NSLOT>(PART, w, rt, lane, mfma_tile(A_ih1[rt], ACT + fi * LDC + kbase_lane));
            __syncthreads();
            if (is_pw) {
                const float gir = get_partial<NSLOT>(PART, 0, 0 * U + pu, pj) + BI1[0 * U + pu];
                const float giz = get_partial<NSLOT>(PART, 0, 1 * U + pu, pj) + BI1[1 * U + pu];
                const float gin = get_partial<NSLOT>(PART, 0, 2 * U + pu, pj) + BI1[2 * U + pu];
                const float hn = gru_update(gir, giz, gin, GH1[(0 * U + pu) * SEG + pj], GH1[(1 * U + pu) * SEG + pj],
                                            GH1[(2 * U + pu) * SEG + pj], HOWN1[pu * SEG + pj]);
                HOWN1[pu * SEG + pj] = hn;
                publish(G1, tag, pj, prow, hn);
            }
            // off the critical path: gh2(t) = W_hh2 . h2(t-1) while the granules travel (HS holds h2(t-1))
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
                put_partial<NSLOT>(PART, w, RT + rt, lane, mfma_tile(A_hh2[rt], HS + fi * LDC + kbase_lane));
            __syncthreads();
#pragma unroll
            for (int i = 0; i < K::GHI; ++i) {
                const int q = tid + NT * i;
                if (q < GR * SEG) GH2[q] = get_partial<NSLOT>(PART, RT, q >> 4, q & 15) + BH2[q >> 4];
            }
            // h1(t) -> HS ; ACT = xi + h1 (:212).  Every wave passed the barrier => the hh2 reads of HS are done.
            bool ok = sweep_layer<true, NL>(grs, soff0 + 0 * LAYER_BYTES, tag, nb, tid, HS, ACT, a.status);
            if (!ok) report_failure(a.status, 0x200u | 1u, blockIdx.x, t, tid);
            if (__syncthreads_or(!ok)) return;

            // =========================== S2: GRU2 (:213-214) ==========================================
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
                put_partial<NSLOT>(PART, w, rt, lane, mfma_tile(A_ih2[rt], ACT + fi * LDC + kbase_lane));
            __syncthreads();
            if (is_pw) {
                const float gir = get_partial<NSLOT>(PART, 0, 0 * U + pu, pj) + c2r;
                const float giz = get_partial<NSLOT>(PART, 0, 1 * U + pu, pj) + c2z;
                const float gin = get_partial<NSLOT>(PART, 0, 2 * U + pu, pj) + c2n;
                const float hn = gru_update(gir, giz, gin, GH2[(0 * U + pu) * SEG + pj], GH2[(1 * U + pu) * SEG + pj],
                                            GH2[(2 * U + pu) * SEG + pj], HOWN2[pu * SEG + pj]);
                HOWN2[pu * SEG + pj] = hn;
                publish(G2, tag, pj, prow, hn);
            }
            // off the critical path: gh1(t+1) = W_hh1 . h1(t)   (HS holds h1(t))
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
                put_partial<NSLOT>(PART, w, RT + rt, lane, mfma_tile(A_hh1[rt], HS + fi * LDC + kbase_lane));
            __syncthreads();
#pragma unroll
            for (int i = 0; i < K::GHI; ++i) {
                const int q = tid + NT * i;
                if (q < GR * SEG) GH1[q] = get_partial<NSLOT>(PART, RT, q >> 4, q & 15) + BH1[q >> 4];
            }
            // h2(t) -> HS ; ACT = x1 + h2 (:216)
            ok = sweep_layer<true, NL>(grs, soff0 + 1 * LAYER_BYTES, tag, nb, tid, HS, ACT, a.status);
            if (!ok) report_failure(a.status, 0x200u | 2u, blockIdx.x, t, tid);
            if (__syncthreads_or(!ok)) return;

            // =========================== S3: fc1 + relu (:217-218) ====================================
            put_partial<NSLOT>(PART, w, 0, lane, mfma_tile(A_fc1, ACT + fi * LDC + kbase_lane));
            __syncthreads();
            if (is_pw) publish(G3, tag, pj, prow, fmaxf(get_partial<NSLOT>(PART, 0, pu, pj) + c3v, 0.f));
            ok = sweep_layer<false, NL>(grs, soff0 + 2 * LAYER_BYTES, tag, nb, tid, ACT, nullptr, a.status);   // ACT reads are done
            if (!ok) report_failure(a.status, 0x200u | 3u, blockIdx.x, t, tid);
            if (__syncthreads_or(!ok)) return;

            // =========================== S4: fc2 + relu (:220-221) ====================================
            put_partial<NSLOT>(PART, w, 0, lane, mfma_tile(A_fc2, ACT + fi * LDC + kbase_lane));
            __syncthreads();
            if (is_pw) publish(G4, tag, pj, prow, fmaxf(get_partial<NSLOT>(PART, 0, pu, pj) + c4v, 0.f));
            ok = sweep_layer<false, NL>(grs, soff0 + 3 * LAYER_BYTES, tag, nb, tid, ACT, nullptr, a.status);
            if (!ok) report_failure(a.status, 0x200u | 4u, blockIdx.x, t, tid);
            if (__syncthreads_or(!ok)) return;

            // =========================== S5: fc3 + sampling (:223-237) ================================
            // late prefetch (keeps 32-64 registers free during S1-S4): cI of the next step, RAW sampling noise
            {
                const int tn = (t + 1 < T) ? t + 1 : t;
                const float *crow = a.cI + ((size_t)tn * Btot + b0 + erc) * H + 2 * ec;
#pragma unroll
                for (int i = 0; i < 16; ++i) cn[i] = *reinterpret_cast<const float2 *>(crow + 32 * i);
            }
            float qn[4][8];                                   // RAW: Exp(1) variates of this wave's 4 segments
            if (MODE == 0) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int sj = 4 * w + s;
                    const int sjj = sj < nb ? sj : nb - 1;
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        qn[s][e] = a.noise[((size_t)t * Btot + b0 + sjj) * C + lane + 64 * e];
                }
            }
            if (MODE == 1) {
                put_partial<NSLOT>(PART, w, 0, lane, mfma_tile_lds(W3 + fi * LDC + kbase_lane, ACT + fi * LDC + kbase_lane));
                put_partial<NSLOT>(PART, w, 1, lane, mfma_tile_lds(W3 + (16 + fi) * LDC + kbase_lane, ACT + fi * LDC + kbase_lane));
                __syncthreads();
                {   // 30 logit rows x 16 segments = 480 sums over 256 threads: rows pu and 16+pu
                    const float lg = get_partial<NSLOT>(PART, 0, pu, pj) + B3[pu];
                    LOG[pj * 32 + pu] = lg;
                    if (a.dbg_logits && wg == 0 && pj < nb) a.dbg_logits[((size_t)t * Btot + b0 + pj) * C + pu] = lg;
                    if (pu < 14) {
                        const float lg2 = get_partial<NSLOT>(PART, 0, 16 + pu, pj) + B3[16 + pu];
                        LOG[pj * 32 + 16 + pu] = lg2;
                        if (a.dbg_logits && wg == 0 && pj < nb) a.dbg_logits[((size_t)t * Btot + b0 + pj) * C + 16 + pu] = lg2;
                    }
                }
                __syncthreads();
                {
                    // utils/distribution.py:102-121.  16-lane group = one segment (pu), lane pj = mixture.
                    float best = (pj < 10) ? mol_gumbel(LOG[pu * 32 + pj], nz0) : -INFINITY;
                    int bidx = pj;
#pragma unroll
                    for (int m = 8; m >= 1; m >>= 1) {
                        const float ob = __shfl_xor(best, m, 16);
                        const int oi = __shfl_xor(bidx, m, 16);
                        if (ob > best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
                    }
                    if (pj == 0 && pu < nb) {
                        float x = mol_sample(LOG[pu * 32 + 10 + bidx], LOG[pu * 32 + 20 + bidx], nz1);
                        if (wg == 0) a.out[(size_t)(b0 + pu) * T + t] = x;
                        if (a.force_x) x = a.force_x[(size_t)(b0 + pu) * T + t];
                        XS[pu] = x;
                    }
                }
                __syncthreads();
            } else {
                put_partial<NSLOT>(PART, w, 0, lane, mfma_tile_lds(W3 + fi * LDC + kbase_lane, ACT + fi * LDC + kbase_lane));
                __syncthreads();
                if (is_pw) publish(G5, tag, pj, prow, get_partial<NSLOT>(PART, 0, pu, pj) + B3[pu]);
                ok = sweep_layer<false, NL>(grs, soff0 + 4 * LAYER_BYTES, tag, nb, tid, ACT, nullptr, a.status);   // ACT <- logits [seg][class]
                if (!ok) report_failure(a.status, 0x200u | 5u, blockIdx.x, t, tid);
                if (__syncthreads_or(!ok)) return;
                // :232-237  softmax -> Categorical (renormalise) -> argmax(p/q); one wave per 4 segments
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int sj = 4 * w + s;
                    if (sj < nb) {                                           // wave-uniform
                        float lg[8];
                        float mx = -INFINITY;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            lg[e] = ACT[sj * LDC + lane + 64 * e];
                            if (a.dbg_logits && wg == 0) a.dbg_logits[((size_t)t * Btot + b0 + sj) * C + lane + 64 * e] = lg[e];
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
                            const float r = (lg[e] / sum2) / qn[s][e];
                            if (r > best) { best = r; bidx = lane + 64 * e; }
                        }
#pragma unroll
                        for (int m = 32; m >= 1; m >>= 1) {
                            const float ob = __shfl_xor(best, m, 64);
                            const int oi = __shfl_xor(bidx, m, 64);
                            if (ob > best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
                        }
                        if (lane == 0) {
                            float x = 2.f * (float)bidx / ((float)C - 1.f) - 1.f;
                            if (wg == 0) a.out[(size_t)(b0 + sj) * T + t] = x;
                            if (a.force_x) x = a.force_x[(size_t)(b0 + sj) * T + t];
                            XS[sj] = x;
                        }
                    }
                }
                __syncthreads();
            }
            // xi(t+1) = W_I[:,0] * x_t + cI(t+1)   (:208-209 with the conditioning part hoisted)
            {
                const float xs = XS[er];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float2 wv = *reinterpret_cast<const float2 *>(WI0 + own_col(i, ec));
                    *reinterpret_cast<float2 *>(ACT + er * LDC + own_col(i, ec)) = make_float2(fmaf(wv.x, xs, cn[i].x), fmaf(wv.y, xs, cn[i].y));
                }
            }
            __syncthreads();
        }
    }
}

template <int U, int MODE, int NL>
static hipError_t launch_cluster_t(const LoopArgs &args, int ncl, hipStream_t stream)
{
    using K = ClusterCfg<U>;
    const size_t lds = (size_t)K::LDS_FLOATS * sizeof(float);
    hipError_t e = hipFuncSetAttribute((const void *)wrnn_cluster_kernel<U, MODE, NL>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    LoopArgs a = args;
    void *params[] = {(void *)&a};
    return hipLaunchCooperativeKernel((const void *)wrnn_cluster_kernel<U, MODE, NL>, dim3(ncl * K::NWGC), dim3(NT), params,
                                      (unsigned)lds, stream);
}

// clusters a U-unit split forms on an n_cus-CU device (0 = does not fit)
int cluster_count(int U, int n_cus)
{
    if (U != 2 && U != 4 && U != 8) return 0;
    int ncl = n_cus / (H / U);
    if (ncl > U / 2) ncl = U / 2;
    if (ncl > MAXCL) ncl = MAXCL;
    while (ncl > 1 && (8 % ncl) != 0) --ncl;
    return ncl;
}

// One launch for all args.NG groups.  `nl` = sweep loads in flight (16, or 8 = split sweep; 0 = default for U).
// Returns hipErrorCooperativeLaunchTooLarge if the grid is not co-resident.
hipError_t launch_cluster(const LoopArgs &args, int U, int ncl, int mode, int nl, hipStream_t stream)
{
    if (ncl < 1) return hipErrorInvalidValue;
    if (nl == 0) nl = (U == 8) ? 8 : 16;
    if (U == 2) return mode == 1 ? launch_cluster_t<2, 1, 16>(args, ncl, stream) : launch_cluster_t<2, 0, 16>(args, ncl, stream);
    if (U == 4) return mode == 1 ? launch_cluster_t<4, 1, 16>(args, ncl, stream) : launch_cluster_t<4, 0, 16>(args, ncl, stream);
    // U = 8 exists for MOL only: RAW keeps 32 more registers of sampling noise live and would spill
    if (U == 8 && mode == 1) return nl == 16 ? launch_cluster_t<8, 1, 16>(args, ncl, stream) : launch_cluster_t<8, 1, 8>(args, ncl, stream);
    return hipErrorInvalidValue;
}

}  // namespace wrnn
