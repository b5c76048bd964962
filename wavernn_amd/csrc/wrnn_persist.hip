// wrnn_persist.hip -- the persistent chip-wide WaveRNN loop kernel for MI355X (gfx950 / CDNA4).
//
// Replaces the `for i in range(seq_len)` loop of fatchord/WaveRNN `WaveRNN.generate()`
// (reference models/fatchord_version.py:201-241) for a group of <= 16 folded segments.
//
// Design (DESIGN.md section "K-loop"):
//   * ONE cooperative launch of NWG = 512/U workgroups (U hidden units per workgroup; U=2 -> 256 WGs =
//     one per CU) lives for all T steps.  The 15.3 MB of fp32 loop weights do not fit one CU's LDS, so
//     they are partitioned over the whole chip: workgroup g owns hidden units [U*g, U*g+U) of both GRUs,
//     the same rows of fc1/fc2 and (MOL) a replica of the 30-row fc3.
//   * weights are gathered ONCE into MFMA A-fragments that stay in VGPRs for the whole kernel
//     (v_mfma_f32_16x16x4_f32: exact f32, rows = gate rows of the owned units, cols = the 16 segments);
//     the 512-wide activation vectors of the 16 segments live in LDS ([seg][k], stride 516) and are the
//     B operands (ds_read_b128).  4 waves (one per SIMD, 512-VGPR budget each) split K = 512; partial
//     tiles are summed through LDS in a fixed order.
//   * after each of the 4 dependent layers (GRU1, GRU2, fc1, fc2; RAW: + fc3) every workgroup publishes
//     its U x nb outputs as 8-byte {step tag, f32} granules (one relaxed agent-scope = sc1 store each) and
//     sweeps all 512 x nb granules of that layer (relaxed agent-scope loads, bounded spin).  No fences,
//     no flags: the data is its own flag (MI355X guide, Guideline 16 form R2).
//   * the sampling tail (fc3 + mixture-of-logistics / softmax sampling) is computed redundantly by every
//     workgroup from identical inputs, so x_t needs no further exchange; wavefront shuffles do the
//     argmax / softmax reductions.
//   * conditioning products that do not depend on the recurrence were hoisted (wrnn_cond.hip).
//   * GRU hidden-to-hidden products (W_hh.h) depend only on the previous step: they run while this
//     workgroup's granules are in flight.
#include "wrnn_device.h"

namespace wrnn {

// LDS carve (floats)
constexpr int OFF_ACT = 0;
constexpr int OFF_HS1 = OFF_ACT + SEG * LDA;
constexpr int OFF_HS2 = OFF_HS1 + SEG * LDA;
constexpr int OFF_PART = OFF_HS2 + SEG * LDA;          // [NW][2][16][16]
constexpr int OFF_GH1 = OFF_PART + NW * 2 * 256;       // [16][SEG]
constexpr int OFF_GH2 = OFF_GH1 + 16 * SEG;
constexpr int OFF_HOWN1 = OFF_GH2 + 16 * SEG;          // [4][SEG]
constexpr int OFF_HOWN2 = OFF_HOWN1 + 4 * SEG;
constexpr int OFF_LOG = OFF_HOWN2 + 4 * SEG;           // [SEG][32]
constexpr int OFF_XS = OFF_LOG + SEG * 32;             // [SEG]
constexpr int OFF_WI0 = OFF_XS + SEG;                  // [H]  I.weight[:,0]
constexpr int LDS_FLOATS = OFF_WI0 + H;

size_t persist_lds_bytes() { return (size_t)LDS_FLOATS * sizeof(float); }

// U: hidden units per workgroup (2 or 4).  MODE: 0 RAW (C == 512), 1 MOL (C == 30).
template <int U, int MODE>
__global__ __launch_bounds__(NT, 1) void wrnn_persist_kernel(const LoopArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *ACT = smem + OFF_ACT, *HS1 = smem + OFF_HS1, *HS2 = smem + OFF_HS2, *PART = smem + OFF_PART;
    float *GH1 = smem + OFF_GH1, *GH2 = smem + OFF_GH2, *HOWN1 = smem + OFF_HOWN1, *HOWN2 = smem + OFF_HOWN2;
    float *LOG = smem + OFF_LOG, *XS = smem + OFF_XS, *WI0 = smem + OFF_WI0;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wg = blockIdx.x;
    const int fi = lane & 15, kq = lane >> 4;          // MFMA fragment row / k-quad of this lane
    const int kbase_lane = KCH * w + 4 * kq;
    const int nb = a.nb, b0 = a.b0, Btot = a.Btot, T = a.T, C = a.C;
    // elementwise / sweep role: row er (segment) and the 16 column pairs own_col(i, ec), i = 0..15
    const int er = tid >> 4, ec = tid & 15;
    const int erc = er < nb ? er : nb - 1;             // clamped row for conditioning reads of unused rows

    // ---- one-time: gather the weight slice into VGPR-resident MFMA A fragments ----------------------
    // GRU tiles: fragment row i = gate*U + u  (gate 0..2 = r,z,n; u = owned unit) ; fc tiles: row i = u.
    float A_ih1[AF], A_hh1[AF], A_ih2[AF], A_hh2[AF], A_fc1[AF], A_fc2[AF], A_o0[AF], A_o1[AF];
    {
        const bool vg = fi < 3 * U;
        const int grow = (fi / U) * H + U * wg + (fi % U);
        load_afrag(A_ih1, a.w_ih1, H, grow, vg, kbase_lane);
        load_afrag(A_hh1, a.w_hh1, H, grow, vg, kbase_lane);
        load_afrag(A_ih2, a.w_ih2, H + AUX, grow, vg, kbase_lane);
        load_afrag(A_hh2, a.w_hh2, H, grow, vg, kbase_lane);
        const bool vf = fi < U;
        const int frow = U * wg + fi;
        load_afrag(A_fc1, a.fc1_w, H + AUX, frow, vf, kbase_lane);
        load_afrag(A_fc2, a.fc2_w, H + AUX, frow, vf, kbase_lane);
        if (MODE == 1) {   // replicated 30-row fc3: rows 0..15 and 16..29
            load_afrag(A_o0, a.fc3_w, H, fi, true, kbase_lane);
            load_afrag(A_o1, a.fc3_w, H, 16 + fi, fi < 14, kbase_lane);
        } else {           // distributed fc3: rows U*wg .. U*wg+U-1
            load_afrag(A_o0, a.fc3_w, H, frow, vf, kbase_lane);
#pragma unroll
            for (int q = 0; q < AF; ++q) A_o1[q] = 0.f;
        }
    }

    // per-thread constants
    const int pu = tid >> 4, pj = tid & 15;             // pointwise role: (owned unit, segment) for tid < 16U
    const bool is_pw = (tid < 16 * U) && (pj < nb);
    const int prow = U * wg + (pu % U);                 // hidden index this pointwise thread owns
    float bi1r = 0, bi1z = 0, bi1n = 0;
    if (tid < 16 * U) { bi1r = a.b_ih1[prow]; bi1z = a.b_ih1[H + prow]; bi1n = a.b_ih1[2 * H + prow]; }
    const bool is_gh = tid < 48 * U;                    // gh-reduce role: fragment row pu, segment pj
    float bh1 = 0, bh2 = 0;
    if (is_gh) {
        const int grow = (pu / U) * H + U * wg + (pu % U);
        bh1 = a.b_hh1[grow];
        bh2 = a.b_hh2[grow];
    }
    float b3a = 0.f, b3b = 0.f;                         // fc3 bias of the logit row(s) this thread reduces
    if (MODE == 1) { b3a = a.fc3_b[pu]; if (tid + NT < 480) b3b = a.fc3_b[(tid + NT) >> 4]; }
    else { if (tid < 16 * U) b3a = a.fc3_b[prow]; }

    u64 *G1 = a.gran + 0 * SEG * H, *G2 = a.gran + 1 * SEG * H, *G3 = a.gran + 2 * SEG * H;
    u64 *G4 = a.gran + 3 * SEG * H, *G5 = a.gran + 4 * SEG * H;
    const __amdgpu_buffer_rsrc_t grs = make_rsrc(a.gran, NGRAN * SEG * H * 8);

    // ---- state init (fatchord_version.py:194-196: h1 = h2 = 0, x = 0) ------------------------------
    for (int q = tid; q < 2 * SEG * LDA; q += NT) HS1[q] = 0.f;    // HS1 and HS2 are contiguous
    if (tid < 8 * SEG) HOWN1[tid] = 0.f;                            // HOWN1 + HOWN2
    if (tid < SEG) XS[tid] = 0.f;
    WI0[2 * tid] = a.I_w0[2 * tid];                                 // I.weight[:,0]: tap of x_{t-1}
    WI0[2 * tid + 1] = a.I_w0[2 * tid + 1];
    float2 cn[16];                                                   // cI of the NEXT step, row er, owned columns
    {
        const float *crow = a.cI + ((size_t)0 * Btot + b0 + erc) * H + 2 * ec;
#pragma unroll
        for (int i = 0; i < 16; ++i) cn[i] = *reinterpret_cast<const float2 *>(crow + 32 * i);
    }
    __syncthreads();
    {                                                                // xi(0)
        const float xs = XS[er];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float2 wv = *reinterpret_cast<const float2 *>(WI0 + own_col(i, ec));
            *reinterpret_cast<float2 *>(ACT + er * LDA + own_col(i, ec)) = make_float2(fmaf(wv.x, xs, cn[i].x), fmaf(wv.y, xs, cn[i].y));
        }
    }
    {   // gh1 of step 0 = W_hh1 . 0 + b_hh1 (run the generic path so every step is identical)
        const f32x4 acc = mfma_tile(A_hh1, HS1 + fi * LDA + kbase_lane);
        store_partial(PART, w, 1, lane, acc);
    }
    __syncthreads();
    if (is_gh) GH1[pu * SEG + pj] = reduce_partial(PART, 1, pu, pj) + bh1;
    __syncthreads();

    for (int t = 0; t < T; ++t) {
        const unsigned tag = (unsigned)t + 1u;
        // ---- prefetch everything of this / the next step that does not depend on the recurrence ----
        {
            const int tn = (t + 1 < T) ? t + 1 : t;
            const float *crow = a.cI + ((size_t)tn * Btot + b0 + erc) * H + 2 * ec;
#pragma unroll
            for (int i = 0; i < 16; ++i) cn[i] = *reinterpret_cast<const float2 *>(crow + 32 * i);
        }
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
        float qn[4][8];                                   // RAW: Exp(1) variates of this wave's 4 segments
        if (MODE == 1) {
            if (pu < nb) {
                const float *nrow = a.noise + (size_t)t * 11 * Btot;
                if (pj < 10) nz0 = nrow[(b0 + pu) * 10 + pj];
                if (pj == 0) nz1 = nrow[10 * Btot + b0 + pu];
            }
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int sj = 4 * w + s;
                const int sjj = sj < nb ? sj : nb - 1;
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    qn[s][e] = a.noise[((size_t)t * Btot + b0 + sjj) * C + lane + 64 * e];
            }
        }

        // =========================== S1: GRU1 (fatchord_version.py:210) ===========================
        {
            const f32x4 acc = mfma_tile(A_ih1, ACT + fi * LDA + kbase_lane);
            store_partial(PART, w, 0, lane, acc);
        }
        __syncthreads();
        if (is_pw) {
            const float gir = reduce_partial(PART, 0, 0 * U + pu, pj) + bi1r;
            const float giz = reduce_partial(PART, 0, 1 * U + pu, pj) + bi1z;
            const float gin = reduce_partial(PART, 0, 2 * U + pu, pj) + bi1n;
            const float hn = gru_update(gir, giz, gin, GH1[(0 * U + pu) * SEG + pj], GH1[(1 * U + pu) * SEG + pj],
                                        GH1[(2 * U + pu) * SEG + pj], HOWN1[pu * SEG + pj]);
            HOWN1[pu * SEG + pj] = hn;
            publish(G1, tag, pj, prow, hn);
        }
        {   // off the critical path: gh2(t) = W_hh2 . h2(t-1) while the granules travel
            const f32x4 acc = mfma_tile(A_hh2, HS2 + fi * LDA + kbase_lane);
            store_partial(PART, w, 1, lane, acc);
        }
        __syncthreads();
        if (is_gh) GH2[pu * SEG + pj] = reduce_partial(PART, 1, pu, pj) + bh2;
        bool ok = sweep(grs, 0, tag, nb, tid, HS1, a.status);
        if (!ok) report_failure(a.status, 0x100u | 1u, wg, t, tid);
        if (__syncthreads_or(!ok)) return;
        if (er < nb) {                                                                    // :212 x = x + h1
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float2 x = *reinterpret_cast<float2 *>(ACT + er * LDA + own_col(i, ec));
                const float2 h = *reinterpret_cast<const float2 *>(HS1 + er * LDA + own_col(i, ec));
                x.x += h.x; x.y += h.y;
                *reinterpret_cast<float2 *>(ACT + er * LDA + own_col(i, ec)) = x;
            }
        }
        __syncthreads();

        // =========================== S2: GRU2 (:213-214) ==========================================
        {
            const f32x4 acc = mfma_tile(A_ih2, ACT + fi * LDA + kbase_lane);
            store_partial(PART, w, 0, lane, acc);
        }
        __syncthreads();
        if (is_pw) {
            const float gir = reduce_partial(PART, 0, 0 * U + pu, pj) + c2r;
            const float giz = reduce_partial(PART, 0, 1 * U + pu, pj) + c2z;
            const float gin = reduce_partial(PART, 0, 2 * U + pu, pj) + c2n;
            const float hn = gru_update(gir, giz, gin, GH2[(0 * U + pu) * SEG + pj], GH2[(1 * U + pu) * SEG + pj],
                                        GH2[(2 * U + pu) * SEG + pj], HOWN2[pu * SEG + pj]);
            HOWN2[pu * SEG + pj] = hn;
            publish(G2, tag, pj, prow, hn);
        }
        {   // off the critical path: gh1(t+1) = W_hh1 . h1(t)
            const f32x4 acc = mfma_tile(A_hh1, HS1 + fi * LDA + kbase_lane);
            store_partial(PART, w, 1, lane, acc);
        }
        __syncthreads();
        if (is_gh) GH1[pu * SEG + pj] = reduce_partial(PART, 1, pu, pj) + bh1;
        ok = sweep(grs, 1, tag, nb, tid, HS2, a.status);
        if (!ok) report_failure(a.status, 0x100u | 2u, wg, t, tid);
        if (__syncthreads_or(!ok)) return;
        if (er < nb) {                                                                    // :216 x = x + h2
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float2 x = *reinterpret_cast<float2 *>(ACT + er * LDA + own_col(i, ec));
                const float2 h = *reinterpret_cast<const float2 *>(HS2 + er * LDA + own_col(i, ec));
                x.x += h.x; x.y += h.y;
                *reinterpret_cast<float2 *>(ACT + er * LDA + own_col(i, ec)) = x;
            }
        }
        __syncthreads();

        // =========================== S3: fc1 + relu (:217-218) ====================================
        {
            const f32x4 acc = mfma_tile(A_fc1, ACT + fi * LDA + kbase_lane);
            store_partial(PART, w, 0, lane, acc);
        }
        __syncthreads();
        if (is_pw) publish(G3, tag, pj, prow, fmaxf(reduce_partial(PART, 0, pu, pj) + c3v, 0.f));
        ok = sweep(grs, 2, tag, nb, tid, ACT, a.status);       // every wave passed the barrier => ACT reads are done
        if (!ok) report_failure(a.status, 0x100u | 3u, wg, t, tid);
        if (__syncthreads_or(!ok)) return;

        // =========================== S4: fc2 + relu (:220-221) ====================================
        {
            const f32x4 acc = mfma_tile(A_fc2, ACT + fi * LDA + kbase_lane);
            store_partial(PART, w, 0, lane, acc);
        }
        __syncthreads();
        if (is_pw) publish(G4, tag, pj, prow, fmaxf(reduce_partial(PART, 0, pu, pj) + c4v, 0.f));
        ok = sweep(grs, 3, tag, nb, tid, ACT, a.status);
        if (!ok) report_failure(a.status, 0x100u | 4u, wg, t, tid);
        if (__syncthreads_or(!ok)) return;

        // =========================== S5: fc3 + sampling (:223-237) ================================
        if (MODE == 1) {
            {
                const f32x4 acc0 = mfma_tile(A_o0, ACT + fi * LDA + kbase_lane);
                store_partial(PART, w, 0, lane, acc0);
                const f32x4 acc1 = mfma_tile(A_o1, ACT + fi * LDA + kbase_lane);
                store_partial(PART, w, 1, lane, acc1);
            }
            __syncthreads();
            {   // 30 logit rows x 16 segments = 480 sums over 256 threads: rows pu and pu+16
                const float lg = reduce_partial(PART, 0, pu, pj) + b3a;
                LOG[pj * 32 + pu] = lg;
                if (a.dbg_logits && wg == 0 && pj < nb) a.dbg_logits[((size_t)t * Btot + b0 + pj) * C + pu] = lg;
                if (pu < 14) {
                    const float lg2 = reduce_partial(PART, 1, pu, pj) + b3b;
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
            {
                const f32x4 acc0 = mfma_tile(A_o0, ACT + fi * LDA + kbase_lane);
                store_partial(PART, w, 0, lane, acc0);
            }
            __syncthreads();
            if (is_pw) publish(G5, tag, pj, prow, reduce_partial(PART, 0, pu, pj) + b3a);
            ok = sweep(grs, 4, tag, nb, tid, ACT, a.status);                 // ACT <- logits [seg][class]
            if (!ok) report_failure(a.status, 0x100u | 5u, wg, t, tid);
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
                        lg[e] = ACT[sj * LDA + lane + 64 * e];
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
                *reinterpret_cast<float2 *>(ACT + er * LDA + own_col(i, ec)) = make_float2(fmaf(wv.x, xs, cn[i].x), fmaf(wv.y, xs, cn[i].y));
            }
        }
        __syncthreads();
    }
}

template <int U, int MODE>
static hipError_t launch_t(const LoopArgs &args, int nwg, hipStream_t stream)
{
    const size_t lds = persist_lds_bytes();
    hipError_t e = hipFuncSetAttribute((const void *)wrnn_persist_kernel<U, MODE>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    LoopArgs a = args;
    void *params[] = {(void *)&a};
    return hipLaunchCooperativeKernel((const void *)wrnn_persist_kernel<U, MODE>, dim3(nwg), dim3(NT), params,
                                      (unsigned)lds, stream);
}

// U = 2 -> 256 workgroups, U = 4 -> 128.  Returns hipErrorCooperativeLaunchTooLarge if not co-resident.
hipError_t launch_persist(const LoopArgs &args, int U, int mode, hipStream_t stream)
{
    if (U == 2) return mode == 1 ? launch_t<2, 1>(args, H / 2, stream) : launch_t<2, 0>(args, H / 2, stream);
    if (U == 4) return mode == 1 ? launch_t<4, 1>(args, H / 4, stream) : launch_t<4, 0>(args, H / 4, stream);
    return hipErrorInvalidValue;
}

}  // namespace wrnn
