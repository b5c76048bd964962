// wrnn_generic.hip -- the DIMENSION-GENERIC loop kernel: any rnn_dims / fc_dims / feat_dims / aux_dims / RAW class count.
//
// The persistent kernels (wrnn_duo.hip, wrnn_loop.hip, wrnn_sparse.hip) and the hoisted-conditioning stream kernel are built for
// the shipped hparams (rnn = fc = 512, feat 80, aux 32: hparams.py:38-44), but the reference's constructor takes any
// (models/fatchord_version.py:93-123).  For every other geometry wrnn_pack_create builds a "generic" pack (k-major copies of the
// eight matrices) and the loop runs here: ONE workgroup per folded segment, the weights re-read from L2 / MALL every step, nothing
// hoisted -- the per-step dataflow of fatchord_version.py:203-237 verbatim:
//     in0 = [x_{t-1} | m_t | a1_t] ; xi = I(in0) ; h1 = GRU(xi, h1) ; x1 = xi + h1 ; h2 = GRU([x1 | a2_t], h2) ; x2 = x1 + h2 ;
//     y1 = relu(fc1([x2 | a3_t])) ; y2 = relu(fc2([y1 | a4_t])) ; logits = fc3(y2) ; sample.
// Every dot product is ONE fmaf chain in ascending k with the bias added after it -- the order of the C oracle (oracle/wrnn_oracle.c).
// Not a fast path: it exists so that non-shipped hparams RUN on the device (and fail loudly nowhere).
#include "wrnn_device.h"

namespace wrnn {

constexpr int GNT = 512;


__device__ __forceinline__ float gblock_max(float v, float *buf, int tid)
{
    v = wave_max64(v);
    __syncthreads();
    if ((tid & 63) == 0) buf[tid >> 6] = v;
    __syncthreads();
    float r = buf[0];
#pragma unroll
    for (int w = 1; w < GNT / 64; ++w) r = fmaxf(r, buf[w]);
    return r;
}
__device__ __forceinline__ float gblock_sum(float v, float *buf, int tid)
{
    v = wave_sum64(v);
    __syncthreads();
    if ((tid & 63) == 0) buf[tid >> 6] = v;
    __syncthreads();
    float r = buf[0];
#pragma unroll
    for (int w = 1; w < GNT / 64; ++w) r += buf[w];
    return r;
}

// one output row: bias + sum_k WT[k][row] * x[k], ascending k
__device__ __forceinline__ float gdot(const float *WT, int ld, int row, const float *x, int K)
{
    float s = 0.f;
#pragma unroll 4
    for (int k = 0; k < K; ++k) s = fmaf(WT[(size_t)k * ld + row], x[k], s);
    return s;
}

constexpr int GRPT = 4;          // classes per thread at most in the RAW sampling: C <= GRPT * GNT = 2048 (also the bound on H, F)

// the three gate rows (r, z, n of unit u) of a GRU matrix pair in one pass over k: gi over K1 inputs, gh over H state values
__device__ __forceinline__ float ggru(const float *WiT, const float *WhT, const float *bi, const float *bh, int H, int u, const float *x, int K1,
                                      const float *hs)
{
    float gir = 0.f, giz = 0.f, gin = 0.f, ghr = 0.f, ghz = 0.f, ghn = 0.f;
#pragma unroll 2
    for (int k = 0; k < K1; ++k) {
        const float xv = x[k];
        const float *wi = WiT + (size_t)k * 3 * H + u;
        gir = fmaf(wi[0], xv, gir); giz = fmaf(wi[H], xv, giz); gin = fmaf(wi[2 * H], xv, gin);
    }
#pragma unroll 2
    for (int k = 0; k < H; ++k) {
        const float hv = hs[k];
        const float *wh = WhT + (size_t)k * 3 * H + u;
        ghr = fmaf(wh[0], hv, ghr); ghz = fmaf(wh[H], hv, ghz); ghn = fmaf(wh[2 * H], hv, ghn);
    }
    return gru_update(gir + bi[u], giz + bi[H + u], gin + bi[2 * H + u], ghr + bh[u], ghz + bh[H + u], ghn + bh[2 * H + u], hs[u]);
}

// MODE: 0 RAW, 1 MOL (C == 30).  Dynamic LDS: in0[1 + M + A] | va[HF] | vb[HF] | tmp[H] | h1s[H] | h2s[H] | lg[C] | red[16], HF = max(H, F) + A
template <int MODE>
__global__ __launch_bounds__(GNT) void wrnn_generic_kernel(const GenArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float gsm[];
    const int H = a.H, F = a.F, M = a.M, A = a.A, C = a.C, T = a.T, B = a.B;
    const int K0 = 1 + M + A, HF = (H > F ? H : F) + A;
    float *in0 = gsm, *va = in0 + ((K0 + 3) & ~3), *vb = va + ((HF + 3) & ~3), *tmp = vb + ((HF + 3) & ~3), *h1s = tmp + ((H + 3) & ~3);
    float *h2s = h1s + ((H + 3) & ~3), *lg = h2s + ((H + 3) & ~3), *red = lg + ((C + 3) & ~3);
    int *ired = reinterpret_cast<int *>(red + 8);
    const int tid = threadIdx.x, b = blockIdx.x;
    for (int r = tid; r < H; r += GNT) { h1s[r] = 0.f; h2s[r] = 0.f; }      // fatchord_version.py:194-196
    if (tid == 0) in0[0] = 0.f;
    __syncthreads();
    const int pos0 = a.seg_pos[b], lim = a.seg_lim[b];
    float keep[GRPT];

    for (int t = 0; t < T; ++t) {
        const int p = pos0 + t;
        const bool live = p < lim;                      // past the utterance: the fold's zero padding (:326-330)
        const float *mrow = a.mels_up + (size_t)(live ? p : 0) * M, *arow = a.aux + (size_t)((live ? p : 0) / a.hop) * 4 * A;
        for (int k = tid; k < M; k += GNT) in0[1 + k] = live ? mrow[k] : 0.f;
        for (int k = tid; k < A; k += GNT) {
            in0[1 + M + k] = live ? arow[k] : 0.f;                  // a1
            va[H + k] = live ? arow[A + k] : 0.f;                   // a2, behind x1
        }
        __syncthreads();
        // ---- I (:208-209): xi -> vb[0..H)
        for (int r = tid; r < H; r += GNT) vb[r] = gdot(a.I_T, H, r, in0, K0) + a.I_b[r];
        __syncthreads();
        // ---- rnn1 (:210) on xi, then x1 = xi + h1 (:212) -> va[0..H)
        for (int u = tid; u < H; u += GNT) tmp[u] = ggru(a.w_ih1T, a.w_hh1T, a.b_ih1, a.b_hh1, H, u, vb, H, h1s);
        __syncthreads();                                // every thread has read the old h1
        for (int u = tid; u < H; u += GNT) { const float hn = tmp[u]; h1s[u] = hn; va[u] = vb[u] + hn; }
        __syncthreads();
        // ---- rnn2 (:213-214) on [x1 | a2], then x2 = x1 + h2 (:216) -> vb[0..H), a3 behind it
        for (int u = tid; u < H; u += GNT) tmp[u] = ggru(a.w_ih2T, a.w_hh2T, a.b_ih2, a.b_hh2, H, u, va, H + A, h2s);
        __syncthreads();
        for (int u = tid; u < H; u += GNT) { const float hn = tmp[u]; h2s[u] = hn; vb[u] = va[u] + hn; }
        for (int k = tid; k < A; k += GNT) vb[H + k] = live ? arow[2 * A + k] : 0.f;      // a3
        __syncthreads();
        // ---- fc1 (:217-218) on [x2 | a3] -> va[0..F), a4 behind it (va is free: x1 | a2 were last read by rnn2)
        for (int r = tid; r < F; r += GNT) va[r] = fmaxf(gdot(a.fc1T, F, r, vb, H + A) + a.fc1_b[r], 0.f);
        for (int k = tid; k < A; k += GNT) va[F + k] = live ? arow[3 * A + k] : 0.f;      // a4
        __syncthreads();
        // ---- fc2 (:220-221) on [y1 | a4] -> vb[0..F)
        for (int r = tid; r < F; r += GNT) vb[r] = fmaxf(gdot(a.fc2T, F, r, va, F + A) + a.fc2_b[r], 0.f);
        __syncthreads();
        // ---- fc3 (:223)
        for (int c = tid; c < C; c += GNT) {
            const float l = gdot(a.fc3T, C, c, vb, F) + a.fc3_b[c];
            lg[c] = l;
            if (a.dbg_logits) a.dbg_logits[((size_t)t * B + b) * C + c] = l;
        }
        __syncthreads();
        // ---- sampling
        if (MODE == 1) {                                // utils/distribution.py:102-121 (C = 30: 10 mixtures)
            if (tid < 64) {
                const float *nrow = a.noise + (size_t)t * 11 * B;
                float best = (tid < 10) ? mol_gumbel(lg[tid], nrow[b * 10 + tid]) : -INFINITY;
                int bidx = tid;
#pragma unroll
                for (int m = 8; m >= 1; m >>= 1) {
                    const float ob = __shfl_xor(best, m, 16);
                    const int oi = __shfl_xor(bidx, m, 16);
                    if (ob > best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
                }
                if (tid == 0) {
                    float x = mol_sample(lg[10 + bidx], lg[20 + bidx], nrow[10 * B + b]);
                    a.out[(size_t)b * T + t] = x;
                    if (a.force_x) x = a.force_x[(size_t)b * T + t];
                    in0[0] = x;
                }
            }
        } else {                                        // :232-237 softmax -> Categorical renormalisation -> argmax(p / q), first max wins
            float lmax = -INFINITY;
            for (int c = tid; c < C; c += GNT) lmax = fmaxf(lmax, lg[c]);
            const float mx = gblock_max(lmax, red, tid);
            float part = 0.f;
#pragma unroll
            for (int q = 0; q < GRPT; ++q) {
                const int c = tid + q * GNT;
                keep[q] = (c < C) ? expf(lg[c] - mx) : 0.f;
                part += keep[q];
            }
            const float sum = gblock_sum(part, red, tid);
            part = 0.f;
#pragma unroll
            for (int q = 0; q < GRPT; ++q) { keep[q] = keep[q] / sum; part += keep[q]; }
            const float sum2 = gblock_sum(part, red, tid);
            float rbest = -INFINITY;
            int bidx = tid;
#pragma unroll
            for (int q = 0; q < GRPT; ++q) {
                const int c = tid + q * GNT;
                if (c < C) {
                    const float rr = (keep[q] / sum2) / a.noise[((size_t)t * B + b) * C + c];
                    if (rr > rbest) { rbest = rr; bidx = c; }          // ascending c: the first maximum stays
                }
            }
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {
                const float ob = __shfl_xor(rbest, m, 64);
                const int oi = __shfl_xor(bidx, m, 64);
                if (ob > rbest || (ob == rbest && oi < bidx)) { rbest = ob; bidx = oi; }
            }
            __syncthreads();
            if ((tid & 63) == 0) { red[tid >> 6] = rbest; ired[tid >> 6] = bidx; }
            __syncthreads();
            if (tid == 0) {
                float br = red[0];
                int bi = ired[0];
                for (int w = 1; w < GNT / 64; ++w)
                    if (red[w] > br || (red[w] == br && ired[w] < bi)) { br = red[w]; bi = ired[w]; }
                float x = 2.f * (float)bi / ((float)C - 1.f) - 1.f;
                a.out[(size_t)b * T + t] = x;
                if (a.force_x) x = a.force_x[(size_t)b * T + t];
                in0[0] = x;
            }
        }
        __syncthreads();
    }
}

size_t generic_lds_bytes(int H, int F, int M, int A, int C)
{
    const int K0 = 1 + M + A, HF = (H > F ? H : F) + A;
    return (size_t)(((K0 + 3) & ~3) + 2 * ((HF + 3) & ~3) + 3 * ((H + 3) & ~3) + ((C + 3) & ~3) + 16 + 16) * sizeof(float);
}

// dims this kernel can take
bool generic_dims_ok(int H, int F, int M, int A, int C, int mode)
{
    if (H < 1 || F < 1 || M < 1 || A < 1 || H > GRPT * GNT || F > GRPT * GNT || C > GRPT * GNT || M + A > 1024) return false;
    if (mode == 1 && C != 30) return false;
    if (mode == 0 && C < 2) return false;
    return generic_lds_bytes(H, F, M, A, C) <= 64 * 1024;
}

hipError_t launch_generic(const GenArgs &args, int mode, hipStream_t stream)
{
    const size_t lds = generic_lds_bytes(args.H, args.F, args.M, args.A, args.C);
    if (mode == 1) hipLaunchKernelGGL(wrnn_generic_kernel<1>, dim3(args.B), dim3(GNT), lds, stream, args);
    else hipLaunchKernelGGL(wrnn_generic_kernel<0>, dim3(args.B), dim3(GNT), lds, stream, args);
    return hipGetLastError();
}

}  // namespace wrnn
