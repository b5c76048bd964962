// wrnn_stream.hip -- "stream" loop kernel: ONE workgroup per folded segment, no inter-workgroup traffic.
//
// Same per-step dataflow as wrnn_persist.hip (reference models/fatchord_version.py:201-241) but every
// workgroup re-reads the k-major weight copies from L2 / Infinity Cache each step (15.3 MB per segment-step),
// thread j owning hidden unit j.  It is the generic path (any number of segments, any RAW class count, any
// device size) and the on-GPU cross-check of the persistent kernel; it is NOT the fast path.
#include "wrnn_device.h"

namespace wrnn {

constexpr int SNT = 512;

struct BlockRed {
    float *buf;   // [2*8] floats of LDS scratch
    int *ibuf;    // [8]
};

__device__ __forceinline__ float block_max(float v, float *buf, int tid)
{
    v = wave_max64(v);
    __syncthreads();
    if ((tid & 63) == 0) buf[tid >> 6] = v;
    __syncthreads();
    float r = buf[0];
#pragma unroll
    for (int w = 1; w < SNT / 64; ++w) r = fmaxf(r, buf[w]);
    return r;
}

__device__ __forceinline__ float block_sum(float v, float *buf, int tid)
{
    v = wave_sum64(v);
    __syncthreads();
    if ((tid & 63) == 0) buf[tid >> 6] = v;
    __syncthreads();
    float r = buf[0];
#pragma unroll
    for (int w = 1; w < SNT / 64; ++w) r += buf[w];
    return r;
}

// MODE: 0 RAW, 1 MOL
template <int MODE>
__global__ __launch_bounds__(SNT) void wrnn_stream_kernel(const LoopArgs a)
{
    __shared__ float v[H], h1s[H], h2s[H], lg[H], red[16 * 32], rbuf[8];
    __shared__ int ibuf[8];
    __shared__ float xs;
    const int tid = threadIdx.x;
    const int b = a.b0 + blockIdx.x;
    const int T = a.T, C = a.C, Btot = a.Btot;
    const float wi0 = a.I_w0[tid];
    const float bi1r = a.b_ih1[tid], bi1z = a.b_ih1[H + tid], bi1n = a.b_ih1[2 * H + tid];
    const float bh1r = a.b_hh1[tid], bh1z = a.b_hh1[H + tid], bh1n = a.b_hh1[2 * H + tid];
    const float bh2r = a.b_hh2[tid], bh2z = a.b_hh2[H + tid], bh2n = a.b_hh2[2 * H + tid];
    float h1 = 0.f, h2 = 0.f;
    h1s[tid] = 0.f;
    h2s[tid] = 0.f;
    if (tid == 0) xs = 0.f;
    __syncthreads();

    for (int t = 0; t < T; ++t) {
        const int f = cond_frame(a.seg_pos, a.seg_lim, b, t, a.hop, a.NF);
        const float xi = fmaf(wi0, xs, a.cI[((size_t)t * Btot + b) * H + tid]);
        v[tid] = xi;
        __syncthreads();
        // ---- GRU1 (:210)
        float gir = 0, giz = 0, gin = 0, ghr = 0, ghz = 0, ghn = 0;
#pragma unroll 4
        for (int k = 0; k < H; ++k) {
            const float xv = v[k], hv = h1s[k];
            const float *wi = a.w_ih1T + (size_t)k * 3 * H + tid, *wh = a.w_hh1T + (size_t)k * 3 * H + tid;
            gir = fmaf(wi[0], xv, gir); giz = fmaf(wi[H], xv, giz); gin = fmaf(wi[2 * H], xv, gin);
            ghr = fmaf(wh[0], hv, ghr); ghz = fmaf(wh[H], hv, ghz); ghn = fmaf(wh[2 * H], hv, ghn);
        }
        h1 = gru_update(gir + bi1r, giz + bi1z, gin + bi1n, ghr + bh1r, ghz + bh1z, ghn + bh1n, h1);
        __syncthreads();
        h1s[tid] = h1;
        const float x1 = xi + h1;                                   // :212
        v[tid] = x1;
        __syncthreads();
        // ---- GRU2 (:213-214), conditioning columns + b_ih hoisted into c2f
        gir = giz = gin = ghr = ghz = ghn = 0.f;
#pragma unroll 4
        for (int k = 0; k < H; ++k) {
            const float xv = v[k], hv = h2s[k];
            const float *wi = a.w_ih2T + (size_t)k * 3 * H + tid, *wh = a.w_hh2T + (size_t)k * 3 * H + tid;
            gir = fmaf(wi[0], xv, gir); giz = fmaf(wi[H], xv, giz); gin = fmaf(wi[2 * H], xv, gin);
            ghr = fmaf(wh[0], hv, ghr); ghz = fmaf(wh[H], hv, ghz); ghn = fmaf(wh[2 * H], hv, ghn);
        }
        {
            const float *c2 = a.c2f + (size_t)f * 3 * H;
            h2 = gru_update(gir + c2[tid], giz + c2[H + tid], gin + c2[2 * H + tid], ghr + bh2r, ghz + bh2z,
                            ghn + bh2n, h2);
        }
        __syncthreads();
        h2s[tid] = h2;
        v[tid] = x1 + h2;                                           // :216
        __syncthreads();
        // ---- fc1 (:217-218), fc2 (:220-221)
        float y = 0.f;
#pragma unroll 8
        for (int k = 0; k < H; ++k) y = fmaf(a.fc1T[(size_t)k * H + tid], v[k], y);
        y = fmaxf(y + a.c3f[(size_t)f * H + tid], 0.f);
        __syncthreads();
        v[tid] = y;
        __syncthreads();
        y = 0.f;
#pragma unroll 8
        for (int k = 0; k < H; ++k) y = fmaf(a.fc2T[(size_t)k * H + tid], v[k], y);
        y = fmaxf(y + a.c4f[(size_t)f * H + tid], 0.f);
        __syncthreads();
        v[tid] = y;
        __syncthreads();
        // ---- fc3 (:223) + sampling
        float x;
        if (MODE == 1) {
            const int row = tid & 31, kc = tid >> 5;                // 16 k-chunks of 32
            float p = 0.f;
            if (row < 30) {
#pragma unroll 8
                for (int k = kc * 32; k < kc * 32 + 32; ++k) p = fmaf(a.fc3T[(size_t)k * C + row], v[k], p);
            }
            red[kc * 32 + row] = p;
            __syncthreads();
            if (tid < 30) {
                float s = red[tid];
                for (int q = 1; q < 16; ++q) s += red[q * 32 + tid];
                s += a.fc3_b[tid];
                lg[tid] = s;
                if (a.dbg_logits) a.dbg_logits[((size_t)t * Btot + b) * C + tid] = s;
            }
            __syncthreads();
            if (tid < 64) {                                          // utils/distribution.py:102-121
                const float *nrow = a.noise + (size_t)t * 11 * Btot;
                float best = (tid < 10) ? mol_gumbel(lg[tid], nrow[b * 10 + tid]) : -INFINITY;
                int bidx = tid;
#pragma unroll
                for (int m = 8; m >= 1; m >>= 1) {
                    const float ob = __shfl_xor(best, m, 16);
                    const int oi = __shfl_xor(bidx, m, 16);
                    if (ob > best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
                }
                if (tid == 0) {
                    x = mol_sample(lg[10 + bidx], lg[20 + bidx], nrow[10 * Btot + b]);
                    a.out[(size_t)b * T + t] = x;
                    if (a.force_x) x = a.force_x[(size_t)b * T + t];
                    xs = x;
                }
            }
        } else {
            float l = -INFINITY;
            if (tid < C) {
                l = 0.f;
#pragma unroll 8
                for (int k = 0; k < H; ++k) l = fmaf(a.fc3T[(size_t)k * C + tid], v[k], l);
                l += a.fc3_b[tid];
                if (a.dbg_logits) a.dbg_logits[((size_t)t * Btot + b) * C + tid] = l;
            }
            // :232-237 softmax -> Categorical renormalisation -> argmax(p/q), first max wins
            const float mx = block_max(l, rbuf, tid);
            float e = (tid < C) ? expf(l - mx) : 0.f;
            const float sum = block_sum(e, rbuf, tid);
            e = e / sum;
            const float sum2 = block_sum(e, rbuf, tid);
            float r = -INFINITY;
            if (tid < C) r = (e / sum2) / a.noise[((size_t)t * Btot + b) * C + tid];
            int bidx = tid;
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {
                const float ob = __shfl_xor(r, m, 64);
                const int oi = __shfl_xor(bidx, m, 64);
                if (ob > r || (ob == r && oi < bidx)) { r = ob; bidx = oi; }
            }
            __syncthreads();
            if ((tid & 63) == 0) { rbuf[tid >> 6] = r; ibuf[tid >> 6] = bidx; }
            __syncthreads();
            if (tid == 0) {
                float br = rbuf[0];
                int bi = ibuf[0];
                for (int w = 1; w < SNT / 64; ++w)
                    if (rbuf[w] > br) { br = rbuf[w]; bi = ibuf[w]; }     // ascending waves: first max wins
                x = 2.f * (float)bi / ((float)C - 1.f) - 1.f;
                a.out[(size_t)b * T + t] = x;
                if (a.force_x) x = a.force_x[(size_t)b * T + t];
                xs = x;
            }
        }
        __syncthreads();
    }
}

hipError_t launch_stream(const LoopArgs &args, int mode, hipStream_t stream)
{
    if (mode == 1) hipLaunchKernelGGL(wrnn_stream_kernel<1>, dim3(args.nb), dim3(SNT), 0, stream, args);
    else hipLaunchKernelGGL(wrnn_stream_kernel<0>, dim3(args.nb), dim3(SNT), 0, stream, args);
    return hipGetLastError();
}

}  // namespace wrnn
