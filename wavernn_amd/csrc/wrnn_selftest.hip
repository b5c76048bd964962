// wrnn_selftest.hip -- device self tests of the primitives the persistent kernel is built from:
//   1. MFMA tile: load_afrag / mfma_tile / store_partial / reduce_partial against a host fp64 reference
//      (asymmetric operands, so a transposed fragment layout cannot pass).
//   2. granule all-gather: NWG co-resident workgroups publish + sweep {tag,value} granules for many rounds,
//      every word checked, bounded spins; reports microseconds per round.
//   3. tanh_sel (the branch-free tanh of the fused stages) == tanhf, bit for bit, over a dense sweep of arguments.
//   4. xor_pair (DPP / v_permlane*_swap butterflies of the RAW sampler's reductions) == __shfl_xor, bit for bit.
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "wrnn_device.h"

namespace wrnn {

__global__ __launch_bounds__(NT, 1) void selftest_mfma_kernel(const float *W /*[16][512]*/, const float *X /*[16][512]*/,
                                                            float *D /*[16][16]*/)
{
    __shared__ __attribute__((aligned(16))) float act[SEG * LDA];
    __shared__ float part[NW * 2 * 256];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, fi = lane & 15, kq = lane >> 4;
    const int kbase_lane = KCH * w + 4 * kq;
    float A[AF];
    load_afrag(A, W, H, fi, true, kbase_lane);
    for (int j = 0; j < SEG; ++j) {
        act[j * LDA + 2 * tid] = X[j * H + 2 * tid];
        act[j * LDA + 2 * tid + 1] = X[j * H + 2 * tid + 1];
    }
    __syncthreads();
    const f32x4 acc = mfma_tile(A, act + fi * LDA + kbase_lane);
    store_partial(part, w, 1, lane, acc);
    __syncthreads();
    D[tid] = reduce_partial(part, 1, tid >> 4, tid & 15);
}

int selftest_mfma(char *msg, size_t n)
{
    std::vector<float> W(16 * H), X(16 * H), D(256);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (auto &v : W) v = rnd();
    for (auto &v : X) v = rnd() * 2.f + 0.25f;
    float *dW, *dX, *dD;
    if (hipMalloc(&dW, W.size() * 4) != hipSuccess || hipMalloc(&dX, X.size() * 4) != hipSuccess || hipMalloc(&dD, 1024) != hipSuccess) {
        snprintf(msg, n, "hipMalloc failed");
        return 1;
    }
    hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(selftest_mfma_kernel, dim3(1), dim3(NT), 0, 0, dW, dX, dD);
    hipError_t e = hipDeviceSynchronize();
    hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
    hipFree(dW); hipFree(dX); hipFree(dD);
    if (e != hipSuccess) { snprintf(msg, n, "kernel failed: %s", hipGetErrorString(e)); return 1; }
    double worst = 0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double ref = 0;
            for (int k = 0; k < H; ++k) ref += (double)W[i * H + k] * (double)X[j * H + k];
            const double err = fabs(ref - (double)D[i * 16 + j]);
            if (err > worst) worst = err;
        }
    snprintf(msg, n, "mfma tile max abs err %.3e", worst);
    return worst < 1e-3 ? 0 : 1;
}

// ---- tanh_sel == tanhf ---------------------------------------------------------------------------------
// 2^24 arguments: every 2^8-th float bit pattern of both signs (all exponents, NaN / inf / denormals included)
__global__ void selftest_tanh_kernel(unsigned *mismatches, unsigned *first)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    const float x = __uint_as_float(i << 8 | (i & 0xFFu));
    const unsigned a = __float_as_uint(tanhf(x)), b = __float_as_uint(tanh_sel(x));
    const bool nan_both = (a & 0x7FFFFFFFu) > 0x7F800000u && (b & 0x7FFFFFFFu) > 0x7F800000u;
    if (a != b && !nan_both) {
        if (atomicAdd(mismatches, 1u) == 0u) *first = __float_as_uint(x);
    }
}

int selftest_tanh(char *msg, size_t n)
{
    unsigned *d, h[2] = {0u, 0u};
    if (hipMalloc(&d, 8) != hipSuccess) { snprintf(msg, n, "hipMalloc failed"); return 1; }
    hipMemcpy(d, h, 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(selftest_tanh_kernel, dim3(1u << 16), dim3(256), 0, 0, d, d + 1);
    hipError_t e = hipDeviceSynchronize();
    hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    hipFree(d);
    if (e != hipSuccess) { snprintf(msg, n, "kernel failed: %s", hipGetErrorString(e)); return 1; }
    snprintf(msg, n, "tanh_sel vs tanhf over 2^24 arguments: %u bit mismatches (first at 0x%08x)", h[0], h[1]);
    return h[0] == 0u ? 0 : 1;
}

// ---- xor_pair / wave_sum64 / wave_max64 / wave_argmax64 == the __shfl_xor forms, bit for bit ----------------------
template <int M> __device__ __forceinline__ unsigned xor_pair_bad(unsigned v)
{
    unsigned a, b;
    xor_pair_u<M>(v, a, b);
    const unsigned o = (unsigned)__shfl_xor((int)v, M, 64);
    return ((a == v && b == o) || (a == o && b == v)) ? 0u : 1u;
}
__global__ void selftest_xor_kernel(unsigned *mismatches)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned h = i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    // floats of mixed sign and magnitude (sums that round differently under another tree), every 7th wave small non-negative values with ties
    float v = __uint_as_float((h & 0x807FFFFFu) | ((100u + (h >> 23 & 31u)) << 23));
    if ((i >> 6) % 7u == 0u) v = (float)(h & 7u) * 0.125f;
    unsigned bad = xor_pair_bad<32>(h) + xor_pair_bad<16>(h) + xor_pair_bad<8>(h) + xor_pair_bad<4>(h) + xor_pair_bad<2>(h) + xor_pair_bad<1>(h);
    float s = v, m = v, best = v;
    int bidx = (int)(threadIdx.x & 63u) + 64 * (int)(h & 7u);
    int bi2 = bidx;
    float b2 = best;
    for (int k = 32; k >= 1; k >>= 1) {
        s += __shfl_xor(s, k, 64);
        m = fmaxf(m, __shfl_xor(m, k, 64));
        const float ob = __shfl_xor(b2, k, 64);
        const int oi = __shfl_xor(bi2, k, 64);
        if (ob > b2 || (ob == b2 && oi < bi2)) { b2 = ob; bi2 = oi; }
    }
    wave_argmax64(best, bidx);
    bad += __float_as_uint(wave_sum64(v)) != __float_as_uint(s);
    bad += __float_as_uint(wave_max64(v)) != __float_as_uint(m);
    bad += __float_as_uint(best) != __float_as_uint(b2) || bidx != bi2;
    if (bad) atomicAdd(mismatches, bad);
}
int selftest_xor(char *msg, size_t n)
{
    unsigned *d, h = 0u;
    if (hipMalloc(&d, 4) != hipSuccess) { snprintf(msg, n, "hipMalloc failed"); return 1; }
    hipMemcpy(d, &h, 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(selftest_xor_kernel, dim3(1u << 12), dim3(256), 0, 0, d);
    hipError_t e = hipDeviceSynchronize();
    hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    hipFree(d);
    if (e != hipSuccess) { snprintf(msg, n, "kernel failed: %s", hipGetErrorString(e)); return 1; }
    snprintf(msg, n, "xor_pair / wave_sum64 / wave_max64 / wave_argmax64 vs __shfl_xor over 2^14 waves: %u mismatches", h);
    return h == 0u ? 0 : 1;
}

// ---- all-gather ------------------------------------------------------------------------------------
__device__ __forceinline__ float ag_value(int k, int j, int round) { return (float)(k * 31 + j * 7 + round % 1000) * 0.25f; }

template <int U>
__global__ __launch_bounds__(NT, 1) void selftest_allgather_kernel(u64 *G /*[2][SEG][H]*/, unsigned *status, int nb, int rounds,
                                                                  unsigned *errors)
{
    __shared__ __attribute__((aligned(16))) float tile[SEG * LDA];
    const int tid = threadIdx.x, wg = blockIdx.x;
    const int pu = tid >> 4, pj = tid & 15;
    unsigned bad = 0;
    const __amdgpu_buffer_rsrc_t grs = make_rsrc(G, 2 * SEG * H * 8);
    for (int r = 0; r < rounds; ++r) {
        u64 *Gr = G + (size_t)(r & 1) * SEG * H;          // two layers alternate, like consecutive layers of a step
        const unsigned tag = (unsigned)r / 2u + 1u;
        if (tid < 16 * U && pj < nb) {
            const int k = U * wg + pu;
            publish(Gr, tag, pj, k, ag_value(k, pj, r));
        }
        const bool ok = sweep(grs, r & 1, tag, nb, tid, tile, status);
        if (!ok) report_failure(status, 0x200u, wg, r, tid);
        if (__syncthreads_or(!ok)) return;
        for (int j = 0; j < nb; ++j) {
            // check columns written by OTHER threads too (tid+1 mod NT), after the barrier
            const int c = 2 * ((tid + 1) % NT);
            if (tile[j * LDA + c] != ag_value(c, j, r) || tile[j * LDA + c + 1] != ag_value(c + 1, j, r)) ++bad;
        }
        __syncthreads();
    }
    if (bad) atomicAdd(errors, bad);
}

int selftest_allgather(int n_cus, char *msg, size_t n, float *us_per_round)
{
    const int U = (n_cus >= H / 2) ? 2 : 4;
    const int nwg = H / U;
    if (n_cus < nwg) { snprintf(msg, n, "device has %d CUs; need >= %d", n_cus, nwg); return 1; }
    const int rounds = 2000, nb = 12;
    u64 *G; unsigned *st;
    if (hipMalloc(&G, (size_t)2 * SEG * H * 8) != hipSuccess || hipMalloc(&st, 256) != hipSuccess) { snprintf(msg, n, "hipMalloc failed"); return 1; }
    hipMemset(G, 0, (size_t)2 * SEG * H * 8);
    hipMemset(st, 0, 256);
    unsigned *errors = st + 32;
    int nb_ = nb, rounds_ = rounds;
    void *params[] = {(void *)&G, (void *)&st, (void *)&nb_, (void *)&rounds_, (void *)&errors};
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipError_t e = (U == 2) ? hipLaunchCooperativeKernel((const void *)selftest_allgather_kernel<2>, dim3(nwg), dim3(NT), params, 0, 0)
                            : hipLaunchCooperativeKernel((const void *)selftest_allgather_kernel<4>, dim3(nwg), dim3(NT), params, 0, 0);
    hipEventRecord(e1, 0);
    if (e != hipSuccess) { snprintf(msg, n, "cooperative launch failed: %s", hipGetErrorString(e)); hipFree(G); hipFree(st); return 1; }
    e = hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned h[64];
    hipMemcpy(h, st, 256, hipMemcpyDeviceToHost);
    hipFree(G); hipFree(st);
    hipEventDestroy(e0); hipEventDestroy(e1);
    if (e != hipSuccess) { snprintf(msg, n, "kernel failed: %s", hipGetErrorString(e)); return 1; }
    *us_per_round = ms * 1000.f / rounds;
    if (h[0] || h[1]) { snprintf(msg, n, "gave up: code 0x%x wg %u round %u tid %u", h[1], h[2], h[3], h[4]); return 1; }
    if (h[32]) { snprintf(msg, n, "%u wrong words", h[32]); return 1; }
    snprintf(msg, n, "all-gather of %d x 512 granules over %d workgroups: %.2f us per round (%d rounds)", nb, nwg, *us_per_round, rounds);
    return 0;
}

}  // namespace wrnn
