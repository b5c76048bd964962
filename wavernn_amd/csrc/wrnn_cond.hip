// wrnn_cond.hip -- hoisted conditioning projections (run once per generate call, before the loop).
//
// In the reference every step concatenates conditioning onto the recurrent input and multiplies the whole
// thing (fatchord_version.py:208-209, :213-214, :217-218, :220-221).  The conditioning columns of those four
// matrices do not depend on the recurrence, so they are computed here for all steps / frames:
//   cI [T][B][H]   = I.bias + I.weight[:,1:81] . m_t + I.weight[:,81:113] . a1_t         (per sample)
//   c2f[NF+1][3H]  = rnn2.bias_ih + rnn2.weight_ih[:,H:H+A] . a2                          (per FRAME)
//   c3f[NF+1][H]   = fc1.bias + fc1.weight[:,H:H+A] . a3 ;  c4f likewise with fc2 / a4
// aux is constant over a mel frame (Stretch2d, :57-61,:84), so c2f/c3f/c4f are frame tables; row NF is the
// zero-conditioning row the fold's zero padding selects (:326-330).  The fold itself (:336-338) is never
// materialised: segment b, step t reads position p = seg_pos[b] + t (see LoopArgs).
#include <stdlib.h>
#include <string.h>

#include "wrnn_device.h"

namespace wrnn {


constexpr int CR = 8;   // (t,b) rows per iteration

// thread r owns output row r of the I layer and keeps its 112 weights in VGPRs; the 112-long conditioning
// vectors of CR (t,b) rows are staged in LDS and broadcast-read (ds_read_b128).
__global__ __launch_bounds__(H) void wrnn_cond_sample_kernel(const CondArgs a)
{
    __shared__ __attribute__((aligned(16))) float in[CR][KCOND];
    const int r = threadIdx.x;
    float wreg[KCOND];
#pragma unroll
    for (int k = 0; k < KCOND; ++k) wreg[k] = a.I_cT[k * H + r];
    const float bias = a.I_b[r];
    const long rows = (long)a.T * a.B;
    for (long base = (long)blockIdx.x * CR; base < rows; base += (long)gridDim.x * CR) {
        __syncthreads();
        for (int q = r; q < CR * KCOND; q += H) {
            const int rr = q / KCOND, k = q % KCOND;
            const long row = base + rr;
            float val = 0.f;
            if (row < rows) {
                const int t = (int)(row / a.B), b = (int)(row % a.B);
                const int p = a.seg_pos[b] + t;
                if (p < a.seg_lim[b]) val = (k < MEL) ? a.mels_up[(size_t)p * MEL + k] : a.aux[(size_t)(p / a.hop) * 4 * AUX + (k - MEL)];
            }
            in[rr][k] = val;
        }
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < CR; ++rr) {
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < KCOND; k += 4) {
                const float4 x = *reinterpret_cast<const float4 *>(&in[rr][k]);
                acc = fmaf(wreg[k + 0], x.x, acc);
                acc = fmaf(wreg[k + 1], x.y, acc);
                acc = fmaf(wreg[k + 2], x.z, acc);
                acc = fmaf(wreg[k + 3], x.w, acc);
            }
            if (base + rr < rows) a.cI[(size_t)(base + rr) * H + r] = acc + bias;
        }
    }
}


// MFMA form of the same product: one workgroup keeps ALL of I.weight[:,1:] (512 x 112) in registers as A fragments
// (wave w owns output rows [128w, 128w+128) = 8 row tiles x 7 k-blocks x 4 = 224 registers) and grid-strides over tiles of
// 16 (t, segment) rows staged in LDS.  One accumulator chain per row tile in ascending k, bias added last -- the same
// order as wrnn_cond_sample_kernel's fmaf chain up to the rounding inside each 4-term MFMA step (measured: RAW class
// indices identical, MoL samples within 1e-5; tests/test_gpu_parity.py), at the MFMA rate.
constexpr int CKB = KCOND / 16;        // 7 k-blocks of 16
constexpr int CLD = KCOND + 4;         // LDS row stride (116 floats, 16-B aligned rows)
__global__ __launch_bounds__(256, 1) void wrnn_cond_mfma_kernel(const CondArgs a)
{
    __shared__ __attribute__((aligned(16))) float in[16 * CLD];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int fi = lane & 15, kq = lane >> 4;
    float A[8][4 * CKB];
    float4 bias[8];
#pragma unroll
    for (int tt = 0; tt < 8; ++tt) {
        const int row = 128 * w + 16 * tt + fi;
#pragma unroll
        for (int r = 0; r < CKB; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) A[tt][4 * r + e] = a.I_cT[(size_t)(16 * r + 4 * kq + e) * H + row];
        bias[tt] = *reinterpret_cast<const float4 *>(a.I_b + 128 * w + 16 * tt + 4 * kq);
    }
    const long rows = (long)a.T * a.B;
    for (long base = (long)blockIdx.x * 16; base < rows; base += (long)gridDim.x * 16) {
        __syncthreads();
        for (int q = tid; q < 16 * KCOND; q += 256) {
            const int rr = q / KCOND, k = q % KCOND;
            const long row = base + rr;
            float val = 0.f;
            if (row < rows) {
                const int t = (int)(row / a.B), b = (int)(row % a.B);
                const int p = a.seg_pos[b] + t;
                if (p < a.seg_lim[b]) val = (k < MEL) ? a.mels_up[(size_t)p * MEL + k] : a.aux[(size_t)(p / a.hop) * 4 * AUX + (k - MEL)];
            }
            in[rr * CLD + k] = val;
        }
        __syncthreads();
        float4 bf[CKB];
#pragma unroll
        for (int r = 0; r < CKB; ++r) bf[r] = *reinterpret_cast<const float4 *>(in + fi * CLD + 16 * r + 4 * kq);
        f32x4 acc[8];
#pragma unroll
        for (int tt = 0; tt < 8; ++tt) acc[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < CKB; ++r) {
#pragma unroll
            for (int tt = 0; tt < 8; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[tt][4 * r + 0], bf[r].x, acc[tt], 0, 0, 0);
#pragma unroll
            for (int tt = 0; tt < 8; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[tt][4 * r + 1], bf[r].y, acc[tt], 0, 0, 0);
#pragma unroll
            for (int tt = 0; tt < 8; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[tt][4 * r + 2], bf[r].z, acc[tt], 0, 0, 0);
#pragma unroll
            for (int tt = 0; tt < 8; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[tt][4 * r + 3], bf[r].w, acc[tt], 0, 0, 0);
        }
        if (base + fi < rows) {
#pragma unroll
            for (int tt = 0; tt < 8; ++tt) {
                const float4 o = make_float4(acc[tt][0] + bias[tt].x, acc[tt][1] + bias[tt].y, acc[tt][2] + bias[tt].z, acc[tt][3] + bias[tt].w);
                *reinterpret_cast<float4 *>(a.cI + (size_t)(base + fi) * H + 128 * w + 16 * tt + 4 * kq) = o;
            }
        }
    }
}

// The same product for the role-split loop kernel (wrnn_loop.hip): one SLAB of steps [t0, t1) of one ROUND of segments
// [rb0, rb0 + B) cut into NG groups, written in the loop kernel's MFMA-fragment order: tile (t, g) = 16 segments x 512 columns
// as [wave][k-block][lane][4] -- which is exactly what lane `lane` of wave w holds in acc[tt] (4 consecutive columns
// 128 w + 16 tt + 4 kq .. of segment fi), so every store instruction writes 1 KB contiguously.  Same arithmetic and order
// as wrnn_cond_mfma_kernel (bit-identical values); rows of segments beyond a group's count get zero conditioning.
__global__ __launch_bounds__(256, 1) void wrnn_cond_frag_kernel(const CondArgs a)
{
    __shared__ __attribute__((aligned(16))) float in[16 * CLD];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int fi = lane & 15, kq = lane >> 4;
    float A[8][4 * CKB];
    float4 bias[8];
#pragma unroll
    for (int tt = 0; tt < 8; ++tt) {
        const int row = 128 * w + 16 * tt + fi;
#pragma unroll
        for (int r = 0; r < CKB; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) A[tt][4 * r + e] = a.I_cT[(size_t)(16 * r + 4 * kq + e) * H + row];
        bias[tt] = *reinterpret_cast<const float4 *>(a.I_b + 128 * w + 16 * tt + 4 * kq);
    }
    const long tiles = (long)(a.t1 - a.t0) * a.NG;
    // this thread's 7 input values of a tile ([segment rr][k], q = tid + 256 i): gathered one tile AHEAD, so the two dependent global
    // loads (segment table -> conditioning row) fly under the previous tile's 224 MFMAs instead of in front of them (round 3: the
    // kernel ran 105 us per slab at one wave per SIMD, a third of it MFMA)
    constexpr int NQ = (16 * KCOND + 255) / 256;
    auto gather = [&](long tile, float (&v)[NQ]) {
        const int t = a.t0 + (int)(tile / a.NG), g = (int)(tile % a.NG);
        const int b0 = a.rb0 + (int)(((long)g * a.B) / a.NG), nb = a.rb0 + (int)(((long)(g + 1) * a.B) / a.NG) - b0;
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int q = tid + 256 * i;
            const int rr = q / KCOND, k = q % KCOND;
            float val = 0.f;
            if (q < 16 * KCOND && rr < nb) {
                const int p = a.seg_pos[b0 + rr] + t;
                if (p < a.seg_lim[b0 + rr]) val = (k < MEL) ? a.mels_up[(size_t)p * MEL + k] : a.aux[(size_t)(p / a.hop) * 4 * AUX + (k - MEL)];
            }
            v[i] = val;
        }
    };
    float cur[NQ] = {}, nxt[NQ] = {};
    if (blockIdx.x < tiles) gather(blockIdx.x, cur);
    for (long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int q = tid + 256 * i;
            if (q < 16 * KCOND) in[(q / KCOND) * CLD + q % KCOND] = cur[i];
        }
        __syncthreads();
        if (tile + gridDim.x < tiles) gather(tile + gridDim.x, nxt);
        float4 bf[CKB];
#pragma unroll
        for (int r = 0; r < CKB; ++r) bf[r] = *reinterpret_cast<const float4 *>(in + fi * CLD + 16 * r + 4 * kq);
        f32x4 acc[8];
#pragma unroll
        for (int tt = 0; tt < 8; ++tt) acc[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < CKB; ++r) {
#pragma unroll
            for (int tt = 0; tt < 8; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[tt][4 * r + 0], bf[r].x, acc[tt], 0, 0, 0);
#pragma unroll
            for (int tt = 0; tt < 8; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[tt][4 * r + 1], bf[r].y, acc[tt], 0, 0, 0);
#pragma unroll
            for (int tt = 0; tt < 8; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[tt][4 * r + 2], bf[r].z, acc[tt], 0, 0, 0);
#pragma unroll
            for (int tt = 0; tt < 8; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[tt][4 * r + 3], bf[r].w, acc[tt], 0, 0, 0);
        }
        float4 *dst = reinterpret_cast<float4 *>(a.cI + (size_t)tile * (SEG * H)) + (w * 8) * 64 + lane;
#pragma unroll
        for (int tt = 0; tt < 8; ++tt)
            dst[tt * 64] = make_float4(acc[tt][0] + bias[tt].x, acc[tt][1] + bias[tt].y, acc[tt][2] + bias[tt].z, acc[tt][3] + bias[tt].w);
#pragma unroll
        for (int i = 0; i < NQ; ++i) cur[i] = nxt[i];
    }
}

// one block per frame (block NF = zero-conditioning row)
__global__ __launch_bounds__(H) void wrnn_cond_frame_kernel(const CondArgs a)
{
    __shared__ float ax[4 * AUX];
    const int f = blockIdx.x, r = threadIdx.x;
    if (r < 4 * AUX) ax[r] = (f < a.NF) ? a.aux[(size_t)f * 4 * AUX + r] : 0.f;
    __syncthreads();
    float g0 = 0.f, g1 = 0.f, g2 = 0.f, y3 = 0.f, y4 = 0.f;
#pragma unroll 8
    for (int k = 0; k < AUX; ++k) {
        g0 = fmaf(a.c2_wT[k * 3 * H + r], ax[AUX + k], g0);
        g1 = fmaf(a.c2_wT[k * 3 * H + H + r], ax[AUX + k], g1);
        g2 = fmaf(a.c2_wT[k * 3 * H + 2 * H + r], ax[AUX + k], g2);
        y3 = fmaf(a.c3_wT[k * H + r], ax[2 * AUX + k], y3);
        y4 = fmaf(a.c4_wT[k * H + r], ax[3 * AUX + k], y4);
    }
    a.c2f[(size_t)f * 3 * H + r] = g0 + a.b_ih2[r];
    a.c2f[(size_t)f * 3 * H + H + r] = g1 + a.b_ih2[H + r];
    a.c2f[(size_t)f * 3 * H + 2 * H + r] = g2 + a.b_ih2[2 * H + r];
    a.c3f[(size_t)f * H + r] = y3 + a.fc1_b[r];
    a.c4f[(size_t)f * H + r] = y4 + a.fc2_b[r];
}

// The same tables for ONE SLAB of steps [a.t0, a.t0 + slab) of a call, per SEGMENT (wrnn_duo.hip, SURVEY.md 8 row f1: nothing that grows
// with the call's length or with the corpus' frame count): row b * a.FPS + j = frame (seg_pos[b] + a.t0) / hop + j of segment b,
// j < a.FPS = (slab - 1) / hop + 2 (every frame the segment can touch in the slab); the last row, a.B * a.FPS, is the zero-conditioning row.
__global__ __launch_bounds__(H) void wrnn_cond_frame_slab_kernel(const CondArgs a)
{
    __shared__ float ax[4 * AUX];
    const int row = blockIdx.x, r = threadIdx.x;
    int f = a.NF;
    if (row < a.B * a.FPS) {
        const int b = row / a.FPS, j = row % a.FPS;
        f = (a.seg_pos[b] + a.t0) / a.hop + j;
    }
    if (r < 4 * AUX) ax[r] = (f < a.NF) ? a.aux[(size_t)f * 4 * AUX + r] : 0.f;
    __syncthreads();
    float g0 = 0.f, g1 = 0.f, g2 = 0.f, y3 = 0.f, y4 = 0.f;
#pragma unroll 8
    for (int k = 0; k < AUX; ++k) {
        g0 = fmaf(a.c2_wT[k * 3 * H + r], ax[AUX + k], g0);
        g1 = fmaf(a.c2_wT[k * 3 * H + H + r], ax[AUX + k], g1);
        g2 = fmaf(a.c2_wT[k * 3 * H + 2 * H + r], ax[AUX + k], g2);
        y3 = fmaf(a.c3_wT[k * H + r], ax[2 * AUX + k], y3);
        y4 = fmaf(a.c4_wT[k * H + r], ax[3 * AUX + k], y4);
    }
    a.c2f[(size_t)row * 3 * H + r] = g0 + a.b_ih2[r];
    a.c2f[(size_t)row * 3 * H + H + r] = g1 + a.b_ih2[H + r];
    a.c2f[(size_t)row * 3 * H + 2 * H + r] = g2 + a.b_ih2[2 * H + r];
    a.c3f[(size_t)row * H + r] = y3 + a.fc1_b[r];
    a.c4f[(size_t)row * H + r] = y4 + a.fc2_b[r];
}

// MOL sampling noise -> the two derived variates the sampler needs (utils/distribution.py:106-108,118-121), once per
// launch instead of once per workgroup and step: mixture columns u -> log(-log u) (Gumbel), logistic column
// u -> log u - log(1-u).  Same device math functions and operation order as the in-loop form (mol_gumbel / mol_sample).
__global__ __launch_bounds__(256) void wrnn_noise_mol_kernel(const float *__restrict__ in, float *__restrict__ out, long n, int B)
{
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float u = in[i];
        const int col = (int)(i % (11L * B));
        out[i] = (col < 10 * B) ? logf(-logf(u)) : (logf(u) - logf(1.0f - u));
    }
}

hipError_t launch_noise_mol(const float *in, float *out, long n, int B, int n_cus, hipStream_t stream)
{
    long blocks = (n + 255) / 256;
    if (blocks > (long)n_cus * 8) blocks = (long)n_cus * 8;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(wrnn_noise_mol_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, in, out, n, B);
    return hipGetLastError();
}

// per-frame aux tables c2f / c3f / c4f (once per call)
hipError_t launch_cond_frames(const CondArgs &a, hipStream_t stream)
{
    hipLaunchKernelGGL(wrnn_cond_frame_kernel, dim3(a.NF + 1), dim3(H), 0, stream, a);
    return hipGetLastError();
}

// the per-segment tables of one slab starting at a.t0 (all a.B segments of the call; a.FPS rows each + the zero row)
hipError_t launch_cond_frames_slab(const CondArgs &a, hipStream_t stream)
{
    hipLaunchKernelGGL(wrnn_cond_frame_slab_kernel, dim3(a.B * a.FPS + 1), dim3(H), 0, stream, a);
    return hipGetLastError();
}

// up to 64 floats handed over BY VALUE (kernel arguments are copied when the launch is enqueued): for small host arrays that live on
// the caller's stack -- an asynchronous copy from them would rely on the runtime staging pageable sources before it returns
struct PutFloats { float v[64]; };
__global__ void wrnn_put_floats_kernel(float *dst, PutFloats src, int n)
{
    if ((int)threadIdx.x < n) dst[threadIdx.x] = src.v[threadIdx.x];
}
hipError_t launch_put_floats(float *dst, const float *src, int n, hipStream_t stream)
{
    if (n < 1 || n > 64) return hipErrorInvalidValue;
    PutFloats p;
    for (int i = 0; i < 64; ++i) p.v[i] = i < n ? src[i] : 0.f;
    hipLaunchKernelGGL(wrnn_put_floats_kernel, dim3(1), dim3(64), 0, stream, dst, p, n);
    return hipGetLastError();
}

// one slab [a.t0, a.t1) of one round (a.rb0, a.B segments, a.NG groups) of cI in fragment order -> a.cI
hipError_t launch_cond_frag(const CondArgs &a, int n_cus, hipStream_t stream)
{
    long blocks = (long)(a.t1 - a.t0) * a.NG;
    if (blocks > n_cus) blocks = n_cus;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(wrnn_cond_frag_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}

// whole-T cI in [t][segment][H] order (stream / block-sparse kernels); valu = the VALU form, kept as the on-GPU cross-check
hipError_t launch_cond(const CondArgs &a, int n_cus, bool valu, hipStream_t stream)
{
    const long rows = (long)a.T * a.B;
    if (valu) {
        long blocks = (rows + CR - 1) / CR;
        const long cap = (long)n_cus * 4;
        if (blocks > cap) blocks = cap;
        if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(wrnn_cond_sample_kernel, dim3((unsigned)blocks), dim3(H), 0, stream, a);
    } else {
        long blocks = (rows + 15) / 16;
        if (blocks > n_cus) blocks = n_cus;
        if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(wrnn_cond_mfma_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    }
    return hipGetLastError();
}

}  // namespace wrnn
