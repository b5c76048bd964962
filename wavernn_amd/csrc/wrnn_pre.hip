// wrnn_pre.hip -- the pre-loop stage of WaveRNN.generate() on MI355X (gfx950): `UpsampleNetwork.forward`
// (reference models/fatchord_version.py:82-89) = MelResNet (:31-48, ResBlock :13-28) on f32 MFMA + the three
// Stretch2d/box-filter up-sampling stages (:51-61, :73-80, :86-88).  Runs once per utterance before the loop; the
// x275 Stretch2d repeat of aux (:83-85) is never materialised (the loop reads aux per FRAME).
//
// K-pre-1  wrnn_resnet_kernel: one workgroup per tile of 16 frames carries the tile through ALL 22 layers in LDS
//          ([frame][channel] rows = the MFMA B operand).  conv_in (k = 5, no padding: :35) is an im2col GEMM with
//          K = 80*5 = 400; every ResBlock layer is a 128x128 GEMM; eval-mode BatchNorm is a per-row scale/shift folded
//          on the host (y = x*s + t, s = w/sqrt(var+eps), t = b - mean*s).  v_mfma_f32_16x16x4_f32, wave w owns output
//          rows [32w, 32w+32) with the full K, so there is no cross-wave reduction; weights stream from L2 per layer.
// K-pre-2  wrnn_upstage_kernel x 3: Stretch2d(s,1) + Conv2d(1,1,(1,2s+1), padding (0,s)) per mel channel, the last
//          stage writes [sample][channel] and applies the `indent` crop (:88).
// K-pre-1g wrnn_resnet_generic_kernel (round 5): the same network for ANY feat_dims / compute_dims / res_out_dims / pad / res_blocks the
//          reference's constructor takes (:64-71; hparams.py:38-44 are only defaults) -- a tile of frames per workgroup in LDS, one output
//          per thread and trip, k ascending in one fmaf chain; the up-sampling stages take the channel count at run time (F = 0).  The
//          shipped dims keep the MFMA kernel.
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <vector>

#include "wrnn_device.h"
#include "../../include/wavernn_amd.h"

namespace wrnn {

constexpr int PC = 128;           // compute_dims == res_out_dims of the MFMA kernel (other dims: wrnn_resnet_generic_kernel)
constexpr int PFEAT = MEL;        // 80
constexpr int PK = 5;             // conv_in taps = 2*pad + 1
constexpr int PPAD = 2;
constexpr int PKIN = PFEAT * PK;  // 400
constexpr int PLDI = 516;         // LDS stride of the im2col tile (K padded to 512)
constexpr int PLDX = 132;         // LDS stride of the activation tiles
constexpr int PNF = 16;           // frames per workgroup (MFMA N)

struct PreArgs {
    const float *mel;             // [PFEAT][N]
    const float *conv_in_w;       // [PC][PKIN]
    const float *bn_in;           // [2][PC] scale, shift
    const float *res_w;           // [blocks][2][PC][PC]
    const float *res_bn;          // [blocks][2][2][PC] scale, shift
    const float *conv_out_w;      // [PC][PC]
    const float *conv_out_b;      // [PC]
    float *aux;                   // [N][PC]
    int N, blocks;
};

// D tile (rows 16*tile .. +15 of the layer output, 16 frames) -> per-lane rows (lane>>4)*4 + v, frame lane&15
__device__ __forceinline__ f32x4 pre_tile(const float *W, int ld, int kvalid, int row0, int nchunks, const float *act, int lda,
                                          int lane)
{
    const int fi = lane & 15, kq = lane >> 4;
    f32x4 total = {0.f, 0.f, 0.f, 0.f};
    for (int kc = 0; kc < nchunks; ++kc) {
        float a[AF];
#pragma unroll
        for (int r = 0; r < AF / 4; ++r) {
            const int col = kc * 128 + 16 * r + 4 * kq;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (col < kvalid) v = *reinterpret_cast<const float4 *>(W + (size_t)(row0 + fi) * ld + col);
            a[4 * r + 0] = v.x; a[4 * r + 1] = v.y; a[4 * r + 2] = v.z; a[4 * r + 3] = v.w;
        }
        const f32x4 part = mfma_tile(a, act + fi * lda + kc * 128 + 4 * kq);
        total = (kc == 0) ? part : total + part;
    }
    return total;
}

__global__ __launch_bounds__(256) void wrnn_resnet_kernel(const PreArgs a)
{
    __shared__ __attribute__((aligned(16))) float XIN[PNF * PLDI];
    __shared__ __attribute__((aligned(16))) float XA[PNF * PLDX];
    __shared__ __attribute__((aligned(16))) float XB[PNF * PLDX];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int j0 = blockIdx.x * PNF;                       // first output frame of this tile
    const int N = a.N;
    // im2col of the zero-padded mel (:185 pads `pad` frames each side): XIN[j][c*5 + t] = m[c][j0 + j + t - pad]
    for (int q = tid; q < PNF * PLDI; q += 256) {
        const int j = q / PLDI, k = q % PLDI;
        float v = 0.f;
        if (k < PKIN) {
            const int c = k / PK, t = k % PK;
            const int f = j0 + j + t - PPAD;
            if (f >= 0 && f < N && j0 + j < N) v = a.mel[(size_t)c * N + f];
        }
        XIN[q] = v;
    }
    __syncthreads();
    const int fi = lane & 15, rq = (lane >> 4) * 4;
    // ---- conv_in -> BN -> ReLU (:43-44) ---------------------------------------------------------------
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        const int row0 = 32 * w + 16 * tt;
        const f32x4 d = pre_tile(a.conv_in_w, PKIN, PKIN, row0, 4, XIN, PLDI, lane);
        const float4 s = *reinterpret_cast<const float4 *>(a.bn_in + row0 + rq);
        const float4 t = *reinterpret_cast<const float4 *>(a.bn_in + PC + row0 + rq);
        float4 y;
        y.x = fmaxf(fmaf(d[0], s.x, t.x), 0.f); y.y = fmaxf(fmaf(d[1], s.y, t.y), 0.f);
        y.z = fmaxf(fmaf(d[2], s.z, t.z), 0.f); y.w = fmaxf(fmaf(d[3], s.w, t.w), 0.f);
        *reinterpret_cast<float4 *>(XA + fi * PLDX + row0 + rq) = y;
    }
    __syncthreads();
    // ---- residual blocks (:21-28): x + BN2(conv2(relu(BN1(conv1(x))))) ---------------------------------
    for (int b = 0; b < a.blocks; ++b) {
        const float *W1 = a.res_w + (size_t)(2 * b) * PC * PC, *W2 = W1 + PC * PC;
        const float *bn1 = a.res_bn + (size_t)(2 * b) * 2 * PC, *bn2 = bn1 + 2 * PC;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int row0 = 32 * w + 16 * tt;
            const f32x4 d = pre_tile(W1, PC, PC, row0, 1, XA, PLDX, lane);
            const float4 s = *reinterpret_cast<const float4 *>(bn1 + row0 + rq);
            const float4 t = *reinterpret_cast<const float4 *>(bn1 + PC + row0 + rq);
            float4 y;
            y.x = fmaxf(fmaf(d[0], s.x, t.x), 0.f); y.y = fmaxf(fmaf(d[1], s.y, t.y), 0.f);
            y.z = fmaxf(fmaf(d[2], s.z, t.z), 0.f); y.w = fmaxf(fmaf(d[3], s.w, t.w), 0.f);
            *reinterpret_cast<float4 *>(XB + fi * PLDX + row0 + rq) = y;
        }
        __syncthreads();
        f32x4 d2[2];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) d2[tt] = pre_tile(W2, PC, PC, 32 * w + 16 * tt, 1, XB, PLDX, lane);
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {                                     // XA is read only as this lane's own residual
            const int row0 = 32 * w + 16 * tt;
            const float4 s = *reinterpret_cast<const float4 *>(bn2 + row0 + rq);
            const float4 t = *reinterpret_cast<const float4 *>(bn2 + PC + row0 + rq);
            float4 r = *reinterpret_cast<float4 *>(XA + fi * PLDX + row0 + rq);
            r.x += fmaf(d2[tt][0], s.x, t.x); r.y += fmaf(d2[tt][1], s.y, t.y);
            r.z += fmaf(d2[tt][2], s.z, t.z); r.w += fmaf(d2[tt][3], s.w, t.w);
            *reinterpret_cast<float4 *>(XA + fi * PLDX + row0 + rq) = r;
        }
        __syncthreads();
    }
    // ---- conv_out + bias (:47) -> aux[frame][channel] ---------------------------------------------------
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        const int row0 = 32 * w + 16 * tt;
        const f32x4 d = pre_tile(a.conv_out_w, PC, PC, row0, 1, XA, PLDX, lane);
        const float4 bo = *reinterpret_cast<const float4 *>(a.conv_out_b + row0 + rq);
        if (j0 + fi < N)
            *reinterpret_cast<float4 *>(a.aux + (size_t)(j0 + fi) * PC + row0 + rq) = make_float4(d[0] + bo.x, d[1] + bo.y, d[2] + bo.z, d[3] + bo.w);
    }
}

// MelResNet for any dims (see the header).  Dynamic LDS: MW[feat][gf + 2 pad] (the zero-padded mel window of the tile: :185 pads `pad` frames
// each side, conv_in has no padding of its own, :35), XA[gf][C], XB[gf][C].  Thread -> (row c, frame j) with j fastest: a wave reads one or a
// few weight rows (broadcast) and consecutive LDS words.
struct GenPreArgs {
    PreArgs p;
    int feat, C, R, pad, gf;
};
__global__ __launch_bounds__(256) void wrnn_resnet_generic_kernel(const GenPreArgs g)
{
    extern __shared__ __attribute__((aligned(16))) float gsm[];
    const PreArgs &a = g.p;
    const int feat = g.feat, C = g.C, R = g.R, pad = g.pad, GF = g.gf, K = 2 * pad + 1, WW = GF + 2 * pad;
    float *MW = gsm, *XA = MW + feat * WW, *XB = XA + GF * C;
    const int tid = threadIdx.x, N = a.N;
    const int j0 = blockIdx.x * GF;
    for (int q = tid; q < feat * WW; q += 256) {
        const int c = q / WW, f = j0 + q % WW - pad;
        MW[q] = (f >= 0 && f < N) ? a.mel[(size_t)c * N + f] : 0.f;
    }
    __syncthreads();
    // conv_in -> BN -> ReLU (:43-44): weight [C][feat][K]
    for (int o = tid; o < C * GF; o += 256) {
        const int c = o / GF, j = o % GF;
        const float *wr = a.conv_in_w + (size_t)c * feat * K;
        float acc = 0.f;
        for (int ci = 0; ci < feat; ++ci)
            for (int t = 0; t < K; ++t) acc = fmaf(wr[ci * K + t], MW[ci * WW + j + t], acc);
        XA[j * C + c] = fmaxf(fmaf(acc, a.bn_in[c], a.bn_in[C + c]), 0.f);
    }
    __syncthreads();
    for (int b = 0; b < a.blocks; ++b) {                  // residual blocks (:21-28)
        const float *W1 = a.res_w + (size_t)(2 * b) * C * C, *W2 = W1 + (size_t)C * C;
        const float *bn1 = a.res_bn + (size_t)(2 * b) * 2 * C, *bn2 = bn1 + 2 * C;
        for (int o = tid; o < C * GF; o += 256) {
            const int c = o / GF, j = o % GF;
            float acc = 0.f;
            for (int k = 0; k < C; ++k) acc = fmaf(W1[(size_t)c * C + k], XA[j * C + k], acc);
            XB[j * C + c] = fmaxf(fmaf(acc, bn1[c], bn1[C + c]), 0.f);
        }
        __syncthreads();
        for (int o = tid; o < C * GF; o += 256) {         // XA[j][c] is read and written by this thread only
            const int c = o / GF, j = o % GF;
            float acc = 0.f;
            for (int k = 0; k < C; ++k) acc = fmaf(W2[(size_t)c * C + k], XB[j * C + k], acc);
            XA[j * C + c] += fmaf(acc, bn2[c], bn2[C + c]);
        }
        __syncthreads();
    }
    for (int o = tid; o < R * GF; o += 256) {             // conv_out + bias (:47) -> aux[frame][channel]
        const int r = o / GF, j = o % GF;
        float acc = 0.f;
        for (int k = 0; k < C; ++k) acc = fmaf(a.conv_out_w[(size_t)r * C + k], XA[j * C + k], acc);
        if (j0 + j < N) a.aux[(size_t)(j0 + j) * R + r] = acc + a.conv_out_b[r];
    }
}

// One Stretch2d(s,1) + Conv2d(1,1,(1,2s+1), padding (0,s), bias=False) stage (:73-80, :86-87) on [PFEAT][n_in] rows.
//   out[c][q] = sum_j w[j] * rep(q + j - s),  rep(u) = in[c][u / s] for 0 <= u < n_in*s, else 0 (the conv's zero padding)
// FIRST: `in` is the un-padded mel [PFEAT][n_in - 2*pad]; frames outside it are the zero padding of :185.
// LAST:  writes out_t[q - indent][c] for indent <= q < n_out - indent (the crop of :88, transposed to [sample][channel]).
// S > 0: the stretch factor as a compile-time constant (the divisions become multiplies); S = 0: runtime `s_rt`.  F: the channel count
// (PFEAT), F = 0: runtime `feat_rt`.
template <bool FIRST, bool LAST, int S, int F = PFEAT>
__global__ __launch_bounds__(256) void wrnn_upstage_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                           const float *__restrict__ taps, int n_in, int s_rt, int pad, int indent, int feat_rt = PFEAT)
{
    const int s = S > 0 ? S : s_rt;
    const int feat = F > 0 ? F : feat_rt;
    const long n_out = (long)n_in * s;
    const long total = LAST ? (n_out - 2L * indent) * feat : n_out * feat;
    const int ld_in = FIRST ? n_in - 2 * pad : n_in;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        int c;
        long q;
        if (LAST) { c = (int)(idx % feat); q = idx / feat + indent; }
        else { c = (int)(idx / n_out); q = idx % n_out; }
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j <= 2 * s; ++j) {
            const long u = q + j - s;
            float v = 0.f;
            if (u >= 0 && u < n_out) {
                int f = (int)(u / s);
                if (FIRST) { f -= pad; v = (f >= 0 && f < ld_in) ? in[(size_t)c * ld_in + f] : 0.f; }
                else v = in[(size_t)c * ld_in + f];
            }
            acc = fmaf(taps[j], v, acc);
        }
        if (LAST) out[(size_t)(q - indent) * feat + c] = acc;
        else out[(size_t)c * n_out + q] = acc;
    }
}

// The LAST stage through an LDS tile (round 3): wrnn_upstage_kernel<.., true, ..> maps consecutive threads to consecutive CHANNELS of
// one sample (the output is [sample][channel]), so each of its 2 s + 1 reads per output walks 80 different rows of the input --
// 0.38 ms per 641-frame utterance (150 GB/s) in profiles/r03k_kernel_stats.csv.  Here a workgroup computes a 64-sample x 80-channel
// tile with consecutive threads on consecutive SAMPLES of one channel (a wave touches ~7 input words), parks it in LDS and
// writes it out row by row, coalesced.  Same taps in the same order per output: bit-identical values.
template <int S>
__global__ __launch_bounds__(256) void wrnn_upstage_last_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                                const float *__restrict__ taps, int n_in, int s_rt, int indent)
{
    constexpr int TQ = 64;
    __shared__ float tile[TQ][PFEAT + 1];
    const int s = S > 0 ? S : s_rt;
    const long n_out = (long)n_in * s, nq = n_out - 2L * indent;
    for (long q0 = (long)blockIdx.x * TQ; q0 < nq; q0 += (long)gridDim.x * TQ) {
        for (int e = threadIdx.x; e < TQ * PFEAT; e += 256) {
            const int c = e / TQ, ql = e % TQ;
            const long q = q0 + ql + indent;
            float acc = 0.f;
            if (q0 + ql < nq) {
#pragma unroll
                for (int j = 0; j <= 2 * s; ++j) {
                    const long u = q + j - s;
                    float v = 0.f;
                    if (u >= 0 && u < n_out) v = in[(size_t)c * n_in + (int)(u / s)];
                    acc = fmaf(taps[j], v, acc);
                }
            }
            tile[ql][c] = acc;
        }
        __syncthreads();
        for (int e = threadIdx.x; e < TQ * PFEAT; e += 256) {
            const int ql = e / PFEAT, c = e % PFEAT;
            if (q0 + ql < nq) out[(size_t)(q0 + ql) * PFEAT + c] = tile[ql][c];
        }
        __syncthreads();
    }
}

}  // namespace wrnn

using namespace wrnn;

// ---------------------------------------------------------------------------------------------------------
// C ABI (declared in include/wavernn_amd.h)
// ---------------------------------------------------------------------------------------------------------
static thread_local char g_pre_err[384] = "";
extern "C" const char *wrnn_pre_last_error(void) { return g_pre_err; }
#define PRE_FAIL(code, ...)                                  \
    do {                                                     \
        snprintf(g_pre_err, sizeof g_pre_err, __VA_ARGS__);  \
        return code;                                         \
    } while (0)
#define PRE_HIP(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) PRE_FAIL(WRNN_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));    \
    } while (0)

struct wrnn_pre {
    int device, blocks, scales[3], total_scale;
    int feat, C, R, pad;            // feat_dims, compute_dims, res_out_dims, pad
    int gf;                         // 0: the shipped dims (MFMA kernel); else frames per workgroup of wrnn_resnet_generic_kernel
    float *dev;
    const float *conv_in_w, *bn_in, *res_w, *res_bn, *conv_out_w, *conv_out_b, *taps[3];
};

static size_t generic_lds_bytes(int feat, int C, int pad, int gf) { return ((size_t)feat * (gf + 2 * pad) + 2 * (size_t)gf * C) * sizeof(float); }

extern "C" int wrnn_pre_create(const wrnn_pre_weights *w, int device, wrnn_pre **out)
{
    if (!w || !out) PRE_FAIL(WRNN_ERR_ARG, "NULL argument");
    if (w->feat_dims < 1 || w->feat_dims > 4096 || w->compute_dims < 1 || w->compute_dims > 4096 || w->res_out_dims < 1 || w->res_out_dims > 4096 ||
        w->pad < 0 || w->pad > 64 || w->res_blocks < 0 || w->res_blocks > 64)
        PRE_FAIL(WRNN_ERR_ARG, "bad UpsampleNetwork dims: feat_dims %d, compute_dims %d, res_out_dims %d, pad %d, res_blocks %d", w->feat_dims,
                 w->compute_dims, w->res_out_dims, w->pad, w->res_blocks);
    for (int i = 0; i < 3; ++i)
        if (w->upsample_factors[i] < 1 || w->upsample_factors[i] > 64) PRE_FAIL(WRNN_ERR_ARG, "bad upsample factor %d", w->upsample_factors[i]);
    if (!w->conv_in_w || !w->bn_in || (w->res_blocks && (!w->res_w || !w->res_bn)) || !w->conv_out_w || !w->conv_out_b || !w->up_w)
        PRE_FAIL(WRNN_ERR_ARG, "NULL weight pointer");
    const int FE = w->feat_dims, C = w->compute_dims, R = w->res_out_dims, PD = w->pad, KT = 2 * PD + 1;
    const bool shipped_dims = FE == PFEAT && C == PC && R == PC && PD == PPAD;
    int gf = 0;
    if (!shipped_dims) {                                        // the largest frame tile whose LDS fits the default 64 KB
        for (gf = 16; gf >= 1 && generic_lds_bytes(FE, C, PD, gf) > 65536; gf >>= 1) {}
        if (gf < 1) PRE_FAIL(WRNN_ERR_ARG, "UpsampleNetwork too wide for the pre-loop kernel (feat_dims %d, compute_dims %d, pad %d)", FE, C, PD);
    }
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device >= n) PRE_FAIL(WRNN_ERR_NO_DEVICE, "no HIP device %d (count %d)", device, n);
    DeviceGuard dg(device);
    PRE_HIP(dg.err);
    const int B = w->res_blocks;
    const double eps = 1e-5;                                    // nn.BatchNorm1d default
    std::vector<float> h;
    auto put = [&](const float *src, size_t cnt) { size_t o = (h.size() + 63) / 64 * 64; h.resize(o + cnt); if (src) memcpy(h.data() + o, src, cnt * 4); return o; };
    // eval-mode BN folded to scale/shift in float32 arithmetic: s = w / sqrt(var + eps), t = b - mean * s
    auto fold = [&](const float *bn, size_t o) {               // bn: [4][C] weight, bias, running_mean, running_var
        for (int c = 0; c < C; ++c) {
            const float s = bn[c] / sqrtf(bn[3 * C + c] + (float)eps);
            h[o + c] = s;
            h[o + C + c] = bn[C + c] - bn[2 * C + c] * s;
        }
    };
    const size_t o_cin = put(w->conv_in_w, (size_t)C * FE * KT);
    const size_t o_bnin = put(nullptr, 2 * (size_t)C);
    fold(w->bn_in, o_bnin);
    const size_t o_resw = put(w->res_w, (size_t)B * 2 * C * C);
    const size_t o_resbn = put(nullptr, (size_t)B * 2 * 2 * C);
    for (int i = 0; i < 2 * B; ++i) fold(w->res_bn + (size_t)i * 4 * C, o_resbn + (size_t)i * 2 * C);
    const size_t o_cow = put(w->conv_out_w, (size_t)R * C), o_cob = put(w->conv_out_b, R);
    size_t o_t[3], toff = 0;
    for (int i = 0; i < 3; ++i) { o_t[i] = put(w->up_w + toff, 2 * w->upsample_factors[i] + 1); toff += 2 * w->upsample_factors[i] + 1; }
    put(nullptr, 64);
    wrnn_pre *p = new wrnn_pre();
    p->device = device; p->blocks = B; p->total_scale = 1;
    p->feat = FE; p->C = C; p->R = R; p->pad = PD; p->gf = gf;
    for (int i = 0; i < 3; ++i) { p->scales[i] = w->upsample_factors[i]; p->total_scale *= w->upsample_factors[i]; }
    hipError_t e = hipMalloc((void **)&p->dev, h.size() * 4);
    if (e != hipSuccess) { delete p; PRE_FAIL(WRNN_ERR_HIP, "hipMalloc failed: %s", hipGetErrorString(e)); }
    e = hipMemcpy(p->dev, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(p->dev); delete p; PRE_FAIL(WRNN_ERR_HIP, "hipMemcpy failed: %s", hipGetErrorString(e)); }
    p->conv_in_w = p->dev + o_cin; p->bn_in = p->dev + o_bnin; p->res_w = p->dev + o_resw; p->res_bn = p->dev + o_resbn;
    p->conv_out_w = p->dev + o_cow; p->conv_out_b = p->dev + o_cob;
    for (int i = 0; i < 3; ++i) p->taps[i] = p->dev + o_t[i];
    *out = p;
    return WRNN_OK;
}

extern "C" void wrnn_pre_destroy(wrnn_pre *p)
{
    if (!p) return;
    (void)hipFree(p->dev);
    delete p;
}

extern "C" int wrnn_pre_hop(const wrnn_pre *p) { return p ? p->total_scale : 0; }

extern "C" size_t wrnn_pre_workspace_bytes(const wrnn_pre *p, int32_t n_frames)
{
    if (!p || n_frames < 1) return 0;
    const size_t nf = (size_t)n_frames + 2 * p->pad;
    return (nf * p->scales[0] + nf * p->scales[0] * p->scales[1]) * p->feat * sizeof(float) + 512;
}

// rows_only: stop in front of the last stage and write its input as [row][channel] (wrnn_pre_upsample_rows)
static int pre_run(const wrnn_pre *p, const float *mel, int32_t n_frames, float *mels_up, float *aux,
                   void *workspace, size_t workspace_bytes, void *stream_, bool rows_only)
{
    if (!p || !mel || !mels_up || !aux || !workspace) PRE_FAIL(WRNN_ERR_ARG, "NULL argument");
    if (n_frames < 1 || (double)n_frames * p->total_scale > 2.0e9) PRE_FAIL(WRNN_ERR_ARG, "bad n_frames %d", n_frames);
    if (workspace_bytes < wrnn_pre_workspace_bytes(p, n_frames)) PRE_FAIL(WRNN_ERR_WORKSPACE, "workspace too small");
    if (((uintptr_t)workspace & 255) != 0) PRE_FAIL(WRNN_ERR_ARG, "workspace must be 256-byte aligned");
    hipStream_t stream = (hipStream_t)stream_;
    DeviceGuard dg(p->device);
    PRE_HIP(dg.err);
    PreArgs a;
    a.mel = mel; a.conv_in_w = p->conv_in_w; a.bn_in = p->bn_in; a.res_w = p->res_w; a.res_bn = p->res_bn;
    a.conv_out_w = p->conv_out_w; a.conv_out_b = p->conv_out_b; a.aux = aux; a.N = n_frames; a.blocks = p->blocks;
    const int FE = p->feat, PD = p->pad;
    if (p->gf == 0) hipLaunchKernelGGL(wrnn_resnet_kernel, dim3((n_frames + PNF - 1) / PNF), dim3(256), 0, stream, a);
    else {
        GenPreArgs g;
        g.p = a; g.feat = FE; g.C = p->C; g.R = p->R; g.pad = PD; g.gf = p->gf;
        hipLaunchKernelGGL(wrnn_resnet_generic_kernel, dim3((n_frames + p->gf - 1) / p->gf), dim3(256), generic_lds_bytes(FE, p->C, PD, p->gf), stream, g);
    }
    const int nf = n_frames + 2 * PD;
    float *s1 = (float *)workspace;                                     // [feat][nf*s0]
    float *s2 = s1 + (size_t)nf * p->scales[0] * FE;                    // [feat][nf*s0*s1]
    auto grid = [](long total) { long b = (total + 255) / 256; return (unsigned)(b > 65535 ? 65535 : (b < 1 ? 1 : b)); };
    const bool feat80 = FE == PFEAT;
    const bool shipped = feat80 && PD == PPAD && p->scales[0] == 5 && p->scales[1] == 5 && p->scales[2] == 11;       // hparams.py: voc_upsample_factors
    const unsigned g1 = grid((long)nf * p->scales[0] * FE), g2 = grid((long)nf * p->scales[0] * p->scales[1] * FE);
    const unsigned g3 = grid((long)n_frames * p->total_scale * 4);      // one 64-sample x 80-channel tile per workgroup and trip
    const int n2 = nf * p->scales[0], n3 = n2 * p->scales[1], indent = PD * p->total_scale;
    if (rows_only) {
        // (LAST = true with indent 0: no crop, the [sample][channel] transpose only)
        if (shipped) {
            hipLaunchKernelGGL((wrnn_upstage_kernel<true, false, 5>), dim3(g1), dim3(256), 0, stream, mel, s1, p->taps[0], nf, 5, PD, 0, PFEAT);
            hipLaunchKernelGGL((wrnn_upstage_kernel<false, true, 5>), dim3(g2), dim3(256), 0, stream, s1, mels_up, p->taps[1], n2, 5, 0, 0, PFEAT);
        } else {
            hipLaunchKernelGGL((wrnn_upstage_kernel<true, false, 0, 0>), dim3(g1), dim3(256), 0, stream, mel, s1, p->taps[0], nf, p->scales[0], PD, 0, FE);
            hipLaunchKernelGGL((wrnn_upstage_kernel<false, true, 0, 0>), dim3(g2), dim3(256), 0, stream, s1, mels_up, p->taps[1], n2, p->scales[1], 0, 0, FE);
        }
    } else if (shipped) {
        hipLaunchKernelGGL((wrnn_upstage_kernel<true, false, 5>), dim3(g1), dim3(256), 0, stream, mel, s1, p->taps[0], nf, 5, PD, 0, PFEAT);
        hipLaunchKernelGGL((wrnn_upstage_kernel<false, false, 5>), dim3(g2), dim3(256), 0, stream, s1, s2, p->taps[1], n2, 5, 0, 0, PFEAT);
        hipLaunchKernelGGL((wrnn_upstage_last_kernel<11>), dim3(g3), dim3(256), 0, stream, s2, mels_up, p->taps[2], n3, 11, indent);
    } else {
        hipLaunchKernelGGL((wrnn_upstage_kernel<true, false, 0, 0>), dim3(g1), dim3(256), 0, stream, mel, s1, p->taps[0], nf, p->scales[0], PD, 0, FE);
        hipLaunchKernelGGL((wrnn_upstage_kernel<false, false, 0, 0>), dim3(g2), dim3(256), 0, stream, s1, s2, p->taps[1], n2, p->scales[1], 0, 0, FE);
        // the last stage: the LDS-tile kernel is built for 80 channels; any other count: the plain stage kernel (LAST: crop + [sample][channel])
        if (feat80) hipLaunchKernelGGL((wrnn_upstage_last_kernel<0>), dim3(g3), dim3(256), 0, stream, s2, mels_up, p->taps[2], n3, p->scales[2], indent);
        else hipLaunchKernelGGL((wrnn_upstage_kernel<false, true, 0, 0>), dim3(grid((long)n_frames * p->total_scale * FE)), dim3(256), 0, stream, s2, mels_up, p->taps[2], n3, p->scales[2], 0, indent, FE);
    }
    PRE_HIP(hipGetLastError());
    return WRNN_OK;
}

extern "C" int wrnn_pre_upsample(const wrnn_pre *p, const float *mel, int32_t n_frames, float *mels_up, float *aux,
                                 void *workspace, size_t workspace_bytes, void *stream)
{
    return pre_run(p, mel, n_frames, mels_up, aux, workspace, workspace_bytes, stream, false);
}

extern "C" int wrnn_pre_upsample_rows(const wrnn_pre *p, const float *mel, int32_t n_frames, float *mel_rows, float *aux,
                                      void *workspace, size_t workspace_bytes, void *stream)
{
    return pre_run(p, mel, n_frames, mel_rows, aux, workspace, workspace_bytes, stream, true);
}
