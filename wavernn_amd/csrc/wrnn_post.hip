// wrnn_post.hip -- the post-loop stage of WaveRNN.generate() on the device, in the reference's float64 semantics:
// gather cast (models/fatchord_version.py:245), mu-law expansion (utils/dsp.py:98-103), equal-power cross-fade +
// overlap-add (`xfade_and_unfold`, fatchord_version.py:342-405), truncation and the linear tail fade (:255-258).
//
// Bit-exact by construction: every transcendental of the reference (pow in decode_mu_law, sqrt/linspace in the fades)
// is evaluated ON THE HOST with the reference's own numpy expressions and handed over as float64 tables; the device only
// gathers, multiplies and adds IEEE doubles.  That works because the RAW loop output lies on the 2^bits-level grid (one
// table entry per class) and the fades depend only on `overlap` / `hop`.
//   out[p] = ( y[i-1][o+stride] * fade_out[...]  +  y[i][o] * fade_in[o] ) * tail[...]     i = p / stride, o = p % stride
// in the reference's accumulation order (fold i-1 is added before fold i, :398-403; x + 0 and 0 + x are exact).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/wavernn_amd.h"

namespace wrnn {

struct PostArgs {
    const float *seg;        // [n_segments][T] loop output
    const int *first;        // [n_utt] first segment of every utterance
    const int *folds;        // [n_utt] segments of every utterance
    const long *out_off;     // [n_utt + 1] offset of every utterance's waveform in `out`
    const double *lut;       // [n_classes] decode_mu_law of every class value, or nullptr (MOL / mu_law off)
    const double *fade_in;   // [overlap]
    const double *fade_out;  // [overlap]
    const double *tail;      // [tail_len] linspace(1, 0, 20*hop)
    double *out;             // [sum wave_len]
    int n_utt, T, overlap, tail_len, n_classes, batched;
};

__device__ __forceinline__ double post_value(const PostArgs &a, int row, int o)
{
    const float y = a.seg[(size_t)row * a.T + o];
    double v;
    if (a.lut) {
        int idx = (int)rintf((y + 1.0f) * (0.5f * (float)(a.n_classes - 1)));     // y = 2*idx/(C-1) - 1 (fatchord_version.py:237)
        idx = idx < 0 ? 0 : (idx >= a.n_classes ? a.n_classes - 1 : idx);
        v = a.lut[idx];
    } else {
        v = (double)y;                                                            // .astype(np.float64), :245
    }
    if (a.batched) {
        if (o < a.overlap) v *= a.fade_in[o];                                      // :394
        if (o >= a.T - a.overlap) v *= a.fade_out[o - (a.T - a.overlap)];          // :395
    }
    return v;
}

// grid.y = utterance, grid-stride over its samples
__global__ __launch_bounds__(256) void wrnn_post_kernel(const PostArgs a)
{
    const int u = blockIdx.y;
    const long base = a.out_off[u];
    const long wave_len = a.out_off[u + 1] - base;
    const int f0 = a.first[u], nf = a.folds[u];
    const int stride = a.T - a.overlap;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < wave_len; p += (long)gridDim.x * blockDim.x) {
        double acc;
        if (a.batched) {
            const int i = (int)(p / stride), o = (int)(p % stride);
            acc = 0.0;
            if (i >= 1 && i - 1 < nf && o + stride < a.T) acc += post_value(a, f0 + i - 1, o + stride);   // earlier fold first
            if (i < nf) acc += post_value(a, f0 + i, o);
        } else {
            acc = post_value(a, f0, (int)p);
        }
        const long tl = p - (wave_len - a.tail_len);
        if (tl >= 0) acc *= a.tail[tl];                                            // :256-258
        a.out[base + p] = acc;
    }
}

}  // namespace wrnn

using namespace wrnn;

static thread_local char g_post_err[256] = "";
extern "C" const char *wrnn_post_last_error(void) { return g_post_err; }

extern "C" int wrnn_post_unfold(const float *segments, int32_t T, int32_t n_utt, const int32_t *first, const int32_t *folds,
                                const int64_t *out_off, const double *lut, int32_t n_classes, const double *fade_in,
                                const double *fade_out, int32_t overlap, const double *tail, int32_t tail_len, int32_t batched,
                                double *out, void *stream)
{
    if (!segments || !first || !folds || !out_off || !tail || !out || n_utt < 1 || T < 1 || tail_len < 0 ||
        (batched && (!fade_in || !fade_out || overlap < 0 || 2 * overlap > T)) || (lut && n_classes < 2)) {
        snprintf(g_post_err, sizeof g_post_err, "bad argument (n_utt=%d T=%d overlap=%d tail_len=%d)", n_utt, T, overlap, tail_len);
        return WRNN_ERR_ARG;
    }
    PostArgs a;
    a.seg = segments; a.first = first; a.folds = folds; a.out_off = (const long *)out_off; a.lut = lut; a.fade_in = fade_in;
    a.fade_out = fade_out; a.tail = tail; a.out = out; a.n_utt = n_utt; a.T = T; a.overlap = batched ? overlap : 0;
    a.tail_len = tail_len; a.n_classes = n_classes; a.batched = batched ? 1 : 0;
    hipLaunchKernelGGL(wrnn_post_kernel, dim3(128, n_utt), dim3(256), 0, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_post_err, sizeof g_post_err, "launch failed: %s", hipGetErrorString(e));
        return WRNN_ERR_HIP;
    }
    return WRNN_OK;
}
