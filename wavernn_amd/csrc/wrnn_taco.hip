// wrnn_taco.hip -- DRAFT (compiles; never run on a GPU): the Tacotron DECODER LOOP of BASELINE config 3 as ONE persistent
// cooperative kernel.
//
// Replaces the per-frame Python loop of `Tacotron.generate()` (reference models/tacotron.py:396-414) around `Decoder.forward`
// (:218-279: PreNet :141-155, attention GRUCell, LSA attention :181-207, rnn_input, two residual LSTMCells, mel_proj) for ONE
// sentence (batch 1).  The HIP-graph replay of wavernn_amd/tacotron.py spends 555 us per decoder step in ~60 tiny launches;
// here the whole loop is one launch:
//   * grid = up to 128 workgroups x 4 waves, cooperative (co-resident); N = 1, so every layer is a set of independent rows
//     (dot products): a WAVE owns a row (or a recurrent UNIT = its 6 / 8 gate rows), lanes split K with 16-byte loads, a DPP /
//     shuffle butterfly sums the 64 partials.  No MFMA: one column.
//   * weights (5.9 M f32 = 23.6 MB) are NOT staged: they are immutable, read with plain loads and stay in L2 / MALL between
//     steps (every row is read by exactly one wave per step).  [Register residency, as in the vocoder kernel, is the follow-up:
//     DESIGN.md section 9 item 3.]
//   * activations (<= 512 floats per layer) cross workgroups through a 36 KB workspace with device-coherent (sc1) stores / loads,
//     layers separated by a flag barrier: every workgroup drains its stores, writes its arrival word, polls all arrival words
//     (no atomics, bounded spins, failure code in the status words -- the conventions of wrnn_loop.hip).  Ten barriers per step.
//   * the recurrent vectors read AND written by one layer (attn_h, h1, h2) are double-buffered by step parity; c1 / c2 and the
//     cumulative attention have a single owner each.
//   * the stop test of :411 (`all mel values < stop_threshold and t > 10`) is evaluated in the kernel (per-workgroup counts of
//     values >= threshold, summed identically by every workgroup after the barrier), so the loop ends without a host round trip.
// Layer order inside a step and every formula follow wavernn_amd/tacotron.py::_decoder_step (the CPU mirror that is bit-exact
// with the reference); the summation ORDER differs (lanes split K), so parity is a tolerance, stated in the test.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "../../include/wavernn_amd.h"
#include "wrnn_device.h"

namespace wrnn {

constexpr int T_NM = 80;        // mel channels
constexpr int T_P1 = 256;       // prenet fc1
constexpr int T_P2 = 128;       // prenet fc2
constexpr int T_DD = 256;       // decoder dims (attention GRU hidden) == encoder sequence width (context)
constexpr int T_LD = 512;       // LSTM dims
constexpr int T_AF = 32;        // attention location filters
constexpr int T_AK = 31;        // ... their taps
constexpr int T_NMAX = 1024;    // encoder positions the workspace is laid out for
constexpr int T_MAXWG = 128;

// workspace (floats)
constexpr int A_PRE_IN = 0, A_PRE1 = 128, A_PRE2 = 384, A_ATTN_H = 512 /* [2][256] */, A_CTX = 1024, A_PQ = 1280,
              A_S = 1536 /* [NMAX] */, A_CUM = 2560, A_ATT = 3584, A_X = 4608, A_X2 = 5120, A_X3 = 5632,
              A_H1 = 6144 /* [2][512] */, A_H2 = 7168, A_C1 = 8192, A_C2 = 8704, A_END = 9216;
// then (unsigned) [T_MAXWG] arrival words, [T_MAXWG] not-below-threshold counts, [8] status
constexpr int U_FLAG = 0, U_CNT = T_MAXWG, U_STATUS = 2 * T_MAXWG, U_END = 2 * T_MAXWG + 8;

struct TacoArgs {
    wrnn_taco_weights w;
    const float *seq, *seq_proj;          // [n][256]
    float *act;                           // workspace floats [A_END]
    unsigned *uw;                         // workspace words [U_END]
    float *mel_out;                       // [max_steps][80][r]
    float *scores_out;                    // [max_steps][n]
    int *steps_done;
    int n, r, max_r, max_steps, nwg;
    float stop_threshold;
};

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

// dot(w[0..K), x[0..K)) with K a multiple of 4: lanes take float4 chunks round-robin; x in LDS.  Result in every lane.
__device__ __forceinline__ float row_dot(const float *__restrict__ wrow, const float *x, int K, int lane)
{
    float s = 0.f;
    for (int k = 4 * lane; k < K; k += 256) {
        const float4 a = *reinterpret_cast<const float4 *>(wrow + k);
        const float4 b = *reinterpret_cast<const float4 *>(x + k);
        s = fmaf(a.x, b.x, s); s = fmaf(a.y, b.y, s); s = fmaf(a.z, b.z, s); s = fmaf(a.w, b.w, s);
    }
    return wave_sum(s);
}

__global__ __launch_bounds__(NT, 1) void wrnn_taco_decoder_kernel(const TacoArgs a)
{
    __shared__ __attribute__((aligned(16))) float xin[1024];          // the layer's input vector(s)
    __shared__ __attribute__((aligned(16))) float sc[T_NMAX];         // attention scores of the step
    __shared__ __attribute__((aligned(16))) float convw[T_AF * 2 * T_AK];
    __shared__ __attribute__((aligned(16))) float Lw[T_DD * T_AF];
    __shared__ float wtmp[NW][64 + T_AF];                              // per wave: location window (62) + filter outputs (32)
    __shared__ float red[NT];
    __shared__ int misc[4];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wg = blockIdx.x, nwg = a.nwg;
    const int gw = wg * NW + w, NWV = nwg * NW;                        // global wave id / count
    const int gt = wg * NT + tid, NGT = nwg * NT;                      // global thread id / count
    const int n = a.n, r = a.r;
    const __amdgpu_buffer_rsrc_t ars = make_rsrc(a.act, (unsigned)(A_END * 4));
    const __amdgpu_buffer_rsrc_t urs = make_rsrc(a.uw, (unsigned)(U_END * 4));
    unsigned *status = a.uw + U_STATUS;
    unsigned phase = 0u;
    bool ok = true;

    auto ld = [&](int off) -> float { return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ars, off * 4, 0, 16 /* sc1 */)); };
    auto st = [&](int off, float v) { __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ars, off * 4, 0, 16 /* sc1 */); };
    // input vector(s) of a layer -> LDS (every thread one float per 256): `cnt` floats from workspace offset `off` to xin + dst
    auto stage = [&](int dst, int off, int cnt) {
        for (int k = tid; k < cnt; k += NT) xin[dst + k] = ld(off + k);
    };
    // flag barrier over all workgroups: everything stored before it (sc1) is visible to every workgroup after it
    auto barrier = [&](unsigned code) -> bool {
        ++phase;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // this thread's sc1 stores are acknowledged
        __syncthreads();
        if (w == 0) {
            if (lane == 0) __builtin_amdgcn_raw_buffer_store_b32(phase, urs, (U_FLAG + wg) * 4, 0, 16 /* sc1 */);
            unsigned spins = 0;
            bool good = true;
            for (;;) {
                bool all = true;
                for (int j = lane; j < nwg; j += 64)
                    all = all && (__builtin_amdgcn_raw_buffer_load_b32(urs, (U_FLAG + j) * 4, 0, 16 /* sc1 */) >= phase);
                if (__all(all)) break;
                if ((++spins & 255u) == 0u && (spins > SPIN_LIMIT || ld_agent32(status) != 0u)) { good = false; break; }
                __builtin_amdgcn_s_sleep(2);
            }
            if (!good && lane == 0) report_failure(status, 0x800u | code, wg, phase, tid);
            if (lane == 0) misc[0] = good ? 1 : 0;
        }
        __syncthreads();
        return misc[0] != 0;
    };
#define BAR(code)                       \
    do {                                \
        if (!barrier(code)) return;     \
    } while (0)

    // one-time: attention location weights -> LDS; zero state (the kernel owns the workspace from step 0)
    for (int k = tid; k < T_AF * 2 * T_AK; k += NT) convw[k] = a.w.attn_conv_w[k];
    for (int k = tid; k < T_DD * T_AF; k += NT) Lw[k] = a.w.attn_L_w[k];
    for (int k = gt; k < A_END; k += NGT) st(k, 0.f);
    BAR(0);

    int step = 0;
    for (; step < a.max_steps; ++step) {
        const int p = step & 1;                                         // recurrent double buffers: read [p], write [p ^ 1]
        // ---- L1 / L2: PreNet (:141-155, eval: no dropout) on the previous frame's last mel column (<GO> = zeros) ----------
        stage(0, A_PRE_IN, T_NM);
        __syncthreads();
        for (int row = gw; row < T_P1; row += NWV) {
            const float s = row_dot(a.w.prenet_fc1_w + (size_t)row * T_NM, xin, T_NM, lane) + a.w.prenet_fc1_b[row];
            if (lane == 0) st(A_PRE1 + row, fmaxf(s, 0.f));
        }
        BAR(1);
        stage(0, A_PRE1, T_P1);
        __syncthreads();
        for (int row = gw; row < T_P2; row += NWV) {
            const float s = row_dot(a.w.prenet_fc2_w + (size_t)row * T_P1, xin, T_P1, lane) + a.w.prenet_fc2_b[row];
            if (lane == 0) st(A_PRE2 + row, fmaxf(s, 0.f));
        }
        BAR(2);
        // ---- L3: attention GRUCell on [context, prenet] (:233-235; ATen gru_cell algebra) ------------------------------
        stage(0, A_CTX, T_DD);
        stage(T_DD, A_PRE2, T_P2);
        stage(512, A_ATTN_H + p * T_DD, T_DD);
        __syncthreads();
        for (int u = gw; u < T_DD; u += NWV) {
            const int KI = T_DD + T_P2;
            const float gir = row_dot(a.w.attn_rnn_w_ih + (size_t)u * KI, xin, KI, lane) + a.w.attn_rnn_b_ih[u];
            const float giz = row_dot(a.w.attn_rnn_w_ih + (size_t)(T_DD + u) * KI, xin, KI, lane) + a.w.attn_rnn_b_ih[T_DD + u];
            const float gin = row_dot(a.w.attn_rnn_w_ih + (size_t)(2 * T_DD + u) * KI, xin, KI, lane) + a.w.attn_rnn_b_ih[2 * T_DD + u];
            const float ghr = row_dot(a.w.attn_rnn_w_hh + (size_t)u * T_DD, xin + 512, T_DD, lane) + a.w.attn_rnn_b_hh[u];
            const float ghz = row_dot(a.w.attn_rnn_w_hh + (size_t)(T_DD + u) * T_DD, xin + 512, T_DD, lane) + a.w.attn_rnn_b_hh[T_DD + u];
            const float ghn = row_dot(a.w.attn_rnn_w_hh + (size_t)(2 * T_DD + u) * T_DD, xin + 512, T_DD, lane) + a.w.attn_rnn_b_hh[2 * T_DD + u];
            const float rg = sigm(gir + ghr), zg = sigm(giz + ghz);
            const float ng = tanhf(gin + rg * ghn);
            const float h = xin[512 + u];
            if (lane == 0) st(A_ATTN_H + (p ^ 1) * T_DD + u, (h - ng) * zg + ng);
        }
        BAR(3);
        // ---- L4: processed query W . attn_h + b (:193) -------------------------------------------------------------------
        stage(0, A_ATTN_H + (p ^ 1) * T_DD, T_DD);
        __syncthreads();
        for (int row = gw; row < T_DD; row += NWV) {
            const float s = row_dot(a.w.attn_W_w + (size_t)row * T_DD, xin, T_DD, lane) + a.w.attn_W_b[row];
            if (lane == 0) st(A_PQ + row, s);
        }
        BAR(4);
        // ---- L5: location-sensitive scores (:194-203): one wave per encoder position ----------------------------------------
        stage(0, A_PQ, T_DD);
        __syncthreads();
        for (int pos = gw; pos < n; pos += NWV) {
            float *win = wtmp[w], *cf = wtmp[w] + 64;
            if (lane < 2 * T_AK) {                                      // window of [cumulative, attention] around pos, zero padded
                const int c = lane / T_AK, k = lane % T_AK, idx = pos + k - T_AK / 2;
                win[lane] = (idx >= 0 && idx < n) ? ld((c == 0 ? A_CUM : A_ATT) + idx) : 0.f;
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane < T_AF) {                                          // conv1d (no bias): filter `lane`
                float s = 0.f;
                for (int q = 0; q < 2 * T_AK; ++q) s = fmaf(convw[lane * 2 * T_AK + q], win[q], s);
                cf[lane] = s;
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            float u = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int d = 4 * lane + e;
                float pl = a.w.attn_L_b[d];
                for (int f = 0; f < T_AF; ++f) pl = fmaf(Lw[d * T_AF + f], cf[f], pl);
                u = fmaf(a.w.attn_v_w[d], tanhf(xin[d] + a.seq_proj[(size_t)pos * T_DD + d] + pl), u);
            }
            u = wave_sum(u);
            if (lane == 0) st(A_S + pos, sigm(u));
            __builtin_amdgcn_wave_barrier();
        }
        BAR(5);
        // ---- L6: normalise (:204), update attention / cumulative (:205-206), context = scores . seq (:207) -----------------
        {
            float part = 0.f;
            for (int k = tid; k < n; k += NT) { sc[k] = ld(A_S + k); part += sc[k]; }
            red[tid] = part;
            __syncthreads();
            for (int m = NT / 2; m >= 1; m >>= 1) {                     // the same tree in every workgroup: the same total everywhere
                if (tid < m) red[tid] += red[tid + m];
                __syncthreads();
            }
            const float total = red[0];
            for (int k = tid; k < n; k += NT) sc[k] = sc[k] / total;
            __syncthreads();
            for (int pos = gt; pos < n; pos += NGT) {                   // owner of position pos (the same thread every step)
                st(A_ATT + pos, sc[pos]);
                st(A_CUM + pos, ld(A_CUM + pos) + sc[pos]);
                a.scores_out[(size_t)step * n + pos] = sc[pos];
            }
            for (int d = gw; d < T_DD; d += NWV) {
                float s = 0.f;
                for (int pos = lane; pos < n; pos += 64) s = fmaf(sc[pos], a.seq[(size_t)pos * T_DD + d], s);
                s = wave_sum(s);
                if (lane == 0) st(A_CTX + d, s);
            }
        }
        BAR(6);
        // ---- L7: rnn_input on [context, attn_h] (:246-247) ---------------------------------------------------------------
        stage(0, A_CTX, T_DD);
        stage(T_DD, A_ATTN_H + (p ^ 1) * T_DD, T_DD);
        __syncthreads();
        for (int row = gw; row < T_LD; row += NWV) {
            const float s = row_dot(a.w.rnn_input_w + (size_t)row * (2 * T_DD), xin, 2 * T_DD, lane) + a.w.rnn_input_b[row];
            if (lane == 0) st(A_X + row, s);
        }
        BAR(7);
        // ---- L8 / L9: the two residual LSTMCells (:249-259; ATen lstm_cell: gates i, f, g, o) -----------------------------------
#pragma unroll 1
        for (int L = 0; L < 2; ++L) {
            const float *wih = L == 0 ? a.w.rnn1_w_ih : a.w.rnn2_w_ih, *whh = L == 0 ? a.w.rnn1_w_hh : a.w.rnn2_w_hh;
            const float *bih = L == 0 ? a.w.rnn1_b_ih : a.w.rnn2_b_ih, *bhh = L == 0 ? a.w.rnn1_b_hh : a.w.rnn2_b_hh;
            const int AX = L == 0 ? A_X : A_X2, AH = L == 0 ? A_H1 : A_H2, AC = L == 0 ? A_C1 : A_C2, AO = L == 0 ? A_X2 : A_X3;
            stage(0, AX, T_LD);
            stage(T_LD, AH + p * T_LD, T_LD);
            __syncthreads();
            for (int u = gw; u < T_LD; u += NWV) {
                float g4[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const size_t row = (size_t)(q * T_LD + u);
                    g4[q] = row_dot(wih + row * T_LD, xin, T_LD, lane) + bih[row] + row_dot(whh + row * T_LD, xin + T_LD, T_LD, lane) + bhh[row];
                }
                if (lane == 0) {
                    const float c = sigm(g4[1]) * ld(AC + u) + sigm(g4[0]) * tanhf(g4[2]);
                    const float h = sigm(g4[3]) * tanhf(c);
                    st(AC + u, c);
                    st(AH + (p ^ 1) * T_LD + u, h);
                    st(AO + u, xin[u] + h);                             // residual (:251, :256)
                }
            }
            BAR(8 + L);
        }
        // ---- L10: mel_proj (:262-263): rows (m, j < r) of the (n_mels, max_r) view; the stop statistics ----------------------
        stage(0, A_X3, T_LD);
        if (tid == 0) misc[1] = 0;
        __syncthreads();
        {
            int notbelow = 0;
            for (int q = gw; q < T_NM * r; q += NWV) {
                const int m = q / r, j = q % r;
                const float s = row_dot(a.w.mel_proj_w + (size_t)(m * a.max_r + j) * T_LD, xin, T_LD, lane);
                if (lane == 0) {
                    a.mel_out[((size_t)step * T_NM + m) * r + j] = s;
                    if (j == r - 1) st(A_PRE_IN + m, s);                 // next step's prenet input (:412)
                    notbelow += !(s < a.stop_threshold);
                }
            }
            if (lane == 0 && notbelow) atomicAdd(&misc[1], notbelow);
            __syncthreads();
            if (tid == 0) __builtin_amdgcn_raw_buffer_store_b32((unsigned)misc[1], urs, (U_CNT + wg) * 4, 0, 16 /* sc1 */);
        }
        BAR(10);
        {   // :411  `if (mel_frames < stop_threshold).all() and t > 10: break`   (t = step * r)
            unsigned cnt = 0;
            for (int j = lane; j < nwg; j += 64) cnt += __builtin_amdgcn_raw_buffer_load_b32(urs, (U_CNT + j) * 4, 0, 16 /* sc1 */);
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) cnt += __shfl_xor(cnt, m, 64);
            if (cnt == 0u && step * r > 10) { ++step; break; }
        }
        // (the counts are rewritten only after the next step's ten barriers: no workgroup can still be reading them)
    }
    if (gt == 0) *a.steps_done = step;
#undef BAR
}

}  // namespace wrnn

using namespace wrnn;

static thread_local char g_taco_err[400] = "";
static void taco_err(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_taco_err, sizeof g_taco_err, fmt, ap);
    va_end(ap);
}

extern "C" const char *wrnn_taco_last_error(void) { return g_taco_err; }

extern "C" size_t wrnn_taco_workspace_bytes(void) { return (size_t)A_END * 4 + (size_t)U_END * 4; }

extern "C" int wrnn_taco_decode(int device, const wrnn_taco_weights *w, const wrnn_taco_call *c)
{
    if (!w || !c) { taco_err("null argument"); return WRNN_ERR_ARG; }
    if (c->struct_bytes != sizeof(wrnn_taco_call) || w->struct_bytes != sizeof(wrnn_taco_weights)) {
        taco_err("struct size mismatch (header / library versions differ)");
        return WRNN_ERR_ARG;
    }
    if (c->n < 1 || c->n > T_NMAX || c->r < 1 || c->r > c->max_r || c->max_steps < 1 || !c->seq || !c->seq_proj || !c->mel_out ||
        !c->scores_out || !c->steps_done || !c->workspace || c->workspace_bytes < wrnn_taco_workspace_bytes()) {
        taco_err("bad call: n=%d (1..%d) r=%d max_r=%d max_steps=%d or a null / short buffer", c->n, T_NMAX, c->r, c->max_r, c->max_steps);
        return WRNN_ERR_ARG;
    }
    if (w->n_mels != T_NM || w->prenet1 != T_P1 || w->prenet2 != T_P2 || w->decoder_dims != T_DD || w->encoder_width != T_DD ||
        w->lstm_dims != T_LD || w->attn_filters != T_AF || w->attn_kernel != T_AK) {
        taco_err("unsupported decoder geometry (the kernel is built for the reference's hparams: 80 / 256 / 128 / 256 / 512 / 32 x 31)");
        return WRNN_ERR_ARG;
    }
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) { taco_err("hipSetDevice: %s", hipGetErrorString(e)); return WRNN_ERR_HIP; }
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) { taco_err("hipGetDeviceProperties: %s", hipGetErrorString(e)); return WRNN_ERR_HIP; }
    TacoArgs a;
    a.w = *w;
    a.seq = c->seq; a.seq_proj = c->seq_proj;
    a.act = reinterpret_cast<float *>(c->workspace);
    a.uw = reinterpret_cast<unsigned *>(reinterpret_cast<char *>(c->workspace) + (size_t)A_END * 4);
    a.mel_out = c->mel_out; a.scores_out = c->scores_out; a.steps_done = c->steps_done;
    a.n = c->n; a.r = c->r; a.max_r = c->max_r; a.max_steps = c->max_steps;
    a.nwg = prop.multiProcessorCount < T_MAXWG ? prop.multiProcessorCount : T_MAXWG;
    a.stop_threshold = c->stop_threshold;
    hipStream_t stream = reinterpret_cast<hipStream_t>(c->stream);
    e = hipMemsetAsync(a.uw, 0, (size_t)U_END * 4, stream);
    if (e != hipSuccess) { taco_err("hipMemsetAsync: %s", hipGetErrorString(e)); return WRNN_ERR_HIP; }
    void *params[] = {(void *)&a};
    e = hipLaunchCooperativeKernel((const void *)wrnn_taco_decoder_kernel, dim3(a.nwg), dim3(NT), params, 0, stream);
    if (e != hipSuccess) { taco_err("cooperative launch of %d workgroups refused: %s", a.nwg, hipGetErrorString(e)); return WRNN_ERR_RESIDENCY; }
    return WRNN_OK;
}

// the status words of the last decode on this workspace (0 = clean; else code / workgroup / barrier of the first failure)
extern "C" int wrnn_taco_status(const void *workspace, unsigned *out4, void *stream)
{
    const char *p = reinterpret_cast<const char *>(workspace) + (size_t)A_END * 4 + (size_t)U_STATUS * 4;
    hipError_t e = hipMemcpyAsync(out4, p, 16, hipMemcpyDeviceToHost, reinterpret_cast<hipStream_t>(stream));
    if (e == hipSuccess) e = hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream));
    if (e != hipSuccess) { taco_err("status read: %s", hipGetErrorString(e)); return WRNN_ERR_HIP; }
    return WRNN_OK;
}
