// wrnn_taco.hip -- the Tacotron side of BASELINE config 3: the DECODER LOOP as ONE persistent
// cooperative kernel, in two forms -- wrnn_taco_decoder_kernel (flag barriers, weights from L2: any CU count; described here) and
// wrnn_taco_resident_kernel (register-resident weights, tagged exchange: what runs on an MI355X; further down) -- and the CBHGs'
// bidirectional GRU as a persistent kernel (wrnn_bigru_kernel).  Parity: tests/test_gpu_config3.py (<= 1e-8 against the CPU mirror
// of the reference on 200 frames).
//
// Replaces the per-frame Python loop of `Tacotron.generate()` (reference models/tacotron.py:396-414) around `Decoder.forward`
// (:218-279: PreNet :141-155, attention GRUCell, LSA attention :181-207, rnn_input, two residual LSTMCells, mel_proj) for ONE
// sentence (batch 1).  The HIP-graph replay of wavernn_amd/tacotron.py spends 555 us per decoder step in ~60 tiny launches;
// here the whole loop is one launch:
//   * grid = up to 128 workgroups x 4 waves, cooperative (co-resident); N = 1, so every layer is a set of independent rows
//     (dot products): a WAVE owns a row (or a recurrent UNIT = its 6 / 8 gate rows), lanes split K with 16-byte loads, a DPP /
//     shuffle butterfly sums the 64 partials.  No MFMA: one column.
//   * weights (5.9 M f32 = 23.6 MB) are NOT staged: they are immutable, read with plain loads and stay in L2 / MALL between
//     steps (every row is read by exactly one wave per step).  [Register residency: wrnn_taco_resident_kernel below.]
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
    unsigned long long *tv;               // resident kernel: tagged vectors [V_END] of 8 bytes
    float *mel_out;                       // [max_steps][80][r]
    float *scores_out;                    // [max_steps][n]
    int *steps_done;
    int n, r, max_r, max_steps, nwg;
    float stop_threshold;
};

__device__ __forceinline__ float wave_sum(float v) { return wave_sum64(v); }      // (DPP / permlane-swap butterflies, wrnn_device.h: the same tree as the __shfl_xor loop, without the LDS crossbar)
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


// =================================================================================================================================
// wrnn_taco_resident_kernel (round 3): the same decoder step with
//   * every weight REGISTER-RESIDENT: 128 workgroups x 4 waves = 512 waves, one LSTM unit / one rnn_input row per wave, one GRU
//     unit / query row / context dim / prenet row / mel row per wave of the first 256 / 128 / 80 r -- ~230 VGPRs of float4 weight
//     chunks per lane, loaded once (5.9 M f32 = 45 KB per wave).  No weight traffic per step (the flag-barrier form above re-reads
//     23.6 MB from L2 / MALL every step, behind each layer's input);
//   * NO flag barriers: a layer's output entries are 8-byte words {value, tag = step + 1} stored device-coherently (one b64 store:
//     atomic), two buffers per vector by step parity; the consumer workgroup polls the entries it stages into LDS until the tags
//     match.  One store -> load latency per layer instead of the barrier's three (drain, arrival word, poll of 128 words);
//   * the attention's location state (previous attention, cumulative attention) kept by EVERY workgroup in its own LDS: each
//     normalises the step's scores itself (same tree, same total), so nothing but the raw scores is exchanged;
//   * the stop test of :411 evaluated by every workgroup on the mel entries it stages for the next step's prenet anyway.
// WAR safety of the parity buffers: a vector written at step t is overwritten at step t + 2, i.e. after the whole chain of step
// t + 1 up to that layer -- which needs the LSTM units of EVERY wave (512 units = 512 waves), and every workgroup runs its layers
// in program order: all reads of step t + 1 before its own LSTM layers are behind it.  (Hence exactly 128 workgroups.)
// Summation order: lanes split K in float4 chunks as row_dot above, the 64 partials are summed by a DPP tree + 4 read-lanes
// (not the shuffle butterfly): results differ from the kernel above in the last bits; both are inside the test's tolerance.
// (Hardware exp / rcp in the gate functions were measured -- profiles/r03ac_taco_profile.json: 28.5 -> 26.6 us per step -- and
// NOT kept: the error against the CPU mirror of the reference grew from 5e-9 to 2e-5 over 200 frames for 0.7 % of config 3.)
// =================================================================================================================================
constexpr int R_NWG = 128, R_NWV = R_NWG * NW;
constexpr int R_MAXR = 8;
// (Measured and dropped in round 3, profiles/r03aj_taco_profile_local_prenet.json: both PreNet layers computed by every workgroup that owns
// attention-GRU units for itself -- two exchange hops less, but 80 KB of LDS and 128 registers per thread more: no faster.)
// tagged vectors: offsets in 8-byte entries, [2 parity buffers][length]
constexpr int VL_MEL = T_NM * R_MAXR, VL_S = T_NMAX;
constexpr int V_MEL = 0, V_PRE1 = V_MEL + 2 * VL_MEL, V_PRE2 = V_PRE1 + 2 * T_P1, V_ATTNH = V_PRE2 + 2 * T_P2, V_CTX = V_ATTNH + 2 * T_DD,
              V_PQ = V_CTX + 2 * T_DD, V_S = V_PQ + 2 * T_DD, V_X = V_S + 2 * VL_S, V_X2 = V_X + 2 * T_LD, V_X3 = V_X2 + 2 * T_LD,
              V_H1 = V_X3 + 2 * T_LD, V_H2 = V_H1 + 2 * T_LD, V_END = V_H2 + 2 * T_LD;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <int CTRL>
__device__ __forceinline__ float dpp_get(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float lane_get(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }
// sum over the 64 lanes, in every lane: quad xor 1, xor 2, half-row mirror, row mirror (16-lane row totals), then the four rows
__device__ __forceinline__ float wave_total(float v)
{
    v += dpp_get<0xB1>(v);
    v += dpp_get<0x4E>(v);
    v += dpp_get<0x141>(v);
    v += dpp_get<0x140>(v);
    return (lane_get(v, 0) + lane_get(v, 16)) + (lane_get(v, 32) + lane_get(v, 48));
}
__device__ __forceinline__ float4 ldw4(const float *w, bool on) { return on ? *reinterpret_cast<const float4 *>(w) : make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float fma4(const float4 a, const float *x, float s)
{
    const float4 b = *reinterpret_cast<const float4 *>(x);
    s = fmaf(a.x, b.x, s); s = fmaf(a.y, b.y, s); s = fmaf(a.z, b.z, s); s = fmaf(a.w, b.w, s);
    return s;
}

// PROF: workgroup 0, thread 0 accumulates shader clocks per layer, [2 L - 2] = until its input is staged, [2 L - 1] = compute + publish
// (L = 1 .. 10), [20] = steps; flushed to the 24 words after the tagged vectors (read by scripts/gpu_taco_profile.py)
template <bool PROF>
__global__ __launch_bounds__(NT, 1) void wrnn_taco_resident_kernel(const TacoArgs a)
{
    __shared__ unsigned long long profl[24];
    unsigned long long plast = 0;
#define PH(k)                                                          \
    do {                                                               \
        if (PROF && blockIdx.x == 0 && threadIdx.x == 0) {             \
            const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
            profl[k] += now_ - plast;                                  \
            plast = now_;                                              \
        }                                                              \
    } while (0)
    if (threadIdx.x < 24) profl[threadIdx.x] = 0;

    __shared__ __attribute__((aligned(16))) float xb[2][1024];        // a layer's input vector(s); layers alternate the two buffers
    __shared__ __attribute__((aligned(16))) float melv[VL_MEL];       // the previous step's mel block (stop test + prenet input)
    __shared__ __attribute__((aligned(16))) float ahv[T_DD];          // attn_h of this step (query layer and rnn_input)
    __shared__ __attribute__((aligned(16))) float sc[T_NMAX];         // scores of the step (raw, then normalised)
    __shared__ __attribute__((aligned(16))) float att[T_NMAX];        // this workgroup's copy of the previous attention ...
    __shared__ __attribute__((aligned(16))) float cum[T_NMAX];        // ... and of the cumulative attention (:205-206)
    __shared__ __attribute__((aligned(16))) float convT[2 * T_AK * T_AF];   // location conv weights, [tap of (channel, k)][filter]
    __shared__ __attribute__((aligned(16))) float LT[T_AF * T_DD];          // L weights, [filter][dim]
    __shared__ float wtmp[NW][64 + T_AF];
    __shared__ float red[NT];
    __shared__ int misc[4];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wg = blockIdx.x;
    const int gw = wg * NW + w;
    const int gt = wg * NT + tid, NGT = R_NWG * NT;
    const int n = a.n, r = a.r, nmel = T_NM * a.r;
    const __amdgpu_buffer_rsrc_t vrs = make_rsrc(a.tv, (unsigned)(V_END * 8));
    unsigned *status = a.uw + U_STATUS;
    bool ok = true;

    // ---------------- resident weights (float4 chunk c of a row: k = 4 lane + 256 c) ----------------
    const int k0 = 4 * lane;
    const bool u256 = gw < T_DD, u128 = gw < T_P2;
    const float4 w_fc1 = ldw4(a.w.prenet_fc1_w + (size_t)gw * T_NM + k0, u256 && k0 < T_NM);
    const float4 w_fc2 = ldw4(a.w.prenet_fc2_w + (size_t)gw * T_P1 + k0, u128);
    float4 g_i[3][2], g_h[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const size_t row = (size_t)(q * T_DD + gw);
        g_i[q][0] = ldw4(a.w.attn_rnn_w_ih + row * (T_DD + T_P2) + k0, u256);
        g_i[q][1] = ldw4(a.w.attn_rnn_w_ih + row * (T_DD + T_P2) + 256 + k0, u256 && k0 < T_P2);
        g_h[q] = ldw4(a.w.attn_rnn_w_hh + row * T_DD + k0, u256);
    }
    const float4 w_q = ldw4(a.w.attn_W_w + (size_t)gw * T_DD + k0, u256);
    float4 w_ri[2], l1i[4][2], l1h[4][2], l2i[4][2], l2h[4][2], w_mp[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        w_ri[c] = ldw4(a.w.rnn_input_w + (size_t)gw * (2 * T_DD) + 256 * c + k0, true);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const size_t row = (size_t)(q * T_LD + gw) * T_LD + 256 * c + k0;
            l1i[q][c] = ldw4(a.w.rnn1_w_ih + row, true); l1h[q][c] = ldw4(a.w.rnn1_w_hh + row, true);
            l2i[q][c] = ldw4(a.w.rnn2_w_ih + row, true); l2h[q][c] = ldw4(a.w.rnn2_w_hh + row, true);
        }
        w_mp[c] = ldw4(a.w.mel_proj_w + (size_t)((gw / r) * a.max_r + gw % r) * T_LD + 256 * c + k0, gw < nmel);
    }
    float seqc[16];                                                   // context dim gw: encoder_seq[pos = lane + 64 i][gw]
#pragma unroll
    for (int i = 0; i < 16; ++i) seqc[i] = (u256 && lane + 64 * i < n) ? a.seq[(size_t)(lane + 64 * i) * T_DD + gw] : 0.f;
    float4 spj[2];                                                    // encoder_seq_proj rows of the positions gw, gw + 512
#pragma unroll
    for (int i = 0; i < 2; ++i) spj[i] = ldw4(a.seq_proj + (size_t)(gw + R_NWV * i) * T_DD + k0, gw + R_NWV * i < n);
    const float4 v_w = ldw4(a.w.attn_v_w + k0, true), L_b = ldw4(a.w.attn_L_b + k0, true);
    // biases of this wave's rows (wave-uniform)
    const float b_fc1 = u256 ? a.w.prenet_fc1_b[gw] : 0.f, b_fc2 = u128 ? a.w.prenet_fc2_b[gw] : 0.f;
    const float b_q = u256 ? a.w.attn_W_b[gw] : 0.f;
    float b_gi[3], b_gh[3], b_l1[4], b_l1h[4], b_l2[4], b_l2h[4];
#pragma unroll
    for (int q = 0; q < 3; ++q) { b_gi[q] = u256 ? a.w.attn_rnn_b_ih[q * T_DD + gw] : 0.f; b_gh[q] = u256 ? a.w.attn_rnn_b_hh[q * T_DD + gw] : 0.f; }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        b_l1[q] = a.w.rnn1_b_ih[q * T_LD + gw]; b_l1h[q] = a.w.rnn1_b_hh[q * T_LD + gw];
        b_l2[q] = a.w.rnn2_b_ih[q * T_LD + gw]; b_l2h[q] = a.w.rnn2_b_hh[q * T_LD + gw];
    }
    const float b_ri = a.w.rnn_input_b[gw];

    // ---------------- LDS: attention weights (transposed: conflict-free lane strides), zero state ----------------
    for (int k = tid; k < T_AF * 2 * T_AK; k += NT) convT[(k % (2 * T_AK)) * T_AF + k / (2 * T_AK)] = a.w.attn_conv_w[k];
    for (int k = tid; k < T_DD * T_AF; k += NT) LT[(k % T_AF) * T_DD + k / T_AF] = a.w.attn_L_w[k];
    for (int k = tid; k < 1024; k += NT) { xb[0][k] = 0.f; xb[1][k] = 0.f; sc[k] = 0.f; att[k] = 0.f; cum[k] = 0.f; }
    for (int k = tid; k < VL_MEL; k += NT) melv[k] = 0.f;
    ahv[tid] = 0.f;
    if (tid < 4) misc[tid] = 0;
    __syncthreads();
    float c1 = 0.f, c2 = 0.f;                                          // this wave's LSTM cell states (uniform over the lanes)

    // entries [ventry, ventry + cnt) -> dst, each polled until it carries `tag`
    auto poll_in = [&](float *dst, int ventry, int cnt, unsigned tag, unsigned code) {
        for (int q0 = 0; q0 < cnt; q0 += NT) {
            const int q = q0 + tid;
            const bool live = q < cnt;
            u32x2 e = {0u, tag};
            if (live) e = __builtin_amdgcn_raw_buffer_load_b64(vrs, (ventry + q) * 8, 0, 16 /* sc1 */);
            unsigned spins = 0;
            while (__any(live && e.y != tag)) {
                if ((++spins & 255u) == 0u && (spins > SPIN_LIMIT || ld_agent32(status) != 0u)) {
                    if (lane == 0) report_failure(status, 0x900u | code, wg, tag, tid);
                    ok = false;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
                if (live) e = __builtin_amdgcn_raw_buffer_load_b64(vrs, (ventry + q) * 8, 0, 16 /* sc1 */);
            }
            if (live) dst[q] = __uint_as_float(e.x);
        }
    };
    auto pub = [&](int ventry, float v, unsigned tag) {
        if (lane == 0) {
            const u32x2 e = {__float_as_uint(v), tag};
            __builtin_amdgcn_raw_buffer_store_b64(e, vrs, ventry * 8, 0, 16 /* sc1 */);
        }
    };
#define STAGED()                            \
    do {                                    \
        if (!ok) misc[0] = 1;               \
        __syncthreads();                    \
        if (misc[0] != 0) return;           \
    } while (0)

    int step = 0;
    for (; step < a.max_steps; ++step) {
        const int p = step & 1;
        const unsigned tag = (unsigned)step + 1u, ptag = (unsigned)step;      // this step's entries / the previous step's
        // ---- the previous step's mel block: stop test of :411, then L1 PreNet fc1 on its last column (<GO> = zeros) -----------
        if (PROF && wg == 0 && tid == 0 && plast == 0) plast = __builtin_amdgcn_s_memtime();
        if (step > 0) poll_in(melv, V_MEL + (p ^ 1) * VL_MEL, nmel, ptag, 1);
        STAGED();
        PH(0);
        {
            int notbelow = 0;
            for (int q = tid; q < nmel; q += NT) notbelow |= !(melv[q] < a.stop_threshold);
            const int any = __syncthreads_or(notbelow);
            if (step > 0 && !any && (step - 1) * r > 10) break;        // `(mel_frames < stop_threshold).all() and t > 10`, t = (step - 1) r
        }
        if (u256) {
            float s = 0.f;
            if (k0 < T_NM) {                                           // the last frame of the block: column j = r - 1 of rows m = k0 .. k0 + 3
                s = fmaf(w_fc1.x, melv[k0 * r + r - 1], s);
                s = fmaf(w_fc1.y, melv[(k0 + 1) * r + r - 1], s);
                s = fmaf(w_fc1.z, melv[(k0 + 2) * r + r - 1], s);
                s = fmaf(w_fc1.w, melv[(k0 + 3) * r + r - 1], s);
            }
            pub(V_PRE1 + p * T_P1 + gw, fmaxf(wave_total(s) + b_fc1, 0.f), tag);
        }
        PH(1);
        // ---- L2: PreNet fc2 ----
        poll_in(xb[0], V_PRE1 + p * T_P1, T_P1, tag, 2);
        STAGED();
        PH(2);
        if (u128) pub(V_PRE2 + p * T_P2 + gw, fmaxf(wave_total(fma4(w_fc2, xb[0] + k0, 0.f)) + b_fc2, 0.f), tag);
        PH(3);
        // ---- L3: attention GRUCell on [context(t-1), prenet] with h = attn_h(t-1) (:233-235) ----
        if (step > 0) {
            poll_in(xb[1], V_CTX + (p ^ 1) * T_DD, T_DD, ptag, 3);
            poll_in(xb[1] + 512, V_ATTNH + (p ^ 1) * T_DD, T_DD, ptag, 3);
        }
        poll_in(xb[1] + T_DD, V_PRE2 + p * T_P2, T_P2, tag, 3);
        STAGED();
        PH(4);
        if (u256) {
            float gi[3], gh[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                float s = fma4(g_i[q][0], xb[1] + k0, 0.f);
                if (k0 < T_P2) s = fma4(g_i[q][1], xb[1] + 256 + k0, s);
                gi[q] = wave_total(s) + b_gi[q];
                gh[q] = wave_total(fma4(g_h[q], xb[1] + 512 + k0, 0.f)) + b_gh[q];
            }
            const float rg = sigm(gi[0] + gh[0]), zg = sigm(gi[1] + gh[1]);
            const float ng = tanhf(gi[2] + rg * gh[2]);
            const float h = xb[1][512 + gw];
            pub(V_ATTNH + p * T_DD + gw, (h - ng) * zg + ng, tag);
        }
        PH(5);
        // ---- L4: processed query (:193) ----
        poll_in(ahv, V_ATTNH + p * T_DD, T_DD, tag, 4);
        STAGED();
        PH(6);
        if (u256) pub(V_PQ + p * T_DD + gw, wave_total(fma4(w_q, ahv + k0, 0.f)) + b_q, tag);
        PH(7);
        // ---- L5: location-sensitive scores (:194-203), one wave per encoder position ----
        poll_in(xb[1], V_PQ + p * T_DD, T_DD, tag, 5);
        STAGED();
        PH(8);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int pos = gw + R_NWV * i;
            if (pos < n) {
                float *win = wtmp[w], *cf = wtmp[w] + 64;
                if (lane < 2 * T_AK) {
                    const int c = lane / T_AK, k = lane % T_AK, idx = pos + k - T_AK / 2;
                    win[lane] = (idx >= 0 && idx < n) ? (c == 0 ? cum[idx] : att[idx]) : 0.f;
                }
                __builtin_amdgcn_wave_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane < T_AF) {
                    float s = 0.f;
                    for (int q = 0; q < 2 * T_AK; ++q) s = fmaf(convT[q * T_AF + lane], win[q], s);
                    cf[lane] = s;
                }
                __builtin_amdgcn_wave_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                float pl0 = L_b.x, pl1 = L_b.y, pl2 = L_b.z, pl3 = L_b.w;
#pragma unroll 8
                for (int f = 0; f < T_AF; ++f) {
                    const float4 lw = *reinterpret_cast<const float4 *>(LT + f * T_DD + k0);
                    const float cv = cf[f];
                    pl0 = fmaf(lw.x, cv, pl0); pl1 = fmaf(lw.y, cv, pl1); pl2 = fmaf(lw.z, cv, pl2); pl3 = fmaf(lw.w, cv, pl3);
                }
                const float4 xq = *reinterpret_cast<const float4 *>(xb[1] + k0);
                const float4 sp = spj[i];
                float u = 0.f;
                u = fmaf(v_w.x, tanhf(xq.x + sp.x + pl0), u);
                u = fmaf(v_w.y, tanhf(xq.y + sp.y + pl1), u);
                u = fmaf(v_w.z, tanhf(xq.z + sp.z + pl2), u);
                u = fmaf(v_w.w, tanhf(xq.w + sp.w + pl3), u);
                pub(V_S + p * VL_S + pos, sigm(wave_total(u)), tag);
                __builtin_amdgcn_wave_barrier();
            }
        }
        PH(9);
        // ---- L6: normalise (:204), this workgroup's copies of attention / cumulative (:205-206), context (:207) ----
        poll_in(sc, V_S + p * VL_S, n, tag, 6);
        STAGED();
        PH(10);
        {
            float part = 0.f;
            for (int k = tid; k < n; k += NT) part += sc[k];
            part = wave_total(part);                                   // the same tree in every workgroup: the same total everywhere
            if (lane == 0) red[w] = part;
            __syncthreads();
            const float total = (red[0] + red[1]) + (red[2] + red[3]);
            for (int k = tid; k < n; k += NT) {
                const float v = sc[k] / total;
                sc[k] = v; att[k] = v; cum[k] += v;
            }
            __syncthreads();
            for (int pos = gt; pos < n; pos += NGT) a.scores_out[(size_t)step * n + pos] = sc[pos];
            if (u256) {
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (64 * i < n) s = fmaf(sc[lane + 64 * i], seqc[i], s);
                pub(V_CTX + p * T_DD + gw, wave_total(s), tag);
            }
        }
        PH(11);
        // ---- L7: rnn_input on [context, attn_h] (:246-247) ----
        poll_in(xb[0], V_CTX + p * T_DD, T_DD, tag, 7);
        STAGED();
        PH(12);
        pub(V_X + p * T_LD + gw, wave_total(fma4(w_ri[1], ahv + k0, fma4(w_ri[0], xb[0] + k0, 0.f))) + b_ri, tag);
        PH(13);
        // ---- L8: residual LSTMCell 1 (:249-251; ATen lstm_cell: gates i, f, g, o) ----
        poll_in(xb[1], V_X + p * T_LD, T_LD, tag, 8);
        if (step > 0) poll_in(xb[1] + T_LD, V_H1 + (p ^ 1) * T_LD, T_LD, ptag, 8);
        STAGED();
        PH(14);
        {
            float g4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                g4[q] = wave_total(fma4(l1i[q][1], xb[1] + 256 + k0, fma4(l1i[q][0], xb[1] + k0, 0.f))) + b_l1[q] +
                        wave_total(fma4(l1h[q][1], xb[1] + T_LD + 256 + k0, fma4(l1h[q][0], xb[1] + T_LD + k0, 0.f))) + b_l1h[q];
            c1 = sigm(g4[1]) * c1 + sigm(g4[0]) * tanhf(g4[2]);
            const float h = sigm(g4[3]) * tanhf(c1);
            pub(V_H1 + p * T_LD + gw, h, tag);
            pub(V_X2 + p * T_LD + gw, xb[1][gw] + h, tag);
        }
        PH(15);
        // ---- L9: residual LSTMCell 2 (:254-256) ----
        poll_in(xb[0], V_X2 + p * T_LD, T_LD, tag, 9);
        if (step > 0) poll_in(xb[0] + T_LD, V_H2 + (p ^ 1) * T_LD, T_LD, ptag, 9);
        STAGED();
        PH(16);
        {
            float g4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                g4[q] = wave_total(fma4(l2i[q][1], xb[0] + 256 + k0, fma4(l2i[q][0], xb[0] + k0, 0.f))) + b_l2[q] +
                        wave_total(fma4(l2h[q][1], xb[0] + T_LD + 256 + k0, fma4(l2h[q][0], xb[0] + T_LD + k0, 0.f))) + b_l2h[q];
            c2 = sigm(g4[1]) * c2 + sigm(g4[0]) * tanhf(g4[2]);
            const float h = sigm(g4[3]) * tanhf(c2);
            pub(V_H2 + p * T_LD + gw, h, tag);
            pub(V_X3 + p * T_LD + gw, xb[0][gw] + h, tag);
        }
        PH(17);
        // ---- L10: mel_proj (:262-263): entry q = m r + j is row (m, j) of the (n_mels, max_r) view ----
        poll_in(xb[1], V_X3 + p * T_LD, T_LD, tag, 10);
        STAGED();
        PH(18);
        if (gw < nmel) {
            const float s = wave_total(fma4(w_mp[1], xb[1] + 256 + k0, fma4(w_mp[0], xb[1] + k0, 0.f)));
            if (lane == 0) a.mel_out[(size_t)step * nmel + gw] = s;
            pub(V_MEL + p * VL_MEL + gw, s, tag);
        }
        for (int q = gw + R_NWV; q < nmel; q += R_NWV) {              // r >= 7 only: the rows past the 512 resident ones, from L2
            const float *wr = a.w.mel_proj_w + (size_t)((q / r) * a.max_r + q % r) * T_LD;
            const float s = wave_total(fma4(ldw4(wr + 256 + k0, true), xb[1] + 256 + k0, fma4(ldw4(wr + k0, true), xb[1] + k0, 0.f)));
            if (lane == 0) a.mel_out[(size_t)step * nmel + q] = s;
            pub(V_MEL + p * VL_MEL + q, s, tag);
        }
        PH(19);
        if (PROF && gt == 0) profl[20] += 1;
    }
    if (gt == 0) *a.steps_done = step;
    if (PROF && wg == 0) {
        __syncthreads();
        if (tid < 24) a.tv[V_END + tid] = profl[tid];
    }
#undef STAGED
#undef PH
}


// =================================================================================================================================
// wrnn_bigru_kernel: the bidirectional GRU that ends a CBHG (reference models/tacotron.py:95 `nn.GRU(channels, channels,
// batch_first=True, bidirectional=True)`, applied at :137) for ONE sequence, hidden size 128.  MIOpen runs it as T x 2 rounds of
// small launches (47 ms for the post-net's 800 frames, twice the decoder loop); here one workgroup per direction keeps W_hh
// (384 x 128 f32 = 192 KB) in the registers of 384 threads (thread j = row j, 128 VGPRs) and h in LDS: a step is 32 broadcast
// 16-byte LDS reads + 128 FMAs per thread, one barrier, the gate math of ATen's gru_cell on 128 threads, one barrier.
// gi = W_ih x + b_ih for all frames is a plain GEMM, done by the caller (F.linear).
// =================================================================================================================================
constexpr int GR_H = 128, GR_NT = 3 * GR_H;

struct BigruArgs {
    const float *gi[2];        // [T][384] per direction (forward, reverse)
    const float *w_hh[2];      // [384][128]
    const float *b_hh[2];      // [384]
    float *out;                // [T][256]: forward h at [:, :128], reverse h at [:, 128:]
    int T;
};

__global__ __launch_bounds__(GR_NT, 1) void wrnn_bigru_kernel(const BigruArgs a)
{
    __shared__ __attribute__((aligned(16))) float hs[GR_H];
    __shared__ float gs[GR_NT];
    const int j = threadIdx.x, dir = blockIdx.x, T = a.T;
    const float *gi = a.gi[dir];
    float wr[GR_H];
#pragma unroll
    for (int k = 0; k < GR_H; k += 4) {
        const float4 v = *reinterpret_cast<const float4 *>(a.w_hh[dir] + (size_t)j * GR_H + k);
        wr[k] = v.x; wr[k + 1] = v.y; wr[k + 2] = v.z; wr[k + 3] = v.w;
    }
    const float bh = a.b_hh[dir][j];
    if (j < GR_H) hs[j] = 0.f;
    __syncthreads();
    float h = 0.f;                                                     // thread u < 128: h[u]
    int t = dir == 0 ? 0 : T - 1;
    const int dt = dir == 0 ? 1 : -1;
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    if (j < GR_H) { g0 = gi[(size_t)t * GR_NT + j]; g1 = gi[(size_t)t * GR_NT + GR_H + j]; g2 = gi[(size_t)t * GR_NT + 2 * GR_H + j]; }
    for (int s = 0; s < T; ++s, t += dt) {
        float n0 = 0.f, n1 = 0.f, n2 = 0.f;                             // the next step's gi, in flight under this step's products
        if (j < GR_H && s + 1 < T) {
            const float *gn = gi + (size_t)(t + dt) * GR_NT;
            n0 = gn[j]; n1 = gn[GR_H + j]; n2 = gn[2 * GR_H + j];
        }
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int k = 0; k < GR_H; k += 4) {
            const float4 hv = *reinterpret_cast<const float4 *>(hs + k);
            a0 = fmaf(wr[k], hv.x, a0); a1 = fmaf(wr[k + 1], hv.y, a1); a2 = fmaf(wr[k + 2], hv.z, a2); a3 = fmaf(wr[k + 3], hv.w, a3);
        }
        gs[j] = ((a0 + a1) + (a2 + a3)) + bh;
        __syncthreads();
        if (j < GR_H) {                                                 // ATen gru_cell: r, z, n; h' = (h - n) z + n
            const float rg = 1.0f / (1.0f + expf(-(g0 + gs[j])));
            const float zg = 1.0f / (1.0f + expf(-(g1 + gs[GR_H + j])));
            const float ng = tanhf(g2 + rg * gs[2 * GR_H + j]);
            h = (h - ng) * zg + ng;
            hs[j] = h;
            a.out[(size_t)t * (2 * GR_H) + dir * GR_H + j] = h;
            g0 = n0; g1 = n1; g2 = n2;
        }
        __syncthreads();
    }
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

// workspace: [A_END floats | U_END words] of the flag-barrier kernel (the status words live there for both kernels), then the
// resident kernel's tagged vectors
constexpr size_t TACO_V1_BYTES = (size_t)A_END * 4 + (size_t)U_END * 4;
static_assert(TACO_V1_BYTES % 8 == 0, "tagged entries are 8-byte aligned");
extern "C" size_t wrnn_taco_workspace_bytes(void) { return TACO_V1_BYTES + (size_t)(V_END + 24) * 8; }

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
    DeviceGuard dg(device);
    hipError_t e = dg.err;
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
    a.tv = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(c->workspace) + TACO_V1_BYTES);
    a.nwg = prop.multiProcessorCount < T_MAXWG ? prop.multiProcessorCount : T_MAXWG;
    a.stop_threshold = c->stop_threshold;
    // the register-resident kernel needs exactly R_NWG co-resident workgroups (one LSTM unit per wave) and r <= R_MAXR frames per step
    const bool resident_ok = prop.multiProcessorCount >= R_NWG && c->r <= R_MAXR;
    if (c->variant < 0 || c->variant > 3 || (c->variant >= 2 && !resident_ok)) {
        taco_err("variant %d: 0 = auto, 1 = flag-barrier kernel, 2 = register-resident kernel (needs >= %d CUs, r <= %d; device has %d, r = %d)",
                 c->variant, R_NWG, R_MAXR, prop.multiProcessorCount, c->r);
        return c->variant >= 2 && c->variant <= 3 ? WRNN_ERR_RESIDENCY : WRNN_ERR_ARG;
    }
    const bool resident = c->variant >= 2 || (c->variant == 0 && resident_ok);
    hipStream_t stream = reinterpret_cast<hipStream_t>(c->stream);
    e = hipMemsetAsync(a.uw, 0, (size_t)U_END * 4 + (resident ? (size_t)V_END * 8 : 0), stream);     // tags 0: never a step's tag
    if (e != hipSuccess) { taco_err("hipMemsetAsync: %s", hipGetErrorString(e)); return WRNN_ERR_HIP; }
    void *params[] = {(void *)&a};
    if (resident) {
        a.nwg = R_NWG;
        e = hipLaunchCooperativeKernel(c->variant == 3 ? (const void *)wrnn_taco_resident_kernel<true> : (const void *)wrnn_taco_resident_kernel<false>,
                                       dim3(R_NWG), dim3(NT), params, 0, stream);
    } else {
        e = hipLaunchCooperativeKernel((const void *)wrnn_taco_decoder_kernel, dim3(a.nwg), dim3(NT), params, 0, stream);
    }
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

// The CBHG's bidirectional GRU (see wrnn_bigru_kernel).  All pointers device pointers; asynchronous on `stream`.
extern "C" int wrnn_bigru(int device, const wrnn_bigru_call *c)
{
    if (!c || c->struct_bytes != sizeof(wrnn_bigru_call)) { taco_err("null argument / struct size mismatch"); return WRNN_ERR_ARG; }
    if (c->hidden != GR_H || c->T < 1 || !c->gi_fwd || !c->gi_rev || !c->w_hh_fwd || !c->w_hh_rev || !c->b_hh_fwd || !c->b_hh_rev || !c->out) {
        taco_err("bad call: hidden=%d (this build: %d) T=%d or a null buffer", c->hidden, GR_H, c->T);
        return WRNN_ERR_ARG;
    }
    DeviceGuard dg(device);
    hipError_t e = dg.err;
    if (e != hipSuccess) { taco_err("hipSetDevice: %s", hipGetErrorString(e)); return WRNN_ERR_HIP; }
    BigruArgs a;
    a.gi[0] = c->gi_fwd; a.gi[1] = c->gi_rev; a.w_hh[0] = c->w_hh_fwd; a.w_hh[1] = c->w_hh_rev; a.b_hh[0] = c->b_hh_fwd; a.b_hh[1] = c->b_hh_rev;
    a.out = c->out; a.T = c->T;
    hipLaunchKernelGGL(wrnn_bigru_kernel, dim3(2), dim3(GR_NT), 0, reinterpret_cast<hipStream_t>(c->stream), a);
    e = hipGetLastError();
    if (e != hipSuccess) { taco_err("wrnn_bigru_kernel launch: %s", hipGetErrorString(e)); return WRNN_ERR_HIP; }
    return WRNN_OK;
}
