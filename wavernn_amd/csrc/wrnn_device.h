// wrnn_device.h -- shared device-side definitions of the MI355X (gfx950) WaveRNN loop kernels.
//
// The arithmetic restated here is the reference's per-step dataflow (fatchord/WaveRNN
// models/fatchord_version.py:203-237, utils/distribution.py:102-121, ATen gru_cell); see DESIGN.md.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wrnn {

// Host side: the entry points run on the device of the pack / call, and leave the calling thread's current device as they found it
// (a caller such as PyTorch tracks its own current device: switching it behind its back sends its next launch to the wrong GPU).
struct DeviceGuard {
    int prev = -1;
    hipError_t err;
    explicit DeviceGuard(int device)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        err = (prev == device) ? hipSuccess : hipSetDevice(device);
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

constexpr int H = 512;        // rnn_dims == fc_dims (this build)
constexpr int MEL = 80;       // feat_dims
constexpr int AUX = 32;       // aux_dims = res_out_dims / 4
constexpr int KCOND = MEL + AUX;   // conditioning inputs of the I layer (112)
constexpr int LAST_SCALE = 11;     // stretch factor of the last up-sampling stage when wrnn_duo_kernel forms it in the loop (hparams.py: voc_upsample_factors[2])
constexpr int SEG = 16;       // segments per persistent launch group (= MFMA N)
constexpr int LDA = 516;      // padded row stride (floats) of the LDS activation tiles
constexpr int MAXCL = 4;      // cluster kernels: at most 4 independent clusters per chip
// role-split loop kernel (wrnn_loop.hip): tag-free exchange buffer [cluster][slot][layer h1 h2 y1 y2 lg x1 x2][ring][SEG*H floats]
constexpr int LMAXG = 8;      // slots (groups in flight) per cluster the exchange buffer is sized for
constexpr int NXLAYER = 8;      // h1, h2, y1, y2, RAW logits, x1 = xi + h1, x2 = x1 + h2, x_t (16 words: samples drawn by role B)
constexpr int XRING = 4;
constexpr size_t XBUF_FLOATS = (size_t)MAXCL * LMAXG * NXLAYER * XRING * SEG * H;
constexpr int STATUS_WORDS = 16;    // 0 abort flag, 1 code, 2 wg, 3 step, 4 detail; 8 = the loop kernel kind a call settled on at step 0 (continuations must match)
constexpr int XCC_WORDS = MAXCL * 128; // wrnn_duo.hip placement handshake
constexpr int MAXWG = 256;    // workgroups of a persistent launch

// Everything the loop kernels read.  All pointers are device pointers.
struct LoopArgs {
    // raw row-major weights (persistent kernel gathers its MFMA A-fragments from these once)
    const float *I_w0;                  // [H]            column 0 of I.weight (the x_{t-1} tap)
    const float *w_ih1, *w_hh1;         // [3H,H]
    const float *b_ih1, *b_hh1;         // [3H]
    const float *w_ih2;                 // [3H,H+AUX] (only the first H columns are used in-loop)
    const float *w_hh2;                 // [3H,H]
    const float *b_hh2;                 // [3H]   (b_ih2 is folded into c2f)
    const float *fc1_w, *fc2_w;         // [H,H+AUX]
    const float *fc3_w, *fc3_b;         // [C,H], [C]
    // k-major (transposed) copies for the stream kernel: [K][rows]
    const float *w_ih1T, *w_hh1T, *w_ih2T, *w_hh2T;   // [H][3H]
    const float *fc1T, *fc2T;                         // [H][H]
    const float *fc3T;                                // [H][C]
    // hoisted conditioning (wrnn_cond kernels)
    const float *cI;                    // [T][Btot][H]   b_I + W_I[:,1:] . [m_t, a1_t]
    const float *c2f, *c3f, *c4f;       // [NF+1][3H], [NF+1][H], [NF+1][H]  per-frame aux projections + bias
    const float *noise;                 // MOL [T][11*Btot]; RAW [T][Btot][C]
    const float *noise_pre;             // MOL, pipelined kernel: [T][11*Btot] derived variates (wrnn_noise_mol_kernel)
    // block-sparse GRU pack (wrnn_sparse.hip): matrix m in {ih1,hh1,ih2,hh2}, block row (16-row block rb, gate g), NBP padded blocks
    const float *sp_vals;               // [4][32][3][NBP][16]  block values (16 rows of one column)
    const int *sp_cols;                 // [4][32][3][NBP]      their column indices (padding: column 0, zero values)
    const float *sp_fc_vals;            // [2][32][NBP][16]     fc1 / fc2 (their x2 / y1 columns) packed the same way, or null: dense fc stages (round 6)
    const int *sp_fc_cols;              // [2][32][NBP]
    const float *force_x;               // optional [Btot][T]
    float *out;                         // [Btot][T]
    float *dbg_logits;                  // optional [T][Btot][C]
    unsigned *status;                   // [STATUS_WORDS]: 0 abort flag, 1 code, 2 wg, 3 step, 4 detail
    u64 *prof;                          // optional [256 workgroups][32] shader-clock totals per phase (wrnn_options.phase_clocks; layouts: the kernels' "PROF" notes)
    // segment table: segment b, step t reads conditioning position p = seg_pos[b] + t; p >= seg_lim[b] is the
    // fold's zero padding (fatchord_version.py:326-330).  One utterance: seg_pos[b] = b*(target+overlap),
    // seg_lim[b] = L.  Several utterances: positions in the concatenated conditioning (each utterance starts
    // on a frame boundary).
    const int *seg_pos, *seg_lim;       // [Btot]
    int Btot, b0, nb, T, hop, NF, C;
    int NG;                             // cluster kernel: number of <= SEG-segment groups the Btot segments form
    // ---- role-split loop kernel (wrnn_loop.hip): one ROUND of segments [rb0, rb0 + Btot) of a call with Nall segments, steps
    //      [t0, t1); NG = groups of the round, G = groups in flight per cluster
    float *xbuf;                        // [XBUF_FLOATS] exchange buffer, sentinel-filled before every launch
    float *state;                       // [clusters*64][G][LGRP] per-group state carried between launches of one round
    const float *cIf;                   // [t1 - cI_t0 ...][NG][SEG*H] hoisted conditioning of the round in fragment order (slab)
    int t0, t1, cI_t0, noise_t0;        // noise / noise_pre row 0 is step noise_t0, cIf row 0 is step cI_t0
    int tuning;                         // A/B switches of the loop kernel (wrnn_options.tuning)
    int rb0, Nall, G, resume;           // resume != 0: restore the per-group state instead of the zero initial state
    const float *fc3f;                  // MOL: fc3.weight in A-fragment order [tile 2][wave 4][k-block 8][lane 64][4] (wrnn_duo.hip's sampler reads it from L2)
    // ---- wrnn_duo.hip (round 4)
    const float *u1;                    // [3H] rnn1.weight_ih . I.weight[:,0]: the x_{t-1} term of rnn1's gi, applied in the gates' pointwise half
    unsigned hop_magic;                 // p / hop == __umulhi(p, hop_magic) >> hop_shift for 0 <= p < 2^31 (0: divide)
    int hop_shift;
    const float *mels_up, *aux_fr;      // [L][MEL], [NF][4 AUX]: the conditioning itself (wrnn_duo.hip forms cI(t) in the loop: SURVEY.md 8 row f1)
    const float *I_cT, *I_b;            // [KCOND][H] transposed I.weight[:, 1:], [H] I.bias
    // the LAST up-sampling stage (Stretch2d(11) + its 23-tap conv + the crop, fatchord_version.py:73-80, :86-88) formed in the loop too:
    // mel_stage != 0: `mels_up` is that stage's INPUT [rows][MEL]; step t of segment b sits at un-cropped position j = seg_pos[b] + t +
    // seg_moff[b], and mel(j) = c[ph][0] row(j/11 - 1) + c[ph][1] row(j/11) + c[ph][2] row(j/11 + 1), ph = j % 11, c = mel_coef[11][3]
    int mel_stage;
    const int *seg_moff;                // [Btot]
    const float *mel_coef;              // [LAST_SCALE][3]: sums of the conv taps that fall on each of the three rows
    int tab_fps, tab_t0;                // wrnn_duo.hip: c2f / c3f / c4f are per-SEGMENT tables of the slab that starts at step tab_t0: row
                                        // (segment index in the call) * tab_fps + frame - (seg_pos + tab_t0) / hop; zero row = Nall * tab_fps
    unsigned *xcc_tab;                  // [MAXCL * 128] zeroed before every launch: XCC id + 1 of every workgroup (placement handshake)
    int kind_tag;                       // 1 wrnn_loop_kernel, 2 wrnn_duo_kernel, 3 wrnn_sparse_kernel, 4 wrnn_chain_kernel, 5 wrnn_octo_kernel: recorded in status[8] by the launch that starts a call at step 0,
                                        // checked by every continuing launch (the two kernels keep different state / ring layouts)
};

// A call's first launch records which loop kernel runs it; a continuation (resume) that was planned onto the other kernel -- e.g. the
// `auto` fallback from two workgroups per CU to one struck on one slice only -- fails loudly instead of resuming from foreign state.
__device__ __forceinline__ void check_kind(const LoopArgs &a)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (!a.resume) a.status[8] = (unsigned)a.kind_tag;
        else if (a.status[8] != (unsigned)a.kind_tag) {
            if (atomicCAS(a.status + 1, 0u, 0x7F0u | (unsigned)a.kind_tag) == 0u) { a.status[2] = a.status[8]; a.status[3] = (unsigned)a.t0; a.status[4] = 0u; }
            __hip_atomic_store(a.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}


// arguments of the dimension-generic loop kernel (wrnn_generic.hip): any rnn / fc / feat / aux dims, nothing hoisted
struct GenArgs {
    // k-major copies: XT[k][rows]
    const float *I_T, *I_b;             // [1 + M + A][H], [H]
    const float *w_ih1T, *w_hh1T, *b_ih1, *b_hh1;     // [H][3H] x 2, [3H] x 2
    const float *w_ih2T, *w_hh2T, *b_ih2, *b_hh2;     // [H + A][3H], [H][3H]
    const float *fc1T, *fc1_b;          // [H + A][F], [F]
    const float *fc2T, *fc2_b;          // [F + A][F], [F]
    const float *fc3T, *fc3_b;          // [F][C], [C]
    const float *mels_up, *aux;         // [L][M], [NF][4 A]
    const float *noise;                 // MOL [T][11 B]; RAW [T][B][C]
    const float *force_x;               // optional [B][T]
    float *out, *dbg_logits;            // [B][T], optional [T][B][C]
    const int *seg_pos, *seg_lim;       // [B]
    int H, F, M, A, C, B, T, hop;
};

// arguments of the hoisted-conditioning kernels (wrnn_cond.hip)
struct CondArgs {
    const float *mels_up;   // [L][MEL]
    const float *aux;       // [NF][4*AUX]
    const float *I_cT;      // [KCOND][H]   transposed I.weight[:,1:]
    const float *I_b;       // [H]
    const float *c2_wT;     // [AUX][3H]    transposed rnn2.weight_ih[:,H:]
    const float *b_ih2;     // [3H]
    const float *c3_wT, *fc1_b;   // [AUX][H], [H]
    const float *c4_wT, *fc2_b;   // [AUX][H], [H]
    float *cI, *c2f, *c3f, *c4f;
    const int *seg_pos, *seg_lim;       // [B] (see LoopArgs)
    int B, T, hop, NF;
    int t0, t1, rb0, NG;                // fragment-order slab form (wrnn_cond_frag_kernel): steps [t0, t1) of the round [rb0, rb0 + B)
    int FPS;                            // per-segment slab tables (wrnn_cond_frame_slab_kernel): rows per segment
};

__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

// ATen CPU gru_cell algebra (what nn.GRUCell runs at fatchord_version.py:210,214):
// r = sig(gh_r + gi_r), z = sig(gh_z + gi_z), n = tanh(gi_n + gh_n * r), h' = (h - n) * z + n
__device__ __forceinline__ float gru_update(float gi_r, float gi_z, float gi_n, float gh_r, float gh_z,
                                            float gh_n, float h)
{
    const float r = sigmoid_f(gh_r + gi_r);
    const float z = sigmoid_f(gh_z + gi_z);
    const float n = tanhf(gi_n + gh_n * r);
    return (h - n) * z + n;
}

// tanhf without the library's branch: both of its paths (|x| >= 0.625: 1 - 2 / (exp(2|x|) + 1) with the hardware reciprocal;
// else the odd polynomial) with the library's own operations and constants, then a select -- the same instructions on the same
// operands as __ocml_tanh_f32 (checked in the ISA), so bit-identical, and straight-line
__device__ __forceinline__ float tanh_sel(float x)
{
    const float ax = fabsf(x);
    const float e = expf(ax + ax);
    const float ra = fmaf(__builtin_amdgcn_rcpf(e + 1.0f), -2.0f, 1.0f);
    const float x2 = x * x;
    float p = fmaf(__uint_as_float(0xbbbac73du), x2, __uint_as_float(0x3ca908c9u));
    p = fmaf(x2, p, __uint_as_float(0xbd5c1c4eu));
    p = fmaf(x2, p, __uint_as_float(0x3e088382u));
    p = fmaf(x2, p, __uint_as_float(0xbeaaaa99u));
    const float rb = fmaf(x2, ax * p, ax);
    return copysignf(!(ax < 0.625f) ? ra : rb, x);
}
__device__ __forceinline__ float gru_update_sel(float gi_r, float gi_z, float gi_n, float gh_r, float gh_z, float gh_n, float h)
{
    const float r = sigmoid_f(gh_r + gi_r);
    const float z = sigmoid_f(gh_z + gi_z);
    const float n = tanh_sel(gi_n + gh_n * r);
    return (h - n) * z + n;
}

// Butterfly exchanges of a 64-lane wave WITHOUT the LDS crossbar (round 6): the RAW sampler's five 6-level reductions were 30 dependent ds_bpermute_b32 --
// ~100+ cycles each behind an lgkmcnt wait, on the slot's chain.  xor_pair<M>(v, a, b) leaves {a, b} = {v of this lane, v of lane ^ M} IN UNSPECIFIED ORDER
// (gfx950's v_permlane32_swap / v_permlane16_swap hand the pair back either way round; M <= 8: DPP row_ror:8, row_shl:4 | row_shr:4 under bank masks,
// quad_perm) -- for a combine that is symmetric in its operands (IEEE a + b == b + a, max, argmax with an index tie-break) the result is bit-identical
// to `v op __shfl_xor(v, M, 64)`: the pairing of every level, hence the summation tree, is the same.  wrnn_selftest(device, 4) checks all six against
// __shfl_xor on the device.
template <int M> __device__ __forceinline__ void xor_pair_u(unsigned v, unsigned &a, unsigned &b)
{
    static_assert(M == 32 || M == 16 || M == 8 || M == 4 || M == 2 || M == 1, "xor_pair: lane mask");
    if constexpr (M == 32) {
        const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
        a = r[0]; b = r[1];
    } else if constexpr (M == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        a = r[0]; b = r[1];
    } else if constexpr (M == 8) {
        a = v; b = __builtin_amdgcn_update_dpp(0u, v, 0x128 /* row_ror:8 */, 0xF, 0xF, false);
    } else if constexpr (M == 4) {
        unsigned t = __builtin_amdgcn_update_dpp(0u, v, 0x104 /* row_shl:4: lane i <- i + 4 */, 0xF, 0x5 /* banks 0, 2 */, false);
        t = __builtin_amdgcn_update_dpp(t, v, 0x114 /* row_shr:4: lane i <- i - 4 */, 0xF, 0xA /* banks 1, 3 */, false);
        a = v; b = t;
    } else if constexpr (M == 2) {
        a = v; b = __builtin_amdgcn_update_dpp(0u, v, 0x4E /* quad_perm [2,3,0,1] */, 0xF, 0xF, false);
    } else {
        a = v; b = __builtin_amdgcn_update_dpp(0u, v, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, false);
    }
}
template <int M> __device__ __forceinline__ void xor_pair(float v, float &a, float &b)
{
    unsigned ua, ub;
    xor_pair_u<M>(__float_as_uint(v), ua, ub);
    a = __uint_as_float(ua); b = __uint_as_float(ub);
}
template <int M> __device__ __forceinline__ float wave_sum_level(float v) { float a, b; xor_pair<M>(v, a, b); return a + b; }
template <int M> __device__ __forceinline__ float wave_max_level(float v) { float a, b; xor_pair<M>(v, a, b); return fmaxf(a, b); }
// == for (m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64), bit for bit
__device__ __forceinline__ float wave_sum64(float v)
{
    v = wave_sum_level<32>(v); v = wave_sum_level<16>(v); v = wave_sum_level<8>(v);
    v = wave_sum_level<4>(v); v = wave_sum_level<2>(v); return wave_sum_level<1>(v);
}
__device__ __forceinline__ float wave_max64(float v)
{
    v = wave_max_level<32>(v); v = wave_max_level<16>(v); v = wave_max_level<8>(v);
    v = wave_max_level<4>(v); v = wave_max_level<2>(v); return wave_max_level<1>(v);
}
// arg max with the smaller index on a tie (== the shuffle form: `if (ob > best || (ob == best && oi < bidx)) take the other`): every lane ends with the wave's pair
template <int M> __device__ __forceinline__ void wave_argmax_level(float &best, int &bidx)
{
    float va, vb;
    unsigned ia, ib;
    xor_pair<M>(best, va, vb);
    xor_pair_u<M>((unsigned)bidx, ia, ib);
    const bool take_b = vb > va || (vb == va && (int)ib < (int)ia);
    best = take_b ? vb : va;
    bidx = (int)(take_b ? ib : ia);
}
__device__ __forceinline__ void wave_argmax64(float &best, int &bidx)
{
    wave_argmax_level<32>(best, bidx); wave_argmax_level<16>(best, bidx); wave_argmax_level<8>(best, bidx);
    wave_argmax_level<4>(best, bidx); wave_argmax_level<2>(best, bidx); wave_argmax_level<1>(best, bidx);
}

// utils/distribution.py:106-108  logit_probs - log(-log(u))
__device__ __forceinline__ float mol_gumbel(float lp, float u) { return lp - logf(-logf(u)); }

// the same two steps on pre-transformed noise (wrnn_noise_mol_kernel): g = log(-log u), l = log u - log(1-u)
__device__ __forceinline__ float mol_gumbel_pre(float lp, float g) { return lp - g; }
__device__ __forceinline__ float mol_sample_pre(float mean, float ls, float l)
{
    const float lsmin = -32.23619130191664f;
    ls = fmaxf(ls, lsmin);
    float x = mean + expf(ls) * l;
    x = fmaxf(x, -1.0f);
    return fminf(x, 1.0f);
}

// utils/distribution.py:113-121  x = clamp(mean + exp(max(ls, ln 1e-14)) * (log u - log(1-u)), -1, 1)
__device__ __forceinline__ float mol_sample(float mean, float ls, float u)
{
    const float lsmin = -32.23619130191664f;   // float(np.log(1e-14))
    ls = fmaxf(ls, lsmin);
    float x = mean + expf(ls) * (logf(u) - logf(1.0f - u));
    x = fmaxf(x, -1.0f);
    return fminf(x, 1.0f);
}

// argmax over the 16 lanes of a DPP row with the first-max tie rule (lowest index wins), as an all-reduce in four DPP
// steps (xor 1, xor 2 inside quads, then half-row mirror and row mirror) instead of eight ds_bpermute shuffles.
template <int CTRL>
__device__ __forceinline__ void argmax_dpp_step(float &best, int &bidx)
{
    const float ob = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, best), CTRL, 0xF, 0xF, true));
    const int oi = __builtin_amdgcn_update_dpp(0, bidx, CTRL, 0xF, 0xF, true);
    if (ob > best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
}
__device__ __forceinline__ void argmax_row16(float &best, int &bidx)
{
    argmax_dpp_step<0xB1>(best, bidx);     // quad_perm [1,0,3,2]
    argmax_dpp_step<0x4E>(best, bidx);     // quad_perm [2,3,0,1]
    argmax_dpp_step<0x141>(best, bidx);    // row_half_mirror
    argmax_dpp_step<0x140>(best, bidx);    // row_mirror
}

// frame of conditioning position p (Stretch2d repeats each frame `hop` times; p >= L is the fold's zero pad)
__device__ __forceinline__ int cond_frame(const int *seg_pos, const int *seg_lim, int b, int t, int hop, int NF)
{
    const int p = seg_pos[b] + t;
    return (p < seg_lim[b]) ? (p / hop) : NF;
}

__device__ __forceinline__ u64 ld_agent(const u64 *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(u64 *p, u64 v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned ld_agent32(const unsigned *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// record the first failure and raise the abort flag every spinning workgroup polls
__device__ __forceinline__ void report_failure(unsigned *status, unsigned code, unsigned wg, unsigned step,
                                               unsigned detail)
{
    // (the compare-and-swap wants its operand pair in VGPRs; built HERE -- the optimiser was seen to build it in the kernel's prologue, far
    // outside the step loop, and to spill it: the only VGPR spill of the RAW duo kernel)
    unsigned c;
    asm volatile("v_mov_b32 %0, %1" : "=v"(c) : "s"(code));
    if (atomicCAS(status + 1, 0u, c) == 0u) {
        status[2] = wg;
        status[3] = step;
        status[4] = detail;
    }
    __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------------------
// MFMA building blocks (v_mfma_f32_16x16x4_f32: exact f32, D = A(16x4) * B(4x16) + C).
// Lane l supplies A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15]; D reg v holds D[(l>>4)*4+v][l&15].
// A workgroup is NW = 4 waves (one per SIMD, so each wave may keep up to 512 VGPRs); wave w owns the
// K chunk [128w, 128w+128): 32 MFMAs per tile.  A fragments (weights) live in VGPRs for the whole kernel;
// B fragments (activations, [seg][k] rows of stride LDA in LDS) are read with ds_read_b128: lane reads
// act[j][kbase + 16r + 4(l>>4) .. +3], r = 0..7, so MFMA (r,e) contracts k = kbase + 16r + 4(l>>4) + e.
// Two accumulators (even / odd r) hide the 40-cycle dependent-accumulator latency behind the 32-cycle issue.
// ---------------------------------------------------------------------------------------------------
constexpr int NW = 4;               // waves per workgroup
constexpr int NT = NW * 64;         // threads per workgroup; thread tid owns activation columns 2tid, 2tid+1
constexpr int KCH = H / NW;         // K chunk per wave (128)
constexpr int AF = KCH / 4;         // A-fragment registers per tile (32)

__device__ __forceinline__ void load_afrag(float (&a)[AF], const float *W, int ld, int row, bool valid,
                                           int kbase_lane /* = KCH*w + 4*(l>>4) */)
{
#pragma unroll
    for (int r = 0; r < AF / 4; ++r) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid) v = *reinterpret_cast<const float4 *>(W + (size_t)row * ld + kbase_lane + 16 * r);
        a[4 * r + 0] = v.x; a[4 * r + 1] = v.y; a[4 * r + 2] = v.z; a[4 * r + 3] = v.w;
    }
}

__device__ __forceinline__ f32x4 mfma_tile(const float (&a)[AF], const float *act_lane /* act + j*LDA + kbase_lane */)
{
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < AF / 4; r += 2) {
        const float4 b0 = *reinterpret_cast<const float4 *>(act_lane + 16 * r);
        const float4 b1 = *reinterpret_cast<const float4 *>(act_lane + 16 * (r + 1));
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * r + 0], b0.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * r + 4], b1.x, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * r + 1], b0.y, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * r + 5], b1.y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * r + 2], b0.z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * r + 6], b1.z, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * r + 3], b0.w, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * r + 7], b1.w, acc1, 0, 0, 0);
    }
    return acc0 + acc1;
}

// partial tile of wave w -> part[(w*2+slot)*256 + i*16 + j]
__device__ __forceinline__ void store_partial(float *part, int w, int slot, int lane, f32x4 acc)
{
    float *p = part + (w * 2 + slot) * 256 + ((lane >> 4) * 4) * 16 + (lane & 15);
    p[0] = acc[0]; p[16] = acc[1]; p[32] = acc[2]; p[48] = acc[3];
}

__device__ __forceinline__ float reduce_partial(const float *part, int slot, int i, int j)
{
    float s = part[(0 * 2 + slot) * 256 + i * 16 + j];
#pragma unroll
    for (int w = 1; w < NW; ++w) s += part[(w * 2 + slot) * 256 + i * 16 + j];
    return s;
}

// ---------------------------------------------------------------------------------------------------
// Inter-workgroup exchange: 8-byte {f32 value, tag} granules (value in the low dword), ONE relaxed
// agent-scope (sc1, write-through) store per granule, sc1 polling loads, no fences (MI355X guide,
// Guideline 16 form R2: the data is its own flag).  Layout per layer: G[seg][hidden index], 4 KB per segment.
// Sweep: thread tid = (row r = tid>>4, c = tid&15) reads the whole row r of the layer with 16
// buffer_load_dwordx4 sc1 (SGPR descriptor + one VGPR offset + immediates; lanes c = 0..15 cover 256
// contiguous bytes per load), i.e. granule pairs (i*16+c)*2, +1 for i = 0..15, and writes the values to the
// same columns of an LDS tile row.  Bounded spin; every 8-byte half carries its own tag.
// ---------------------------------------------------------------------------------------------------
constexpr unsigned SPIN_LIMIT = 4000000u;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}

__device__ __forceinline__ void publish(u64 *G, unsigned tag, int j, int k, float val)
{
    st_agent(G + j * H + k, ((u64)tag << 32) | (u64)__float_as_uint(val));
}

// column (float index) of piece i owned by thread-in-row c: 2 floats
__device__ __forceinline__ int own_col(int i, int c) { return (i * 16 + c) * 2; }

__device__ __forceinline__ bool sweep(__amdgpu_buffer_rsrc_t rs, int layer, unsigned tag, int nb, int tid,
                                      float *dst, unsigned *status)
{
    const int r = tid >> 4, c = tid & 15;
    if (r >= nb) return true;
    const int voff = r * (H * 8) + c * 16;
    const int soff = layer * (SEG * H * 8);
    u32x4 x[16];
    unsigned spins = 0;
    for (;;) {
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + i * 256, soff, 16 /* sc1 */);
        bool ok = true;
#pragma unroll
        for (int i = 0; i < 16; ++i) ok &= (x[i].y == tag) & (x[i].w == tag);
        if (ok) break;
        ++spins;
        if ((spins & 255u) == 0u) {
            if (spins > SPIN_LIMIT || ld_agent32(status) != 0u) return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i)
        *reinterpret_cast<float2 *>(dst + r * LDA + own_col(i, c)) = make_float2(__uint_as_float(x[i].x), __uint_as_float(x[i].z));
    return true;
}

}  // namespace wrnn
