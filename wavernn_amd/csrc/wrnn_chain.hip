// wrnn_chain.hip -- the persistent WaveRNN loop kernel (MOL and 9-bit RAW, dense weights) for SMALL batches: <= 128 folded segments (what `auto` runs on
// it: one group per 64-CU cluster up to 64 segments = one utterance of BASELINE config 2 (N = 481 -> 12 segments; N = 1001 -> 24) or config 3's
// sentence (19); two groups per cluster up to 128), on MI355X (gfx950 / CDNA4).  Round 5.
//
// With one group of <= 16 segments per cluster nothing can be pipelined: a step of reference models/fatchord_version.py:201-241 IS the
// latency of its chain x_{t-1} -> h1 -> h2 -> fc1 -> fc2 -> sample.  wrnn_duo.hip (two workgroups per CU, four roles, built for throughput:
// 12.2 us per step with one group in flight) pays for that chain four cross-XCD hops and the round trip ih -> hh -> ih of every gh.  This
// kernel is wrnn_sparse.hip's single-stream form with dense stages:
//
//   * up to 4 clusters of 64 CUs (two XCDs), ONE workgroup of 4 waves per CU (512 registers per lane, every weight register-resident),
//     ONE group per cluster.  Workgroup J of the cluster's first XCD owns units [16 J, 16 J + 16) of rnn1 -- its rows of W_ih AND of
//     W_hh: gh(t + 1) = W_hh . h(t) + b_hh stays in the registers of the thread that needs it, no exchange --, forms the I-layer
//     conditioning cI of those rows (wave 0; as wrnn_duo.hip, incl. wrnn_options.mel_stage) and multiplies cI through W_ih ahead of
//     time: when x_{t-1} arrives only the x u1 term and the cell are left.  Workgroup J of the second XCD owns the same units of rnn2 AND
//     rows [16 J, 16 J + 16) of fc1 and of fc2: x2 -> y1 -> y2 never leave that XCD's L2 (plain stores; two cross-XCD hops per step are
//     left: x1 and y2).  fc3 + the mixture-of-logistics sampling (utils/distribution.py:87-123) run on rnn1's workgroup 0, which waits for
//     the chain anyway: x_t reaches the rnn1 workgroups through their own L2.
//   * a stage = this wave's 8 fragments of the layer (16-byte sc1 loads, the sentinel is the arrival flag) -> 96 or 32 MFMAs (K split over
//     the 4 waves) -> partial tiles through LDS in the accumulators' order, one LDS-only barrier -> the pointwise half of thread (unit,
//     segment) -> publish: wrnn_duo.hip's stage with nothing pipelined around it.
//   * exchange: wrnn_ring.h's buffer geometry; sentinel layers h1 x1 (re-armed two steps ahead by the wave that publishes them, behind its
//     poll of x_{t-1}: that x needed everything of step t - 1) and h2 x2 y1 y2 (behind the poll of y1(t)), drained at the top of the next
//     step; cI without a sentinel inside a launch (formed at the end of step t for step t + 2, drained at the top of step t + 1, read
//     behind the poll of h1(t + 1)); x_t as tagged 8-byte words in two entries.  The rules are wrnn_sparse.hip's (model:
//     tests/test_sparse_exchange_model.py); what differs is who publishes what (modelled with 1-4 slots and the RAW form in tests/test_chain_exchange_model.py).
//   * 9-bit RAW (fatchord_version.py:231-237): fc3 has 512 rows -- rnn1's workgroup J (idle while the chain runs through rnn2) owns rows [16 J, 16 J + 16): one
//     more stage (32 MFMAs on y2) and one more same-XCD hop (the 512 logits, layer 16) in front of the sampling, which is wrnn_duo.hip's: softmax ->
//     Categorical (renormalise) -> argmax(p / q) in the reference's operation order, on rnn1's workgroups 4 s .. 4 s + 3 for slot s, one segment per wave; the GRU cells use the library exp / tanh.
//     Bit-identical to wrnn_duo_kernel's RAW output (same stage arithmetic), i.e. to the reference on every fixture.
//   * conditioning slabs, state between launches (4 floats per (unit, segment)), step-range continuation: as wrnn_duo.hip / wrnn_sparse.hip.
//   * TWO (up to four) groups per cluster (wrnn_options.depth; `auto`: 65 .. 128 segments): every stage becomes a loop over the cluster's slots --
//     gates(0) gates(1) | fc1(0) fc1(1) | fc2(0) .. | gh(0) .. -- so one slot's hops fly under the other slot's stages; the per-slot state of a
//     thread (h, gh, W_ih . cI, its residual word: 8 floats) lives in LDS, the stage code exists once; rnn1's workgroup i samples slot i, wave w
//     forms the cI rows of slots w, w + 4.  Measured (profiles/r05i_probe_chain_depths.json): 13.8 us per step for 128 segments (wrnn_duo_kernel:
//     16.9); with three / four slots the single instruction stream per SIMD loses to the duo kernel's two (21.8 / 27.8 vs 19.8 / 23.5 us), and
//     so does moving the fc1 rows to the rnn1 workgroups to balance the MFMA count of the two XCDs (15.7 us at two slots: two more cross-XCD hops).
// Summation order per output: wrnn_duo.hip's (k ascending within a wave's quarter of K, the four quarters added in wave order).
#include <type_traits>

#include "wrnn_ring.h"

namespace wrnn {

constexpr int CHCL = 4;                      // clusters of 64 CUs
constexpr int CHWG = 64;                     // workgroups per cluster (one per CU): 32 x rnn1, 32 x rnn2 (+ fc rows)
constexpr int CHMAXG = 4;                    // slots (groups in flight) per cluster
constexpr int CHPART = 2 * NW * 3 * 256;     // two ping-pong sets of [wave][tile 0..2][lane][4]
constexpr int CHSTATE_SLOT = NT * 4 + SEG;   // saved state of a workgroup's slot: per thread {h, gh_r, gh_z, gh_n}, then x_{t1-1} (rnn1)
static_assert(CHCL * CHWG <= XCC_WORDS, "placement table");
static_assert(CHCL * CHMAXG <= LMAXG * MAXCL, "one exchange-buffer region per (slot, cluster)");
static_assert(SEG * LDC <= 2 * XT, "RAW logits fit the fc3 region");

struct ChLds {
    int off_seg, off_geo, off_st, off_part, off_misc, off_prof, off_b3, off_y2t, off_f3, total;
};
__host__ __device__ inline ChLds ch_lds(int G)
{
    ChLds l;
    int o = 0;
    l.off_seg = o;  o += G * 64;             // ints per slot: 16 positions | 16 limits | 16 table-row bases of this slab | 16 mel offsets
    l.off_geo = o;  o += 2 * CHMAXG;         // ints: first segment (in the call) and segment count of every slot
    l.off_st = o;   o += G * 8 * NT;         // per slot and thread: h, gh_r, gh_z, gh_n, (rnn1) W_ih . cI r z n, own cI word
    l.off_part = o; o += CHPART;
    l.off_misc = o; o += 64;                 // placement table of the cluster (ints)
    l.off_prof = o; o += 64;                 // [32] u64 phase clocks (profiling builds)
    l.off_b3 = o;   o += 32;                 // sampling workgroups: fc3.bias
    l.off_y2t = o;  o += 256;                // rnn2 (MOL, fc3 folded into the fc2 stage): the workgroup's y2 tile [row][segment]
    o = (o + 3) & ~3;
    l.off_f3 = o;   o += 2 * XT;             // ... MOL: fc3 (30 x 512 = two 16-row tiles) in A-fragment order; RAW: the gathered logits [segment][class], stride LDC
    l.total = o;
    return l;
}

typedef unsigned ch_u32x2 __attribute__((ext_vector_type(2)));
#define CHX(k)                                                                 \
    do {                                                                       \
        if (PROF && tid == 0) {                                                \
            const u64 now_ = __builtin_amdgcn_s_memtime();                     \
            PROFL[k] += now_ - plast;                                          \
            plast = now_;                                                      \
        }                                                                      \
    } while (0)

// ---------------------------------------------------------------------------------------------------------------------------------
// One workgroup: units [16 J, 16 J + 16) of rnn1 (LA) or of rnn2 + rows [16 J, 16 J + 16) of fc1 and of fc2.  Slot i of cluster cl = group cl + ncl i of the round = region i CHCL + cl of the exchange buffer.
// loc_a / loc_b: the cluster's rnn1 / rnn2 workgroups were all seen on one XCD.
// PROF (thread 0, shader clocks per segment of a step, summed over the slots, program order): rnn1: 0 drain + wait x_{t-1}, 1 cell + publish,
// 2 wait h1(t), 3 gh stage, 4 wait cI(t+1), 5 W_ih . cI stage, 6 cI(t+2) formed, 7 wait y2(t), 8 fc3 + sampling (RAW: the logits stages; the sampling is not clocked); rnn2:
// 0 drain + wait x1(t), 1 gate stage + cell + publish, 2 wait x2, 3 fc1, 4 wait y1, 5 fc2, 6 wait h2 (there), 7 gh stage; 15 = steps
// ---------------------------------------------------------------------------------------------------------------------------------
template <int MODE, bool LA, bool PROF>
__device__ __forceinline__ void ch_role(const LoopArgs &a, float *smem, const int cl, const int ncl, const int J, const bool loc_a, const bool loc_b)
{
    const int G = a.G;
    const ChLds L = ch_lds(G);
    float *PART = smem + L.off_part, *fc3b = smem + L.off_b3, *F3 = smem + L.off_f3, *ST = smem + L.off_st, *Y2T = smem + L.off_y2t;
    int *SEGT = reinterpret_cast<int *>(smem + L.off_seg), *GEO = reinterpret_cast<int *>(smem + L.off_geo);
    u64 *PROFL = reinterpret_cast<u64 *>(smem + L.off_prof);
    u64 plast = 0;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fi = lane & 15, kq = lane >> 4;
    const int kbase_lane = KCH * w + 4 * kq;
    const int pu = 4 * w + (tid & 3), pj = (tid >> 2) & 15;      // pointwise role: (unit 16 J + pu, segment pj)
    const int prow = LU * J + pu;
    const int T0 = a.t0, T1 = a.t1;
    unsigned *const status = a.status;
    const int hop = a.hop, resume = a.resume, Tall = a.T, Nall = a.Nall, noise_t0 = a.noise_t0, C = a.C;
    const unsigned magic = a.hop_magic;
    const int mshift = a.hop_shift;
    const int zrow = a.Nall * a.tab_fps;
    float *const outp = a.out, *const dbgl = a.dbg_logits;
    const float *const forcex = a.force_x, *const noise_pre = a.noise_pre, *const noise_raw = a.noise;
    const float *const mels_up = a.mels_up, *const aux_fr = a.aux_fr, *const mel_coef = a.mel_coef;
    const int mel_stage = a.mel_stage;
    const int NR = a.Btot, NGR = a.NG;
    constexpr bool HAS_FC1 = !LA;
    constexpr bool MOL = MODE == 1;
    float *const state_wg = a.state + ((size_t)(cl * CHWG + (LA ? 0 : 32) + J) * G) * CHSTATE_SLOT;

    // ---- weights: three gate tiles of W_ih and of W_hh; one tile of fc1 (HAS_FC1) and of fc2 (rnn2)
    float A_ih[3][AF], A_hh[3][AF], A_fc1[AF], A_fc2[AF];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        load_afrag(A_ih[g], LA ? a.w_ih1 : a.w_ih2, LA ? H : H + AUX, g * H + LU * J + fi, true, kbase_lane);
        load_afrag(A_hh[g], LA ? a.w_hh1 : a.w_hh2, H, g * H + LU * J + fi, true, kbase_lane);
    }
    if constexpr (HAS_FC1) load_afrag(A_fc1, a.fc1_w, H + AUX, LU * J + fi, true, kbase_lane);
    if constexpr (!LA) load_afrag(A_fc2, a.fc2_w, H + AUX, LU * J + fi, true, kbase_lane);
    float A_f3[AF];                                     // RAW, rnn1: fc3 rows (classes) [16 J, 16 J + 16)
    float A_f3m[2][AF];                                 // MOL, rnn1: both 16-row tiles of fc3 (30 x 512), this wave's quarter of K -- AGPRs (round 6; they were read from LDS MFMA by MFMA)
    if constexpr (LA && MOL) {
        if (a.tuning & 32) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int k = 0; k < AF; ++k) A_f3m[q][k] = 0.f;
        } else {
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const float4 v = *reinterpret_cast<const float4 *>(a.fc3f + q * XT + frag_off(w, 0, lane) + 256 * r);
                    A_f3m[q][4 * r + 0] = v.x; A_f3m[q][4 * r + 1] = v.y; A_f3m[q][4 * r + 2] = v.z; A_f3m[q][4 * r + 3] = v.w;
                }
        }
    }
    // MOL, rnn2 (round 6): fc3 FOLDED into the fc2 stage -- this workgroup multiplies ITS 16 rows of y2 through fc3's columns [16 J, 16 J + 16) (two 16-row
    // tiles x 4 MFMAs, wave 0) and publishes the two partial logit tiles instead of y2; the sampling workgroup adds the 32 workgroups' tiles instead of running
    // 64 MFMAs per wave on its slot's chain.  A3[q][i]: row 16 q + fi of fc3, column 16 J + 4 i + kq (K-slot kq of MFMA i <-> y2 row 4 i + kq).
    bool fold3 = false;                                 // (set below: one slot only -- with two the rnn2 workgroups are the busy side: 13.84 vs 13.57 us per step; A/B: tuning bit 4 = off)
    float A3[2][4];
    if constexpr (!LA && MOL) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i) A3[q][i] = (16 * q + fi < C) ? a.fc3_w[(size_t)(16 * q + fi) * H + LU * J + 4 * i + kq] : 0.f;
    }
    float b3 = 0.f;
    if constexpr (LA && !MOL) {
        load_afrag(A_f3, a.fc3_w, H, LU * J + fi, true, kbase_lane);
        b3 = a.fc3_b[prow];
    }
    // constants of the pointwise role: rnn1: b_ih1, u1 = W_ih1 . w0 (the x_{t-1} term), w0 (rnn2's b_ih2 is inside c2f); b_hh
    float cb_r = 0.f, cb_z = 0.f, cb_n = 0.f, ux_r = 0.f, ux_z = 0.f, ux_n = 0.f, w0o = 0.f;
    if constexpr (LA) {
        cb_r = a.b_ih1[prow]; cb_z = a.b_ih1[H + prow]; cb_n = a.b_ih1[2 * H + prow];
        ux_r = a.u1[prow]; ux_z = a.u1[H + prow]; ux_n = a.u1[2 * H + prow];
        w0o = a.I_w0[prow];
    }
    const float *bhp = LA ? a.b_hh1 : a.b_hh2;
    const float bh_r = bhp[prow], bh_z = bhp[H + prow], bh_n = bhp[2 * H + prow];
    CondTile ct;
    if constexpr (LA) cond_tile_init(ct, a.I_cT, a.I_b, J, lane);
    for (int q = tid; q < L.off_f3; q += NT) smem[q] = 0.f;
    __syncthreads();
    int nact = 0;
    for (int i = 0; i < G; ++i) {
        const int g = cl + ncl * i;
        if (g >= NGR) break;
        nact = i + 1;
        const int b0 = (int)(((long)g * NR) / NGR), nb = (int)(((long)(g + 1) * NR) / NGR) - b0;
        if (tid == 0) { GEO[2 * i] = a.rb0 + b0; GEO[2 * i + 1] = nb; }
        if (tid < SEG) {
            const int sc = a.rb0 + b0 + (tid < nb ? tid : nb - 1);
            const int pos = a.seg_pos[sc];
            SEGT[i * 64 + tid] = pos;
            SEGT[i * 64 + SEG + tid] = a.seg_lim[sc];
            SEGT[i * 64 + 2 * SEG + tid] = sc * a.tab_fps - (pos + a.tab_t0) / a.hop;
            SEGT[i * 64 + 3 * SEG + tid] = a.mel_stage ? a.seg_moff[sc] : 0;
        }
        // per-slot state of the thread: h, gh (fatchord_version.py:194-195: h = 0, so gh(0) = b_hh), or what the previous launch saved
        float4 sv = make_float4(0.f, bh_r, bh_z, bh_n);
        if (resume) sv = *reinterpret_cast<const float4 *>(state_wg + (size_t)i * CHSTATE_SLOT + tid * 4);
        ST[(i * 8 + 0) * NT + tid] = sv.x; ST[(i * 8 + 1) * NT + tid] = sv.y; ST[(i * 8 + 2) * NT + tid] = sv.z; ST[(i * 8 + 3) * NT + tid] = sv.w;
    }
    fold3 = MOL && nact == 1 && (a.tuning & 16) == 0;
    // MOL: rnn1's workgroup i runs fc3 + the sampling of slot i.  RAW: FOUR workgroups per slot -- workgroup J samples segments 4 (J & 3) .. + 3 of slot
    // J >> 2, one segment per wave (the 512-class softmax of 16 segments on one workgroup took ~8 us of the chain: profiles/r05k_probe_raw_chain.json)
    const bool sampler = LA && J < (MOL ? nact : 4 * nact);
    const bool f3_lds = (a.tuning & 32) != 0;           // A/B: fc3's tiles read from LDS, as round 5
    if (MOL && sampler) {                               // fc3 -> LDS (fragment order as in the pack)
        for (int q = tid; q < 2 * XT / 4; q += NT) reinterpret_cast<float4 *>(F3)[q] = reinterpret_cast<const float4 *>(a.fc3f)[q];
        if (tid < 32) fc3b[tid] = tid < 30 ? a.fc3_b[tid] : 0.f;
    }
    float *const LGT = F3;                              // RAW: the gathered logits of the slot being sampled
    __syncthreads();

    const __amdgpu_buffer_rsrc_t xrs = make_rsrc(a.xbuf, (unsigned)(DXBUF_FLOATS * 4));
    const __amdgpu_buffer_rsrc_t crs = make_rsrc(a.c2f, 0x7FFFF000u);
    const __amdgpu_buffer_rsrc_t nrs = make_rsrc(noise_pre, 0x7FFFF000u);
    const __amdgpu_buffer_rsrc_t f1rs = make_rsrc(a.c3f, 0x7FFFF000u), f2rs = make_rsrc(a.c4f, 0x7FFFF000u);
    const int voff_frag = frag_off(w, 0, lane) * 4;      // this lane's first fragment of a layer (bytes)
    const int voff_own = (((J * 64) + 16 * w + pj) * 4 + (tid & 3)) * 4;      // the layer word of (unit 16 J + pu, segment pj) = its publish position
    // (x1 and y2 cross the XCDs -- rnn2 / the samplers sit on the other one -- and are written through; x2 and y1 stay in rnn2's L2)
    const bool loc_x2 = loc_b, loc_y1 = loc_b;

    bool dead = false;
    int pp = 0;
    int t = T0;
    auto cbase_of = [&](int i) { return (i * CHCL + cl) * DSLOTB; };
    auto nb_of = [&](int i) { return GEO[2 * i + 1]; };

    // one stage: the layer at byte offset `so` -> three (NTL = 3) or one tile(s) of this workgroup x the slot's 16 segments; the sums of the
    // thread's (unit, segment) come back in s0..s2, `ownw` = the thread's own word of that layer (the residual input of the gate stages)
    auto stage = [&](auto NTC, auto AGC, const float (&A0)[AF], const float (&A1)[AF], const float (&A2)[AF], int so, int nb, unsigned code, int ts, int px_wait,
                     float &s0, float &s1, float &s2, unsigned &ownw, bool want_own) {
        constexpr int NTL = decltype(NTC)::value;
        constexpr bool AG = decltype(AGC)::value != 0;  // the tiles live in AGPRs (wrnn_ring.h: mfma3s_ag)
        const bool live = fi < nb, plive = pj < nb;
        u32x4 x[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) x[r] = __builtin_amdgcn_raw_buffer_load_b128(xrs, voff_frag + r * 1024, so, 16 /* sc1 */);
        if (want_own) ownw = __builtin_amdgcn_raw_buffer_load_b32(xrs, voff_own, so, 16 /* sc1 */);
        if (__builtin_expect(!frag_there(x, live), 0))
            wait_for([&] { return frag_there(x, live); },
                     [&] {
#pragma unroll
                         for (int r = 0; r < 8; ++r) x[r] = __builtin_amdgcn_raw_buffer_load_b128(xrs, voff_frag + r * 1024, so, 16 /* sc1 */);
                         // (the own word with every re-load: asked for once, beside the FIRST look, it is the sentinel whenever the layer is waited for, and the
                         // poll below then cost a whole round trip behind the arrival: 10.5 -> 9.75 us per step for one utterance, profiles/r06af_chain_ab.log.
                         // Measured with it, no gain: TWO requests in flight half a round trip apart -- 9.76 vs 9.74; 15.7 vs 14.3 with two slots.)
                         if (want_own) ownw = __builtin_amdgcn_raw_buffer_load_b32(xrs, voff_own, so, 16 /* sc1 */);
                     },
                     status, dead, code, ts);
        if (want_own && __builtin_expect(__any(plive && ownw == SENT), 0))
            wait_for([&] { return !__any(plive && ownw == SENT); }, [&] { ownw = __builtin_amdgcn_raw_buffer_load_b32(xrs, voff_own, so, 16 /* sc1 */); },
                     status, dead, code | 0x80u, ts);
        CHX(px_wait);
        float b[32];
        frag_to_b(x, b);
        float *PW = PART + pp * (NW * 3 * 256);
        if constexpr (NTL == 3) {
            f32x4 o0, o1, o2;
            if constexpr (AG) mfma3s_ag(A0, A1, A2, b, o0, o1, o2);
            else mfma3s(A0, A1, A2, b, o0, o1, o2);
            put_partial<3>(PW, w, 0, lane, o0);
            put_partial<3>(PW, w, 1, lane, o1);
            put_partial<3>(PW, w, 2, lane, o2);
        } else if constexpr (AG) put_partial<3>(PW, w, 0, lane, mfma1_ag(A0, b));
        else put_partial<3>(PW, w, 0, lane, mfma1(A0, b));
        lds_barrier();
        s0 = get_partial<3>(PW, 0, pu, pj);
        if constexpr (NTL == 3) { s1 = get_partial<3>(PW, 1, pu, pj); s2 = get_partial<3>(PW, 2, pu, pj); }
        pp ^= 1;
    };
    using N1 = std::integral_constant<int, 1>;
    using N3 = std::integral_constant<int, 3>;
    using VG = std::integral_constant<int, 0>;          // weight tiles in VGPRs: W_ih, fc1 (128 registers) | in AGPRs: W_hh, fc2 / RAW's fc3 rows (128)
    using AGR = std::integral_constant<int, 1>;
    // re-arm this wave's quarter of the workgroup's 1 KB block of `layer` of slot i in entry (t + 2) % 4 (lanes 16 q .. 16 q + 15)
    auto rearm1 = [&](int i, int layer, int q, bool local) {
        if (kq == q) {
            const u32x4 sv = {SENT, SENT, SENT, SENT};
            const int vo = layer * DLAYERB + J * 1024 + w * 256 + fi * 16, so = cbase_of(i) + ((t + DAHEAD_IH) & (DRING - 1)) * XTB;
            if (local) __builtin_amdgcn_raw_buffer_store_b128(sv, xrs, vo, so, 0);
            else __builtin_amdgcn_raw_buffer_store_b128(sv, xrs, vo, so, 16 /* sc1 */);
        }
    };
    // fc1 + relu (:216-218) -> y1 of slot i (this workgroup's 16 rows)
    auto fc1_stage = [&](int i) {
        if constexpr (HAS_FC1) {
            const int nb = nb_of(i), sb = cbase_of(i) + (t & (DRING - 1)) * XTB;
            const int fr = table_row(SEGT[i * 64 + pj] + t, SEGT[i * 64 + SEG + pj], SEGT[i * 64 + 2 * SEG + pj], magic, mshift, hop, zrow);
            const float cv = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(f1rs, (fr * H + prow) * 4, 0, 0));
            float s0, s1, s2;
            unsigned dummy = 0u;
            stage(N1{}, VG{}, A_fc1, A_fc1, A_fc1, sb + 6 * DLAYERB, nb, 0x801u, t, 2, s0, s1, s2, dummy, false);
            publish4l(xrs, sb + 2 * DLAYERB + J * 1024, tid, fmaxf(s0 + cv, 0.f), pj < nb, loc_y1);
            CHX(3);
        }
    };
    // gh(t + 1) = W_hh . h(t) + b_hh of slot i: stays with the thread
    auto gh_stage = [&](int i) {
        float s0, s1, s2;
        unsigned dummy = 0u;
        stage(N3{}, AGR{}, A_hh[0], A_hh[1], A_hh[2], cbase_of(i) + (LA ? 0 : 1) * DLAYERB + (t & (DRING - 1)) * XTB, nb_of(i), 0x840u | (LA ? 0u : 8u), t, LA ? 2 : 6, s0, s1, s2,
              dummy, false);
        ST[(i * 8 + 1) * NT + tid] = s0 + bh_r; ST[(i * 8 + 2) * NT + tid] = s1 + bh_z; ST[(i * 8 + 3) * NT + tid] = s2 + bh_n;
        CHX(LA ? 3 : 7);
    };

    if (PROF && tid == 0) plast = __builtin_amdgcn_s_memtime();
    if constexpr (LA) {
        // ---------------- rnn1 ----------------
        auto cond_step = [&](int tt) {                  // wave w: cI(tt) of the workgroup's 16 rows for slots w, w + 4 (fatchord_version.py:203-209 without the x_{t-1} column)
#pragma unroll 1
            for (int i = w; i < nact; i += NW) {
                const int p = SEGT[i * 64 + fi] + tt;
                const bool valid = fi < nb_of(i) && p < SEGT[i * 64 + SEG + fi];
                const int fr = magic ? (int)(__umulhi((unsigned)p, magic) >> mshift) : p / hop;
                f32x4 v;
                if (mel_stage) {
                    const int j = p + SEGT[i * 64 + 3 * SEG + fi];
                    const int row = j / LAST_SCALE;
                    v = cond_tile_rows(ct, mels_up + (size_t)(row - 1) * MEL, mel_coef + 3 * (j - row * LAST_SCALE), aux_fr + (size_t)fr * (4 * AUX), valid, lane);
                } else v = cond_tile(ct, mels_up + (size_t)p * MEL, aux_fr + (size_t)fr * (4 * AUX), valid, lane);
                const u32x4 q = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
                const int so = cbase_of(i) + 4 * DLAYERB + (tt & (DRING - 1)) * XTB;
                if (loc_a) __builtin_amdgcn_raw_buffer_store_b128(q, xrs, J * 1024 + lane * 16, so, 0);
                else __builtin_amdgcn_raw_buffer_store_b128(q, xrs, J * 1024 + lane * 16, so, 16 /* sc1 */);
            }
        };
        auto front = [&](int i, int ts) {               // W_ih1 . cI(ts) of slot i + the thread's own cI word (xi - w0 x)
            unsigned ow = 0u;
            float s0, s1, s2;
            stage(N3{}, VG{}, A_ih[0], A_ih[1], A_ih[2], cbase_of(i) + 4 * DLAYERB + (ts & (DRING - 1)) * XTB, nb_of(i), 0x820u, ts, 4, s0, s1, s2, ow, true);
            ST[(i * 8 + 4) * NT + tid] = s0; ST[(i * 8 + 5) * NT + tid] = s1; ST[(i * 8 + 6) * NT + tid] = s2; ST[(i * 8 + 7) * NT + tid] = __uint_as_float(ow);
            CHX(5);
        };
        auto back = [&](int i) {                        // the chain: sampling of slot i -> here
            const int nb = nb_of(i), cb = cbase_of(i);
            const bool plive = pj < nb;
            float xv = 0.f;
            if (t > T0) {                               // x_{t-1}: a tagged word {x, tag = t} from the slot's sampling workgroup
                const int sx = cb + 7 * DLAYERB + ((t - 1) & 1) * XTB;
                ch_u32x2 xq = __builtin_amdgcn_raw_buffer_load_b64(xrs, pj * 8, sx, 16 /* sc1 */);
                if (__builtin_expect(__any(plive && xq.y != (unsigned)t), 0))
                    wait_for([&] { return !__any(plive && xq.y != (unsigned)t); }, [&] { xq = __builtin_amdgcn_raw_buffer_load_b64(xrs, pj * 8, sx, 16 /* sc1 */); },
                             status, dead, 0x830u, t);
                xv = __uint_as_float(xq.x);
            } else if (resume) xv = state_wg[(size_t)i * CHSTATE_SLOT + NT * 4 + pj];
            CHX(0);
            // the GRU cell of (unit 16 J + pu, segment pj) (ATen gru_cell; hardware exp / rcp as the duo kernel's MoL path) -> x1 (to rnn2's XCD), h1
            const float gir = ST[(i * 8 + 4) * NT + tid] + fmaf(xv, ux_r, cb_r), giz = ST[(i * 8 + 5) * NT + tid] + fmaf(xv, ux_z, cb_z),
                        gin = ST[(i * 8 + 6) * NT + tid] + fmaf(xv, ux_n, cb_n);
            const float xin = fmaf(w0o, xv, ST[(i * 8 + 7) * NT + tid]);      // xi of the owned unit (:208-209)
            // MoL: hardware exp / rcp (inside the 1e-5 tolerance); RAW (class indices compared bit for bit): the library forms
            const float h = MOL ? gru_update_fast(gir, giz, gin, ST[(i * 8 + 1) * NT + tid], ST[(i * 8 + 2) * NT + tid], ST[(i * 8 + 3) * NT + tid], ST[(i * 8 + 0) * NT + tid])
                                : gru_update(gir, giz, gin, ST[(i * 8 + 1) * NT + tid], ST[(i * 8 + 2) * NT + tid], ST[(i * 8 + 3) * NT + tid], ST[(i * 8 + 0) * NT + tid]);
            ST[(i * 8 + 0) * NT + tid] = h;
            const int sb = cb + (t & (DRING - 1)) * XTB;
            publish4l(xrs, sb + 5 * DLAYERB + J * 1024, tid, xin + h, plive, false);      // x1 = xi + h1 (:212)
            publish4l(xrs, sb + 0 * DLAYERB + J * 1024, tid, h, plive, loc_a);
            // ring hygiene: behind the poll of x_{t-1}, which needed everything of step t - 1: every reader is past the data of step t - 2
            rearm1(i, 0, 0, loc_a);
            rearm1(i, 5, 1, false);
            if constexpr (!MOL) rearm1(i, 16, 2, loc_a);
            CHX(1);
        };
        auto sample = [&](int i) {
            const int nb = nb_of(i), b0g = GEO[2 * i], cb = cbase_of(i);
            const bool live = fi < nb, plive = pj < nb;
            const int sb = cb + (t & (DRING - 1)) * XTB;
            // (the step's noise FIRST -- in front of the y2 requests and of every re-load of the poll: behind them the two words came back from L2 / HBM
            // on the slot's chain -- round 6)
            const int su = tid >> 4, sm = tid & 15;     // sampling role: 16-lane row = segment su, lane sm = mixture
            const int suc = su < nb ? su : nb - 1;
            const int nvo = (int)(((size_t)(t - noise_t0) * 11 * Nall) * 4);
            const float nz0 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(nrs, ((b0g + suc) * 10 + (sm < 10 ? sm : 9)) * 4, nvo, 0));
            const float nz1 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(nrs, (10 * Nall + b0g + suc) * 4, nvo, 0));
            // x[r] = this wave's fragments of y2 -- or, fc3 folded into the fc2 stage (fold3): the tile-0 partial logits of workgroups 8 w + r (rows 4 kq + e of
            // segment fi: the accumulators' layout; the same addresses), x1[r] = their tile-1 partial logits (layer 16)
            u32x4 x[8], x1[8];
            auto ask = [&] {
#pragma unroll
                for (int r = 0; r < 8; ++r) x[r] = __builtin_amdgcn_raw_buffer_load_b128(xrs, voff_frag + r * 1024, sb + 3 * DLAYERB, 16 /* sc1 */);
                if (fold3) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) x1[r] = __builtin_amdgcn_raw_buffer_load_b128(xrs, voff_frag + r * 1024, sb + 16 * DLAYERB, 16 /* sc1 */);
                } else {
#pragma unroll
                    for (int r = 0; r < 8; ++r) x1[r] = u32x4{0u, 0u, 0u, 0u};
                }
            };
            ask();
            if (__builtin_expect(!(frag_there(x, live) && frag_there(x1, live)), 0))
                wait_for([&] { return frag_there(x, live) && frag_there(x1, live); }, ask, status, dead, 0x850u, t);
            CHX(7);
            float *PW = PART + pp * (NW * 3 * 256);
            if (fold3) {
                f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    s0 += f32x4{__uint_as_float(x[r].x), __uint_as_float(x[r].y), __uint_as_float(x[r].z), __uint_as_float(x[r].w)};
                    s1 += f32x4{__uint_as_float(x1[r].x), __uint_as_float(x1[r].y), __uint_as_float(x1[r].z), __uint_as_float(x1[r].w)};
                }
                put_partial<3>(PW, w, 0, lane, s0);
                put_partial<3>(PW, w, 1, lane, s1);
            }
            float b[32];
            frag_to_b(x, b);
            if (fold3) {
            } else if (f3_lds) {
                put_partial<3>(PW, w, 0, lane, mfma1_lds(F3 + frag_off(w, 0, lane), b));
                put_partial<3>(PW, w, 1, lane, mfma1_lds(F3 + XT + frag_off(w, 0, lane), b));
            } else {
                put_partial<3>(PW, w, 0, lane, mfma1_ag(A_f3m[0], b));
                put_partial<3>(PW, w, 1, lane, mfma1_ag(A_f3m[1], b));
            }
            lds_barrier();
            if (dbgl && plive) {                        // test hook: the 30 logits of every segment
                dbgl[((size_t)t * Nall + b0g + pj) * C + pu] = get_partial<3>(PW, 0, pu, pj) + fc3b[pu];
                if (pu < 14) dbgl[((size_t)t * Nall + b0g + pj) * C + 16 + pu] = get_partial<3>(PW, 1, pu, pj) + fc3b[16 + pu];
            }
            {   // utils/distribution.py:102-121: lane sm < 10 of a 16-lane row sums the partial tiles of ITS mixture logit straight from LDS,
                // Gumbel-max over the row, then lane 0 fetches mean and log-scale of the winner
                float best = (sm < 10) ? mol_gumbel_pre(get_partial<3>(PW, 0, sm < 10 ? sm : 0, su) + fc3b[sm < 10 ? sm : 0], nz0) : -INFINITY;
                int bidx = sm;
                argmax_row16(best, bidx);
                if (sm == 0 && su < nb) {
                    const float mean = get_partial<3>(PW, (10 + bidx) >> 4, (10 + bidx) & 15, su) + fc3b[10 + bidx];
                    const float ls = get_partial<3>(PW, (20 + bidx) >> 4, (20 + bidx) & 15, su) + fc3b[20 + bidx];
                    float xv = mol_sample_pre(mean, ls, nz1);
                    outp[(size_t)(b0g + su) * Tall + t] = xv;
                    if (forcex) xv = forcex[(size_t)(b0g + su) * Tall + t];
                    const ch_u32x2 q = {__float_as_uint(xv), (unsigned)t + 1u};       // one 8-byte word {x_t, tag}: its own flag, two entries, no re-arm
                    if (loc_a) __builtin_amdgcn_raw_buffer_store_b64(q, xrs, su * 8, cb + 7 * DLAYERB + (t & 1) * XTB, 0);
                    else __builtin_amdgcn_raw_buffer_store_b64(q, xrs, su * 8, cb + 7 * DLAYERB + (t & 1) * XTB, 16 /* sc1 */);
                }
            }
            pp ^= 1;
            CHX(8);
        };

        // ---- RAW: the workgroup's 16 logit rows of slot i (fc3 on y2, :223) -> layer 16, read by the slot's sampling workgroup
        auto logits = [&](int i) {
            const int nb = nb_of(i), sb = cbase_of(i) + (t & (DRING - 1)) * XTB;
            float s0, s1, s2;
            unsigned dummy = 0u;
            stage(N1{}, AGR{}, A_f3, A_f3, A_f3, sb + 3 * DLAYERB, nb, 0x851u, t, 7, s0, s1, s2, dummy, false);
            publish4l(xrs, sb + 16 * DLAYERB + J * 1024, tid, s0 + b3, pj < nb, loc_a);
            CHX(8);
        };
        // ---- RAW: fatchord_version.py:231-237: softmax -> Categorical (renormalise) -> argmax(p / q) -- wrnn_duo.hip's code: one segment per wave;
        //      per segment the operation order is the reference's (class indices compared bit for bit); wave w of sampling workgroup J: segment 4 (J & 3) + w
        auto sample_raw = [&](int i) {
            const int nb = nb_of(i), b0g = GEO[2 * i], cb = cbase_of(i);
            const bool live = fi < nb;
            const int so = cb + 16 * DLAYERB + (t & (DRING - 1)) * XTB;
            u32x4 x[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) x[r] = __builtin_amdgcn_raw_buffer_load_b128(xrs, voff_frag + r * 1024, so, 16 /* sc1 */);
            if (__builtin_expect(!frag_there(x, live), 0))
                wait_for([&] { return frag_there(x, live); },
                         [&] {
#pragma unroll
                             for (int r = 0; r < 8; ++r) x[r] = __builtin_amdgcn_raw_buffer_load_b128(xrs, voff_frag + r * 1024, so, 16 /* sc1 */);
                         },
                         status, dead, 0x852u, t);
            {
                float *lp = LGT + fi * LDC + kbase_lane;
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    *reinterpret_cast<float4 *>(lp + 16 * r) = make_float4(__uint_as_float(x[r].x), __uint_as_float(x[r].y), __uint_as_float(x[r].z), __uint_as_float(x[r].w));
            }
            lds_barrier();
            {
                constexpr int NS = 1;
                const int h0 = 4 * (J & 3) + w;         // this wave's segment
                float qn[NS][8], lg[NS][8], mx[NS], sum[NS], sum2[NS], best[NS];
                int bidx[NS];
                const size_t tn = (size_t)(t - noise_t0);
#pragma unroll
                for (int s4 = 0; s4 < NS; ++s4) {
                    const int sjc = (h0 + s4 < nb) ? h0 + s4 : nb - 1;
#pragma unroll
                    for (int e = 0; e < 8; ++e) qn[s4][e] = noise_raw[(tn * Nall + b0g + sjc) * C + lane + 64 * e];
                    mx[s4] = -INFINITY;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        lg[s4][e] = LGT[sjc * LDC + lane + 64 * e];
                        mx[s4] = fmaxf(mx[s4], lg[s4][e]);
                    }
                }
                if (dbgl) {
#pragma unroll
                    for (int s4 = 0; s4 < NS; ++s4)
                        if (h0 + s4 < nb)
#pragma unroll
                            for (int e = 0; e < 8; ++e) dbgl[((size_t)t * Nall + b0g + h0 + s4) * C + lane + 64 * e] = lg[s4][e];
                }
#pragma unroll
                for (int s4 = 0; s4 < NS; ++s4) mx[s4] = wave_max64(mx[s4]);      // (xor_pair butterflies, wrnn_device.h: == the __shfl_xor form bit for bit)
#pragma unroll
                for (int s4 = 0; s4 < NS; ++s4) {
                    sum[s4] = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) { lg[s4][e] = expf(lg[s4][e] - mx[s4]); sum[s4] += lg[s4][e]; }
                }
#pragma unroll
                for (int s4 = 0; s4 < NS; ++s4) sum[s4] = wave_sum64(sum[s4]);
#pragma unroll
                for (int s4 = 0; s4 < NS; ++s4) {
                    sum2[s4] = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) { lg[s4][e] = lg[s4][e] / sum[s4]; sum2[s4] += lg[s4][e]; }
                }
#pragma unroll
                for (int s4 = 0; s4 < NS; ++s4) sum2[s4] = wave_sum64(sum2[s4]);
#pragma unroll
                for (int s4 = 0; s4 < NS; ++s4) {
                    best[s4] = -INFINITY;
                    bidx[s4] = 0;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float rr = (lg[s4][e] / sum2[s4]) / qn[s4][e];
                        if (rr > best[s4]) { best[s4] = rr; bidx[s4] = lane + 64 * e; }
                    }
                }
#pragma unroll
                for (int s4 = 0; s4 < NS; ++s4) wave_argmax64(best[s4], bidx[s4]);
                if (lane == 0) {
#pragma unroll
                    for (int s4 = 0; s4 < NS; ++s4) {
                        const int sj = h0 + s4;
                        if (sj < nb) {
                            float xv = 2.f * (float)bidx[s4] / ((float)C - 1.f) - 1.f;
                            outp[(size_t)(b0g + sj) * Tall + t] = xv;
                            if (forcex) xv = forcex[(size_t)(b0g + sj) * Tall + t];
                            const ch_u32x2 q = {__float_as_uint(xv), (unsigned)t + 1u};
                            if (loc_a) __builtin_amdgcn_raw_buffer_store_b64(q, xrs, sj * 8, cb + 7 * DLAYERB + (t & 1) * XTB, 0);
                            else __builtin_amdgcn_raw_buffer_store_b64(q, xrs, sj * 8, cb + 7 * DLAYERB + (t & 1) * XTB, 16 /* sc1 */);
                        }
                    }
                }
            }
            lds_barrier();                              // LGT is read by every wave before the next step's gather overwrites it
        };

        cond_step(T0);                                  // the two steps a launch starts with; every later cI is formed at the end of step t for t + 2
        if (T0 + 1 < T1) cond_step(T0 + 1);
#pragma unroll 1
        for (int i = 0; i < nact; ++i) front(i, T0);
        for (; t < T1; ++t) {
            if (PROF && tid == 0) PROFL[15] += 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // last step's re-arm stores and the cI formed at its end are out before this step publishes
#pragma unroll 1
            for (int i = 0; i < nact; ++i) back(i);
#pragma unroll 1
            for (int i = 0; i < nact; ++i) gh_stage(i); // (needs h1(t) of every rnn1 workgroup: one hop behind the publications above)
            if (t + 1 < T1) {
#pragma unroll 1
                for (int i = 0; i < nact; ++i) front(i, t + 1);
            }
            if (t + 2 < T1) { cond_step(t + 2); CHX(6); }
            if constexpr (MOL) {
                if (sampler) sample(J);
            } else {
#pragma unroll 1
                for (int i = 0; i < nact; ++i) logits(i);
                if (sampler) sample_raw(J >> 2);
            }
        }
        // ---- what the next launch needs: x_{T1-1} of every slot; the sentinel in the cI entries of steps T1 and T1 + 1 (its first two steps are polled)
#pragma unroll 1
        for (int i = 0; i < nact; ++i) {
            const int nb = nb_of(i);
            const int sx = cbase_of(i) + 7 * DLAYERB + ((T1 - 1) & 1) * XTB;
            ch_u32x2 xq = __builtin_amdgcn_raw_buffer_load_b64(xrs, fi * 8, sx, 16 /* sc1 */);
            wait_for([&] { return !__any(fi < nb && xq.y != (unsigned)T1); }, [&] { xq = __builtin_amdgcn_raw_buffer_load_b64(xrs, fi * 8, sx, 16 /* sc1 */); },
                     status, dead, 0x861u, T1);
            if (tid < SEG) state_wg[(size_t)i * CHSTATE_SLOT + NT * 4 + tid] = fi < nb ? __uint_as_float(xq.x) : 0.f;
        }
#pragma unroll 1
        for (int i = w; i < nact; i += NW) {
            const u32x4 q = {SENT, SENT, SENT, SENT};
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int so = cbase_of(i) + 4 * DLAYERB + ((T1 + e) & (DRING - 1)) * XTB;
                if (loc_a) __builtin_amdgcn_raw_buffer_store_b128(q, xrs, J * 1024 + lane * 16, so, 0);
                else __builtin_amdgcn_raw_buffer_store_b128(q, xrs, J * 1024 + lane * 16, so, 16 /* sc1 */);
            }
        }
    } else {
        // ---------------- rnn2 (+ fc1, fc2) ----------------
        auto gates = [&](int i) {                       // the whole gate stage is on the slot's chain (x1 -> here)
            const int nb = nb_of(i), sb = cbase_of(i) + (t & (DRING - 1)) * XTB;
            const int fr = table_row(SEGT[i * 64 + pj] + t, SEGT[i * 64 + SEG + pj], SEGT[i * 64 + 2 * SEG + pj], magic, mshift, hop, zrow);
            const int vo = (fr * 3 * H + prow) * 4;
            const float c0 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(crs, vo, 0, 0));
            const float c1 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(crs, vo, H * 4, 0));
            const float c2 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(crs, vo, 2 * H * 4, 0));
            unsigned ow = 0u;
            float s0, s1, s2;
            stage(N3{}, VG{}, A_ih[0], A_ih[1], A_ih[2], sb + 5 * DLAYERB, nb, 0x828u, t, 0, s0, s1, s2, ow, true);
            const float h = MOL ? gru_update_fast(s0 + c0, s1 + c1, s2 + c2, ST[(i * 8 + 1) * NT + tid], ST[(i * 8 + 2) * NT + tid], ST[(i * 8 + 3) * NT + tid], ST[(i * 8 + 0) * NT + tid])
                                : gru_update(s0 + c0, s1 + c1, s2 + c2, ST[(i * 8 + 1) * NT + tid], ST[(i * 8 + 2) * NT + tid], ST[(i * 8 + 3) * NT + tid], ST[(i * 8 + 0) * NT + tid]);
            ST[(i * 8 + 0) * NT + tid] = h;
            publish4l(xrs, sb + 6 * DLAYERB + J * 1024, tid, __uint_as_float(ow) + h, pj < nb, loc_x2);      // x2 = x1 + h2 (:216)
            publish4l(xrs, sb + 1 * DLAYERB + J * 1024, tid, h, pj < nb, loc_b);
            CHX(1);
        };
        auto fc2_stage = [&](int i) {                   // fc2 + relu (:220-221) -> y2 (to the sampling workgroup on the other XCD)
            const int nb = nb_of(i), sb = cbase_of(i) + (t & (DRING - 1)) * XTB;
            const int fr = table_row(SEGT[i * 64 + pj] + t, SEGT[i * 64 + SEG + pj], SEGT[i * 64 + 2 * SEG + pj], magic, mshift, hop, zrow);
            const float cv = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(f2rs, (fr * H + prow) * 4, 0, 0));
            float s0, s1, s2;
            unsigned dummy = 0u;
            stage(N1{}, AGR{}, A_fc2, A_fc2, A_fc2, sb + 2 * DLAYERB, nb, 0x802u, t, 4, s0, s1, s2, dummy, false);
            const float y2v = fmaxf(s0 + cv, 0.f);
            if (fold3) {
                Y2T[pu * 16 + pj] = y2v;                // (the next workgroup barrier -- the gh stage's -- lies between wave 0's reads below and the next writes)
                lds_barrier();
                if (w == 0) {
                    float yb[4];
#pragma unroll
                    for (int k4 = 0; k4 < 4; ++k4) yb[k4] = Y2T[(4 * k4 + kq) * 16 + fi];
                    const u32x4 sv = {SENT, SENT, SENT, SENT};
                    const int so2 = cbase_of(i) + ((t + DAHEAD_IH) & (DRING - 1)) * XTB;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int k4 = 0; k4 < 4; ++k4) c = __builtin_amdgcn_mfma_f32_16x16x4f32(A3[q][k4], yb[k4], c, 0, 0, 0);
                        const u32x4 qv = {__float_as_uint(c[0]), __float_as_uint(c[1]), __float_as_uint(c[2]), __float_as_uint(c[3])};
                        // partial logits, rows 16 q + 4 kq + e of segment fi: layer 3 (tile 0) / 16 (tile 1), block J, to the sampling workgroup's XCD
                        if (fi < nb) __builtin_amdgcn_raw_buffer_store_b128(qv, xrs, (q == 0 ? 3 : 16) * DLAYERB + J * 1024 + lane * 16, sb, 16 /* sc1 */);
                        __builtin_amdgcn_raw_buffer_store_b128(sv, xrs, (q == 0 ? 3 : 16) * DLAYERB + J * 1024 + lane * 16, so2, 16 /* sc1 */);      // ring hygiene (as rearm1)
                    }
                }
            } else publish4l(xrs, sb + 3 * DLAYERB + J * 1024, tid, y2v, pj < nb, false);
            // ring hygiene: behind the last poll of the slot's step (y1(t): everybody is past the readers of step t - 2) and behind the publication
            rearm1(i, 1, 0, loc_b);
            rearm1(i, 6, 1, loc_x2);
            rearm1(i, 2, 2, loc_y1);
            if (!fold3) rearm1(i, 3, 3, false);
            CHX(5);
        };
        for (; t < T1; ++t) {
            if (PROF && tid == 0) PROFL[15] += 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll 1
            for (int i = 0; i < nact; ++i) gates(i);
#pragma unroll 1
            for (int i = 0; i < nact; ++i) fc1_stage(i);
#pragma unroll 1
            for (int i = 0; i < nact; ++i) fc2_stage(i);
#pragma unroll 1
            for (int i = 0; i < nact; ++i) gh_stage(i); // (h2(t) arrived with x2(t); needed at the cell of step t + 1: under the sampling and the x_t / x1 hops)
        }
    }
    if (PROF && tid == 0 && a.prof) {
        for (int k = 0; k < 16; ++k) a.prof[(size_t)(blockIdx.x & 255) * 32 + k] += PROFL[k];
    }
    for (int i = 0; i < nact; ++i)
        *reinterpret_cast<float4 *>(state_wg + (size_t)i * CHSTATE_SLOT + tid * 4) =
            make_float4(ST[(i * 8 + 0) * NT + tid], ST[(i * 8 + 1) * NT + tid], ST[(i * 8 + 2) * NT + tid], ST[(i * 8 + 3) * NT + tid]);
}
#undef CHX

// Grid = 4 clusters x 64 workgroups of 256 threads (one per CU), cooperative launch; a cluster without a group leaves at once.  Placement
// (speed only, verified at run time): block b is observed to run on XCD b % 8; cluster cl = XCDs 2 cl (its rnn1 workgroups) and 2 cl + 1
// (rnn2, fc1, fc2); the 32 blocks of an XCD: unit blocks J = 0 .. 31.
template <int MODE, bool PROF>
__global__ __launch_bounds__(NT, 1) void wrnn_chain_kernel(const LoopArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    check_kind(a);
    const int b = blockIdx.x;
    const int xcd = b % 8, J = b / 8;
    const int cl = xcd >> 1, layer = xcd & 1;
    if (cl >= a.NG) return;
    const int wgi = layer * 32 + J;
    bool loc_a = false, loc_b = false;
    {   // placement handshake (as wrnn_duo.hip): a half of the cluster seen on ONE XCD exchanges its own layers through that XCD's L2 with plain stores
        int *TAB = reinterpret_cast<int *>(smem) + ch_lds(a.G).off_misc;
        const int tid = threadIdx.x;
        unsigned *tab = a.xcc_tab + cl * CHWG;
        if (tid == 0) {
            const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xFu;          // HW_REG_XCC_ID
            __hip_atomic_store(tab + wgi, xcc + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        unsigned v = 1u;
        if (tid < CHWG) {
            unsigned spins = 0;
            v = ld_agent32(tab + tid);
            while (v == 0u && ++spins < 200000u) {
                __builtin_amdgcn_s_sleep(2);
                v = ld_agent32(tab + tid);
            }
            TAB[tid] = (int)v;
        }
        __syncthreads();
        const int ok_a = (tid < 32) ? (v != 0u && (int)v == TAB[0]) : 1;
        const int ok_b = (tid >= 32 && tid < CHWG) ? (v != 0u && (int)v == TAB[32]) : 1;
        loc_a = __syncthreads_and(ok_a) != 0;
        loc_b = __syncthreads_and(ok_b) != 0;
        if (a.tuning & 256) { loc_a = false; loc_b = false; }       // A/B: everything written through
        __syncthreads();
    }
    const int ncl = CHCL;
    if (layer == 0) ch_role<MODE, true, PROF>(a, smem, cl, ncl, J, loc_a, loc_b);
    else ch_role<MODE, false, PROF>(a, smem, cl, ncl, J, loc_a, loc_b);
}

int chain_clusters(int n_cus) { return n_cus >= CHCL * CHWG ? CHCL : 0; }
int chain_max_depth() { return CHMAXG; }
size_t chain_state_floats(int G) { return (size_t)CHCL * CHWG * G * CHSTATE_SLOT; }
size_t chain_xbuf_bytes(int G) { return (size_t)G * CHCL * DSLOTB; }

hipError_t launch_chain(const LoopArgs &args, int mode, hipStream_t stream)
{
    if ((mode == 1 && !args.fc3f) || !args.u1 || !args.xcc_tab || args.NG < 1 || args.G < 1 || args.G > CHMAXG || args.NG > CHCL * args.G) return hipErrorInvalidValue;
    const size_t lds = (size_t)ch_lds(args.G).total * sizeof(float);
    const bool prof = args.prof && !(args.tuning & 64);
    const void *fn = mode == 1 ? (prof ? (const void *)wrnn_chain_kernel<1, true> : (const void *)wrnn_chain_kernel<1, false>)
                               : (prof ? (const void *)wrnn_chain_kernel<0, true> : (const void *)wrnn_chain_kernel<0, false>);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    LoopArgs a = args;
    void *params[] = {(void *)&a};
    return hipLaunchCooperativeKernel(fn, dim3(CHCL * CHWG), dim3(NT), params, (unsigned)lds, stream);
}

}  // namespace wrnn
