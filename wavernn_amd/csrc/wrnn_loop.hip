// wrnn_loop.hip -- the ROLE-SPLIT pipelined persistent WaveRNN loop kernel (MOL and RAW) for MI355X (gfx950 / CDNA4).
//
// Round 3 (measured: profiles/r03i_probe_fused_{mol,raw}.json; bit-identical results, tests/test_gpu_parity.py::test_fused_stages_equal_unfused_bitwise):
//  (1) RAW: the sampling stages alternate between the roles with >= 2 groups in flight, as MoL's do (role B holds its 16 fc3 rows
//      too, re-arms the logit rows of the slots it samples, hands x_t over through layer 7): 1.32x at depth 2, 1.43x at depth 4.
//      wrnn_options.tuning bit 3 = role A alone, as before.
//  (2) fused stages: the GATES / GH pointwise half of the previous GROUP is interleaved with the MFMA tiles of the next stage of
//      the same phase (sched_group_barrier 1 MFMA : 6 / 2 VALU); branch-free tanh (tanh_sel: the library's two paths + select,
//      same instructions), publish4_nb / issue_sel (stores / loads that must not happen go to an out-of-range buffer offset),
//      frame-index write to a scratch LDS word.  On with >= 4 groups in flight (+1-2 %), off below (it costs 5-9 % at depth 2);
//      tuning bit 2 switches it off everywhere.
//
// Replaces the `for i in range(seq_len)` loop of fatchord/WaveRNN `WaveRNN.generate()` (reference
// models/fatchord_version.py:201-241) for one ROUND of folded segments (<= clusters x G groups of <= 16) over a range of
// steps [t0, t1).  Second generation of the clustered design (wrnn_cluster.hip / wrnn_pipe.hip); what changed and why
// (profiles/r01t_summary.md: MFMA busy 25.7 %, 30 % of the issued MFMA rows were padding, sweeps 30 % of a group-step):
//
//   * ROLES instead of padding.  A cluster is still 64 workgroups holding one fp32 copy of the loop weights in registers,
//     but workgroup wg = 2 J + role owns SIXTEEN hidden units [16 J, 16 J + 16) of ONE recurrent layer:
//         role A: rnn1 (W_ih 3 gate tiles + W_hh 3 gate tiles) + the same 16 rows of fc1,
//         role B: rnn2 (3 + 3 tiles)                            + the same 16 rows of fc2,
//     so every v_mfma_f32_16x16x4_f32 tile is 16 live rows (gate g of the 16 owned units; 7 tiles per workgroup against
//     10 two-thirds-full ones before; 224 weight registers, which fit the AGPR half of the register file -- 9 tiles did not
//     and spilled the B operands).  fc3 lives in LDS in A-fragment order: MOL -- every workgroup holds all 30 rows (2 tiles)
//     and samples redundantly, no 5th exchange; RAW -- role-A workgroup J owns logit rows [16 J, 16 J + 16) and the 512
//     logits are a 5th exchange.  Only role A needs x_t (for xi): with one group in flight (and in RAW) role A alone runs
//     fc3 + sampling; with >= 2 groups in flight (MOL) the sampling stages alternate between the roles by slot parity, role
//     B handing its x_t to role A through a 16-word exchange layer, so both roles run 3.5 stages per group-step.
//   * TAG-FREE exchange in MFMA-FRAGMENT ORDER.  A layer (h1, h2, y1, y2, RAW logits) of a group is 16 segments x 512 f32 =
//     32 KB stored as [wave w][k-block r][lane][4]: exactly the B fragments wave w feeds its MFMAs, so a consumer wave
//     loads its operands with 8 buffer_load_dwordx4 (sc1) straight into registers -- no LDS staging, no sweep barrier, half
//     the bytes of the 8-byte {tag, value} granules.  Arrival is detected by value: every slot is pre-filled with a
//     sentinel NaN pattern (0xFFFFFFFF, which no finite activation takes) and a consumer re-reads until none of its 32
//     words is the sentinel.  Slots form a ring of 4 by step; a producer wave re-arms its own words of slot (t+3) % 4 at the
//     end of step t (it holds step t-1, which every consumer is provably done with), and drains its stores (s_waitcnt
//     vmcnt(0)) at the end of step t+1, before it publishes anything of step t+2 -- so the re-arm is visible before any
//     poll of step t+3 can start (such a poll follows data that depends on those publications).
//     A producer's 16 units x 16 segments are one contiguous 1 KB block of the layer: 64 lanes x 16-byte sc1 stores.
//   * no activation tile in LDS at all: the residual sums x1 = xi + h1, x2 = x1 + h2 (:212,:216) are re-formed in registers
//     from the conditioning slab and the ring slots (still valid: re-armed two steps later), each wave touching only its own
//     K chunk, so the only workgroup barrier of a stage is the one between writing and summing the four waves' partial
//     tiles; partial buffers ping-pong.  5 barriers per group-step (was 14); 4.3 KB of LDS per group in flight.
//   * the hoisted I-layer conditioning cI arrives in fragment order and in SLABS of a few hundred steps (wrnn_cond.hip):
//     the host loops slabs x rounds, the kernel saves / restores its per-group state, the workspace no longer scales with T.
// Arithmetic and summation order per output are those of wrnn_cluster.hip / wrnn_pipe.hip (K split over 4 waves, two
// accumulator chains per tile, partials added in wave order), so results are bit-identical to those kernels.
#include <type_traits>

#include "wrnn_ring.h"

namespace wrnn {

constexpr int LOOP_PUBFIRST_DEPTH = 2;       // (measured: profiles/r03v_probe_pubfirst.json)
constexpr int LOGS = 36;                     // row stride of the logits scratch: writer lane (row 4w + (l & 3), segment (l >> 2) & 15) -> bank 4 * segment + row: conflict-free
constexpr int LPART = 2 * NW * 3 * 256;      // two ping-pong sets of [wave][slot 0..2][16][16]
struct LoopLds {
    int off_part, off_log, off_wi0, off_f3, off_lgt, off_misc, off_prof, total;
};
__host__ __device__ inline LoopLds loop_lds(int mode, int G)
{
    LoopLds l;
    int o = G * LGRP;
    l.off_part = o; o += LPART;
    l.off_log = o;  o += SEG * LOGS;                       // [segment][LOGS] logits of the group being sampled (528 floats: keeps 16-B alignment)
    l.off_wi0 = o;  o += H;
    l.off_f3 = o;   o += (mode == 1 ? 2 : 1) * XT;         // fc3 in A-fragment order [tile][wave][r][lane][4]
    l.off_lgt = o;  o += (mode == 0) ? SEG * LDC : 0;      // RAW: the gathered logits as [segment][class] rows
    l.off_misc = o; o += 16 + 2 * LMAXG;                   // [0] failure flag; [16 + 2 i], [17 + 2 i]: first segment / segment count of slot i
    l.off_prof = o; o += 2 * 32;                           // [4 phases][8] u64 phase clocks (profiling builds)
    l.total = o;
    return l;
}

// The whole life of one workgroup in one role (compile-time, so the two roles are disjoint code with separate register
// allocations).  MODE: 0 RAW (C == 512), 1 MOL (C == 30).
// PROF: thread 0 accumulates shader clocks per (phase, segment of the stage) in LDS and writes them to a.prof at the end:
//   [phase 0..3][0 issue, 1 barrier wait of the previous stage's back half, 2 its pointwise + publish, 3 load wait / poll,
//                4 operand build, 5 MFMA + partial tiles, 6 stages, 7 stages whose first check found a sentinel]
template <int MODE, bool roleA, bool PROF>
__device__ __forceinline__ void loop_role(const LoopArgs &a, float *smem, int cl, int wg, int ncl)
{
    constexpr bool MOL = MODE == 1;
    const int G = a.G;
    const LoopLds L = loop_lds(MODE, G);
    float *PART = smem + L.off_part, *LOG = smem + L.off_log, *WI0 = smem + L.off_wi0, *F3 = smem + L.off_f3, *LGT = smem + L.off_lgt;
    int *FAIL = reinterpret_cast<int *>(smem + L.off_misc);
    int *GEO = FAIL + 16;                                // [2 i] first segment of slot i's group in the call's table, [2 i + 1] its count
    u64 *PROFL = reinterpret_cast<u64 *>(smem + L.off_prof);
    u64 plast = 0;
#define PH(k)                                                                  \
    do {                                                                       \
        if (PROF && tid == 0) {                                                \
            const u64 now_ = __builtin_amdgcn_s_memtime();                     \
            PROFL[k] += now_ - plast;                                          \
            plast = now_;                                                      \
        }                                                                      \
    } while (0)

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int J = wg >> 1;                              // owned units [16 J, 16 J + 16)
    const int fi = lane & 15, kq = lane >> 4;
    const int kbase_lane = KCH * w + 4 * kq;
    const int pu = 4 * w + (tid & 3), pj = (tid >> 2) & 15;   // pointwise role: (owned unit, segment)
    const int prow = LU * J + pu;                       // hidden index of the pointwise role
    const int T0 = a.t0, T1 = a.t1, C = a.C;
    const int NR = a.Btot, Nall = a.Nall;               // segments of this round / of the whole call (row stride of noise, logits)
    const int NGR = a.NG;                               // groups of this round
    const bool leader = wg == (roleA ? 0 : 1);           // the workgroup of this role that writes out / publishes x_t

    // ---- one-time: weight slice -> register-resident MFMA A fragments (7 tiles = 224 registers) ------------------------
    float A_ih[3][AF], A_hh[3][AF], A_fc[AF];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        const int grow = g * H + LU * J + fi;
        load_afrag(A_ih[g], roleA ? a.w_ih1 : a.w_ih2, roleA ? H : H + AUX, grow, true, kbase_lane);
        load_afrag(A_hh[g], roleA ? a.w_hh1 : a.w_hh2, H, grow, true, kbase_lane);
    }
    load_afrag(A_fc, roleA ? a.fc1_w : a.fc2_w, H + AUX, LU * J + fi, true, kbase_lane);
    // per-thread constants of the pointwise role
    float bi_r = 0.f, bi_z = 0.f, bi_n = 0.f;           // role A: b_ih1 (role B's b_ih2 is inside c2f)
    if constexpr (roleA) { bi_r = a.b_ih1[prow]; bi_z = a.b_ih1[H + prow]; bi_n = a.b_ih1[2 * H + prow]; }
    const float *bhh = roleA ? a.b_hh1 : a.b_hh2;
    const float bh_r = bhh[prow], bh_z = bhh[H + prow], bh_n = bhh[2 * H + prow];
    const float b3a = MOL ? a.fc3_b[pu] : a.fc3_b[prow];                             // MOL: logit rows pu and 16 + pu; RAW: row of the pointwise role
    const float b3b = (MOL && 16 + pu < 30) ? a.fc3_b[16 + pu] : 0.f;

    for (int q = tid; q < L.total; q += NT) smem[q] = 0.f;
    __syncthreads();
    WI0[2 * tid] = a.I_w0[2 * tid];
    WI0[2 * tid + 1] = a.I_w0[2 * tid + 1];
    // fc3 -> LDS in A-fragment order: F3[tile s][wave][r][lane (row fi, k-quad kq)][4] = fc3_w[row(s, fi)][128 wave + 16 r + 4 kq ..]
    for (int q = tid; q < (MOL ? 2 : 1) * (XT / 4); q += NT) {      // (both roles: with >= 2 groups in flight role B samples too)
        const int l6 = q & 63, r = (q >> 6) & 7, wv = (q >> 9) & 3, sidx = q >> 11;
        const int rfi = l6 & 15, rkq = l6 >> 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        int row = -1;
        if (MOL) { if (16 * sidx + rfi < 30) row = 16 * sidx + rfi; }
        else { row = LU * J + rfi; }                                   // RAW: role-A workgroup J owns logit rows [16 J, 16 J + 16)
        if (row >= 0) v = *reinterpret_cast<const float4 *>(a.fc3_w + (size_t)row * H + KCH * wv + 16 * r + 4 * rkq);
        reinterpret_cast<float4 *>(F3)[q] = v;
    }

    // ---- groups of this round run by this cluster: slot i <-> group cl + ncl * i ---------------------------------------
    int nact = 0;
    for (int i = 0; i < G; ++i)
        if (cl + ncl * i < NGR) nact = i + 1;
    const size_t state_wg = ((size_t)(cl * LNWGC + wg) * G) * LGRP;
    for (int i = 0; i < nact; ++i) {
        float *GP = smem + i * LGRP;
        const int g = cl + ncl * i;
        const int b0 = (int)(((long)g * NR) / NGR), nb = (int)(((long)(g + 1) * NR) / NGR) - b0;
        if (tid == 0) { GEO[2 * i] = a.rb0 + b0; GEO[2 * i + 1] = nb; }
        if (a.resume) {
            const float4 *src = reinterpret_cast<const float4 *>(a.state + state_wg + (size_t)i * LGRP);
            for (int q = tid; q < LGRP / 4; q += NT) reinterpret_cast<float4 *>(GP)[q] = src[q];
        } else {
            // fatchord_version.py:194-196: h1 = h2 = 0, x = 0  =>  gh = W_hh . 0 + b_hh = b_hh
            GP[tid] = bh_r; GP[256 + tid] = bh_z; GP[512 + tid] = bh_n;
            GP[O_HOWN + tid] = 0.f;
            if (tid < SEG) {
                GP[O_XS + tid] = 0.f;
                int *SP = reinterpret_cast<int *>(GP + O_SP);
                const int sc = a.rb0 + b0 + (tid < nb ? tid : nb - 1);
                SP[tid] = a.seg_pos[sc];
                SP[SEG + tid] = a.seg_lim[sc];
                const int p0 = SP[tid] + T0;
                reinterpret_cast<int *>(GP + O_FR)[SEG * (T0 & 1) + tid] = (p0 < SP[SEG + tid]) ? (p0 / a.hop) : a.NF;
            }
        }
    }
    __syncthreads();

    const __amdgpu_buffer_rsrc_t xrs = make_rsrc(a.xbuf, (unsigned)(XBUF_FLOATS * 4));
    // exchange layout: [cluster][slot][layer 0..4 = h1 h2 y1 y2 lg][ring 0..2][XT floats]
    const int xcl = cl * LMAXG;
    float touch = 0.f;
    int pp = 0;                                          // partial-tile ping-pong
    bool ok = true;
    unsigned fcode = 0u;                                 // phase whose poll gave up (reported on the exit path)
    int t = T0;

#define XLAYER(i, layer, ring) ((((xcl + (i)) * NXLAYER + (layer)) * XRING + (ring)) * XT)
#define PARTOF(q) (PART + (q) * (NW * 3 * 256))

    // ---- software pipeline, one stage deep.  A STAGE = (phase, group).  Its FRONT half issues the stage's loads, then -- while
    //      they fly -- runs the BACK half of the previous stage (the workgroup barrier, the 4-wave partial sum, the pointwise
    //      math, the publish), then checks the loads, runs the MFMA tiles and writes this wave's partial tiles.  One wave per
    //      SIMD cannot hide a load behind another wave; this hides every stage's loads behind the previous stage's barrier +
    //      pointwise half instead.  The pending back half is described by (bk, bi, bpp, bt, bc*).
    enum { BK_NONE = 0, BK_GATES, BK_GH, BK_RELU, BK_SAMPLE };
    // the exchanged layer a stage polls, by phase (role A phase 0 polls nothing: xi comes from the conditioning slab)
    auto stage_layer = [](int ph) -> int { return roleA ? (ph == 1 ? 0 : (ph == 2 ? 6 : 3)) : (ph == 0 ? 5 : (ph == 1 ? 1 : (ph == 2 ? 2 : 3))); };
    constexpr int NPH = 4;
    // Who runs fc3 + sampling (phase 3) for slot i.  Only role A needs x_t (for xi), so one group in flight (and RAW) is sampled by
    // role A; with >= 2 groups in flight the MOL sampling stages ALTERNATE -- even slots role A, odd slots role B, which hands
    // x_t to role A through a 16-word exchange layer -- so both roles run 3.5 stages per group-step instead of 4 and 3.
    const bool alternate = nact >= 2 && (MOL || (a.tuning & 8) == 0);      // (tuning bit 3: RAW sampled by role A alone, as before)
    auto samples = [&](int i) -> bool { return alternate ? (((i & 1) == 0) == roleA) : roleA; };
    // RAW with alternating roles: the softmax + Categorical sampling of a slot (16 segments x 512 classes, 2 KB of noise per segment
    // and step) runs REDUNDANTLY in all 32 workgroups of the sampling role; in `solo` form every workgroup of that role still publishes its 16
    // logit rows, but ONE of them (unit block (slot >> 1) % 32: a different one per slot) gathers the logits, samples and hands x_t
    // to the role-A workgroups through the 16-word layer 7 -- for every slot, not only the role-B-sampled ones.  MEASURED
    // (profiles/r03n_probe_raw_solo.json): no faster -- 54.9 vs 54.3 us per step at depth 4: a RAW step is bound by the latency of a
    // slot's chain through the sampling stage, not by the busy time of the 31 workgroups this relieves -- so it is OFF unless
    // tuning bit 4 asks for it (it does cut the noise reads 32-fold).
    const bool solo = !MOL && alternate && (a.tuning & 16) != 0;
    auto sole = [&](int i) -> bool { return samples(i) && J == ((i >> 1) & (LNJ - 1)); };
    auto from_ring = [&](int i) -> bool { return alternate && ((i & 1) || solo); };      // role A reads x_{t-1} of slot i from layer 7
    // last stage of a step this workgroup executes (the ring hygiene point)
    const int last_ph = (alternate || roleA) ? 3 : 2;
    const int last_i = !alternate ? nact - 1 : (roleA ? ((nact - 1) & ~1) : (((nact - 1) & 1) ? nact - 1 : nact - 2));
    unsigned xtw = 0u;                                   // x_{t-1} word of segment fi, polled from role B (alternate sampling)
    auto poll_xt = [&](int i, int ring, int nb, unsigned &v) -> bool {
        unsigned spins = 0;
        while (__any(fi < nb && v == SENT)) {
            if ((++spins & 255u) == 0u) {
                if (spins > SPIN_LIMIT || ld_agent32(a.status) != 0u) return false;
            }
            __builtin_amdgcn_s_sleep(1);
            v = __builtin_amdgcn_raw_buffer_load_b32(xrs, fi * 4, XLAYER(i, 7, ring) * 4, 16 /* sc1 */);
        }
        return true;
    };
    u32x4 x[8];                                          // fragments of the polled layer; issued ONE STAGE AHEAD (before the previous
    bool xahead = false;                                 // stage's MFMA tiles) whenever that stage is in the same step
    const bool lookahead = (a.tuning & 1) == 0, full_fence = (a.tuning & 2) != 0;      // A/B switches (wrnn_options.tuning)
    // fused stages (below): measured (profiles/r03i_probe_fused_*.json) 1-2 % faster with >= 4 groups in flight, 5-9 % SLOWER with 2 (the
    // publish of a fused half leaves later, inside the MFMA block, and at depth 2 a step is bound by the latency of that chain)
    const bool fuse_on = (a.tuning & 4) == 0 && nact >= 4;
    // PUBLISH FIRST (round 3): with <= LOOP_PUBFIRST_DEPTH groups in flight a step is bounded by the latency of the slots' chains, and
    // every hop of a chain ends with a back half (barrier, pointwise, publish) that used to wait behind the next stage's load issue
    // and conditioning reads; there a stage now STARTS with the pending back half.  (Deeper pipelines are busy-bound: loads first,
    // so they fly under the back half.)  tuning bit 5 = never, bit 6 = always.
    const bool pubfirst = (a.tuning & 64) != 0 || ((a.tuning & 32) == 0 && nact <= LOOP_PUBFIRST_DEPTH);
    // MoL: hardware exp / rcp in the GRU pointwise math (gru_update_fast, as wrnn_duo.hip; inside the 1e-5 tolerance); RAW keeps the
    // library forms (bit-exact class indices).  tuning bit 7 = the library forms for MoL too.
    const bool fast_pw = MOL && (a.tuning & 128) == 0;
    int bk = BK_NONE, bi = 0, bpp = 0, bt = 0, cur_ph = 0;
    float bc0 = 0.f, bc1 = 0.f, bc2 = 0.f;

    // The pending half of kind K (GATES / GH; the short RELU half is not worth it) as straight-line code -- no barrier, no branch: stores that must not happen go
    // to a dropped buffer offset / a scratch LDS word -- so that it can be interleaved with the MFMA tiles of the next stage.
    auto pointwise_straight = [&](auto KC) {
        constexpr int K = decltype(KC)::value;
        float *GP = smem + bi * LGRP;
        const float *PB = PARTOF(bpp);
        const int nb = GEO[2 * bi + 1];
        const int bring = bt % XRING;
        if constexpr (K == BK_GATES) {
            const float gir = get_partial<3>(PB, 0, pu, pj) + bc0;
            const float giz = get_partial<3>(PB, 1, pu, pj) + bc1;
            const float gin = get_partial<3>(PB, 2, pu, pj) + bc2;
            const float xo = GP[O_XO + tid];
            const float hn = fast_pw ? gru_update_fast(gir, giz, gin, GP[tid], GP[256 + tid], GP[512 + tid], GP[O_HOWN + tid])
                                     : gru_update_sel(gir, giz, gin, GP[tid], GP[256 + tid], GP[512 + tid], GP[O_HOWN + tid]);
            GP[O_HOWN + tid] = hn;
            publish4_nb(xrs, (XLAYER(bi, roleA ? 0 : 1, bring) + 256 * J) * 4, tid, hn, pj < nb);
            publish4_nb(xrs, (XLAYER(bi, roleA ? 5 : 6, bring) + 256 * J) * 4, tid, xo + hn, pj < nb);
        } else if constexpr (K == BK_GH) {
            const float g0 = get_partial<3>(PB, 0, pu, pj), g1 = get_partial<3>(PB, 1, pu, pj), g2 = get_partial<3>(PB, 2, pu, pj);
            const int *SP = reinterpret_cast<const int *>(GP + O_SP);
            const int sl = tid & (SEG - 1);
            const int p1 = SP[sl] + bt + 1;
            const int fr = (p1 < SP[SEG + sl]) ? (p1 / a.hop) : a.NF;
            int *frp = tid < SEG ? reinterpret_cast<int *>(GP + O_FR) + SEG * ((bt + 1) & 1) + tid : FAIL + 8;   // FAIL[8]: scratch word
            GP[tid] = g0 + bh_r;
            GP[256 + tid] = g1 + bh_z;
            GP[512 + tid] = g2 + bh_n;
            *frp = fr;
        } else static_assert(K == BK_GATES || K == BK_GH, "fusable kinds");

        bk = BK_NONE;
    };

    // KINDS: bit mask (1 << BK_*) of the halves that can be pending at the call site (compile-time: every site inlines only those,
    // and RAW's heavy sampling half exists at one site only)
    auto run_back = [&](auto KINDS) -> bool {
        constexpr unsigned KM = decltype(KINDS)::value;
        if (bk == BK_NONE) return true;
        float *GP = smem + bi * LGRP;
        const float *PB = PARTOF(bpp);
        const int nb = GEO[2 * bi + 1];
        const int bring = bt % XRING;
        if (!ok) FAIL[0] = 1;
        // every wave's partial tiles of the stage are in LDS.  Only LDS is handed over here, so the barrier waits for LDS
        // traffic alone (lgkmcnt): __syncthreads() would also drain vmcnt -- i.e. wait for the very loads this stage has just
        // put in flight -- through its workgroup-scope fences.
        if (full_fence) __syncthreads();
        else lds_barrier();
        if (FAIL[0] != 0) return false;
        PH(8 * cur_ph + 1);
        if ((KM & (1u << BK_GATES)) && bk == BK_GATES) {   // GRU cell pointwise (ATen gru_cell) -> publish h1 (role A) / h2 (role B)
            const float gir = get_partial<3>(PB, 0, pu, pj) + bc0;
            const float giz = get_partial<3>(PB, 1, pu, pj) + bc1;
            const float gin = get_partial<3>(PB, 2, pu, pj) + bc2;
            const float hn = fast_pw ? gru_update_fast(gir, giz, gin, GP[tid], GP[256 + tid], GP[512 + tid], GP[O_HOWN + tid])
                                     : gru_update(gir, giz, gin, GP[tid], GP[256 + tid], GP[512 + tid], GP[O_HOWN + tid]);
            GP[O_HOWN + tid] = hn;
            publish4(xrs, (XLAYER(bi, roleA ? 0 : 1, bring) + 256 * J) * 4, tid, hn, pj < nb);
            // ... and the residual sum of the owned units, so its consumers load ONE layer: x1 = xi + h1 (:212) for role B's rnn2
            // gates, x2 = x1 + h2 (:216) for role A's fc1
            publish4(xrs, (XLAYER(bi, roleA ? 5 : 6, bring) + 256 * J) * 4, tid, GP[O_XO + tid] + hn, pj < nb);
        } else if ((KM & (1u << BK_GH)) && bk == BK_GH) {  // gh(t+1) = W_hh . h(t) + b_hh of the owned (unit, segment)
            // (all twelve partials are read before the first store: the compiler cannot tell GP from PB and would otherwise
            // serialise read -> wait -> store per gate)
            const float g0 = get_partial<3>(PB, 0, pu, pj), g1 = get_partial<3>(PB, 1, pu, pj), g2 = get_partial<3>(PB, 2, pu, pj);
            GP[tid] = g0 + bh_r;
            GP[256 + tid] = g1 + bh_z;
            GP[512 + tid] = g2 + bh_n;
            if (tid < SEG) {   // conditioning frame of every segment at the NEXT step (Stretch2d: constant over a hop; the fold's zero
                               // pad -> NF), into the other half of FR: it is first read two stages' barriers from here
                const int *SP = reinterpret_cast<const int *>(GP + O_SP);
                const int p1 = SP[tid] + bt + 1;
                reinterpret_cast<int *>(GP + O_FR)[SEG * ((bt + 1) & 1) + tid] = (p1 < SP[SEG + tid]) ? (p1 / a.hop) : a.NF;
            }
        } else if ((KM & (1u << BK_RELU)) && bk == BK_RELU) {   // fc1 / fc2 + relu -> publish y1 (role A) / y2 (role B)
            publish4(xrs, (XLAYER(bi, roleA ? 2 : 3, bring) + 256 * J) * 4, tid, fmaxf(get_partial<3>(PB, 0, pu, pj) + bc0, 0.f), pj < nb);
        } else if constexpr ((KM & (1u << BK_SAMPLE)) != 0u) {  // fc3 logits -> sample x_t
            if (bk == BK_SAMPLE) {
            float *XS = GP + O_XS;
            const int b0 = GEO[2 * bi];                                   // first segment of the group in the call's segment table
            if constexpr (MOL) {
                {   // 30 logit rows x 16 segments: thread (rows pu and 16 + pu, segment pj) -- the partial tiles' conflict-free reader mapping
                    const int row = pu, sj = pj;
                    const float lg = get_partial<3>(PB, 0, row, sj) + b3a;
                    const float lg2 = get_partial<3>(PB, 1, row, sj) + b3b;      // (rows 30, 31 of the second tile: zero weights, unused)
                    LOG[sj * LOGS + row] = lg;
                    if (a.dbg_logits && leader && sj < nb) a.dbg_logits[((size_t)bt * Nall + b0 + sj) * C + row] = lg;
                    if (row < 14) {
                        LOG[sj * LOGS + 16 + row] = lg2;
                        if (a.dbg_logits && leader && sj < nb) a.dbg_logits[((size_t)bt * Nall + b0 + sj) * C + 16 + row] = lg2;
                    }
                }
                lds_barrier();
                {   // utils/distribution.py:102-121: 16-lane row = one segment (su), lane sm = mixture; bc0 / bc1 = this thread's noise
                    const int su = tid >> 4, sm = tid & 15;
                    float best = (sm < 10) ? mol_gumbel_pre(LOG[su * LOGS + sm], bc0) : -INFINITY;
                    int bidx = sm;
                    argmax_row16(best, bidx);
                    if (sm == 0 && su < nb) {
                        float x = mol_sample_pre(LOG[su * LOGS + 10 + bidx], LOG[su * LOGS + 20 + bidx], bc1);
                        if (leader) a.out[(size_t)(b0 + su) * a.T + bt] = x;
                        if (a.force_x) x = a.force_x[(size_t)(b0 + su) * a.T + bt];
                        XS[su] = x;
                        if (!roleA && leader)                                       // role B sampled this slot: hand x_t to role A
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(x), xrs, su * 4, XLAYER(bi, 7, bring) * 4, 16 /* sc1 */);
                    }
                }
            } else {
                const size_t tn = (size_t)(bt - a.noise_t0);
                if (solo && !sole(bi)) {                 // not this slot's sampler: publish the owned logit rows, done
                    publish4(xrs, (XLAYER(bi, 4, bring) + 256 * J) * 4, tid, get_partial<3>(PB, 0, pu, pj) + b3a, pj < nb);
                } else {
                // this wave samples segments 4 w .. 4 w + 3 (clamped to the group: a ragged group re-does its last segment, results
                // discarded): their Exp(1) variates are requested FIRST, so the 2 KB per segment arrive behind the logit exchange
                float qn[4][8];
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    const int sjc = (4 * w + s4 < nb) ? 4 * w + s4 : nb - 1;
#pragma unroll
                    for (int e = 0; e < 8; ++e) qn[s4][e] = a.noise[(tn * Nall + b0 + sjc) * C + lane + 64 * e];
                }
                publish4(xrs, (XLAYER(bi, 4, bring) + 256 * J) * 4, tid, get_partial<3>(PB, 0, pu, pj) + b3a, pj < nb);
                {   // the 512 logits of every segment -> LGT [segment][class]
                    u32x4 x[8];
                    float b[32];
                    issue(xrs, XLAYER(bi, 4, bring) * 4, w, lane, x);
                    unsigned spins = 0;
                    ok = ok && finish(xrs, XLAYER(bi, 4, bring) * 4, w, lane, nb, x, b, a.status, spins);
                    if (!ok && fcode == 0u) fcode = 0x400u | 7u;
                    float *lp = LGT + fi * LDC + kbase_lane;
#pragma unroll
                    for (int r = 0; r < 8; ++r) *reinterpret_cast<float4 *>(lp + 16 * r) = make_float4(b[4 * r], b[4 * r + 1], b[4 * r + 2], b[4 * r + 3]);
                }
                if (!ok) FAIL[0] = 1;
                lds_barrier();
                if (FAIL[0] != 0) return false;
                // fatchord_version.py:232-237: softmax -> Categorical (renormalise) -> argmax(p / q).  One wave per 4 segments, the
                // four handled in lock step (straight-line code: four independent butterfly chains in flight instead of one); per
                // segment the operation order is unchanged (bit-exact class indices)
                {
                    float lg[4][8], mx[4], sum[4], sum2[4], best[4];
                    int bidx[4];
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) {
                        const int sjc = (4 * w + s4 < nb) ? 4 * w + s4 : nb - 1;
                        mx[s4] = -INFINITY;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            lg[s4][e] = LGT[sjc * LDC + lane + 64 * e];
                            mx[s4] = fmaxf(mx[s4], lg[s4][e]);
                        }
                    }
                    if (a.dbg_logits && (leader || solo)) {
#pragma unroll
                        for (int s4 = 0; s4 < 4; ++s4)
                            if (4 * w + s4 < nb)
#pragma unroll
                                for (int e = 0; e < 8; ++e) a.dbg_logits[((size_t)bt * Nall + b0 + 4 * w + s4) * C + lane + 64 * e] = lg[s4][e];
                    }
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) mx[s4] = wave_max64(mx[s4]);      // (xor_pair butterflies, wrnn_device.h: == the __shfl_xor form bit for bit)
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) {
                        sum[s4] = 0.f;
#pragma unroll
                        for (int e = 0; e < 8; ++e) { lg[s4][e] = expf(lg[s4][e] - mx[s4]); sum[s4] += lg[s4][e]; }
                    }
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) sum[s4] = wave_sum64(sum[s4]);
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) {
                        sum2[s4] = 0.f;
#pragma unroll
                        for (int e = 0; e < 8; ++e) { lg[s4][e] = lg[s4][e] / sum[s4]; sum2[s4] += lg[s4][e]; }
                    }
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) sum2[s4] = wave_sum64(sum2[s4]);
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) {
                        best[s4] = -INFINITY;
                        bidx[s4] = 0;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float rr = (lg[s4][e] / sum2[s4]) / qn[s4][e];
                            if (rr > best[s4]) { best[s4] = rr; bidx[s4] = lane + 64 * e; }
                        }
                    }
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) wave_argmax64(best[s4], bidx[s4]);
                    if (lane == 0) {
#pragma unroll
                        for (int s4 = 0; s4 < 4; ++s4) {
                            const int sj = 4 * w + s4;
                            if (sj < nb) {
                                float x = 2.f * (float)bidx[s4] / ((float)C - 1.f) - 1.f;
                                if (leader || solo) a.out[(size_t)(b0 + sj) * a.T + bt] = x;
                                if (a.force_x) x = a.force_x[(size_t)(b0 + sj) * a.T + bt];
                                XS[sj] = x;
                                if (solo || (!roleA && leader))                             // hand x_t to the role-A workgroups
                                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(x), xrs, sj * 4, XLAYER(bi, 7, bring) * 4, 16 /* sc1 */);
                            }
                        }
                    }
                }
                lds_barrier();                         // LGT is read by every wave before the next group overwrites it
                }
            }
            // x_t is read by EVERY wave in this slot's first stage of the next step.  When this sampling half is the one that runs
            // inside that very stage (role A's last sampling stage of a step is slot 0's: one group in flight, or two with
            // alternating roles) the read follows at once; otherwise another stage's barrier lies in between.
            if (roleA && last_i == 0) lds_barrier();
            }
        }
        bk = BK_NONE;
        PH(8 * cur_ph + 2);
        return true;
    };

    // A STAGE of phase PH, compile-time: its loads, operands and tiles are straight-line code, and the pending back half can
    // only be the phase's own kind (previous group) or the previous phase's (its last group).  Phases of a step:
    //   role A: 0 = P1 rnn1 gates on xi (:208-210)      1 = P2 gh1(t+1) = W_hh1 . h1       2 = P3 fc1 + relu on x2 = (xi + h1) + h2 (:216-218)
    //   role B: 0 = P2 rnn2 gates on x1 = xi + h1 (:212-214)   1 = P3 gh2(t+1) = W_hh2 . h2   2 = P4 fc2 + relu on y1 (:220-221)
    //   both:   3 = P5 fc3 on y2 (:223); its back half samples x_t (:225-237)
    constexpr unsigned KM_SAMPLE = MOL ? (1u << BK_SAMPLE) : 0u;           // RAW runs its sampling half at once: never pending
    int ring = 0, tc = 0;
    auto stage = [&](auto PHC, int i) -> bool {
        constexpr int ph = decltype(PHC)::value;
        constexpr unsigned KM = ph == 0 ? ((1u << BK_GATES) | KM_SAMPLE | (roleA ? 0u : (1u << BK_RELU)))
                              : ph == 1 ? ((1u << BK_GH) | (1u << BK_GATES))
                              : ph == 2 ? ((1u << BK_RELU) | (1u << BK_GH))
                                        : (KM_SAMPLE | (1u << BK_RELU));
        float *GP = smem + i * LGRP;
        const int g = cl + ncl * i;
        const int nb = GEO[2 * i + 1];
        const float *cIg = a.cIf + ((size_t)tc * NGR + g) * XT;
        float4 c[8];
        float v0 = 0.f, v1 = 0.f, v2 = 0.f;           // conditioning / noise values the back half needs
        if (PROF && tid == 0 && plast == 0) plast = __builtin_amdgcn_s_memtime();
        if (pubfirst) {                                 // (cur_ph still names the stage the half belongs to: its clocks go there)
            if (!run_back(std::integral_constant<unsigned, KM>{})) return false;
        }
        cur_ph = ph;
        // ---------------- front, part 1 --------------------------------------------------------------------------------
        // The polled layer: normally its loads were issued ONE STAGE AGO (before the previous stage's MFMA tiles), so they
        // have landed.  They are consumed FIRST, before anything else touches vector memory: vmcnt retires in order, so a
        // wait for these loads placed after the back half's publish stores would also wait ~1 us for the stores'
        // write-through acknowledgements (measured: profiles/r02h_*).  If a word is still the sentinel, the poll comes
        // after the back half -- it may be waiting for this workgroup's own pending publication.
        constexpr bool polled = !(roleA && ph == 0);
        const int xl = stage_layer(ph);
        float b[32];
        bool ready = false;
        int la_soff = 0;
        if (polled) {
            if (xahead) ready = try_finish(lane, nb, x, b);
            else issue(xrs, XLAYER(i, xl, ring) * 4, w, lane, x);
        }
        // conditioning / noise values of the back half, the conditioning slab (role A, phase 0)
        if constexpr (roleA) {
            if (ph == 0) {
                v0 = bi_r; v1 = bi_z; v2 = bi_n;
                load_cI(cIg, w, lane, c);
                if (from_ring(i) && t > T0)              // x_{t-1} of a slot sampled elsewhere (at the first step of a launch: from the state)
                    xtw = __builtin_amdgcn_raw_buffer_load_b32(xrs, fi * 4, XLAYER(i, 7, (t + 3) % XRING) * 4, 16 /* sc1 */);
            }
            else if (ph == 2) v0 = a.c3f[(size_t)reinterpret_cast<const int *>(GP + O_FR)[SEG * (t & 1) + pj] * H + prow];
        } else {
            const int fr = reinterpret_cast<const int *>(GP + O_FR)[SEG * (t & 1) + pj];   // conditioning frame of (segment pj, step t)
            if (ph == 0) {
                v0 = a.c2f[(size_t)fr * 3 * H + prow];                       // aux columns of rnn2 + b_ih2: per-frame table
                v1 = a.c2f[(size_t)fr * 3 * H + H + prow];
                v2 = a.c2f[(size_t)fr * 3 * H + 2 * H + prow];
            } else if (ph == 2) v0 = a.c4f[(size_t)fr * H + prow];
        }
        if (MOL && ph == 3) {
            // this step's sampling noise, pre-transformed (wrnn_noise_mol_kernel): thread (segment tid >> 4, mixture tid & 15)
            const int b0 = GEO[2 * i];
            const int su = tid >> 4, sm = tid & 15;
            const float *nrow = a.noise_pre + (size_t)(t - a.noise_t0) * 11 * Nall;
            const int suc = su < nb ? su : nb - 1;
            v0 = nrow[(size_t)(b0 + suc) * 10 + (sm < 10 ? sm : 9)];
            v1 = nrow[(size_t)10 * Nall + b0 + suc];
        }
        PH(8 * ph + 0);
        // ---------------- the previous stage's back half -------------------------------------------------------------
        // FUSED when it is this phase's own kind (the previous GROUP's half: no data in common with this stage) and this stage's
        // operands are here: only its barrier runs now, its pointwise math + publish are interleaved with this stage's MFMA tiles
        // below (the wave issues in order and is alone on its SIMD: otherwise the VALU idles through every 96-MFMA block and the
        // MFMA pipe through every pointwise half).
        constexpr int KOWN = ph == 0 ? BK_GATES : BK_GH;       // (phases 0, 1 only)
        const bool fuse = !pubfirst && fuse_on && ph < 2 && bk == KOWN && (!polled || ready);
        if (fuse) {
            if (!ok) FAIL[0] = 1;
            lds_barrier();
            if (FAIL[0] != 0) return false;
            PH(8 * ph + 1);
        } else if (!run_back(std::integral_constant<unsigned, KM>{})) return false;
        // ---------------- front, part 2: operands -> MFMA tiles -> this wave's partial tiles -----------------------
        if (polled) {
            unsigned spins = 0;
            if (!ready) {
                ok = ok && finish(xrs, XLAYER(i, xl, ring) * 4, w, lane, nb, x, b, a.status, spins);
                if (!ok && fcode == 0u) fcode = 0x400u | (roleA ? 0u : 8u) | (unsigned)ph;
            }
            if (PROF && tid == 0) { PROFL[8 * ph + 6] += 1; PROFL[8 * ph + 7] += !ready; }
        }
        PH(8 * ph + 3);
        if (ph == last_ph && i == last_i) {
            // ---- ring hygiene, once per step, at the point where it is free: the last layer this role polls in the step (A: y2,
            //      B: y1) has just arrived, and it depends on every store this wave issued before polling for it.
            //      (1) Drain: every store of this wave so far -- in particular the re-arm it issued one step ago for the
            //      slot of step t+2 -- is acknowledged before anything of step t+1 is published (a consumer polls a word
            //      for step t+2 only after consuming data that depends on this wave's step-(t+1) publications).  The
            //      publishes themselves then need no drain and never stall on a stage's prefetched loads.
            //      (2) Re-arm this wave's own words of slot (t+3) % 4: it holds step t-1, which every consumer is done
            //      with (no workgroup publishes y2 of step t before it has finished step t-1).
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const int ringn = (t + 3) % XRING;
#pragma unroll 1
            for (int i2 = 0; i2 < nact; ++i2)
            {
                rearm(xrs, (XLAYER(i2, 0, ringn) + 256 * J + 64 * w) * 4, lane, roleA ? 0 : 1, roleA ? 2 : 3, roleA ? 5 : 6,
                      (!MOL && samples(i2)) ? 4 : -1);       // RAW: the logit rows this role publishes for the slots it samples
                if ((solo ? sole(i2) : (!roleA && leader && alternate && (i2 & 1))) && lane == 48) {   // the 4 x_t words this wave publishes (segments 4 w ..)
                    const u32x4 q = {SENT, SENT, SENT, SENT};
                    __builtin_amdgcn_raw_buffer_store_b128(q, xrs, 16 * w, XLAYER(i2, 7, ringn) * 4, 16 /* sc1 */);
                }
            }
        }
        {   // the NEXT stage's polled layer, one stage ahead: its 8 loads fly during this stage's MFMA tiles and the back half
            // that follows (a publication is normally several stages old by the time it is polled; if it is not there yet,
            // finish() polls as before).  Not across a step boundary (nothing of the next step is published yet).
            int nph = ph, ni = i;
            do {
                if (++ni >= nact) { ++nph; ni = 0; }
            } while (nph == 3 && nph < NPH && !samples(ni));
            xahead = lookahead && nph < NPH && !(roleA && nph == 0) && (MOL || nph != 3);
            la_soff = XLAYER(ni, stage_layer(nph < NPH ? nph : 0), ring) * 4;
            // a fused GATES half publishes inside the MFMA block: the look-ahead loads are issued there, AFTER its stores
            // (vmcnt retires in order -- the next stage's wait for these loads must not also wait for younger stores)
            if (xahead && !(fuse && ph == 0)) issue(xrs, la_soff, w, lane, x);
        }
        if (roleA && ph == 0) {
            float xs = GP[O_XS + fi];
            if (from_ring(i) && t > T0) {
                ok = ok && poll_xt(i, (t + 3) % XRING, nb, xtw);
                if (!ok && fcode == 0u) fcode = 0x400u | 0x20u;
                xs = (fi < nb) ? __uint_as_float(xtw) : 0.f;
                GP[O_XS + fi] = xs;                      // (every wave writes the same 16 values) kept for the launch's saved state
            }
            make_xi(c, WI0, xs, w, lane, b);             // xi(t) (:208-209)
        }
        if (ph == 0 && w == (J >> 3)) {
            // the owned units' slice of this GRU's input (role A: xi, role B: x1) -> LDS in publish order, for the residual
            // sum the gates' back half publishes: unit block J = k-block r = J & 7 of wave J >> 3
#pragma unroll
            for (int r = 0; r < 8; ++r)
                if (r == (J & 7)) *reinterpret_cast<float4 *>(GP + O_XO + 4 * lane) = make_float4(b[4 * r], b[4 * r + 1], b[4 * r + 2], b[4 * r + 3]);
        }
        PH(8 * ph + 4);
        float *PW = PARTOF(pp);
        if (ph == 0 || (ph == 1)) {                                          // three gate tiles of W_ih (ph 0) / W_hh (ph 1)
            f32x4 o0, o1, o2;
            if (fuse) {
                if constexpr (ph < 2) {
                    pointwise_straight(std::integral_constant<int, KOWN>{});
                    if (ph == 0) issue_sel(xrs, la_soff, w, lane, x, xahead);
                    if (ph == 0) mfma3(A_ih[0], A_ih[1], A_ih[2], b, o0, o1, o2);
                    else mfma3(A_hh[0], A_hh[1], A_hh[2], b, o0, o1, o2);
                    interleave_mfma_valu<96, (ph == 0 ? 6 : 2)>();
                }
            } else {
                if (ph == 0) mfma3(A_ih[0], A_ih[1], A_ih[2], b, o0, o1, o2);
                else mfma3(A_hh[0], A_hh[1], A_hh[2], b, o0, o1, o2);
            }
            put_partial<3>(PW, w, 0, lane, o0);
            put_partial<3>(PW, w, 1, lane, o1);
            put_partial<3>(PW, w, 2, lane, o2);
            bk = ph == 0 ? BK_GATES : BK_GH;
        } else if (ph == 2) {
            put_partial<3>(PW, w, 0, lane, mfma1(A_fc, b));
            bk = BK_RELU;
        } else {
            put_partial<3>(PW, w, 0, lane, mfma1_lds(F3 + frag_off(w, 0, lane), b));
            if constexpr (MOL) put_partial<3>(PW, w, 1, lane, mfma1_lds(F3 + XT + frag_off(w, 0, lane), b));
            bk = BK_SAMPLE;
        }
        if (roleA && ph == 2 && t + 1 < T1) {   // pull the next step's conditioning block of this group (32 KB) into this XCD's L2
            asm volatile("" ::"v"(touch));
            touch = a.cIf[((size_t)(tc + 1) * NGR + g) * XT + 32 * tid];
        }
        PH(8 * ph + 5);
        bi = i; bpp = pp; bt = t; bc0 = v0; bc1 = v1; bc2 = v2;
        pp ^= 1;
        if constexpr (!MOL && ph == 3) {                                     // RAW: the (heavy) sampling half is not deferred
            if (!run_back(std::integral_constant<unsigned, (1u << BK_SAMPLE)>{})) return false;
        }
        return true;
    };

    for (; t < T1; ++t) {
        ring = t % XRING;
        tc = t - a.cI_t0;                               // row of the conditioning slab
#pragma unroll 1
        for (int i = 0; i < nact; ++i)
            if (!stage(std::integral_constant<int, 0>{}, i)) goto bail;
#pragma unroll 1
        for (int i = 0; i < nact; ++i)
            if (!stage(std::integral_constant<int, 1>{}, i)) goto bail;
#pragma unroll 1
        for (int i = 0; i < nact; ++i)
            if (!stage(std::integral_constant<int, 2>{}, i)) goto bail;
        if constexpr (NPH == 4) {
#pragma unroll 1
            for (int i = 0; i < nact; ++i)
                if (samples(i)) {                       // (else the other role samples this slot)
                    if (!stage(std::integral_constant<int, 3>{}, i)) goto bail;
                }
        }
    }
    // the last stage's back half
    if (!run_back(std::integral_constant<unsigned, (roleA ? KM_SAMPLE : (KM_SAMPLE | (1u << BK_RELU)))>{})) goto bail;
    if (roleA && alternate && T1 > T0) {                                // x_{T1-1} of the slots role B sampled -> this launch's saved state
#pragma unroll 1
        for (int i = solo ? 0 : 1; i < nact; i += solo ? 1 : 2) {
            const int nb = GEO[2 * i + 1];
            unsigned v = __builtin_amdgcn_raw_buffer_load_b32(xrs, fi * 4, XLAYER(i, 7, (T1 - 1) % XRING) * 4, 16 /* sc1 */);
            ok = ok && poll_xt(i, (T1 - 1) % XRING, nb, v);
            smem[i * LGRP + O_XS + fi] = (fi < nb) ? __uint_as_float(v) : 0.f;
        }
        if (!ok) { fcode = 0x400u | 0x21u; FAIL[0] = 1; }
        __syncthreads();
        if (FAIL[0] != 0) goto bail;
    }
    asm volatile("" ::"v"(touch));
    // ---- save the per-group state for the next slab of steps ------------------------------------------------------
    __syncthreads();
    for (int i = 0; i < nact; ++i) {
        const float4 *GP = reinterpret_cast<const float4 *>(smem + i * LGRP);
        float4 *dst = reinterpret_cast<float4 *>(a.state + state_wg + (size_t)i * LGRP);
        for (int q = tid; q < LGRP / 4; q += NT) dst[q] = GP[q];
    }
    if (PROF && tid == 0 && a.prof && blockIdx.x < MAXWG) {
        for (int k = 0; k < 32; ++k) a.prof[(size_t)blockIdx.x * 32 + k] += PROFL[k];
    }
    return;
bail:   // a bounded spin expired (or another workgroup raised the abort flag): record the first failure, leave
    if (fcode != 0u) report_failure(a.status, fcode, blockIdx.x, t, tid);
#undef PH
#undef XLAYER
#undef PARTOF
}

// Grid = clusters x 64 workgroups of 256 threads, cooperative launch.  Workgroup wg of a cluster: role A if wg is even.
template <int MODE, bool PROF>
__global__ __launch_bounds__(NT, 1) void wrnn_loop_kernel(const LoopArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int cl, wg;                                         // cluster, workgroup-in-cluster: whole XCDs per cluster (speed only)
    const int ncl = gridDim.x / LNWGC;
    check_kind(a);
    {
        const int b = blockIdx.x, nblk = gridDim.x;
        if (nblk % 8 == 0 && ncl >= 1 && 8 % ncl == 0) {
            const int xpc = 8 / ncl, per_xcd = nblk / 8;
            const int xcd = b % 8;
            cl = xcd / xpc;
            wg = (xcd % xpc) * per_xcd + b / 8;
        } else {
            cl = b / LNWGC;
            wg = b % LNWGC;
        }
    }
    if ((wg & 1) == 0) loop_role<MODE, true, PROF>(a, smem, cl, wg, ncl);
    else loop_role<MODE, false, PROF>(a, smem, cl, wg, ncl);
}

size_t loop_lds_bytes(int mode, int G) { return (size_t)loop_lds(mode, G).total * sizeof(float); }

// deepest pipeline: the exchange buffer has LMAXG slots per cluster (LDS is not the limit: 4.3 KB per group)
int loop_max_depth(int mode)
{
    int g = LMAXG;
    while (g > 1 && loop_lds_bytes(mode, g) > 160 * 1024) --g;
    return g;
}

size_t loop_state_floats(int G) { return (size_t)MAXCL * LNWGC * G * LGRP; }

int loop_clusters(int n_cus)
{
    int ncl = n_cus / LNWGC;
    if (ncl > MAXCL) ncl = MAXCL;
    while (ncl > 1 && (8 % ncl) != 0) --ncl;
    return ncl;
}

// One cooperative launch: clusters x 64 workgroups; args.G groups in flight per cluster, steps [args.t0, args.t1).
hipError_t launch_loop(const LoopArgs &args, int ncl, int mode, hipStream_t stream)
{
    if (ncl < 1 || args.G < 1 || args.G > loop_max_depth(mode)) return hipErrorInvalidValue;
    const size_t lds = loop_lds_bytes(mode, args.G);
    const void *fn = mode == 1 ? (args.prof ? (const void *)wrnn_loop_kernel<1, true> : (const void *)wrnn_loop_kernel<1, false>)
                               : (const void *)wrnn_loop_kernel<0, false>;      // phase clocks: MOL only
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    LoopArgs a = args;
    void *params[] = {(void *)&a};
    return hipLaunchCooperativeKernel(fn, dim3(ncl * LNWGC), dim3(NT), params, (unsigned)lds, stream);
}

}  // namespace wrnn
