// wrnn_duo.hip -- the TWO-WORKGROUPS-PER-CU form of the persistent WaveRNN loop kernel (MOL and 9-bit RAW) for MI355X (gfx950 / CDNA4).
//
// Same path and the same tag-free sentinel exchange in MFMA-fragment order as wrnn_loop.hip (reference
// models/fatchord_version.py:201-241); every role of wrnn_loop.hip is cut in two along the line between its critical and its
// off-critical half so that a workgroup fits 256 registers per lane and TWO workgroups share a CU (two waves per SIMD: one
// wave's MFMAs run under the other's VALU / LDS / memory / wait time):
//
//     A-ih : rnn1 W_ih (3 gate tiles) + fc1 (1 tile) = 128 weight registers   stages P0 (gates -> h1, x1 = xi + h1), P2 (fc1 -> y1)
//     B-ih : rnn2 W_ih (3 gate tiles) + fc2 (1 tile) = 128                    stages P0 (gates -> h2, x2 = x1 + h2), P2 (fc2 -> y2)
//     A-hh : rnn1 W_hh (3 gate tiles)                =  96 (+ 32: its rows of stage  P1 (gh1(t+1) = W_hh1 . h1(t) + b_hh); forms cI(t+2) of its 16 rows for
//            I.weight[:, 1:])                                                 every slot -- from the mel (or, wrnn_options.mel_stage, from the x25
//                                                                             signal: the last up-sampling stage too) and the frame's aux row
//     B-hh : rnn2 W_hh (3 gate tiles)                =  96 (RAW: + 32, its    stage  P1 (gh2(t+1)); MOL: workgroup J < groups in flight also runs fc3 +
//            16 rows of fc3)                                                  the sampling of slot J (first fc3 tile in LDS, second from L2); RAW: every
//                                                                             workgroup computes its 16 logit rows of every slot (ring layer 16),
//                                                                             workgroup J gathers the 512 logits of slot J and samples in the reference's order
//
// Round 4 -- the LEAN form.  Round 3's phase clocks (profiles/r03ab_*, r03ag_*) showed the ih workgroup's serial instruction stream
// -- 16.7 k cycles per group-step at depth 4 AND at depth 8, 5.2 k of them MFMA -- bounding the step while the hh workgroup idled
// half of the time, and the ISA showed why: ~500 non-MFMA instructions and ~40 branches per gate stage (failure-flag plumbing,
// run-time stage kinds, an integer division for the conditioning frame, 64-bit address arithmetic, the xi operand build, the
// residual slice through LDS).  This version keeps the exchange protocol and the data layout and removes that work:
//   * xi is never built.  rnn1's gi = W_ih . (cI + w0 x) + b = W_ih . cI + x (W_ih . w0) + b: the conditioning slab is the MFMA
//     operand as it comes from memory, and the x_{t-1} term -- one FMA per gate with the pack's pre-multiplied vector
//     u1 = W_ih1 . I.weight[:,0] -- moves into the gates' pointwise half.  x_{t-1} is therefore needed AFTER the gate tiles, not
//     before them: the sampling -> rnn1 hop of a slot's chain is shorter by one MFMA block.  (fp32 rounding differs from the
//     reference's order by ~1e-7 relative: MoL tolerance 1e-5; the RAW class indices still agree bit for bit on every golden and sweep.)
//   * the residual input of the owned units (xi / x1 for the published sums x1 = xi + h1, x2 = x1 + h2) is one 4-byte load per
//     thread from the layer the stage consumes anyway, not an LDS slice written by one wave behind eight compares;
//   * stage kinds are compile-time at every call site (a step is: gates of slot 0 | gates of the others | fc of slot 0 | fc of the
//     others), per-slot byte offsets are 32-bit buffer offsets, the conditioning frame is a multiply-high with a magic constant,
//     constants of the pointwise role live in registers;
//   * a bounded spin that expires reports and lets the wave run on (sticky `dead`: later polls are skipped; every wave keeps
//     its barrier sequence, the launch ends, wrnn_status() reports) -- no failure flag is checked on the hot path;
//   * one accumulator chain per gate tile (k ascending, the oracle's order; consecutive MFMAs of a chain are 96 cycles apart).
// Placement (speed only, verified at run time): with 4 clusters a cluster is two XCDs; all rnn1 workgroups (A-ih, A-hh) of a cluster
// sit on one XCD and all rnn2 workgroups on the other, the ih and the hh workgroup of the same units on one CU.  Every workgroup
// records its XCC id; where producer and consumers of a layer are seen to share an XCD (h1, gh1 | h2, gh2, y2) the producer's
// stores are plain write-back stores that stay in that XCD's L2 (consumers always read with sc1 = L2-served loads): no fabric traffic
// for those layers.  Everything else is written through (sc1) as before.
//
// Ring discipline: 4 ring entries per layer (by step), every word the sentinel until its step's value is stored.  (Round 3 ran 8 entries
// and re-armed 4 ahead; with the layers that stay inside one XCD written by plain stores the ring's footprint decides whether those lines
// ever leave the L2: 8 entries x 4 slots of h, gh, cI, y2 are 6-7 MB per XCD against 4 MB of L2 -- every stored byte was eventually evicted
// to the fabric, profiles/r04h_summary.md -- 4 entries and a 2-entry gh ring are ~2 MB.)  Two facts bound the skew inside a cluster: an hh
// workgroup polls h(t) of every ih workgroup, so it never runs ahead of one; an ih workgroup in step t + 1 holds x(t + 1), which needed
// everything every ih workgroup publishes in step t, which needed every hh workgroup's gh(t), i.e. every hh workgroup has consumed h(t - 1).
//   * layers an ih workgroup publishes (h, y, residual sum; DAHEAD_IH = 2): after the last poll of its step t a wave re-arms its own
//     words of entry (t + 2) % 4 -- data of step t - 2, consumed by every hh workgroup (above) and by every ih workgroup (x(t - 1)
//     needed it); the wave drains its stores at the top of step t + 1, before it publishes anything of that step; the entry is polled
//     for step t + 2 by consumers that have consumed this wave's step t + 1 data: the re-arm is visible before the poll (with the drain
//     at the re-arm site it need not be).  (Three ahead would be safe too -- the re-arm site lies behind the fc stage's poll, which needed
//     every hh workgroup's gh of this step -- two ahead keeps a step of margin on both sides.)  tests/test_duo_exchange_model.py runs
//     these rules as a discrete-event model under adversarial timing.
//   * layers an hh workgroup publishes with a sentinel (x_t, RAW logits; DAHEAD_HH = 3): once this step's y2 is there a wave drains, then
//     re-arms its words of entry (t + 3) % 4 -- data of step t - 1, consumed before the y2(t) the wave has just polled could exist; the
//     next drain is at step t + 1, and a consumer that polls the entry for step t + 3 has seen something this wave published in step t + 2.
//   * cI (layer 4) carries NO sentinel inside a launch and is never re-armed: rnn1's hh workgroup forms cI(t + 2) at the top of its step t
//     (the entry held cI(t - 2): the workgroup has polled h1(t - 1) of every ih workgroup, so every reader is past it) and drains before it
//     publishes anything of that step.  An ih workgroup reads cI(s) after its fc stage of step s - 1, whose x2(s - 1) needed x1(s - 1) of
//     EVERY ih workgroup, each of which needed gh1(s - 1) of its hh workgroup -- published in that workgroup's step s - 2, behind the drain
//     of cI(s): the value is there, no poll.  Only the first two steps of a launch lack that chain (their gh comes from the saved
//     state): a launch leaves the sentinel in the entries of its T1 and T1 + 1, and the next launch's ih workgroups poll those as usual.
//   * the gh words {r, z, n, tag = consuming step + 1} carry a tag instead of relying on a sentinel: no re-arm, and two ring entries
//     suffice (the hh workgroup cannot start gh(t+3) before the ih workgroup has consumed gh(t+1): it needs h(t+2), which needs gh(t+2)).
// A launch ends with every entry in the state the next step expects (kernel end drains everything): continuations need no refill.
#include <type_traits>

#include "wrnn_ring.h"

namespace wrnn {

// A/B build switches (measured: profiles/r04c_*, r04g_*, r04p_*)
#ifndef DUO_XR_FIRST
#define DUO_XR_FIRST 1                       // 1 = the residual sum (on a slot's chain) is published before h (read a step later)
#endif

#ifndef DUO_ABLATE
#define DUO_ABLATE 0                          // TIMING EXPERIMENTS ONLY (wrong results): bit 0 = fc stages without their MFMAs, bit 1 = gh stages with one
#endif                                        // tile of three, bit 2 = hh stages load ONE fragment of eight, bit 3 = gate stages with one tile of three, bit 4 = no GRU
                                              // pointwise math, bit 5 = ih stages load one fragment of eight (register-load variant)
#ifndef DUO_EI_DEPTH
#define DUO_EI_DEPTH 4                        // slots in flight from which the ih workgroups ISSUE a stage's operand loads in front of the pending back half (launch_duo; measured: r06v)
#endif

constexpr int DNWGC = 4 * LNJ;               // workgroups per cluster (128)
#ifndef DUO_GH_SHIFT
#define DUO_GH_SHIFT 1                        // hh workgroups: the gh stage of the last slot runs at the top of the next step (see duo_hh's step loop)
#endif
constexpr int DLOGS = 36;                    // as LOGS of wrnn_loop.hip
constexpr int DPART = 2 * NW * 3 * 256;      // two ping-pong sets of [wave][tile 0..2][lane][4]
// saved state of an ih workgroup's slot in global memory (the slot layout of wrnn_loop.hip, LGRP floats): [0, 768) gh(t1) of the
// finished launch, [768, 1024) h, [1024, 1040) x_{t1-1} (A-ih), [1040, 1072) segment table
static_assert(O_HOWN == 768 && O_XS == 1024 && O_SP == 1040, "saved state layout");

struct DuoLds {
    int off_h, off_seg, off_xs, off_part, off_log, off_misc, off_prof, off_f3, total;
};
__host__ __device__ inline DuoLds duo_lds(int G)
{
    DuoLds l;
    int o = 0;
    l.off_h = o;    o += G * 256;            // h of the owned (unit, segment), per slot (thread-private words)
    l.off_seg = o;  o += G * 48;             // ints: [slot][16 positions | 16 limits | 16 table-row bases (this slab's per-segment aux tables; rnn1's hh workgroups: mel offsets)]
    l.off_xs = o;   o += G * 16;             // A-ih: x_{t0-1} of a continuing launch, x_{t1-1} at its end
    l.off_part = o; o += DPART;
    l.off_log = o;  o += SEG * DLOGS;
    l.off_misc = o; o += 2 * LMAXG + 2 * DNWGC;      // [2 i], [2 i + 1]: first segment / count of slot i; then the placement table (ints)
    l.off_prof = o; o += 2 * 16;             // [16] u64 phase clocks (profiling builds)
    o = (o + 3) & ~3;
    l.off_f3 = o;   o += SEG * LDC;          // sampling workgroups: MOL fc3's first tile in A-fragment order (XT floats: its L2 latency off the slot's
                                             // chain); RAW the gathered logits as [segment][class] rows of stride LDC
    l.total = o;
    return l;
}

#define DPARTOF(q) (PART + (q) * (NW * 3 * 256))
#define PHX(k)                                                                 \
    do {                                                                       \
        if (PROF && tid == 0) {                                                \
            const u64 now_ = __builtin_amdgcn_s_memtime();                     \
            PROFL[k] += now_ - plast;                                          \
            plast = now_;                                                      \
            if (trp && t >= tr_t0 && t < tr_t0 + 3 && ntr < 510)               \
                trp[1 + ntr++] = (now_ << 16) | (u64)(((tri & 15) << 8) | ((k) & 255));   \
        }                                                                      \
    } while (0)
// TRACE (profiling builds, wrnn_options.tuning bit 14): every PHX of three steps in the middle of a launch is also written out as
// (shader clock << 16 | slot << 8 | segment code) to phase_clocks[256 * 32 + block * 512 ...] (word 0: the count) -- the time line of a
// workgroup's stages, to be laid beside the other workgroup of its CU (scripts/gpu_duo_trace.py)
#define TRACE_DECL                                                                                                   \
    u64 *const trp = (PROF && (a.tuning & 16384) && a.prof) ? a.prof + 256 * 32 + (size_t)blockIdx.x * 512 : nullptr; \
    int ntr = 0, tri = 0;                                                                                            \
    const int tr_t0 = a.t0 + (a.t1 - a.t0) / 2;

// This wave's 8 fragments of a layer (byte offset soff) -> its 8 KB of LDS at `dst`, lane-linear per fragment (dst + r * 1 KB + lane * 16):
// LDS-DMA, no VGPR is written.  The instruction offset (12 bits: < 4 KB) is added to BOTH addresses, hence two base pairs.
typedef __attribute__((address_space(3))) void lds_void;
__device__ __forceinline__ void prefetch8(__amdgpu_buffer_rsrc_t rs, float *dst, int voff, int soff)
{
    lds_void *d0 = (lds_void *)dst, *d1 = (lds_void *)(dst + 1024);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, d0, 16, voff, soff, 0, 16 /* sc1 */);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, d0, 16, voff, soff, 1024, 16);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, d0, 16, voff, soff, 2048, 16);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, d0, 16, voff, soff, 3072, 16);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, d1, 16, voff + 4096, soff, 0, 16);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, d1, 16, voff + 4096, soff, 1024, 16);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, d1, 16, voff + 4096, soff, 2048, 16);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, d1, 16, voff + 4096, soff, 3072, 16);
}

// What every role needs of the launch geometry: slot i <-> group cl + ncl * i of the round
struct DuoGeo {
    int nact;
    u64 nbpack;                              // segment count of slot i in byte i
};

// ---------------------------------------------------------------------------------------------------------------------------------
// ih workgroup: LA = rnn1 (W_ih + fc1; its gi has the x_{t-1} term) or rnn2 (W_ih + fc2).  Owns the GRU state of its 16 units (h, the
// gate pointwise math) and publishes h / the residual sum / relu(fc).
// A stage starts with the previous stage's back half (publish first: faster or equal at every depth, profiles/r04g_probe_*.json).
// LP: the stage's operand fragments are fetched INTO LDS one stage ahead (buffer_load ... lds, no register) instead of being loaded into
// registers behind that back half -- round 6; which one a launch runs: launch_duo.
// EI (register-load form only): the loads are ISSUED in front of the pending back half -- i.e. in front of its publish stores: vmcnt retires in order, a
// load issued behind a write-through store waits ~1 us for that store's acknowledgement -- and looked at behind it, as before (round 6).
// PROF (thread 0, shader clocks per segment of a stage, [gates: 0-7, fc: 8-15]): 0 front issue, 1 barrier wait, 2 back half, 3 operand
// wait / poll, 4 ring hygiene, 5 MFMA tiles + partial writes, 6 stages, 7 stages whose operand was not there at the first look
// ---------------------------------------------------------------------------------------------------------------------------------
template <int MODE, bool LA, bool LP, bool EI, bool PROF>
__device__ __forceinline__ void duo_ih(const LoopArgs &a, float *smem, const int cl, const int J, const int ncl, const bool loc_h, const bool loc_y)
{
    const int G = a.G;
    const DuoLds L = duo_lds(G);
    float *HS = smem + L.off_h, *XS = smem + L.off_xs, *PART = smem + L.off_part;
    int *SEGT = reinterpret_cast<int *>(smem + L.off_seg);
    int *GEO = reinterpret_cast<int *>(smem + L.off_misc);
    u64 *PROFL = reinterpret_cast<u64 *>(smem + L.off_prof);
    u64 plast = 0;
    TRACE_DECL
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fi = lane & 15, kq = lane >> 4;
    const int kbase_lane = KCH * w + 4 * kq;
    const int pu = 4 * w + (tid & 3), pj = (tid >> 2) & 15;   // pointwise role: (owned unit, segment)
    const int prow = LU * J + pu;
    const int T0 = a.t0, T1 = a.t1;
    const int NR = a.Btot, NGR = a.NG;
    // (everything the stage lambdas need of the launch arguments as local values: they must not hold a reference to the argument
    // struct -- with one the compiler was seen to copy the whole struct to scratch and read every field from there)
    unsigned *const status = a.status;
    float *const state = a.state;
    u64 *const profp = a.prof;
    const float *const bhhp = LA ? a.b_hh1 : a.b_hh2;
    const int resume = a.resume, hop = a.hop;
    const int zrow = a.Nall * a.tab_fps;
    constexpr int L_H = LA ? 0 : 1, L_XR = LA ? 5 : 6, L_Y = LA ? 2 : 3, L_GH = LA ? 8 : 12;       // layers this role publishes / reads gh from
    constexpr int L_P0 = LA ? 4 : 5, L_P2 = LA ? 6 : 2;                                            // layers its stages poll (rnn1's gates: cI, formed in the loop by rnn1's hh workgroups)

    float A_ih[3][AF], A_fc[AF];
#pragma unroll
    for (int g = 0; g < 3; ++g) load_afrag(A_ih[g], LA ? a.w_ih1 : a.w_ih2, LA ? H : H + AUX, g * H + LU * J + fi, true, kbase_lane);
    load_afrag(A_fc, LA ? a.fc1_w : a.fc2_w, H + AUX, LU * J + fi, true, kbase_lane);
    // constants of the pointwise role (rnn1: b_ih1, the x_{t-1} vector u1 = W_ih1 . w0, w0 of the owned unit; rnn2's b_ih2 is inside c2f)
    float cb_r = 0.f, cb_z = 0.f, cb_n = 0.f, ux_r = 0.f, ux_z = 0.f, ux_n = 0.f, w0o = 0.f;
    if constexpr (LA) {
        cb_r = a.b_ih1[prow]; cb_z = a.b_ih1[H + prow]; cb_n = a.b_ih1[2 * H + prow];
        ux_r = a.u1[prow]; ux_z = a.u1[H + prow]; ux_n = a.u1[2 * H + prow];
        w0o = a.I_w0[prow];
    }

    for (int q = tid; q < L.total; q += NT) smem[q] = 0.f;
    __syncthreads();
    DuoGeo geo;
    geo.nact = 0; geo.nbpack = 0;
    // saved state: the slot layout of wrnn_loop.hip ([cluster][2 J + layer][slot][LGRP])
    const size_t state_wg = ((size_t)(cl * LNWGC + 2 * J + (LA ? 0 : 1)) * G) * LGRP;
    for (int i = 0; i < G; ++i) {
        const int g = cl + ncl * i;
        if (g >= NGR) break;
        geo.nact = i + 1;
        const int b0 = (int)(((long)g * NR) / NGR), nb = (int)(((long)(g + 1) * NR) / NGR) - b0;
        if (tid == 0) { GEO[2 * i] = a.rb0 + b0; GEO[2 * i + 1] = nb; }
        geo.nbpack |= (u64)(unsigned)nb << (8 * i);
        if (a.resume) {
            const float *sg = a.state + state_wg + (size_t)i * LGRP;
            HS[i * 256 + tid] = sg[O_HOWN + tid];
            if (tid < SEG) XS[i * 16 + tid] = sg[O_XS + tid];
            if (tid < 2 * SEG) SEGT[i * 48 + tid] = reinterpret_cast<const int *>(sg + O_SP)[tid];
        } else if (tid < SEG) {                         // fatchord_version.py:194-196: h1 = h2 = 0, x = 0 (LDS is zero)
            const int sc = a.rb0 + b0 + (tid < nb ? tid : nb - 1);
            SEGT[i * 48 + tid] = a.seg_pos[sc];
            SEGT[i * 48 + SEG + tid] = a.seg_lim[sc];
        }
    }
    __syncthreads();
    for (int i = 0; i < geo.nact; ++i) {                // this slab's table rows of every segment (after the table above is in place)
        if (tid < SEG) {
            const int nb = (int)((geo.nbpack >> (8 * i)) & 255u);
            const int sc = GEO[2 * i] + (tid < nb ? tid : nb - 1);
            SEGT[i * 48 + 2 * SEG + tid] = sc * a.tab_fps - (SEGT[i * 48 + tid] + a.tab_t0) / a.hop;
        }
    }
    __syncthreads();
    const int nact = geo.nact;
    const u64 nbpack = geo.nbpack;

    const __amdgpu_buffer_rsrc_t xrs = make_rsrc(a.xbuf, (unsigned)(DXBUF_FLOATS * 4));
    const __amdgpu_buffer_rsrc_t crs = make_rsrc(a.c2f, 0x7FFFF000u);                              // rnn2: per-frame table of its aux columns + b_ih2
    const __amdgpu_buffer_rsrc_t frs = make_rsrc(LA ? a.c3f : a.c4f, 0x7FFFF000u);                 // per-frame table of fc1 / fc2
    const int voff_frag = frag_off(w, 0, lane) * 4;     // this lane's first fragment of a layer (bytes)
    const int voff_own = (((J * 64) + 16 * w + pj) * 4 + (tid & 3)) * 4;      // the layer word of (owned unit pu, segment pj) = its publish position
    const int voff_gh = (256 * (J & 7) + tid) * 16;     // this thread's {r, z, n, tag} word in layer L_GH + (J >> 3)
    const int cbase = cl * DSLOTB;                      // slot i: cbase + i * MAXCL * DSLOTB
    const unsigned magic = a.hop_magic;
    const int mshift = a.hop_shift;

    const int locbits = (loc_h ? 1 : 0) | (loc_y ? 2 : 0);                          // by re-arm lane group: h | y | residual sum (never local)
    // Round 6 (profiles/r06y_sweep.log, r06aa_reprobe.log; A/B: wrnn_options.tuning bit 7 / bit 3 switch them OFF): the residual input word of the owned
    // unit is requested BEHIND the operand check (beside the operand loads it was the sentinel whenever the layer had not arrived yet, and the back half
    // then paid a whole round trip for it: 15.4 vs 16.4 us per step at 2 slots), and rnn1 asks for x_{t-1} AGAIN at the top of the stage that runs the
    // pending back half (22.5 vs 22.8 us at 4 slots).  Also measured, no effect: a static wave priority for the ih workgroups (s_setprio 1 .. 3), the
    // back halves at priority 3; the hh workgroups at a higher priority cost 4.5 %.
    const bool late_xo = (a.tuning & 128) == 0, reprobe = (a.tuning & 8) == 0;
    bool dead = false;
    int pp = 0;
    int t = T0;
    int cur = 0;

    // what a stage's front leaves for its back half, one stage later
    struct Carry {
        float c0, c1, c2;                               // rnn2 gates: c2f row (aux columns + b_ih2); fc: c3f / c4f value
        u32x4 gw;                                       // gh word of the slot (gates)
        unsigned xo, xt;                                // residual input word; x_{t-1} word (rnn1)
        int i, pp, t;
    };
    Carry cy;
    cy.c0 = cy.c1 = cy.c2 = 0.f; cy.gw = u32x4{0u, 0u, 0u, 0u}; cy.xo = cy.xt = 0u; cy.i = 0; cy.pp = 0; cy.t = T0;
    // (Round 6, measured and dropped -- profiles/r06ab_defer.log: rnn1's gates back half DEFERRED behind the next gate stage's front when its x_{t-1} was
    // not there -- wave 0 deciding for the workgroup through a word in LDS -- instead of waited for: 15.5 vs 16.4 us per step at 2 slots, but 23.8 vs
    // 23.5 at 4 and 42.7 vs 41.4 at 8, and the code alone cost 0.7 us at 4 slots and 3 us at 8 against the kernel without it.)

    auto slot_nb = [&](int i) -> int { return (int)((nbpack >> (8 * i)) & 255u); };

    // ---------------- back half of a gate stage: 4-wave partial sum, GRU cell pointwise (ATen gru_cell) -> publish h and the residual sum
    auto back_gates = [&](const Carry &c) {
        const int bi = c.i, bt = c.t;
        const float *PB = DPARTOF(c.pp);
        const int nb = slot_nb(bi);
        const int sb = cbase + bi * (MAXCL * DSLOTB) + (bt & (DRING - 1)) * XTB;
        const bool live = pj < nb;
        lds_barrier();
        PHX(cur + 1);
        const float pr = get_partial<3>(PB, 0, pu, pj), pz = get_partial<3>(PB, 1, pu, pj), pn = get_partial<3>(PB, 2, pu, pj);
        const float hprev = HS[bi * 256 + tid];
        float ghr, ghz, ghn;
        u32x4 g = c.gw;
        if (bt > T0) {                                  // gh(t) from the hh workgroup of the same unit block (published during step t - 1)
            const unsigned tag = (unsigned)bt + 1u;
            if (__builtin_expect(__any(live && g.w != tag), 0))
                wait_for([&] { return !__any(live && g.w != tag); },
                         [&] { g = __builtin_amdgcn_raw_buffer_load_b128(xrs, voff_gh, cbase + bi * (MAXCL * DSLOTB) + (L_GH + (J >> 3)) * DLAYERB + (bt & (DGHRING - 1)) * XTB, 16 /* sc1 */); },
                         status, dead, 0x500u | (LA ? 0u : 8u) | 6u, bt);
            ghr = __uint_as_float(g.x); ghz = __uint_as_float(g.y); ghn = __uint_as_float(g.z);
        } else if (resume) {                          // first step of a continuing launch: gh(t0), saved by the launch that ended there
            const float *sg = state + state_wg + (size_t)bi * LGRP;
            ghr = sg[tid]; ghz = sg[256 + tid]; ghn = sg[512 + tid];
        } else {                                        // t = 0: gh = W_hh . 0 + b_hh = b_hh
            ghr = bhhp[prow]; ghz = bhhp[H + prow]; ghn = bhhp[2 * H + prow];
        }
        float gir, giz, gin, xin;
        unsigned xw = c.xo;                             // the GRU input of the owned unit (cI / x1): another wave's quarter of the layer, normally validated long ago
        if (__builtin_expect(__any(live && xw == SENT), 0))
            wait_for([&] { return !__any(live && xw == SENT); },
                     [&] { xw = __builtin_amdgcn_raw_buffer_load_b32(xrs, voff_own, sb + L_P0 * DLAYERB, 16 /* sc1 */); },
                     status, dead, 0x528u, bt);
        if constexpr (LA) {
            float xv;
            if (bt > T0) {                              // x_{t-1}, sampled by a B-hh workgroup
                unsigned xt = c.xt;
                if (__builtin_expect(__any(live && xt == SENT), 0))
                    wait_for([&] { return !__any(live && xt == SENT); },
                             [&] { xt = __builtin_amdgcn_raw_buffer_load_b32(xrs, pj * 4, cbase + bi * (MAXCL * DSLOTB) + 7 * DLAYERB + ((bt - 1) & (DRING - 1)) * XTB, 16 /* sc1 */); },
                             status, dead, 0x520u, bt);
                xv = __uint_as_float(xt);
            } else xv = XS[bi * 16 + pj];
            gir = pr + fmaf(xv, ux_r, cb_r); giz = pz + fmaf(xv, ux_z, cb_z); gin = pn + fmaf(xv, ux_n, cb_n);
            xin = fmaf(w0o, xv, __uint_as_float(xw));               // xi of the owned unit (:208-209)
        } else {
            gir = pr + c.c0; giz = pz + c.c1; gin = pn + c.c2;
            xin = __uint_as_float(xw);
        }
        // MoL: hardware exp / rcp (inside the 1e-5 tolerance); RAW (class indices compared bit for bit): the library forms
        const float hn = (DUO_ABLATE & 16) ? 0.25f * (gir + giz) + 0.1f * (gin + ghr + ghz + ghn) + 0.5f * hprev
                         : MODE == 1 ? gru_update_fast(gir, giz, gin, ghr, ghz, ghn, hprev) : gru_update(gir, giz, gin, ghr, ghz, ghn, hprev);
        HS[bi * 256 + tid] = hn;
        if (DUO_XR_FIRST) publish4l(xrs, sb + L_XR * DLAYERB + J * 1024, tid, xin + hn, live, false);    // x1 = xi + h1 (:212) / x2 = x1 + h2 (:216): to the other XCD, written through
        publish4l(xrs, sb + L_H * DLAYERB + J * 1024, tid, hn, live, loc_h);
        if (!DUO_XR_FIRST) publish4l(xrs, sb + L_XR * DLAYERB + J * 1024, tid, xin + hn, live, false);
        PHX(cur + 2);
    };
    // ---------------- back half of an fc stage: fc1 / fc2 + relu -> publish y1 / y2
    auto back_relu = [&](const Carry &c) {
        const float *PB = DPARTOF(c.pp);
        const int sb = cbase + c.i * (MAXCL * DSLOTB) + (c.t & (DRING - 1)) * XTB;
        lds_barrier();
        PHX(cur + 1);
        const float yv = fmaxf(get_partial<3>(PB, 0, pu, pj) + c.c0, 0.f);
        publish4l(xrs, sb + L_Y * DLAYERB + J * 1024, tid, yv, pj < slot_nb(c.i), loc_y);
        PHX(cur + 2);
    };

    // LDS prefetch (LP): the operand fragments of the NEXT stage -- (gates, i + 1) ... (gates, n - 1), (fc, 0) ... (fc, n - 1), then
    // the next step's (gates, 0) -- are requested into this wave's 8 KB of the (here unused) fc3 region while the current stage's MFMA tiles run,
    // and read back with 8 ds_read_b128 at the top of their stage: the L2 / fabric latency of a stage's operand (and the write-through
    // acknowledgements of the stores in front of it: vmcnt retires in order) leaves the workgroup's serial instruction stream, at no register.
    // Protocol-wise the request sits where the next stage's loads used to be issued, minus one MFMA block and one back half in which this
    // workgroup polls nothing another workgroup re-arms: whatever made the entry safe to load there makes it safe here (header; a word
    // that is still the sentinel is polled for as before).
    static_assert(SEG * LDC >= NW * 2048, "the prefetch blocks fit the fc3 region");
    float *const PFW = smem + L.off_f3 + w * 2048;
    auto prefetch_next = [&](auto PHC, int i) {
        constexpr int ph = decltype(PHC)::value;
        const int ring = t & (DRING - 1);
        int nso;
        if constexpr (ph == 0) nso = (i + 1 < nact) ? cbase + (i + 1) * (MAXCL * DSLOTB) + ring * XTB + L_P0 * DLAYERB : cbase + ring * XTB + L_P2 * DLAYERB;
        else nso = (i + 1 < nact) ? cbase + (i + 1) * (MAXCL * DSLOTB) + ring * XTB + L_P2 * DLAYERB
                                  : (t + 1 < T1 ? cbase + ((t + 1) & (DRING - 1)) * XTB + L_P0 * DLAYERB : -1);
        if (nso >= 0) prefetch8(xrs, PFW, voff_frag, nso);
    };
    if constexpr (LP) {
        if (T0 < T1) prefetch8(xrs, PFW, voff_frag, cbase + (T0 & (DRING - 1)) * XTB + L_P0 * DLAYERB);
    }

    // A STAGE (phase ph = 0 gates / 2 fc, slot i), BK = the kind of the pending back half (0 none / 1 gates / 2 relu; compile-time)
    auto stage = [&](auto PHC, auto BKC, int i) {
        constexpr int ph = decltype(PHC)::value;
        constexpr int BK = decltype(BKC)::value;
        const int nb = slot_nb(i);
        const int ring = t & (DRING - 1);
        const int sbase = cbase + i * (MAXCL * DSLOTB);
        const int sb = sbase + ring * XTB;
        u32x4 x[8];
        Carry nc;
        nc.c0 = nc.c1 = nc.c2 = 0.f; nc.gw = u32x4{0u, 0u, 0u, 0u}; nc.xo = nc.xt = 0u;
        nc.i = i; nc.pp = pp; nc.t = t;
        if (PROF && tid == 0 && plast == 0) plast = __builtin_amdgcn_s_memtime();
        tri = cy.i;
        if constexpr (LP) {
            // this stage's fragments were requested into LDS one stage ago (prefetch8 below): everything older is out too -- the
            // previous stage's small loads (its back half, next, reads them) and, at the top of a step, last step's re-arm stores
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int r = 0; r < 8; ++r) x[r] = *reinterpret_cast<const u32x4 *>(PFW + r * 256 + lane * 4);
            if (PROF) {                                 // (profiling builds: the time at the top of a stage counts as its "issue" segment)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                cur = ph == 0 ? 0 : 8;
                PHX(cur + 0);
            }
        }
        if constexpr (LA && BK == 1) {
            // the pending gates back half needs x_{t-1} of its slot: the word requested in ITS front is ~one MFMA block old -- ask again now, under the
            // barrier and the partial sums, instead of finding the stale sentinel behind them and paying a whole round trip there
            if (reprobe && cy.t > T0)
                cy.xt = __builtin_amdgcn_raw_buffer_load_b32(xrs, pj * 4, cbase + cy.i * (MAXCL * DSLOTB) + 7 * DLAYERB + ((cy.t - 1) & (DRING - 1)) * XTB, 16 /* sc1 */);
        }
        if constexpr (!LP && EI) {
            const int so0 = sb + (ph == 0 ? L_P0 : L_P2) * DLAYERB;
#pragma unroll
            for (int r = 0; r < 8; ++r) x[r] = __builtin_amdgcn_raw_buffer_load_b128(xrs, voff_frag + r * 1024, so0, 16 /* sc1 */);
        }
        if constexpr (BK == 1) back_gates(cy);
        if constexpr (BK == 2) back_relu(cy);
        cur = ph == 0 ? 0 : 8;
        tri = i;
        // ---------------- front: this stage's loads ----------------
        int soff_x;                                     // where the operand fragments come from (for the re-load of a poll)
        if constexpr (ph == 0) {
            soff_x = sb + L_P0 * DLAYERB;
            if constexpr (!LP && !EI) {
#pragma unroll
                for (int r = 0; r < 8; ++r) x[r] = ((DUO_ABLATE & 32) && r > 0) ? x[0] : __builtin_amdgcn_raw_buffer_load_b128(xrs, voff_frag + r * 1024, soff_x, 16 /* sc1 */);
            }
            if (!late_xo) nc.xo = __builtin_amdgcn_raw_buffer_load_b32(xrs, voff_own, soff_x, 16 /* sc1 */);
            if constexpr (LA) {
                if (t > T0) nc.xt = __builtin_amdgcn_raw_buffer_load_b32(xrs, pj * 4, sbase + 7 * DLAYERB + ((t - 1) & (DRING - 1)) * XTB, 16 /* sc1 */);
            } else {
                const int fr = table_row(SEGT[i * 48 + pj] + t, SEGT[i * 48 + SEG + pj], SEGT[i * 48 + 2 * SEG + pj], magic, mshift, hop, zrow);
                const int vo = (fr * 3 * H + prow) * 4;
                nc.c0 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(crs, vo, 0, 0));
                nc.c1 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(crs, vo, H * 4, 0));
                nc.c2 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(crs, vo, 2 * H * 4, 0));
            }
            if (t > T0) nc.gw = __builtin_amdgcn_raw_buffer_load_b128(xrs, voff_gh, sbase + (L_GH + (J >> 3)) * DLAYERB + (t & (DGHRING - 1)) * XTB, 16 /* sc1 */);
        } else {
            soff_x = sb + L_P2 * DLAYERB;
            if constexpr (!LP && !EI) {
#pragma unroll
                for (int r = 0; r < 8; ++r) x[r] = ((DUO_ABLATE & 32) && r > 0) ? x[0] : __builtin_amdgcn_raw_buffer_load_b128(xrs, voff_frag + r * 1024, soff_x, 16 /* sc1 */);
            }
            const int fr = table_row(SEGT[i * 48 + pj] + t, SEGT[i * 48 + SEG + pj], SEGT[i * 48 + 2 * SEG + pj], magic, mshift, hop, zrow);
            nc.c0 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(frs, (fr * H + prow) * 4, 0, 0));
        }
        PHX(cur + 0);
        // ---------------- operands ----------------
        {
            const bool live = fi < nb;
            const bool there = frag_there(x, live);
            if (PROF && tid == 0) { PROFL[cur + 6] += 1; PROFL[cur + 7] += !there; }
            if (__builtin_expect(!there, 0))
                wait_for([&] { return frag_there(x, live); },
                         [&] {
#pragma unroll
                             for (int r = 0; r < 8; ++r) x[r] = __builtin_amdgcn_raw_buffer_load_b128(xrs, voff_frag + r * 1024, soff_x, 16 /* sc1 */);
                         },
                         status, dead, 0x500u | (LA ? 0u : 8u) | (unsigned)ph, t);
        }
        PHX(cur + 3);
        if (ph == 0 && late_xo) nc.xo = __builtin_amdgcn_raw_buffer_load_b32(xrs, voff_own, soff_x, 16 /* sc1 */);      // (this wave's quarter is there: the other waves' usually too)
        if (ph == 0 && i == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // ring hygiene: last step's re-arm stores are out before anything of this step is published
        if (ph == 2 && i == nact - 1) {
            // ring hygiene, once per step, after the last layer this workgroup polls in the step has arrived (see the header): re-arm this
            // wave's own words of entry (t + 2) % 4 in the three layers it publishes, for every slot (drained at the top of the next step)
            const int which = lane >> 4;
            const int layer = which == 0 ? L_H : (which == 1 ? L_Y : L_XR);
            const bool lloc = ((locbits >> which) & 1) != 0;          // (NOT a select between the captured flags: a select of pointers to
                                                                      // captured locals keeps the whole closure -- every local -- in scratch)
            const int vo = layer * DLAYERB + J * 1024 + w * 256 + (lane & 15) * 16;
            const int so = cbase + ((t + DAHEAD_IH) & (DRING - 1)) * XTB;
            const u32x4 q = {SENT, SENT, SENT, SENT};
            if (which < 3) {
#pragma unroll 1
                for (int i2 = 0; i2 < nact; ++i2) {
                    if (lloc) __builtin_amdgcn_raw_buffer_store_b128(q, xrs, vo, so + i2 * (MAXCL * DSLOTB), 0);
                    else __builtin_amdgcn_raw_buffer_store_b128(q, xrs, vo, so + i2 * (MAXCL * DSLOTB), 16 /* sc1 */);
                }
            }
        }
        if constexpr (LP) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // (the fragments are in registers: the wave's LDS block is free)
            prefetch_next(PHC, i);
        }
        PHX(cur + 4);
        float b[32];
        frag_to_b(x, b);
        float *PW = DPARTOF(pp);
        if constexpr (ph == 0) {
            f32x4 o0, o1, o2;
            if constexpr ((DUO_ABLATE & 8) != 0) { o0 = mfma1(A_ih[0], b); o1 = o0; o2 = o0; }
            else mfma3s(A_ih[0], A_ih[1], A_ih[2], b, o0, o1, o2);
            put_partial<3>(PW, w, 0, lane, o0);
            put_partial<3>(PW, w, 1, lane, o1);
            put_partial<3>(PW, w, 2, lane, o2);
        } else {
            if constexpr ((DUO_ABLATE & 1) != 0) put_partial<3>(PW, w, 0, lane, f32x4{b[0], b[1], b[2], b[3]});
            else put_partial<3>(PW, w, 0, lane, mfma1(A_fc, b));
        }
        PHX(cur + 5);
        cy = nc;
        pp ^= 1;
    };

    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    // (Round 6, measured and dropped -- profiles/r06z_lag.log: the fc stage of slot k interleaved `lag` gate stages behind its gate stage, g0 g1 f0 g2 f1 ..:
    // 25.4 us per step at 4 slots with lag 2, 23.6 with lag 3 against 23.8 for this order in the same build; that build took the kind of the pending
    // back half as a run-time value -- two call sites instead of five -- which alone cost 1.1 us per step at 4 slots and 4 us at 8.)
    for (; t < T1; ++t) {
        if (t == T0) stage(I0{}, I0{}, 0);
        else stage(I0{}, I2{}, 0);
#pragma unroll 1
        for (int i = 1; i < nact; ++i) stage(I0{}, I1{}, i);
        stage(I2{}, I1{}, 0);
#pragma unroll 1
        for (int i = 1; i < nact; ++i) stage(I2{}, I2{}, i);
    }
    cur = 8;
    back_relu(cy);
    // ---- what the next launch of this round needs from the ring: gh(T1) of every slot (published during step T1 - 1) -> the saved
    //      state in global memory, and, for rnn1, x_{T1-1}
#pragma unroll 1
    for (int i = 0; i < nact; ++i) {
        const int nb = slot_nb(i);
        const bool live = pj < nb;
        const int sbase = cbase + i * (MAXCL * DSLOTB);
        const int so = sbase + (L_GH + (J >> 3)) * DLAYERB + (T1 & (DGHRING - 1)) * XTB;
        u32x4 gq = __builtin_amdgcn_raw_buffer_load_b128(xrs, voff_gh, so, 16 /* sc1 */);
        const unsigned tag = (unsigned)T1 + 1u;
        wait_for([&] { return !__any(live && gq.w != tag); },
                 [&] { gq = __builtin_amdgcn_raw_buffer_load_b128(xrs, voff_gh, so, 16 /* sc1 */); }, status, dead, 0x521u, T1);
        float *sg = state + state_wg + (size_t)i * LGRP;
        sg[tid] = __uint_as_float(gq.x); sg[256 + tid] = __uint_as_float(gq.y); sg[512 + tid] = __uint_as_float(gq.z);
        sg[O_HOWN + tid] = HS[i * 256 + tid];
        if (tid < 2 * SEG) reinterpret_cast<int *>(sg + O_SP)[tid] = SEGT[i * 48 + tid];
        if constexpr (LA) {
            const int sx = sbase + 7 * DLAYERB + ((T1 - 1) & (DRING - 1)) * XTB;
            unsigned v = __builtin_amdgcn_raw_buffer_load_b32(xrs, fi * 4, sx, 16 /* sc1 */);
            wait_for([&] { return !__any(fi < nb && v == SENT); },
                     [&] { v = __builtin_amdgcn_raw_buffer_load_b32(xrs, fi * 4, sx, 16 /* sc1 */); }, status, dead, 0x522u, T1);
            if (tid < SEG) sg[O_XS + tid] = (tid < nb) ? __uint_as_float(v) : 0.f;
        }
    }
    if (PROF && tid == 0 && profp) {
        for (int k = 0; k < 16; ++k) profp[(size_t)(blockIdx.x & 255) * 32 + k] += PROFL[k];
        if (trp) trp[0] = (u64)ntr;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// hh workgroup: gh(t+1) = W_hh . h(t) + b_hh of its 16 units for every slot (off the critical path); rnn2's workgroup J, for slot J:
// fc3 + mixture-of-logistics sampling (utils/distribution.py:87-123); rnn1's workgroups: the I-layer conditioning cI(t+1) of their 16
// rows for every slot, formed here from the up-sampled mel and the frame's aux (cond_tile: one wave, 28 MFMAs) and published through
// ring layer 4 one step ahead -- nothing of the conditioning is materialised per call.  Keeps no state between launches.
// PROF: as duo_ih, [gh stages: 0-7, the sampling stage: 8-15]
// ---------------------------------------------------------------------------------------------------------------------------------
template <int MODE, bool LA, bool LP, bool PROF>
__device__ __forceinline__ void duo_hh(const LoopArgs &a, float *smem, const int cl, const int J, const int ncl, const bool loc_h)
{
    const int G = a.G;
    const DuoLds L = duo_lds(G);
    float *PART = smem + L.off_part, *LOG = smem + L.off_log, *F3 = smem + L.off_f3;
    int *SEGT = reinterpret_cast<int *>(smem + L.off_seg);
    int *GEO = reinterpret_cast<int *>(smem + L.off_misc);
    u64 *PROFL = reinterpret_cast<u64 *>(smem + L.off_prof);
    u64 plast = 0;
    TRACE_DECL
    int cur = 0;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fi = lane & 15, kq = lane >> 4;
    const int kbase_lane = KCH * w + 4 * kq;
    const int pu = 4 * w + (tid & 3), pj = (tid >> 2) & 15;
    const int prow = LU * J + pu;
    const int T0 = a.t0, T1 = a.t1, C = a.C;
    const int NR = a.Btot, Nall = a.Nall, NGR = a.NG;
    unsigned *const status = a.status;                  // (local values for the stage lambdas: see duo_ih)
    u64 *const profp = a.prof;
    float *const outp = a.out, *const dbgl = a.dbg_logits;
    const float *const forcex = a.force_x, *const noise_pre = a.noise_pre, *const fc3f = a.fc3f, *const noise_raw = a.noise;
    constexpr bool MOL = MODE == 1;
    const int Tall = a.T, noise_t0 = a.noise_t0;
    const float *const mels_up = a.mels_up, *const aux_fr = a.aux_fr, *const mel_coef = a.mel_coef;
    const int mel_stage = a.mel_stage;
    const int hop = a.hop;
    const unsigned magic = a.hop_magic;
    const int mshift = a.hop_shift;
    CondTile ct;
    if constexpr (LA) cond_tile_init(ct, a.I_cT, a.I_b, J, lane);
    constexpr int L_H = LA ? 0 : 1, L_GH = LA ? 8 : 12;

    float A_hh[3][AF];
#pragma unroll
    for (int g = 0; g < 3; ++g) load_afrag(A_hh[g], LA ? a.w_hh1 : a.w_hh2, H, g * H + LU * J + fi, true, kbase_lane);

    const float *bhh = LA ? a.b_hh1 : a.b_hh2;
    const float bh_r = bhh[prow], bh_z = bhh[H + prow], bh_n = bhh[2 * H + prow];
    float b3a = 0.f, b3b = 0.f;                         // MOL: logit rows pu and 16 + pu; RAW: logit row (class) 16 J + pu of the owned block
    float A_f3[AF];                                     // RAW: fc3 rows [16 J, 16 J + 16) of rnn2's hh workgroup J (the 512 logits are a sixth exchange)
    if constexpr (!LA) {
        if constexpr (MOL) {
            b3a = a.fc3_b[pu];
            b3b = (16 + pu < 30) ? a.fc3_b[16 + pu] : 0.f;
        } else {
            b3a = a.fc3_b[prow];
            load_afrag(A_f3, a.fc3_w, H, LU * J + fi, true, kbase_lane);
        }
    }

    for (int q = tid; q < L.total; q += NT) smem[q] = 0.f;
    __syncthreads();
    int nact = 0;
    u64 nbpack = 0;
    for (int i = 0; i < G; ++i) {
        const int g = cl + ncl * i;
        if (g >= NGR) break;
        nact = i + 1;
        const int b0 = (int)(((long)g * NR) / NGR), nb = (int)(((long)(g + 1) * NR) / NGR) - b0;
        if (tid == 0) { GEO[2 * i] = a.rb0 + b0; GEO[2 * i + 1] = nb; }
        nbpack |= (u64)(unsigned)nb << (8 * i);
        if (LA && tid < SEG) {                          // segment table of the slot (the conditioning it forms)
            const int sc = a.rb0 + b0 + (tid < nb ? tid : nb - 1);
            SEGT[i * 48 + tid] = a.seg_pos[sc];
            SEGT[i * 48 + SEG + tid] = a.seg_lim[sc];
            SEGT[i * 48 + 2 * SEG + tid] = a.mel_stage ? a.seg_moff[sc] : 0;      // (the ih workgroups keep their table-row bases here)
        }
    }
    __syncthreads();
    // rnn2's hh workgroup J samples slot J (y2 comes from rnn2's ih workgroups: the same XCD under the placement above)
    // MOL: rnn2's hh workgroup J samples slot J.  RAW (round 5): FOUR workgroups per slot -- workgroup J samples segments 4 (J & 3) .. + 3 of slot
    // J >> 2, one segment per wave: the 512-class softmax -> Categorical -> argmax(p / q) of 16 segments took one workgroup ~8 us on the slot's chain
    // (4 segments per wave in lock step; profiles/r05k_probe_raw_chain.json), the other hh workgroups were waiting meanwhile
    const bool sampler = !LA && J < (MOL ? nact : 4 * nact);
    const int my_slot = MOL ? J : (J >> 2);
    const int quad = J & 3;
    if (MOL && sampler) {                               // fc3's first tile -> LDS (fragment order as in the pack)
        for (int q = tid; q < XT / 4; q += NT) reinterpret_cast<float4 *>(F3)[q] = reinterpret_cast<const float4 *>(a.fc3f)[q];
    }
    constexpr bool order1 = MOL;                        // MOL: round 6's stage order (see the step loop); RAW: round 4's
    float *const LGT = F3;                              // RAW: the gathered logits of the slot being sampled, [segment][class] rows of stride LDC
    __syncthreads();

    const __amdgpu_buffer_rsrc_t xrs = make_rsrc(a.xbuf, (unsigned)(DXBUF_FLOATS * 4));
    const int voff_frag = frag_off(w, 0, lane) * 4;
    const int voff_gh = (256 * (J & 7) + tid) * 16;
    const int cbase = cl * DSLOTB;
    auto slot_nb = [&](int i) -> int { return (int)((nbpack >> (8 * i)) & 255u); };

    const bool fresh_y2 = (a.tuning & 16) != 0;

    const bool prio_smp = (a.tuning & 32) == 0;          // the sampling stage (on its slot's chain) runs at wave priority 3 (round 6: 16.1 vs 16.4 us per step at 2 slots, 22.56 vs 22.72 at 4; A/B: tuning bit 5 = off)
    bool dead = false;
    int pp = 0;
    int t = T0;
    u32x4 x[8];
    bool xahead = false;
    struct Carry { float c0, c1; int i, pp, t; };
    Carry cy;
    cy.c0 = cy.c1 = 0.f; cy.i = 0; cy.pp = 0; cy.t = T0;

    // ---------------- back half of a gh stage: gh(t+1) of the owned (unit, segment) -> ring entry (t + 1), consumed by the ih workgroup J at step t + 1
    auto back_gh = [&](const Carry &c) {
        const float *PB = DPARTOF(c.pp);
        lds_barrier();
        PHX(cur + 1);
        const float g0 = get_partial<3>(PB, 0, pu, pj) + bh_r, g1 = get_partial<3>(PB, 1, pu, pj) + bh_z, g2 = get_partial<3>(PB, 2, pu, pj) + bh_n;
        if (pj < slot_nb(c.i)) {                        // one 16-byte word {r, z, n, tag} per (unit, segment): one store, its own flag (hand-off form R2)
            const u32x4 q = {__float_as_uint(g0), __float_as_uint(g1), __float_as_uint(g2), (unsigned)c.t + 2u};      // tag: consuming step (t + 1) + 1
            const int so = cbase + c.i * (MAXCL * DSLOTB) + (L_GH + (J >> 3)) * DLAYERB + ((c.t + 1) & (DGHRING - 1)) * XTB;
            if (loc_h) __builtin_amdgcn_raw_buffer_store_b128(q, xrs, voff_gh, so, 0);
            else __builtin_amdgcn_raw_buffer_store_b128(q, xrs, voff_gh, so, 16 /* sc1 */);
        }
        PHX(cur + 2);
    };
    // ---------------- RAW: back half of a logits stage: the owned 16 classes x 16 segments of fc3 (:223) -> ring layer 16 (read by the slot's sampler)
    auto back_lg = [&](const Carry &c) {
        const float *PB = DPARTOF(c.pp);
        lds_barrier();
        PHX(cur + 1);
        publish4l(xrs, cbase + c.i * (MAXCL * DSLOTB) + 16 * DLAYERB + (c.t & (DRING - 1)) * XTB + J * 1024, tid, get_partial<3>(PB, 0, pu, pj) + b3a,
                  pj < slot_nb(c.i), loc_h);
        PHX(cur + 2);
    };
    // ---------------- back half of the sampling stage: fc3 logits -> sample x_t (utils/distribution.py:102-121)
    auto back_sample = [&](const Carry &c) {
        const float *PB = DPARTOF(c.pp);
        const int bi = c.i, bt = c.t;
        const int nb = slot_nb(bi);
        const int b0 = GEO[2 * bi];
        lds_barrier();
        PHX(cur + 1);
        {   // 30 logit rows x 16 segments: thread (rows pu and 16 + pu, segment pj) -- the partial tiles' conflict-free reader mapping
            const float lg = get_partial<3>(PB, 0, pu, pj) + b3a;
            const float lg2 = get_partial<3>(PB, 1, pu, pj) + b3b;
            LOG[pj * DLOGS + pu] = lg;
            if (dbgl && pj < nb) dbgl[((size_t)bt * Nall + b0 + pj) * C + pu] = lg;
            if (pu < 14) {
                LOG[pj * DLOGS + 16 + pu] = lg2;
                if (dbgl && pj < nb) dbgl[((size_t)bt * Nall + b0 + pj) * C + 16 + pu] = lg2;
            }
        }
        lds_barrier();
        {   // 16-lane row = one segment (su), lane sm = mixture; c0 / c1 = this thread's pre-transformed noise
            const int su = tid >> 4, sm = tid & 15;
            float best = (sm < 10) ? mol_gumbel_pre(LOG[su * DLOGS + sm], c.c0) : -INFINITY;
            int bidx = sm;
            argmax_row16(best, bidx);
            if (sm == 0 && su < nb) {
                float xv = mol_sample_pre(LOG[su * DLOGS + 10 + bidx], LOG[su * DLOGS + 20 + bidx], c.c1);
                outp[(size_t)(b0 + su) * Tall + bt] = xv;
                if (forcex) xv = forcex[(size_t)(b0 + su) * Tall + bt];
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(xv), xrs, su * 4, cbase + bi * (MAXCL * DSLOTB) + 7 * DLAYERB + (bt & (DRING - 1)) * XTB, 16 /* sc1 */);
            }
        }
        if (prio_smp) __builtin_amdgcn_s_setprio(0);
        PHX(cur + 2);
    };
    // ---------------- rnn1: cI(tt) of the owned 16 rows, slots w, w + 4 by wave w (no barrier: a wave forms, publishes and later re-arms its own slots)
    auto cond_step = [&](int tt) {
        if constexpr (LA) {
#pragma unroll 1
            for (int i = w; i < nact; i += NW) {
                const int p = SEGT[i * 48 + fi] + tt;
                const bool valid = fi < slot_nb(i) && p < SEGT[i * 48 + SEG + fi];
                const int fr = magic ? (int)(__umulhi((unsigned)p, magic) >> mshift) : p / hop;
                f32x4 v;
                if (mel_stage) {                        // the last up-sampling stage here too: three rows of its input, this position's tap sums
                    const int j = p + SEGT[i * 48 + 2 * SEG + fi];
                    const int row = j / LAST_SCALE;
                    v = cond_tile_rows(ct, mels_up + (size_t)(row - 1) * MEL, mel_coef + 3 * (j - row * LAST_SCALE), aux_fr + (size_t)fr * (4 * AUX), valid, lane);
                } else v = cond_tile(ct, mels_up + (size_t)p * MEL, aux_fr + (size_t)fr * (4 * AUX), valid, lane);
                const u32x4 q = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
                const int so = cbase + i * (MAXCL * DSLOTB) + 4 * DLAYERB + (tt & (DRING - 1)) * XTB;
                if (loc_h) __builtin_amdgcn_raw_buffer_store_b128(q, xrs, J * 1024 + lane * 16, so, 0);
                else __builtin_amdgcn_raw_buffer_store_b128(q, xrs, J * 1024 + lane * 16, so, 16 /* sc1 */);
            }
        }
    };
    // layer 4 carries no sentinel inside a launch (see the header: cI is formed two steps ahead and drained before this workgroup's next gh
    // publication, which every reader's inputs depend on).  The first two steps of a launch have no such dependency (their gh comes from
    // the saved state): the launch that ends at T1 leaves the sentinel in the entries of steps T1 and T1 + 1 -- nobody reads them any more --
    // and the next launch's ih workgroups poll them like any other layer.
    auto cond_leave = [&]() {
        if constexpr (LA) {
            const u32x4 q = {SENT, SENT, SENT, SENT};
#pragma unroll 1
            for (int i = w; i < nact; i += NW) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int so = cbase + i * (MAXCL * DSLOTB) + 4 * DLAYERB + ((T1 + e) & (DRING - 1)) * XTB;
                    if (loc_h) __builtin_amdgcn_raw_buffer_store_b128(q, xrs, J * 1024 + lane * 16, so, 0);
                    else __builtin_amdgcn_raw_buffer_store_b128(q, xrs, J * 1024 + lane * 16, so, 16 /* sc1 */);
                }
            }
        }
    };
    // Round 6: the gh block -- 96 back-to-back MFMAs, off every chain -- is BROKEN UP: a compare + scalar branch behind every three MFMAs and a 64-cycle
    // yield (s_sleep 1) behind every twelve.  f32 MFMA and VALU do not overlap on a SIMD (scripts/micro/mfma_valu_coexec.hip: the times ADD), and an
    // instruction of the ih wave beside an MFMA stream of this wave waits for the MFMA in flight: the block starts ~one hop behind the ih workgroup's
    // h publication and ran into that workgroup's NEXT back half, which is on its slot's chain (profiles/r06x_trace_d4.txt: back halves of 1.3 k cycles
    // stretched to 3-5 k).  Measured on one box (profiles/r06bb_gh_branchy.log, us per step at 2 / 4 / 8 slots): 14.4 / 20.6 / 36.0 against 15.2 / 22.3 /
    // 38.1 for the unbroken block.  Also measured: the yields pinned at compile time without the branches (no gain: r06ay), the branches without the yields
    // (21.8 / 37.5), 16-64 idle cycles (s_nop) behind every three MFMAs instead (21.4 at 4 slots, slower at 8: r06ba), the whole block held back by 1-1.5 k
    // cycles (-1.6 % at 4 slots only: r06at), static wave priorities (nothing).  wrnn_options.tuning bit 20: the unbroken block; bits 21-25: yield length / spacing (A/B).
    const bool gh_yield = (a.tuning & 0x100000) == 0 && nact >= 2;      // (one slot: 11.9 vs 11.6 us per step -- nothing of the ih workgroup to let through)
    const int gh_chunk = ((a.tuning >> 21) & 3) == 3 ? 0 : 1 + ((a.tuning >> 21) & 3), gh_gran = 3 ^ ((a.tuning >> 23) & 7);
    enum { BK_NONE = 0, BK_GH = 1, BK_SAMPLE = 2, BK_ANY = 3, BK_LG = 4 };
    int pend = BK_NONE;                                 // run-time kind of the pending half, read only where two kinds can meet (BK_ANY sites)
    int last_here = nact - 1;                           // the last gh stage in front of a step's other stages (see the step loop)
    bool deferred = false;                              // the running gh stage is the deferred one of the previous step

    // kind 1: gh stage of slot i (polls h(t)); kind 3: MOL sampling stage of my_slot (polls y2(t)); RAW: kind 2: logits stage of slot i (polls y2(t)),
    // kind 4: sampling stage of my_slot (polls the 512 logits)
    auto stage = [&](auto KC, auto BKC, int i) {
        constexpr int kind = decltype(KC)::value;
        constexpr int BK = decltype(BKC)::value;
        const int nb = slot_nb(i);
        const int ring = t & (DRING - 1);
        const int sbase = cbase + i * (MAXCL * DSLOTB);
        const int soff_x = sbase + (kind == 1 ? L_H : (kind == 4 ? 16 : 3)) * DLAYERB + ring * XTB;
        Carry nc;
        nc.c0 = nc.c1 = 0.f; nc.i = i; nc.pp = pp; nc.t = t;
        auto run_back = [&] {
            if constexpr (BK == BK_GH) back_gh(cy);
            else if constexpr (BK == BK_SAMPLE) back_sample(cy);
            else if constexpr (BK == BK_LG) back_lg(cy);
            else if constexpr (BK == BK_ANY) {
                if (pend == BK_GH) back_gh(cy);
                else if (pend == BK_SAMPLE) { if constexpr (!LA && MOL) back_sample(cy); }
                else if (pend == BK_LG) { if constexpr (!LA && !MOL) back_lg(cy); }
            }
        };
        if (PROF && tid == 0 && plast == 0) plast = __builtin_amdgcn_s_memtime();
        tri = cy.i;
        if (kind == 3 && fresh_y2) {
            // the sampling stage is on its slot's chain and its look-ahead request went out in FRONT of the gh stage's MFMA block: ask again now -- the pending
            // back half covers most of the L2 round trip -- instead of finding the old sentinel behind it and polling from there (round 6)
#pragma unroll
            for (int r = 0; r < 8; ++r) x[r] = __builtin_amdgcn_raw_buffer_load_b128(xrs, voff_frag + r * 1024, soff_x, 16 /* sc1 */);
            xahead = true;
        }
        run_back();                                     // (publish first: profiles/r04g_probe_*.json)
        cur = kind == 1 ? 0 : 8;
        tri = i;
        if (kind == 3 && prio_smp) __builtin_amdgcn_s_setprio(3);

        if (!xahead) {
#pragma unroll
            for (int r = 0; r < 8; ++r) x[r] = ((DUO_ABLATE & 4) && r > 0) ? x[0] : __builtin_amdgcn_raw_buffer_load_b128(xrs, voff_frag + r * 1024, soff_x, 16 /* sc1 */);
        }
        if constexpr (kind == 3) {
            // this step's sampling noise, pre-transformed (wrnn_noise_mol_kernel): thread (segment tid >> 4, mixture tid & 15)
            const int b0 = GEO[2 * i];
            const int su = tid >> 4, sm = tid & 15;
            const float *nrow = noise_pre + (size_t)(t - noise_t0) * 11 * Nall;
            const int suc = su < nb ? su : nb - 1;
            nc.c0 = nrow[(size_t)(b0 + suc) * 10 + (sm < 10 ? sm : 9)];
            nc.c1 = nrow[(size_t)10 * Nall + b0 + suc];
        }
        PHX(cur + 0);
        {
            const bool live = fi < nb;
            const bool there = frag_there(x, live);
            if (PROF && tid == 0) { PROFL[cur + 6] += 1; PROFL[cur + 7] += !there; }
            if (__builtin_expect(!there, 0))
                wait_for([&] { return frag_there(x, live); },
                         [&] {
#pragma unroll
                             for (int r = 0; r < 8; ++r) x[r] = __builtin_amdgcn_raw_buffer_load_b128(xrs, voff_frag + r * 1024, soff_x, 16 /* sc1 */);
                         },
                         status, dead, 0x600u | (LA ? 0u : 8u) | (unsigned)kind, t);
        }
        PHX(cur + 3);
        if (kind == 2 && i == nact - 1) {
            // RAW ring hygiene (see the header): drain, then re-arm this wave's quarter of the workgroup's logit block of every slot in entry (t + 3) % 4
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane < 16) {
                const u32x4 q = {SENT, SENT, SENT, SENT};
                const int so = cbase + 16 * DLAYERB + ((t + DAHEAD_HH) & (DRING - 1)) * XTB;
#pragma unroll 1
                for (int i2 = 0; i2 < nact; ++i2) {
                    if (loc_h) __builtin_amdgcn_raw_buffer_store_b128(q, xrs, J * 1024 + w * 256 + lane * 16, so + i2 * (MAXCL * DSLOTB), 0);
                    else __builtin_amdgcn_raw_buffer_store_b128(q, xrs, J * 1024 + w * 256 + lane * 16, so + i2 * (MAXCL * DSLOTB), 16 /* sc1 */);
                }
            }
        }
        if (sampler && (kind == 3 || kind == 4)) {
            // ring hygiene (see the header): drain, then re-arm the x_t words of the slot this workgroup samples in entry (t + 3) % 4
            // (the gh words carry a step tag instead of relying on a sentinel: nothing to re-arm)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (MOL) {
                if (lane == 48) {                       // the 4 x_t words of segments 4 w ..
                    const u32x4 q = {SENT, SENT, SENT, SENT};
                    __builtin_amdgcn_raw_buffer_store_b128(q, xrs, 16 * w, sbase + 7 * DLAYERB + ((t + DAHEAD_HH) & (DRING - 1)) * XTB, 16 /* sc1 */);
                }
            } else if (lane == 48) {                    // RAW: this wave's ONE word (segment 4 quad + w)
                __builtin_amdgcn_raw_buffer_store_b32(SENT, xrs, (4 * quad + w) * 4, sbase + 7 * DLAYERB + ((t + DAHEAD_HH) & (DRING - 1)) * XTB, 16 /* sc1 */);
            }
        }
        float b[32];
        frag_to_b(x, b);
        {   // the next stage's polled layer, one stage ahead (not across a step boundary: nothing of the next step is published yet)
            xahead = false;
            if (kind == 1 || kind == 2) {
                int so = -1;
                if (kind == 1 && order1) {                                                                                  // MOL: gh(0 .. my_slot) | sampling | gh(my_slot + 1 ..)
                    if (sampler && i == my_slot) so = cbase + my_slot * (MAXCL * DSLOTB) + 3 * DLAYERB + ring * XTB;
                    else if (i + 1 < nact) so = sbase + MAXCL * DSLOTB + L_H * DLAYERB + ring * XTB;
                } else if (kind == 1 && deferred) so = -1;                                                                  // (the next stage belongs to the next step)
                else if (i < (kind == 1 ? last_here : nact - 1)) so = sbase + MAXCL * DSLOTB + (kind == 1 ? L_H : 3) * DLAYERB + ring * XTB;
                else if (kind == 1 && !LA && !MOL) so = cbase + 3 * DLAYERB + ring * XTB;                                   // RAW: the logits stage of slot 0
                else if (sampler) so = cbase + my_slot * (MAXCL * DSLOTB) + (MOL ? 3 : 16) * DLAYERB + ring * XTB;      // the sampling stage
                if (so >= 0) {
                    xahead = true;
#pragma unroll
                    for (int r = 0; r < 8; ++r) x[r] = ((DUO_ABLATE & 4) && r > 0) ? x[0] : __builtin_amdgcn_raw_buffer_load_b128(xrs, voff_frag + r * 1024, so, 16 /* sc1 */);
                }
            }
        }
        PHX(cur + 4);
        float *PW = DPARTOF(pp);
        if constexpr (kind == 1) {
            f32x4 o0, o1, o2;
            if constexpr ((DUO_ABLATE & 2) != 0) { o0 = mfma1(A_hh[0], b); o1 = o0; o2 = o0; }
            else if (gh_yield) {
                f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0;
#pragma unroll
                for (int k = 0; k < 32; ++k) {
                    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A_hh[0][k], b[k], c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A_hh[1][k], b[k], c1, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(A_hh[2][k], b[k], c2, 0, 0, 0);
                    // (gh_gran / gh_chunk are RUN-TIME values on purpose: the test below compiles to a compare + a taken scalar branch behind every k -- behind every
                    // three MFMAs -- and it is THAT broken-up stream, more than the three yields, that lets the ih wave's instructions through; see gh_yield)
                    if ((k & gh_gran) == gh_gran && k < 31) { for (int q = 0; q < gh_chunk; ++q) __builtin_amdgcn_s_sleep(1); }
                }
                o0 = c0; o1 = c1; o2 = c2;
            } else mfma3s(A_hh[0], A_hh[1], A_hh[2], b, o0, o1, o2);
            put_partial<3>(PW, w, 0, lane, o0);
            put_partial<3>(PW, w, 1, lane, o1);
            put_partial<3>(PW, w, 2, lane, o2);
            pend = BK_GH;
        } else if constexpr (kind == 2) {
            put_partial<3>(PW, w, 0, lane, mfma1(A_f3, b));
            pend = BK_LG;
        } else if constexpr (kind == 4) {
            // fatchord_version.py:231-237: softmax -> Categorical (renormalise) -> argmax(p / q) -- the loop kernel's code, ONE segment per wave (4 quad + w);
            // per segment the operation order is the reference's (class indices compared bit for bit)
            const int b0 = GEO[2 * i];
            {
                float *lp = LGT + fi * LDC + kbase_lane;
#pragma unroll
                for (int r = 0; r < 8; ++r) *reinterpret_cast<float4 *>(lp + 16 * r) = make_float4(b[4 * r], b[4 * r + 1], b[4 * r + 2], b[4 * r + 3]);
            }
            lds_barrier();
            // The Exp(1) variates (2 KB per segment and step, streamed from HBM) are requested before the softmax passes that do not need them yet.
            {
                constexpr int NS = 1;
                const int h0 = 4 * quad + w;            // this wave's segment
                float qn[NS][8], lg[NS][8], mx[NS], sum[NS], sum2[NS], best[NS];
                int bidx[NS];
                const size_t tn = (size_t)(t - noise_t0);
#pragma unroll
                for (int s4 = 0; s4 < NS; ++s4) {
                    const int sjc = (h0 + s4 < nb) ? h0 + s4 : nb - 1;
#pragma unroll
                    for (int e = 0; e < 8; ++e) qn[s4][e] = noise_raw[(tn * Nall + b0 + sjc) * C + lane + 64 * e];
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
                            for (int e = 0; e < 8; ++e) dbgl[((size_t)t * Nall + b0 + h0 + s4) * C + lane + 64 * e] = lg[s4][e];
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
                            outp[(size_t)(b0 + sj) * Tall + t] = xv;
                            if (forcex) xv = forcex[(size_t)(b0 + sj) * Tall + t];
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(xv), xrs, sj * 4, sbase + 7 * DLAYERB + ring * XTB, 16 /* sc1 */);
                        }
                    }
                }
            }
            lds_barrier();                              // LGT is read by every wave before the next step's gather overwrites it
#pragma unroll
            for (int r = 0; r < 8; ++r) x[r] = u32x4{0u, 0u, 0u, 0u};      // (the look-ahead registers are dead through the sampling: say so)
            pend = BK_NONE;
        } else {
            // fc3 (30 x 512: two 16-row tiles in A-fragment order).  Tile 0 sits in LDS (copied once per launch): requesting it from L2
            // here costs ~0.8 us per step on a slot's chain (25.35 vs 24.59 us per step at depth 4 with the request moved ahead of the
            // wait for y2, which in turn cost 5 VGPR spills: profiles/r04c_probe_*.json); tile 1 comes from L2 under tile 0's MFMAs,
            // into the registers of the (now idle) look-ahead fragments
            {
                const u32x4 *fp = reinterpret_cast<const u32x4 *>(fc3f + XT + frag_off(w, 0, lane));
#pragma unroll
                for (int r = 0; r < 8; ++r) x[r] = fp[64 * r];
            }
            put_partial<3>(PW, w, 0, lane, mfma1_lds(F3 + frag_off(w, 0, lane), b));
            float4 av1[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) av1[r] = make_float4(__uint_as_float(x[r].x), __uint_as_float(x[r].y), __uint_as_float(x[r].z), __uint_as_float(x[r].w));
            put_partial<3>(PW, w, 1, lane, mfma1_frag(av1, b));
            pend = BK_SAMPLE;
        }
        PHX(cur + 5);
        cy = nc;
        pp ^= 1;
    };

    using K1 = std::integral_constant<int, 1>;
    using K2 = std::integral_constant<int, 2>;
    using K3 = std::integral_constant<int, 3>;
    using K4 = std::integral_constant<int, 4>;
    using BGH = std::integral_constant<int, BK_GH>;
    using BLG = std::integral_constant<int, BK_LG>;
    using BANY = std::integral_constant<int, BK_ANY>;
    cond_step(T0);                                      // (the two steps a launch starts with; every later one is formed two steps ahead)
    if (T0 + 1 < T1) cond_step(T0 + 1);
    // Order of a step's stages.  The gh stages are off every chain (their output is read a step later); the stages behind them -- the MOL
    // sampling of this workgroup's slot, RAW's logits -- are ON their slot's chain, and with the plain order gh(0) .. gh(n-1) | sample the gh
    // stage of the LAST slot (whose h2 arrives last) sits right in front of them: y2 of slot 0 is there ~one stage earlier than this
    // workgroup is free.  So the last slot's gh stage is deferred to the top of the NEXT step (its product, gh(t+1) of that slot, is needed
    // late in that step): gh(n-1)@t-1 | gh(0) .. gh(n-2) | sample ...  -- the same two call sites (every call site of `stage` is a full
    // inlined copy), one more loop iteration at the end of a launch for the last deferred stage.
    const bool shift = DUO_GH_SHIFT && nact >= 2 && !order1;                 // (RAW)
    last_here = shift ? nact - 2 : nact - 1;
    if constexpr (order1) {
        // Round 6 (MOL).  gh(0) .. gh(n - 1) in slot order, nothing deferred across the step boundary, and a sampling workgroup runs its
        // sampling stage right behind gh(my_slot): y2 of slot J is published by the J-th fc stage of rnn2's ih workgroups, about when
        // gh(0 .. J) are through -- round 4's order (the last slot's gh stage deferred to the top of the next step, the sampling stage
        // behind gh(n - 2)) made the sampler of slot 0 sit through the gh stages of slots whose h2 arrives after its y2.  Measured
        // (profiles/r06l_*): 16.4 / 19.4 / 22.7 us per step at depth 2 / 3 / 4 against 17.0 / 19.8 / 23.5.  (Also measured, no effect: the
        // gh stages held back until the LAST slot's h is there, i.e. until the ih workgroup of the CU has left its MFMA-dense gates
        // phase -- profiles/r06k_*.)
        for (; t < T1; ++t) {
            if constexpr (LA) {
                if (t + 2 < T1) cond_step(t + 2);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // ... and out before anything of this step is published
            }
#pragma unroll 1
            for (int i = 0; i < nact; ++i) {
                stage(K1{}, BANY{}, i);
                if constexpr (!LA) {
                    if (sampler && i == my_slot) stage(K3{}, BGH{}, my_slot);
                }
            }
        }
    } else
    for (; t <= T1; ++t) {
        const bool late = shift && t > T0;              // this iteration starts with the deferred stage of step t - 1
        if (t == T1 && !late) break;
        if constexpr (LA) {
            if (t + 2 < T1) cond_step(t + 2);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // ... and out before anything of this step is published
        }
        deferred = late;
        if (late) --t;
        stage(K1{}, BANY{}, late ? nact - 1 : 0);
        if (late) ++t;
        deferred = false;
        if (t == T1) break;
#pragma unroll 1
        for (int i = late ? 0 : 1; i <= last_here; ++i) stage(K1{}, BGH{}, i);
        if constexpr (!LA && MOL) {
            if (sampler) stage(K3{}, BGH{}, my_slot);
        }
        if constexpr (!LA && !MOL) {
            stage(K2{}, BGH{}, 0);
#pragma unroll 1
            for (int i = 1; i < nact; ++i) stage(K2{}, BLG{}, i);
            if (sampler) stage(K4{}, BLG{}, my_slot);
        }
    }
    cur = 0;
    if (pend == BK_GH) back_gh(cy);
    else if (pend == BK_SAMPLE) { if constexpr (!LA && MOL) back_sample(cy); }
    else if (pend == BK_LG) { if constexpr (!LA && !MOL) back_lg(cy); }
    cond_leave();
    if (PROF && tid == 0 && profp) {
        for (int k = 0; k < 16; ++k) profp[(size_t)(blockIdx.x & 255) * 32 + 16 + k] += PROFL[k];
        if (trp) trp[0] = (u64)ntr;
    }
}
#undef DPARTOF
#undef PHX
#undef TRACE_DECL

// Grid = clusters x 128 workgroups of 256 threads (two per CU), cooperative launch.  Whole XCDs per cluster (speed only: nothing
// depends on the placement; what the placement is, is looked at below).
template <int MODE, bool LP, bool EI, bool PROF>
__global__ __launch_bounds__(NT, 2) void wrnn_duo_kernel(const LoopArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int cl, wg;
    const int ncl = gridDim.x / DNWGC;
    check_kind(a);
    {
        const int b = blockIdx.x, nblk = gridDim.x;
        if (nblk % 8 == 0 && ncl >= 1 && 8 % ncl == 0) {
            const int xpc = 8 / ncl, per_xcd = nblk / 8;
            const int xcd = b % 8;
            cl = xcd / xpc;
            wg = (xcd % xpc) * per_xcd + b / 8;
        } else {
            cl = b / DNWGC;
            wg = b % DNWGC;
        }
    }
    if (cl >= a.NG) return;                             // a cluster without a group of this round (the grid is always 4 clusters: see launch_duo)
    // role / unit block of workgroup wg.  Speed only: block b is observed to run on XCD b % 8 and the blocks of an XCD to be dealt
    // round-robin over its 32 CUs, i.e. (with 64 blocks per XCD) local blocks q and q + 32 share a CU; with 4 clusters a cluster is two
    // XCDs = the two halves of its 128 blocks.  First half: rnn1, second half: rnn2; within a half q < 32: the ih workgroup of unit block
    // q, q >= 32: the hh workgroup of unit block q - 32.  A CU carries the ih (128 MFMAs per group-step and wave) and the hh workgroup (96)
    // of the same units; h, gh and (rnn2) y2 never leave the XCD.  (Measured and dropped, profiles/r04b_probe.json: all ih workgroups
    // of a cluster on one XCD -- x1, x2, y1 through one L2 -- is 1.0 us per step SLOWER at depth 1 and 3-8 % slower at depth 4-8: a plain
    // store is not visible in its L2 sooner than a write-through store is visible across the fabric, and two ih workgroups per CU
    // share the matrix pipe badly.)
    const int layer = wg / (DNWGC / 2), q = wg % (DNWGC / 2);
    const int hh = q >> 5, J = q & 31;
    // ---- placement handshake: every workgroup records its XCC id; a layer whose producers and consumers all sit on one XCD is
    //      exchanged through that XCD's L2 with plain stores.  Every workgroup of a cluster reads the same 128 words -> the same verdict.
    bool loc_a = false, loc_b = false;
    {
        int *TAB = reinterpret_cast<int *>(smem) + 2 * LMAXG;         // scratch (the roles clear their LDS afterwards)
        const int tid = threadIdx.x;
        unsigned *tab = a.xcc_tab + cl * DNWGC;
        if (tid == 0) {
            const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xFu;          // HW_REG_XCC_ID
            __hip_atomic_store(tab + wg, xcc + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!PROF && a.prof && blockIdx.x < 2048 && (a.tuning & 64)) {                   // placement read-out (test / profiling hook)
                const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
                a.prof[blockIdx.x] = ((u64)xcc << 32) | hw | ((u64)(unsigned)((layer | (hh << 1)) | (J << 2) | (cl << 8)) << 40);
            }
        }
        unsigned v = 1u;
        if (tid < DNWGC) {
            unsigned spins = 0;
            v = ld_agent32(tab + tid);
            while (v == 0u && ++spins < 200000u) {
                __builtin_amdgcn_s_sleep(2);
                v = ld_agent32(tab + tid);
            }
            TAB[tid] = (int)v;
        }
        __syncthreads();
        const int ok_a = (tid < DNWGC / 2) ? (v != 0u && (int)v == TAB[0]) : 1;
        const int ok_b = (tid >= DNWGC / 2 && tid < DNWGC) ? (v != 0u && (int)v == TAB[DNWGC / 2]) : 1;
        loc_a = __syncthreads_and(ok_a) != 0;
        loc_b = __syncthreads_and(ok_b) != 0;
        if (a.tuning & 256) { loc_a = false; loc_b = false; }        // A/B: everything written through, as round 3
        __syncthreads();
    }
    // loc_a / loc_b: every workgroup of the first / second half of the cluster was seen on one XCC -> h, gh (ih <-> hh workgroups of a
    // layer) and y2 (rnn2's ih workgroups -> the sampling workgroups) stay in that XCD's L2
#ifndef DUO_ROLES
#define DUO_ROLES 15                          // (register-allocation diagnosis: compile with a subset of the four roles, -DDUO_ROLES=<mask>)
#endif
    if (layer == 0) {
        if (hh == 0) { if constexpr ((DUO_ROLES & 1) != 0) duo_ih<MODE, true, LP, EI, PROF>(a, smem, cl, J, ncl, loc_a, false); }
        else { if constexpr ((DUO_ROLES & 2) != 0) duo_hh<MODE, true, LP, PROF>(a, smem, cl, J, ncl, loc_a); }
    } else {
        if (hh == 0) { if constexpr ((DUO_ROLES & 4) != 0) duo_ih<MODE, false, LP, EI, PROF>(a, smem, cl, J, ncl, loc_b, loc_b); }
        else { if constexpr ((DUO_ROLES & 8) != 0) duo_hh<MODE, false, LP, PROF>(a, smem, cl, J, ncl, loc_b); }
    }
}

size_t duo_lds_bytes(int G) { return (size_t)duo_lds(G).total * sizeof(float); }
size_t duo_xbuf_bytes(int G) { return (size_t)G * MAXCL * DNX * DLAYER_ENTRIES * XT * sizeof(float); }     // the slots a launch with depth G touches: a prefix
size_t duo_xbuf_bytes_max() { return DXBUF_FLOATS * sizeof(float); }
int duo_max_depth() { return LMAXG; }

// clusters this device can host at two workgroups per CU (64 CUs per cluster, as wrnn_loop_kernel)
int duo_clusters(int n_cus)
{
    int ncl = n_cus / LNWGC;
    if (ncl > MAXCL) ncl = MAXCL;
    while (ncl > 1 && (8 % ncl) != 0) --ncl;
    return ncl;
}

// How the ih workgroups get a stage's operand (round 6, profiles/r06l_*: us per step at 1 / 2 / 3 / 4 / 8 slots in flight -- register loads
// behind the back half 12.3 / 16.4 / 19.4 / 22.7 / 41.3, LDS prefetch one stage ahead 12.4 / 16.6 / 19.7 / 23.6 / 38.3): the prefetch pays
// once a step is the workgroups' busy time, not the latency of a slot's chain (its 8 LDS-DMA issues and the read-back cost what the exposed
// L2 latency costs at 4 slots; at 8 they are 7 % cheaper).
hipError_t launch_duo(const LoopArgs &args, int ncl, int mode, hipStream_t stream)
{
    if (ncl < 1 || args.G < 1 || args.G > LMAXG || (mode == 1 && !args.fc3f) || !args.u1 || !args.xcc_tab) return hipErrorInvalidValue;
    const size_t lds = duo_lds_bytes(args.G);
    // wrnn_options.tuning (A/B switches): bit 0 = operands by LATE register loads (behind the back half), bit 1 = by LDS prefetch one stage ahead (on request only
    // since r06v: early register loads beat it at 8 slots); bit 8 = every layer
    // written through (no XCD-local plain stores); bit 6 = placement read-out through the phase-clock buffer; bit 14 (profiling builds):
    // the stage time line ("TRACE")
    // ... bit 9 = register loads issued in front of the pending back half (EI; default from DUO_EI_DEPTH slots in flight on: r06v -- 22.7 vs 22.9 us per
    // step at 4 slots, 37.4 at 8 against 38.8 with the LDS prefetch and 41.9 with late register loads; 0.1-0.2 us slower below 4)
    const bool lp = (args.tuning & 2) != 0;
    const bool ei = !lp && ((args.tuning & 512) ? true : ((args.tuning & 1) ? false : args.G >= DUO_EI_DEPTH));
    const bool prof = mode == 1 && args.prof && !(args.tuning & 64);              // phase clocks: MOL builds only
    const void *fn = mode == 1 ? (lp ? (prof ? (const void *)wrnn_duo_kernel<1, true, false, true> : (const void *)wrnn_duo_kernel<1, true, false, false>)
                                : ei ? (prof ? (const void *)wrnn_duo_kernel<1, false, true, true> : (const void *)wrnn_duo_kernel<1, false, true, false>)
                                     : (prof ? (const void *)wrnn_duo_kernel<1, false, false, true> : (const void *)wrnn_duo_kernel<1, false, false, false>))
                               : (lp ? (const void *)wrnn_duo_kernel<0, true, false, false> : ei ? (const void *)wrnn_duo_kernel<0, false, true, false> : (const void *)wrnn_duo_kernel<0, false, false, false>);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    LoopArgs a = args;
    void *params[] = {(void *)&a};
    return hipLaunchCooperativeKernel(fn, dim3(ncl * DNWGC), dim3(NT), params, (unsigned)lds, stream);
}

}  // namespace wrnn
