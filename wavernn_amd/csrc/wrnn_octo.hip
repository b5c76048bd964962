// wrnn_octo.hip -- a WAVE-SPECIALISED form of the persistent dense WaveRNN loop kernel (MOL) for MI355X (gfx950 / CDNA4).  Round 6: built, bit-identical to
// wrnn_duo_kernel on every split, and SLOWER than it (29.4 vs 22.9 us per step at 256 segments, 16.8 vs 12.5 with one slot in flight: DESIGN.md 9.1 d,
// profiles/r06u_octo_clocks.log) -- kept on request only (WRNN_ALGO_OCTO), as the measured answer to "what if the MFMAs had a wave of their own".
//
// Same path (reference models/fatchord_version.py:201-241, utils/distribution.py:87-123), the same exchange layers, ring discipline and
// fragment-order layout as wrnn_duo.hip -- what changes is WHO does what on a CU.  Measured on this chip (scripts/micro/mfma_valu_coexec.hip,
// octo_feasibility.hip; profiles/r06o_octo_feasibility.log): a wave's own VALU / LDS / memory instructions never overlap its own MFMAs, and a wave that
// issues MFMAs back to back is not slowed by the other wave of its SIMD while that wave's instructions take ~2.3 x as long.  wrnn_duo_kernel's
// two workgroups per CU therefore take turns more than they overlap: each of its waves carries ~250 (gates) / ~165 (fc) / ~150 (gh) non-MFMA
// instructions per stage around its MFMA block and the matrix pipe is busy 0.53-0.59 of the time.  Here ONE 512-thread workgroup per CU holds
//
//     waves 0-3, the MATRIX waves : W_ih AND W_hh of the CU's 16 units (192 weight registers; the fc1 / fc2 tile as an A operand in LDS).
//                                   Per block: the operand fragments arrive in the wave's 8 KB of LDS by LDS-DMA (requested one block ahead),
//                                   8 ds_read_b128, one sentinel compare, the next request, 96 (gates, gh) or 32 (fc) MFMAs, the partial
//                                   tiles into LDS, one LDS counter.  No pointwise math, no store to memory.
//     waves 4-7, the SERVICE waves: everything else -- the 4-wave partial sums, the GRU cell, relu, every publish and re-arm, rnn1: the I-layer
//                                   conditioning cI(t + 2) (cond_tile, as wrnn_duo_kernel's rnn1 hh role); rnn2's CU s < slots in flight: fc3 (both
//                                   tiles in ITS registers) + the MoL sampling of slot s.  They store (write-through: ~1 us to the acknowledgement)
//                                   and vmcnt retires in order, so in the common path they LOAD nothing but the x_{t-1} words: the residual input
//                                   of the owned units comes from the matrix waves' registers with the partial tiles, the aux-table values live
//                                   in LDS and are re-read when the table row changes (once per hop).
//
// The two kinds meet through LDS only: a ring of ONPX partial-tile buffers and one counter PER WAVE (a plain LDS store by its only writer; LDS
// operations of a wave execute in order, so the counter follows the data without a fence) -- no s_barrier inside the loop, the matrix waves run
// up to ONPX blocks ahead of the service waves.  gh never leaves the CU (wrnn_duo_kernel: a 16-byte word per (unit, segment) and step through L2,
// 256 of the 672 KB a group-step stored): the service thread that reduced gh(t + 1) keeps it in its own LDS word until its gate job of step t + 1.
//
// Why it loses (phase clocks, profiles/r06u_octo_clocks.log): (1) the chain of a slot is longer -- every stage hands over twice inside the CU (DMA ->
// LDS -> registers in front of the MFMAs, partial tiles -> counter -> service wave behind them), the fc block with its A operand in LDS takes 1.7 k
// clocks for 32 MFMAs, the sampling (64 MFMAs on a service wave that shares its SIMD with a matrix wave) 6.5 k: 16.8 us per step with one slot
// in flight against 12.5; (2) one job queue per kind: a gate job that waits for x_{t-1} holds up the gh and fc jobs of the other slots behind it
// (wrnn_duo_kernel's hh workgroup keeps working meanwhile), so at four slots the matrix waves still wait for operands 31 % of the time; (3) the
// matrix waves' blocks run at 38-41 clocks per MFMA beside a busy service wave, not the 34 of the micro-benchmark: 12 k clocks of work per
// slot-step where wrnn_duo_kernel's ih workgroup has 13.3 k -- even the throughput bound would be a gain of 10 %, not the 35 % the
// micro-benchmark's cut promised.
//
// Cluster = 64 CUs (two XCDs under the placement the kernel looks at): rnn1 on one, rnn2 on the other; CU (layer, J) owns units [16 J, 16 J + 16).
// Step t of a CU, n slots in flight -- matrix waves: ih(0) .. ih(n - 1) | hh(0) fc(0) hh(1) fc(1) ..; service waves: the back halves in the
// same order (gates -> h, residual sum | gh -> LDS | fc -> relu -> y) plus, once per step, cI(t + 2) / the sampling of the own slot / the ring hygiene.
//
// Ring discipline (wrnn_duo.hip's, re-derived for this cut):
//   * sentinel layers a CU's service waves publish (h, residual sum, y): FOUR entries by step; a wave re-arms its own words of entry (t + 2) % 4
//     (data of step t - 2) at the END of its step t and drains at the top of step t + 1 before it publishes anything of that step.  Every
//     reader is past step t - 2 by then: the wave has run a gate job of step t, whose x_{t-1} (rnn1) / x1(t) (rnn2) needed fc2(t - 1) of every
//     rnn2 CU, hence fc1(t - 1) of every rnn1 CU, and a CU's matrix waves run their blocks in step order; a reader that requests the entry for
//     step t + 2 has validated something this wave published in step t + 1, behind the drain (also for the look-ahead requests: the block
//     in front of the request was validated first).
//   * x_t (the sampler's 16 words): re-armed THREE ahead behind the poll of y2(t) and a drain, as wrnn_duo.hip.
//   * cI (layer 4): no sentinel inside a launch, never re-armed: formed two steps ahead at the top of the service waves' step and drained
//     before anything of that step is published; a matrix wave requests cI(s) after it has validated an operand that was published in step
//     s - 1 (the last block of step s - 1: x2 of that step), i.e. behind the drain of the step that formed cI(s + 1), let alone cI(s).
//     The first two steps of a launch are polled (the previous launch leaves the sentinel in the entries of T1 and T1 + 1).
// A launch ends with every entry in the state the next step expects; the state between launches is wrnn_duo_kernel's (gh(T1), h, x_{T1-1},
// the segment table) in the same layout.  (The first form of this kernel summed the waves' block counts into one counter per kind: four waves that
// drift apart by up to ONPX blocks make such a sum meaningless -- it passed every test with XCD-local layers and failed every run without.)
#include <type_traits>

#include "wrnn_ring.h"

namespace wrnn {

constexpr int ONT = 512;                     // threads per workgroup: 4 matrix + 4 service waves
constexpr int ONPX = 3;                      // partial-tile buffers between the two kinds (12 KB each)
constexpr int OLOGS = 36;                    // LOG row stride (as DLOGS of wrnn_duo.hip)
constexpr int OPXB = NW * 3 * 256 + 256;     // floats of one partial-tile buffer: [wave][tile 0..2][lane][4], then the CU's own block of the gate stage's operand layer
constexpr int OXO = NW * 3 * 256;            // ... at this offset: fragment [lane][4] = units 4 (lane >> 4) .. + 3 of segment lane & 15 (what x[J & 7] of matrix wave J >> 3 holds)

struct OctoLds {
    int off_op, off_px, off_fcw, off_h, off_gh, off_cv, off_seg, off_xs, off_py, off_log, off_misc, off_cnt, off_prof, total;
};
__host__ __device__ constexpr OctoLds octo_lds(int G)
{
    OctoLds l{};
    int o = 0;
    l.off_op = o;   o += NW * 2048;          // matrix wave w: its 8 fragments of the block's operand (LDS-DMA target)
    l.off_px = o;   o += ONPX * OPXB;
    l.off_fcw = o;  o += XT;                 // fc1 / fc2 rows of the CU's 16 units in A-fragment order
    l.off_h = o;    o += G * 256;            // h of the owned (unit, segment), per slot (thread-private words of the service waves)
    l.off_gh = o;   o += G * 768;            // gh(t + 1) = W_hh h(t) + b_hh of the owned (unit, segment): [slot][r z n][256] (thread-private)
    l.off_cv = o;   o += G * 5 * 256;        // aux-table values of the owned (unit, segment), per slot: [c2f r z n | c3f / c4f | the table row they belong to] (thread-private)
    l.off_seg = o;  o += G * 64;             // ints: [slot][16 positions | 16 limits | 16 table-row bases | 16 mel offsets]
    l.off_xs = o;   o += G * 16;             // rnn1: x_{t0-1} of a continuing launch
    l.off_py = o;   o += NW * 2 * 256;       // sampler: fc3 partial tiles of the service waves
    l.off_log = o;  o += SEG * OLOGS;
    l.off_misc = o; o += 2 * LMAXG + LNWGC;  // [2 i], [2 i + 1]: first segment / count of slot i; then the placement table (ints)
    o = (o + 3) & ~3;
    l.off_cnt = o;  o += 12;                 // [0..3] blocks produced by matrix wave w, [4..7] blocks consumed by service wave w, [8..11] rendezvous count of service wave w
    o = (o + 1) & ~1;
    l.off_prof = o; o += 2 * 24;             // u64: [8] matrix wave 0, [16] service wave 0 (profiling builds)
    l.total = (o + 3) & ~3;
    return l;
}

// phase clocks (profiling builds: wrnn_options.phase_clocks): lane 0 of wave 0 of each kind adds the shader clocks since the last mark to slot k
#define OPH(k)                                                   \
    do {                                                         \
        if (PROF && w == 0 && lane == 0) {                       \
            const u64 now_ = __builtin_amdgcn_s_memtime();       \
            pacc[k] += now_ - plast;                             \
            plast = now_;                                        \
        }                                                        \
    } while (0)
// (the matrix waves have no register to spare: their last mark lives in LDS too, slot 7)
#define OPHM(k)                                                  \
    do {                                                         \
        if (PROF && w == 0 && lane == 0) {                       \
            const u64 now_ = __builtin_amdgcn_s_memtime();       \
            pacc[k] += now_ - pacc[7];                           \
            pacc[7] = now_;                                      \
        }                                                        \
    } while (0)
typedef __attribute__((address_space(3))) void lds_void_o;
__device__ __forceinline__ void octo_dma8(__amdgpu_buffer_rsrc_t rs, float *dst, int voff, int soff)
{
    lds_void_o *d0 = (lds_void_o *)dst, *d1 = (lds_void_o *)(dst + 1024);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, d0, 16, voff, soff, 0, 16 /* sc1 */);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, d0, 16, voff, soff, 1024, 16);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, d0, 16, voff, soff, 2048, 16);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, d0, 16, voff, soff, 3072, 16);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, d1, 16, voff + 4096, soff, 0, 16);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, d1, 16, voff + 4096, soff, 1024, 16);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, d1, 16, voff + 4096, soff, 2048, 16);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, d1, 16, voff + 4096, soff, 3072, 16);
}

// LDS counters, ONE PER WAVE (a wave is the only writer of its word: a plain store of its new count; the four waves of a kind drift apart by up to
// ONPX blocks, so a sum over the waves says nothing about any one of them -- the first form of this kernel summed and read partial tiles that one
// matrix wave had not written yet).  A reader takes the four words of a kind with one ds_read_b128 and waits for the slowest.  The LDS executes a wave's
// operations in order: data first, then the counter on the producer side; counter first, then the data on the consumer side (compiler barriers only).
__device__ __forceinline__ u32x4 ocnt_ld4(const unsigned *p) { return *reinterpret_cast<const volatile u32x4 *>(p); }
__device__ __forceinline__ bool ocnt_reached(u32x4 c, unsigned target)
{
    return (int)(c.x - target) >= 0 && (int)(c.y - target) >= 0 && (int)(c.z - target) >= 0 && (int)(c.w - target) >= 0;
}
__device__ __forceinline__ void ocnt_set(unsigned *p, unsigned v, int lane)
{
    asm volatile("" ::: "memory");
    if (lane == 0) *reinterpret_cast<volatile unsigned *>(p) = v;
    asm volatile("" ::: "memory");
}
// bounded like every wait of the loop kernels: the abort flag of the launch is looked at now and then; a wave that gives up is `dead` and skips
// every later wait (the launch ends, wrnn_status() reports)
__device__ __forceinline__ void ocnt_wait(const unsigned *p4, unsigned target, unsigned *status, bool &dead, unsigned code, int step)
{
    unsigned spins = 0;
    while (!dead && !ocnt_reached(ocnt_ld4(p4), target)) {
        if ((++spins & 1023u) == 0u) {
            if (ld_agent32(status) != 0u) { dead = true; break; }
            if (spins > 4u * SPIN_LIMIT) { report_failure(status, code, blockIdx.x, step, threadIdx.x); dead = true; break; }
        }
        __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
}

// ---------------------------------------------------------------------------------------------------------------------------------
// matrix waves (w = 0 .. 3: K chunk [128 w, 128 w + 128) of every tile)
// ---------------------------------------------------------------------------------------------------------------------------------
template <bool LA, bool PROF>
__device__ __forceinline__ void octo_matrix(const LoopArgs &a, float *smem, const int cl, const int J, const int ncl, const int w, const int lane)
{
    const int G = a.G;
    const OctoLds L = octo_lds(G);
    float *const OP = smem + L.off_op + w * 2048;
    float *const PX = smem + L.off_px;
    const float *const FCW = smem + L.off_fcw + frag_off(w, 0, lane);
    unsigned *const CNT = reinterpret_cast<unsigned *>(smem + L.off_cnt);
    const int fi = lane & 15, kq = lane >> 4;
    const int kbase_lane = KCH * w + 4 * kq;
    const int T0 = a.t0, T1 = a.t1;
    const int NR = a.Btot, NGR = a.NG;
    unsigned *const status = a.status;
    const int tun = a.tuning;
    constexpr int L_IH = LA ? 4 : 5, L_HH = LA ? 0 : 1, L_FC = LA ? 6 : 2;       // the layers the three block kinds multiply

    float A_ih[3][AF], A_hh[3][AF];
#pragma unroll
    for (int g = 0; g < 3; ++g) load_afrag(A_ih[g], LA ? a.w_ih1 : a.w_ih2, LA ? H : H + AUX, g * H + LU * J + fi, true, kbase_lane);
#pragma unroll
    for (int g = 0; g < 3; ++g) load_afrag(A_hh[g], LA ? a.w_hh1 : a.w_hh2, H, g * H + LU * J + fi, true, kbase_lane);

    int nact = 0;
    u64 nbpack = 0;
    for (int i = 0; i < G; ++i) {
        const int g = cl + ncl * i;
        if (g >= NGR) break;
        nact = i + 1;
        const int b0 = (int)(((long)g * NR) / NGR), nb = (int)(((long)(g + 1) * NR) / NGR) - b0;
        nbpack |= (u64)(unsigned)nb << (8 * i);
    }
    const __amdgpu_buffer_rsrc_t xrs = make_rsrc(a.xbuf, (unsigned)(DXBUF_FLOATS * 4));
    const int voff_frag = frag_off(w, 0, lane) * 4;
    const int cbase = cl * DSLOTB;
    bool dead = false;
    u64 *const pacc = reinterpret_cast<u64 *>(smem + L.off_prof);
    if (PROF && w == 0 && lane == 0) pacc[7] = __builtin_amdgcn_s_memtime();
    // PROF: 0 operand wait (DMA + poll), 1 LDS read + check + next request,
                                                        // 2 partial-buffer wait, 3 MFMA + partial tiles (gates / gh), 4 the same (fc), 5 blocks, 6 blocks whose operand was polled for
    unsigned kblk = 0;                                  // blocks produced so far (this wave)
    int pxi = 0;                                        // kblk % ONPX
    int t = T0;

    if (T0 < T1 && nact > 0) octo_dma8(xrs, OP, voff_frag, cbase + (T0 & (DRING - 1)) * XTB + L_IH * DLAYERB);

    // one block: kind 0 = W_ih . (cI | x1), 1 = W_hh . h, 2 = fc . (x2 | y1) of slot i at step t
    auto block = [&](auto KC, int i) {
        constexpr int kind = decltype(KC)::value;
        const int nb = (int)((nbpack >> (8 * i)) & 255u);
        const int ring = t & (DRING - 1);
        const int sbase = cbase + i * (MAXCL * DSLOTB);
        const int soff_x = sbase + ring * XTB + (kind == 0 ? L_IH : (kind == 1 ? L_HH : L_FC)) * DLAYERB;
        u32x4 x[8];
        u32x4 cons_seen;
        {   // the block's fragments, requested one block ago, are in this wave's LDS block; a word that is still the sentinel: request again, look again
            // (ONE loop that defines x from LDS in every turn: a register-reloading poll loop keeps two copies of x alive -- 32 registers this role does not have)
            const bool live = fi < nb;
            unsigned spins = 0;
            if (tun & 512) {                            // (A/B and race hunting: no look-ahead -- the operand is requested here)
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                octo_dma8(xrs, OP, voff_frag, soff_x);
            }
            for (;;) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                OPHM(0);
#pragma unroll
                for (int r = 0; r < 8; ++r) x[r] = *reinterpret_cast<const u32x4 *>(OP + r * 256 + lane * 4);
                cons_seen = ocnt_ld4(CNT + 4);              // (read with the fragments: one LDS round trip; looked at below)
                if (__builtin_expect(frag_there(x, live) || dead, 1)) break;
                if (PROF && spins == 0 && lane == 0 && w == 0) pacc[6] += 1;
                if ((++spins & 255u) == 0u) {
                    if (ld_agent32(status) != 0u) dead = true;
                    else if (spins > SPIN_LIMIT) { report_failure(status, 0x700u | (LA ? 0u : 8u) | (unsigned)kind, blockIdx.x, t, threadIdx.x); dead = true; }
                }
                __builtin_amdgcn_s_sleep(1);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                octo_dma8(xrs, OP, voff_frag, soff_x);
            }
        }
        {   // the NEXT block's operand -> the (now free) LDS block.  Order of a step: ih(0) .. ih(n - 1) | hh(0) fc(0) hh(1) fc(1) ..
            int nso;
            if (kind == 0) nso = (i + 1 < nact) ? soff_x + MAXCL * DSLOTB : cbase + ring * XTB + L_HH * DLAYERB;
            else if (kind == 1) nso = sbase + ring * XTB + L_FC * DLAYERB;
            else nso = (i + 1 < nact) ? sbase + MAXCL * DSLOTB + ring * XTB + L_HH * DLAYERB
                                      : (t + 1 < T1 ? cbase + ((t + 1) & (DRING - 1)) * XTB + L_IH * DLAYERB : -1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (nso >= 0) octo_dma8(xrs, OP, voff_frag, nso);
        }
        float b[32];
        frag_to_b(x, b);
        OPHM(1);
        if (__builtin_expect(!ocnt_reached(cons_seen, kblk + 1u - (unsigned)ONPX), 0))
            ocnt_wait(CNT + 4, kblk + 1u - (unsigned)ONPX, status, dead, 0x710u | (LA ? 0u : 8u) | (unsigned)kind, t);     // the partial buffer is free: every service wave has read block kblk - ONPX
        OPHM(2);
        float *P = PX + pxi * OPXB;
        if constexpr (kind == 0) {
            // the CU's own block of the operand layer (cI / x1 of the owned units = the GRU's residual input) -> the service waves, with the partial tiles
            if (w == (J >> 3)) {
                u32x4 own;
                switch (J & 7) {
                case 0: own = x[0]; break;
                case 1: own = x[1]; break;
                case 2: own = x[2]; break;
                case 3: own = x[3]; break;
                case 4: own = x[4]; break;
                case 5: own = x[5]; break;
                case 6: own = x[6]; break;
                default: own = x[7]; break;
                }
                *reinterpret_cast<u32x4 *>(P + OXO + lane * 4) = own;
            }
        }
        if constexpr (kind == 2) {
            put_partial<3>(P, w, 0, lane, mfma1_lds(FCW, b));
        } else {
            f32x4 o0, o1, o2;
            if constexpr (kind == 0) mfma3s(A_ih[0], A_ih[1], A_ih[2], b, o0, o1, o2);
            else mfma3s(A_hh[0], A_hh[1], A_hh[2], b, o0, o1, o2);
            put_partial<3>(P, w, 0, lane, o0);
            put_partial<3>(P, w, 1, lane, o1);
            put_partial<3>(P, w, 2, lane, o2);
        }
        ocnt_set(CNT + w, kblk + 1u, lane);
        if (PROF) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); if (lane == 0 && w == 0) pacc[5] += 1; }
        OPHM(kind == 2 ? 4 : 3);
        ++kblk;
        pxi = (pxi + 1 == ONPX) ? 0 : pxi + 1;
    };
    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, 1>;
    using K2 = std::integral_constant<int, 2>;
    for (; t < T1; ++t) {
#pragma unroll 1
        for (int i = 0; i < nact; ++i) block(K0{}, i);
#pragma unroll 1
        for (int i = 0; i < nact; ++i) {
            block(K1{}, i);
            block(K2{}, i);
        }
    }
    if (PROF && w == 0 && lane == 0 && a.prof)
        for (int k = 0; k < 7; ++k) a.prof[(size_t)(blockIdx.x & 255) * 32 + k] += pacc[k];
}

// ---------------------------------------------------------------------------------------------------------------------------------
// service waves (w = 0 .. 3: thread (unit pu = 4 w + (tid & 3), segment pj = (tid >> 2) & 15) of the CU's 16 units)
// ---------------------------------------------------------------------------------------------------------------------------------
template <bool LA, bool PROF>
__device__ __forceinline__ void octo_service(const LoopArgs &a, float *smem, const int cl, const int J, const int ncl, const int w, const int lane,
                                             const bool loc_h)
{
    const int G = a.G;
    const OctoLds L = octo_lds(G);
    float *const PX = smem + L.off_px, *const HS = smem + L.off_h, *const GH = smem + L.off_gh, *const XS = smem + L.off_xs;
    float *const PY = smem + L.off_py, *const LOG = smem + L.off_log;
    int *const SEGT = reinterpret_cast<int *>(smem + L.off_seg);
    int *const GEO = reinterpret_cast<int *>(smem + L.off_misc);
    unsigned *const CNT = reinterpret_cast<unsigned *>(smem + L.off_cnt);
    const int tid = w * 64 + lane;                      // 0 .. 255 among the service waves
    const int fi = lane & 15, kq = lane >> 4;
    const int kbase_lane = KCH * w + 4 * kq;
    const int pu = 4 * w + (tid & 3), pj = (tid >> 2) & 15;
    const int prow = LU * J + pu;
    const int T0 = a.t0, T1 = a.t1, C = a.C;
    const int NR = a.Btot, Nall = a.Nall, NGR = a.NG;
    unsigned *const status = a.status;
    float *const state = a.state;
    float *const outp = a.out, *const dbgl = a.dbg_logits;
    const float *const forcex = a.force_x, *const noise_pre = a.noise_pre;
    const int Tall = a.T, noise_t0 = a.noise_t0;
    const float *const mels_up = a.mels_up, *const aux_fr = a.aux_fr, *const mel_coef = a.mel_coef;
    const int mel_stage = a.mel_stage;
    const int resume = a.resume, hop = a.hop;
    const int zrow = a.Nall * a.tab_fps;
    const unsigned magic = a.hop_magic;
    const int mshift = a.hop_shift;
    const int tun = a.tuning;
    constexpr int L_H = LA ? 0 : 1, L_XR = LA ? 5 : 6, L_Y = LA ? 2 : 3, L_IN = LA ? 4 : 5;       // layers this CU publishes; the GRU input of the owned unit

    // constants of the pointwise role
    const float *bhh = LA ? a.b_hh1 : a.b_hh2;
    const float bh_r = bhh[prow], bh_z = bhh[H + prow], bh_n = bhh[2 * H + prow];
    float cb_r = 0.f, cb_z = 0.f, cb_n = 0.f, ux_r = 0.f, ux_z = 0.f, ux_n = 0.f, w0o = 0.f;
    CondTile ct;
    if constexpr (LA) {
        cb_r = a.b_ih1[prow]; cb_z = a.b_ih1[H + prow]; cb_n = a.b_ih1[2 * H + prow];
        ux_r = a.u1[prow]; ux_z = a.u1[H + prow]; ux_n = a.u1[2 * H + prow];
        w0o = a.I_w0[prow];
        cond_tile_init(ct, a.I_cT, a.I_b, J, lane);
    }

    // geometry (the whole workgroup filled GEO / SEGT / HS / GH / XS before the roles split: wrnn_octo_kernel)
    int nact = 0;
    u64 nbpack = 0;
    for (int i = 0; i < G; ++i) {
        const int g = cl + ncl * i;
        if (g >= NGR) break;
        nact = i + 1;
        const int b0 = (int)(((long)g * NR) / NGR), nb = (int)(((long)(g + 1) * NR) / NGR) - b0;
        nbpack |= (u64)(unsigned)nb << (8 * i);
    }
    auto slot_nb = [&](int i) -> int { return (int)((nbpack >> (8 * i)) & 255u); };
    const size_t state_wg = ((size_t)(cl * LNWGC + 2 * J + (LA ? 0 : 1)) * G) * LGRP;

    // rnn2's CU J < slots in flight samples slot J: fc3 (30 rows: two tiles) in this wave's registers
    const bool sampler = !LA && J < nact;
    float A3[2][AF];
    float b3a = 0.f, b3b = 0.f;
    if constexpr (!LA) {
        load_afrag(A3[0], a.fc3_w, H, fi, sampler, kbase_lane);
        load_afrag(A3[1], a.fc3_w, H, 16 + fi, sampler && 16 + fi < 30, kbase_lane);
        if (sampler) {
            b3a = a.fc3_b[pu];
            b3b = (16 + pu < 30) ? a.fc3_b[16 + pu] : 0.f;
        }
    }

    const __amdgpu_buffer_rsrc_t xrs = make_rsrc(a.xbuf, (unsigned)(DXBUF_FLOATS * 4));
    const __amdgpu_buffer_rsrc_t crs = make_rsrc(a.c2f, 0x7FFFF000u);
    const __amdgpu_buffer_rsrc_t frs = make_rsrc(LA ? a.c3f : a.c4f, 0x7FFFF000u);
    const int voff_frag = frag_off(w, 0, lane) * 4;
    const int voff_own = (((J * 64) + 16 * w + pj) * 4 + (tid & 3)) * 4;      // the layer word of (owned unit pu, segment pj) = its publish position
    const int cbase = cl * DSLOTB;

    bool dead = false;
    u64 *const pacc = reinterpret_cast<u64 *>(smem + L.off_prof) + 8;
    u64 plast = PROF ? __builtin_amdgcn_s_memtime() : 0;
    // PROF: 0 wait for a block of the matrix waves (gates), 1 gates: partial sums, 2 gates: wait for x_{t-1} / the input word, 3 gates: cell + publish,
    // 4 gh job (wait), 5 gh job (work), 6 fc job (wait), 7 fc job (work), 8 pre-loads + loop, 9 cI forming + drain, 10 sampling: wait for y2, 11 sampling: rest, 12 re-arm
    unsigned kblk = 0, ksync = 0;
    int pxi = 0;
    int t = T0;

    // The service waves store (write-through: ~1 us until the acknowledgement) and vmcnt retires in order: a load a wave issues behind its publishes
    // waits for those acknowledgements.  So in the common path they load NOTHING but the x_{t-1} words (rnn1, polled for anyway): the residual input
    // of the owned units comes from the matrix waves with the partial tiles (OXO), and the aux-table values of the owned (unit, segment) --
    // c2f (rnn2's gates), c3f / c4f (fc) -- change once per hop and live in this thread's LDS words, re-read when the table row changes.
    float *const CV = smem + L.off_cv;
    auto aux_values = [&](int i, int tt) {              // (gates job of slot i, step tt: refresh if the row changed; every thread for its own segment)
        const int fr = table_row(SEGT[i * 64 + pj] + tt, SEGT[i * 64 + SEG + pj], SEGT[i * 64 + 2 * SEG + pj], magic, mshift, hop, zrow);
        float *cv = CV + i * 1280 + tid;
        if (__builtin_expect(__any(fr + 1 != __float_as_int(cv[1024])), 0)) {
            if constexpr (!LA) {
                const int vo = (fr * 3 * H + prow) * 4;
                cv[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(crs, vo, 0, 0));
                cv[256] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(crs, vo, H * 4, 0));
                cv[512] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(crs, vo, 2 * H * 4, 0));
            }
            cv[768] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(frs, (fr * H + prow) * 4, 0, 0));
            cv[1024] = __int_as_float(fr + 1);          // (0 = nothing cached: the LDS starts zeroed)
        }
    };
    auto pre_xt = [&](int i, int tt) -> unsigned {      // rnn1: x_{tt-1} of slot i, requested one job ahead
        unsigned v = 0u;
        if constexpr (LA) {
            if (tun & 1024) return SENT;                // (A/B and race hunting: no look-ahead)
            if (tt > T0) v = __builtin_amdgcn_raw_buffer_load_b32(xrs, pj * 4, cbase + i * (MAXCL * DSLOTB) + 7 * DLAYERB + ((tt - 1) & (DRING - 1)) * XTB, 16 /* sc1 */);
        }
        return v;
    };
    auto take = [&](unsigned code) -> const float * {      // the next block of the matrix waves is complete: its partial tiles
        ocnt_wait(CNT + 0, kblk + 1u, status, dead, code, t);       // every matrix wave has written its tiles of block kblk
        return PX + pxi * OPXB;
    };
    auto release = [&]() {                               // ... read: the buffer goes back
        ++kblk;
        ocnt_set(CNT + 4 + w, kblk, lane);
        pxi = (pxi + 1 == ONPX) ? 0 : pxi + 1;
    };

    // ---------------- gates: 4-wave partial sum, GRU cell pointwise (ATen gru_cell) -> publish the residual sum and h
    auto job_gates = [&](int i, unsigned xt_pre) {
        const int nb = slot_nb(i);
        const bool live = pj < nb;
        const int sbase = cbase + i * (MAXCL * DSLOTB);
        const int sb = sbase + (t & (DRING - 1)) * XTB;
        OPH(8);
        const float *PB = take(0x720u | (LA ? 0u : 8u));
        OPH(0);
        const float pr = get_partial<3>(PB, 0, pu, pj), pz = get_partial<3>(PB, 1, pu, pj), pn = get_partial<3>(PB, 2, pu, pj);
        const float xown = PB[OXO + ((pu >> 2) * 16 + pj) * 4 + (pu & 3)];       // cI / x1 of (owned unit, segment): validated by the matrix waves with the whole layer
        release();
        aux_values(i, t);
        if (PROF) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        OPH(1);
        const float ghr = GH[i * 768 + tid], ghz = GH[i * 768 + 256 + tid], ghn = GH[i * 768 + 512 + tid];
        const float hprev = HS[i * 256 + tid];
        float gir, giz, gin, xin;
        if constexpr (LA) {
            float xv;
            if (t > T0) {                                // x_{t-1}, sampled by rnn2's CU i
                unsigned xt = xt_pre;
                if (__builtin_expect(__any(live && xt == SENT), 0))
                    wait_for([&] { return !__any(live && xt == SENT); },
                             [&] { xt = __builtin_amdgcn_raw_buffer_load_b32(xrs, pj * 4, sbase + 7 * DLAYERB + ((t - 1) & (DRING - 1)) * XTB, 16 /* sc1 */); },
                             status, dead, 0x721u, t);
                xv = __uint_as_float(xt);
            } else xv = XS[i * 16 + pj];
            gir = pr + fmaf(xv, ux_r, cb_r); giz = pz + fmaf(xv, ux_z, cb_z); gin = pn + fmaf(xv, ux_n, cb_n);
            xin = fmaf(w0o, xv, xown);                              // xi of the owned unit (:208-209)
        } else {
            const float *cv = CV + i * 1280 + tid;
            gir = pr + cv[0]; giz = pz + cv[256]; gin = pn + cv[512];
            xin = xown;
        }
        if (PROF) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        OPH(2);
        const float hn = gru_update_fast(gir, giz, gin, ghr, ghz, ghn, hprev);
        HS[i * 256 + tid] = hn;
        publish4l(xrs, sb + L_XR * DLAYERB + J * 1024, tid, xin + hn, live, false);    // x1 = xi + h1 (:212) / x2 = x1 + h2 (:216): to the other XCD, written through
        publish4l(xrs, sb + L_H * DLAYERB + J * 1024, tid, hn, live, loc_h);
        OPH(3);
    };
    // ---------------- gh(t + 1) = W_hh h(t) + b_hh of the owned (unit, segment): stays in this thread's LDS words
    auto job_gh = [&](int i) {
        OPH(8);
        const float *PB = take(0x722u | (LA ? 0u : 8u));
        OPH(4);
        const float g0 = get_partial<3>(PB, 0, pu, pj) + bh_r, g1 = get_partial<3>(PB, 1, pu, pj) + bh_z, g2 = get_partial<3>(PB, 2, pu, pj) + bh_n;
        release();
        GH[i * 768 + tid] = g0; GH[i * 768 + 256 + tid] = g1; GH[i * 768 + 512 + tid] = g2;
        if (PROF) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        OPH(5);
    };
    // ---------------- fc1 / fc2 + relu -> publish y1 / y2
    auto job_fc = [&](int i) {
        const int sb = cbase + i * (MAXCL * DSLOTB) + (t & (DRING - 1)) * XTB;
        OPH(8);
        const float *PB = take(0x723u | (LA ? 0u : 8u));
        OPH(6);
        const float y = fmaxf(get_partial<3>(PB, 0, pu, pj) + CV[i * 1280 + 768 + tid], 0.f);       // (the aux value: refreshed by this step's gates job of the slot)
        release();
        publish4l(xrs, sb + L_Y * DLAYERB + J * 1024, tid, y, pj < slot_nb(i), LA ? false : loc_h);
        OPH(7);
    };
    // rendezvous of the four service waves (the sampling stage's partial sum and its logit rows)
    auto service_sync = [&]() {
        ++ksync;
        ocnt_set(CNT + 8 + w, ksync, lane);
        ocnt_wait(CNT + 8, ksync, status, dead, 0x72Fu, t);
    };
    // ---------------- rnn1: cI(tt) of the owned 16 rows, slots w, w + 4 by wave w (as wrnn_duo.hip's rnn1 hh role)
    auto cond_step = [&](int tt) {
        if constexpr (LA) {
#pragma unroll 1
            for (int i = w; i < nact; i += NW) {
                const int p = SEGT[i * 64 + fi] + tt;
                const bool valid = fi < slot_nb(i) && p < SEGT[i * 64 + SEG + fi];
                const int fr = magic ? (int)(__umulhi((unsigned)p, magic) >> mshift) : p / hop;
                f32x4 v;
                if (mel_stage) {
                    const int j = p + SEGT[i * 64 + 3 * SEG + fi];
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
    // ---------------- rnn2's CU J = slot: fc3 + mixture-of-logistics sampling (utils/distribution.py:102-121) -> x_t
    u32x4 xs[8];                                         // y2 fragments of the own slot, requested behind its fc job
    auto sample_request = [&]() {
        if constexpr (!LA) {
            const int so = cbase + J * (MAXCL * DSLOTB) + 3 * DLAYERB + (t & (DRING - 1)) * XTB + ((tun & 2048) ? 0x70000000 : 0);     // (bit 11: out of range -> zeros; the sampling stage's own loads follow)
#pragma unroll
            for (int r = 0; r < 8; ++r) xs[r] = __builtin_amdgcn_raw_buffer_load_b128(xrs, voff_frag + r * 1024, so, 16 /* sc1 */);
        }
    };
    auto sample = [&]() {
        if constexpr (!LA) {
            const int i = J;
            const int nb = slot_nb(i);
            const int b0 = GEO[2 * i];
            const int sbase = cbase + i * (MAXCL * DSLOTB);
            const int so = sbase + 3 * DLAYERB + (t & (DRING - 1)) * XTB;
            // this step's sampling noise, pre-transformed (wrnn_noise_mol_kernel): thread (segment tid >> 4, mixture tid & 15)
            const int su = tid >> 4, sm = tid & 15;
            const float *nrow = noise_pre + (size_t)(t - noise_t0) * 11 * Nall;
            const int suc = su < nb ? su : nb - 1;
            const float n0 = nrow[(size_t)(b0 + suc) * 10 + (sm < 10 ? sm : 9)];
            const float n1 = nrow[(size_t)10 * Nall + b0 + suc];
            OPH(8);
            if (tun & 2048) {
#pragma unroll
                for (int r = 0; r < 8; ++r) xs[r] = __builtin_amdgcn_raw_buffer_load_b128(xrs, voff_frag + r * 1024, so, 16 /* sc1 */);
            }
            {
                const bool live = fi < nb;
                if (__builtin_expect(!frag_there(xs, live), 0))
                    wait_for([&] { return frag_there(xs, live); },
                             [&] {
#pragma unroll
                                 for (int r = 0; r < 8; ++r) xs[r] = __builtin_amdgcn_raw_buffer_load_b128(xrs, voff_frag + r * 1024, so, 16 /* sc1 */);
                             },
                             status, dead, 0x72Bu, t);
            }
            // ring hygiene (header): drain, then re-arm the x_t words of this slot in entry (t + 3) % 4
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            OPH(10);
            if (lane == 48) {
                const u32x4 q = {SENT, SENT, SENT, SENT};
                __builtin_amdgcn_raw_buffer_store_b128(q, xrs, 16 * w, sbase + 7 * DLAYERB + ((t + DAHEAD_HH) & (DRING - 1)) * XTB, 16 /* sc1 */);
            }
            float b[32];
            frag_to_b(xs, b);
            put_partial<2>(PY, w, 0, lane, mfma1(A3[0], b));
            put_partial<2>(PY, w, 1, lane, mfma1(A3[1], b));
            service_sync();
            {   // 30 logit rows x 16 segments: thread (rows pu and 16 + pu, segment pj)
                const float lg = get_partial<2>(PY, 0, pu, pj) + b3a;
                const float lg2 = get_partial<2>(PY, 1, pu, pj) + b3b;
                LOG[pj * OLOGS + pu] = lg;
                if (dbgl && pj < nb) dbgl[((size_t)t * Nall + b0 + pj) * C + pu] = lg;
                if (pu < 14) {
                    LOG[pj * OLOGS + 16 + pu] = lg2;
                    if (dbgl && pj < nb) dbgl[((size_t)t * Nall + b0 + pj) * C + 16 + pu] = lg2;
                }
            }
            service_sync();
            {   // 16-lane row = one segment (su), lane sm = mixture
                float best = (sm < 10) ? mol_gumbel_pre(LOG[su * OLOGS + sm], n0) : -INFINITY;
                int bidx = sm;
                argmax_row16(best, bidx);
                if (sm == 0 && su < nb) {
                    float xv = mol_sample_pre(LOG[su * OLOGS + 10 + bidx], LOG[su * OLOGS + 20 + bidx], n1);
                    outp[(size_t)(b0 + su) * Tall + t] = xv;
                    if (forcex) xv = forcex[(size_t)(b0 + su) * Tall + t];
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(xv), xrs, su * 4, sbase + 7 * DLAYERB + (t & (DRING - 1)) * XTB, 16 /* sc1 */);
                }
            }
            OPH(11);
        }
    };

    cond_step(T0);                                      // (the two steps a launch starts with; every later one is formed two steps ahead)
    if (T0 + 1 < T1) cond_step(T0 + 1);
    unsigned xt_next = pre_xt(0, T0);
    for (; t < T1; ++t) {
        if constexpr (LA) {
            if (t + 2 < T1) cond_step(t + 2);
        }
        // ring hygiene: last step's re-arm stores (and the cI just formed) are out before anything of this step is published
        OPH(8);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        OPH(9);
#pragma unroll 1
        for (int i = 0; i < nact; ++i) {
            const unsigned cur = xt_next;
            if (i + 1 < nact) xt_next = pre_xt(i + 1, t);
            job_gates(i, cur);
        }
        bool sampled = !sampler;
#pragma unroll 1
        for (int i = 0; i < nact; ++i) {
            job_gh(i);
            if (i + 1 == nact && t + 1 < T1) xt_next = pre_xt(0, t + 1);
            job_fc(i);
            if constexpr (!LA) {
                if (sampler) {
                    if (i == J) sample_request();
                    else if (i == J + 1) { sample(); sampled = true; }
                }
            }
        }
        if constexpr (!LA) {
            if (!sampled) sample();                     // (the sampler of the last slot)
        }
        OPH(8);
        {   // ring hygiene, once per step, behind every poll of the step: re-arm this wave's own words of entry (t + 2) % 4 in the three layers it
            // publishes, for every slot (drained at the top of the next step)
            const int which = lane >> 4;
            const int layer = which == 0 ? L_H : (which == 1 ? L_Y : L_XR);
            const bool lloc = LA ? (which == 0 && loc_h) : (which < 2 && loc_h);       // h1 | h2, y2 stay inside the XCD (when the placement was seen)
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
        OPH(12);
    }
    if (PROF && w == 0 && lane == 0 && a.prof)
        for (int k = 0; k < 16; ++k) a.prof[(size_t)(blockIdx.x & 255) * 32 + 16 + k] += pacc[k];
    // ---- what the next launch of this round needs: gh(T1), h, the segment table and, for rnn1, x_{T1-1} -> the saved state (wrnn_duo.hip's layout)
#pragma unroll 1
    for (int i = 0; i < nact; ++i) {
        const int nb = slot_nb(i);
        const int sbase = cbase + i * (MAXCL * DSLOTB);
        float *sg = state + state_wg + (size_t)i * LGRP;
        sg[tid] = GH[i * 768 + tid]; sg[256 + tid] = GH[i * 768 + 256 + tid]; sg[512 + tid] = GH[i * 768 + 512 + tid];
        sg[O_HOWN + tid] = HS[i * 256 + tid];
        if (tid < 2 * SEG) reinterpret_cast<int *>(sg + O_SP)[tid] = SEGT[i * 64 + tid];
        if constexpr (LA) {
            const int sx = sbase + 7 * DLAYERB + ((T1 - 1) & (DRING - 1)) * XTB;
            unsigned v = __builtin_amdgcn_raw_buffer_load_b32(xrs, fi * 4, sx, 16 /* sc1 */);
            wait_for([&] { return !__any(fi < nb && v == SENT); },
                     [&] { v = __builtin_amdgcn_raw_buffer_load_b32(xrs, fi * 4, sx, 16 /* sc1 */); }, status, dead, 0x724u, T1);
            if (tid < SEG) sg[O_XS + tid] = (tid < nb) ? __uint_as_float(v) : 0.f;
        }
    }
    cond_leave();
    (void)resume;
}

// Grid = clusters x 64 workgroups of 512 threads (one per CU), cooperative launch.  Whole XCDs per cluster (speed only).
template <bool PROF>
__global__ __launch_bounds__(ONT, 1) void wrnn_octo_kernel(const LoopArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int cl, wg;
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
    if (cl >= a.NG) return;                             // a cluster without a group of this round
    // first half of a cluster's 64 workgroups: rnn1, second half: rnn2 (with 4 clusters: one XCD each)
    const int layer = wg / LNJ, J = wg % LNJ;
    const int tid = threadIdx.x, lane = tid & 63, w8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = a.G;
    const OctoLds L = octo_lds(G);
    // ---- placement handshake (as wrnn_duo.hip): a layer whose producers and consumers all sit on one XCD is exchanged through that XCD's L2
    bool loc_a = false, loc_b = false;
    {
        int *TAB = reinterpret_cast<int *>(smem);
        unsigned *tab = a.xcc_tab + cl * LNWGC;
        if (tid == 0) {
            const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xFu;          // HW_REG_XCC_ID
            __hip_atomic_store(tab + wg, xcc + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        unsigned v = 1u;
        if (tid < LNWGC) {
            unsigned spins = 0;
            v = ld_agent32(tab + tid);
            while (v == 0u && ++spins < 200000u) {
                __builtin_amdgcn_s_sleep(2);
                v = ld_agent32(tab + tid);
            }
            TAB[tid] = (int)v;
        }
        __syncthreads();
        const int ok_a = (tid < LNJ) ? (v != 0u && (int)v == TAB[0]) : 1;
        const int ok_b = (tid >= LNJ && tid < LNWGC) ? (v != 0u && (int)v == TAB[LNJ]) : 1;
        loc_a = __syncthreads_and(ok_a) != 0;
        loc_b = __syncthreads_and(ok_b) != 0;
        if (a.tuning & 256) { loc_a = false; loc_b = false; }        // A/B: everything written through
        __syncthreads();
    }
    // ---- LDS: zero, then the fc tile in A-fragment order, the slots' geometry and state (all eight waves; the roles split afterwards and
    //      never meet at a barrier again)
    for (int q = tid; q < L.total; q += ONT) smem[q] = 0.f;
    __syncthreads();
    {
        const float *Wfc = layer == 0 ? a.fc1_w : a.fc2_w;
        float *FCW = smem + L.off_fcw;
        for (int q = tid; q < XT / 4; q += ONT) {       // float4 q = (wave wq, k-block r, lane lq): W[16 J + (lq & 15)][128 wq + 16 r + 4 (lq >> 4) ..]
            const int lq = q & 63, r = (q >> 6) & 7, wq = q >> 9;
            reinterpret_cast<float4 *>(FCW)[q] = *reinterpret_cast<const float4 *>(Wfc + (size_t)(LU * J + (lq & 15)) * (H + AUX) + KCH * wq + 16 * r + 4 * (lq >> 4));
        }
        float *HS = smem + L.off_h, *GH = smem + L.off_gh, *XS = smem + L.off_xs;
        int *SEGT = reinterpret_cast<int *>(smem + L.off_seg);
        int *GEO = reinterpret_cast<int *>(smem + L.off_misc);
        const size_t state_wg = ((size_t)(cl * LNWGC + 2 * J + layer) * G) * LGRP;
        const float *bhh = layer == 0 ? a.b_hh1 : a.b_hh2;
        const int NR = a.Btot, NGR = a.NG;
        for (int i = 0; i < G; ++i) {
            const int g = cl + ncl * i;
            if (g >= NGR) break;
            const int b0 = (int)(((long)g * NR) / NGR), nb = (int)(((long)(g + 1) * NR) / NGR) - b0;
            if (tid == 0) { GEO[2 * i] = a.rb0 + b0; GEO[2 * i + 1] = nb; }
            if (tid < 256) {
                const int pu = 4 * (tid >> 6) + (tid & 3);
                const int prow = LU * J + pu;
                if (a.resume) {
                    const float *sg = a.state + state_wg + (size_t)i * LGRP;
                    HS[i * 256 + tid] = sg[O_HOWN + tid];
                    GH[i * 768 + tid] = sg[tid]; GH[i * 768 + 256 + tid] = sg[256 + tid]; GH[i * 768 + 512 + tid] = sg[512 + tid];
                    if (tid < SEG) XS[i * 16 + tid] = sg[O_XS + tid];
                    if (tid < 2 * SEG) SEGT[i * 64 + tid] = reinterpret_cast<const int *>(sg + O_SP)[tid];
                } else {                                // fatchord_version.py:194-196: h1 = h2 = 0, x = 0 (LDS is zero); gh(0) = W_hh . 0 + b_hh
                    GH[i * 768 + tid] = bhh[prow]; GH[i * 768 + 256 + tid] = bhh[H + prow]; GH[i * 768 + 512 + tid] = bhh[2 * H + prow];
                    if (tid < SEG) {
                        const int sc = a.rb0 + b0 + (tid < nb ? tid : nb - 1);
                        SEGT[i * 64 + tid] = a.seg_pos[sc];
                        SEGT[i * 64 + SEG + tid] = a.seg_lim[sc];
                    }
                }
            }
        }
        __syncthreads();
        for (int i = 0; i < G; ++i) {                   // this slab's table rows and the mel offsets of every segment (after the table above is in place)
            const int g = cl + ncl * i;
            if (g >= NGR) break;
            const int b0 = (int)(((long)g * NR) / NGR), nb = (int)(((long)(g + 1) * NR) / NGR) - b0;
            if (tid < SEG) {
                const int sc = a.rb0 + b0 + (tid < nb ? tid : nb - 1);
                SEGT[i * 64 + 2 * SEG + tid] = sc * a.tab_fps - (SEGT[i * 64 + tid] + a.tab_t0) / a.hop;
                SEGT[i * 64 + 3 * SEG + tid] = a.mel_stage ? a.seg_moff[sc] : 0;
            }
        }
        __syncthreads();
    }
#ifndef OCTO_ROLES
#define OCTO_ROLES 15                         // (register-allocation diagnosis: compile with a subset of the four roles, -DOCTO_ROLES=<mask>)
#endif
    const int w = w8 & 3;
    if (w8 < 4) {
        if (layer == 0) { if constexpr ((OCTO_ROLES & 1) != 0) octo_matrix<true, PROF>(a, smem, cl, J, ncl, w, lane); }
        else { if constexpr ((OCTO_ROLES & 2) != 0) octo_matrix<false, PROF>(a, smem, cl, J, ncl, w, lane); }
    } else {
        if (layer == 0) { if constexpr ((OCTO_ROLES & 4) != 0) octo_service<true, PROF>(a, smem, cl, J, ncl, w, lane, loc_a); }
        else { if constexpr ((OCTO_ROLES & 8) != 0) octo_service<false, PROF>(a, smem, cl, J, ncl, w, lane, loc_b); }
    }
}

size_t octo_lds_bytes(int G) { return (size_t)octo_lds(G).total * sizeof(float); }
constexpr int OMAXG = 4;                     // slots in flight per cluster: 8.3 KB of LDS per slot beside 114 KB of operand blocks, partial tiles and the fc tile
static_assert(sizeof(float) * (size_t)octo_lds(OMAXG).total <= 160 * 1024, "LDS carve");
int octo_max_depth() { return OMAXG; }
// clusters this device can host at one 512-thread workgroup per CU (64 CUs per cluster)
int octo_clusters(int n_cus)
{
    int ncl = n_cus / LNWGC;
    if (ncl > MAXCL) ncl = MAXCL;
    while (ncl > 1 && (8 % ncl) != 0) --ncl;
    return ncl;
}

hipError_t launch_octo(const LoopArgs &args, int ncl, int mode, hipStream_t stream)
{
    if (ncl < 1 || args.G < 1 || args.G > OMAXG || mode != 1 || !args.u1 || !args.xcc_tab) return hipErrorInvalidValue;
    const size_t lds = octo_lds_bytes(args.G);
    const void *fn = args.prof ? (const void *)wrnn_octo_kernel<true> : (const void *)wrnn_octo_kernel<false>;       // (phase clocks: wrnn_options.phase_clocks)
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    LoopArgs a = args;
    void *params[] = {(void *)&a};
    return hipLaunchCooperativeKernel(fn, dim3(ncl * LNWGC), dim3(ONT), params, (unsigned)lds, stream);
}

}  // namespace wrnn
