// Feasibility of the WAVE-SPECIALISED cut of the dense loop kernel (round 6, DESIGN.md 9.1 d): ONE 512-thread workgroup per CU, two waves per SIMD --
// waves 0-3 ("matrix waves") hold rnn W_ih AND W_hh of the CU's 16 units (192 weight registers) and do nothing but fetch an operand (LDS-DMA), check it,
// run a 96-MFMA block and drop the partial tiles into LDS; waves 4-7 ("service waves") hold the fc tile and do everything else: the 4-wave partial sums, the
// GRU pointwise math, the publishes, the fc stage (32 MFMAs, its own partial sum).  The two kinds meet through LDS counters only (no s_barrier in the loop).
// This micro-benchmark has no inter-CU dependency (every operand is "there"): it measures the THROUGHPUT bound of the cut -- shader clocks per slot-step of
// a matrix wave and of a service wave, together and alone -- against wrnn_duo_kernel's 13.3 k clocks per slot-step (profiles/r04o_phase_clocks.log).
// Build + run: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I wavernn_amd/csrc scripts/micro/octo_feasibility.hip -o /tmp/octo && /tmp/octo
#include <cstdio>
#include <vector>
#include "wrnn_ring.h"
using namespace wrnn;

#define PH(q) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); ph[q] += n_ - pl; pl = n_; } while (0)
constexpr int NPX = 3;                         // partial-tile buffers of the matrix waves (12 KB each)
typedef __attribute__((address_space(3))) void lds_void;
__device__ __forceinline__ void dma8(__amdgpu_buffer_rsrc_t rs, float *dst, int voff, int soff)
{
    lds_void *d0 = (lds_void *)dst, *d1 = (lds_void *)(dst + 1024);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, d0, 16, voff, soff, 0, 16);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, d0, 16, voff, soff, 1024, 16);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, d0, 16, voff, soff, 2048, 16);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, d0, 16, voff, soff, 3072, 16);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, d1, 16, voff + 4096, soff, 0, 16);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, d1, 16, voff + 4096, soff, 1024, 16);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, d1, 16, voff + 4096, soff, 2048, 16);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, d1, 16, voff + 4096, soff, 3072, 16);
}
__device__ __forceinline__ unsigned cnt_ld(unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void cnt_add(unsigned *p)
{
    asm volatile("" ::: "memory");
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
}
__device__ int g_nosleep;
__device__ __forceinline__ void cnt_wait(unsigned *p, unsigned target, bool nosleep = false)
{
    while ((int)(cnt_ld(p) - target) < 0) { if (!nosleep) __builtin_amdgcn_s_sleep(1); else __builtin_amdgcn_s_sleep(8); }
    asm volatile("" ::: "memory");
}

// mode 0: both kinds; 1: matrix waves only (the service waves only acknowledge); 2: service waves only (the matrix waves only bump their counter)
template <bool FCX>
__global__ __launch_bounds__(512, 1) void k(const float *W, float *xbuf, int steps, int nslot, int mode, unsigned long long *out, float *sink)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *OPX = smem;                          // [4 waves][2][2048] operand blocks of the matrix waves
    float *PX = OPX + 4 * 2 * 2048;             // [NPX][4 waves][3][256]
    float *PY = PX + NPX * 4 * 3 * 256;         // [2][4][256]   (FCX: the fc tile in A-fragment order, 8192 floats)
    float *GH = PY + (FCX ? 8192 : 2 * 4 * 256);               // [4 slots][3][256]
    float *HS = GH + 4 * 3 * 256;               // [4 slots][256]
    unsigned *CNT = reinterpret_cast<unsigned *>(HS + 4 * 256);      // 0 prodX, 1 consX, 2 prodY
    float *FCW = PY;
    const int tid = threadIdx.x, lane = tid & 63, w8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool matrix = w8 < 4;
    const int w = w8 & 3;
    const int t4 = tid & 255;
    const int fi = lane & 15, kq = lane >> 4;
    const int kbase_lane = KCH * w + 4 * kq;
    const int pu = 4 * w + (t4 & 3), pj = (t4 >> 2) & 15;
    const int J = blockIdx.x & 31;
    for (int q = tid; q < (int)(CNT + 4 - reinterpret_cast<unsigned *>(smem)); q += 512) smem[q] = 0.f;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t xrs = make_rsrc(xbuf, 64u << 20);
    const int voff_frag = frag_off(w, 0, lane) * 4;
    float acc_sink = 0.f;
    unsigned long long t0 = 0, t1 = 0;
    if (matrix) {
        float A_ih[3][AF], A_hh[3][AF];
#pragma unroll
        for (int g = 0; g < 3; ++g) load_afrag(A_ih[g], W, H, g * H + LU * J + fi, true, kbase_lane);
#pragma unroll
        for (int g = 0; g < 3; ++g) load_afrag(A_hh[g], W + 3 * H * H, H, g * H + LU * J + fi, true, kbase_lane);
        float *OP = OPX + w * 4096;
        unsigned kblk = 0;
        unsigned long long ph[5] = {0, 0, 0, 0, 0}, pl;
        dma8(xrs, OP, voff_frag, 0);
        __syncthreads();
        t0 = __builtin_amdgcn_s_memtime();
        for (int t = 0; t < steps; ++t) {
#pragma unroll 1
            for (int i = 0; i < nslot; ++i) {
#pragma unroll
                for (int which = 0; which < (FCX ? 3 : 2); ++which) {
                    u32x4 x[8];
                    float b[32];
                    if (mode != 2) {
                        pl = __builtin_amdgcn_s_memtime();
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        PH(0);
                        const float *src = OP + (kblk & 1) * 2048;
#pragma unroll
                        for (int r = 0; r < 8; ++r) x[r] = *reinterpret_cast<const u32x4 *>(src + r * 256 + lane * 4);
                        const bool there = frag_there(x, true);
                        if (!there) acc_sink += 1.f;
                        // the next block's operand -> the other half
                        dma8(xrs, OP + ((kblk + 1) & 1) * 2048, voff_frag, (((kblk + 1) & 7) * 17 + (which ? 1 : 5)) * XTB);
                        frag_to_b(x, b);
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        PH(1);
                        cnt_wait(CNT + 1, 4u * (kblk + 1u - NPX));          // the partial buffer is free (all four service waves have read block kblk - NPX)
                        PH(2);
                        float *P = PX + (kblk % NPX) * (4 * 3 * 256);
                        f32x4 o0, o1, o2;
                        if (which == 0) mfma3s(A_ih[0], A_ih[1], A_ih[2], b, o0, o1, o2);
                        else if (which == (FCX ? 2 : 1)) mfma3s(A_hh[0], A_hh[1], A_hh[2], b, o0, o1, o2);
                        else o0 = mfma1_lds(FCW + frag_off(w, 0, lane), b);
                        put_partial<3>(P, w, 0, lane, o0);
                        if (which != 1 || !FCX) {
                            put_partial<3>(P, w, 1, lane, o1);
                            put_partial<3>(P, w, 2, lane, o2);
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        PH(3);
                    } else cnt_wait(CNT + 1, 4u * (kblk + 1u - NPX));
                    cnt_add(CNT + 0);
                    ++kblk;
                }
            }
        }
        t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0 && blockIdx.x == 0 && w == 0) for (int q = 0; q < 4; ++q) out[8 + q] = ph[q];
    } else {
        float A_fc[AF];
        load_afrag(A_fc, W + 6 * H * H, H, LU * J + fi, true, kbase_lane);
        const float cb = W[pu], ux = W[H + pu];
        unsigned kblk = 0, kfc = 0;
        __syncthreads();
        t0 = __builtin_amdgcn_s_memtime();
        u32x4 x[8];
        if (!FCX) {
#pragma unroll
            for (int r = 0; r < 8; ++r) x[r] = __builtin_amdgcn_raw_buffer_load_b128(xrs, voff_frag + r * 1024, 3 * XTB, 16);
        }
        for (int t = 0; t < steps; ++t) {
#pragma unroll 1
            for (int i = 0; i < nslot; ++i) {
                const int sb = (i * 17) * XTB * 4 + (t & 3) * XTB;
                // ---- back half of the gates block
                {
                    const unsigned xo = __builtin_amdgcn_raw_buffer_load_b32(xrs, (J * 256 + t4) * 4, sb + 4 * XTB, 16);
                    const unsigned xt = __builtin_amdgcn_raw_buffer_load_b32(xrs, pj * 4, sb + 7 * XTB, 16);
                    cnt_wait(CNT + 0, 4u * (kblk + 1u), mode == 5);
                    float hn = 0.f;
                    if (mode != 1) {
                        const float *P = PX + (kblk % NPX) * (4 * 3 * 256);
                        const float pr = get_partial<3>(P, 0, pu, pj), pz = get_partial<3>(P, 1, pu, pj), pn = get_partial<3>(P, 2, pu, pj);
                        const float ghr = GH[((i & 3) * 3 + 0) * 256 + t4], ghz = GH[((i & 3) * 3 + 1) * 256 + t4], ghn = GH[((i & 3) * 3 + 2) * 256 + t4];
                        const float hprev = HS[(i & 3) * 256 + t4];
                        const float xv = __uint_as_float(xt);
                        hn = gru_update_fast(pr + fmaf(xv, ux, cb), pz + fmaf(xv, ux, cb), pn + fmaf(xv, ux, cb), ghr, ghz, ghn, hprev);
                        HS[(i & 3) * 256 + t4] = hn;
                    }
                    cnt_add(CNT + 1);
                    ++kblk;
                    if (mode != 1 && mode != 3) {
                        publish4l(xrs, sb + 5 * XTB + J * 1024, t4, __uint_as_float(xo) + hn, true, false);
                        publish4l(xrs, sb + 0 * XTB + J * 1024, t4, hn, true, true);
                    }
                }
                // ---- the fc stage of a slot (operand requested one job ahead)
                if (FCX) {
                    const float c0 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrs, (i * 512 + pu) * 4, 9 * XTB, 0));
                    cnt_wait(CNT + 0, 4u * (kblk + 1u), mode == 5);
                    float yv = 0.f;
                    if (mode != 1) yv = fmaxf(get_partial<3>(PX + (kblk % NPX) * (4 * 3 * 256), 0, pu, pj) + c0, 0.f);
                    cnt_add(CNT + 1);
                    ++kblk;
                    if (mode != 1 && mode != 3) publish4l(xrs, sb + 2 * XTB + J * 1024, t4, yv, true, false);
                    else acc_sink += yv;
                } else if (mode != 1 && mode != 4) {
                    const float c0 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrs, (i * 512 + pu) * 4, 9 * XTB, 0));
                    const bool there = frag_there(x, true);
                    if (!there) acc_sink += 1.f;
                    float b[32];
                    frag_to_b(x, b);
                    const f32x4 o = mfma1(A_fc, b);
#pragma unroll
                    for (int r = 0; r < 8; ++r) x[r] = __builtin_amdgcn_raw_buffer_load_b128(xrs, voff_frag + r * 1024, sb + 6 * XTB, 16);
                    float *P = PY + (kfc & 1) * (4 * 256);
                    put_partial<1>(P, w, 0, lane, o);
                    cnt_add(CNT + 2);
                    cnt_wait(CNT + 2, 4u * (kfc + 1u));
                    ++kfc;
                    const float yv = fmaxf(get_partial<1>(P, 0, pu, pj) + c0, 0.f);
                    if (mode != 3) publish4l(xrs, sb + 2 * XTB + J * 1024, t4, yv, true, false);
                    else acc_sink += yv;
                }
                // ---- back half of the gh block
                {
                    cnt_wait(CNT + 0, 4u * (kblk + 1u), mode == 5);
                    if (mode != 1) {
                        const float *P = PX + (kblk % NPX) * (4 * 3 * 256);
                        GH[((i & 3) * 3 + 0) * 256 + t4] = get_partial<3>(P, 0, pu, pj) + cb;
                        GH[((i & 3) * 3 + 1) * 256 + t4] = get_partial<3>(P, 1, pu, pj) + cb;
                        GH[((i & 3) * 3 + 2) * 256 + t4] = get_partial<3>(P, 2, pu, pj) + cb;
                    }
                    cnt_add(CNT + 1);
                    ++kblk;
                }
            }
        }
        t1 = __builtin_amdgcn_s_memtime();
        if (!FCX) acc_sink += __uint_as_float(x[0].x);
    }
    if (lane == 0 && blockIdx.x == 0) out[w8] = t1 - t0;
    sink[blockIdx.x * 512 + tid] = acc_sink;
}

int main()
{
    unsigned long long *out, h[16];
    float *sink, *W, *xb;
    (void)hipMalloc(&out, 128); (void)hipMalloc(&sink, 256 * 512 * 4);
    (void)hipMalloc(&W, 7 * H * H * 4); (void)hipMalloc(&xb, 64u << 20);
    (void)hipMemset(W, 0, 7 * H * H * 4); (void)hipMemset(xb, 0, 64u << 20);
    const int lds = (4 * 2 * 2048 + NPX * 4 * 3 * 256 + 8192 + 4 * 3 * 256 + 4 * 256 + 16) * 4;
    (void)hipFuncSetAttribute((const void *)k<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute((const void *)k<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int steps = 300;
    const char *names[] = {"matrix + service waves", "matrix waves alone", "service waves alone", "both, no publishes", "both, no fc stage", "both, s_sleep 8 in service spins"};
    for (int fcx = 0; fcx < 2; ++fcx)
    for (int nslot : {4, 8})
        for (int mode = 0; mode < 6; ++mode) {
            if (fcx && mode == 4) continue;
            (void)hipMemset(out, 0, 128);
            if (fcx) hipLaunchKernelGGL(k<true>, dim3(256), dim3(512), lds, 0, W, xb, steps, nslot, mode, out, sink);
            else hipLaunchKernelGGL(k<false>, dim3(256), dim3(512), lds, 0, W, xb, steps, nslot, mode, out, sink);
            if (hipMemcpy(h, out, 128, hipMemcpyDeviceToHost) != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
            printf("%s%d slots, %-24s matrix wave %8.1f  service wave %8.1f  clocks per slot-step (duo: 13.3 k)\n", fcx ? "fc on the matrix waves (A operand in LDS): " : "", nslot, names[mode], (double)h[0] / (steps * nslot),
                   (double)h[4] / (steps * nslot));
            printf("      matrix wave 0, clocks per BLOCK: operand wait %7.1f  lds read + check + dma issue %7.1f  buffer wait %7.1f  mfma + partials %7.1f\n", (double)h[8] / ((2 + fcx) * steps * nslot),
                   (double)h[9] / ((2 + fcx) * steps * nslot), (double)h[10] / ((2 + fcx) * steps * nslot), (double)h[11] / ((2 + fcx) * steps * nslot));
        }
    return 0;
}
