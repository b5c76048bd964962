// Does VALU work hide behind v_mfma_f32_16x16x4_f32 on gfx950 -- (a) from the SAME wave, interleaved in program order, (b) from ANOTHER wave of the same SIMD?
// One workgroup per CU.  (a) 256 threads: every wave runs 96-MFMA blocks with K independent v_fma_f32 behind each MFMA.  (b) 512 threads (two waves per SIMD):
// waves 0-3 run bare MFMA blocks, waves 4-7 run V v_fma_f32 per iteration; each kind is also timed alone.  Prints shader clocks (s_memtime) per block.
// Build + run: hipcc --offload-arch=gfx950 -O2 scripts/micro/mfma_valu_coexec.hip -o /tmp/coexec && /tmp/coexec
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)
#define REP96(x) REP32(x) REP32(x) REP32(x)
#define MF "v_mfma_f32_16x16x4_f32 %0, %3, %4, %0\n v_mfma_f32_16x16x4_f32 %1, %3, %4, %1\n v_mfma_f32_16x16x4_f32 %2, %3, %4, %2\n"
#define VF "v_fma_f32 %5, %5, %6, %6\n"
template <int K>
__device__ __forceinline__ void block(f32x4 &c0, f32x4 &c1, f32x4 &c2, float a, float b, float &v0, float &v1, float &v2, float &v3, float &v4, float &v5, float m)
{
    // 32 x (3 MFMAs on three accumulators, each followed by K VALU on K different registers: no VALU dependency stalls)
#define ONE(vreg) "v_fma_f32 %" #vreg ", %" #vreg ", %11, %11\n"
#define M1(acc) "v_mfma_f32_16x16x4_f32 %" #acc ", %3, %4, %" #acc "\n"
#define GAP ((K >= 1) ? ONE(5) : "") ((K >= 2) ? ONE(6) : "") ((K >= 3) ? ONE(7) : "") ((K >= 4) ? ONE(8) : "") ((K >= 5) ? ONE(9) : "") ((K >= 6) ? ONE(10) : "")
    if constexpr (K == 0) { asm volatile(REP32(M1(0) M1(1) M1(2)) : "+v"(c0), "+v"(c1), "+v"(c2) : "v"(a), "v"(b), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(v4), "v"(v5), "v"(m)); }
    if constexpr (K == 1) { asm volatile(REP32(M1(0) ONE(5) M1(1) ONE(5) M1(2) ONE(5)) : "+v"(c0), "+v"(c1), "+v"(c2) : "v"(a), "v"(b), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(v4), "v"(v5), "v"(m)); }
    if constexpr (K == 2) { asm volatile(REP32(M1(0) ONE(5) ONE(6) M1(1) ONE(5) ONE(6) M1(2) ONE(5) ONE(6)) : "+v"(c0), "+v"(c1), "+v"(c2) : "v"(a), "v"(b), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(v4), "v"(v5), "v"(m)); }
    if constexpr (K == 4) { asm volatile(REP32(M1(0) ONE(5) ONE(6) ONE(7) ONE(8) M1(1) ONE(5) ONE(6) ONE(7) ONE(8) M1(2) ONE(5) ONE(6) ONE(7) ONE(8)) : "+v"(c0), "+v"(c1), "+v"(c2) : "v"(a), "v"(b), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(v4), "v"(v5), "v"(m)); }
    if constexpr (K == 6) { asm volatile(REP32(M1(0) ONE(5) ONE(6) ONE(7) ONE(8) ONE(9) ONE(10) M1(1) ONE(5) ONE(6) ONE(7) ONE(8) ONE(9) ONE(10) M1(2) ONE(5) ONE(6) ONE(7) ONE(8) ONE(9) ONE(10)) : "+v"(c0), "+v"(c1), "+v"(c2) : "v"(a), "v"(b), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(v4), "v"(v5), "v"(m)); }
}
__device__ __forceinline__ void valu_block(float &v0, float &v1, float &v2, float &v3, float m)
{
    asm volatile(REP96("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4\n") : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(m));
}
// mode 0..4: same-wave interleave K = 0, 1, 2, 4, 6.  mode 10: waves 0-3 MFMA blocks, waves 4-7 VALU blocks (384 v_fma each).  mode 11: only the MFMA waves work.  mode 12: only the VALU waves.
__global__ void k(int mode, int iters, unsigned long long *out, float *sink)
{
    const int w = threadIdx.x >> 6;
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0;
    float a = threadIdx.x * 1e-3f, b = 1e-3f, v0 = 1, v1 = 2, v2 = 3, v3 = 4, v4 = 5, v5 = 6, m = 0.999f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (mode == 0) block<0>(c0, c1, c2, a, b, v0, v1, v2, v3, v4, v5, m);
        else if (mode == 1) block<1>(c0, c1, c2, a, b, v0, v1, v2, v3, v4, v5, m);
        else if (mode == 2) block<2>(c0, c1, c2, a, b, v0, v1, v2, v3, v4, v5, m);
        else if (mode == 3) block<4>(c0, c1, c2, a, b, v0, v1, v2, v3, v4, v5, m);
        else if (mode == 4) block<6>(c0, c1, c2, a, b, v0, v1, v2, v3, v4, v5, m);
        else {
            const bool mf = w < 4;
            if (mf && mode != 12) block<0>(c0, c1, c2, a, b, v0, v1, v2, v3, v4, v5, m);
            if (!mf && mode != 11) valu_block(v0, v1, v2, v3, m);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[w] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + v0 + v1 + v2 + v3 + v4 + v5;
}
int main()
{
    unsigned long long *out, h[8];
    float *sink;
    (void)hipMalloc(&out, 64); (void)hipMalloc(&sink, 256 * 512 * 4);
    const int iters = 200;
    const char *names[] = {"same wave, 96 MFMA + 0 VALU per block", "same wave, 1 VALU per MFMA", "same wave, 2 VALU per MFMA", "same wave, 4 VALU per MFMA", "same wave, 6 VALU per MFMA"};
    for (int mode = 0; mode < 5; ++mode) {
        hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, mode, iters, out, sink);
        (void)hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
        printf("%-44s %8.1f clocks per 96-MFMA block (wave 0)\n", names[mode], (double)h[0] / iters);
    }
    for (int mode = 10; mode <= 12; ++mode) {
        (void)hipMemset(out, 0, 64);
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, iters, out, sink);
        (void)hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
        printf("two waves per SIMD, mode %d (10 both, 11 MFMA waves only, 12 VALU waves only): MFMA wave %8.1f, VALU wave (384 v_fma) %8.1f clocks per iteration\n", mode,
               (double)h[0] / iters, (double)h[4] / iters);
    }
    return 0;
}
