// Where does `buffer_load_dwordx4 ... lds` put its data?  (gfx950; checks the layout wrnn_duo.hip's prefetch8 relies on: destination =
// M0 base + instruction offset + lane * 16, the instruction offset -- 12 bits -- added to BOTH the global and the LDS address.)
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 scripts/micro/lds_dma_layout.hip -o /tmp/lds_dma && /tmp/lds_dma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
__global__ void k(const unsigned *src, unsigned *dump)
{
    extern __shared__ __attribute__((aligned(16))) unsigned smem[];
    for (int q = threadIdx.x; q < 4 * 2048; q += blockDim.x) smem[q] = 0xDEADBEEFu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(src), 0, 1 << 20, 0x00020000);
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned *dst = smem + w * 2048;
    const int voff = (w * 8 * 64 + lane) * 16;
    lds_void *d0 = (lds_void *)dst, *d1 = (lds_void *)(dst + 1024);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, d0, 16, voff, 0, 0, 16);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, d0, 16, voff, 0, 1024, 16);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, d0, 16, voff, 0, 2048, 16);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, d0, 16, voff, 0, 3072, 16);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, d1, 16, voff + 4096, 0, 0, 16);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, d1, 16, voff + 4096, 0, 1024, 16);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, d1, 16, voff + 4096, 0, 2048, 16);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, d1, 16, voff + 4096, 0, 3072, 16);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int q = threadIdx.x; q < 4 * 2048; q += blockDim.x) dump[q] = smem[q];
}
int main()
{
    const int N = 4 * 2048;
    std::vector<unsigned> h(N), o(N);
    for (int i = 0; i < N; ++i) h[i] = i;
    unsigned *d, *dd;
    hipMalloc(&d, 1 << 20); hipMalloc(&dd, N * 4);
    hipMemset(d, 0, 1 << 20);
    hipMemcpy(d, h.data(), N * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), N * 4, 0, d, dd);
    hipMemcpy(o.data(), dd, N * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < N; ++i) if (o[i] != (unsigned)i) { if (bad < 8) printf("lds word %d holds %u (0x%x)\n", i, o[i], o[i]); ++bad; }
    printf("lds_dma_layout: %s (%d of %d words differ from the lane-linear image)\n", bad ? "UNEXPECTED" : "OK", bad, N);
    return bad != 0;
}
