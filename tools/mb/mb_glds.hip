// micro-benchmark: per-CU throughput of global_load_lds (LDS-DMA) from an L2-resident buffer vs
// ordinary global_load_dwordx4 -> VGPR, at 1/2 workgroups of 4/8 waves per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int NW, int DEPTH>
__global__ __launch_bounds__(NW * 64) void glds_kernel(const unsigned char* __restrict__ src, size_t span, int iters, unsigned* out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned char* base = src + ((size_t)blockIdx.x * 65536) % span;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const unsigned char* g = base + ((size_t)(it * DEPTH + d) * NW + wave) * 1024 % 65536 + lane * 16;
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)g, (LDS_AS void*)(smem + (d * NW + wave) * 1024), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (threadIdx.x == 0) out[blockIdx.x] = smem[0];
}

template <int NW, int DEPTH>
__global__ __launch_bounds__(NW * 64) void vload_kernel(const unsigned char* __restrict__ src, size_t span, int iters, unsigned* out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned char* base = src + ((size_t)blockIdx.x * 65536) % span;
    u32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        u32x4 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
            v[d] = *(const u32x4*)(base + ((size_t)(it * DEPTH + d) * NW + wave) * 1024 % 65536 + lane * 16);
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) acc ^= v[d];
    }
    if (acc[0] == 0x12345678) out[blockIdx.x] = acc[1];
}

template <typename K> void run(const char* name, K kern, int grid, int threads, int lds, const unsigned char* src, size_t span, unsigned* out, double bytes_per_iter_block)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 2000;
    kern<<<grid, threads, lds>>>(src, span, 10, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    kern<<<grid, threads, lds>>>(src, span, iters, out);
    hipEventRecord(b);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, a, b);
    double tot = bytes_per_iter_block * iters * grid;
    printf("%-44s grid %4d: %8.1f GB/s total  %6.1f GB/s per CU\n", name, grid, tot / ms / 1e6, tot / ms / 1e6 / 256.0);
}

int main()
{
    unsigned char* src; unsigned* out;
    const size_t span = 16u << 20;     // 16 MB: L2 + MALL resident
    hipMalloc(&src, span + (1 << 20)); hipMemset(src, 1, span + (1 << 20)); hipMalloc(&out, 4096 * 4);
    hipFuncSetAttribute((const void*)glds_kernel<4, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void*)glds_kernel<8, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    run("glds 4 waves x depth 8, 1 WG/CU", glds_kernel<4, 8>, 256, 256, 32768, src, span, out, 4 * 8 * 1024.0);
    run("glds 4 waves x depth 8, 2 WG/CU", glds_kernel<4, 8>, 512, 256, 32768, src, span, out, 4 * 8 * 1024.0);
    run("glds 8 waves x depth 8, 1 WG/CU", glds_kernel<8, 8>, 256, 512, 65536, src, span, out, 8 * 8 * 1024.0);
    run("glds 8 waves x depth 8, 2 WG/CU", glds_kernel<8, 8>, 512, 512, 65536, src, span, out, 8 * 8 * 1024.0);
    run("vload 4 waves x depth 8, 1 WG/CU", vload_kernel<4, 8>, 256, 256, 0, src, span, out, 4 * 8 * 1024.0);
    run("vload 4 waves x depth 8, 2 WG/CU", vload_kernel<4, 8>, 512, 256, 0, src, span, out, 4 * 8 * 1024.0);
    run("vload 8 waves x depth 8, 2 WG/CU", vload_kernel<8, 8>, 512, 512, 0, src, span, out, 8 * 8 * 1024.0);
    run("vload 8 waves x depth 8, 4 WG/CU", vload_kernel<8, 8>, 1024, 512, 0, src, span, out, 8 * 8 * 1024.0);
    return 0;
}
