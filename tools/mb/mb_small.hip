// micro-benchmark of the small per-iteration kernels (latency floor study).  Build:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I smallk_amd/csrc tools/mb/mb_small.hip -o tools/mb/mb_small
#include "../../smallk_amd/csrc/kernels.hip"
#include <cstdio>
#include <vector>
using namespace smk;

__global__ void empty_kernel(int) {}
__global__ __launch_bounds__(1024) void empty1024(int) {}

template <typename F> float time_launches(F f, int n, hipStream_t st)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 20; ++i) f(i);
    hipStreamSynchronize(st);
    hipEventRecord(a, st);
    for (int i = 0; i < n; ++i) f(i);
    hipEventRecord(b, st);
    hipStreamSynchronize(st);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1000.f / n;
}

std::string g_err_;
void smk::set_error(const std::string& m) { g_err_ = m; }

int main()
{
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    const int k = 32, KP = 32; const i64 M = 65536, N = 16384;
    double *Wt, *H, *G, *scratch, *P, *gscr;
    hipMalloc(&Wt, KP * M * 8); hipMalloc(&H, KP * N * 8); hipMalloc(&G, KP * KP * 8);
    hipMalloc(&scratch, 2 * k * 1024 * 8 + 1024); hipMalloc(&P, M * 32 * 8); hipMalloc(&gscr, 256 * KP * KP * 8);
    std::vector<double> h(KP * M, 0.01), g(KP * KP, 0.5);
    for (int i = 0; i < KP; ++i) g[i * KP + i] = 20.0;
    hipMemcpy(Wt, h.data(), KP * M * 8, hipMemcpyHostToDevice);
    hipMemcpy(H, h.data(), KP * N * 8, hipMemcpyHostToDevice);
    hipMemcpy(P, h.data(), KP * M * 8, hipMemcpyHostToDevice);
    hipMemcpy(G, g.data(), KP * KP * 8, hipMemcpyHostToDevice);
    hipMemset(scratch, 0, 2 * k * 1024 * 8);
    int* ff; hipMalloc(&ff, 4); hipMemset(ff, 0x7f, 4);
    PartialView R{P, 1, 0, 32, 1};

    printf("empty<<<1,64>>>          %.2f us/launch\n", time_launches([&](int) { empty_kernel<<<1, 64, 0, st>>>(0); }, 2000, st));
    printf("empty<<<512,1024>>>      %.2f us/launch\n", time_launches([&](int) { empty1024<<<512, 1024, 0, st>>>(0); }, 2000, st));
    printf("empty<<<2048,256>>>      %.2f us/launch\n", time_launches([&](int) { empty_kernel<<<2048, 256, 0, st>>>(0); }, 2000, st));
    const int nblk = 512;
    double* ss = scratch; double* nz = scratch + k * nblk;
    printf("hals_w_col c=0           %.2f us/launch\n", time_launches([&](int) { hals_w_col_kernel<32><<<nblk, 1024, 0, st>>>(Wt, k, M, R, G, 0, nblk, ss, nz); }, 1000, st));
    printf("hals_w_col c=5           %.2f us/launch\n", time_launches([&](int) { hals_w_col_kernel<32><<<nblk, 1024, 0, st>>>(Wt, k, M, R, G, 5, nblk, ss, nz); }, 1000, st));
    printf("hals_w full update (33)  %.2f us/update\n", time_launches([&](int) { launch_hals_w_update(Wt, k, M, R, G, scratch, 256, ff, st); }, 100, st));
    printf("hals_sweep n=16384       %.2f us\n", time_launches([&](int) { launch_hals_sweep(H, k, N, R, G, st); }, 500, st));
    printf("gram m=65536             %.2f us\n", time_launches([&](int) { launch_gram(Wt, k, M, G, gscr, 256, st); }, 500, st));
    printf("gram n=16384             %.2f us\n", time_launches([&](int) { launch_gram(H, k, N, G, gscr, 256, st); }, 500, st));
    void* pk; hipMalloc(&pk, packed_bytes(STORE_BF16, k, M, 3));
    printf("pack m=65536             %.2f us\n", time_launches([&](int) { launch_pack(Wt, k, M, STORE_BF16, 3, pk, st); }, 500, st));
    printf("grad_pg m=65536          %.2f us\n", time_launches([&](int) { launch_grad_pg(Wt, k, M, R, G, nullptr, gscr, gscr + 4096, 0, st); }, 500, st));
    return 0;
}
