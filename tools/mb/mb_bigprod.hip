// phase profile of the streaming kernel (s_memtime stamps per stage): wait / barrier / issue / compute
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSMK_BP_PROFILE -I smallk_amd/csrc tools/mb/mb_bigprod.hip -o tools/mb/mb_bigprod
#include "../../smallk_amd/csrc/kernels.hip"
#include "../../smallk_amd/csrc/bigprod.hip"
#include <cstdio>
#include <vector>
using namespace smk;
std::string g_err_;
void smk::set_error(const std::string& m) { g_err_ = m; }

int main(int argc, char** argv)
{
    const int k = argc > 1 ? atoi(argv[1]) : 64;
    const i64 len = argc > 2 ? atoll(argv[2]) : 65536, ncols = argc > 3 ? atoll(argv[3]) : 32768;
    const int nsplit = argc > 4 ? atoi(argv[4]) : 3;
    const int storage = (argc > 5 && atoi(argv[5]) == 0) ? STORE_F32 : STORE_BF16;
    const int es = storage == STORE_BF16 ? 2 : 4;
    hipStream_t st; hipStreamCreate(&st);
    const i64 ld = round_up(len, ROW_PAD), cp = round_up(ncols, COL_PAD);
    void* B; hipMalloc(&B, (size_t)ld * cp * es);
    launch_fill_uniform(B, storage, ld, len, ncols, ld, cp, 0, 0, len, 42, storage == STORE_BF16 ? 1 : 0, st);
    double* X; hipMalloc(&X, (size_t)kp_of(k) * len * 8);
    launch_fill_uniform(B, storage, ld, len, ncols, ld, cp, 0, 0, len, 42, storage == STORE_BF16 ? 1 : 0, st);
    hipMemset(X, 0, (size_t)kp_of(k) * len * 8);
    void* Xp; hipMalloc(&Xp, packed_bytes(storage, k, len, nsplit));
    launch_pack(X, k, len, storage, nsplit, Xp, st);
    BigProdPlan pl = plan_bigprod(storage, k, len, ncols, nsplit, 256);
    double* P; hipMalloc(&P, pl.p_elems * 8);
    const size_t nslots = (size_t)8192 * 8 * 4;
    unsigned long long* prof; hipMalloc(&prof, nslots * 8); hipMemset(prof, 0, nslots * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(g_bp_prof), &prof, sizeof(prof));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch_bigprod(pl, B, ld, Xp, P, st);
    hipStreamSynchronize(st);
    hipEventRecord(a, st);
    launch_bigprod(pl, B, ld, Xp, P, st);
    hipEventRecord(b, st);
    hipStreamSynchronize(st);
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> h(nslots);
    hipMemcpy(h.data(), prof, nslots * 8, hipMemcpyDeviceToHost);
    double s[4] = {0, 0, 0, 0}; size_t n = 0;
    for (size_t i = 0; i < nslots / 4; ++i) {
        if (h[i * 4 + 3] == 0) continue;
        for (int j = 0; j < 4; ++j) s[j] += (double)h[i * 4 + j];
        ++n;
    }
    const double tot = s[0] + s[1] + s[2] + s[3];
    printf("k=%d len=%ld ncols=%ld nsplit=%d variant=%d S=%d: %.3f ms  %.0f GB/s | per-wave cycles (avg over %zu waves): wait %.0f (%.0f%%) barrier %.0f (%.0f%%) issue %.0f (%.0f%%) compute %.0f (%.0f%%) | per stage: %.0f cycles\n",
           k, (long)len, (long)ncols, nsplit, pl.variant, pl.S, ms, (double)len * ncols * es / ms / 1e6, n, s[0] / n, 100 * s[0] / tot,
           s[1] / n, 100 * s[1] / tot, s[2] / n, 100 * s[2] / tot, s[3] / n, 100 * s[3] / tot, tot / n / (double)pl.nst);
    return 0;
}
