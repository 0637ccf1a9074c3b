// bandwidth sweep over streaming-kernel variants, with a result check against the first variant
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I smallk_amd/csrc tools/mb/mb_bp_sweep.hip -o tools/mb/mb_bp_sweep
//   mb_bp_sweep k len ncols storage(0 f32 | 1 bf16) variant [variant ...]
#include "../../smallk_amd/csrc/bigprod.hip"
#include "../../smallk_amd/csrc/kernels.hip"
#include <cstdio>
#include <vector>
#include <cmath>
using namespace smk;
#define STEP(name) do { hipError_t e_ = hipDeviceSynchronize(); if (getenv("MB_TRACE") || e_ != hipSuccess) { fprintf(stderr, "[%s] %s\n", name, hipGetErrorString(e_)); fflush(stderr); } } while (0)
std::string g_err_;
void smk::set_error(const std::string& m) { g_err_ = m; }
// kernels.hip refers to the wide gather product of wide.hip, which this tool does not link
int smk::launch_spmm_gather_wide(const i64*, const unsigned*, const double*, i64, const double*, int, double*, int, hipStream_t) { return -100; }

int main(int argc, char** argv)
{
    if (argc < 6) { printf("usage: k len ncols storage variant...\n"); return 1; }
    const int k = atoi(argv[1]);
    const i64 len = atoll(argv[2]), ncols = atoll(argv[3]);
    const int storage = atoi(argv[4]) == 0 ? STORE_F32 : STORE_BF16;
    const int es = storage == STORE_BF16 ? 2 : 4, nsplit = getenv("MB_NSPLIT") ? atoi(getenv("MB_NSPLIT")) : 3;
    hipStream_t st; hipStreamCreate(&st);
    const i64 ld = round_up(len, ROW_PAD), cp = round_up(ncols, COL_PAD);
    void* B; hipMalloc(&B, (size_t)ld * cp * es);
    launch_fill_uniform(B, storage, ld, len, ncols, ld, cp, 0, 0, len, 42, storage == STORE_BF16 ? 1 : 0, st);
    STEP("fill");
    if (getenv("MB_ZERO")) hipMemset(B, 0, (size_t)ld * cp * es);      // clock / power experiment: no data toggling
    const int reps = getenv("MB_REPS") ? atoi(getenv("MB_REPS")) : 5;
    const int KP = kp_of(k);
    double* X; hipMalloc(&X, (size_t)KP * len * 8);
    {   // X: uniform values in the live rows
        std::vector<double> hx((size_t)KP * len, 0.0);
        for (i64 c = 0; c < len; ++c) for (int r = 0; r < k; ++r) hx[c * KP + r] = (double)((c * 131 + r * 17) % 1000) / 1000.0 + 1e-7 * r;
        hipMemcpy(X, hx.data(), hx.size() * 8, hipMemcpyHostToDevice);
    }
    void* Xp; hipMalloc(&Xp, packed_bytes(storage, k, len, nsplit));
    double *xs = nullptr, *os = nullptr;
    const float ascale = 16384.f;                 // fill_uniform: max |B| < 1
    if (nsplit == NSPLIT_F16X2) {                 // row scales from the Gram diagonal, as the solver does
        double *G, *scr;
        hipMalloc(&G, (size_t)KP * KP * 8); hipMalloc(&scr, gram_scratch_elems(k, 1024) * 8);
        hipMalloc(&xs, 128 * 8); hipMalloc(&os, 128 * 8);
        launch_gram(X, k, len, G, scr, 1024, st, xs, os, (double)ascale);
    }
    STEP("gram");
    if (launch_pack(X, k, len, storage, nsplit, Xp, st, xs)) { printf("pack failed: %s\n", g_err_.c_str()); return 1; }
    STEP("pack");
    // exact products for the sampled columns (fp32 storage only)
    const i64 nsamp = 64;
    std::vector<double> exact;
    if (storage == STORE_F32 && !getenv("MB_ZERO")) {
        std::vector<double> hx((size_t)KP * len);
        hipMemcpy(hx.data(), X, hx.size() * 8, hipMemcpyDeviceToHost);
        std::vector<float> colv((size_t)len);
        exact.assign((size_t)nsamp * kt_of(k) * 32, 0.0);
        for (i64 j = 0; j < 8; ++j) {
            const i64 col = (j * 7919) % ncols;
            hipMemcpy(colv.data(), (const float*)B + col * ld, (size_t)len * 4, hipMemcpyDeviceToHost);
            for (int r = 0; r < k; ++r) {
                double acc = 0;          // fp64: 2^-53 sqrt(len) is far below every form measured here
                for (i64 c = 0; c < len; ++c) acc += hx[c * KP + r] * (double)colv[c];
                exact[j * kt_of(k) * 32 + r] = (double)acc;
            }
        }
    }
    std::vector<double> ref;
    for (int a = 5; a < argc; ++a) {
        setenv("SMK_BP_VARIANT", argv[a], 1);
        BigProdPlan pl = plan_bigprod(storage, k, len, ncols, nsplit, 256);
        pl.oscale = os; pl.ascale = ascale;
        double* P; hipMalloc(&P, pl.p_elems * 8);
        hipMemset(P, 0, pl.p_elems * 8);
        if (launch_bigprod(pl, B, ld, Xp, P, st)) { printf("variant %s: launch failed: %s\n", argv[a], g_err_.c_str()); hipFree(P); continue; }
        STEP("first launch");
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, st);
        for (int r = 0; r < reps; ++r) launch_bigprod(pl, B, ld, Xp, P, st);
        hipEventRecord(e1, st);
        hipStreamSynchronize(st);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
        // summed result for a sample of columns
        std::vector<double> got((size_t)nsamp * pl.kt * 32, 0.0), slab((size_t)pl.kt * 32);
        for (i64 j = 0; j < nsamp; ++j) {
            const i64 col = (j * 7919) % ncols;
            for (int s = 0; s < pl.S; ++s) {
                hipMemcpy(slab.data(), P + ((i64)s * pl.ncols_pad + col) * pl.kt * 32, slab.size() * 8, hipMemcpyDeviceToHost);
                for (size_t e = 0; e < slab.size(); ++e) got[j * slab.size() + e] += slab[e];
            }
        }
        double maxrel = 0.0;
        if (ref.empty()) ref = got;
        else for (size_t e = 0; e < got.size(); ++e) { const double d = fabs(got[e] - ref[e]) / (fabs(ref[e]) + 1e-300); if (ref[e] != 0.0 && d > maxrel) maxrel = d; }
        double maxex = 0.0;
        for (size_t e = 0; e < exact.size(); ++e)
            if (exact[e] != 0.0) maxex = std::max(maxex, fabs(got[e] - exact[e]) / fabs(exact[e]));
        printf("k=%d len=%ld ncols=%ld %s nsplit %d variant %d (requested %s) S=%d: %.3f ms  %.0f GB/s  max rel diff vs first %.2e  vs exact %.2e\n", k, (long)len, (long)ncols,
               storage == STORE_BF16 ? "bf16" : "f32", nsplit, pl.variant, argv[a], pl.S, ms, (double)len * ncols * es / ms / 1e6, maxrel, maxex);
        fflush(stdout);
        hipFree(P);
    }
    return 0;
}
