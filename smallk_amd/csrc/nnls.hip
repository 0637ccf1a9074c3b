// smallk_amd/csrc/nnls.hip -- NnlsBlockpivot on the device (nnls.hpp:144-244, src/nnls.cpp:18-74,
// nmf_solver_bpp.hpp:146-219, normal_eq.hpp:27-54): the per-column masked Gauss-Jordan kernel and the
// inverse-based kernel for k in (32, 64].
#include "devutil.h"
#include "gram_inverse.h"
#include "nnls_masked.h"

namespace smk {

// --------------------------------------------------------------------------
// Diagnostics (SMK_NNLS_STATS=1; tools/nnls_sets.py, smk_debug_nnls_stats): histograms of the work block pivoting does, kept in
// 256 device counters -- [0..15] exchanges per column (15 = 15 or more), [16..80] size t of the FIRST compact solve of a column
// (16 + t), [96..160] size of every later solve, [176] solves in the complement form, [177] in the direct form, [178] columns,
// [179] solves with every variable passive, [180] with none.  nullptr (the default) costs one uniform branch per column.
// --------------------------------------------------------------------------
unsigned long long* nnls_stats_ptr()
{
    static unsigned long long* buf = [] {
        const char* e = getenv("SMK_NNLS_STATS");
        unsigned long long* p = nullptr;
        if (e && atoi(e) != 0 && hipMalloc((void**)&p, 256 * sizeof(unsigned long long)) == hipSuccess) (void)hipMemset(p, 0, 256 * sizeof(unsigned long long));
        else p = nullptr;
        return p;
    }();
    return buf;
}
int nnls_stats_read(unsigned long long* out256, int reset)
{
    unsigned long long* p = nnls_stats_ptr();
    if (!p) return -1;
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    if (out256 && hipMemcpy(out256, p, 256 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return -2;
    if (reset) (void)hipMemset(p, 0, 256 * sizeof(unsigned long long));
    return 0;
}

// the masked elimination by itself (device code: nnls_masked.h)
// KP = 64: left alone the compiler takes 256 VGPRs + 40 AGPRs (one wave per SIMD) and every
// readlane -> FMA dependency is exposed; capping at 168 registers (3 waves per SIMD, 516 B of
// scratch per lane) is 1.45x faster on a 262144 x 8192 k = 64 BPP iteration.  No gain at KP <= 32.
template <int KP>
__global__ __launch_bounds__(256, (KP == 64 ? 3 : 1)) void nnls_bpp_kernel(double* __restrict__ X, double* __restrict__ Y, int k, i64 N,
                                                       PartialView R, const double* __restrict__ G,
                                                       int* __restrict__ fail_flag, int iter_tag, i64 col_begin,
                                                       const int* __restrict__ skip_if, double* __restrict__ Gp, NnlsPack pk,
                                                       unsigned long long* __restrict__ stats, NnlsRiders rd = NnlsRiders())
{
    if (skip_if && *skip_if != 0) return;           // the inverse-based kernel took this launch
    nnls_bpp_body<KP>(X, Y, k, N, R, G, fail_flag, iter_tag, col_begin, Gp, pk, stats, rd);
}

// --------------------------------------------------------------------------
// k in (32, 64]: block principal pivoting through the INVERSE of the Gram matrix.
//
// On noise-like data every column has its own passive set F (|F| ~ 45-58 of 64, 8192 distinct sets in
// 8192 columns) and needs ~2 solves, so neither the reference's "all passive" shortcut nor its grouping
// by identical sets (nmf_solver_bpp.hpp:29-142) ever fires; a factorisation per column and pivot of the
// |F| x |F| system costs O(k^3).  With Ginv = G^-1 (one 64 x 64 inversion per launch) and v = Ginv r:
//     Z = complement of F,  S = Ginv[Z,Z]:   y_Z = -S^-1 v_Z,   x = v + Ginv[:,Z] y_Z   (x_Z = 0, y_F = 0)
// (block elimination on G x = r + y), i.e. one |Z| x |Z| solve plus |Z| rank-one terms -- and both x and
// the dual y come out of it.  When |F| < |Z| the direct form G[F,F] x_F = r_F, y = G[:,F] x_F - r is
// cheaper; both are the same "compact solve + accumulate" on (M, T, s) = (Ginv, Z, -v) or (G, F, r) with
// t = min(|F|, |Z|) <= 32 rows in the first t lanes of the wave.  One wave per column; the set is
// wave-uniform, so index lists, loop bounds and pivot broadcasts are scalar.
// The inverse is taken only when every Gauss-Jordan pivot of G is positive and not tiny (gram_inverse_kernel
// sets status = 1); otherwise nnls_bpp_kernel<64> above runs and reproduces the reference's non-SPD failure.
// --------------------------------------------------------------------------
// Ginv = G^-1 (k x k live, 64 x 64 storage, pads zero) by in-place Gauss-Jordan, one workgroup of 256 threads:
// thread (r, cq) keeps the 16 entries a[r][16 cq .. 16 cq + 15] in registers; per pivot only the pivot row and
// column travel through LDS (double buffered: one barrier per step).  (KP = 64 runs gram_inverse64_kernel below: in this
// form the compiler indexes a[] at run time for the pivot column and keeps it in scratch memory, 113 us at k = 64.)
// status = 1 when every pivot p_j satisfies p_j > 1e-9 * G[j][j] (SPD and usable), else 0.
template <int KP>
__global__ __launch_bounds__(256) void gram_inverse_kernel(const double* __restrict__ G, int k,
                                                           double* __restrict__ Ginv, int* __restrict__ status)
{
    static_assert(KP == 64 || KP == 128, "256 threads x (KP * KP / 256) entries");
    constexpr int CQ = 256 / KP;                 // threads per matrix row
    constexpr int EPT = KP / CQ;                 // entries per thread: a[r][EPT cq .. EPT cq + EPT - 1]
    __shared__ __attribute__((aligned(16))) double rowj[2][KP];
    __shared__ double colj[2][KP];
    __shared__ double diag0[KP];
    __shared__ int bad;
    const int tid = threadIdx.x;
    const int r = tid / CQ, cq = tid % CQ;
    if (tid == 0) bad = 0;
    if (tid < KP) diag0[tid] = (tid < k) ? G[tid * KP + tid] : 1.0;
    double a[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int c = cq * EPT + e;
        a[e] = (r < k && c < k) ? G[c * KP + r] : ((r == c) ? 1.0 : 0.0);
    }
    auto publish = [&](int j) {                 // row j and column j of the current matrix -> LDS buffer j & 1
        const int buf = j & 1, jq = j / EPT, je = j % EPT;
        if (r == j) {
#pragma unroll
            for (int e = 0; e < EPT; ++e) rowj[buf][cq * EPT + e] = a[e];
        }
        if (cq == jq) {
            double v = a[0];
#pragma unroll
            for (int e = 1; e < EPT; ++e) v = (e == je) ? a[e] : v;
            colj[buf][r] = v;
        }
    };
    publish(0);
    __syncthreads();
    for (int j = 0; j < k; ++j) {
        const int buf = j & 1;
        const double piv = rowj[buf][j];
        if (tid == 0 && !(piv > 1.0e-9 * diag0[j])) bad = 1;
        const double ip = 1.0 / piv;
        const double f = colj[buf][r] * ip;
#pragma unroll
        for (int e = 0; e < EPT; e += 2) {
            const f64x2_t rr = *(const f64x2_t*)&rowj[buf][cq * EPT + e];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int c = cq * EPT + e + u;
                double val;
                if (r == j) val = (c == j) ? ip : rr[u] * ip;
                else val = (c == j) ? -f : __builtin_fma(-f, rr[u], a[e + u]);
                a[e + u] = val;
            }
        }
        if (j + 1 < k) publish(j + 1);
        __syncthreads();
    }
    // symmetrise: write the matrix out, then average each entry with its transposed partner
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int c = cq * EPT + e;
        Ginv[c * KP + r] = (r < k && c < k) ? a[e] : 0.0;
    }
    __syncthreads();
    __threadfence_block();
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int c = cq * EPT + e;
        const double up = Ginv[c * KP + r], lo = Ginv[r * KP + c];
        a[e] = 0.5 * (up + lo);
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < EPT; ++e) Ginv[(cq * EPT + e) * KP + r] = a[e];
    if (tid == 0) *status = bad ? 0 : 1;
}

// (the elimination itself: gram_inverse.h -- it also rides in the sparse product launch)
template <int KP>
__global__ __launch_bounds__(256) void gram_inverse64_kernel(const double* __restrict__ G, int k, double* __restrict__ Ginv,
                                                             int* __restrict__ status)
{
    __shared__ __attribute__((aligned(16))) double lds[GRAM_INVERSE_LDS(KP)];
    gram_inverse64_body<KP>(G, k, Ginv, status, lds);
}

template <int KP, int NT, int WPS, bool FINE = true>
__global__ __launch_bounds__(NT, WPS) void nnls_bpp_inv_kernel(double* __restrict__ X, double* __restrict__ Y, int k, i64 N,
                                                          PartialView R, const double* __restrict__ G,
                                                          const double* __restrict__ Ginv,
                                                          const int* __restrict__ status,
                                                          int* __restrict__ fail_flag, int iter_tag, i64 col_begin,
                                                          unsigned long long* __restrict__ stats,
                                                          const unsigned* __restrict__ worklist)
{
    static_assert(KP == 64 || KP == 32, "one wave per column, one lane per component");
    constexpr int NW = NT / 64;
    if (*status == 0) return;
    // worklist != nullptr: only the columns nnls_bpp_g16_kernel handed over (worklist[0] of them, col_begin + worklist[1 + i]);
    // usually few or none, so the count is read before the matrices are copied
    const i64 nwork = worklist ? (i64)worklist[0] : 0;
    if (worklist && (i64)blockIdx.x * NW >= nwork) return;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* gs = lds;                               // gs[c*KP + i]  = G[i][c]
    double* gis = lds + KP * KP;                    // gis[c*KP + i] = Ginv[i][c]
    for (int t = threadIdx.x; t < KP * KP; t += NT) { gs[t] = G[t]; gis[t] = Ginv[t]; }
    const int lane = threadIdx.x & 63;
    // KP = 32 (k in (16, 32], round 4): the upper half of the wave has no component; it runs along masked (comp_ok, the set
    // masks) and reads the matrices at ln = lane mod KP so that it stays inside them
    const int ln = lane & (KP - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double* sv = lds + 2 * KP * KP + wave * 128;               // wave-private: 64 doubles of values
    int* sidx = (int*)(sv + 64);                               //               64 ints of indices
    __syncthreads();

    const unsigned long long kmask = (k >= 64) ? ~0ull : ((1ull << k) - 1ull);
    const bool comp_ok = lane < k;
    const int max_iter = 5 * k;
    int failed_any = 0;

    // A wave works through its columns one after the other, and a column is load -> ~0.3 us of arithmetic -> store: without
    // the loads of column c + 1 in flight during the arithmetic of column c every column costs a round trip to memory on top
    // (1 M columns at k = 32, 256 columns per wave: 1.19 ms per launch, 4.6 us per column and wave, of which the arithmetic
    // is a tenth; profiles/r05_s_1m_kernel_stats.md).  The right-hand side and the start of the NEXT column are requested
    // before the current one is touched.
    // positions idx walk the columns [col_begin, N) or the work list
    const i64 col_stride = (i64)gridDim.x * NW;
    const i64 nidx = worklist ? nwork : (N - col_begin);
    auto col_of = [&](i64 idx) -> i64 { return col_begin + (worklist ? (i64)worklist[1 + idx] : idx); };
    i64 idx = (i64)blockIdx.x * NW + wave;
    RhsPending rq;
    rq.t[0] = rq.t[1] = rq.t[2] = rq.t[3] = rq.tail = 0.0;
    double x_next = 0.0;
    i64 col_nx = idx < nidx ? col_of(idx) : 0;
    if (idx < nidx && comp_ok) {
        rhs_issue(R, col_nx, lane, rq);
        x_next = X[col_nx * KP + lane];
    }
    for (; idx < nidx; idx += col_stride) {
        const i64 col = col_nx;
        double rhs = comp_ok ? rhs_finish(R, rq) : 0.0, x = x_next, y = 0.0;
        if (idx + col_stride < nidx) {
            col_nx = col_of(idx + col_stride);
            if (comp_ok) {
                rhs_issue(R, col_nx, lane, rq);
                x_next = X[col_nx * KP + lane];
            }
        }
        unsigned long long F = __ballot(comp_ok && x > 0.0) & kmask;       // passive_set = (X > 0), nnls.hpp:157

        // v = Ginv r, the unconstrained solution: needed by the complement form only (and when every variable is passive).
        // A column whose passive sets stay small -- sparse factors: a document in two or three topics, a node in one community
        // -- never takes that form, and the KP-step product was a third of its time (s_1m: 1.19 ms per launch of 1 M columns,
        // profiles/r05_s_1m_kernel_stats.md), so it is computed when the first solve of the column asks for it.
        double v = 0.0;
        bool have_v = false;
        auto need_v = [&]() {
            if (have_v) return;
            have_v = true;
            sv[lane] = rhs;
            double v0 = 0.0, v1 = 0.0;
#pragma unroll 8
            for (int c = 0; c < KP; c += 2) {
                const f64x2_t rr = *(const f64x2_t*)(sv + c);              // broadcast read
                v0 = __builtin_fma(gis[c * KP + ln], rr[0], v0);
                v1 = __builtin_fma(gis[(c + 1) * KP + ln], rr[1], v1);
            }
            v = v0 + v1;
        };

        int failed = 0;
        int nsolve = 0;                                                     // diagnostics only
        // compact solve of M[T,T] u = s_T on the first TB lanes (rows >= t are identity rows), then
        // out = base + M[:,T] u.  TB is a compile-time bound: no branch inside the elimination.
        auto compact = [&](auto tb_tag, const double* M, int t, int tl, double sc, double base, double& u_out) -> double {
            constexpr int TB = decltype(tb_tag)::value;
            const bool live = lane < t;
            double a[TB];
#pragma unroll
            for (int b = 0; b < TB; ++b) {
                const int tb = __builtin_amdgcn_readlane(tl, b);           // lanes >= t carry tl = 0: harmless
                const double mv = M[tb * KP + tl];
                a[b] = (live && b < t) ? mv : ((b == lane) ? 1.0 : 0.0);
            }
            double d = 1.0;
#pragma unroll
            for (int j = 0; j < TB; ++j) {
                const double piv = readlane_f64(a[j], j);
                if (!(piv > 0.0)) failed = 1;
                const double ip = fast_rcp(piv);
                if (lane == j) d = a[j];
                const double f = (lane == j) ? 0.0 : a[j] * ip;
#pragma unroll
                for (int c = j + 1; c < TB; ++c) a[c] = __builtin_fma(-f, readlane_f64(a[c], j), a[c]);
                sc = __builtin_fma(-f, readlane_f64(sc, j), sc);
            }
            const double u = live ? sc * fast_rcp(d) : 0.0;
            double out = base;
#pragma unroll
            for (int b = 0; b < TB; ++b) {
                const int tb = __builtin_amdgcn_readlane(tl, b);
                out = __builtin_fma(M[tb * KP + ln], readlane_f64(u, b), out);       // u = 0 beyond t
            }
            u_out = u;
            return out;
        };
        // one block-pivot solve for the passive set Fs: leaves x (zero outside Fs) and y (zero inside Fs)
        auto solve = [&](unsigned long long Fs) {
            const unsigned long long Zs = ~Fs & kmask;
            const int p = __popcll(Fs), q = __popcll(Zs);
            const bool inF = (Fs >> lane) & 1ull;
            if (stats && lane == 0) {
                const int tt = q <= p ? q : p;
                nnls_stat(stats, (nsolve == 0 ? 16 : 96) + tt);
                nnls_stat(stats, q == 0 ? 179 : p == 0 ? 180 : q <= p ? 176 : 177);
            }
            ++nsolve;
            if (q == 0) { need_v(); x = v; y = 0.0; return; }
            if (p == 0) { x = 0.0; y = comp_ok ? -rhs : 0.0; return; }
            const bool comp = q <= p;                                       // complement form on Ginv
            if (comp) need_v();
            const unsigned long long T = comp ? Zs : Fs;
            const int t = comp ? q : p;
            const double* M = comp ? gis : gs;
            const bool inT = (T >> lane) & 1ull;
            // compact index list: lane l < t gets the l-th member of T
            const int rank = __popcll(T & ((1ull << lane) - 1ull));
            if (inT) sidx[rank] = lane;
            sv[lane] = comp ? -v : rhs;
            const int tl = (lane < t) ? sidx[lane] : 0;
            const double sc = (lane < t) ? sv[tl] : 0.0;
            const double base = comp ? v : -rhs;
            double u = 0.0, out;
            // the elimination costs ~TB^2 / 2 broadcast + FMA pairs: bounds in steps of 4 (FINE) waste a quarter less
            // than steps of 8 on the |Z| = 6 .. 20 of noise-like data
            if (FINE && t <= 4) out = compact(std::integral_constant<int, 4>{}, M, t, tl, sc, base, u);
            else if (t <= 8) out = compact(std::integral_constant<int, 8>{}, M, t, tl, sc, base, u);
            else if (FINE && t <= 12) out = compact(std::integral_constant<int, 12>{}, M, t, tl, sc, base, u);
            else if (t <= 16) out = compact(std::integral_constant<int, 16>{}, M, t, tl, sc, base, u);
            else if (FINE && t <= 20) out = compact(std::integral_constant<int, 20>{}, M, t, tl, sc, base, u);
            else if (t <= 24) out = compact(std::integral_constant<int, 24>{}, M, t, tl, sc, base, u);
            else if (FINE && t <= 28) out = compact(std::integral_constant<int, 28>{}, M, t, tl, sc, base, u);
            else out = compact(std::integral_constant<int, 32>{}, M, t, tl, sc, base, u);
            // u back to component positions
            if (lane < t) sv[tl] = u;
            const double ut = inT ? sv[lane] : 0.0;
            if (comp) { x = inF ? out : 0.0; y = ut; }
            else      { x = ut; y = (comp_ok && !inF) ? out : 0.0; }
        };

        solve(F);
        bool passive = (F >> lane) & 1ull;
        unsigned long long nonopt = __ballot(comp_ok && !passive && (y < 0.0));
        unsigned long long infeas = __ballot(comp_ok && passive && (x < 0.0));
        int ng = __popcll(nonopt) + __popcll(infeas);
        int Pc = 3, Ninf = k + 1;                    // PBAR = 3, nnls.hpp:152,170
        int iter = 0;
        while (ng > 0) {                             // uniform
            if (iter >= max_iter) { failed = 1; break; }
            // UpdatePassiveSet, src/nnls.cpp:18-74
            if (ng < Ninf) { Pc = 3; Ninf = ng; F = (F | nonopt) & ~infeas; }
            else if (Pc >= 1) { Pc -= 1; F = (F | nonopt) & ~infeas; }
            else {
                const int r1 = nonopt ? (63 - __clzll(nonopt)) : 0;
                const int r2 = infeas ? (63 - __clzll(infeas)) : 0;
                F ^= (1ull << (r1 > r2 ? r1 : r2));
            }
            F &= kmask;
            passive = (F >> lane) & 1ull;
            solve(F);
            if (fabs(x) < 1.0e-12) x = 0.0;          // ZeroizeSmallValues, nnls.hpp:213,224
            if (fabs(y) < 1.0e-12) y = 0.0;          // :225
            nonopt = __ballot(comp_ok && !passive && (y < 0.0));
            infeas = __ballot(comp_ok && passive && (x < 0.0));
            ng = __popcll(nonopt) + __popcll(infeas);
            ++iter;
        }
        if (fabs(x) < 1.0e-12) x = 0.0;              // columns that never pivot are zeroized too (nnls_bpp_kernel's note)
        if (fabs(y) < 1.0e-12) y = 0.0;
        if (comp_ok) {
            X[col * KP + lane] = x;
            if (Y) Y[col * KP + lane] = y;
        }
        if (stats && lane == 0) { nnls_stat(stats, iter < 15 ? iter : 15); nnls_stat(stats, 178); }
        failed_any |= failed;
    }
    if (failed_any && lane == 0) atomicMin(fail_flag, iter_tag);
}

// k in (64, 128]: the same algorithm with two components per lane (lane i owns i and 64 + i), passive sets as two
// 64-bit words, Ginv (128 KiB) in LDS and G read through the caches (it is needed only by the direct form, i.e. for
// solutions with more zeros than positives).  The compact dimension min(|F|, |Z|) is at most 64 = one row per lane.
// There is no masked Gauss-Jordan fallback at this width: a Gram matrix that is not (numerically) positive definite
// reports failure for the whole launch -- the reference fails only when a passive block it actually meets is not
// SPD (normal_eq.hpp:35-50), which for a rank-deficient factor is the usual case.
template <int NT>
__global__ __launch_bounds__(NT) void nnls_bpp_inv128_kernel(double* __restrict__ X, double* __restrict__ Y, int k, i64 N,
                                                             PartialView R, const double* __restrict__ G,
                                                             const double* __restrict__ Ginv,
                                                             const int* __restrict__ status,
                                                             int* __restrict__ fail_flag, int iter_tag, i64 col_begin)
{
    constexpr int KP = 128;
    constexpr int NW = NT / 64;
    if (*status == 0) {
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicMin(fail_flag, iter_tag);
        return;
    }
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* gis = lds;                              // gis[c*KP + i] = Ginv[i][c]
    for (int t = threadIdx.x; t < KP * KP; t += NT) gis[t] = Ginv[t];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double* sv = lds + KP * KP + wave * (KP + KP / 2);          // wave-private: 128 doubles of values
    int* sidx = (int*)(sv + KP);                                //               128 ints of indices
    __syncthreads();

    const int k1 = k - 64;                                       // live components in the upper word (k > 64 here)
    const unsigned long long m0 = ~0ull, m1 = (k1 >= 64) ? ~0ull : ((1ull << k1) - 1ull);
    const bool ok0 = true, ok1 = lane < k1;
    const unsigned long long below = (1ull << lane) - 1ull;
    const int max_iter = 5 * k;
    int failed_any = 0;

    for (i64 col = col_begin + (i64)blockIdx.x * NW + wave; col < N; col += (i64)gridDim.x * NW) {
        const double r0 = rhs_elem(R, col, lane), r1 = ok1 ? rhs_elem(R, col, 64 + lane) : 0.0;
        double x0 = X[col * KP + lane], x1 = ok1 ? X[col * KP + 64 + lane] : 0.0;
        double y0 = 0.0, y1 = 0.0;
        unsigned long long F0 = __ballot(ok0 && x0 > 0.0) & m0, F1 = __ballot(ok1 && x1 > 0.0) & m1;

        sv[lane] = r0;
        sv[64 + lane] = r1;
        double v0 = 0.0, v1 = 0.0;
#pragma unroll 4
        for (int c = 0; c < KP; c += 2) {
            const f64x2_t rr = *(const f64x2_t*)(sv + c);
            v0 = __builtin_fma(gis[c * KP + lane], rr[0], v0);
            v1 = __builtin_fma(gis[c * KP + 64 + lane], rr[0], v1);
            v0 = __builtin_fma(gis[(c + 1) * KP + lane], rr[1], v0);
            v1 = __builtin_fma(gis[(c + 1) * KP + 64 + lane], rr[1], v1);
        }

        int failed = 0;
        auto compact = [&](auto tb_tag, const double* M, int t, int tl, double sc, double b0, double b1, double& u_out,
                           double& o0, double& o1) {
            constexpr int TB = decltype(tb_tag)::value;
            const bool live = lane < t;
            double a[TB];
#pragma unroll
            for (int b = 0; b < TB; ++b) {
                const int tb = __builtin_amdgcn_readlane(tl, b);
                const double mv = M[tb * KP + tl];
                a[b] = (live && b < t) ? mv : ((b == lane) ? 1.0 : 0.0);
            }
            double d = 1.0;
#pragma unroll
            for (int j = 0; j < TB; ++j) {
                const double piv = readlane_f64(a[j], j);
                if (!(piv > 0.0)) failed = 1;
                const double ip = fast_rcp(piv);
                if (lane == j) d = a[j];
                const double f = (lane == j) ? 0.0 : a[j] * ip;
#pragma unroll
                for (int c = j + 1; c < TB; ++c) a[c] = __builtin_fma(-f, readlane_f64(a[c], j), a[c]);
                sc = __builtin_fma(-f, readlane_f64(sc, j), sc);
            }
            const double u = live ? sc * fast_rcp(d) : 0.0;
            o0 = b0;
            o1 = b1;
#pragma unroll
            for (int b = 0; b < TB; ++b) {
                const int tb = __builtin_amdgcn_readlane(tl, b);
                const double ub = readlane_f64(u, b);
                o0 = __builtin_fma(M[tb * KP + lane], ub, o0);
                o1 = __builtin_fma(M[tb * KP + 64 + lane], ub, o1);
            }
            u_out = u;
        };
        auto solve = [&](unsigned long long Fa, unsigned long long Fb) {
            const unsigned long long Za = ~Fa & m0, Zb = ~Fb & m1;
            const int p = __popcll(Fa) + __popcll(Fb), q = __popcll(Za) + __popcll(Zb);
            const bool inF0 = (Fa >> lane) & 1ull, inF1 = (Fb >> lane) & 1ull;
            if (q == 0) { x0 = v0; x1 = v1; y0 = y1 = 0.0; return; }
            if (p == 0) { x0 = x1 = 0.0; y0 = -r0; y1 = ok1 ? -r1 : 0.0; return; }
            const bool comp = q <= p;
            const unsigned long long Ta = comp ? Za : Fa, Tb = comp ? Zb : Fb;
            const int t = comp ? q : p;
            const double* M = comp ? gis : G;
            const bool inT0 = (Ta >> lane) & 1ull, inT1 = (Tb >> lane) & 1ull;
            const int rank0 = __popcll(Ta & below), rank1 = __popcll(Ta) + __popcll(Tb & below);
            if (inT0) sidx[rank0] = lane;
            if (inT1) sidx[rank1] = 64 + lane;
            sv[lane] = comp ? -v0 : r0;
            sv[64 + lane] = comp ? -v1 : r1;
            const int tl = (lane < t) ? sidx[lane] : 0;
            const double sc = (lane < t) ? sv[tl] : 0.0;
            double u = 0.0, o0 = 0.0, o1 = 0.0;
            const double b0 = comp ? v0 : -r0, b1 = comp ? v1 : -r1;
            if (t <= 8) compact(std::integral_constant<int, 8>{}, M, t, tl, sc, b0, b1, u, o0, o1);
            else if (t <= 16) compact(std::integral_constant<int, 16>{}, M, t, tl, sc, b0, b1, u, o0, o1);
            else if (t <= 32) compact(std::integral_constant<int, 32>{}, M, t, tl, sc, b0, b1, u, o0, o1);
            else compact(std::integral_constant<int, 64>{}, M, t, tl, sc, b0, b1, u, o0, o1);
            if (lane < t) sv[tl] = u;
            const double ut0 = inT0 ? sv[lane] : 0.0, ut1 = inT1 ? sv[64 + lane] : 0.0;
            if (comp) { x0 = inF0 ? o0 : 0.0; x1 = inF1 ? o1 : 0.0; y0 = ut0; y1 = ut1; }
            else      { x0 = ut0; x1 = ut1; y0 = !inF0 ? o0 : 0.0; y1 = (ok1 && !inF1) ? o1 : 0.0; }
        };

        solve(F0, F1);
        auto sets = [&](unsigned long long& no0, unsigned long long& no1, unsigned long long& in0, unsigned long long& in1) {
            const bool pa = (F0 >> lane) & 1ull, pb = (F1 >> lane) & 1ull;
            no0 = __ballot(ok0 && !pa && (y0 < 0.0));
            no1 = __ballot(ok1 && !pb && (y1 < 0.0));
            in0 = __ballot(ok0 && pa && (x0 < 0.0));
            in1 = __ballot(ok1 && pb && (x1 < 0.0));
        };
        unsigned long long no0, no1, in0, in1;
        sets(no0, no1, in0, in1);
        int ng = __popcll(no0) + __popcll(no1) + __popcll(in0) + __popcll(in1);
        int Pc = 3, Ninf = k + 1;
        int iter = 0;
        while (ng > 0) {
            if (iter >= max_iter) { failed = 1; break; }
            if (ng < Ninf) { Pc = 3; Ninf = ng; F0 = (F0 | no0) & ~in0; F1 = (F1 | no1) & ~in1; }
            else if (Pc >= 1) { Pc -= 1; F0 = (F0 | no0) & ~in0; F1 = (F1 | no1) & ~in1; }
            else {
                // backup rule: the largest index among the non-optimal / infeasible components changes sides
                const unsigned long long hi = no1 | in1, lo = no0 | in0;
                if (hi) F1 ^= 1ull << (63 - __clzll(hi));
                else F0 ^= 1ull << (63 - __clzll(lo));
            }
            F0 &= m0;
            F1 &= m1;
            solve(F0, F1);
            if (fabs(x0) < 1.0e-12) x0 = 0.0;
            if (fabs(x1) < 1.0e-12) x1 = 0.0;
            if (fabs(y0) < 1.0e-12) y0 = 0.0;
            if (fabs(y1) < 1.0e-12) y1 = 0.0;
            sets(no0, no1, in0, in1);
            ng = __popcll(no0) + __popcll(no1) + __popcll(in0) + __popcll(in1);
            ++iter;
        }
        if (fabs(x0) < 1.0e-12) x0 = 0.0;            // columns that never pivot are zeroized too (nnls_bpp_kernel's note)
        if (fabs(x1) < 1.0e-12) x1 = 0.0;
        if (fabs(y0) < 1.0e-12) y0 = 0.0;
        if (fabs(y1) < 1.0e-12) y1 = 0.0;
        X[col * KP + lane] = x0;
        if (ok1) X[col * KP + 64 + lane] = x1;
        if (Y) {
            Y[col * KP + lane] = y0;
            if (ok1) Y[col * KP + 64 + lane] = y1;
        }
        failed_any |= failed;
    }
    if (failed_any && lane == 0) atomicMin(fail_flag, iter_tag);
}

bool nnls_uses_tiles(int k)
{
    // default: from k = 65 (measured on 16384 x 8192, 12 iterations: k = 80 / 100 / 128 3.96 / 4.53 / 5.27 -> 2.39 / 2.84 / 3.12 ms per
    // iteration against nnls_bpp_inv128_kernel); SMK_NNLS_TILE128=0 keeps that kernel, =2 also sends k in (32, 64] here (A/B)
    static const int level = [] { const char* e = getenv("SMK_NNLS_TILE128"); return e ? atoi(e) : 1; }();
    return is_wide(k) || (level >= 1 && k > 64) || (level >= 2 && k > 32);
}

size_t nnls_scratch_elems(int k) { return (size_t)kp_of(k) * kp_of(k) + 8; }     // k <= 128; above: nnls_wide_scratch_elems
// k in (16, 32] also solves through the inverse of the Gram matrix (round 4; SMK_NNLS_INV32=0: the masked elimination, as before)
// Workgroups of the inverse-based kernels = resident workgroups x rounds.  Every workgroup first copies G and Ginv into LDS (16 /
// 64 KB) and a wave prefetches its next column, so few columns per wave waste both: one round up to 65536 columns (s_reuters 159 ->
// 154 us per iteration, 32768 x 8192 at k = 32 466 -> 453), four above, where more workgroups in flight hide more latency (10^6
// columns: 3.39 against 3.56 ms per iteration; C4 whole: no difference).  SMK_NNLS_ROUNDS overrides.
static inline int nnls_rounds(i64 ncols)
{
    static const int forced = [] { const char* e = getenv("SMK_NNLS_ROUNDS"); return e ? atoi(e) : 0; }();
    return forced > 0 ? forced : (ncols <= 65536 ? 1 : 4);
}

bool nnls_inverse_at_32()
{
    static const bool on = [] { const char* e = getenv("SMK_NNLS_INV32"); return !(e && e[0] == '0'); }();
    return on;
}

// k > 32: Ginv and the path selector into `scratch` (nnls_scratch_elems(k) doubles).  One workgroup, ~0.1 ms: the
// solver runs it on a side stream beside the streaming product that separates the Gram matrix from its NNLS.
int launch_gram_inverse(const double* G, int k, double* scratch, hipStream_t st)
{
    const int KPv = kp_of(k);
    if (KPv < 32 || KPv > 128 || !scratch) return 0;
    if (KPv == 32 && !nnls_inverse_at_32()) return 0;
    static const bool old64 = [] { const char* e = getenv("SMK_GRAM_INVERSE_OLD"); return e && atoi(e) != 0; }();
    if (KPv == 32) gram_inverse64_kernel<32><<<1, 256, 0, st>>>(G, k, scratch, (int*)(scratch + 32 * 32));
    else if (KPv == 64 && !old64) gram_inverse64_kernel<64><<<1, 256, 0, st>>>(G, k, scratch, (int*)(scratch + 64 * 64));
    else if (KPv == 64) gram_inverse_kernel<64><<<1, 256, 0, st>>>(G, k, scratch, (int*)(scratch + 64 * 64));
    else gram_inverse_kernel<128><<<1, 256, 0, st>>>(G, k, scratch, (int*)(scratch + 128 * 128));
    SMK_HIP(hipGetLastError());
    return 0;
}

// solves columns [col_begin, col_end) of X (col_end <= N); other columns are untouched.
// `scratch`: nnls_scratch_elems(k) doubles (the inverse of G and the path selector for k > 32);
// inverse_ready != 0: launch_gram_inverse(G, ...) has already been ordered before this call.
int launch_nnls_bpp(double* X, double* Y, int k, i64 col_begin, i64 col_end, PartialView R, const double* G,
                    int* fail_flag, int iter_tag, double* scratch, int inverse_ready, int num_cus, hipStream_t st,
                    double* gram_partials, int* gram_nblk, const NnlsPack* pack, unsigned* defer_ws, const NnlsRiders* riders)
{
    if (gram_nblk) *gram_nblk = 0;
    if (riders && riders->pg_nblk) *riders->pg_nblk = 0;
    if (nnls_uses_tiles(k)) return launch_nnls_bpp_wide(X, Y, k, col_begin, col_end, R, G, fail_flag, iter_tag, scratch, inverse_ready, num_cus, st);
    const int KPv = kp_of(k);
    const int gpb = 256 / KPv;
    const i64 ncols = col_end - col_begin;
    if (ncols <= 0) return 0;
    const int grid = (int)((ncols + gpb - 1) / gpb);
    const i64 N = col_end;
    static const int inv_mode = [] { const char* e = getenv("SMK_NNLS_INV"); return e ? atoi(e) : 1; }();
    const int* skip_if = nullptr;
    if (KPv == 128) {
        if (!scratch) { set_error("nnls: k > 64 needs the scratch buffer"); return -100; }
        constexpr int NT = 1024;
        double* Ginv = scratch;
        int* status = (int*)(scratch + 128 * 128);
        if (!inverse_ready) { int irc = launch_gram_inverse(G, k, scratch, st); if (irc) return irc; }
        const int lds = (128 * 128 + (NT / 64) * (128 + 64)) * (int)sizeof(double);
        SMK_HIP(hipFuncSetAttribute((const void*)nnls_bpp_inv128_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        i64 g2 = (ncols + NT / 64 - 1) / (NT / 64);
        if (g2 > (i64)num_cus * 4) g2 = (i64)num_cus * 4;
        nnls_bpp_inv128_kernel<NT><<<(unsigned)g2, NT, lds, st>>>(X, Y, k, N, R, G, Ginv, status, fail_flag, iter_tag, col_begin);
        SMK_HIP(hipGetLastError());
        return 0;
    }
    if (KPv == 32 && inv_mode && scratch && nnls_inverse_at_32()) {
        // k in (16, 32]: the same kernel, one wave per column with its upper half idle.  The masked elimination costs a full
        // 32-step Gauss-Jordan per exchange and column (99 us per launch on 8192 x 4096 at k = 32, two thirds of the iteration);
        // through the inverse an exchange is a t x t solve with t <= 16 (profiles/r04_gram_inverse.txt)
        double* Ginv = scratch;
        int* status = (int*)(scratch + 32 * 32);
        if (!inverse_ready) { int irc = launch_gram_inverse(G, k, scratch, st); if (irc) return irc; }
        constexpr int NT = 512;
        const int lds = (2 * 32 * 32 + (NT / 64) * 128) * (int)sizeof(double);
        i64 g2 = (ncols + NT / 64 - 1) / (NT / 64);
        const i64 cap = (i64)num_cus * 2 * nnls_rounds(ncols);
        if (g2 > cap) g2 = cap;
        // round 6: four columns per wave (nnls_g16.hip); SMK_NNLS_G16=0 keeps the wave-per-column kernel
        const int g16 = launch_nnls_bpp_g16(X, Y, k, col_begin, col_end, R, G, Ginv, status, fail_flag, iter_tag, nullptr, num_cus, st, nnls_stats_ptr());
        if (g16 < 0) return g16;
        if (g16 == 2) return 0;                   // that launch solves by masked elimination itself when the inverse was rejected
        if (g16 == 0) nnls_bpp_inv_kernel<32, 512, 4><<<(unsigned)g2, NT, lds, st>>>(X, Y, k, N, R, G, Ginv, status, fail_flag, iter_tag, col_begin, nnls_stats_ptr(), nullptr);
        SMK_HIP(hipGetLastError());
        skip_if = status;
    }
    if (KPv == 64 && inv_mode && scratch) {
        double* Ginv = scratch;
        int* status = (int*)(scratch + 64 * 64);
        if (!inverse_ready) { int irc = launch_gram_inverse(G, k, scratch, st); if (irc) return irc; }
        // round 6: the columns whose exchanges stay at t <= 16 are solved four per wave (nnls_g16.hip); the others arrive here
        // through the work list
        const unsigned* worklist = nullptr;
        if (defer_ws && inv_mode == 1) {
            const int g16 = launch_nnls_bpp_g16(X, Y, k, col_begin, col_end, R, G, Ginv, status, fail_flag, iter_tag, defer_ws, num_cus, st, nnls_stats_ptr());
            if (g16 < 0) return g16;
            if (g16 > 0) worklist = defer_ws;
        }
        auto run = [&](auto kern, int NT, int wg_per_cu) -> int {
            const int lds = (2 * 64 * 64 + (NT / 64) * 128) * (int)sizeof(double);
            SMK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            i64 g2 = (ncols + NT / 64 - 1) / (NT / 64);
            const i64 cap = (i64)num_cus * wg_per_cu * nnls_rounds(ncols);
            if (g2 > cap) g2 = cap;
            kern<<<(unsigned)g2, NT, lds, st>>>(X, Y, k, N, R, G, Ginv, status, fail_flag, iter_tag, col_begin, nnls_stats_ptr(), worklist);
            SMK_HIP(hipGetLastError());
            return 0;
        };
        int rc;
        if (inv_mode == 2) rc = run(nnls_bpp_inv_kernel<64, 768, 3>, 768, 1);
        else if (inv_mode == 3) rc = run(nnls_bpp_inv_kernel<64, 512, 2>, 512, 1);
        else if (inv_mode == 4) rc = run(nnls_bpp_inv_kernel<64, 512, 4, false>, 512, 2);   // solve bounds in steps of 8 (A/B)
        else rc = run(nnls_bpp_inv_kernel<64, 512, 4>, 512, 2);
        if (rc) return rc;
        skip_if = status;
    }
    int grid1 = grid;
    // the fallback behind the inverse-based kernel almost always exits on its first load: two workgroups per CU (6.1 -> ~3 us
    // of launch at 1500 workgroups, profiles/r05_s_reuters_kernel_stats.md) and more trips on the rare launch that does run
    if (skip_if && grid1 > num_cus * 2) grid1 = num_cus * 2;
    // KP = 16, all columns, at most NNLS_GRAM_MAX workgroups: the launch also leaves the Gram partials of the solved factor
    double* gp = nullptr;
    if (gram_partials && gram_nblk && KPv == 16 && col_begin == 0 && grid1 == grid && grid <= NNLS_GRAM_MAX) { gp = gram_partials; *gram_nblk = grid; }
    NnlsPack pk;                                           // only together with the Gram partials (one trip per workgroup, all columns)
    if (pack && gp) pk = *pack;
    NnlsRiders rd;                                         // only the launch that solves by itself (k <= 16) carries them
    if (riders && !skip_if && KPv <= 16) {
        rd = *riders;
        if (rd.pg_nblk) *rd.pg_nblk = rd.pg_part ? grid1 : 0;
    }
    KP_DISPATCH(KPv, (nnls_bpp_kernel<KP><<<grid1, 256, 0, st>>>(X, Y, k, N, R, G, fail_flag, iter_tag, col_begin, skip_if, gp, pk, skip_if ? nullptr : nnls_stats_ptr(), rd)));
    SMK_HIP(hipGetLastError());
    return 0;
}

}  // namespace smk
