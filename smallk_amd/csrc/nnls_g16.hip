// smallk_amd/csrc/nnls_g16.hip -- NnlsBlockpivot (nnls.hpp:144-244, src/nnls.cpp:18-74, nmf_solver_bpp.hpp:146-219) for
// k in (16, 64] with FOUR columns per wave: a column per 16-lane DPP row (round 6).
//
// Why: the wave-per-column kernel (nnls.hip: nnls_bpp_inv_kernel) is bound by VALU issue, not by memory or LDS -- counters on the
// 10^6-column solves of `s_1m` (k = 32): 557 vector instructions per column, SQ_ACTIVE_INST_VALU = 97 % of the launch, 0.13 of the
// HBM rate (profiles/r06_nnls_counters.md) -- and most of those instructions serve few lanes: the compact system of an exchange
// has t = min(|F|, |Z|) rows (measured mean 7.5 at k = 32 on sparse factors, 3.3 at k = 64 on noise-like data in steady state),
// one row per lane, so 48 - 60 of the 64 lanes idle through the elimination, and every pivot-row value crosses the wave as two
// v_readlane.  Grouping columns by identical passive set (the reference's BppSolveNormalEq, nmf_solver_bpp.hpp:29-142) does not
// apply: 33 - 100 % of the columns of a launch have a set of their own (profiles/r06_nnls_passive_sets.md).
//
// Here a 16-lane DPP row owns a column: lane l holds components l, l + 16, ... (E = KP / 16 of them), the compact system
// (t <= 16 rows: t <= KP / 2 always, so every solve at KP = 32 and the t <= 16 ones at KP = 64) sits one row per lane of the
// row, and pivot-row values travel by `v_mov_b64_dpp row_newbcast` -- one instruction per fp64 value, the one DPP control
// gfx90a+ keeps for 64-bit data.  (Default at KP = 32; KP = 64 is selectable, SMK_NNLS_G16=2 -- see g16_level below.)  The four columns of a wave run the same unrolled elimination (bound TB = the largest t of the
// four, rounded up to 4), each on its own system, form (complement on Ginv / direct on G) and state machine.  A column of KP = 64
// whose exchange needs t > 16 is handed over untouched to the wave-per-column kernel through a work list.
// Arithmetic and order of operations are those of nnls_bpp_inv_kernel (same compact elimination, same accumulation order of
// Ginv r and of M[:, T] u): the results are bit-identical to that kernel's (tests/test_gpu_nnls.py compares them).
#include "devutil.h"
#include "nnls_masked.h"

namespace smk {

template <int J>
__device__ __forceinline__ double row_bcast_f64(double v)     // lane J of every 16-lane row -> the whole row
{
    return __builtin_amdgcn_mov_dpp(v, 0x150 + J, 0xF, 0xF, true);
}
template <int J>
__device__ __forceinline__ int row_bcast_i32(int v)
{
    return __builtin_amdgcn_mov_dpp(v, 0x150 + J, 0xF, 0xF, true);
}
// J is a constant after unrolling: one case survives
__device__ __forceinline__ double row_bcast(double v, int J)
{
    switch (J) {
#define SMK_BC(n) case n: return row_bcast_f64<n>(v);
        SMK_BC(0) SMK_BC(1) SMK_BC(2) SMK_BC(3) SMK_BC(4) SMK_BC(5) SMK_BC(6) SMK_BC(7)
        SMK_BC(8) SMK_BC(9) SMK_BC(10) SMK_BC(11) SMK_BC(12) SMK_BC(13) SMK_BC(14) SMK_BC(15)
#undef SMK_BC
        default: return v;
    }
}
__device__ __forceinline__ int row_bcast(int v, int J)
{
    switch (J) {
#define SMK_BC(n) case n: return row_bcast_i32<n>(v);
        SMK_BC(0) SMK_BC(1) SMK_BC(2) SMK_BC(3) SMK_BC(4) SMK_BC(5) SMK_BC(6) SMK_BC(7)
        SMK_BC(8) SMK_BC(9) SMK_BC(10) SMK_BC(11) SMK_BC(12) SMK_BC(13) SMK_BC(14) SMK_BC(15)
#undef SMK_BC
        default: return v;
    }
}

template <int KP> struct G16Mask { typedef unsigned type; };
template <> struct G16Mask<64> { typedef unsigned long long type; };
__device__ __forceinline__ int g16_popc(unsigned m) { return __popc(m); }
__device__ __forceinline__ int g16_popc(unsigned long long m) { return __popcll(m); }
__device__ __forceinline__ int g16_top(unsigned m) { return m ? 31 - __clz((int)m) : 0; }
__device__ __forceinline__ int g16_top(unsigned long long m) { return m ? 63 - __clzll((long long)m) : 0; }

// S1: the right-hand side is one fp64 slab (every full-size launch: C4 whole, the sparse workloads), so the loads of the NEXT four
// columns are issued before the current four are touched; otherwise the slabs are summed on arrival (rhs_elem's order)
template <int KP, int NT, int WGS, bool S1, bool FINE = false>
__global__ __launch_bounds__(NT, WGS) void nnls_bpp_g16_kernel(double* __restrict__ X, double* __restrict__ Y, int k, i64 N, PartialView R,
                                                          const double* __restrict__ G, const double* __restrict__ Ginv,
                                                          const int* __restrict__ status, int* __restrict__ fail_flag, int iter_tag,
                                                          i64 col_begin, unsigned* __restrict__ defer,
                                                          unsigned long long* __restrict__ stats)
{
    static_assert(KP == 16 || KP == 32 || KP == 64, "a column per 16-lane row, KP / 16 components per lane");
    constexpr int E = KP / 16;
    constexpr int NW = NT / 64;
    typedef typename G16Mask<KP>::type mask_t;
    if (*status == 0) {
        // the inverse is not usable: the masked elimination solves.  At KP = 32 with 256-thread workgroups it runs right here (the
        // body of nnls_bpp_kernel<32>, nnls_masked.h) -- the decision is the device's, and a separate launch that returns at once
        // in every ordinary run cost 4 - 6 us per solve (8 % of an iteration on the Reuters shape); otherwise that launch follows
        if constexpr (KP == 32 && NT == 256) nnls_bpp_body<32>(X, Y, k, N, R, G, fail_flag, iter_tag, col_begin, nullptr, NnlsPack(), stats, NnlsRiders());
        return;
    }
    extern __shared__ __attribute__((aligned(16))) double lds[];
    // both matrices with PERMUTED columns: entry (r, c) at r * KP + pos(c), pos(c) = E (c mod 16) + c / 16, so that the E components
    // of a lane are adjacent (one or two ds_read_b128 per matrix row in the accumulation loops)
    double* gs = lds;                                           // G    (direct form)
    double* gis = lds + KP * KP;                                // Ginv (complement form)
    for (int t = threadIdx.x; t < KP * KP; t += NT) {
        const int r = t / KP, c = t % KP;
        const int p = r * KP + E * (c & 15) + (c >> 4);
        gs[p] = G[t];
        gis[p] = Ginv[t];
    }
    const int lane = threadIdx.x & 63;
    const int l = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double* svg = lds + 2 * KP * KP + (wave * 4 + g) * (KP + KP / 2);      // per column: KP doubles of values ...
    int* sig = (int*)(svg + KP);                                           // ... and KP ints of indices
    __syncthreads();

    const mask_t kmask = (k >= (int)(8 * sizeof(mask_t))) ? ~(mask_t)0 : (((mask_t)1 << k) - (mask_t)1);
    const int max_iter = 5 * k;
    bool ok[E];
#pragma unroll
    for (int e = 0; e < E; ++e) ok[e] = (l + 16 * e) < k;
    int failed_any = 0;

    auto group_mask = [&](const bool (&pred)[E]) -> mask_t {   // bit c = pred of component c of THIS lane's column
        mask_t m = 0;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const unsigned long long b = __ballot(pred[e]);
            m |= (mask_t)((unsigned)(b >> (16 * g)) & 0xFFFFu) << (16 * e);
        }
        return m;
    };

    const i64 nquads = (N - col_begin + 3) / 4;
    const i64 qstride = (i64)gridDim.x * NW;
    i64 quad = (i64)blockIdx.x * NW + wave;
    double rn[E], xn[E];                                        // S1: the next quad's right-hand side and start
#pragma unroll
    for (int e = 0; e < E; ++e) rn[e] = xn[e] = 0.0;
    if constexpr (S1) {
        const i64 col = col_begin + quad * 4 + g;
        if (quad < nquads && col < N) {
#pragma unroll
            for (int e = 0; e < E; ++e)
                if (ok[e]) { rn[e] = ((const double*)R.p)[col * R.kpp + l + 16 * e]; xn[e] = X[col * KP + l + 16 * e]; }
        }
    }
    for (; quad < nquads; quad += qstride) {
        const i64 col = col_begin + quad * 4 + g;
        const bool col_ok = col < N;
        double rhs[E], x[E], y[E], v[E];
#pragma unroll
        for (int e = 0; e < E; ++e) { y[e] = 0.0; v[e] = 0.0; }
        if constexpr (S1) {
#pragma unroll
            for (int e = 0; e < E; ++e) { rhs[e] = rn[e]; x[e] = xn[e]; rn[e] = xn[e] = 0.0; }
            const i64 coln = col + qstride * 4;
            if (quad + qstride < nquads && coln < N) {
#pragma unroll
                for (int e = 0; e < E; ++e)
                    if (ok[e]) { rn[e] = ((const double*)R.p)[coln * R.kpp + l + 16 * e]; xn[e] = X[coln * KP + l + 16 * e]; }
            }
        } else {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                rhs[e] = x[e] = 0.0;
                if (col_ok && ok[e]) { rhs[e] = rhs_elem(R, col, l + 16 * e); x[e] = X[col * KP + l + 16 * e]; }
            }
        }
        bool pr[E];
#pragma unroll
        for (int e = 0; e < E; ++e) pr[e] = ok[e] && x[e] > 0.0;
        mask_t F = group_mask(pr) & kmask;                      // passive_set = (X > 0), nnls.hpp:157

        bool have_v = false;                                    // wave-uniform
        bool deferred = false;                                  // per column
        int failed = 0;
        int nsolve = 0;                                         // diagnostics only

        // v = Ginv r for the four columns (needed by the complement form and when every variable is passive); accumulation order
        // of nnls_bpp_inv_kernel: even c into one sum, odd c into another
        auto need_v = [&]() {
            have_v = true;
            double v0[E], v1[E];
#pragma unroll
            for (int e = 0; e < E; ++e) v0[e] = v1[e] = 0.0;
            // a real loop over the E source registers (16 broadcasts each): fully unrolled, the scheduler hoists all KP broadcasts
            // and matrix rows ahead of the sums and the kernel needs 80 more registers
#pragma unroll 1
            for (int h = 0; h < E; ++h) {
                double src = rhs[0];
#pragma unroll
                for (int e = 1; e < E; ++e) src = (h == e) ? rhs[e] : src;
                const double* grow = gis + (16 * h) * KP + E * l;
#pragma unroll
                for (int j = 0; j < 16; j += 2) {
                    const double r0 = row_bcast(src, j);
                    const double r1 = row_bcast(src, j + 1);
                    if constexpr (E == 1) {
                        v0[0] = __builtin_fma(grow[j * KP], r0, v0[0]);
                        v1[0] = __builtin_fma(grow[(j + 1) * KP], r1, v1[0]);
                    } else {
#pragma unroll
                        for (int e = 0; e < E; e += 2) {
                            const f64x2_t m0 = *(const f64x2_t*)(grow + j * KP + e);
                            const f64x2_t m1 = *(const f64x2_t*)(grow + (j + 1) * KP + e);
                            v0[e] = __builtin_fma(m0[0], r0, v0[e]);
                            v0[e + 1] = __builtin_fma(m0[1], r0, v0[e + 1]);
                            v1[e] = __builtin_fma(m1[0], r1, v1[e]);
                            v1[e + 1] = __builtin_fma(m1[1], r1, v1[e + 1]);
                        }
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < E; ++e) v[e] = v0[e] + v1[e];
        };

        // compact solve of M[T,T] u = s_T on the first TB lanes of each row (rows >= t are identity rows), then
        // out = base + M[:,T] u.  TB is a compile-time bound: no branch inside the elimination.
        auto compact = [&](auto tb_tag, int mbase, int t, int tl, double sc, const double (&base)[E], double (&out)[E]) -> double {
            constexpr int TB = decltype(tb_tag)::value;
            const bool live = l < t;
            const int ptl = E * (tl & 15) + (tl >> 4);
            double a[TB];
#pragma unroll
            for (int b = 0; b < TB; ++b) {
                const int tb = row_bcast(tl, b);                               // lanes >= t carry tl = 0: harmless
                const double mv = lds[mbase + tb * KP + ptl];
                a[b] = (live && b < t) ? mv : ((b == l) ? 1.0 : 0.0);
            }
            double d = 1.0;
#pragma unroll
            for (int j = 0; j < TB; ++j) {
                const double piv = row_bcast(a[j], j);
                if (!(piv > 0.0)) failed = 1;
                const double ip = fast_rcp(piv);
                if (l == j) d = a[j];
                const double f = (l == j) ? 0.0 : a[j] * ip;
#pragma unroll
                for (int c = j + 1; c < TB; ++c) a[c] = __builtin_fma(-f, row_bcast(a[c], j), a[c]);
                sc = __builtin_fma(-f, row_bcast(sc, j), sc);
            }
            const double u = live ? sc * fast_rcp(d) : 0.0;
#pragma unroll
            for (int e = 0; e < E; ++e) out[e] = base[e];
#pragma unroll
            for (int b = 0; b < TB; ++b) {
                const int tb = row_bcast(tl, b);
                const double ub = row_bcast(u, b);                             // u = 0 beyond t
                const double* mrow = lds + mbase + tb * KP + E * l;
                if constexpr (E == 1) out[0] = __builtin_fma(mrow[0], ub, out[0]);
                else {
#pragma unroll
                    for (int e = 0; e < E; e += 2) {
                        const f64x2_t mm = *(const f64x2_t*)(mrow + e);
                        out[e] = __builtin_fma(mm[0], ub, out[e]);
                        out[e + 1] = __builtin_fma(mm[1], ub, out[e + 1]);
                    }
                }
            }
            return u;
        };

        // one block-pivot solve per column with `act` set: leaves x (zero outside the passive set) and y (zero inside)
        auto solve = [&](mask_t Fs, bool& act) {
            const mask_t Zs = ~Fs & kmask;
            const int p = g16_popc(Fs), q = g16_popc(Zs);
            const bool comp = q <= p;                                           // complement form on Ginv (also q == 0)
            const bool trivial = q == 0 || p == 0;
            int t = (act && !trivial) ? (comp ? q : p) : 0;
            if (stats && l == 0 && act) {
                nnls_stat(stats, (nsolve == 0 ? 16 : 96) + (comp ? q : p));
                nnls_stat(stats, q == 0 ? 179 : p == 0 ? 180 : comp ? 176 : 177);
            }
            ++nsolve;
            if constexpr (KP == 64) {
                if (t > 16) { deferred = true; act = false; t = 0; }           // this column goes to the wave-per-column kernel
            }
            if (!have_v && __ballot(act && comp) != 0ull) need_v();
            // the largest compact system of the four columns bounds the unrolled elimination
            const int tmax = max(max(__builtin_amdgcn_readlane(t, 0), __builtin_amdgcn_readlane(t, 16)),
                                 max(__builtin_amdgcn_readlane(t, 32), __builtin_amdgcn_readlane(t, 48)));
            const mask_t T = comp ? Zs : Fs;
            bool inT[E], inF[E];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                inT[e] = (T >> (l + 16 * e)) & (mask_t)1;
                inF[e] = (Fs >> (l + 16 * e)) & (mask_t)1;
            }
            double out[E], ut[E];
#pragma unroll
            for (int e = 0; e < E; ++e) out[e] = ut[e] = 0.0;
            if (tmax > 0) {
                // compact index list: lane l < t of the row gets the l-th member of T
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int c = l + 16 * e;
                    if (t > 0 && inT[e]) sig[g16_popc((mask_t)(T & (((mask_t)1 << c) - (mask_t)1)))] = c;
                    svg[c] = comp ? -v[e] : rhs[e];
                }
                __builtin_amdgcn_wave_barrier();
                const int tl = (l < t) ? sig[l] : 0;
                const double sc = (l < t) ? svg[tl] : 0.0;
                double base[E];
#pragma unroll
                for (int e = 0; e < E; ++e) base[e] = comp ? v[e] : -rhs[e];
                const int mbase = comp ? KP * KP : 0;
                double u;
                // bounds in steps of 2 (SMK_NNLS_G16_FINE=0: steps of 4): the elimination costs ~TB^2 / 2 broadcast + FMA pairs and the
                // bound is the LARGEST system of the four columns
                if (FINE && tmax <= 2) u = compact(std::integral_constant<int, 2>{}, mbase, t, tl, sc, base, out);
                else if (tmax <= 4) u = compact(std::integral_constant<int, 4>{}, mbase, t, tl, sc, base, out);
                else if (FINE && tmax <= 6) u = compact(std::integral_constant<int, 6>{}, mbase, t, tl, sc, base, out);
                else if (tmax <= 8) u = compact(std::integral_constant<int, 8>{}, mbase, t, tl, sc, base, out);
                else if (FINE && tmax <= 10) u = compact(std::integral_constant<int, 10>{}, mbase, t, tl, sc, base, out);
                else if (tmax <= 12) u = compact(std::integral_constant<int, 12>{}, mbase, t, tl, sc, base, out);
                else if (FINE && tmax <= 14) u = compact(std::integral_constant<int, 14>{}, mbase, t, tl, sc, base, out);
                else u = compact(std::integral_constant<int, 16>{}, mbase, t, tl, sc, base, out);
                // u back to component positions
                __builtin_amdgcn_wave_barrier();
                if (l < t) svg[tl] = u;
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int e = 0; e < E; ++e) ut[e] = (t > 0 && inT[e]) ? svg[l + 16 * e] : 0.0;
                __builtin_amdgcn_wave_barrier();
            }
            if (act) {
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    if (q == 0) { x[e] = v[e]; y[e] = 0.0; }
                    else if (p == 0) { x[e] = 0.0; y[e] = ok[e] ? -rhs[e] : 0.0; }
                    else if (comp) { x[e] = inF[e] ? out[e] : 0.0; y[e] = ut[e]; }
                    else { x[e] = ut[e]; y[e] = (ok[e] && !inF[e]) ? out[e] : 0.0; }
                }
            }
        };

        auto sets = [&](mask_t Fc, mask_t& nonopt, mask_t& infeas) {
            bool pn[E], pi[E];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const bool pas = (Fc >> (l + 16 * e)) & (mask_t)1;
                pn[e] = ok[e] && !pas && (y[e] < 0.0);
                pi[e] = ok[e] && pas && (x[e] < 0.0);
            }
            nonopt = group_mask(pn);
            infeas = group_mask(pi);
        };

        bool active = col_ok;
        solve(F, active);
        mask_t nonopt, infeas;
        sets(F, nonopt, infeas);
        int ng = g16_popc(nonopt) + g16_popc(infeas);
        int Pc = 3, Ninf = k + 1;                    // PBAR = 3, nnls.hpp:152,170
        int iter = 0;
        active = active && ng > 0;
        while (__ballot(active) != 0ull) {
            if (active && iter >= max_iter) { failed = 1; active = false; }
            if (active) {
                // UpdatePassiveSet, src/nnls.cpp:18-74
                if (ng < Ninf) { Pc = 3; Ninf = ng; F = (F | nonopt) & ~infeas; }
                else if (Pc >= 1) { Pc -= 1; F = (F | nonopt) & ~infeas; }
                else {
                    const int r1 = g16_top(nonopt), r2 = g16_top(infeas);
                    F ^= ((mask_t)1 << (r1 > r2 ? r1 : r2));
                }
                F &= kmask;
            }
            solve(F, active);
            if (active) {
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    if (fabs(x[e]) < 1.0e-12) x[e] = 0.0;          // ZeroizeSmallValues, nnls.hpp:213,224
                    if (fabs(y[e]) < 1.0e-12) y[e] = 0.0;          // :225
                }
            }
            mask_t no2, in2;
            sets(F, no2, in2);
            if (active) {
                nonopt = no2;
                infeas = in2;
                ng = g16_popc(nonopt) + g16_popc(infeas);
                ++iter;
                if (ng == 0) active = false;
            }
        }
        if (col_ok && !deferred) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                if (fabs(x[e]) < 1.0e-12) x[e] = 0.0;              // columns that never pivot are zeroized too (nnls_bpp_kernel's note)
                if (fabs(y[e]) < 1.0e-12) y[e] = 0.0;
                if (ok[e]) {
                    X[col * KP + l + 16 * e] = x[e];
                    if (Y) Y[col * KP + l + 16 * e] = y[e];
                }
            }
            if (stats && l == 0) { nnls_stat(stats, iter < 15 ? iter : 15); nnls_stat(stats, 178); }
            failed_any |= failed;
        }
        if constexpr (KP == 64) {
            if (deferred && l == 0) defer[1 + atomicAdd(&defer[0], 1u)] = (unsigned)(col - col_begin);
        }
    }
    if (failed_any && l == 0) atomicMin(fail_flag, iter_tag);
}

// SMK_NNLS_G16: 0 = off (a wave per column everywhere), 1 = k in (16, 32] (default), 2 = also k in (32, 64].
// At KP = 64 the kernel is built and bit-identical but NOT the default: measured against the wave-per-column kernel on one box
// (profiles/r06_nnls_g16_k64.txt) it gains 1.5 % on a C4 shard in steady state (W-side launch 460 -> 295 us), nothing on C4 whole or
// on rank 0 of an emulated 8, and LOSES 2 % over the first twenty iterations of a cold start (every column's first exchange has
// t ~ 26 > 16 and is handed over after a wasted first solve) and 5 - 10 % on 16384 x 8192 (three more launches per side on a
// launch-bound iteration).
static int g16_level()
{
    static const int level = [] { const char* e = getenv("SMK_NNLS_G16"); return e ? atoi(e) : 1; }();
    return level;
}

// the four-columns-per-wave launch; returns 1 when it was issued, 2 when it was issued AND carries the masked-elimination fallback
// itself (no separate fallback launch needed) (0: not applicable, < 0: error).  KP = 64: `defer` receives the
// columns that must still be solved by nnls_bpp_inv_kernel (defer[0] = count, zeroed here; defer[1 ..] = column - col_begin)
int launch_nnls_bpp_g16(double* X, double* Y, int k, i64 col_begin, i64 col_end, PartialView R, const double* G, const double* Ginv,
                        const int* status, int* fail_flag, int iter_tag, unsigned* defer, int num_cus, hipStream_t st,
                        unsigned long long* stats)
{
    const int KPv = kp_of(k);
    if (g16_level() < 1 || (KPv != 32 && KPv != 64)) return 0;
    if (KPv == 64 && (g16_level() < 2 || !defer)) return 0;
    const i64 ncols = col_end - col_begin;
    if (ncols <= 0) return 0;
    const bool s1 = R.S == 1 && R.f64;
    const i64 nquads = (ncols + 3) / 4;
    static const int wgs = [] { const char* e = getenv("SMK_NNLS_G16_WGS"); return e ? atoi(e) : 0; }();
    if (KPv == 32) {
        // shapes (SMK_NNLS_G16_SHAPE, A/B; s_1m it/s on one box): 3 = 256 threads, three workgroups per CU at <= 168 registers (no
        // spills), elimination bounds in steps of 2 (default: 441); 0 = the same with bounds in steps of 4 (428); 1 = 512 threads at
        // <= 128 registers (four waves per SIMD, 144 bytes of scratch per lane: 395); 2 = 512 threads, two waves per SIMD (408)
        static const int shape = [] { const char* e = getenv("SMK_NNLS_G16_SHAPE"); return e ? atoi(e) : 3; }();
        auto run = [&](auto kern, int NT, int wg_per_cu) -> int {
            const int lds = (2 * 32 * 32 + (NT / 64) * 4 * 48) * (int)sizeof(double);
            i64 g2 = (nquads + NT / 64 - 1) / (NT / 64);
            const i64 cap = (i64)num_cus * (wgs > 0 ? wgs : wg_per_cu);
            if (g2 > cap) g2 = cap;
            kern<<<(unsigned)g2, NT, lds, st>>>(X, Y, k, col_end, R, G, Ginv, status, fail_flag, iter_tag, col_begin, nullptr, stats);
            return 0;
        };
        if (shape == 1) { if (s1) run(nnls_bpp_g16_kernel<32, 512, 4, true>, 512, 2); else run(nnls_bpp_g16_kernel<32, 512, 4, false>, 512, 2); }
        else if (shape == 2) { if (s1) run(nnls_bpp_g16_kernel<32, 512, 2, true>, 512, 1); else run(nnls_bpp_g16_kernel<32, 512, 2, false>, 512, 1); }
        else if (shape == 0) { if (s1) run(nnls_bpp_g16_kernel<32, 256, 3, true>, 256, 3); else run(nnls_bpp_g16_kernel<32, 256, 3, false>, 256, 3); }
        else { if (s1) run(nnls_bpp_g16_kernel<32, 256, 3, true, true>, 256, 3); else run(nnls_bpp_g16_kernel<32, 256, 3, false, true>, 256, 3); }
        SMK_HIP(hipGetLastError());
        return (shape == 1 || shape == 2) ? 1 : 2;
    } else {
        constexpr int NT = 512;
        const int lds = (2 * 64 * 64 + (NT / 64) * 4 * 96) * (int)sizeof(double);
        SMK_HIP(hipMemsetAsync(defer, 0, sizeof(unsigned), st));
        i64 g2 = (nquads + NT / 64 - 1) / (NT / 64);
        const i64 cap = (i64)num_cus * (wgs > 0 ? wgs : 1);
        if (g2 > cap) g2 = cap;
        if (s1) {
            SMK_HIP(hipFuncSetAttribute((const void*)nnls_bpp_g16_kernel<64, NT, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            nnls_bpp_g16_kernel<64, NT, 1, true><<<(unsigned)g2, NT, lds, st>>>(X, Y, k, col_end, R, G, Ginv, status, fail_flag, iter_tag, col_begin, defer, stats);
        } else {
            SMK_HIP(hipFuncSetAttribute((const void*)nnls_bpp_g16_kernel<64, NT, 1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            nnls_bpp_g16_kernel<64, NT, 1, false><<<(unsigned)g2, NT, lds, st>>>(X, Y, k, col_end, R, G, Ginv, status, fail_flag, iter_tag, col_begin, defer, stats);
        }
    }
    SMK_HIP(hipGetLastError());
    return 1;
}

}  // namespace smk
