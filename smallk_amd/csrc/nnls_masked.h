// smallk_amd/csrc/nnls_masked.h -- the masked Gauss-Jordan block-pivoting solve of one column per KP-lane group (device code shared by
// nnls.hip and nnls_g16.hip; moved out of nnls.hip in round 6, unchanged).
#pragma once
#include "devutil.h"

namespace smk {

// ==========================================================================
// NNLS by block principal pivoting, one column per GS-lane group (GS = KP).
// Lane i of a group owns component i of the column: x_i, y_i, rhs_i, its
// passive bit, and row i of the masked Gram matrix in registers.  The passive
// sub-system G[F,F] x_F = rhs_F is solved by Gauss-Jordan elimination on the
// masked matrix (non-passive rows/columns replaced by identity) -- pivots are
// the Cholesky pivots, so "pivot <= 0" is exactly the reference's non-SPD
// failure (normal_eq.hpp:35-50).  Pivot-row values are broadcast with
// v_readlane (GS = 64) or ds_bpermute (GS < 64).
// Per-column state machine = NnlsBlockpivot (nnls.hpp:144-244) restricted to one
// column: columns are independent in the reference except for the shared
// iteration cap (5k), which here is per column.  The backup rule toggles the
// TRUE largest index (the reference's MaxRowIndex is off by 32 for k >= 64,
// bit_matrix.cpp:456-468; the NNLS optimum is unique so results agree).
// ==========================================================================
// GS = 16: a column group is one DPP row, and row_newbcast:N (the one DPP control gfx90a+ keeps for 64-bit data) puts lane
// N of every row into all 16 lanes of that row -- two VALU moves, no trip through the LDS crossbar
template <int J>
__device__ __forceinline__ double row16_bcast(double v)
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x150 + J, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x150 + J, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

template <int GS>
__device__ __forceinline__ double group_bcast(double v, int src /* compile-time after unroll */)
{
    if constexpr (GS == 64) {
        int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
        int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
        return __hiloint2double(hi, lo);
    } else if constexpr (GS == 16) {
        switch (src) {          // src is a constant after unrolling: one case survives
#define SMK_BC(n) case n: return row16_bcast<n>(v);
            SMK_BC(0) SMK_BC(1) SMK_BC(2) SMK_BC(3) SMK_BC(4) SMK_BC(5) SMK_BC(6) SMK_BC(7)
            SMK_BC(8) SMK_BC(9) SMK_BC(10) SMK_BC(11) SMK_BC(12) SMK_BC(13) SMK_BC(14) SMK_BC(15)
#undef SMK_BC
            default: return v;
        }
    } else {
        return __shfl(v, src, GS);
    }
}

template <int GS>
__device__ __forceinline__ unsigned long long group_ballot(bool pred, int lane)
{
    unsigned long long b = __ballot(pred);
    if constexpr (GS == 64) return b;
    else {
        const int shift = (lane / GS) * GS;
        return (b >> shift) & ((1ull << GS) - 1ull);
    }
}

// The body of nnls_bpp_kernel<KP> as a device function of a 256-thread workgroup (round 6): the standalone kernel (nnls.hip) calls
// it, and so does nnls_bpp_g16_kernel<32> (nnls_g16.hip) when the Gram inverse was rejected -- the decision is taken on the device
// (the status word), and until round 6 that cost a second launch per solve that returned at once in every ordinary run.
template <int KP>
__device__ __forceinline__ void nnls_bpp_body(double* __restrict__ X, double* __restrict__ Y, int k, i64 N,
                                              PartialView R, const double* __restrict__ G,
                                              int* __restrict__ fail_flag, int iter_tag, i64 col_begin,
                                              double* __restrict__ Gp, NnlsPack pk,
                                              unsigned long long* __restrict__ stats, NnlsRiders rd)
{
    constexpr int GS = KP;
    constexpr int GPB = 256 / GS;                   // column groups per block
    __shared__ double gs[KP * KP];                  // gs[c*KP + i] = G[i][c] (symmetric)
    const int lane = threadIdx.x & 63;
    const int i = threadIdx.x % GS;                 // component owned by this lane
    // the loads of the first trip -- right-hand side (one load per row split) and start -- are issued together with those of G,
    // ahead of the barrier that publishes G: two dependent round trips to memory become one (a launch is ~12 us of such latencies)
    double rhs_first = 0.0, x_first = 0.0;
    {
        const i64 col = col_begin + (i64)blockIdx.x * GPB + threadIdx.x / GS;
        if ((i64)blockIdx.x * GPB < N - col_begin && i < k) {
            const i64 cc = col < N ? col : (N - 1);
            rhs_first = rhs_elem(R, cc, i);
            x_first = X[cc * KP + i];
        }
    }
    for (int t = threadIdx.x; t < KP * KP; t += blockDim.x) gs[t] = G[t];
    __syncthreads();

    // NnlsPack: the row scale of component i, from the diagonal of the system matrix (computed here, ahead of the solve)
    double pack_xs = 1.0;
    if constexpr (KP == 16) {
        if (pk.out) {
            const double g = gs[i * KP + i];
            if (g > 0.0 && g < 1.0e300) {
                const double bound = pk.anorm / sqrt(g);
                if (bound > 1.0e-290 && bound < 1.0e290) {
                    int ex = 0;
                    (void)frexp(bound, &ex);                    // bound < 2^ex
                    pack_xs = ldexp(1.0, 15 - ex);
                }
            }
        }
    }
    // KP = 16, Gp != nullptr: the Gram matrix X X' of the SOLVED columns comes out of this launch too.  A wave holds 4 columns
    // x 16 components with lane = component + 16 column -- exactly the A (and B) operand of v_mfma_f64_16x16x4 -- so one
    // matrix instruction per trip accumulates the wave's 16 x 16 partial; the four waves are added through LDS and the
    // workgroup leaves its partial in Gp[blockIdx] (the layout gram_reduce_kernel sums).  Saves the separate Gram launch over
    // the factor on latency-bound problems (C2: 4.6 us + a launch gap per side).
    typedef double f64x4_acc __attribute__((ext_vector_type(4)));
    f64x4_acc gacc = {0.0, 0.0, 0.0, 0.0};
    double pg_sum = 0.0;                            // NnlsRiders::pg_part
    // grid-stride over blocks of GPB columns: the k in (32, 64] fallback is launched with a small grid so that its
    // usual early exit costs 2 us, not one workgroup per 4 columns; every other launch covers its columns in one trip
    for (i64 vb = blockIdx.x; vb * GPB < N - col_begin; vb += gridDim.x) {
    const i64 col = col_begin + vb * GPB + threadIdx.x / GS;
    const bool col_ok = col < N;
    const bool comp_ok = i < k;
    const i64 cc = col_ok ? col : (N - 1);

    double rhs = 0.0, x = 0.0, y = 0.0;
    if (vb == (i64)blockIdx.x) {
        rhs = rhs_first;
        x = x_first;
    } else if (comp_ok) {
        rhs = rhs_elem(R, cc, i);
        x = X[cc * KP + i];
    }
    if (rd.pg_part) {                               // NnlsRiders: projected gradient of the warm start (uniform branch)
        double acc = 0.0;
#pragma unroll
        for (int c = 0; c < KP; ++c) acc += gs[c * KP + i] * group_bcast<GS>(x, c);
        const double gq = comp_ok ? (acc - rhs) : 0.0;
        if (col_ok && comp_ok && (gq < 0.0 || x > 0.0)) pg_sum += gq * gq;
    }
    bool passive = comp_ok && (x > 0.0);            // passive_set = (X > 0), nnls.hpp:157
    const unsigned long long kmask = (k >= 64) ? ~0ull : ((1ull << k) - 1ull);
    int failed = 0;                                 // per trip

    auto solve = [&](unsigned long long F) {
        // masked matrix row i
        double a[KP];
#pragma unroll
        for (int c = 0; c < KP; ++c) {
            const bool pc = (F >> c) & 1ull;
            a[c] = (passive && pc) ? gs[c * KP + i] : ((c == i) ? 1.0 : 0.0);
        }
        double b = passive ? rhs : 0.0;
#pragma unroll
        for (int j = 0; j < KP; ++j) {
            // wave-uniform skip when no group in this wave has j passive
            const bool pj = (F >> j) & 1ull;
            if (__ballot(pj) == 0ull) continue;
            const double piv = group_bcast<GS>(a[j], j);
            if (pj && !(piv > 0.0)) failed = 1;
            // v_rcp_f64 + two Newton steps instead of a division: half the dependent chain of the sequential pivots (round 6: C2
            // 13 120 -> 13 310 it/s over three runs each; all 116 NNLS cases and the goldens unchanged at their tolerances)
            const double f = (i == j || !pj) ? 0.0 : a[j] * fast_rcp(piv);
#pragma unroll
            for (int c = j + 1; c < KP; ++c) a[c] -= f * group_bcast<GS>(a[c], j);
            b -= f * group_bcast<GS>(b, j);
        }
        double d = 1.0;
#pragma unroll
        for (int c = 0; c < KP; ++c)
            if (c == i) d = a[c];
        x = passive ? (b * fast_rcp(d)) : 0.0;
    };

    auto residual = [&]() {          // y = G x - rhs
        double acc = 0.0;
#pragma unroll
        for (int c = 0; c < KP; ++c) acc += gs[c * KP + i] * group_bcast<GS>(x, c);
        y = comp_ok ? (acc - rhs) : 0.0;
    };

    unsigned long long F = group_ballot<GS>(passive, lane) & kmask;
    if (stats && i == 0 && col_ok) { nnls_stat(stats, 16 + __popcll(F)); nnls_stat(stats, 178); }
    solve(F);
    residual();

    unsigned long long nonopt = group_ballot<GS>(comp_ok && !passive && (y < 0.0), lane);
    unsigned long long infeas = group_ballot<GS>(comp_ok && passive && (x < 0.0), lane);
    int ng = __popcll(nonopt) + __popcll(infeas);
    int Pc = 3, Ninf = k + 1;                       // PBAR = 3, nnls.hpp:152,170
    const int max_iter = 5 * k;
    int iter = 0;
    bool active = col_ok && ng > 0;

    while (__ballot(active) != 0ull) {
        if (active) {
            if (iter >= max_iter) { failed = 1; active = false; }
        }
        if (active) {
            // UpdatePassiveSet, src/nnls.cpp:18-74
            if (ng < Ninf) { Pc = 3; Ninf = ng; F = (F | nonopt) & ~infeas; }
            else if (Pc >= 1) { Pc -= 1; F = (F | nonopt) & ~infeas; }
            else {
                const int r1 = nonopt ? (63 - __clzll(nonopt)) : 0;
                const int r2 = infeas ? (63 - __clzll(infeas)) : 0;
                F ^= (1ull << (r1 > r2 ? r1 : r2));
            }
            F &= kmask;
            passive = (F >> i) & 1ull;
        }
        // all lanes execute the cross-lane code; inactive groups keep their state
        const double x_keep = x, y_keep = y;
        solve(F);
        if (fabs(x) < 1.0e-12) x = 0.0;             // ZeroizeSmallValues, nnls.hpp:213,224
        residual();
        if (fabs(y) < 1.0e-12) y = 0.0;             // :225
        if (!active) { x = x_keep; y = y_keep; }
        const unsigned long long no2 = group_ballot<GS>(comp_ok && !passive && (y < 0.0), lane);
        const unsigned long long in2 = group_ballot<GS>(comp_ok && passive && (x < 0.0), lane);
        if (active) {
            nonopt = no2;
            infeas = in2;
            ng = __popcll(nonopt) + __popcll(infeas);
            ++iter;
            if (ng == 0) active = false;
        }
    }
    if (stats && i == 0 && col_ok) nnls_stat(stats, iter < 15 ? iter : 15);

    // The reference zeroizes the WHOLE X and Y after every pivoting round (nnls.hpp:224-225), i.e. also the columns that never
    // pivot -- as soon as ANY column of the solve does, which a workgroup cannot know.  Some column pivots in practically every
    // solve of a run that has not converged, so the columns that never pivot are zeroized here as well (the columns that did
    // pivot already are).  Differs from the reference only in solves where no column at all pivots, by entries below 1e-12.
    if (fabs(x) < 1.0e-12) x = 0.0;
    if (fabs(y) < 1.0e-12) y = 0.0;
    if (col_ok && comp_ok) {
        X[col * KP + i] = x;
        if (Y) Y[col * KP + i] = y;
    }
    if (rd.snap_x && col_ok && i < rd.k2) rd.snap_x[col * rd.k2 + i] = comp_ok ? x : 0.0;
    if (failed && col_ok) atomicMin(fail_flag, iter_tag);
    if constexpr (KP == 16) {
        if (Gp) {
            const double xg = (col_ok && comp_ok) ? x : 0.0;
            gacc = __builtin_amdgcn_mfma_f64_16x16x4f64(xg, xg, gacc, 0, 0, 0);
        }
        // The packed operand of the product that follows (pack_f16x2_kernel's layout, KT = 1: this workgroup's 16 columns are
        // chunk pair q = blockIdx.x), with row scales that need no pass over the solved factor -- see NnlsPack (common.h):
        // x_i <= anorm / sqrt(G_ii) at a KKT point of a problem whose other factor is non-negative, so 2^15 / (the next power
        // of two above that bound) keeps every entry below 2^15 with 2x to spare.  One trip per workgroup (the launcher
        // guarantees it), so the barrier below is uniform.
        if (pk.out) {
            __shared__ double xsh[256];
            xsh[(threadIdx.x / GS) * 16 + i] = (col_ok && comp_ok) ? x * pack_xs : 0.0;
            if (blockIdx.x == 0 && threadIdx.x < 16) {
                pk.xscale[i] = pack_xs;
                pk.oscale[i] = 1.0 / (pack_xs * pk.ascale);
            }
            __syncthreads();
            // fragment lane l = (r, h) holds rows 8 h .. 8 h + 7 of component r as 8 halves (16 bytes) per term; thread t converts
            // entries 2 p, 2 p + 1 of lane l = t & 63 with p = t >> 6, so the four waves share the work and nobody waits for one
            {
                const int l = threadIdx.x & 63, p2 = (threadIdx.x >> 6) * 2, r = l & 31, h = l >> 5;
                double r0 = 0.0, r1 = 0.0;
                if (r < 16) { r0 = xsh[(h * 8 + p2) * 16 + r]; r1 = xsh[(h * 8 + p2 + 1) * 16 + r]; }
                // (finite values only: a solve that met a non-positive pivot leaves inf / NaN behind and has raised the failure
                // flag itself; that run fails as the reference's does and must not be repeated)
                const double a0 = fabs(r0), a1 = fabs(r1);
                if ((a0 >= 65504.0 && a0 < 1.0e300) || (a1 >= 65504.0 && a1 < 1.0e300)) atomicMin(fail_flag, NNLS_PACK_OVERFLOW);
                typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
                unsigned char* dst = pk.out + (size_t)vb * 2048 + l * 16 + p2 * 2;
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2) {
                    f16x2_t hh;
                    hh[0] = (_Float16)(float)r0;
                    hh[1] = (_Float16)(float)r1;
                    r0 = (r0 - (double)(float)hh[0]) * 2048.0;          // F16X2_LO_SCALE: the low term is carried 2^11 up
                    r1 = (r1 - (double)(float)hh[1]) * 2048.0;
                    *(f16x2_t*)(dst + t2 * 1024) = hh;
                }
                // rows past the last column up to the padded length: zero chunk pairs, written by the last workgroup
                if (vb == (i64)gridDim.x - 1) {
                    const f16x2_t zero = {0, 0};
                    for (i64 q = (i64)gridDim.x; q < pk.nq; ++q) {
                        *(f16x2_t*)(pk.out + (size_t)q * 2048 + l * 16 + p2 * 2) = zero;
                        *(f16x2_t*)(pk.out + (size_t)q * 2048 + 1024 + l * 16 + p2 * 2) = zero;
                    }
                }
            }
        }
    }
    }
    if (rd.pg_part) {                               // one partial per workgroup, lanes and waves added in a fixed order
        __shared__ double pg_sh[8];
        const double t = block_sum(pg_sum, pg_sh);
        if (threadIdx.x == 0) rd.pg_part[blockIdx.x] = t;
    }
    if constexpr (KP == 16) {
        if (Gp) {
            // D: column = lane & 15, row = (lane >> 4) + 4 reg; waves added in a fixed order through LDS (gs is free now)
            const int wave = threadIdx.x >> 6, kc = lane >> 4, r16 = lane & 15;
            __syncthreads();
            for (int w = 0; w < 4; ++w) {
                if (wave == w) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int idx = r16 * KP + kc + 4 * r;
                        gs[idx] = (w == 0) ? gacc[r] : gs[idx] + gacc[r];
                    }
                }
                __syncthreads();
            }
            Gp[(i64)blockIdx.x * KP * KP + threadIdx.x] = gs[threadIdx.x];
            // the diagonal once more, compact, behind the partials: what reduce_pack_f16x2_k16_kernel's packers add up
            if (threadIdx.x < KP) Gp[(i64)NNLS_GRAM_MAX * KP * KP + (i64)blockIdx.x * KP + threadIdx.x] = gs[threadIdx.x * (KP + 1)];
        }
    }
}

}  // namespace smk
