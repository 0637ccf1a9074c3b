// The Gram matrix of a factor as device functions of 256-thread workgroups: launched by themselves (kernels.hip: gram_mfma_kernel,
// gram_reduce_kernel) or riding in the two launches of a sparse gather product -- the partial sums as extra workgroups of
// spmm_seg_kernel (both only read the factor), their reduction as extra workgroups of the fix-up launch behind it (common.h:
// GramRide).  Four launches become two on a launch-bound iteration (the Reuters shape under HALS: 102 -> 84 us).
// Reference: the W'W / HH' products of nmf_solver_hals.hpp:166-199, nmf_solver_mu.hpp:121-164.
#pragma once
#include "common.h"
#include "devutil.h"

namespace smk {

typedef __attribute__((ext_vector_type(4))) double f64x4_t;

template <int KP>
__device__ __forceinline__ void gram_mfma_body(const double* __restrict__ X, i64 N, i64 cols_per_wave, double* __restrict__ Gp,
                                               i64 blk, double* __restrict__ red /* KP * KP doubles of LDS (unused on the KP = 64 short-factor path) */)
{
    constexpr int T = KP / 16;
    if (KP == 64 && cols_per_wave <= 32) {
        // KP = 64, short factors (<= 128 columns per workgroup: N <= 32768 with 256 partials; longer ones would re-read more than the
        // L1 holds -- C4's W side 133 us against ~90): wave w owns tile ROW w of the result (4 of the 16 tiles) over ALL columns of the workgroup, instead of all 16
        // tiles over a quarter of the columns: the same matrix instructions per wave, the loads four times (the four waves read
        // the same lines at the same time), and no sum over the waves -- that sum (four turns of 4096 LDS read-modify-writes
        // between barriers) was 8 of the 19 us this launch took on a 4096-column factor.
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const i64 c_begin = blk * 4 * cols_per_wave;
        i64 c_end = c_begin + 4 * cols_per_wave;
        if (c_end > N) c_end = N;
        f64x4_t acc[T];
#pragma unroll
        for (int b = 0; b < T; ++b) acc[b] = f64x4_t{0.0, 0.0, 0.0, 0.0};
        const int kc = lane >> 4, r16 = lane & 15;
        for (i64 c0 = c_begin; c0 < c_end; c0 += 16) {
            double f[4][T];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const i64 col = c0 + 4 * u + kc;
                const bool ok = col < c_end;
#pragma unroll
                for (int t = 0; t < T; ++t) f[u][t] = ok ? X[col * KP + 16 * t + r16] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                double fa = f[u][0];                            // the own tile row's operand (wave is uniform: a select, not an index)
#pragma unroll
                for (int t = 1; t < T; ++t) fa = (wave == t) ? f[u][t] : fa;
#pragma unroll
                for (int b = 0; b < T; ++b) acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa, f[u][b], acc[b], 0, 0, 0);
            }
        }
        if constexpr (KP == 64) {
            double* out = Gp + blk * KP * KP;
#pragma unroll
            for (int b = 0; b < T; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) out[(16 * b + r16) * KP + 16 * wave + kc + 4 * r] = acc[b][r];
        }
        return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const i64 wg = blk * 4 + wave;
    const i64 c_begin = wg * cols_per_wave;
    i64 c_end = c_begin + cols_per_wave;
    if (c_end > N) c_end = N;
    f64x4_t acc[T][T];
#pragma unroll
    for (int a = 0; a < T; ++a)
#pragma unroll
        for (int b = 0; b < T; ++b) acc[a][b] = f64x4_t{0.0, 0.0, 0.0, 0.0};
    const int kc = lane >> 4, r16 = lane & 15;
    // 16 columns (4 MFMA k-steps) per trip: all loads of the trip are issued before its MFMAs.  (Round 6 tried keeping the loads of
    // trip i + 1 in flight during the MFMAs of trip i -- the launch is one workgroup per CU and runs at 3 TB/s on 10^6 columns -- and
    // measured it SLOWER: 85 -> 106 us per launch, s_1m 440 -> 431 it/s.  Reverted.)
    for (i64 c0 = c_begin; c0 < c_end; c0 += 16) {
        double f[4][T];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const i64 col = c0 + 4 * u + kc;
            const bool ok = col < c_end;
#pragma unroll
            for (int t = 0; t < T; ++t) f[u][t] = ok ? X[col * KP + 16 * t + r16] : 0.0;
        }
        // only the tile blocks on and above the diagonal (round 6): block (b, a) is the transpose of block (a, b) -- the same products
        // in the same order, bit for bit -- and the fp64 matrix instructions are what this kernel waits for (3 of 4 at KP = 32, 10
        // of 16 at KP = 64)
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int a = 0; a < T; ++a)
#pragma unroll
                for (int b = a; b < T; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[u][a], f[u][b], acc[a][b], 0, 0, 0);
    }
    // deterministic in-block sum of the 4 waves (the blocks below the diagonal are filled from their mirror images)
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int a = 0; a < T; ++a)
#pragma unroll
                for (int b = a; b < T; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * a + kc + 4 * r, colm = 16 * b + r16;
                        const int idx = colm * KP + row;
                        red[idx] = (w == 0) ? acc[a][b][r] : red[idx] + acc[a][b][r];
                        if (a != b) {
                            const int idm = row * KP + colm;
                            red[idm] = (w == 0) ? acc[a][b][r] : red[idm] + acc[a][b][r];
                        }
                    }
        }
        __syncthreads();
    }
    double* out = Gp + blk * KP * KP;
    for (int i = threadIdx.x; i < KP * KP; i += 256) out[i] = red[i];
}

// G[e] = sum_b Gp[b][e]: 16 elements per block, 16 thread groups stride the partials, fixed order
__device__ __forceinline__ void gram_reduce_body(const double* __restrict__ Gp, int nblk, int elems, double* __restrict__ G, int KP,
                                                 double* __restrict__ xscale, double* __restrict__ oscale, double ascale, int blk,
                                                 double (*sh)[17] /* 16 x 17 doubles of LDS */)
{
    const int el = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int e = blk * 16 + el;
    double s = 0.0;
    if (e < elems) {
#pragma unroll 8
        for (int b = g; b < nblk; b += 16) s += Gp[(i64)b * elems + e];
    }
    sh[g][el] = s;
    __syncthreads();
    if (g == 0 && e < elems) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += sh[i][el];
        G[e] = t;
        // fp16 two-term operand: every entry of row r is bounded by sqrt(G_rr); scale the row so that this bound lands
        // in [2^13, 2^14] (fp16 tops out at 65504, its 11-bit precision holds down to 2^-14)
        if (xscale && KP > 0 && e % (KP + 1) == 0) {
            const int r = e / (KP + 1);
            int ex = 0;
            double xs = 1.0;
            if (t > 0.0 && t < 1.0e300) {
                (void)frexp(t, &ex);                    // t = f 2^ex, f in [0.5, 1)  ->  sqrt(t) <= 2^ceil(ex / 2)
                const int half = (ex >= 0) ? (ex + 1) / 2 : -((-ex) / 2);
                xs = ldexp(1.0, 14 - half);
            }
            xscale[r] = xs;
            if (oscale) oscale[r] = 1.0 / (xs * ascale);
        }
    }
}


}  // namespace smk
