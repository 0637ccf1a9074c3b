// smallk_amd/csrc/wide.hip -- ranks above 128 (valid in the reference: k <= n is its only bound, nmf_options.cpp:47-52).
//
// The kernels of kernels.hip / nnls.hip keep a column's k values in a few lanes and the k x k Gram matrix in LDS or in
// registers; neither survives k > 128 (512 KB of Gram matrix at k = 256).  This file is the general path: KP = k rounded
// up to a multiple of 64 (up to 2048), one WAVE per column with V = KP / 64 values per lane (element e of a column lives
// in lane e % 64, slot e / 64, so every load of a column or of a Gram row is one coalesced 512-byte line per slot), the
// Gram matrix read through the caches.  Block principal pivoting: k <= 256 a wave per column with the block of an exchange
// as 16 x 16 tiles in LDS (blocked Cholesky on the f64 matrix cores), above that a workgroup per column (Cholesky of the
// passive block in LDS or in a global scratch panel).  Same arithmetic as the narrow kernels (reference file:line cited
// there); the streaming products are the same kernels at every k (one pass over A per 64 factor rows).
#include "devutil.h"

#include <cfloat>

namespace smk {

typedef double f64x4_t __attribute__((ext_vector_type(4)));

// --------------------------------------------------------------------------------------------------------------------
// Gram partials: grid (nblk, KP / 64, KP / 64); a workgroup owns one 64 x 64 tile of G for its share of the N columns of X,
// both 64-entry slices of 16 columns at a time through LDS (coalesced 512-byte reads), wave w rows 16 w .. 16 w + 15 of the
// tile on v_mfma_f64_16x16x4.  (Round 2's kernel gave a workgroup a 16 x 64 strip and read its operands straight from
// global memory: 2.7 GB of L2 reads per Gram matrix at k = 512, 250 us; this one reads 1 GB.)
// --------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gram_wide_kernel(const double* __restrict__ X, int KP, i64 N, i64 cols_per_wg,
                                                        double* __restrict__ Gp)
{
    __shared__ double sA[16][64 + 1], sB[16][64 + 1];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int ia = blockIdx.y, ib = blockIdx.z;
    const i64 c_begin = (i64)blockIdx.x * cols_per_wg;
    i64 c_end = c_begin + cols_per_wg;
    if (c_end > N) c_end = N;
    f64x4_t acc[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[b] = f64x4_t{0.0, 0.0, 0.0, 0.0};
    const int cc = tid >> 4, r4 = (tid & 15) * 4;
    for (i64 c0 = c_begin; c0 < c_end; c0 += 16) {
        const i64 col = c0 + cc;
        const bool ok = col < c_end;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            sA[cc][r4 + u] = ok ? X[col * KP + 64 * ia + r4 + u] : 0.0;
            sB[cc][r4 + u] = ok ? X[col * KP + 64 * ib + r4 + u] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const double fa = sA[4 * kk + (lane >> 4)][16 * w + (lane & 15)];
#pragma unroll
            for (int b = 0; b < 4; ++b)
                acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa, sB[4 * kk + (lane >> 4)][16 * b + (lane & 15)], acc[b], 0, 0, 0);
        }
        __syncthreads();
    }
    // D: column = lane & 15, row = (lane >> 4) + 4 reg; G is stored column-major with leading dimension KP
    double* out = Gp + (i64)blockIdx.x * KP * KP;
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            out[(i64)(64 * ib + 16 * b + (lane & 15)) * KP + 64 * ia + 16 * w + (lane >> 4) + 4 * r] = acc[b][r];
}

int gram_wide_blocks(int KP, i64 N, int max_blocks)
{
    i64 cap = ((i64)1 << 24) / ((i64)KP * KP);          // partials stay under 128 MB
    if (cap < 1) cap = 1;
    i64 nblk = (N + 255) / 256;
    if (nblk > cap) nblk = cap;
    if (nblk > max_blocks) nblk = max_blocks;
    if (nblk < 1) nblk = 1;
    return (int)nblk;
}

// partials only; the caller runs gram_reduce_kernel over KP * KP elements
int launch_gram_wide_partials(const double* X, int KP, i64 N, double* scratch, int max_blocks, int* nblk_out, hipStream_t st)
{
    const int nblk = gram_wide_blocks(KP, N, max_blocks);
    i64 cpw = (N + nblk - 1) / nblk;                    // columns per workgroup, in whole chunks of 16
    cpw = (cpw + 15) / 16 * 16;
    gram_wide_kernel<<<dim3(nblk, KP / 64, KP / 64), 256, 0, st>>>(X, KP, N, cpw, scratch);
    SMK_HIP(hipGetLastError());
    *nblk_out = nblk;
    return 0;
}

// --------------------------------------------------------------------------------------------------------------------
// one wave per column, V values per lane
// --------------------------------------------------------------------------------------------------------------------
template <int V>
__device__ __forceinline__ void wide_load(const double* __restrict__ X, i64 col, int KP, int lane, double (&x)[V])
{
#pragma unroll
    for (int v = 0; v < V; ++v) x[v] = X[col * KP + 64 * v + lane];
}
template <int V>
__device__ __forceinline__ void wide_store(double* __restrict__ X, i64 col, int KP, int lane, const double (&x)[V])
{
#pragma unroll
    for (int v = 0; v < V; ++v) X[col * KP + 64 * v + lane] = x[v];
}
template <int V>
__device__ __forceinline__ void wide_rhs(const PartialView& R, i64 col, int k, int lane, double (&b)[V])
{
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const int e = 64 * v + lane;
        b[v] = (e < k && e < R.kpp) ? rhs_elem(R, col, e) : 0.0;
    }
}
// (G x)_r for the column held by this wave; the sum is wave-uniform
template <int V>
__device__ __forceinline__ double wide_dot(const double* __restrict__ Grow, const double (&x)[V], int lane)
{
    double a = 0.0;
#pragma unroll
    for (int v = 0; v < V; ++v) a = __builtin_fma(Grow[64 * v + lane], x[v], a);
    return wave_sum(a);
}

// Y = S M for all the columns of a launch at once, M symmetric k x k, S either the summed partial products R (block pivoting:
// V = R Ginv, i.e. v = Ginv r of every column) or a factor itself (MU and the gradients: Y = X G).  The per-column kernels
// formed these as matrix-vector products, streaming the k x k matrix from L2 for every column -- a quarter of a warm
// block-pivoting iteration at k = 192, most of an MU iteration.  One workgroup per 64 columns x 64 entries, chunks of 16
// along c staged in LDS, f64 matrix cores (wave w: columns 16 w .. 16 w + 15, four 16 x 16 tiles).
// Y[(col - col_begin) * KP + e]; entries e >= k come out as zero when M's padding is zero.
template <bool FROM_VIEW>
__global__ __launch_bounds__(256) void rows_times_sym_wide_kernel(PartialView R, const double* __restrict__ S,
                                                                  const double* __restrict__ M, const int* __restrict__ status, int k,
                                                                  int KP, i64 col_begin, i64 N, double* __restrict__ Y)
{
    __shared__ double sR[16][64 + 1], sG[16][64 + 1];
    if (status && *status == 0) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const i64 c0 = col_begin + (i64)blockIdx.x * 64;
    const int e0 = blockIdx.y * 64;
    f64x4_t acc[4] = {};
    for (int cb = 0; cb < k; cb += 16) {
        {   // S chunk: thread -> column tid / 4, entries 4 (tid % 4) .. + 3
            const i64 col = c0 + (tid >> 2);
            const int cc = (tid & 3) * 4;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                double v = 0.0;
                if (col < N && cb + cc + u < k) {
                    if constexpr (FROM_VIEW) v = (cb + cc + u < R.kpp) ? rhs_elem(R, col, cb + cc + u) : 0.0;
                    else v = S[col * KP + cb + cc + u];
                }
                sR[cc + u][tid >> 2] = v;
            }
            // M chunk: thread -> row tid / 16, entries 4 (tid % 16) .. + 3
            const int gr = tid >> 4, ge = (tid & 15) * 4;
#pragma unroll
            for (int u = 0; u < 4; ++u) sG[gr][ge + u] = (cb + gr < k) ? M[(size_t)(cb + gr) * KP + e0 + ge + u] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const double a = sR[4 * kk + (lane >> 4)][16 * w + (lane & 15)];
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
                acc[jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, sG[4 * kk + (lane >> 4)][16 * jt + (lane & 15)], acc[jt], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const i64 col = c0 + 16 * w + (lane >> 4) + 4 * v;
            if (col < N) Y[(size_t)(col - col_begin) * KP + e0 + 16 * jt + (lane & 15)] = acc[jt][v];
        }
}

// MU: x <- x .* R ./ (G x + 1e-13)   (mu_update_kernel)
template <int V>
__global__ __launch_bounds__(256) void mu_wide_kernel(double* __restrict__ X, int k, int KP, i64 N, PartialView R,
                                                      const double* __restrict__ G)
{
    const int lane = threadIdx.x & 63;
    const i64 j = (i64)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= N) return;
    double x[V], b[V], d[V];
    wide_load<V>(X, j, KP, lane, x);
    wide_rhs<V>(R, j, k, lane, b);
#pragma unroll
    for (int v = 0; v < V; ++v) d[v] = 0.0;
    for (int r = 0; r < k; ++r) {
        const double dot = wide_dot<V>(G + (i64)r * KP, x, lane);
#pragma unroll
        for (int v = 0; v < V; ++v)
            if (v == (r >> 6) && lane == (r & 63)) d[v] = dot;
    }
#pragma unroll
    for (int v = 0; v < V; ++v)
        if (64 * v + lane < k) x[v] = x[v] * (b[v] / (d[v] + 1.0e-13));
    wide_store<V>(X, j, KP, lane, x);
}

// HALS H sweep, Gauss-Seidel over the rows inside the column   (hals_sweep_kernel)
template <int V>
__global__ __launch_bounds__(256) void hals_sweep_wide_kernel(double* __restrict__ X, int k, int KP, i64 N, PartialView R,
                                                              const double* __restrict__ G)
{
    const int lane = threadIdx.x & 63;
    const i64 j = (i64)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= N) return;
    double x[V], b[V];
    wide_load<V>(X, j, KP, lane, x);
    wide_rhs<V>(R, j, k, lane, b);
    for (int r = 0; r < k; ++r) {
        const double* Grow = G + (i64)r * KP;
        const double dot = wide_dot<V>(Grow, x, lane);
        const double grr = Grow[r];
#pragma unroll
        for (int v = 0; v < V; ++v)
            if (v == (r >> 6) && lane == (r & 63)) {
                double t = x[v] + (b[v] - dot) / grr;
                if (isnan(t) || t < 0.0) t = 0.0;
                x[v] = t;
            }
    }
    wide_store<V>(X, j, KP, lane, x);
}

// gradient G x - R (optionally stored) and its projected-gradient partial sum per workgroup   (grad_pg_kernel)
template <int V>
__global__ __launch_bounds__(256) void grad_pg_wide_kernel(const double* __restrict__ X, int k, int KP, i64 N, PartialView R,
                                                           const double* __restrict__ G, double* __restrict__ grad_out,
                                                           double* __restrict__ partials)
{
    __shared__ double sh[16];
    const int lane = threadIdx.x & 63;
    const i64 j = (i64)blockIdx.x * 4 + (threadIdx.x >> 6);
    const bool valid = j < N;
    const i64 jc = valid ? j : (N - 1);
    double x[V], b[V], g[V];
    wide_load<V>(X, jc, KP, lane, x);
    wide_rhs<V>(R, jc, k, lane, b);
#pragma unroll
    for (int v = 0; v < V; ++v) g[v] = 0.0;
    for (int r = 0; r < k; ++r) {
        const double dot = wide_dot<V>(G + (i64)r * KP, x, lane);
#pragma unroll
        for (int v = 0; v < V; ++v)
            if (v == (r >> 6) && lane == (r & 63)) g[v] = dot - b[v];
    }
    double sum = 0.0;
    if (valid) {
#pragma unroll
        for (int v = 0; v < V; ++v)
            if (64 * v + lane < k && (g[v] < 0.0 || x[v] > 0.0)) sum += g[v] * g[v];
        if (grad_out) wide_store<V>(grad_out, j, KP, lane, g);
    }
    const double t = block_sum(sum, sh);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

#define WIDE_DISPATCH(KPV, CALL)                                   \
    switch ((KPV) / 64) {                                          \
        case 3: { constexpr int V = 3; CALL; } break;              \
        case 4: { constexpr int V = 4; CALL; } break;              \
        case 5: { constexpr int V = 5; CALL; } break;              \
        case 6: { constexpr int V = 6; CALL; } break;              \
        case 7: { constexpr int V = 7; CALL; } break;              \
        case 8: { constexpr int V = 8; CALL; } break;              \
        case 9: { constexpr int V = 9; CALL; } break;              \
        case 10: { constexpr int V = 10; CALL; } break;            \
        case 11: { constexpr int V = 11; CALL; } break;            \
        case 12: { constexpr int V = 12; CALL; } break;            \
        case 13: { constexpr int V = 13; CALL; } break;            \
        case 14: { constexpr int V = 14; CALL; } break;            \
        case 15: { constexpr int V = 15; CALL; } break;            \
        case 16: { constexpr int V = 16; CALL; } break;            \
        case 17: { constexpr int V = 17; CALL; } break;            \
        case 18: { constexpr int V = 18; CALL; } break;            \
        case 19: { constexpr int V = 19; CALL; } break;            \
        case 20: { constexpr int V = 20; CALL; } break;            \
        case 21: { constexpr int V = 21; CALL; } break;            \
        case 22: { constexpr int V = 22; CALL; } break;            \
        case 23: { constexpr int V = 23; CALL; } break;            \
        case 24: { constexpr int V = 24; CALL; } break;            \
        case 25: { constexpr int V = 25; CALL; } break;            \
        case 26: { constexpr int V = 26; CALL; } break;            \
        case 27: { constexpr int V = 27; CALL; } break;            \
        case 28: { constexpr int V = 28; CALL; } break;            \
        case 29: { constexpr int V = 29; CALL; } break;            \
        case 30: { constexpr int V = 30; CALL; } break;            \
        case 31: { constexpr int V = 31; CALL; } break;            \
        case 32: { constexpr int V = 32; CALL; } break;            \
        default: set_error("wide kernels: KP must be 192 .. 2048"); return -100; \
    }

// x <- x .* R ./ (y + 1e-13) with y = (X G) of the same column from rows_times_sym_wide_kernel; a thread per entry
__global__ __launch_bounds__(256) void mu_apply_wide_kernel(double* __restrict__ X, int k, int KP, i64 N, PartialView R,
                                                            const double* __restrict__ Y)
{
    const i64 gid = (i64)blockIdx.x * 256 + threadIdx.x;
    const i64 col = gid / KP;
    const int e = (int)(gid % KP);
    if (col >= N || e >= k) return;
    const double b = (e < R.kpp) ? rhs_elem(R, col, e) : 0.0;
    X[col * KP + e] = X[col * KP + e] * (b / (Y[col * KP + e] + 1.0e-13));
}

// gradient y - R (optionally stored) and its projected-gradient partial sum per workgroup, y = (X G) as above
__global__ __launch_bounds__(256) void grad_pg_apply_wide_kernel(const double* __restrict__ X, int k, int KP, i64 N, PartialView R,
                                                                 const double* __restrict__ Y, double* __restrict__ grad_out,
                                                                 double* __restrict__ partials)
{
    __shared__ double sh[16];
    double sum = 0.0;
    for (i64 gid = (i64)blockIdx.x * 256 + threadIdx.x; gid < N * KP; gid += (i64)gridDim.x * 256) {
        const i64 col = gid / KP;
        const int e = (int)(gid % KP);
        double g = 0.0;
        if (e < k) {
            const double b = (e < R.kpp) ? rhs_elem(R, col, e) : 0.0;
            g = Y[gid] - b;
            if (g < 0.0 || X[gid] > 0.0) sum += g * g;
        }
        if (grad_out) grad_out[gid] = g;
    }
    const double t = block_sum(sum, sh);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

int launch_mu_update_wide(double* X, int k, i64 N, PartialView R, const double* G, hipStream_t st, double* tmp)
{
    const int KP = kp_of(k);
    if (tmp) {      // Y = X G on the matrix cores, then the elementwise rule
        rows_times_sym_wide_kernel<false><<<dim3((unsigned)((N + 63) / 64), (unsigned)(KP / 64)), 256, 0, st>>>(PartialView{}, X, G, nullptr, k, KP, 0, N,
                                                                                                         tmp);
        SMK_HIP(hipGetLastError());
        mu_apply_wide_kernel<<<(unsigned)((N * KP + 255) / 256), 256, 0, st>>>(X, k, KP, N, R, tmp);
        SMK_HIP(hipGetLastError());
        return 0;
    }
    const unsigned grid = (unsigned)((N + 3) / 4);
    WIDE_DISPATCH(KP, (mu_wide_kernel<V><<<grid, 256, 0, st>>>(X, k, KP, N, R, G)));
    SMK_HIP(hipGetLastError());
    return 0;
}
int launch_hals_sweep_wide(double* X, int k, i64 N, PartialView R, const double* G, hipStream_t st)
{
    const int KP = kp_of(k);
    const unsigned grid = (unsigned)((N + 3) / 4);
    WIDE_DISPATCH(KP, (hals_sweep_wide_kernel<V><<<grid, 256, 0, st>>>(X, k, KP, N, R, G)));
    SMK_HIP(hipGetLastError());
    return 0;
}
// partial sums land in pg_partials[0 .. *grid_out)
int launch_grad_pg_wide(const double* X, int k, i64 N, PartialView R, const double* G, double* grad_out, double* pg_partials,
                        int* grid_out, hipStream_t st, double* tmp)
{
    const int KP = kp_of(k);
    const unsigned grid = (unsigned)((N + 3) / 4);
    if (tmp) {
        rows_times_sym_wide_kernel<false><<<dim3((unsigned)((N + 63) / 64), (unsigned)(KP / 64)), 256, 0, st>>>(PartialView{}, X, G, nullptr, k, KP, 0, N,
                                                                                                         tmp);
        SMK_HIP(hipGetLastError());
        grad_pg_apply_wide_kernel<<<grid, 256, 0, st>>>(X, k, KP, N, R, tmp, grad_out, pg_partials);   // as many partials as before
        SMK_HIP(hipGetLastError());
        *grid_out = (int)grid;
        return 0;
    }
    WIDE_DISPATCH(KP, (grad_pg_wide_kernel<V><<<grid, 256, 0, st>>>(X, k, KP, N, R, G, grad_out, pg_partials)));
    SMK_HIP(hipGetLastError());
    *grid_out = (int)grid;
    return 0;
}

// --------------------------------------------------------------------------------------------------------------------
// HALS W sweep, one launch per column c of W (hals_w_col_kernel): a wave per row, rows strided over the grid.  Launch c
// first applies the normalisation of column c - 1 (norm from the per-workgroup partials of launch c - 1), then updates
// column c and leaves its partial sum of squares / count of clamped entries in ss[c][blk], nz[c][blk].
// --------------------------------------------------------------------------------------------------------------------
template <int V>
__global__ __launch_bounds__(256) void hals_w_col_wide_kernel(double* __restrict__ Wt, int k, int KP, i64 M, PartialView R,
                                                              const double* __restrict__ G, int c, int nblk,
                                                              double* __restrict__ ss, double* __restrict__ nz)
{
    __shared__ double sh[34];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double scale_prev = 1.0, fill_prev = -1.0;
    if (c > 0) {
        double s2 = 0.0, zc = 0.0;
        for (int t = threadIdx.x; t < nblk; t += blockDim.x) {
            s2 += ss[(i64)(c - 1) * nblk + t];
            zc += nz[(i64)(c - 1) * nblk + t];
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            s2 += __shfl_down(s2, off, 64);
            zc += __shfl_down(zc, off, 64);
        }
        if (lane == 0) { sh[wave] = s2; sh[16 + wave] = zc; }
        __syncthreads();
        s2 = (sh[0] + sh[1]) + (sh[2] + sh[3]);
        zc = (sh[16] + sh[17]) + (sh[18] + sh[19]);
        __syncthreads();
        if (zc >= (double)M) {                      // all-zero column guard (nmf_solver_hals.hpp:105-111)
            const double eps = DBL_EPSILON;
            const double nrm = sqrt((double)M * eps * eps);
            fill_prev = eps * (1.0 / nrm);
        } else {
            scale_prev = 1.0 / sqrt(s2);
        }
    }
    double gc[V];
#pragma unroll
    for (int v = 0; v < V; ++v) gc[v] = (c < k) ? G[(i64)c * KP + 64 * v + lane] : 0.0;
    const double gcc = (c < k) ? G[(i64)c * KP + c] : 1.0;
    double v2 = 0.0, zero = 0.0;
    for (i64 i = (i64)blockIdx.x * 4 + wave; i < M; i += (i64)gridDim.x * 4) {
        double w[V];
        wide_load<V>(Wt, i, KP, lane, w);
        if (c > 0) {
            const int p = c - 1;
#pragma unroll
            for (int v = 0; v < V; ++v)
                if (v == (p >> 6) && lane == (p & 63)) {
                    w[v] = (fill_prev >= 0.0) ? fill_prev : w[v] * scale_prev;
                    Wt[i * KP + p] = w[v];
                }
        }
        if (c < k) {
            double a = 0.0;
#pragma unroll
            for (int v = 0; v < V; ++v) a = __builtin_fma(gc[v], w[v], a);
            const double dot = wave_sum(a);
            const double rhs = rhs_elem(R, i, c);
#pragma unroll
            for (int v = 0; v < V; ++v)
                if (v == (c >> 6) && lane == (c & 63)) {
                    double t = w[v] + (rhs - dot) / gcc;
                    if (isnan(t) || t < 0.0) { t = 0.0; zero += 1.0; }
                    Wt[i * KP + c] = t;
                    v2 += t * t;
                }
        }
    }
    if (c < k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            v2 += __shfl_down(v2, off, 64);
            zero += __shfl_down(zero, off, 64);
        }
        if (lane == 0) { sh[wave] = v2; sh[16 + wave] = zero; }
        __syncthreads();
        if (threadIdx.x == 0) {
            ss[(i64)c * nblk + blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
            nz[(i64)c * nblk + blockIdx.x] = (sh[16] + sh[17]) + (sh[18] + sh[19]);
        }
    }
}

int hals_w_wide_blocks(i64 M)
{
    i64 nblk = (M + 3) / 4;
    if (nblk > 1024) nblk = 1024;
    if (nblk < 1) nblk = 1;
    return (int)nblk;
}

int launch_hals_w_update_wide(double* Wt, int k, i64 M, PartialView R, const double* G, double* scratch, hipStream_t st)
{
    const int KP = kp_of(k), nblk = hals_w_wide_blocks(M);
    double* ss = scratch;
    double* nz = scratch + (i64)k * nblk;
    for (int c = 0; c <= k; ++c) {
        WIDE_DISPATCH(KP, (hals_w_col_wide_kernel<V><<<nblk, 256, 0, st>>>(Wt, k, KP, M, R, G, c, nblk, ss, nz)));
    }
    SMK_HIP(hipGetLastError());
    return 0;
}

// --------------------------------------------------------------------------------------------------------------------
// HALS W sweep for k > 64, by blocks of 16 columns.  The per-column launches above read every row of W in full to form
// w_i . G[:, c] -- k passes over W per sweep (k = 100 on 16384 rows: 101 launches of 14 us, 43 % of the iteration; k = 512:
// 9.8 of 18.5 ms).  Here the dots of a block's 16 columns come from ONE product Y = W G[:, block] on the matrix cores
// (rows_times_cols16_kernel, one pass over W per block), and a column's launch corrects its entry for what changed since:
//     w_i . G[:, c] = Y[i][c - c0] + sum_s Delta[i][s] G[col_s][c],
// Delta[i][0] = the normalisation of column c0 - 1 (applied by launch c0, after the product saw the raw column), Delta[i][1 + q]
// = final minus old value of block column c0 + q (update, then normalisation by the next launch).  A launch is a thread per row
// and touches ~150 bytes per row; the global column norms still cost one launch per column (same update order, same guards,
// same partial sums as hals_w_col_kernel / nmf_solver_hals.hpp:66-117).
// scratch: [ss: k x nblk][nz: k x nblk][Yt: 16 x M][Dl: 17 x M]
// --------------------------------------------------------------------------------------------------------------------
constexpr int HW_NB = 16;

// Yt[q][i] = sum_e Wt[i][e] G[c0 + q][e], q < ncb <= 16 (zero above): 64 rows per workgroup, chunks of 16 along e through LDS
__global__ __launch_bounds__(256) void rows_times_cols16_kernel(const double* __restrict__ Wt, const double* __restrict__ G, int k, int KP,
                                                                int c0, int ncb, i64 M, double* __restrict__ Yt)
{
    __shared__ double sR[16][64 + 1], sG[16][16 + 1];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const i64 r0 = (i64)blockIdx.x * 64;
    f64x4_t acc = {0.0, 0.0, 0.0, 0.0};
    for (int cb = 0; cb < k; cb += 16) {
        {
            const i64 row = r0 + (tid >> 2);
            const int cc = (tid & 3) * 4;
#pragma unroll
            for (int u = 0; u < 4; ++u) sR[cc + u][tid >> 2] = (row < M && cb + cc + u < k) ? Wt[row * KP + cb + cc + u] : 0.0;
            const int ee = tid >> 4, q = tid & 15;
            sG[ee][q] = (q < ncb && cb + ee < k) ? G[(size_t)(c0 + q) * KP + cb + ee] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sR[4 * kk + (lane >> 4)][16 * w + (lane & 15)], sG[4 * kk + (lane >> 4)][lane & 15], acc, 0, 0, 0);
        __syncthreads();
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const i64 row = r0 + 16 * w + (lane >> 4) + 4 * v;
        if (row < M) Yt[(i64)(lane & 15) * M + row] = acc[v];
    }
}

// launch c of the sweep (0 .. k): normalise column c - 1 from its partial sums, then update column c; a thread per row
__global__ __launch_bounds__(256) void hals_w_blk_col_kernel(double* __restrict__ Wt, int k, int KP, i64 M, PartialView R,
                                                             const double* __restrict__ G, int c, int c0, int nblk,
                                                             double* __restrict__ ss, double* __restrict__ nz,
                                                             const double* __restrict__ Yt, double* __restrict__ Dl)
{
    __shared__ double sh[34];
    __shared__ double sg[HW_NB + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double scale_prev = 1.0, fill_prev = -1.0;
    if (c > 0) {                                                // exactly the prologue of hals_w_col_wide_kernel
        double s2 = 0.0, zc = 0.0;
        for (int t = threadIdx.x; t < nblk; t += blockDim.x) {
            s2 += ss[(i64)(c - 1) * nblk + t];
            zc += nz[(i64)(c - 1) * nblk + t];
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            s2 += __shfl_down(s2, off, 64);
            zc += __shfl_down(zc, off, 64);
        }
        if (lane == 0) { sh[wave] = s2; sh[16 + wave] = zc; }
        __syncthreads();
        s2 = (sh[0] + sh[1]) + (sh[2] + sh[3]);
        zc = (sh[16] + sh[17]) + (sh[18] + sh[19]);
        __syncthreads();
        if (zc >= (double)M) {                                  // all-zero column guard (nmf_solver_hals.hpp:105-111)
            const double eps = DBL_EPSILON;
            const double nrm = sqrt((double)M * eps * eps);
            fill_prev = eps * (1.0 / nrm);
        } else {
            scale_prev = 1.0 / sqrt(s2);
        }
    }
    // G[col_s][c] of the columns that changed since the block's product: slot 0 = c0 - 1, slot 1 + q = c0 + q (q < c - c0)
    const int nq = c - c0;                                      // finished block columns before c
    if (c < k && threadIdx.x <= nq) {
        const int col = (threadIdx.x == 0) ? c0 - 1 : c0 + (int)threadIdx.x - 1;
        sg[threadIdx.x] = (col >= 0) ? G[(size_t)col * KP + c] : 0.0;
    }
    __syncthreads();
    const double gcc = (c < k) ? G[(size_t)c * KP + c] : 1.0;
    double v2 = 0.0, zero = 0.0;
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < M; i += (i64)gridDim.x * 256) {
        if (c > 0) {
            const int p = c - 1;
            const double wp = Wt[i * KP + p];
            const double np = (fill_prev >= 0.0) ? fill_prev : wp * scale_prev;
            Wt[i * KP + p] = np;
            if (c < k) {
                if (p < c0) Dl[i] = np - wp;                                    // slot 0: the product saw the raw column c0 - 1
                else Dl[(i64)(1 + p - c0) * M + i] += np - wp;                  // update (stored by launch p) + normalisation
            }
        }
        if (c < k) {
            double dot = Yt[(i64)nq * M + i];
            if (c0 > 0) dot = __builtin_fma(Dl[i], sg[0], dot);
            for (int q = 0; q < nq; ++q) dot = __builtin_fma(Dl[(i64)(1 + q) * M + i], sg[1 + q], dot);
            const double wold = Wt[i * KP + c];
            const double rhs = rhs_elem(R, i, c);
            double t = wold + (rhs - dot) / gcc;
            if (isnan(t) || t < 0.0) { t = 0.0; zero += 1.0; }
            Wt[i * KP + c] = t;
            Dl[(i64)(1 + nq) * M + i] = t - wold;
            v2 += t * t;
        }
    }
    if (c < k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            v2 += __shfl_down(v2, off, 64);
            zero += __shfl_down(zero, off, 64);
        }
        if (lane == 0) { sh[wave] = v2; sh[16 + wave] = zero; }
        __syncthreads();
        if (threadIdx.x == 0) {
            ss[(i64)c * nblk + blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
            nz[(i64)c * nblk + blockIdx.x] = (sh[16] + sh[17]) + (sh[18] + sh[19]);
        }
    }
}

int hals_w_blocked_blocks(i64 M)
{
    i64 nblk = (M + 255) / 256;
    if (nblk > 1024) nblk = 1024;
    if (nblk < 1) nblk = 1;
    return (int)nblk;
}
size_t hals_w_blocked_scratch_elems(int k, i64 M) { return (size_t)2 * k * hals_w_blocked_blocks(M) + (size_t)(2 * HW_NB + 1) * M; }

int launch_hals_w_update_blocked(double* Wt, int k, i64 M, PartialView R, const double* G, double* scratch, hipStream_t st)
{
    const int KP = kp_of(k), nblk = hals_w_blocked_blocks(M);
    double* ss = scratch;
    double* nz = ss + (i64)k * nblk;
    double* Yt = nz + (i64)k * nblk;
    double* Dl = Yt + (i64)HW_NB * M;
    int c0 = 0;
    for (int c = 0; c <= k; ++c) {
        if (c < k && c % HW_NB == 0) {
            c0 = c;
            const int ncb = k - c0 < HW_NB ? k - c0 : HW_NB;
            rows_times_cols16_kernel<<<(unsigned)((M + 63) / 64), 256, 0, st>>>(Wt, G, k, KP, c0, ncb, M, Yt);
        }
        hals_w_blk_col_kernel<<<nblk, 256, 0, st>>>(Wt, k, KP, M, R, G, c, c0, nblk, ss, nz, Yt, Dl);
    }
    SMK_HIP(hipGetLastError());
    return 0;
}

// --------------------------------------------------------------------------------------------------------------------
// sparse gather product: out[:, j] = sum over the stored entries p of column j of val[p] * X[:, row[p]]; a wave per column
// --------------------------------------------------------------------------------------------------------------------
template <int V>
__global__ __launch_bounds__(256) void spmm_gather_wide_kernel(const i64* __restrict__ colptr, const unsigned* __restrict__ rowidx,
                                                               const double* __restrict__ val, i64 ncols,
                                                               const double* __restrict__ X, int KP, double* __restrict__ P, int kpp)
{
    const int lane = threadIdx.x & 63;
    const i64 j = (i64)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= ncols) return;
    double acc[V];
#pragma unroll
    for (int v = 0; v < V; ++v) acc[v] = 0.0;
    const i64 p0 = colptr[j], p1 = colptr[j + 1];
    for (i64 p = p0; p < p1; ++p) {
        const double a = val[p];
        const double* xr = X + (i64)rowidx[p] * KP;
#pragma unroll
        for (int v = 0; v < V; ++v) acc[v] = __builtin_fma(a, xr[64 * v + lane], acc[v]);
    }
#pragma unroll
    for (int v = 0; v < V; ++v)
        if (64 * v + lane < kpp) P[j * kpp + 64 * v + lane] = acc[v];
}

int launch_spmm_gather_wide(const i64* colptr, const unsigned* rowidx, const double* val, i64 ncols, const double* X, int k,
                            double* P, int kpp, hipStream_t st)
{
    const int KP = kp_of(k);
    const unsigned grid = (unsigned)((ncols + 3) / 4);
    WIDE_DISPATCH(KP, (spmm_gather_wide_kernel<V><<<grid, 256, 0, st>>>(colptr, rowidx, val, ncols, X, KP, P, kpp)));
    SMK_HIP(hipGetLastError());
    return 0;
}

// --------------------------------------------------------------------------------------------------------------------
// NNLS by block principal pivoting (nnls.hpp:144-244, src/nnls.cpp:18-74, normal_eq.hpp:27-54), one workgroup per column.
//
// As for k in (32, 128] (nnls.hip) the common work is moved into the inverse of the Gram matrix: Ginv = G^-1 once per
// launch (launch_chol_wide + inv_cols_wide_kernel), V = R Ginv for all columns (rows_times_sym_wide_kernel), and a passive set F is solved either
// directly, G[F,F] x_F = r_F, y = G[:,F] x_F - r, or through its complement Z: y_Z = -(Ginv[Z,Z])^-1 v_Z,
// x = v + Ginv[:,Z] y_Z -- whichever block is smaller.  The block (t = min(|F|, |Z|) rows) is gathered into a panel
// (LDS when t <= 128, else this workgroup's panel of global scratch; the code is the same through generic pointers),
// factored by a right-looking Cholesky and solved by forward / back substitution on an LDS vector.  A pivot <= 0 is the
// reference's "not SPD" failure.  When G itself is not safely invertible (a pivot of its Cholesky below 1e-9 of the
// diagonal) only the direct form is used, which is the reference's own computation.
// State machine as in nnls_bpp_kernel: PBAR = 3, backup rule on the largest index, 5 k pivots at most, 1e-12 zeroing
// after every exchange.
// --------------------------------------------------------------------------------------------------------------------
constexpr int WIDE_MAX = MAX_K;
constexpr int WIDE_TL = 128;                       // largest block kept in LDS (128 KiB)

// in-place Cholesky of the column-major lower triangle Mp (leading dimension t), then L z = b, L' x = z on zs (LDS).
// Every thread of the workgroup calls it; returns true when a pivot is not positive (uniform).
__device__ __forceinline__ bool chol_solve_panel(double* Mp, int t, double* zs, int* s_bad)
{
    const int tid = threadIdx.x, nt = blockDim.x;
    if (tid == 0) *s_bad = 0;
    __syncthreads();
    for (int j = 0; j < t; ++j) {
        const double piv = Mp[(size_t)j * t + j];
        if (!(piv > 0.0)) {                                     // uniform: every thread reads the same entry
            if (tid == 0) *s_bad = 1;
            break;
        }
        const double d = sqrt(piv), id = 1.0 / d;
        __syncthreads();                                        // everyone has read the pivot
        for (int i = j + 1 + tid; i < t; i += nt) Mp[(size_t)j * t + i] *= id;
        if (tid == 0) Mp[(size_t)j * t + j] = d;
        __syncthreads();
        {   // trailing update on a 16 x (nt / 16) thread grid: entry (i, l), j < l <= i < t
            const int ti = tid & 15, tl = tid >> 4, nl = nt >> 4;
            for (int l = j + 1 + tl; l < t; l += nl) {
                const double mlj = Mp[(size_t)j * t + l];
                for (int i = l + ti; i < t; i += 16) Mp[(size_t)l * t + i] -= Mp[(size_t)j * t + i] * mlj;
            }
        }
        __syncthreads();
    }
    __syncthreads();
    if (*s_bad) return true;
    for (int j = 0; j < t; ++j) {                               // L z = b
        const double zj = zs[j] / Mp[(size_t)j * t + j];
        __syncthreads();
        if (tid == 0) zs[j] = zj;
        for (int i = j + 1 + tid; i < t; i += nt) zs[i] -= Mp[(size_t)j * t + i] * zj;
        __syncthreads();
    }
    for (int j = t - 1; j >= 0; --j) {                          // L' x = z
        const double xj = zs[j] / Mp[(size_t)j * t + j];
        __syncthreads();
        if (tid == 0) zs[j] = xj;
        for (int i = tid; i < j; i += nt) zs[i] -= Mp[(size_t)i * t + j] * xj;
        __syncthreads();
    }
    return false;
}

// L = chol(G) (lower, column-major, leading dimension KP) as a sequence of launches, right-looking by panels of CH_NB columns
// (one workgroup walking the whole matrix through L2 took 1.1 ms at k = 192 and 15 ms at k = 512 -- half of a warm
// block-pivoting iteration; this takes 0.2 / 0.7 ms).  Panel J: `chol_panel_kernel` factors the CH_NB x CH_NB diagonal block
// in LDS (every workgroup for itself; workgroup 0 stores it in a side buffer -- not in place, the others may still be reading
// the block -- which `chol_diag_store_kernel` moves into L at the end) and solves its 256 rows of the panel against it;
// `chol_trail_kernel` subtracts the panel's outer product from the trailing lower triangle in 64 x 64 tiles.  *status stays
// non-zero when every pivot exceeds 1e-9 of its diagonal entry (the guard of gram_inverse_kernel); a smaller pivot clears it,
// which every later launch and the block-pivoting kernels read.
constexpr int CH_NB = 32;

__global__ __launch_bounds__(256) void chol_panel_kernel(double* __restrict__ L, const double* __restrict__ G, int k, int KP, int J,
                                                         int* __restrict__ status, double* __restrict__ Dblk)
{
    __shared__ double sD[CH_NB][CH_NB + 1];                     // sD[c][r] = block entry (r, c), r >= c
    __shared__ double sdiag[CH_NB], sguard[CH_NB];
    __shared__ int s_bad;
    if (*status == 0) return;
    const int tid = threadIdx.x;
    const int nb = (k - J < CH_NB) ? (k - J) : CH_NB;
    for (int q = tid; q < CH_NB * CH_NB; q += 256) {
        const int r = q % CH_NB, c = q / CH_NB;
        sD[c][r] = (r < nb && c < nb && r >= c) ? L[(size_t)(J + c) * KP + J + r] : (r == c ? 1.0 : 0.0);
    }
    if (tid == 0) s_bad = 0;
    if (tid < nb) sguard[tid] = 1.0e-9 * G[(size_t)(J + tid) * KP + J + tid];
    __syncthreads();
    const int r = tid % CH_NB, lq = tid / CH_NB;                // 32 rows x 8 column groups
    for (int j = 0; j < nb; ++j) {
        __syncthreads();                                        // the updates of step j - 1 have landed
        const double piv = sD[j][j];
        if (!(piv > sguard[j])) {                               // uniform: every thread reads the same entries
            if (tid == 0) s_bad = 1;
            break;
        }
        double id = __builtin_amdgcn_rsq(piv);                  // 1 / sqrt(piv): hardware estimate + two Newton steps
        id = id * __builtin_fma(-0.5 * piv * id, id, 1.5);
        id = id * __builtin_fma(-0.5 * piv * id, id, 1.5);
        if (lq == 0) {
            if (r > j) sD[j][r] *= id;                          // the diagonal entry itself stays the pivot until the end
            else if (r == j) sdiag[j] = id;
        }
        __syncthreads();
        const double mrj = sD[j][r];
        for (int l = j + 1 + lq; l <= r; l += 8) sD[l][r] -= mrj * sD[j][l];
    }
    __syncthreads();
    if (!s_bad && tid < nb) sD[tid][tid] *= sdiag[tid];         // sqrt(pivot); sdiag keeps 1 / L_jj for the rows below
    __syncthreads();
    if (s_bad) {
        if (blockIdx.x == 0 && tid == 0) *status = 0;
        return;
    }
    // the factored block goes to a side buffer, NOT back into L: the other workgroups of this launch read the unfactored block
    // from L whenever they happen to start (chol_diag_store_kernel moves all blocks into L after the last panel)
    if (blockIdx.x == 0)
        for (int q = tid; q < nb * nb; q += 256) {
            const int rr = q % nb, c = q / nb;
            if (rr >= c) Dblk[(size_t)(J + c) * CH_NB + rr] = sD[c][rr];
        }
    // this thread's row of the panel: x <- x L_JJ^-T
    const int i = J + CH_NB + blockIdx.x * 256 + tid;
    if (i >= k || nb < CH_NB) return;
    double x[CH_NB];
#pragma unroll
    for (int c = 0; c < CH_NB; ++c) x[c] = L[(size_t)(J + c) * KP + i];
#pragma unroll
    for (int c = 0; c < CH_NB; ++c) {
        double a = x[c];
#pragma unroll
        for (int q = 0; q < c; ++q) a = __builtin_fma(-x[q], sD[q][c], a);
        x[c] = a * sdiag[c];
    }
#pragma unroll
    for (int c = 0; c < CH_NB; ++c) L[(size_t)(J + c) * KP + i] = x[c];
}

// trailing update behind panel J: C(i, l) -= sum_c L(i, J + c) L(l, J + c) for i >= l >= J + CH_NB, one 64 x 64 tile per
// workgroup (tiles of the lower triangle only), a 4 x 4 block per thread
__global__ __launch_bounds__(256) void chol_trail_kernel(double* __restrict__ L, int k, int KP, int J, const int* __restrict__ status)
{
    __shared__ double sA[CH_NB][64 + 1], sB[CH_NB][64 + 1];
    if (*status == 0) return;
    const int base = J + CH_NB;
    int ti = 0, b = blockIdx.x;
    while (b > ti) { b -= ti + 1; ++ti; }                       // tile row ti, tile column b <= ti
    const int i0 = base + 64 * ti, l0 = base + 64 * b;
    const int tid = threadIdx.x;
    for (int q = tid; q < CH_NB * 64; q += 256) {
        const int rr = q % 64, c = q / 64;
        sA[c][rr] = (i0 + rr < k) ? L[(size_t)(J + c) * KP + i0 + rr] : 0.0;
        sB[c][rr] = (l0 + rr < k) ? L[(size_t)(J + c) * KP + l0 + rr] : 0.0;
    }
    __syncthreads();
    const int tr = tid % 16, tc = tid / 16;                     // rows tr + 16 u, columns tc + 16 v
    double acc[4][4] = {};
#pragma unroll 8
    for (int c = 0; c < CH_NB; ++c) {
        double a[4], bb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { a[u] = sA[c][tr + 16 * u]; bb[u] = sB[c][tc + 16 * u]; }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[u][v] = __builtin_fma(a[u], bb[v], acc[u][v]);
    }
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + tr + 16 * u, l = l0 + tc + 16 * v;
            if (i < k && l <= i) L[(size_t)l * KP + i] -= acc[u][v];
        }
}

__global__ __launch_bounds__(256) void chol_diag_store_kernel(double* __restrict__ L, const double* __restrict__ Dblk, int k, int KP,
                                                              const int* __restrict__ status)
{
    if (*status == 0) return;
    const int J = blockIdx.x * CH_NB;
    const int nb = (k - J < CH_NB) ? (k - J) : CH_NB;
    for (int q = threadIdx.x; q < nb * nb; q += 256) {
        const int rr = q % nb, c = q / nb;
        if (rr >= c) L[(size_t)(J + c) * KP + J + rr] = Dblk[(size_t)(J + c) * CH_NB + rr];
    }
}

// L = chol(G) on the stream: copy, then two launches per panel
static int launch_chol_wide(const double* G, int k, int KP, double* L, int* status, double* Dblk, hipStream_t st)
{
    SMK_HIP(hipMemcpyAsync(L, G, (size_t)KP * KP * sizeof(double), hipMemcpyDeviceToDevice, st));
    SMK_HIP(hipMemsetAsync(status, 1, sizeof(int), st));         // any non-zero value reads as "invertible"
    for (int J = 0; J < k; J += CH_NB) {
        const int rest = k - J - CH_NB;                         // rows below the diagonal block
        const int g1 = rest > 0 ? (rest + 255) / 256 : 1;
        chol_panel_kernel<<<g1, 256, 0, st>>>(L, G, k, KP, J, status, Dblk);
        if (rest > 0) {
            const int nt = (rest + 63) / 64;
            chol_trail_kernel<<<nt * (nt + 1) / 2, 256, 0, st>>>(L, k, KP, J, status);
        }
    }
    chol_diag_store_kernel<<<(k + CH_NB - 1) / CH_NB, 256, 0, st>>>(L, Dblk, k, KP, status);
    SMK_HIP(hipGetLastError());
    return 0;
}

// column c of Ginv = G^-1 from L: L z = e_c, L' x = z; one workgroup per column
__global__ __launch_bounds__(256) void inv_cols_wide_kernel(const double* __restrict__ L, int k, int KP, double* __restrict__ Ginv,
                                                            const int* __restrict__ status)
{
    __shared__ double z[WIDE_MAX];
    if (*status == 0) return;
    const int c = blockIdx.x, tid = threadIdx.x;
    for (int e = tid; e < k; e += 256) z[e] = (e == c) ? 1.0 : 0.0;
    __syncthreads();
    for (int j = c; j < k; ++j) {                               // z_j = 0 for j < c
        const double zj = z[j] / L[(size_t)j * KP + j];
        __syncthreads();
        if (tid == 0) z[j] = zj;
        for (int i = j + 1 + tid; i < k; i += 256) z[i] -= L[(size_t)j * KP + i] * zj;
        __syncthreads();
    }
    for (int j = k - 1; j >= 0; --j) {
        const double xj = z[j] / L[(size_t)j * KP + j];
        __syncthreads();
        if (tid == 0) z[j] = xj;
        for (int i = tid; i < j; i += 256) z[i] -= L[(size_t)i * KP + j] * xj;
        __syncthreads();
    }
    for (int e = tid; e < KP; e += 256) Ginv[(size_t)c * KP + e] = (e < k) ? z[e] : 0.0;
}

__global__ __launch_bounds__(256) void nnls_wide_kernel(double* __restrict__ X, double* __restrict__ Y, int k, int KP, i64 N,
                                                        PartialView R, const double* __restrict__ G,
                                                        const double* __restrict__ Ginv, const int* __restrict__ status,
                                                        int* __restrict__ fail_flag, int iter_tag, i64 col_begin,
                                                        double* __restrict__ panels, int tl_cap, int skip_if_invertible,
                                                        const double* __restrict__ V)
{
    extern __shared__ __attribute__((aligned(16))) double lds_panel[];          // tl_cap * tl_cap doubles, then the vectors
    if (skip_if_invertible && Ginv != nullptr && *status != 0) return;          // the wave kernel took this launch
    // [panel][xs ys rs zs vs: 5 x KP doubles][idx: KP ints][pas nonopt infeas: 3 x KP bytes] (nnls_wide_lds_bytes)
    double* const xs = lds_panel + (size_t)tl_cap * tl_cap;
    double* const ys = xs + KP;
    double* const rs = ys + KP;
    double* const zs = rs + KP;
    double* const vs = zs + KP;
    int* const idx = (int*)(vs + KP);
    unsigned char* const pas = (unsigned char*)(idx + KP);
    unsigned char* const nonopt = pas + KP;
    unsigned char* const infeas = nonopt + KP;
    __shared__ int s_t, s_ng, s_bad, s_last, s_comp;
    double* gpanel = panels + (size_t)blockIdx.x * KP * KP;
    const int tid = threadIdx.x;
    const bool use_inv = Ginv != nullptr && *status != 0;
    int failed_any = 0;

    for (i64 col = col_begin + blockIdx.x; col < N; col += gridDim.x) {
        for (int e = tid; e < k; e += 256) {
            rs[e] = rhs_elem(R, col, e);
            const double x0 = X[col * KP + e];
            xs[e] = x0;
            pas[e] = x0 > 0.0;                      // passive_set = (X > 0), nnls.hpp:157
        }
        __syncthreads();
        if (use_inv) {                              // v = Ginv r (rows_times_sym_wide_kernel)
            for (int e = tid; e < k; e += 256) vs[e] = V[(size_t)(col - col_begin) * KP + e];
            __syncthreads();
        }

        // one block-pivot solve for the current passive set, then the two violation sets and their size
        auto solve_and_classify = [&](bool zeroize) -> bool {
            if (tid == 0) {
                int p = 0;
                for (int e = 0; e < k; ++e) p += pas[e] ? 1 : 0;
                const int comp = (use_inv && (k - p) <= p) ? 1 : 0;       // the smaller block
                int t = 0;
                for (int e = 0; e < k; ++e)
                    if ((pas[e] != 0) != (comp != 0)) idx[t++] = e;      // direct: the passive ones; complement: the others
                s_t = t;
                s_comp = comp;
            }
            __syncthreads();
            const int t = s_t;
            const bool comp = s_comp != 0;
            const double* Msrc = comp ? Ginv : G;
            double* Mp = (t <= tl_cap) ? lds_panel : gpanel;
#pragma unroll 4
            for (int q = tid; q < t * t; q += 256) {            // lower triangle, column-major: (i, l), i >= l, at Mp[l t + i]
                const int i = q % t, l = q / t;
                if (i >= l) Mp[(size_t)l * t + i] = Msrc[(size_t)idx[l] * KP + idx[i]];
            }
            for (int a = tid; a < t; a += 256) zs[a] = comp ? -vs[idx[a]] : rs[idx[a]];
            __syncthreads();
            const bool bad = (t > 0) ? chol_solve_panel(Mp, t, zs, &s_bad) : false;
            // out = base + Msrc[:, T] u  with base = v (complement) or -r (direct)
            for (int e = tid; e < k; e += 256) {
                double acc = comp ? vs[e] : -rs[e];
                if (!bad) {
                    double a1 = 0.0, a2 = 0.0, a3 = 0.0;
                    int a = 0;
#pragma unroll 2
                    for (; a + 4 <= t; a += 4) {
                        acc = __builtin_fma(Msrc[(size_t)idx[a] * KP + e], zs[a], acc);
                        a1 = __builtin_fma(Msrc[(size_t)idx[a + 1] * KP + e], zs[a + 1], a1);
                        a2 = __builtin_fma(Msrc[(size_t)idx[a + 2] * KP + e], zs[a + 2], a2);
                        a3 = __builtin_fma(Msrc[(size_t)idx[a + 3] * KP + e], zs[a + 3], a3);
                    }
                    for (; a < t; ++a) acc = __builtin_fma(Msrc[(size_t)idx[a] * KP + e], zs[a], acc);
                    acc = (acc + a1) + (a2 + a3);
                }
                double x, y;
                if (comp) { x = pas[e] ? acc : 0.0; y = 0.0; }         // y on Z is u, scattered below
                else      { x = 0.0; y = pas[e] ? 0.0 : acc; }         // x on F is u, scattered below
                xs[e] = x;
                ys[e] = y;
            }
            __syncthreads();
            if (!bad)
                for (int a = tid; a < t; a += 256) {
                    if (comp) ys[idx[a]] = zs[a];
                    else xs[idx[a]] = zs[a];
                }
            __syncthreads();
            for (int e = tid; e < k; e += 256) {
                double x = xs[e], y = ys[e];
                if (zeroize) {                                         // ZeroizeSmallValues, nnls.hpp:213,224-225
                    if (fabs(x) < 1.0e-12) x = 0.0;
                    if (fabs(y) < 1.0e-12) y = 0.0;
                    xs[e] = x;
                    ys[e] = y;
                }
                nonopt[e] = (!pas[e]) && (y < 0.0);
                infeas[e] = pas[e] && (x < 0.0);
            }
            __syncthreads();
            if (tid == 0) {
                int ng = 0, last = -1;
                for (int e = 0; e < k; ++e)
                    if (nonopt[e] || infeas[e]) { ++ng; last = e; }
                s_ng = ng;
                s_last = last;
            }
            __syncthreads();
            return bad;
        };

        bool failed = solve_and_classify(false);
        int ng = s_ng, Pc = 3, Ninf = k + 1, iter = 0;          // PBAR = 3, nnls.hpp:152,170
        const int max_iter = 5 * k;
        while (ng > 0 && !failed) {
            if (iter >= max_iter) { failed = true; break; }
            // UpdatePassiveSet, src/nnls.cpp:18-74
            const bool full = (ng < Ninf) || (Pc >= 1);
            if (ng < Ninf) { Pc = 3; Ninf = ng; }
            else if (Pc >= 1) { Pc -= 1; }
            const int last = s_last;
            __syncthreads();
            if (full) {
                for (int e = tid; e < k; e += 256) {
                    if (nonopt[e]) pas[e] = 1;
                    if (infeas[e]) pas[e] = 0;
                }
            } else if (tid == 0 && last >= 0) {
                pas[last] = !pas[last];                          // backup rule: the largest index in either set
            }
            __syncthreads();
            failed = solve_and_classify(true);
            ng = s_ng;
            ++iter;
        }
        for (int e = tid; e < k; e += 256) {          // columns that never pivot are zeroized too (nnls_bpp_kernel's note)
            const double xo = xs[e], yo = ys[e];
            X[col * KP + e] = fabs(xo) < 1.0e-12 ? 0.0 : xo;
            if (Y) Y[col * KP + e] = fabs(yo) < 1.0e-12 ? 0.0 : yo;
        }
        failed_any |= failed ? 1 : 0;
        __syncthreads();
    }
    if (failed_any && tid == 0) atomicMin(fail_flag, iter_tag);
}

// --------------------------------------------------------------------------------------------------------------------
// The same algorithm with one WAVE per column and no workgroup barrier at all (k <= 256, Gram matrix safely invertible):
// the block solved per exchange has at most k / 2 rows (the smaller of F and Z), its lower triangle is kept PACKED in this
// wave's slice of LDS (t (t + 1) / 2 doubles: 66 KB at t = 128), sets are built with ballots, and every step of the
// Cholesky and of the substitutions is wave-synchronous (LDS operations of one wave execute in order; WAVE_SYNC only stops
// the compiler from moving them).  The workgroup kernel above spends its time in ~7 t barriers per solve; here a solve
// costs its ~t^3 / 6 multiply-adds over 64 lanes plus ~4 t LDS round trips.
// --------------------------------------------------------------------------------------------------------------------
#define WAVE_SYNC() __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront")

#ifdef SMK_WIDE_PROFILE                                        // phase clocks of the wave kernel (diagnostic build only)
__device__ unsigned long long g_wprof[16];
#define WP_NOW() __builtin_amdgcn_s_memtime()
#define WP_ADD(slot, t0) do { const unsigned long long t1_ = WP_NOW(); wp[slot] += t1_ - (t0); (t0) = t1_; } while (0)
#else
#define WP_ADD(slot, t0) do { } while (0)
#endif

// The block of an exchange as 16 x 16 tiles of its lower triangle (tile (I, L), L <= I, at tile_off(I, L); entry (r, c) of a
// tile at c * 17 + r: columns padded to 17 doubles so that both the operand reads of the matrix instruction -- 16 rows of 4
// columns -- and its result layout -- 4 rows of 16 columns -- spread over the LDS banks).  Rows and columns t .. 16 tp - 1
// are those of the identity.
constexpr int TILE_ELEMS = 16 * 17;
__host__ __device__ static inline int tile_count(int tp) { return tp * (tp + 1) / 2; }
__device__ __forceinline__ int tile_off(int I, int L) { return (I * (I + 1) / 2 + L) * TILE_ELEMS; }

// A column is solved by a GROUP of NW waves: NW = 1, a wave per column (four columns per workgroup, no workgroup barrier),
// or NW = 4, the whole workgroup on one column.  group_sync orders the group's LDS / scratch traffic.
template <int NW> __device__ __forceinline__ void group_sync()
{
    if constexpr (NW == 1) WAVE_SYNC();
    else __syncthreads();
}

// T <- the tiles of Msrc[idx, idx] (t rows; tp = ceil(t / 16)); 16 loads per lane in flight, four tiles per wave and round
template <int NW>
__device__ __forceinline__ void tiles_gather(double* __restrict__ T, const double* __restrict__ Msrc, int KP, const int* idx, int t,
                                             int tp, int lane, int wave)
{
    const int r16 = lane & 15, q4 = lane >> 4;
    const int nt = tile_count(tp);
    for (int ti = 4 * wave; ti < nt; ti += 4 * NW) {
        int I = 0, L = ti;
        while (L > I) { L -= I + 1; ++I; }                      // tile ti = (I, L)
        double g[4][4];
        int offs[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bool on = ti + s < nt;
            const int i = 16 * I + r16;
            offs[s] = on ? tile_off(I, L) : -1;
            const int ii = (on && i < t) ? idx[i] : -1;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int l = 16 * L + q4 + 4 * u;
                g[s][u] = (ii >= 0 && l < t) ? Msrc[(size_t)idx[l] * KP + ii] : ((i == l) ? 1.0 : 0.0);
            }
            if (++L > I) { ++I; L = 0; }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
            if (offs[s] >= 0) {
#pragma unroll
                for (int u = 0; u < 4; ++u) T[offs[s] + (q4 + 4 * u) * 17 + r16] = g[s][u];
            }
    }
}

// ---- pieces of the tiled Cholesky (one wave each) ----
// diagonal tile (J, J) in registers: lane r16 holds row r16 (the four lane groups carry copies); leaves 1 / L_jj on the
// diagonal.  Returns true (and stores nothing) when a pivot is not positive.
__device__ __forceinline__ bool tile_factor_diag(double* __restrict__ T, int J, int lane)
{
    const int r16 = lane & 15, q4 = lane >> 4;
    const int ojj = tile_off(J, J);
    double d[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) d[c] = T[ojj + c * 17 + r16];
    bool bad = false;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const double piv = readlane_f64(d[j], j);
        bad |= !(piv > 0.0);                                    // no early exit: what follows a bad pivot is never used
        double id = __builtin_amdgcn_rsq(piv);                  // 1 / sqrt(piv): hardware estimate + two Newton steps
        id = id * __builtin_fma(-0.5 * piv * id, id, 1.5);
        id = id * __builtin_fma(-0.5 * piv * id, id, 1.5);
        d[j] = (r16 == j) ? id : d[j] * id;
#pragma unroll
        for (int l = j + 1; l < 16; ++l) d[l] = __builtin_fma(-d[j], readlane_f64(d[j], l), d[l]);
    }
    if (!bad && q4 == 0) {
#pragma unroll
        for (int c = 0; c < 16; ++c) T[ojj + c * 17 + r16] = d[c];
    }
    return bad;
}
// panel: the rows of tiles (I0 + 0 .. 3, J) <- row L_JJ^-T, one row per lane (lane group q4 takes tile I0 + q4)
__device__ __forceinline__ void tile_panel_rows(double* __restrict__ T, int I0, int J, int tp, int lane)
{
    const int r16 = lane & 15, q4 = lane >> 4;
    const int ojj = tile_off(J, J);
    const int I = I0 + q4;
    const bool on = I < tp;
    const int o = tile_off(on ? I : J + 1, J);
    double x[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) x[c] = T[o + c * 17 + r16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {                              // right-looking: the updates behind a column are independent
        x[c] *= T[ojj + c * 17 + c];
#pragma unroll
        for (int q = c + 1; q < 16; ++q) x[q] = __builtin_fma(-x[c], T[ojj + c * 17 + q], x[q]);
    }
    if (on) {
#pragma unroll
        for (int c = 0; c < 16; ++c) T[o + c * 17 + r16] = x[c];
    }
}
// trailing tile: C(I, L) -= P_I P_L' with P = tile column J, on the f64 matrix cores (A = -P_I: lane l has row l & 15, column
// 4 kk + (l >> 4); the result lane has column l & 15, rows (l >> 4) + 4 v)
__device__ __forceinline__ void tile_trailing(double* __restrict__ T, int I, int L, int J, int lane)
{
    const int r16 = lane & 15, q4 = lane >> 4;
    const int oa = tile_off(I, J), ob = tile_off(L, J), oc = tile_off(I, L);
    f64x4_t acc;
#pragma unroll
    for (int v = 0; v < 4; ++v) acc[v] = T[oc + r16 * 17 + q4 + 4 * v];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-T[oa + (4 * kk + q4) * 17 + r16], T[ob + (4 * kk + q4) * 17 + r16], acc, 0, 0, 0);
#pragma unroll
    for (int v = 0; v < 4; ++v) T[oc + r16 * 17 + q4 + 4 * v] = acc[v];
}

// two trailing tiles at once: all loads first, the two chains of matrix instructions interleaved, then the stores (a block in
// global scratch waits one round trip for the pair instead of one per tile)
__device__ __forceinline__ void tile_trailing2(double* __restrict__ T, int I1, int L1, int I2, int L2, int J, int lane)
{
    const int r16 = lane & 15, q4 = lane >> 4;
    const int oa1 = tile_off(I1, J), ob1 = tile_off(L1, J), oc1 = tile_off(I1, L1);
    const int oa2 = tile_off(I2, J), ob2 = tile_off(L2, J), oc2 = tile_off(I2, L2);
    f64x4_t acc1, acc2;
    double a1[4], b1[4], a2[4], b2[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) { acc1[v] = T[oc1 + r16 * 17 + q4 + 4 * v]; acc2[v] = T[oc2 + r16 * 17 + q4 + 4 * v]; }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        a1[kk] = -T[oa1 + (4 * kk + q4) * 17 + r16]; b1[kk] = T[ob1 + (4 * kk + q4) * 17 + r16];
        a2[kk] = -T[oa2 + (4 * kk + q4) * 17 + r16]; b2[kk] = T[ob2 + (4 * kk + q4) * 17 + r16];
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[kk], b1[kk], acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[kk], b2[kk], acc2, 0, 0, 0);
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) { T[oc1 + r16 * 17 + q4 + 4 * v] = acc1[v]; T[oc2 + r16 * 17 + q4 + 4 * v] = acc2[v]; }
}

// In-place Cholesky of the tiled block, right-looking by tile columns.  Returns true when a pivot is not positive (*s_bad,
// cleared by the caller before the group's last sync, carries that to the other waves).
// NW = 1: diagonal tile, panel, trailing tiles, one after the other.  NW > 1, with look-ahead: behind panel J the tiles of
// column J + 1 are brought up to date first (all waves); then wave 0 factors the NEXT diagonal tile while the other waves
// finish the rest of the trailing triangle (wave 0 joins them for what exceeds its own 16 dependent pivots' worth).
template <int NW>
__device__ __forceinline__ bool tiles_cholesky(double* __restrict__ T, int tp, int lane, int wave, int* s_bad)
{
    if constexpr (NW == 1) {
        for (int J = 0; J < tp; ++J) {
            if (tile_factor_diag(T, J, lane)) return true;
            WAVE_SYNC();
            if (J + 1 == tp) break;
            for (int I0 = J + 1; I0 < tp; I0 += 4) tile_panel_rows(T, I0, J, tp, lane);
            WAVE_SYNC();
            for (int I = J + 1; I < tp; ++I)
                for (int L = J + 1; L <= I; ++L) tile_trailing(T, I, L, J, lane);
            WAVE_SYNC();
        }
        return false;
    } else {
        constexpr int AHEAD = 12 * (NW - 1);                    // trailing tiles the other waves take while wave 0 factors (~12 each)
        if (tp > 0 && wave == 0 && tile_factor_diag(T, 0, lane) && lane == 0) *s_bad = 1;
        __syncthreads();
        if (*s_bad) return true;                                // uniform over the group
        for (int J = 0; J + 1 < tp; ++J) {
            for (int I0 = J + 1 + 4 * wave; I0 < tp; I0 += 4 * NW) tile_panel_rows(T, I0, J, tp, lane);
            __syncthreads();
            for (int I = J + 1 + wave; I < tp; I += NW) tile_trailing(T, I, J + 1, J, lane);     // tile column J + 1 first
            __syncthreads();
            if (wave == 0 && tile_factor_diag(T, J + 1, lane) && lane == 0) *s_bad = 1;
            {   // the rest of the trailing triangle: tiles (I, L), J + 2 <= L <= I, a wave's tiles two at a time
                int p = 0, pI = -1, pL = -1;
                for (int I = J + 2; I < tp; ++I)
                    for (int L = J + 2; L <= I; ++L, ++p) {
                        const int owner = (p < AHEAD) ? 1 + p % (NW - 1) : (p - AHEAD) % NW;
                        if (owner != wave) continue;
                        if (pI < 0) { pI = I; pL = L; }
                        else { tile_trailing2(T, pI, pL, I, L, J, lane); pI = -1; }
                    }
                if (pI >= 0) tile_trailing(T, pI, pL, J, lane);
            }
            __syncthreads();
            if (*s_bad) return true;
        }
        return false;
    }
}

// zs <- (L L')^-1 zs on the factored tiles (zs has 16 tp entries; those beyond t are solved against the identity rows); one wave
__device__ __forceinline__ void tiles_solve(const double* __restrict__ T, double* __restrict__ zs, int tp, int lane)
{
    const int r16 = lane & 15, q4 = lane >> 4;
    for (int I = 0; I < tp; ++I) {                              // L z = b
        double part = 0.0;
        for (int L = 0; L < I; ++L) {
            const int o = tile_off(I, L);
#pragma unroll
            for (int u = 0; u < 4; ++u) part = __builtin_fma(T[o + (q4 + 4 * u) * 17 + r16], zs[16 * L + q4 + 4 * u], part);
        }
        part += __shfl_xor(part, 16);
        part += __shfl_xor(part, 32);
        double y = zs[16 * I + r16] - part;
        const int od = tile_off(I, I);
        double d[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) d[c] = T[od + c * 17 + r16];       // row r16 of the diagonal tile
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const double zj = readlane_f64(y, j) * readlane_f64(d[j], j);
            y = (r16 == j) ? zj : ((r16 > j) ? __builtin_fma(-d[j], zj, y) : y);
        }
        WAVE_SYNC();
        if (q4 == 0) zs[16 * I + r16] = y;
        WAVE_SYNC();
    }
    for (int I = tp - 1; I >= 0; --I) {                         // L' x = z; lane r16 is a COLUMN here
        double part = 0.0;
        for (int L = I + 1; L < tp; ++L) {
            const int o = tile_off(L, I);
#pragma unroll
            for (int u = 0; u < 4; ++u) part = __builtin_fma(T[o + r16 * 17 + q4 + 4 * u], zs[16 * L + q4 + 4 * u], part);
        }
        part += __shfl_xor(part, 16);
        part += __shfl_xor(part, 32);
        double y = zs[16 * I + r16] - part;
        const int od = tile_off(I, I);
        double dc[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) dc[r] = T[od + r16 * 17 + r];      // column r16 of the diagonal tile
#pragma unroll
        for (int j = 15; j >= 0; --j) {
            const double xj = readlane_f64(y, j) * readlane_f64(dc[j], j);
            y = (r16 == j) ? xj : ((r16 < j) ? __builtin_fma(-dc[j], xj, y) : y);
        }
        WAVE_SYNC();
        if (q4 == 0) zs[16 * I + r16] = y;
        WAVE_SYNC();
    }
}

// gather + factor + solve of one exchange on the tiles at T (LDS or the group's panel of global scratch); zs in, zs out
template <int NW>
__device__ __forceinline__ bool tiles_exchange(double* __restrict__ T, const double* __restrict__ Msrc, int KP, const int* idx, int t,
                                               double* __restrict__ zs, int lane, int wave, int* s_bad
#ifdef SMK_WIDE_PROFILE
                                               , unsigned long long* wp, unsigned long long& tp
#endif
)
{
    const int tpx = (t + 15) / 16;
    tiles_gather<NW>(T, Msrc, KP, idx, t, tpx, lane, wave);
    group_sync<NW>();
    WP_ADD(2, tp);
    const bool bad = tiles_cholesky<NW>(T, tpx, lane, wave, s_bad);
    WP_ADD(3, tp);
    if (!bad && (NW == 1 || wave == 0)) tiles_solve(T, zs, tpx, lane);
    group_sync<NW>();
    WP_ADD(4, tp);
    return bad;
}

// per-group LDS slice: [tiles: tile_count(tp_lds)][xs ys rs zs vs: 5 x kq doubles][idx: kq ints][pas: kq bytes][8 ints], kq = k
// rounded up to 64.  Blocks of more than 16 tp_lds rows go to the workgroup's panel of global scratch (NW = 4 only).
static inline size_t tile_slice_bytes(int k, int tp_lds)
{
    const int kq = (k + 63) / 64 * 64;
    return ((size_t)tile_count(tp_lds) * TILE_ELEMS * 8 + (size_t)5 * kq * 8 + (size_t)kq * 4 + (size_t)kq + 32 + 15) / 16 * 16;
}

template <int NW>
__global__ __launch_bounds__(256) void nnls_wide_tile_kernel(double* __restrict__ X, double* __restrict__ Y, int k, int KP, i64 N,
                                                             PartialView R, const double* __restrict__ G,
                                                             const double* __restrict__ Ginv, const int* __restrict__ status,
                                                             int* __restrict__ fail_flag, int iter_tag, i64 col_begin,
                                                             int slice_bytes, const double* __restrict__ V, int tp_lds,
                                                             double* __restrict__ panels)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char wave_lds[];
    if (*status == 0) return;                                   // nnls_wide_kernel takes this launch
    const int lane = threadIdx.x & 63;
    const int wave = NW == 1 ? 0 : (int)(threadIdx.x >> 6);     // this wave inside its group
    const int grp = NW == 1 ? (int)(threadIdx.x >> 6) : 0, ngrp = NW == 1 ? (int)(blockDim.x >> 6) : 1;
    const int gt = NW == 1 ? lane : (int)threadIdx.x;           // thread inside the group
    constexpr int GS = 64 * NW;
    const int kq = (k + 63) / 64 * 64;
    unsigned char* base = wave_lds + (size_t)grp * slice_bytes;
    double* Tl = (double*)base;
    double* xs = Tl + tile_count(tp_lds) * TILE_ELEMS;
    double* ys = xs + kq;
    double* rs = ys + kq;
    double* zs = rs + kq;
    double* vs = zs + kq;
    int* idx = (int*)(vs + kq);
    unsigned char* pas = (unsigned char*)(idx + kq);
    int* sc = (int*)(pas + kq);                                 // t, comp, ng, last, bad (NW > 1: wave 0 decides, all read)
    double* Tg = panels ? panels + (size_t)blockIdx.x * KP * KP : nullptr;
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    int failed_any = 0;
#ifdef SMK_WIDE_PROFILE
    unsigned long long wp[12] = {};
    unsigned long long tp = WP_NOW();
#endif

    for (i64 col = col_begin + (i64)blockIdx.x * ngrp + grp; col < N; col += (i64)gridDim.x * ngrp) {
#ifdef SMK_WIDE_PROFILE
        wp[8] += 1;
        tp = WP_NOW();
#endif
        for (int e = gt; e < kq; e += GS) {
            const bool in = e < k;
            const double x0 = in ? X[col * KP + e] : 0.0;
            rs[e] = in ? rhs_elem(R, col, e) : 0.0;
            vs[e] = in ? V[(size_t)(col - col_begin) * KP + e] : 0.0;      // v = Ginv r (rows_times_sym_wide_kernel)
            xs[e] = x0;
            pas[e] = in && x0 > 0.0;                            // passive_set = (X > 0), nnls.hpp:157
        }
        group_sync<NW>();
        WP_ADD(0, tp);

        int ng = 0, last = -1;
        // one block-pivot solve for the current passive set; leaves xs, ys and the violation count / largest violator
        auto solve_and_classify = [&](bool zeroize) -> bool {
            int t = 0;
            bool comp = false;
            if (NW == 1 || wave == 0) {
                int p = 0;
                for (int e0 = 0; e0 < kq; e0 += 64) p += __popcll(__ballot(pas[e0 + lane] != 0));
                comp = (k - p) <= p;                            // the smaller block
                for (int e0 = 0; e0 < kq; e0 += 64) {
                    const int e = e0 + lane;
                    const bool sel = (e < k) && ((pas[e] != 0) != comp);
                    const unsigned long long m = __ballot(sel);
                    if (sel) idx[t + __popcll(m & lt_mask)] = e;
                    t += __popcll(m);
                }
                if (NW > 1 && lane == 0) { sc[0] = t; sc[1] = comp ? 1 : 0; sc[4] = 0; }
            }
            group_sync<NW>();
            if constexpr (NW > 1) { t = sc[0]; comp = sc[1] != 0; }
#ifdef SMK_WIDE_PROFILE
            wp[9] += 1;
            wp[10] += t;
#endif
            WP_ADD(1, tp);
            const double* Msrc = comp ? Ginv : G;
            const int tpx = (t + 15) / 16;
            for (int a = gt; a < 16 * tpx; a += GS) zs[a] = (a < t) ? (comp ? -vs[idx[a]] : rs[idx[a]]) : 0.0;
            bool bad;
#ifdef SMK_WIDE_PROFILE
            if (tpx <= tp_lds) bad = tiles_exchange<NW>(Tl, Msrc, KP, idx, t, zs, lane, wave, sc + 4, wp, tp);
            else { wp[11] += 1; bad = tiles_exchange<NW>(Tg, Msrc, KP, idx, t, zs, lane, wave, sc + 4, wp, tp); }
#else
            if (tpx <= tp_lds) bad = tiles_exchange<NW>(Tl, Msrc, KP, idx, t, zs, lane, wave, sc + 4);
            else bad = tiles_exchange<NW>(Tg, Msrc, KP, idx, t, zs, lane, wave, sc + 4);
#endif
            // out = base + Msrc[:, T] u  with base = v (complement) or -r (direct); four entries per thread and trip: one trip up to
            // kq = 4 GS (a wave per column: k <= 256; the workgroup on a column: k <= 1024), more above (round 5: the single trip left
            // the entries e >= 1024 of a column untouched -- the defect that capped block pivoting at k = 1024)
            for (int eo = 0; eo < kq; eo += 4 * GS) {
                const int gte = gt + eo;                        // this thread's first entry of the trip
                double acc[4][2] = {};
                if (!bad) {
                    int a = 0;
                    if (kq <= 2 * GS) {        // (eo == 0 here)                         // at most two entries per thread: 16 rows' loads in flight
                        for (; a + 16 <= t; a += 16) {
                            double g[2][16];
#pragma unroll
                            for (int u = 0; u < 16; ++u) {
                                const size_t row = (size_t)idx[a + u] * KP + gte;
#pragma unroll
                                for (int v = 0; v < 2; ++v) g[v][u] = (gte + GS * v < kq) ? Msrc[row + GS * v] : 0.0;
                            }
#pragma unroll
                            for (int u = 0; u < 16; ++u) {
                                const double za = zs[a + u];
#pragma unroll
                                for (int v = 0; v < 2; ++v) acc[v][u & 1] = __builtin_fma(g[v][u], za, acc[v][u & 1]);
                            }
                        }
                    }
                    for (; a + 8 <= t; a += 8) {
                        double g[4][8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const size_t row = (size_t)idx[a + u] * KP + gte;
#pragma unroll
                            for (int v = 0; v < 4; ++v) g[v][u] = (gte + GS * v < kq) ? Msrc[row + GS * v] : 0.0;
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const double za = zs[a + u];
#pragma unroll
                            for (int v = 0; v < 4; ++v) acc[v][u & 1] = __builtin_fma(g[v][u], za, acc[v][u & 1]);
                        }
                    }
                    for (; a < t; ++a) {
                        const size_t row = (size_t)idx[a] * KP + gte;
                        const double za = zs[a];
#pragma unroll
                        for (int v = 0; v < 4; ++v)
                            if (gte + GS * v < kq) acc[v][0] = __builtin_fma(Msrc[row + GS * v], za, acc[v][0]);
                    }
                }
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int e = gte + GS * v;
                    if (e < kq) {
                        const double o = (comp ? vs[e] : -rs[e]) + (acc[v][0] + acc[v][1]);
                        const bool pe = pas[e] != 0;
                        xs[e] = comp ? (pe ? o : 0.0) : 0.0;    // the block's own entries are scattered below
                        ys[e] = comp ? 0.0 : (pe ? 0.0 : o);
                    }
                }
            }
            group_sync<NW>();
            WP_ADD(5, tp);
            if (!bad)
                for (int a = gt; a < t; a += GS) {
                    if (comp) ys[idx[a]] = zs[a];
                    else xs[idx[a]] = zs[a];
                }
            group_sync<NW>();
            ng = 0;
            last = -1;
            if (NW == 1 || wave == 0) {
                for (int e0 = 0; e0 < kq; e0 += 64) {
                    const int e = e0 + lane;
                    double x = xs[e], y = ys[e];
                    if (zeroize) {                              // ZeroizeSmallValues, nnls.hpp:213,224-225
                        if (fabs(x) < 1.0e-12) x = 0.0;
                        if (fabs(y) < 1.0e-12) y = 0.0;
                        xs[e] = x;
                        ys[e] = y;
                    }
                    const bool pe = pas[e] != 0, in = e < k;
                    const bool viol = in && ((!pe && y < 0.0) || (pe && x < 0.0));
                    const unsigned long long m = __ballot(viol);
                    if (m) { ng += __popcll(m); last = e0 + 63 - __clzll(m); }
                    // remember the violation in bit 1 of the mask byte for the exchange below
                    pas[e] = (unsigned char)((pe ? 1 : 0) | (viol ? 2 : 0));
                }
                if (NW > 1 && lane == 0) { sc[2] = ng; sc[3] = last; }
            }
            group_sync<NW>();
            if constexpr (NW > 1) { ng = sc[2]; last = sc[3]; }
            WP_ADD(6, tp);
            return bad;
        };
        // NOTE: after solve_and_classify pas[e] carries bit 0 = passive, bit 1 = violates; the (pas[e] != 0) tests above see
        // only clean bytes because every exchange below rewrites them to 0 / 1 first.

        bool failed = solve_and_classify(false);
        int Pc = 3, Ninf = k + 1, iter = 0;                     // PBAR = 3, nnls.hpp:152,170
        const int max_iter = 5 * k;
        while (ng > 0 && !failed) {
            if (iter >= max_iter) { failed = true; break; }
            // UpdatePassiveSet, src/nnls.cpp:18-74
            const bool full = (ng < Ninf) || (Pc >= 1);
            if (ng < Ninf) { Pc = 3; Ninf = ng; }
            else if (Pc >= 1) { Pc -= 1; }
            if (full) {
                for (int e = gt; e < kq; e += GS) {
                    const unsigned char b = pas[e];
                    pas[e] = (b & 2) ? (unsigned char)((b & 1) ^ 1) : (unsigned char)(b & 1);   // violators change side
                }
            } else {
                for (int e = gt; e < kq; e += GS) {
                    const unsigned char b = pas[e];
                    pas[e] = (e == last) ? (unsigned char)((b & 1) ^ 1) : (unsigned char)(b & 1);   // backup rule: the largest violator
                }
            }
            group_sync<NW>();
            failed = solve_and_classify(true);
            ++iter;
        }
        for (int e = gt; e < k; e += GS) {            // columns that never pivot are zeroized too (nnls_bpp_kernel's note)
            const double xo = xs[e], yo = ys[e];
            X[col * KP + e] = fabs(xo) < 1.0e-12 ? 0.0 : xo;
            if (Y) Y[col * KP + e] = fabs(yo) < 1.0e-12 ? 0.0 : yo;
        }
        failed_any |= failed ? 1 : 0;
        group_sync<NW>();
        WP_ADD(7, tp);
    }
#ifdef SMK_WIDE_PROFILE
    if (lane == 0 && wave == 0)
        for (int q = 0; q < 12; ++q) atomicAdd(&g_wprof[q], wp[q]);
#endif
    if (failed_any && gt == 0) atomicMin(fail_flag, iter_tag);
}

#ifdef SMK_WIDE_PROFILE
extern "C" int smk_debug_wide_profile(unsigned long long* out)
{
    unsigned long long zero[16] = {};
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wprof), sizeof(zero)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_wprof), zero, sizeof(zero)) == hipSuccess ? 0 : -1;
}
#endif

// workgroups per CU: the LDS panel only has to hold the smaller of a passive set and its complement, <= k / 2 rows
static inline int nnls_wide_vec_bytes(int k) { return kp_of(k) * (5 * 8 + 4 + 3); }
static inline int nnls_wide_tl(int k)
{
    const int h = (k + 1) / 2;
    int cap = WIDE_TL;                                                       // 128 rows = 128 KiB beside <= 24 KiB of vectors
    while (cap * cap * 8 + nnls_wide_vec_bytes(k) > 156 * 1024) cap -= 8;    // k > 512: 112 rows
    return h < cap ? h : cap;
}
static inline int nnls_wide_lds_bytes(int k) { return nnls_wide_tl(k) * nnls_wide_tl(k) * 8 + nnls_wide_vec_bytes(k); }
static inline int nnls_wide_wgs_per_cu(int k)
{
    const int lds = nnls_wide_lds_bytes(k) + 1024;
    const int w = (160 * 1024) / lds;
    return w < 1 ? 1 : (w > 2 ? 2 : w);
}
// scratch: [panels: wgs x KP x KP][L: KP x KP][Ginv: KP x KP][status: 8][diagonal blocks of L: KP x 32][V: ncols x KP]
size_t nnls_wide_scratch_elems(int k, int num_cus, i64 ncols)
{
    const size_t KP = (size_t)kp_of(k);
    return ((size_t)num_cus * nnls_wide_wgs_per_cu(k) + 2) * KP * KP + 8 + KP * CH_NB + (size_t)ncols * KP;
}

// SMK_NNLS_INV=0: the direct form only (nnls_wide_kernel: the reference's own computation on the passive block)
static inline bool wide_use_inverse()
{
    static const int mode = [] { const char* e = getenv("SMK_NNLS_INV"); return e ? atoi(e) : 1; }();
    return mode != 0;
}

int launch_gram_inverse_wide(const double* G, int k, double* scratch, int num_cus, hipStream_t st)
{
    if (!scratch || !wide_use_inverse()) return 0;
    const int KP = kp_of(k);
    double* L = scratch + (size_t)num_cus * nnls_wide_wgs_per_cu(k) * KP * KP;
    double* Ginv = L + (size_t)KP * KP;
    int* status = (int*)(Ginv + (size_t)KP * KP);
    double* Dblk = Ginv + (size_t)KP * KP + 8;
    if (launch_chol_wide(G, k, KP, L, status, Dblk, st)) return -100;
    inv_cols_wide_kernel<<<k, 256, 0, st>>>(L, k, KP, Ginv, status);
    SMK_HIP(hipGetLastError());
    return 0;
}

int launch_nnls_bpp_wide(double* X, double* Y, int k, i64 col_begin, i64 col_end, PartialView R, const double* G, int* fail_flag,
                         int iter_tag, double* scratch, int inverse_ready, int num_cus, hipStream_t st)
{
    const i64 ncols = col_end - col_begin;
    if (ncols <= 0) return 0;
    if (!scratch) { set_error("nnls: k > 128 needs the scratch panels"); return -100; }
    const int KP = kp_of(k);
    const int wgs = num_cus * nnls_wide_wgs_per_cu(k), tl = nnls_wide_tl(k);
    double* L = scratch + (size_t)wgs * KP * KP;
    double* Ginv = L + (size_t)KP * KP;
    int* status = (int*)(Ginv + (size_t)KP * KP);
    double* V = Ginv + (size_t)KP * KP + 8 + (size_t)KP * CH_NB; // ncols x KP: the caller sized the scratch for its columns
    const bool use_inv = wide_use_inverse();
    if (use_inv) {
        if (!inverse_ready && launch_gram_inverse_wide(G, k, scratch, num_cus, st)) return -100;
        rows_times_sym_wide_kernel<true><<<dim3((unsigned)((ncols + 63) / 64), (unsigned)(KP / 64)), 256, 0, st>>>(R, nullptr, Ginv, status, k, KP,
                                                                                                              col_begin, col_end, V);
        SMK_HIP(hipGetLastError());
    }
    const int lds = nnls_wide_lds_bytes(k);
    static std::atomic<unsigned long long> attr_set{0};       // per device (DeviceOnce)
    if (DeviceOnce once{attr_set}) {
        SMK_HIP(hipFuncSetAttribute((const void*)nnls_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
        once.done();
    }
    // G invertible: the tile kernel -- a wave per column with the block's tiles in LDS where four values per lane cover a
    // column (k <= 256) and SMK_WIDE_NW=1 asks for it, else the workgroup on one column with as many tile rows in LDS as fit
    // and larger blocks in the workgroup's panel of global scratch.  nnls_wide_kernel below runs only when G is not safely
    // invertible (direct form only: the reference's own computation, including its "not SPD" failure).
    int took_wave = 0;
    if (use_inv) {
        static const int nw_env = [] { const char* e = getenv("SMK_WIDE_NW"); return e ? atoi(e) : 0; }();
        const int tp_full = ((k + 1) / 2 + 15) / 16;
        // measured, 12 iterations at 16384 x 8192: k = 160 4.4 ms per iteration with a wave per column (three columns per CU)
        // against 5.0 with the workgroup on one; k = 192 6.2 against 5.8; k = 256 13.7 against 11.0
        const bool fits3 = k <= 256 && tile_slice_bytes(k, tp_full) * 3 <= 156 * 1024;
        const bool one_wave = (nw_env == 1 && k <= 256) || (nw_env == 0 && fits3);
        static std::atomic<unsigned long long> attr_tile{0};       // per device (DeviceOnce)
        if (DeviceOnce once{attr_tile}) {
            SMK_HIP(hipFuncSetAttribute((const void*)nnls_wide_tile_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
            SMK_HIP(hipFuncSetAttribute((const void*)nnls_wide_tile_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
            once.done();
        }
        if (one_wave) {
            const size_t slice = tile_slice_bytes(k, tp_full);
            int waves = (int)((156 * 1024) / slice);
            if (waves > 4) waves = 4;
            i64 g2 = (ncols + waves - 1) / waves;
            if (g2 > (i64)num_cus * 2) g2 = (i64)num_cus * 2;
            nnls_wide_tile_kernel<1><<<(unsigned)g2, 64 * waves, (unsigned)(slice * waves), st>>>(X, Y, k, KP, col_end, R, G, Ginv, status, fail_flag,
                                                                                              iter_tag, col_begin, (int)slice, V, tp_full,
                                                                                              nullptr);
        } else {
            int tp_lds = tp_full, per_cu = 1;
            if (tile_slice_bytes(k, tp_full) <= 78 * 1024) per_cu = 2;             // whole blocks in LDS, two columns per CU
            else
                while (tile_slice_bytes(k, tp_lds) > 156 * 1024) --tp_lds;
            const size_t slice = tile_slice_bytes(k, tp_lds);
            i64 g4 = (i64)num_cus * per_cu;
            if (tp_lds < tp_full && g4 > wgs) g4 = wgs;                           // one scratch panel per workgroup
            if (g4 > ncols) g4 = ncols;
            nnls_wide_tile_kernel<4><<<(unsigned)g4, 256, (unsigned)slice, st>>>(X, Y, k, KP, col_end, R, G, Ginv, status, fail_flag, iter_tag,
                                                                            col_begin, (int)slice, V, tp_lds,
                                                                            tp_lds < tp_full ? scratch : nullptr);
        }
        SMK_HIP(hipGetLastError());
        took_wave = 1;
    }
    i64 grid = wgs;                                              // as many workgroups as their LDS panels let be resident
    if (grid > ncols) grid = ncols;
    nnls_wide_kernel<<<(unsigned)grid, 256, lds, st>>>(X, Y, k, KP, col_end, R, G, use_inv ? Ginv : nullptr, status, fail_flag,
                                                       iter_tag, col_begin, scratch, tl, took_wave, V);
    SMK_HIP(hipGetLastError());
    return 0;
}

}  // namespace smk
