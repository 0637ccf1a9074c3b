// smallk_amd/csrc/wide.hip -- ranks above 128 (valid in the reference: k <= n is its only bound, nmf_options.cpp:47-52).
//
// The kernels of kernels.hip / nnls.hip keep a column's k values in a few lanes and the k x k Gram matrix in LDS or in
// registers; neither survives k > 128 (512 KB of Gram matrix at k = 256).  This file is the general path: KP = k rounded
// up to a multiple of 64 (up to 512), one WAVE per column with V = KP / 64 values per lane (element e of a column lives
// in lane e % 64, slot e / 64, so every load of a column or of a Gram row is one coalesced 512-byte line per slot), the
// Gram matrix read through the caches, and a workgroup per column for block principal pivoting (Cholesky of the passive
// block in a global scratch panel).  Same arithmetic as the narrow kernels (reference file:line cited there); built for
// correctness at any rank, not for speed -- the streaming products, which dominate, are the same kernels at every k
// (one pass over A per 64 factor rows).
#include "devutil.h"

#include <cfloat>

namespace smk {

typedef double f64x4_t __attribute__((ext_vector_type(4)));

// --------------------------------------------------------------------------------------------------------------------
// Gram partials: grid (nblk, KP / 16, KP / 64); a workgroup owns tile row a (16 rows of G) x 4 tile columns (64 columns of
// G) for its share of the N columns of X, one v_mfma_f64_16x16x4 per tile and 4 columns of X (as gram_mfma_rows_kernel)
// --------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gram_wide_kernel(const double* __restrict__ X, int KP, i64 N, i64 cols_per_wave,
                                                        double* __restrict__ Gp)
{
    __shared__ double red[16 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int a = blockIdx.y, cb = blockIdx.z;
    const i64 wg = (i64)blockIdx.x * 4 + wave;
    const i64 c_begin = wg * cols_per_wave;
    i64 c_end = c_begin + cols_per_wave;
    if (c_end > N) c_end = N;
    f64x4_t acc[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[b] = f64x4_t{0.0, 0.0, 0.0, 0.0};
    const int kc = lane >> 4, r16 = lane & 15;
    for (i64 c0 = c_begin; c0 < c_end; c0 += 4) {
        const i64 col = c0 + kc;
        const bool ok = col < c_end;
        const double fa = ok ? X[col * KP + 16 * a + r16] : 0.0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const double fb = ok ? X[col * KP + 64 * cb + 16 * b + r16] : 0.0;
            acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa, fb, acc[b], 0, 0, 0);
        }
    }
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = kc + 4 * r, colm = 16 * b + r16;
                    const int idx = colm * 16 + row;
                    red[idx] = (w == 0) ? acc[b][r] : red[idx] + acc[b][r];
                }
        }
        __syncthreads();
    }
    double* out = Gp + (i64)blockIdx.x * KP * KP;
    for (int i = threadIdx.x; i < 16 * 64; i += 256) {
        const int colm = i / 16, row = i % 16;
        out[(i64)(64 * cb + colm) * KP + 16 * a + row] = red[i];
    }
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
    i64 cpw = (N + (i64)nblk * 4 - 1) / ((i64)nblk * 4);
    cpw = (cpw + 15) / 16 * 16;
    gram_wide_kernel<<<dim3(nblk, KP / 16, KP / 64), 256, 0, st>>>(X, KP, N, cpw, scratch);
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
        default: set_error("wide kernels: KP must be 192 .. 512"); return -100; \
    }

int launch_mu_update_wide(double* X, int k, i64 N, PartialView R, const double* G, hipStream_t st)
{
    const int KP = kp_of(k);
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
                        int* grid_out, hipStream_t st)
{
    const int KP = kp_of(k);
    const unsigned grid = (unsigned)((N + 3) / 4);
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
// The passive block G[F,F] is gathered into this workgroup's panel of global scratch (column-major lower triangle, leading
// dimension t = |F|), factored by a right-looking Cholesky (a pivot <= 0 is the reference's "not SPD" failure), and
// solved by forward / back substitution on vectors in LDS; y = G x - r through the caches.  State machine as in
// nnls_bpp_kernel: PBAR = 3, backup rule on the largest index, 5 k pivots at most, 1e-12 zeroing after every exchange.
// --------------------------------------------------------------------------------------------------------------------
constexpr int WIDE_MAX = 512;

__global__ __launch_bounds__(256) void nnls_wide_kernel(double* __restrict__ X, double* __restrict__ Y, int k, int KP, i64 N,
                                                        PartialView R, const double* __restrict__ G,
                                                        int* __restrict__ fail_flag, int iter_tag, i64 col_begin,
                                                        double* __restrict__ scratch)
{
    __shared__ double xs[WIDE_MAX], ys[WIDE_MAX], rs[WIDE_MAX], zs[WIDE_MAX];
    __shared__ int idx[WIDE_MAX];
    __shared__ unsigned char pas[WIDE_MAX], nonopt[WIDE_MAX], infeas[WIDE_MAX];
    __shared__ int s_t, s_ng, s_bad, s_last;
    double* M = scratch + (size_t)blockIdx.x * KP * KP;
    const int tid = threadIdx.x;
    int failed_any = 0;

    for (i64 col = col_begin + blockIdx.x; col < N; col += gridDim.x) {
        for (int e = tid; e < k; e += 256) {
            rs[e] = rhs_elem(R, col, e);
            const double x0 = X[col * KP + e];
            xs[e] = x0;
            pas[e] = x0 > 0.0;                      // passive_set = (X > 0), nnls.hpp:157
        }
        __syncthreads();

        // x_F = G[F,F]^-1 r_F, x elsewhere 0; y = G x - r; then the two violation sets and their size
        auto solve_and_classify = [&](bool zeroize) {
            if (tid == 0) {
                int t = 0;
                for (int e = 0; e < k; ++e)
                    if (pas[e]) idx[t++] = e;
                s_t = t;
                s_bad = 0;
            }
            __syncthreads();
            const int t = s_t;
            for (int q = tid; q < t * t; q += 256) {            // lower triangle, column-major: (i, l), i >= l, at M[l t + i]
                const int i = q % t, l = q / t;
                if (i >= l) M[(size_t)l * t + i] = G[(i64)idx[l] * KP + idx[i]];
            }
            for (int a = tid; a < t; a += 256) zs[a] = rs[idx[a]];
            __syncthreads();
            for (int j = 0; j < t; ++j) {
                const double piv = M[(size_t)j * t + j];
                if (!(piv > 0.0)) {                             // uniform: every thread reads the same entry
                    if (tid == 0) s_bad = 1;
                    break;
                }
                const double d = sqrt(piv), id = 1.0 / d;
                __syncthreads();                                // everyone has read the pivot
                for (int i = j + 1 + tid; i < t; i += 256) M[(size_t)j * t + i] *= id;
                if (tid == 0) M[(size_t)j * t + j] = d;
                __syncthreads();
                const int w = t - j - 1;
                for (int q = tid; q < w * w; q += 256) {
                    const int i = j + 1 + q % w, l = j + 1 + q / w;
                    if (i >= l) M[(size_t)l * t + i] -= M[(size_t)j * t + i] * M[(size_t)j * t + l];
                }
                __syncthreads();
            }
            __syncthreads();
            const bool bad = s_bad != 0;
            if (!bad) {
                for (int j = 0; j < t; ++j) {                   // L z = b
                    const double zj = zs[j] / M[(size_t)j * t + j];
                    __syncthreads();
                    if (tid == 0) zs[j] = zj;
                    for (int i = j + 1 + tid; i < t; i += 256) zs[i] -= M[(size_t)j * t + i] * zj;
                    __syncthreads();
                }
                for (int j = t - 1; j >= 0; --j) {              // L' x = z
                    const double xj = zs[j] / M[(size_t)j * t + j];
                    __syncthreads();
                    if (tid == 0) zs[j] = xj;
                    for (int i = tid; i < j; i += 256) zs[i] -= M[(size_t)i * t + j] * xj;
                    __syncthreads();
                }
            }
            for (int e = tid; e < k; e += 256) xs[e] = 0.0;
            __syncthreads();
            if (!bad)
                for (int a = tid; a < t; a += 256) {
                    double v = zs[a];
                    if (zeroize && fabs(v) < 1.0e-12) v = 0.0;      // ZeroizeSmallValues, nnls.hpp:213,224
                    xs[idx[a]] = v;
                }
            __syncthreads();
            for (int e = tid; e < k; e += 256) {                    // y = G x - r (G symmetric: row idx[a] read along e)
                double acc = 0.0;
                for (int a = 0; a < t; ++a) acc = __builtin_fma(G[(i64)idx[a] * KP + e], xs[idx[a]], acc);
                double y = acc - rs[e];
                if (zeroize && fabs(y) < 1.0e-12) y = 0.0;          // :225
                ys[e] = y;
                nonopt[e] = (!pas[e]) && (y < 0.0);
                infeas[e] = pas[e] && (xs[e] < 0.0);
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
        for (int e = tid; e < k; e += 256) {
            X[col * KP + e] = xs[e];
            if (Y) Y[col * KP + e] = ys[e];
        }
        failed_any |= failed ? 1 : 0;
        __syncthreads();
    }
    if (failed_any && tid == 0) atomicMin(fail_flag, iter_tag);
}

size_t nnls_wide_scratch_elems(int k, int num_cus)
{
    const size_t KP = (size_t)kp_of(k);
    return (size_t)(2 * num_cus) * KP * KP;
}

int launch_nnls_bpp_wide(double* X, double* Y, int k, i64 col_begin, i64 col_end, PartialView R, const double* G, int* fail_flag,
                         int iter_tag, double* scratch, int num_cus, hipStream_t st)
{
    const i64 ncols = col_end - col_begin;
    if (ncols <= 0) return 0;
    if (!scratch) { set_error("nnls: k > 128 needs the scratch panels"); return -100; }
    i64 grid = 2 * (i64)num_cus;
    if (grid > ncols) grid = ncols;
    nnls_wide_kernel<<<(unsigned)grid, 256, 0, st>>>(X, Y, k, kp_of(k), col_end, R, G, fail_flag, iter_tag, col_begin, scratch);
    SMK_HIP(hipGetLastError());
    return 0;
}

}  // namespace smk
