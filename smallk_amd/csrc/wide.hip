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
        case 9: { constexpr int V = 9; CALL; } break;              \
        case 10: { constexpr int V = 10; CALL; } break;            \
        case 11: { constexpr int V = 11; CALL; } break;            \
        case 12: { constexpr int V = 12; CALL; } break;            \
        case 13: { constexpr int V = 13; CALL; } break;            \
        case 14: { constexpr int V = 14; CALL; } break;            \
        case 15: { constexpr int V = 15; CALL; } break;            \
        case 16: { constexpr int V = 16; CALL; } break;            \
        default: set_error("wide kernels: KP must be 192 .. 1024"); return -100; \
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
//
// As for k in (32, 128] (nnls.hip) the common work is moved into the inverse of the Gram matrix: Ginv = G^-1 once per
// launch (chol_wide_kernel + inv_cols_wide_kernel), v = Ginv r per column, and a passive set F is solved either
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

// L = chol(G) (lower, column-major, leading dimension KP) by one workgroup; status = 1 when every pivot exceeds 1e-9 of
// its diagonal entry (the guard of gram_inverse_kernel), else 0
__global__ __launch_bounds__(1024) void chol_wide_kernel(const double* __restrict__ G, int k, int KP, double* __restrict__ L,
                                                         int* __restrict__ status)
{
    __shared__ int bad;
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int q = tid; q < k * k; q += nt) {
        const int i = q % k, l = q / k;
        if (i >= l) L[(size_t)l * KP + i] = G[(size_t)l * KP + i];
    }
    if (tid == 0) bad = 0;
    __syncthreads();
    for (int j = 0; j < k; ++j) {
        const double piv = L[(size_t)j * KP + j];
        if (!(piv > 1.0e-9 * G[(size_t)j * KP + j])) {
            if (tid == 0) bad = 1;
            break;
        }
        const double d = sqrt(piv), id = 1.0 / d;
        __syncthreads();
        for (int i = j + 1 + tid; i < k; i += nt) L[(size_t)j * KP + i] *= id;
        if (tid == 0) L[(size_t)j * KP + j] = d;
        __syncthreads();
        const int w = k - j - 1;
        for (int q = tid; q < w * w; q += nt) {
            const int i = j + 1 + q % w, l = j + 1 + q / w;
            if (i >= l) L[(size_t)l * KP + i] -= L[(size_t)j * KP + i] * L[(size_t)j * KP + l];
        }
        __syncthreads();
    }
    __syncthreads();
    if (tid == 0) *status = bad ? 0 : 1;
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
                                                        double* __restrict__ panels, int tl_cap, int skip_if_invertible)
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
        if (use_inv) {                              // v = Ginv r
            for (int e = tid; e < k; e += 256) {
                double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;          // four chains: the loads of 8 rows are in flight together
                int c = 0;
#pragma unroll 2
                for (; c + 4 <= k; c += 4) {
                    a0 = __builtin_fma(Ginv[(size_t)c * KP + e], rs[c], a0);
                    a1 = __builtin_fma(Ginv[(size_t)(c + 1) * KP + e], rs[c + 1], a1);
                    a2 = __builtin_fma(Ginv[(size_t)(c + 2) * KP + e], rs[c + 2], a2);
                    a3 = __builtin_fma(Ginv[(size_t)(c + 3) * KP + e], rs[c + 3], a3);
                }
                for (; c < k; ++c) a0 = __builtin_fma(Ginv[(size_t)c * KP + e], rs[c], a0);
                vs[e] = (a0 + a1) + (a2 + a3);
            }
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
        for (int e = tid; e < k; e += 256) {
            X[col * KP + e] = xs[e];
            if (Y) Y[col * KP + e] = ys[e];
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

__host__ __device__ static inline int tri_elems(int t) { return t * (t + 1) / 2; }
__device__ __forceinline__ int tri_off(int l, int t) { return l * t - (l * (l - 1)) / 2; }      // start of column l (entry (l, l))

// per-wave LDS slice: [panel: tri(tl)] [xs ys rs zs vs: 5 x kq doubles] [idx: kq ints] [pas: kq bytes], kq = k rounded up to 64
static inline size_t wave_slice_bytes(int k)
{
    const int kq = (k + 63) / 64 * 64, tl = (k + 1) / 2;
    return (size_t)tri_elems(tl) * 8 + (size_t)5 * kq * 8 + (size_t)kq * 4 + (size_t)kq;
}

__global__ __launch_bounds__(256) void nnls_wide_wave_kernel(double* __restrict__ X, double* __restrict__ Y, int k, int KP, i64 N,
                                                             PartialView R, const double* __restrict__ G,
                                                             const double* __restrict__ Ginv, const int* __restrict__ status,
                                                             int* __restrict__ fail_flag, int iter_tag, i64 col_begin,
                                                             int slice_bytes)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char wave_lds[];
    if (*status == 0) return;                                   // the workgroup kernel takes this launch
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int kq = (k + 63) / 64 * 64, tl = (k + 1) / 2;
    unsigned char* base = wave_lds + (size_t)wave * slice_bytes;
    double* Mp = (double*)base;
    double* xs = Mp + tri_elems(tl);
    double* ys = xs + kq;
    double* rs = ys + kq;
    double* zs = rs + kq;
    double* vs = zs + kq;
    int* idx = (int*)(vs + kq);
    unsigned char* pas = (unsigned char*)(idx + kq);
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    int failed_any = 0;

    for (i64 col = col_begin + (i64)blockIdx.x * nwaves + wave; col < N; col += (i64)gridDim.x * nwaves) {
        for (int e = lane; e < kq; e += 64) {
            const bool in = e < k;
            const double x0 = in ? X[col * KP + e] : 0.0;
            rs[e] = in ? rhs_elem(R, col, e) : 0.0;
            xs[e] = x0;
            pas[e] = in && x0 > 0.0;                            // passive_set = (X > 0), nnls.hpp:157
        }
        WAVE_SYNC();
        for (int e = lane; e < kq; e += 64) {                   // v = Ginv r
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;      // 8 independent loads of Ginv in flight per lane
            if (e < k) {
                int c = 0;
                for (; c + 8 <= k; c += 8) {
                    double g[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) g[u] = Ginv[(size_t)(c + u) * KP + e];
                    a0 = __builtin_fma(g[0], rs[c], a0);     a1 = __builtin_fma(g[1], rs[c + 1], a1);
                    a2 = __builtin_fma(g[2], rs[c + 2], a2); a3 = __builtin_fma(g[3], rs[c + 3], a3);
                    a0 = __builtin_fma(g[4], rs[c + 4], a0); a1 = __builtin_fma(g[5], rs[c + 5], a1);
                    a2 = __builtin_fma(g[6], rs[c + 6], a2); a3 = __builtin_fma(g[7], rs[c + 7], a3);
                }
                for (; c < k; ++c) a0 = __builtin_fma(Ginv[(size_t)c * KP + e], rs[c], a0);
            }
            vs[e] = (a0 + a1) + (a2 + a3);
        }
        WAVE_SYNC();

        int ng = 0, last = -1;
        // one block-pivot solve for the current passive set; leaves xs, ys and the violation count / largest violator
        auto solve_and_classify = [&](bool zeroize) -> bool {
            int p = 0;
            for (int e0 = 0; e0 < kq; e0 += 64) p += __popcll(__ballot(pas[e0 + lane] != 0));
            const bool comp = (k - p) <= p;                     // the smaller block
            int t = 0;
            for (int e0 = 0; e0 < kq; e0 += 64) {
                const int e = e0 + lane;
                const bool sel = (e < k) && ((pas[e] != 0) != comp);
                const unsigned long long m = __ballot(sel);
                if (sel) idx[t + __popcll(m & lt_mask)] = e;
                t += __popcll(m);
            }
            WAVE_SYNC();
            const double* Msrc = comp ? Ginv : G;
            {   // packed lower triangle, column l: rows l .. t - 1; four columns' loads issued together
                int l = 0;
                for (; l + 4 <= t; l += 4) {
                    double g[4][2];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int i = l + u + lane + 64 * h;
                            g[u][h] = (i < t) ? Msrc[(size_t)idx[l + u] * KP + idx[i]] : 0.0;
                        }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int i = l + u + lane + 64 * h;
                            if (i < t) Mp[tri_off(l + u, t) + (i - l - u)] = g[u][h];
                        }
                }
                for (; l < t; ++l) {
                    const int il = idx[l], o = tri_off(l, t);
                    for (int i = l + lane; i < t; i += 64) Mp[o + (i - l)] = Msrc[(size_t)il * KP + idx[i]];
                }
            }
            for (int a = lane; a < t; a += 64) zs[a] = comp ? -vs[idx[a]] : rs[idx[a]];
            WAVE_SYNC();
            bool bad = false;
            for (int j = 0; j < t; ++j) {                       // right-looking Cholesky; the diagonal keeps 1 / L_jj
                const int oj = tri_off(j, t);
                const double piv = Mp[oj];
                if (!(piv > 0.0)) { bad = true; break; }
                double id = __builtin_amdgcn_rsq(piv);          // 1 / sqrt(piv): hardware estimate + two Newton steps
                id = id * __builtin_fma(-0.5 * piv * id, id, 1.5);
                id = id * __builtin_fma(-0.5 * piv * id, id, 1.5);
                WAVE_SYNC();
                for (int i = j + 1 + lane; i < t; i += 64) Mp[oj + (i - j)] *= id;
                if (lane == 0) Mp[oj] = id;
                WAVE_SYNC();
                const int ti = lane & 15, tq = lane >> 4;       // 16 rows x 4 columns of the trailing block at a time
                for (int l = j + 1 + tq; l < t; l += 4) {
                    const double mlj = Mp[oj + (l - j)];
                    const int ol = tri_off(l, t);
                    for (int i = l + ti; i < t; i += 16) Mp[ol + (i - l)] -= Mp[oj + (i - j)] * mlj;
                }
                WAVE_SYNC();
            }
            if (!bad) {
                // substitutions with the right-hand side in registers (t <= 128: two entries per lane): the value of step j
                // is a lane broadcast, not an LDS round trip
                double z0 = (lane < t) ? zs[lane] : 0.0, z1 = (64 + lane < t) ? zs[64 + lane] : 0.0;
                for (int j = 0; j < t; ++j) {                   // L z = b
                    const int oj = tri_off(j, t);
                    const double zj = readlane_f64(j < 64 ? z0 : z1, j & 63) * Mp[oj];
                    if (j < 64) { if (lane == j) z0 = zj; } else { if (lane == j - 64) z1 = zj; }
                    const int i0 = lane, i1 = 64 + lane;
                    if (i0 > j && i0 < t) z0 = __builtin_fma(-Mp[oj + (i0 - j)], zj, z0);
                    if (i1 > j && i1 < t) z1 = __builtin_fma(-Mp[oj + (i1 - j)], zj, z1);
                }
                for (int j = t - 1; j >= 0; --j) {              // L' x = z
                    const double xj = readlane_f64(j < 64 ? z0 : z1, j & 63) * Mp[tri_off(j, t)];
                    if (j < 64) { if (lane == j) z0 = xj; } else { if (lane == j - 64) z1 = xj; }
                    const int i0 = lane, i1 = 64 + lane;
                    if (i0 < j) z0 = __builtin_fma(-Mp[tri_off(i0, t) + (j - i0)], xj, z0);
                    if (i1 < j) z1 = __builtin_fma(-Mp[tri_off(i1, t) + (j - i1)], xj, z1);
                }
                WAVE_SYNC();
                if (lane < t) zs[lane] = z0;
                if (64 + lane < t) zs[64 + lane] = z1;
                WAVE_SYNC();
            }
            // out = base + Msrc[:, T] u  with base = v (complement) or -r (direct)
            for (int e = lane; e < kq; e += 64) {
                double acc = comp ? vs[e] : -rs[e], a1 = 0.0, a2 = 0.0, a3 = 0.0;
                if (!bad && e < k) {
                    int a = 0;
                    for (; a + 8 <= t; a += 8) {
                        double g[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) g[u] = Msrc[(size_t)idx[a + u] * KP + e];
                        acc = __builtin_fma(g[0], zs[a], acc);      a1 = __builtin_fma(g[1], zs[a + 1], a1);
                        a2 = __builtin_fma(g[2], zs[a + 2], a2);    a3 = __builtin_fma(g[3], zs[a + 3], a3);
                        acc = __builtin_fma(g[4], zs[a + 4], acc);  a1 = __builtin_fma(g[5], zs[a + 5], a1);
                        a2 = __builtin_fma(g[6], zs[a + 6], a2);    a3 = __builtin_fma(g[7], zs[a + 7], a3);
                    }
                    for (; a < t; ++a) acc = __builtin_fma(Msrc[(size_t)idx[a] * KP + e], zs[a], acc);
                    acc = (acc + a1) + (a2 + a3);
                }
                const bool pe = pas[e] != 0;
                xs[e] = comp ? (pe ? acc : 0.0) : 0.0;          // the block's own entries are scattered below
                ys[e] = comp ? 0.0 : (pe ? 0.0 : acc);
            }
            WAVE_SYNC();
            if (!bad)
                for (int a = lane; a < t; a += 64) {
                    if (comp) ys[idx[a]] = zs[a];
                    else xs[idx[a]] = zs[a];
                }
            WAVE_SYNC();
            ng = 0;
            last = -1;
            for (int e0 = 0; e0 < kq; e0 += 64) {
                const int e = e0 + lane;
                double x = xs[e], y = ys[e];
                if (zeroize) {                                  // ZeroizeSmallValues, nnls.hpp:213,224-225
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
            WAVE_SYNC();
            return bad;
        };
        // NOTE: after solve_and_classify pas[e] carries bit 0 = passive, bit 1 = violates; the (pas[e] != 0) tests above see
        // only clean bytes because every exchange below rewrites them to 0 / 1 first.
        auto clean = [&]() {
            for (int e = lane; e < kq; e += 64) pas[e] &= 1;
            WAVE_SYNC();
        };

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
                for (int e = lane; e < kq; e += 64) {
                    const unsigned char b = pas[e];
                    pas[e] = (b & 2) ? (unsigned char)((b & 1) ^ 1) : (unsigned char)(b & 1);   // violators change side
                }
            } else {
                for (int e = lane; e < kq; e += 64) {
                    const unsigned char b = pas[e];
                    pas[e] = (e == last) ? (unsigned char)((b & 1) ^ 1) : (unsigned char)(b & 1);   // backup rule: the largest violator
                }
            }
            WAVE_SYNC();
            failed = solve_and_classify(true);
            ++iter;
        }
        clean();
        for (int e = lane; e < k; e += 64) {
            X[col * KP + e] = xs[e];
            if (Y) Y[col * KP + e] = ys[e];
        }
        failed_any |= failed ? 1 : 0;
        WAVE_SYNC();
    }
    if (failed_any && lane == 0) atomicMin(fail_flag, iter_tag);
}

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
// scratch: [panels: wgs x KP x KP][L: KP x KP][Ginv: KP x KP][status: 8]
size_t nnls_wide_scratch_elems(int k, int num_cus)
{
    const size_t KP = (size_t)kp_of(k);
    return ((size_t)num_cus * nnls_wide_wgs_per_cu(k) + 2) * KP * KP + 8;
}

int launch_nnls_bpp_wide(double* X, double* Y, int k, i64 col_begin, i64 col_end, PartialView R, const double* G, int* fail_flag,
                         int iter_tag, double* scratch, int num_cus, hipStream_t st)
{
    const i64 ncols = col_end - col_begin;
    if (ncols <= 0) return 0;
    if (!scratch) { set_error("nnls: k > 128 needs the scratch panels"); return -100; }
    const int KP = kp_of(k);
    const int wgs = num_cus * nnls_wide_wgs_per_cu(k), tl = nnls_wide_tl(k);
    double* L = scratch + (size_t)wgs * KP * KP;
    double* Ginv = L + (size_t)KP * KP;
    int* status = (int*)(Ginv + (size_t)KP * KP);
    static const bool use_inv = [] { const char* e = getenv("SMK_NNLS_INV"); return !(e && e[0] == '0'); }();
    if (use_inv) {
        chol_wide_kernel<<<1, 1024, 0, st>>>(G, k, KP, L, status);
        SMK_HIP(hipGetLastError());
        inv_cols_wide_kernel<<<k, 256, 0, st>>>(L, k, KP, Ginv, status);
        SMK_HIP(hipGetLastError());
    }
    const int lds = nnls_wide_lds_bytes(k);
    static bool attr_set = false;
    if (!attr_set) {
        SMK_HIP(hipFuncSetAttribute((const void*)nnls_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
        attr_set = true;
    }
    // k <= 256 and G invertible: one wave per column, no workgroup barriers; otherwise (and as the fallback that
    // reproduces a "not SPD" failure) one workgroup per column
    int took_wave = 0;
    if (use_inv && k <= 256) {
        const size_t slice = (wave_slice_bytes(k) + 15) / 16 * 16;
        int waves = (int)((150 * 1024) / slice);
        if (waves > 4) waves = 4;
        if (waves >= 1) {
            const int wlds = (int)(slice * waves);
            static int attr_wave = 0;
            if (attr_wave < wlds) {
                SMK_HIP(hipFuncSetAttribute((const void*)nnls_wide_wave_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
                attr_wave = 152 * 1024;
            }
            i64 g2 = (ncols + waves - 1) / waves;
            if (g2 > (i64)num_cus * 2) g2 = (i64)num_cus * 2;
            nnls_wide_wave_kernel<<<(unsigned)g2, 64 * waves, wlds, st>>>(X, Y, k, KP, col_end, R, G, Ginv, status, fail_flag, iter_tag,
                                                                     col_begin, (int)slice);
            SMK_HIP(hipGetLastError());
            took_wave = 1;
        }
    }
    i64 grid = wgs;                                              // as many workgroups as their LDS panels let be resident
    if (grid > ncols) grid = ncols;
    nnls_wide_kernel<<<(unsigned)grid, 256, lds, st>>>(X, Y, k, KP, col_end, R, G, use_inv ? Ginv : nullptr, status, fail_flag,
                                                       iter_tag, col_begin, scratch, tl, took_wave);
    SMK_HIP(hipGetLastError());
    return 0;
}

}  // namespace smk
