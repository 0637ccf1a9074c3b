// smallk_amd/csrc/rank2.hip -- the RANK2 iteration (nmf_solver_rank2.hpp:353-455) as few, fused launches.
//
// HierNMF2 spends its time in thousands of rank-2 iterations on small and medium node matrices: every kernel of the
// iteration is launch-latency bound (4 - 12 us each), so what counts is how many there are.  Per iteration:
//   rank2_solve_kernel      closed-form 2x2 solve + optimal active set of one factor, a compact N x 2 copy of the result
//                           for the gather product, and per-workgroup partial sums of the Gram matrix of the result --
//                           summed by the kernel that consumes the matrix (see Gram2 below), not by a reduce launch
//   rank2_normalize_kernel  NormalizeAndScale of H, W, the stored AH' and HH' (:418-437); also writes W'W of the
//                           normalised W (D^-1 W'W D^-1) and the compact copy of the normalised W
//   rank2_progress_kernel   both projected-gradient sums (projected_gradient.hpp:125-171) as per-workgroup partials that
//                           the host adds up after its read-back, the failure flag, and the snapshot of (W, H, W'W) that
//                           lets the driver undo a speculative iteration, in one launch
// The two products with A run between them (spmm.hip for sparse A, bigprod.hip for dense A).
#include "devutil.h"
#include "rank2_math.h"

namespace smk {

// ---- grid-wide 2 x 2 sums without a reduce launch ------------------------------------------------------------------
// A kernel that produces a rank-2 Gram matrix leaves per-workgroup partial sums ([workgroups][4]: s00, s01, s11); the
// kernel that CONSUMES it sums them itself, every workgroup redundantly and in the same fixed order (bit-identical in all
// of them; the kernel boundary makes the partials visible -- no fences, no atomics), and its workgroup 0 also stores the
// finished matrix for everybody downstream.  That costs nb x 24 bytes of L2 reads per workgroup, so it is used up to
// R2_INLINE_BLOCKS partials; larger problems take the separate 5 us reduce launch, which is noise next to their products.
// (Tried and dropped: a last-arriver reduce -- __threadfence() per workgroup writes back the whole L2, 80x slower; ticket
// atomics + self-validating slots cost ~18 ns per workgroup on the one contended address, 70 us at 1 M columns.)
constexpr int R2_INLINE_BLOCKS = 768;

struct Gram2 {
    const double* G;      // finished KP x KP matrix (nb == 0), else where workgroup 0 stores it (may be null)
    const double* Gp;     // [nb][4] partial sums
    int nb;
};

static __device__ __forceinline__ void gram2_load(const Gram2& g, double* sh /* >= 16 */, double& g00, double& g01, double& g11)
{
    constexpr int KP = 8;
    if (g.nb == 0) { g00 = g.G[0]; g01 = g.G[1]; g11 = g.G[KP + 1]; return; }
    double a = 0.0, b = 0.0, c = 0.0;
    for (int i = threadIdx.x; i < g.nb; i += blockDim.x) {
        const f64x2_t u = *(const f64x2_t*)(g.Gp + (i64)i * 4);
        a += u[0];
        b += u[1];
        c += g.Gp[(i64)i * 4 + 2];
    }
    a = block_sum(a, sh);
    b = block_sum(b, sh);
    c = block_sum(c, sh);
    if (threadIdx.x == 0) { sh[9] = a; sh[10] = b; sh[11] = c; }
    __syncthreads();
    g00 = sh[9]; g01 = sh[10]; g11 = sh[11];
    __syncthreads();
}

// ==========================================================================
// SystemSolveH :25-135 / SystemSolveW :139-212 (one fast Givens rotation, cosine or sine branch) followed by the
// optimal active set :216-318.  One thread per column of X (KP = 8 layout, rows 0 and 1 live).
// Xc (optional): compact copy, 16 B per column.  Gp (optional): [workgroups][4] partial sums of X X'.
// ==========================================================================
__global__ __launch_bounds__(256) void rank2_solve_kernel(double* __restrict__ X, double* __restrict__ Xc, i64 N, PartialView R,
                                                          Gram2 gin, double* __restrict__ Gstore, int side,
                                                          int* __restrict__ fail_flag, int iter_tag,
                                                          double* __restrict__ Gp)
{
    constexpr int KP = 8;
    __shared__ double shg[4][3];
    __shared__ double sh[16];
    double a00, a01, a11;
    gram2_load(gin, sh, a00, a01, a11);
    if (gin.nb != 0 && Gstore && blockIdx.x == 0 && threadIdx.x < KP * KP) {
        const int e = threadIdx.x;
        Gstore[e] = (e == 0) ? a00 : (e == 1 || e == KP) ? a01 : (e == KP + 1) ? a11 : 0.0;
    }
    const R2Solve sv = r2_prepare(a00, a01, a11, side);           // rank2_math.h: the reference's formulas
    if (sv.bad) {
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicMin(fail_flag, iter_tag);
        return;
    }
    const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = j < N;
    double x0 = 0.0, x1 = 0.0;
    if (valid) {
        r2_apply(sv, side, rhs_elem(R, j, 0), rhs_elem(R, j, 1), x0, x1);
        f64x2_t v;
        v[0] = x0;
        v[1] = x1;
        *(f64x2_t*)(X + j * KP) = v;
        if (Xc) *(f64x2_t*)(Xc + j * 2) = v;
    }
    if (!Gp) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double s00 = wave_sum(x0 * x0), s01 = wave_sum(x0 * x1), s11 = wave_sum(x1 * x1);
    if (lane == 0) { shg[wave][0] = s00; shg[wave][1] = s01; shg[wave][2] = s11; }
    __syncthreads();
    if (threadIdx.x < 3)
        Gp[(i64)blockIdx.x * 4 + threadIdx.x] = (shg[0][threadIdx.x] + shg[1][threadIdx.x]) + (shg[2][threadIdx.x] + shg[3][threadIdx.x]);
}

// sum [nb][4] partials into a finished KP x KP matrix (problems above R2_INLINE_BLOCKS workgroups, sharded runs)
__global__ __launch_bounds__(256) void rank2_gram_finish_kernel(const double* __restrict__ Gp, int nb, double* __restrict__ G)
{
    constexpr int KP = 8;
    __shared__ double sh[16];
    double g00, g01, g11;
    gram2_load(Gram2{nullptr, Gp, nb}, sh, g00, g01, g11);
    if (threadIdx.x < KP * KP) {
        const int e = threadIdx.x;
        G[e] = (e == 0) ? g00 : (e == 1 || e == KP) ? g01 : (e == KP + 1) ? g11 : 0.0;
    }
}

// partial-sum buffers of the fused RANK2 kernels: two of [blocks][4] doubles (the H-side and the W-side Gram matrix are
// alive at the same time)
size_t rank2_gram_scratch_elems(i64 N) { return (size_t)((N + 255) / 256) * 4 * 2 + 16; }

// X <- closed-form solve of G X = R (side 0: H, side 1: W').  The left-hand side G is given finished (gin_nb == 0: Gin)
// or as gin_nb partial sums (Gin_p) which this kernel sums itself; in that case workgroup 0 also stores the finished
// matrix in Gin.  Gp_out (optional): partial sums of X X' of the result; *nb_out = how many.  finish != 0 (or too many
// partials for the consumer to sum in line): they are summed into Gout by a second launch and *nb_out = 0.
int launch_rank2_solve(double* X, double* Xc, i64 N, PartialView R, double* Gin, const double* Gin_p, int gin_nb, int side,
                       int* fail_flag, int iter_tag, double* Gp_out, int* nb_out, double* Gout, int finish, hipStream_t st)
{
    const int grid = (int)((N + 255) / 256);
    if (nb_out) *nb_out = 0;
    if (grid < 1) return 0;
    rank2_solve_kernel<<<grid, 256, 0, st>>>(X, Xc, N, R, Gram2{Gin, Gin_p, gin_nb}, Gin, side, fail_flag, iter_tag, Gp_out);
    SMK_HIP(hipGetLastError());
    if (!Gp_out) return 0;
    if (finish || grid > R2_INLINE_BLOCKS) {
        rank2_gram_finish_kernel<<<1, 256, 0, st>>>(Gp_out, grid, Gout);
        SMK_HIP(hipGetLastError());
    } else if (nb_out) {
        *nb_out = grid;
    }
    return 0;
}

// ==========================================================================
// Per-iteration NormalizeAndScale of RANK2 (nmf_solver_rank2.hpp:418-437) in one launch: H rows *= nu, W columns /= nu,
// the stored AH' *= nu per column, HH'_ij *= nu_i nu_j; nu_c = sqrt(Graw[c][c]) with Graw = W'W of the W just solved.
// Also Gw <- D^-1 Graw D^-1 (= W'W of the normalised W, no second pass over W) and the compact copy of the normalised W.
// A zero norm reports -2 through fail_flag and leaves that component unscaled (the reference throws).
// ==========================================================================
__global__ __launch_bounds__(256) void rank2_normalize_kernel(double* __restrict__ H, i64 n, double* __restrict__ Wt,
                                                              double* __restrict__ Wc, i64 m, void* __restrict__ P, int S,
                                                              i64 slab, int kpp, int f64, double* __restrict__ Gh,
                                                              Gram2 graw, double* __restrict__ Gw,
                                                              int* __restrict__ fail_flag)
{
    constexpr int KP = 8;
    __shared__ double sh[16];
    double g00, g01, g11;
    gram2_load(graw, sh, g00, g01, g11);
    const double nu0 = sqrt(g00), nu1 = sqrt(g11);
    const bool ok0 = !(fabs(nu0) < DBL_EPSILON), ok1 = !(fabs(nu1) < DBL_EPSILON);
    const double h0 = ok0 ? nu0 : 1.0, h1 = ok1 ? nu1 : 1.0;
    const double w0 = ok0 ? 1.0 / nu0 : 1.0, w1 = ok1 ? 1.0 / nu1 : 1.0;
    const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j == 0) {
        if (!ok0 || !ok1) atomicMin(fail_flag, -2);
        Gh[0] *= nu0 * nu0;
        Gh[1] *= nu0 * nu1;
        Gh[KP] *= nu0 * nu1;
        Gh[KP + 1] *= nu1 * nu1;
    }
    if (blockIdx.x == 0 && threadIdx.x < KP * KP) {
        const int e = threadIdx.x;
        Gw[e] = (e == 0) ? g00 / (nu0 * nu0) : (e == 1 || e == KP) ? g01 / (nu0 * nu1) : (e == KP + 1) ? g11 / (nu1 * nu1) : 0.0;
    }
    if (j < n) {
        f64x2_t v = *(f64x2_t*)(H + j * KP);
        v[0] *= h0;
        v[1] *= h1;
        *(f64x2_t*)(H + j * KP) = v;
    }
    if (j < m) {
        f64x2_t v = *(f64x2_t*)(Wt + j * KP);
        v[0] *= w0;
        v[1] *= w1;
        *(f64x2_t*)(Wt + j * KP) = v;
        if (Wc) *(f64x2_t*)(Wc + j * 2) = v;
        for (int s = 0; s < S; ++s) {
            if (f64) {
                double* p = (double*)P + s * slab + j * kpp;
                p[0] *= nu0;
                p[1] *= nu1;
            } else {
                float* p = (float*)P + s * slab + j * kpp;
                p[0] = (float)((double)p[0] * nu0);
                p[1] = (float)((double)p[1] * nu1);
            }
        }
    }
}

// Graw: W'W of the W just solved, finished (graw_nb == 0) or as graw_nb partial sums (Graw_p)
int launch_rank2_normalize(double* H, i64 n, double* Wt, double* Wc, i64 m, PartialView R, double* Gh, const double* Graw,
                           const double* Graw_p, int graw_nb, double* Gw, int* fail_flag, hipStream_t st)
{
    const i64 cnt = n > m ? n : m;
    const int grid = (int)((cnt + 255) / 256);
    rank2_normalize_kernel<<<grid, 256, 0, st>>>(H, n, Wt, Wc, m, const_cast<void*>(R.p), R.S, R.slab, R.kpp, R.f64, Gh,
                                                 Gram2{Graw, Graw_p, graw_nb}, Gw, fail_flag);
    SMK_HIP(hipGetLastError());
    return 0;
}

// compact N x 2 copy of the live rows of a KP = 8 factor (solver.Init: the first W'A gathers from W0)
__global__ __launch_bounds__(256) void rank2_compact_kernel(const double* __restrict__ X, double* __restrict__ Xc, i64 N)
{
    const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < N) *(f64x2_t*)(Xc + j * 2) = *(const f64x2_t*)(X + j * 8);
}
int launch_rank2_compact(const double* X, double* Xc, i64 N, hipStream_t st)
{
    const int grid = (int)((N + 255) / 256);
    if (grid < 1) return 0;
    rank2_compact_kernel<<<grid, 256, 0, st>>>(X, Xc, N);
    SMK_HIP(hipGetLastError());
    return 0;
}

// ==========================================================================
// Stopping rule of one RANK2 iteration in one launch: gradW = W HH' - AH', gradH = W'W H - W'A and their projected sums
// (entries with g < 0 or x > 0) as per-workgroup partials ([workgroups][2]: W side, H side -- the host adds them up in
// index order after the read-back it does anyway), the failure flag mirrored as a double behind them, and -- snap !=
// nullptr -- the snapshot of (W, H, W'W) in launch_snapshot's layout.
// ==========================================================================
__global__ __launch_bounds__(256) void rank2_progress_kernel(const double* __restrict__ Wt, i64 m, PartialView R2,
                                                             const double* __restrict__ Gh, const double* __restrict__ H, i64 n,
                                                             PartialView R1, const double* __restrict__ Gw,
                                                             double* __restrict__ partials, const int* __restrict__ flag,
                                                             double* __restrict__ snap)
{
    constexpr int KP = 8;
    __shared__ double sh[16];
    const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    f64x2_t* s2 = (f64x2_t*)snap;
    double sw = 0.0, shh = 0.0;
    if (j < m) {
        const f64x2_t w = *(const f64x2_t*)(Wt + j * KP);
        const double g0 = (Gh[0] * w[0] + Gh[KP] * w[1]) - rhs_elem(R2, j, 0);
        const double g1 = (Gh[1] * w[0] + Gh[KP + 1] * w[1]) - rhs_elem(R2, j, 1);
        if (g0 < 0.0 || w[0] > 0.0) sw += g0 * g0;
        if (g1 < 0.0 || w[1] > 0.0) sw += g1 * g1;
        if (snap) s2[j] = w;
    }
    if (j < n) {
        const f64x2_t h = *(const f64x2_t*)(H + j * KP);
        const double g0 = (Gw[0] * h[0] + Gw[KP] * h[1]) - rhs_elem(R1, j, 0);
        const double g1 = (Gw[1] * h[0] + Gw[KP + 1] * h[1]) - rhs_elem(R1, j, 1);
        if (g0 < 0.0 || h[0] > 0.0) shh += g0 * g0;
        if (g1 < 0.0 || h[1] > 0.0) shh += g1 * g1;
        if (snap) s2[m + j] = h;
    }
    if (snap && blockIdx.x == 0 && threadIdx.x < KP * KP / 2) s2[m + n + threadIdx.x] = ((const f64x2_t*)Gw)[threadIdx.x];
    const double tw = block_sum(sw, sh);
    const double th = block_sum(shh, sh);
    if (threadIdx.x == 0) {
        partials[(i64)blockIdx.x * 2] = tw;
        partials[(i64)blockIdx.x * 2 + 1] = th;
        if (blockIdx.x == 0) partials[(i64)gridDim.x * 2] = flag ? (double)*flag : 0.0;
    }
}

int rank2_progress_blocks(i64 m, i64 n) { return (int)(((m > n ? m : n) + 255) / 256); }
// [blocks][2] partial sums + the flag
size_t rank2_progress_scratch_elems(i64 m, i64 n) { return (size_t)rank2_progress_blocks(m, n) * 2 + 2; }

int launch_rank2_progress(const double* Wt, i64 m, PartialView R2, const double* Gh, const double* H, i64 n, PartialView R1,
                          const double* Gw, double* partials, const int* flag, double* snap, hipStream_t st)
{
    const int grid = rank2_progress_blocks(m, n);
    if (grid < 1) return 0;
    rank2_progress_kernel<<<grid, 256, 0, st>>>(Wt, m, R2, Gh, H, n, R1, Gw, partials, flag, snap);
    SMK_HIP(hipGetLastError());
    return 0;
}

}  // namespace smk
