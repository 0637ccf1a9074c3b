/*
 * oracle/nmf_oracle.c -- CPU restatement (fp64, column-major) of the dense NMF
 * hot path of smallk.  TEST INFRASTRUCTURE ONLY.
 *
 *   * Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may
 *     load this file's shared library.  The product (smallk_amd/) never does.
 *   * PARITY UNPINNED: the reference tree ships no golden vectors for this path
 *     (its fixtures live in the external `smallk_data` repository, which is not
 *     in /root/reference) and the reference itself cannot be compiled in this
 *     image (it needs Elemental's El.hpp, an empty submodule).  This file is
 *     therefore a line-by-line behavioural restatement checked only against
 *     (i) an independent numpy/scipy restatement (tests/golden/make_golden.py),
 *     (ii) the reference's own property tests (KKT / residual thresholds,
 *     tests/src/test_bpp.cpp, test_dense_nmf.cpp).  See DESIGN.md section 3.
 *
 * Every function cites the reference file:line (relative to /root/reference)
 * whose behaviour it restates.  No reference source text is reproduced.
 *
 * Build: see oracle/Makefile  (gcc -O3 -march=x86-64-v3 -fopenmp -shared).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef int64_t i64;

/* ---- enums: common/include/nmf.hpp:17-41 -------------------------------- */
enum { ORC_OK = 0, ORC_NOTINITIALIZED = -1, ORC_INITIALIZED = -2, ORC_BAD_PARAM = -3,
       ORC_FAILURE = -4, ORC_SIZE_TOO_LARGE = -5 };
enum { ORC_MU = 0, ORC_HALS = 1, ORC_RANK2 = 2, ORC_BPP = 3 };
enum { ORC_PG_RATIO = 0, ORC_DELTA_FNORM = 1 };

/* NmfOptions, common/include/nmf.hpp:55-69 (bools widened to int for the FFI) */
typedef struct {
    double tol;
    int algorithm;
    int prog_est_algorithm;
    int height, width, k;
    int min_iter, max_iter, tolcount, max_threads;
    int verbose, normalize;
} orc_options;

/* NmfStats, common/include/nmf.hpp:43-53 */
typedef struct {
    unsigned long long elapsed_us;
    int iteration_count;
} orc_stats;

#define AT(buf, ld, r, c) ((buf)[(i64)(c) * (i64)(ld) + (i64)(r)])

static double now_us(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}

/* ======================================================================== */
/* Synthetic data: counter-based uniform [0,1) generator shared bit-for-bit  */
/* with the device generator (smallk_amd/csrc/fill.hip).  Element (r,c) of a  */
/* matrix with GLOBAL leading dimension `gld` is hashed from its global       */
/* linear index, so any shard/sub-block is reproducible.  SURVEY 8(d).        */
/* ======================================================================== */
static inline uint64_t orc_mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static inline float orc_bf16_round(float f)
{
    uint32_t b;
    memcpy(&b, &f, 4);
    b += 0x7FFFu + ((b >> 16) & 1u);   /* round to nearest even */
    b &= 0xFFFF0000u;
    memcpy(&f, &b, 4);
    return f;
}

/* quant: 0 = 24-bit uniform (exact in fp32 and fp64), 1 = rounded to bf16 */
double orc_uniform_value(uint64_t seed, uint64_t gidx, int quant)
{
    uint64_t h = orc_mix64(seed * 0xD1342543DE82EF95ull + gidx);
    float f = (float)(h >> 40) * (1.0f / 16777216.0f);
    if (quant == 1) f = orc_bf16_round(f);
    return (double)f;
}

/* Fill the rows x cols block whose top-left element is global (r0,c0) of a
 * matrix with global height `gheight`. */
void orc_fill_uniform(double* buf, i64 ld, i64 rows, i64 cols,
                      i64 r0, i64 c0, i64 gheight, uint64_t seed, int quant)
{
#pragma omp parallel for schedule(static)
    for (i64 c = 0; c < cols; ++c)
        for (i64 r = 0; r < rows; ++r)
            AT(buf, ld, r, c) =
                orc_uniform_value(seed, (uint64_t)((c0 + c) * gheight + (r0 + r)), quant);
}

/* Structured synthetic data ("planted-low-rank variant ... for convergence sanity"; the formula of
 * tests/golden/make_golden.py:make_A with the order of operations fixed): element (r, c) of
 *   A = Ws Hs + noise * U,  Ws = uniform(gheight x kstar, seed + 1) with entries <= thr dropped,
 *   Hs = uniform(kstar x gwidth, seed + 2) likewise, U = uniform(gheight x gwidth, seed),
 * fp64 sum in increasing j with one fma per term (a zero factor leaves the sum unchanged, so zero terms may be skipped),
 * the noise term last with one more fma, then rounded to the storage type.  The device twin
 * (smallk_amd/csrc/kernels.hip:fill_planted_kernel) does the same operations: identical bits. */
void orc_fill_planted(double* buf, i64 ld, i64 rows, i64 cols, i64 r0, i64 c0, i64 gheight,
                      uint64_t seed, int kstar, double thr, double noise, int quant)
{
    const float thrf = (float)thr;
    float* ws = (float*)malloc((size_t)rows * kstar * sizeof(float));     /* [r][j] */
#pragma omp parallel for schedule(static)
    for (i64 r = 0; r < rows; ++r)
        for (int j = 0; j < kstar; ++j) {
            float w = (float)orc_uniform_value(seed + 1, (uint64_t)((i64)j * gheight + r0 + r), 0);
            ws[r * kstar + j] = w > thrf ? w : 0.f;
        }
#pragma omp parallel
    {
        float* hs = (float*)malloc((size_t)kstar * sizeof(float));
#pragma omp for schedule(static)
        for (i64 c = 0; c < cols; ++c) {
            for (int j = 0; j < kstar; ++j) {
                float h = (float)orc_uniform_value(seed + 2, (uint64_t)((c0 + c) * (i64)kstar + j), 0);
                hs[j] = h > thrf ? h : 0.f;
            }
            for (i64 r = 0; r < rows; ++r) {
                double acc = 0.0;
                const float* w = ws + r * kstar;
                for (int j = 0; j < kstar; ++j)
                    if (hs[j] != 0.f) acc = fma((double)w[j], (double)hs[j], acc);
                float u = (float)orc_uniform_value(seed, (uint64_t)((c0 + c) * gheight + (r0 + r)), 0);
                float v = (float)fma(noise, (double)u, acc);
                if (quant == 1) v = orc_bf16_round(v);
                AT(buf, ld, r, c) = (double)v;
            }
        }
        free(hs);
    }
    free(ws);
}

/* Round a buffer in place to what the device stores (fp32 or bf16). */
void orc_quantize(double* buf, i64 count, int quant)
{
#pragma omp parallel for schedule(static)
    for (i64 i = 0; i < count; ++i) {
        float f = (float)buf[i];
        if (quant == 1) f = orc_bf16_round(f);
        buf[i] = (double)f;
    }
}

/* ======================================================================== */
/* Dense helpers: the BLAS-like subset behind dense_matrix_ops.hpp            */
/* (Gemm :255-270, Gemv :287-296, Axpy, Norm :118-140, Nrm2, Scal).           */
/* Textbook dgemm/dgemv semantics; Elemental+BLAS itself is not in the tree   */
/* (.gitmodules:1-6), so summation order differs from a real build at the     */
/* 1e-13 level only.                                                          */
/* ======================================================================== */

/* C(MxN) = alpha * op(A) * op(B) + beta * C ; op = transpose when t? != 0.
 * A is (ta ? K x M : M x K), B is (tb ? N x K : K x N). */
void orc_gemm(int ta, int tb, i64 M, i64 N, i64 K, double alpha,
              const double* A, i64 lda, const double* B, i64 ldb,
              double beta, double* C, i64 ldc)
{
    if (beta == 0.0) {
#pragma omp parallel for schedule(static)
        for (i64 j = 0; j < N; ++j)
            for (i64 i = 0; i < M; ++i) AT(C, ldc, i, j) = 0.0;
    } else if (beta != 1.0) {
#pragma omp parallel for schedule(static)
        for (i64 j = 0; j < N; ++j)
            for (i64 i = 0; i < M; ++i) AT(C, ldc, i, j) *= beta;
    }
    if (M == 0 || N == 0 || K == 0) return;

    if (ta && !tb) {
        /* C = A' * B : dot products down contiguous columns.  Block the long
         * K dimension so the A panel stays cache resident. */
        const i64 KB = 1024;
        for (i64 k0 = 0; k0 < K; k0 += KB) {
            i64 kb = (K - k0 < KB) ? (K - k0) : KB;
#pragma omp parallel for schedule(static)
            for (i64 j = 0; j < N; ++j) {
                const double* b = &AT(B, ldb, k0, j);
                for (i64 i = 0; i < M; ++i) {
                    const double* a = &AT(A, lda, k0, i);
                    double s = 0.0;
                    for (i64 p = 0; p < kb; ++p) s += a[p] * b[p];
                    AT(C, ldc, i, j) += alpha * s;
                }
            }
        }
    } else if (!ta && tb) {
        /* C = A * B' : rank-1 (axpy) updates, each thread owns a row block of C */
        const i64 RB = 256;
        i64 nblk = (M + RB - 1) / RB;
#pragma omp parallel for schedule(dynamic, 1)
        for (i64 blk = 0; blk < nblk; ++blk) {
            i64 i0 = blk * RB, ib = (M - i0 < RB) ? (M - i0) : RB;
            for (i64 p = 0; p < K; ++p) {
                const double* a = &AT(A, lda, i0, p);
                for (i64 j = 0; j < N; ++j) {
                    double s = alpha * AT(B, ldb, j, p);
                    double* c = &AT(C, ldc, i0, j);
                    for (i64 i = 0; i < ib; ++i) c[i] += s * a[i];
                }
            }
        }
    } else if (!ta && !tb) {
        /* C = A * B : column j of C is a combination of columns of A */
        if (N >= M) {
#pragma omp parallel for schedule(static)
            for (i64 j = 0; j < N; ++j) {
                double* c = &AT(C, ldc, 0, j);
                for (i64 p = 0; p < K; ++p) {
                    double s = alpha * AT(B, ldb, p, j);
                    const double* a = &AT(A, lda, 0, p);
                    for (i64 i = 0; i < M; ++i) c[i] += s * a[i];
                }
            }
        } else {
            const i64 RB = 512;
            i64 nblk = (M + RB - 1) / RB;
#pragma omp parallel for schedule(dynamic, 1)
            for (i64 blk = 0; blk < nblk; ++blk) {
                i64 i0 = blk * RB, ib = (M - i0 < RB) ? (M - i0) : RB;
                for (i64 j = 0; j < N; ++j) {
                    double* c = &AT(C, ldc, i0, j);
                    for (i64 p = 0; p < K; ++p) {
                        double s = alpha * AT(B, ldb, p, j);
                        const double* a = &AT(A, lda, i0, p);
                        for (i64 i = 0; i < ib; ++i) c[i] += s * a[i];
                    }
                }
            }
        }
    } else {
        /* C = A' * B' (not used on the hot path; kept for completeness) */
        for (i64 j = 0; j < N; ++j)
            for (i64 i = 0; i < M; ++i) {
                double s = 0.0;
                for (i64 p = 0; p < K; ++p) s += AT(A, lda, p, i) * AT(B, ldb, j, p);
                AT(C, ldc, i, j) += alpha * s;
            }
    }
}

static void mat_axpy(double alpha, i64 M, i64 N, const double* X, i64 ldx, double* Y, i64 ldy)
{
#pragma omp parallel for schedule(static)
    for (i64 j = 0; j < N; ++j)
        for (i64 i = 0; i < M; ++i) AT(Y, ldy, i, j) += alpha * AT(X, ldx, i, j);
}

static void mat_copy(i64 M, i64 N, const double* X, i64 ldx, double* Y, i64 ldy)
{
    for (i64 j = 0; j < N; ++j) memcpy(&AT(Y, ldy, 0, j), &AT(X, ldx, 0, j), (size_t)M * sizeof(double));
}

/* Frobenius norm (dense_matrix_ops.hpp:118-140, FROBENIUS_NORM) */
double orc_fnorm(i64 M, i64 N, const double* X, i64 ldx)
{
    double s = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : s)
    for (i64 j = 0; j < N; ++j) {
        double t = 0.0;
        for (i64 i = 0; i < M; ++i) t += AT(X, ldx, i, j) * AT(X, ldx, i, j);
        s += t;
    }
    return sqrt(s);
}

/* out(N x M) = in(M x N)'  (dense_matrix_ops.hpp Transpose) */
static void mat_transpose(i64 M, i64 N, const double* in, i64 ldi, double* out, i64 ldo)
{
    const i64 TB = 32;
#pragma omp parallel for schedule(static)
    for (i64 j0 = 0; j0 < N; j0 += TB)
        for (i64 i0 = 0; i0 < M; i0 += TB)
            for (i64 j = j0; j < j0 + TB && j < N; ++j)
                for (i64 i = i0; i < i0 + TB && i < M; ++i) AT(out, ldo, j, i) = AT(in, ldi, i, j);
}

/* ======================================================================== */
/* Cholesky / HPD solve: normal_eq.hpp:27-54 -> El::HPDSolve(UPPER, NORMAL),  */
/* i.e. LAPACK dpotrf('U') + two triangular solves.  Returns 0 when a pivot   */
/* is not strictly positive (Elemental raises NonHPSDMatrixException, which   */
/* normal_eq.hpp:41-52 turns into `false`).                                   */
/* ======================================================================== */
static int chol_upper(int n, double* M, int ld)
{
    for (int j = 0; j < n; ++j) {
        double d = AT(M, ld, j, j);
        for (int p = 0; p < j; ++p) d -= AT(M, ld, p, j) * AT(M, ld, p, j);
        if (!(d > 0.0)) return 0;
        d = sqrt(d);
        AT(M, ld, j, j) = d;
        for (int c = j + 1; c < n; ++c) {
            double s = AT(M, ld, j, c);
            for (int p = 0; p < j; ++p) s -= AT(M, ld, p, j) * AT(M, ld, p, c);
            AT(M, ld, j, c) = s / d;
        }
    }
    return 1;
}

/* Solve U'U x = b in place for nrhs right-hand sides. */
static void chol_solve_upper(int n, const double* U, int ld, double* B, i64 ldb, i64 nrhs)
{
#pragma omp parallel for schedule(static) if (nrhs > 64)
    for (i64 c = 0; c < nrhs; ++c) {
        double* b = &AT(B, ldb, 0, c);
        for (int i = 0; i < n; ++i) {          /* U' y = b */
            double s = b[i];
            for (int p = 0; p < i; ++p) s -= AT(U, ld, p, i) * b[p];
            b[i] = s / AT(U, ld, i, i);
        }
        for (int i = n - 1; i >= 0; --i) {     /* U x = y */
            double s = b[i];
            for (int p = i + 1; p < n; ++p) s -= AT(U, ld, i, p) * b[p];
            b[i] = s / AT(U, ld, i, i);
        }
    }
}

/* SolveNormalEq(LHS, RHS, X): normal_eq.hpp:58-74 (copies LHS, X = RHS, solves) */
static int solve_normal_eq_full(int k, const double* LHS, int ldl,
                                const double* RHS, i64 ldr, double* X, i64 ldx, i64 ncols)
{
    double* M = (double*)malloc((size_t)k * k * sizeof(double));
    for (int j = 0; j < k; ++j)
        for (int i = 0; i < k; ++i) M[(size_t)j * k + i] = AT(LHS, ldl, i, j);
    mat_copy(k, ncols, RHS, ldr, X, ldx);
    int ok = chol_upper(k, M, k);
    if (ok) chol_solve_upper(k, M, k, X, ldx, ncols);
    else fprintf(stderr, "Cholesky factorization failure - matrix was not symmetric positive-definite.\n");
    free(M);
    return ok;
}

/* ======================================================================== */
/* BitMatrix: bit_matrix.hpp:21-153 / bit_matrix.cpp.  Column-packed 32-bit   */
/* words, ldim = ceil(height/32), tail bits forced to zero through MASK       */
/* (bit_matrix.cpp:28-44).                                                    */
/* ======================================================================== */
typedef struct {
    int height;
    i64 width;
    int ldim;        /* words per column */
    int full_wds;    /* height / 32 */
    uint32_t mask;   /* mask for the partial last word, 0 if none */
    uint32_t* w;
} bitmat;

static bitmat bm_alloc(int height, i64 width)
{
    bitmat b;
    b.height = height;
    b.width = width;
    b.full_wds = height / 32;
    int extra = height - 32 * b.full_wds;
    b.ldim = b.full_wds + (extra ? 1 : 0);
    b.mask = extra ? ((1u << extra) - 1u) : 0u;
    b.w = (uint32_t*)calloc((size_t)(b.ldim > 0 ? b.ldim : 1) * (size_t)(width > 0 ? width : 1), sizeof(uint32_t));
    return b;
}
static void bm_free(bitmat* b) { free(b->w); b->w = NULL; }

static inline int popcount32(uint32_t x) { return __builtin_popcount(x); }

/* BitMatrix::SumColumns (bit_matrix.cpp, popcount per column; population_count.hpp:22-40) */
static int bm_colsum(const bitmat* b, i64 c)
{
    int s = 0;
    for (int q = 0; q < b->ldim; ++q) s += popcount32(b->w[c * b->ldim + q]);
    return s;
}

/* BitMatrix::MaxRowIndex, bit_matrix.cpp:432-468.  QUIRK restated on purpose:
 * for a set bit found in a *full* word with index r_wd > 0 the reference returns
 * (r_wd-1)*32 + q, i.e. 32 less than the true row; and an empty column gives 0. */
static unsigned bm_max_row_index(const bitmat* b, i64 c)
{
    const uint32_t* col = &b->w[c * b->ldim];
    int r_wd_start = b->ldim - 1;
    if (b->mask > 0) {
        int extra = b->height - 32 * b->full_wds;
        uint32_t wd = b->mask & col[r_wd_start];
        for (int q = extra - 1; q >= 0; --q)
            if (wd & (1u << q)) return (unsigned)(b->full_wds * 32 + q);
        --r_wd_start;
    }
    for (int r_wd = r_wd_start; r_wd >= 0; --r_wd) {
        uint32_t wd = col[r_wd];
        for (int q = 31; q >= 0; --q)
            if (wd & (1u << q)) return (r_wd > 0) ? (unsigned)((r_wd - 1) * 32 + q) : (unsigned)q;
    }
    return 0;
}

static inline void bm_toggle(bitmat* b, unsigned r, i64 c) { b->w[c * b->ldim + r / 32] ^= (1u << (r % 32)); }
static inline int bm_test(const bitmat* b, unsigned r, i64 c) { return (b->w[c * b->ldim + r / 32] >> (r % 32)) & 1u; }

/* BitMatrix = (Dense > 0) / (Dense < 0): bit_matrix_ops.hpp Apply(), tail masked */
static void bm_from_compare(bitmat* b, const double* X, i64 ldx, int greater)
{
#pragma omp parallel for schedule(static)
    for (i64 c = 0; c < b->width; ++c) {
        for (int q = 0; q < b->ldim; ++q) b->w[c * b->ldim + q] = 0;
        for (int r = 0; r < b->height; ++r) {
            double v = AT(X, ldx, r, c);
            int bit = greater ? (v > 0.0) : (v < 0.0);
            if (bit) b->w[c * b->ldim + r / 32] |= (1u << (r % 32));
        }
    }
}

/* ======================================================================== */
/* BppSolveNormalEqNoGroup: nmf_solver_bpp.hpp:146-219.                       */
/* cols[0..ncols) index into the full passive set; RHSsub/Xsub are k x ncols. */
/* ======================================================================== */
static int bpp_solve_normal_eq_nogroup(int k, i64 ncols, const i64* cols, const bitmat* passive,
                                       const double* LHS, int ldl, const double* RHSsub, i64 ldr,
                                       double* Xsub, i64 ldx)
{
    /* AllCols(passive_set, col_indices), bit_matrix_ops.cpp:57-72 */
    int all = 1;
    for (i64 c = 0; c < ncols; ++c)
        if (bm_colsum(passive, cols[c]) != k) { all = 0; break; }
    if (all) return solve_normal_eq_full(k, LHS, ldl, RHSsub, ldr, Xsub, ldx, ncols);

    for (i64 c = 0; c < ncols; ++c) memset(&AT(Xsub, ldx, 0, c), 0, (size_t)k * sizeof(double));
    int success = 1;
#pragma omp parallel
    {
        double* Lsub = (double*)malloc((size_t)k * k * sizeof(double));
        double* rsub = (double*)malloc((size_t)k * sizeof(double));
        int* ri = (int*)malloc((size_t)k * sizeof(int));
#pragma omp for schedule(dynamic, 16)
        for (i64 c = 0; c < ncols; ++c) {
            int nr = 0;
            for (int r = 0; r < k; ++r)
                if (bm_test(passive, (unsigned)r, cols[c])) ri[nr++] = r;   /* RowIndices */
            if (nr == 0) continue;
            for (int j = 0; j < nr; ++j)
                for (int i = 0; i < nr; ++i) Lsub[(size_t)j * nr + i] = AT(LHS, ldl, ri[i], ri[j]);
            for (int i = 0; i < nr; ++i) rsub[i] = AT(RHSsub, ldr, ri[i], c);
            if (!chol_upper(nr, Lsub, nr)) {
#pragma omp atomic write
                success = 0;
                continue;
            }
            chol_solve_upper(nr, Lsub, nr, rsub, nr, 1);
            for (int i = 0; i < nr; ++i) AT(Xsub, ldx, ri[i], c) = rsub[i];
        }
        free(Lsub); free(rsub); free(ri);
    }
    if (!success) fprintf(stderr, "Cholesky factorization failure - matrix was not symmetric positive-definite.\n");
    return success;
}

static void zeroize_small(i64 M, i64 N, double* X, i64 ldx, double tol)
{   /* dense_matrix_ops.hpp:371-394 */
#pragma omp parallel for schedule(static)
    for (i64 c = 0; c < N; ++c)
        for (i64 r = 0; r < M; ++r)
            if (fabs(AT(X, ldx, r, c)) < tol) AT(X, ldx, r, c) = 0.0;
}

/* ======================================================================== */
/* NnlsBlockpivot: nnls.hpp:144-244; UpdatePassiveSet: src/nnls.cpp:18-74;    */
/* BppUpdateSets: nnls.hpp:43-140.  Solves LHS*X = RHS, X >= 0 for all        */
/* columns; X holds the warm start on entry; Y = LHS*X - RHS on exit.         */
/* Returns 1 on success, 0 on failure (pivot limit 5k or non-SPD subproblem). */
/* `pivots_out` (optional) receives the number of outer pivoting rounds.      */
/* ======================================================================== */
int orc_nnls_blockpivot(int k, i64 ncols, const double* LHS, int ldl,
                        const double* RHS, i64 ldr, double* X, i64 ldx,
                        double* Y, i64 ldy, int* pivots_out)
{
    const int PBAR = 3;
    const unsigned MAX_ITER = (unsigned)k * 5u;
    int ok = 1;

    bitmat passive = bm_alloc(k, ncols);
    bitmat nonopt = bm_alloc(k, ncols);
    bitmat infeas = bm_alloc(k, ncols);
    bm_from_compare(&passive, X, ldx, 1);                       /* passive = (X > 0) */

    i64* idx = (i64*)malloc((size_t)(ncols > 0 ? ncols : 1) * sizeof(i64));
    for (i64 i = 0; i < ncols; ++i) idx[i] = i;
    int* P = (int*)malloc((size_t)(ncols > 0 ? ncols : 1) * sizeof(int));
    int* Ninf = (int*)malloc((size_t)(ncols > 0 ? ncols : 1) * sizeof(int));
    int* not_good = (int*)malloc((size_t)(ncols > 0 ? ncols : 1) * sizeof(int));
    double* RHSsub = (double*)malloc((size_t)k * (size_t)(ncols > 0 ? ncols : 1) * sizeof(double));
    double* Xsub = (double*)malloc((size_t)k * (size_t)(ncols > 0 ? ncols : 1) * sizeof(double));
    double* Ysub = (double*)malloc((size_t)k * (size_t)(ncols > 0 ? ncols : 1) * sizeof(double));
    unsigned iter = 0;

    for (i64 c = 0; c < ncols; ++c) memset(&AT(X, ldx, 0, c), 0, (size_t)k * sizeof(double));
    if (!bpp_solve_normal_eq_nogroup(k, ncols, idx, &passive, LHS, ldl, RHS, ldr, X, ldx)) { ok = 0; goto done; }

    /* Y = LHS*X - RHS */
    orc_gemm(0, 0, k, ncols, k, 1.0, LHS, ldl, X, ldx, 0.0, Y, ldy);
    mat_axpy(-1.0, k, ncols, RHS, ldr, Y, ldy);

    for (i64 c = 0; c < ncols; ++c) { P[c] = PBAR; Ninf[c] = k + 1; }

    /* nonopt = (Y<0) & ~passive ; infeas = (X<0) & passive */
    bm_from_compare(&nonopt, Y, ldy, 0);
    bm_from_compare(&infeas, X, ldx, 0);
    i64 n_nonopt_cols = 0;
    for (i64 c = 0; c < ncols; ++c) {
        int s = 0;
        for (int q = 0; q < passive.ldim; ++q) {
            uint32_t p = passive.w[c * passive.ldim + q];
            nonopt.w[c * nonopt.ldim + q] &= ~p;
            infeas.w[c * infeas.ldim + q] &= p;
            s += popcount32(nonopt.w[c * nonopt.ldim + q]) + popcount32(infeas.w[c * infeas.ldim + q]);
        }
        not_good[c] = s;
    }
    for (i64 c = 0; c < ncols; ++c)
        if (not_good[c] > 0) idx[n_nonopt_cols++] = c;          /* not_opt_cols.Find() */

    while (n_nonopt_cols > 0) {
        if (iter >= MAX_ITER) { ok = 0; goto done; }            /* nnls.hpp:195-196 */

        /* UpdatePassiveSet, src/nnls.cpp:18-74 (per non-optimal column) */
        for (i64 t = 0; t < n_nonopt_cols; ++t) {
            i64 c = idx[t];
            int rule;
            if (not_good[c] < Ninf[c]) rule = 1;                /* cols1 */
            else if (P[c] >= 1) rule = 2;                       /* cols2 */
            else rule = 3;                                      /* cols3 */
            if (rule == 1) { P[c] = PBAR; Ninf[c] = not_good[c]; }
            if (rule == 2) { P[c] -= 1; }
            if (rule == 1 || rule == 2) {
                for (int q = 0; q < passive.ldim; ++q) {
                    passive.w[c * passive.ldim + q] |= nonopt.w[c * nonopt.ldim + q];
                    passive.w[c * passive.ldim + q] &= ~infeas.w[c * infeas.ldim + q];
                }
            } else {
                unsigned r1 = bm_max_row_index(&nonopt, c);
                unsigned r2 = bm_max_row_index(&infeas, c);
                bm_toggle(&passive, r1 > r2 ? r1 : r2, c);
            }
        }

        /* gather RHS and X columns (SubmatrixFromCols) */
        for (i64 t = 0; t < n_nonopt_cols; ++t) {
            memcpy(&RHSsub[(size_t)t * k], &AT(RHS, ldr, 0, idx[t]), (size_t)k * sizeof(double));
            memcpy(&Xsub[(size_t)t * k], &AT(X, ldx, 0, idx[t]), (size_t)k * sizeof(double));
        }
        if (!bpp_solve_normal_eq_nogroup(k, n_nonopt_cols, idx, &passive, LHS, ldl, RHSsub, k, Xsub, k)) { ok = 0; goto done; }
        zeroize_small(k, n_nonopt_cols, Xsub, k, 1.0e-12);

        /* Ysub = LHS*Xsub - RHSsub */
        orc_gemm(0, 0, k, n_nonopt_cols, k, 1.0, LHS, ldl, Xsub, k, 0.0, Ysub, k);
        mat_axpy(-1.0, k, n_nonopt_cols, RHSsub, k, Ysub, k);

        for (i64 t = 0; t < n_nonopt_cols; ++t) {               /* OverwriteCols */
            memcpy(&AT(Y, ldy, 0, idx[t]), &Ysub[(size_t)t * k], (size_t)k * sizeof(double));
            memcpy(&AT(X, ldx, 0, idx[t]), &Xsub[(size_t)t * k], (size_t)k * sizeof(double));
        }
        zeroize_small(k, ncols, X, ldx, 1.0e-12);               /* whole matrices, nnls.hpp:224-225 */
        zeroize_small(k, ncols, Y, ldy, 1.0e-12);

        /* BppUpdateSets (masked by not_opt_mask: columns that were non-optimal
         * on entry to this round; all other columns get empty sets) */
        for (i64 c = 0; c < ncols; ++c) {
            for (int q = 0; q < passive.ldim; ++q) { nonopt.w[c * nonopt.ldim + q] = 0; infeas.w[c * infeas.ldim + q] = 0; }
            not_good[c] = 0;
        }
        for (i64 t = 0; t < n_nonopt_cols; ++t) {
            i64 c = idx[t];
            int s = 0;
            for (int r = 0; r < k; ++r) {
                int p = bm_test(&passive, (unsigned)r, c);
                if (!p && AT(Y, ldy, r, c) < 0.0) { nonopt.w[c * nonopt.ldim + r / 32] |= (1u << (r % 32)); ++s; }
                if (p && AT(X, ldx, r, c) < 0.0) { infeas.w[c * infeas.ldim + r / 32] |= (1u << (r % 32)); ++s; }
            }
            not_good[c] = s;
        }
        i64 nn = 0;
        for (i64 c = 0; c < ncols; ++c)
            if (not_good[c] > 0) idx[nn++] = c;
        n_nonopt_cols = nn;
        ++iter;
    }

done:
    if (pivots_out) *pivots_out = (int)iter;
    bm_free(&passive); bm_free(&nonopt); bm_free(&infeas);
    free(idx); free(P); free(Ninf); free(not_good); free(RHSsub); free(Xsub); free(Ysub);
    return ok;
}

/* ======================================================================== */
/* ProjectedGradientNorm: projected_gradient.hpp:125-171                      */
/* ======================================================================== */
static double pg_sum(i64 M, i64 N, const double* G, i64 ldg, const double* X, i64 ldx)
{
    double s = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : s)
    for (i64 c = 0; c < N; ++c) {
        double t = 0.0;
        for (i64 r = 0; r < M; ++r) {
            double g = AT(G, ldg, r, c);
            if (g < 0.0 || AT(X, ldx, r, c) > 0.0) t += g * g;
        }
        s += t;
    }
    return s;
}

double orc_projected_gradient_norm(i64 m, i64 n, int k, const double* gradW, i64 ldgw,
                                   const double* gradH, i64 ldgh, const double* W, i64 ldw,
                                   const double* H, i64 ldh)
{
    double nw = pg_sum(m, k, gradW, ldgw, W, ldw);
    double nh = pg_sum(k, n, gradH, ldgh, H, ldh);
    return sqrt(nw + nh);       /* NaN is reported by the caller (reference throws) */
}

/* ======================================================================== */
/* NormalizeAndScale: normalize.hpp:25-53, 90-114, 118-140                    */
/* returns 0 when a column norm is below DBL_EPSILON (reference throws).      */
/* ======================================================================== */
int orc_normalize_and_scale(i64 m, i64 n, int k, double* W, i64 ldw, double* H, i64 ldh)
{
    for (int c = 0; c < k; ++c) {
        double s = 0.0;
        for (i64 r = 0; r < m; ++r) s += AT(W, ldw, r, c) * AT(W, ldw, r, c);
        double nrm = sqrt(s);
        if (fabs(nrm) < DBL_EPSILON) return 0;
        double inv = 1.0 / nrm;
        for (i64 r = 0; r < m; ++r) AT(W, ldw, r, c) *= inv;
        for (i64 j = 0; j < n; ++j) AT(H, ldh, c, j) *= nrm;
    }
    return 1;
}

/* ======================================================================== */
/* Solver state shared by the three algorithms                                */
/* ======================================================================== */
typedef struct {
    i64 m, n; int k;
    const double* A; i64 lda;
    /* sparse A (NmfSparse, common/src/nmf.cpp:232-300): CSC of A and, for BPP, of A' (SparseMatrix   */
    /* Transpose, sparse_matrix_ops.hpp:36-127); A == NULL then                                        */
    const unsigned *cp, *ri; const double* va;
    unsigned *cpt, *rit; double* vat;
    double *WtW, *HHt;          /* k x k */
    double *WtA;                /* k x n */
    double *AHt;                /* m x k */
    double *T1;                 /* scratch: k x n (WtWH) */
    double *T2;                 /* scratch: m x k (WHHt) */
    /* BPP only */
    double *At;                 /* n x m */
    double *Wt, *gradWt, *HAt;  /* k x m */
} solver_ws;

static double* dalloc(size_t n) { return (double*)calloc(n ? n : 1, sizeof(double)); }

/* ---- the three products that touch A.  Dense: Gemm -> BLAS.  Sparse: the reference's own loops,     */
/* sparse_gemm_ba_impl.hpp (B'A and BA: one dot / axpy per stored entry, column by column of A) and    */
/* sparse_gemm_ab_impl.hpp (AB': scatter of every stored entry into row i of the result).              */
/* wall time spent in the big products of the last orc_nmf / orc_nmf_sparse call (bench.py prices the CPU baseline  */
/* from it: the rest of an iteration is NNLS / element-wise work that scales with m + n, the products with m n)     */
static double g_big_us = 0.0;
static int g_big_calls = 0;
void orc_big_product_time(double* seconds, int* calls)
{
    if (seconds) *seconds = g_big_us * 1e-6;
    if (calls) *calls = g_big_calls;
}
#define BIG_T0 const double big_t0_ = now_us()
#define BIG_T1 do { g_big_us += now_us() - big_t0_; g_big_calls += 1; } while (0)

static void prod_WtA(const solver_ws* s, const double* W, i64 ldw, double* out)      /* k x n, ld k */
{
    const int k = s->k;
    if (s->A) { BIG_T0; orc_gemm(1, 0, k, s->n, s->m, 1.0, W, ldw, s->A, s->lda, 0.0, out, k); BIG_T1; return; }
#pragma omp parallel for schedule(dynamic, 64)
    for (i64 j = 0; j < s->n; ++j) {
        double* o = out + j * k;
        for (int r = 0; r < k; ++r) o[r] = 0.0;
        for (unsigned p = s->cp[j]; p < s->cp[j + 1]; ++p) {
            const double v = s->va[p];
            const i64 i = s->ri[p];
            for (int r = 0; r < k; ++r) o[r] += v * W[r * ldw + i];
        }
    }
}

static void prod_AHt(const solver_ws* s, const double* H, i64 ldh, double* out)      /* m x k, ld m */
{
    const int k = s->k;
    if (s->A) { BIG_T0; orc_gemm(0, 1, s->m, k, s->n, 1.0, s->A, s->lda, H, ldh, 0.0, out, s->m); BIG_T1; return; }
    for (i64 e = 0; e < s->m * k; ++e) out[e] = 0.0;
    for (i64 j = 0; j < s->n; ++j)
        for (unsigned p = s->cp[j]; p < s->cp[j + 1]; ++p) {
            const double v = s->va[p];
            const i64 i = s->ri[p];
            for (int r = 0; r < k; ++r) out[r * s->m + i] += v * H[j * ldh + r];
        }
}

static void prod_HAt(const solver_ws* s, const double* H, i64 ldh, double* out)      /* k x m, ld k */
{
    const int k = s->k;
    if (s->A) { BIG_T0; orc_gemm(0, 0, k, s->m, s->n, 1.0, H, ldh, s->At, s->n, 0.0, out, k); BIG_T1; return; }
#pragma omp parallel for schedule(dynamic, 64)
    for (i64 i = 0; i < s->m; ++i) {
        double* o = out + i * k;
        for (int r = 0; r < k; ++r) o[r] = 0.0;
        for (unsigned p = s->cpt[i]; p < s->cpt[i + 1]; ++p) {
            const double v = s->vat[p];
            const i64 j = s->rit[p];
            for (int r = 0; r < k; ++r) o[r] += v * H[j * ldh + r];
        }
    }
}

static void sparse_transpose(solver_ws* s)
{
    const unsigned nnz = s->cp[s->n];
    s->cpt = (unsigned*)calloc((size_t)s->m + 1, sizeof(unsigned));
    s->rit = (unsigned*)malloc((size_t)(nnz ? nnz : 1) * sizeof(unsigned));
    s->vat = (double*)malloc((size_t)(nnz ? nnz : 1) * sizeof(double));
    for (unsigned p = 0; p < nnz; ++p) s->cpt[s->ri[p] + 1] += 1;
    for (i64 r = 0; r < s->m; ++r) s->cpt[r + 1] += s->cpt[r];
    unsigned* fill = (unsigned*)malloc((size_t)(s->m ? s->m : 1) * sizeof(unsigned));
    memcpy(fill, s->cpt, (size_t)s->m * sizeof(unsigned));
    for (i64 j = 0; j < s->n; ++j)
        for (unsigned p = s->cp[j]; p < s->cp[j + 1]; ++p) {
            const unsigned q = fill[s->ri[p]]++;
            s->rit[q] = (unsigned)j;
            s->vat[q] = s->va[p];
        }
    free(fill);
}

/* ---- MU: nmf_solver_mu.hpp:98-114 (Init), :121-164 (iteration), :27-71 ---- */
static void mu_init(solver_ws* s, const double* W, i64 ldw)
{
    prod_WtA(s, W, ldw, s->WtA);
    orc_gemm(1, 0, s->k, s->k, s->m, 1.0, W, ldw, W, ldw, 0.0, s->WtW, s->k);
}

static int mu_iter(solver_ws* s, double* W, i64 ldw, double* H, i64 ldh,
                   double* gradW, double* gradH)
{
    const double EPS = 1.0e-13;                                   /* SolverMU::EPSILON :22 */
    const i64 m = s->m, n = s->n; const int k = s->k;
    orc_gemm(0, 0, k, n, k, 1.0, s->WtW, k, H, ldh, 0.0, s->T1, k);          /* WtWH */
#pragma omp parallel for schedule(static)
    for (i64 c = 0; c < n; ++c)
        for (int r = 0; r < k; ++r)
            AT(H, ldh, r, c) *= (AT(s->WtA, k, r, c) / (AT(s->T1, k, r, c) + EPS));
    orc_gemm(0, 1, k, k, n, 1.0, H, ldh, H, ldh, 0.0, s->HHt, k);             /* HHt */
    prod_AHt(s, H, ldh, s->AHt);       /* AHt */
    orc_gemm(0, 0, m, k, k, 1.0, W, ldw, s->HHt, k, 0.0, s->T2, m);           /* WHHt */
#pragma omp parallel for schedule(static)
    for (int c = 0; c < k; ++c)
        for (i64 r = 0; r < m; ++r)
            AT(W, ldw, r, c) *= (AT(s->AHt, m, r, c) / (AT(s->T2, m, r, c) + EPS));
    prod_WtA(s, W, ldw, s->WtA);       /* WtA (new W) */
    orc_gemm(1, 0, k, k, m, 1.0, W, ldw, W, ldw, 0.0, s->WtW, k);             /* WtW */
    orc_gemm(0, 0, m, k, k, 1.0, W, ldw, s->HHt, k, 0.0, gradW, m);           /* gradW = W*HHt - AHt */
    mat_axpy(-1.0, m, k, s->AHt, m, gradW, m);
    orc_gemm(0, 0, k, n, k, 1.0, s->WtW, k, H, ldh, 0.0, gradH, k);           /* gradH = WtW*H - WtA */
    mat_axpy(-1.0, k, n, s->WtA, k, gradH, k);
    return 1;
}

/* ---- HALS: nmf_solver_hals.hpp:142-159 (Init), :166-199, :66-117, :26-62 -- */
static void hals_init(solver_ws* s, const double* H, i64 ldh)
{
    orc_gemm(0, 1, s->k, s->k, s->n, 1.0, H, ldh, H, ldh, 0.0, s->HHt, s->k);
    prod_AHt(s, H, ldh, s->AHt);
}

static void hals_update_w(solver_ws* s, double* W, i64 ldw)
{
    const i64 m = s->m; const int k = s->k;
    double* whht = s->T2;                                   /* m x 1 column */
    for (int c = 0; c < k; ++c) {
        /* WHHt_c = W * HHt(:,c) using the CURRENT W (Gauss-Seidel) */
#pragma omp parallel for schedule(static)
        for (i64 r = 0; r < m; ++r) {
            double acc = 0.0;
            for (int j = 0; j < k; ++j) acc += AT(W, ldw, r, j) * AT(s->HHt, k, j, c);
            whht[r] = acc;
        }
        const double hcc = AT(s->HHt, k, c, c);
        i64 num_zeros = 0;
        double ss = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : num_zeros, ss)
        for (i64 r = 0; r < m; ++r) {
            double w = AT(W, ldw, r, c) + (AT(s->AHt, m, r, c) - whht[r]) / hcc;
            if (isnan(w) || w < 0.0) { w = 0.0; ++num_zeros; }
            AT(W, ldw, r, c) = w;
            ss += w * w;
        }
        if (num_zeros == m) {                               /* all-zero column guard :105-111 */
            for (i64 r = 0; r < m; ++r) AT(W, ldw, r, c) = DBL_EPSILON;
            ss = (double)m * DBL_EPSILON * DBL_EPSILON;
        }
        double inv = 1.0 / sqrt(ss);                        /* Norm(W_c, FROBENIUS), Scal :113-115 */
#pragma omp parallel for schedule(static)
        for (i64 r = 0; r < m; ++r) AT(W, ldw, r, c) *= inv;
    }
}

static void hals_update_h(solver_ws* s, double* H, i64 ldh)
{
    const i64 n = s->n; const int k = s->k;
    /* Rows are updated in order r = 0..k-1, each using the current H; the
     * update of row r touches each column independently, so looping columns
     * outermost is the same computation (:26-62). */
#pragma omp parallel for schedule(static)
    for (i64 c = 0; c < n; ++c) {
        double* h = &AT(H, ldh, 0, c);
        for (int r = 0; r < k; ++r) {
            double acc = 0.0;
            for (int q = 0; q < k; ++q) acc += AT(s->WtW, k, r, q) * h[q];
            double v = h[r] + (AT(s->WtA, k, r, c) - acc) / AT(s->WtW, k, r, r);
            if (isnan(v) || v < 0.0) v = 0.0;
            h[r] = v;
        }
    }
}

static int hals_iter(solver_ws* s, double* W, i64 ldw, double* H, i64 ldh,
                     double* gradW, double* gradH)
{
    const i64 m = s->m, n = s->n; const int k = s->k;
    hals_update_w(s, W, ldw);
    orc_gemm(1, 0, k, k, m, 1.0, W, ldw, W, ldw, 0.0, s->WtW, k);
    prod_WtA(s, W, ldw, s->WtA);
    hals_update_h(s, H, ldh);
    orc_gemm(0, 0, k, n, k, 1.0, s->WtW, k, H, ldh, 0.0, gradH, k);
    mat_axpy(-1.0, k, n, s->WtA, k, gradH, k);
    orc_gemm(0, 1, k, k, n, 1.0, H, ldh, H, ldh, 0.0, s->HHt, k);
    prod_AHt(s, H, ldh, s->AHt);
    orc_gemm(0, 0, m, k, k, 1.0, W, ldw, s->HHt, k, 0.0, gradW, m);
    mat_axpy(-1.0, m, k, s->AHt, m, gradW, m);
    return 1;
}

/* ---- BPP: nmf_solver_bpp.hpp:310-335 (Init), :342-377 (iteration) --------- */
static void bpp_init(solver_ws* s, const double* W, i64 ldw)
{
    if (s->A) mat_transpose(s->m, s->n, s->A, s->lda, s->At, s->n);
    else sparse_transpose(s);
    orc_gemm(1, 0, s->k, s->k, s->m, 1.0, W, ldw, W, ldw, 0.0, s->WtW, s->k);
    prod_WtA(s, W, ldw, s->WtA);
    mat_transpose(s->m, s->k, W, ldw, s->Wt, s->k);
}

static int bpp_iter(solver_ws* s, double* W, i64 ldw, double* H, i64 ldh,
                    double* gradW, double* gradH)
{
    const i64 m = s->m, n = s->n; const int k = s->k;
    if (!orc_nnls_blockpivot(k, n, s->WtW, k, s->WtA, k, H, ldh, gradH, k, NULL)) return 0;
    orc_gemm(0, 1, k, k, n, 1.0, H, ldh, H, ldh, 0.0, s->HHt, k);
    prod_HAt(s, H, ldh, s->HAt);
    if (!orc_nnls_blockpivot(k, m, s->HHt, k, s->HAt, k, s->Wt, k, s->gradWt, k, NULL)) return 0;
    mat_transpose(k, m, s->Wt, k, W, ldw);
    mat_transpose(k, m, s->gradWt, k, gradW, m);
    orc_gemm(1, 0, k, k, m, 1.0, W, ldw, W, ldw, 0.0, s->WtW, k);
    prod_WtA(s, W, ldw, s->WtA);
    orc_gemm(0, 0, k, n, k, 1.0, s->WtW, k, H, ldh, 0.0, gradH, k);
    mat_axpy(-1.0, k, n, s->WtA, k, gradH, k);
    return 1;
}

/* ---- RANK2: nmf_solver_rank2.hpp:25-135 (SystemSolveH), :139-212 (SystemSolveW), -------- */
/*      :216-318 (OptimalActiveSetH/W), :323-461 (Init / iteration)                       */
/* side 0: solve G * x = b for every column of B (2 x n), X is 2 x n (ld ldx)              */
/* side 1: solve x' * G = b' for every row; stored here as columns too (X, B are 2 x m)    */
static int rank2_system_solve(int side, i64 N, double* X, i64 ldx, const double* G /*2x2, ld 2*/,
                              const double* B, i64 ldb)
{
    const double eps = DBL_EPSILON;
    const double a00 = G[0], a10 = G[1], a01 = G[2], a11 = G[3];
    if (fabs(a00) < eps && fabs(a01) < eps) { fprintf(stderr, "SystemSolve%c: singular matrix\n", side ? 'W' : 'H'); return 0; }
    double a2, b2, d2, t;
    const int cosine = fabs(a00) >= fabs(a01);
    if (side == 0) {
        if (cosine) { t = -a10 / a00; a2 = a00 - t * a10; b2 = a01 - t * a11; d2 = a11 + t * a01; }
        else        { t = -a00 / a10; a2 = -a10 + t * a00; b2 = -a11 + t * a01; d2 = a01 + t * a11; }
    } else {
        if (cosine) { t = a01 / a00; a2 = a00 + t * a01; b2 = a10 + t * a11; d2 = a11 - t * a10; }
        else        { t = a00 / a01; a2 = -a01 - t * a00; b2 = -a11 - t * a10; d2 = a10 - t * a11; }
    }
    const double inv_a2 = 1.0 / a2, inv_d2 = 1.0 / d2;
    if (fabs(d2 / a2) < eps) return 0;
#pragma omp parallel for schedule(static)
    for (i64 i = 0; i < N; ++i) {
        const double b0 = AT(B, ldb, 0, i), b1 = AT(B, ldb, 1, i);
        double e2, f2;
        if (side == 0) {
            if (cosine) { e2 = b0 - t * b1; f2 = b1 + t * b0; }
            else        { e2 = -b1 + t * b0; f2 = b0 + t * b1; }
        } else {
            if (cosine) { e2 = b0 + t * b1; f2 = b1 - t * b0; }
            else        { e2 = -b1 - t * b0; f2 = b0 - t * b1; }
        }
        const double x1 = f2 * inv_d2;
        AT(X, ldx, 1, i) = x1;
        AT(X, ldx, 0, i) = (e2 - b2 * x1) * inv_a2;
    }
    return 1;
}

/* OptimalActiveSetH / W (:216-318): columns whose unconstrained solution has a non-positive
 * entry are replaced by the better of the two single-variable solutions */
static void rank2_optimal_active_set(i64 N, double* X, i64 ldx, const double* G, const double* B, i64 ldb)
{
    const double g00 = G[0], g11 = G[3];
    const double inv0 = 1.0 / g00, inv1 = 1.0 / g11, sq0 = sqrt(g00), sq1 = sqrt(g11);
#pragma omp parallel for schedule(static)
    for (i64 i = 0; i < N; ++i) {
        double v1 = AT(B, ldb, 0, i) * inv0, v2 = AT(B, ldb, 1, i) * inv1;
        const double vv1 = v1 * sq0, vv2 = v2 * sq1;
        if (vv1 >= vv2) v2 = 0.0; else v1 = 0.0;
        if (AT(X, ldx, 0, i) <= 0.0 || AT(X, ldx, 1, i) <= 0.0) { AT(X, ldx, 0, i) = v1; AT(X, ldx, 1, i) = v2; }
    }
}

static void rank2_init(solver_ws* s, const double* W, i64 ldw)
{
    orc_gemm(1, 0, 2, 2, s->m, 1.0, W, ldw, W, ldw, 0.0, s->WtW, 2);
    prod_WtA(s, W, ldw, s->WtA);
}

static int rank2_iter(solver_ws* s, double* W, i64 ldw, double* H, i64 ldh, double* gradW, double* gradH)
{
    const i64 m = s->m, n = s->n;
    if (!rank2_system_solve(0, n, H, ldh, s->WtW, s->WtA, 2)) return 0;
    rank2_optimal_active_set(n, H, ldh, s->WtW, s->WtA, 2);
    orc_gemm(0, 1, 2, 2, n, 1.0, H, ldh, H, ldh, 0.0, s->HHt, 2);
    prod_AHt(s, H, ldh, s->AHt);
    /* W side works on rows of W; reuse the column routines on W' (2 x m) */
    double* Wt = s->T1;        /* 2 x m scratch (T1 has k*n >= ? doubles: allocated max(k*n, k*m) below) */
    double* Bt = s->T2;        /* 2 x m */
    mat_transpose(m, 2, s->AHt, m, Bt, 2);
    if (!rank2_system_solve(1, m, Wt, 2, s->HHt, Bt, 2)) return 0;
    /* OptimalActiveSetW tests the freshly solved W */
    rank2_optimal_active_set(m, Wt, 2, s->HHt, Bt, 2);
    mat_transpose(2, m, Wt, 2, W, ldw);
    /* NormalizeAndScale(W, H, ScaleFactors) every iteration (:418) */
    double sf[2];
    for (int c = 0; c < 2; ++c) {
        double ss = 0.0;
        for (i64 r = 0; r < m; ++r) ss += AT(W, ldw, r, c) * AT(W, ldw, r, c);
        const double nrm = sqrt(ss);
        if (fabs(nrm) < DBL_EPSILON) return -1;              /* reference throws runtime_error */
        const double inv = 1.0 / nrm;
        for (i64 r = 0; r < m; ++r) AT(W, ldw, r, c) *= inv;
        sf[c] = nrm;
    }
    for (i64 j = 0; j < n; ++j) { AT(H, ldh, 0, j) *= sf[0]; AT(H, ldh, 1, j) *= sf[1]; }
    /* keep HHt and AHt consistent with the rescaled H (:424-437) */
    const double e00 = s->HHt[0], e01 = s->HHt[2], e11 = s->HHt[3];
    s->HHt[0] = e00 * sf[0] * sf[0]; s->HHt[2] = e01 * sf[0] * sf[1]; s->HHt[1] = e01 * sf[0] * sf[1]; s->HHt[3] = e11 * sf[1] * sf[1];
    for (int c = 0; c < 2; ++c)
        for (i64 r = 0; r < m; ++r) AT(s->AHt, m, r, c) *= sf[c];
    orc_gemm(0, 0, m, 2, 2, 1.0, W, ldw, s->HHt, 2, 0.0, gradW, m);
    mat_axpy(-1.0, m, 2, s->AHt, m, gradW, m);
    orc_gemm(1, 0, 2, 2, m, 1.0, W, ldw, W, ldw, 0.0, s->WtW, 2);
    prod_WtA(s, W, ldw, s->WtA);
    orc_gemm(0, 0, 2, n, 2, 1.0, s->WtW, 2, H, ldh, 0.0, gradH, 2);
    mat_axpy(-1.0, 2, n, s->WtA, 2, gradH, 2);
    return 1;
}

/* ======================================================================== */
/* IsValid: common/src/nmf_options.cpp:23-112                                 */
/* ======================================================================== */
int orc_is_valid(const orc_options* o)
{
    if (o->k <= 0) return 0;
    if (o->height <= 0 || o->width <= 0) return 0;
    if (o->k > o->width) return 0;
    if (o->tol <= 0.0 || o->tol >= 1.0) return 0;
    if (o->min_iter <= 0 || o->max_iter <= 0 || o->tolcount <= 0) return 0;
    if (o->algorithm != ORC_MU && o->algorithm != ORC_HALS && o->algorithm != ORC_RANK2 && o->algorithm != ORC_BPP) return 0;
    if (o->algorithm == ORC_RANK2 && o->k != 2) return 0;
    if (o->prog_est_algorithm != ORC_PG_RATIO && o->prog_est_algorithm != ORC_DELTA_FNORM) return 0;
    return 1;
}

/* ======================================================================== */
/* Nmf() + RunNmf() + NmfSolve<>: common/src/nmf.cpp:173-229, :55-111;        */
/* common/include/nmf_solve_generic.hpp:34-140; progress estimators           */
/* progress_estimator_generic.hpp:30-69 (DeltaW), :74-109 (PgRatio).          */
/* W (m x k) and H (k x n) are in/out.  `metrics` (optional, length max_iter) */
/* receives the progress metric of every iteration that computed one (NaN     */
/* elsewhere).             */
/* ======================================================================== */
static int nmf_driver(const orc_options* o, const double* A, i64 lda, const unsigned* cp, const unsigned* ri,
                      const double* va, double* W, i64 ldw, double* H, i64 ldh, orc_stats* stats, double* metrics);

int orc_nmf(const orc_options* o, const double* A, i64 lda, double* W, i64 ldw,
            double* H, i64 ldh, orc_stats* stats, double* metrics)
{
    if (!A) return ORC_BAD_PARAM;
    return nmf_driver(o, A, lda, NULL, NULL, NULL, W, ldw, H, ldh, stats, metrics);
}

/* NmfSparse (common/src/nmf.cpp:232-300): the same driver on a CSC matrix (32-bit indices as in the    */
/* reference); only the three products that touch A differ.                                             */
int orc_nmf_sparse(const orc_options* o, const unsigned* col_offsets, const unsigned* row_indices, const double* data,
                   double* W, i64 ldw, double* H, i64 ldh, orc_stats* stats, double* metrics)
{
    if (!col_offsets || !row_indices || !data) return ORC_BAD_PARAM;
    return nmf_driver(o, NULL, o->height, col_offsets, row_indices, data, W, ldw, H, ldh, stats, metrics);
}

static int nmf_driver(const orc_options* o, const double* A, i64 lda, const unsigned* cp, const unsigned* ri,
                      const double* va, double* W, i64 ldw, double* H, i64 ldh, orc_stats* stats, double* metrics)
{
    if (!orc_is_valid(o)) return ORC_BAD_PARAM;
    const i64 m = o->height, n = o->width; const int k = o->k;
    if ((uint64_t)m * (uint64_t)k > 0x7FFFFFFFull) return ORC_SIZE_TOO_LARGE;   /* nmf.cpp:194-210 */
    if ((uint64_t)n * (uint64_t)k > 0x7FFFFFFFull) return ORC_SIZE_TOO_LARGE;
    if (ldw < m || ldh < k || lda < m) return ORC_BAD_PARAM;   /* reference throws logic_error :213-219 */
#ifdef _OPENMP
    if (o->max_threads > 0) omp_set_num_threads(o->max_threads);   /* thread_utils.hpp:34-41 */
#endif

    solver_ws s;
    memset(&s, 0, sizeof(s));
    s.m = m; s.n = n; s.k = k; s.A = A; s.lda = lda;
    s.cp = cp; s.ri = ri; s.va = va;
    s.WtW = dalloc((size_t)k * k); s.HHt = dalloc((size_t)k * k);
    s.WtA = dalloc((size_t)k * n); s.AHt = dalloc((size_t)m * k);
    s.T1 = dalloc((size_t)k * (n > m ? n : m));  s.T2 = dalloc((size_t)m * k);
    if (o->algorithm == ORC_BPP) {
        s.At = A ? dalloc((size_t)m * n) : NULL; s.Wt = dalloc((size_t)k * m);
        s.gradWt = dalloc((size_t)k * m); s.HAt = dalloc((size_t)k * m);
    }
    double* gradH = dalloc((size_t)k * n);
    double* gradW = dalloc((size_t)m * k);
    double* Wprev = NULL;
    double pg0 = 1.0;
    if (metrics) for (int i = 0; i < o->max_iter; ++i) metrics[i] = NAN;

    double t0 = now_us();
    g_big_us = 0.0; g_big_calls = 0;

    /* solver.Init, progress_est->Init */
    if (o->algorithm == ORC_MU) mu_init(&s, W, ldw);
    else if (o->algorithm == ORC_HALS) hals_init(&s, H, ldh);
    else if (o->algorithm == ORC_RANK2) rank2_init(&s, W, ldw);
    else bpp_init(&s, W, ldw);
    if (o->prog_est_algorithm == ORC_DELTA_FNORM) {
        /* DeltaW::Init: Wprev = 0; Compute(W) -> Wprev = W (:38-45, :58-69) */
        Wprev = dalloc((size_t)m * k);
        mat_copy(m, k, W, ldw, Wprev, m);
    }

    int success = 0, result = ORC_OK;
    int iter = 0, success_count = 0;
    for (iter = 0; iter < o->max_iter; ++iter) {
        int ok;
        if (o->algorithm == ORC_MU) ok = mu_iter(&s, W, ldw, H, ldh, gradW, gradH);
        else if (o->algorithm == ORC_HALS) ok = hals_iter(&s, W, ldw, H, ldh, gradW, gradH);
        else if (o->algorithm == ORC_RANK2) ok = rank2_iter(&s, W, ldw, H, ldh, gradW, gradH);
        else ok = bpp_iter(&s, W, ldw, H, ldh, gradW, gradH);
        if (ok <= 0) {
            fprintf(stderr, "\tNMF solver failure on iteration %d\n", iter + 1);
            result = ORC_FAILURE;
            goto finish;
        }

        int do_update = (iter >= o->min_iter) || (iter == 0);
        double metric = 1.0;
        if (do_update) {
            if (o->prog_est_algorithm == ORC_DELTA_FNORM) {
                /* Wprev = Wprev - W ; ratio = |Wprev|_F / |W|_F ; Wprev = W */
                mat_axpy(-1.0, m, k, W, ldw, Wprev, m);
                double nd = orc_fnorm(m, k, Wprev, m);
                double nc = orc_fnorm(m, k, W, ldw);
                metric = nd / nc;
                mat_copy(m, k, W, ldw, Wprev, m);
            } else {
                double pg = orc_projected_gradient_norm(m, n, k, gradW, m, gradH, k, W, ldw, H, ldh);
                if (isnan(pg)) { result = ORC_FAILURE; goto finish; }     /* reference throws */
                if (iter == 0) { pg0 = pg; metric = 1.0; }
                else metric = pg / pg0;
            }
            if (metrics) metrics[iter] = metric;
        }
        if (iter < o->min_iter) {
            if (o->verbose) printf("%d:\tprogress metric: \t(min_iter)\n", iter + 1);
            continue;
        }
        if (o->verbose && ((iter + 1 < 10) || ((iter + 1) % 10 == 0)))
            printf("%d:\tprogress metric:\t%g\n", iter + 1, metric);      /* nmf_progress_estimation.hpp:22-33 */
        if (metric <= o->tol) {
            if (++success_count >= o->tolcount) {
                success = 1;
                if (o->verbose) printf("\nSolution converged after %d iterations.\n\n", iter + 1);
                break;
            }
        } else {
            success_count = 0;
        }
    }

    if (o->normalize) {
        if (!orc_normalize_and_scale(m, n, k, W, ldw, H, ldh)) { result = ORC_FAILURE; goto finish; }
    }
    if (!success && iter == o->max_iter) success = 1;
    result = success ? ORC_OK : ORC_FAILURE;

finish:
    if (stats) {
        stats->elapsed_us = (unsigned long long)(now_us() - t0);
        stats->iteration_count = iter;
    }
    free(s.WtW); free(s.HHt); free(s.WtA); free(s.AHt); free(s.T1); free(s.T2);
    free(s.At); free(s.Wt); free(s.gradWt); free(s.HAt);
    free(s.cpt); free(s.rit); free(s.vat);
    free(gradH); free(gradW); free(Wprev);
    return result;
}

void orc_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
