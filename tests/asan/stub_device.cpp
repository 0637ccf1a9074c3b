// tests/asan/stub_device.cpp -- TEST INFRASTRUCTURE ONLY.
// A host-only stand-in for the device half of libsmallk_amd (solver.cpp + the .hip files) so that the ~4000 lines of
// host C++ above it -- facade.cpp (namespace smallk, ::Nmf, CSV / MatrixMarket I/O, flat API), hierclust.cpp (tree
// search, priority scores, writers), flatclust.cpp and the three command line tools -- can be compiled with
// -fsanitize=address,undefined and run on a machine without a GPU (tests/asan/Makefile, tests/test_asan_host.py).
// Every factorisation is delegated to the CPU oracle (oracle/nmf_oracle.c); nothing here ships in the product.
#include "../../include/smallk_amd.h"
#include "../../smallk_amd/csrc/common.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

extern "C" {
typedef struct { double tol; int algorithm, prog_est_algorithm, height, width, k, min_iter, max_iter, tolcount, max_threads, verbose, normalize; } orc_options;
typedef struct { unsigned long long elapsed_us; int iteration_count; } orc_stats;
int orc_nmf(const orc_options*, const double* A, int64_t lda, double* W, int64_t ldw, double* H, int64_t ldh, orc_stats*, double* metrics);
int orc_nmf_sparse(const orc_options*, const unsigned* cp, const unsigned* ri, const double* va, double* W, int64_t ldw, double* H,
                   int64_t ldh, orc_stats*, double* metrics);
void orc_fill_uniform(double* buf, int64_t ld, int64_t rows, int64_t cols, int64_t r0, int64_t c0, int64_t gheight, uint64_t seed, int quant);
int orc_normalize_and_scale(int64_t m, int64_t n, int k, double* W, int64_t ldw, double* H, int64_t ldh);
}

static thread_local std::string g_err;
static bool g_init = false;
namespace smk {
void set_error(const std::string& m) { g_err = m; }
int device_sort_desc(const double* const*, int* const*, double* const*, int, i64, hipStream_t) { return -1; }   // host sorts
int device_priority_score(const double*, const double*, i64, i64, double*, hipStream_t) { return -1; }          // host arithmetic
void device_priority_release() {}
}

struct smk_matrix {
    int64_t m = 0, n = 0;
    bool sparse = false;
    std::vector<double> dense;                       // m x n column-major
    std::vector<unsigned> cp, ri;
    std::vector<double> va;
};
struct smk_solver {
    smk_options o;
    const smk_matrix* a;
    std::vector<double> W, H;                        // m x k, k x n
};

static orc_options to_orc(const smk_options& o)
{
    return orc_options{o.tol, o.algorithm, o.prog_est_algorithm, o.height, o.width, o.k, o.min_iter, o.max_iter, o.tolcount,
                       o.max_threads, o.verbose, o.normalize};
}

extern "C" {

int smk_initialize(int) { g_init = true; return SMK_OK; }
int smk_is_initialized(void) { return g_init ? SMK_INITIALIZED : SMK_NOTINITIALIZED; }
void smk_finalize(void) { g_init = false; }
const char* smk_last_error(void) { return g_err.c_str(); }
int smk_device_cu_count(void) { return 1; }
int smk_set_stream(void*) { return SMK_OK; }
// a second "device" is just a second host thread here: the two-device HierNMF2 step runs under the sanitizers too
int smk_thread_context_begin(int) { return SMK_OK; }
void smk_thread_context_end(void) {}
int smk_device_count(void) { return 1; }
int smk_current_device(void) { return 0; }
int smk_device_synchronize(void) { return 0; }
int smk_matrix_clone(const smk_matrix* src, smk_matrix** out)
{
    if (!src || !out) return SMK_BAD_PARAM;
    *out = new smk_matrix(*src);
    return SMK_OK;
}

int smk_is_valid(const smk_options* o, int vm)
{   // same checks and messages as solver.cpp / nmf_options.cpp:23-112
    if (!o) return 0;
    if (o->k <= 0) { fprintf(stderr, "nmflib error: k-value must be a positive integer\n"); return 0; }
    if (vm) {
        if (o->height <= 0) { fprintf(stderr, "nmflib error: matrix height must be a positive integer\n"); return 0; }
        if (o->width <= 0) { fprintf(stderr, "nmflib error: matrix width must be a positive integer\n"); return 0; }
        if (o->k > o->width) { fprintf(stderr, "nmflib error: k value cannot exceed the number of columns\n"); return 0; }
    }
    if (o->tol <= 0.0 || o->tol >= 1.0) { fprintf(stderr, "nmflib error: tolerance must be in the interval (0.0, 1.0)\n"); return 0; }
    if (o->min_iter <= 0 || o->max_iter <= 0 || o->tolcount <= 0) { fprintf(stderr, "nmflib error: iteration counts must be positive\n"); return 0; }
    if (o->algorithm < 0 || o->algorithm > 3) { fprintf(stderr, "nmflib error: unknown NMF algorithm specified\n"); return 0; }
    if (o->algorithm == SMK_ALG_RANK2 && o->k != 2) { fprintf(stderr, "nmflib error: RANK2 algorithm requires k == 2\n"); return 0; }
    if (o->prog_est_algorithm != 0 && o->prog_est_algorithm != 1) { fprintf(stderr, "nmflib error: unknown stopping criterion specified\n"); return 0; }
    return 1;
}

void smk_uniform_fill_host(double* buf, int64_t ld, int64_t rows, int64_t cols, int64_t r0, int64_t c0, int64_t gh, uint64_t seed, int quant)
{
    orc_fill_uniform(buf, ld, rows, cols, r0, c0, gh, seed, quant);
}

int smk_matrix_create(smk_matrix** out, int64_t h, int64_t wg, int64_t c0, int64_t nc, int)
{
    if (!out || h <= 0 || wg <= 0 || nc <= 0 || c0 != 0 || nc != wg) return SMK_BAD_PARAM;
    smk_matrix* a = new smk_matrix;
    a->m = h; a->n = nc;
    a->dense.assign((size_t)h * nc, 0.0);
    *out = a;
    return SMK_OK;
}
int smk_matrix_upload_f64(smk_matrix* a, const double* host, int64_t ld)
{
    if (!a || !host || ld < a->m || a->sparse) return SMK_BAD_PARAM;
    for (int64_t c = 0; c < a->n; ++c)
        for (int64_t r = 0; r < a->m; ++r) a->dense[(size_t)c * a->m + r] = (double)(float)host[(size_t)c * ld + r];   // fp32 storage
    return SMK_OK;
}
int smk_matrix_create_sparse(smk_matrix** out, int64_t h, int64_t wg, int64_t c0, int64_t nc, int64_t nnz, const unsigned* cp,
                             const unsigned* ri, const double* va)
{
    if (!out || h <= 0 || wg <= 0 || nc != wg || c0 != 0 || nnz < 0 || !cp) return SMK_BAD_PARAM;
    smk_matrix* a = new smk_matrix;
    a->m = h; a->n = nc; a->sparse = true;
    a->cp.assign(cp, cp + nc + 1);
    a->ri.assign(ri, ri + nnz);
    a->va.assign(va, va + nnz);
    for (unsigned r : a->ri)
        if ((int64_t)r >= h) { delete a; g_err = "row index out of range"; return SMK_BAD_PARAM; }
    *out = a;
    return SMK_OK;
}
void smk_matrix_destroy(smk_matrix* a) { delete a; }
int64_t smk_matrix_nnz(const smk_matrix* a) { return a ? (int64_t)a->va.size() : 0; }

int smk_matrix_gather_cols(const smk_matrix* src, const unsigned* cols, int64_t ncols, smk_matrix** out, unsigned* n2o, int64_t* nh)
{
    if (!out) return SMK_BAD_PARAM;
    *out = nullptr;
    if (!src || !cols || ncols <= 0) { g_err = "SubMatrixColsCompact: empty column set"; return SMK_BAD_PARAM; }
    for (int64_t j = 0; j < ncols; ++j)
        if ((int64_t)cols[j] >= src->n) { g_err = "SubMatrixColsCompact: column index out of range"; return SMK_BAD_PARAM; }
    smk_matrix* a = new smk_matrix;
    a->n = ncols;
    if (!src->sparse) {
        a->m = src->m;
        a->dense.resize((size_t)src->m * ncols);
        for (int64_t j = 0; j < ncols; ++j)
            std::copy(src->dense.begin() + (size_t)cols[j] * src->m, src->dense.begin() + (size_t)(cols[j] + 1) * src->m,
                      a->dense.begin() + (size_t)j * src->m);
        if (n2o) for (int64_t r = 0; r < src->m; ++r) n2o[r] = (unsigned)r;
        if (nh) *nh = src->m;
        *out = a;
        return SMK_OK;
    }
    const unsigned UNUSED = 0xFFFFFFFFu;
    std::vector<unsigned> o2n((size_t)src->m, UNUSED);
    size_t total = 0;
    for (int64_t j = 0; j < ncols; ++j)
        for (unsigned p = src->cp[cols[j]]; p < src->cp[cols[j] + 1]; ++p, ++total) o2n[src->ri[p]] = 0;
    if (total == 0) { delete a; g_err = "SparseMatrix::SubMatrixColsCompact: submatrix is the zero matrix"; return SMK_BAD_PARAM; }
    int64_t h = 0;
    for (int64_t r = 0; r < src->m; ++r)
        if (o2n[(size_t)r] != UNUSED) { o2n[(size_t)r] = (unsigned)h; if (n2o) n2o[h] = (unsigned)r; ++h; }
    a->sparse = true; a->m = h;
    a->cp.resize((size_t)ncols + 1);
    for (int64_t j = 0; j < ncols; ++j) {
        a->cp[(size_t)j] = (unsigned)a->ri.size();
        for (unsigned p = src->cp[cols[j]]; p < src->cp[cols[j] + 1]; ++p) { a->ri.push_back(o2n[src->ri[p]]); a->va.push_back(src->va[p]); }
    }
    a->cp[(size_t)ncols] = (unsigned)a->ri.size();
    if (nh) *nh = h;
    *out = a;
    return SMK_OK;
}

int smk_solver_create(smk_solver** out, const smk_options* o, const smk_matrix* a)
{
    if (!out || !o || !a) return SMK_BAD_PARAM;
    *out = nullptr;
    if (!g_init) return SMK_NOTINITIALIZED;
    if (!smk_is_valid(o, 1)) return SMK_BAD_PARAM;
    if (o->height != a->m || o->width != a->n) { g_err = "options/matrix dimension mismatch"; return SMK_BAD_PARAM; }
    smk_solver* s = new smk_solver;
    s->o = *o; s->a = a;
    *out = s;
    return SMK_OK;
}
void smk_solver_destroy(smk_solver* s) { delete s; }
int smk_solver_set_factors(smk_solver* s, const double* W0, int64_t ldW, const double* H0, int64_t ldH)
{
    if (!s || !W0 || !H0 || ldW < s->a->m || ldH < s->o.k) return SMK_BAD_PARAM;
    const int64_t m = s->a->m, n = s->a->n, k = s->o.k;
    s->W.resize((size_t)m * k); s->H.resize((size_t)k * n);
    for (int64_t c = 0; c < k; ++c) std::copy(W0 + c * ldW, W0 + c * ldW + m, s->W.begin() + c * m);
    for (int64_t c = 0; c < n; ++c) std::copy(H0 + c * ldH, H0 + c * ldH + k, s->H.begin() + c * k);
    return SMK_OK;
}
int smk_solver_set_factors_uniform(smk_solver* s, uint64_t seed_w, uint64_t seed_h)
{
    if (!s) return SMK_BAD_PARAM;
    const int64_t m = s->a->m, n = s->a->n, k = s->o.k;
    s->W.resize((size_t)m * k); s->H.resize((size_t)k * n);
    orc_fill_uniform(s->W.data(), m, m, k, 0, 0, m, seed_w, 0);
    orc_fill_uniform(s->H.data(), k, k, n, 0, 0, k, seed_h, 0);
    return SMK_OK;
}
int smk_solver_run(smk_solver* s, smk_stats* st)
{
    if (!s || s->W.empty()) return SMK_BAD_PARAM;
    orc_options o = to_orc(s->o);
    orc_stats os{0, 0};
    const int64_t m = s->a->m, k = s->o.k;
    int rc = s->a->sparse ? orc_nmf_sparse(&o, s->a->cp.data(), s->a->ri.data(), s->a->va.data(), s->W.data(), m, s->H.data(), k, &os, nullptr)
                          : orc_nmf(&o, s->a->dense.data(), m, s->W.data(), m, s->H.data(), k, &os, nullptr);
    if (st) { st->elapsed_us = os.elapsed_us; st->iteration_count = os.iteration_count; }
    return rc;
}
int smk_solver_get_factors(smk_solver* s, int, double* W, int64_t ldW, double* H, int64_t ldH)
{
    if (!s || !W || !H) return SMK_BAD_PARAM;
    const int64_t m = s->a->m, n = s->a->n, k = s->o.k;
    for (int64_t c = 0; c < k; ++c) std::copy(s->W.begin() + c * m, s->W.begin() + (c + 1) * m, W + c * ldW);
    for (int64_t c = 0; c < n; ++c) std::copy(s->H.begin() + c * k, s->H.begin() + (c + 1) * k, H + c * ldH);
    return SMK_OK;
}
// NnlsHals (nnls.hpp:249-316): HALS sweeps over H with W fixed until PG(H) < tol * PG(H after sweep 1)
int smk_solver_nnls_hals(smk_solver* s, double tol, int, int max_iter, int* iterations)
{
    if (!s || s->a->sparse) return SMK_UNSUPPORTED;
    const int64_t m = s->a->m, n = s->a->n;
    const int k = s->o.k;
    std::vector<double> G((size_t)k * k, 0.0), R((size_t)k * n, 0.0);
    for (int a = 0; a < k; ++a)
        for (int b = 0; b < k; ++b)
            for (int64_t i = 0; i < m; ++i) G[(size_t)b * k + a] += s->W[(size_t)a * m + i] * s->W[(size_t)b * m + i];
    for (int64_t j = 0; j < n; ++j)
        for (int a = 0; a < k; ++a)
            for (int64_t i = 0; i < m; ++i) R[(size_t)j * k + a] += s->W[(size_t)a * m + i] * s->a->dense[(size_t)j * m + i];
    double pg0 = 0.0;
    bool ok = false;
    int it = 0;
    for (it = 0; it < max_iter; ++it) {
        double sum = 0.0;
        for (int64_t j = 0; j < n; ++j) {
            double* h = &s->H[(size_t)j * k];
            for (int r = 0; r < k; ++r) {
                double dot = 0.0;
                for (int c = 0; c < k; ++c) dot += G[(size_t)c * k + r] * h[c];
                double v = h[r] + (R[(size_t)j * k + r] - dot) / G[(size_t)r * k + r];
                h[r] = (std::isnan(v) || v < 0.0) ? 0.0 : v;
            }
            for (int r = 0; r < k; ++r) {
                double g = -R[(size_t)j * k + r];
                for (int c = 0; c < k; ++c) g += G[(size_t)c * k + r] * h[c];
                if (g < 0.0 || h[r] > 0.0) sum += g * g;
            }
        }
        const double pg = std::sqrt(sum);
        if (it == 0) { pg0 = pg; continue; }
        if (pg < tol * pg0) { ok = true; orc_normalize_and_scale(m, n, k, s->W.data(), m, s->H.data(), k); break; }
    }
    if (iterations) *iterations = ok ? it + 1 : it;
    return ok ? SMK_OK : SMK_FAILURE;
}

int smk_nmf_dense(const smk_options* o, const double* A, int64_t ldA, double* W, int64_t ldW, double* H, int64_t ldH, smk_stats* st, int)
{
    if (!g_init) return SMK_NOTINITIALIZED;
    if (!o || !smk_is_valid(o, 1) || !A || !W || !H) return SMK_BAD_PARAM;
    smk_matrix* a = nullptr;
    smk_solver* s = nullptr;
    int rc = smk_matrix_create(&a, o->height, o->width, 0, o->width, 0);
    if (rc == SMK_OK) rc = smk_matrix_upload_f64(a, A, ldA);
    if (rc == SMK_OK) rc = smk_solver_create(&s, o, a);
    if (rc == SMK_OK) rc = smk_solver_set_factors(s, W, ldW, H, ldH);
    if (rc == SMK_OK) { rc = smk_solver_run(s, st); if (rc == SMK_OK || rc == SMK_FAILURE) smk_solver_get_factors(s, 0, W, ldW, H, ldH); }
    smk_solver_destroy(s);
    smk_matrix_destroy(a);
    return rc;
}
int smk_nmf_dense_sharded(const smk_options* o, const double* A, int64_t ldA, double* W, int64_t ldW, double* H, int64_t ldH,
                          smk_stats* st, int storage, int, const int*, int)
{
    return smk_nmf_dense(o, A, ldA, W, ldW, H, ldH, st, storage);
}
int smk_nmf_sparse(const smk_options* o, unsigned h, unsigned w, unsigned nz, const unsigned* cp, const unsigned* ri, const double* va,
                   double* W, int64_t ldW, double* H, int64_t ldH, smk_stats* st)
{
    if (!g_init) return SMK_NOTINITIALIZED;
    if (!o || !smk_is_valid(o, 1)) return SMK_BAD_PARAM;
    smk_matrix* a = nullptr;
    smk_solver* s = nullptr;
    int rc = smk_matrix_create_sparse(&a, h, w, 0, w, nz, cp, ri, va);
    if (rc == SMK_OK) rc = smk_solver_create(&s, o, a);
    if (rc == SMK_OK) rc = smk_solver_set_factors(s, W, ldW, H, ldH);
    if (rc == SMK_OK) { rc = smk_solver_run(s, st); if (rc == SMK_OK || rc == SMK_FAILURE) smk_solver_get_factors(s, 0, W, ldW, H, ldH); }
    smk_solver_destroy(s);
    smk_matrix_destroy(a);
    return rc;
}

}  // extern "C"
