// oracle/ref_sparse_glue.cpp -- TEST INFRASTRUCTURE ONLY.
// extern "C" handles onto the pieces of the reference that compile from their own sources with the
// standard library alone (no Elemental), built in place from /root/reference by `make -C oracle ref`
// into oracle/_ref/libref_sparse.so:
//   IsValid(NmfOptions, bool)                       common/src/nmf_options.cpp:23-112
//   SparseMatrix<T>::BeginLoad/Load/EndLoad/Compress common/include/sparse_matrix_impl.hpp:109-260
//   Transpose(SparseMatrix)                         common/include/sparse_matrix_ops.hpp:36-127
//   SparseMatrix<T>::SubMatrixColsCompact           common/include/sparse_matrix_impl.hpp:478-592
//   LoadMatrixMarketFile                            common/include/sparse_matrix_io.hpp:117-259
//   IsSparse / IsDense                              common/src/file_loader.cpp:20-40
//   Random (mt19937 wrappers)                       common/include/random.hpp:22-183
// The tests compare smk_is_valid, smk_load_matrix_market, the product's CSC transpose and its
// sparse column subsets (host and device cuts) with these, value for value.
#include <cstring>
#include <string>
#include <vector>
#include "nmf.hpp"
#include "sparse_matrix_decl.hpp"
#include "sparse_matrix_impl.hpp"
#include "sparse_matrix_ops.hpp"
#include "sparse_matrix_io.hpp"
#include "file_loader.hpp"
#include "random.hpp"

typedef SparseMatrix<double> SM;

extern "C" {

int ref_is_valid(double tol, int algorithm, int prog_est_algorithm, int height, int width, int k, int min_iter,
                 int max_iter, int tolcount, int max_threads, int verbose, int normalize, int validate_matrix)
{
    NmfOptions o;
    o.tol = tol;
    o.algorithm = (NmfAlgorithm)algorithm;
    o.prog_est_algorithm = (NmfProgressAlgorithm)prog_est_algorithm;
    o.height = height; o.width = width; o.k = k;
    o.min_iter = min_iter; o.max_iter = max_iter; o.tolcount = tolcount; o.max_threads = max_threads;
    o.verbose = verbose != 0; o.normalize = normalize != 0;
    return IsValid(o, validate_matrix != 0) ? 1 : 0;
}

// ---- opaque SparseMatrix<double> handles ----
void* ref_sm_from_triplets(unsigned height, unsigned width, unsigned count, const unsigned* rows, const unsigned* cols,
                           const double* vals)
{
    SM* a = new SM;
    a->Reserve(height, width, count);
    a->BeginLoad();
    for (unsigned i = 0; i < count; ++i) a->Load(rows[i], cols[i], vals[i]);
    a->EndLoad();
    return a;
}

void* ref_sm_from_csc(unsigned height, unsigned width, unsigned nz, const unsigned* col_offsets,
                      const unsigned* row_indices, const double* data)
{
    return new SM(height, width, nz, col_offsets, row_indices, data);
}

void ref_sm_free(void* h) { delete (SM*)h; }
unsigned ref_sm_height(void* h) { return ((SM*)h)->Height(); }
unsigned ref_sm_width(void* h) { return ((SM*)h)->Width(); }
unsigned ref_sm_size(void* h) { return ((SM*)h)->Size(); }

void ref_sm_copy_out(void* h, unsigned* col_offsets, unsigned* row_indices, double* data)
{
    SM* a = (SM*)h;
    const unsigned nz = a->Size();
    std::memcpy(col_offsets, a->LockedColBuffer(), sizeof(unsigned) * (a->Width() + 1));
    if (nz) {
        std::memcpy(row_indices, a->LockedRowBuffer(), sizeof(unsigned) * nz);
        std::memcpy(data, a->LockedDataBuffer(), sizeof(double) * nz);
    }
}

void* ref_sm_transpose(void* h)
{
    SM* b = new SM;
    Transpose(*(SM*)h, *b);
    return b;
}

// returns the submatrix handle; old_to_new has Height() entries, new_to_old at most Height()
void* ref_sm_submatrix_cols_compact(void* h, const unsigned* cols, unsigned ncols, unsigned* old_to_new,
                                    unsigned* new_to_old, unsigned* new_height)
{
    SM* a = (SM*)h;
    std::vector<unsigned> ci(cols, cols + ncols), o2n, n2o;
    SM* r = new SM;
    try {
        a->SubMatrixColsCompact(*r, ci, o2n, n2o);
    } catch (...) {
        delete r;
        return nullptr;
    }
    if (old_to_new) std::memcpy(old_to_new, o2n.data(), sizeof(unsigned) * o2n.size());
    if (new_to_old) std::memcpy(new_to_old, n2o.data(), sizeof(unsigned) * n2o.size());
    *new_height = (unsigned)n2o.size();
    return r;
}

void* ref_sm_load_matrix_market(const char* path, unsigned* height, unsigned* width, unsigned* nnz)
{
    SM* a = new SM;
    unsigned h = 0, w = 0, nz = 0;
    bool ok = false;
    try {
        ok = LoadMatrixMarketFile(std::string(path), *a, h, w, nz);
    } catch (...) {
        ok = false;
    }
    if (!ok) { delete a; return nullptr; }
    *height = h; *width = w; *nnz = nz;
    return a;
}

int ref_is_sparse_file(const char* path) { return IsSparse(std::string(path)) ? 1 : 0; }
int ref_is_dense_file(const char* path) { return IsDense(std::string(path)) ? 1 : 0; }

// Random::RandomDouble(center, radius) stream after SeedFromInt(seed)
void ref_random_doubles(int seed, double center, double radius, unsigned count, double* out)
{
    Random rng;
    rng.SeedFromInt(seed);
    for (unsigned i = 0; i < count; ++i) out[i] = rng.RandomDouble(center, radius);
}

}  // extern "C"
