// smallk_amd/csrc/sparse_subset.hip -- SparseMatrix::SubMatrixColsCompact on the device.
//
// A HierNMF2 node factors the columns `cols` of the resident sparse A with the rows that have no
// stored entry in those columns removed (common/include/sparse_matrix_impl.hpp:479-590).  The host
// version of that cut (counting sort for the transpose included) plus the upload cost more than a
// quarter of a C5-shaped run; here both CSC(A_sub) and CSC(A_sub') are assembled from the resident
// CSC(A) / CSC(A') with a handful of streaming kernels and three prefix sums, and only the row map
// (4 bytes per kept row) goes back to the host.
//
// Entry order inside a column of A_sub is the order in A; inside a column of A_sub' (= a row of
// A_sub) it is increasing new column index -- identical to what the host path produces, PROVIDED
// `cols` is strictly increasing (the caller checks; HierNMF2 document lists always are).
#include "common.h"

#include <hipcub/hipcub.hpp>

namespace smk {

namespace {

__global__ __launch_bounds__(256) void sub_len_kernel(const i64* __restrict__ colptr, const unsigned* __restrict__ cols,
                                                      i64 ncols, i64* __restrict__ len, int* __restrict__ colmap)
{
    const i64 j = (i64)blockIdx.x * 256 + threadIdx.x;
    if (j > ncols) return;
    if (j == ncols) { len[j] = 0; return; }
    const unsigned c = cols[j];
    len[j] = colptr[c + 1] - colptr[c];
    colmap[c] = (int)j;
}

// one wave per selected column: copy its entries, flag the rows it touches
__global__ __launch_bounds__(256) void sub_copy_kernel(const i64* __restrict__ colptr, const unsigned* __restrict__ rowidx,
                                                       const double* __restrict__ val, const unsigned* __restrict__ cols,
                                                       i64 ncols, const i64* __restrict__ cp, unsigned* __restrict__ ri,
                                                       double* __restrict__ va, int* __restrict__ used)
{
    const int lane = threadIdx.x & 63;
    for (i64 j = (i64)blockIdx.x * 4 + (threadIdx.x >> 6); j < ncols; j += (i64)gridDim.x * 4) {
        const i64 s0 = colptr[cols[j]], d0 = cp[j], cnt = cp[j + 1] - d0;
        for (i64 t = lane; t < cnt; t += 64) {
            const unsigned r = rowidx[s0 + t];
            ri[d0 + t] = r;
            va[d0 + t] = val[s0 + t];
            used[r] = 1;
        }
    }
}

__global__ __launch_bounds__(256) void sub_remap_kernel(unsigned* __restrict__ ri, i64 total, const int* __restrict__ o2n)
{
    for (i64 p = (i64)blockIdx.x * 256 + threadIdx.x; p < total; p += (i64)gridDim.x * 256) ri[p] = (unsigned)o2n[ri[p]];
}

__global__ __launch_bounds__(256) void sub_rowmap_kernel(const int* __restrict__ used, const int* __restrict__ o2n, i64 m,
                                                         unsigned* __restrict__ n2o)
{
    for (i64 r = (i64)blockIdx.x * 256 + threadIdx.x; r < m; r += (i64)gridDim.x * 256)
        if (used[r]) n2o[o2n[r]] = (unsigned)r;
}

// row i of A_sub = row n2o[i] of A restricted to the selected columns: count, then fill
__global__ __launch_bounds__(256) void sub_rowcount_kernel(const i64* __restrict__ colptr_t, const unsigned* __restrict__ rowidx_t,
                                                           const unsigned* __restrict__ n2o, i64 nh,
                                                           const int* __restrict__ colmap, i64* __restrict__ rlen)
{
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    if (i > nh) return;
    if (i == nh) { rlen[i] = 0; return; }
    const unsigned r = n2o[i];
    i64 cnt = 0;
    for (i64 p = colptr_t[r]; p < colptr_t[r + 1]; ++p) cnt += (colmap[rowidx_t[p]] >= 0);
    rlen[i] = cnt;
}

__global__ __launch_bounds__(256) void sub_rowfill_kernel(const i64* __restrict__ colptr_t, const unsigned* __restrict__ rowidx_t,
                                                          const double* __restrict__ val_t, const unsigned* __restrict__ n2o,
                                                          i64 nh, const int* __restrict__ colmap, const i64* __restrict__ cpt,
                                                          unsigned* __restrict__ rit, double* __restrict__ vat)
{
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    if (i >= nh) return;
    const unsigned r = n2o[i];
    i64 q = cpt[i];
    for (i64 p = colptr_t[r]; p < colptr_t[r + 1]; ++p) {
        const int c = colmap[rowidx_t[p]];
        if (c >= 0) {
            rit[q] = (unsigned)c;
            vat[q] = val_t[p];
            ++q;
        }
    }
}

template <typename T>
hipError_t exclusive_sum(const T* in, T* out, i64 count, void*& temp, size_t& temp_cap, hipStream_t st)
{
    size_t need = 0;
    hipError_t e = hipcub::DeviceScan::ExclusiveSum(nullptr, need, in, out, (int)count, st);
    if (e != hipSuccess) return e;
    if (need > temp_cap) {
        if (temp) (void)smk::dev_free(temp);
        temp = nullptr;
        temp_cap = 0;
        if ((e = smk::dev_malloc(&temp, need)) != hipSuccess) return e;
        temp_cap = need;
    }
    return hipcub::DeviceScan::ExclusiveSum(temp, need, in, out, (int)count, st);
}

}  // namespace

int device_sparse_subset(const SparseDev& src, const unsigned* cols_host, i64 ncols, SparseDev* out,
                         unsigned* new_to_old_host, hipStream_t st)
{
    *out = SparseDev();
    const i64 m = src.m;
    unsigned *d_cols = nullptr, *ri = nullptr, *rit = nullptr, *n2o = nullptr;
    int *colmap = nullptr, *used = nullptr, *o2n = nullptr;
    i64 *len = nullptr, *cp = nullptr, *rlen = nullptr, *cpt = nullptr;
    double *va = nullptr, *vat = nullptr;
    void* temp = nullptr;
    size_t temp_cap = 0;
    int rc = 0;
    hipError_t e = hipSuccess;
    i64 total = 0, total_t = 0;
    int nh = 0;
#define SUB_TRY(expr)                                                                             \
    do {                                                                                          \
        if (!rc && (e = (expr)) != hipSuccess) {                                                  \
            set_error(std::string("sparse subset: ") + #expr + ": " + hipGetErrorString(e));      \
            rc = -100;                                                                            \
        }                                                                                         \
    } while (0)
    SUB_TRY(smk::dev_malloc(&d_cols, (size_t)ncols * 4));
    SUB_TRY(smk::dev_malloc(&len, (size_t)(ncols + 1) * 8));
    SUB_TRY(smk::dev_malloc(&cp, (size_t)(ncols + 1) * 8));
    SUB_TRY(smk::dev_malloc(&colmap, (size_t)src.n * 4));
    SUB_TRY(smk::dev_malloc(&used, (size_t)(m + 1) * 4));
    SUB_TRY(smk::dev_malloc(&o2n, (size_t)(m + 1) * 4));
    SUB_TRY(hipMemcpyAsync(d_cols, cols_host, (size_t)ncols * 4, hipMemcpyHostToDevice, st));
    SUB_TRY(hipMemsetAsync(colmap, 0xFF, (size_t)src.n * 4, st));
    SUB_TRY(hipMemsetAsync(used, 0, (size_t)(m + 1) * 4, st));
    if (!rc) sub_len_kernel<<<(unsigned)((ncols + 256) / 256), 256, 0, st>>>(src.colptr, d_cols, ncols, len, colmap);
    SUB_TRY(exclusive_sum(len, cp, ncols + 1, temp, temp_cap, st));
    SUB_TRY(hipMemcpyAsync(&total, cp + ncols, 8, hipMemcpyDeviceToHost, st));
    SUB_TRY(hipStreamSynchronize(st));
    if (!rc && total == 0) {
        set_error("SparseMatrix::SubMatrixColsCompact: submatrix is the zero matrix");
        rc = -3;
    }
    SUB_TRY(smk::dev_malloc(&ri, (size_t)total * 4));
    SUB_TRY(smk::dev_malloc(&va, (size_t)total * 8));
    if (!rc) {
        const unsigned grid = (unsigned)std::min<i64>((ncols + 3) / 4, 65536);
        sub_copy_kernel<<<grid, 256, 0, st>>>(src.colptr, src.rowidx, src.val, d_cols, ncols, cp, ri, va, used);
    }
    SUB_TRY(exclusive_sum(used, o2n, m + 1, temp, temp_cap, st));
    SUB_TRY(hipMemcpyAsync(&nh, o2n + m, 4, hipMemcpyDeviceToHost, st));
    SUB_TRY(hipStreamSynchronize(st));
    SUB_TRY(smk::dev_malloc(&n2o, (size_t)(nh > 0 ? nh : 1) * 4));
    SUB_TRY(smk::dev_malloc(&rlen, (size_t)(nh + 1) * 8));
    SUB_TRY(smk::dev_malloc(&cpt, (size_t)(nh + 1) * 8));
    SUB_TRY(smk::dev_malloc(&rit, (size_t)total * 4));
    SUB_TRY(smk::dev_malloc(&vat, (size_t)total * 8));
    if (!rc) {
        const unsigned g1 = (unsigned)std::min<i64>((total + 255) / 256, 8192), g2 = (unsigned)std::min<i64>((m + 255) / 256, 8192);
        sub_remap_kernel<<<g1, 256, 0, st>>>(ri, total, o2n);
        sub_rowmap_kernel<<<g2, 256, 0, st>>>(used, o2n, m, n2o);
        sub_rowcount_kernel<<<(unsigned)((nh + 256) / 256), 256, 0, st>>>(src.colptr_t, src.rowidx_t, n2o, nh, colmap, rlen);
    }
    SUB_TRY(exclusive_sum(rlen, cpt, (i64)nh + 1, temp, temp_cap, st));
    if (!rc)
        sub_rowfill_kernel<<<(unsigned)((nh + 255) / 256), 256, 0, st>>>(src.colptr_t, src.rowidx_t, src.val_t, n2o, nh, colmap,
                                                                         cpt, rit, vat);
    SUB_TRY(hipGetLastError());
    SUB_TRY(hipMemcpyAsync(&total_t, cpt + nh, 8, hipMemcpyDeviceToHost, st));
    if (new_to_old_host) SUB_TRY(hipMemcpyAsync(new_to_old_host, n2o, (size_t)nh * 4, hipMemcpyDeviceToHost, st));
    SUB_TRY(hipStreamSynchronize(st));
    if (!rc && total_t != total) {
        set_error("sparse subset: transpose holds a different number of entries than the matrix");
        rc = -100;
    }
#undef SUB_TRY
    void* scratch[] = {d_cols, len, colmap, used, o2n, n2o, rlen, temp};
    for (void* p : scratch)
        if (p) (void)smk::dev_free(p);
    if (rc) {
        void* res[] = {cp, ri, va, cpt, rit, vat};
        for (void* p : res)
            if (p) (void)smk::dev_free(p);
        return rc;
    }
    out->m = nh; out->n = ncols; out->nnz = total;
    out->colptr = cp; out->rowidx = ri; out->val = va;
    out->colptr_t = cpt; out->rowidx_t = rit; out->val_t = vat;
    return 0;
}

}  // namespace smk
