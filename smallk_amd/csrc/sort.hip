// smallk_amd/csrc/sort.hip -- device argsort for the HierNMF2 priority score.
//
// compute_priority (clust_hier_util.hpp:105-173) ranks the m entries of three topic vectors
// (desc_ordered :37-47: decreasing value, ties by increasing index) and sorts one weight vector.
// For a 1M-term vocabulary those four host sorts cost more than the rank-2 factorisation they
// score, while the GPU sits idle; a stable LSD radix sort of (key, index) pairs gives exactly the
// reference's permutation: descending keys, equal keys keep their input (= index) order.
// -0.0 is canonicalised to +0.0 first because the radix order separates them and `>` does not.
#include "common.h"

#include <hipcub/hipcub.hpp>

namespace smk {

__global__ void prep_keys_kernel(const double* __restrict__ in, double* __restrict__ keys, int* __restrict__ idx, i64 n)
{
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
        keys[i] = in[i] + 0.0;
        if (idx) idx[i] = (int)i;
    }
}

// Host arrays in, host arrays out.  For each of the `count` key vectors (length n, vector v starts at
// keys_host[v]): idx_host[v] receives the stable descending argsort.  With idx_host[v] == nullptr
// the vector is sorted keys-only and written back in place to sorted_host[v].
int device_sort_desc(const double* const* keys_host, int* const* idx_host, double* const* sorted_host, int count, i64 n,
                     hipStream_t st)
{
    if (n <= 0 || count <= 0) return 0;
    double *d_in = nullptr, *d_keys = nullptr, *d_keys_out = nullptr;
    int *d_idx = nullptr, *d_idx_out = nullptr;
    void* d_temp = nullptr;
    size_t temp_pairs = 0, temp_keys = 0;
    hipError_t e = hipcub::DeviceRadixSort::SortPairsDescending(nullptr, temp_pairs, d_keys, d_keys_out, d_idx, d_idx_out,
                                                               (int)n, 0, 64, st);
    if (e == hipSuccess)
        e = hipcub::DeviceRadixSort::SortKeysDescending(nullptr, temp_keys, d_keys, d_keys_out, (int)n, 0, 64, st);
    const size_t temp_bytes = temp_pairs > temp_keys ? temp_pairs : temp_keys;
    int rc = 0;
    auto fail = [&](const char* what, hipError_t err) {
        set_error(std::string(what) + ": " + hipGetErrorString(err));
        rc = -100;
    };
    if (e != hipSuccess) fail("hipcub size query", e);
    if (!rc && (e = smk::dev_malloc(&d_in, (size_t)n * 8)) != hipSuccess) fail("hipMalloc", e);
    if (!rc && (e = smk::dev_malloc(&d_keys, (size_t)n * 8)) != hipSuccess) fail("hipMalloc", e);
    if (!rc && (e = smk::dev_malloc(&d_keys_out, (size_t)n * 8)) != hipSuccess) fail("hipMalloc", e);
    if (!rc && (e = smk::dev_malloc(&d_idx, (size_t)n * 4)) != hipSuccess) fail("hipMalloc", e);
    if (!rc && (e = smk::dev_malloc(&d_idx_out, (size_t)n * 4)) != hipSuccess) fail("hipMalloc", e);
    if (!rc && (e = smk::dev_malloc(&d_temp, temp_bytes ? temp_bytes : 16)) != hipSuccess) fail("hipMalloc", e);
    const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    for (int v = 0; v < count && !rc; ++v) {
        const bool pairs = idx_host[v] != nullptr;
        if ((e = hipMemcpyAsync(d_in, keys_host[v], (size_t)n * 8, hipMemcpyHostToDevice, st)) != hipSuccess) { fail("H2D", e); break; }
        prep_keys_kernel<<<grid, 256, 0, st>>>(d_in, d_keys, pairs ? d_idx : nullptr, n);
        size_t tb = temp_bytes;
        if (pairs)
            e = hipcub::DeviceRadixSort::SortPairsDescending(d_temp, tb, d_keys, d_keys_out, d_idx, d_idx_out, (int)n, 0, 64, st);
        else
            e = hipcub::DeviceRadixSort::SortKeysDescending(d_temp, tb, d_keys, d_keys_out, (int)n, 0, 64, st);
        if (e != hipSuccess) { fail("hipcub radix sort", e); break; }
        if (pairs) e = hipMemcpyAsync(idx_host[v], d_idx_out, (size_t)n * 4, hipMemcpyDeviceToHost, st);
        else e = hipMemcpyAsync(sorted_host[v], d_keys_out, (size_t)n * 8, hipMemcpyDeviceToHost, st);
        if (e != hipSuccess) { fail("D2H", e); break; }
        if ((e = hipStreamSynchronize(st)) != hipSuccess) { fail("sync", e); break; }
    }
    void* ptrs[] = {d_in, d_keys, d_keys_out, d_idx, d_idx_out, d_temp};
    for (void* p : ptrs)
        if (p) (void)smk::dev_free(p);
    return rc;
}


// ==========================================================================
// compute_priority (clust_hier_util.hpp:105-173) entirely on the device.  The host version spent 30 - 50 ms per call at
// 1 M terms in O(n) loops around the three device argsorts (logs, rank look-ups, two NDCG sums, the ideal sum) and in
// the transfers between them; here the topic vectors go up once (24 n bytes), everything else stays in HBM and three
// doubles come back.  Sums are two-level with a fixed order (per-workgroup partials, then one workgroup adds them in
// index order), so a score is reproducible run to run; against the sequential host sum it differs by rounding only.
// ==========================================================================
namespace {

struct PrioWs {
    i64 n = 0;
    double *in = nullptr, *keys = nullptr, *keys_out = nullptr, *weight = nullptr, *wpart = nullptr, *partials = nullptr, *result = nullptr;
    int *idx = nullptr, *idxp = nullptr, *idx1 = nullptr, *idx2 = nullptr, *pos1 = nullptr, *pos2 = nullptr, *seq = nullptr, *zfirst = nullptr;
    void* temp = nullptr;
    size_t temp_bytes = 0;
    double* host3 = nullptr;        // pinned
    void release()
    {
        void* ptrs[] = {in, keys, keys_out, weight, wpart, partials, result, idx, idxp, idx1, idx2, pos1, pos2, seq, zfirst, temp};
        for (void* p : ptrs) if (p) (void)smk::dev_free(p);
        if (host3) (void)hipHostFree(host3);
        *this = PrioWs();
    }
};
thread_local PrioWs g_prio;

constexpr int PRIO_BLOCKS = 1024;

__global__ __launch_bounds__(256) void prio_inverse_kernel(const int* __restrict__ ip, const int* __restrict__ i1, const int* __restrict__ i2,
                                                           int* __restrict__ seq, int* __restrict__ pos1, int* __restrict__ pos2, i64 n,
                                                           const double* __restrict__ wp, int* __restrict__ zfirst)
{
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) {
        const int t = ip[i];
        seq[t] = (int)i;
        pos1[i1[i]] = (int)i;
        pos2[i2[i]] = (int)i;
        if (wp[t] == 0.0) atomicMin(zfirst, (int)i);       // first position of the parent order that holds a zero
    }
}

__global__ __launch_bounds__(256) void prio_weights_kernel(const int* __restrict__ ip, const int* __restrict__ pos1,
                                                           const int* __restrict__ pos2, i64 n, i64 n_part, const int* __restrict__ zfirst,
                                                           double* __restrict__ weight, double* __restrict__ wpart)
{
    const i64 z = *zfirst;
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) {
        const int t = ip[i];
        const int mp = max(pos1[t], pos2[t]);
        double discount = log((double)(n - mp));
        if (discount == 0.0) discount = log(2.0);
        const double w = (i < z) ? log((double)(n - i)) : 1.0;
        const double wq = (i < n_part) ? log((double)(n_part - i)) : 0.0;
        weight[i] = w / discount;
        wpart[i] = wq / discount;
    }
}

// partial sums of term(i) over a CONTIGUOUS range per workgroup (so that the two-level sum runs in index order)
template <int MODE>   // 0: NDCG numerator of child order `test` (wpart[seq[test[i]]] / log2(i + 1)), 1: ideal sum of sorted weights
__global__ __launch_bounds__(256) void prio_sum_kernel(const int* __restrict__ seq, const int* __restrict__ test,
                                                       const double* __restrict__ v, i64 n, double* __restrict__ partials)
{
    __shared__ double sh[4];
    const i64 per = (n + gridDim.x - 1) / gridDim.x;
    const i64 lo = (i64)blockIdx.x * per, hi = (lo + per < n) ? lo + per : n;
    double acc = 0.0;
    for (i64 i = lo + threadIdx.x; i < hi; i += 256) {
        double s = (MODE == 0) ? v[seq[test[i]]] : v[i];
        if (i > 0) s /= log2((double)(i + 1));
        acc += s;
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__global__ __launch_bounds__(256) void prio_final_kernel(const double* __restrict__ partials, int nb, double* __restrict__ out3)
{
    // three lists of nb partials; thread 0 of each wave adds one list in index order (sequential: fixed, and nb is small)
    const int w = threadIdx.x >> 6;
    if (w < 3 && (threadIdx.x & 63) == 0) {
        double t = 0.0;
        for (int i = 0; i < nb; ++i) t += partials[(i64)w * nb + i];
        out3[w] = t;
    }
}

}  // namespace

void device_priority_release() { g_prio.release(); }

// *score = compute_priority(wp, wc[0..n), wc[n..2n)); n_part = number of nonzero entries of wp (counted by the caller).
// Returns 0, or a negative code when the device path is not available (the caller then takes the host path).
int device_priority_score(const double* wp, const double* wc, i64 n, i64 n_part, double* score, hipStream_t st)
{
    if (n <= 0 || n > 0x7FFFFFFF) return -1;
    PrioWs& w = g_prio;
    hipError_t e = hipSuccess;
    if (w.n < n) {
        w.release();
        size_t tp = 0, tk = 0;
        e = hipcub::DeviceRadixSort::SortPairsDescending(nullptr, tp, w.keys, w.keys_out, w.idx, w.idxp, (int)n, 0, 64, st);
        if (e == hipSuccess) e = hipcub::DeviceRadixSort::SortKeysDescending(nullptr, tk, w.keys, w.keys_out, (int)n, 0, 64, st);
        if (e != hipSuccess) return -2;
        w.temp_bytes = (tp > tk ? tp : tk) + 16;
        bool ok = true;
        auto A = [&](void** p, size_t bytes) { if (ok && smk::dev_malloc(p, bytes) != hipSuccess) ok = false; };
        A((void**)&w.in, (size_t)n * 24); A((void**)&w.keys, (size_t)n * 8); A((void**)&w.keys_out, (size_t)n * 8);
        A((void**)&w.weight, (size_t)n * 8); A((void**)&w.wpart, (size_t)n * 8); A((void**)&w.partials, (size_t)PRIO_BLOCKS * 3 * 8);
        A((void**)&w.result, 64);
        int** ips[] = {&w.idx, &w.idxp, &w.idx1, &w.idx2, &w.pos1, &w.pos2, &w.seq};
        for (int** ip : ips) A((void**)ip, (size_t)n * 4);
        A((void**)&w.zfirst, 64); A(&w.temp, w.temp_bytes);
        if (ok && hipHostMalloc((void**)&w.host3, 64) != hipSuccess) ok = false;
        if (!ok) { w.release(); set_error("device priority score: out of memory"); return -3; }
        w.n = n;
    }
    const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    const int nb = (int)((n + 255) / 256 < PRIO_BLOCKS ? (n + 255) / 256 : PRIO_BLOCKS);
    auto H = [&](hipError_t x) { if (e == hipSuccess) e = x; };
    H(hipMemcpyAsync(w.in, wp, (size_t)n * 8, hipMemcpyHostToDevice, st));
    H(hipMemcpyAsync(w.in + n, wc, (size_t)n * 16, hipMemcpyHostToDevice, st));
    int* outs[3] = {w.idxp, w.idx1, w.idx2};
    for (int v = 0; v < 3 && e == hipSuccess; ++v) {
        prep_keys_kernel<<<grid, 256, 0, st>>>(w.in + (i64)v * n, w.keys, w.idx, n);
        size_t tb = w.temp_bytes;
        H(hipcub::DeviceRadixSort::SortPairsDescending(w.temp, tb, w.keys, w.keys_out, w.idx, outs[v], (int)n, 0, 64, st));
    }
    const int big = 0x7FFFFFFF;
    H(hipMemcpyAsync(w.zfirst, &big, sizeof(int), hipMemcpyHostToDevice, st));
    if (e == hipSuccess) {
        prio_inverse_kernel<<<grid, 256, 0, st>>>(w.idxp, w.idx1, w.idx2, w.seq, w.pos1, w.pos2, n, w.in, w.zfirst);
        prio_weights_kernel<<<grid, 256, 0, st>>>(w.idxp, w.pos1, w.pos2, n, n_part, w.zfirst, w.weight, w.wpart);
        prio_sum_kernel<0><<<nb, 256, 0, st>>>(w.seq, w.idx1, w.wpart, n, w.partials);
        prio_sum_kernel<0><<<nb, 256, 0, st>>>(w.seq, w.idx2, w.wpart, n, w.partials + nb);
        size_t tb = w.temp_bytes;
        H(hipcub::DeviceRadixSort::SortKeysDescending(w.temp, tb, w.weight, w.keys_out, (int)n, 0, 64, st));
        prio_sum_kernel<1><<<nb, 256, 0, st>>>(nullptr, nullptr, w.keys_out, n, w.partials + 2 * (i64)nb);
        prio_final_kernel<<<1, 256, 0, st>>>(w.partials, nb, w.result);
        H(hipGetLastError());
    }
    H(hipMemcpyAsync(w.host3, w.result, 24, hipMemcpyDeviceToHost, st));
    H(hipStreamSynchronize(st));
    if (e != hipSuccess) { set_error(std::string("device priority score: ") + hipGetErrorString(e)); return -4; }
    const double c1 = w.host3[0], c2 = w.host3[1], ideal = w.host3[2];
    *score = (c1 / ideal) * (c2 / ideal);
    return 0;
}


// ==========================================================================
// CSC(A') from CSC(A) on the device (Transpose(SparseMatrix), sparse_matrix_ops.hpp:36-127, is a counting sort by row
// that keeps the column order inside a row): a STABLE radix sort of the row indices carrying the entry position, then two
// gathers.  Same entry order as the host counting sort, so every product sums in the same order; 16 M entries take a few
// milliseconds where the host sort took 0.1 - 0.15 s of a C5-sized upload.
// ==========================================================================
namespace {
__global__ __launch_bounds__(256) void tr_colids_kernel(const i64* __restrict__ colptr, i64 ncols, unsigned* __restrict__ col_of,
                                                        unsigned* __restrict__ pos)
{
    // one wave per column: its entries get the column id; pos[p] = p
    const i64 wave = ((i64)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const i64 nw = ((i64)gridDim.x * 256) >> 6;
    for (i64 j = wave; j < ncols; j += nw)
        for (i64 p = colptr[j] + lane; p < colptr[j + 1]; p += 64) { col_of[p] = (unsigned)j; pos[p] = (unsigned)p; }
}
__global__ __launch_bounds__(256) void tr_gather_kernel(const unsigned* __restrict__ perm, const unsigned* __restrict__ col_of,
                                                        const double* __restrict__ val, i64 nnz, unsigned* __restrict__ rowidx_t,
                                                        double* __restrict__ val_t)
{
    for (i64 q = (i64)blockIdx.x * 256 + threadIdx.x; q < nnz; q += (i64)gridDim.x * 256) {
        const unsigned p = perm[q];
        rowidx_t[q] = col_of[p];
        val_t[q] = val[p];
    }
}
// colptr_t[r] = first position of the sorted row list holding a row >= r
__global__ __launch_bounds__(256) void tr_offsets_kernel(const unsigned* __restrict__ rows_sorted, i64 nnz, i64 height,
                                                         i64* __restrict__ colptr_t)
{
    for (i64 r = (i64)blockIdx.x * 256 + threadIdx.x; r <= height; r += (i64)gridDim.x * 256) {
        i64 lo = 0, hi = nnz;
        while (lo < hi) {
            const i64 mid = (lo + hi) >> 1;
            if ((i64)rows_sorted[mid] < r) lo = mid + 1; else hi = mid;
        }
        colptr_t[r] = lo;
    }
}
}  // namespace

int device_csc_transpose(i64 height, i64 ncols, i64 nnz, const i64* colptr, const unsigned* rowidx, const double* val,
                         i64* colptr_t, unsigned* rowidx_t, double* val_t, hipStream_t st)
{
    if (nnz <= 0) { return hipMemsetAsync(colptr_t, 0, (size_t)(height + 1) * sizeof(i64), st) == hipSuccess ? 0 : -100; }
    if (nnz > 0x7FFFFFFF) return -1;                    // positions travel as 32-bit payloads
    unsigned *col_of = nullptr, *pos = nullptr, *rows_sorted = nullptr, *perm = nullptr;
    void* temp = nullptr;
    size_t tb = 0;
    int bits = 1;
    while (((i64)1 << bits) < height && bits < 32) ++bits;
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(nullptr, tb, rowidx, rows_sorted, pos, perm, (int)nnz, 0, bits, st);
    int rc = 0;
    auto fail = [&](const char* what) { set_error(std::string("device CSC transpose: ") + what); rc = -100; };
    if (e != hipSuccess) fail("size query");
    if (!rc && smk::dev_malloc((void**)&col_of, (size_t)nnz * 4) != hipSuccess) fail("hipMalloc");
    if (!rc && smk::dev_malloc((void**)&pos, (size_t)nnz * 4) != hipSuccess) fail("hipMalloc");
    if (!rc && smk::dev_malloc((void**)&rows_sorted, (size_t)nnz * 4) != hipSuccess) fail("hipMalloc");
    if (!rc && smk::dev_malloc((void**)&perm, (size_t)nnz * 4) != hipSuccess) fail("hipMalloc");
    if (!rc && smk::dev_malloc(&temp, tb + 16) != hipSuccess) fail("hipMalloc");
    if (!rc) {
        const int g1 = (int)((ncols * 64 + 255) / 256 < 8192 ? (ncols * 64 + 255) / 256 : 8192);
        tr_colids_kernel<<<g1 > 0 ? g1 : 1, 256, 0, st>>>(colptr, ncols, col_of, pos);
        e = hipcub::DeviceRadixSort::SortPairs(temp, tb, rowidx, rows_sorted, pos, perm, (int)nnz, 0, bits, st);     // stable
        if (e != hipSuccess) fail("radix sort");
    }
    if (!rc) {
        const int g2 = (int)((nnz + 255) / 256 < 8192 ? (nnz + 255) / 256 : 8192);
        tr_gather_kernel<<<g2, 256, 0, st>>>(perm, col_of, val, nnz, rowidx_t, val_t);
        const int g3 = (int)((height + 256) / 256 < 4096 ? (height + 256) / 256 : 4096);
        tr_offsets_kernel<<<g3, 256, 0, st>>>(rows_sorted, nnz, height, colptr_t);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) fail("kernels");
    }
    void* ptrs[] = {col_of, pos, rows_sorted, perm, temp};
    for (void* p : ptrs) if (p) (void)smk::dev_free(p);
    return rc;
}

}  // namespace smk
