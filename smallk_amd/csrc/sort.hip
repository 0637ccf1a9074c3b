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
    if (!rc && (e = hipMalloc(&d_in, (size_t)n * 8)) != hipSuccess) fail("hipMalloc", e);
    if (!rc && (e = hipMalloc(&d_keys, (size_t)n * 8)) != hipSuccess) fail("hipMalloc", e);
    if (!rc && (e = hipMalloc(&d_keys_out, (size_t)n * 8)) != hipSuccess) fail("hipMalloc", e);
    if (!rc && (e = hipMalloc(&d_idx, (size_t)n * 4)) != hipSuccess) fail("hipMalloc", e);
    if (!rc && (e = hipMalloc(&d_idx_out, (size_t)n * 4)) != hipSuccess) fail("hipMalloc", e);
    if (!rc && (e = hipMalloc(&d_temp, temp_bytes ? temp_bytes : 16)) != hipSuccess) fail("hipMalloc", e);
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
        if (p) (void)hipFree(p);
    return rc;
}

}  // namespace smk
