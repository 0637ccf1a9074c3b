// smallk_amd/csrc/kernels.hip -- hand-written gfx950 (MI355X / CDNA4) kernels for the
// dense NMF inner loop.  Wave = 64 lanes everywhere.  No CUDA compatibility paths.
//
// Kernel map (reference call sites in parentheses, paths relative to /root/reference); the streaming
// products live in bigprod.hip, the NNLS kernels in nnls.hip, shared device helpers in devutil.h:
//   bigprod_kernel      W'A and H*At streaming products  (nmf_solver_{mu,hals,bpp}.hpp Gemm calls
//                       on A: mu :131,:143  hals :173,:187  bpp :354,:367) -- bf16/f32 MFMA, LDS
//                       staged by global_load_lds, 2- to 5-deep ring, counted vmcnt.
//   pack_kernel         fp64 factor -> MFMA A-operand fragments (bf16 hi/mid/lo split, or f32)
//   gram_*              W'W, HH'                         (Gemm TRANSPOSE,NORMAL / NORMAL,TRANSPOSE)
//   mu_update_kernel    Update_H_MU / Update_W_MU        (nmf_solver_mu.hpp:27-71)
//   hals_sweep_kernel   UpdateH_Hals                     (nmf_solver_hals.hpp:26-62)
//   hals_w_col_kernel   UpdateW_Hals                     (nmf_solver_hals.hpp:66-117)
//   nnls_bpp_kernel     NnlsBlockpivot + UpdatePassiveSet + BppUpdateSets + masked SPD solves
//                       (nnls.hpp:144-244, src/nnls.cpp:18-74, nnls.hpp:43-140,
//                        nmf_solver_bpp.hpp:146-219, normal_eq.hpp:27-54)
//   grad_pg_kernel      gradients + ProjectedGradientNorm (projected_gradient.hpp:125-171)
//   scale_rows_kernel   NormalizeAndScale                (normalize.hpp:25-53,90-140)
//   delta_fnorm_kernel  ProgEstGenericDeltaW::Compute    (progress_estimator_generic.hpp:58-69)
#include "common.h"
#include "devutil.h"
#include "gram_inverse.h"
#include "gram_body.h"
#include <climits>

namespace smk {

template <typename T> __device__ inline T store_cast(float f);
template <> __device__ inline float store_cast<float>(float f) { return f; }
template <> __device__ inline unsigned short store_cast<unsigned short>(float f) { return f32_to_bf16_rne(f); }

template <typename T>
__global__ __launch_bounds__(256) void fill_uniform_kernel(T* __restrict__ buf, i64 ld, i64 rows, i64 cols,
                                                           i64 rows_pad, i64 cols_pad, i64 r0, i64 c0,
                                                           i64 gheight, uint64_t seed, int quant)
{
    const i64 total = rows_pad * cols_pad;
    for (i64 idx = (i64)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (i64)gridDim.x * blockDim.x) {
        i64 c = idx / rows_pad, r = idx - c * rows_pad;
        float v = 0.f;
        if (r < rows && c < cols) v = uniform_value(seed, (uint64_t)((c0 + c) * gheight + (r0 + r)), quant);
        buf[c * ld + r] = store_cast<T>(v);
    }
}

int launch_fill_uniform(void* buf, int storage, i64 ld, i64 rows, i64 cols, i64 rows_pad, i64 cols_pad,
                        i64 r0, i64 c0, i64 gheight, uint64_t seed, int quant, hipStream_t st)
{
    i64 total = rows_pad * cols_pad;
    int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (grid < 1) grid = 1;
    if (storage == STORE_BF16)
        fill_uniform_kernel<unsigned short><<<grid, 256, 0, st>>>((unsigned short*)buf, ld, rows, cols, rows_pad,
                                                                  cols_pad, r0, c0, gheight, seed, quant);
    else
        fill_uniform_kernel<float><<<grid, 256, 0, st>>>((float*)buf, ld, rows, cols, rows_pad, cols_pad, r0, c0,
                                                         gheight, seed, quant);
    SMK_HIP(hipGetLastError());
    return 0;
}

// Structured synthetic data ("planted-low-rank variant for convergence sanity"): element (r, c) of
//   A = Ws Hs + noise * U,   Ws = (m x kstar uniform, seed + 1, entries <= thr dropped), Hs = (kstar x n_global uniform,
//   seed + 2, entries <= thr dropped), U = the uniform matrix of `seed`,
// every factor keyed by its GLOBAL element index, the sum taken in fp64 with one fused multiply-add per j in increasing
// j and the noise term added last with one more -- oracle/nmf_oracle.c:orc_fill_planted does the same operations in the same
// order, so the stored fp32 / bf16 values are the oracle's bit for bit and shards agree with the whole.
// A 64 x 64 tile per workgroup, 64 planted components at a time through LDS (the factors are generated in place: no
// buffers), a row and 16 columns per thread.
template <typename T>
__global__ __launch_bounds__(256) void fill_planted_kernel(T* __restrict__ buf, i64 ld, i64 rows, i64 cols, i64 rows_pad,
                                                           i64 cols_pad, i64 c0, i64 gheight, uint64_t seed, int kstar,
                                                           float thr, double noise, int quant)
{
    __shared__ float ws[64][65];     // [j][row]
    __shared__ float hs[64][64];     // [j][col]
    const i64 tr0 = (i64)blockIdx.x * 64, tc0 = (i64)blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    double acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0;
    for (int j0 = 0; j0 < kstar; j0 += 64) {
        __syncthreads();
        for (int jj = ty; jj < 64; jj += 4) {
            const int j = j0 + jj;
            float w = 0.f, h = 0.f;
            if (j < kstar) {
                if (tr0 + tx < rows) w = uniform_value(seed + 1, (uint64_t)((i64)j * gheight + tr0 + tx), 0);
                if (tc0 + tx < cols) h = uniform_value(seed + 2, (uint64_t)((c0 + tc0 + tx) * (i64)kstar + j), 0);
            }
            ws[jj][tx] = w > thr ? w : 0.f;
            hs[jj][tx] = h > thr ? h : 0.f;
        }
        __syncthreads();
        const int jn = kstar - j0 < 64 ? kstar - j0 : 64;
        for (int jj = 0; jj < jn; ++jj) {
            const double w = (double)ws[jj][tx];
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_fma(w, (double)hs[jj][ty * 16 + i], acc[i]);
        }
    }
    const i64 r = tr0 + tx;
    if (r >= rows_pad) return;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const i64 c = tc0 + ty * 16 + i;
        if (c >= cols_pad) continue;
        float v = 0.f;
        if (r < rows && c < cols) {
            const float u = uniform_value(seed, (uint64_t)((c0 + c) * gheight + r), 0);
            v = (float)__builtin_fma(noise, (double)u, acc[i]);
            if (quant == 1) v = bf16_bits_to_f32(f32_to_bf16_rne(v));
        }
        buf[c * ld + r] = store_cast<T>(v);
    }
}

int launch_fill_planted(void* buf, int storage, i64 ld, i64 rows, i64 cols, i64 rows_pad, i64 cols_pad, i64 c0,
                        i64 gheight, uint64_t seed, int kstar, double thr, double noise, int quant, hipStream_t st)
{
    dim3 grid((unsigned)((rows_pad + 63) / 64), (unsigned)((cols_pad + 63) / 64));
    if (grid.y > 65535u) return -101; /* SMK_UNSUPPORTED: more than 4 M columns */
    if (storage == STORE_BF16)
        fill_planted_kernel<unsigned short><<<grid, 256, 0, st>>>((unsigned short*)buf, ld, rows, cols, rows_pad, cols_pad,
                                                                  c0, gheight, seed, kstar, (float)thr, noise, quant);
    else
        fill_planted_kernel<float><<<grid, 256, 0, st>>>((float*)buf, ld, rows, cols, rows_pad, cols_pad, c0, gheight,
                                                         seed, kstar, (float)thr, noise, quant);
    SMK_HIP(hipGetLastError());
    return 0;
}

// fp64 (host layout, staged on device) -> storage dtype
template <typename T>
__global__ __launch_bounds__(256) void convert_f64_kernel(const double* __restrict__ src, i64 ld_src,
                                                          T* __restrict__ dst, i64 ld_dst, i64 rows, i64 cols)
{
    const i64 total = rows * cols;
    for (i64 idx = (i64)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (i64)gridDim.x * blockDim.x) {
        i64 c = idx / rows, r = idx - c * rows;
        dst[c * ld_dst + r] = store_cast<T>((float)src[c * ld_src + r]);
    }
}

int launch_convert_f64(const double* src, i64 ld_src, void* dst, int storage, i64 ld_dst, i64 rows, i64 cols,
                       hipStream_t st)
{
    i64 total = rows * cols;
    if (total == 0) return 0;
    int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (storage == STORE_BF16)
        convert_f64_kernel<unsigned short><<<grid, 256, 0, st>>>(src, ld_src, (unsigned short*)dst, ld_dst, rows, cols);
    else
        convert_f64_kernel<float><<<grid, 256, 0, st>>>(src, ld_src, (float*)dst, ld_dst, rows, cols);
    SMK_HIP(hipGetLastError());
    return 0;
}

// dst(cols x rows) = src(rows x cols)'   64x64 tiles through LDS
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T* __restrict__ src, i64 ld_src, T* __restrict__ dst,
                                                        i64 ld_dst, i64 rows, i64 cols)
{
    __shared__ T tile[64][65];
    const i64 r0 = (i64)blockIdx.x * 64, c0 = (i64)blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int cc = ty; cc < 64; cc += 4) {
        i64 r = r0 + tx, c = c0 + cc;
        tile[cc][tx] = (r < rows && c < cols) ? src[c * ld_src + r] : T(0);
    }
    __syncthreads();
    for (int rr = ty; rr < 64; rr += 4) {
        i64 r = r0 + rr, c = c0 + tx;
        if (r < rows && c < cols) dst[r * ld_dst + c] = tile[tx][rr];
    }
}

int launch_transpose_store(const void* src, i64 ld_src, void* dst, i64 ld_dst, int storage, i64 rows, i64 cols,
                           hipStream_t st)
{
    dim3 grid((unsigned)((rows + 63) / 64), (unsigned)((cols + 63) / 64));
    if (storage == STORE_BF16)
        transpose_kernel<unsigned short><<<grid, 256, 0, st>>>((const unsigned short*)src, ld_src,
                                                               (unsigned short*)dst, ld_dst, rows, cols);
    else
        transpose_kernel<float><<<grid, 256, 0, st>>>((const float*)src, ld_src, (float*)dst, ld_dst, rows, cols);
    SMK_HIP(hipGetLastError());
    return 0;
}

int launch_transpose_f64(const double* src, i64 ld_src, double* dst, i64 ld_dst, i64 rows, i64 cols, hipStream_t st)
{
    dim3 grid((unsigned)((rows + 63) / 64), (unsigned)((cols + 63) / 64));
    transpose_kernel<double><<<grid, 256, 0, st>>>(src, ld_src, dst, ld_dst, rows, cols);
    SMK_HIP(hipGetLastError());
    return 0;
}

// dst[:, j] = src[:, cols[j]] for whole padded columns (col_bytes is a multiple of 16):
// the HierNMF2 node submatrix (SubMatrixColsCompact, dense_matrix_impl.hpp:224-281) without
// leaving HBM.  One uint4 per thread, fully coalesced on both sides.
__global__ __launch_bounds__(256) void gather_cols_kernel(const uint4* __restrict__ src, i64 ld_src16,
                                                          const unsigned* __restrict__ cols, uint4* __restrict__ dst,
                                                          i64 ld_dst16, i64 col16)
{
    const i64 j = blockIdx.y;
    const uint4* s = src + (i64)cols[j] * ld_src16;
    uint4* d = dst + j * ld_dst16;
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < col16; i += (i64)gridDim.x * 256)
        d[i] = s[i];
}
int launch_gather_cols(const void* src, i64 ld_src_bytes, const unsigned* cols_dev, i64 ncols, void* dst,
                       i64 ld_dst_bytes, i64 col_bytes, hipStream_t st)
{
    if (ncols <= 0) return 0;
    const i64 col16 = col_bytes / 16;
    unsigned gx = (unsigned)((col16 + 255) / 256);
    if (gx > 64) gx = 64;
    for (i64 j0 = 0; j0 < ncols; j0 += 65535) {      // gridDim.y limit
        const i64 nj = ncols - j0 < 65535 ? ncols - j0 : 65535;
        gather_cols_kernel<<<dim3(gx, (unsigned)nj), 256, 0, st>>>((const uint4*)src, ld_src_bytes / 16, cols_dev + j0,
                                                                   (uint4*)((char*)dst + j0 * ld_dst_bytes),
                                                                   ld_dst_bytes / 16, col16);
    }
    SMK_HIP(hipGetLastError());
    return 0;
}

// Snapshot / restore of the factors around a speculative iteration: the k2 = round_up(k, 2) live rows
// of W' (KP x m) and H (KP x n) -- pad rows are zero and stay zero -- plus the KP x KP Gram matrix, in
// one launch.  `pack` != 0: factors -> compact buffer, else the reverse.
__global__ __launch_bounds__(256) void snapshot_kernel(double* __restrict__ Wt, i64 m, double* __restrict__ H, i64 n,
                                                       double* __restrict__ G, double* __restrict__ buf, int KP, int k2,
                                                       int pack)
{
    const int h2 = k2 / 2;                                 // 16-byte pairs per column
    const i64 nw = m * h2, nh = n * h2, ng = (i64)KP * KP / 2;
    f64x2_t* b2 = (f64x2_t*)buf;
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < nw + nh + ng; i += (i64)gridDim.x * 256) {
        f64x2_t* p;
        if (i < nw) p = (f64x2_t*)(Wt + (i / h2) * KP) + (i % h2);
        else if (i < nw + nh) p = (f64x2_t*)(H + ((i - nw) / h2) * KP) + ((i - nw) % h2);
        else p = (f64x2_t*)G + (i - nw - nh);
        if (pack) b2[i] = *p;
        else *p = b2[i];
    }
}
size_t snapshot_elems(int k, i64 m, i64 n)
{
    const int KP = kp_of(k), k2 = (k + 1) / 2 * 2;
    return (size_t)((m + n) * k2 + (i64)KP * KP);
}
int launch_snapshot(double* Wt, i64 m, double* H, i64 n, double* G, double* buf, int k, int pack, hipStream_t st)
{
    const int KP = kp_of(k), k2 = (k + 1) / 2 * 2;
    const i64 total = (m + n) * (k2 / 2) + (i64)KP * KP / 2;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    snapshot_kernel<<<grid, 256, 0, st>>>(Wt, m, H, n, G, buf, KP, k2, pack);
    SMK_HIP(hipGetLastError());
    return 0;
}

// dst (k x N, ld k) = first k rows of src (KP x N, ld KP): the host-facing layout of a factor
__global__ __launch_bounds__(256) void compact_rows_kernel(const double* __restrict__ src, int KP, double* __restrict__ dst,
                                                           int k, i64 N)
{
    const i64 total = N * k;
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < total; i += (i64)gridDim.x * 256)
        dst[i] = src[(i / k) * KP + (i % k)];
}
int launch_compact_rows(const double* src, int KP, double* dst, int k, i64 N, hipStream_t st)
{
    const i64 total = N * k;
    if (total <= 0) return 0;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    compact_rows_kernel<<<grid, 256, 0, st>>>(src, KP, dst, k, N);
    SMK_HIP(hipGetLastError());
    return 0;
}

__global__ void zero_f64_kernel(double* p, i64 n)
{
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) p[i] = 0.0;
}
int launch_zero_f64(double* p, i64 n, hipStream_t st)
{
    if (n <= 0) return 0;
    int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    zero_f64_kernel<<<grid, 256, 0, st>>>(p, n);
    SMK_HIP(hipGetLastError());
    return 0;
}

// Sharded runs: the H-side projected-gradient sum and a "some rank failed" indicator travel in one 2-element
// all-reduce (scal[6..7]); afterwards every rank holds the global sum in scal[1] and a consistent failure flag.
// wpart: the W-side projected-gradient sum is a partial one too (BPP with row-sharded W solves) and travels in scal[5]
__global__ void dist_scalars_kernel(double* __restrict__ scal, int* __restrict__ flag, int tag, int unpack, int wpart)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (!unpack) {
        scal[5] = wpart ? scal[0] : 0.0;
        scal[6] = scal[1];
        scal[7] = (*flag != 0x7FFFFFFF) ? 1.0 : 0.0;
    } else {
        if (wpart) scal[0] = scal[5];
        scal[1] = scal[6];
        if (scal[7] > 0.0) atomicMin(flag, tag);
    }
}
int launch_dist_scalars(double* scal, int* flag, int tag, int unpack, int wpart, hipStream_t st)
{
    dist_scalars_kernel<<<1, 64, 0, st>>>(scal, flag, tag, unpack, wpart);
    SMK_HIP(hipGetLastError());
    return 0;
}

// ---- in-process stand-in for the RCCL collectives (comm.cpp: several shards on one device) ----
struct RankPtrs { void* p[16]; };
template <typename T>
__global__ __launch_bounds__(256) void local_allreduce_kernel(RankPtrs rp, int world, i64 count)
{
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < count; i += (i64)gridDim.x * 256) {
        T s = ((const T*)rp.p[0])[i];
        for (int r = 1; r < world; ++r) s += ((const T*)rp.p[r])[i];        // fixed rank order
        for (int r = 0; r < world; ++r) ((T*)rp.p[r])[i] = s;
    }
}
// every rank's `per` elements at send[r] land at recv[q] + r * per on every rank q
template <typename T>
__global__ __launch_bounds__(256) void local_allgather_kernel(RankPtrs send, RankPtrs recv, int world, i64 per)
{
    const i64 total = per * world;
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < total; i += (i64)gridDim.x * 256) {
        const int src = (int)(i / per);
        const T v = ((const T*)send.p[src])[i - (i64)src * per];
        for (int r = 0; r < world; ++r) ((T*)recv.p[r])[i] = v;
    }
}
// recv[r] (per elements) <- sum over ranks q of send[q][r * per ...], fixed rank order
template <typename T>
__global__ __launch_bounds__(256) void local_reduce_scatter_kernel(RankPtrs send, RankPtrs recv, int world, i64 per)
{
    const i64 total = per * world;
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < total; i += (i64)gridDim.x * 256) {
        const int dst = (int)(i / per);
        T s = ((const T*)send.p[0])[i];
        for (int r = 1; r < world; ++r) s += ((const T*)send.p[r])[i];
        ((T*)recv.p[dst])[i - (i64)dst * per] = s;
    }
}
int launch_local_reduce_scatter(void* const* sends, void* const* recvs, int world, i64 per, int f64, hipStream_t st)
{
    if (world > 16) { set_error("local communicator: at most 16 ranks"); return -100; }
    RankPtrs rp, rq;
    for (int r = 0; r < 16; ++r) { rp.p[r] = r < world ? sends[r] : nullptr; rq.p[r] = r < world ? recvs[r] : nullptr; }
    i64 grid = (per * world + 255) / 256;
    if (grid > 4096) grid = 4096;
    if (grid < 1) grid = 1;
    if (f64) local_reduce_scatter_kernel<double><<<(unsigned)grid, 256, 0, st>>>(rp, rq, world, per);
    else local_reduce_scatter_kernel<float><<<(unsigned)grid, 256, 0, st>>>(rp, rq, world, per);
    SMK_HIP(hipGetLastError());
    return 0;
}
int launch_local_allreduce(void* const* ptrs, int world, i64 count, int f64, hipStream_t st)
{
    if (world > 16) { set_error("local communicator: at most 16 ranks"); return -100; }
    RankPtrs rp;
    for (int r = 0; r < 16; ++r) rp.p[r] = r < world ? ptrs[r] : nullptr;
    const int grid = (int)((count + 255) / 256 < 2048 ? (count + 255) / 256 : 2048);
    if (grid < 1) return 0;
    if (f64) local_allreduce_kernel<double><<<grid, 256, 0, st>>>(rp, world, count);
    else local_allreduce_kernel<float><<<grid, 256, 0, st>>>(rp, world, count);
    SMK_HIP(hipGetLastError());
    return 0;
}
int launch_local_allgather(void* const* sends, void* const* recvs, int world, i64 count_per_rank, int f64, hipStream_t st)
{
    if (world > 16) { set_error("local communicator: at most 16 ranks"); return -100; }
    RankPtrs rp, rq;
    for (int r = 0; r < 16; ++r) { rp.p[r] = r < world ? sends[r] : nullptr; rq.p[r] = r < world ? recvs[r] : nullptr; }
    const i64 total = count_per_rank * world;
    const int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    if (grid < 1) return 0;
    if (f64) local_allgather_kernel<double><<<grid, 256, 0, st>>>(rp, rq, world, count_per_rank);
    else local_allgather_kernel<float><<<grid, 256, 0, st>>>(rp, rq, world, count_per_rank);
    SMK_HIP(hipGetLastError());
    return 0;
}

// sum the S slabs into one slab (fp32 or fp64) ahead of a cross-GPU sum
template <typename T>
__global__ __launch_bounds__(256) void reduce_partials_kernel(const double* __restrict__ p, int S, i64 slab, i64 count,
                                                              T* __restrict__ out)
{
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (i64)gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int t = 0; t < S; ++t) s += p[t * slab + i];
        out[i] = (T)s;
    }
}

// columns [c0, c0 + N) of the partial products (every slab holds ncols_pad columns of kpp doubles)
int launch_reduce_partials(PartialView pv, int k, i64 c0, i64 N, void* out /* [.][kpp], same column index */, int out_f64,
                           hipStream_t st)
{
    i64 count = N * pv.kpp;
    if (count <= 0) return 0;
    int grid = (int)((count + 255) / 256 < 4096 ? (count + 255) / 256 : 4096);
    const double* src = (const double*)pv.p + c0 * pv.kpp;
    if (out_f64) reduce_partials_kernel<double><<<grid, 256, 0, st>>>(src, pv.S, pv.slab, count, (double*)out + c0 * pv.kpp);
    else reduce_partials_kernel<float><<<grid, 256, 0, st>>>(src, pv.S, pv.slab, count, (float*)out + c0 * pv.kpp);
    SMK_HIP(hipGetLastError());
    return 0;
}

__global__ void sum_partials_kernel(const double* __restrict__ partials, int n, double* __restrict__ out)
{
    __shared__ double sh[16];
    double t = block_sum_array(partials, n, sh);
    if (threadIdx.x == 0) *out = t;
}

// two partial arrays in one launch (block b sums array b into out[b]); block 0 also mirrors the
// solver's failure flag into out[flag_slot] so that one 64-byte read-back carries everything
__global__ void sum_partials2_kernel(const double* __restrict__ p0, int n0, const double* __restrict__ p1, int n1,
                                     double* __restrict__ out, const int* __restrict__ flag, int flag_slot)
{
    __shared__ double sh[16];
    const double t = (blockIdx.x == 0) ? block_sum_array(p0, n0, sh) : block_sum_array(p1, n1, sh);
    if (threadIdx.x == 0) {
        out[blockIdx.x] = t;
        if (blockIdx.x == 0 && flag) out[flag_slot] = (double)*flag;
    }
}

// ==========================================================================
// Gram matrix  G(KP x KP, ld KP) = X X'   (X: KP x N fp64), deterministic two stage
// ==========================================================================
template <int KP>
__global__ __launch_bounds__(256) void gram_partial_kernel(const double* __restrict__ X, i64 N,
                                                           i64 cols_per_block, double* __restrict__ Gp)
{
    constexpr int CB = 32;                       // columns per LDS chunk
    constexpr int T = (KP >= 16) ? KP / 16 : 1;  // per-thread tile edge
    constexpr int GRID = (KP >= 16) ? 16 : KP;   // threads per tile edge
    __shared__ double xs[CB][KP + 1];
    const int tid = threadIdx.x;
    const int ti = tid / GRID, tj = tid % GRID;
    const bool active = tid < GRID * GRID;
    double acc[T][T];
#pragma unroll
    for (int a = 0; a < T; ++a)
#pragma unroll
        for (int b = 0; b < T; ++b) acc[a][b] = 0.0;

    const i64 c_begin = (i64)blockIdx.x * cols_per_block;
    i64 c_end = c_begin + cols_per_block;
    if (c_end > N) c_end = N;
    for (i64 c0 = c_begin; c0 < c_end; c0 += CB) {
        const int nc = (int)((c_end - c0 < CB) ? (c_end - c0) : CB);
        // coalesced: the chunk is nc*KP contiguous doubles
        for (int idx = tid; idx < CB * KP; idx += 256) {
            const int cc = idx / KP, r = idx % KP;
            xs[cc][r] = (cc < nc) ? X[(c0 + cc) * KP + r] : 0.0;
        }
        __syncthreads();
        if (active) {
#pragma unroll 4
            for (int cc = 0; cc < CB; ++cc) {
                double xa[T], xb[T];
#pragma unroll
                for (int a = 0; a < T; ++a) xa[a] = xs[cc][ti * T + a];
#pragma unroll
                for (int b = 0; b < T; ++b) xb[b] = xs[cc][tj * T + b];
#pragma unroll
                for (int a = 0; a < T; ++a)
#pragma unroll
                    for (int b = 0; b < T; ++b) acc[a][b] += xa[a] * xb[b];
            }
        }
        __syncthreads();
    }
    if (active) {
        double* out = Gp + (i64)blockIdx.x * KP * KP;
#pragma unroll
        for (int a = 0; a < T; ++a)
#pragma unroll
            for (int b = 0; b < T; ++b) out[(tj * T + b) * KP + (ti * T + a)] = acc[a][b];
    }
}

// fp64 matrix-core version (KP >= 16): v_mfma_f64_16x16x4_f64, 4 columns of X per instruction.
// A operand lane l: X[16*ti + (l&15)][c0 + (l>>4)], B operand the same with tj -- the Gram matrix
// needs no second operand load.  D: col = lane&15, row = (lane>>4) + 4*reg (f64 layout).

template <int KP>
__global__ __launch_bounds__(256) void gram_mfma_kernel(const double* __restrict__ X, i64 N, i64 cols_per_wave,
                                                        double* __restrict__ Gp)
{
    __shared__ double red[KP * KP];
    gram_mfma_body<KP>(X, N, cols_per_wave, Gp, (i64)blockIdx.x, red);        // gram_body.h
}

// KP = 128 (k in (64, 128]): the 8 x 8 grid of 16 x 16 tiles does not fit one wave's registers, so a workgroup
// computes TA = 2 tile rows (blockIdx.y picks them) against all 8 tile columns; 4 passes over X share the launch.
template <int KP, int TA>
__global__ __launch_bounds__(256) void gram_mfma_rows_kernel(const double* __restrict__ X, i64 N, i64 cols_per_wave,
                                                             double* __restrict__ Gp)
{
    constexpr int T = KP / 16;
    __shared__ double red[TA * 16 * KP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int a0 = blockIdx.y * TA;
    const i64 wg = (i64)blockIdx.x * 4 + wave;
    const i64 c_begin = wg * cols_per_wave;
    i64 c_end = c_begin + cols_per_wave;
    if (c_end > N) c_end = N;
    f64x4_t acc[TA][T];
#pragma unroll
    for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int b = 0; b < T; ++b) acc[a][b] = f64x4_t{0.0, 0.0, 0.0, 0.0};
    const int kc = lane >> 4, r16 = lane & 15;
    for (i64 c0 = c_begin; c0 < c_end; c0 += 4) {
        const i64 col = c0 + kc;
        const bool ok = col < c_end;
        double f[T], fa[TA];
#pragma unroll
        for (int t = 0; t < T; ++t) f[t] = ok ? X[col * KP + 16 * t + r16] : 0.0;
#pragma unroll
        for (int a = 0; a < TA; ++a) fa[a] = ok ? X[col * KP + 16 * (a0 + a) + r16] : 0.0;   // same cache lines as f[]
#pragma unroll
        for (int a = 0; a < TA; ++a)
#pragma unroll
            for (int b = 0; b < T; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a], f[b], acc[a][b], 0, 0, 0);
    }
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int a = 0; a < TA; ++a)
#pragma unroll
                for (int b = 0; b < T; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * a + kc + 4 * r, colm = 16 * b + r16;       // row inside this pass
                        const int idx = colm * (TA * 16) + row;
                        red[idx] = (w == 0) ? acc[a][b][r] : red[idx] + acc[a][b][r];
                    }
        }
        __syncthreads();
    }
    double* out = Gp + (i64)blockIdx.x * KP * KP;
    for (int i = threadIdx.x; i < TA * 16 * KP; i += 256) {
        const int colm = i / (TA * 16), row = i % (TA * 16);
        out[colm * KP + a0 * 16 + row] = red[i];
    }
}

// Gram partials AND the packed MFMA operand of the same factor in one pass (k <= 64, bf16 fragments): a wave walks
// its column range once for the fp64-MFMA Gram partials and once more (L1/L2-warm) for the hi/mid/lo fragments;
// gram_reduce_kernel then sums the partials.  Replaces gram_mfma + pack (two launches, two reads of the factor).
// (A last-arriver reduce inside this kernel was tried: one workgroup summing 64-256 partials is slower than the
// separate 4.7 us reduce launch.)
template <int KP, int NSPLIT>
__global__ __launch_bounds__(256) void gram_pack_kernel(const double* __restrict__ X, int k, i64 N, i64 cols_per_wave,
                                                        double* __restrict__ Gp, int KT, i64 nq,
                                                        unsigned char* __restrict__ out)
{
    constexpr int T = KP / 16;
    __shared__ double red[KP * KP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const i64 wg = (i64)blockIdx.x * 4 + wave;
    const i64 c_begin = wg * cols_per_wave;             // multiple of 16
    i64 c_end = c_begin + cols_per_wave;
    if (c_end > N) c_end = N;
    f64x4_t acc[T][T];
#pragma unroll
    for (int a = 0; a < T; ++a)
#pragma unroll
        for (int b = 0; b < T; ++b) acc[a][b] = f64x4_t{0.0, 0.0, 0.0, 0.0};
    const int kc = lane >> 4, r16 = lane & 15;
    for (i64 c0 = c_begin; c0 < c_end; c0 += 16) {
        double f[4][T];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const i64 col = c0 + 4 * u + kc;
            const bool ok = col < c_end;
#pragma unroll
            for (int t = 0; t < T; ++t) f[u][t] = ok ? X[col * KP + 16 * t + r16] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int a = 0; a < T; ++a)
#pragma unroll
                for (int b = a; b < T; ++b)         // upper blocks only (gram_mfma_kernel's note)
                    acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[u][a], f[u][b], acc[a][b], 0, 0, 0);
    }
    // ---- fragments of the same columns: chunk pair q covers rows 16 q .. 16 q + 15 of the operand (= columns of X)
    {
        const i64 q0 = c_begin / 16;
        i64 q1 = (c_begin + cols_per_wave) / 16;
        if (q1 > nq) q1 = nq;
        // the very last wave also writes the zero padding up to nq
        if (c_begin < N && c_begin + cols_per_wave >= N) q1 = nq;
        const int h = lane >> 5;
        for (i64 q = q0; q < q1; ++q)
            for (int kt = 0; kt < KT; ++kt) {
                const int r = kt * 32 + (lane & 31);
                const i64 row0 = (2 * q + h) * 8;
                double res[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const i64 row = row0 + e;
                    res[e] = (row < N && r < k) ? X[row * KP + r] : 0.0;
                }
#pragma unroll
                for (int sp = 0; sp < NSPLIT; ++sp) {
                    unsigned short hb[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        hb[e] = f32_to_bf16_rne((float)res[e]);
                        res[e] -= (double)bf16_bits_to_f32(hb[e]);
                    }
                    u32x4_t w;
#pragma unroll
                    for (int e = 0; e < 4; ++e) w[e] = (unsigned)hb[2 * e] | ((unsigned)hb[2 * e + 1] << 16);
                    *(u32x4_t*)(out + (((q * NSPLIT + sp) * KT + kt) * 64 + lane) * 16) = w;
                }
            }
    }
    // ---- deterministic in-block sum of the 4 waves (blocks below the diagonal from their mirror images)
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int a = 0; a < T; ++a)
#pragma unroll
                for (int b = a; b < T; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * a + kc + 4 * r, colm = 16 * b + r16;
                        const int idx = colm * KP + row;
                        red[idx] = (w == 0) ? acc[a][b][r] : red[idx] + acc[a][b][r];
                        if (a != b) {
                            const int idm = row * KP + colm;
                            red[idm] = (w == 0) ? acc[a][b][r] : red[idm] + acc[a][b][r];
                        }
                    }
        }
        __syncthreads();
    }
    double* mine = Gp + (i64)blockIdx.x * KP * KP;
    for (int i = threadIdx.x; i < KP * KP; i += 256) mine[i] = red[i];
}

// G[e] = sum_b Gp[b][e]: 16 elements per block, 16 thread groups stride the partials, fixed order (gram_body.h)
__global__ __launch_bounds__(256) void gram_reduce_kernel(const double* __restrict__ Gp, int nblk, int elems,
                                                          double* __restrict__ G, int KP = 0,
                                                          double* __restrict__ xscale = nullptr,
                                                          double* __restrict__ oscale = nullptr, double ascale = 1.0)
{
    __shared__ double sh[16][17];
    gram_reduce_body(Gp, nblk, elems, G, KP, xscale, oscale, ascale, (int)blockIdx.x, sh);
}

// KP = 8 (k <= 8: rank-2 hierarchical clustering lives here): one column (64 bytes) per thread per
// step, the 36 products of the upper triangle in registers, wave sums by DPP, four waves through LDS.
// Streams X once at HBM rate; the LDS-chunked kernel above spends its time in barriers at this width.
__global__ __launch_bounds__(256) void gram_stream8_kernel(const double* __restrict__ X, i64 N, double* __restrict__ Gp)
{
    constexpr int KP = 8;
    __shared__ double sh[4][36];
    double acc[36];
#pragma unroll
    for (int q = 0; q < 36; ++q) acc[q] = 0.0;
    for (i64 c = (i64)blockIdx.x * 256 + threadIdx.x; c < N; c += (i64)gridDim.x * 256) {
        double x[KP];
        const f64x2_t* p = (const f64x2_t*)(X + c * KP);
#pragma unroll
        for (int j = 0; j < KP / 2; ++j) {
            const f64x2_t v = p[j];
            x[2 * j] = v[0];
            x[2 * j + 1] = v[1];
        }
        int q = 0;
#pragma unroll
        for (int a = 0; a < KP; ++a)
#pragma unroll
            for (int b = a; b < KP; ++b) acc[q++] += x[a] * x[b];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < 36; ++q) {
        const double t = wave_sum(acc[q]);
        if (lane == 0) sh[wave][q] = t;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int a = threadIdx.x / KP, b = threadIdx.x % KP;
        const int lo = a < b ? a : b, hi = a < b ? b : a;
        const int q = lo * KP - lo * (lo - 1) / 2 + (hi - lo);       // index of (lo, hi) in the packed triangle
        Gp[(i64)blockIdx.x * KP * KP + threadIdx.x] = (sh[0][q] + sh[1][q]) + (sh[2][q] + sh[3][q]);
    }
}

// max |A| over a padded fp32 allocation (padding is zero): picks the power-of-two scale of the fp16 two-term product
__global__ __launch_bounds__(256) void absmax_f32_kernel(const float* __restrict__ A, i64 elems, unsigned* __restrict__ out)
{
    float m = 0.f;
    const i64 n4 = elems >> 2;
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n4; i += (i64)gridDim.x * 256) {
        const f32x4_t v = ((const f32x4_t*)A)[i];
        m = fmaxf(fmaxf(m, fabsf(v[0])), fmaxf(fabsf(v[1]), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(elems & 3)) m = fmaxf(m, fabsf(A[(n4 << 2) + threadIdx.x]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));      // non-negative floats order like their bits
}

// per-column max |A| folded into two words: out[0] = bits of the largest column maximum, out[1] = bits of the smallest
// NON-ZERO column maximum (0xFFFFFFFF if every column is zero).  Their ratio is the spread of the column scales: fp32-class
// products lose the small columns of an iteration once that spread passes ~2^28 (tests: columns spanning 2^+-20).
template <typename T>
__global__ __launch_bounds__(256) void colrange_kernel(const T* __restrict__ A, i64 ld, i64 rows, i64 cols, unsigned* __restrict__ out)
{
    const i64 wave = ((i64)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const i64 nw = ((i64)gridDim.x * 256) >> 6;
    for (i64 j = wave; j < cols; j += nw) {
        float m = 0.f;
        for (i64 r = lane; r < rows; r += 64) {
            float v;
            if constexpr (sizeof(T) == 2) v = bf16_bits_to_f32(A[j * ld + r]);
            else v = A[j * ld + r];
            m = fmaxf(m, fabsf(v));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if (lane == 0 && m > 0.f && m < 3.0e38f) {
            atomicMax(out, __float_as_uint(m));
            atomicMin(out + 1, __float_as_uint(m));
        }
    }
}

int launch_colrange(const void* A, int storage, i64 ld, i64 rows, i64 cols, unsigned* out2, hipStream_t st)
{
    // initialised on the device: an async copy from this frame's stack could be read after the frame is gone
    SMK_HIP(hipMemsetAsync(out2, 0, sizeof(unsigned), st));
    SMK_HIP(hipMemsetAsync(out2 + 1, 0xFF, sizeof(unsigned), st));
    i64 grid = (cols * 64 + 255) / 256;
    if (grid > 8192) grid = 8192;
    if (grid < 1) grid = 1;
    if (storage == STORE_BF16) colrange_kernel<unsigned short><<<(unsigned)grid, 256, 0, st>>>((const unsigned short*)A, ld, rows, cols, out2);
    else colrange_kernel<float><<<(unsigned)grid, 256, 0, st>>>((const float*)A, ld, rows, cols, out2);
    SMK_HIP(hipGetLastError());
    return 0;
}

// The largest sum of squares of a column (the a-priori bound on NNLS solutions that lets nnls_bpp_kernel<16> pack its own
// result, common.h NnlsPack): a wave per column, fp64 sums, the maximum through the bit pattern (non-negative doubles order
// like unsigned integers).  Infinities / NaNs in A leave the maximum alone (such a run fails elsewhere).
template <typename T>
__global__ __launch_bounds__(256) void colnorm2_max_kernel(const T* __restrict__ A, i64 ld, i64 rows, i64 cols, unsigned long long* __restrict__ out)
{
    const i64 wave = ((i64)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const i64 nw = ((i64)gridDim.x * 256) >> 6;
    for (i64 j = wave; j < cols; j += nw) {
        double s0 = 0.0, s1 = 0.0;
        i64 r = lane;
        for (; r + 64 < rows; r += 128) {
            float v0, v1;
            if constexpr (sizeof(T) == 2) { v0 = bf16_bits_to_f32(A[j * ld + r]); v1 = bf16_bits_to_f32(A[j * ld + r + 64]); }
            else { v0 = A[j * ld + r]; v1 = A[j * ld + r + 64]; }
            s0 += (double)v0 * (double)v0;
            s1 += (double)v1 * (double)v1;
        }
        if (r < rows) {
            float v0;
            if constexpr (sizeof(T) == 2) v0 = bf16_bits_to_f32(A[j * ld + r]);
            else v0 = A[j * ld + r];
            s0 += (double)v0 * (double)v0;
        }
        double s = s0 + s1;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0 && s > 0.0 && s < 1.0e300) atomicMax(out, (unsigned long long)__double_as_longlong(s));
    }
}

int launch_colnorm2_max(const void* A, int storage, i64 ld, i64 rows, i64 cols, double* out, hipStream_t st)
{
    SMK_HIP(hipMemsetAsync(out, 0, sizeof(double), st));
    i64 grid = (cols * 64 + 255) / 256;
    if (grid > 8192) grid = 8192;
    if (grid < 1) grid = 1;
    if (storage == STORE_BF16) colnorm2_max_kernel<unsigned short><<<(unsigned)grid, 256, 0, st>>>((const unsigned short*)A, ld, rows, cols, (unsigned long long*)out);
    else colnorm2_max_kernel<float><<<(unsigned)grid, 256, 0, st>>>((const float*)A, ld, rows, cols, (unsigned long long*)out);
    SMK_HIP(hipGetLastError());
    return 0;
}

// the counter-based uniform start of a factor, generated where it is used: X is KP x N (ld KP, pad rows zero); entry (r, j) of
// a k x N column-major host matrix has global index j * k + r (H), entry (i, c) of an N x k one has c * N + i (W kept transposed)
__global__ __launch_bounds__(256) void fill_factor_uniform_kernel(double* __restrict__ X, int KP, int k, i64 N, unsigned long long seed, int transposed)
{
    const i64 total = N * KP;
    for (i64 idx = (i64)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (i64)gridDim.x * blockDim.x) {
        const i64 j = idx / KP;
        const int r = (int)(idx % KP);
        double v = 0.0;
        if (r < k) v = (double)uniform_value(seed, (uint64_t)(transposed ? (i64)r * N + j : j * k + r), 0);
        X[idx] = v;
    }
}
int launch_fill_factor_uniform(double* X, int k, i64 N, unsigned long long seed, int transposed, hipStream_t st)
{
    const int KP = kp_of(k);
    i64 grid = (N * KP + 255) / 256;
    if (grid > 4096) grid = 4096;
    if (grid < 1) grid = 1;
    fill_factor_uniform_kernel<<<(unsigned)grid, 256, 0, st>>>(X, KP, k, N, seed, transposed);
    SMK_HIP(hipGetLastError());
    return 0;
}

// run-time guard of the product form (solver.cpp): sum of squared differences between the fast-form product (P1, all row
// splits) and the accurate-form product of the same `ncols` sampled columns, and the accurate product's sum of squares
__global__ __launch_bounds__(256) void guard_compare_kernel(PartialView fast, const unsigned* __restrict__ cols, int ncols,
                                                            const double* __restrict__ acc, int S_acc, i64 slab_acc, int kpp, int k,
                                                            double* __restrict__ out2)
{
    __shared__ double sh[16];
    double num = 0.0, den = 0.0;
    for (int idx = threadIdx.x; idx < ncols * k; idx += blockDim.x) {
        const int i = idx / k, r = idx % k;
        const double f = rhs_elem(fast, (i64)cols[i], r);
        double a = 0.0;
        for (int s = 0; s < S_acc; ++s) a += acc[s * slab_acc + (i64)i * kpp + r];
        num += (f - a) * (f - a);
        den += a * a;
    }
    const double tn = block_sum(num, sh);
    const double td = block_sum(den, sh);
    if (threadIdx.x == 0) { out2[0] = tn; out2[1] = td; }
}
int launch_guard_compare(PartialView fast, const unsigned* cols, int ncols, const double* acc, int S_acc, i64 slab_acc, int kpp, int k,
                         double* out2, hipStream_t st)
{
    guard_compare_kernel<<<1, 256, 0, st>>>(fast, cols, ncols, acc, S_acc, slab_acc, kpp, k, out2);
    SMK_HIP(hipGetLastError());
    return 0;
}

int launch_absmax_f32(const float* A, i64 elems, unsigned* out, hipStream_t st)
{
    SMK_HIP(hipMemsetAsync(out, 0, sizeof(unsigned), st));
    i64 grid = (elems / 4 + 255) / 256;
    if (grid > 4096) grid = 4096;
    if (grid < 1) grid = 1;
    absmax_f32_kernel<<<(unsigned)grid, 256, 0, st>>>(A, elems, out);
    SMK_HIP(hipGetLastError());
    return 0;
}

size_t gram_scratch_elems(int k, int max_blocks)
{
    int KP = kp_of(k);
    if (is_wide(k)) return (size_t)gram_wide_blocks(KP, (i64)1 << 40, max_blocks) * KP * KP + 8;
    return (size_t)max_blocks * KP * KP + 8;             // + the ticket word of the fused Gram/pack kernel
}

// partial Gram matrices of the columns [0, N) of X into scratch ([*nblk_out][KP * KP]); several calls with scratch
// offsets (row segments of a factor) followed by ONE launch_gram_reduce give the Gram matrix of the union
// the blocking of gram_mfma_kernel<16 / 32 / 64> for N columns (also used by the launches that carry the partial sums as riders)
void gram_partial_shape(i64 N, int max_blocks, int* nblk_out, i64* cpw_out)
{
    int nblk = (int)((N + 63) / 64);
    if (nblk > max_blocks) nblk = max_blocks;
    if (nblk < 1) nblk = 1;
    i64 cpw = (N + (i64)nblk * 4 - 1) / ((i64)nblk * 4);
    cpw = (cpw + 15) / 16 * 16;
    *nblk_out = nblk;
    *cpw_out = cpw;
}

int launch_gram_partials(const double* X, int k, i64 N, double* scratch, int max_blocks, int* nblk_out, hipStream_t st)
{
    const int KP = kp_of(k);
    int nblk;
    if (is_wide(k)) {
        const int rc = launch_gram_wide_partials(X, KP, N, scratch, max_blocks, &nblk, st);
        if (rc) return rc;
    } else if (KP >= 16) {
        // one 16-column trip per wave while that keeps the partials within max_blocks (N <= 64 max_blocks), 64 and more columns
        // per wave above: a factor of 4096 columns was 16 workgroups of 4 sequential trips, 26 us at KP = 64 on an idle chip
        i64 cpw;
        gram_partial_shape(N, max_blocks, &nblk, &cpw);
        switch (KP) {
            case 16: gram_mfma_kernel<16><<<nblk, 256, 0, st>>>(X, N, cpw, scratch); break;
            case 32: gram_mfma_kernel<32><<<nblk, 256, 0, st>>>(X, N, cpw, scratch); break;
            case 64: gram_mfma_kernel<64><<<nblk, 256, 0, st>>>(X, N, cpw, scratch); break;
            default: gram_mfma_rows_kernel<128, 2><<<dim3(nblk, 4), 256, 0, st>>>(X, N, cpw, scratch); break;
        }
    } else {
        nblk = (int)((N + 255) / 256);
        if (nblk > max_blocks) nblk = max_blocks;
        if (nblk < 1) nblk = 1;
        gram_stream8_kernel<<<nblk, 256, 0, st>>>(X, N, scratch);
    }
    SMK_HIP(hipGetLastError());
    *nblk_out = nblk;
    return 0;
}

int launch_gram_reduce(const double* scratch, int nblk, int k, double* G, hipStream_t st, double* xscale, double* oscale,
                       double ascale)
{
    const int KP = kp_of(k), elems = KP * KP;
    gram_reduce_kernel<<<(elems + 15) / 16, 256, 0, st>>>(scratch, nblk, elems, G, KP, xscale, oscale, ascale);
    SMK_HIP(hipGetLastError());
    return 0;
}

// the row scales of the fp16 two-term operand from a FINISHED Gram matrix (sharded runs: after its all-reduce);
// the same rule as gram_reduce_kernel
__global__ void gram_scales_kernel(const double* __restrict__ G, int KP, int k, double* __restrict__ xscale,
                                   double* __restrict__ oscale, double ascale)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= KP) return;
    const double t = G[(i64)r * KP + r];
    int ex = 0;
    double xs = 1.0;
    if (t > 0.0 && t < 1.0e300) {
        (void)frexp(t, &ex);
        const int half = (ex >= 0) ? (ex + 1) / 2 : -((-ex) / 2);
        xs = ldexp(1.0, 14 - half);
    }
    xscale[r] = xs;
    if (oscale) oscale[r] = 1.0 / (xs * ascale);
}

int launch_gram_scales(const double* G, int k, double* xscale, double* oscale, double ascale, hipStream_t st)
{
    const int KP = kp_of(k);
    gram_scales_kernel<<<(KP + 63) / 64, 64, 0, st>>>(G, KP, k, xscale, oscale, ascale);
    SMK_HIP(hipGetLastError());
    return 0;
}

int launch_gram(const double* X, int k, i64 N, double* G, double* scratch, int max_blocks, hipStream_t st, double* xscale,
                double* oscale, double ascale)
{
    int nblk = 0;
    const int rc = launch_gram_partials(X, k, N, scratch, max_blocks, &nblk, st);
    if (rc) return rc;
    return launch_gram_reduce(scratch, nblk, k, G, st, xscale, oscale, ascale);
}

// gram scratch: [max_blocks][KP*KP] partials followed by one ticket word.  Returns 1 when this shape has no fused
// kernel (the caller then runs launch_gram + launch_pack).
int launch_gram_pack(const double* X, int k, i64 N, double* G, double* scratch, int max_blocks, int storage, int nsplit,
                     void* packed, hipStream_t st)
{
    const int KP = kp_of(k);
    const bool bf16_frag = storage == STORE_BF16 || nsplit >= 2;
    if (KP < 16 || KP > 64 || !bf16_frag || nsplit < 1 || nsplit > 3) return 1;     // (the fp16 form needs the finished Gram diagonal first)
    static const bool enabled = [] { const char* e = getenv("SMK_FUSED_GRAM"); return !(e && e[0] == '0'); }();
    if (!enabled) return 1;
    int nblk = (int)((N + 63) / 64);                     // one 16-column trip per wave while the partials stay within max_blocks, as launch_gram
    if (nblk > max_blocks) nblk = max_blocks;
    if (nblk < 1) nblk = 1;
    i64 cpw = (N + (i64)nblk * 4 - 1) / ((i64)nblk * 4);
    cpw = (cpw + 15) / 16 * 16;
    const int KT = kt_of(k);
    const i64 nq = round_up(N, ROW_PAD) / 16;
#define SMK_GP(KPX, NSX) gram_pack_kernel<KPX, NSX><<<nblk, 256, 0, st>>>(X, k, N, cpw, scratch, KT, nq, (unsigned char*)packed)
    switch (KP * 10 + nsplit) {
        case 161: SMK_GP(16, 1); break; case 162: SMK_GP(16, 2); break; case 163: SMK_GP(16, 3); break;
        case 321: SMK_GP(32, 1); break; case 322: SMK_GP(32, 2); break; case 323: SMK_GP(32, 3); break;
        case 641: SMK_GP(64, 1); break; case 642: SMK_GP(64, 2); break; default: SMK_GP(64, 3); break;
    }
#undef SMK_GP
    SMK_HIP(hipGetLastError());
    gram_reduce_kernel<<<(KP * KP + 15) / 16, 256, 0, st>>>(scratch, nblk, KP * KP, G);
    SMK_HIP(hipGetLastError());
    return 0;
}

// ==========================================================================
// column-tile kernels.  Thread (column j, sub-lane s) owns X[4s..4s+3, j].
// G is symmetric, so row r of G at "my" columns is gs[r*KP + 4s .. +3]: one
// 32-byte LDS read, identical addresses across the columns of a wave.
// ==========================================================================
#define COLTILE_PROLOGUE(KP)                                                          \
    constexpr int LPC = KP / 4;                                                       \
    extern __shared__ __attribute__((aligned(16))) double gs[];   /* KP * KP doubles */ \
    for (int i_ = threadIdx.x; i_ < KP * KP; i_ += blockDim.x) gs[i_] = G[i_];        \
    __syncthreads();                                                                  \
    const i64 gtid = (i64)blockIdx.x * blockDim.x + threadIdx.x;                      \
    const i64 j = gtid / LPC;                                                         \
    const int s = (int)(gtid % LPC);                                                  \
    const bool valid = j < N;                                                         \
    const i64 jc = valid ? j : (N - 1);

template <int KP>
__device__ __forceinline__ double dot_row(const double* gs, int r, int s, const double (&x)[4])
{
    double g[4];
    load4(gs + r * KP + 4 * s, g);
    return (g[0] * x[0] + g[1] * x[1]) + (g[2] * x[2] + g[3] * x[3]);
}

// ---- MU: x <- x .* R ./ (G x + 1e-13)      (nmf_solver_mu.hpp:22, :27-71) -------------
template <int KP>
__global__ __launch_bounds__(256) void mu_update_kernel(double* __restrict__ X, int k, i64 N, PartialView R,
                                                        const double* __restrict__ G)
{
    COLTILE_PROLOGUE(KP)
    double x[4], b[4], d[4] = {0, 0, 0, 0};
    load4(X + jc * KP + 4 * s, x);
    load_rhs4(R, jc, 4 * s, b);
#pragma unroll
    for (int r = 0; r < KP; ++r) {
        if (r < k) {
            const double dot = group_sum<LPC>(dot_row<KP>(gs, r, s, x));
            if (s == r / 4) d[r % 4] = dot;
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (4 * s + e < k) x[e] = x[e] * (b[e] / (d[e] + 1.0e-13));
    if (valid) store4(X + j * KP + 4 * s, x);
}

// ---- HALS H sweep: rows r = 0..k-1 in order, Gauss-Seidel inside the column ----------
//      (nmf_solver_hals.hpp:26-62)
template <int KP>
__global__ __launch_bounds__(256) void hals_sweep_kernel(double* __restrict__ X, int k, i64 N, PartialView R,
                                                         const double* __restrict__ G)
{
    COLTILE_PROLOGUE(KP)
    double x[4], b[4];
    load4(X + jc * KP + 4 * s, x);
    load_rhs4(R, jc, 4 * s, b);
#pragma unroll
    for (int r = 0; r < KP; ++r) {
        if (r < k) {
            const double dot = group_sum<LPC>(dot_row<KP>(gs, r, s, x));
            if (s == r / 4) {
                double v = x[r % 4] + (b[r % 4] - dot) / gs[r * KP + r];
                if (isnan(v) || v < 0.0) v = 0.0;
                x[r % 4] = v;
            }
        }
    }
    if (valid) store4(X + j * KP + 4 * s, x);
}

// ---- epilogue of the two HALS sweeps (round 6): packed operand + Gram partial from the tile a workgroup has just updated ---------
// Until round 6 a HALS iteration at k <= 32 ran gram_pack_kernel after each sweep: a launch that re-reads the factor to write
// (a) the MFMA operand fragments of the streaming product that follows and (b) partial Gram matrices.  Both sweeps hold the new
// values in registers when they finish, so they leave them in an LDS tile ([column of the factor][component], leading dimension
// KP + 1: conflict-free for the row-wise writes and the fragment reads) and this function does (a) and (b) from there --
// gram_pack_kernel's arithmetic and layouts: fragment lane (r = component, h) holds 8 consecutive operand rows as bf16 terms, the
// Gram partial is v_mfma_f64_16x16x4 over the tile's columns, the waves' shares added in wave order.  C3: two launches of 11 us
// fewer per iteration (profiles/r06_c3_epilogues.txt).  ncols: multiple of 16, (ncols / waves) a multiple of 4.
template <int KP>
__device__ __forceinline__ void tile_pack_gram(const double* tile, int ncols, i64 col0, i64 N, int k, int KT, int nsplit, i64 nq,
                                               bool last_wg, unsigned char* __restrict__ out, double* __restrict__ Gp_block,
                                               double* red /* LDS, KP * KP */)
{
    constexpr int T = KP / 16;
    constexpr int LD = KP + 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    // ---- fragments: chunk pair q covers operand rows 16 q .. 16 q + 15 (= columns of the factor)
    {
        const int npairs = ncols / 16;
        const i64 q0 = col0 / 16;
        const int h = lane >> 5;
        for (int qi = wave; qi < npairs; qi += nwaves) {
            const i64 q = q0 + qi;
            if (q >= nq) continue;
            for (int kt = 0; kt < KT; ++kt) {
                const int r = kt * 32 + (lane & 31);
                const int lrow0 = (2 * qi + h) * 8;
                double res[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) res[e] = (col0 + lrow0 + e < N && r < k) ? tile[(lrow0 + e) * LD + r] : 0.0;
#pragma unroll
                for (int sp = 0; sp < 3; ++sp) {
                    if (sp < nsplit) {
                        unsigned short hb[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            hb[e] = f32_to_bf16_rne((float)res[e]);
                            res[e] -= (double)bf16_bits_to_f32(hb[e]);
                        }
                        u32x4_t w;
#pragma unroll
                        for (int e = 0; e < 4; ++e) w[e] = (unsigned)hb[2 * e] | ((unsigned)hb[2 * e + 1] << 16);
                        *(u32x4_t*)(out + (((q * nsplit + sp) * KT + kt) * 64 + lane) * 16) = w;
                    }
                }
            }
        }
        if (last_wg) {               // zero padding of the operand up to nq chunk pairs
            const u32x4_t z = {0u, 0u, 0u, 0u};
            for (i64 q = q0 + npairs + wave; q < nq; q += nwaves)
                for (int kt = 0; kt < KT; ++kt)
                    for (int sp = 0; sp < nsplit; ++sp) *(u32x4_t*)(out + (((q * nsplit + sp) * KT + kt) * 64 + lane) * 16) = z;
        }
    }
    // ---- Gram partial of the tile's columns (the expensive half of this epilogue: +9.4 us in the W sweep, +6.7 us in the H sweep on
    // C3 against +4.7 / +1.9 us for the fragments -- profiles/r06_c3_epilogues.txt)
    f64x4_t acc[T][T];
#pragma unroll
    for (int a = 0; a < T; ++a)
#pragma unroll
        for (int b = 0; b < T; ++b) acc[a][b] = f64x4_t{0.0, 0.0, 0.0, 0.0};
    const int kc = lane >> 4, r16 = lane & 15;
    const int cpw = ncols / nwaves;
    for (int c0 = wave * cpw; c0 < (wave + 1) * cpw; c0 += 4) {
        double f[T];
#pragma unroll
        for (int t = 0; t < T; ++t) f[t] = tile[(c0 + kc) * LD + 16 * t + r16];          // columns past N hold zeros
#pragma unroll
        for (int a = 0; a < T; ++a)
#pragma unroll
            for (int b = a; b < T; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[a], f[b], acc[a][b], 0, 0, 0);   // upper blocks only (gram_mfma_kernel's note)
    }
    for (int w = 0; w < nwaves; ++w) {
        if (wave == w) {
#pragma unroll
            for (int a = 0; a < T; ++a)
#pragma unroll
                for (int b = a; b < T; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * a + kc + 4 * r, colm = 16 * b + r16;
                        const int idx = colm * KP + row;
                        red[idx] = (w == 0) ? acc[a][b][r] : red[idx] + acc[a][b][r];
                        if (a != b) {
                            const int idm = row * KP + colm;
                            red[idm] = (w == 0) ? acc[a][b][r] : red[idm] + acc[a][b][r];
                        }
                    }
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < KP * KP; i += blockDim.x) Gp_block[i] = red[i];
}

// hals_sweep_kernel + the epilogue above (UpdateH_Hals, nmf_solver_hals.hpp:26-62)
template <int KP>
__global__ __launch_bounds__(256) void hals_sweep_pack_kernel(double* __restrict__ X, int k, i64 N, PartialView R,
                                                              const double* __restrict__ G, unsigned char* __restrict__ pack_out,
                                                              double* __restrict__ Gp, int KT, i64 nq, int nsplit)
{
    COLTILE_PROLOGUE(KP)
    constexpr int CPW = 256 / LPC;                  // columns per workgroup
    __shared__ double tile[CPW * (KP + 1)];
    __shared__ double red[KP * KP];
    double x[4], b[4];
    load4(X + jc * KP + 4 * s, x);
    load_rhs4(R, jc, 4 * s, b);
#pragma unroll
    for (int r = 0; r < KP; ++r) {
        if (r < k) {
            const double dot = group_sum<LPC>(dot_row<KP>(gs, r, s, x));
            if (s == r / 4) {
                double v = x[r % 4] + (b[r % 4] - dot) / gs[r * KP + r];
                if (isnan(v) || v < 0.0) v = 0.0;
                x[r % 4] = v;
            }
        }
    }
    if (valid) store4(X + j * KP + 4 * s, x);
    {
        double* t = tile + (threadIdx.x / LPC) * (KP + 1) + 4 * s;
#pragma unroll
        for (int e = 0; e < 4; ++e) t[e] = valid ? x[e] : 0.0;
    }
    __syncthreads();
    tile_pack_gram<KP>(tile, CPW, (i64)blockIdx.x * CPW, N, k, KT, nsplit, nq, blockIdx.x == gridDim.x - 1, pack_out,
                       Gp + (i64)blockIdx.x * KP * KP, red);
}

// ---- gradient g = G x - R, projected-gradient partial sums ---------------------------
//      (mu :156-161, hals :181-195, bpp :370-371; projected_gradient.hpp:125-171)
// one workgroup of the gradient / projected-gradient pass over the column tile `bid` of X
template <int KP>
__device__ __forceinline__ void grad_pg_body(const double* __restrict__ X, int k, i64 N, const PartialView& R,
                                             const double* __restrict__ G, double* __restrict__ grad_out,
                                             double* __restrict__ partials, int bid, double* gs, double* sh)
{
    constexpr int LPC = KP / 4;
    for (int i_ = threadIdx.x; i_ < KP * KP; i_ += blockDim.x) gs[i_] = G[i_];
    __syncthreads();
    const i64 gtid = (i64)bid * blockDim.x + threadIdx.x;
    const i64 j = gtid / LPC;
    const int s = (int)(gtid % LPC);
    const bool valid = j < N;
    const i64 jc = valid ? j : (N - 1);
    double x[4], b[4], g[4] = {0, 0, 0, 0};
    load4(X + jc * KP + 4 * s, x);
    load_rhs4(R, jc, 4 * s, b);
#pragma unroll
    for (int r = 0; r < KP; ++r) {
        if (r < k) {
            const double dot = group_sum<LPC>(dot_row<KP>(gs, r, s, x));
            if (s == r / 4) g[r % 4] = dot - b[r % 4];
        }
    }
    double sum = 0.0;
    if (valid) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (4 * s + e < k && (g[e] < 0.0 || x[e] > 0.0)) sum += g[e] * g[e];
        if (grad_out) store4(grad_out + j * KP + 4 * s, g);
    }
    const double t = block_sum(sum, sh);
    if (threadIdx.x == 0) partials[bid] = t;
}

template <int KP>
__global__ __launch_bounds__(256) void grad_pg_kernel(const double* __restrict__ X, int k, i64 N, PartialView R,
                                                      const double* __restrict__ G, double* __restrict__ grad_out,
                                                      double* __restrict__ partials)
{
    __shared__ double sh[16];
    extern __shared__ __attribute__((aligned(16))) double gs[];      // KP * KP doubles
    grad_pg_body<KP>(X, k, N, R, G, grad_out, partials, blockIdx.x, gs, sh);
}

// both factors in one launch: workgroups [0, grid1) take side 1 (W'), the rest side 2 (H)
template <int KP>
__global__ __launch_bounds__(256) void grad_pg2_kernel(const double* __restrict__ X1, i64 N1, PartialView R1,
                                                       const double* __restrict__ G1, double* __restrict__ part1,
                                                       int grid1, const double* __restrict__ X2, i64 N2, PartialView R2,
                                                       const double* __restrict__ G2, double* __restrict__ part2, int k)
{
    __shared__ double sh[16];
    extern __shared__ __attribute__((aligned(16))) double gs[];      // KP * KP doubles
    if ((int)blockIdx.x < grid1) grad_pg_body<KP>(X1, k, N1, R1, G1, nullptr, part1, blockIdx.x, gs, sh);
    else grad_pg_body<KP>(X2, k, N2, R2, G2, nullptr, part2, blockIdx.x - grid1, gs, sh);
}

// The progress check of a PG_RATIO run in TWO launches and no copy packet (round 6): both gradient sums as grad_pg2_kernel plus the
// snapshot of (W', H, W'W) that lets the driver undo a speculated iteration (snapshot_kernel's layout; the tiles are in registers
// anyway), then sum_partials2_host_kernel: the two totals in sum_partials2_kernel's order and the failure flag, written straight
// into the PINNED HOST slot the driver reads after the event.  Before: four stream operations per checked iteration (gradients,
// sums, a 64-byte copy, the snapshot); the reference forms its gradients and evaluates the rule every iteration
// (nmf_solve_generic.hpp:98-121).  (ONE launch with a last-arriver sum was built first and measured 20 - 30 % SLOWER on every
// configuration: the agent-scope fence each workgroup needs before its ticket writes back the XCD's L2, which this very kernel
// fills with the snapshot -- profiles/r06_progress_check.txt.)
template <int KP>
__global__ __launch_bounds__(256) void grad_pg2_snap_kernel(const double* __restrict__ X1, i64 N1, PartialView R1,
                                                            const double* __restrict__ G1, double* __restrict__ part1,
                                                            int grid1, const double* __restrict__ X2, i64 N2, PartialView R2,
                                                            const double* __restrict__ G2, double* __restrict__ part2, int k,
                                                            double* __restrict__ snap, int k2, int skip1)
{
    constexpr int LPC = KP / 4;
    __shared__ double sh[16];
    extern __shared__ __attribute__((aligned(16))) double gs[];      // KP * KP doubles
    const bool side1 = (int)blockIdx.x < grid1;
    const int bid = side1 ? (int)blockIdx.x : (int)blockIdx.x - grid1;
    // skip1 (BPP): the gradient of side 1 is the dual Y of the NNLS solve that produced W' (nmf_solver_bpp.hpp:362-366) -- zero on the
    // passive set by construction and non-negative elsewhere, so its projected-gradient sum is exactly 0 and the S slabs of the
    // right-hand side need not be read again; the workgroups of side 1 only take the snapshot
    if (side1 && skip1) { if (threadIdx.x == 0) part1[bid] = 0.0; }
    else if (side1) grad_pg_body<KP>(X1, k, N1, R1, G1, nullptr, part1, bid, gs, sh);
    else grad_pg_body<KP>(X2, k, N2, R2, G2, nullptr, part2, bid, gs, sh);
    if (snap) {
        // snapshot_kernel's layout: [W': N1 columns of k2 / 2 pairs][H: N2 columns][W'W = G2: KP * KP / 2 pairs]
        const int h2 = k2 / 2;
        const i64 gtid = (i64)bid * blockDim.x + threadIdx.x;
        const i64 j = gtid / LPC;
        const int sl = (int)(gtid % LPC);
        const double* X = side1 ? X1 : X2;
        const i64 N = side1 ? N1 : N2;
        f64x2_t* b2 = (f64x2_t*)snap + (side1 ? 0 : N1 * h2);
        if (j < N) {
            const f64x2_t a = *(const f64x2_t*)(X + j * KP + 4 * sl), b = *(const f64x2_t*)(X + j * KP + 4 * sl + 2);
            if (2 * sl < h2) b2[j * h2 + 2 * sl] = a;
            if (2 * sl + 1 < h2) b2[j * h2 + 2 * sl + 1] = b;
        }
        if (!side1 && bid == 0) {        // gs holds G2 = W'W
            f64x2_t* g2 = (f64x2_t*)snap + (N1 + N2) * h2;
            for (int i = threadIdx.x; i < KP * KP / 2; i += blockDim.x) g2[i] = *(const f64x2_t*)(gs + 2 * i);
        }
    }
}

// sum_partials2_kernel in ONE workgroup that also writes the result where the host reads it (pinned memory: no copy packet)
__global__ __launch_bounds__(256) void sum_partials2_host_kernel(const double* __restrict__ p0, int n0, const double* __restrict__ p1, int n1,
                                                                 double* __restrict__ out, double* __restrict__ host_out,
                                                                 const int* __restrict__ flag, int flag_slot, double host_tag)
{
    __shared__ double sh[16];
    const double t0 = block_sum_array(p0, n0, sh);
    const double t1 = block_sum_array(p1, n1, sh);
    if (threadIdx.x == 0) {
        const double f = flag ? (double)*flag : 0.0;
        out[0] = t0; out[1] = t1; out[flag_slot] = f;
        host_out[0] = t0; host_out[1] = t1; host_out[flag_slot] = f;
        if (host_tag != 0.0) {               // the host polls the slot: the tag goes out last
            __threadfence_system();
            __hip_atomic_store(&host_out[7], host_tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

__global__ __launch_bounds__(256) void pg_from_grad_kernel(const double* __restrict__ X, const double* __restrict__ Y,
                                                           i64 count, double* __restrict__ partials)
{
    __shared__ double sh[16];
    double sum = 0.0;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (i64)gridDim.x * blockDim.x) {
        const double g = Y[i];
        if (g < 0.0 || X[i] > 0.0) sum += g * g;
    }
    const double t = block_sum(sum, sh);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
}


static inline int coltile_grid(int KP, i64 N) { return (int)((N * (KP / 4) + 255) / 256); }
// the column-tile kernels keep G in dynamic LDS: KP * KP doubles (128 KiB for KP = 128: opt in once per kernel)
template <typename K>
static int coltile_lds(K kern, int KP)
{
    const int bytes = KP * KP * (int)sizeof(double);
    if (bytes > 48 * 1024) SMK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    return bytes;
}
#define COLTILE_CASE(KPX, KERN, GRID, ...)                                                \
    case KPX: {                                                                           \
        const int lds_ = coltile_lds(KERN<KPX>, KPX);                                     \
        if (lds_ < 0) return lds_;                                                        \
        KERN<KPX><<<(GRID), 256, lds_, st>>>(__VA_ARGS__);                                \
    } break;
#define COLTILE_LAUNCH(KERN, GRID, ...)                                                   \
    switch (KPv) {                                                                        \
        COLTILE_CASE(8, KERN, GRID, __VA_ARGS__)                                          \
        COLTILE_CASE(16, KERN, GRID, __VA_ARGS__)                                         \
        COLTILE_CASE(32, KERN, GRID, __VA_ARGS__)                                         \
        COLTILE_CASE(64, KERN, GRID, __VA_ARGS__)                                         \
        default: { COLTILE_CASE(128, KERN, GRID, __VA_ARGS__) }                           \
    }

int launch_mu_update(double* X, int k, i64 N, PartialView R, const double* G, hipStream_t st, double* wide_tmp)
{
    if (is_wide(k)) return launch_mu_update_wide(X, k, N, R, G, st, wide_tmp);
    const int KPv = kp_of(k), grid = coltile_grid(KPv, N);
    COLTILE_LAUNCH(mu_update_kernel, grid, X, k, N, R, G);
    SMK_HIP(hipGetLastError());
    return 0;
}

int launch_hals_sweep(double* X, int k, i64 N, PartialView R, const double* G, hipStream_t st, HalsEpilogue* ep)
{
    if (ep) ep->done = false;
    if (is_wide(k)) return launch_hals_sweep_wide(X, k, N, R, G, st);
    const int KPv = kp_of(k), grid = coltile_grid(KPv, N);
    if (ep && ep->pack_out && ep->Gp && (KPv == 16 || KPv == 32) && grid <= ep->max_blocks) {
        const int lds = KPv * KPv * (int)sizeof(double);
        if (KPv == 16) hals_sweep_pack_kernel<16><<<grid, 256, lds, st>>>(X, k, N, R, G, ep->pack_out, ep->Gp, ep->KT, ep->nq, ep->nsplit);
        else hals_sweep_pack_kernel<32><<<grid, 256, lds, st>>>(X, k, N, R, G, ep->pack_out, ep->Gp, ep->KT, ep->nq, ep->nsplit);
        SMK_HIP(hipGetLastError());
        ep->nblk = grid;
        ep->done = true;
        return 0;
    }
    COLTILE_LAUNCH(hals_sweep_kernel, grid, X, k, N, R, G);
    SMK_HIP(hipGetLastError());
    return 0;
}

// the per-workgroup partial sums only (*grid_out of them); several calls with partials offsets (row segments of a
// factor) followed by ONE launch_sum_partials give the sum over the union
int launch_grad_pg_partials(const double* X, int k, i64 N, PartialView R, const double* G, double* grad_out,
                            double* pg_partials, int* grid_out, hipStream_t st, double* wide_tmp)
{
    if (is_wide(k)) return launch_grad_pg_wide(X, k, N, R, G, grad_out, pg_partials, grid_out, st, wide_tmp);
    const int KPv = kp_of(k), grid = coltile_grid(KPv, N);
    COLTILE_LAUNCH(grad_pg_kernel, grid, X, k, N, R, G, grad_out, pg_partials);
    SMK_HIP(hipGetLastError());
    *grid_out = grid;
    return 0;
}

int launch_sum_partials(const double* partials, int n, double* out, hipStream_t st)
{
    sum_partials_kernel<<<1, 256, 0, st>>>(partials, n, out);
    SMK_HIP(hipGetLastError());
    return 0;
}

int launch_grad_pg(const double* X, int k, i64 N, PartialView R, const double* G, double* grad_out,
                   double* pg_partials, double* pg_accum, int slot, hipStream_t st, double* wide_tmp)
{
    int g = 0;
    const int rc = launch_grad_pg_partials(X, k, N, R, G, grad_out, pg_partials, &g, st, wide_tmp);
    if (rc) return rc;
    return launch_sum_partials(pg_partials, g, pg_accum + slot, st);
}

// projected-gradient sums of both factors: pg_accum[0] (side 1), pg_accum[1] (side 2), and the failure flag
// as a double in pg_accum[flag_slot]; two launches instead of four
int launch_grad_pg2(const double* X1, i64 N1, PartialView R1, const double* G1, double* part1, const double* X2, i64 N2,
                    PartialView R2, const double* G2, double* part2, int k, double* pg_accum, const int* flag,
                    int flag_slot, hipStream_t st)
{
    const int KPv = kp_of(k), g1 = coltile_grid(KPv, N1), g2 = coltile_grid(KPv, N2);
    COLTILE_LAUNCH(grad_pg2_kernel, g1 + g2, X1, N1, R1, G1, part1, g1, X2, N2, R2, G2, part2, k);
    SMK_HIP(hipGetLastError());
    sum_partials2_kernel<<<2, 256, 0, st>>>(part1, g1, part2, g2, pg_accum, flag, flag_slot);
    SMK_HIP(hipGetLastError());
    return 0;
}

// totals of a deferred progress check (common.h: NnlsRiders, launch_pg_defer_sum)
__global__ __launch_bounds__(256) void pg_defer_sum_kernel(const double* __restrict__ part, int n, double* __restrict__ out,
                                                           double* __restrict__ host_out, const int* __restrict__ flag, int flag_slot,
                                                           int tag_limit, const double* __restrict__ G, double* __restrict__ snap_g, int kk,
                                                           double host_tag)
{
    __shared__ double sh[16];
    const double t1 = block_sum_array(part, n, sh);
    if (snap_g)
        for (int i = threadIdx.x; i < kk; i += blockDim.x) snap_g[i] = G[i];
    if (threadIdx.x == 0) {
        int fv = flag ? *flag : INT_MAX;
        if (fv != INT_MAX && fv > tag_limit) fv = INT_MAX;        // a failure of the speculated NEXT iteration is not this check's
        const double f = (double)fv;
        out[0] = 0.0; out[1] = t1; out[flag_slot] = f;
        host_out[0] = 0.0; host_out[1] = t1; host_out[flag_slot] = f;
        if (host_tag != 0.0) {               // the host polls the slot: the tag goes out last
            __threadfence_system();
            __hip_atomic_store(&host_out[7], host_tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
int launch_pg_defer_sum(const double* part, int n, double* out, double* host_out, const int* flag, int flag_slot, int tag_limit,
                        const double* G, double* snap_g, int kk, hipStream_t st, double host_tag)
{
    pg_defer_sum_kernel<<<1, 256, 0, st>>>(part, n, out, host_out, flag, flag_slot, tag_limit, G, snap_g, kk, host_tag);
    SMK_HIP(hipGetLastError());
    return 0;
}

int launch_grad_pg2_fused(const double* X1, i64 N1, PartialView R1, const double* G1, double* part1, const double* X2, i64 N2,
                          PartialView R2, const double* G2, double* part2, int k, double* pg_accum, const int* flag,
                          int flag_slot, double* snap, double* host_out, hipStream_t st, int skip1, double host_tag)
{
    const int KPv = kp_of(k), g2 = coltile_grid(KPv, N2);
    const int g1 = (skip1 && !snap) ? 0 : coltile_grid(KPv, N1);         // nothing to do for side 1 without a snapshot
    const int k2 = (k + 1) / 2 * 2;
    COLTILE_LAUNCH(grad_pg2_snap_kernel, g1 + g2, X1, N1, R1, G1, part1, g1, X2, N2, R2, G2, part2, k, snap, k2, skip1);
    SMK_HIP(hipGetLastError());
    sum_partials2_host_kernel<<<1, 256, 0, st>>>(part1, g1, part2, g2, pg_accum, host_out, flag, flag_slot, host_tag);
    SMK_HIP(hipGetLastError());
    return 0;
}

int launch_pg_from_grad(const double* X, const double* Y, int k, i64 N, double* pg_partials, double* pg_accum,
                        int slot, hipStream_t st)
{
    const i64 count = N * kp_of(k);
    int grid = (int)((count + 255) / 256 < 1024 ? (count + 255) / 256 : 1024);
    if (grid < 1) grid = 1;
    pg_from_grad_kernel<<<grid, 256, 0, st>>>(X, Y, count, pg_partials);
    SMK_HIP(hipGetLastError());
    sum_partials_kernel<<<1, 256, 0, st>>>(pg_partials, grid, pg_accum + slot);
    SMK_HIP(hipGetLastError());
    return 0;
}

// ==========================================================================
// HALS W update (nmf_solver_hals.hpp:66-117) on Wt (KP x M): one launch per
// column c (k sequential grid-wide reductions are inherent: column c's L2 norm
// feeds every later column; a dependent kernel boundary, ~1.5 us, is the
// cheapest grid-wide sync on this chip).  Kernel c first applies the pending
// normalisation of column c-1, then updates column c un-normalised and emits
// per-block partial sums of squares / zero counts.
//   scratch: ss[k][nblk], nz[k][nblk]
// ==========================================================================
// two block-wide sums at once (1024-thread blocks); results broadcast to every thread
__device__ __forceinline__ void block_sum2_bcast(double& a, double& b, double* sh /* >= 34 doubles */)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_down(a, off, 64);
        b += __shfl_down(b, off, 64);
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nw = (blockDim.x + 63) >> 6;
    if (lane == 0) { sh[w] = a; sh[16 + w] = b; }
    __syncthreads();
    double ta = 0.0, tb = 0.0;
    for (int i = 0; i < nw; ++i) { ta += sh[i]; tb += sh[16 + i]; }     // same order in every thread
    __syncthreads();
    a = ta;
    b = tb;
}

template <int KP>
__global__ __launch_bounds__(1024) void hals_w_col_kernel(double* __restrict__ Wt, int k, i64 M, PartialView R,
                                                          const double* __restrict__ G, int c, int nblk,
                                                          double* __restrict__ ss, double* __restrict__ nz)
{
    constexpr int LPC = KP / 4;
    constexpr int RPB = 1024 / LPC;                // rows per block pass
    __shared__ double sh[34];
    const int s = threadIdx.x % LPC;
    const int own_c = (c < k && s == c / 4) ? (c % 4) : -1;            // which of my 4 slots is column c
    const int own_p = (c > 0 && s == (c - 1) / 4) ? ((c - 1) % 4) : -1;
    const i64 rows_per_block = (M + nblk - 1) / nblk;
    const i64 r_begin = (i64)blockIdx.x * rows_per_block;
    i64 r_end = r_begin + rows_per_block;
    if (r_end > M) r_end = M;

    // ---- issue every load of the first pass before touching the previous column's norm
    i64 i = r_begin + threadIdx.x / LPC;
    bool valid = i < r_end;
    i64 ic = valid ? i : (r_end > r_begin ? r_end - 1 : 0);
    double w[4], gc[4] = {0, 0, 0, 0};
    double gcc = 1.0, rhs = 0.0;
    load4(Wt + ic * KP + 4 * s, w);
    if (c < k) {
        load4(G + (i64)c * KP + 4 * s, gc);         // HHt(4s.., c) (symmetric)
        gcc = G[(i64)c * KP + c];
        if (own_c >= 0) rhs = rhs_elem(R, ic, c);
    }

    // ---- norm of the previous column from the per-block partials of the previous launch
    double scale_prev = 1.0, fill_prev = -1.0;
    if (c > 0) {
        double s2 = 0.0, zc = 0.0;
        for (int t = threadIdx.x; t < nblk; t += blockDim.x) {
            s2 += ss[(i64)(c - 1) * nblk + t];
            zc += nz[(i64)(c - 1) * nblk + t];
        }
        block_sum2_bcast(s2, zc, sh);
        if (zc >= (double)M) {                      // all-zero column guard (:105-111)
            const double eps = DBL_EPSILON;
            const double nrm = sqrt((double)M * eps * eps);
            fill_prev = eps * (1.0 / nrm);
        } else {
            scale_prev = 1.0 / sqrt(s2);
        }
    }

    double v2 = 0.0, zero = 0.0;
    for (i64 i0 = r_begin; i0 < r_end; i0 += RPB) {
        if (i0 != r_begin) {                        // later passes (M > nblk * RPB)
            i = i0 + threadIdx.x / LPC;
            valid = i < r_end;
            ic = valid ? i : (r_end - 1);
            load4(Wt + ic * KP + 4 * s, w);
            if (own_c >= 0) rhs = rhs_elem(R, ic, c);
        }
        double* pw = Wt + ic * KP + 4 * s;
        if (own_p >= 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (e == own_p) {
                    w[e] = (fill_prev >= 0.0) ? fill_prev : w[e] * scale_prev;
                    if (valid) pw[e] = w[e];
                }
        }
        if (c < k) {
            const double dot = group_sum<LPC>((gc[0] * w[0] + gc[1] * w[1]) + (gc[2] * w[2] + gc[3] * w[3]));
            if (own_c >= 0 && valid) {
                double wc = 0.0;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (e == own_c) wc = w[e];
                double v = wc + (rhs - dot) / gcc;
                if (isnan(v) || v < 0.0) { v = 0.0; zero += 1.0; }
                pw[own_c] = v;
                v2 += v * v;
            }
        }
    }
    if (c < k) {
        block_sum2_bcast(v2, zero, sh);
        if (threadIdx.x == 0) {
            ss[(i64)c * nblk + blockIdx.x] = v2;
            nz[(i64)c * nblk + blockIdx.x] = zero;
        }
    }
}

static inline int hals_w_blocks(int KP, i64 M)
{
    const i64 rpb = 1024 / (KP / 4);
    i64 nblk = (M + rpb - 1) / rpb;
    if (nblk > 1024) nblk = 1024;
    if (nblk < 1) nblk = 1;
    return (int)nblk;
}

// --------------------------------------------------------------------------
// Fused HALS W update: ONE persistent launch, every row of W lives in the
// registers of one thread for the whole sweep; the k column norms are exchanged
// between workgroups through self-validating 8-byte granules (the partial sum
// of squares IS the flag: slots are pre-set to an all-ones NaN pattern, a
// relaxed agent-scope (sc1) store publishes, relaxed agent-scope loads poll).
// Every workgroup sums the same slots in the same order, so all of them derive
// bit-identical norms.  At most one workgroup per CU (grid <= CU count): all
// resident by construction; every spin is bounded and reports through
// fail_flag instead of hanging.
// --------------------------------------------------------------------------
constexpr unsigned long long kSlotEmpty = ~0ull;

// one column step of the fused sweep; C is a compile-time column index so that w[] stays in VGPRs
template <int KP, int NT, int C>
__device__ __forceinline__ void hals_w_fused_step(double (&w)[KP], double& rhs, bool& dead, const double* gs,
                                                  double* sh, int k, i64 M, i64 row, bool valid,
                                                  const PartialView& R, unsigned long long* __restrict__ slots,
                                                  int nblk, int lane, int wave, unsigned spin_max,
                                                  unsigned long long* __restrict__ gslots)
{
    constexpr int NW = NT / 64;
    if (C >= k || dead) return;                     // uniform
    const double gcc = gs[C * KP + C];
    double d0 = 0.0, d1 = 0.0;
#pragma unroll
    for (int j = 0; j < KP; j += 2) {
        d0 += w[j] * gs[C * KP + j];
        d1 += w[j + 1] * gs[C * KP + j + 1];
    }
    double v = w[C] + (rhs - (d0 + d1)) / gcc;
    if (isnan(v) || v < 0.0) v = 0.0;
    if (!valid) v = 0.0;
    double v2 = v * v;
    // prefetch next column's right-hand side while the norm is being exchanged
    const double rhs_next = (C + 1 < k && valid) ? rhs_elem(R, row, C + 1) : 0.0;

    // block partial -> slot.  sh[] is double buffered by column parity (one barrier less).
    double* shc = sh + (C & 1) * 20;
    v2 = wave_sum(v2);
    if (lane == 0) shc[wave] = v2;
    __syncthreads();
    if (wave == 0) {
        double t = (lane < NW) ? shc[lane] : 0.0;
        t = wave_sum(t);
        unsigned long long* col_slots = slots + (i64)C * nblk;
        if (lane == 0)
            __hip_atomic_store(col_slots + blockIdx.x, (unsigned long long)__double_as_longlong(t),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        double acc = 0.0;
        bool ok = true;
        if (gslots) {
            // Two-level exchange (round 4, SMK_HALS_EXCHANGE=2; NOT the default: it measured slower, see launch_hals_w_update).  Every
            // workgroup polling every slot is an all-to-all of nblk x nblk loads per column (3.9 us per column at 256 workgroups: the
            // guide's price for a broadcast + fan-in).  Here a workgroup polls only the
            // slots of its GROUP (workgroups b with b % 8 == its own: 32 of 256 -- on this chip block b is observed to run on
            // XCD b % 8, so these polls stay inside one XCD's L2; nothing depends on that, any placement gives the same bits),
            // the group's first workgroup publishes the group sum, and everybody polls the 8 group sums.
            const int grp = (int)(blockIdx.x & 7u), ngrp = nblk < 8 ? nblk : 8;
            const int members = (nblk - grp + 7) / 8;                       // <= 64 for nblk <= 512
            unsigned long long bits = lane < members ? kSlotEmpty : 0ull;
            for (unsigned spin = 0; spin < spin_max; ++spin) {
                if (bits == kSlotEmpty) bits = __hip_atomic_load(col_slots + grp + 8 * lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (!__any(bits == kSlotEmpty)) break;
            }
            if (bits == kSlotEmpty) { ok = false; bits = 0ull; }
            const double gsum = wave_sum(__longlong_as_double((long long)bits));
            unsigned long long* col_g = gslots + (i64)C * 8;
            if ((int)blockIdx.x < ngrp && lane == 0)                        // the group's first workgroup publishes its sum
                __hip_atomic_store(col_g + grp, (unsigned long long)__double_as_longlong(gsum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned long long gb = lane < ngrp ? kSlotEmpty : 0ull;
            for (unsigned spin = 0; spin < spin_max; ++spin) {
                if (gb == kSlotEmpty) gb = __hip_atomic_load(col_g + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (!__any(gb == kSlotEmpty)) break;
            }
            if (gb == kSlotEmpty) { ok = false; gb = 0ull; }
            acc = __longlong_as_double((long long)gb);
        } else
        // gather every workgroup's partial: up to 4 slots per lane polled together (bounded spin)
        for (int b0 = 0; b0 < nblk; b0 += 256) {
            unsigned long long bits[4];
            bool have[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { have[u] = (b0 + u * 64 + lane) < nblk; bits[u] = have[u] ? kSlotEmpty : 0ull; }
            for (unsigned spin = 0; spin < spin_max; ++spin) {
                bool pending = false;
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (have[u] && bits[u] == kSlotEmpty) {
                        bits[u] = __hip_atomic_load(col_slots + b0 + u * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        pending |= (bits[u] == kSlotEmpty);
                    }
                if (!__any(pending)) break;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (have[u] && bits[u] == kSlotEmpty) { ok = false; bits[u] = 0ull; }
                acc += __longlong_as_double((long long)bits[u]);
            }
        }
        acc = wave_sum(acc);
        const bool all_ok = __all(ok);
        if (lane == 0) { shc[16] = acc; shc[17] = all_ok ? 0.0 : 1.0; }
    }
    __syncthreads();
    const double nu2 = shc[16];
    if (shc[17] != 0.0) dead = true;
    if (nu2 == 0.0) {                               // whole column clamped to zero (:105-111)
        const double eps = DBL_EPSILON;
        v = eps * (1.0 / sqrt((double)M * eps * eps));
    } else {
        v = v * (1.0 / sqrt(nu2));
    }
    w[C] = v;
    rhs = rhs_next;
}

template <int KP, int NT, int... Cs>
__device__ __forceinline__ void hals_w_fused_all(std::integer_sequence<int, Cs...>, double (&w)[KP], double& rhs,
                                                 bool& dead, const double* gs, double* sh, int k, i64 M, i64 row,
                                                 bool valid, const PartialView& R,
                                                 unsigned long long* __restrict__ slots, int nblk, int lane, int wave,
                                                 unsigned spin_max, unsigned long long* __restrict__ gslots)
{
    (hals_w_fused_step<KP, NT, Cs>(w, rhs, dead, gs, sh, k, M, row, valid, R, slots, nblk, lane, wave, spin_max, gslots), ...);
}

template <int KP, int NT>
__global__ __launch_bounds__(NT) void hals_w_fused_kernel(double* __restrict__ Wt, int k, i64 M, PartialView R,
                                                          const double* __restrict__ G,
                                                          unsigned long long* __restrict__ slots,
                                                          unsigned long long* __restrict__ slots_other, int nblk,
                                                          int* __restrict__ fail_flag, unsigned spin_max, int two_level,
                                                          unsigned char* __restrict__ pack_out, double* __restrict__ Gp_out,
                                                          int KT, i64 nq, int nsplit)
{
    __shared__ __attribute__((aligned(16))) double gs[KP * KP];
    extern __shared__ double ep_tile[];             // NT x (KP + 1) doubles, only when pack_out != nullptr (tile_pack_gram)
    __shared__ double sh[40];
    for (int t = threadIdx.x; t < KP * KP; t += NT) gs[t] = G[t];
    // the other slot buffer (used by the previous sweep, which is complete) is re-armed for the next one: no memset
    // launch per iteration
    if ((int)threadIdx.x < k) slots_other[(i64)threadIdx.x * nblk + blockIdx.x] = kSlotEmpty;
    // the group sums of the two-level exchange live behind the k x 1024 slots of each buffer; workgroup 0 re-arms the other buffer's
    unsigned long long* gslots = two_level ? slots + (i64)k * 1024 : nullptr;
    if (two_level && blockIdx.x == 0 && (int)threadIdx.x < k * 8) (slots_other + (i64)k * 1024)[threadIdx.x] = kSlotEmpty;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const i64 row = (i64)blockIdx.x * NT + threadIdx.x;
    const bool valid = row < M;
    double w[KP];
    {
        const double* p = Wt + (valid ? row : 0) * KP;
#pragma unroll
        for (int j = 0; j < KP; j += 2) {
            const f64x2_t v = *(const f64x2_t*)(p + j);
            w[j] = valid ? v[0] : 0.0;
            w[j + 1] = valid ? v[1] : 0.0;
        }
    }
    double rhs = valid ? rhs_elem(R, row, 0) : 0.0;
    __syncthreads();

    bool dead = false;
    hals_w_fused_all<KP, NT>(std::make_integer_sequence<int, KP>{}, w, rhs, dead, gs, sh, k, M, row, valid, R, slots,
                             nblk, lane, wave, spin_max, gslots);
    if (dead) {
        if (threadIdx.x == 0) atomicMin(fail_flag, -3);
        return;
    }
    if (valid) {
        double* p = Wt + row * KP;
#pragma unroll
        for (int j = 0; j < KP; j += 2) {
            f64x2_t v;
            v[0] = w[j];
            v[1] = w[j + 1];
            *(f64x2_t*)(p + j) = v;
        }
    }
    if constexpr (KP == 16 || KP == 32) {
        if (pack_out) {                             // uniform: the packed operand and the Gram partial of this workgroup's NT rows
            double* t = ep_tile + threadIdx.x * (KP + 1);
#pragma unroll
            for (int j = 0; j < KP; ++j) t[j] = valid ? w[j] : 0.0;
            __syncthreads();                        // (also: gs is free now and serves as the reduction buffer)
            tile_pack_gram<KP>(ep_tile, NT, (i64)blockIdx.x * NT, M, k, KT, nsplit, nq, blockIdx.x == gridDim.x - 1, pack_out,
                               Gp_out + (i64)blockIdx.x * KP * KP, gs);
        }
    }
}

static inline bool hals_w_use_blocked(int k)
{
    static const bool on = [] { const char* e = getenv("SMK_HALS_W_BLOCKED"); return !(e && e[0] == '0'); }();
    return on && k > 64;
}

size_t hals_w_scratch_elems(int k, i64 M)
{
    if (is_wide(k)) return std::max((size_t)(2 * (i64)k * hals_w_wide_blocks(M)), hals_w_blocked_scratch_elems(k, M));
    const size_t multi = (size_t)(2 * (i64)k * hals_w_blocks(kp_of(k), M));
    const size_t fused = (size_t)2 * k * (1024 + 8);      // two slot buffers (8 bytes per slot), generous, each followed by its k x 8 group sums
    const size_t blocked = k > 64 ? hals_w_blocked_scratch_elems(k, M) : 0;
    return std::max(std::max(multi, fused), blocked);
}

// `parity` alternates between the two slot buffers of the fused sweep (both must be all-ones before the first call:
// hals_w_scratch_init); force_multi: the one-launch-per-column path (also SMK_HALS_W=multi).
int hals_w_scratch_init(double* scratch, int k, i64 M, hipStream_t st)
{
    SMK_HIP(hipMemsetAsync(scratch, 0xFF, hals_w_scratch_elems(k, M) * sizeof(double), st));
    return 0;
}

int launch_hals_w_update(double* Wt, int k, i64 M, PartialView R, const double* G, double* scratch, int num_cus,
                         int* fail_flag, int parity, int force_multi, hipStream_t st, HalsEpilogue* ep)
{
    if (ep) ep->done = false;
    if (hals_w_use_blocked(k)) return launch_hals_w_update_blocked(Wt, k, M, R, G, scratch, st);
    if (is_wide(k)) return launch_hals_w_update_wide(Wt, k, M, R, G, scratch, st);
    const int KPv = kp_of(k);
    static int mode = -1;                            // SMK_HALS_W=multi forces the one-launch-per-column path
    static unsigned spin_max = 1u << 22;             // SMK_HALS_SPIN=<n>: bound of the exchange polls (tests)
    if (mode < 0) {
        const char* env = getenv("SMK_HALS_W");
        mode = (env && env[0] == 'm') ? 0 : 1;
        const char* sp = getenv("SMK_HALS_SPIN");
        if (sp && atoi(sp) > 0) spin_max = (unsigned)atoi(sp);
    }
    // fused path: at most one workgroup per CU so that all of them are resident by construction.
    // Smallest workgroup (256 threads: cheapest in-block sync, measured best) that still covers M
    // rows with <= num_cus workgroups; the register budget caps it at 512 for KP = 32 and 256 for 64.
    int nt = 0;
    const int nt_max = (KPv == 64) ? 256 : 1024;
    static const int nt_min = [] { const char* e = getenv("SMK_HALS_NT"); return e ? atoi(e) : 256; }();
    for (int cand = 256; cand <= nt_max; cand *= 2)
        if (cand >= nt_min && (M + cand - 1) / cand <= (i64)num_cus) { nt = cand; break; }
    if (mode == 1 && nt != 0 && !force_multi && KPv <= 64) {
        const i64 nblk_f = (M + nt - 1) / nt;
        unsigned long long* slots = (unsigned long long*)scratch + (size_t)(parity & 1) * k * (1024 + 8);
        unsigned long long* other = (unsigned long long*)scratch + (size_t)((parity & 1) ^ 1) * k * (1024 + 8);
        const int nb = (int)nblk_f;
        // Default: every workgroup polls every slot.  SMK_HALS_EXCHANGE=2 selects the two-level exchange (group sums first): built in
        // round 4 on the expectation that 40 polled slots instead of 256 would cut the 3.9 us per column -- measured on C3 it is
        // SLOWER, 155 us per sweep against 125 (4.8 us per column): the second dependent store -> poll hop costs more than the
        // all-to-all's contention.  Kept selectable as the record of that measurement (profiles/r04_hals_exchange_two_level.txt).
        static const int mode = [] { const char* e = getenv("SMK_HALS_EXCHANGE"); return e ? atoi(e) : 1; }();
        const int two_level = (mode == 2 && nb >= 16 && nb <= 512 && k * 8 <= nt) ? 1 : 0;
        // the epilogue (packed operand + Gram partial per workgroup) rides along at KP = 16 / 32 with 256-thread workgroups
        const bool with_ep = ep && ep->pack_out && ep->Gp && (KPv == 16 || KPv == 32) && nt == 256 && nb <= ep->max_blocks;
        const int ep_lds = with_ep ? 256 * (KPv + 1) * (int)sizeof(double) : 0;
        unsigned char* ep_out = with_ep ? ep->pack_out : nullptr;
        double* ep_gp = with_ep ? ep->Gp : nullptr;
        const int ep_kt = with_ep ? ep->KT : 1, ep_ns = with_ep ? ep->nsplit : 3;
        const i64 ep_nq = with_ep ? ep->nq : 0;
        if (with_ep) {
            if (KPv == 16) SMK_HIP(hipFuncSetAttribute((const void*)hals_w_fused_kernel<16, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, ep_lds));
            else SMK_HIP(hipFuncSetAttribute((const void*)hals_w_fused_kernel<32, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, ep_lds));
            ep->nblk = nb;
            ep->done = true;
        }
#define SMK_FUSED(KPX, NTX) hals_w_fused_kernel<KPX, NTX><<<nb, NTX, (NTX == 256 ? ep_lds : 0), st>>>(Wt, k, M, R, G, slots, other, nb, fail_flag, spin_max, two_level, (NTX == 256 ? ep_out : nullptr), ep_gp, ep_kt, ep_nq, ep_ns)
        switch (KPv) {
            case 8: if (nt == 256) SMK_FUSED(8, 256); else if (nt == 512) SMK_FUSED(8, 512); else SMK_FUSED(8, 1024); break;
            case 16: if (nt == 256) SMK_FUSED(16, 256); else if (nt == 512) SMK_FUSED(16, 512); else SMK_FUSED(16, 1024); break;
            case 32: if (nt == 256) SMK_FUSED(32, 256); else if (nt == 512) SMK_FUSED(32, 512); else SMK_FUSED(32, 1024); break;
            default: SMK_FUSED(64, 256); break;
        }
#undef SMK_FUSED
        SMK_HIP(hipGetLastError());
        return 0;
    }
    const int nblk = hals_w_blocks(KPv, M);
    double* ss = scratch;
    double* nz = scratch + (i64)k * nblk;
    for (int c = 0; c <= k; ++c) {
        KP_DISPATCH128(KPv, (hals_w_col_kernel<KP><<<nblk, 1024, 0, st>>>(Wt, k, M, R, G, c, nblk, ss, nz)));
    }
    SMK_HIP(hipGetLastError());
    return 0;
}

// ==========================================================================
// Sparse A (CSC, fp64 values): the two big products become gathers
//   out[:, j] = sum_{p in column j of B} val[p] * X[:, row[p]]
// with B = A (X = W', out = W'A) or B = A' (X = H, out = (AH')'), i.e. the reference's
// sparse Gemm variants (sparse_gemm_ab_impl.hpp / sparse_gemm_ba_impl.hpp) in gather form.
// KP/4 lanes per output column, 4 fp64 values (32 B) of X per lane per nonzero.
// ==========================================================================
template <int KP>
__global__ __launch_bounds__(256) void spmm_gather_kernel(const i64* __restrict__ colptr,
                                                          const unsigned* __restrict__ rowidx,
                                                          const double* __restrict__ val, i64 ncols,
                                                          const double* __restrict__ X, double* __restrict__ P, int kpp,
                                                          InvRide ride)
{
    constexpr int LPC = KP / 4;
    i64 blk = blockIdx.x;
    if constexpr (KP == 32 || KP == 64) {
        // workgroup 0 of a launch that carries the Gram inverse (common.h: InvRide) inverts; the product starts at workgroup 1
        if (ride.G) {
            if (blk == 0) {
                __shared__ __attribute__((aligned(16))) double inv_lds[GRAM_INVERSE_LDS(KP)];
                gram_inverse64_body<KP>(ride.G, ride.k, ride.Ginv, (int*)(ride.Ginv + KP * KP), inv_lds);
                return;
            }
            --blk;
        }
    }
    const i64 gtid = blk * blockDim.x + threadIdx.x;
    const i64 j = gtid / LPC;
    const int s = (int)(gtid % LPC);
    if (j >= ncols) return;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    const i64 p0 = colptr[j], p1 = colptr[j + 1];
    for (i64 p = p0; p < p1; ++p) {
        const double v = val[p];
        double x[4];
        load4(X + (i64)rowidx[p] * KP + 4 * s, x);
        acc[0] += v * x[0];
        acc[1] += v * x[1];
        acc[2] += v * x[2];
        acc[3] += v * x[3];
    }
    if (4 * s < kpp) store4(P + j * kpp + 4 * s, acc);
}

// k <= 2 (the RANK2 / HierNMF2 hot loop): LPC lanes per output column.  Lane l takes the stored entries p0 + l,
// p0 + l + LPC, ... of its column, so the LPC lanes read consecutive values / row indices (and consecutive columns are
// consecutive in memory: the wave streams them in full lines), every lane has all of its gathers in flight at once, and
// the 2 x LPC partial sums are joined by DPP moves.  One lane per column (round 2) serialised ~16 dependent
// gathers per lane and touched a different line of val / rowidx in every lane.
// `ldx` = row pitch of X in doubles (KP, or 2 for the compact copy the RANK2 kernels keep).
template <int LPC>
__global__ __launch_bounds__(256) void spmm_gather2_kernel(const i64* __restrict__ colptr,
                                                           const unsigned* __restrict__ rowidx,
                                                           const double* __restrict__ val, i64 ncols,
                                                           const double* __restrict__ X, int ldx,
                                                           double* __restrict__ P, int kpp)
{
    const i64 gtid = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const i64 j = gtid / LPC;
    const int l = (int)(gtid % LPC);
    const bool valid = j < ncols;
    double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
    if (valid) {
        const i64 p0 = colptr[j], p1 = colptr[j + 1];
        i64 p = p0 + l;
        for (; p + LPC < p1; p += 2 * LPC) {
            const double v0 = val[p], v1 = val[p + LPC];
            const f64x2_t x0 = *(const f64x2_t*)(X + (i64)rowidx[p] * ldx);
            const f64x2_t x1 = *(const f64x2_t*)(X + (i64)rowidx[p + LPC] * ldx);
            a0 += v0 * x0[0]; a1 += v0 * x0[1];
            b0 += v1 * x1[0]; b1 += v1 * x1[1];
        }
        if (p < p1) {
            const double v0 = val[p];
            const f64x2_t x0 = *(const f64x2_t*)(X + (i64)rowidx[p] * ldx);
            a0 += v0 * x0[0]; a1 += v0 * x0[1];
        }
    }
    const double r0 = group_sum<LPC>(a0 + b0), r1 = group_sum<LPC>(a1 + b1);      // whole waves take part in the DPP moves
    if (!valid || l != 0) return;
    double* out = P + j * kpp;
    f64x2_t r;
    r[0] = r0;
    r[1] = r1;
    *(f64x2_t*)out = r;
    for (int e = 2; e < kpp && e < 8; e += 2) { f64x2_t z; z[0] = 0.0; z[1] = 0.0; *(f64x2_t*)(out + e) = z; }
}

int launch_spmm_gather(const i64* colptr, const unsigned* rowidx, const double* val, i64 ncols, i64 nnz_hint, const double* X,
                       int ldx, int k, double* P, int kpp, hipStream_t st, const InvRide* ride_in)
{
    if (is_wide(k)) return launch_spmm_gather_wide(colptr, rowidx, val, ncols, X, k, P, kpp, st);
    const int KPv = kp_of(k);
    if (k <= 2 && (ldx == 2 || ldx == KPv)) {
        if (ncols <= 0) return 0;
        // lanes per column by the average column length (nnz_hint <= 0: unknown -> 8)
        static const int forced = [] { const char* e = getenv("SMK_SPMM2_LPC"); return e ? atoi(e) : 0; }();
        const double avg = nnz_hint > 0 ? (double)nnz_hint / (double)ncols : 12.0;
        int lpc = forced ? forced : (avg <= 3.0 ? 2 : avg <= 6.0 ? 4 : avg <= 12.0 ? 8 : 16);
        const i64 threads = ncols * lpc;
        const unsigned grid2 = (unsigned)((threads + 255) / 256);
        switch (lpc) {
            case 1: spmm_gather2_kernel<1><<<grid2, 256, 0, st>>>(colptr, rowidx, val, ncols, X, ldx, P, kpp); break;
            case 2: spmm_gather2_kernel<2><<<grid2, 256, 0, st>>>(colptr, rowidx, val, ncols, X, ldx, P, kpp); break;
            case 4: spmm_gather2_kernel<4><<<grid2, 256, 0, st>>>(colptr, rowidx, val, ncols, X, ldx, P, kpp); break;
            case 16: spmm_gather2_kernel<16><<<grid2, 256, 0, st>>>(colptr, rowidx, val, ncols, X, ldx, P, kpp); break;
            default: spmm_gather2_kernel<8><<<grid2, 256, 0, st>>>(colptr, rowidx, val, ncols, X, ldx, P, kpp); break;
        }
        SMK_HIP(hipGetLastError());
        return 0;
    }
    if (ldx != KPv) { set_error("spmm: unsupported row pitch of the gathered factor"); return -100; }
    int grid = (int)((ncols * (KPv / 4) + 255) / 256);
    if (grid == 0) return 0;
    InvRide ride;
    if (ride_in && ride_in->G && (KPv == 32 || KPv == 64)) { ride = *ride_in; ++grid; }
    KP_DISPATCH128(KPv, (spmm_gather_kernel<KP><<<grid, 256, 0, st>>>(colptr, rowidx, val, ncols, X, P, kpp, ride)));
    SMK_HIP(hipGetLastError());
    return ride.G ? 1 : 0;                   // 1: the launch carried the Gram inverse
}

// ==========================================================================
// NormalizeAndScale (normalize.hpp:118-140): nu_c = ||W(:,c)||_2 = sqrt(WtW[c][c]);
// Wt row c /= nu_c (invert=1), H row c *= nu_c (invert=0).
// ==========================================================================
__global__ __launch_bounds__(256) void scale_rows_kernel(double* __restrict__ X, int k, i64 N,
                                                         const double* __restrict__ G, int KP, int invert,
                                                         int* __restrict__ fail_flag)
{
    const i64 total = N * KP;
    for (i64 idx = (i64)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (i64)gridDim.x * blockDim.x) {
        const int r = (int)(idx % KP);
        if (r >= k) continue;
        const double nu = sqrt(G[r * KP + r]);
        if (fabs(nu) < DBL_EPSILON) {               // reference throws (normalize.hpp:41-42)
            if (invert) atomicMin(fail_flag, -2);
            continue;
        }
        X[idx] = invert ? X[idx] * (1.0 / nu) : X[idx] * nu;
    }
}

int launch_scale_rows(double* X, int k, i64 N, const double* G, int invert, int* fail_flag, hipStream_t st)
{
    const i64 total = N * kp_of(k);
    int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    if (grid < 1) grid = 1;
    scale_rows_kernel<<<grid, 256, 0, st>>>(X, k, N, G, kp_of(k), invert, fail_flag);
    SMK_HIP(hipGetLastError());
    return 0;
}

// ==========================================================================
// DELTA_FNORM progress (progress_estimator_generic.hpp:58-69)
// ==========================================================================
__global__ __launch_bounds__(256) void delta_fnorm_kernel(const double* __restrict__ W, double* __restrict__ Wprev,
                                                          i64 count, double* __restrict__ partials)
{
    __shared__ double sh[16];
    double d2 = 0.0, w2 = 0.0;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (i64)gridDim.x * blockDim.x) {
        const double w = W[i];
        const double d = Wprev[i] - w;
        d2 += d * d;
        w2 += w * w;
        Wprev[i] = w;
    }
    const double t1 = block_sum(d2, sh);
    const double t2 = block_sum(w2, sh);
    if (threadIdx.x == 0) {
        partials[blockIdx.x] = t1;
        partials[gridDim.x + blockIdx.x] = t2;
    }
}

int launch_delta_fnorm(const double* W, double* Wprev, i64 count, double* partials, double* out2, hipStream_t st)
{
    int grid = (int)((count + 255) / 256 < 512 ? (count + 255) / 256 : 512);
    if (grid < 1) grid = 1;
    delta_fnorm_kernel<<<grid, 256, 0, st>>>(W, Wprev, count, partials);
    SMK_HIP(hipGetLastError());
    sum_partials_kernel<<<1, 256, 0, st>>>(partials, grid, out2);
    sum_partials_kernel<<<1, 256, 0, st>>>(partials + grid, grid, out2 + 1);
    SMK_HIP(hipGetLastError());
    return 0;
}

}  // namespace smk
