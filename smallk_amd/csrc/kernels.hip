// smallk_amd/csrc/kernels.hip -- hand-written gfx950 (MI355X / CDNA4) kernels for the
// dense NMF inner loop.  Wave = 64 lanes everywhere.  No CUDA compatibility paths.
//
// Kernel map (reference call sites in parentheses, paths relative to /root/reference):
//   bigprod_kernel      W'A and H*At streaming products  (nmf_solver_{mu,hals,bpp}.hpp Gemm calls
//                       on A: mu :131,:143  hals :173,:187  bpp :354,:367) -- bf16/f32 MFMA, LDS
//                       staged by global_load_lds, 3-deep ring, counted vmcnt.
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
#include <cfloat>
#include <cstdlib>
#include <utility>

namespace smk {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(8))) float f32x8_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((ext_vector_type(2))) double f64x2_t;

#ifndef NT_AUX
#define NT_AUX 2
#endif
#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

// --------------------------------------------------------------------------
// synthetic data: counter based uniform [0,1) -- bit-identical to
// oracle/nmf_oracle.c:orc_uniform_value (matrixgen UNIFORM semantics)
// --------------------------------------------------------------------------
__host__ __device__ inline uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__host__ __device__ inline unsigned short f32_to_bf16_rne(float f)
{
    unsigned b = __builtin_bit_cast(unsigned, f);
    b += 0x7FFFu + ((b >> 16) & 1u);
    return (unsigned short)(b >> 16);
}

__host__ __device__ inline float bf16_bits_to_f32(unsigned short h)
{
    unsigned b = ((unsigned)h) << 16;
    return __builtin_bit_cast(float, b);
}

__host__ __device__ inline float uniform_value(uint64_t seed, uint64_t gidx, int quant)
{
    uint64_t h = mix64(seed * 0xD1342543DE82EF95ull + gidx);
    float f = (float)(h >> 40) * (1.0f / 16777216.0f);
    if (quant == 1) f = bf16_bits_to_f32(f32_to_bf16_rne(f));
    return f;
}

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

// ==========================================================================
// Streaming product  P[s](k x ncols) = X(k x len)[:, rows of split s] * B[rows of split s, :]
//
//   B  : len x ncols, column-major, bf16 or f32, the contraction runs down the
//        CONTIGUOUS dimension (pass 1: B = A, X = W';  pass 2: B = A', X = H).
//   X  : pre-packed MFMA A-operand fragments (pack_kernel), 1 KiB per
//        (chunk-pair q, split term s, k-tile kt), lane-linear.
//   Workgroup = 4 waves, tile = 128 columns (32 per wave) x MB=64 rows per stage.
//   Stages are staged into a 3-deep LDS ring by global_load_lds (16 B / lane,
//   full 128-B lines per column), one s_barrier per stage, counted vmcnt so two
//   stages stay in flight.  B chunks are XOR-swizzled on the SOURCE side so the
//   ds_read_b128 fragment reads are bank-conflict free.
//   MFMA: v_mfma_f32_32x32x16_bf16 (bf16) / v_mfma_f32_32x32x2_f32 (f32).
//   The contraction order inside a stage is permuted (lane half h takes chunk
//   2q+h) -- identical on both operands, so the result is the plain dot product.
//   HBM-bound: algorithmic bytes = len*ncols*sizeof(B elt) per launch.
// ==========================================================================
template <int EBYTES, int KT, int NSPLIT, int MB_, int NSTAGE_, int CW_, int WK_, int NWL_>
struct BPCfg {
    static constexpr int MB = MB_;          // rows per stage
    static constexpr int CW = CW_;          // 32-column MFMA tiles per wave
    static constexpr int WK = WK_;          // wave groups along k: waves = 4*WK, each owns KT/WK k-tiles
    static constexpr int KTW = KT / WK_;
    static constexpr int NWC = 4 * WK_;     // compute waves per workgroup
    // NWL > 0: wave specialisation -- NWL extra waves only issue the LDS-DMA loads (a vector-memory
    // instruction occupies its wave ~100 cycles; keeping it off the MFMA waves is worth more than the
    // registers the loader waves waste).  NWL = 0: every wave loads and computes.
    static constexpr int NWL = NWL_;
    static constexpr int NLD = NWL_ > 0 ? NWL_ : NWC;   // waves that issue loads
    static constexpr int NW = NWC + NWL_;   // waves per workgroup
    static constexpr int E = 16 / EBYTES;
    static constexpr int CPC = MB / E;      // 16-B chunks per column per stage
    // XOR swizzle of the chunk index so that 16 lanes reading 16 different columns hit 16 distinct
    // 16-byte slots of the 256-byte LDS bank row (column pitch = CPC*16 bytes)
    static constexpr int SWZ_SH = (CPC >= 16) ? 0 : (CPC == 8) ? 1 : (CPC == 4) ? 2 : 3;
    static constexpr int SWZ_MASK = (CPC >= 16 ? 16 : CPC) - 1;
    // fp32 B with a 3-term X operand = "bf16x3" emulation: the fp32 tile is split into bf16
    // hi/mid/lo in registers and multiplied on the bf16 MFMA (6 products of significance <= 2^-16),
    // 2.7x less matrix-core time than v_mfma_f32_32x32x2_f32 and a 16x shorter rounding chain.
    static constexpr bool EMU = (EBYTES == 4 && NSPLIT == 3);
    static constexpr int QS = (EBYTES == 2 || EMU) ? MB / 16 : CPC / 2;   // MFMA steps per stage
    static constexpr int NB = 128 * CW;     // columns per workgroup
    static constexpr int B_BYTES = NB * MB * EBYTES;
    static constexpr int X_BYTES = QS * NSPLIT * KT * 1024;
    static constexpr int STAGE_BYTES = B_BYTES + X_BYTES;
    static constexpr int NSTAGE = NSTAGE_;
    static constexpr int PD = NSTAGE_ - 1;          // stages in flight ahead of the consumer
    static constexpr int TI = STAGE_BYTES / 1024;   // wave-level 1-KiB loads per stage
    static constexpr int LPS = TI / NLD;            // per loading wave
    static_assert(TI % NLD == 0, "loads per stage must split evenly over the loading waves");
    static_assert(KT % WK_ == 0, "k tiles must split evenly over the wave groups");
    static_assert(LPS * PD <= 63, "vmcnt is a 6-bit counter");
    static_assert(STAGE_BYTES * NSTAGE <= 160 * 1024, "LDS ring exceeds 160 KiB");
};

#ifdef SMK_BP_PROFILE
__device__ unsigned long long* g_bp_prof = nullptr;   // [wg][wave][4] cycles: wait, barrier, issue, compute
#define BP_T(x) const unsigned long long x = __builtin_readcyclecounter()
#else
#define BP_T(x)
#endif

template <int N> __device__ __forceinline__ void wait_vmcnt()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// wait until at most `ahead` younger stages (LPS loads each) are still in flight
template <int LPS, int PD> __device__ __forceinline__ void wait_stage(int ahead)
{
    if constexpr (PD >= 4) { if (ahead >= 4) { wait_vmcnt<4 * LPS>(); return; } }
    if constexpr (PD >= 3) { if (ahead == 3) { wait_vmcnt<3 * LPS>(); return; } }
    if constexpr (PD >= 2) { if (ahead == 2) { wait_vmcnt<2 * LPS>(); return; } }
    if (ahead == 1) { wait_vmcnt<LPS>(); return; }
    wait_vmcnt<0>();
}

template <int EBYTES, int KT, int NSPLIT, int MB, int NSTAGE, int CW, int WK, int NWL>
__global__ __launch_bounds__(64 * (4 * WK + NWL), 1) void bigprod_kernel(const unsigned char* __restrict__ B, i64 ldb_bytes,
                                                         const unsigned char* __restrict__ Xp,
                                                         double* __restrict__ P, i64 stages, i64 nst,
                                                         i64 tiles, i64 ncols_pad, int S, int logS)
{
    using C = BPCfg<EBYTES, KT, NSPLIT, MB, NSTAGE, CW, WK, NWL>;
    constexpr int KTW = C::KTW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    // ---- XCD-aware block -> (tile, split): all blocks of one split share an XCD's L2
    // (block b is dispatched to XCD b % 8; used for speed only).
    const int bid = blockIdx.x;
    const int xcd = bid & 7;
    const i64 grp = bid >> 3;
    i64 tile;
    int split;
    if (S <= 8) {
        split = xcd & (S - 1);
        tile = grp * (8 >> logS) + (xcd >> logS);
    } else {
        const int sub = S >> 3;
        split = (int)(grp % sub) * 8 + xcd;
        tile = grp / sub;
    }
    if (tile >= tiles) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_loader = (NWL == 0) || (wave >= C::NWC);          // wave-uniform
    const bool is_compute = wave < C::NWC;
    const int lw = (NWL == 0) ? wave : (wave - C::NWC);             // index among the loading waves

    i64 st0 = (i64)split * nst;
    i64 st1 = st0 + nst;
    if (st1 > stages) st1 = stages;
    const int my_nst = (st1 > st0) ? (int)(st1 - st0) : 0;

    // per-lane source offsets for this wave's loads (constant across stages)
    const i64 col0 = tile * C::NB;
    i64 src_off[C::LPS];
    int is_b[C::LPS];
#pragma unroll
    for (int i = 0; i < C::LPS; ++i) {
        const int t = lw + C::NLD * i;             // wave-level load index within the stage
        if (t * 1024 < C::B_BYTES) {
            const int p = t * 64 + lane;           // chunk position inside the LDS B tile
            const int j = p / C::CPC;
            const int pc = p % C::CPC;
            const int swz = (j >> C::SWZ_SH) & C::SWZ_MASK;
            const int lc = pc ^ swz;
            src_off[i] = (col0 + j) * ldb_bytes + (i64)lc * 16;
            is_b[i] = 1;
        } else {
            src_off[i] = (i64)(t * 1024 - C::B_BYTES) + lane * 16;   // offset inside the X stage block
            is_b[i] = 0;
        }
    }

    auto issue = [&](int s_local) {
        const i64 stage = st0 + s_local;
        const int buf = s_local % C::NSTAGE;
        unsigned char* lbase = smem + buf * C::STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < C::LPS; ++i) {
            const int t = lw + C::NLD * i;
            const unsigned char* g = is_b[i] ? (B + src_off[i] + stage * (C::MB * EBYTES))
                                             : (Xp + stage * C::X_BYTES + src_off[i]);
            // B is streamed once: non-temporal policy (aux = 2) keeps it from displacing the X slice in L2
            if (is_b[i])
                __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)g, (LDS_AS void*)(lbase + t * 1024), 16, 0, NT_AUX);
            else
                __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)g, (LDS_AS void*)(lbase + t * 1024), 16, 0, 0);
        }
    };

    // Accumulators.  The leading (hi) term is accumulated in fp32 by the MFMA for ONE stage and
    // then added into fp64 running sums by the VALU while the next stage's MFMAs run into the
    // other fp32 set (accA/accB ping-pong): the fp32 rounding chain never exceeds one stage.
    // The mid/lo split terms are 2^-8 / 2^-16 smaller and stay in fp32 for the whole split.
    constexpr int NT = CW * KTW;                   // 32x32 output tiles per wave
    const int cwv = wave & 3;                      // column group of this wave
    const int kw = wave >> 2;                      // k-tile group of this wave
    constexpr int NS1 = (NSPLIT > 1) ? NSPLIT - 1 : 1;
    f32x16_t accA[NT], accB[NT];
    f32x16_t accs[NS1][NT];
    double dacc[NT][16];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            accA[n][r] = 0.f;
            accB[n][r] = 0.f;
            dacc[n][r] = 0.0;
#pragma unroll
            for (int s = 0; s < NS1; ++s) accs[s][n][r] = 0.f;
        }
    }

    // fragment read addresses: this wave owns columns [wave*32*CW, +32*CW) of the tile
    const int h = lane >> 5;
    int bfrag_base[CW], swz_r[CW];
#pragma unroll
    for (int c = 0; c < CW; ++c) {
        const int jl = (cwv * CW + c) * 32 + (lane & 31);
        swz_r[c] = (jl >> C::SWZ_SH) & C::SWZ_MASK;
        bfrag_base[c] = jl * C::CPC * 16;
    }

    auto flush = [&](f32x16_t (&a)[NT]) {
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                dacc[n][r] += (double)a[n][r];
                a[n][r] = 0.f;
            }
    };

#ifdef SMK_BP_PROFILE
    unsigned long long prof_[4] = {0, 0, 0, 0};
#endif
    // one stage: wait for its data, refill the ring slot freed by the previous stage, MFMAs into
    // `cur`; the previous stage's fp32 sums (`prev`) are folded into fp64 after the first step.
    auto stage_body = [&](int t, f32x16_t (&cur)[NT], f32x16_t (&prev)[NT], bool flush_prev) {
        int ahead = my_nst - 1 - t;
        if (ahead > C::PD - 1) ahead = C::PD - 1;
        BP_T(t0_);
        if (is_loader) wait_stage<C::LPS, C::PD>(ahead);
        BP_T(t1_);
        __builtin_amdgcn_s_barrier();
        BP_T(t2_);
        if (is_loader && t + C::PD < my_nst) issue(t + C::PD);
        BP_T(t3_);
        if (!is_compute) return;

        const unsigned char* sb = smem + (t % C::NSTAGE) * C::STAGE_BYTES;
        const unsigned char* sx = sb + C::B_BYTES;
        u32x4_t bq[2][CW];
        u32x4_t aq[2][NSPLIT][KTW];
#pragma unroll
        for (int q = 0; q < C::QS; ++q) {
            if constexpr (C::EMU) {
                // 16 rows per step: this lane half owns rows 16q + 8h .. +7 = fp32 chunks 4q+2h, 4q+2h+1
                bf16x8_t bhi[CW], bmid[CW], blo[CW];
#pragma unroll
                for (int c = 0; c < CW; ++c) {
                    const int lc0 = 4 * q + 2 * h;
                    const f32x4_t f0 = *(const f32x4_t*)(sb + bfrag_base[c] + (((lc0) ^ swz_r[c]) << 4));
                    const f32x4_t f1 = *(const f32x4_t*)(sb + bfrag_base[c] + (((lc0 + 1) ^ swz_r[c]) << 4));
                    f32x8_t x;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { x[e] = f0[e]; x[4 + e] = f1[e]; }
                    bhi[c] = __builtin_convertvector(x, bf16x8_t);
                    x -= __builtin_convertvector(bhi[c], f32x8_t);
                    bmid[c] = __builtin_convertvector(x, bf16x8_t);
                    x -= __builtin_convertvector(bmid[c], f32x8_t);
                    blo[c] = __builtin_convertvector(x, bf16x8_t);
                }
#pragma unroll
                for (int kt = 0; kt < KTW; ++kt) {
                    bf16x8_t a[3];
#pragma unroll
                    for (int s = 0; s < 3; ++s)
                        a[s] = __builtin_bit_cast(bf16x8_t, *(const u32x4_t*)(sx + ((q * 3 + s) * KT + kw * KTW + kt) * 1024 + lane * 16));
#pragma unroll
                    for (int c = 0; c < CW; ++c) {
                        const int n = c * KTW + kt;
                        cur[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bhi[c], cur[n], 0, 0, 0);
                        accs[0][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bmid[c], accs[0][n], 0, 0, 0);
                        accs[0][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], bhi[c], accs[0][n], 0, 0, 0);
                        accs[1][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], blo[c], accs[1][n], 0, 0, 0);
                        accs[1][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], bmid[c], accs[1][n], 0, 0, 0);
                        accs[1][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], bhi[c], accs[1][n], 0, 0, 0);
                    }
                }
            } else {
            const int lc = 2 * q + h;
            if constexpr (EBYTES == 2) {
                // software-pipelined fragment reads: the ds_reads of step q+1 are issued before the
                // MFMAs of step q so that LDS latency hides behind the matrix pipe
                if (q == 0) {
#pragma unroll
                    for (int c = 0; c < CW; ++c) bq[0][c] = *(const u32x4_t*)(sb + bfrag_base[c] + ((lc ^ swz_r[c]) << 4));
#pragma unroll
                    for (int s = 0; s < NSPLIT; ++s)
#pragma unroll
                        for (int kt = 0; kt < KTW; ++kt)
                            aq[0][s][kt] = *(const u32x4_t*)(sx + ((0 * NSPLIT + s) * KT + kw * KTW + kt) * 1024 + lane * 16);
                }
                if (q + 1 < C::QS) {
                    const int lcn = 2 * (q + 1) + h;
#pragma unroll
                    for (int c = 0; c < CW; ++c)
                        bq[(q + 1) & 1][c] = *(const u32x4_t*)(sb + bfrag_base[c] + ((lcn ^ swz_r[c]) << 4));
#pragma unroll
                    for (int s = 0; s < NSPLIT; ++s)
#pragma unroll
                        for (int kt = 0; kt < KTW; ++kt)
                            aq[(q + 1) & 1][s][kt] =
                                *(const u32x4_t*)(sx + (((q + 1) * NSPLIT + s) * KT + kw * KTW + kt) * 1024 + lane * 16);
                }
#pragma unroll
                for (int s = 0; s < NSPLIT; ++s)
#pragma unroll
                    for (int kt = 0; kt < KTW; ++kt) {
                        const bf16x8_t afr = __builtin_bit_cast(bf16x8_t, aq[q & 1][s][kt]);
#pragma unroll
                        for (int c = 0; c < CW; ++c) {
                            const bf16x8_t bfr = __builtin_bit_cast(bf16x8_t, bq[q & 1][c]);
                            const int n = c * KTW + kt;
                            if (s == 0) cur[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr, bfr, cur[n], 0, 0, 0);
                            else accs[s - 1][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr, bfr, accs[s - 1][n], 0, 0, 0);
                        }
                    }
            } else {
                u32x4_t braw[CW];
#pragma unroll
                for (int c = 0; c < CW; ++c) braw[c] = *(const u32x4_t*)(sb + bfrag_base[c] + ((lc ^ swz_r[c]) << 4));
#pragma unroll
                for (int kt = 0; kt < KTW; ++kt) {
                    const f32x4_t afr = *(const f32x4_t*)(sx + (q * KT + kw * KTW + kt) * 1024 + lane * 16);
#pragma unroll
                    for (int c = 0; c < CW; ++c) {
                        const f32x4_t bfr = __builtin_bit_cast(f32x4_t, braw[c]);
                        const int n = c * KTW + kt;
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            cur[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(afr[e], bfr[e], cur[n], 0, 0, 0);
                    }
                }
            }
            }
            if (q == 0 && flush_prev) flush(prev);
        }
#ifdef SMK_BP_PROFILE
        {
            asm volatile("s_nop 0" ::: "memory");
            const unsigned long long t4_ = __builtin_readcyclecounter();
            prof_[0] += t1_ - t0_; prof_[1] += t2_ - t1_; prof_[2] += t3_ - t2_; prof_[3] += t4_ - t3_;
        }
#endif
    };

    if (is_loader) {
#pragma unroll
        for (int i = 0; i < C::PD; ++i)
            if (i < my_nst) issue(i);
    }

    int t = 0;
    for (; t + 1 < my_nst; t += 2) {
        stage_body(t, accA, accB, t > 0);
        stage_body(t + 1, accB, accA, true);
    }
    if (t < my_nst) {
        stage_body(t, accA, accB, t > 0);
        flush(accA);
    } else if (my_nst > 0) {
        flush(accB);
    }

#ifdef SMK_BP_PROFILE
    if (g_bp_prof && lane == 0) {
        unsigned long long* o = g_bp_prof + ((size_t)blockIdx.x * C::NW + wave) * 4;
        o[0] = prof_[0]; o[1] = prof_[1]; o[2] = prof_[2]; o[3] = prof_[3];
    }
#endif
    if (!is_compute) return;
    // epilogue: fp64 totals (+ the small split terms), stored k-contiguous as doubles.
    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
#pragma unroll
    for (int c = 0; c < CW; ++c) {
        const i64 jg = col0 + (cwv * CW + c) * 32 + (lane & 31);
        double* pout = P + ((i64)split * ncols_pad + jg) * (KT * 32) + kw * KTW * 32;
#pragma unroll
        for (int kt = 0; kt < KTW; ++kt) {
            const int n = c * KTW + kt;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int i = 0; i < 4; i += 2) {
                    f64x2_t v;
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        double tsum = dacc[n][4 * g + i + u];
                        if constexpr (NSPLIT > 1) {
                            float small = accs[NSPLIT - 2][n][4 * g + i + u];
#pragma unroll
                            for (int s = NSPLIT - 3; s >= 0; --s) small += accs[s][n][4 * g + i + u];
                            tsum += (double)small;
                        }
                        v[u] = tsum;
                    }
                    *(f64x2_t*)(pout + kt * 32 + 8 * g + 4 * h + i) = v;
                }
            }
        }
    }
}

// ---- packing of the skinny operand -------------------------------------------------
// out layout: [q][s][kt][lane = (r, h)][16 B], chunk = 2q + h covers rows chunk*E .. +E-1,
// r = k index inside tile kt.  bf16: hi = bf16(x), mid = bf16(x-hi), lo = bf16(x-hi-mid).
template <int EBYTES, int NSPLIT>
__global__ __launch_bounds__(256) void pack_kernel(const double* __restrict__ X, int k, int ldx, i64 N, int KT, i64 nq,
                                                   unsigned char* __restrict__ out)
{
    constexpr int E = 16 / EBYTES;
    const i64 gid = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = (int)(gid & 63);
    const i64 rest = gid >> 6;
    const int kt = (int)(rest % KT);
    const i64 q = rest / KT;
    if (q >= nq) return;
    const int r = kt * 32 + (lane & 31);
    const i64 row0 = (2 * q + (lane >> 5)) * E;
    double v[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const i64 row = row0 + e;
        v[e] = (row < N && r < k) ? X[row * ldx + r] : 0.0;
    }
    if constexpr (EBYTES == 2) {
        double res[E];
#pragma unroll
        for (int e = 0; e < E; ++e) res[e] = v[e];
#pragma unroll
        for (int s = 0; s < NSPLIT; ++s) {
            unsigned short hbits[E];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                hbits[e] = f32_to_bf16_rne((float)res[e]);
                res[e] -= (double)bf16_bits_to_f32(hbits[e]);
            }
            u32x4_t w;
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = (unsigned)hbits[2 * e] | ((unsigned)hbits[2 * e + 1] << 16);
            *(u32x4_t*)(out + (((q * NSPLIT + s) * KT + kt) * 64 + lane) * 16) = w;
        }
    } else {
        f32x4_t w;
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = (float)v[e];
        *(f32x4_t*)(out + ((q * KT + kt) * 64 + lane) * 16) = w;
    }
}

// The packed operand layout does not depend on the stage height: it is a sequence of 1-KiB
// blocks indexed by the global chunk-pair q; rows are padded to a multiple of 128.
// operand format: bf16 fragments (E = 8) for bf16 storage and for the fp32 "bf16x3" emulation
// (nsplit == 3); native fp32 fragments (E = 4, one term) otherwise
static inline bool pack_is_bf16(int storage, int nsplit) { return storage == STORE_BF16 || nsplit == 3; }

static inline i64 pack_nq(int storage, int nsplit, i64 N)
{
    const i64 E = pack_is_bf16(storage, nsplit) ? 8 : 4;
    return round_up(N, ROW_PAD) / (2 * E);
}

size_t packed_bytes(int storage, int k, i64 N, int nsplit)
{
    if (!pack_is_bf16(storage, nsplit)) nsplit = 1;
    return (size_t)pack_nq(storage, nsplit, N) * nsplit * kt_of(k) * 1024;
}

int launch_pack(const double* X, int k, i64 N, int storage, int nsplit, void* out, hipStream_t st)
{
    const int KT = kt_of(k);
    const i64 nq = pack_nq(storage, nsplit, N);
    const i64 threads = nq * KT * 64;
    const int grid = (int)((threads + 255) / 256);
    if (grid == 0) return 0;
    if (pack_is_bf16(storage, nsplit)) {
        if (nsplit == 3) pack_kernel<2, 3><<<grid, 256, 0, st>>>(X, k, kp_of(k), N, KT, nq, (unsigned char*)out);
        else if (nsplit == 2) pack_kernel<2, 2><<<grid, 256, 0, st>>>(X, k, kp_of(k), N, KT, nq, (unsigned char*)out);
        else pack_kernel<2, 1><<<grid, 256, 0, st>>>(X, k, kp_of(k), N, KT, nq, (unsigned char*)out);
    } else {
        pack_kernel<4, 1><<<grid, 256, 0, st>>>(X, k, kp_of(k), N, KT, nq, (unsigned char*)out);
    }
    SMK_HIP(hipGetLastError());
    return 0;
}

constexpr bool bp_fits(int ebytes, int kt, int nsplit, int mb, int nstage, int cw, int wk, int nwl);
// ---- kernel variants (tile shape / pipeline depth); chosen per plan, SMK_BP_VARIANT overrides ----
struct BPVariant { int mb, nstage, cw, wk, nwl; };
static const BPVariant kVariants[] = {
    {64, 3, 1, 1, 0},    // 0: 128 cols x 64 rows, 3-deep ring
    {64, 4, 1, 1, 0},    // 1
    {64, 5, 1, 1, 0},    // 2
    {128, 2, 1, 1, 0},   // 3
    {128, 3, 1, 1, 0},   // 4
    {64, 3, 2, 1, 0},    // 5: 256 cols per workgroup
    {64, 2, 1, 1, 0},    // 6: two workgroups per CU
    {32, 2, 1, 1, 0},    // 7: 32-row stages (fp32: one 128-B line per column per stage)
    {32, 3, 1, 1, 0},    // 8
    {32, 4, 1, 1, 0},    // 9
    {64, 2, 1, 2, 0},    // 10: k in (32,64]: 8 waves, the two k tiles on different waves
    {64, 3, 1, 2, 0},    // 11
    {64, 4, 1, 2, 0},    // 12
    {32, 4, 1, 2, 0},    // 13
    {64, 2, 2, 2, 0},    // 14: 256 columns, 8 waves
    // wave-specialised: +4 (or +2) loader waves that only issue the LDS-DMA loads
    {64, 3, 1, 1, 4},    // 15
    {64, 4, 1, 1, 4},    // 16
    {64, 5, 1, 1, 4},    // 17
    {64, 3, 1, 2, 4},    // 18: k in (32,64]
    {64, 4, 1, 2, 4},    // 19
    {32, 4, 1, 1, 2},    // 20: fp32 emulation, k <= 32
    {32, 4, 1, 2, 4},    // 21: fp32 emulation, k in (32,64]
    {32, 5, 1, 2, 4},    // 22
    {64, 2, 1, 1, 4},    // 23
};
static const int kNumVariants = (int)(sizeof(kVariants) / sizeof(kVariants[0]));

BigProdPlan plan_bigprod(int storage, int k, i64 len, i64 ncols, int nsplit, int num_cus)
{
    BigProdPlan pl;
    pl.storage = storage;
    pl.kt = kt_of(k);
    // fp32 storage: nsplit 3 selects the bf16x3 emulation (default), 1 the native fp32 MFMA
    pl.nsplit = storage == STORE_BF16 ? nsplit : (nsplit == 3 ? 3 : 1);
    // measured best on MI355X: bf16 -> 64-row stages, 2-deep ring, 2 workgroups per CU (C3: 5.98 TB/s)
    int v = (storage == STORE_BF16) ? 6 : 7;
    // k in (32,64]: 8 compute waves (one k tile each) + 4 loader waves
    if (pl.kt == 2) v = (storage == STORE_BF16) ? 18 : 21;
    const char* env = getenv("SMK_BP_VARIANT");
    if (env) v = atoi(env);
    if (v < 0 || v >= kNumVariants) v = 6;
    // variants that do not fit the 160 KiB LDS for this dtype / k fall back to variant 0
    auto fits = [&](int vv) {
        return bp_fits(storage == STORE_BF16 ? 2 : 4, pl.kt, pl.nsplit, kVariants[vv].mb, kVariants[vv].nstage,
                       kVariants[vv].cw, kVariants[vv].wk, kVariants[vv].nwl);
    };
    if (!fits(v)) v = (pl.kt == 2) ? 11 : 0;
    if (!fits(v) && pl.kt == 2) v = 13;
    if (!fits(v)) v = 0;
    if (!fits(v)) v = 7;
    pl.variant = v;
    const int MB = kVariants[v].mb;
    const int NB = 128 * kVariants[v].cw;
    pl.stages = (len + MB - 1) / MB;
    pl.tiles = (ncols + NB - 1) / NB;
    pl.ncols_pad = round_up(ncols, COL_PAD);
    // enough workgroups for >= 4 rounds over the CUs, but keep >= 8 stages per split
    int S = 1;
    while (pl.tiles * S < 2 * (i64)num_cus && S < 64 && pl.stages / (2 * S) >= 8) S *= 2;
    const char* envS = getenv("SMK_BP_SPLITS");
    if (envS && atoi(envS) > 0) { S = 1; while (S < atoi(envS) && S < 64) S *= 2; }
    pl.S = S;
    pl.nst = (pl.stages + S - 1) / S;
    pl.p_elems = (size_t)S * pl.ncols_pad * pl.kt * 32;   // doubles
    return pl;
}

template <int EBYTES, int KT, int NSPLIT, int MB, int NSTAGE, int CW, int WK, int NWL>
static int launch_bigprod_t(const BigProdPlan& pl, const void* B, i64 ldb, const void* Xp, double* P, hipStream_t st)
{
    using C = BPCfg<EBYTES, KT, NSPLIT, MB, NSTAGE, CW, WK, NWL>;
    constexpr int lds = C::STAGE_BYTES * C::NSTAGE;
    static bool attr_set = false;
    auto kern = bigprod_kernel<EBYTES, KT, NSPLIT, MB, NSTAGE, CW, WK, NWL>;
    if (!attr_set) {
        SMK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_set = true;
    }
    int logS = 0;
    while ((1 << logS) < pl.S) ++logS;
    i64 grid;
    if (pl.S <= 8) {
        const i64 per = 8 >> logS;                       // tiles per group of 8 blocks
        grid = (pl.tiles + per - 1) / per * 8;
    } else {
        grid = pl.tiles * pl.S;
    }
    kern<<<(unsigned)grid, 64 * C::NW, lds, st>>>((const unsigned char*)B, ldb * EBYTES, (const unsigned char*)Xp, P,
                                           pl.stages, pl.nst, pl.tiles, pl.ncols_pad, pl.S, logS);
    SMK_HIP(hipGetLastError());
    return 0;
}

constexpr bool bp_fits(int ebytes, int kt, int nsplit, int mb, int nstage, int cw, int wk, int nwl)
{
    if (kt % wk != 0) return false;
    const int cpc = mb / (16 / ebytes);
    const int qs = (ebytes == 2 || nsplit == 3) ? mb / 16 : cpc / 2;
    if (qs < 1) return false;
    const int stage = 128 * cw * mb * ebytes + qs * nsplit * kt * 1024;
    const int ti = stage / 1024;
    const int nw = nwl > 0 ? nwl : 4 * wk;
    return (ti % nw == 0) && (ti / nw * (nstage - 1) <= 63) && (stage * nstage <= 160 * 1024);
}

template <int EBYTES, int KT, int NSPLIT, int MB, int NSTAGE, int CW, int WK, int NWL = 0>
static int launch_bigprod_if(const BigProdPlan& pl, const void* B, i64 ldb, const void* Xp, double* P, hipStream_t st)
{
    if constexpr (bp_fits(EBYTES, KT, NSPLIT, MB, NSTAGE, CW, WK, NWL))
        return launch_bigprod_t<EBYTES, KT, NSPLIT, MB, NSTAGE, CW, WK, NWL>(pl, B, ldb, Xp, P, st);
    else {
        set_error("bigprod variant does not fit LDS");
        return -100;
    }
}

template <int EBYTES, int KT, int NSPLIT>
static int launch_bigprod_v(const BigProdPlan& pl, const void* B, i64 ldb, const void* Xp, double* P, hipStream_t st)
{
    switch (pl.variant) {
        case 1: return launch_bigprod_if<EBYTES, KT, NSPLIT, 64, 4, 1, 1>(pl, B, ldb, Xp, P, st);
        case 2: return launch_bigprod_if<EBYTES, KT, NSPLIT, 64, 5, 1, 1>(pl, B, ldb, Xp, P, st);
        case 3: return launch_bigprod_if<EBYTES, KT, NSPLIT, 128, 2, 1, 1>(pl, B, ldb, Xp, P, st);
        case 4: return launch_bigprod_if<EBYTES, KT, NSPLIT, 128, 3, 1, 1>(pl, B, ldb, Xp, P, st);
        case 5: return launch_bigprod_if<EBYTES, KT, NSPLIT, 64, 3, 2, 1>(pl, B, ldb, Xp, P, st);
        case 6: return launch_bigprod_if<EBYTES, KT, NSPLIT, 64, 2, 1, 1>(pl, B, ldb, Xp, P, st);
        case 7: return launch_bigprod_if<EBYTES, KT, NSPLIT, 32, 2, 1, 1>(pl, B, ldb, Xp, P, st);
        case 8: return launch_bigprod_if<EBYTES, KT, NSPLIT, 32, 3, 1, 1>(pl, B, ldb, Xp, P, st);
        case 9: return launch_bigprod_if<EBYTES, KT, NSPLIT, 32, 4, 1, 1>(pl, B, ldb, Xp, P, st);
        case 10: return launch_bigprod_if<EBYTES, KT, NSPLIT, 64, 2, 1, 2>(pl, B, ldb, Xp, P, st);
        case 11: return launch_bigprod_if<EBYTES, KT, NSPLIT, 64, 3, 1, 2>(pl, B, ldb, Xp, P, st);
        case 12: return launch_bigprod_if<EBYTES, KT, NSPLIT, 64, 4, 1, 2>(pl, B, ldb, Xp, P, st);
        case 13: return launch_bigprod_if<EBYTES, KT, NSPLIT, 32, 4, 1, 2>(pl, B, ldb, Xp, P, st);
        case 14: return launch_bigprod_if<EBYTES, KT, NSPLIT, 64, 2, 2, 2>(pl, B, ldb, Xp, P, st);
        case 15: return launch_bigprod_if<EBYTES, KT, NSPLIT, 64, 3, 1, 1, 4>(pl, B, ldb, Xp, P, st);
        case 16: return launch_bigprod_if<EBYTES, KT, NSPLIT, 64, 4, 1, 1, 4>(pl, B, ldb, Xp, P, st);
        case 17: return launch_bigprod_if<EBYTES, KT, NSPLIT, 64, 5, 1, 1, 4>(pl, B, ldb, Xp, P, st);
        case 18: return launch_bigprod_if<EBYTES, KT, NSPLIT, 64, 3, 1, 2, 4>(pl, B, ldb, Xp, P, st);
        case 19: return launch_bigprod_if<EBYTES, KT, NSPLIT, 64, 4, 1, 2, 4>(pl, B, ldb, Xp, P, st);
        case 20: return launch_bigprod_if<EBYTES, KT, NSPLIT, 32, 4, 1, 1, 2>(pl, B, ldb, Xp, P, st);
        case 21: return launch_bigprod_if<EBYTES, KT, NSPLIT, 32, 4, 1, 2, 4>(pl, B, ldb, Xp, P, st);
        case 22: return launch_bigprod_if<EBYTES, KT, NSPLIT, 32, 5, 1, 2, 4>(pl, B, ldb, Xp, P, st);
        case 23: return launch_bigprod_if<EBYTES, KT, NSPLIT, 64, 2, 1, 1, 4>(pl, B, ldb, Xp, P, st);
        default: break;
    }
    return launch_bigprod_if<EBYTES, KT, NSPLIT, 64, 3, 1, 1>(pl, B, ldb, Xp, P, st);
}

int launch_bigprod(const BigProdPlan& pl, const void* B, i64 ldb, const void* Xp, double* P, hipStream_t st)
{
    if (pl.storage == STORE_BF16) {
        if (pl.kt == 1) {
            if (pl.nsplit == 3) return launch_bigprod_v<2, 1, 3>(pl, B, ldb, Xp, P, st);
            if (pl.nsplit == 2) return launch_bigprod_v<2, 1, 2>(pl, B, ldb, Xp, P, st);
            return launch_bigprod_v<2, 1, 1>(pl, B, ldb, Xp, P, st);
        } else {
            if (pl.nsplit == 3) return launch_bigprod_v<2, 2, 3>(pl, B, ldb, Xp, P, st);
            if (pl.nsplit == 2) return launch_bigprod_v<2, 2, 2>(pl, B, ldb, Xp, P, st);
            return launch_bigprod_v<2, 2, 1>(pl, B, ldb, Xp, P, st);
        }
    } else {
        if (pl.nsplit == 3) {
            if (pl.kt == 1) return launch_bigprod_v<4, 1, 3>(pl, B, ldb, Xp, P, st);
            return launch_bigprod_v<4, 2, 3>(pl, B, ldb, Xp, P, st);
        }
        if (pl.kt == 1) return launch_bigprod_v<4, 1, 1>(pl, B, ldb, Xp, P, st);
        return launch_bigprod_v<4, 2, 1>(pl, B, ldb, Xp, P, st);
    }
}

// sum the S slabs into one fp32 slab (used before a cross-GPU all-reduce)
__global__ __launch_bounds__(256) void reduce_partials_kernel(const double* __restrict__ p, int S, i64 slab, i64 count,
                                                              float* __restrict__ out)
{
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (i64)gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int t = 0; t < S; ++t) s += p[t * slab + i];
        out[i] = (float)s;
    }
}

int launch_reduce_partials(PartialView pv, int k, i64 N, float* out, hipStream_t st)
{
    i64 count = N * pv.kpp;
    if (count == 0) return 0;
    int grid = (int)((count + 255) / 256 < 4096 ? (count + 255) / 256 : 4096);
    reduce_partials_kernel<<<grid, 256, 0, st>>>((const double*)pv.p, pv.S, pv.slab, count, out);
    SMK_HIP(hipGetLastError());
    return 0;
}

// ==========================================================================
// Layout of the factor-side matrices: fp64, column-major KP x N with leading
// dimension KP (k padded to 8/16/32/64, pad rows are zero and stay zero).
// "Column tile" kernels give each column to LPC = KP/4 adjacent lanes, 4
// consecutive values (32 B) per lane, so a wave reads 2 KiB contiguous.
// Dot products over the column are summed across the lane group with DPP
// moves (no LDS traffic).
// ==========================================================================
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

// sum over the LPC (2/4/8/16) adjacent lanes of a group; every lane gets the total
template <int LPC>
__device__ __forceinline__ double group_sum(double v)
{
    if constexpr (LPC >= 2) v += dpp_f64<0xB1>(v);    // quad_perm [1,0,3,2]
    if constexpr (LPC >= 4) v += dpp_f64<0x4E>(v);    // quad_perm [2,3,0,1]
    if constexpr (LPC >= 8) v += dpp_f64<0x141>(v);   // row_half_mirror
    if constexpr (LPC >= 16) v += dpp_f64<0x140>(v);  // row_mirror
    return v;
}

__device__ __forceinline__ double readlane_f64(double v, int srclane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), srclane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), srclane);
    return __hiloint2double(hi, lo);
}

// sum over the 64 lanes of a wave, result in every lane: DPP row reductions + 4 readlanes
// (no LDS-crossbar permutes, fixed order)
__device__ __forceinline__ double wave_sum(double v)
{
    v = group_sum<16>(v);
    return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}

// the 4 values of column j owned by sub-lane s
__device__ __forceinline__ void load4(const double* __restrict__ p, double (&x)[4])
{
    const f64x2_t a = *(const f64x2_t*)p, b = *(const f64x2_t*)(p + 2);
    x[0] = a[0]; x[1] = a[1]; x[2] = b[0]; x[3] = b[1];
}
__device__ __forceinline__ void store4(double* __restrict__ p, const double (&x)[4])
{
    f64x2_t a, b;
    a[0] = x[0]; a[1] = x[1]; b[0] = x[2]; b[1] = x[3];
    *(f64x2_t*)p = a;
    *(f64x2_t*)(p + 2) = b;
}

// the same 4 entries of the summed partial product (zero beyond kpp)
__device__ __forceinline__ void load_rhs4(const PartialView& R, i64 j, int e0, double (&b)[4])
{
    b[0] = b[1] = b[2] = b[3] = 0.0;
    if (e0 >= R.kpp) return;
    if (R.f64) {
        for (int s = 0; s < R.S; ++s) {
            double t[4];
            load4((const double*)R.p + s * R.slab + j * R.kpp + e0, t);
            b[0] += t[0]; b[1] += t[1]; b[2] += t[2]; b[3] += t[3];
        }
    } else {
        for (int s = 0; s < R.S; ++s) {
            const f32x4_t t = *(const f32x4_t*)((const float*)R.p + s * R.slab + j * R.kpp + e0);
            b[0] += (double)t[0]; b[1] += (double)t[1]; b[2] += (double)t[2]; b[3] += (double)t[3];
        }
    }
}

// one element (row i of column j) of the summed partial product
__device__ __forceinline__ double rhs_elem(const PartialView& R, i64 j, int i)
{
    double v = 0.0;
    if (R.f64) {
        for (int s = 0; s < R.S; ++s) v += ((const double*)R.p)[s * R.slab + j * R.kpp + i];
    } else {
        for (int s = 0; s < R.S; ++s) v += (double)((const float*)R.p)[s * R.slab + j * R.kpp + i];
    }
    return v;
}

// block-wide sum of one double; result valid in thread 0
__device__ __forceinline__ double block_sum(double v, double* sh /* >= 4 doubles */)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) sh[w] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        for (int i = 0; i < nw; ++i) t += sh[i];
    }
    __syncthreads();
    return t;
}

// sum `n` partials (fixed order per thread stride) -> broadcast to the whole block
__device__ __forceinline__ double block_sum_array(const double* __restrict__ p, int n, double* sh)
{
    double v = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) v += p[i];
    double t = block_sum(v, sh);
    if (threadIdx.x == 0) sh[8] = t;
    __syncthreads();
    t = sh[8];
    __syncthreads();
    return t;
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
typedef __attribute__((ext_vector_type(4))) double f64x4_t;

template <int KP>
__global__ __launch_bounds__(256) void gram_mfma_kernel(const double* __restrict__ X, i64 N, i64 cols_per_wave,
                                                        double* __restrict__ Gp)
{
    constexpr int T = KP / 16;
    __shared__ double red[KP * KP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const i64 wg = (i64)blockIdx.x * 4 + wave;
    const i64 c_begin = wg * cols_per_wave;
    i64 c_end = c_begin + cols_per_wave;
    if (c_end > N) c_end = N;
    f64x4_t acc[T][T];
#pragma unroll
    for (int a = 0; a < T; ++a)
#pragma unroll
        for (int b = 0; b < T; ++b) acc[a][b] = f64x4_t{0.0, 0.0, 0.0, 0.0};
    const int kc = lane >> 4, r16 = lane & 15;
    // 16 columns (4 MFMA k-steps) per trip: all loads of the trip are issued before its MFMAs
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
                for (int b = 0; b < T; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[u][a], f[u][b], acc[a][b], 0, 0, 0);
    }
    // deterministic in-block sum of the 4 waves
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int a = 0; a < T; ++a)
#pragma unroll
                for (int b = 0; b < T; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * a + kc + 4 * r, colm = 16 * b + r16;
                        const int idx = colm * KP + row;
                        red[idx] = (w == 0) ? acc[a][b][r] : red[idx] + acc[a][b][r];
                    }
        }
        __syncthreads();
    }
    double* out = Gp + (i64)blockIdx.x * KP * KP;
    for (int i = threadIdx.x; i < KP * KP; i += 256) out[i] = red[i];
}

// G[e] = sum_b Gp[b][e]: 16 elements per block, 16 thread groups stride the partials, fixed order
__global__ __launch_bounds__(256) void gram_reduce_kernel(const double* __restrict__ Gp, int nblk, int elems,
                                                          double* __restrict__ G)
{
    __shared__ double sh[16][17];
    const int el = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int e = blockIdx.x * 16 + el;
    double s = 0.0;
    if (e < elems) {
#pragma unroll 8
        for (int b = g; b < nblk; b += 16) s += Gp[(i64)b * elems + e];
    }
    sh[g][el] = s;
    __syncthreads();
    if (g == 0 && e < elems) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += sh[i][el];
        G[e] = t;
    }
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

size_t gram_scratch_elems(int k, int max_blocks)
{
    int KP = kp_of(k);
    return (size_t)max_blocks * KP * KP;
}

int launch_gram(const double* X, int k, i64 N, double* G, double* scratch, int max_blocks, hipStream_t st)
{
    const int KP = kp_of(k);
    const int elems = KP * KP;
    int nblk;
    if (KP >= 16) {
        nblk = (int)((N + 255) / 256);               // >= 64 columns per wave
        if (nblk > max_blocks) nblk = max_blocks;
        if (nblk < 1) nblk = 1;
        i64 cpw = (N + (i64)nblk * 4 - 1) / ((i64)nblk * 4);
        cpw = (cpw + 15) / 16 * 16;
        switch (KP) {
            case 16: gram_mfma_kernel<16><<<nblk, 256, 0, st>>>(X, N, cpw, scratch); break;
            case 32: gram_mfma_kernel<32><<<nblk, 256, 0, st>>>(X, N, cpw, scratch); break;
            default: gram_mfma_kernel<64><<<nblk, 256, 0, st>>>(X, N, cpw, scratch); break;
        }
    } else {
        nblk = (int)((N + 255) / 256);
        if (nblk > max_blocks) nblk = max_blocks;
        if (nblk < 1) nblk = 1;
        gram_stream8_kernel<<<nblk, 256, 0, st>>>(X, N, scratch);
    }
    SMK_HIP(hipGetLastError());
    gram_reduce_kernel<<<(elems + 15) / 16, 256, 0, st>>>(scratch, nblk, elems, G);
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
    __shared__ __attribute__((aligned(16))) double gs[KP * KP];                       \
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
    __shared__ __attribute__((aligned(16))) double gs[KP * KP];
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
    __shared__ __attribute__((aligned(16))) double gs[KP * KP];
    if ((int)blockIdx.x < grid1) grad_pg_body<KP>(X1, k, N1, R1, G1, nullptr, part1, blockIdx.x, gs, sh);
    else grad_pg_body<KP>(X2, k, N2, R2, G2, nullptr, part2, blockIdx.x - grid1, gs, sh);
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

#define KP_DISPATCH(KPV, CALL)              \
    switch (KPV) {                          \
        case 8: { constexpr int KP = 8; CALL; } break;   \
        case 16: { constexpr int KP = 16; CALL; } break; \
        case 32: { constexpr int KP = 32; CALL; } break; \
        default: { constexpr int KP = 64; CALL; } break; \
    }

static inline int coltile_grid(int KP, i64 N) { return (int)((N * (KP / 4) + 255) / 256); }

int launch_mu_update(double* X, int k, i64 N, PartialView R, const double* G, hipStream_t st)
{
    const int KPv = kp_of(k), grid = coltile_grid(KPv, N);
    KP_DISPATCH(KPv, (mu_update_kernel<KP><<<grid, 256, 0, st>>>(X, k, N, R, G)));
    SMK_HIP(hipGetLastError());
    return 0;
}

int launch_hals_sweep(double* X, int k, i64 N, PartialView R, const double* G, hipStream_t st)
{
    const int KPv = kp_of(k), grid = coltile_grid(KPv, N);
    KP_DISPATCH(KPv, (hals_sweep_kernel<KP><<<grid, 256, 0, st>>>(X, k, N, R, G)));
    SMK_HIP(hipGetLastError());
    return 0;
}

int launch_grad_pg(const double* X, int k, i64 N, PartialView R, const double* G, double* grad_out,
                   double* pg_partials, double* pg_accum, int slot, hipStream_t st)
{
    const int KPv = kp_of(k), grid = coltile_grid(KPv, N);
    KP_DISPATCH(KPv, (grad_pg_kernel<KP><<<grid, 256, 0, st>>>(X, k, N, R, G, grad_out, pg_partials)));
    SMK_HIP(hipGetLastError());
    sum_partials_kernel<<<1, 256, 0, st>>>(pg_partials, grid, pg_accum + slot);
    SMK_HIP(hipGetLastError());
    return 0;
}

// projected-gradient sums of both factors: pg_accum[0] (side 1), pg_accum[1] (side 2), and the failure flag
// as a double in pg_accum[flag_slot]; two launches instead of four
int launch_grad_pg2(const double* X1, i64 N1, PartialView R1, const double* G1, double* part1, const double* X2, i64 N2,
                    PartialView R2, const double* G2, double* part2, int k, double* pg_accum, const int* flag,
                    int flag_slot, hipStream_t st)
{
    const int KPv = kp_of(k), g1 = coltile_grid(KPv, N1), g2 = coltile_grid(KPv, N2);
    KP_DISPATCH(KPv, (grad_pg2_kernel<KP><<<g1 + g2, 256, 0, st>>>(X1, N1, R1, G1, part1, g1, X2, N2, R2, G2, part2, k)));
    SMK_HIP(hipGetLastError());
    sum_partials2_kernel<<<2, 256, 0, st>>>(part1, g1, part2, g2, pg_accum, flag, flag_slot);
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
                                                  int nblk, int lane, int wave)
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
        // gather every workgroup's partial: up to 4 slots per lane polled together (bounded spin)
        double acc = 0.0;
        bool ok = true;
        for (int b0 = 0; b0 < nblk; b0 += 256) {
            unsigned long long bits[4];
            bool have[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { have[u] = (b0 + u * 64 + lane) < nblk; bits[u] = have[u] ? kSlotEmpty : 0ull; }
            for (unsigned spin = 0; spin < (1u << 22); ++spin) {
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
                                                 unsigned long long* __restrict__ slots, int nblk, int lane, int wave)
{
    (hals_w_fused_step<KP, NT, Cs>(w, rhs, dead, gs, sh, k, M, row, valid, R, slots, nblk, lane, wave), ...);
}

template <int KP, int NT>
__global__ __launch_bounds__(NT) void hals_w_fused_kernel(double* __restrict__ Wt, int k, i64 M, PartialView R,
                                                          const double* __restrict__ G,
                                                          unsigned long long* __restrict__ slots, int nblk,
                                                          int* __restrict__ fail_flag)
{
    __shared__ __attribute__((aligned(16))) double gs[KP * KP];
    __shared__ double sh[40];
    for (int t = threadIdx.x; t < KP * KP; t += NT) gs[t] = G[t];

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
                             nblk, lane, wave);
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
}

size_t hals_w_scratch_elems(int k, i64 M)
{
    const size_t multi = (size_t)(2 * (i64)k * hals_w_blocks(kp_of(k), M));
    const size_t fused = (size_t)k * 1024;          // slots (8 bytes each), generous
    return multi > fused ? multi : fused;
}

int launch_hals_w_update(double* Wt, int k, i64 M, PartialView R, const double* G, double* scratch, int num_cus,
                         int* fail_flag, hipStream_t st)
{
    const int KPv = kp_of(k);
    static int mode = -1;                            // SMK_HALS_W=multi forces the one-launch-per-column path
    if (mode < 0) {
        const char* env = getenv("SMK_HALS_W");
        mode = (env && env[0] == 'm') ? 0 : 1;
    }
    // fused path: at most one workgroup per CU so that all of them are resident by construction.
    // Smallest workgroup (256 threads: cheapest in-block sync, measured best) that still covers M
    // rows with <= num_cus workgroups; the register budget caps it at 512 for KP = 32 and 256 for 64.
    int nt = 0;
    const int nt_max = (KPv == 64) ? 256 : (KPv == 32) ? 512 : 1024;
    for (int cand = 256; cand <= nt_max; cand *= 2)
        if ((M + cand - 1) / cand <= (i64)num_cus) { nt = cand; break; }
    if (mode == 1 && nt != 0) {
        const i64 nblk_f = (M + nt - 1) / nt;
        unsigned long long* slots = (unsigned long long*)scratch;
        SMK_HIP(hipMemsetAsync(slots, 0xFF, (size_t)k * nblk_f * sizeof(unsigned long long), st));
        const int nb = (int)nblk_f;
#define SMK_FUSED(KPX, NTX) hals_w_fused_kernel<KPX, NTX><<<nb, NTX, 0, st>>>(Wt, k, M, R, G, slots, nb, fail_flag)
        switch (KPv) {
            case 8: if (nt == 256) SMK_FUSED(8, 256); else if (nt == 512) SMK_FUSED(8, 512); else SMK_FUSED(8, 1024); break;
            case 16: if (nt == 256) SMK_FUSED(16, 256); else if (nt == 512) SMK_FUSED(16, 512); else SMK_FUSED(16, 1024); break;
            case 32: if (nt == 256) SMK_FUSED(32, 256); else SMK_FUSED(32, 512); break;
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
        KP_DISPATCH(KPv, (hals_w_col_kernel<KP><<<nblk, 1024, 0, st>>>(Wt, k, M, R, G, c, nblk, ss, nz)));
    }
    SMK_HIP(hipGetLastError());
    return 0;
}

// ==========================================================================
// NNLS by block principal pivoting, one column per GS-lane group (GS = KP).
// Lane i of a group owns component i of the column: x_i, y_i, rhs_i, its
// passive bit, and row i of the masked Gram matrix in registers.  The passive
// sub-system G[F,F] x_F = rhs_F is solved by Gauss-Jordan elimination on the
// masked matrix (non-passive rows/columns replaced by identity) -- pivots are
// the Cholesky pivots, so "pivot <= 0" is exactly the reference's non-SPD
// failure (normal_eq.hpp:35-50).  Pivot-row values are broadcast with
// v_readlane (GS = 64) or ds_bpermute (GS < 64).
// Per-column state machine = NnlsBlockpivot (nnls.hpp:144-244) restricted to one
// column: columns are independent in the reference except for the shared
// iteration cap (5k), which here is per column.  The backup rule toggles the
// TRUE largest index (the reference's MaxRowIndex is off by 32 for k >= 64,
// bit_matrix.cpp:456-468; the NNLS optimum is unique so results agree).
// ==========================================================================
template <int GS>
__device__ __forceinline__ double group_bcast(double v, int src /* compile-time after unroll */)
{
    if constexpr (GS == 64) {
        int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
        int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
        return __hiloint2double(hi, lo);
    } else {
        return __shfl(v, src, GS);
    }
}

template <int GS>
__device__ __forceinline__ unsigned long long group_ballot(bool pred, int lane)
{
    unsigned long long b = __ballot(pred);
    if constexpr (GS == 64) return b;
    else {
        const int shift = (lane / GS) * GS;
        return (b >> shift) & ((1ull << GS) - 1ull);
    }
}

// KP = 64: left alone the compiler takes 256 VGPRs + 40 AGPRs (one wave per SIMD) and every
// readlane -> FMA dependency is exposed; capping at 168 registers (3 waves per SIMD, 516 B of
// scratch per lane) is 1.45x faster on a 262144 x 8192 k = 64 BPP iteration.  No gain at KP <= 32.
template <int KP>
__global__ __launch_bounds__(256, (KP == 64 ? 3 : 1)) void nnls_bpp_kernel(double* __restrict__ X, double* __restrict__ Y, int k, i64 N,
                                                       PartialView R, const double* __restrict__ G,
                                                       int* __restrict__ fail_flag, int iter_tag, i64 col_begin)
{
    constexpr int GS = KP;
    constexpr int GPB = 256 / GS;                   // column groups per block
    __shared__ double gs[KP * KP];                  // gs[c*KP + i] = G[i][c] (symmetric)
    for (int t = threadIdx.x; t < KP * KP; t += blockDim.x) gs[t] = G[t];
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int i = threadIdx.x % GS;                 // component owned by this lane
    const i64 col = col_begin + (i64)blockIdx.x * GPB + threadIdx.x / GS;
    const bool col_ok = col < N;
    const bool comp_ok = i < k;
    const i64 cc = col_ok ? col : (N - 1);

    double rhs = 0.0, x = 0.0, y = 0.0;
    if (comp_ok) {
        rhs = rhs_elem(R, cc, i);
        x = X[cc * KP + i];
    }
    bool passive = comp_ok && (x > 0.0);            // passive_set = (X > 0), nnls.hpp:157
    const unsigned long long kmask = (k >= 64) ? ~0ull : ((1ull << k) - 1ull);
    int failed = 0;

    auto solve = [&](unsigned long long F) {
        // masked matrix row i
        double a[KP];
#pragma unroll
        for (int c = 0; c < KP; ++c) {
            const bool pc = (F >> c) & 1ull;
            a[c] = (passive && pc) ? gs[c * KP + i] : ((c == i) ? 1.0 : 0.0);
        }
        double b = passive ? rhs : 0.0;
#pragma unroll
        for (int j = 0; j < KP; ++j) {
            // wave-uniform skip when no group in this wave has j passive
            const bool pj = (F >> j) & 1ull;
            if (__ballot(pj) == 0ull) continue;
            const double piv = group_bcast<GS>(a[j], j);
            if (pj && !(piv > 0.0)) failed = 1;
            const double f = (i == j || !pj) ? 0.0 : a[j] / piv;
#pragma unroll
            for (int c = j + 1; c < KP; ++c) a[c] -= f * group_bcast<GS>(a[c], j);
            b -= f * group_bcast<GS>(b, j);
        }
        double d = 1.0;
#pragma unroll
        for (int c = 0; c < KP; ++c)
            if (c == i) d = a[c];
        x = passive ? (b / d) : 0.0;
    };

    auto residual = [&]() {          // y = G x - rhs
        double acc = 0.0;
#pragma unroll
        for (int c = 0; c < KP; ++c) acc += gs[c * KP + i] * group_bcast<GS>(x, c);
        y = comp_ok ? (acc - rhs) : 0.0;
    };

    unsigned long long F = group_ballot<GS>(passive, lane) & kmask;
    solve(F);
    residual();

    unsigned long long nonopt = group_ballot<GS>(comp_ok && !passive && (y < 0.0), lane);
    unsigned long long infeas = group_ballot<GS>(comp_ok && passive && (x < 0.0), lane);
    int ng = __popcll(nonopt) + __popcll(infeas);
    int Pc = 3, Ninf = k + 1;                       // PBAR = 3, nnls.hpp:152,170
    const int max_iter = 5 * k;
    int iter = 0;
    bool active = col_ok && ng > 0;

    while (__ballot(active) != 0ull) {
        if (active) {
            if (iter >= max_iter) { failed = 1; active = false; }
        }
        if (active) {
            // UpdatePassiveSet, src/nnls.cpp:18-74
            if (ng < Ninf) { Pc = 3; Ninf = ng; F = (F | nonopt) & ~infeas; }
            else if (Pc >= 1) { Pc -= 1; F = (F | nonopt) & ~infeas; }
            else {
                const int r1 = nonopt ? (63 - __clzll(nonopt)) : 0;
                const int r2 = infeas ? (63 - __clzll(infeas)) : 0;
                F ^= (1ull << (r1 > r2 ? r1 : r2));
            }
            F &= kmask;
            passive = (F >> i) & 1ull;
        }
        // all lanes execute the cross-lane code; inactive groups keep their state
        const double x_keep = x, y_keep = y;
        solve(F);
        if (fabs(x) < 1.0e-12) x = 0.0;             // ZeroizeSmallValues, nnls.hpp:213,224
        residual();
        if (fabs(y) < 1.0e-12) y = 0.0;             // :225
        if (!active) { x = x_keep; y = y_keep; }
        const unsigned long long no2 = group_ballot<GS>(comp_ok && !passive && (y < 0.0), lane);
        const unsigned long long in2 = group_ballot<GS>(comp_ok && passive && (x < 0.0), lane);
        if (active) {
            nonopt = no2;
            infeas = in2;
            ng = __popcll(nonopt) + __popcll(infeas);
            ++iter;
            if (ng == 0) active = false;
        }
    }

    if (col_ok && comp_ok) {
        X[col * KP + i] = x;
        if (Y) Y[col * KP + i] = y;
    }
    if (failed && col_ok) atomicMin(fail_flag, iter_tag);
}

// solves columns [col_begin, col_end) of X (col_end <= N); other columns are untouched
int launch_nnls_bpp(double* X, double* Y, int k, i64 col_begin, i64 col_end, PartialView R, const double* G,
                    int* fail_flag, int iter_tag, hipStream_t st)
{
    const int KPv = kp_of(k);
    const int gpb = 256 / KPv;
    const i64 ncols = col_end - col_begin;
    if (ncols <= 0) return 0;
    const int grid = (int)((ncols + gpb - 1) / gpb);
    const i64 N = col_end;
    KP_DISPATCH(KPv, (nnls_bpp_kernel<KP><<<grid, 256, 0, st>>>(X, Y, k, N, R, G, fail_flag, iter_tag, col_begin)));
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
                                                          const double* __restrict__ X, double* __restrict__ P, int kpp)
{
    constexpr int LPC = KP / 4;
    const i64 gtid = (i64)blockIdx.x * blockDim.x + threadIdx.x;
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

// k <= 2 (the RANK2 / HierNMF2 hot loop): one lane per output column, 16 bytes of the X row per stored
// entry instead of two lanes x 32 bytes, two independent accumulation chains per lane so that two
// gathers are in flight.  `ldx` = row pitch of X in doubles (KP, or 2 for a compact copy).
__global__ __launch_bounds__(256) void spmm_gather2_kernel(const i64* __restrict__ colptr,
                                                           const unsigned* __restrict__ rowidx,
                                                           const double* __restrict__ val, i64 ncols,
                                                           const double* __restrict__ X, int ldx,
                                                           double* __restrict__ P, int kpp)
{
    const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ncols) return;
    double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
    const i64 p0 = colptr[j], p1 = colptr[j + 1];
    i64 p = p0;
    for (; p + 1 < p1; p += 2) {
        const double v0 = val[p], v1 = val[p + 1];
        const f64x2_t x0 = *(const f64x2_t*)(X + (i64)rowidx[p] * ldx);
        const f64x2_t x1 = *(const f64x2_t*)(X + (i64)rowidx[p + 1] * ldx);
        a0 += v0 * x0[0]; a1 += v0 * x0[1];
        b0 += v1 * x1[0]; b1 += v1 * x1[1];
    }
    if (p < p1) {
        const double v0 = val[p];
        const f64x2_t x0 = *(const f64x2_t*)(X + (i64)rowidx[p] * ldx);
        a0 += v0 * x0[0]; a1 += v0 * x0[1];
    }
    double* out = P + j * kpp;
    f64x2_t r;
    r[0] = a0 + b0;
    r[1] = a1 + b1;
    *(f64x2_t*)out = r;
    for (int e = 2; e < kpp && e < 8; e += 2) { f64x2_t z; z[0] = 0.0; z[1] = 0.0; *(f64x2_t*)(out + e) = z; }
}

int launch_spmm_gather(const i64* colptr, const unsigned* rowidx, const double* val, i64 ncols, const double* X,
                       int k, double* P, int kpp, hipStream_t st)
{
    const int KPv = kp_of(k);
    static const bool rank2_path = [] { const char* e = getenv("SMK_SPMM2"); return !(e && e[0] == '0'); }();
    if (k <= 2 && rank2_path) {
        const int grid2 = (int)((ncols + 255) / 256);
        if (grid2 == 0) return 0;
        spmm_gather2_kernel<<<grid2, 256, 0, st>>>(colptr, rowidx, val, ncols, X, KPv, P, kpp);
        SMK_HIP(hipGetLastError());
        return 0;
    }
    const int grid = (int)((ncols * (KPv / 4) + 255) / 256);
    if (grid == 0) return 0;
    KP_DISPATCH(KPv, (spmm_gather_kernel<KP><<<grid, 256, 0, st>>>(colptr, rowidx, val, ncols, X, P, kpp)));
    SMK_HIP(hipGetLastError());
    return 0;
}

// ==========================================================================
// RANK2 (nmf_solver_rank2.hpp): closed-form 2x2 solves by one fast Givens rotation
// (SystemSolveH :25-135 / SystemSolveW :139-212) followed by the optimal active set
// (:216-318).  One thread per column of X (KP = 8 layout, rows 0 and 1 live).
// ==========================================================================
// Gp != nullptr: the kernel also leaves per-workgroup partial sums of X X' (the Gram matrix every
// RANK2 step needs right after the solve) in Gp[block][64], so the solved factor is not re-read.
__global__ __launch_bounds__(256) void rank2_solve_kernel(double* __restrict__ X, i64 N, PartialView R,
                                                          const double* __restrict__ G, int side,
                                                          int* __restrict__ fail_flag, int iter_tag,
                                                          double* __restrict__ Gp)
{
    constexpr int KP = 8;
    __shared__ double shg[4][3];
    const double eps = DBL_EPSILON;
    const double a00 = G[0], a10 = G[1], a01 = G[KP], a11 = G[KP + 1];
    bool bad = (fabs(a00) < eps) && (fabs(a01) < eps);          // "singular matrix"
    const bool cosine = fabs(a00) >= fabs(a01);
    double t, a2, b2, d2;
    if (side == 0) {
        if (cosine) { t = -a10 / a00; a2 = a00 - t * a10; b2 = a01 - t * a11; d2 = a11 + t * a01; }
        else        { t = -a00 / a10; a2 = -a10 + t * a00; b2 = -a11 + t * a01; d2 = a01 + t * a11; }
    } else {
        if (cosine) { t = a01 / a00; a2 = a00 + t * a01; b2 = a10 + t * a11; d2 = a11 - t * a10; }
        else        { t = a00 / a01; a2 = -a01 - t * a00; b2 = -a11 - t * a10; d2 = a10 - t * a11; }
    }
    const double inv_a2 = 1.0 / a2, inv_d2 = 1.0 / d2;
    if (fabs(d2 / a2) < eps) bad = true;
    if (bad) {
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicMin(fail_flag, iter_tag);
        return;
    }
    const double inv0 = 1.0 / a00, inv1 = 1.0 / a11, sq0 = sqrt(a00), sq1 = sqrt(a11);
    const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = j < N;
    if (!valid && !Gp) return;
    double x0 = 0.0, x1 = 0.0;
    if (valid) {
        const double b0 = rhs_elem(R, j, 0), b1 = rhs_elem(R, j, 1);
        double e2, f2;
        if (side == 0) {
            if (cosine) { e2 = b0 - t * b1; f2 = b1 + t * b0; }
            else        { e2 = -b1 + t * b0; f2 = b0 + t * b1; }
        } else {
            if (cosine) { e2 = b0 + t * b1; f2 = b1 - t * b0; }
            else        { e2 = -b1 - t * b0; f2 = b0 - t * b1; }
        }
        x1 = f2 * inv_d2;
        x0 = (e2 - b2 * x1) * inv_a2;
        if (x0 <= 0.0 || x1 <= 0.0) {               // OptimalActiveSet
            double v1 = b0 * inv0, v2 = b1 * inv1;
            if (v1 * sq0 >= v2 * sq1) v2 = 0.0; else v1 = 0.0;
            x0 = v1;
            x1 = v2;
        }
        f64x2_t v;
        v[0] = x0;
        v[1] = x1;
        *(f64x2_t*)(X + j * KP) = v;
    }
    if (!Gp) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double s00 = wave_sum(x0 * x0), s01 = wave_sum(x0 * x1), s11 = wave_sum(x1 * x1);
    if (lane == 0) { shg[wave][0] = s00; shg[wave][1] = s01; shg[wave][2] = s11; }
    __syncthreads();
    if (threadIdx.x < KP * KP) {
        const int e = threadIdx.x;
        const int q = (e == 0) ? 0 : (e == 1 || e == KP) ? 1 : (e == KP + 1) ? 2 : -1;
        Gp[(i64)blockIdx.x * KP * KP + e] = (q < 0) ? 0.0 : (shg[0][q] + shg[1][q]) + (shg[2][q] + shg[3][q]);
    }
}

// Gout != nullptr: also Gout = X X' (KP x KP), through `scratch` (rank2_gram_scratch_elems(N) doubles)
int launch_rank2_solve(double* X, i64 N, PartialView R, const double* G, int side, int* fail_flag, int iter_tag,
                       double* Gout, double* scratch, hipStream_t st)
{
    const int grid = (int)((N + 255) / 256);
    rank2_solve_kernel<<<grid, 256, 0, st>>>(X, N, R, G, side, fail_flag, iter_tag, Gout ? scratch : nullptr);
    SMK_HIP(hipGetLastError());
    if (Gout) {
        gram_reduce_kernel<<<4, 256, 0, st>>>(scratch, grid, 64, Gout);
        SMK_HIP(hipGetLastError());
    }
    return 0;
}
size_t rank2_gram_scratch_elems(i64 N) { return (size_t)((N + 255) / 256) * 64; }

// Per-iteration NormalizeAndScale of RANK2 (nmf_solver_rank2.hpp:418-437) in one launch: H rows *= nu,
// W columns /= nu, the stored AH' *= nu per column, HH'_ij *= nu_i nu_j; nu_c = sqrt(Gw[c][c]).
// A zero norm reports -2 through fail_flag and leaves that component unscaled (the reference throws).
__global__ __launch_bounds__(256) void rank2_normalize_kernel(double* __restrict__ H, i64 n, double* __restrict__ Wt, i64 m,
                                                              void* __restrict__ P, int S, i64 slab, int kpp, int f64,
                                                              double* __restrict__ Gh, const double* __restrict__ Gw,
                                                              int* __restrict__ fail_flag)
{
    constexpr int KP = 8;
    const double nu0 = sqrt(Gw[0]), nu1 = sqrt(Gw[KP + 1]);
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

// W'W of the NORMALISED W without another pass over W: (D^-1 W'W D^-1)_ij = Gw_ij / (nu_i nu_j).
// Runs after the consumer of the un-normalised Gram matrix (rank2_normalize_kernel) on the stream.
__global__ void rank2_gw_normalize_kernel(double* __restrict__ Gw)
{
    constexpr int KP = 8;
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double nu0 = sqrt(Gw[0]), nu1 = sqrt(Gw[KP + 1]);
    Gw[0] = Gw[0] / (nu0 * nu0);
    Gw[1] = Gw[1] / (nu0 * nu1);
    Gw[KP] = Gw[KP] / (nu0 * nu1);
    Gw[KP + 1] = Gw[KP + 1] / (nu1 * nu1);
}

int launch_rank2_normalize(double* H, i64 n, double* Wt, i64 m, PartialView R, double* Gh, double* Gw, int* fail_flag,
                           hipStream_t st)
{
    const i64 cnt = n > m ? n : m;
    const int grid = (int)((cnt + 255) / 256);
    rank2_normalize_kernel<<<grid, 256, 0, st>>>(H, n, Wt, m, const_cast<void*>(R.p), R.S, R.slab, R.kpp, R.f64, Gh, Gw,
                                                 fail_flag);
    SMK_HIP(hipGetLastError());
    rank2_gw_normalize_kernel<<<1, 64, 0, st>>>(Gw);
    SMK_HIP(hipGetLastError());
    return 0;
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
