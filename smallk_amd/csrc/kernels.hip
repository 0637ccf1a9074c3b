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

namespace smk {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((ext_vector_type(2))) double f64x2_t;

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
template <int EBYTES, int KT, int NSPLIT>
struct BPCfg {
    static constexpr int MB = 64;
    static constexpr int E = 16 / EBYTES;
    static constexpr int CPC = MB / E;      // 16-B chunks per column per stage
    static constexpr int QS = CPC / 2;      // chunk-pair steps per stage
    static constexpr int NB = 128;
    static constexpr int B_BYTES = NB * MB * EBYTES;
    static constexpr int X_BYTES = QS * NSPLIT * KT * 1024;
    static constexpr int STAGE_BYTES = B_BYTES + X_BYTES;
    static constexpr int NSTAGE = 3;
    static constexpr int TI = STAGE_BYTES / 1024;   // wave-level 1-KiB loads per stage
    static constexpr int LPS = TI / 4;              // per wave
    static_assert(TI % 4 == 0, "loads per stage must split evenly over 4 waves");
};

template <int N> __device__ __forceinline__ void wait_vmcnt()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int EBYTES, int KT, int NSPLIT>
__global__ __launch_bounds__(256, 1) void bigprod_kernel(const unsigned char* __restrict__ B, i64 ldb_bytes,
                                                         const unsigned char* __restrict__ Xp,
                                                         double* __restrict__ P, i64 stages, i64 nst,
                                                         i64 tiles, i64 ncols_pad, int S, int logS)
{
    using C = BPCfg<EBYTES, KT, NSPLIT>;
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

    i64 st0 = (i64)split * nst;
    i64 st1 = st0 + nst;
    if (st1 > stages) st1 = stages;
    const int my_nst = (st1 > st0) ? (int)(st1 - st0) : 0;

    // per-lane source offsets for this wave's B loads (constant across stages)
    const i64 col0 = tile * C::NB;
    i64 b_off[C::LPS];     // byte offset of the lane's 16-B chunk relative to stage row 0
    int is_b[C::LPS];
#pragma unroll
    for (int i = 0; i < C::LPS; ++i) {
        const int t = wave + 4 * i;                // wave-level load index within the stage
        if (t * 1024 < C::B_BYTES) {
            const int p = t * 64 + lane;           // chunk position inside the LDS B tile
            const int j = p / C::CPC;
            const int pc = p % C::CPC;
            const int swz = (C::CPC == 8) ? ((j >> 1) & 7) : (j & 15);
            const int lc = pc ^ swz;
            b_off[i] = (col0 + j) * ldb_bytes + (i64)lc * 16;
            is_b[i] = 1;
        } else {
            b_off[i] = (i64)(t * 1024 - C::B_BYTES) + lane * 16;   // offset inside the X stage block
            is_b[i] = 0;
        }
    }

    auto issue = [&](int s_local) {
        const i64 stage = st0 + s_local;
        const int buf = s_local % C::NSTAGE;
        unsigned char* lbase = smem + buf * C::STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < C::LPS; ++i) {
            const int t = wave + 4 * i;
            const unsigned char* g = is_b[i] ? (B + b_off[i] + stage * (C::MB * EBYTES))
                                             : (Xp + stage * C::X_BYTES + b_off[i]);
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)g, (LDS_AS void*)(lbase + t * 1024), 16, 0, 0);
        }
    };

    // Accumulators.  The leading (hi) term is accumulated in fp32 by the MFMA for ONE stage
    // (64 rows = 4 dependent MFMAs) and then added into fp64 running sums by the VALU while the
    // next stage's MFMAs run into the other fp32 set (accA/accB ping-pong): the fp32 rounding
    // chain never exceeds one stage.  The mid/lo split terms are 2^-8 / 2^-16 smaller and stay
    // in fp32 for the whole split.
    f32x16_t accA[KT], accB[KT];
    f32x16_t accs[NSPLIT > 1 ? NSPLIT - 1 : 1][KT];
    double dacc[KT][16];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            accA[kt][r] = 0.f;
            accB[kt][r] = 0.f;
            dacc[kt][r] = 0.0;
#pragma unroll
            for (int s = 0; s < (NSPLIT > 1 ? NSPLIT - 1 : 1); ++s) accs[s][kt][r] = 0.f;
        }
    }

    // fragment read addresses
    const int jl = wave * 32 + (lane & 31);
    const int h = lane >> 5;
    const int swz_r = (C::CPC == 8) ? ((jl >> 1) & 7) : (jl & 15);
    const int bfrag_base = jl * C::CPC * 16;

    auto flush = [&](f32x16_t (&a)[KT]) {
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                dacc[kt][r] += (double)a[kt][r];
                a[kt][r] = 0.f;
            }
    };

    // one stage: wait for its data, release the ring slot two stages ahead, MFMAs into `cur`;
    // the previous stage's fp32 sums (`prev`) are folded into fp64 right after the first step.
    auto stage_body = [&](int t, f32x16_t (&cur)[KT], f32x16_t (&prev)[KT], bool flush_prev) {
        if (t + 1 < my_nst) wait_vmcnt<C::LPS>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (t + 2 < my_nst) issue(t + 2);

        const unsigned char* sb = smem + (t % C::NSTAGE) * C::STAGE_BYTES;
        const unsigned char* sx = sb + C::B_BYTES;
#pragma unroll
        for (int q = 0; q < C::QS; ++q) {
            const int lc = 2 * q + h;
            const u32x4_t braw = *(const u32x4_t*)(sb + bfrag_base + ((lc ^ swz_r) << 4));
            if constexpr (EBYTES == 2) {
                const bf16x8_t bfr = __builtin_bit_cast(bf16x8_t, braw);
#pragma unroll
                for (int s = 0; s < NSPLIT; ++s)
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt) {
                        const u32x4_t araw = *(const u32x4_t*)(sx + ((q * NSPLIT + s) * KT + kt) * 1024 + lane * 16);
                        const bf16x8_t afr = __builtin_bit_cast(bf16x8_t, araw);
                        if (s == 0) cur[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr, bfr, cur[kt], 0, 0, 0);
                        else accs[s - 1][kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr, bfr, accs[s - 1][kt], 0, 0, 0);
                    }
            } else {
                const f32x4_t bfr = __builtin_bit_cast(f32x4_t, braw);
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {
                    const f32x4_t afr = *(const f32x4_t*)(sx + (q * KT + kt) * 1024 + lane * 16);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        cur[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(afr[e], bfr[e], cur[kt], 0, 0, 0);
                }
            }
            if (q == 0 && flush_prev) flush(prev);
        }
    };

    if (my_nst > 0) issue(0);
    if (my_nst > 1) issue(1);

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

    // epilogue: fp64 totals (+ the small split terms), stored k-contiguous as doubles.
    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
    const i64 jg = col0 + jl;
    double* pout = P + ((i64)split * ncols_pad + jg) * (KT * 32);
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int i = 0; i < 4; i += 2) {
                f64x2_t v;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    double tsum = dacc[kt][4 * g + i + u];
                    if constexpr (NSPLIT > 1) {
                        float small = accs[NSPLIT - 2][kt][4 * g + i + u];
#pragma unroll
                        for (int s = NSPLIT - 3; s >= 0; --s) small += accs[s][kt][4 * g + i + u];
                        tsum += (double)small;
                    }
                    v[u] = tsum;
                }
                *(f64x2_t*)(pout + kt * 32 + 8 * g + 4 * h + i) = v;
            }
        }
    }
}

// ---- packing of the skinny operand -------------------------------------------------
// out layout: [q][s][kt][lane = (r, h)][16 B], chunk = 2q + h covers rows chunk*E .. +E-1,
// r = k index inside tile kt.  bf16: hi = bf16(x), mid = bf16(x-hi), lo = bf16(x-hi-mid).
template <int EBYTES, int NSPLIT>
__global__ __launch_bounds__(256) void pack_kernel(const double* __restrict__ X, int k, i64 N, int KT, i64 nq,
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
        v[e] = (row < N && r < k) ? X[row * k + r] : 0.0;
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

static inline i64 pack_nq(int storage, i64 N)
{
    const i64 MB = 64;
    const i64 E = storage == STORE_BF16 ? 8 : 4;
    i64 stages = (N + MB - 1) / MB;
    return stages * (MB / E / 2);
}

size_t packed_bytes(int storage, int k, i64 N, int nsplit)
{
    if (storage != STORE_BF16) nsplit = 1;
    return (size_t)pack_nq(storage, N) * nsplit * kt_of(k) * 1024;
}

int launch_pack(const double* X, int k, i64 N, int storage, int nsplit, void* out, hipStream_t st)
{
    const int KT = kt_of(k);
    const i64 nq = pack_nq(storage, N);
    const i64 threads = nq * KT * 64;
    const int grid = (int)((threads + 255) / 256);
    if (grid == 0) return 0;
    if (storage == STORE_BF16) {
        if (nsplit == 3) pack_kernel<2, 3><<<grid, 256, 0, st>>>(X, k, N, KT, nq, (unsigned char*)out);
        else if (nsplit == 2) pack_kernel<2, 2><<<grid, 256, 0, st>>>(X, k, N, KT, nq, (unsigned char*)out);
        else pack_kernel<2, 1><<<grid, 256, 0, st>>>(X, k, N, KT, nq, (unsigned char*)out);
    } else {
        pack_kernel<4, 1><<<grid, 256, 0, st>>>(X, k, N, KT, nq, (unsigned char*)out);
    }
    SMK_HIP(hipGetLastError());
    return 0;
}

BigProdPlan plan_bigprod(int storage, int k, i64 len, i64 ncols, int nsplit, int num_cus)
{
    BigProdPlan pl;
    pl.storage = storage;
    pl.kt = kt_of(k);
    pl.nsplit = storage == STORE_BF16 ? nsplit : 1;
    pl.stages = (len + 63) / 64;
    pl.tiles = (ncols + 127) / 128;
    pl.ncols_pad = pl.tiles * 128;
    // enough workgroups for >= 4 rounds over the CUs, but keep >= 8 stages per split
    int S = 1;
    while (pl.tiles * S < 4 * (i64)num_cus && S < 64 && pl.stages / (2 * S) >= 8) S *= 2;
    pl.S = S;
    pl.nst = (pl.stages + S - 1) / S;
    pl.p_elems = (size_t)S * pl.ncols_pad * pl.kt * 32;   // doubles
    return pl;
}

template <int EBYTES, int KT, int NSPLIT>
static int launch_bigprod_t(const BigProdPlan& pl, const void* B, i64 ldb, const void* Xp, double* P, hipStream_t st)
{
    using C = BPCfg<EBYTES, KT, NSPLIT>;
    constexpr int lds = C::STAGE_BYTES * C::NSTAGE;
    static bool attr_set = false;
    auto kern = bigprod_kernel<EBYTES, KT, NSPLIT>;
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
    kern<<<(unsigned)grid, 256, lds, st>>>((const unsigned char*)B, ldb * EBYTES, (const unsigned char*)Xp, P,
                                           pl.stages, pl.nst, pl.tiles, pl.ncols_pad, pl.S, logS);
    SMK_HIP(hipGetLastError());
    return 0;
}

int launch_bigprod(const BigProdPlan& pl, const void* B, i64 ldb, const void* Xp, double* P, hipStream_t st)
{
    if (pl.storage == STORE_BF16) {
        if (pl.kt == 1) {
            if (pl.nsplit == 3) return launch_bigprod_t<2, 1, 3>(pl, B, ldb, Xp, P, st);
            if (pl.nsplit == 2) return launch_bigprod_t<2, 1, 2>(pl, B, ldb, Xp, P, st);
            return launch_bigprod_t<2, 1, 1>(pl, B, ldb, Xp, P, st);
        } else {
            if (pl.nsplit == 3) return launch_bigprod_t<2, 2, 3>(pl, B, ldb, Xp, P, st);
            if (pl.nsplit == 2) return launch_bigprod_t<2, 2, 2>(pl, B, ldb, Xp, P, st);
            return launch_bigprod_t<2, 2, 1>(pl, B, ldb, Xp, P, st);
        }
    } else {
        if (pl.kt == 1) return launch_bigprod_t<4, 1, 1>(pl, B, ldb, Xp, P, st);
        return launch_bigprod_t<4, 2, 1>(pl, B, ldb, Xp, P, st);
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
// Gram matrix  G(KP x KP, ld KP) = X X'   (X: k x N fp64), deterministic two stage
// ==========================================================================
template <int KP>
__global__ __launch_bounds__(256) void gram_partial_kernel(const double* __restrict__ X, int k, i64 N,
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
        // coalesced: the chunk is nc*k contiguous doubles
        for (int idx = tid; idx < CB * KP; idx += 256) {
            const int cc = idx / KP, r = idx % KP;
            double v = 0.0;
            if (cc < nc && r < k) v = X[(c0 + cc) * k + r];
            xs[cc][r] = v;
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

__global__ __launch_bounds__(256) void gram_reduce_kernel(const double* __restrict__ Gp, int nblk, int elems,
                                                          double* __restrict__ G)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= elems) return;
    double s = 0.0;
    for (int b = 0; b < nblk; ++b) s += Gp[(i64)b * elems + i];
    G[i] = s;
}

size_t gram_scratch_elems(int k, int max_blocks)
{
    int KP = kp_of(k);
    return (size_t)max_blocks * KP * KP;
}

int launch_gram(const double* X, int k, i64 N, double* G, double* scratch, int max_blocks, hipStream_t st)
{
    const int KP = kp_of(k);
    int nblk = (int)((N + 127) / 128);
    if (nblk > max_blocks) nblk = max_blocks;
    if (nblk < 1) nblk = 1;
    i64 cpb = (N + nblk - 1) / nblk;
    cpb = (cpb + 31) / 32 * 32;
    nblk = (int)((N + cpb - 1) / cpb);
    if (nblk < 1) nblk = 1;
    switch (KP) {
        case 8: gram_partial_kernel<8><<<nblk, 256, 0, st>>>(X, k, N, cpb, scratch); break;
        case 16: gram_partial_kernel<16><<<nblk, 256, 0, st>>>(X, k, N, cpb, scratch); break;
        case 32: gram_partial_kernel<32><<<nblk, 256, 0, st>>>(X, k, N, cpb, scratch); break;
        default: gram_partial_kernel<64><<<nblk, 256, 0, st>>>(X, k, N, cpb, scratch); break;
    }
    SMK_HIP(hipGetLastError());
    const int elems = KP * KP;
    gram_reduce_kernel<<<(elems + 255) / 256, 256, 0, st>>>(scratch, nblk, elems, G);
    SMK_HIP(hipGetLastError());
    return 0;
}

// ==========================================================================
// column-per-thread helpers
// ==========================================================================
template <int KP>
__device__ __forceinline__ void load_col(const double* __restrict__ X, int k, i64 j, double (&x)[KP])
{
    const double* p = X + j * k;
#pragma unroll
    for (int r = 0; r < KP; ++r) x[r] = (r < k) ? p[r] : 0.0;
}

template <int KP>
__device__ __forceinline__ void load_rhs(const PartialView& R, int k, i64 j, double (&b)[KP])
{
#pragma unroll
    for (int r = 0; r < KP; ++r) b[r] = 0.0;
    if (R.f64) {
        for (int s = 0; s < R.S; ++s) {
            const double* p = (const double*)R.p + s * R.slab + j * R.kpp;
#pragma unroll
            for (int r = 0; r < KP; ++r)
                if (r < k) b[r] += p[r];
        }
    } else {
        for (int s = 0; s < R.S; ++s) {
            const float* p = (const float*)R.p + s * R.slab + j * R.kpp;
#pragma unroll
            for (int r = 0; r < KP; ++r)
                if (r < k) b[r] += (double)p[r];
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

// ---- MU: x <- x .* R ./ (G x + 1e-13)      (nmf_solver_mu.hpp:22, :27-71) -------------
template <int KP>
__global__ __launch_bounds__(256) void mu_update_kernel(double* __restrict__ X, int k, i64 N, PartialView R,
                                                        const double* __restrict__ G)
{
    __shared__ double gs[KP * KP];
    for (int i = threadIdx.x; i < KP * KP; i += blockDim.x) gs[i] = G[i];
    __syncthreads();
    const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= N) return;
    double x[KP], b[KP];
    load_col<KP>(X, k, j, x);
    load_rhs<KP>(R, k, j, b);
    double* px = X + j * k;
#pragma unroll
    for (int r = 0; r < KP; ++r) {
        if (r < k) {
            double acc = 0.0;
#pragma unroll
            for (int q = 0; q < KP; ++q) acc += gs[q * KP + r] * x[q];
            px[r] = x[r] * (b[r] / (acc + 1.0e-13));
        }
    }
}

// ---- HALS H sweep: rows r = 0..k-1 in order, Gauss-Seidel inside the column ----------
//      (nmf_solver_hals.hpp:26-62)
template <int KP>
__global__ __launch_bounds__(256) void hals_sweep_kernel(double* __restrict__ X, int k, i64 N, PartialView R,
                                                         const double* __restrict__ G)
{
    __shared__ double gs[KP * KP];
    for (int i = threadIdx.x; i < KP * KP; i += blockDim.x) gs[i] = G[i];
    __syncthreads();
    const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= N) return;
    double x[KP], b[KP];
    load_col<KP>(X, k, j, x);
    load_rhs<KP>(R, k, j, b);
#pragma unroll
    for (int r = 0; r < KP; ++r) {
        if (r < k) {
            double acc = 0.0;
#pragma unroll
            for (int q = 0; q < KP; ++q) acc += gs[q * KP + r] * x[q];
            double v = x[r] + (b[r] - acc) / gs[r * KP + r];
            if (isnan(v) || v < 0.0) v = 0.0;
            x[r] = v;
        }
    }
    double* px = X + j * k;
#pragma unroll
    for (int r = 0; r < KP; ++r)
        if (r < k) px[r] = x[r];
}

// ---- gradient g = G x - R, projected-gradient partial sums ---------------------------
//      (mu :156-161, hals :181-195, bpp :370-371; projected_gradient.hpp:125-171)
template <int KP>
__global__ __launch_bounds__(256) void grad_pg_kernel(const double* __restrict__ X, int k, i64 N, PartialView R,
                                                      const double* __restrict__ G, double* __restrict__ grad_out,
                                                      double* __restrict__ partials)
{
    __shared__ double gs[KP * KP];
    __shared__ double sh[16];
    for (int i = threadIdx.x; i < KP * KP; i += blockDim.x) gs[i] = G[i];
    __syncthreads();
    const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    double sum = 0.0;
    if (j < N) {
        double x[KP], b[KP];
        load_col<KP>(X, k, j, x);
        load_rhs<KP>(R, k, j, b);
#pragma unroll
        for (int r = 0; r < KP; ++r) {
            if (r < k) {
                double acc = 0.0;
#pragma unroll
                for (int q = 0; q < KP; ++q) acc += gs[q * KP + r] * x[q];
                const double g = acc - b[r];
                if (grad_out) grad_out[j * k + r] = g;
                if (g < 0.0 || x[r] > 0.0) sum += g * g;
            }
        }
    }
    const double t = block_sum(sum, sh);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
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

int launch_mu_update(double* X, int k, i64 N, PartialView R, const double* G, hipStream_t st)
{
    const int grid = (int)((N + 255) / 256);
    KP_DISPATCH(kp_of(k), (mu_update_kernel<KP><<<grid, 256, 0, st>>>(X, k, N, R, G)));
    SMK_HIP(hipGetLastError());
    return 0;
}

int launch_hals_sweep(double* X, int k, i64 N, PartialView R, const double* G, hipStream_t st)
{
    const int grid = (int)((N + 255) / 256);
    KP_DISPATCH(kp_of(k), (hals_sweep_kernel<KP><<<grid, 256, 0, st>>>(X, k, N, R, G)));
    SMK_HIP(hipGetLastError());
    return 0;
}

int launch_grad_pg(const double* X, int k, i64 N, PartialView R, const double* G, double* grad_out,
                   double* pg_partials, double* pg_accum, int slot, hipStream_t st)
{
    const int grid = (int)((N + 255) / 256);
    KP_DISPATCH(kp_of(k), (grad_pg_kernel<KP><<<grid, 256, 0, st>>>(X, k, N, R, G, grad_out, pg_partials)));
    SMK_HIP(hipGetLastError());
    sum_partials_kernel<<<1, 256, 0, st>>>(pg_partials, grid, pg_accum + slot);
    SMK_HIP(hipGetLastError());
    return 0;
}

int launch_pg_from_grad(const double* X, const double* Y, int k, i64 N, double* pg_partials, double* pg_accum,
                        int slot, hipStream_t st)
{
    const i64 count = N * k;
    int grid = (int)((count + 255) / 256 < 1024 ? (count + 255) / 256 : 1024);
    if (grid < 1) grid = 1;
    pg_from_grad_kernel<<<grid, 256, 0, st>>>(X, Y, count, pg_partials);
    SMK_HIP(hipGetLastError());
    sum_partials_kernel<<<1, 256, 0, st>>>(pg_partials, grid, pg_accum + slot);
    SMK_HIP(hipGetLastError());
    return 0;
}

// ==========================================================================
// HALS W update (nmf_solver_hals.hpp:66-117) on Wt (k x M): one launch per
// column c (k sequential grid-wide reductions are inherent: column c's L2 norm
// feeds every later column).  Kernel c first applies the pending normalisation
// of column c-1, then updates column c un-normalised and emits per-block
// partial sums of squares / zero counts; kernel boundaries are the grid sync.
//   scratch: ss[k][nblk], nz[k][nblk]
// ==========================================================================
template <int KP>
__global__ __launch_bounds__(256) void hals_w_col_kernel(double* __restrict__ Wt, int k, i64 M, PartialView R,
                                                         const double* __restrict__ G, int c, int nblk,
                                                         double* __restrict__ ss, double* __restrict__ nz)
{
    __shared__ double sh[16];
    __shared__ double gcol[KP];
    if (threadIdx.x < KP) gcol[threadIdx.x] = (c < k) ? G[c * KP + threadIdx.x] : 0.0;
    double scale_prev = 1.0, fill_prev = -1.0;
    if (c > 0) {
        const double s2 = block_sum_array(ss + (i64)(c - 1) * nblk, nblk, sh);
        const double zc = block_sum_array(nz + (i64)(c - 1) * nblk, nblk, sh);
        if (zc >= (double)M) {                      // all-zero column guard (:105-111)
            const double eps = DBL_EPSILON;
            const double nrm = sqrt((double)M * eps * eps);
            fill_prev = eps * (1.0 / nrm);
        } else {
            scale_prev = 1.0 / sqrt(s2);
        }
    }
    __syncthreads();
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    double v2 = 0.0, zero = 0.0;
    if (i < M) {
        double w[KP];
        load_col<KP>(Wt, k, i, w);
        double* pw = Wt + i * k;
        if (c > 0) {
            double wp = 0.0;
#pragma unroll
            for (int r = 0; r < KP; ++r)
                if (r == c - 1) wp = w[r];
            wp = (fill_prev >= 0.0) ? fill_prev : wp * scale_prev;
#pragma unroll
            for (int r = 0; r < KP; ++r)
                if (r == c - 1) w[r] = wp;
            pw[c - 1] = wp;
        }
        if (c < k) {
            double acc = 0.0, wc = 0.0;
#pragma unroll
            for (int r = 0; r < KP; ++r) {
                acc += w[r] * gcol[r];
                if (r == c) wc = w[r];
            }
            const double rhs = rhs_elem(R, i, c);
            double v = wc + (rhs - acc) / gcol[c];
            if (isnan(v) || v < 0.0) { v = 0.0; zero = 1.0; }
            pw[c] = v;
            v2 = v * v;
        }
    }
    if (c < k) {
        const double t2 = block_sum(v2, sh);
        const double tz = block_sum(zero, sh);
        if (threadIdx.x == 0) {
            ss[(i64)c * nblk + blockIdx.x] = t2;
            nz[(i64)c * nblk + blockIdx.x] = tz;
        }
    }
}

size_t hals_w_scratch_elems(int k, i64 M)
{
    i64 nblk = (M + 255) / 256;
    return (size_t)(2 * (i64)k * nblk);
}

int launch_hals_w_update(double* Wt, int k, i64 M, PartialView R, const double* G, double* scratch, hipStream_t st)
{
    const int nblk = (int)((M + 255) / 256);
    double* ss = scratch;
    double* nz = scratch + (i64)k * nblk;
    const int KPv = kp_of(k);
    for (int c = 0; c <= k; ++c) {
        KP_DISPATCH(KPv, (hals_w_col_kernel<KP><<<nblk, 256, 0, st>>>(Wt, k, M, R, G, c, nblk, ss, nz)));
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

template <int KP>
__global__ __launch_bounds__(256) void nnls_bpp_kernel(double* __restrict__ X, double* __restrict__ Y, int k, i64 N,
                                                       PartialView R, const double* __restrict__ G,
                                                       int* __restrict__ fail_flag, int iter_tag)
{
    constexpr int GS = KP;
    constexpr int GPB = 256 / GS;                   // column groups per block
    __shared__ double gs[KP * KP];                  // gs[c*KP + i] = G[i][c] (symmetric)
    for (int t = threadIdx.x; t < KP * KP; t += blockDim.x) gs[t] = G[t];
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int i = threadIdx.x % GS;                 // component owned by this lane
    const i64 col = (i64)blockIdx.x * GPB + threadIdx.x / GS;
    const bool col_ok = col < N;
    const bool comp_ok = i < k;
    const i64 cc = col_ok ? col : (N - 1);

    double rhs = 0.0, x = 0.0, y = 0.0;
    if (comp_ok) {
        rhs = rhs_elem(R, cc, i);
        x = X[cc * k + i];
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
        X[col * k + i] = x;
        if (Y) Y[col * k + i] = y;
    }
    if (failed && col_ok) atomicMin(fail_flag, iter_tag);
}

int launch_nnls_bpp(double* X, double* Y, int k, i64 N, PartialView R, const double* G, int* fail_flag,
                    int iter_tag, hipStream_t st)
{
    const int KPv = kp_of(k);
    const int gpb = 256 / KPv;
    const int grid = (int)((N + gpb - 1) / gpb);
    KP_DISPATCH(KPv, (nnls_bpp_kernel<KP><<<grid, 256, 0, st>>>(X, Y, k, N, R, G, fail_flag, iter_tag)));
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
    const i64 total = N * k;
    for (i64 idx = (i64)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (i64)gridDim.x * blockDim.x) {
        const int r = (int)(idx % k);
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
    const i64 total = N * k;
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
