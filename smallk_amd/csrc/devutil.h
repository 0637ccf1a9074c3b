// smallk_amd/csrc/devutil.h -- device-side helpers shared by the .hip translation units
// (vector types, bf16 rounding, DPP / readlane reductions, access to the partial products).
#pragma once
#include "common.h"
#include <cfloat>
#include <cstdlib>
#include <utility>

namespace smk {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
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
    if constexpr (LPC >= 32) v += __shfl_xor(v, 16, 64);   // the neighbouring DPP row (LDS crossbar; KP = 128 only)
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
    if (R.kpp - e0 < 4) {                 // compact rank-2 products (kpp = 2): only the entries that exist
        for (int e = 0; e0 + e < R.kpp; ++e)
            for (int s = 0; s < R.S; ++s)
                b[e] += R.f64 ? ((const double*)R.p)[s * R.slab + j * R.kpp + e0 + e] : (double)((const float*)R.p)[s * R.slab + j * R.kpp + e0 + e];
        return;
    }
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

// one element (row i of column j) of the summed partial product.
// The slabs are read in batches whose loads are all in flight together: a loop of `v += p[s * slab]` waits for every load
// before it issues the next (the adds are ordered), i.e. S dependent round trips to memory per element -- 16 on the H side of
// a 1/8 column shard of C4, where that chain WAS the NNLS launch (51 us for 2 columns per wave, profiles/r05_c4_planted_rank0_of_8_*).
// Indices are clamped instead of predicated so that a batch is branch-free; the adds are predicated (sum in slab order).
struct RhsPending { double t[4]; double tail; };

// issue the loads of element (i, j): the first four slabs stay un-summed in registers (no arithmetic, hence no wait: the caller
// can compute on something else until rhs_finish), slabs 4 .. S-1 are summed here, eight loads at a time
__device__ __forceinline__ void rhs_issue(const PartialView& R, i64 j, int i, RhsPending& q)
{
    q.tail = 0.0;
    const i64 off = j * R.kpp + i;
    if (R.f64) {
        const double* p = (const double*)R.p + off;
        q.t[0] = p[0];
        if (R.S == 1) { q.t[1] = q.t[2] = q.t[3] = 0.0; return; }
#pragma unroll
        for (int u = 1; u < 4; ++u) q.t[u] = p[(i64)(u < R.S ? u : 0) * R.slab];
        for (int s0 = 4; s0 < R.S; s0 += 8) {
            double t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = p[(i64)(s0 + u < R.S ? s0 + u : R.S - 1) * R.slab];
#pragma unroll
            for (int u = 0; u < 8; ++u) if (s0 + u < R.S) q.tail += t[u];
        }
    } else {            // fp32 on the wire (SMK_COMM_F64=0): summed at once
        const float* p = (const float*)R.p + off;
        double v = 0.0;
        for (int s = 0; s < R.S; ++s) v += (double)p[(i64)s * R.slab];
        q.t[0] = v;
        q.t[1] = q.t[2] = q.t[3] = 0.0;
    }
}
__device__ __forceinline__ double rhs_finish(const PartialView& R, const RhsPending& q)
{
    double v = q.t[0];
    if (R.f64 && R.S > 1) {
        v += q.t[1];
        if (R.S > 2) v += q.t[2];
        if (R.S > 3) v += q.t[3];
        if (R.S > 4) v += q.tail;
    }
    return v;
}
__device__ __forceinline__ double rhs_elem(const PartialView& R, i64 j, int i)
{
    RhsPending q;
    rhs_issue(R, j, i, q);
    return rhs_finish(R, q);
}

// 1 / x to full precision: v_rcp_f64 + two Newton steps (the block-pivoting kernels divide by pivots)
__device__ __forceinline__ double fast_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-x, r, 1.0);
    return __builtin_fma(r, e, r);
}
// diagnostics of the block-pivoting kernels (nnls.hip: nnls_stats_ptr)
__device__ __forceinline__ void nnls_stat(unsigned long long* stats, int slot) { atomicAdd(&stats[slot], 1ull); }

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

#define KP_DISPATCH(KPV, CALL)              \
    switch (KPV) {                          \
        case 8: { constexpr int KP = 8; CALL; } break;   \
        case 16: { constexpr int KP = 16; CALL; } break; \
        case 32: { constexpr int KP = 32; CALL; } break; \
        default: { constexpr int KP = 64; CALL; } break; \
    }
// kernels that also exist for k in (64, 128]
#define KP_DISPATCH128(KPV, CALL)           \
    switch (KPV) {                          \
        case 8: { constexpr int KP = 8; CALL; } break;   \
        case 16: { constexpr int KP = 16; CALL; } break; \
        case 32: { constexpr int KP = 32; CALL; } break; \
        case 64: { constexpr int KP = 64; CALL; } break; \
        default: { constexpr int KP = 128; CALL; } break; \
    }

}  // namespace smk
