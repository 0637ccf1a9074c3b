// The Gram inverse of block pivoting at k in (16, 64] as a device function of ONE 256-thread workgroup: launched by itself
// (nnls.hip: gram_inverse64_kernel) or riding as workgroup 0 of the sparse product that follows the Gram matrix in every BPP
// schedule (spmm_seg.hip: InvRide) -- the inverse is needed one launch later, and a second stream with its two event hops costs a
// small problem more than the 13 us it hides (the Reuters shape: 125 us per iteration beside the product, 118 in stream order,
// 100 riding).  Reference: the normal-equation solves of nnls.hpp:144-244.
#pragma once
#include <type_traits>

#include "common.h"
#include "devutil.h"

namespace smk {

#define GRAM_INVERSE_LDS(KP) (5 * (KP) + 2)

// KP = 64: the same elimination with the pivot index split as j = 16 j0 + JE and the sixteen values of JE unrolled, so that the
// one entry of a thread's row segment that belongs to the pivot COLUMN is a compile-time register.  With a run-time index
// the compiler kept a[] in scratch memory -- sixteen scratch stores and a dependent scratch load per pivot step, 1.3 - 1.8 us
// each: 113 us for k = 64, half of a block-pivoting iteration on an 8192 x 4096 matrix; now 47 us (k = 40: 83 -> 37 us;
// profiles/r04_gram_inverse.txt).  The division is not what is left (rcp + Newton steps instead: 47.2 us): a step is an LDS
// round trip, sixteen multiply-adds, the publication of the next row / column and a barrier.
// Same operations in the same order as gram_inverse_kernel<64>: the result is bit-identical.
template <int KP>
__device__ __forceinline__ void gram_inverse64_body(const double* __restrict__ G, int k, double* __restrict__ Ginv,
                                                    int* __restrict__ status, double* __restrict__ lds /* GRAM_INVERSE_LDS(KP) doubles, 16-byte aligned */)
{
    static_assert(KP == 64 || KP == 32, "256 threads hold the matrix as KP rows x (256 / KP) segments");
    constexpr int CQ = 256 / KP, EPT = KP / CQ;               // KP = 64: 4 segments of 16 entries; KP = 32: 8 segments of 4
    double* const rowj = lds;                   // [2][KP]
    double* const colj = lds + 2 * KP;          // [2][KP]
    double* const diag0 = lds + 4 * KP;         // [KP]
    int& bad = *(int*)(lds + 5 * KP);
    const int tid = threadIdx.x;
    const int r = tid / CQ, cq = tid % CQ;
    if (tid == 0) bad = 0;
    if (tid < KP) diag0[tid] = (tid < k) ? G[tid * KP + tid] : 1.0;
    double a[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int c = cq * EPT + e;
        a[e] = (r < k && c < k) ? G[c * KP + r] : ((r == c) ? 1.0 : 0.0);
    }
    if (r == 0) {
#pragma unroll
        for (int e = 0; e < EPT; ++e) rowj[cq * EPT + e] = a[e];
    }
    if (cq == 0) colj[r] = a[0];
    __syncthreads();
    auto step = [&](int j, auto je_tag) {
        constexpr int JE = decltype(je_tag)::value;
        constexpr int JN = (JE + 1) % EPT;
        const int buf = j & 1;
        const double piv = rowj[buf * KP + j];
        if (tid == 0 && !(piv > 1.0e-9 * diag0[j])) bad = 1;
        const double ip = 1.0 / piv;
        const double f = colj[buf * KP + r] * ip;
        const bool my_row = r == j, my_colq = cq == j / EPT;
#pragma unroll
        for (int e = 0; e < EPT; e += 2) {
            const f64x2_t rr = *(const f64x2_t*)&rowj[buf * KP + cq * EPT + e];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                double val = my_row ? rr[u] * ip : __builtin_fma(-f, rr[u], a[e + u]);
                if (e + u == JE) {                              // the entry in the pivot column (for the threads of that column block)
                    const double on_col = my_row ? ip : -f;
                    val = my_colq ? on_col : val;
                }
                a[e + u] = val;
            }
        }
        if (j + 1 < k) {                                        // row and column j + 1 of the updated matrix -> the other buffer
            const int nb = buf ^ 1;
            if (r == j + 1) {
#pragma unroll
                for (int e = 0; e < EPT; ++e) rowj[nb * KP + cq * EPT + e] = a[e];
            }
            if (cq == (j + 1) / EPT) colj[nb * KP + r] = a[JN];
        }
        __syncthreads();
    };
    for (int j0 = 0; j0 < k; j0 += EPT) {                       // k is uniform: every thread takes the same steps (barriers inside)
#define SMK_GI_STEP(n) if constexpr (n < EPT) { if (j0 + n < k) step(j0 + n, std::integral_constant<int, n>{}); }
        SMK_GI_STEP(0) SMK_GI_STEP(1) SMK_GI_STEP(2) SMK_GI_STEP(3) SMK_GI_STEP(4) SMK_GI_STEP(5) SMK_GI_STEP(6) SMK_GI_STEP(7)
        SMK_GI_STEP(8) SMK_GI_STEP(9) SMK_GI_STEP(10) SMK_GI_STEP(11) SMK_GI_STEP(12) SMK_GI_STEP(13) SMK_GI_STEP(14) SMK_GI_STEP(15)
#undef SMK_GI_STEP
    }
    // symmetrise: write the matrix out, then average each entry with its transposed partner
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int c = cq * EPT + e;
        Ginv[c * KP + r] = (r < k && c < k) ? a[e] : 0.0;
    }
    __syncthreads();
    __threadfence_block();
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int c = cq * EPT + e;
        const double up = Ginv[c * KP + r], lo = Ginv[r * KP + c];
        a[e] = 0.5 * (up + lo);
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < EPT; ++e) Ginv[(cq * EPT + e) * KP + r] = a[e];
    if (tid == 0) *status = bad ? 0 : 1;
}


}  // namespace smk
