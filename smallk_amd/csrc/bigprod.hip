// smallk_amd/csrc/bigprod.hip -- the streaming products W'A and H*At (the dominant kernels) and the
// packing of their skinny operand.  See kernels.hip for the kernel map.
#include "devutil.h"
#include "gram_inverse.h"
#include <type_traits>

namespace smk {

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

// TRB ("transposed source", bf16; the fp32 counterpart is bigprod_f3_kernel with TAIL = 2): the SAME product with B = A' taken from A itself, so that MU and HALS need no stored
// transpose (the reference's MU / HALS call Gemm(NORMAL, TRANSPOSE) on A, nmf_solver_mu.hpp:121-164, nmf_solver_hals.hpp:166-199;
// only its BPP keeps At).  B then points at A (m x n column-major), the tile's 128 "columns" are 128 consecutive ROWS of A (the
// contiguous direction) and a stage is 64 COLUMNS of A: the contraction runs down the strided direction.  A stage is fetched as
// 1-KiB pieces of [8 columns][64 rows] -- 128 contiguous bytes per column, full lines, fetched by 8 adjacent lanes -- and the
// MFMA operand (8 contraction-consecutive values per lane) comes out of LDS through ds_read_b64_tr_b16, the gfx950
// transposing read: the 16 lanes of a group hand in the four quarters of four 32-byte rows and each receives one column of
// that 4 x 16 block.  A half-wave reads 4 rows x 64 bytes, laid out (half-swapped rows) on one whole bank row: no conflicts.
// Everything else (X operand, ring, splits, epilogue) is unchanged.
template <int EBYTES, int KT, int NSPLIT, int MB, int NSTAGE, int CW, int WK, int NWL, bool TRB = false>
__global__ __launch_bounds__(64 * (4 * WK + NWL), 1) void bigprod_kernel(const unsigned char* __restrict__ B, i64 ldb_bytes,
                                                         const unsigned char* __restrict__ Xp,
                                                         double* __restrict__ P, i64 stages, i64 nst,
                                                         i64 tiles, i64 ncols_pad, int S, int logS, int pstride, int accum,
                                                         InvRide ride)
{
    using C = BPCfg<EBYTES, KT, NSPLIT, MB, NSTAGE, CW, WK, NWL>;
    constexpr int KTW = C::KTW;
    static_assert(!TRB || (EBYTES == 2 && CW == 1 && MB % 16 == 0), "transposed source: bf16, 128-row tiles");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    // ---- XCD-aware block -> (tile, split): all blocks of one split share an XCD's L2
    // (block b is dispatched to XCD b % 8; used for speed only).
    int bid = blockIdx.x;
    if constexpr (!TRB) {
        // the Gram inverse of the next block-pivoting launch rides along (common.h: InvRide): eight more workgroups in front (the
        // tile mapping keeps its XCDs), the first four waves of the first one invert, in the launch's own LDS
        static_assert(C::STAGE_BYTES * C::NSTAGE >= GRAM_INVERSE_LDS(64) * 8, "the inversion needs 2.6 KB of LDS");
        if (ride.G) {
            if (bid < 8) {
                if (bid == 0 && threadIdx.x < 256) {
                    if (ride.k <= 32) gram_inverse64_body<32>(ride.G, ride.k, ride.Ginv, (int*)(ride.Ginv + 32 * 32), (double*)smem);
                    else gram_inverse64_body<64>(ride.G, ride.k, ride.Ginv, (int*)(ride.Ginv + 64 * 64), (double*)smem);
                }
                return;
            }
            bid -= 8;
        }
    }
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
            if constexpr (TRB) {
                // piece t: columns 8 (t / 2) .. + 7 of the stage, rows 64 (t % 2) .. + 63 of the tile.  Eight ADJACENT lanes fetch the
                // 128 contiguous bytes of one column (lane = 8 jj + y: with the two 64-byte halves on lanes 32 apart the pass ran
                // 11 - 18 % below the stored-transpose pass); the LDS row of column jj holds its eight 16-byte chunks x = y ^ 4 (jj / 2 % 2):
                // the 64-byte halves of rows 2, 3, 6, 7 are swapped so that the four rows a transposing read touches (pitch 128 B)
                // fall on four different quarters of the 256-byte bank row
                const int jj = lane >> 3, x = (lane & 7) ^ (((jj >> 1) & 1) << 2);
                src_off[i] = (i64)((t >> 1) * 8 + jj) * ldb_bytes + (col0 + (t & 1) * 64 + x * 8) * 2;
            } else {
            const int p = t * 64 + lane;           // chunk position inside the LDS B tile
            const int j = p / C::CPC;
            const int pc = p % C::CPC;
            const int swz = (j >> C::SWZ_SH) & C::SWZ_MASK;
            const int lc = pc ^ swz;
            src_off[i] = (col0 + j) * ldb_bytes + (i64)lc * 16;
            }
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
            const unsigned char* g = is_b[i] ? (B + src_off[i] + (TRB ? stage * C::MB * ldb_bytes : stage * (C::MB * EBYTES)))
                                             : (Xp + stage * C::X_BYTES + src_off[i]);
            // B is streamed once: non-temporal policy (aux = 2) keeps it from displacing the X slice in L2
            if (is_b[i]) {
                if (accum & 2) __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)g, (LDS_AS void*)(lbase + t * 1024), 16, 0, 0);        // a matrix that fits the Infinity Cache: keep it there
                else __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)g, (LDS_AS void*)(lbase + t * 1024), 16, 0, NT_AUX);
            } else
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
    // transposed source: this wave's 32 rows are sub-tile (cwv & 1) of the pieces with t % 2 == cwv / 2; lane half h takes the
    // columns 8 h .. 8 h + 7 of a 16-column MFMA step = piece pair 2 q + h; inside a 16-lane group lane p hands in row p / 4,
    // quarter p % 4 of the group's 4 x 16 block and receives column p
    const int tr_row = (lane & 15) >> 2;                                           // row of the 4 x 16 block this lane hands in (+ 4 in the second read)
    const int tr_chunk = ((cwv & 1) * 4 + ((lane >> 4) & 1) * 2 + ((lane & 3) >> 1)) ^ (((tr_row >> 1) & 1) << 2);
    const int tr_base = (h * 2 + (cwv >> 1)) * 1024 + tr_row * 128 + tr_chunk * 16 + (lane & 1) * 8;
    auto tr_frag = [&](const unsigned char* sb, int q) -> u32x4_t {
        typedef short s16x4_t __attribute__((ext_vector_type(4)));
        const unsigned char* a = sb + tr_base + q * 4096;
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4_t*)(a));
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4_t*)(a + 512));     // rows 4 .. 7: the same swap pattern
        typedef short s16x8_t __attribute__((ext_vector_type(8)));
        const s16x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(u32x4_t, v);
    };
    (void)tr_base;

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
                    for (int c = 0; c < CW; ++c) {
                        if constexpr (TRB) bq[0][c] = tr_frag(sb, 0);
                        else bq[0][c] = *(const u32x4_t*)(sb + bfrag_base[c] + ((lc ^ swz_r[c]) << 4));
                    }
#pragma unroll
                    for (int s = 0; s < NSPLIT; ++s)
#pragma unroll
                        for (int kt = 0; kt < KTW; ++kt)
                            aq[0][s][kt] = *(const u32x4_t*)(sx + ((0 * NSPLIT + s) * KT + kw * KTW + kt) * 1024 + lane * 16);
                }
                if (q + 1 < C::QS) {
                    const int lcn = 2 * (q + 1) + h;
#pragma unroll
                    for (int c = 0; c < CW; ++c) {
                        if constexpr (TRB) bq[(q + 1) & 1][c] = tr_frag(sb, q + 1);
                        else bq[(q + 1) & 1][c] = *(const u32x4_t*)(sb + bfrag_base[c] + ((lcn ^ swz_r[c]) << 4));
                    }
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
        double* pout = P + ((i64)split * ncols_pad + jg) * pstride + kw * KTW * 32;
        const int prows = pstride < 32 ? pstride : 1 << 30;          // rows of this group a column of P has room for
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
                    if (kw * KTW * 32 + kt * 32 + 8 * g >= prows) continue;      // k <= 16: a column of P holds 8 / 16 rows, not 32 (kpp_of)
                    if (accum & 1) v += *(const f64x2_t*)(pout + kt * 32 + 8 * g + 4 * h + i);   // a later row chunk of the same product
                    *(f64x2_t*)(pout + kt * 32 + 8 * g + 4 * h + i) = v;
                }
            }
        }
    }
}

// ---- packing of the skinny operand -------------------------------------------------
// Destination of chunk pair q.  Identity, or -- a row-sharded W (solver.cpp) -- the source holds this rank's blocks back
// to back (block j = rows [j blk, (j+1) blk) of the source) and block j belongs at rows ((j world + rank) blk ...) of the
// operand: bq = chunk pairs per block.
struct PackMap {
    i64 bq = 0;
    int world = 1, rank = 0;
    __host__ __device__ i64 dest(i64 q) const { return bq > 0 ? ((q / bq) * world + rank) * bq + q % bq : q; }
};
// out layout: [q][s][kt][lane = (r, h)][16 B], chunk = 2q + h covers rows chunk*E .. +E-1,
// r = k index inside tile kt.  bf16: hi = bf16(x), mid = bf16(x-hi), lo = bf16(x-hi-mid).
template <int EBYTES, int NSPLIT>
__global__ __launch_bounds__(256) void pack_kernel(const double* __restrict__ X, int k, int ldx, i64 N, int KT, i64 nq,
                                                   unsigned char* __restrict__ out, PackMap pm)
{
    constexpr int E = 16 / EBYTES;
    const i64 gid = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = (int)(gid & 63);
    const i64 rest = gid >> 6;
    const int kt = (int)(rest % KT);
    i64 q = rest / KT;
    if (q >= nq) return;
    const int r = kt * 32 + (lane & 31);
    const i64 row0 = (2 * q + (lane >> 5)) * E;
    q = pm.dest(q);
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

// The low terms of both operands are stored multiplied by 2^11 (so that they are normal fp16 numbers whenever the
// high term is: full 22-bit operands over a 2^28 range below the scale) and the cross-term accumulator is divided
// by 2^11 once, in the epilogue.
constexpr float F16X2_LO_SCALE = 2048.f;
// residual of the fp16 two-term split, (a s - hi) 2^11, for both halves of a packed fp16 pair: one v_fma_mix_f32 each
// (fp16 source read in place, fp32 accumulate; exact because a s - hi is representable)
static __device__ __forceinline__ void f16x2_residual(unsigned hi_pair, float xl0, float xl1, float& r0, float& r1)
{
#ifndef SMK_F16_RESID_ASM
#define SMK_F16_RESID_ASM 1
#endif
#if SMK_F16_RESID_ASM
    const float neg = -F16X2_LO_SCALE;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hi_pair), "v"(neg), "v"(xl0));
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hi_pair), "v"(neg), "v"(xl1));
#else
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    const h2_t h = __builtin_bit_cast(h2_t, hi_pair);
    r0 = __builtin_fmaf((float)h[0], -F16X2_LO_SCALE, xl0);
    r1 = __builtin_fmaf((float)h[1], -F16X2_LO_SCALE, xl1);
#endif
}

// fp16 two-term form: the same fragment layout with _Float16 entries; row r of the operand (factor row k0 + r) is
// multiplied by xscale[k0 + r] (a power of two from the Gram diagonal, gram_reduce_kernel) before the split so that its
// entries sit in fp16's range with full 11-bit precision: hi = fp16(x), lo = fp16(x - hi)  (22 significant bits).
__global__ __launch_bounds__(256) void pack_f16x2_kernel(const double* __restrict__ X, int k, int ldx, i64 N, int KT, i64 nq,
                                                         const double* __restrict__ xscale, unsigned char* __restrict__ out,
                                                         PackMap pm)
{
    const i64 gid = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = (int)(gid & 63);
    const i64 rest = gid >> 6;
    const int kt = (int)(rest % KT);
    i64 q = rest / KT;
    if (q >= nq) return;
    const int r = kt * 32 + (lane & 31);
    const i64 row0 = (2 * q + (lane >> 5)) * 8;
    q = pm.dest(q);
    const double sc = (r < k) ? xscale[r] : 0.0;
    double res[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const i64 row = row0 + e;
        res[e] = (row < N && r < k) ? X[row * ldx + r] * sc : 0.0;
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        f16x8_t h;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            h[e] = (_Float16)(float)res[e];
            res[e] = (res[e] - (double)(float)h[e]) * F16X2_LO_SCALE;     // the low term is carried 2^11 up
        }
        *(f16x8_t*)(out + (((q * 2 + s) * KT + kt) * 64 + lane) * 16) = h;
    }
}

// k in (8, 16], fp16 two-term form, Gram partials left behind by the NNLS launch (nnls_bpp_kernel<16>): reduce + pack as ONE
// launch.  Workgroups 0 .. 15 are gram_reduce_kernel's (the matrix, the row scales and the output scales of the product);
// every other workgroup packs, after adding up the 16 diagonal entries itself in gram_reduce_kernel's order (same sums, same
// power-of-two scales).  No workgroup waits for another.  C2: 4.6 + 5.3 us and a launch boundary become one launch.
__global__ __launch_bounds__(256) void reduce_pack_f16x2_k16_kernel(const double* __restrict__ Gp, int nblk, double* __restrict__ G,
                                                                    double* __restrict__ xscale, double* __restrict__ oscale,
                                                                    double ascale, const double* __restrict__ X, int k, int ldx, i64 N,
                                                                    i64 nq, unsigned char* __restrict__ out)
{
    constexpr int KP = 16, ELEMS = 256;
    __shared__ double sh[16][17];
    __shared__ double sxs[16];
    const int el = threadIdx.x & 15, g = threadIdx.x >> 4;
    const bool reducer = blockIdx.x < 16;
    const int e = reducer ? (int)blockIdx.x * 16 + el : el * (KP + 1);     // packers: diagonal entry el
    double s = 0.0;
    // packers read the compact copy of the diagonal the NNLS launch left behind the partials (contiguous 128-byte rows)
    const double* src = reducer ? Gp + e : Gp + (i64)NNLS_GRAM_MAX * ELEMS + el;
    const int stride = reducer ? ELEMS : KP;
    for (int b0 = g; b0 < nblk; b0 += 16 * 32) {                // 32 loads in flight, added in gram_reduce_kernel's order
        double v[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) v[u] = (b0 + 16 * u < nblk) ? src[(i64)(b0 + 16 * u) * stride] : 0.0;
#pragma unroll
        for (int u = 0; u < 32; ++u) s += v[u];                 // (s + 0.0 == s: the padding changes nothing)
    }
    sh[g][el] = s;
    __syncthreads();
    if (g == 0) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += sh[i][el];
        if (reducer) G[e] = t;
        if (!reducer || e % (KP + 1) == 0) {            // the rule of gram_reduce_kernel
            const int r = e / (KP + 1);
            int ex = 0;
            double xs = 1.0;
            if (t > 0.0 && t < 1.0e300) {
                (void)frexp(t, &ex);
                const int half = (ex >= 0) ? (ex + 1) / 2 : -((-ex) / 2);
                xs = ldexp(1.0, 14 - half);
            }
            if (reducer) {
                xscale[r] = xs;
                oscale[r] = 1.0 / (xs * ascale);
            } else {
                sxs[r] = xs;
            }
        }
    }
    if (reducer) return;
    __syncthreads();
    const i64 gid = (i64)(blockIdx.x - 16) * blockDim.x + threadIdx.x;     // pack_f16x2_kernel with KT = 1
    const int lane = (int)(gid & 63);
    const i64 q = gid >> 6;
    if (q >= nq) return;
    const int r = lane & 31;
    const i64 row0 = (2 * q + (lane >> 5)) * 8;
    const double sc = (r < k) ? sxs[r & 15] : 0.0;
    double res[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const i64 row = row0 + i;
        res[i] = (row < N && r < k) ? X[row * ldx + r] * sc : 0.0;
    }
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
        f16x8_t h;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            h[i] = (_Float16)(float)res[i];
            res[i] = (res[i] - (double)(float)h[i]) * F16X2_LO_SCALE;
        }
        *(f16x8_t*)(out + ((q * 2 + t2) * 64 + lane) * 16) = h;
    }
}

// The packed operand layout does not depend on the stage height: it is a sequence of 1-KiB
// blocks indexed by the global chunk-pair q; rows are padded to a multiple of 128.
// operand format: bf16 fragments (E = 8) for bf16 storage and for the fp32 "bf16x3" emulation
// (nsplit == 3); native fp32 fragments (E = 4, one term) otherwise
static inline bool pack_is_bf16(int storage, int nsplit) { return storage == STORE_BF16 || nsplit >= 2; }
static inline int pack_terms(int nsplit) { return nsplit == NSPLIT_F16X2 ? 2 : nsplit; }

static inline i64 pack_nq(int storage, int nsplit, i64 N)
{
    const i64 E = pack_is_bf16(storage, nsplit) ? 8 : 4;
    return round_up(N, ROW_PAD) / (2 * E);
}

i64 packed_chunk_pairs_f16x2(int storage, i64 N) { return pack_nq(storage, NSPLIT_F16X2, N); }

size_t packed_bytes(int storage, int k, i64 N, int nsplit)
{
    if (nsplit == NSPLIT_F64) return 16;         // the accurate form reads the factor itself
    if (!pack_is_bf16(storage, nsplit)) nsplit = 1;
    size_t total = 0;
    for (int k0 = 0; k0 < k; k0 += 64)          // groups of 64 factor rows, each packed with its own k-tile count
        total += (size_t)pack_nq(storage, nsplit, N) * pack_terms(nsplit) * kt_of(k - k0 < 64 ? k - k0 : 64) * 1024;
    return total;
}

// bytes from the start of a group's packed block to the fragments of row r0 (a multiple of 16): the layout is
// chunk-pair major, so a row range of the operand is a contiguous byte range
size_t packed_row_offset(int storage, int kg, int nsplit, i64 r0)
{
    const i64 E = pack_is_bf16(storage, nsplit) ? 8 : 4;
    if (!pack_is_bf16(storage, nsplit)) nsplit = 1;
    return (size_t)(r0 / (2 * E)) * pack_terms(nsplit) * kt_of(kg) * 1024;
}

static int launch_pack_rows_mapped(const double* X, int ldx, int k0, int kg, i64 N, i64 nq, int storage, int nsplit, void* out,
                                   hipStream_t st, const double* xscale, PackMap pm)
{
    const int KT = kt_of(kg);
    const i64 threads = nq * KT * 64;
    const int grid = (int)((threads + 255) / 256);
    if (grid == 0) return 0;
    const double* Xg = X + k0;                   // the kernel sees rows [k0, k0 + kg) as rows [0, kg)
    if (nsplit == NSPLIT_F16X2) {
        if (!xscale) { set_error("pack: the fp16 two-term form needs the row scales"); return -100; }
        pack_f16x2_kernel<<<grid, 256, 0, st>>>(Xg, kg, ldx, N, KT, nq, xscale + k0, (unsigned char*)out, pm);
    } else if (pack_is_bf16(storage, nsplit)) {
        if (nsplit == 3) pack_kernel<2, 3><<<grid, 256, 0, st>>>(Xg, kg, ldx, N, KT, nq, (unsigned char*)out, pm);
        else if (nsplit == 2) pack_kernel<2, 2><<<grid, 256, 0, st>>>(Xg, kg, ldx, N, KT, nq, (unsigned char*)out, pm);
        else pack_kernel<2, 1><<<grid, 256, 0, st>>>(Xg, kg, ldx, N, KT, nq, (unsigned char*)out, pm);
    } else {
        pack_kernel<4, 1><<<grid, 256, 0, st>>>(Xg, kg, ldx, N, KT, nq, (unsigned char*)out, pm);
    }
    SMK_HIP(hipGetLastError());
    return 0;
}

int launch_pack_rows(const double* X, int ldx, int k0, int kg, i64 N, int storage, int nsplit, void* out, hipStream_t st,
                     const double* xscale)
{
    return launch_pack_rows_mapped(X, ldx, k0, kg, N, pack_nq(storage, nsplit, N), storage, nsplit, out, st, xscale, PackMap{});
}

// rows [0, N) of X are this rank's blocks of `blk` rows back to back (nblocks of them; rows >= N are padding and pack
// as zeros); block j goes to rows ((j world + rank) blk ...) of the operand that starts at `out`
int launch_pack_own_blocks(const double* X, int ldx, int k0, int kg, i64 N, i64 blk, int nblocks, int world, int rank,
                           int storage, int nsplit, void* out, hipStream_t st, const double* xscale)
{
    const i64 E2 = pack_is_bf16(storage, nsplit) ? 16 : 8;       // rows per chunk pair
    PackMap pm;
    pm.bq = blk / E2; pm.world = world; pm.rank = rank;
    return launch_pack_rows_mapped(X, ldx, k0, kg, N, (i64)nblocks * pm.bq, storage, nsplit, out, st, xscale, pm);
}

// see reduce_pack_f16x2_k16_kernel; returns 1 when the shape is not the fused one
int launch_reduce_pack_f16x2(const double* Gp, int nblk, int k, double* G, double* xscale, double* oscale, double ascale,
                             const double* X, i64 N, int storage, void* out, hipStream_t st)
{
    if (kp_of(k) != 16 || nblk < 1) return 1;
    const i64 nq = pack_nq(storage, NSPLIT_F16X2, N);
    const i64 threads = nq * 64;
    const unsigned grid = 16u + (unsigned)((threads + 255) / 256);
    reduce_pack_f16x2_k16_kernel<<<grid, 256, 0, st>>>(Gp, nblk, G, xscale, oscale, ascale, X, k, 16, N, nq, (unsigned char*)out);
    SMK_HIP(hipGetLastError());
    return 0;
}

int launch_pack(const double* X, int k, i64 N, int storage, int nsplit, void* out, hipStream_t st, const double* xscale)
{
    // all groups of a k-row factor, back to back (group g: rows [64 g, 64 g + 64) of [0, k))
    size_t off = 0;
    for (int k0 = 0; k0 < k; k0 += 64) {
        const int kg = k - k0 < 64 ? k - k0 : 64;
        int rc = launch_pack_rows(X, kp_of(k), k0, kg, N, storage, nsplit, (unsigned char*)out + off, st, xscale);
        if (rc) return rc;
        off += packed_bytes(storage, kg, N, nsplit);
    }
    return 0;
}


// ==========================================================================
// fp32 A as three bf16 planes ("bf16x3"), second generation.
//
// Same staging (LDS ring filled by global_load_lds, XOR-swizzled 16-byte chunks, packed X fragments) and
// the same result layout as bigprod_kernel, but organised around what limited that kernel for fp32 input
// at k in (32, 64] (VALU issue, not HBM -- DESIGN.md 5.1):
//   * a compute wave owns 32 columns and ALL k tiles, so the fp32 -> bf16 hi/mid/lo split of a tile is done
//     once (it was done once per k-tile wave group) and each B fragment is read from LDS once;
//   * the five small products (significance 2^-8 and 2^-16) share ONE fp32 accumulator per tile;
//   * the leading product is folded into the fp64 sums every FOLD stages (FOLD * MB rows, the chain length the
//     bf16 path has) by plain VALU adds, and the first MFMA after a fold takes C = 0, so there is no second
//     accumulator set and no zeroing;
//   * 4 compute waves (+ NWL loader waves) per 128-column tile.
// ==========================================================================
template <int KT, int MB_, int NSTAGE_, int NWL_, int NS = 3>
struct F3Cfg {
    static constexpr int MB = MB_;
    static constexpr int NWC = 4;
    static constexpr int NWL = NWL_;
    static constexpr int NLD = NWL_ > 0 ? NWL_ : NWC;
    static constexpr int NW = NWC + NWL_;
    static constexpr int CPC = MB / 4;              // 16-byte chunks per column per stage
    static constexpr int SWZ_SH = (CPC >= 16) ? 0 : (CPC == 8) ? 1 : (CPC == 4) ? 2 : 3;
    static constexpr int SWZ_MASK = (CPC >= 16 ? 16 : CPC) - 1;
    static constexpr int QS = MB / 16;              // 16-row MFMA steps per stage
    static constexpr int NB = 128;
    static constexpr int B_BYTES = NB * MB * 4;
    static constexpr int X_BYTES = QS * NS * KT * 1024;
    static constexpr int STAGE_BYTES = B_BYTES + X_BYTES;
    static constexpr int NSTAGE = NSTAGE_;
    static constexpr int PD = NSTAGE_ - 1;
    static constexpr int TI = STAGE_BYTES / 1024;
    static constexpr int LPS = TI / NLD;
    static constexpr bool OK = (TI % NLD == 0) && (LPS * PD <= 63) && (STAGE_BYTES * NSTAGE <= 160 * 1024);
};

// TAIL workgroups of bigprod_f3_kernel (256 threads): workgroup w of 16 adds up entries 16 w .. 16 w + 15 of nblk partial
// 16 x 16 Gram matrices -- gram_reduce_kernel's sums in gram_reduce_kernel's order (16 groups of threads stride the partials,
// the 16 group sums are added in order), 32 loads in flight.  C2: the Gram matrix of the factor an NNLS launch has just solved
// is needed by the NEXT NNLS launch, one streaming pass later, so its reduction rides in that pass instead of sitting in
// front of it as a launch of its own.
static __device__ __forceinline__ void gram_reduce_tail(const double* __restrict__ Gp, int nblk, double* __restrict__ G, int w,
                                                        double* __restrict__ sh /* 16 x 17 doubles of LDS */)
{
    constexpr int ELEMS = 256;
    const int el = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int e = w * 16 + el;
    double s = 0.0;
    for (int b0 = g; b0 < nblk; b0 += 16 * 32) {
        double v[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) v[u] = (b0 + 16 * u < nblk) ? Gp[(i64)(b0 + 16 * u) * ELEMS + e] : 0.0;
#pragma unroll
        for (int u = 0; u < 32; ++u) s += v[u];                 // (s + 0.0 == s: the padding changes nothing)
    }
    sh[g * 17 + el] = s;
    __syncthreads();
    if (g == 0) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += sh[i * 17 + el];
        G[e] = t;
    }
}

// the totals of a deferred progress check (pg_defer_sum_kernel's arithmetic: block_sum_array over 256 threads)
static __device__ __forceinline__ void check_totals_tail(const BigProdPlan::TailCheck& tc, double* __restrict__ sh /* >= 16 doubles */)
{
    const double t1 = block_sum_array(tc.part, tc.n, sh);
    if (tc.snap_g)
        for (int i = threadIdx.x; i < tc.kk; i += blockDim.x) tc.snap_g[i] = tc.G[i];
    if (threadIdx.x == 0) {
        int fv = tc.flag ? *tc.flag : INT_MAX;
        if (fv != INT_MAX && fv > tc.tag_limit) fv = INT_MAX;     // a failure of the speculated NEXT iteration is not this check's
        const double f = (double)fv;
        tc.out[0] = 0.0; tc.out[1] = t1; tc.out[tc.flag_slot] = f;
        tc.host_out[0] = 0.0; tc.host_out[1] = t1; tc.host_out[tc.flag_slot] = f;
        if (tc.tag != 0.0) {                 // the host polls the slot: the tag goes out last
            __threadfence_system();
            __hip_atomic_store(&tc.host_out[7], tc.tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

template <int KT, int MB, int NSTAGE, int NWL, int FOLD, int WPS, int NS, int FMT, int TAIL = 0>
__global__ __launch_bounds__(64 * (4 + NWL), WPS) void bigprod_f3_kernel(const unsigned char* __restrict__ B, i64 ldb_bytes,
                                                                        const unsigned char* __restrict__ Xp,
                                                                        double* __restrict__ P, i64 stages, i64 nst,
                                                                        i64 tiles, i64 ncols_pad, int S, int logS, int pstride,
                                                                        const double* __restrict__ oscale, float ascale, int accum,
                                                                        const double* __restrict__ tail_gp, int tail_nblk,
                                                                        double* __restrict__ tail_g, BigProdPlan::TailCheck tc, InvRide ride)
{
    using C = F3Cfg<KT, MB, NSTAGE, NWL, NS>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    static_assert(TAIL != 2 || MB == 32, "transposed source (TAIL == 2): 32-column stages, pieces of two columns");
    int bid = blockIdx.x;
    if constexpr (TAIL != 2) {
        // the Gram inverse of the next block-pivoting launch rides along (common.h: InvRide): eight more workgroups in front (the
        // tile mapping keeps its XCDs), the first four waves of the first one invert, in the launch's own LDS
        static_assert(C::STAGE_BYTES * NSTAGE >= GRAM_INVERSE_LDS(64) * 8, "the inversion needs 2.6 KB of LDS");
        if (ride.G) {
            if (bid < 8) {
                if (bid == 0 && threadIdx.x < 256) {
                    if (ride.k <= 32) gram_inverse64_body<32>(ride.G, ride.k, ride.Ginv, (int*)(ride.Ginv + 32 * 32), (double*)smem);
                    else gram_inverse64_body<64>(ride.G, ride.k, ride.Ginv, (int*)(ride.Ginv + 64 * 64), (double*)smem);
                }
                return;
            }
            bid -= 8;
        }
    }
    if constexpr (TAIL == 1) {
        // the first 16 workgroups (two per XCD, so the tile mapping below keeps its XCD of every other workgroup) reduce
        static_assert(NWL == 0 && C::STAGE_BYTES * NSTAGE >= 16 * 17 * 8, "the tail needs 256 threads and 2176 bytes of LDS");
        // (measured on C2, 512 product workgroups = every slot of the chip: reducers first 25.8 us per launch, reducers last 26.3,
        // the reduction spread over the first 256 product workgroups behind their first stage loads 26.9; without a tail 24.9)
        if (bid < 16) { gram_reduce_tail(tail_gp, tail_nblk, tail_g, bid, (double*)smem); return; }
        // a deferred progress check rides along: eight more workgroups (the tile mapping keeps its XCDs), the first one adds up
        const int ntail = tc.part ? 24 : 16;
        if (bid < ntail) { if (bid == 16) check_totals_tail(tc, (double*)smem); return; }
        bid -= ntail;
    }
    const int xcd = bid & 7;
    const i64 grp = bid >> 3;
    i64 tile;
    int split;
    // (the other assignment -- every row split of a column tile on ONE XCD, so that a tile's pages of B are touched through one
    // XCD only while the operand chunks go through every L2 -- measures the same within noise on five C4-sized shapes:
    // profiles/r04_xcd_mapping_sweep.txt)
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
    const bool is_loader = (NWL == 0) || (wave >= C::NWC);
    const bool is_compute = wave < C::NWC;
    const int lw = (NWL == 0) ? wave : (wave - C::NWC);

    i64 st0 = (i64)split * nst;
    i64 st1 = st0 + nst;
    if (st1 > stages) st1 = stages;
    const int my_nst = (st1 > st0) ? (int)(st1 - st0) : 0;

    const i64 col0 = tile * C::NB;
    // one running pointer per load of a stage: issue() is called for consecutive stages only, so each call costs one
    // 64-bit add per load (the select between the two operands and the stage multiply are paid once, here)
    const unsigned char* gsrc[C::LPS];
    int is_b[C::LPS];
#pragma unroll
    for (int i = 0; i < C::LPS; ++i) {
        const int t = lw + C::NLD * i;
        if (t * 1024 < C::B_BYTES) {
            if constexpr (TAIL == 2) {
                // transposed source (fp32): B = A, the tile's columns are 128 consecutive ROWS of A, a stage is MB COLUMNS of A.
                // Piece t = columns 2 t, 2 t + 1 of the stage, 512 contiguous bytes (four lines) each, fetched by 32 adjacent
                // lanes; the LDS image is the natural [column][row] one with a 512-byte pitch
                gsrc[i] = B + (st0 * C::MB + 2 * t + (lane >> 5)) * ldb_bytes + (col0 + (lane & 31) * 4) * 4;
            } else {
            const int p = t * 64 + lane;
            const int j = p / C::CPC;
            const int pc = p % C::CPC;
            const int swz = (j >> C::SWZ_SH) & C::SWZ_MASK;
            gsrc[i] = B + (col0 + j) * ldb_bytes + (i64)(pc ^ swz) * 16 + st0 * (C::MB * 4);
            }
            is_b[i] = 1;
        } else {
            gsrc[i] = Xp + st0 * C::X_BYTES + (i64)(t * 1024 - C::B_BYTES) + lane * 16;
            is_b[i] = 0;
        }
    }
    auto issue = [&](int s_local) {
        unsigned char* lbase = smem + (s_local % C::NSTAGE) * C::STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < C::LPS; ++i) {
            const int t = lw + C::NLD * i;
            if (is_b[i]) {
                if (accum & 2) __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)gsrc[i], (LDS_AS void*)(lbase + t * 1024), 16, 0, 0);        // a matrix that fits the Infinity Cache: keep it there
                else __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)gsrc[i], (LDS_AS void*)(lbase + t * 1024), 16, 0, NT_AUX);
                if constexpr (TAIL == 2) gsrc[i] += C::MB * ldb_bytes;
                else gsrc[i] += C::MB * 4;
            } else {
                __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)gsrc[i], (LDS_AS void*)(lbase + t * 1024), 16, 0, 0);
                gsrc[i] += C::X_BYTES;
            }
        }
    };

    f32x16_t hi[KT], sm[KT];
    double dacc[KT][16];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { hi[kt][r] = 0.f; sm[kt][r] = 0.f; dacc[kt][r] = 0.0; }

    const int h = lane >> 5;
    const int jl = wave * 32 + (lane & 31);                  // this lane's column inside the tile (compute waves)
    const int swz_r = (jl >> C::SWZ_SH) & C::SWZ_MASK;
    const int bfrag_base = jl * C::CPC * 16;

    auto fold = [&]() {
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) dacc[kt][r] += (double)hi[kt][r];
    };

#ifdef SMK_BP_PROFILE
    unsigned long long prof_[4] = {0, 0, 0, 0};
#endif
    // one stage; FIRST: the leading accumulators restart from zero (C = 0 on their first MFMA)
    auto stage_body = [&](int t, auto first_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        int ahead = my_nst - 1 - t;
        if (ahead > C::PD - 1) ahead = C::PD - 1;
        BP_T(t0_);
        if (is_loader) wait_stage<C::LPS, C::PD>(ahead);
        BP_T(t1_);
        __builtin_amdgcn_s_barrier();
        BP_T(t2_);
        if (is_loader && t + C::PD < my_nst) issue(t + C::PD);
        BP_T(t3_);
#ifdef SMK_BP_PROFILE
        prof_[0] += t1_ - t0_; prof_[1] += t2_ - t1_; prof_[2] += t3_ - t2_;
#endif
        if (!is_compute) return;
        const unsigned char* sb = smem + (t % C::NSTAGE) * C::STAGE_BYTES;
        const unsigned char* sx = sb + C::B_BYTES;
#pragma unroll
        for (int q = 0; q < C::QS; ++q) {
            // rows 16q + 8h .. +7 of this lane's column = fp32 chunks 4q + 2h, 4q + 2h + 1
            const int lc0 = 4 * q + 2 * h;
            (void)lc0;
            f32x8_t x;
            if constexpr (TAIL == 2) {
                // the lane's row of A in the 8 columns 16 q + 8 h .. + 7 of the stage: eight 4-byte reads 512 bytes apart (a half-wave
                // reads 32 consecutive words per instruction: conflict-free)
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = *(const float*)(sb + (16 * q + 8 * h + e) * 512 + jl * 4);
            } else {
            const f32x4_t f0 = *(const f32x4_t*)(sb + bfrag_base + (((lc0) ^ swz_r) << 4));
            const f32x4_t f1 = *(const f32x4_t*)(sb + bfrag_base + (((lc0 + 1) ^ swz_r) << 4));
#pragma unroll
            for (int e = 0; e < 4; ++e) { x[e] = f0[e]; x[4 + e] = f1[e]; }
            }
            if constexpr (FMT == 1) {
                // fp16 two-term form: hi = fp16(a s), lo = fp16(a s - hi); products hi*hi | hi*lo + lo*hi
                f16x8_t a[KT][2];
#pragma unroll
                for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                    for (int s = 0; s < 2; ++s)
                        a[kt][s] = __builtin_bit_cast(f16x8_t, *(const u32x4_t*)(sx + ((q * 2 + s) * KT + kt) * 1024 + lane * 16));
                // hi = fp16(a s); lo = fp16((a s - hi) 2^11), the residual as one mixed-precision fma per entry (exact)
                const f32x8_t xs = x * ascale, xl = x * (ascale * F16X2_LO_SCALE);
                const f16x8_t bh = __builtin_convertvector(xs, f16x8_t);
                {
                    const u32x4_t hp = __builtin_bit_cast(u32x4_t, bh);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float r0, r1;
                        f16x2_residual(hp[e], xl[2 * e], xl[2 * e + 1], r0, r1);
                        x[2 * e] = r0;
                        x[2 * e + 1] = r1;
                    }
                }
                const f16x8_t bl = __builtin_convertvector(x, f16x8_t);
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {
                    if (FIRST && q == 0) {
                        const f32x16_t zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        hi[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[kt][0], bh, zero, 0, 0, 0);
                    } else {
                        hi[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[kt][0], bh, hi[kt], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) sm[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[kt][0], bl, sm[kt], 0, 0, 0);
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) sm[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[kt][1], bh, sm[kt], 0, 0, 0);
            } else {
            bf16x8_t a[KT][NS];
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    a[kt][s] = __builtin_bit_cast(bf16x8_t, *(const u32x4_t*)(sx + ((q * NS + s) * KT + kt) * 1024 + lane * 16));
            const bf16x8_t bhi = __builtin_convertvector(x, bf16x8_t);
            x -= __builtin_convertvector(bhi, f32x8_t);
            const bf16x8_t bmid = __builtin_convertvector(x, bf16x8_t);
            bf16x8_t blo = bmid;
            if constexpr (NS == 3) {
                x -= __builtin_convertvector(bmid, f32x8_t);
                blo = __builtin_convertvector(x, bf16x8_t);
            }
            // independent accumulators alternate so that no MFMA waits on the one just issued
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                if (FIRST && q == 0) {
                    const f32x16_t zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    hi[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kt][0], bhi, zero, 0, 0, 0);
                } else {
                    hi[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kt][0], bhi, hi[kt], 0, 0, 0);
                }
            }
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) sm[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kt][0], bmid, sm[kt], 0, 0, 0);
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) sm[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kt][1], bhi, sm[kt], 0, 0, 0);
            if constexpr (NS == 3) {
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) sm[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kt][0], blo, sm[kt], 0, 0, 0);
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) sm[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kt][1], bmid, sm[kt], 0, 0, 0);
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) sm[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kt][NS - 1], bhi, sm[kt], 0, 0, 0);
            }
            }
        }
#ifdef SMK_BP_PROFILE
        asm volatile("s_nop 0" ::: "memory");
        prof_[3] += __builtin_readcyclecounter() - t3_;
#endif
    };

    if (is_loader) {
#pragma unroll
        for (int i = 0; i < C::PD; ++i)
            if (i < my_nst) issue(i);
    }
    int t = 0;
    for (; t + FOLD <= my_nst; t += FOLD) {
        stage_body(t, std::true_type{});
#pragma unroll
        for (int u = 1; u < FOLD; ++u) stage_body(t + u, std::false_type{});
        if (is_compute) fold();
    }
    if (t < my_nst) {
        stage_body(t, std::true_type{});
        for (++t; t < my_nst; ++t) stage_body(t, std::false_type{});
        if (is_compute) fold();
    }
#ifdef SMK_BP_PROFILE
    if (g_bp_prof && lane == 0) {
        unsigned long long* o = g_bp_prof + ((size_t)blockIdx.x * C::NW + wave) * 4;
        o[0] = prof_[0]; o[1] = prof_[1]; o[2] = prof_[2]; o[3] = prof_[3];
    }
#endif
    if (!is_compute) return;

    // epilogue (same layout as bigprod_kernel): col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const i64 jg = col0 + jl;
    double* pout = P + ((i64)split * ncols_pad + jg) * pstride;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int i = 0; i < 4; i += 2) {
                f64x2_t v;
                constexpr double cross = (FMT == 1) ? 1.0 / F16X2_LO_SCALE : 1.0;
                v[0] = dacc[kt][4 * g + i] + cross * (double)sm[kt][4 * g + i];
                v[1] = dacc[kt][4 * g + i + 1] + cross * (double)sm[kt][4 * g + i + 1];
                if constexpr (FMT == 1) {
                    const f64x2_t os = *(const f64x2_t*)(oscale + kt * 32 + 8 * g + 4 * h + i);
                    v[0] *= os[0];
                    v[1] *= os[1];
                }
                if (pstride < 32 && kt * 32 + 8 * g >= pstride) continue;      // k <= 16: a column of P holds 8 / 16 rows, not 32 (kpp_of)
                if (accum & 1) v += *(const f64x2_t*)(pout + kt * 32 + 8 * g + 4 * h + i);
                *(f64x2_t*)(pout + kt * 32 + 8 * g + 4 * h + i) = v;
            }
}


// --------------------------------------------------------------------------
// Software-pipelined form of the kernel above (MB = 32: two 16-row steps per stage).  Measured on the
// non-pipelined form with one compute wave per SIMD: 90 % of a wave's time is its compute phase, ~1400
// cycles per stage for 24 MFMAs (768 cycles of matrix pipe) -- the fragment reads and the bf16 split of a
// step sit in front of its MFMAs instead of beside them.  Here the reads + split of step u + 1 are issued
// behind the MFMAs of step u (register double buffer), across stages too: the barrier that publishes stage
// t + 1 sits between the two steps of stage t, so a ring slot is reused two barriers after its last read
// (in flight: NSTAGE - 2 stages).  sched_group_barrier pins the interleave: 1 MFMA : 1 LDS read : 4 VALU.
// --------------------------------------------------------------------------
static __device__ __forceinline__ f32x16_t mfma16(bf16x8_t a, bf16x8_t b, f32x16_t c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
static __device__ __forceinline__ f32x16_t mfma16(f16x8_t a, f16x8_t b, f32x16_t c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
template <int KT, int NSTAGE, int NWL, int FOLD, int PIN, int NS, int FMT>
__global__ __launch_bounds__(64 * (4 + NWL), 2) void bigprod_f3p_kernel(const unsigned char* __restrict__ B, i64 ldb_bytes,
                                                                        const unsigned char* __restrict__ Xp,
                                                                        double* __restrict__ P, i64 stages, i64 nst,
                                                                        i64 tiles, i64 ncols_pad, int S, int logS, int pstride,
                                                                        const double* __restrict__ oscale, float ascale, int accum)
{
    using C = F3Cfg<KT, 32, NSTAGE, NWL, NS>;
    constexpr int PDN = NSTAGE - 2;                 // stages in flight ahead of the published one
    static_assert(PDN >= 1, "ring too shallow");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

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
    const bool is_compute = wave < C::NWC;
    const int lw = (NWL == 0) ? wave : (wave - C::NWC);

    i64 st0 = (i64)split * nst;
    i64 st1 = st0 + nst;
    if (st1 > stages) st1 = stages;
    const int my_nst = (st1 > st0) ? (int)(st1 - st0) : 0;

    const i64 col0 = tile * C::NB;
    const unsigned char* gsrc[C::LPS];             // running pointers, as in bigprod_f3_kernel
    int is_b[C::LPS];
#pragma unroll
    for (int i = 0; i < C::LPS; ++i) {
        const int t = lw + C::NLD * i;
        if (t * 1024 < C::B_BYTES) {
            const int p = t * 64 + lane;
            const int j = p / C::CPC;
            const int pc = p % C::CPC;
            const int swz = (j >> C::SWZ_SH) & C::SWZ_MASK;
            gsrc[i] = B + (col0 + j) * ldb_bytes + (i64)(pc ^ swz) * 16 + st0 * (C::MB * 4);
            is_b[i] = 1;
        } else {
            gsrc[i] = Xp + st0 * C::X_BYTES + (i64)(t * 1024 - C::B_BYTES) + lane * 16;
            is_b[i] = 0;
        }
    }
    auto issue = [&](int s_local) {
        unsigned char* lbase = smem + (s_local % C::NSTAGE) * C::STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < C::LPS; ++i) {
            const int t = lw + C::NLD * i;
            if (is_b[i]) {
                if (accum & 2) __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)gsrc[i], (LDS_AS void*)(lbase + t * 1024), 16, 0, 0);        // a matrix that fits the Infinity Cache: keep it there
                else __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)gsrc[i], (LDS_AS void*)(lbase + t * 1024), 16, 0, NT_AUX);
                gsrc[i] += C::MB * 4;
            } else {
                if constexpr ((PIN & 16) == 0)      // experiment bit 16: the X fragments are not fetched at all
                    __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)gsrc[i], (LDS_AS void*)(lbase + t * 1024), 16, 0, 0);
                gsrc[i] += C::X_BYTES;
            }
        }
    };
    f32x16_t hi[KT], sm[KT];
    double dacc[KT][16];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { hi[kt][r] = 0.f; sm[kt][r] = 0.f; dacc[kt][r] = 0.0; }

    const int h = lane >> 5;
    const int jl = (wave & 3) * 32 + (lane & 31);
    const int swz_r = (jl >> C::SWZ_SH) & C::SWZ_MASK;
    const int bfrag_base = jl * C::CPC * 16;

    using V = std::conditional_t<FMT == 1, f16x8_t, bf16x8_t>;   // fp16 two-term form or bf16 terms
    struct Frag { V bh, bm, bl; V a[KT][NS]; };
    auto lds_read = [&](int slot, int q, f32x4_t& r0, f32x4_t& r1, Frag& f) {
        const unsigned char* sb = smem + slot * C::STAGE_BYTES;
        const unsigned char* sx = sb + C::B_BYTES;
        const int lc0 = 4 * q + 2 * h;             // rows 16q + 8h .. +7 = fp32 chunks lc0, lc0 + 1
        r0 = *(const f32x4_t*)(sb + bfrag_base + (((lc0) ^ swz_r) << 4));
        r1 = *(const f32x4_t*)(sb + bfrag_base + (((lc0 + 1) ^ swz_r) << 4));
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int s = 0; s < NS; ++s)
                f.a[kt][s] = __builtin_bit_cast(V, *(const u32x4_t*)(sx + ((q * NS + s) * KT + kt) * 1024 + lane * 16));
    };
    auto split3 = [&](const f32x4_t& r0, const f32x4_t& r1, Frag& f) {
        f32x8_t x;
#pragma unroll
        for (int e = 0; e < 4; ++e) { x[e] = r0[e]; x[4 + e] = r1[e]; }
        if constexpr ((PIN & 2) != 0) {             // experiment: no split (wrong numbers, same traffic)
            f.bh = __builtin_bit_cast(V, r0);
            f.bm = __builtin_bit_cast(V, r1);
            f.bl = f.bh;
            return;
        }
        if constexpr (FMT == 1) {       // as in bigprod_f3_kernel
            const f32x8_t xs = x * ascale, xl = x * (ascale * F16X2_LO_SCALE);
            f.bh = __builtin_convertvector(xs, V);
            const u32x4_t hp = __builtin_bit_cast(u32x4_t, f.bh);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                        float r0, r1;
                        f16x2_residual(hp[e], xl[2 * e], xl[2 * e + 1], r0, r1);
                        x[2 * e] = r0;
                        x[2 * e + 1] = r1;
                    }
            f.bm = __builtin_convertvector(x, V);
            f.bl = f.bm;
            return;
        }
        f.bh = __builtin_convertvector(x, V);
        x -= __builtin_convertvector(f.bh, f32x8_t);
        f.bm = __builtin_convertvector(x, V);
        f.bl = f.bm;
        if constexpr (NS == 3) {
            x -= __builtin_convertvector(f.bm, f32x8_t);
            f.bl = __builtin_convertvector(x, V);
        }
    };
    auto mma = [&](const Frag& f) {
        if constexpr ((PIN & 4) != 0) {             // experiment: no matrix work, keep every read alive
            u32x4_t acc = __builtin_bit_cast(u32x4_t, f.bh) ^ __builtin_bit_cast(u32x4_t, f.bm) ^ __builtin_bit_cast(u32x4_t, f.bl);
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int s3 = 0; s3 < NS; ++s3) acc ^= __builtin_bit_cast(u32x4_t, f.a[kt][s3]);
            hi[0][0] += __builtin_bit_cast(float, acc[0] ^ acc[1] ^ acc[2] ^ acc[3]);
            return;
        }
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) hi[kt] = mfma16(f.a[kt][0], f.bh, hi[kt]);
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) sm[kt] = mfma16(f.a[kt][0], f.bm, sm[kt]);
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) sm[kt] = mfma16(f.a[kt][1], f.bh, sm[kt]);
        if constexpr (NS == 3) {
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) sm[kt] = mfma16(f.a[kt][0], f.bl, sm[kt]);
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) sm[kt] = mfma16(f.a[kt][1], f.bm, sm[kt]);
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) sm[kt] = mfma16(f.a[kt][NS - 1], f.bh, sm[kt]);
        }
    };
    // interleave of one half stage: 6 KT MFMAs, 2 + 3 KT fragment reads, ~44 VALU of the split
    auto pin_schedule = [&]() {
        if constexpr ((PIN & 1) == 0) return;
#pragma unroll
        for (int i = 0; i < (NS == 3 ? 6 : 3) * KT; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);              // one MFMA
            if (i < 2 + NS * KT) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // one LDS read
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);              // four VALU
        }
    };

    static_assert(NWL > 0, "the pipelined kernel keeps loading and computing on different waves");
    if (!is_compute) {
        // ---- loader waves: one barrier per stage, in step with the compute waves below ----
        if constexpr ((PIN & 8) != 0) __builtin_amdgcn_s_setprio(3);
#pragma unroll
        for (int i = 0; i < PDN; ++i)
            if (i < my_nst) issue(i);
        for (int s = 0; s < my_nst; ++s) {
            int ahead = my_nst - 1 - s;
            if (ahead > PDN - 1) ahead = PDN - 1;
            wait_stage<C::LPS, PDN>(ahead);
            __builtin_amdgcn_s_barrier();                       // publishes stage s; stage s - 2 is consumed
            if (s + PDN < my_nst) issue(s + PDN);
        }
        return;
    }
    if (my_nst > 0) {
        __builtin_amdgcn_s_barrier();                           // stage 0 published
        Frag fA, fB;
        f32x4_t r0, r1;
        lds_read(0, 0, r0, r1, fA);
        split3(r0, r1, fA);
        for (int t = 0; t < my_nst; ++t) {
            const int slot = t % C::NSTAGE;
            const int slot_next = (t + 1) % C::NSTAGE;         // past the end: stale but valid LDS, never used
            lds_read(slot, 1, r0, r1, fB);
            mma(fA);
            split3(r0, r1, fB);
            pin_schedule();
            if (t + 1 < my_nst) __builtin_amdgcn_s_barrier();   // stage t + 1 published
            lds_read(slot_next, 0, r0, r1, fA);
            mma(fB);
            split3(r0, r1, fA);
            pin_schedule();
            if ((t % FOLD) == FOLD - 1 || t == my_nst - 1) {
#pragma unroll
                for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { dacc[kt][r] += (double)hi[kt][r]; hi[kt][r] = 0.f; }
            }
        }
    }

    const i64 jg = col0 + jl;
    double* pout = P + ((i64)split * ncols_pad + jg) * pstride;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int i = 0; i < 4; i += 2) {
                f64x2_t v;
                constexpr double cross = (FMT == 1) ? 1.0 / F16X2_LO_SCALE : 1.0;
                v[0] = dacc[kt][4 * g + i] + cross * (double)sm[kt][4 * g + i];
                v[1] = dacc[kt][4 * g + i + 1] + cross * (double)sm[kt][4 * g + i + 1];
                if constexpr (FMT == 1) {
                    const f64x2_t os = *(const f64x2_t*)(oscale + kt * 32 + 8 * g + 4 * h + i);
                    v[0] *= os[0];
                    v[1] *= os[1];
                }
                if (pstride < 32 && kt * 32 + 8 * g >= pstride) continue;      // k <= 16: a column of P holds 8 / 16 rows, not 32 (kpp_of)
                if (accum & 1) v += *(const f64x2_t*)(pout + kt * 32 + 8 * g + 4 * h + i);
                *(f64x2_t*)(pout + kt * 32 + 8 * g + 4 * h + i) = v;
            }
}

template <int KT, int NS, int NSTAGE, int NWL, int FOLD, int PIN = 1, int FMT = 0>
static int launch_f3p_t(const BigProdPlan& pl, const void* B, i64 ldb, const void* Xp, double* P, hipStream_t st)
{
    using C = F3Cfg<KT, 32, NSTAGE, NWL, NS>;
    if constexpr (!C::OK) {
        set_error("bigprod f3p variant does not fit");
        return -100;
    } else {
        constexpr int lds = C::STAGE_BYTES * C::NSTAGE;
        static std::atomic<unsigned long long> attr_set{0};       // per device (DeviceOnce)
        auto kern = bigprod_f3p_kernel<KT, NSTAGE, NWL, FOLD, PIN, NS, FMT>;
        if (DeviceOnce once{attr_set}) {
            SMK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            once.done();
        }
        int logS = 0;
        while ((1 << logS) < pl.S) ++logS;
        i64 grid;
        if (pl.S <= 8) {
            const i64 per = 8 >> logS;
            grid = (pl.tiles + per - 1) / per * 8;
        } else {
            grid = pl.tiles * pl.S;
        }
        kern<<<(unsigned)grid, 64 * C::NW, lds, st>>>((const unsigned char*)B, ldb * 4, (const unsigned char*)Xp, P, pl.stages,
                                                      pl.nst, pl.tiles, pl.ncols_pad, pl.S, logS, pl.pstride, pl.oscale, (float)pl.ascale, pl.accum | (pl.temporal ? 2 : 0));
        SMK_HIP(hipGetLastError());
        return 0;
    }
}

struct F3Variant { int mb, nstage, nwl, fold, wps; };
static const F3Variant kF3Variants[] = {
    {32, 2, 0, 2, 2},   // 100: no loader waves, two workgroups per CU
    {32, 3, 0, 2, 2},   // 101
    {32, 4, 4, 2, 2},   // 102: 4 loader waves, one workgroup per CU, deep ring
    {32, 5, 4, 2, 2},   // 103
    {32, 2, 2, 2, 3},   // 104: 2 loader waves, two workgroups per CU (168 registers)
    {64, 2, 0, 1, 2},   // 105: 64-row stages
    {64, 2, 4, 1, 2},   // 106
    {32, 4, 4, 4, 2},   // 107: fold every 4 stages (128-row chains)
    {32, 2, 0, 4, 2},   // 108
    {32, 4, 2, 2, 2},   // 109
    // software-pipelined kernel (bigprod_f3p_kernel); wps unused
    {32, 4, 4, 4, 0},   // 110: 4 loader waves, 4-deep ring, fold every 4 stages
    {32, 5, 4, 4, 0},   // 111
    {32, 5, 4, 2, 0},   // 112
    {32, 4, 2, 4, 0},   // 113: 2 loader waves
    {32, 5, 2, 4, 0},   // 114
    {32, 3, 2, 4, 0},   // 115: shallow ring, two workgroups per CU fit
    {32, 5, 4, 8, 0},   // 116: fold every 8 stages
    {32, 5, 4, 4, 0},   // 117: as 111 without the pinned interleave (compiler's own schedule)
    {32, 5, 4, 4, 0},   // 118: EXPERIMENT no bf16 split
    {32, 5, 4, 4, 0},   // 119: EXPERIMENT no MFMA
    {32, 5, 4, 4, 0},   // 120: EXPERIMENT neither
    {32, 5, 4, 4, 0},   // 121: loader waves at priority 3
    {32, 5, 4, 4, 0},   // 122: loader priority, compiler schedule
    {32, 5, 4, 4, 0},   // 123: EXPERIMENT no X fetch (full compute)
    {32, 5, 4, 4, 0},   // 124: EXPERIMENT no X fetch, no MFMA, no split
    {32, 2, 0, 8, 2},   // 125: as 108, folding into fp64 every 8 stages (256-row fp32 chains)
    {32, 3, 0, 4, 2},   // 126: as 108 with a 3-deep ring
    {32, 2, 0, 16, 2},  // 127: as 125, folding every 16 stages (512-row fp32 chains)
    {32, 2, 0, 2, 2},   // 128: as 108, folding every 2 stages (64-row fp32 chains)
    {32, 2, 0, 1, 2},   // 129: as 108, folding every stage (32-row fp32 chains: two roundings per accumulator and fold)
};
static const int kNumF3 = (int)(sizeof(kF3Variants) / sizeof(kF3Variants[0]));

template <int KT, int NS, int MB, int NSTAGE, int NWL, int FOLD, int WPS, int FMT = 0, int TAIL = 0>
static int launch_f3_t(const BigProdPlan& pl, const void* B, i64 ldb, const void* Xp, double* P, hipStream_t st)
{
    using C = F3Cfg<KT, MB, NSTAGE, NWL, NS>;
    if constexpr (!C::OK) {
        set_error("bigprod f3 variant does not fit");
        return -100;
    } else {
        constexpr int lds = C::STAGE_BYTES * C::NSTAGE;
        static std::atomic<unsigned long long> attr_set{0};       // per device (DeviceOnce)
        auto kern = bigprod_f3_kernel<KT, MB, NSTAGE, NWL, FOLD, WPS, NS, FMT, TAIL>;
        if (DeviceOnce once{attr_set}) {
            SMK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            once.done();
        }
        int logS = 0;
        while ((1 << logS) < pl.S) ++logS;
        i64 grid;
        if (pl.S <= 8) {
            const i64 per = 8 >> logS;
            grid = (pl.tiles + per - 1) / per * 8;
        } else {
            grid = pl.tiles * pl.S;
        }
        if (TAIL == 1) grid += pl.tail_check.part ? 24 : 16;
        InvRide ride;
        if (TAIL != 2 && pl.inv_ride.G && pl.inv_ride.k > 16 && pl.inv_ride.k <= 64) { ride = pl.inv_ride; grid += 8; }
        kern<<<(unsigned)grid, 64 * C::NW, lds, st>>>((const unsigned char*)B, ldb * 4, (const unsigned char*)Xp, P, pl.stages,
                                                      pl.nst, pl.tiles, pl.ncols_pad, pl.S, logS, pl.pstride, pl.oscale, (float)pl.ascale, pl.accum | (pl.temporal ? 2 : 0),
                                                      pl.tail_gp, pl.tail_nblk, pl.tail_g, pl.tail_check, ride);
        SMK_HIP(hipGetLastError());
        return ride.G ? 1 : 0;                   // 1: the launch carried the Gram inverse
    }
}

template <int KT, int NS>
static int launch_f3(const BigProdPlan& pl, const void* B, i64 ldb, const void* Xp, double* P, hipStream_t st)
{
    switch (pl.variant - 100) {
        case 0: return launch_f3_t<KT, NS, 32, 2, 0, 2, 2>(pl, B, ldb, Xp, P, st);
        case 1: return launch_f3_t<KT, NS, 32, 3, 0, 2, 2>(pl, B, ldb, Xp, P, st);
        case 2: return launch_f3_t<KT, NS, 32, 4, 4, 2, 2>(pl, B, ldb, Xp, P, st);
        case 3: return launch_f3_t<KT, NS, 32, 5, 4, 2, 2>(pl, B, ldb, Xp, P, st);
        case 4: return launch_f3_t<KT, NS, 32, 2, 2, 2, 3>(pl, B, ldb, Xp, P, st);
        case 5: return launch_f3_t<KT, NS, 64, 2, 0, 1, 2>(pl, B, ldb, Xp, P, st);
        case 6: return launch_f3_t<KT, NS, 64, 2, 4, 1, 2>(pl, B, ldb, Xp, P, st);
        case 7: return launch_f3_t<KT, NS, 32, 4, 4, 4, 2>(pl, B, ldb, Xp, P, st);
        case 8: return launch_f3_t<KT, NS, 32, 2, 0, 4, 2>(pl, B, ldb, Xp, P, st);
        case 9: return launch_f3_t<KT, NS, 32, 4, 2, 2, 2>(pl, B, ldb, Xp, P, st);
        case 10: return launch_f3p_t<KT, NS, 4, 4, 4>(pl, B, ldb, Xp, P, st);
        case 11: return launch_f3p_t<KT, NS, 5, 4, 4>(pl, B, ldb, Xp, P, st);
        case 12: return launch_f3p_t<KT, NS, 5, 4, 2>(pl, B, ldb, Xp, P, st);
        case 13: return launch_f3p_t<KT, NS, 4, 2, 4>(pl, B, ldb, Xp, P, st);
        case 14: return launch_f3p_t<KT, NS, 5, 2, 4>(pl, B, ldb, Xp, P, st);
        case 15: return launch_f3p_t<KT, NS, 3, 2, 4>(pl, B, ldb, Xp, P, st);
        case 16: return launch_f3p_t<KT, NS, 5, 4, 8>(pl, B, ldb, Xp, P, st);
        case 17: return launch_f3p_t<KT, NS, 5, 4, 4, 0>(pl, B, ldb, Xp, P, st);
        case 18: return launch_f3p_t<KT, NS, 5, 4, 4, 2>(pl, B, ldb, Xp, P, st);
        case 19: return launch_f3p_t<KT, NS, 5, 4, 4, 4>(pl, B, ldb, Xp, P, st);
        case 20: return launch_f3p_t<KT, NS, 5, 4, 4, 6>(pl, B, ldb, Xp, P, st);
        case 21: return launch_f3p_t<KT, NS, 5, 4, 4, 9>(pl, B, ldb, Xp, P, st);
        case 22: return launch_f3p_t<KT, NS, 5, 4, 4, 8>(pl, B, ldb, Xp, P, st);
        case 23: return launch_f3p_t<KT, NS, 5, 4, 4, 16>(pl, B, ldb, Xp, P, st);
        case 24: return launch_f3p_t<KT, NS, 5, 4, 4, 22>(pl, B, ldb, Xp, P, st);
        case 25: return launch_f3_t<KT, NS, 32, 2, 0, 8, 2>(pl, B, ldb, Xp, P, st);
        case 26: return launch_f3_t<KT, NS, 32, 3, 0, 4, 2>(pl, B, ldb, Xp, P, st);
        case 27: return launch_f3_t<KT, NS, 32, 2, 0, 16, 2>(pl, B, ldb, Xp, P, st);
        default: break;
    }
    set_error("unknown bigprod f3 variant");
    return -100;
}

// the launches that go to bigprod_f3_kernel from the stored transpose (launch_bigprod's dispatch): they can carry the Gram inverse
bool bigprod_supports_ride(const BigProdPlan& pl)
{
    if (pl.nsplit == NSPLIT_F64) return !pl.tr;                  // the accurate form (bigprod_f64_kernel at k > 2), either storage
    if (pl.tr) return false;
    if (pl.storage == STORE_BF16 || pl.variant < 100) return true;  // bigprod_kernel (bf16 storage; fp32 storage in its older forms)
    if (pl.storage != STORE_F32) return false;
    const int v = pl.variant - 100;
    if (pl.nsplit == NSPLIT_F16X2) return v == 8 || (v >= 25 && v <= 29);
    if (pl.nsplit == 2 || pl.nsplit == 3) return (v >= 0 && v <= 9) || (v >= 25 && v <= 27);
    return false;
}

// fp16 two-term form: the variants that won for the bf16 forms
// the variants (fold intervals 4 / 8 / 2 / 1 of the two-workgroup kernel, one k tile) that can carry the Gram reduction
bool bigprod_supports_tail(const BigProdPlan& pl)
{
    return pl.nsplit == NSPLIT_F16X2 && pl.storage == STORE_F32 && pl.kt == 1 &&
           (pl.variant == 108 || pl.variant == 125 || pl.variant == 128 || pl.variant == 129);
}

template <int KT>
static int launch_f3_f16(const BigProdPlan& pl, const void* B, i64 ldb, const void* Xp, double* P, hipStream_t st)
{
    if constexpr (KT == 1) {
        if (pl.tail_nblk > 0) {
            if (!pl.tail_gp || !pl.tail_g) { set_error("bigprod: tail without buffers"); return -100; }
            switch (pl.variant - 100) {
                case 8: return launch_f3_t<KT, 2, 32, 2, 0, 4, 2, 1, 1>(pl, B, ldb, Xp, P, st);
                case 25: return launch_f3_t<KT, 2, 32, 2, 0, 8, 2, 1, 1>(pl, B, ldb, Xp, P, st);
                case 28: return launch_f3_t<KT, 2, 32, 2, 0, 2, 2, 1, 1>(pl, B, ldb, Xp, P, st);
                case 29: return launch_f3_t<KT, 2, 32, 2, 0, 1, 2, 1, 1>(pl, B, ldb, Xp, P, st);
                default: break;
            }
            set_error("bigprod: this variant cannot carry the Gram reduction");
            return -100;
        }
    }
    if (pl.tail_nblk > 0) { set_error("bigprod: this shape cannot carry the Gram reduction"); return -100; }
    switch (pl.variant - 100) {
        case 8: return launch_f3_t<KT, 2, 32, 2, 0, 4, 2, 1>(pl, B, ldb, Xp, P, st);
        case 10: return launch_f3p_t<KT, 2, 4, 4, 4, 1, 1>(pl, B, ldb, Xp, P, st);
        case 11: return launch_f3p_t<KT, 2, 5, 4, 4, 1, 1>(pl, B, ldb, Xp, P, st);
        case 15: return launch_f3p_t<KT, 2, 3, 2, 4, 1, 1>(pl, B, ldb, Xp, P, st);
        case 25: return launch_f3_t<KT, 2, 32, 2, 0, 8, 2, 1>(pl, B, ldb, Xp, P, st);
        case 26: return launch_f3_t<KT, 2, 32, 3, 0, 4, 2, 1>(pl, B, ldb, Xp, P, st);
        case 27: return launch_f3_t<KT, 2, 32, 2, 0, 16, 2, 1>(pl, B, ldb, Xp, P, st);
        case 28: return launch_f3_t<KT, 2, 32, 2, 0, 2, 2, 1>(pl, B, ldb, Xp, P, st);
        case 29: return launch_f3_t<KT, 2, 32, 2, 0, 1, 2, 1>(pl, B, ldb, Xp, P, st);
        default: break;
    }
    set_error("unknown bigprod f16 variant");
    return -100;
}

// ==========================================================================
// The accurate form (NSPLIT_F64): P = X B with B's entries (bf16 / fp32, exact in fp64) against the fp64 factor itself on
// the fp64 matrix cores, fp64 accumulation from the first product on.  About 3x the time of the 16-bit forms (78 TFLOP/s
// of fp64 MFMA against an HBM-bound stream), but the result is the fp64 product of the stored data to rounding (1e-16),
// where the 16-bit forms sit at 1e-8 .. 4e-8 -- which HALS at high rank amplifies past the 1e-4 parity bar in long runs.
// Workgroup = 4 waves, tile = 64 columns (16 per wave) x 64 rows per stage, both operands staged through LDS
// (B: coalesced along the contraction, X: 64 x kg slab); one group of <= 64 factor rows per launch, as the other forms.
// ==========================================================================
// KT16: live 16-row tiles of the group (a template parameter: with a run-time bound on the tile loop the compiler keeps ONE
// accumulator block in AGPRs and copies all four through it around every matrix instruction -- 62 VALU instructions and
// the full result latency per MFMA, 8 TFLOP/s; unrolled it is four independent chains)
template <int EBYTES, int KT16>
__global__ __launch_bounds__(256) void bigprod_f64_kernel(const unsigned char* __restrict__ B, i64 ldb_bytes,
                                                          const double* __restrict__ X, int ldx, int kvalid, i64 len,
                                                          double* __restrict__ P, i64 stages, i64 nst, i64 tiles, i64 ncols_pad,
                                                          int S, int pstride, int ktw, int accum, InvRide ride)
{
    constexpr int MB = 64, NB = 64, KG = 64;
    __shared__ float Bs[NB][MB + 1];
    __shared__ __attribute__((aligned(16))) double Xs[MB][KG + 2];
    i64 bid = blockIdx.x;
    if (ride.G) {            // the Gram inverse of the next block-pivoting launch rides along (common.h: InvRide): workgroup 0 inverts
        if (bid == 0) {
            if (ride.k <= 32) gram_inverse64_body<32>(ride.G, ride.k, ride.Ginv, (int*)(ride.Ginv + 32 * 32), &Xs[0][0]);
            else gram_inverse64_body<64>(ride.G, ride.k, ride.Ginv, (int*)(ride.Ginv + 64 * 64), &Xs[0][0]);
            return;
        }
        --bid;
    }
    const i64 tile = bid / S;
    const int split = (int)(bid % S);
    if (tile >= tiles) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    i64 st0 = (i64)split * nst, st1 = st0 + nst;
    if (st1 > stages) st1 = stages;
    const i64 col0 = tile * NB;
    typedef __attribute__((ext_vector_type(4))) double f64x4;
    f64x4 acc[KT16];
#pragma unroll
    for (int t = 0; t < KT16; ++t) acc[t] = f64x4{0.0, 0.0, 0.0, 0.0};
    const int l15 = lane & 15, l4 = lane >> 4;
    // a stage's operands travel global -> registers -> LDS; the registers of stage st + 1 are loaded while the matrix cores
    // work on stage st.  B tile: consecutive threads take consecutive rows of a column (contiguous in memory; padding rows /
    // columns are zero).  X slab: 64 contraction rows x the group's factor rows (zero beyond the factor / the length).
    float bv[(NB * MB) / 256];
    double xv[(MB * KG) / 256];
    auto fetch = [&](i64 st) {
        const i64 r0 = st * MB;
#pragma unroll
        for (int i = 0; i < (NB * MB) / 256; ++i) {
            const int idx = i * 256 + tid, c = idx / MB, r = idx % MB;
            const unsigned char* src = B + (col0 + c) * ldb_bytes + (r0 + r) * EBYTES;
            if constexpr (EBYTES == 2) bv[i] = bf16_bits_to_f32(*(const unsigned short*)src);
            else bv[i] = *(const float*)src;
        }
#pragma unroll
        for (int i = 0; i < (MB * KG) / 256; ++i) {
            const int idx = i * 256 + tid, r = idx / KG, kk = idx % KG;
            xv[i] = (r0 + r < len && kk < kvalid) ? X[(r0 + r) * ldx + kk] : 0.0;
        }
    };
    if (st0 < st1) fetch(st0);
    for (i64 st = st0; st < st1; ++st) {
#pragma unroll
        for (int i = 0; i < (NB * MB) / 256; ++i) {
            const int idx = i * 256 + tid;
            Bs[idx / MB][idx % MB] = bv[i];
        }
#pragma unroll
        for (int i = 0; i < (MB * KG) / 256; ++i) {
            const int idx = i * 256 + tid;
            Xs[idx / KG][idx % KG] = xv[i];
        }
        __syncthreads();
        if (st + 1 < st1) fetch(st + 1);
#pragma unroll
        for (int q = 0; q < MB / 4; ++q) {
            const double b = (double)Bs[16 * wave + l15][4 * q + l4];
#pragma unroll
            for (int t = 0; t < KT16; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(Xs[4 * q + l4][16 * t + l15], b, acc[t], 0, 0, 0);
        }
        __syncthreads();
    }
    // D: column = lane & 15 (column of B), row = (lane >> 4) + 4 reg (factor row inside the 16-row tile)
    double* pout = P + ((i64)split * ncols_pad + col0 + 16 * wave + l15) * pstride;
    // ktw 16-row tiles are written: the group's 32-row k tiles in full (rows past the live ones as zeros, like the other forms)
#pragma unroll
    for (int t = 0; t < 4; ++t)
        if (t < ktw)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * t + l4 + 4 * r;
                if (pstride < 32 && row >= pstride) continue;             // k <= 16: a column of P holds 8 / 16 rows (kpp_of)
                double v = (t < KT16) ? acc[t < KT16 ? t : 0][r] : 0.0;      // tiles past the live ones are written as zeros
                if (accum) v += pout[row];
                pout[row] = v;
            }
}

// The accurate form for a factor of ONE or TWO rows (dense RANK2: every node factorisation of HierNMF2 on dense A): 2 fp64
// multiply-adds per stored entry are nothing for the vector ALUs (78 TFLOP/s of fp64 against 2 x 1.5 T entries/s at the HBM
// rate), so this product can stream AND be the fp64 product of the stored data to rounding -- no packed operand, no matrix
// cores, no LDS in the loop.  Same plan as bigprod_f64_kernel (64-column plan tiles, 64-row stages, S row splits, P layout);
// a workgroup takes HALF a plan tile: 4 waves x 8 columns.  Lane l owns 4 consecutive rows (one 16-byte fp32 / 8-byte bf16
// load per column) of a block of 256 rows; the 8 column loads and the lane's rows of the factor (from L2: 16 B per row) of
// block b + 1 are issued before the multiply-adds of block b (register double buffer); the factor rows of a block travel
// through a double-buffered LDS slab, one barrier per block.  Two accumulators per column per lane, joined at the end.
// Measured on the C3 matrix (tools/r2_dense_rate.py): fp32 A 6.2 - 6.6 TB/s (the bf16x3 MFMA form: 5.9 - 6.1), bf16 A
// 4.9 - 5.4 (6.1 - 6.2).  (The first version staged the factor rows through LDS with two barriers per block and loaded 8 of 16 columns at
// a time: 4.4 / 3.6 TB/s.)
template <int EBYTES>
__global__ __launch_bounds__(256) void bigprod_f64_k2_kernel(const unsigned char* __restrict__ B, i64 ldb_bytes,
                                                             const double* __restrict__ X, int ldx, int kvalid, i64 len,
                                                             double* __restrict__ P, i64 stages, i64 nst, i64 tiles, i64 ncols_pad,
                                                             int S, int pstride, int ktw, int accum)
{
    constexpr int CW = EBYTES == 2 ? 16 : 8, NBW = 4 * CW;      // columns per wave / per workgroup: 128 bytes in flight per lane and buffer
                                                            // either way (fp32: 8 x 16 B, a workgroup = half a plan tile; bf16: 16 x 8 B, a whole one)
    constexpr int HALVES = 64 / NBW;                        // workgroups per 64-column plan tile
    constexpr int RPL = 4;                                // rows per lane and block: one 16-byte (fp32) / 8-byte (bf16) load per column
    constexpr int RB = 64 * RPL;                          // rows per block
    // 4 rows = 4 words (fp32) or 2 words (bf16: a register pair -- half the registers of the fp32 variant, so more waves cover
    // the narrower loads).  Two fixed typedefs and .x/.y/.z/.w on purpose: with a vector type whose length depended on the
    // template parameter the compiler dropped the element index (every row of a lane read word 0; tools/r2_dense_dbg.py).
    typedef __attribute__((ext_vector_type(4))) unsigned words4_t;
    typedef __attribute__((ext_vector_type(2))) unsigned words2_t;
    typedef typename std::conditional<EBYTES == 4, words4_t, words2_t>::type u32x4v;
    __shared__ double red[NBW][2];
    const i64 half = blockIdx.x / S;                      // this workgroup's NBW columns
    const int split = (int)(blockIdx.x % S);
    if (half >= HALVES * tiles) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    i64 st0 = (i64)split * nst, st1 = st0 + nst;
    if (st1 > stages) st1 = stages;
    const i64 ra = st0 * 64, rz = st1 * 64;               // rows of this split (the matrix is zero padded to whole stages)
    const i64 col0 = half * NBW + CW * wave;
    const unsigned char* bcol = B + col0 * ldb_bytes;
    double acc[CW][2];
#pragma unroll
    for (int c = 0; c < CW; ++c) acc[c][0] = acc[c][1] = 0.0;

    // the factor rows of a block go through LDS (one row per thread per 256 rows, double buffered: ONE barrier per block);
    // the column loads of block b + 1 are in flight while block b is consumed
    __shared__ double xs[2][RB][2];
    u32x4v v[2][CW];
    auto fetch = [&](int buf, i64 rb) {
        const i64 r = rb + (i64)RPL * lane;
        const bool rows_in = r < rz;                      // RPL divides 64: a lane's rows are inside the split or outside together
#pragma unroll
        for (int c = 0; c < CW; ++c) {
            v[buf][c] = u32x4v{};
            if (rows_in) v[buf][c] = __builtin_nontemporal_load((const u32x4v*)(bcol + c * ldb_bytes + r * EBYTES));
        }
    };
    double xr[RB / 256][2];
    auto load_x = [&](i64 rb) {
#pragma unroll
        for (int u = 0; u < RB / 256; ++u) {
            const i64 r = rb + tid + 256 * u;
            const bool live = r < len && r < rz;
            xr[u][0] = live ? X[r * ldx] : 0.0;
            xr[u][1] = (live && kvalid > 1) ? X[r * ldx + 1] : 0.0;
        }
    };
    auto store_x = [&](int buf) {
#pragma unroll
        for (int u = 0; u < RB / 256; ++u) { xs[buf][tid + 256 * u][0] = xr[u][0]; xs[buf][tid + 256 * u][1] = xr[u][1]; }
    };
    auto consume = [&](int buf) {
        double x[RPL][2];
#pragma unroll
        for (int i = 0; i < RPL; ++i) { x[i][0] = xs[buf][RPL * lane + i][0]; x[i][1] = xs[buf][RPL * lane + i][1]; }
#pragma unroll
        for (int c = 0; c < CW; ++c)
#pragma unroll
            for (int i = 0; i < RPL; ++i) {
                double b;
                if constexpr (EBYTES == 2) {
                    const unsigned w = (i >> 1) ? v[buf][c].y : v[buf][c].x;
                    b = (double)__builtin_bit_cast(float, (i & 1) ? (w & 0xFFFF0000u) : (w << 16));
                } else {
                    const unsigned w = i == 0 ? v[buf][c].x : i == 1 ? v[buf][c].y : i == 2 ? v[buf][c].z : v[buf][c].w;
                    b = (double)__builtin_bit_cast(float, w);
                }
                acc[c][0] = fma(b, x[i][0], acc[c][0]);
                acc[c][1] = fma(b, x[i][1], acc[c][1]);
            }
    };
    if (ra < rz) { fetch(0, ra); load_x(ra); store_x(0); }
    __syncthreads();
    int cur = 0;
    for (i64 rb = ra; rb < rz; rb += RB) {
        const bool more = rb + RB < rz;
        if (more) {
            if (cur == 0) fetch(1, rb + RB); else fetch(0, rb + RB);
            load_x(rb + RB);
        }
        if (cur == 0) consume(0); else consume(1);
        if (more) store_x(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
    // join the 64 lanes of a column (fixed butterfly order: the result does not depend on scheduling)
#pragma unroll
    for (int c = 0; c < CW; ++c)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            double t = acc[c][j];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
            if (lane == 0) red[CW * wave + c][j] = t;
        }
    __syncthreads();
    if (tid < NBW) {
        double* pout = P + ((i64)split * ncols_pad + half * NBW + tid) * pstride;
        double v0 = red[tid][0], v1 = red[tid][1];
        if (accum) { v0 += pout[0]; v1 += pout[1]; }
        pout[0] = v0;
        pout[1] = v1;
        if (!accum)                        // the rest of the group's 32-row k tiles as zeros, like the other forms
            for (int r = 2; r < 16 * ktw && (pstride >= 32 || r < pstride); ++r) pout[r] = 0.0;
    }
}

static int launch_bigprod_f64(const BigProdPlan& pl, const void* B, i64 ldb, const void* X, double* P, i64 len, hipStream_t st)
{
    const i64 grid = pl.tiles * pl.S;
    if (grid <= 0) return 0;
    // the factor's live rows of this group: kg of them, never past the padded rank
    const int kvalid = pl.kg < pl.ldx - pl.k0 ? pl.kg : pl.ldx - pl.k0;
    const int ktw = 2 * kt_of(pl.kg);                      // 16-row tiles covering the group's 32-row k tiles
    if (kvalid <= 2) {                                     // rank 1 / 2: the vector-ALU kernel at the streaming rate
        if (pl.storage == STORE_BF16)
            bigprod_f64_k2_kernel<2><<<(unsigned)grid, 256, 0, st>>>((const unsigned char*)B, ldb * 2, (const double*)X, pl.ldx, kvalid, len, P,
                                                                    pl.stages, pl.nst, pl.tiles, pl.ncols_pad, pl.S, pl.pstride, ktw, pl.accum);
        else
            bigprod_f64_k2_kernel<4><<<(unsigned)(2 * grid), 256, 0, st>>>((const unsigned char*)B, ldb * 4, (const double*)X, pl.ldx, kvalid, len, P,
                                                                    pl.stages, pl.nst, pl.tiles, pl.ncols_pad, pl.S, pl.pstride, ktw, pl.accum);
        SMK_HIP(hipGetLastError());
        return 0;
    }
    const int kt16 = (kvalid + 15) / 16;                     // live 16-row tiles of this group (1 .. 4)
    InvRide ride;
    if (pl.inv_ride.G && pl.inv_ride.k > 16 && pl.inv_ride.k <= 64) ride = pl.inv_ride;
    const i64 grid_r = grid + (ride.G ? 1 : 0);
#define SMK_F64(EB, T) bigprod_f64_kernel<EB, T><<<(unsigned)grid_r, 256, 0, st>>>((const unsigned char*)B, ldb * EB, (const double*)X, pl.ldx, kvalid, \
                                                                              len, P, pl.stages, pl.nst, pl.tiles, pl.ncols_pad, pl.S, pl.pstride, ktw, pl.accum, ride)
    if (pl.storage == STORE_BF16) {
        switch (kt16) { case 1: SMK_F64(2, 1); break; case 2: SMK_F64(2, 2); break; case 3: SMK_F64(2, 3); break; default: SMK_F64(2, 4); break; }
    } else {
        switch (kt16) { case 1: SMK_F64(4, 1); break; case 2: SMK_F64(4, 2); break; case 3: SMK_F64(4, 3); break; default: SMK_F64(4, 4); break; }
    }
#undef SMK_F64
    SMK_HIP(hipGetLastError());
    return ride.G ? 1 : 0;                       // 1: the launch carried the Gram inverse
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

int plan_bigprod_groups(int storage, int k, i64 len, i64 ncols, int nsplit, int num_cus, BigProdPlan* out)
{
    int ng = 0;
    size_t off = 0;
    const int pstride = kpp_of(k);
    for (int k0 = 0; k0 < k; k0 += 64, ++ng) {
        const int kg = k - k0 < 64 ? k - k0 : 64;
        BigProdPlan pl = plan_bigprod(storage, kg, len, ncols, nsplit, num_cus);
        pl.k0 = k0; pl.kg = kg; pl.pstride = pstride; pl.pack_offset = off;
        if (ng > 0) {                              // one P layout for all groups: the first group's row splits
            pl.S = out[0].S;
            pl.nst = (pl.stages + pl.S - 1) / pl.S;
        }
        off += packed_bytes(storage, kg, len, nsplit);
        out[ng] = pl;
    }
    for (int g = 0; g < ng; ++g) out[g].p_elems = (size_t)out[0].S * out[0].ncols_pad * pstride;
    return ng;
}

// the H*A' pass of a single-copy (bf16) matrix: same geometry as the stored-transpose pass -- 128 rows of A per tile, 64 columns
// of A per stage -- on the two kernel shapes that exist for the transposed source (k <= 32: 4 waves, 2-deep ring, two workgroups
// per CU; k in (32, 64]: 8 compute + 4 loader waves, 3-deep)
int plan_bigprod_groups_tr(int storage, int k, i64 len, i64 ncols, int nsplit, int num_cus, BigProdPlan* out)
{
    int ng = 0;
    size_t off = 0;
    const int pstride = kpp_of(k);
    const bool f32 = storage == STORE_F32;
    // forms the transposed-source kernels exist for (fp32 A: the fp16 two-term form or bf16x3; bf16 A: 1 .. 3 bf16 terms).  Anything
    // else -- the accurate form above all -- must not be re-labelled here: the caller would then hand an fp64 factor to a kernel
    // that reads packed 16-bit fragments (plan_products materialises the stored transpose instead)
    if (f32 ? (nsplit != NSPLIT_F16X2 && nsplit != 3) : (nsplit < 1 || nsplit > 3)) return -1;
    for (int k0 = 0; k0 < k; k0 += 64, ++ng) {
        const int kg = k - k0 < 64 ? k - k0 : 64;
        BigProdPlan pl;
        pl.storage = storage;
        pl.kt = kt_of(kg);
        pl.k0 = k0; pl.kg = kg; pl.pstride = pstride; pl.pack_offset = off;
        pl.nsplit = nsplit;
        pl.tr = 1;
        if (f32) {
            // the two-workgroup shape of bigprod_f3_kernel (32-column stages, 2-deep ring); the fold interval follows the
            // contraction length as in plan_bigprod
            pl.variant = len >= 65536 ? 125 : len >= 16384 ? 108 : len >= 4096 ? 128 : 129;
            pl.mb = 32;
        } else {
            pl.variant = pl.kt == 2 ? 18 : 6;
            pl.mb = 64;
        }
        pl.nb = 128;
        pl.len = len;
        pl.stages = (len + pl.mb - 1) / pl.mb;
        pl.tiles = (ncols + 127) / 128;
        pl.ncols_pad = round_up(ncols, COL_PAD);
        int S = 1;
        while (pl.tiles * S < 2 * (i64)num_cus && S < 64 && pl.stages / (2 * S) >= 8) S *= 2;
        const char* envS = getenv("SMK_BP_SPLITS");
        if (envS && atoi(envS) > 0) { S = 1; while (S < atoi(envS) && S < 64) S *= 2; }
        if (ng > 0) S = out[0].S;
        pl.S = S;
        pl.nst = (pl.stages + S - 1) / S;
        off += packed_bytes(storage, kg, len, nsplit);
        out[ng] = pl;
    }
    for (int g = 0; g < ng; ++g) out[g].p_elems = (size_t)out[0].S * out[0].ncols_pad * pstride;
    return ng;
}

BigProdPlan plan_bigprod(int storage, int k, i64 len, i64 ncols, int nsplit, int num_cus)
{
    BigProdPlan pl;
    pl.storage = storage;
    pl.kt = kt_of(k);
    pl.k0 = 0; pl.kg = k; pl.pstride = kpp_of(k); pl.pack_offset = 0;
    // fp32 storage: nsplit 3 selects the bf16x3 emulation (default), 1 the native fp32 MFMA
    pl.nsplit = storage == STORE_BF16 ? nsplit : (nsplit >= 2 ? nsplit : 1);
    if (nsplit == NSPLIT_F64) {            // the accurate form: fp64 matrix cores, 64 x 64 tiles
        pl.nsplit = NSPLIT_F64;
        pl.variant = 200;
        pl.mb = 64; pl.nb = 64;
        pl.stages = (len + 63) / 64;
        pl.tiles = (ncols + 63) / 64;
        pl.ncols_pad = round_up(ncols, COL_PAD);
        int S = 1;
        while (pl.tiles * S < 4 * (i64)num_cus && S < 64 && pl.stages / (2 * S) >= 4) S *= 2;
        pl.S = S;
        pl.nst = (pl.stages + S - 1) / S;
        pl.len = len;
        pl.p_elems = (size_t)S * pl.ncols_pad * pl.pstride;
        return pl;
    }
    // measured best on MI355X: bf16 -> 64-row stages, 2-deep ring, 2 workgroups per CU (C3: 5.98 TB/s)
    int v = (storage == STORE_BF16) ? 6 : 7;
    // k in (32,64]: 8 compute waves (one k tile each) + 4 loader waves
    if (pl.kt == 2) v = (storage == STORE_BF16) ? 18 : 21;
    // fp32 A as bf16 planes: the second-generation kernels (power-bound at ~4.1 TB/s for k = 64)
    if (storage == STORE_F32 && pl.nsplit >= 2) v = (pl.kt == 2) ? 108 : 115;
    // fp16 two-term form: the two-workgroup kernel wins or ties at every shape measured (C2, 32768 x 8192 k = 32, both
    // passes of a C4 shard); the pipelined ones stay selectable (110, 111, 115)
    // The error of this form is the fp32 accumulation of the 22-bit hi * hi products (one rounding per MFMA), so it falls with
    // the length of the fp32 chains (stages between two folds into fp64) and, relative to the result, with the number of
    // folds: the fold interval follows the contraction length so that the product stays at 1e-8 .. 2e-8 of the exact one at
    // every size (profiles/r04_fold_interval_sweep.txt, max over 512 entries against an fp64 host product: len 1500 9.2e-8
    // with 8 stages per fold, 1.9e-8 with 1; len 8192 4.4e-8 -> 1.3e-8 with 2; len 65536 1.7e-8 and len 262144 1.0e-8 with 8).
    // Short contractions are latency-bound, the extra folds cost nothing there; at C4 (len >= 65536) folding every 8 stages
    // instead of 4 is worth 0 .. 2 %.
    if (storage == STORE_F32 && pl.nsplit == NSPLIT_F16X2) v = len >= 65536 ? 125 : len >= 16384 ? 108 : len >= 4096 ? 128 : 129;
    const char* env = getenv("SMK_BP_VARIANT");
    if (env) v = atoi(env);
    const char* env2 = getenv("SMK_BP_VARIANT_K64");       // only for k in (32, 64]
    if (env2 && pl.kt == 2) v = atoi(env2);
    if (storage == STORE_F32 && pl.nsplit == 2 && v < 100) v = (pl.kt == 2) ? 108 : 115;   // the 2-term forms exist only there
    if (pl.nsplit != NSPLIT_F16X2 && (v == 128 || v == 129)) v = 108;       // the short-chain variants exist for the fp16 form only
    if (pl.nsplit == NSPLIT_F16X2 && v != 108 && v != 110 && v != 111 && v != 115 && v != 125 && v != 126 && v != 127 && v != 128 && v != 129) v = 125;
    if (v >= 100 && storage == STORE_F32 && pl.nsplit >= 2) {
        auto f3_fits = [&](int vv) {
            if (vv < 100 || vv >= 100 + kNumF3) return false;
            const F3Variant& f = kF3Variants[vv - 100];
            const int stage = 128 * f.mb * 4 + (f.mb / 16) * pack_terms(pl.nsplit) * pl.kt * 1024;
            const int ti = stage / 1024, nld = f.nwl > 0 ? f.nwl : 4;
            return ti % nld == 0 && (ti / nld) * (f.nstage - 1) <= 63 && stage * f.nstage <= 160 * 1024;
        };
        if (!f3_fits(v)) v = (pl.nsplit == NSPLIT_F16X2) ? 125 : 108;
        if (!f3_fits(v)) v = 115;
        if (!f3_fits(v)) v = 110;
        pl.variant = v;
        const int MB = kF3Variants[v - 100].mb;
        pl.mb = MB; pl.nb = 128;
        pl.stages = (len + MB - 1) / MB;
        pl.tiles = (ncols + 127) / 128;
        pl.ncols_pad = round_up(ncols, COL_PAD);
        int S = 1;
        while (pl.tiles * S < 2 * (i64)num_cus && S < 64 && pl.stages / (2 * S) >= 8) S *= 2;
        const char* envS = getenv("SMK_BP_SPLITS");
        if (envS && atoi(envS) > 0) { S = 1; while (S < atoi(envS) && S < 64) S *= 2; }
        pl.S = S;
        pl.nst = (pl.stages + S - 1) / S;
        pl.p_elems = (size_t)S * pl.ncols_pad * pl.pstride;
        return pl;
    }
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
    pl.mb = MB; pl.nb = NB;
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
    pl.p_elems = (size_t)S * pl.ncols_pad * pl.pstride;   // doubles
    return pl;
}

template <int EBYTES, int KT, int NSPLIT, int MB, int NSTAGE, int CW, int WK, int NWL, bool TRB = false>
static int launch_bigprod_t(const BigProdPlan& pl, const void* B, i64 ldb, const void* Xp, double* P, hipStream_t st)
{
    using C = BPCfg<EBYTES, KT, NSPLIT, MB, NSTAGE, CW, WK, NWL>;
    constexpr int lds = C::STAGE_BYTES * C::NSTAGE;
    static std::atomic<unsigned long long> attr_set{0};       // per device (DeviceOnce)
    auto kern = bigprod_kernel<EBYTES, KT, NSPLIT, MB, NSTAGE, CW, WK, NWL, TRB>;
    if (DeviceOnce once{attr_set}) {
        SMK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        once.done();
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
    InvRide ride;
    if (!TRB && pl.inv_ride.G && pl.inv_ride.k > 16 && pl.inv_ride.k <= 64) { ride = pl.inv_ride; grid += 8; }
    kern<<<(unsigned)grid, 64 * C::NW, lds, st>>>((const unsigned char*)B, ldb * EBYTES, (const unsigned char*)Xp, P,
                                           pl.stages, pl.nst, pl.tiles, pl.ncols_pad, pl.S, logS, pl.pstride, pl.accum | (pl.temporal ? 2 : 0), ride);
    SMK_HIP(hipGetLastError());
    return ride.G ? 1 : 0;                       // 1: the launch carried the Gram inverse
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

// transposed source: B = A (bf16), ldb = its column stride
template <int KT, int NSPLIT, int MB, int NSTAGE, int WK, int NWL>
static int launch_bigprod_if_tr(const BigProdPlan& pl, const void* B, i64 ldb, const void* Xp, double* P, hipStream_t st)
{
    if constexpr (bp_fits(2, KT, NSPLIT, MB, NSTAGE, 1, WK, NWL)) {
        BigProdPlan q = pl;                    // the plan was made for 64-column stages
        q.mb = MB;
        q.stages = (pl.len + MB - 1) / MB;
        q.nst = (q.stages + q.S - 1) / q.S;
        return launch_bigprod_t<2, KT, NSPLIT, MB, NSTAGE, 1, WK, NWL, true>(q, B, ldb, Xp, P, st);
    } else {
        set_error("bigprod (transposed source): this shape does not fit");
        return -100;
    }
}
template <int KT, int NSPLIT>
static int launch_bigprod_tr_v(const BigProdPlan& pl, const void* B, i64 ldb, const void* Xp, double* P, hipStream_t st)
{
    // the kernel shapes built for the transposed source (numbers = the variants of kVariants; SMK_BP_TR_VARIANT picks one)
    static const int v = [] { const char* e = getenv("SMK_BP_TR_VARIANT"); return e ? atoi(e) : -1; }();
    if constexpr (KT == 1) {
        switch (v) {
            case 0: return launch_bigprod_t<2, 1, NSPLIT, 64, 3, 1, 1, 0, true>(pl, B, ldb, Xp, P, st);
            case 1: return launch_bigprod_t<2, 1, NSPLIT, 64, 4, 1, 1, 0, true>(pl, B, ldb, Xp, P, st);
            case 15: return launch_bigprod_t<2, 1, NSPLIT, 64, 3, 1, 1, 4, true>(pl, B, ldb, Xp, P, st);
            case 6: return launch_bigprod_t<2, 1, NSPLIT, 64, 2, 1, 1, 0, true>(pl, B, ldb, Xp, P, st);
            case 17: return launch_bigprod_t<2, 1, NSPLIT, 64, 5, 1, 1, 4, true>(pl, B, ldb, Xp, P, st);
            case 23: return launch_bigprod_t<2, 1, NSPLIT, 64, 2, 1, 1, 4, true>(pl, B, ldb, Xp, P, st);
            case 20: return launch_bigprod_if_tr<1, NSPLIT, 32, 4, 1, 2>(pl, B, ldb, Xp, P, st);
            case 24: return launch_bigprod_if_tr<1, NSPLIT, 32, 5, 1, 2>(pl, B, ldb, Xp, P, st);
            case 25: return launch_bigprod_if_tr<1, NSPLIT, 32, 6, 1, 2>(pl, B, ldb, Xp, P, st);
            case 26: return launch_bigprod_if_tr<1, NSPLIT, 32, 3, 1, 2>(pl, B, ldb, Xp, P, st);
            // 16: four loader waves and a 4-deep ring, one workgroup per CU.  The shape of the stored-transpose pass (6: 2-deep, two
            // workgroups per CU) runs 400 us on C3 against 339; with loader waves 354 - 363 (profiles/r05_single_copy_c3.txt)
            default: return launch_bigprod_t<2, 1, NSPLIT, 64, 4, 1, 1, 4, true>(pl, B, ldb, Xp, P, st);
        }
    } else {
        switch (v) {
            case 11: return launch_bigprod_t<2, 2, NSPLIT, 64, 3, 1, 2, 0, true>(pl, B, ldb, Xp, P, st);
            case 19: return launch_bigprod_t<2, 2, NSPLIT, 64, 4, 1, 2, 4, true>(pl, B, ldb, Xp, P, st);
            default: return launch_bigprod_t<2, 2, NSPLIT, 64, 3, 1, 2, 4, true>(pl, B, ldb, Xp, P, st);      // 18
        }
    }
}
template <int KT>
static int launch_bigprod_tr(const BigProdPlan& pl, const void* B, i64 ldb, const void* Xp, double* P, hipStream_t st)
{
    if (pl.nsplit == 3) return launch_bigprod_tr_v<KT, 3>(pl, B, ldb, Xp, P, st);
    if (pl.nsplit == 2) return launch_bigprod_tr_v<KT, 2>(pl, B, ldb, Xp, P, st);
    return launch_bigprod_tr_v<KT, 1>(pl, B, ldb, Xp, P, st);
}

// transposed source, fp32 storage: bigprod_f3_kernel<..., TAIL = 2> in its two-workgroup shape, fold interval by variant
template <int KT>
static int launch_f3_tr(const BigProdPlan& pl, const void* B, i64 ldb, const void* Xp, double* P, hipStream_t st)
{
    if (pl.nsplit == NSPLIT_F16X2) {
        if (!pl.oscale) { set_error("bigprod: the fp16 two-term form needs the output scales"); return -100; }
        switch (pl.variant) {
            case 125: return launch_f3_t<KT, 2, 32, 2, 0, 8, 2, 1, 2>(pl, B, ldb, Xp, P, st);
            case 128: return launch_f3_t<KT, 2, 32, 2, 0, 2, 2, 1, 2>(pl, B, ldb, Xp, P, st);
            case 129: return launch_f3_t<KT, 2, 32, 2, 0, 1, 2, 1, 2>(pl, B, ldb, Xp, P, st);
            default: return launch_f3_t<KT, 2, 32, 2, 0, 4, 2, 1, 2>(pl, B, ldb, Xp, P, st);
        }
    }
    // bf16x3: a stage is 16 KB of A + 6 KB (k <= 32) or 12 KB of operand fragments; 22 pieces do not split over four loading
    // waves, so k <= 32 takes the shape with two loader waves (variant 104's), k in (32, 64] the plain one
    constexpr int NWL = KT == 1 ? 2 : 0, WPS = KT == 1 ? 3 : 2;
    switch (pl.variant) {
        case 125: return launch_f3_t<KT, 3, 32, 2, NWL, 8, WPS, 0, 2>(pl, B, ldb, Xp, P, st);
        case 128: return launch_f3_t<KT, 3, 32, 2, NWL, 2, WPS, 0, 2>(pl, B, ldb, Xp, P, st);
        case 129: return launch_f3_t<KT, 3, 32, 2, NWL, 1, WPS, 0, 2>(pl, B, ldb, Xp, P, st);
        default: return launch_f3_t<KT, 3, 32, 2, NWL, 4, WPS, 0, 2>(pl, B, ldb, Xp, P, st);
    }
}

int launch_bigprod(const BigProdPlan& pl, const void* B, i64 ldb, const void* Xp, double* P, hipStream_t st)
{
    if (pl.tr) {
        if (pl.storage == STORE_F32) {
            if (pl.nsplit != 3 && pl.nsplit != NSPLIT_F16X2) { set_error("bigprod: the transposed source of fp32 storage takes the fp16 two-term form or bf16x3"); return -100; }
            return pl.kt == 1 ? launch_f3_tr<1>(pl, B, ldb, Xp, P, st) : launch_f3_tr<2>(pl, B, ldb, Xp, P, st);
        }
        if (pl.nsplit < 1 || pl.nsplit > 3) { set_error("bigprod: the transposed source of bf16 storage takes 1 .. 3 operand terms"); return -100; }
        return pl.kt == 1 ? launch_bigprod_tr<1>(pl, B, ldb, Xp, P, st) : launch_bigprod_tr<2>(pl, B, ldb, Xp, P, st);
    }
    if (pl.nsplit == NSPLIT_F64) return launch_bigprod_f64(pl, B, ldb, Xp, P, pl.len, st);
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
        if (pl.variant >= 100) {
            if (pl.nsplit == NSPLIT_F16X2) {
                if (!pl.oscale) { set_error("bigprod: the fp16 two-term form needs the output scales"); return -100; }
                return pl.kt == 1 ? launch_f3_f16<1>(pl, B, ldb, Xp, P, st) : launch_f3_f16<2>(pl, B, ldb, Xp, P, st);
            }
            if (pl.nsplit == 3) return pl.kt == 1 ? launch_f3<1, 3>(pl, B, ldb, Xp, P, st) : launch_f3<2, 3>(pl, B, ldb, Xp, P, st);
            return pl.kt == 1 ? launch_f3<1, 2>(pl, B, ldb, Xp, P, st) : launch_f3<2, 2>(pl, B, ldb, Xp, P, st);
        }
        if (pl.nsplit == 3) {
            if (pl.kt == 1) return launch_bigprod_v<4, 1, 3>(pl, B, ldb, Xp, P, st);
            return launch_bigprod_v<4, 2, 3>(pl, B, ldb, Xp, P, st);
        }
        if (pl.kt == 1) return launch_bigprod_v<4, 1, 1>(pl, B, ldb, Xp, P, st);
        return launch_bigprod_v<4, 2, 1>(pl, B, ldb, Xp, P, st);
    }
}

}  // namespace smk
