// smallk_amd/csrc/bigprod.hip -- the streaming products W'A and H*At (the dominant kernels) and the
// packing of their skinny operand.  See kernels.hip for the kernel map.
#include "devutil.h"

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

}  // namespace smk
