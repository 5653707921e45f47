// bf16 MFMA GEMM for gfx950 with fused epilogue (bias / exact GELU / residual-or-accumulate) and split-K.
//
// One kernel template covers the three operand forms a Linear layer needs (include/mi355_vlm.h):
//   NT  y  = x W^T      both operands K-contiguous            -> ds_read_b128 fragments
//   NN  dx = dy W       B is K-strided ([K][N] row-major)     -> ds_read_b64_tr_b16 fragments for B
//   TN  dW = dy^T x     A and B K-strided                     -> transposing reads for both
// so no operand is ever transposed in HBM.
//
// Structure (tile BM x BN x 64, WM x WN waves, every wave owns (BM/WM) x (BN/WN) outputs as 16x16x32 MFMA tiles):
//   * HBM -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds, 1 KiB per wave-instruction), two LDS stages, the
//     next K-tile's DMA in flight while the current one feeds the MFMAs; one barrier per K-tile.
//   * The DMA destination is lane-linear, so the bank-conflict swizzle is applied on the per-lane SOURCE
//     address and again on the fragment read (both-sides rule): row-major-K tiles use 128-B rows with
//     chunk' = chunk ^ ((row>>1)&7); K-strided tiles use (2*cols)-byte rows with chunk' = chunk ^ (f(k)<<1).
//   * Out-of-range rows/cols/K are zero-filled by the buffer range check (offset 0x80000000 > num_records),
//     so any M and any K,N multiple of 8 work without a tail path in the main loop.
//   * The kernel is LDS-bandwidth-bound at 64x64 per wave (fragment reads + DMA writes ~ MFMA time), so the large
//     configurations give every wave 128x64 outputs: 25 % fewer fragment bytes and half the DMA bytes per MFMA.
//   * Epilogue: accumulators -> LDS (fp32, 64x64 at a time per wave) -> row-contiguous 16-B global stores with
//     bias/GELU/residual fused, or raw fp32 slabs when K is split (few output tiles + long K: weight gradients).
//   * Workgroup -> tile map: XCD-aware (blocks b and b+8 share an L2) then 3-row super-groups.
#include <type_traits>

#include "common.h"

namespace {

// -DGEMM_SWAP=1 (round-3 experiment, NOT the default): the B-matrix fragment as the MFMA's first operand, so that an accumulator tile holds four consecutive
// COLUMNS of one row per lane and the epilogue's staging writes are 16 16-byte LDS stores per 64x64 sub-block instead of 64 4-byte ones.  Correct (all GEMM tests),
// level in isolation (gate-up + SwiGLU 606.6 against 606 us, plain 523) and SLOWER in the step: 192.7 / 193.1 / 192.6 against 190.6 / 190.9 / 191.1 ms in three
// same-box pairs -- the main loops are the same instructions with the operand registers exchanged, and that is enough to move the step by 1 %.
#ifndef GEMM_SWAP
#define GEMM_SWAP 0
#endif
#ifndef GEMM_C_AUX_WALK
#define GEMM_C_AUX_WALK 2
#endif
#ifndef GEMM_C_AUX_OLD
#define GEMM_C_AUX_OLD 0
#endif
#ifndef MI355_GEMM_WALK_DEFAULT
#define MI355_GEMM_WALK_DEFAULT 2  // see walk_on()
#endif
#ifndef MI355_GEMM_PP_DEFAULT
#define MI355_GEMM_PP_DEFAULT 0  // see pp_mask()
#endif
#if GEMM_SWAP
#define MFMA_CT(af, bf, c) (c) = __builtin_amdgcn_mfma_f32_16x16x32_bf16((bf), (af), (c), 0, 0, 0)
#else
#define MFMA_CT(af, bf, c) (c) = __builtin_amdgcn_mfma_f32_16x16x32_bf16((af), (bf), (c), 0, 0, 0)
#endif
constexpr int EPI_LD = 68;             // fp32 row pitch of the epilogue staging (bank-spread, 16-B aligned)
constexpr unsigned OOB = 0x80000000u;  // beyond num_records (0x7fffffff): load returns zeros

// BK_ = K extent of one LDS stage, NS_ = stages of the ring: the DMA of K-tile t+NS-1 is issued while tile t is being
// multiplied, and the wait in front of the per-tile barrier is a COUNTED vmcnt that leaves the NS-2 youngest tiles in
// flight (a vmcnt(0) there -- what __syncthreads() emits -- is the ceiling of the simple structure).
template <int BM_, int BN_, int WM_, int WN_, int BK_, int NS_>
struct TileCfg {
    static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, BK = BK_, NS = NS_, NW = WM_ * WN_, NTHREADS = NW * 64;
    static constexpr int WTM = BM / WM, WTN = BN / WN;  // wave tile
    static constexpr int FM = WTM / 16, FN = WTN / 16;  // 16x16 accumulator tiles per wave
    static constexpr int KK = BK / 32;                  // MFMA k-steps per stage
    static constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
    static constexpr int A_PPW = A_BYTES / 1024 / NW, B_PPW = B_BYTES / 1024 / NW;  // DMA pieces per wave per K-tile
    static constexpr int DMA_PER_TILE = A_PPW + B_PPW;
    static constexpr int EPI_BYTES = NW * 64 * EPI_LD * 4;
    static constexpr int SMEM = NS * STAGE > EPI_BYTES ? NS * STAGE : EPI_BYTES;
    static constexpr int MIN_WAVES = (NW == 8) ? 2 : (SMEM <= 80 * 1024 ? 2 : 1);
    static_assert(BK == 32 || BK == 64, "BK must be 32 or 64");
    static_assert(WTM % 64 == 0 && WTN % 64 == 0, "wave tile must be a multiple of 64x64");
    static_assert(A_BYTES % (1024 * NW) == 0 && B_BYTES % (1024 * NW) == 0, "pieces must divide evenly over the waves");
    static_assert((NS - 2) * DMA_PER_TILE <= 63, "vmcnt field is 6 bits");
};
using Cfg128 = TileCfg<128, 128, 2, 2, 64, 2>;      // 68 KiB LDS, 2 workgroups / CU, vmcnt(0) structure
using Cfg256 = TileCfg<256, 256, 2, 4, 64, 2>;      // 136 KiB LDS, 1 workgroup (8 waves) / CU
using Cfg256a = TileCfg<256, 256, 2, 4, 32, 4>;     // 136 KiB LDS, 4-stage ring, alternating wave groups
using Cfg256b = TileCfg<256, 256, 2, 4, 32, 5>;     // 160 KiB LDS (all of it), 5-stage ring, ONE barrier per phase (tile hint 4)
using Cfg256w = TileCfg<256, 256, 2, 2, 32, 4>;     // 128 KiB LDS, FOUR waves of 128x128 (one per SIMD, 512 registers each), tile hint 5

#ifndef GEMM_PROF
#define GEMM_PROF 0  // profiling builds only: in-kernel cycle stamps of the alternating loop (tools/gemm_prof.py)
#endif
#if GEMM_PROF
__device__ unsigned long long g_gemm_prof[32];
#define GP_T() __builtin_readcyclecounter()
#define GP_ADD(i, t0) do { const unsigned long long now_ = __builtin_readcyclecounter(); gp[i] += now_ - (t0); (t0) = now_; } while (0)
#else
#define GP_T() 0ull
#define GP_ADD(i, t0) do { } while (0)
#endif

#ifndef GEMM_TL
#define GEMM_TL 0  // profiling builds only: per-workgroup wall-clock stamps (100 MHz s_memrealtime) of prologue / main loop / write-out, tools/gemm_timeline.py
#endif
#if GEMM_TL
__device__ unsigned long long g_gemm_tl[32768][8];
#define TL_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 32768) g_gemm_tl[blockIdx.x][i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define TL_STAMP(i) do { } while (0)
#endif

struct GemmParams {
    const bf16_t* A;
    const bf16_t* B;
    void* C;
    const float* bias;
    const void* R;
    int64_t M, N, K, lda, ldb, ldc, ldr;
    int tiles_m, tiles_n, epilogue;
    int ksplit;  // > 1: K is split over ksplit workgroups per tile, each writes an fp32 slab into ws
    float* ws;   // [ksplit][M][N] fp32
    int ablate;  // profiling only (tile_hint >> 8): 1 = no DMA after the prologue, 2 = no fragment reads after the first, 4 = no barrier
    // MI355_EPI_ATTN_DELTA (mi355_gemm_bf16_attn_delta): C = d(ctx) of an attention block, R = the forward's ctx; the epilogue also leaves the
    // attention backward's row constants delta[b, h, s] = sum_d dctx * ctx, -delta and -lse * log2(e), each fp32 [B, Hq, S]
    const float* ad_lse;
    float* ad_delta;
    float* ad_nl2;
    float* ad_ndl;
    int ad_S, ad_Hq;
    int split3;  // fp32 output kernels only: C is bf16 [M, 3N], the value as [hi | lo | hi] (MI355_DT_SPLIT3)
};

__device__ __forceinline__ float gelu_erf(float x) { return x * 0.5f * (1.0f + erff(x * 0.70710678118654752440f)); }

__device__ __forceinline__ int tr_f(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }

// swizzle of a row-major-K image: 128-B rows (BK 64): chunk ^ ((row>>1)&7); 64-B rows (BK 32): chunk ^ ((-(row>>2))&3)
template <int BKc>
__device__ __forceinline__ int swz_rowk(int chunk, int row) {
    return BKc == 64 ? chunk ^ ((row >> 1) & 7) : chunk ^ ((0 - (row >> 2)) & 3);
}

// Per-lane byte offsets (relative to the tile's corner) of this wave's DMA pieces of one operand tile, plus the
// K-extent each piece needs for validity.  Non-TR: tile [EXT rows][BK k]; TR: tile [BK k][EXT cols] (2*EXT-byte rows).
template <bool TR, int EXT, int BKc, int PPW>
__device__ __forceinline__ void piece_offsets(int wave, int lane, int64_t ld, int64_t ext_left, unsigned (&voff)[PPW], int (&kneed)[PPW]) {
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int pi = wave * PPW + j;
        if constexpr (!TR) {
            constexpr int CH = BKc * 2 / 16, RPP = 64 / CH;  // chunks per row, rows per 1-KiB piece
            const int r = pi * RPP + lane / CH;
            const int c = swz_rowk<BKc>(lane % CH, r);
            voff[j] = (r < ext_left) ? (unsigned)(r * ld * 2 + c * 16) : OOB;
            kneed[j] = c * 8;  // valid iff kneed < K - k0
        } else {
            constexpr int CH = EXT * 2 / 16, RPP = 64 / CH;
            const int kr = pi * RPP + lane / CH;
            const int c = (lane % CH) ^ (tr_f(kr) << 1);
            voff[j] = (c * 8 < ext_left) ? (unsigned)(kr * ld * 2 + c * 16) : OOB;
            kneed[j] = kr;
        }
    }
}

// A pointer every lane agrees on, moved to scalar registers.  The buffer resource of an LDS-DMA must be scalar; a base pointer that went
// through a select (the SwiGLU-forward row remap picks another B base) otherwise reaches the DMA in vector registers and every piece
// is wrapped in a readfirstlane waterfall loop.
__device__ __forceinline__ const bf16_t* uniform_ptr(const bf16_t* ptr) {
    const unsigned long long v = (unsigned long long)ptr;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const bf16_t*)(((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ void dma_piece(const void* base, unsigned voff, char* lds_dst) {
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(lds_dst), 16, voff, 0, 0, 0);
}

// fragment of a row-major-K tile: 16 rows starting at r0, k-step kk (32 wide)
template <int BKc>
__device__ __forceinline__ bf16x8 frag_rowk(const char* tile, int r0, int kk, int lane) {
    const int r = r0 + (lane & 15);
    const int c = kk * 4 + (lane >> 4);
    return *reinterpret_cast<const bf16x8*>(tile + r * (BKc * 2) + (swz_rowk<BKc>(c, r) << 4));
}

// fragment of a K-strided tile [BK k][EXT cols]: 16 cols starting at c0, k-step kk, via two transposing reads
template <int EXT>
__device__ __forceinline__ bf16x8 frag_tr(const char* tile, int c0, int kk, int lane) {
    constexpr int ROWB = EXT * 2;
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int f = q | ((g & 1) << 2);
    const int chunk = ((c0 >> 3) + (p >> 1)) ^ (f << 1);
    const int row = kk * 32 + 8 * g + q;
    const char* a0 = tile + row * ROWB + (chunk << 4) + (p & 1) * 8;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(a0));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(a0 + 4 * ROWB));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

// The same fragment requested from an asm statement.  hipcc (ROCm 7.2) puts `s_waitcnt vmcnt(0)` in front of every ds_read_tr16_b64 INTRINSIC
// that follows an LDS-DMA (it cannot tell the read from the DMA's destination; plain ds_read_b128 loads are not affected), which drains the
// whole prefetch ring in every phase of the K-strided forms.  An asm read is invisible to that logic -- and to the compiler's lgkmcnt
// bookkeeping, so the two 64-bit halves stay separate values until `tr_join`, which runs after a wait that names them.
struct TrHalves {
    bf16x4 lo, hi;
};
template <int EXT>
__device__ __forceinline__ void frag_tr_issue(TrHalves& f, const char* tile, int c0, int kk, int lane) {
    constexpr int ROWB = EXT * 2;
    static_assert(4 * ROWB < 65536, "ds offset field is 16 bits");
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int fq = q | ((g & 1) << 2);
    const int chunk = ((c0 >> 3) + (p >> 1)) ^ (fq << 1);
    const int row = kk * 32 + 8 * g + q;
    const unsigned addr = (unsigned)(uintptr_t)LDS_PTR(tile + row * ROWB + (chunk << 4) + (p & 1) * 8);
    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:%c3" : "=&v"(f.lo), "=&v"(f.hi) : "v"(addr), "i"(4 * ROWB));
}
__device__ __forceinline__ bf16x8 tr_join(const TrHalves& f) { return __builtin_shufflevector(f.lo, f.hi, 0, 1, 2, 3, 4, 5, 6, 7); }
#define TRH(x) "+v"((x).lo), "+v"((x).hi)
// counted wait that names the fragments it covers (N = LDS instructions issued after them; LDS returns in issue order), then joins them
template <int N, int CNT>
__device__ __forceinline__ void tr_wait_join(TrHalves (&h)[CNT], bf16x8 (&out)[CNT]) {
    static_assert(N >= 0 && N <= 15 && CNT % 2 == 0, "lgkmcnt field is 4 bits; fragments are named in pairs");
#pragma unroll
    for (int k = 0; k < CNT; k += 2) asm volatile("s_waitcnt lgkmcnt(%4)" : TRH(h[k]), TRH(h[k + 1]) : "n"(N) : "memory");
#pragma unroll
    for (int k = 0; k < CNT; ++k) out[k] = tr_join(h[k]);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else static_assert(N == 0, "add the immediate");
}

// position b of a round-robin-over-XCDs numbering -> position in a numbering where each XCD owns one contiguous chunk
__device__ __forceinline__ int xcd_chunked(int b, int n) {
    const int xcd = b & 7, q = n >> 3, r = n & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}

// One output tile (and one K split) of one problem.  `pid` = tile index in XCD-chunked order, `split` = K-split index.
template <class T, bool A_TR, bool B_TR, int OUT_DT>
__device__ __forceinline__ void gemm_tile(const GemmParams& p, const int pid, const int split, char* smem) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    TL_STAMP(0);
#if GEMM_TL
    if (threadIdx.x == 0 && blockIdx.x < 32768) {
        g_gemm_tl[blockIdx.x][6] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | __builtin_amdgcn_s_getreg((31 << 11) | 4);  // XCC_ID, HW_ID
        g_gemm_tl[blockIdx.x][7] = (unsigned long long)pid;
    }
#endif

    // ---- tile index -> (tm, tn): row super-groups ----------------------------------------------------
    // super-group height by in-step A/B on the VLM step (ms/step, two runs each): 2: 245.9 / 246.2, 3: 243.8 / 243.8, 4: 245.3 / 245.3, 6: 243.8 / 244.3,
    // 8: 245.2 / 245.6, 16: 249.7 / 248.0.  Profiling bits 5-7 of the tile hint select another height.
    const int gsel = (p.ablate >> 5) & 7;
    // Round 3, NT gate-up shape alone (profiles/r03_tile_walk.json): fabric reads 1 383 / 1 116 / 996 / 946 / 1 010 / 1 390 MB at heights 2 / 3 / 4 / 6 / 8 / 16, time
    // 602 / 595 / 582 / 584 / 583 / 585 us -- the walk moves the re-fetch by 1.5x and the time by 2 %; in the step, all NT launches at 6 against 3: 194.5 / 194.6 against
    // 194.9 / 195.2 ms.  NT launches therefore walk 6-row groups, the K-strided forms stay at 3 (selector 6 = 3 rows for A/B runs).
    const int GROUP_M = gsel == 1 ? 4 : gsel == 2 ? 16 : gsel == 3 ? 2 : gsel == 4 ? 6 : gsel == 5 ? 8 : gsel == 6 ? 3 : (!A_TR && !B_TR) ? 6 : 3;
    const int in_group = GROUP_M * p.tiles_n;
    const int first_m = (pid / in_group) * GROUP_M;
    const int gsz = min(p.tiles_m - first_m, GROUP_M);
    const int tm = first_m + (pid % in_group) % gsz;
    const int tn = (pid % in_group) / gsz;
    const int64_t m0 = (int64_t)tm * T::BM, n0 = (int64_t)tn * T::BN;

    // ---- DMA plan -------------------------------------------------------------------------------------
    unsigned voffA[T::A_PPW], voffB[T::B_PPW];
    int kneedA[T::A_PPW], kneedB[T::B_PPW];
    constexpr int BK = T::BK;
    // SPLIT (NT on the 8-wave 256x256 tile): the requests of a SIMD's two waves (w and w + 4) are divided by OPERAND, as in gemm_nt_persist_kernel -- waves 0-3 request
    // the A-panel pieces of both (their own in the "A" set below, their partner's in the "B" set), behind the barrier; waves 4-7 the B-panel pieces of both, a phase later
    constexpr bool SPLIT = T::NW == 8 && T::BK == 64 && !A_TR && !B_TR && T::A_PPW == T::B_PPW;
    const bool a_side = !SPLIT || wave < 4;
    const int w1 = SPLIT ? (wave & 3) : wave, w2 = SPLIT ? (wave & 3) + 4 : wave;  // whose pieces the first / second set holds
    if (!SPLIT || a_side) piece_offsets<A_TR, T::BM, BK, T::A_PPW>(w1, lane, p.lda, p.M - m0, voffA, kneedA);
    else piece_offsets<B_TR, T::BN, BK, T::B_PPW>(w1, lane, p.ldb, p.N - n0, voffA, kneedA);
    if (SPLIT && a_side) piece_offsets<A_TR, T::BM, BK, T::A_PPW>(w2, lane, p.lda, p.M - m0, voffB, kneedB);
    else piece_offsets<B_TR, T::BN, BK, T::B_PPW>(w2, lane, p.ldb, p.N - n0, voffB, kneedB);
    const bf16_t* baseA = A_TR ? p.A + m0 : p.A + m0 * p.lda;
    const bf16_t* baseB = B_TR ? p.B + n0 : p.B + n0 * p.ldb;
    if constexpr (!B_TR) {
        if (p.epilogue == MI355_EPI_SWIGLU_FWD && !(SPLIT && a_side)) {
            // gate-up projection with the activation in the epilogue: every 64 output columns of the tile are [32 lin1 rows | the 32
            // lin_gate rows of the SAME hidden units] of the fused weight [lin1 (N/2 rows) | lin_gate (N/2 rows)], so a wave's 64x64
            // staging block holds u and g side by side.  Only the row each DMA lane fetches changes.
            constexpr int CH = BK * 2 / 16, RPP = 64 / CH;
            const int64_t nh = p.N >> 1;
#pragma unroll
            for (int j = 0; j < T::B_PPW; ++j) {
                const int r = (w2 * T::B_PPW + j) * RPP + lane / CH;
                const int c = swz_rowk<BK>(lane % CH, r);
                const int64_t hid = (n0 >> 1) + (r >> 6) * 32 + (r & 31);
                const int64_t grow = ((r >> 5) & 1) * nh + hid;
                voffB[j] = hid < nh ? (unsigned)(grow * p.ldb * 2 + c * 16) : OOB;
                kneedB[j] = c * 8;
            }
            if constexpr (SPLIT) {  // (the requesting wave's first set: the partner's B pieces)
#pragma unroll
                for (int j = 0; j < T::B_PPW; ++j) {
                    const int r = (w1 * T::B_PPW + j) * RPP + lane / CH;
                    const int c = swz_rowk<BK>(lane % CH, r);
                    const int64_t hid = (n0 >> 1) + (r >> 6) * 32 + (r & 31);
                    const int64_t grow = ((r >> 5) & 1) * nh + hid;
                    voffA[j] = hid < nh ? (unsigned)(grow * p.ldb * 2 + c * 16) : OOB;
                    kneedA[j] = c * 8;
                }
            }
            baseB = p.B;
        }
    }
    if constexpr (SPLIT) {  // both sets of a wave come from ONE operand
        if (a_side) baseB = baseA;
        else baseA = baseB;
    }
    baseA = uniform_ptr(baseA);
    baseB = uniform_ptr(baseB);
    const int64_t stepA = A_TR ? (int64_t)BK * p.lda : BK;
    const int64_t stepB = B_TR ? (int64_t)BK * p.ldb : BK;

    auto issue_tile = [&](int t, int stage) {
        const int64_t krem = p.K - (int64_t)t * BK;
        const bf16_t* pa = baseA + t * stepA;
        const bf16_t* pb = baseB + t * stepB;
        char* dA = smem + stage * T::STAGE + ((SPLIT && !a_side) ? T::A_BYTES : 0) + w1 * T::A_PPW * 1024;
        char* dB = smem + stage * T::STAGE + ((SPLIT && a_side) ? 0 : T::A_BYTES) + w2 * T::B_PPW * 1024;
#pragma unroll
        for (int j = 0; j < T::A_PPW; ++j) dma_piece(pa, kneedA[j] < krem ? voffA[j] : OOB, dA + j * 1024);
#pragma unroll
        for (int j = 0; j < T::B_PPW; ++j) dma_piece(pb, kneedB[j] < krem ? voffB[j] : OOB, dB + j * 1024);
    };

    f32x4 acc[T::FM][T::FN];
#pragma unroll
    for (int i = 0; i < T::FM; ++i)
#pragma unroll
        for (int j = 0; j < T::FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int wr0 = (wave / T::WN) * T::WTM, wc0 = (wave % T::WN) * T::WTN;
    const int nt_all = (int)((p.K + BK - 1) / BK);
    const int per = (nt_all + p.ksplit - 1) / p.ksplit;
    const int t0 = split * per;
    const int nt = min(nt_all, t0 + per);

    bool extra_barrier = false;
    if constexpr (std::is_same_v<T, Cfg256b>) {
        // ---- alternating groups, ONE barrier per phase.  The two groups stay one barrier apart (waves 4-7 take an extra one
        // up front), so at every barrier one group has just finished a load segment and the other a 16-MFMA cluster; a phase
        // is { load segment ; lgkmcnt(0) ; s_barrier ; 16 MFMAs }.  With half the barriers the hazards are covered by distance:
        //   WAR  the DMA of tile t+3 (issued during tile t) lands in the stage of tile t-2 (5 stages), which both groups left
        //        at least one barrier before the issuing wave's current one;
        //   RAW  a wave confirms its pieces of tile t+1 (counted vmcnt; they were requested 3-4 phases earlier) in phase 0
        //        of tile t; the group that runs ahead reads tile t+1 two barriers later, after the group behind has passed
        //        its own phase-0 wait.
        static_assert(T::NW == 8 && T::KK == 1 && T::NS == 5 && T::A_PPW == 2 && T::B_PPW == 2, "single-barrier loop: 8 waves, 5 stages, 2+2 pieces");
        constexpr int HM = T::FM / 2;
        const bool late = wave >= 4;
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (t0 + i < nt) issue_tile(t0 + i, i);
        if (t0 < nt) {
            const int younger = nt - 1 - t0;
            if (younger >= 2) wait_vmcnt<8>(); else if (younger >= 1) wait_vmcnt<4>(); else wait_vmcnt<0>();
        }
        __builtin_amdgcn_s_barrier();  // tile t0 landed for every wave
        if (late) __builtin_amdgcn_s_barrier();
        bf16x8 a[HM], b[T::FN];
        constexpr bool TR_ASM = A_TR && B_TR;  // weight gradients: transposing reads from asm (see frag_tr_issue)
        [[maybe_unused]] TrHalves ah[HM], bh[T::FN];
        for (int t = t0; t < nt; ++t) {
            const char* sA = smem + ((t - t0) % T::NS) * T::STAGE;
            const char* sB = sA + T::A_BYTES;
            const int nxt = t + 3;
            const int nst = (nxt - t0) % T::NS;
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
                auto read_frags = [&]() {
                    if constexpr (TR_ASM) {
                        if (ph == 0) {
#pragma unroll
                            for (int j = 0; j < T::FN; ++j) frag_tr_issue<T::BN>(bh[j], sB, wc0 + j * 16, 0, lane);
                        }
#pragma unroll
                        for (int i = 0; i < HM; ++i) frag_tr_issue<T::BM>(ah[i], sA, wr0 + (ph * HM + i) * 16, 0, lane);
                        return;
                    }
                    if (ph == 0) {
#pragma unroll
                        for (int j = 0; j < T::FN; ++j) b[j] = B_TR ? frag_tr<T::BN>(sB, wc0 + j * 16, 0, lane) : frag_rowk<BK>(sB, wc0 + j * 16, 0, lane);
                    }
#pragma unroll
                    for (int i = 0; i < HM; ++i) {
                        const int r0 = wr0 + (ph * HM + i) * 16;
                        a[i] = A_TR ? frag_tr<T::BM>(sA, r0, 0, lane) : frag_rowk<BK>(sA, r0, 0, lane);
                    }
                };
                if (!(p.ablate & 2)) read_frags();
                if (nxt < nt) {
                    const int64_t krem = p.K - (int64_t)nxt * BK;
                    if (ph == 0) {
                        const bf16_t* pa = baseA + nxt * stepA;
                        char* dA = smem + nst * T::STAGE + wave * T::A_PPW * 1024;
#pragma unroll
                        for (int j = 0; j < T::A_PPW; ++j) dma_piece(pa, kneedA[j] < krem ? voffA[j] : OOB, dA + j * 1024);
                    } else {
                        const bf16_t* pb = baseB + nxt * stepB;
                        char* dB = smem + nst * T::STAGE + T::A_BYTES + wave * T::B_PPW * 1024;
#pragma unroll
                        for (int j = 0; j < T::B_PPW; ++j) dma_piece(pb, kneedB[j] < krem ? voffB[j] : OOB, dB + j * 1024);
                    }
                }
                if (p.ablate & 2) read_frags();  // experiment: DMA issue in front of the fragment reads
                if (ph == 0 && t + 1 < nt) {  // this wave's share of tile t+1 has landed; tile t+2 and the A pieces of t+3 may be in flight
                    if (nxt < nt) wait_vmcnt<6>(); else if (t + 2 < nt) wait_vmcnt<4>(); else wait_vmcnt<0>();
                }
                if constexpr (TR_ASM) {
                    static_assert(HM == 4 && T::FN == 4, "operand lists below are written for 4 + 4 fragments");
                    if (ph == 0) {
                        asm volatile("s_waitcnt lgkmcnt(0)" : TRH(bh[0]), TRH(bh[1]), TRH(bh[2]), TRH(bh[3]), TRH(ah[0]), TRH(ah[1]), TRH(ah[2]), TRH(ah[3])::"memory");
#pragma unroll
                        for (int j = 0; j < T::FN; ++j) b[j] = tr_join(bh[j]);
                    } else {
                        asm volatile("s_waitcnt lgkmcnt(0)" : TRH(ah[0]), TRH(ah[1]), TRH(ah[2]), TRH(ah[3])::"memory");
                    }
#pragma unroll
                    for (int i = 0; i < HM; ++i) a[i] = tr_join(ah[i]);
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                if (p.ablate & 1) __builtin_amdgcn_s_setprio(1);  // measured: raising the cluster's priority costs 2-3 % in this loop
#pragma unroll
                for (int i = 0; i < HM; ++i)
#pragma unroll
                    for (int j = 0; j < T::FN; ++j)
                        acc[ph * HM + i][j] = MFMA_CT(a[i], b[j], acc[ph * HM + i][j]);
                if (p.ablate & 1) __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        extra_barrier = !late;  // the early group balances the late group's extra barrier
    } else if constexpr (T::NW == 4 && T::BK == 32) {
        // ---- one wave per SIMD, 128x128 outputs per wave (256 accumulator registers): a third fewer fragment bytes per MFMA than the
        // 128x64 wave tiles, and nobody to alternate with, so the loads ride in the shadow of the wave's own MFMAs: per K-tile (64 MFMAs)
        //   phase 0: 32 MFMAs (rows 0-63)   beside the 4 fragment reads of rows 64-127 and the wave's 8 DMA pieces of tile t+3
        //   counted vmcnt (tile t+1 landed) ; s_barrier
        //   phase 1: 32 MFMAs (rows 64-127) beside the 12 fragment reads of tile t+1 (its B fragments go to the other register set)
        // The DMA is issued for every tile number, past the end of K with out-of-range offsets (zero fill into a stage nobody reads),
        // so the body has no branches and the vmcnt immediate is one constant.
        static_assert(T::KK == 1 && T::NS == 4 && T::FM == 8 && T::FN == 8 && T::A_PPW == 4 && T::B_PPW == 4, "4-wave loop: 128x128 per wave, 4+4 pieces");
        constexpr int HM = 4;
        auto issue_piece = [&](int tl, int stage, int g) {  // piece g (0-3: A, 4-7: B) of this wave's share of tile tl, any tile number
            const int64_t krem = p.K - (int64_t)tl * BK;
            if (g < T::A_PPW) {
                dma_piece(baseA + tl * stepA, (tl < nt && kneedA[g] < krem) ? voffA[g] : OOB, smem + stage * T::STAGE + (wave * T::A_PPW + g) * 1024);
            } else {
                const int h = g - T::A_PPW;
                dma_piece(baseB + tl * stepB, (tl < nt && kneedB[h] < krem) ? voffB[h] : OOB, smem + stage * T::STAGE + T::A_BYTES + (wave * T::B_PPW + h) * 1024);
            }
        };
        // The accumulators are pinned to the accumulation registers by the operand constraint: with the builtin, hipcc keeps part of the
        // 256 values in vector registers and moves four in and four out around every MFMA (measured: 594 TFLOP/s on the gate-up shape).
        auto mma = [&](f32x4& c, const bf16x8& av, const bf16x8& bv) {
#if GEMM_SWAP
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %1, %0" : "+a"(c) : "v"(av), "v"(bv));  // B fragment first: see MFMA_CT
#else
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(av), "v"(bv));
#endif
        };
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int g = 0; g < 8; ++g) issue_piece(t0 + i, i, g);
        wait_vmcnt<16>();
        __builtin_amdgcn_s_barrier();  // tile t0 landed for every wave
        bf16x8 a0[HM], a1[HM], b[2][T::FN];
#pragma unroll
        for (int j = 0; j < T::FN; ++j) b[0][j] = B_TR ? frag_tr<T::BN>(smem + T::A_BYTES, wc0 + j * 16, 0, lane) : frag_rowk<BK>(smem + T::A_BYTES, wc0 + j * 16, 0, lane);
#pragma unroll
        for (int i = 0; i < HM; ++i) a0[i] = A_TR ? frag_tr<T::BM>(smem, wr0 + i * 16, 0, lane) : frag_rowk<BK>(smem, wr0 + i * 16, 0, lane);
        auto body = [&](auto par, int t) {
            constexpr int CUR = decltype(par)::value;
            const char* sA = smem + ((t - t0) & 3) * T::STAGE;
            const char* nA = smem + ((t + 1 - t0) & 3) * T::STAGE;
            const int nst = (t + 3 - t0) & 3;
            // -------- phase 0: 8 groups of { 4 MFMAs, 1 DMA piece, (first four) 1 fragment of the lower rows }
#pragma unroll
            for (int g = 0; g < 8; ++g) {
#pragma unroll
                for (int q = 0; q < 4; ++q) mma(acc[g >> 1][(g & 1) * 4 + q], a0[g >> 1], b[CUR][(g & 1) * 4 + q]);
                issue_piece(t + 3, nst, g);
                if (g < HM) a1[g] = A_TR ? frag_tr<T::BM>(sA, wr0 + (HM + g) * 16, 0, lane) : frag_rowk<BK>(sA, wr0 + (HM + g) * 16, 0, lane);
                __builtin_amdgcn_sched_barrier(0);
            }
            wait_vmcnt<16>();  // this wave's pieces of tile t+1 have landed (t+2, t+3 in flight)
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // -------- phase 1: 12 groups of { 2 MFMAs, 1 fragment of tile t+1 }, then the last 8 MFMAs
#pragma unroll
            for (int g = 0; g < 16; ++g) {
#pragma unroll
                for (int q = 0; q < 2; ++q) mma(acc[HM + (g >> 2)][(g & 3) * 2 + q], a1[g >> 2], b[CUR][(g & 3) * 2 + q]);
                if (g < T::FN) b[CUR ^ 1][g] = B_TR ? frag_tr<T::BN>(nA + T::A_BYTES, wc0 + g * 16, 0, lane) : frag_rowk<BK>(nA + T::A_BYTES, wc0 + g * 16, 0, lane);
                else if (g < T::FN + HM) a0[g - T::FN] = A_TR ? frag_tr<T::BM>(nA, wr0 + (g - T::FN) * 16, 0, lane) : frag_rowk<BK>(nA, wr0 + (g - T::FN) * 16, 0, lane);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // Weight gradients (both operands K-strided): the same schedule with every fragment requested from asm (two ds_read_b64_tr_b16 per fragment;
        // the intrinsic form makes hipcc drain the DMA ring with vmcnt(0) in front of each, see frag_tr_issue) and joined behind a wait that names
        // the halves.  Phase 0's four reads are joined at the barrier, phase 1's twelve at the end of the tile: both sit under 32 MFMAs.
        [[maybe_unused]] TrHalves h_a1[HM], h_b[T::FN], h_a0[HM];
        auto body_tr = [&](auto par, int t) {
            constexpr int CUR = decltype(par)::value;
            const char* sA = smem + ((t - t0) & 3) * T::STAGE;
            const char* nA = smem + ((t + 1 - t0) & 3) * T::STAGE;
            const int nst = (t + 3 - t0) & 3;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
#pragma unroll
                for (int q = 0; q < 4; ++q) mma(acc[g >> 1][(g & 1) * 4 + q], a0[g >> 1], b[CUR][(g & 1) * 4 + q]);
                if ((g & 1) == 0) issue_piece(t + 3, nst, g >> 1);  // the A pieces here, the B pieces in phase 1: one request per 8 MFMAs over the whole K-tile (all eight in phase 0
                                                                     // were one per 4 MFMAs from four waves at once, then none for 32 MFMAs; the ring is three tiles deep, nothing waits for them)
                if (g < HM) frag_tr_issue<T::BM>(h_a1[g], sA, wr0 + (HM + g) * 16, 0, lane);
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : TRH(h_a1[0]), TRH(h_a1[1]), TRH(h_a1[2]), TRH(h_a1[3])::"memory");
#pragma unroll
            for (int i = 0; i < HM; ++i) a1[i] = tr_join(h_a1[i]);
            wait_vmcnt<12>();  // tile t + 1 has landed: tile t + 2 and the four A pieces of t + 3 may be in flight
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < 16; ++g) {
#pragma unroll
                for (int q = 0; q < 2; ++q) mma(acc[HM + (g >> 2)][(g & 3) * 2 + q], a1[g >> 2], b[CUR][(g & 3) * 2 + q]);
                if ((g & 3) == 1) issue_piece(t + 3, nst, T::A_PPW + (g >> 2));
                if (g < T::FN) frag_tr_issue<T::BN>(h_b[g], nA + T::A_BYTES, wc0 + g * 16, 0, lane);
                else if (g < T::FN + HM) frag_tr_issue<T::BM>(h_a0[g - T::FN], nA, wr0 + (g - T::FN) * 16, 0, lane);
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : TRH(h_b[0]), TRH(h_b[1]), TRH(h_b[2]), TRH(h_b[3]), TRH(h_b[4]), TRH(h_b[5]), TRH(h_b[6]), TRH(h_b[7]),
                         TRH(h_a0[0]), TRH(h_a0[1]), TRH(h_a0[2]), TRH(h_a0[3])::"memory");
#pragma unroll
            for (int j = 0; j < T::FN; ++j) b[CUR ^ 1][j] = tr_join(h_b[j]);
#pragma unroll
            for (int i = 0; i < HM; ++i) a0[i] = tr_join(h_a0[i]);
        };
        int t = t0;
        if constexpr (A_TR && B_TR) {
            for (; t + 1 < nt; t += 2) {
                body_tr(std::integral_constant<int, 0>{}, t);
                body_tr(std::integral_constant<int, 1>{}, t + 1);
            }
            if (t < nt) body_tr(std::integral_constant<int, 0>{}, t);
        } else {
            for (; t + 1 < nt; t += 2) {
                body(std::integral_constant<int, 0>{}, t);
                body(std::integral_constant<int, 1>{}, t + 1);
            }
            if (t < nt) body(std::integral_constant<int, 0>{}, t);
        }
        wait_vmcnt<0>();  // the zero-fill DMAs past the last tile must not land in the epilogue's staging
    } else if constexpr (T::BK == 32) {
        // ---- alternating-group loop (8 waves, BK = 32, NS-stage ring).  Waves 4-7 run ONE barrier behind waves 0-3, so on
        // every SIMD one wave is in its 16-MFMA cluster while its partner is in the load segment (fragment reads, two
        // LDS-DMA issues, waits).  A phase = { load segment ; lgkmcnt(0) ; s_barrier ; 16 MFMAs ; s_barrier }, two phases per
        // K-tile (row halves mh = 0, 1).  Hazards: a wave's reads of tile t are complete before the barrier that precedes
        // its last MFMA cluster of that tile, and the slot is refilled (tile t+NS) only after that wave group's following
        // barrier, which the other group reaches after completing ITS reads; tile t+1 is waited for (counted vmcnt) in
        // phase 1 of tile t, a full cluster + barrier before anyone reads it.
        static_assert(T::NW == 8 && T::KK == 1 && T::NS >= 3 && T::A_PPW == 2 && T::B_PPW == 2, "alternating loop: 8 waves, 2+2 DMA pieces per tile");
        constexpr int HM = T::FM / 2;
        const bool late = wave >= 4;  // wave-uniform (readfirstlane above)
#pragma unroll
        for (int i = 0; i < T::NS - 1; ++i)
            if (t0 + i < nt) issue_tile(t0 + i, i);
        if (t0 < nt) {
            const int younger = nt - 1 - t0;
            if (younger >= 2) wait_vmcnt<8>(); else if (younger >= 1) wait_vmcnt<4>(); else wait_vmcnt<0>();
        }
        __builtin_amdgcn_s_barrier();  // tile t0 landed for every wave
        if (late) __builtin_amdgcn_s_barrier();
        bf16x8 a[HM], b[T::FN];
        constexpr bool TR_ASM = A_TR && B_TR;  // weight gradients: both operands K-strided
        [[maybe_unused]] TrHalves ah[HM], bh[T::FN];
        [[maybe_unused]] unsigned long long gp[8] = {};
        [[maybe_unused]] const unsigned long long gp_start = GP_T();
        for (int t = t0; t < nt; ++t) {
            const char* sA = smem + ((t - t0) % T::NS) * T::STAGE;
            const char* sB = sA + T::A_BYTES;
            const int nxt = t + T::NS - 1;
            const int nst = (nxt - t0) % T::NS;
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
                [[maybe_unused]] unsigned long long gt = GP_T();
                // -------- load segment
                if constexpr (TR_ASM) {
                    if (ph == 0) {
#pragma unroll
                        for (int j = 0; j < T::FN; ++j) frag_tr_issue<T::BN>(bh[j], sB, wc0 + j * 16, 0, lane);
                    }
#pragma unroll
                    for (int i = 0; i < HM; ++i) frag_tr_issue<T::BM>(ah[i], sA, wr0 + (ph * HM + i) * 16, 0, lane);
                } else if (!(p.ablate & 2) || t == t0) {
                    if (ph == 0) {
#pragma unroll
                        for (int j = 0; j < T::FN; ++j) b[j] = B_TR ? frag_tr<T::BN>(sB, wc0 + j * 16, 0, lane) : frag_rowk<BK>(sB, wc0 + j * 16, 0, lane);
                    }
#pragma unroll
                    for (int i = 0; i < HM; ++i) {
                        const int r0 = wr0 + (ph * HM + i) * 16;
                        a[i] = A_TR ? frag_tr<T::BM>(sA, r0, 0, lane) : frag_rowk<BK>(sA, r0, 0, lane);
                    }
                }
                GP_ADD(0, gt);  // fragment reads issued
                if (nxt < nt && !(p.ablate & 1)) {  // two of this wave's four DMA pieces of tile t+NS-1 per phase: A pieces, then B pieces
                    const int64_t krem = p.K - (int64_t)nxt * BK;
                    if (ph == 0) {
                        const bf16_t* pa = baseA + nxt * stepA;
                        char* dA = smem + nst * T::STAGE + wave * T::A_PPW * 1024;
#pragma unroll
                        for (int j = 0; j < T::A_PPW; ++j) dma_piece(pa, kneedA[j] < krem ? voffA[j] : OOB, dA + j * 1024);
                    } else {
                        const bf16_t* pb = baseB + nxt * stepB;
                        char* dB = smem + nst * T::STAGE + T::A_BYTES + wave * T::B_PPW * 1024;
#pragma unroll
                        for (int j = 0; j < T::B_PPW; ++j) dma_piece(pb, kneedB[j] < krem ? voffB[j] : OOB, dB + j * 1024);
                    }
                }
                GP_ADD(1, gt);  // DMA issued
                if (ph == 1 && t + 1 < nt) {  // this wave's share of tile t+1 has landed (tiles t+2.. may stay in flight)
                    const int younger = min(nt - 1, t + T::NS - 1) - (t + 1);
                    if (younger >= 2) wait_vmcnt<8>(); else if (younger >= 1) wait_vmcnt<4>(); else wait_vmcnt<0>();
                }
                GP_ADD(2, gt);  // vmcnt wait
                const bool wait_late = !TR_ASM && ((p.ablate & 8) || ((p.ablate & 16) && ph == 0));
                if constexpr (TR_ASM) {  // the wait names the halves it covers, so no use of them can be placed in front of it
                    static_assert(HM == 4 && T::FN == 4, "operand lists below are written for 4 + 4 fragments");
                    if (ph == 0) {
                        asm volatile("s_waitcnt lgkmcnt(0)" : TRH(bh[0]), TRH(bh[1]), TRH(bh[2]), TRH(bh[3]), TRH(ah[0]), TRH(ah[1]), TRH(ah[2]), TRH(ah[3])::"memory");
#pragma unroll
                        for (int j = 0; j < T::FN; ++j) b[j] = tr_join(bh[j]);
                    } else {
                        asm volatile("s_waitcnt lgkmcnt(0)" : TRH(ah[0]), TRH(ah[1]), TRH(ah[2]), TRH(ah[3])::"memory");
                    }
#pragma unroll
                    for (int i = 0; i < HM; ++i) a[i] = tr_join(ah[i]);
                } else if (!wait_late) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                GP_ADD(3, gt);  // lgkmcnt wait
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                GP_ADD(4, gt);  // barrier in front of the cluster
                if (wait_late) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                // -------- MFMA cluster
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < HM; ++i)
#pragma unroll
                    for (int j = 0; j < T::FN; ++j)
                        acc[ph * HM + i][j] = MFMA_CT(a[i], b[j], acc[ph * HM + i][j]);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                GP_ADD(5, gt);  // MFMA cluster issued
                if (!(p.ablate & 4)) __builtin_amdgcn_s_barrier();
                GP_ADD(6, gt);  // barrier behind the cluster
            }
        }
#if GEMM_PROF
        if ((threadIdx.x & 63) == 0 && (wave == 0 || wave == 4)) {
            const int o = wave == 0 ? 0 : 16;
            for (int i = 0; i < 7; ++i) atomicAdd(&g_gemm_prof[o + i], gp[i]);
            atomicAdd(&g_gemm_prof[o + 7], __builtin_readcyclecounter() - gp_start);
            atomicAdd(&g_gemm_prof[o + 8], (unsigned long long)(2 * (nt - t0)));
        }
#endif
        extra_barrier = !late && !(p.ablate & 4);  // the early group balances the late group's extra barrier
    } else {
    // ---- main loop: 4 phases per K-tile = (k-step kk, half of the wave's rows mh); the fragments of phase p+1 are read
    // from LDS while the MFMAs of phase p run (two register sets for the A half, two for B), so the matrix pipe is not
    // idle during fragment reads.  The per-tile barrier sits in front of the LAST phase's MFMA cluster: by then every
    // read of tile t has been issued (and is waited for), tile t+1's DMA is checked with a counted vmcnt, the freed stage
    // is refilled and the first fragments of tile t+1 are requested -- all under the remaining MFMAs of tile t.
    constexpr int HM = T::FM / 2;
    static_assert(T::KK == 2, "the pipelined loop is written for BK = 64");
    auto loadA = [&](bf16x8 (&a)[HM], const char* sA, int kk, int mh) {
        if (p.ablate & 2) return;
#pragma unroll
        for (int i = 0; i < HM; ++i) {
            const int r0 = wr0 + (mh * HM + i) * 16;
            a[i] = A_TR ? frag_tr<T::BM>(sA, r0, kk, lane) : frag_rowk<BK>(sA, r0, kk, lane);
        }
    };
    auto loadB = [&](bf16x8 (&b)[T::FN], const char* sB, int kk) {
        if (p.ablate & 2) return;
#pragma unroll
        for (int j = 0; j < T::FN; ++j) b[j] = B_TR ? frag_tr<T::BN>(sB, wc0 + j * 16, kk, lane) : frag_rowk<BK>(sB, wc0 + j * 16, kk, lane);
    };
    auto mma = [&](const bf16x8 (&a)[HM], const bf16x8 (&b)[T::FN], int mh) {
        // raised priority around the cluster: kept on the 4-wave tile; on the 8-wave tile it costs 1.5 ms/step (206.0 / 206.4 without vs 207.1 / 208.3)
        if constexpr (T::NW != 8) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < HM; ++i)  // (B fragment held over consecutive MFMAs instead of the A fragment: level in the step, 192.4 / 192.7 / 192.4 against 192.8 / 192.0 / 192.3)
#pragma unroll
            for (int j = 0; j < T::FN; ++j)
                acc[mh * HM + i][j] = MFMA_CT(a[i], b[j], acc[mh * HM + i][j]);
        if constexpr (T::NW != 8) __builtin_amdgcn_s_setprio(0);
    };
    auto wait_tile = [&](int younger) {  // returns once at most min(NS-2, younger) younger tiles of this wave are in flight
        if (T::NS >= 4 && younger >= 2) wait_vmcnt<2 * T::DMA_PER_TILE>();
        else if (T::NS >= 3 && younger >= 1) wait_vmcnt<T::DMA_PER_TILE>();
        else wait_vmcnt<0>();
    };

#pragma unroll
    for (int i = 0; i < T::NS - 1; ++i)
        if (t0 + i < nt) issue_tile(t0 + i, i);
    bf16x8 aE[HM], aO[HM], b0[T::FN], b1[T::FN];
    if constexpr (A_TR && B_TR) {
        // weight gradients: the same four-phase schedule with the transposing reads issued from asm (see frag_tr_issue) and a counted,
        // operand-naming lgkmcnt in front of every MFMA cluster: 2 LDS instructions per fragment, so with the reads of the following
        // phase(s) already in flight a cluster waits for all but the 2*HM (phases 0, 2) or 2*FN + 2*HM (phases 1, 3) youngest.
        constexpr int YA = 2 * HM < 15 ? 2 * HM : 15, YAB = 2 * T::FN + 2 * HM < 15 ? 2 * T::FN + 2 * HM : 15;
        TrHalves hE[HM], hO[HM], h0[T::FN], h1[T::FN];
        auto ldA = [&](TrHalves (&h)[HM], const char* sA, int kk, int mh) {
#pragma unroll
            for (int i = 0; i < HM; ++i) frag_tr_issue<T::BM>(h[i], sA, wr0 + (mh * HM + i) * 16, kk, lane);
        };
        auto ldB = [&](TrHalves (&h)[T::FN], const char* sB, int kk) {
#pragma unroll
            for (int j = 0; j < T::FN; ++j) frag_tr_issue<T::BN>(h[j], sB, wc0 + j * 16, kk, lane);
        };
        if (t0 < nt) {
            wait_tile(nt - 1 - t0);
            __builtin_amdgcn_s_barrier();
            if (t0 + T::NS - 1 < nt) issue_tile(t0 + T::NS - 1, T::NS - 1);
            ldB(h0, smem + T::A_BYTES, 0);
            ldA(hE, smem, 0, 0);
        }
        for (int t = t0; t < nt; ++t) {
            const char* sA = smem + ((t - t0) % T::NS) * T::STAGE;
            const char* sB = sA + T::A_BYTES;
            ldA(hO, sA, 0, 1);
            tr_wait_join<YA>(h0, b0);
            tr_wait_join<YA>(hE, aE);
            mma(aE, b0, 0);  // phase 0
            ldB(h1, sB, 1);
            ldA(hE, sA, 1, 0);
            tr_wait_join<YAB>(hO, aO);
            mma(aO, b0, 1);  // phase 1
            ldA(hO, sA, 1, 1);
            tr_wait_join<YA>(h1, b1);
            tr_wait_join<YA>(hE, aE);
            mma(aE, b1, 0);  // phase 2
            if (t + 1 < nt) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's reads of tile t are complete ...
                wait_tile(nt - 2 - t);                              // ... and its share of tile t+1 has landed
                __builtin_amdgcn_s_barrier();                       // ... for every wave
                if (t + T::NS < nt) issue_tile(t + T::NS, (t - t0) % T::NS);
                const char* nA = smem + ((t + 1 - t0) % T::NS) * T::STAGE;
                ldB(h0, nA + T::A_BYTES, 0);
                ldA(hE, nA, 0, 0);
                tr_wait_join<YAB>(hO, aO);  // landed before the barrier; the wait only names them
            } else {
                tr_wait_join<0>(hO, aO);
            }
            mma(aO, b1, 1);  // phase 3
        }
    } else {
#pragma unroll
    for (int i = 0; i < HM; ++i) aE[i] = aO[i] = (bf16x8){1, 2, 3, 4, 5, 6, 7, 8};
#pragma unroll
    for (int j = 0; j < T::FN; ++j) b0[j] = b1[j] = (bf16x8){8, 7, 6, 5, 4, 3, 2, 1};
    if (t0 < nt) {
        wait_tile(nt - 1 - t0);
        __builtin_amdgcn_s_barrier();
        TL_STAMP(1);
        if (t0 + T::NS - 1 < nt) issue_tile(t0 + T::NS - 1, T::NS - 1);
        loadB(b0, smem + T::A_BYTES, 0);
        loadA(aE, smem, 0, 0);
    }
    for (int t = t0; t < nt; ++t) {
        const char* sA = smem + ((t - t0) % T::NS) * T::STAGE;
        const char* sB = sA + T::A_BYTES;
        loadA(aO, sA, 0, 1);
        mma(aE, b0, 0);  // phase 0
        if constexpr (SPLIT) {  // the B panels of tile t + 1 (its A panels went out behind the last barrier; tile t0 + 1: in front of the loop)
            static_assert(!SPLIT || T::NS == 2, "written for the two-stage ring");
            if (!a_side && t > t0 && t + 1 < nt && !(p.ablate & 1)) issue_tile(t + 1, (t + 1 - t0) % T::NS);
        }
        loadB(b1, sB, 1);
        loadA(aE, sA, 1, 0);
        mma(aO, b0, 1);  // phase 1
        loadA(aO, sA, 1, 1);
        mma(aE, b1, 0);  // phase 2
        if (t + 1 < nt) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's reads of tile t are complete ...
            wait_tile(nt - 2 - t);                              // ... and its share of tile t+1 has landed
            if (!(p.ablate & 4)) __builtin_amdgcn_s_barrier();  // ... for every wave
            if (t + T::NS < nt && !(p.ablate & 1) && a_side) issue_tile(t + T::NS, (t - t0) % T::NS);
            const char* nA = smem + ((t + 1 - t0) % T::NS) * T::STAGE;
            loadB(b0, nA + T::A_BYTES, 0);
            loadA(aE, nA, 0, 0);
        }
        mma(aO, b1, 1);  // phase 3
    }
    }  // K-contiguous operand(s): compiler-issued reads

    }
    if (extra_barrier) __builtin_amdgcn_s_barrier();
    TL_STAMP(2);
#ifdef GEMM_EPI_ABL  // profiling builds only (tools/build_variant.sh epi1 gemm_p2 "-DGEMM_EPI_ABL=1", tools/time_nt_shapes.py): no write-out at all -- what is left is
    // prologue + main loop.  Round 4, batch 160: QKV 907 -> 767 us, gate-up + SwiGLU 1 490 -> 1 100, down dgrad + SwiGLU backward 975 -> 575, dctx 455 -> 368.
    if (GEMM_EPI_ABL & 1) {
        float keep = 0.f;
#pragma unroll
        for (int i = 0; i < T::FM; ++i)
#pragma unroll
            for (int j = 0; j < T::FN; ++j) keep += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (keep == 12345.678f) reinterpret_cast<float*>(p.C)[0] = keep;  // keeps the main loop alive
        return;
    }
#endif
    // ---- epilogue: acc -> LDS (fp32, 64x64 per wave at a time) -> coalesced rows ---------------------------------
    __syncthreads();
    TL_STAMP(3);
    float* stg = reinterpret_cast<float*>(smem) + wave * 64 * EPI_LD;
    // "No fused form" and "this 64x64 sub-block lies inside the matrix and is 16-byte addressable" are resolved once, outside the store
    // passes: the plain epilogue (with or without bias / residual) gets a straight-line copy of them for interior sub-blocks.  With the kind
    // tested inside every pass (all fused forms inlined behind run-time branches) the passes were instruction-issue-bound: 22 us per
    // 256x256 tile with four waves, ~5 us with eight.  The fused forms keep the run-time dispatch (compile time: every extra copy of the
    // passes costs about a minute over the 48 kernels of this file).
    const bool aligned_io = ((p.ldc & 7) == 0) && ((p.ldr & 7) == 0 || p.R == nullptr);
    auto run_epilogue = [&](auto kind_c) __attribute__((always_inline)) {
        constexpr int KSEL = decltype(kind_c)::value;
        const int KIND = KSEL >= 0 ? KSEL : p.epilogue;
        // residual rows of interior sub-blocks are fetched one sub-block ahead (eight 16-byte loads per lane), so that their HBM latency is
        // paid once per tile instead of in every store pass (measured: +7.6 us per 256x256 tile on eight waves, +17 us on four).  NT only:
        // that is where the step's residual adds are (out-projection, down-projection); in the NN / TN kernels the 38 extra registers of the
        // look-ahead changed the main loop's allocation and cost 1.2 ms/step each in the per-kernel profile, for a path they never take.
        constexpr bool RES_AHEAD = (KSEL == MI355_EPI_NONE || KSEL == MI355_EPI_ATTN_DELTA) && OUT_DT == MI355_DT_BF16 && !A_TR && !B_TR;
        constexpr int SUBS_N = T::WTN / 64, SUBS = (T::WTM / 64) * SUBS_N;
        const bool res_ahead = RES_AHEAD && p.R != nullptr && aligned_io && p.ksplit == 1;
        [[maybe_unused]] u32x4 rnext[8];
        auto sub_inside = [&](int sm, int sn) { return aligned_io && m0 + wr0 + sm * 64 + 64 <= p.M && n0 + wc0 + sn * 64 + 64 <= p.N; };
        auto fetch_residual = [&](int sm, int sn) __attribute__((always_inline)) {
            const bf16_t* r0 = reinterpret_cast<const bf16_t*>(p.R) + (m0 + wr0 + sm * 64 + (lane >> 3)) * p.ldr + n0 + wc0 + sn * 64 + (lane & 7) * 8;
#pragma unroll
            for (int tpass = 0; tpass < 8; ++tpass) rnext[tpass] = *reinterpret_cast<const u32x4*>(r0 + (int64_t)tpass * 8 * p.ldr);
        };
        if constexpr (RES_AHEAD) {
            if (res_ahead && sub_inside(0, 0)) fetch_residual(0, 0);
        }
#pragma unroll
        for (int sm = 0; sm < T::WTM / 64; ++sm) {
#pragma unroll
            for (int sn = 0; sn < T::WTN / 64; ++sn) {
                // SwiGLU backward: the sub-block's 16 loads of the forward gate-up pair go out before its staging writes, all in flight at once,
                // instead of two dependent loads inside each of the eight store passes
                constexpr bool GU_EARLY = KSEL == MI355_EPI_SWIGLU_BWD && OUT_DT == MI355_DT_BF16;
                [[maybe_unused]] u32x4 ucur[8], gcur[8];
                if constexpr (GU_EARLY) {
                    if (sub_inside(sm, sn)) {
                        const bf16_t* r0 = reinterpret_cast<const bf16_t*>(p.R) + (m0 + wr0 + sm * 64 + (lane >> 3)) * p.ldr + n0 + wc0 + sn * 64 + (lane & 7) * 8;
#pragma unroll
                        for (int tpass = 0; tpass < 8; ++tpass) {
                            ucur[tpass] = *reinterpret_cast<const u32x4*>(r0 + (int64_t)tpass * 8 * p.ldr);
                            gcur[tpass] = *reinterpret_cast<const u32x4*>(r0 + (int64_t)tpass * 8 * p.ldr + p.N);
                        }
                    }
                }
#if GEMM_SWAP
                // every MFMA takes the B fragment as its first operand (MFMA_CT), so a 16x16 accumulator tile is the TRANSPOSE of the usual layout: the lane holds
                // C[lane & 15][4 (lane >> 4) + e], four consecutive COLUMNS of one row -- one 16-byte LDS store per tile (16 per sub-block) instead of four 4-byte ones (64)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        *reinterpret_cast<f32x4*>(stg + (i * 16 + (lane & 15)) * EPI_LD + j * 16 + (lane >> 4) * 4) = acc[sm * 4 + i][sn * 4 + j];
#else
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            stg[(i * 16 + (lane >> 4) * 4 + e) * EPI_LD + j * 16 + (lane & 15)] = acc[sm * 4 + i][sn * 4 + j][e];
#endif
                __builtin_amdgcn_wave_barrier();
                const int64_t gn = n0 + wc0 + sn * 64 + (lane & 7) * 8;
                const int64_t gm0 = m0 + wr0 + sm * 64;
                [[maybe_unused]] u32x4 rcur[8];
                if constexpr (RES_AHEAD) {
#pragma unroll
                    for (int tpass = 0; tpass < 8; ++tpass) rcur[tpass] = rnext[tpass];
                    const int nxt = sm * SUBS_N + sn + 1;
                    if (nxt < SUBS && res_ahead && sub_inside(nxt / SUBS_N, nxt % SUBS_N)) fetch_residual(nxt / SUBS_N, nxt % SUBS_N);
                }
                auto passes = [&](auto full_c) __attribute__((always_inline)) {
                    constexpr bool FULL = decltype(full_c)::value;
                    if (p.ksplit > 1) {  // raw fp32 partial sums; residual / conversion happen in splitk_reduce_kernel
                        if constexpr (OUT_DT == MI355_DT_F32) {
                            float* slab = p.ws + (int64_t)split * p.M * p.N;
#pragma unroll
                            for (int tpass = 0; tpass < 8; ++tpass) {
                                const int row = tpass * 8 + (lane >> 3);
                                const int64_t gm = gm0 + row;
                                if (!FULL && (gm >= p.M || gn >= p.N)) continue;  // N % 8 == 0 is required for split-K
                                *reinterpret_cast<f32x4*>(slab + gm * p.N + gn) = *reinterpret_cast<const f32x4*>(stg + row * EPI_LD + (lane & 7) * 8);
                                *reinterpret_cast<f32x4*>(slab + gm * p.N + gn + 4) = *reinterpret_cast<const f32x4*>(stg + row * EPI_LD + (lane & 7) * 8 + 4);
                            }
                        }
                    } else {
                        const bool vec_ok = FULL || ((gn + 8 <= p.N) && aligned_io);
#pragma unroll
                        for (int tpass = 0; tpass < 8; ++tpass) {
                            const int row = tpass * 8 + (lane >> 3);
                            const int64_t gm = gm0 + row;
                            if (!FULL && (gm >= p.M || gn >= p.N)) continue;
                            float v[8];
                            const f32x4 v0 = *reinterpret_cast<const f32x4*>(stg + row * EPI_LD + (lane & 7) * 8);
                            const f32x4 v1 = *reinterpret_cast<const f32x4*>(stg + row * EPI_LD + (lane & 7) * 8 + 4);
                            v[0] = v0[0]; v[1] = v0[1]; v[2] = v0[2]; v[3] = v0[3];
                            v[4] = v1[0]; v[5] = v1[1]; v[6] = v1[2]; v[7] = v1[3];
                            const int nvalid = FULL ? 8 : (int)min((int64_t)8, p.N - gn);
                            if (p.bias) {
#pragma unroll
                                for (int e = 0; e < 8; ++e)
                                    if (e < nvalid) v[e] += p.bias[gn + e];
                            }
                            if (KIND == MI355_EPI_GELU_ERF) {
#pragma unroll
                                for (int e = 0; e < 8; ++e) v[e] = gelu_erf(v[e]);
                            }
                            if constexpr (OUT_DT == MI355_DT_BF16) {
                                bf16_t* c = reinterpret_cast<bf16_t*>(p.C) + gm * p.ldc + gn;
                                const bf16_t* r = p.R ? reinterpret_cast<const bf16_t*>(p.R) + gm * p.ldr + gn : nullptr;
                                if (KIND == MI355_EPI_GELU_DUAL_ERF || KIND == MI355_EPI_GELU_DUAL_TANH) {
                                    // v = acc + bias = the pre-activation: C gets it (the backward needs it), R (an OUTPUT here) gets gelu of its bf16 value
                                    u32x4 y1, act;
#pragma unroll
                                    for (int e = 0; e < 4; ++e) {
                                        y1[e] = pack_bf2(v[2 * e], v[2 * e + 1]);
                                        const float lo = __uint_as_float(y1[e] << 16), hi = __uint_as_float(y1[e] & 0xffff0000u);
                                        act[e] = KIND == MI355_EPI_GELU_DUAL_ERF ? pack_bf2(gelu_val<0>(lo), gelu_val<0>(hi)) : pack_bf2(gelu_val<1>(lo), gelu_val<1>(hi));
                                    }
                                    *reinterpret_cast<u32x4*>(c) = y1;
                                    *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(const_cast<void*>(p.R)) + gm * p.ldr + gn) = act;
                                    continue;
                                }
                                if (KIND == MI355_EPI_GELU_BWD_ERF || KIND == MI355_EPI_GELU_BWD_TANH) {
                                    // acc = d(act); R = the forward's pre-activation: C = bf16(acc) * gelu'(R)  (== dgrad GEMM -> mi355_gelu_bwd)
                                    const u32x4 xv = *reinterpret_cast<const u32x4*>(r);
                                    u32x4 o;
#pragma unroll
                                    for (int e = 0; e < 4; ++e) {
                                        const float x0 = __uint_as_float(xv[e] << 16), x1 = __uint_as_float(xv[e] & 0xffff0000u);
                                        const float d0 = bf2f(f2bf(v[2 * e])), d1 = bf2f(f2bf(v[2 * e + 1]));
                                        o[e] = KIND == MI355_EPI_GELU_BWD_ERF ? pack_bf2(d0 * gelu_grad<0>(x0), d1 * gelu_grad<0>(x1))
                                                                                     : pack_bf2(d0 * gelu_grad<1>(x0), d1 * gelu_grad<1>(x1));
                                    }
                                    *reinterpret_cast<u32x4*>(c) = o;
                                    continue;
                                }
                                if (KIND == MI355_EPI_SWIGLU_FWD) {
                                    // this wave's 64 staging columns = [u (32) | g (32)] of hidden units hid0..hid0+31 (see the DMA plan): lanes 0-3 of each
                                    // row group write their 8 u into C[:, hid], lanes 4-7 their 8 g into C[:, N/2 + hid]; EVERY lane then computes four of
                                    // the group's 32 activations a = u * silu(g) (its own operand half from registers, the partner's from the staging row)
                                    // and writes them into R[:, hid] -- with the gate lanes idle the pass was bound by what the value lanes had to issue.
                                    // u, g are rounded to bf16 first, so a equals mi355_swiglu_fwd on the stored gate-up output bit for bit.
                                    const int l8 = lane & 7;
                                    const int64_t nh = p.N >> 1;
                                    const int64_t hid = (n0 >> 1) + ((wc0 + sn * 64) >> 6) * 32 + (l8 & 3) * 8;
                                    if (hid >= nh) continue;
                                    bf16_t* gu_row = reinterpret_cast<bf16_t*>(p.C) + gm * p.ldc;
                                    u32x4 own;
#pragma unroll
                                    for (int e = 0; e < 4; ++e) own[e] = pack_bf2(v[2 * e], v[2 * e + 1]);
                                    const int hi4 = l8 >> 2;  // value lanes take units 0-3 of their eight, gate lanes units 4-7
                                    *reinterpret_cast<u32x4*>(gu_row + (hi4 ? nh : 0) + hid) = own;
                                    // partner operand of my four units: staging columns (other half) + (l8 & 3) * 8 + 4 * hi4
                                    const f32x4 pv = *reinterpret_cast<const f32x4*>(stg + row * EPI_LD + (hi4 ? 0 : 32) + (l8 & 3) * 8 + 4 * hi4);
                                    float a4[4];
                                    const unsigned own0 = hi4 ? own[2] : own[0], own1 = hi4 ? own[3] : own[1];  // selects, not an indexed vector and a divergent branch
#pragma unroll
                                    for (int e = 0; e < 4; ++e) {
                                        const unsigned w = (e >> 1) ? own1 : own0;
                                        const float mine = (e & 1) ? __uint_as_float(w & 0xffff0000u) : __uint_as_float(w << 16);
                                        const float other = bf2f(f2bf(pv[e]));
                                        a4[e] = swiglu_act(hi4 ? other : mine, hi4 ? mine : other);
                                    }
                                    const u32x2 av = {pack_bf2(a4[0], a4[1]), pack_bf2(a4[2], a4[3])};
                                    *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(const_cast<void*>(p.R)) + gm * p.ldr + hid + 4 * hi4) = av;
                                    continue;
                                }
                                if (KIND == MI355_EPI_SWIGLU_BWD) {
                                    // acc = d(act) for hidden units gn..gn+7; R = the forward's gate-up output [u | g] (ldr = 2N): write
                                    // d(gate-up) = [acc * g*sig(g) | acc * u * sig(g) (1 + g (1 - sig(g)))] into C (ldc = 2N).  acc is rounded to
                                    // bf16 first, so the result equals mi355_swiglu_bwd on the stored bf16 d(act) bit for bit.
                                    u32x4 uv, gv;
                                    if constexpr (FULL && GU_EARLY) {
                                        uv = ucur[tpass];
                                        gv = gcur[tpass];
                                    } else {
                                        uv = *reinterpret_cast<const u32x4*>(r);
                                        gv = *reinterpret_cast<const u32x4*>(r + p.N);
                                    }
                                    float du[8], dg[8];
#pragma unroll
                                    for (int e = 0; e < 4; ++e) {
#pragma unroll
                                        for (int hlf = 0; hlf < 2; ++hlf) {
                                            const float u_ = hlf ? __uint_as_float(uv[e] & 0xffff0000u) : __uint_as_float(uv[e] << 16);
                                            const float g_ = hlf ? __uint_as_float(gv[e] & 0xffff0000u) : __uint_as_float(gv[e] << 16);
                                            swiglu_grads(bf2f(f2bf(v[2 * e + hlf])), u_, g_, du[2 * e + hlf], dg[2 * e + hlf]);
                                        }
                                    }
                                    u32x4 o0, o1;
#pragma unroll
                                    for (int e = 0; e < 4; ++e) {
                                        o0[e] = pack_bf2(du[2 * e], du[2 * e + 1]);
                                        o1[e] = pack_bf2(dg[2 * e], dg[2 * e + 1]);
                                    }
                                    *reinterpret_cast<u32x4*>(c) = o0;
                                    *reinterpret_cast<u32x4*>(c + p.N) = o1;
                                    continue;
                                }
                                if constexpr (KSEL == MI355_EPI_ATTN_DELTA) {
                                    // acc = d(ctx); R = the forward's ctx: besides the plain store, the row's dot product over this sub-block's 64 columns
                                    // (half a 128-wide head) of the ROUNDED d(ctx) with ctx -- the eight lanes of a row group fold it, lane 0 of the group parks
                                    // it in the staging row's padding column `sm` (columns 64..67 are never staged into)
                                    u32x4 o, cv;
#pragma unroll
                                    for (int e = 0; e < 4; ++e) o[e] = pack_bf2(v[2 * e], v[2 * e + 1]);
                                    if constexpr (FULL && RES_AHEAD) cv = rcur[tpass];
                                    else cv = *reinterpret_cast<const u32x4*>(r);
                                    float dsum = 0.f;
#pragma unroll
                                    for (int e = 0; e < 4; ++e)
                                        dsum += __uint_as_float(o[e] << 16) * __uint_as_float(cv[e] << 16) + __uint_as_float(o[e] & 0xffff0000u) * __uint_as_float(cv[e] & 0xffff0000u);
                                    *reinterpret_cast<u32x4*>(c) = o;
                                    dsum += __shfl_xor(dsum, 1, 64);
                                    dsum += __shfl_xor(dsum, 2, 64);
                                    dsum += __shfl_xor(dsum, 4, 64);
                                    if ((lane & 7) == 0) stg[row * EPI_LD + 64 + sm] = dsum;
                                    continue;
                                }
                                if (vec_ok) {
                                    if (r) {
                                        u32x4 rv;
                                        if constexpr (FULL && RES_AHEAD) rv = rcur[tpass];  // ksplit == 1 on this branch: fetched a sub-block ago
                                        else rv = *reinterpret_cast<const u32x4*>(r);
#pragma unroll
                                        for (int e = 0; e < 4; ++e) {
                                            v[2 * e] += __uint_as_float(rv[e] << 16);
                                            v[2 * e + 1] += __uint_as_float(rv[e] & 0xffff0000u);
                                        }
                                    }
                                    u32x4 o;
#pragma unroll
                                    for (int e = 0; e < 4; ++e) o[e] = pack_bf2(v[2 * e], v[2 * e + 1]);
#if defined(GEMM_EPI_ABL) && (GEMM_EPI_ABL & 2)  // profiling: the whole write-out except the store instruction itself
                                    if (o[0] == 0x7fc12345u && p.M < 0)
#endif
                                    *reinterpret_cast<u32x4*>(c) = o;
                                } else {
                                    for (int e = 0; e < nvalid; ++e) c[e] = f2bf(v[e] + (r ? bf2f(r[e]) : 0.f));
                                }
                            } else {
                                // (NT on the 8- / 4x64-wave tiles only: one more copy of the passes stops the unroller on the 4-wave 128x128 tile and its accumulators land in scratch)
                                constexpr bool SPLIT3_OK = !A_TR && !B_TR && !std::is_same_v<T, Cfg256w>;
                                if (SPLIT3_OK && p.split3) {  // (N % 8 == 0, 16-byte aligned C, no residual: checked by the entry point)
                                    float lo[8];
#pragma unroll
                                    for (int e = 0; e < 8; ++e) lo[e] = v[e] - bf2f(f2bf(v[e]));
                                    const u32x4 hv = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
                                    const u32x4 lv = {pack_bf2(lo[0], lo[1]), pack_bf2(lo[2], lo[3]), pack_bf2(lo[4], lo[5]), pack_bf2(lo[6], lo[7])};
                                    bf16_t* c3 = reinterpret_cast<bf16_t*>(p.C) + gm * p.ldc + gn;
                                    *reinterpret_cast<u32x4*>(c3) = hv;
                                    *reinterpret_cast<u32x4*>(c3 + p.N) = lv;
                                    *reinterpret_cast<u32x4*>(c3 + 2 * p.N) = hv;
                                } else {
                                float* c = reinterpret_cast<float*>(p.C) + gm * p.ldc + gn;
                                const float* r = p.R ? reinterpret_cast<const float*>(p.R) + gm * p.ldr + gn : nullptr;
                                if (vec_ok) {
                                    f32x4 o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
                                    if (r) {
                                        o0 += *reinterpret_cast<const f32x4*>(r);
                                        o1 += *reinterpret_cast<const f32x4*>(r + 4);
                                    }
                                    *reinterpret_cast<f32x4*>(c) = o0;
                                    *reinterpret_cast<f32x4*>(c + 4) = o1;
                                } else {
                                    for (int e = 0; e < nvalid; ++e) c[e] = v[e] + (r ? r[e] : 0.f);
                                }
                                }
                            }
                        }
                    }
                };
                if constexpr (KSEL >= 0) {
                    if (aligned_io && gm0 + 64 <= p.M && n0 + wc0 + sn * 64 + 64 <= p.N) passes(std::true_type{});
                    else passes(std::false_type{});
                } else {
                    passes(std::false_type{});
                }
                __builtin_amdgcn_wave_barrier();  // the staging area is rewritten by the next 64x64 sub-block
            }
        }
        if constexpr (KSEL == MI355_EPI_ATTN_DELTA) {
            // every wave has parked 128 half-head row sums; a head's two halves sit in two neighbouring waves: one thread per (row, head) adds them
            // in a fixed order and writes the three row-constant arrays of the attention backward
            static_assert(T::NTHREADS == 512 && T::BM == 256 && T::BN == 256 && T::WN == 4 && T::WTN == 64, "delta write-out: 2 x 4 waves of 128 x 64 on a 256 x 256 tile");
            __syncthreads();
            const int t_ = threadIdx.x, row = t_ & 255, hsel = t_ >> 8;
            const float* fs = reinterpret_cast<const float*>(smem);
            const int w0 = (row >> 7) * 4 + 2 * hsel, r64 = row & 63, smi = (row >> 6) & 1;
            const float sum = fs[(w0 * 64 + r64) * EPI_LD + 64 + smi] + fs[((w0 + 1) * 64 + r64) * EPI_LD + 64 + smi];
            const int64_t gm = m0 + row;
            const int64_t head = (n0 >> 7) + hsel;
            if (gm < p.M && head < p.ad_Hq) {
                const int64_t bb = gm / p.ad_S, sq = gm - bb * p.ad_S;
                const int64_t di = (bb * p.ad_Hq + head) * p.ad_S + sq;
                p.ad_delta[di] = sum;
                p.ad_ndl[di] = -sum;
                p.ad_nl2[di] = -p.ad_lse[di] * 1.4426950408889634f;
            }
        }
    };
    // a third copy of the passes for the SwiGLU-forward form (231.9 vs 233.6 ms/step); not on the 4-wave tile, where a third copy of its four
    // sub-blocks stops the unroller and the accumulators land in scratch
    constexpr bool SWIGLU_FWD_COPY = !A_TR && !B_TR && OUT_DT == MI355_DT_BF16 && T::NW == 8;
    constexpr bool SWIGLU_BWD_COPY = !A_TR && OUT_DT == MI355_DT_BF16 && T::NW == 8;  // the down-projection's dgrad (NT on W^T in the step, NN otherwise)
    constexpr bool DELTA_COPY = !A_TR && !B_TR && OUT_DT == MI355_DT_BF16 && T::NW == 8 && T::BK == 64;  // the out-projection's dgrad (NT on W^T), tile 2 only
    if (p.epilogue == MI355_EPI_NONE) {
        run_epilogue(std::integral_constant<int, MI355_EPI_NONE>{});
    } else if (DELTA_COPY && p.epilogue == MI355_EPI_ATTN_DELTA) {
        if constexpr (DELTA_COPY) run_epilogue(std::integral_constant<int, MI355_EPI_ATTN_DELTA>{});
    } else if (SWIGLU_FWD_COPY && p.epilogue == MI355_EPI_SWIGLU_FWD) {
        if constexpr (SWIGLU_FWD_COPY) run_epilogue(std::integral_constant<int, MI355_EPI_SWIGLU_FWD>{});  // the step's largest forward GEMM
    } else if (SWIGLU_BWD_COPY && p.epilogue == MI355_EPI_SWIGLU_BWD) {
        if constexpr (SWIGLU_BWD_COPY) run_epilogue(std::integral_constant<int, MI355_EPI_SWIGLU_BWD>{});
    } else {
        run_epilogue(std::integral_constant<int, -1>{});  // other fused forms: kind read at run time, bounds checked per row
    }
#if GEMM_TL
    TL_STAMP(4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TL_STAMP(5);
#endif
}

template <class T, bool A_TR, bool B_TR, int OUT_DT>
__global__ __launch_bounds__(T::NTHREADS, T::MIN_WAVES) void gemm_bf16_kernel(GemmParams p) {
    __shared__ __attribute__((aligned(16))) char smem[T::SMEM];
    const int nwg = p.tiles_m * p.tiles_n;
    const int split = blockIdx.x / nwg;
    gemm_tile<T, A_TR, B_TR, OUT_DT>(p, xcd_chunked(blockIdx.x - split * nwg, nwg), split, smem);
}

// ---------------------------------------------------------------------------------------------- persistent NT kernel (tile hint 7)
// tools/gemm_timeline.py (per-workgroup wall-clock stamps, round 4) on the QKV projection of the step, 256x256 tile, K = 1024: of a tile's 31.0 us turn on its CU the
// main loop is 24.8; the rest is prologue 1.9 (entry -> first K-tile landed), barrier 0.6, write-out 2.45 (LDS-bandwidth: 512 KiB of fp32 staging traffic per tile),
// stores retiring 0.5 and 0.8 between a workgroup's end and its successor's entry.  This kernel keeps ONE workgroup per CU alive over its share of the tiles:
//   * the K-tile stream runs across tile boundaries (the last two barriers of a tile request the next tile's first two K-tiles), so there is no prologue, no
//     retire wait and no workgroup turnover;
//   * the write-out stages through a 32-KiB region BESIDE the two stages (4 KiB per wave: 32 rows x 64 columns of packed bf16 at a time, swizzled, not padded), so it
//     needs no barrier and does not collide with the stream; the B fragment is the MFMA's first operand, so a lane holds four consecutive columns of a row and
//     packs them itself: a quarter of the LDS bytes of the fp32 staging.
// Results are bit-identical to gemm_bf16_kernel (same MFMA sequence per output, same rounding).  NT form, bf16 output, K % 64 == 0, N % 8 == 0, no bias.
#if GEMM_PART == 2 || !defined(GEMM_PART)
#if GEMM_TL
#define TLQ(q, i) do { if (threadIdx.x == 0 && (q) < 32768) { g_gemm_tl[q][i] = __builtin_amdgcn_s_memrealtime(); g_gemm_tl[q][(i) + 1] = __builtin_readcyclecounter(); } } while (0)  // + the shader clock's counter: cycles / time = the clock the tile ran at
#else
#define TLQ(q, i) do { } while (0)
#endif
template <int KIND, bool RES, bool WALK>  // KIND: MI355_EPI_NONE (RES: + bf16 residual, added in fp32 before the rounding), MI355_EPI_SWIGLU_FWD, MI355_EPI_SWIGLU_BWD (see gemm_tile's epilogue for both)
__global__ __launch_bounds__(512, 2) void gemm_nt_persist_kernel(GemmParams p, int ntiles) {
    using T = Cfg256;
    constexpr int BK = 64, HM = T::FM / 2;
    static_assert(T::A_PPW == 4 && T::B_PPW == 4 && T::FM == 8 && T::FN == 4 && T::NS == 2, "written for 8 waves of 128x64 on two 64-KiB stages");
    __shared__ __attribute__((aligned(16))) char smem[2 * T::STAGE + 32768];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char* const stg = smem + 2 * T::STAGE + wave * 4096;
    const int wr0 = (wave / T::WN) * T::WTM, wc0 = (wave % T::WN) * T::WTN;
    const int nt = (int)(p.K / BK);
    const int G = gridDim.x;

    // the walk: by default tile blockIdx.x + j G of the XCD-chunked order in 6-row groups.  WALK (ablate bit 3, MI355_GEMM_WALK): weight-stationary per XCD, as in
    // gemm_nt_pp_kernel -- column panels four at a time, an XCD owns a contiguous run of (column group, row panel) pairs, its 32 workgroups are 4 column panels x 8 row
    // streams, stream r takes the run's pairs r, r + 8, ...; at K = 1 024 an XCD's four weight panels (2 MB) stay in its L2 for the whole launch.  A column group's
    // missing panels (N / 256 not a multiple of 4) are walked as empty tiles (everything beyond the matrix: zero operands, no stores).  Needs the full grid of 256.
    constexpr bool walk = WALK;
    // cache policy of the output stores (0 default, 2 = nt, 16 = sc1, 18 = both).  Round 6, same-box bench.py pairs with the walk by shape: base 435.7 / 436.4 ms, nt on the
    // weight-stationary launches 434.3 / 435.1, sc1 436.1 / 435.3, sc1 nt 434.5 / 434.8: nt there (the tile is written once and next read by another kernel; the L2 keeps the
    // weight panels instead)
    constexpr int CAUX = WALK ? GEMM_C_AUX_WALK : GEMM_C_AUX_OLD;
    const int wk_xcd = blockIdx.x & 7, wk_l = blockIdx.x >> 3, wk_ci = wk_l & 3;
    const int64_t wk_pairs = (int64_t)((p.tiles_n + 3) >> 2) * p.tiles_m;
    const int wk_lo = (int)(wk_pairs * wk_xcd / 8) + (wk_l >> 2), wk_hi = (int)(wk_pairs * (wk_xcd + 1) / 8);
    const int wk_J = wk_hi > wk_lo ? (wk_hi - wk_lo + 7) / 8 : 0;
    const int qend = walk ? (int)blockIdx.x + wk_J * G : ntiles;  // this workgroup's tiles: q = blockIdx.x + j G < qend
    auto origin = [&](int q, int64_t& m0, int64_t& n0) {
        if (walk) {
            const int pp = wk_lo + 8 * ((q - (int)blockIdx.x) / G);
            const int cg = pp / p.tiles_m;
            m0 = (int64_t)(pp - cg * p.tiles_m) * T::BM;
            n0 = (int64_t)(cg * 4 + wk_ci) * T::BN;
            return;
        }
        const int pid = xcd_chunked(q, ntiles);
        constexpr int GROUP_M = 6;
        const int in_group = GROUP_M * p.tiles_n;
        const int first_m = (pid / in_group) * GROUP_M;
        const int gsz = min(p.tiles_m - first_m, GROUP_M);
        m0 = (int64_t)(first_m + (pid % in_group) % gsz) * T::BM;
        n0 = (int64_t)((pid % in_group) / gsz) * T::BN;
    };
    // ---- the request side of the stream: tile qi, K-tile ti, element number si (stage si & 1).  The LDS-DMA requests of a SIMD's two waves (w and w + 4) are divided by
    // OPERAND, not by wave: wave w requests the A-panel pieces of both (8 per K-tile), right behind the barrier that frees the stage; wave w + 4 requests the B-panel
    // pieces of both (weights: they come from L2) a phase later, under its partner's MFMAs.  Issued symmetrically (every wave its own 4 + 4 behind the barrier), both waves
    // of a SIMD stand in the CU's one address unit's queue at the same time and the matrix pipe idles; with the two request bursts half a K-tile apart, one wave of every
    // SIMD always multiplies.  (All 16 requests on wave w, none on w + 4: between the two -- the requesting wave's burst plus its own 64 MFMAs is the K-tile's critical path.)
    // The SwiGLU-backward form keeps all 16 requests on wave w (BY_OPERAND false): with the second request point in the K-tile its register allocation spills 72 values.
    constexpr bool BY_OPERAND = KIND != MI355_EPI_SWIGLU_BWD;
    const bool a_side = wave < 4;
    const int wlo = wave & 3;
    unsigned voff[2 * T::A_PPW];  // this wave's operand: pieces of waves wlo (0-3) and wlo + 4 (4-7)
    [[maybe_unused]] unsigned voff2[2 * T::B_PPW];  // !BY_OPERAND: wave w's B-panel pieces
    static_assert(T::A_PPW == T::B_PPW, "one offset array serves either operand");
    const bf16_t* baseX = p.A;
    [[maybe_unused]] const bf16_t* baseY = p.B;
    auto plan = [&](int q) {
        int64_t m0, n0;
        origin(q, m0, n0);
        int l = lane;
        asm volatile("" : "+v"(l));  // recomputed from the lane number once per tile: hoisted, the per-piece rows and chunks would occupy 16 registers across the main loop
        if (a_side) {  // (wave-uniform)
            baseX = uniform_ptr(p.A + m0 * p.lda);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                unsigned v[T::A_PPW];
                int unused[T::A_PPW];
                piece_offsets<false, T::BM, BK, T::A_PPW>(wlo + 4 * h, l, p.lda, p.M - m0, v, unused);
#pragma unroll
                for (int j = 0; j < T::A_PPW; ++j) voff[h * T::A_PPW + j] = v[j];
                if constexpr (!BY_OPERAND) {
                    piece_offsets<false, T::BN, BK, T::B_PPW>(wlo + 4 * h, l, p.ldb, p.N - n0, v, unused);
#pragma unroll
                    for (int j = 0; j < T::B_PPW; ++j) voff2[h * T::B_PPW + j] = v[j];
                }
            }
            if constexpr (!BY_OPERAND) baseY = uniform_ptr(p.B + n0 * p.ldb);
        } else if constexpr (BY_OPERAND) {
            baseX = KIND == MI355_EPI_SWIGLU_FWD ? uniform_ptr(p.B) : uniform_ptr(p.B + n0 * p.ldb);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int w = wlo + 4 * h;
                unsigned v[T::B_PPW];
                if constexpr (KIND == MI355_EPI_SWIGLU_FWD) {
                    // every 64 output columns of the tile are [32 lin1 rows | the 32 lin_gate rows of the SAME hidden units] of the fused weight (see gemm_tile)
                    const int64_t nh = p.N >> 1;
#pragma unroll
                    for (int j = 0; j < T::B_PPW; ++j) {
                        const int r = (w * T::B_PPW + j) * 8 + l / 8;
                        const int c = swz_rowk<BK>(l % 8, r);
                        const int64_t hid = (n0 >> 1) + (r >> 6) * 32 + (r & 31);
                        v[j] = hid < nh ? (unsigned)((((r >> 5) & 1) * nh + hid) * p.ldb * 2 + c * 16) : OOB;
                    }
                } else {
                    int unused[T::B_PPW];
                    piece_offsets<false, T::BN, BK, T::B_PPW>(w, l, p.ldb, p.N - n0, v, unused);
                }
#pragma unroll
                for (int j = 0; j < T::B_PPW; ++j) voff[h * T::B_PPW + j] = v[j];
            }
        }
    };
    int qi = blockIdx.x, ti = 0, si = 0;
    auto request_next = [&]() {  // this wave's operand of element si
        if (qi < qend) {
            char* dst = smem + (si & 1) * T::STAGE + (a_side ? 0 : T::A_BYTES) + wlo * T::A_PPW * 1024;
            const bf16_t* px = baseX + (int64_t)ti * BK;
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < T::A_PPW; ++j) dma_piece(px, voff[h * T::A_PPW + j], dst + (h * 4 * T::A_PPW + j) * 1024);
            if constexpr (!BY_OPERAND) {
                const bf16_t* py = baseY + (int64_t)ti * BK;
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int j = 0; j < T::B_PPW; ++j) dma_piece(py, voff2[h * T::B_PPW + j], dst + T::A_BYTES + (h * 4 * T::B_PPW + j) * 1024);
            }
            if (++ti == nt) {
                ti = 0;
                qi += G;
                if (qi < qend) plan(qi);
            }
        }
        ++si;
    };

    int qc = blockIdx.x, sc = 0;  // the multiplying side: tile qc, element number sc
    if (qc >= qend) return;
    plan(qi);
    f32x4 acc[T::FM][T::FN];
    bf16x8 aE[HM], aO[HM], b0[T::FN], b1[T::FN];
    auto loadA = [&](bf16x8 (&a)[HM], const char* sA, int kk, int mh) {
#pragma unroll
        for (int i = 0; i < HM; ++i) a[i] = frag_rowk<BK>(sA, wr0 + (mh * HM + i) * 16, kk, lane);
    };
    auto loadB = [&](bf16x8 (&b)[T::FN], const char* sB, int kk) {
#pragma unroll
        for (int j = 0; j < T::FN; ++j) b[j] = frag_rowk<BK>(sB, wc0 + j * 16, kk, lane);
    };
    // B fragment first: the lane holds C[row lane & 15][4 (lane >> 4) + e].  ZERO: the tile's first k-step starts from the constant 0 (no accumulator is cleared anywhere)
    auto mma = [&](bool zero, const bf16x8 (&a)[HM], const bf16x8 (&b)[T::FN], int mh) {
        if (zero) {  // wave-uniform
#pragma unroll
            for (int i = 0; i < HM; ++i)
#pragma unroll
                for (int j = 0; j < T::FN; ++j) acc[mh * HM + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < HM; ++i)
#pragma unroll
                for (int j = 0; j < T::FN; ++j) acc[mh * HM + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[mh * HM + i][j], 0, 0, 0);
        }
    };
    bool more_tiles = false;
    auto kstep = [&](int t) {  // one element of the stream: 64 MFMAs in four phases (k-step, row half), the next element's first fragments requested under the last
        const char* sA = smem + (sc & 1) * T::STAGE;
        const char* sB = sA + T::A_BYTES;
        loadA(aO, sA, 0, 1);
        mma(t == 0, aE, b0, 0);
        if (BY_OPERAND && !a_side && sc > 0) request_next();  // the B panels of element sc + 1 (its A panels were requested behind the last barrier; elements 0 and 1: before the loop)
        loadB(b1, sB, 1);
        loadA(aE, sA, 1, 0);
        mma(t == 0, aO, b0, 1);
        loadA(aO, sA, 1, 1);
        mma(false, aE, b1, 0);
        const bool last = t + 1 == nt;
        if (!last || more_tiles) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's reads of the element are complete ...
            wait_vmcnt<0>();                                    // ... its share of the next element has landed (and the previous tile's stores have retired)
            __builtin_amdgcn_s_barrier();                       // ... for every wave
            if (a_side) request_next();                         // the A panels of element sc + 2 into the stage just left: at the end of a tile that is the NEXT tile's stream
            if (!last) {
                const char* nA = smem + ((sc + 1) & 1) * T::STAGE;
                loadB(b0, nA + T::A_BYTES, 0);
                loadA(aE, nA, 0, 0);
            }
        }
        mma(false, aO, b1, 1);
        ++sc;
    };
    if (BY_OPERAND || a_side) request_next();  // elements 0 and 1: both operands
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (BY_OPERAND || a_side) request_next();
    loadB(b0, smem + T::A_BYTES, 0);
    loadA(aE, smem, 0, 0);
    for (;;) {
        TLQ(qc, 0);
        more_tiles = qc + G < qend;
        for (int t = 0; t < nt; ++t) kstep(t);
        TLQ(qc, 2);
        // ---- write-out of tile qc: 4 sub-blocks of 32 rows x 64 columns per wave; packed bf16 through this wave's own 4 KiB.  LDS operations of a wave execute in
        // order, so sub-block sb + 1 is staged right behind the row reads of sub-block sb, whose stores go out while it is being packed; rows and columns beyond the
        // matrix are dropped by the buffer range check (offset OOB), so the passes have no branches.
        int le = lane;
        asm volatile("" : "+v"(le));  // derived per tile, not kept across the main loop
        const int g = le >> 4, r16 = le & 15;
        // staging addresses of this lane.  Write side: row r = ii * 16 + r16 of a 32 x 64 sub-block (128-byte rows), 16-byte chunk (2 j + g / 2) ^ (r & 7), 8-byte half
        // (g & 1) ^ (r >> 3 & 1): the 16 rows one store instruction covers hit 32 distinct banks.  Read side: row pass * 8 + lane / 8, chunk (lane & 7) ^ (row & 7).
        const int wr_off = r16 * 128 + ((((g & 1) ^ (r16 >> 3)) & 1) << 3), wr_x = r16 & 7;
        const int rd_row = le >> 3, rd_ch = le & 7;
        const int rd_off = rd_row * 128 + ((rd_ch ^ (rd_row & 7)) << 4);  // (pass * 8 + rd_row) & 7 == rd_row & 7
        int64_t m0, n0;
        origin(qc, m0, n0);
        const int64_t rows_left = p.M - m0 - wr0;
        u32x4 rd[4];
        [[maybe_unused]] u32x2 rres[2][T::FN];   // RES: the residual of a sub-block in the accumulators' layout (8-byte loads)
        [[maybe_unused]] u32x4 uv[4], gv[4];     // SwiGLU backward: the forward's gate-up pair of a sub-block's rows (16-byte loads)
        [[maybe_unused]] u32x2 partner[4];       // SwiGLU forward: the other operand of the four units this lane activates
        [[maybe_unused]] float psum[4][4];       // attention delta: this wave's half-head row sums (sub-block, pass), complete in every lane of the row's eight
        const int hi4 = rd_ch >> 2;
        // column base of this wave in C (and R), validity of this lane's 8 columns
        int64_t cbase, cbase2 = 0;
        bool col_ok;
        if constexpr (KIND == MI355_EPI_SWIGLU_FWD) {
            const int64_t nh = p.N >> 1, hid = (n0 >> 1) + (wc0 >> 6) * 32 + (rd_ch & 3) * 8;
            col_ok = hid < nh;
            cbase = (hi4 ? nh : 0) + hid;  // gate-up output: u into [:, hid], g into [:, N/2 + hid]
            cbase2 = hid + 4 * hi4;        // activation output (R): four units per lane
        } else {
            cbase = n0 + wc0 + rd_ch * 8;
            col_ok = cbase < p.N;
        }
        auto rsrc_c = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<bf16_t*>(p.C) + (m0 + wr0) * p.ldc, 0, 0x7fffffff, 0x00020000);
        auto rsrc_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(reinterpret_cast<const bf16_t*>(p.R)) + (m0 + wr0) * p.ldr, 0, 0x7fffffff, 0x00020000);
        auto fetch_res = [&](int sb) {  // residual of sub-block sb, requested a sub-block before it is packed
            if constexpr (RES) {
#pragma unroll
                for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                    for (int j = 0; j < T::FN; ++j) {
                        const int row = sb * 32 + ii * 16 + r16;
                        const int64_t col = n0 + wc0 + j * 16 + 4 * g;
                        rres[ii][j] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc_r, (row < rows_left && col < p.N) ? (unsigned)((row * p.ldr + col) * 2) : OOB, 0, 0));
                    }
            }
        };
        auto fetch_ug = [&](int sb) {  // gate-up rows of sub-block sb, requested when it is staged, used a sub-block later (after the stores of sb - 1 have consumed theirs)
            if constexpr (KIND == MI355_EPI_SWIGLU_BWD || KIND == MI355_EPI_ATTN_DELTA) {
#pragma unroll
                for (int pass = 0; pass < 4; ++pass) {
                    const int row = sb * 32 + pass * 8 + rd_row;
                    const bool ok = col_ok && row < rows_left;
                    uv[pass] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_r, ok ? (unsigned)((row * p.ldr + cbase) * 2) : OOB, 0, 0));  // delta: the forward's ctx
                    if constexpr (KIND == MI355_EPI_SWIGLU_BWD)
                        gv[pass] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_r, ok ? (unsigned)((row * p.ldr + cbase + p.N) * 2) : OOB, 0, 0));
                }
            }
        };
        auto stores = [&](int sb) {
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const u32x4 o = (pass & 1) ? (u32x4){rd[pass][2], rd[pass][3], rd[pass][0], rd[pass][1]} : rd[pass];  // rows 8-15 of a 16-row tile were written with their halves exchanged
                const int row = sb * 32 + pass * 8 + rd_row;
                const bool ok = col_ok && row < rows_left;
                if constexpr (KIND == MI355_EPI_SWIGLU_BWD) {
                    // o = d(act) of hidden units cbase..+7, rounded to bf16: d(gate-up) = [d * g sig(g) | d * u sig(g) (1 + g (1 - sig(g)))] into C[:, unit], C[:, N + unit] (ldc = 2N)
                    u32x4 o0, o1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float du[2], dg[2];
#pragma unroll
                        for (int hlf = 0; hlf < 2; ++hlf) {
                            const float d_ = hlf ? __uint_as_float(o[e] & 0xffff0000u) : __uint_as_float(o[e] << 16);
                            const float u_ = hlf ? __uint_as_float(uv[pass][e] & 0xffff0000u) : __uint_as_float(uv[pass][e] << 16);
                            const float g_ = hlf ? __uint_as_float(gv[pass][e] & 0xffff0000u) : __uint_as_float(gv[pass][e] << 16);
                            swiglu_grads(d_, u_, g_, du[hlf], dg[hlf]);
                        }
                        o0[e] = pack_bf2(du[0], du[1]);
                        o1[e] = pack_bf2(dg[0], dg[1]);
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(o0, rsrc_c, ok ? (unsigned)((row * p.ldc + cbase) * 2) : OOB, 0, CAUX);
                    __builtin_amdgcn_raw_buffer_store_b128(o1, rsrc_c, ok ? (unsigned)((row * p.ldc + cbase + p.N) * 2) : OOB, 0, CAUX);
                } else {
                    __builtin_amdgcn_raw_buffer_store_b128(o, rsrc_c, ok ? (unsigned)((row * p.ldc + cbase) * 2) : OOB, 0, CAUX);
                    if constexpr (KIND == MI355_EPI_ATTN_DELTA) {
                        // o = d(ctx), rounded; uv = ctx: the row's dot product over this lane's 8 columns, folded over the row's eight lanes = half a head (the wave's 64 columns)
                        float dsum = 0.f;
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            dsum += __uint_as_float(o[e] << 16) * __uint_as_float(uv[pass][e] << 16) + __uint_as_float(o[e] & 0xffff0000u) * __uint_as_float(uv[pass][e] & 0xffff0000u);
                        dsum += __shfl_xor(dsum, 1, 64);
                        dsum += __shfl_xor(dsum, 2, 64);
                        dsum += __shfl_xor(dsum, 4, 64);
                        psum[sb][pass] = dsum;
                    }
                    if constexpr (KIND == MI355_EPI_SWIGLU_FWD) {
                        // this lane's 8 columns are u (chunks 0-3) or g (chunks 4-7) of 8 hidden units; it activates four of them: units 0-3 (u lanes) resp. 4-7 (g lanes),
                        // the other operand = the partner lane's chunk, read back from the staging rows; a = u * silu(g) on the ROUNDED operands, as mi355_swiglu_fwd sees them
                        // (operands are picked with selects, word by word: written as `hi4 ? act(other, mine) : act(mine, other)` on an indexed `o[2 * hi4 + ..]`, every lane
                        // ran both sides of a divergent branch -- twice the exponentials -- behind a chain of compares for the index)
                        const u32x2 pv = partner[pass];
                        const unsigned own0 = hi4 ? o[2] : o[0], own1 = hi4 ? o[3] : o[1];
                        const unsigned uw[2] = {hi4 ? pv[0] : own0, hi4 ? pv[1] : own1}, gw[2] = {hi4 ? own0 : pv[0], hi4 ? own1 : pv[1]};
                        float a4[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float u_ = (e & 1) ? __uint_as_float(uw[e >> 1] & 0xffff0000u) : __uint_as_float(uw[e >> 1] << 16);
                            const float g_ = (e & 1) ? __uint_as_float(gw[e >> 1] & 0xffff0000u) : __uint_as_float(gw[e >> 1] << 16);
                            a4[e] = swiglu_act(u_, g_);
                        }
                        const u32x2 av = {pack_bf2(a4[0], a4[1]), pack_bf2(a4[2], a4[3])};
                        __builtin_amdgcn_raw_buffer_store_b64(av, rsrc_r, ok ? (unsigned)((row * p.ldr + cbase2) * 2) : OOB, 0, 0);
                    }
                }
            }
        };
        fetch_res(0);
#pragma unroll
        for (int sb = 0; sb < 4; ++sb) {
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int j = 0; j < T::FN; ++j) {
                    f32x4 v = acc[sb * 2 + ii][j];
                    if constexpr (RES) {
                        v[0] += __uint_as_float(rres[ii][j][0] << 16);
                        v[1] += __uint_as_float(rres[ii][j][0] & 0xffff0000u);
                        v[2] += __uint_as_float(rres[ii][j][1] << 16);
                        v[3] += __uint_as_float(rres[ii][j][1] & 0xffff0000u);
                    }
                    const u32x2 pk = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
                    *reinterpret_cast<u32x2*>(stg + ii * 2048 + wr_off + (((2 * j + (g >> 1)) ^ wr_x) << 4)) = pk;
                }
            if (sb > 0) stores(sb - 1);
            if (sb < 3) fetch_res(sb + 1);
            fetch_ug(sb);
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                rd[pass] = *reinterpret_cast<const u32x4*>(stg + pass * 1024 + rd_off);
                if constexpr (KIND == MI355_EPI_SWIGLU_FWD) {
                    // partner chunk rd_ch ^ 4 of the same row, its half hi4 (units 4-7 of the partner's eight for g lanes, 0-3 for u lanes); odd passes hold rows whose halves are exchanged
                    const int prow = rd_row;  // (pass * 8 + rd_row) & 7
                    partner[pass] = *reinterpret_cast<const u32x2*>(stg + pass * 1024 + rd_row * 128 + ((((rd_ch ^ 4) ^ (prow & 7))) << 4) + (((hi4 ^ pass) & 1) << 3));
                }
            }
        }
        stores(3);
        if constexpr (KIND == MI355_EPI_ATTN_DELTA) {
            // a head's two halves sit in two neighbouring waves: every wave parks its 128 row sums in its own staging region (behind its last row reads: LDS operations
            // execute in order), one barrier, then one thread per (row, head) adds the halves in a fixed order and writes the attention backward's three row constants.
            // The regions are next written in the following tile's write-out, which every wave reaches only through that tile's K-loop barriers.
            static_assert(T::NTHREADS == 512 && T::WN == 4 && T::WTN == 64, "delta write-out: 2 x 4 waves of 128 x 64 on a 256 x 256 tile");
            float* part = reinterpret_cast<float*>(stg);
#pragma unroll
            for (int sb = 0; sb < 4; ++sb)
#pragma unroll
                for (int pass = 0; pass < 4; ++pass)
                    if (rd_ch == 0) part[sb * 32 + pass * 8 + rd_row] = psum[sb][pass];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const int t_ = threadIdx.x, row = t_ & 255, hsel = t_ >> 8;
            const float* p0 = reinterpret_cast<const float*>(smem + 2 * T::STAGE + ((row >> 7) * 4 + 2 * hsel) * 4096);
            const float sum = p0[row & 127] + p0[1024 + (row & 127)];
            const int64_t gm = m0 + row, head = (n0 >> 7) + hsel;
            if (gm < p.M && head < p.ad_Hq) {
                const int64_t bb = gm / p.ad_S, sq = gm - bb * p.ad_S;
                const int64_t di = (bb * p.ad_Hq + head) * p.ad_S + sq;
                p.ad_delta[di] = sum;
                p.ad_ndl[di] = -sum;
                p.ad_nl2[di] = -p.ad_lse[di] * 1.4426950408889634f;
            }
        }
        TLQ(qc, 4);
        qc += G;
        if (qc >= qend) break;
        const char* nA = smem + (sc & 1) * T::STAGE;
        loadB(b0, nA + T::A_BYTES, 0);
        loadA(aE, nA, 0, 0);
    }
}

int launch_persist(GemmParams p, hipStream_t s) {
    using T = Cfg256;
    p.tiles_m = (int)((p.M + T::BM - 1) / T::BM);
    p.tiles_n = (int)((p.N + T::BN - 1) / T::BN);
    const int64_t tiles = (int64_t)p.tiles_m * p.tiles_n;
    MI355_REQUIRE(tiles < 0x7fffffffLL / 16, "mi355_gemm_bf16: grid too large");
    const dim3 grid((unsigned)((tiles < 256 && !(p.ablate & 8)) ? tiles : 256)), block(T::NTHREADS);  // one workgroup per CU
#define PERSIST_LAUNCH(KIND_, RES_)                                                                                                      \
    do {                                                                                                                                 \
        if (p.ablate & 8) hipLaunchKernelGGL((gemm_nt_persist_kernel<KIND_, RES_, true>), grid, block, 0, s, p, (int)tiles);               \
        else hipLaunchKernelGGL((gemm_nt_persist_kernel<KIND_, RES_, false>), grid, block, 0, s, p, (int)tiles);                           \
    } while (0)
    if (p.epilogue == MI355_EPI_SWIGLU_FWD) PERSIST_LAUNCH(MI355_EPI_SWIGLU_FWD, false);
    else if (p.epilogue == MI355_EPI_SWIGLU_BWD) PERSIST_LAUNCH(MI355_EPI_SWIGLU_BWD, false);
    else if (p.epilogue == MI355_EPI_ATTN_DELTA) PERSIST_LAUNCH(MI355_EPI_ATTN_DELTA, false);
    else if (p.R) PERSIST_LAUNCH(MI355_EPI_NONE, true);
    else PERSIST_LAUNCH(MI355_EPI_NONE, false);
#undef PERSIST_LAUNCH
    MI355_LAUNCH_CHECK("mi355_gemm_bf16(persistent)");
    return 0;
}
#endif  // part 2

// ---------------------------------------------------------------------------------------------- ping-pong persistent NT kernel (tile hint 8)
// gemm_nt_persist_kernel's two waves per SIMD (w and w + 4: the upper and the lower 128 rows of the 256 x 256 tile) reach their write-out together, so the
// matrix pipe of every SIMD idles for the whole write-out: 2.45 of a tile's 27.5 us with the plain epilogue at K = 1 024, a quarter of the tile with the SwiGLU
// forms (round 5: 1.26 PF plain, 1.11 / 0.84 PF fused).  VALU / LDS / store issue and the matrix pipe are separate resources of a SIMD; what keeps them from
// overlapping is only that both waves are in the same phase.  This kernel runs the two wave groups E K-tiles apart:
//   * time is counted in SLOTS (one K-tile of the stream, one workgroup barrier each).  A group's tile takes NT = K / 64 multiplying slots and E write-out
//     slots (the write-out cut into E chunks, one per slot); the lower group (waves 4-7) starts E slots late.  So whenever one group writes out, the other
//     multiplies: the SIMD's matrix pipe belongs to the multiplying wave alone and the write-out's vector / LDS / store instructions issue beside it.
//   * the stream stays ONE stream: a slot's stage holds the weight panel's K-tile k = slot mod NT (both groups use it), the upper 128 rows of the activation
//     panel if the upper group multiplies in that slot and the lower 128 rows if the lower group does.  A tile therefore starts at whatever k the stream is
//     at and wraps around: every output is the sum over all K-tiles in an order ROTATED by a multiple of E, so results equal the per-tile kernel's up to the
//     fp32 summation order (bit-identical whenever the products are exact; tests/test_kernels_gpu.py), and are the same bits on every run.
//   * requests: every wave asks for its own four pieces of the weight panel and, for slots in which its group multiplies, its own four pieces of the activation
//     panel; waves 0-3 right behind the barrier that frees the stage, waves 4-7 half a slot later (the CU has one address unit: see gemm_nt_persist_kernel).
// NT form, bf16 output, K % 64 == 0, K / 64 >= E, N % 8 == 0, no bias.
//
// STATUS (round 6): correct on every epilogue kind (tools/experimental/pp_check.py: bit-identical to tile 2 on integer operands, fp32-summation noise on random ones), and
// SLOWER than gemm_nt_persist_kernel wherever the write-out matters, so the library does not choose it (MI355_GEMM_PP_DEFAULT 0; tile hint 8 / MI355_GEMM_PP select it).
// Batch 160, isolated, us (hint 7 -> hint 8): QKV 771 -> 881, out-proj + residual 424 -> 453, gate-up + SwiGLU 1 310 -> 1 451, down dgrad + SwiGLU backward 905 -> 1 063,
// LM head forward 21.8 -> 22.6 ms; long-K shapes level (dqkv 733 -> 752, LM head dgrad 19.3 -> 18.6 ms: the walk).  In the step: +1.5 ... +7 ms per kind.  Why:
//   * with NO write-out work at all (ablate bit 0) the same launches take 732 / 360 / 1 073 / 621 us and 19.3 ms: -5 ... -31 % -- the overlap is there to be had;
//   * but a slot in which only one group multiplies does not get shorter than ~0.9 us (against 1.55 us for a slot with both): the stream is two stages deep, a slot's
//     panels are requested one slot ahead, and under load an LDS-DMA request takes about that long to land -- the lone multiplying wave waits for data, not for the pipe;
//   * and whatever the writing group puts into the memory pipeline lengthens it further: a sub-block's four stores +0.43 us per slot (ablate bit 1: stores dropped),
//     its staging +0.17 us; the SwiGLU backward's gate-up loads ~2 us per slot unless requested a slot ahead, and requested ahead (two register sets) the kernel spills.
//   So a write-out of E chunks costs 2 E slots of 1.5-1.6 us in which the SIMD's matrix pipe does half the work -- more than the 2.45 us it costs with both waves idle.
//   What would change it: a third stage (192 KB: does not fit beside 256 x 256 tiles), or panels requested two slots ahead into the half of a stage the writing group does
//   not use.  Tried and dropped on the way: the multiplying group issuing every request while the other writes out (main loop -8 % from the extra scalar state), eight
//   16-row slots for the SwiGLU backward (1 294 us), its operands a slot ahead in two register sets (36 spilled values, 2 715 us).
#if GEMM_PART == 6 || !defined(GEMM_PART)
template <int KIND, bool RES>
__global__ __launch_bounds__(512, 2) void gemm_nt_pp_kernel(GemmParams p, int ntiles) {
    using T = Cfg256;
    constexpr int BK = 64, HM = T::FM / 2;
    constexpr int E = KIND == MI355_EPI_ATTN_DELTA ? 5 : 4;  // write-out slots: one 32-row sub-block each (+ the row constants' combine)
    static_assert(T::A_PPW == 4 && T::B_PPW == 4 && T::FM == 8 && T::FN == 4 && T::NS == 2, "written for 8 waves of 128x64 on two 64-KiB stages");
    __shared__ __attribute__((aligned(16))) char smem[2 * T::STAGE + 32768];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool grpY = wave >= 4;
    char* const stg = smem + 2 * T::STAGE + wave * 4096;
    const int wr0 = (wave / T::WN) * T::WTM, wc0 = (wave % T::WN) * T::WTN;
    const int NT = (int)(p.K / BK), P = NT + E;
    // ---- the walk: weight-stationary per XCD.  The column panels are taken four at a time (a column group); an XCD owns a contiguous run of (column group, row
    // panel) pairs; its 32 workgroups are 4 column panels x 8 row streams, stream r taking the run's pairs r, r + 8, ...  In a slot the XCD's workgroups therefore
    // ask for 4 weight K-tiles and 8 activation K-tiles between them (the slot number is the same K-tile for all of them), and at K = 1 024 the four weight panels
    // (2 MB) stay in the 4-MB L2 for the whole run: only the activation panels stream.  blockIdx.x & 7 names the XCD under round-robin dispatch (speed only).
    const int nsr = p.tiles_m, ncp = p.tiles_n, ncg = (ncp + 3) >> 2;
    const int xcd = blockIdx.x & 7, wlx = blockIdx.x >> 3, ci = wlx & 3;
    const int64_t pairs = (int64_t)ncg * nsr;
    const int lo = (int)(pairs * xcd / 8) + (wlx >> 2), hi = (int)(pairs * (xcd + 1) / 8);
    struct Walk {
        int p, sr, cg;
    };
    auto settle = [&](Walk& w) {  // skip pairs whose column group has no panel for this workgroup's column
        while (w.p < hi && w.cg * 4 + ci >= ncp) {
            w.p += 8;
            w.sr += 8;
            while (w.sr >= nsr) w.sr -= nsr, ++w.cg;
        }
    };
    auto walk_init = [&](Walk& w) {
        w.p = lo;
        w.cg = lo / nsr;
        w.sr = lo - w.cg * nsr;
        settle(w);
    };
    auto walk_next = [&](Walk& w) {
        w.p += 8;
        w.sr += 8;
        while (w.sr >= nsr) w.sr -= nsr, ++w.cg;
        settle(w);
    };
    int J = 0;  // tiles of this workgroup
    {
        Walk w;
        for (walk_init(w); w.p < hi; walk_next(w)) ++J;
    }
    if (J == 0) return;
    auto tile_m0 = [&](const Walk& w) { return (int64_t)w.sr * T::BM; };
    auto tile_n0 = [&](const Walk& w) { return (int64_t)(w.cg * 4 + ci) * T::BN; };

    // ---- the request side, divided by OPERAND as in gemm_nt_persist_kernel: waves 0-3 ask for the activation panel (both halves: it comes from HBM, so early, right
    // behind the barrier that frees the stage), waves 4-7 for the weight panel (L2) half a slot later.  A wave therefore follows two streams of four pieces each:
    //   waves 0-3: stream 1 = rows 32 w .. + 31 of the upper group's tile (on the upper group's clock), stream 2 = the same rows of the lower half (lower group's clock: E behind);
    //   waves 4-7: streams 1 / 2 = weight-panel rows 64 (w - 4) .. + 31 / + 32 .. + 63, both on the upper group's clock (the weight stream follows the upper group's tile
    //              index: while the upper group writes out, the lower one is still on that tile).
    // rstage / rkt = stage and K-tile of the slot the next call asks for; ph / tj = that slot's phase and tile on the stream's clock.  Per-lane piece offsets do not
    // depend on the tile (rows r0 + 8 j + lane / 8, 16-byte chunk swizzled by the row): a tile contributes a scalar base and a scalar row limit.
    // (Tried: the multiplying group asking for everything while the other one writes out, so that no request sits behind a chunk's stores in a wave's in-order
    // vmcnt -- slower: the main loop pays for the extra scalar state, and the stores cost the same.)
    unsigned voff1[4], voff2[4];
    const int lr = lane >> 3;
    const int wl = wave & 3;
    {
        int unused[4];
        if (!grpY) {
            piece_offsets<false, T::BM, BK, 4>(wl, lane, p.lda, 1 << 30, voff1, unused);
            piece_offsets<false, T::BM, BK, 4>(wl + 4, lane, p.lda, 1 << 30, voff2, unused);
        } else if constexpr (KIND == MI355_EPI_SWIGLU_FWD) {
            // every 64 output columns of the tile are [32 lin1 rows | the 32 lin_gate rows of the SAME hidden units] of the fused weight (see gemm_tile): the tile's
            // row r is weight row ((r >> 5) & 1) N/2 + n0 / 2 + (r >> 6) 32 + (r & 31); this wave's rows are 64 wl + 8 j + lr: stream 1 = lin1, stream 2 = lin_gate
            const int64_t nh = p.N >> 1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = wl * 64 + j * 8 + lr;
                voff1[j] = (unsigned)((wl * 32 + j * 8 + lr) * p.ldb * 2 + swz_rowk<BK>(lane & 7, r) * 16);
                voff2[j] = (unsigned)((nh + wl * 32 + j * 8 + lr) * p.ldb * 2 + swz_rowk<BK>(lane & 7, r + 32) * 16);
            }
        } else {
            piece_offsets<false, T::BN, BK, 4>(2 * wl, lane, p.ldb, 1 << 30, voff1, unused);
            piece_offsets<false, T::BN, BK, 4>(2 * wl + 1, lane, p.ldb, 1 << 30, voff2, unused);
        }
    }
    const bf16_t *base1 = p.A, *base2 = p.A;
    int lim1 = 0, lim2 = 0;  // rows of the stream's panel inside the matrix, minus the stream's first row: piece j is inside iff lr < lim - 8 j
    Walk w1, w2, wC;
    auto plan1 = [&]() {
        if (!grpY) {
            const int64_t m0 = tile_m0(w1);
            base1 = uniform_ptr(p.A + m0 * p.lda);
            lim1 = (int)min((int64_t)T::BM, p.M - m0) - wl * 32;
        } else {
            const int64_t n0 = tile_n0(w1);
            if constexpr (KIND == MI355_EPI_SWIGLU_FWD) {
                base1 = uniform_ptr(p.B + (n0 >> 1) * p.ldb);
                lim1 = lim2 = (int)min((int64_t)32, (p.N >> 1) - (n0 >> 1) - wl * 32);  // hidden units n0 / 2 + 32 wl + 8 j + lr of either half
            } else {
                base1 = uniform_ptr(p.B + n0 * p.ldb);
                lim1 = (int)min((int64_t)T::BN, p.N - n0) - wl * 64;
                lim2 = lim1 - 32;
            }
            base2 = base1;
        }
    };
    auto plan2 = [&]() {  // (waves 0-3 only)
        const int64_t m0 = tile_m0(w2);
        base2 = uniform_ptr(p.A + m0 * p.lda);
        lim2 = (int)min((int64_t)T::BM, p.M - m0) - 128 - wl * 32;
    };
    int rstage = 0, rkt = 0, ph1 = 0, tj1 = 0, ph2 = -E, tj2 = 0;
    auto request = [&]() __attribute__((always_inline)) {
        char* st = smem + rstage * T::STAGE;
        const int64_t kofs = (int64_t)rkt * BK;
        if (!grpY) {
            if (ph1 < NT && tj1 < J) {  // the upper group multiplies in that slot
#pragma unroll
                for (int j = 0; j < 4; ++j) dma_piece(base1 + kofs, lr < lim1 - 8 * j ? voff1[j] : OOB, st + (wl * 4 + j) * 1024);
            }
            if (ph2 >= 0 && ph2 < NT && tj2 < J) {  // the lower group does
#pragma unroll
                for (int j = 0; j < 4; ++j) dma_piece(base2 + kofs, lr < lim2 - 8 * j ? voff2[j] : OOB, st + (16 + wl * 4 + j) * 1024);
            }
            if (++ph2 == P) {
                ph2 = 0;
                if (++tj2 < J) {
                    walk_next(w2);
                    plan2();
                }
            }
        } else if (tj1 < J) {  // somebody multiplies in that slot
#pragma unroll
            for (int j = 0; j < 4; ++j) dma_piece(base1 + kofs, lr < lim1 - 8 * j ? voff1[j] : OOB, st + T::A_BYTES + (wl * 8 + j) * 1024);
#pragma unroll
            for (int j = 0; j < 4; ++j) dma_piece(base2 + kofs, lr < lim2 - 8 * j ? voff2[j] : OOB, st + T::A_BYTES + (wl * 8 + 4 + j) * 1024);
        }
        rstage ^= 1;
        if (++rkt == NT) rkt = 0;
        if (++ph1 == P) {
            ph1 = 0;
            if (++tj1 < J) {
                walk_next(w1);
                plan1();
            }
        }
    };

    // ---- the multiplying side (gemm_nt_persist_kernel's K-tile, unchanged)
    f32x4 acc[T::FM][T::FN];
    bf16x8 aE[HM], aO[HM], b0[T::FN], b1[T::FN];
    auto loadA = [&](bf16x8 (&a)[HM], const char* sA, int kk, int mh) {
#pragma unroll
        for (int i = 0; i < HM; ++i) a[i] = frag_rowk<BK>(sA, wr0 + (mh * HM + i) * 16, kk, lane);
    };
    auto loadB = [&](bf16x8 (&b)[T::FN], const char* sB, int kk) {
#pragma unroll
        for (int j = 0; j < T::FN; ++j) b[j] = frag_rowk<BK>(sB, wc0 + j * 16, kk, lane);
    };
    auto mma = [&](bool zero, const bf16x8 (&a)[HM], const bf16x8 (&b)[T::FN], int mh) {
        if (zero) {  // wave-uniform
#pragma unroll
            for (int i = 0; i < HM; ++i)
#pragma unroll
                for (int j = 0; j < T::FN; ++j) acc[mh * HM + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < HM; ++i)
#pragma unroll
                for (int j = 0; j < T::FN; ++j) acc[mh * HM + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[mh * HM + i][j], 0, 0, 0);
        }
    };
    int cstage = 0;
    bool started = false;  // a slot has passed (the lower group's requests run one slot behind the upper group's: none in slot 0)
    // the end of every slot, whatever the wave did in it: own reads and requests complete, everybody's too, the upper group's requests into the stage just left,
    // and -- if this wave multiplies in the next slot -- that slot's first fragments
    auto slot_end = [&](bool prefetch, auto keep_c) __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);  // the slot's third MFMA group stays in front of the wait (hipcc otherwise sinks most of it below the barrier, and the requests behind it)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        wait_vmcnt<decltype(keep_c)::value>();  // this wave's requests have landed; a write-out chunk's stores (its youngest operations) may stay in flight for another slot
        __builtin_amdgcn_s_barrier();
        if (!grpY) request();
        if (prefetch) {
            const char* nA = smem + (cstage ^ 1) * T::STAGE;
            loadB(b0, nA + T::A_BYTES, 0);
            loadA(aE, nA, 0, 0);
        }
        cstage ^= 1;
        started = true;
    };
    auto compute_slot = [&](bool first, bool last) {  // 64 MFMAs in four phases (k-step, row half)
        const char* sA = smem + cstage * T::STAGE;
        const char* sB = sA + T::A_BYTES;
        loadA(aO, sA, 0, 1);
        mma(first, aE, b0, 0);
        if (grpY) request();
        loadB(b1, sB, 1);
        loadA(aE, sA, 1, 0);
        mma(first, aO, b0, 1);
        loadA(aO, sA, 1, 1);
        mma(false, aE, b1, 0);
        slot_end(!last, std::integral_constant<int, 0>{});
        mma(false, aO, b1, 1);
    };

    // ---- write-out chunk c of this group's tile cj: sub-block c (32 rows x 64 columns per wave) through the wave's own 4 KiB of packed bf16, as in gemm_nt_persist_kernel
    [[maybe_unused]] float psum[4][4];  // attention delta: this wave's half-head row sums (sub-block, pass)
    auto epi_chunk = [&](auto sb_c, const int cj) __attribute__((always_inline)) {
        constexpr int sb = decltype(sb_c)::value;
        int le = lane;
        asm volatile("" : "+v"(le));  // derived per chunk, not kept across the main loop
        const int g = le >> 4, r16 = le & 15;
        const int wr_off = r16 * 128 + ((((g & 1) ^ (r16 >> 3)) & 1) << 3), wr_x = r16 & 7;
        const int rd_row = le >> 3, rd_ch = le & 7;
        const int rd_off = rd_row * 128 + ((rd_ch ^ (rd_row & 7)) << 4);
        const int64_t m0 = tile_m0(wC), n0 = tile_n0(wC);
        const int64_t rows_left = p.M - m0 - wr0;
        const int hi4 = rd_ch >> 2;
        int64_t cbase, cbase2 = 0;
        bool col_ok;
        if constexpr (KIND == MI355_EPI_SWIGLU_FWD) {
            const int64_t nh = p.N >> 1, hid = (n0 >> 1) + (wc0 >> 6) * 32 + (rd_ch & 3) * 8;
            col_ok = hid < nh;
            cbase = (hi4 ? nh : 0) + hid;
            cbase2 = hid + 4 * hi4;
        } else {
            cbase = n0 + wc0 + rd_ch * 8;
            col_ok = cbase < p.N;
        }
        auto rsrc_c = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<bf16_t*>(p.C) + (m0 + wr0) * p.ldc, 0, 0x7fffffff, 0x00020000);
        auto rsrc_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(reinterpret_cast<const bf16_t*>(p.R)) + (m0 + wr0) * p.ldr, 0, 0x7fffffff, 0x00020000);
        u32x4 rd[4];
        [[maybe_unused]] u32x2 rres[2][T::FN];
        [[maybe_unused]] u32x4 uv[4], gv[4];
        [[maybe_unused]] u32x2 partner[4];
        if constexpr (RES) {
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int j = 0; j < T::FN; ++j) {
                    const int row = sb * 32 + ii * 16 + r16;
                    const int64_t col = n0 + wc0 + j * 16 + 4 * g;
                    rres[ii][j] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc_r, (row < rows_left && col < p.N) ? (unsigned)((row * p.ldr + col) * 2) : OOB, 0, 0));
                }
        }
        if constexpr (KIND == MI355_EPI_SWIGLU_BWD || KIND == MI355_EPI_ATTN_DELTA) {
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const int row = sb * 32 + pass * 8 + rd_row;
                const bool ok = col_ok && row < rows_left;
                uv[pass] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_r, ok ? (unsigned)((row * p.ldr + cbase) * 2) : OOB, 0, 0));
                if constexpr (KIND == MI355_EPI_SWIGLU_BWD)
                    gv[pass] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_r, ok ? (unsigned)((row * p.ldr + cbase + p.N) * 2) : OOB, 0, 0));
            }
        }
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < T::FN; ++j) {
                f32x4 v = acc[sb * 2 + ii][j];
                if constexpr (RES) {
                    v[0] += __uint_as_float(rres[ii][j][0] << 16);
                    v[1] += __uint_as_float(rres[ii][j][0] & 0xffff0000u);
                    v[2] += __uint_as_float(rres[ii][j][1] << 16);
                    v[3] += __uint_as_float(rres[ii][j][1] & 0xffff0000u);
                }
                const u32x2 pk = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
                *reinterpret_cast<u32x2*>(stg + ii * 2048 + wr_off + (((2 * j + (g >> 1)) ^ wr_x) << 4)) = pk;
            }
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            rd[pass] = *reinterpret_cast<const u32x4*>(stg + pass * 1024 + rd_off);
            if constexpr (KIND == MI355_EPI_SWIGLU_FWD)
                partner[pass] = *reinterpret_cast<const u32x2*>(stg + pass * 1024 + rd_row * 128 + ((((rd_ch ^ 4) ^ (rd_row & 7))) << 4) + (((hi4 ^ pass) & 1) << 3));
        }
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const u32x4 o = (pass & 1) ? (u32x4){rd[pass][2], rd[pass][3], rd[pass][0], rd[pass][1]} : rd[pass];  // rows 8-15 of a 16-row tile were written with their halves exchanged
            const int row = sb * 32 + pass * 8 + rd_row;
            const bool ok = col_ok && row < rows_left && !(p.ablate & 2);  // (profiling: bit 1 = no stores)
            if constexpr (KIND == MI355_EPI_SWIGLU_BWD) {
                u32x4 o0, o1;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float du[2], dg[2];
#pragma unroll
                    for (int hlf = 0; hlf < 2; ++hlf) {
                        const float d_ = hlf ? __uint_as_float(o[e] & 0xffff0000u) : __uint_as_float(o[e] << 16);
                        const float u_ = hlf ? __uint_as_float(uv[pass][e] & 0xffff0000u) : __uint_as_float(uv[pass][e] << 16);
                        const float g_ = hlf ? __uint_as_float(gv[pass][e] & 0xffff0000u) : __uint_as_float(gv[pass][e] << 16);
                        swiglu_grads(d_, u_, g_, du[hlf], dg[hlf]);
                    }
                    o0[e] = pack_bf2(du[0], du[1]);
                    o1[e] = pack_bf2(dg[0], dg[1]);
                }
                __builtin_amdgcn_raw_buffer_store_b128(o0, rsrc_c, ok ? (unsigned)((row * p.ldc + cbase) * 2) : OOB, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(o1, rsrc_c, ok ? (unsigned)((row * p.ldc + cbase + p.N) * 2) : OOB, 0, 0);
            } else {
                __builtin_amdgcn_raw_buffer_store_b128(o, rsrc_c, ok ? (unsigned)((row * p.ldc + cbase) * 2) : OOB, 0, 0);
                if constexpr (KIND == MI355_EPI_ATTN_DELTA) {
                    float dsum = 0.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        dsum += __uint_as_float(o[e] << 16) * __uint_as_float(uv[pass][e] << 16) + __uint_as_float(o[e] & 0xffff0000u) * __uint_as_float(uv[pass][e] & 0xffff0000u);
                    dsum += __shfl_xor(dsum, 1, 64);
                    dsum += __shfl_xor(dsum, 2, 64);
                    dsum += __shfl_xor(dsum, 4, 64);
                    psum[sb][pass] = dsum;
                }
                if constexpr (KIND == MI355_EPI_SWIGLU_FWD) {
                    const u32x2 pv = partner[pass];
                    const unsigned own0 = hi4 ? o[2] : o[0], own1 = hi4 ? o[3] : o[1];
                    const unsigned uw[2] = {hi4 ? pv[0] : own0, hi4 ? pv[1] : own1}, gw[2] = {hi4 ? own0 : pv[0], hi4 ? own1 : pv[1]};
                    float a4[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float u_ = (e & 1) ? __uint_as_float(uw[e >> 1] & 0xffff0000u) : __uint_as_float(uw[e >> 1] << 16);
                        const float g_ = (e & 1) ? __uint_as_float(gw[e >> 1] & 0xffff0000u) : __uint_as_float(gw[e >> 1] << 16);
                        a4[e] = swiglu_act(u_, g_);
                    }
                    const u32x2 av = {pack_bf2(a4[0], a4[1]), pack_bf2(a4[2], a4[3])};
                    __builtin_amdgcn_raw_buffer_store_b64(av, rsrc_r, ok ? (unsigned)((row * p.ldr + cbase2) * 2) : OOB, 0, 0);
                }
            }
        }
        if constexpr (KIND == MI355_EPI_ATTN_DELTA) {
            if (sb == 3) {  // park the 128 row sums behind the last row reads (LDS operations of a wave execute in order); read by chunk 4, a slot barrier later
                float* part = reinterpret_cast<float*>(stg);
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                    for (int pass = 0; pass < 4; ++pass)
                        if (rd_ch == 0) part[s4 * 32 + pass * 8 + rd_row] = psum[s4][pass];
            }
        }
    };
    auto delta_combine = [&](const int cj) {
        // the row constants: a head's two halves sit in two neighbouring waves of the group, whose 128 row sums were parked in their staging regions in chunk 3
        // (behind their last row reads; a slot barrier ago).  One thread per (row, head) of the group's 128 rows x 2 heads.
        if constexpr (KIND == MI355_EPI_ATTN_DELTA) {
            const int64_t m0 = tile_m0(wC), n0 = tile_n0(wC);
            const int t_ = threadIdx.x & 255, row = t_ & 127, hsel = t_ >> 7;
            const float* p0 = reinterpret_cast<const float*>(smem + 2 * T::STAGE + ((wave >> 2) * 4 + 2 * hsel) * 4096);
            const float sum = p0[row] + p0[1024 + row];
            const int64_t gm = m0 + wr0 + row, head = (n0 >> 7) + hsel;
            if (gm < p.M && head < p.ad_Hq) {
                const int64_t bb = gm / p.ad_S, sq = gm - bb * p.ad_S;
                const int64_t di = (bb * p.ad_Hq + head) * p.ad_S + sq;
                p.ad_delta[di] = sum;
                p.ad_ndl[di] = -sum;
                p.ad_nl2[di] = -p.ad_lse[di] * 1.4426950408889634f;
            }
        }
    };
    auto other_slot = [&](auto chunk_c, const int cj, bool next_is_compute) __attribute__((always_inline)) {  // a write-out chunk (chunk >= 0) or nothing at all
        constexpr int chunk = decltype(chunk_c)::value;
        if (grpY && started) request();
        if constexpr (chunk >= 0 && chunk < 4) {
            if (!(p.ablate & 1)) epi_chunk(chunk_c, cj);  // (profiling: bit 0 = no write-out at all)
        }
        if constexpr (chunk == 4) delta_combine(cj);
        constexpr int STORES = (chunk < 0 || chunk >= 4) ? 0 : (KIND == MI355_EPI_SWIGLU_FWD || KIND == MI355_EPI_SWIGLU_BWD) ? 8 : 4;
        slot_end(next_is_compute, std::integral_constant<int, STORES>{});
    };

    walk_init(w1);
    walk_init(w2);
    walk_init(wC);
    plan1();
    if (!grpY) plan2();
    request();  // slots 0 and 1
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    request();
    if (!grpY) {
        loadB(b0, smem + T::A_BYTES, 0);
        loadA(aE, smem, 0, 0);
    }
    const int lead = grpY ? E : 0;
    using std::integral_constant;
    for (int i = 0; i < lead; ++i) other_slot(integral_constant<int, -1>{}, 0, i == lead - 1);
    for (int cj = 0; cj < J; ++cj) {
        for (int t = 0; t < NT; ++t) compute_slot(t == 0, t + 1 == NT);
        const bool more = cj + 1 < J;
        other_slot(integral_constant<int, 0>{}, cj, false);
        other_slot(integral_constant<int, 1>{}, cj, false);
        other_slot(integral_constant<int, 2>{}, cj, false);
        other_slot(integral_constant<int, 3>{}, cj, E == 4 && more);
        if constexpr (E == 5) other_slot(integral_constant<int, 4>{}, cj, more);
        walk_next(wC);
    }
    for (int i = 0; i < E - lead; ++i) other_slot(integral_constant<int, -1>{}, 0, false);
}

int launch_pp(GemmParams p, hipStream_t s) {
    using T = Cfg256;
    p.tiles_m = (int)((p.M + T::BM - 1) / T::BM);
    p.tiles_n = (int)((p.N + T::BN - 1) / T::BN);
    const int64_t tiles = (int64_t)p.tiles_m * p.tiles_n;
    MI355_REQUIRE(tiles < 0x7fffffffLL / 16, "mi355_gemm_bf16: grid too large");
    const dim3 grid(256), block(T::NTHREADS);  // one workgroup per CU: 8 XCDs x (4 column panels x 8 row streams); workgroups without a tile leave at once
    if (p.epilogue == MI355_EPI_SWIGLU_FWD) hipLaunchKernelGGL((gemm_nt_pp_kernel<MI355_EPI_SWIGLU_FWD, false>), grid, block, 0, s, p, (int)tiles);
    else if (p.epilogue == MI355_EPI_SWIGLU_BWD) hipLaunchKernelGGL((gemm_nt_pp_kernel<MI355_EPI_SWIGLU_BWD, false>), grid, block, 0, s, p, (int)tiles);
    else if (p.epilogue == MI355_EPI_ATTN_DELTA) hipLaunchKernelGGL((gemm_nt_pp_kernel<MI355_EPI_ATTN_DELTA, false>), grid, block, 0, s, p, (int)tiles);
    else if (p.R) hipLaunchKernelGGL((gemm_nt_pp_kernel<MI355_EPI_NONE, true>), grid, block, 0, s, p, (int)tiles);
    else hipLaunchKernelGGL((gemm_nt_pp_kernel<MI355_EPI_NONE, false>), grid, block, 0, s, p, (int)tiles);
    MI355_LAUNCH_CHECK("mi355_gemm_bf16(ping-pong)");
    return 0;
}
#endif  // part 6

// Several independent problems of one operand form in ONE launch (the four weight gradients of a transformer block:
// each alone has fewer output tiles than the chip has CUs, together they fill it).  The concatenated tile list is cut
// into one contiguous chunk per XCD, so the tiles that share operand panels stay behind one L2.
constexpr int MAX_GROUP = 8;
struct GroupTable {
    GemmParams g[MAX_GROUP];
    int start[MAX_GROUP + 1];  // first tile of each problem in the concatenated list
    int count;
};

template <class T, bool A_TR, bool B_TR, int OUT_DT>
__global__ __launch_bounds__(T::NTHREADS, T::MIN_WAVES) void gemm_grouped_kernel(GroupTable tbl) {
    __shared__ __attribute__((aligned(16))) char smem[T::SMEM];
    const int v = xcd_chunked(blockIdx.x, tbl.start[tbl.count]);
    int g = 0;
#pragma unroll
    for (int i = 1; i < MAX_GROUP; ++i)
        if (i < tbl.count && v >= tbl.start[i]) g = i;
    const GemmParams p = tbl.g[g];
    gemm_tile<T, A_TR, B_TR, OUT_DT>(p, v - tbl.start[g], 0, smem);
}

// C = sum_s slab[s] (+ R), converted to the output dtype.  4 columns per thread (N % 8 == 0).
template <int OUT_DT>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(int64_t M, int64_t N, int ksplit, const float* __restrict__ ws,
                                                            void* __restrict__ C, int64_t ldc, const void* __restrict__ R, int64_t ldr) {
    const int64_t nv = N >> 2, total = M * nv;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t m = i / nv, n = (i - m * nv) * 4;
        f32x4 acc = *reinterpret_cast<const f32x4*>(ws + m * N + n);
        for (int sidx = 1; sidx < ksplit; ++sidx) acc += *reinterpret_cast<const f32x4*>(ws + ((int64_t)sidx * M + m) * N + n);
        if constexpr (OUT_DT == MI355_DT_BF16) {
            bf16_t* c = reinterpret_cast<bf16_t*>(C) + m * ldc + n;
            if (R) {
                const bf16_t* r = reinterpret_cast<const bf16_t*>(R) + m * ldr + n;
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] += bf2f(r[e]);
            }
            *reinterpret_cast<u32x2*>(c) = (u32x2){pack_bf2(acc[0], acc[1]), pack_bf2(acc[2], acc[3])};
        } else {
            float* c = reinterpret_cast<float*>(C) + m * ldc + n;
            if (R) acc += *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(R) + m * ldr + n);
            *reinterpret_cast<f32x4*>(c) = acc;
        }
    }
}

// K-split heuristic: wgrad-like problems (few output tiles, very long K) leave most CUs idle.
int choose_ksplit(int64_t tiles, int64_t slots, int64_t K, int64_t M, int64_t N, int epilogue, const float* bias, int64_t ldc, int64_t ldr, int64_t ws_bytes) {
    if (bias || epilogue != MI355_EPI_NONE || (N & 7) || (ldc & 3) || (ldr & 3)) return 1;
    const int64_t kt = (K + 63) / 64;
    if (tiles > slots / 2 || kt < 32) return 1;  // over half the resident slots already: the reduce pass costs more than it buys
    int64_t ks = (slots + slots / 4 + tiles - 1) / tiles;
    if (ks > kt / 8) ks = kt / 8;
    if (ks > 16) ks = 16;
    while (ks > 1 && ks * M * N * 4 > ws_bytes) --ks;
    return ks < 2 ? 1 : (int)ks;
}

template <class T, bool A_TR, bool B_TR>
int launch(GemmParams p, int out_dtype, void* workspace, int64_t workspace_bytes, hipStream_t s) {
    p.tiles_m = (int)((p.M + T::BM - 1) / T::BM);
    p.tiles_n = (int)((p.N + T::BN - 1) / T::BN);
    const int64_t tiles = (int64_t)p.tiles_m * p.tiles_n;
    MI355_REQUIRE(tiles < 0x7fffffffLL / 16, "mi355_gemm_bf16: grid too large");
    constexpr int WG_PER_CU = (T::MIN_WAVES * 256 / T::NTHREADS) > 0 ? (T::MIN_WAVES * 256 / T::NTHREADS) : 1;
    const int64_t slots = 256 * WG_PER_CU;  // workgroups resident on the chip at once
    p.ws = (float*)workspace;
    p.ksplit = workspace ? choose_ksplit(tiles, slots, p.K, p.M, p.N, p.epilogue, p.bias, p.ldc, p.R ? p.ldr : 0, workspace_bytes) : 1;
    const int grid = (int)(tiles * p.ksplit);
    if (p.ksplit > 1) {
        hipLaunchKernelGGL((gemm_bf16_kernel<T, A_TR, B_TR, MI355_DT_F32>), dim3(grid), dim3(T::NTHREADS), 0, s, p);
        const int64_t work = p.M * (p.N >> 2);
        const int rgrid = (int)((work + 255) / 256 > 2048 ? 2048 : (work + 255) / 256);
        if (out_dtype == MI355_DT_BF16)
            hipLaunchKernelGGL(splitk_reduce_kernel<MI355_DT_BF16>, dim3(rgrid), dim3(256), 0, s, p.M, p.N, p.ksplit, p.ws, p.C, p.ldc, p.R, p.ldr);
        else
            hipLaunchKernelGGL(splitk_reduce_kernel<MI355_DT_F32>, dim3(rgrid), dim3(256), 0, s, p.M, p.N, p.ksplit, p.ws, p.C, p.ldc, p.R, p.ldr);
        MI355_LAUNCH_CHECK("mi355_gemm_bf16(split-K)");
        return 0;
    }
    if (out_dtype == MI355_DT_BF16)
        hipLaunchKernelGGL((gemm_bf16_kernel<T, A_TR, B_TR, MI355_DT_BF16>), dim3(grid), dim3(T::NTHREADS), 0, s, p);
    else
        hipLaunchKernelGGL((gemm_bf16_kernel<T, A_TR, B_TR, MI355_DT_F32>), dim3(grid), dim3(T::NTHREADS), 0, s, p);
    MI355_LAUNCH_CHECK("mi355_gemm_bf16");
    return 0;
}

template <class T>
int launch_form(int form, const GemmParams& p, int out_dtype, void* ws, int64_t ws_bytes, hipStream_t s) {
    switch (form) {
        case MI355_GEMM_NT: return launch<T, false, false>(p, out_dtype, ws, ws_bytes, s);
        case MI355_GEMM_NN: return launch<T, false, true>(p, out_dtype, ws, ws_bytes, s);
        default: return launch<T, true, true>(p, out_dtype, ws, ws_bytes, s);
    }
}

template <class T, bool A_TR, bool B_TR>
int launch_grouped(GroupTable& tbl, int out_dtype, hipStream_t s) {
    int64_t total = 0;
    for (int i = 0; i < tbl.count; ++i) {
        GemmParams& p = tbl.g[i];
        p.tiles_m = (int)((p.M + T::BM - 1) / T::BM);
        p.tiles_n = (int)((p.N + T::BN - 1) / T::BN);
        tbl.start[i] = (int)total;
        total += (int64_t)p.tiles_m * p.tiles_n;
        MI355_REQUIRE(total < 0x7fffffffLL / 16, "mi355_gemm_bf16_grouped: grid too large");
    }
    for (int i = tbl.count; i <= MAX_GROUP; ++i) tbl.start[i] = (int)total;
    if (out_dtype == MI355_DT_BF16)
        hipLaunchKernelGGL((gemm_grouped_kernel<T, A_TR, B_TR, MI355_DT_BF16>), dim3((unsigned)total), dim3(T::NTHREADS), 0, s, tbl);
    else
        hipLaunchKernelGGL((gemm_grouped_kernel<T, A_TR, B_TR, MI355_DT_F32>), dim3((unsigned)total), dim3(T::NTHREADS), 0, s, tbl);
    MI355_LAUNCH_CHECK("mi355_gemm_bf16_grouped");
    return 0;
}

template <class T>
int launch_grouped_form(int form, GroupTable& tbl, int out_dtype, hipStream_t s) {
    switch (form) {
        case MI355_GEMM_NT: return launch_grouped<T, false, false>(tbl, out_dtype, s);
        case MI355_GEMM_NN: return launch_grouped<T, false, true>(tbl, out_dtype, s);
        default: return launch_grouped<T, true, true>(tbl, out_dtype, s);
    }
}

int check_operands(const char* who, int form, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B, int64_t ldb, const void* C) {
    MI355_REQUIRE(M > 0 && N > 0 && K > 0, "%s: empty problem M=%ld N=%ld K=%ld", who, (long)M, (long)N, (long)K);
    MI355_REQUIRE(A && B && C, "%s: null operand", who);
    MI355_REQUIRE((lda & 7) == 0 && (ldb & 7) == 0, "%s: lda/ldb must be multiples of 8 (16-byte rows)", who);
    MI355_REQUIRE(((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0, "%s: A/B must be 16-byte aligned", who);
    if (form == MI355_GEMM_NT) MI355_REQUIRE((K & 7) == 0, "%s(NT): K must be a multiple of 8", who);
    if (form == MI355_GEMM_NN) MI355_REQUIRE((K & 7) == 0 && (N & 7) == 0, "%s(NN): K,N must be multiples of 8", who);
    if (form == MI355_GEMM_TN) MI355_REQUIRE((M & 7) == 0 && (N & 7) == 0, "%s(TN): M,N must be multiples of 8", who);
    // a tile's DMA offsets are 31-bit: 256 rows (or 64 k-rows) of one operand must span < 2 GiB
    MI355_REQUIRE(lda * 2 * 256 < 0x7fffffffLL && ldb * 2 * 256 < 0x7fffffffLL, "%s: leading dimension too large", who);
    return 0;
}

// ---------------------------------------------------------------------------------------------- colsum
// Column sums (bias gradients): a workgroup owns 512 columns x `rows_per_block` rows; a thread keeps 8 adjacent columns, a wave reads one kilobyte
// of a row per instruction (16-byte loads, whole lines), the four waves take rows r, r+1, r+2, r+3; partial sums meet in LDS and leave as one
// fp32 atomic per column and workgroup.  (Round 4: the first version read one 2-byte element per thread -- 128-byte row segments -- and ran at
// 1.8 TB/s: 4.7 ms of the 53-ms ViT-B/16 training step.)
template <int DT>
__global__ __launch_bounds__(256) void colsum_kernel(int64_t M, int64_t N, const void* Xv, int64_t ldx, float* out, int rows_per_block) {
    __shared__ float red[4][512];
    const int chunk = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int64_t col = (int64_t)blockIdx.x * 512 + chunk * 8;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    const int64_t r1 = min(M, r0 + rows_per_block);
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bool vec = col + 8 <= N && (ldx & 7) == 0 && ((uintptr_t)Xv & 15) == 0;
    if (col < N) {
        for (int64_t r = r0 + rl; r < r1; r += 4) {
            if constexpr (DT == MI355_DT_BF16) {
                const bf16_t* px = reinterpret_cast<const bf16_t*>(Xv) + r * ldx + col;
                if (vec) {
                    const u32x4 v = *reinterpret_cast<const u32x4*>(px);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        s[2 * e] += __uint_as_float(v[e] << 16);
                        s[2 * e + 1] += __uint_as_float(v[e] & 0xffff0000u);
                    }
                } else {
                    for (int e = 0; e < 8 && col + e < N; ++e) s[e] += bf2f(px[e]);
                }
            } else {
                const float* px = reinterpret_cast<const float*>(Xv) + r * ldx + col;
                if (vec) {
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(px), v1 = *reinterpret_cast<const f32x4*>(px + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        s[e] += v0[e];
                        s[4 + e] += v1[e];
                    }
                } else {
                    for (int e = 0; e < 8 && col + e < N; ++e) s[e] += px[e];
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[rl][chunk * 8 + e] = s[e];
    __syncthreads();
    for (int c = threadIdx.x; c < 512; c += 256) {
        const int64_t gc = (int64_t)blockIdx.x * 512 + c;
        if (gc < N) atomicAdd(out + gc, red[0][c] + red[1][c] + red[2][c] + red[3][c]);
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------- translation-unit split
// This file is compiled once per tile configuration (-DGEMM_PART=1..5: that configuration's kernels behind two bridge functions) and once
// for the entry points (-DGEMM_PART=0), so the objects build in parallel; parameter blocks cross the bridge as untyped pointers (every part
// is this same source, so the layouts agree).  Without -DGEMM_PART everything lands in one object.
#ifndef GEMM_PART
#define GEMM_PART -1
#endif
#define GEMM_BRIDGE_DECL(N)                                                                                                                       \
    extern "C" __attribute__((visibility("hidden"))) int mi355_gemm_part##N(int form, const void* params, int out_dtype, void* ws, int64_t ws_bytes, void* stream); \
    extern "C" __attribute__((visibility("hidden"))) int mi355_gemm_grouped_part##N(int form, void* table, int out_dtype, void* stream);
#define GEMM_BRIDGE_DEF(N, CFG)                                                                                                                   \
    extern "C" int mi355_gemm_part##N(int form, const void* params, int out_dtype, void* ws, int64_t ws_bytes, void* stream) {                     \
        return launch_form<CFG>(form, *static_cast<const GemmParams*>(params), out_dtype, ws, ws_bytes, (hipStream_t)stream);                      \
    }                                                                                                                                             \
    extern "C" int mi355_gemm_grouped_part##N(int form, void* table, int out_dtype, void* stream) {                                                \
        return launch_grouped_form<CFG>(form, *static_cast<GroupTable*>(table), out_dtype, (hipStream_t)stream);                                   \
    }
GEMM_BRIDGE_DECL(1) GEMM_BRIDGE_DECL(2) GEMM_BRIDGE_DECL(3) GEMM_BRIDGE_DECL(4) GEMM_BRIDGE_DECL(5)
extern "C" __attribute__((visibility("hidden"))) int mi355_gemm_persist_part2(const void* params, void* stream);
extern "C" __attribute__((visibility("hidden"))) int mi355_gemm_pp_part6(const void* params, void* stream);
#if GEMM_PART == 6 || GEMM_PART == -1
extern "C" int mi355_gemm_pp_part6(const void* params, void* stream) { return launch_pp(*static_cast<const GemmParams*>(params), (hipStream_t)stream); }
#endif
#if GEMM_PART == 2 || GEMM_PART == -1
extern "C" int mi355_gemm_persist_part2(const void* params, void* stream) { return launch_persist(*static_cast<const GemmParams*>(params), (hipStream_t)stream); }
#endif
#if GEMM_PART == 1 || GEMM_PART == -1
GEMM_BRIDGE_DEF(1, Cfg128)
#endif
#if GEMM_PART == 2 || GEMM_PART == -1
GEMM_BRIDGE_DEF(2, Cfg256)
#endif
#if GEMM_PART == 3 || GEMM_PART == -1
GEMM_BRIDGE_DEF(3, Cfg256a)
#endif
#if GEMM_PART == 4 || GEMM_PART == -1
GEMM_BRIDGE_DEF(4, Cfg256b)
#endif
#if GEMM_PART == 5 || GEMM_PART == -1
GEMM_BRIDGE_DEF(5, Cfg256w)
#endif

#if GEMM_TL
extern "C" int mi355_debug_gemm_tl(unsigned long long* out, int n) {  // out[n][8]
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    if (n > 32768 || hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gemm_tl), sizeof(unsigned long long) * 8 * n) != hipSuccess) return 2;
    return 0;
}
#endif
#if GEMM_PART <= 0
#if GEMM_PROF
extern "C" int mi355_debug_gemm_prof(unsigned long long* out, int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gemm_prof), sizeof(unsigned long long) * 32) != hipSuccess) return 2;
    if (reset) {
        unsigned long long z[32] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_prof), z, sizeof(z)) != hipSuccess) return 3;
    }
    return 0;
}
#endif

// tiles from which an eligible NT launch takes the persistent kernel by itself (MI355_GEMM_PERSIST_MIN_TILES: 1 = always -- the tests run ragged shapes through it --,
// a huge value = never: A/B runs); read per call, the tests change it inside one process
static int64_t persist_min_tiles() {
    const char* e = getenv("MI355_GEMM_PERSIST_MIN_TILES");
    return e && *e ? atoll(e) : 512;
}

// which epilogue kinds of the persistent NT kernel take the ping-pong form (tile hint 8) by themselves: a bit mask, 1 plain, 2 + residual, 4 SwiGLU forward,
// 8 SwiGLU backward, 16 attention delta (MI355_GEMM_PP; read per call: A/B runs and tests change it inside one process)
static bool walk_on(int64_t N, int64_t K) {
    // MI355_GEMM_WALK: the persistent NT kernel's weight-stationary walk (gemm_nt_persist_kernel<.., WALK>) for launches the library chooses by itself: 0 never, 1 always,
    // 2 (default) by shape: few column panels under a long K (the dgrads into d = 1024, the out / down projections), or very many column panels (the LM head).  Round 6,
    // batch 160, same-box bench.py runs: never 443.3 / 443.6 / 444.7 / 445.9 ms, by shape 438.5 / 438.6 / 441.4-441.8, always 441.6 / 440.6; per shape class (16 + mask):
    // N <= 1024 with K >= 3072 -3.7 ms, + K = 2048 -0.2, the LM head 0 to -1.2, N = 2048 0, N = 3072 / 4096 / 6144 (K = 1024) +1.0 to +1.9 ms each.  Read per call.
    const char* e = getenv("MI355_GEMM_WALK");
    const int mode = e && *e ? atoi(e) : MI355_GEMM_WALK_DEFAULT;
    if (mode >= 16) {  // exploration: 16 + a mask of shape classes
        const int m = mode - 16;
        return ((m & 1) && N <= 1024 && K >= 3072) || ((m & 2) && N >= 64 * 256) || ((m & 4) && N <= 1024 && K == 2048) || ((m & 8) && N == 2048) || ((m & 16) && N == 3072) ||
               ((m & 32) && N == 4096) || ((m & 64) && N == 6144);
    }
    return mode == 1 || (mode == 2 && ((N <= 1024 && K >= 2048) || N >= 64 * 256));
}
static int pp_mask() {
    const char* e = getenv("MI355_GEMM_PP");
    return e && *e ? atoi(e) : MI355_GEMM_PP_DEFAULT;
}

extern "C" int mi355_gemm_bf16(int form, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B,
                               int64_t ldb, void* C, int64_t ldc, int out_dtype, const float* bias,
                               const void* residual, int64_t ldr, int epilogue, void* workspace, int64_t workspace_bytes,
                               int tile_hint, void* stream) {
    MI355_REQUIRE(form >= 0 && form <= 2, "mi355_gemm_bf16: bad form %d", form);
    const bool split3 = out_dtype == MI355_DT_SPLIT3;
    if (split3) {
        MI355_REQUIRE(form == MI355_GEMM_NT && (epilogue == MI355_EPI_NONE || epilogue == MI355_EPI_GELU_ERF) && !residual && (N & 7) == 0 && ldc >= 3 * N && (ldc & 7) == 0 &&
                          ((uintptr_t)C & 15) == 0,
                      "mi355_gemm_bf16(split3 output): NT form, plain / GELU epilogue, no residual, N %% 8 == 0, C bf16 [M, 3N] with 16-byte aligned rows");
        MI355_REQUIRE((tile_hint & 0xff) != 5, "mi355_gemm_bf16(split3 output): not on tile hint 5");
        out_dtype = MI355_DT_F32;  // the fp32-output kernels carry it (no split-K: the slabs' reduce kernel writes fp32)
        workspace = nullptr;
        workspace_bytes = 0;
    }
    MI355_REQUIRE(out_dtype == MI355_DT_BF16 || out_dtype == MI355_DT_F32, "mi355_gemm_bf16: bad out_dtype");
    if (int rc = check_operands("mi355_gemm_bf16", form, M, N, K, A, lda, B, ldb, C)) return rc;
    MI355_REQUIRE(workspace == nullptr || ((uintptr_t)workspace & 15) == 0, "mi355_gemm_bf16: workspace must be 16-byte aligned");
    MI355_REQUIRE(epilogue == MI355_EPI_NONE || epilogue == MI355_EPI_GELU_ERF || epilogue == MI355_EPI_SWIGLU_BWD || epilogue == MI355_EPI_SWIGLU_FWD || (epilogue >= MI355_EPI_GELU_DUAL_ERF && epilogue <= MI355_EPI_GELU_BWD_TANH), "mi355_gemm_bf16: unknown epilogue %d", epilogue);
    if (epilogue >= MI355_EPI_GELU_DUAL_ERF && epilogue <= MI355_EPI_GELU_BWD_TANH)
        MI355_REQUIRE(out_dtype == MI355_DT_BF16 && residual && (N & 7) == 0 && (ldc & 7) == 0 && (ldr & 7) == 0 && ldr >= N &&
                          (((uintptr_t)residual | (uintptr_t)C) & 15) == 0 && (epilogue <= MI355_EPI_GELU_DUAL_TANH || !bias),
                      "mi355_gemm_bf16(GELU dual / backward epilogue): bf16 output, N %% 8 == 0, `residual` = the second [M, N] operand (activation output, resp. "
                      "the forward's pre-activation), no bias in the backward form");
    if (epilogue == MI355_EPI_SWIGLU_FWD)
        MI355_REQUIRE(form == MI355_GEMM_NT && out_dtype == MI355_DT_BF16 && residual && !bias && (N & 63) == 0 && ldc >= N && ldr >= N / 2 && (ldc & 7) == 0 &&
                          (ldr & 7) == 0 && (((uintptr_t)residual | (uintptr_t)C) & 15) == 0,
                      "mi355_gemm_bf16(SwiGLU forward epilogue): NT form, bf16 gate-up output [M, N] with N = 2F, F %% 32 == 0, `residual` = the activation output [M, F], no bias");
    if (epilogue == MI355_EPI_SWIGLU_BWD)
        MI355_REQUIRE(out_dtype == MI355_DT_BF16 && residual && !bias && (N & 7) == 0 && ldc >= 2 * N && ldr >= 2 * N && (ldc & 7) == 0 && (ldr & 7) == 0 &&
                          (((uintptr_t)residual | (uintptr_t)C) & 15) == 0,
                      "mi355_gemm_bf16(SwiGLU backward epilogue): bf16 output [M, 2N] (ldc >= 2N), residual = the forward gate-up output [M, 2N], N %% 8 == 0, no bias");
    const int ablate = tile_hint >> 8;
    tile_hint &= 0xff;
    MI355_REQUIRE((tile_hint >= 0 && tile_hint <= 5) || tile_hint == 7 || tile_hint == 8, "mi355_gemm_bf16: tile_hint must be 0 (auto), 1 (128x128), 2 (256x256), 3 (256x256, alternating wave groups), 4 (3 with one barrier per phase), 5 (256x256, four waves of 128x128), 7 (2 as a persistent workgroup per CU) or 8 (7 with the two wave groups a write-out apart)");
    GemmParams p;
    p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = C; p.bias = bias; p.R = residual;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldr = ldr;
    p.epilogue = epilogue; p.tiles_m = p.tiles_n = 0; p.ksplit = 1; p.ws = nullptr; p.ablate = ablate;
    p.ad_lse = nullptr; p.ad_delta = p.ad_nl2 = p.ad_ndl = nullptr; p.ad_S = p.ad_Hq = 0;
    p.split3 = split3;
    hipStream_t s = (hipStream_t)stream;
    int cfg = tile_hint;
    if (cfg == 0) {
        // measured on MI355X over the VLM step's shapes (tools/gemm_sweep.py): the alternating-group 256x256 kernel wins for
        // forward / dgrad shapes and for the large weight gradients; small weight gradients (few tiles, K = tokens) do
        // better on 128x128 tiles with split-K; tiny problems stay on 128x128.
        // Tile 4 (one barrier per phase, 5 stages) wins every isolated sweep (+5-14 % NT, +10-13 % TN on warm operands) but
        // LOSES inside the training step, where operands arrive cold from the previous kernel: A/B in one process,
        // per form, 246.2 ms/step with tile 3 everywhere vs +0.9 (NT) / +4.3 (NN) / +0.5 (TN) ms with tile 4.  The step decides.
        // The same A/B puts the forward projections (NT) on tile 2 (BK 64, two stages, fragments prefetched one phase ahead):
        // 239.3 vs 243.8-246 ms/step with NT on tile 3, although tile 3 is 10-20 % faster on every isolated NT shape.
        // 256x256 tiles need at least half of the 256 CUs' worth of tiles; below that (small token counts: config 5 at B = 8 has 92 tiles for
        // its N = 1024 outputs) four times as many 128x128 tiles, two workgroups per CU, fill the chip instead
        const int64_t tiles256 = ((M + 255) / 256) * ((N + 255) / 256);
        if (M < 256 || N < 256) cfg = 1;
        else if (form != MI355_GEMM_TN && tiles256 < 128) cfg = 1;  // measured: 92 tiles -> 128x128 wins (+2.8 % on the config-5 step), 180 tiles -> 256x256 wins
        // weight gradients: the 4-wave loop (128x128 per wave: a third fewer transposing reads per MFMA than the 8-wave tiles, every read issued from asm).  Round 4, batch 160:
        // the block's grouped launch 3 327 -> 3 028 us, the LM head's 25.9 -> 22.7 ms in isolation; in the step 469.0 / 468.9 -> 456.9 / 458.0 ms (same box, bit-identical)
        else if (form == MI355_GEMM_TN) cfg = (M * N > 5 * 1024 * 1024) ? 5 : 1;
        else if (form == MI355_GEMM_NT) cfg = 2;
        else cfg = 3;
    }
    // the persistent form of tile 2 (hint 7): same bits, no prologue / barrier / workgroup turnover per tile and a quarter of the staging traffic.  Round 4, batch 160, isolated:
    // QKV 890-917 -> 818-825 us, gate-up + SwiGLU 1 489-1 494 -> 1 418-1 422, down dgrad + SwiGLU backward 959-986 -> 920, dqkv dgrad 788-818 -> 756-764.  Chosen by itself from two
    // rounds of tiles upward (below that every workgroup has one tile and split-K may pay instead)
    const bool persist_ok = form == MI355_GEMM_NT && out_dtype == MI355_DT_BF16 && (epilogue == MI355_EPI_NONE || epilogue == MI355_EPI_SWIGLU_FWD || epilogue == MI355_EPI_SWIGLU_BWD) &&
                            !bias && (K & 63) == 0 && K >= 128 && (N & 7) == 0 && (ldc & 7) == 0 && ((uintptr_t)C & 15) == 0 && M >= 256 && N >= 256 &&
                            (!residual || ((ldr & 7) == 0 && ((uintptr_t)residual & 15) == 0)) && ldc * 2 * 256 < 0x7fffffffLL && ldr * 2 * 256 < 0x7fffffffLL;
    if (cfg == 2 && tile_hint == 0 && persist_ok && ((M + 255) / 256) * ((N + 255) / 256) >= persist_min_tiles()) {
        const int kind_bit = epilogue == MI355_EPI_SWIGLU_FWD ? 4 : epilogue == MI355_EPI_SWIGLU_BWD ? 8 : residual ? 2 : 1;
        cfg = (pp_mask() & kind_bit) ? 8 : 7;
        if (walk_on(N, K)) p.ablate |= 8;
    }
    if (cfg == 8) {  // ping-pong form: needs at least as many K-tiles as write-out slots
        if (persist_ok && K >= 64 * 5) return mi355_gemm_pp_part6(&p, s);
        cfg = 7;
    }
    if (cfg == 7) {
        const bool ok = persist_ok;
        if (ok) return mi355_gemm_persist_part2(&p, s);
        cfg = 2;
    }
    switch (cfg) {
        case 2: return mi355_gemm_part2(form, &p, out_dtype, workspace, workspace_bytes, s);
        case 3: return mi355_gemm_part3(form, &p, out_dtype, workspace, workspace_bytes, s);
        case 4: return mi355_gemm_part4(form, &p, out_dtype, workspace, workspace_bytes, s);
        case 5: return mi355_gemm_part5(form, &p, out_dtype, workspace, workspace_bytes, s);
        default: return mi355_gemm_part1(form, &p, out_dtype, workspace, workspace_bytes, s);
    }
}

extern "C" int mi355_gemm_bf16_attn_delta(int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                                          const void* ctx, int64_t ldctx, int S, int Hq, int D, const float* lse, float* delta, float* neg_lse_log2e,
                                          float* neg_delta, void* stream) {
    if (int rc = check_operands("mi355_gemm_bf16_attn_delta", MI355_GEMM_NT, M, N, K, A, lda, B, ldb, C)) return rc;
    MI355_REQUIRE(D == 128 && S > 0 && Hq > 0 && N == (int64_t)Hq * D && M % S == 0, "mi355_gemm_bf16_attn_delta: C is d(ctx) [B*S, Hq*128] (got N = %ld, Hq = %d, D = %d, M = %ld, S = %d)",
                  (long)N, Hq, D, (long)M, S);
    MI355_REQUIRE(M >= 256, "mi355_gemm_bf16_attn_delta: at least 256 token rows (256 x 256 tiles)");
    MI355_REQUIRE(ctx && lse && delta && neg_lse_log2e && neg_delta, "mi355_gemm_bf16_attn_delta: null pointer");
    MI355_REQUIRE((ldc & 7) == 0 && (ldctx & 7) == 0 && ldc >= N && ldctx >= N && (((uintptr_t)ctx | (uintptr_t)C) & 15) == 0,
                  "mi355_gemm_bf16_attn_delta: ctx / C rows must be 16-byte aligned and hold N columns");
    GemmParams p;
    p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = C; p.bias = nullptr; p.R = ctx;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldr = ldctx;
    p.epilogue = MI355_EPI_ATTN_DELTA; p.tiles_m = p.tiles_n = 0; p.ksplit = 1; p.ws = nullptr; p.ablate = 0;
    p.ad_lse = lse; p.ad_delta = delta; p.ad_nl2 = neg_lse_log2e; p.ad_ndl = neg_delta; p.ad_S = S; p.ad_Hq = Hq;
    p.split3 = 0;
    // the persistent form (same bits) from two rounds of tiles upward
    if ((K & 63) == 0 && K >= 128 && ((M + 255) / 256) * ((N + 255) / 256) >= persist_min_tiles() && ldc * 2 * 256 < 0x7fffffffLL && ldctx * 2 * 256 < 0x7fffffffLL) {
        if ((pp_mask() & 16) && K >= 64 * 5) return mi355_gemm_pp_part6(&p, (hipStream_t)stream);
        if (walk_on(N, K)) p.ablate |= 8;
        return mi355_gemm_persist_part2(&p, (hipStream_t)stream);
    }
    return mi355_gemm_part2(MI355_GEMM_NT, &p, MI355_DT_BF16, nullptr, 0, (hipStream_t)stream);
}

extern "C" int mi355_gemm_bf16_grouped(int form, int count, const mi355_gemm_problem* problems, int out_dtype, int tile_hint,
                                       void* stream) {
    MI355_REQUIRE(form >= 0 && form <= 2, "mi355_gemm_bf16_grouped: bad form %d", form);
    MI355_REQUIRE(count >= 1 && count <= MAX_GROUP && problems, "mi355_gemm_bf16_grouped: count must be 1..%d", MAX_GROUP);
    MI355_REQUIRE(out_dtype == MI355_DT_BF16 || out_dtype == MI355_DT_F32, "mi355_gemm_bf16_grouped: bad out_dtype");
    MI355_REQUIRE(tile_hint == 0 || tile_hint == 1 || tile_hint == 3 || tile_hint == 4 || tile_hint == 5, "mi355_gemm_bf16_grouped: tile_hint must be 0 (auto), 1 (128x128), 3, 4 or 5 (256x256)");
    GroupTable tbl;
    tbl.count = count;
    int64_t tiles256 = 0;
    bool small = false;
    for (int i = 0; i < count; ++i) {
        const mi355_gemm_problem& q = problems[i];
        if (int rc = check_operands("mi355_gemm_bf16_grouped", form, q.M, q.N, q.K, q.A, q.lda, q.B, q.ldb, q.C)) return rc;
        GemmParams& p = tbl.g[i];
        p.A = (const bf16_t*)q.A; p.B = (const bf16_t*)q.B; p.C = q.C; p.bias = nullptr; p.R = q.residual;
        p.M = q.M; p.N = q.N; p.K = q.K; p.lda = q.lda; p.ldb = q.ldb; p.ldc = q.ldc; p.ldr = q.ldr;
        p.epilogue = MI355_EPI_NONE; p.tiles_m = p.tiles_n = 0; p.ksplit = 1; p.ws = nullptr; p.ablate = 0;
        p.ad_lse = nullptr; p.ad_delta = p.ad_nl2 = p.ad_ndl = nullptr; p.ad_S = p.ad_Hq = 0;
        p.split3 = 0;
        tiles256 += ((q.M + 255) / 256) * ((q.N + 255) / 256);
        small |= q.M < 256 || q.N < 256;
    }
    for (int i = count; i < MAX_GROUP; ++i) tbl.g[i] = tbl.g[0];
    // 256x256 tiles (one workgroup per CU) once they cover most of the chip; otherwise 128x128 (two per CU, 4x the tiles)
    const int cfg = tile_hint ? tile_hint : ((small || tiles256 < 160) ? 1 : (form == MI355_GEMM_TN ? 5 : 3));
    hipStream_t s = (hipStream_t)stream;
    if (cfg == 5) return mi355_gemm_grouped_part5(form, &tbl, out_dtype, s);
    if (cfg == 4) return mi355_gemm_grouped_part4(form, &tbl, out_dtype, s);
    if (cfg == 3) return mi355_gemm_grouped_part3(form, &tbl, out_dtype, s);
    return mi355_gemm_grouped_part1(form, &tbl, out_dtype, s);
}

extern "C" int mi355_colsum(int64_t M, int64_t N, const void* X, int x_dtype, int64_t ldx, float* out, int accumulate,
                            void* stream) {
    MI355_REQUIRE(M > 0 && N > 0 && X && out, "mi355_colsum: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (!accumulate) {
        if (hipMemsetAsync(out, 0, N * sizeof(float), s) != hipSuccess) {
            mi355_set_error("mi355_colsum: memset failed");
            return 2;
        }
    }
    const int64_t col_blocks = (N + 511) / 512;
    int rows_per_block = 512;  // enough workgroups to fill the chip a few times over, at least 32 rows each
    while (rows_per_block > 32 && col_blocks * ((M + rows_per_block - 1) / rows_per_block) < 2048) rows_per_block >>= 1;
    dim3 grid((unsigned)col_blocks, (unsigned)((M + rows_per_block - 1) / rows_per_block));
    if (x_dtype == MI355_DT_BF16)
        hipLaunchKernelGGL(colsum_kernel<MI355_DT_BF16>, grid, dim3(256), 0, s, M, N, X, ldx, out, rows_per_block);
    else
        hipLaunchKernelGGL(colsum_kernel<MI355_DT_F32>, grid, dim3(256), 0, s, M, N, X, ldx, out, rows_per_block);
    MI355_LAUNCH_CHECK("mi355_colsum");
    return 0;
}
#endif  // GEMM_PART <= 0
