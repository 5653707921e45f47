// Shared device helpers of the attention kernels (attention.hip; tools/experimental/attention_fwd2.hip): tile configuration, LDS images and their swizzles,
// LDS-DMA tile copies, fragment reads, and the statements that own registers of the accumulator file by name.  gfx950 only.
#pragma once
#include <type_traits>

#include "common.h"

namespace {

constexpr float MASK_T = -2.0e38f;  // finite "masked" score in the log2 domain
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
constexpr unsigned OOB = 0x80000000u;
constexpr int ATTN_MAX_TILES = 512;  // key-mask words kept in LDS by the backward dQ pass: S <= 64 * 512

template <int D>
struct Cfg {
    static constexpr int ROWB = D * 2;            // bytes per tile row
    static constexpr int CH = ROWB / 16;          // 16-byte chunks per row
    static constexpr int KS = D / 16;             // k-steps of the QK^T product
    static constexpr int DT = D / 32;             // 32-row tiles of O^T
    static constexpr int TILE = 64 * ROWB;        // bytes of one 64-key tile
    static constexpr int RPP = 1024 / ROWB;       // rows per 1-KiB DMA piece
    static constexpr int PPW = TILE / 1024 / 4;   // pieces per wave per tile
    // D = 128: ONE image serves both the row reads and the transposing reads of the backward kernels (half the LDS-DMA)
    static constexpr bool UNI = (D == 128);
};

// swizzles (chunk index XOR) -- row-read image (32x32 A-operand pattern) and transposed-read image
template <int D> __device__ __forceinline__ int swz_row(int chunk, int row) { return D == 128 ? chunk ^ (row & 15) : chunk ^ ((row >> 1) & 7); }
template <int D> __device__ __forceinline__ int swz_tr(int chunk, int row) { return D == 128 ? chunk ^ ((row & 3) << 2) : chunk ^ (((row >> 1) & 1) << 2); }

// unified image (256-byte rows = one bank row, 16 chunks): chunk ^ f(row) with f = the two 2-bit fields of row&15 swapped.
// f is a bijection over any 16 aligned rows (ds_read_b128 of 16 lanes = 16 rows at one chunk: 16 distinct slots), and its
// high field follows row&3 (a transposing read's 16 lanes = 4 consecutive rows x 2 adjacent chunks x 2 halves: 8 distinct slots).
__device__ __forceinline__ int swz_uni(int chunk, int row) { return chunk ^ (((row & 3) << 2) | ((row >> 2) & 3)); }
constexpr int IMG_ROW = 0, IMG_TR = 1, IMG_UNI = 2;

// DMA one 64-row tile (rows = tokens tok0.., D contiguous elements at column col0) into LDS; IMG picks the swizzle
template <int D, int IMG>
__device__ __forceinline__ void dma_tile(const bf16_t* base, int64_t ld, int rows_valid, char* lds, int wave, int lane) {
    using C = Cfg<D>;
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(base), 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int j = 0; j < C::PPW; ++j) {
        const int pi = wave * C::PPW + j;
        const int row = pi * C::RPP + lane / C::CH;
        const int pos = lane % C::CH;
        const int c = IMG == IMG_UNI ? swz_uni(pos, row) : IMG == IMG_TR ? swz_tr<D>(pos, row) : swz_row<D>(pos, row);
        const unsigned voff = row < rows_valid ? (unsigned)(row * ld * 2 + c * 16) : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(lds + pi * 1024), 16, voff, 0, 0, 0);
    }
}

// A-operand fragment (32 rows x 16 k) of a row image: row = r0 + (lane&31), k-step ks
template <int D>
__device__ __forceinline__ bf16x8 frag_rows(const char* img, int r0, int ks, int lane) {
    const int row = r0 + (lane & 31);
    const int chunk = 2 * ks + (lane >> 5);
    return *reinterpret_cast<const bf16x8*>(img + row * Cfg<D>::ROWB + (swz_row<D>(chunk, row) << 4));
}

// A-operand fragment of the TRANSPOSE of a tr image: rows of A = 32 columns c0.. of the image, k = image rows in the
// order the accumulator-as-operand trick needs: element j <-> image row  k0 + 8*(j>>2) + 4*(lane>>5) + (j&3).
template <int D>
__device__ __forceinline__ bf16x8 frag_cols(const char* img, int c0, int k0, int lane) {
    const int g = lane >> 4, q4 = (lane >> 2) & 3, p = lane & 3, h = g >> 1;
    const int row = k0 + 4 * h + q4;
    const int col = c0 + 16 * (g & 1) + 4 * p;
    const char* a = img + row * Cfg<D>::ROWB + (swz_tr<D>(col >> 3, row) << 4) + (p & 1) * 8;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(a));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(a + 8 * Cfg<D>::ROWB));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

// ---- precomputed per-lane LDS offsets -----------------------------------------------------------------------------
// The XOR swizzles above defeat the compiler's immediate-offset folding (it re-derives ~6-10 VALU ops per LDS read, and
// the attention loops were VALU-bound at 18 VALU per MFMA).  Both fragment addresses factor into
//     row image:  tile_base + r0*ROWB + ( lane_row ^ (ks << 5) )               lane_row = r*ROWB + ((h ^ swz(r)) << 4)
//     tr  image:  tile_base + k0*ROWB + lane_col[dt]   (+ 8*ROWB second half)   lane_col[dt] = lane part + ((dt ^ x) << 6)
// with tile_base / r0 / k0 multiples of 4 KiB resp. ROWB (they never touch bits 4..7, so they commute with the XOR).
template <int D>
struct LaneOff {
    int row;
    int col[Cfg<D>::DT];
    int rowu;               // unified image (D = 128 only)
    int colu[Cfg<D>::DT];
};
template <int D>
__device__ __forceinline__ LaneOff<D> lane_offsets(int lane) {
    using C = Cfg<D>;
    LaneOff<D> o;
    const int r = lane & 31, h = lane >> 5;
    const int sw = D == 128 ? (r & 15) : ((r >> 1) & 7);
    o.row = r * C::ROWB + ((h ^ sw) << 4);
    const int g = lane >> 4, q4 = (lane >> 2) & 3, p = lane & 3;
    const int x = D == 128 ? q4 : ((q4 >> 1) & 1);
    const int base = (4 * (g >> 1) + q4) * C::ROWB + ((2 * (g & 1) + (p >> 1)) << 4) + (p & 1) * 8;
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt) o.col[dt] = base + ((dt ^ x) << 6);
    // unified image: row r -> chunk ^ ((r&3)<<2 | (r>>2)&3); a transposing read touches rows 4*(g>>1) + q4 (+8: low field ^ 2)
    o.rowu = r * C::ROWB + ((h ^ (((r & 3) << 2) | ((r >> 2) & 3))) << 4);
    const int baseu = (4 * (g >> 1) + q4) * C::ROWB + (((2 * (g & 1) + (p >> 1)) ^ (g >> 1)) << 4) + (p & 1) * 8;
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt) o.colu[dt] = baseu + ((dt ^ q4) << 6);
    return o;
}
// row-image fragment: vx = (lane_row + image_offset) ^ (ks << 5), imm = r0 * ROWB
__device__ __forceinline__ bf16x8 lds_frag(const char* smem, int vx, int imm) { return *reinterpret_cast<const bf16x8*>(smem + vx + imm); }
// tr-image fragment: v = lane_col[dt] + image_offset, imm = k0 * ROWB
template <int D>
__device__ __forceinline__ bf16x8 lds_frag_tr(const char* smem, int v, int imm) {
    const char* a = smem + v + imm;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(a));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(a + 8 * Cfg<D>::ROWB));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

// unified-image transposed fragment: v = lane_colu[dt] + image_offset, imm = k0 * ROWB (k0 a multiple of 16); the rows of
// the second read are 8 further down, where the low swizzle field differs by 2 (byte bit 5)
template <int D>
__device__ __forceinline__ bf16x8 lds_frag_tr_uni(const char* smem, int v, int imm) {
    const char* a = smem + v + imm;
    const char* a2 = smem + (v ^ 0x20) + imm + 8 * Cfg<D>::ROWB;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(a));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(a2));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}
// Transposed fragment read from an asm statement.  hipcc (ROCm 7.2) puts `s_waitcnt vmcnt(0)` in front of the first
// ds_read_tr16_b64 INTRINSIC that follows an LDS-DMA (it cannot tell the read from the DMA's destination; plain ds_read_b128
// loads are not affected): with one wave per SIMD that exposes the whole latency of the tile just requested, every trip.
// An asm read is invisible to that logic -- and to the compiler's lgkmcnt bookkeeping: the two halves stay separate 64-bit
// values until `tr_wait<N>` (the statement that carries the counted s_waitcnt and names both halves) has run.  LDS returns
// data in issue order, so N = the LDS operations issued after this fragment's reads (compiler-issued reads in between only
// make the wait stricter).
struct TrHalves {
    bf16x4 lo, hi;
};
template <int IMM0, int IMM1>
__device__ __forceinline__ void tr_issue(TrHalves& f, unsigned a0, unsigned a1) {
    static_assert(IMM0 >= 0 && IMM1 < 65536, "ds offset field is 16 bits");
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%c4\n\tds_read_b64_tr_b16 %1, %3 offset:%c5" : "=&v"(f.lo), "=&v"(f.hi) : "v"(a0), "v"(a1), "i"(IMM0), "i"(IMM1));
}
template <int N>
__device__ __forceinline__ bf16x8 tr_wait(TrHalves& f) {
    static_assert(N >= 0 && N <= 15, "lgkmcnt field is 4 bits");
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f.lo), "+v"(f.hi) : "n"(N) : "memory");
    return __builtin_shufflevector(f.lo, f.hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
// Workgroup id -> (query block, head) of the per-block attention kernels.  Under the causal mask a block's work grows with its index (2, 4, ... tiles), so
// the heaviest blocks go first -- not head by head (the light blocks of the LAST heads then run on a half-empty chip: 116 tile-times where 108 are the
// ideal on 64 slots per XCD at the headline shape) but over groups of GROUP consecutive heads: all heaviest blocks of the group, then the next lighter
// ones, ...  (GROUP = 8: 110; 16: 110; 32: 108 -- but from 32 on the K / V rows of the heads running together no longer fit an XCD's 4 MB L2; measured forward at the headline shape: 250 us head by head, 241 at 8, 237 at 16, 242 at 32, 244 at 64).
#ifndef ATTN_HEAD_GROUP
#define ATTN_HEAD_GROUP 16
#endif
__device__ __forceinline__ void heavy_first(int vid, int nqb, int heads, int group, int& qb, int& bh) {
    const int per = group * nqb, sg = vid / per, r = vid - sg * per;
    const int gsz = min(group, heads - sg * group);
    qb = nqb - 1 - r / gsz;
    bh = sg * group + r % gsz;
}

// Row-image fragment (ds_read_b128) from an asm statement, for loops that interleave LDS-DMA requests with their fragment reads: hipcc orders
// every compiler-visible LDS read behind an LDS-DMA in front of it and so cannot request fragments ahead across one.  Same contract as tr_issue /
// tr_wait: the value is dead until rowfrag_wait<N> (N = LDS operations issued after this read) has named it.
template <int IMM>
__device__ __forceinline__ void rowfrag_issue(u32x4& f, unsigned a) {
    static_assert(IMM >= 0 && IMM < 65536, "ds offset field is 16 bits");
    asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=&v"(f) : "v"(a), "i"(IMM));
}
template <int N>
__device__ __forceinline__ bf16x8 rowfrag_wait(u32x4& f) {
    static_assert(N >= 0 && N <= 15, "lgkmcnt field is 4 bits");
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "n"(N) : "memory");
    return __builtin_bit_cast(bf16x8, f);
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// pack accumulator registers 8s..8s+7 to a bf16 B-operand fragment
__device__ __forceinline__ bf16x8 pack_frag(const f32x16& x, int s) {
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack_bf2(x[8 * s + 2 * e], x[8 * s + 2 * e + 1]);
    return __builtin_bit_cast(bf16x8, o);
}

// B-operand fragments of a [32 rows][D] row-major global matrix (rows on the lane): row = r0 + (lane&31)
template <int D>
__device__ __forceinline__ void load_rows_frag(const bf16_t* base, int64_t ld, int row, bool valid, int lane, bf16x8 (&f)[Cfg<D>::KS]) {
#pragma unroll
    for (int ks = 0; ks < Cfg<D>::KS; ++ks) {
        u32x4 v = {0, 0, 0, 0};
        if (valid) v = *reinterpret_cast<const u32x4*>(base + (int64_t)row * ld + 16 * ks + 8 * (lane >> 5));
        f[ks] = __builtin_bit_cast(bf16x8, v);
    }
}

// ---- registers owned by name in the accumulator file -----------------------------------------------------------------
// The backward kernels need > 256 registers per lane.  Two things hipcc (ROCm 7.2) does with that were the whole cost of the
// first version of these kernels: (1) it homes loop-carried MFMA accumulators in VGPRs and copies all 16 registers of a tile
// into and out of the AGPRs around every MFMA (840 v_accvgpr_* per loop trip of the dK/dV pass); (2) it issues each LDS
// fragment read immediately in front of the MFMA that consumes it, so with one wave per SIMD every MFMA waits a full LDS
// latency (~10k cycles per trip for 2k cycles of MFMA).  So here:
//   * dK^T / dV^T / dQ^T tiles and the register-resident B operands (K, V resp. Q, dO rows) are literal AGPRs at the TOP of
//     the accumulator file, a[256-OWNED ...], touched only by the statements below.  Every statement lists the whole owned
//     range as clobbered: that reserves it in the kernel descriptor and keeps compiler values that live across a statement
//     out of it; hipcc allocates AGPRs for its own purposes from a0 upward and stays below (tools/audit_agpr.py, run by
//     tests/test_abi_cpu.py, fails the build if a compiler instruction names a register of an owned range).
//   * every MFMA is a volatile statement with a "memory" clobber, so LDS reads keep their source order relative to the
//     MFMAs: the kernels issue fragment reads PD MFMAs ahead into a ring of R register slots, and the compiler only adds
//     the counted lgkmcnt waits.
#define AGPR_CL_64 "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255"
#define AGPR_CL_96 AGPR_CL_64, "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191"
#define AGPR_CL_128 AGPR_CL_96, "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159"
#define AGPR_CL_192 AGPR_CL_128, "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127"
#ifndef ATTN_RING
#define ATTN_RING 8   // fragment-ring slots of the backward kernels
#define ATTN_PD 6     // fragments requested ahead of the MFMA that consumes them (2 * PD <= 15: lgkmcnt is 4 bits)
#endif
#ifndef ATTN_ABL
#define ATTN_ABL 0  // profiling builds only: 1 = no fragment reads, 2 = no B phase, 4 = no C MFMAs, 8 = no A MFMAs
#endif
#define OWNED_ASM(OWNED, ...)                                                     \
    do {                                                                          \
        static_assert((OWNED) == 64 || (OWNED) == 96 || (OWNED) == 128 || (OWNED) == 192, "no clobber list of this size"); \
        if constexpr ((OWNED) == 64) asm volatile(__VA_ARGS__ : AGPR_CL_64, "memory");        \
        else if constexpr ((OWNED) == 96) asm volatile(__VA_ARGS__ : AGPR_CL_96, "memory");   \
        else if constexpr ((OWNED) == 128) asm volatile(__VA_ARGS__ : AGPR_CL_128, "memory"); \
        else asm volatile(__VA_ARGS__ : AGPR_CL_192, "memory");                               \
    } while (0)

#if ATTN_ABL & 16
__device__ unsigned long long g_prof[16];
#define PROF_T() (prof_on ? __builtin_readcyclecounter() : 0ull)
#define PROF_ADD(i, t0) do { if (prof_on) prof_acc[i] += __builtin_readcyclecounter() - (t0); } while (0)
#else
#define PROF_T() 0ull
#define PROF_ADD(i, t0) do { (void)(t0); } while (0)
#endif
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}
// owned tile (16 registers from owned offset OFF) += A x B, both operands in VGPRs.
// NOP (s_nop 1): for a B register written by the VALU instruction just before (v_cvt_pk -> MFMA operand); hipcc pads nothing inside asm.
template <int OWNED, int OFF, bool NOP = true>
__device__ __forceinline__ void mfma_owned(const bf16x8& a, const bf16x8& b) {
    static_assert(OFF % 16 == 0 && OFF + 16 <= OWNED, "tile outside the owned range");
    constexpr int R0 = 256 - OWNED + OFF;
    if constexpr (NOP) OWNED_ASM(OWNED, "s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(a), "v"(b), "i"(R0), "i"(R0 + 15));
    else OWNED_ASM(OWNED, "v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(a), "v"(b), "i"(R0), "i"(R0 + 15));
}
// compiler-allocated tile (16 VGPRs: the VALU reads it right after, no accvgpr copies) (+)= A x ownedB, where ownedB is the
// 4-register B operand at owned offset OFF.  FIRST: start from zero (srcC = 0) instead of accumulating.
template <int OWNED, int OFF, bool FIRST>
__device__ __forceinline__ void mfma_ownedB(f32x16& acc, const bf16x8& a) {
    static_assert(OFF % 4 == 0 && OFF + 4 <= OWNED, "operand outside the owned range");
    constexpr int R0 = 256 - OWNED + OFF;
    if constexpr (FIRST) OWNED_ASM(OWNED, "v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], 0" : "=&v"(acc) : "v"(a), "i"(R0), "i"(R0 + 3));
    else OWNED_ASM(OWNED, "v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], %0" : "+v"(acc) : "v"(a), "i"(R0), "i"(R0 + 3));
}
// the same, the chain's first product: acc = A x ownedB + init (a compiler-allocated tile of initial values -- row constants such as -lse)
template <int OWNED, int OFF>
__device__ __forceinline__ void mfma_ownedB_init(f32x16& acc, const bf16x8& a, const f32x16& init) {
    static_assert(OFF % 4 == 0 && OFF + 4 <= OWNED, "operand outside the owned range");
    constexpr int R0 = 256 - OWNED + OFF;
    OWNED_ASM(OWNED, "v_mfma_f32_32x32x16_bf16 %0, %1, a[%c3:%c4], %2" : "=&v"(acc) : "v"(a), "v"(init), "i"(R0), "i"(R0 + 3));
}
// wait states between the last MFMA into a compiler-allocated tile and its first VALU read (16-pass XDL -> read: 18)
__device__ __forceinline__ void tiles_settle(f32x16& x, f32x16& y) { asm volatile("s_nop 15\n\ts_nop 3" : "+v"(x), "+v"(y)); }
template <int OWNED, int OFF, int COUNT>
__device__ __forceinline__ void owned_zero() {
    static_for<COUNT>([&](auto r) { OWNED_ASM(OWNED, "v_accvgpr_write_b32 a[%c0], 0" ::"i"(256 - OWNED + OFF + r.value)); });
}
// a 4-register operand (8 bf16) into owned offset OFF
template <int OWNED, int OFF>
__device__ __forceinline__ void owned_write4(const bf16x8& v) {
    const u32x4 w = __builtin_bit_cast(u32x4, v);
    OWNED_ASM(OWNED, "v_accvgpr_write_b32 a[%c4], %0\n\tv_accvgpr_write_b32 a[%c5], %1\n\tv_accvgpr_write_b32 a[%c6], %2\n\tv_accvgpr_write_b32 a[%c7], %3"
              ::"v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "i"(256 - OWNED + OFF), "i"(256 - OWNED + OFF + 1), "i"(256 - OWNED + OFF + 2), "i"(256 - OWNED + OFF + 3));
}
template <int OWNED>
__device__ __forceinline__ void owned_settle() { OWNED_ASM(OWNED, "s_nop 15\n\ts_nop 3" ::); }
template <int OWNED, int R>
__device__ __forceinline__ float owned_read() {
    float x;
    asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(x) : "i"(256 - OWNED + R));
    return x;
}

__device__ __forceinline__ int acc_row(int e, int lane) { return (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5); }

// ================================================================================================ forward
// position b of a round-robin-over-XCDs numbering -> position in a numbering where each XCD owns one contiguous chunk: workgroups
// with adjacent VIRTUAL ids (the query blocks of one head, the two query heads of one kv head) then share an XCD and its L2, so
// that K / V of a (batch, kv head) come over the fabric once per XCD instead of once per workgroup (FETCH_SIZE, profiles/).
__device__ __forceinline__ int xcd_chunked(int b, int n) {
    const int xcd = b & 7, q = n >> 3, r = n & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}

}  // namespace
