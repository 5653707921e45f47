// Attention forward, second generation (head_dim 128, an even number of query heads per kv head): one wave per SIMD, persistent workgroups.
//
// Why a second kernel.  The first forward (attention.hip: 4 waves x 32 queries of ONE head, two workgroups per CU, compiler-scheduled) spends
// 5 600 cycles per wave on a 64-key tile that holds 1 024 cycles of MFMA (profiles/r02_pmc_sq_counters.json: matrix pipe 22 % busy, 10.4 vector
// instructions per MFMA, every tile's LDS-DMA issue and the softmax serialised in front of / behind the products).  This kernel changes the shape
// of the work a wave does:
//   * a wave owns 32 query rows of BOTH query heads of a kv-head pair (group g = head 2p + g): 64 MFMAs per 64-key tile against the same K / V
//     tile, so every LDS-DMA piece and every barrier is amortised over twice the matrix work, and K / V come over the fabric once per pair;
//   * one wave per SIMD with the whole register file: O^T of both groups (128 registers) and the Q rows (64) are OWNED accumulator registers
//     (attn_common.h), S^T tiles and packed P live in VGPRs, every MFMA is an asm statement, LDS fragment reads run PD MFMAs ahead in a ring;
//   * a lone wave can hide about 24 cycles of vector issue behind a 32-cycle MFMA, and costs grow faster than linearly past that (DESIGN.md
//     section 5, tools/microbench/mfma_gap.hip).  So the softmax is cut to exp2 + row-sum add + half a bf16 pack + half a max3 per score, and
//     laid out so that the exp2 are spread evenly over the MFMA gaps (two in every other gap):
//       - the scale log2(e) / sqrt(d) is folded into the Q rows once per block (bf16(q * c): one more rounding of a bf16 operand, far inside
//         the reference's own noise -- it rounds the scores themselves to bf16, twice);
//       - the running reference m of a row is the INITIAL ACCUMULATOR of its score products (a 16-register tile per group holding -m), so the
//         MFMA chain delivers s * c - m and no subtraction is ever issued;
//       - the reference moves only when a row's scores outgrow it by 2^8 (or, before anything is accumulated, in either direction): O, l, the
//         tile at hand and the initial-accumulator tile are then re-based by one factor (three instructions per owned register -- rare, and the
//         data-dependent branch tests force it explicitly);
//       - the two groups' element streams (exp2 -> add -> pack, one slot apart) interleave over the 64 slots of a tile:
//           slot   0 .. 15   S(t,1) = K_t Q_1^T        16 .. 31   O_0 += V_t^T P(t,0)     32 .. 47   S(t+1,0) = K_t+1 Q_0^T     48 .. 63   O_1 += V_t^T P(t,1)
//           stream group 0 of tile t: positions 7 .. 32 in slots 0 .. 25     group 1 of tile t: slots 25 .. 57     group 0 of tile t+1: positions 0 .. 6 in slots 57 .. 63
//                  (exp2 pairs in the odd slots, adds + pack in the even ones: a lone wave pays dearly for mixing the two kinds in one gap)
//           max    group 1: slots 10 .. 22 (even), decision 24                 group 0 of t+1: slots 42 .. 54 (even), decision 56
//   * masks are two 32-bit words per lane and tile (set to the fill value / set to zero), applied with v_bfe_i32 + v_bfi_b32 per score and only on
//     tiles that touch the diagonal, the sequence end or padded keys;
//   * a workgroup walks work items (batch, kv head, head pair, run of 128-query blocks), heaviest block first, with ONE 4-stage K/V tile stream
//     running across block and item boundaries; the next block's Q rows are requested before the current block's output is written.
// Semantics are those of attention.hip (reference qwen3_attention.py:121-146): masked scores take a finite fill value, so a row whose visible
// keys are ALL masked attends uniformly to all S keys.  Under the causal mask such rows are exactly the queries in front of the first real key of
// a left-padded batch row (without it: every row of a batch row that is all padding); they are known before the first tile, get score 0 for every
// existing key, and their batch rows walk every tile.
#include "attn_common.h"

namespace {

#ifndef F2_ABL
#define F2_ABL 0  // timing-only builds (wrong results): 1 = no softmax vector work in the tile, 2 = no LDS fragment reads after the first, 4 = no O^T products, 8 = no S^T products,
                  // 32 = no exp2, 64 = no row-sum adds, 128 = no bf16 packs, 256 = no maxima / decisions
#endif
constexpr float F2_THR = 8.0f;  // log2 units a row's scores may outgrow its reference before everything is re-based
constexpr int F2_CUS = 256;
constexpr unsigned F2_FILL = __builtin_bit_cast(unsigned, MASK_T);

#if ATTN_ABL & 16
__device__ unsigned long long g_prof2[32];
#endif
template <int NP, int NE>
__device__ __forceinline__ void f2_fillers(float* f) {
#pragma unroll
    for (int k = 0; k < NP; ++k) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[k % 8]));
#pragma unroll
    for (int k = 0; k < NE; ++k) asm volatile("v_exp_f32_e32 %0, %0" : "+v"(f[8 + k % 4]));
}
template <int N>
__device__ __forceinline__ void f2_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int OWNED, int R>
__device__ __forceinline__ void f2_owned_write(float x) { OWNED_ASM(OWNED, "v_accvgpr_write_b32 a[%c1], %0" ::"v"(x), "i"(256 - OWNED + R)); }
// score tile = A x ownedB + init  (the chain's first product; `init` holds -reference in all 16 registers)
template <int OWNED, int OFF>
__device__ __forceinline__ void f2_mfma_init(f32x16& acc, const bf16x8& a, const f32x16& init) {
    constexpr int R0 = 256 - OWNED + OFF;
    OWNED_ASM(OWNED, "v_mfma_f32_32x32x16_bf16 %0, %1, a[%c3:%c4], %2" : "=&v"(acc) : "v"(a), "v"(init), "i"(R0), "i"(R0 + 3));
}
// bit e of `word` set -> x = fill  (two instructions, no compare)
__device__ __forceinline__ float f2_set_if(float x, unsigned word, int e, unsigned fill_bits) {
    const unsigned sel = (unsigned)__builtin_amdgcn_sbfe((int)word, (unsigned)e, 1u);  // 0 or ~0
    return __uint_as_float((__float_as_uint(x) & ~sel) | (fill_bits & sel));
}
// Single instructions placed by hand.  Left to hipcc, fmaxf on MFMA outputs grows two canonicalising v_max (three instructions per maximum), and
// an exp2 whose input has been ready for a while is hoisted out of the slot it was written in (seven of them ended up in one clump in front of the
// tile).  A volatile statement stays where it is written; none of the consumers sits in the instruction right behind its producer (the
// transcendental-result hazard needs one instruction in between: every result here is consumed at least one MFMA slot later).
__device__ __forceinline__ float f2_exp2(float x) {
    asm volatile("v_exp_f32_e32 %0, %0" : "+v"(x));
    return x;
}
__device__ __forceinline__ float f2_add(float acc, float x) {
    asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(acc) : "v"(x));
    return acc;
}
__device__ __forceinline__ unsigned f2_pack(float lo, float hi) {
    unsigned r;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ float f2_max3(float m, float x, float y) {
    asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(m) : "v"(x), "v"(y));
    return m;
}
// the 16 accumulator rows of a lane (bits 0-3, 8-11, 16-19, 24-27 of a 32-key word already shifted by 4 * half-wave) gathered into 16 bits
__device__ __forceinline__ unsigned f2_gather16(unsigned w) { return (w & 0xFu) | ((w >> 4) & 0xF0u) | ((w >> 8) & 0xF00u) | ((w >> 12) & 0xF000u); }

template <int D>
__global__ __launch_bounds__(256, 1) void attn_fwd2_kernel(int B, int S, int Hq, int Hkv, const bf16_t* __restrict__ q, int64_t ldq,
                                                           const bf16_t* __restrict__ k, int64_t ldk, const bf16_t* __restrict__ v, int64_t ldv,
                                                           bf16_t* __restrict__ o, int64_t ldo, float* __restrict__ lse,
                                                           const uint8_t* __restrict__ key_mask, int causal, float scale_log2, int bpw, int nitems) {
    static_assert(D == 128, "built for head_dim 128");
    using C = Cfg<D>;
    constexpr int KS = C::KS, DT = C::DT;
    constexpr int OWNED = 192;  // a[64:255]:  O^T tiles of group g at 64 g + 16 dt | Q rows of group g at 128 + 32 g + 4 ks
    constexpr int NST = 4, STAGE = 2 * C::TILE, PIECES = 2 * C::PPW;  // a stage = K row image | V transposed-read image
    constexpr int RING = 8, PD = 6;
    causal &= 0xff;
    __shared__ __attribute__((aligned(16))) char smem[NST * STAGE + 2 * ATTN_MAX_TILES * 8];
    unsigned long long* kmw_all = reinterpret_cast<unsigned long long*>(smem + NST * STAGE);  // key-padding words of the item, double-buffered
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rep = Hq / Hkv, pairs = rep >> 1;
    const int nqb = (S + 127) / 128, nchunk = (nqb + bpw - 1) / bpw;
    const int ntiles_all = (S + 63) / 64;
    const int G = (int)gridDim.x;

    // ---- work items -----------------------------------------------------------------------------------------------------------------
    struct Item {
        int b, hkv, pr, qb_hi, qb_lo, allt;  // allt: key 0 of the batch row is padding -- fully masked rows may exist, every tile is walked
    };
    auto decode = [&](int item) {
        Item it;
        const int chunk = item % nchunk, rest = item / nchunk;
        it.pr = rest % pairs;
        const int bh = rest / pairs;
        it.hkv = bh % Hkv;
        it.b = bh / Hkv;
        it.qb_hi = nqb - chunk * bpw;
        it.qb_lo = max(0, it.qb_hi - bpw);
        it.allt = 0;
        if (key_mask) {  // one aligned scalar word (the host checked the alignment)
            const int64_t off = (int64_t)it.b * S;
            const unsigned w = reinterpret_cast<const unsigned*>(key_mask)[off >> 2];
            it.allt = ((w >> (8 * (int)(off & 3))) & 0xffu) == 0;
        }
        return it;
    };
    auto tiles_of = [&](const Item& it, int qb) { return (causal && !it.allt) ? min(ntiles_all, (qb * 128 + 127) / 64 + 1) : ntiles_all; };

    // ---- the K / V tile stream ------------------------------------------------------------------------------------------------------
    // per-lane parts of the DMA source offsets (a wave's piece j of an image covers rows 16 wave + 4 j + (lane >> 4), 16-byte chunk lane & 15):
    // K row image: chunk ^ (row & 15) = (chunk ^ (lane >> 4)) ^ 4 j;  V transposed-read image: chunk ^ ((row & 3) << 2), the same for every piece
    const unsigned dk_row = (unsigned)((wave * 16 + (lane >> 4)) * (int)ldk * 2), dk_c = (unsigned)(((lane & 15) ^ (lane >> 4)) << 4);
    const unsigned dv_off = (unsigned)((wave * 16 + (lane >> 4)) * (int)ldv * 2) + (unsigned)(((lane & 15) ^ (((lane >> 4) & 3) << 2)) << 4);
    int s_item = (int)blockIdx.x, s_qb = 0, s_kt = 0, s_stage = 0, issued = 0;
    Item s_it = {};
    bool s_live = s_item < nitems;
    if (s_live) {
        s_it = decode(s_item);
        s_qb = s_it.qb_hi - 1;
    }
    auto issue_next = [&]() {
        if (!s_live) return;
        char* st_ = smem + s_stage * STAGE;
        const bf16_t* kp = k + ((int64_t)s_it.b * S + (int64_t)s_kt * 64) * ldk + (int64_t)s_it.hkv * D;
        const bf16_t* vp = v + ((int64_t)s_it.b * S + (int64_t)s_kt * 64) * ldv + (int64_t)s_it.hkv * D;
        if (S - s_kt * 64 >= 64) {  // whole tile: no per-lane bound, the piece's rows go into the scalar offset
            auto kr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(kp), 0, 0x7fffffff, 0x00020000);
            auto vr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(vp), 0, 0x7fffffff, 0x00020000);
#pragma unroll
            for (int j = 0; j < C::PPW; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(kr, LDS_PTR(st_ + (wave * C::PPW + j) * 1024), 16, dk_row + (dk_c ^ (unsigned)(j << 6)), (int)(4 * j * ldk * 2), 0, 0);
#pragma unroll
            for (int j = 0; j < C::PPW; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(vr, LDS_PTR(st_ + C::TILE + (wave * C::PPW + j) * 1024), 16, dv_off, (int)(4 * j * ldv * 2), 0, 0);
        } else {
            dma_tile<D, IMG_ROW>(kp, ldk, S - s_kt * 64, st_, wave, lane);
            dma_tile<D, IMG_TR>(vp, ldv, S - s_kt * 64, st_ + C::TILE, wave, lane);
        }
        ++issued;
        s_stage = s_stage == NST - 1 ? 0 : s_stage + 1;
        if (++s_kt == tiles_of(s_it, s_qb)) {
            s_kt = 0;
            if (--s_qb < s_it.qb_lo) {
                s_item += G;
                s_live = s_item < nitems;
                if (s_live) {
                    s_it = decode(s_item);
                    s_qb = s_it.qb_hi - 1;
                }
            }
        }
    };
    issue_next();
    issue_next();
    issue_next();

    const LaneOff<D> lo = lane_offsets<D>(lane);
    const unsigned lds0 = (unsigned)(unsigned long long)LDS_PTR(smem);
    // causal mask of a diagonal 32 x 32 sub-tile: bit ee set <=> key row acc_row(ee) lies behind this lane's query
    unsigned tri16 = 0;
#pragma unroll
    for (int ee = 0; ee < 16; ++ee) tri16 |= (acc_row(ee, lane) > (lane & 31) ? 1u : 0u) << ee;
    [[maybe_unused]] const bool prof_on = threadIdx.x == 0;
    [[maybe_unused]] unsigned long long prof_acc[32] = {};
    [[maybe_unused]] const unsigned long long t_wg = PROF_T();
    int step = 0;    // tiles consumed so far (global over the stream)
    int cstage = 0;  // stage of the tile consumed at this step
    int item_no = 0;

    // Q rows of a block (both groups), requested one block ahead
    bf16x8 tq[2][KS];
    auto request_q = [&](const Item& it_, int qb_) {
        const int qg_ = qb_ * 128 + wave * 32 + (lane & 31);
        const int h0 = it_.hkv * rep + 2 * it_.pr;
        load_rows_frag<D>(q + (int64_t)it_.b * S * ldq + (int64_t)h0 * D, ldq, qg_, qg_ < S, lane, tq[0]);
        load_rows_frag<D>(q + (int64_t)it_.b * S * ldq + (int64_t)(h0 + 1) * D, ldq, qg_, qg_ < S, lane, tq[1]);
    };
    if ((int)blockIdx.x < nitems) {
        const Item it0 = decode((int)blockIdx.x);
        request_q(it0, it0.qb_hi - 1);
    }

    for (int item = (int)blockIdx.x; item < nitems; item += G, ++item_no) {
        [[maybe_unused]] const unsigned long long t_item = PROF_T();
        const Item it = decode(item);
        const int b = it.b, hkv = it.hkv;
        unsigned long long* kmw = kmw_all + (item_no & 1) * ATTN_MAX_TILES;
        int first_real = 0;  // index of the batch row's first real key (S: none)
        if (key_mask) {
            // key-padding bits of every 64-key tile of this batch row (1 = real token; keys beyond S read 0).  The other buffer may still be read by
            // a wave that is one step behind; this one was last read a whole item ago.
            for (int t = wave; t < ntiles_all; t += 4) {
                const int kgl = t * 64 + lane;
                const unsigned long long bits = __ballot(kgl < S && key_mask[(int64_t)b * S + kgl] != 0);
                if (lane == 0) kmw[t] = bits;
            }
            if (it.allt) {  // rare: a left-padded (or empty) batch row
                first_real = S;
                for (int t = 0; t < ntiles_all; ++t) {
                    const int kgl = t * 64 + lane;
                    const unsigned long long bits = __ballot(kgl < S && key_mask[(int64_t)b * S + kgl] != 0);
                    if (bits) {
                        first_real = t * 64 + (int)__builtin_ctzll(bits);
                        break;
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the words are read by every wave behind the next raw barrier
        }
        PROF_ADD(12, t_item);
        for (int qb = it.qb_hi - 1; qb >= it.qb_lo; --qb) {
            [[maybe_unused]] const unsigned long long t_bp = PROF_T();
            const int q0 = qb * 128;
            const int qw = q0 + wave * 32;
            const int qg = qw + (lane & 31);
            const bool qvalid = qg < S;
            const int ntiles = tiles_of(it, qb);
            // tiles in which this wave's 32 queries see at least one key (the rest of the block it only keeps the stream going)
            const int nt_w = (causal && !it.allt) ? min(ntiles, (qw + 31) / 64 + 1) : ntiles;
            // rows whose visible keys are all padding: score 0 for every key that exists
            const bool qrow = it.allt && (causal ? qg < first_real : first_real >= S);

            // ---- block prologue: Q rows times log2(e) / sqrt(d) (rounded to bf16 again) into their owned registers, O = 0
            static_for<2>([&](auto gc) {
                static_for<KS>([&](auto ks) {
                    u32x4 w = __builtin_bit_cast(u32x4, tq[gc.value][ks.value]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) w[e] = pack_bf2(__uint_as_float(w[e] << 16) * scale_log2, __uint_as_float(w[e] & 0xffff0000u) * scale_log2);
                    owned_write4<OWNED, 128 + 32 * gc.value + 4 * ks.value>(__builtin_bit_cast(bf16x8, w));
                });
            });
            PROF_ADD(20, t_bp);
            owned_zero<OWNED, 0, 128>();
            PROF_ADD(7, t_bp);
            if (prof_on) prof_acc[11] += 1;

            float mref[2] = {0.f, 0.f}, l[2] = {0.f, 0.f};
            f32x16 ninit[2];      // -mref in every register: the initial accumulator of a group's score products
            f32x16 sacc[2][2];    // [group][32-key sub-tile]: s*c - mref, then p
            unsigned pw[2][16];   // [group]: packed P^T, words 4 (2 st + s) .. + 3 = B operand of k-step (st, s)
            float ps[2][2];       // [group]: two partial row sums of the tile in flight
            float tmx[2];         // running maxima of the tile whose maxima are being taken
#pragma unroll
            for (int g = 0; g < 2; ++g) {
#pragma unroll
                for (int e = 0; e < 16; ++e) ninit[g][e] = 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) pw[g][e] = 0;
                ps[g][0] = ps[g][1] = 0.f;
            }
            tmx[0] = tmx[1] = -INFINITY;

            // mask words of tile t for this lane, bit e = 16 st + ee:  mset -> the score becomes the fill value, zset -> it becomes 0 (rows whose
            // visible keys are all padding: every existing key counts alike).  kb: bit per key of the tile, 1 = real and existing.
            auto tile_masks = [&](int t, unsigned long long kb, unsigned& mset, unsigned& zset) {
                const int nv = S - t * 64;  // keys of this tile that exist
                const unsigned long long exist = nv >= 64 ? ~0ull : ((1ull << (nv < 0 ? 0 : nv)) - 1ull);
                unsigned m16[2], x16[2];
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    const unsigned sh = 4u * (unsigned)(lane >> 5);
                    const unsigned vis = f2_gather16((st ? (unsigned)(kb >> 32) : (unsigned)kb) >> sh);
                    const unsigned exi = f2_gather16((st ? (unsigned)(exist >> 32) : (unsigned)exist) >> sh);
                    const int x = qw - t * 64 - st * 32;  // the wave's first query against the sub-tile's first key (a multiple of 32): 0 = diagonal
                    const unsigned cz = (!causal || x >= 32) ? 0u : (x == 0 ? tri16 : 0xFFFFu);
                    m16[st] = (~vis & 0xFFFFu) | cz;
                    x16[st] = exi;
                }
                const unsigned mall = m16[0] | (m16[1] << 16), eall = x16[0] | (x16[1] << 16);
                mset = qrow ? ~eall : mall;
                zset = qrow ? eall : 0u;
            };

            // ---- the softmax pieces, each written for ONE MFMA slot ---------------------------------------------------------------------------
            // maxima over elements 2k, 2k+1 of sub-tile st of sacc[g] (masks applied first on boundary tiles)
            auto max_pair = [&](auto gc, auto bndc, auto stc, auto kc, unsigned mset, unsigned zset, bool zany) {
                constexpr int g = decltype(gc)::value, st = decltype(stc)::value, kk = decltype(kc)::value;
                constexpr bool BND = decltype(bndc)::value;
                static_for<2>([&](auto i) {
                    constexpr int ee = 2 * kk + i.value, e = 16 * st + ee;
                    if constexpr (BND) {
                        float t = f2_set_if(sacc[g][st][ee], mset, e, F2_FILL);
                        if (zany) t = f2_set_if(t, zset, e, 0u);
                        sacc[g][st][ee] = t;
                    }
                });
                tmx[kk & 1] = f2_max3(tmx[kk & 1], sacc[g][st][2 * kk], sacc[g][st][2 * kk + 1]);
            };
            // the decision: does any row's tile maximum leave the window its reference allows?  If so re-base O, l, the tile and the initial accumulator.
            auto decide = [&](auto gc) {
                constexpr int g = decltype(gc)::value;
                if constexpr (F2_ABL & (256 | 512)) return;
                const float t0 = f2_max3(tmx[0], tmx[1], tmx[1]);
                const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(t0), __float_as_uint(t0), false, false);
                const float t = f2_max3(__uint_as_float(r[0]), __uint_as_float(r[1]), __uint_as_float(r[1]));  // the two half-waves hold the two halves of a query's keys
                // up: always.  Down: only while nothing is accumulated, and never onto the fill value
                const bool want = t > F2_THR || (l[g] == 0.f && t < -F2_THR && t > 0.5f * MASK_T);
                if (__builtin_expect(__any(want), 0)) {  // (placed out of line: a taken branch over this block in every half-tile cost more than the maxima themselves)
                    const float delta = want ? t : 0.f;
                    const float alpha = __builtin_amdgcn_exp2f(-delta);
                    mref[g] += delta;
                    l[g] *= alpha;
                    static_for<64>([&](auto r_) {
                        constexpr int R = 64 * g + r_.value;
                        f2_owned_write<OWNED, R>(owned_read<OWNED, R>() * alpha);
                    });
#pragma unroll
                    for (int st = 0; st < 2; ++st)
#pragma unroll
                        for (int ee = 0; ee < 16; ++ee) {
                            // a masked score stays AT the fill value (it must keep reading as "nothing" whatever the reference does)
                            const float x = sacc[g][st][ee];
                            sacc[g][st][ee] = x < 0.5f * MASK_T ? x : x - delta;
                        }
                    const float nm = -mref[g];
#pragma unroll
                    for (int e = 0; e < 16; ++e) ninit[g][e] = nm;
                }
                tmx[0] = tmx[1] = -INFINITY;
            };
            // element stream of group g at stream position w (0 .. 32).  A lone wave pays for a transcendental beside plain vector instructions in one
            // MFMA gap far more than for either kind alone (tools/microbench/mfma_gap.hip: 2 exp2 = 33 cycles a gap, 4-6 plain = 37, 4 plain + 1 exp2 = 48),
            // so the kinds alternate: even positions take the exp2 of elements w, w + 1; odd positions their row-sum adds and their bf16 pack.
            auto stream = [&](auto gc, auto wc) {
                constexpr int g = decltype(gc)::value, w = decltype(wc)::value;
                if constexpr (w >= 0 && w < 32 && (w & 1) == 0) {
                    if constexpr (w == 0) ps[g][0] = ps[g][1] = 0.f;
                    constexpr int st = w / 16, ee = w % 16;
                    if constexpr (!(F2_ABL & 32)) {
                        sacc[g][st][ee] = f2_exp2(sacc[g][st][ee]);
                        sacc[g][st][ee + 1] = f2_exp2(sacc[g][st][ee + 1]);
                    }
                }
                if constexpr (w >= 1 && w < 32 && (w & 1) == 1) {
                    constexpr int e = w - 1, st = e / 16, ee = e % 16;
                    if constexpr (!(F2_ABL & 64)) {
                        ps[g][0] = f2_add(ps[g][0], sacc[g][st][ee]);
                        ps[g][1] = f2_add(ps[g][1], sacc[g][st][ee + 1]);
                    }
                    if constexpr (!(F2_ABL & 128)) pw[g][e / 2] = f2_pack(sacc[g][st][ee], sacc[g][st][ee + 1]);
                }
                if constexpr (w == 32) l[g] += ps[g][0] + ps[g][1];
            };
            // maxima of sub-tile st of group g: pairs 3 j, 3 j + 1, 3 j + 2 (the third of them exists for j < 2)
            auto max_trio = [&](auto gc, auto bndc, auto stc, auto jc, unsigned mset, unsigned zset, bool zany) {
                constexpr int j = decltype(jc)::value;
                if constexpr (F2_ABL & (256 | 1024)) return;
                max_pair(gc, bndc, stc, std::integral_constant<int, 3 * j>{}, mset, zset, zany);
                max_pair(gc, bndc, stc, std::integral_constant<int, 3 * j + 1>{}, mset, zset, zany);
                if constexpr (3 * j + 2 < 8) max_pair(gc, bndc, stc, std::integral_constant<int, 3 * j + 2>{}, mset, zset, zany);
            };

            for (int kt = 0; kt < ntiles; ++kt) {
                // tiles `step` and `step + 1` have landed (mine, then everybody's); at most one younger tile stays in flight
                [[maybe_unused]] const unsigned long long t_w = PROF_T();
                if (issued >= step + 3) f2_wait_vmcnt<PIECES>(); else f2_wait_vmcnt<0>();
                PROF_ADD(1, t_w);
                [[maybe_unused]] const unsigned long long t_b = PROF_T();
                __builtin_amdgcn_s_barrier();
                PROF_ADD(2, t_b);
                [[maybe_unused]] const unsigned long long t_i = PROF_T();
                issue_next();  // tile step + 3 into the stage of tile step - 1, whose last reader passed this barrier
                PROF_ADD(3, t_i);
                if (prof_on) prof_acc[9] += 1;
                [[maybe_unused]] const unsigned long long t_s = PROF_T();
                const int st_cur = cstage, st_nxt = cstage == NST - 1 ? 0 : cstage + 1;
                cstage = st_nxt;
                ++step;
                if (kt >= nt_w) continue;
                const bool last = kt + 1 == nt_w;
                auto kbits_of = [&](int t) -> unsigned long long {
                    const int nv = S - t * 64;
                    if (!key_mask) return nv >= 64 ? ~0ull : ((1ull << nv) - 1ull);
                    const unsigned long long w = kmw[t];
                    return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(w >> 32)) << 32) | (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)w);
                };
                auto bnd_of = [&](int t, unsigned long long kb) { return (kb != ~0ull) || it.allt || (causal && t * 64 + 63 > qw); };
                const unsigned long long kb_cur = kbits_of(kt);
                const unsigned long long kb_nxt = last ? ~0ull : kbits_of(kt + 1);
                const int koff = st_cur * STAGE, voff = koff + C::TILE, koff_n = st_nxt * STAGE;
                int vkx[KS], vv[DT];
                const int dko = koff_n - koff;  // the next tile's K image: the XOR of a k-step touches bits 5-7 only, stage offsets are multiples of 32 KiB
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) vkx[ks] = (lo.row + koff) ^ (ks << 5);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) vv[dt] = lo.col[dt] + voff;
                const bool zany = it.allt != 0;
                PROF_ADD(4, t_s);

                [[maybe_unused]] const unsigned long long t_p = PROF_T();
                if (kt == 0) {
                    // first tile of the block: S(0, group 0), its maxima, the decision and stream positions 0 .. 6 -- what slots 32 .. 63 of a previous tile
                    // would have left behind
                    unsigned ms, zs;
                    tile_masks(kt, kb_cur, ms, zs);
                    bf16x8 f0[RING] = {};
                    static_for<PD>([&](auto ic) { f0[ic.value % RING] = lds_frag(smem, vkx[ic.value % 8], (ic.value / 8) * 32 * C::ROWB); });
                    static_for<16>([&](auto ic) {
                        constexpr int i = decltype(ic)::value, st = i / 8, ks = i % 8;
                        if constexpr (i + PD < 16) f0[(i + PD) % RING] = lds_frag(smem, vkx[(i + PD) % 8], ((i + PD) / 8) * 32 * C::ROWB);
                        if constexpr (ks == 0) f2_mfma_init<OWNED, 128 + 4 * ks>(sacc[0][st], f0[i % RING], ninit[0]);
                        else mfma_ownedB<OWNED, 128 + 4 * ks, false>(sacc[0][st], f0[i % RING]);
                        __builtin_amdgcn_sched_barrier(0);
                    });
                    PROF_ADD(21, t_p);
                    tiles_settle(sacc[0][0], sacc[0][1]);
                    tmx[0] = tmx[1] = -INFINITY;
                    static_for<2>([&](auto stc) { static_for<8>([&](auto kc) { max_pair(std::integral_constant<int, 0>{}, std::true_type{}, stc, kc, ms, zs, zany); }); });
                    decide(std::integral_constant<int, 0>{});
                    static_for<7>([&](auto wc) { stream(std::integral_constant<int, 0>{}, wc); });
                }
                PROF_ADD(5, t_p);

                // the tile: 64 slots (LAST: slots 0 .. 31, the rest of group 1's stream, then its 16 output products)
                auto tile_body = [&](auto bndc, auto lastc) {
                    constexpr bool BND = decltype(bndc)::value, LAST = decltype(lastc)::value;
                    constexpr int NI = LAST ? 48 : 64;
#ifdef F2_FILL_PLAIN
                    float fill[12] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 0.f, 0.f, 0.f, 0.f};
#endif
                    TrHalves fr[RING] = {};  // one ring for both kinds of fragment: a ds_read_b128 or the two halves of a transposing read
                    unsigned ms1 = 0, zs1 = 0, ms0 = 0, zs0 = 0;
                    if constexpr (BND) {
                        tile_masks(kt, kb_cur, ms1, zs1);
                        if constexpr (!LAST) tile_masks(kt + 1, kb_nxt, ms0, zs0);
                    }
                    // item kinds: 0 = S^T product (one ds_read_b128), 1 = O^T product (two transposing reads)
                    auto kind = [](int i) constexpr { return LAST ? (i < 16 ? 0 : 1) : ((i / 16) & 1); };
                    auto load = [&](auto ic) {
                        constexpr int i = decltype(ic)::value;
                        if constexpr ((F2_ABL & 2) && i >= RING) return;
                        if constexpr (i < NI) {
                            if constexpr (kind(i) == 0) {
                                constexpr int j = i % 16, st = j / 8, ks = j % 8;
                                const bf16x8 x = lds_frag(smem, i < 16 ? vkx[ks] : vkx[ks] + dko, st * 32 * C::ROWB);
                                fr[i % RING].lo = __builtin_shufflevector(x, x, 0, 1, 2, 3);
                                fr[i % RING].hi = __builtin_shufflevector(x, x, 4, 5, 6, 7);
                            } else {
                                constexpr int j = i % 16, st = j / 8, s = (j / 4) % 2, dt = j % 4, imm = (st * 32 + 16 * s) * C::ROWB;
                                tr_issue<imm, imm + 8 * C::ROWB>(fr[i % RING], lds0 + vv[dt], lds0 + vv[dt]);
                            }
                        }
                    };
                    static_for<PD>([&](auto ic) { load(ic); });
                    static_for<NI>([&](auto ic) {
                        constexpr int i = decltype(ic)::value;
#ifdef F2_FILL_PLAIN
                        float* fillp = fill;  // (named here: clang does not capture a variable that only a discarded branch of a nested generic lambda uses)
#endif
                        load(std::integral_constant<int, i + PD>{});
                        if constexpr (kind(i) == 0) {
                            constexpr int j = i % 16, st = j / 8, ks = j % 8;
                            constexpr int g = i < 16 ? 1 : 0;  // slots 0-15: group 1 of this tile; slots 32-47: group 0 of the next
                            const bf16x8 a = __builtin_shufflevector(fr[i % RING].lo, fr[i % RING].hi, 0, 1, 2, 3, 4, 5, 6, 7);
                            if constexpr (F2_ABL & 8) {
                                asm volatile("" : "+v"(sacc[g][st]) : "v"(a));
                            } else if constexpr (ks == 0) f2_mfma_init<OWNED, 128 + 32 * g + 4 * ks>(sacc[g][st], a, ninit[g]);
                            else mfma_ownedB<OWNED, 128 + 32 * g + 4 * ks, false>(sacc[g][st], a);
                        } else {
                            constexpr int j = i % 16, st = j / 8, s = (j / 4) % 2, dt = j % 4;
                            constexpr int g = i < 32 ? 0 : 1;  // slots 16-31: group 0; the last 16: group 1
                            // LDS instructions issued after this fragment's two reads: the items requested since
                            constexpr int younger = [&]() constexpr {
                                int n = 0;
                                for (int jj = i + 1; jj <= i + PD && jj < NI; ++jj) n += kind(jj) == 0 ? 1 : 2;
                                return n < 15 ? n : 15;
                            }();
                            const u32x4 w = {pw[g][4 * (2 * st + s)], pw[g][4 * (2 * st + s) + 1], pw[g][4 * (2 * st + s) + 2], pw[g][4 * (2 * st + s) + 3]};
                            if constexpr (F2_ABL & 2) {
                                const bf16x8 a = __builtin_shufflevector(fr[i % RING].lo, fr[i % RING].hi, 0, 1, 2, 3, 4, 5, 6, 7);
                                if constexpr (!(F2_ABL & 4)) mfma_owned<OWNED, 64 * g + 16 * dt, false>(a, __builtin_bit_cast(bf16x8, w));
                            } else if constexpr (F2_ABL & 4) {
                                const bf16x8 a = tr_wait<younger>(fr[i % RING]);
                                asm volatile("" ::"v"(a), "v"(w));
                            } else
                                mfma_owned<OWNED, 64 * g + 16 * dt, false>(tr_wait<younger>(fr[i % RING]), __builtin_bit_cast(bf16x8, w));
                        }
                        if constexpr (F2_ABL & 1) {
#ifdef F2_FILL_PLAIN
                            // timing experiment: independent fillers in every gap instead of the softmax
                            f2_fillers<F2_FILL_PLAIN, F2_FILL_EXP>(fillp);
#endif
                            __builtin_amdgcn_sched_barrier(0);
                            return;
                        }
                        // ---- the vector work riding on this slot: exp2 pairs in the odd slots, everything else in the even ones
                        // group 0 of this tile: stream positions 7 .. 32 in slots 0 .. 25
                        if constexpr (i <= 25) stream(std::integral_constant<int, 0>{}, std::integral_constant<int, i + 7>{});
                        // group 1 of this tile: maxima in slots 10 .. 22 (its scores are complete after slot 7 / 15), decision 24, stream from 25
                        if constexpr (i == 10 || i == 12 || i == 14)
                            max_trio(std::integral_constant<int, 1>{}, bndc, std::integral_constant<int, 0>{}, std::integral_constant<int, (i - 10) / 2>{}, ms1, zs1, zany);
                        if constexpr (i == 18 && !(F2_ABL & 2048)) tiles_settle(sacc[1][0], sacc[1][1]);
                        if constexpr (i == 18 || i == 20 || i == 22)
                            max_trio(std::integral_constant<int, 1>{}, bndc, std::integral_constant<int, 1>{}, std::integral_constant<int, (i - 18) / 2>{}, ms1, zs1, zany);
                        if constexpr (i == 24) decide(std::integral_constant<int, 1>{});
                        if constexpr (i >= 25 && (LAST ? i <= 31 : i <= 57)) stream(std::integral_constant<int, 1>{}, std::integral_constant<int, i - 25>{});
                        if constexpr (LAST && i == 31) {
                            // nothing follows the block's last tile: the rest of group 1's stream has no product to hide behind
                            __builtin_amdgcn_sched_barrier(0);
                            static_for<26>([&](auto wc) { stream(std::integral_constant<int, 1>{}, std::integral_constant<int, wc.value + 7>{}); });
                        }
                        if constexpr (!LAST) {
                            // group 0 of the next tile: maxima in slots 42 .. 54, decision 56, stream positions 0 .. 6 in slots 57 .. 63
                            if constexpr (i == 42 || i == 44 || i == 46)
                                max_trio(std::integral_constant<int, 0>{}, bndc, std::integral_constant<int, 0>{}, std::integral_constant<int, (i - 42) / 2>{}, ms0, zs0, zany);
                            if constexpr (i == 50 && !(F2_ABL & 2048)) tiles_settle(sacc[0][0], sacc[0][1]);
                            if constexpr (i == 50 || i == 52 || i == 54)
                                max_trio(std::integral_constant<int, 0>{}, bndc, std::integral_constant<int, 1>{}, std::integral_constant<int, (i - 50) / 2>{}, ms0, zs0, zany);
                            if constexpr (i == 56) decide(std::integral_constant<int, 0>{});
                            if constexpr (i >= 57) stream(std::integral_constant<int, 0>{}, std::integral_constant<int, i - 57>{});
                        }
                        __builtin_amdgcn_sched_barrier(0);  // keep this slot's vector work where it is written
                    });
                };
                const bool bnd = bnd_of(kt, kb_cur) || (!last && bnd_of(kt + 1, kb_nxt));
                [[maybe_unused]] const unsigned long long t_body = PROF_T();
                if (last) tile_body(std::true_type{}, std::true_type{});
                else if (bnd) tile_body(std::true_type{}, std::false_type{});
                else tile_body(std::false_type{}, std::false_type{});
#if ATTN_ABL & 16
                asm volatile("s_nop 0" ::: "memory");
                if (prof_on) { prof_acc[last ? 13 : bnd ? 14 : 6] += __builtin_readcyclecounter() - t_body; prof_acc[last ? 15 : 10] += 1; }
#endif
            }
            [[maybe_unused]] const unsigned long long t_e = PROF_T();

            // ---- the next block's Q rows: requested now, consumed after this block's output has been written
            {
                bool more = true;
                Item itn = it;
                int qbn = qb - 1;
                if (qbn < it.qb_lo) {
                    more = item + G < nitems;
                    if (more) {
                        itn = decode(item + G);
                        qbn = itn.qb_hi - 1;
                    }
                }
                if (more) request_q(itn, qbn);
            }
            PROF_ADD(16, t_e);
            [[maybe_unused]] const unsigned long long t_e1 = PROF_T();
            // ---- block epilogue: O / l as whole 16-byte pieces (one v_permlane32_swap per word pairs the half-waves' 8-byte pieces), lse
            owned_settle<OWNED>();
            PROF_ADD(17, t_e1);
            static_for<2>([&](auto gc) {
                constexpr int g = decltype(gc)::value;
                [[maybe_unused]] const unsigned long long t_e2 = PROF_T();
                const int hq = hkv * rep + 2 * it.pr + g;
                float ls = l[g];
                const auto rr = __builtin_amdgcn_permlane32_swap(__float_as_uint(ls), __float_as_uint(ls), false, false);
                ls = __uint_as_float(rr[0]) + __uint_as_float(rr[1]);
                const float inv = 1.0f / ls;
                bf16_t* orow = o + ((int64_t)b * S + qg) * ldo + (int64_t)hq * D + 8 * (lane >> 5);
                static_for<DT * 2>([&](auto i) {
                    constexpr int dt = i.value / 2, gp = i.value % 2, r0 = 64 * g + 16 * dt + 8 * gp;
                    const unsigned a0 = pack_bf2(owned_read<OWNED, r0>() * inv, owned_read<OWNED, r0 + 1>() * inv);
                    const unsigned a1 = pack_bf2(owned_read<OWNED, r0 + 2>() * inv, owned_read<OWNED, r0 + 3>() * inv);
                    const unsigned b0 = pack_bf2(owned_read<OWNED, r0 + 4>() * inv, owned_read<OWNED, r0 + 5>() * inv);
                    const unsigned b1 = pack_bf2(owned_read<OWNED, r0 + 6>() * inv, owned_read<OWNED, r0 + 7>() * inv);
                    const auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false), s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                    const u32x4 w = {s0[0], s1[0], s0[1], s1[1]};
                    if (qvalid) *reinterpret_cast<u32x4*>(orow + dt * 32 + 16 * gp) = w;
                });
                // a row whose visible keys are all padding reports the fill value as its maximum, as the first-generation kernel does
                const float mout = qrow ? MASK_T : mref[g];
                if (qvalid && lane < 32) lse[((int64_t)b * Hq + hq) * S + qg] = (mout + __builtin_amdgcn_logf(ls)) * LN2;
                PROF_ADD(18 + g, t_e2);
            });
            PROF_ADD(8, t_e);
        }
    }
#if ATTN_ABL & 16
    PROF_ADD(0, t_wg);
    if (prof_on && (blockIdx.x & 15) == 3)  // one workgroup in 16 reports
        for (int i = 0; i < 32; ++i) atomicAdd(&g_prof2[i], prof_acc[i]);
#endif
}

}  // namespace

#if ATTN_ABL & 16
extern "C" int mi355_debug_prof2(unsigned long long* out, int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof2), sizeof(unsigned long long) * 32) != hipSuccess) return 2;
    if (reset) {
        unsigned long long z[32] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_prof2), z, sizeof(z)) != hipSuccess) return 3;
    }
    return 0;
}
#endif

// Host entry of the second-generation forward; returns -1 when the shape is not its (the caller then launches the first-generation kernel).
int mi355_attn_fwd2_launch(int B, int S, int Hq, int Hkv, int D, const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                           void* o, int64_t ldo, float* lse, const uint8_t* key_mask, int causal, float scale_log2, hipStream_t s) {
    const int rep = Hq / Hkv;
    if (D != 128 || (rep & 1) || S > 64 * ATTN_MAX_TILES) return -1;
    if (key_mask && ((uintptr_t)key_mask & 3)) return -1;  // the per-item flag is read as an aligned scalar word
    if (ldk * 2 * 64 >= (1ll << 31) || ldv * 2 * 64 >= (1ll << 31)) return -1;  // tile-relative byte offsets are 32-bit
    const int pairs = rep / 2, nqb = (S + 127) / 128;
    const int64_t heads = (int64_t)B * Hkv * pairs;
    const int bpw = heads >= F2_CUS ? nqb : 1;  // enough (batch, head pair)s to fill the chip: one item = a whole head pair, K / V stay in L2 for its blocks
    const int64_t nitems = heads * ((nqb + bpw - 1) / bpw);
    if (nitems >= 0x7fffffffLL) return -1;
    const int grid = (int)(nitems < F2_CUS ? nitems : F2_CUS);
    hipLaunchKernelGGL(attn_fwd2_kernel<128>, dim3(grid), dim3(256), 0, s, B, S, Hq, Hkv, (const bf16_t*)q, ldq, (const bf16_t*)k, ldk, (const bf16_t*)v, ldv,
                       (bf16_t*)o, ldo, lse, key_mask, causal, scale_log2, bpw, (int)nitems);
    return 0;
}

// C entry of the experiment (tools/experimental/exp.py): arguments of mi355_attn_fwd; -1 = not this kernel's shape
extern "C" int mi355_exp_attn_fwd2(int B, int S, int Hq, int Hkv, int D, const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                                   void* o, int64_t ldo, float* lse, const uint8_t* key_mask, int causal, float scale, void* stream) {
    const int rc = mi355_attn_fwd2_launch(B, S, Hq, Hkv, D, q, ldq, k, ldk, v, ldv, o, ldo, lse, key_mask, causal, scale * LOG2E, (hipStream_t)stream);
    if (rc == 0) MI355_LAUNCH_CHECK("mi355_exp_attn_fwd2");
    return rc;
}
