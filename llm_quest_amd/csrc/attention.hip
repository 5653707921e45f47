// Flash-style attention forward + backward for gfx950, token-major operands (no head transposes in HBM).
//
// Forward (and the dQ pass of backward) put the QUERY on the MFMA lane:  S^T = K Q^T  via
// mfma_f32_32x32x16_bf16(A = K rows from LDS, B = Q rows held in registers), so each lane owns one query column,
// the online-softmax row statistics are lane-local (one cross-half exchange), and the fp32 accumulator tile of
// P^T is, after a pairwise bf16 pack, directly the B operand of  O^T += V^T P^T  (no LDS round trip for P).
// V (and K in the dQ pass) is consumed K-strided through ds_read_b64_tr_b16 from a row-major LDS image.
// K/V tiles (64 keys) arrive by LDS-DMA (buffer_load ... lds) with the bank swizzle on the source address,
// double-buffered: the next tile's DMA is in flight under the current tile's MFMAs.
//
// Masking follows the reference (qwen3_attention.py:130-142): masked scores take a FINITE fill value, so a row
// whose visible keys are all masked degenerates to uniform attention over all S keys exactly as upstream; keys
// beyond S do not exist and get -inf.  Scores stay in fp32 (the reference rounds them to bf16 twice).
#include "common.h"

namespace {

constexpr float MASK_T = -2.0e38f;  // finite "masked" score in the log2 domain
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
constexpr unsigned OOB = 0x80000000u;

template <int D>
struct Cfg {
    static constexpr int ROWB = D * 2;            // bytes per tile row
    static constexpr int CH = ROWB / 16;          // 16-byte chunks per row
    static constexpr int KS = D / 16;             // k-steps of the QK^T product
    static constexpr int DT = D / 32;             // 32-row tiles of O^T
    static constexpr int TILE = 64 * ROWB;        // bytes of one 64-key tile
    static constexpr int RPP = 1024 / ROWB;       // rows per 1-KiB DMA piece
    static constexpr int PPW = TILE / 1024 / 4;   // pieces per wave per tile
};

// swizzles (chunk index XOR) -- row-read image (32x32 A-operand pattern) and transposed-read image
template <int D> __device__ __forceinline__ int swz_row(int chunk, int row) { return D == 128 ? chunk ^ (row & 15) : chunk ^ ((row >> 1) & 7); }
template <int D> __device__ __forceinline__ int swz_tr(int chunk, int row) { return D == 128 ? chunk ^ ((row & 3) << 2) : chunk ^ (((row >> 1) & 1) << 2); }

// DMA one 64-row tile (rows = tokens tok0.., D contiguous elements at column col0) into LDS; TRIMG picks the swizzle
template <int D, bool TRIMG>
__device__ __forceinline__ void dma_tile(const bf16_t* base, int64_t ld, int rows_valid, char* lds, int wave, int lane) {
    using C = Cfg<D>;
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(base), 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int j = 0; j < C::PPW; ++j) {
        const int pi = wave * C::PPW + j;
        const int row = pi * C::RPP + lane / C::CH;
        const int pos = lane % C::CH;
        const int c = TRIMG ? swz_tr<D>(pos, row) : swz_row<D>(pos, row);
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

__device__ __forceinline__ int acc_row(int e, int lane) { return (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5); }

// ================================================================================================ forward
template <int D>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(int B, int S, int Hq, int Hkv, const bf16_t* __restrict__ q, int64_t ldq,
                                                          const bf16_t* __restrict__ k, int64_t ldk, const bf16_t* __restrict__ v,
                                                          int64_t ldv, bf16_t* __restrict__ o, int64_t ldo, float* __restrict__ lse,
                                                          const uint8_t* __restrict__ key_mask, int causal, float scale_log2) {
    using C = Cfg<D>;
    __shared__ __attribute__((aligned(16))) char smem[4 * C::TILE];  // 2 stages x (K row image, V tr image)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nqb = (S + 127) / 128;
    // heaviest (latest) query blocks first under the causal mask
    const int qb = nqb - 1 - (int)(blockIdx.x % nqb);
    const int bh = blockIdx.x / nqb;
    const int hq = bh % Hq, b = bh / Hq;
    const int hkv = hq / (Hq / Hkv);
    const int q0 = qb * 128;
    const int qg = q0 + wave * 32 + (lane & 31);
    const bool qvalid = qg < S;

    bf16x8 qf[C::KS];
    load_rows_frag<D>(q + (int64_t)b * S * ldq + (int64_t)hq * D, ldq, qg, qvalid, lane, qf);

    const bf16_t* kbase = k + (int64_t)b * S * ldk + (int64_t)hkv * D;
    const bf16_t* vbase = v + (int64_t)b * S * ldv + (int64_t)hkv * D;
    const int ntiles_all = (S + 63) / 64;
    int ntiles = causal ? min(ntiles_all, (q0 + 127) / 64 + 1) : ntiles_all;

    f32x16 oacc[C::DT];
#pragma unroll
    for (int i = 0; i < C::DT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[i][e] = 0.f;
    float m = -INFINITY, l = 0.f;

    auto issue = [&](int kt, int stage) {
        char* ks_ = smem + stage * 2 * C::TILE;
        dma_tile<D, false>(kbase + (int64_t)kt * 64 * ldk, ldk, S - kt * 64, ks_, wave, lane);
        dma_tile<D, true>(vbase + (int64_t)kt * 64 * ldv, ldv, S - kt * 64, ks_ + C::TILE, wave, lane);
    };

    const LaneOff<D> lo = lane_offsets<D>(lane);
    issue(0, 0);
    for (int kt = 0; kt < ntiles; ++kt) {
        __syncthreads();
        if (kt + 1 < ntiles) issue(kt + 1, (kt + 1) & 1);
        const int koff = (kt & 1) * 2 * C::TILE, voff = koff + C::TILE;
        // key-padding bits of this tile (1 = real token); keys beyond S read as padding here and are removed below
        unsigned long long kbits = ~0ull;
        if (key_mask) {
            const int kg = kt * 64 + lane;
            kbits = __ballot(kg < S && key_mask[(int64_t)b * S + kg] != 0);
        }
        // a wave whose 32 queries all precede this tile has nothing visible here (unless a row is still fully masked)
        const bool wave_active = !causal || (kt * 64 <= q0 + wave * 32 + 31) || __any(m == MASK_T);
        if (wave_active) {
            int vkx[C::KS], vv[C::DT];
#pragma unroll
            for (int ks = 0; ks < C::KS; ++ks) vkx[ks] = (lo.row + koff) ^ (ks << 5);
#pragma unroll
            for (int dt = 0; dt < C::DT; ++dt) vv[dt] = lo.col[dt] + voff;
            f32x16 sacc[2];
#pragma unroll
            for (int st = 0; st < 2; ++st) {
#pragma unroll
                for (int e = 0; e < 16; ++e) sacc[st][e] = 0.f;
#pragma unroll
                for (int ks = 0; ks < C::KS; ++ks)
                    sacc[st] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_frag(smem, vkx[ks], st * 32 * C::ROWB), qf[ks], sacc[st], 0, 0, 0);
            }
            // masks are needed only on the diagonal / tail / padded tiles (wave-uniform test)
            const bool boundary = (kbits != ~0ull) || (kt * 64 + 64 > S) || (causal && kt * 64 + 63 > q0 + wave * 32);
            float mnew;
            if (boundary) {
                float tmax = -INFINITY;
#pragma unroll
                for (int st = 0; st < 2; ++st)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int kl = st * 32 + acc_row(e, lane);
                        const int kg = kt * 64 + kl;
                        float t = sacc[st][e] * scale_log2;
                        const bool masked = (causal && kg > qg) || !((kbits >> kl) & 1ull);
                        t = masked ? MASK_T : t;
                        t = kg < S ? t : -INFINITY;
                        sacc[st][e] = t;
                        tmax = fmaxf(tmax, t);
                    }
                tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
                mnew = fmaxf(m, tmax);
#pragma unroll
                for (int st = 0; st < 2; ++st)
#pragma unroll
                    for (int e = 0; e < 16; ++e) sacc[st][e] = __builtin_amdgcn_exp2f(sacc[st][e] - mnew);
            } else {
                float rmax = sacc[0][0];
#pragma unroll
                for (int st = 0; st < 2; ++st)
#pragma unroll
                    for (int e = 0; e < 16; ++e) rmax = fmaxf(rmax, sacc[st][e]);
                rmax = fmaxf(rmax, __shfl_xor(rmax, 32, 64));
                mnew = fmaxf(m, rmax * scale_log2);
#pragma unroll
                for (int st = 0; st < 2; ++st)
#pragma unroll
                    for (int e = 0; e < 16; ++e) sacc[st][e] = __builtin_amdgcn_exp2f(fmaf(sacc[st][e], scale_log2, -mnew));
            }
            float psum = 0.f;
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int e = 0; e < 16; ++e) psum += sacc[st][e];
            if (__any(mnew != m)) {  // the running max moved for some row of this wave: rescale (else alpha == 1 exactly)
                const float alpha = __builtin_amdgcn_exp2f(m - mnew);
                l *= alpha;
#pragma unroll
                for (int i = 0; i < C::DT; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e) oacc[i][e] *= alpha;
                m = mnew;
            }
            l += psum;
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const bf16x8 pf = pack_frag(sacc[st], s);
#pragma unroll
                    for (int dt = 0; dt < C::DT; ++dt)
                        oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_frag_tr<D>(smem, vv[dt], (st * 32 + 16 * s) * C::ROWB), pf, oacc[dt], 0, 0, 0);
                }
        }
        // reference semantics for rows whose visible keys are all padding: keep going over the causally hidden tiles
        if (key_mask && causal && kt + 1 == ntiles && ntiles < ntiles_all) {
            if (__syncthreads_or(m == MASK_T)) {
                ntiles = ntiles_all;
                issue(kt + 1, (kt + 1) & 1);  // that stage was last read one iteration ago: free
            }
        }
    }

    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    if (qvalid) {
        bf16_t* orow = o + ((int64_t)b * S + qg) * ldo + (int64_t)hq * D;
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d = dt * 32 + 8 * g4 + 4 * (lane >> 5);
                u32x2 pk = {pack_bf2(oacc[dt][4 * g4] * inv, oacc[dt][4 * g4 + 1] * inv),
                            pack_bf2(oacc[dt][4 * g4 + 2] * inv, oacc[dt][4 * g4 + 3] * inv)};
                *reinterpret_cast<u32x2*>(orow + d) = pk;
            }
        if (lane < 32) lse[((int64_t)b * Hq + hq) * S + qg] = (m + __builtin_amdgcn_logf(l)) * LN2;
    }
}

// ================================================================================================ backward
// delta[b,h,q] = sum_d dO[q,d] * O[q,d]   (one wave per (token, head))
template <int D>
__global__ __launch_bounds__(256) void attn_delta_kernel(int B, int S, int Hq, const bf16_t* __restrict__ o, int64_t ldo,
                                                         const bf16_t* __restrict__ d_o, int64_t lddo, float* __restrict__ delta) {
    const int lane = threadIdx.x & 63;
    const int64_t total = (int64_t)B * S * Hq;
    for (int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); item < total; item += (int64_t)gridDim.x * 4) {
        const int64_t tok = item / Hq;
        const int h = (int)(item - tok * Hq);
        float s = 0.f;
        for (int c = lane * 2; c < D; c += 128) {
            const unsigned a = *reinterpret_cast<const unsigned*>(o + tok * ldo + (int64_t)h * D + c);
            const unsigned g = *reinterpret_cast<const unsigned*>(d_o + tok * lddo + (int64_t)h * D + c);
            s += __uint_as_float(a << 16) * __uint_as_float(g << 16) + __uint_as_float(a & 0xffff0000u) * __uint_as_float(g & 0xffff0000u);
        }
        s = wave_sum(s);
        if (lane == 0) {
            const int64_t bb = tok / S, sq = tok - bb * S;
            delta[(bb * Hq + h) * S + sq] = s;
        }
    }
}

// ---- dQ pass: query on the lane, exactly the forward's structure with three products per key sub-tile:
//   S^T = K Q^T ;  dP^T = V dO^T ;  dQ^T += K^T dS^T   with  P^T = exp2(S^T*c - lse2[q]),  dS^T = P^T * (dP^T - delta[q]).
template <int D>
__global__ __launch_bounds__(256, 1) void attn_bwd_dq_kernel(int B, int S, int Hq, int Hkv, const bf16_t* __restrict__ q, int64_t ldq,
                                                             const bf16_t* __restrict__ k, int64_t ldk, const bf16_t* __restrict__ v,
                                                             int64_t ldv, const bf16_t* __restrict__ d_o, int64_t lddo,
                                                             const float* __restrict__ lse, const float* __restrict__ delta,
                                                             bf16_t* __restrict__ dq, int64_t lddq, const uint8_t* __restrict__ key_mask,
                                                             int causal, float scale, float scale_log2) {
    using C = Cfg<D>;
    __shared__ __attribute__((aligned(16))) char smem[6 * C::TILE];  // 2 stages x (K row, K tr, V row)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nqb = (S + 127) / 128;
    const int qb = nqb - 1 - (int)(blockIdx.x % nqb);
    const int bh = blockIdx.x / nqb;
    const int hq = bh % Hq, b = bh / Hq;
    const int hkv = hq / (Hq / Hkv);
    const int q0 = qb * 128;
    const int qg = q0 + wave * 32 + (lane & 31);
    const bool qvalid = qg < S;

    bf16x8 qf[C::KS], dof[C::KS];
    load_rows_frag<D>(q + (int64_t)b * S * ldq + (int64_t)hq * D, ldq, qg, qvalid, lane, qf);
    load_rows_frag<D>(d_o + (int64_t)b * S * lddo + (int64_t)hq * D, lddo, qg, qvalid, lane, dof);
    const float lse2 = qvalid ? lse[((int64_t)b * Hq + hq) * S + qg] * LOG2E : 0.f;
    const float dlt = qvalid ? delta[((int64_t)b * Hq + hq) * S + qg] : 0.f;

    const bf16_t* kbase = k + (int64_t)b * S * ldk + (int64_t)hkv * D;
    const bf16_t* vbase = v + (int64_t)b * S * ldv + (int64_t)hkv * D;
    const int ntiles_all = (S + 63) / 64;
    const int ntiles = causal ? min(ntiles_all, (q0 + 127) / 64 + 1) : ntiles_all;

    f32x16 dqacc[C::DT];
#pragma unroll
    for (int i = 0; i < C::DT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) dqacc[i][e] = 0.f;

    auto issue = [&](int kt, int stage) {
        char* st_ = smem + stage * 3 * C::TILE;
        dma_tile<D, false>(kbase + (int64_t)kt * 64 * ldk, ldk, S - kt * 64, st_, wave, lane);
        dma_tile<D, true>(kbase + (int64_t)kt * 64 * ldk, ldk, S - kt * 64, st_ + C::TILE, wave, lane);
        dma_tile<D, false>(vbase + (int64_t)kt * 64 * ldv, ldv, S - kt * 64, st_ + 2 * C::TILE, wave, lane);
    };

    const LaneOff<D> lo = lane_offsets<D>(lane);
    issue(0, 0);
    for (int kt = 0; kt < ntiles; ++kt) {
        __syncthreads();
        if (kt + 1 < ntiles) issue(kt + 1, (kt + 1) & 1);
        const int kroff = (kt & 1) * 3 * C::TILE, ktoff = kroff + C::TILE, vroff = kroff + 2 * C::TILE;
        unsigned long long kbits = ~0ull;
        if (key_mask) {
            const int kg = kt * 64 + lane;
            kbits = __ballot(kg < S && key_mask[(int64_t)b * S + kg] != 0);
        }
        const bool wave_active = !causal || (kt * 64 <= q0 + wave * 32 + 31);
        if (!wave_active) continue;
        const bool boundary = (kbits != ~0ull) || (kt * 64 + 64 > S) || (causal && kt * 64 + 63 > q0 + wave * 32);
        int vkx[C::KS], vvx[C::KS], vkt[C::DT];
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) {
            vkx[ks] = (lo.row + kroff) ^ (ks << 5);
            vvx[ks] = (lo.row + vroff) ^ (ks << 5);
        }
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) vkt[dt] = lo.col[dt] + ktoff;
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            f32x16 sacc, pacc;
#pragma unroll
            for (int e = 0; e < 16; ++e) { sacc[e] = 0.f; pacc[e] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < C::KS; ++ks) {
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_frag(smem, vkx[ks], st * 32 * C::ROWB), qf[ks], sacc, 0, 0, 0);
                pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_frag(smem, vvx[ks], st * 32 * C::ROWB), dof[ks], pacc, 0, 0, 0);
            }
            if (boundary) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int kl = st * 32 + acc_row(e, lane);
                    const int kg = kt * 64 + kl;
                    const bool masked = (causal && kg > qg) || !((kbits >> kl) & 1ull) || kg >= S;
                    const float p = masked ? 0.f : __builtin_amdgcn_exp2f(fmaf(sacc[e], scale_log2, -lse2));
                    sacc[e] = p * (pacc[e] - dlt) * scale;  // dS^T (gradient w.r.t. the raw QK^T product)
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float p = __builtin_amdgcn_exp2f(fmaf(sacc[e], scale_log2, -lse2));
                    sacc[e] = p * (pacc[e] - dlt) * scale;
                }
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 dsf = pack_frag(sacc, s);
#pragma unroll
                for (int dt = 0; dt < C::DT; ++dt)
                    dqacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_frag_tr<D>(smem, vkt[dt], (st * 32 + 16 * s) * C::ROWB), dsf, dqacc[dt], 0, 0, 0);
            }
        }
    }
    if (qvalid) {
        bf16_t* row = dq + ((int64_t)b * S + qg) * lddq + (int64_t)hq * D;
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d = dt * 32 + 8 * g4 + 4 * (lane >> 5);
                u32x2 pk = {pack_bf2(dqacc[dt][4 * g4], dqacc[dt][4 * g4 + 1]), pack_bf2(dqacc[dt][4 * g4 + 2], dqacc[dt][4 * g4 + 3])};
                *reinterpret_cast<u32x2*>(row + d) = pk;
            }
    }
}

// ---- dK/dV pass: KEY on the lane.  One workgroup = 128 keys of one (batch, kv head); each wave owns 32 keys and keeps
// dK^T, dV^T for them in accumulators while the workgroup sweeps the group's query heads x 64-query tiles:
//   S = Q K^T (A = Q rows from LDS, B = K rows held in registers), dP = dO V^T (A = dO rows, B = V rows in registers),
//   P = exp2(S*c - lse2[q]) (row constants come from LDS), dS = P*(dP - delta[q])*scale,
//   dV^T += dO^T P  and  dK^T += Q^T dS   (A = transposed reads of the dO / Q tile, B = the accumulator tiles of P / dS).
template <int D>
__global__ __launch_bounds__(256, 1) void attn_bwd_dkv_kernel(int B, int S, int Hq, int Hkv, const bf16_t* __restrict__ q, int64_t ldq,
                                                              const bf16_t* __restrict__ k, int64_t ldk, const bf16_t* __restrict__ v,
                                                              int64_t ldv, const bf16_t* __restrict__ d_o, int64_t lddo,
                                                              const float* __restrict__ lse, const float* __restrict__ delta,
                                                              bf16_t* __restrict__ dk, int64_t lddk, bf16_t* __restrict__ dv, int64_t lddv,
                                                              const uint8_t* __restrict__ key_mask, int causal, float scale, float scale_log2) {
    using C = Cfg<D>;
    // 2 stages x (Q row, Q tr, dO row, dO tr) + 2 stages x 64 x (lse2, delta)
    __shared__ __attribute__((aligned(16))) char smem[8 * C::TILE + 2 * 64 * 8];
    float* rowc = reinterpret_cast<float*>(smem + 8 * C::TILE);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nkb = (S + 127) / 128;
    const int kb = blockIdx.x % nkb;
    const int bh = blockIdx.x / nkb;
    const int hkv = bh % Hkv, b = bh / Hkv;
    const int rep = Hq / Hkv;
    const int k0 = kb * 128;
    const int kg = k0 + wave * 32 + (lane & 31);
    const bool kvalid = kg < S;
    const bool kreal = kvalid && (key_mask == nullptr || key_mask[(int64_t)b * S + kg] != 0);

    bf16x8 kf[C::KS], vf[C::KS];
    load_rows_frag<D>(k + (int64_t)b * S * ldk + (int64_t)hkv * D, ldk, kg, kvalid, lane, kf);
    load_rows_frag<D>(v + (int64_t)b * S * ldv + (int64_t)hkv * D, ldv, kg, kvalid, lane, vf);

    f32x16 dkacc[C::DT], dvacc[C::DT];
#pragma unroll
    for (int i = 0; i < C::DT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) { dkacc[i][e] = 0.f; dvacc[i][e] = 0.f; }

    const int nqt_all = (S + 63) / 64;
    const int qt0 = causal ? k0 / 64 : 0;  // first query tile that can see any key of this block
    const int per_head = nqt_all - qt0;
    const int nit = per_head * rep;

    auto issue = [&](int it, int stage) {
        const int hq = hkv * rep + it / per_head;
        const int qt = qt0 + it % per_head;
        char* st_ = smem + stage * 4 * C::TILE;
        const bf16_t* qb_ = q + ((int64_t)b * S + (int64_t)qt * 64) * ldq + (int64_t)hq * D;
        const bf16_t* ob_ = d_o + ((int64_t)b * S + (int64_t)qt * 64) * lddo + (int64_t)hq * D;
        dma_tile<D, false>(qb_, ldq, S - qt * 64, st_, wave, lane);
        dma_tile<D, true>(qb_, ldq, S - qt * 64, st_ + C::TILE, wave, lane);
        dma_tile<D, false>(ob_, lddo, S - qt * 64, st_ + 2 * C::TILE, wave, lane);
        dma_tile<D, true>(ob_, lddo, S - qt * 64, st_ + 3 * C::TILE, wave, lane);
        if (threadIdx.x < 64) {
            const int qq = qt * 64 + threadIdx.x;
            const int64_t idx = ((int64_t)b * Hq + hq) * S + qq;
            rowc[stage * 128 + threadIdx.x] = qq < S ? lse[idx] * LOG2E : 0.f;
            rowc[stage * 128 + 64 + threadIdx.x] = qq < S ? delta[idx] : 0.f;
        }
    };

    const LaneOff<D> lo = lane_offsets<D>(lane);
    if (nit > 0) issue(0, 0);
    for (int it = 0; it < nit; ++it) {
        __syncthreads();
        if (it + 1 < nit) issue(it + 1, (it + 1) & 1);
        const int qt = qt0 + it % per_head;
        const int qroff = (it & 1) * 4 * C::TILE, qtoff = qroff + C::TILE, oroff = qroff + 2 * C::TILE, otoff = qroff + 3 * C::TILE;
        const float* rc = rowc + (it & 1) * 128;
        // queries of this tile all precede this wave's keys -> nothing visible
        if (causal && qt * 64 + 63 < k0 + wave * 32) continue;
        // masks only where the tile touches the diagonal, the sequence end, or this wave holds padded / out-of-range keys
        const bool boundary = (causal && qt * 64 < k0 + wave * 32 + 31) || (qt * 64 + 64 > S) || __any(!kreal);
        int vqx[C::KS], vox[C::KS], vqt[C::DT], vot[C::DT];
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) {
            vqx[ks] = (lo.row + qroff) ^ (ks << 5);
            vox[ks] = (lo.row + oroff) ^ (ks << 5);
        }
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) {
            vqt[dt] = lo.col[dt] + qtoff;
            vot[dt] = lo.col[dt] + otoff;
        }
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            f32x16 sacc, pacc;
#pragma unroll
            for (int e = 0; e < 16; ++e) { sacc[e] = 0.f; pacc[e] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < C::KS; ++ks) {
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_frag(smem, vqx[ks], st * 32 * C::ROWB), kf[ks], sacc, 0, 0, 0);
                pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_frag(smem, vox[ks], st * 32 * C::ROWB), vf[ks], pacc, 0, 0, 0);
            }
            // per-query constants of the 16 accumulator rows: 4 x 16-byte LDS reads each (rows 8*g4 + 4*h + 0..3)
            float l2r[16], dlr[16];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(rc + st * 32 + 8 * g4 + 4 * (lane >> 5));
                const f32x4 c = *reinterpret_cast<const f32x4*>(rc + 64 + st * 32 + 8 * g4 + 4 * (lane >> 5));
#pragma unroll
                for (int e = 0; e < 4; ++e) { l2r[4 * g4 + e] = a[e]; dlr[4 * g4 + e] = c[e]; }
            }
            if (boundary) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int qq = qt * 64 + st * 32 + acc_row(e, lane);
                    const bool masked = (causal && kg > qq) || !kreal || qq >= S;
                    const float p = masked ? 0.f : __builtin_amdgcn_exp2f(fmaf(sacc[e], scale_log2, -l2r[e]));
                    sacc[e] = p;
                    pacc[e] = p * (pacc[e] - dlr[e]) * scale;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float p = __builtin_amdgcn_exp2f(fmaf(sacc[e], scale_log2, -l2r[e]));
                    sacc[e] = p;
                    pacc[e] = p * (pacc[e] - dlr[e]) * scale;
                }
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 pf = pack_frag(sacc, s);
                const bf16x8 dsf = pack_frag(pacc, s);
#pragma unroll
                for (int dt = 0; dt < C::DT; ++dt) {
                    dvacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_frag_tr<D>(smem, vot[dt], (st * 32 + 16 * s) * C::ROWB), pf, dvacc[dt], 0, 0, 0);
                    dkacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_frag_tr<D>(smem, vqt[dt], (st * 32 + 16 * s) * C::ROWB), dsf, dkacc[dt], 0, 0, 0);
                }
            }
        }
    }
    if (kvalid) {
        bf16_t* krow = dk + ((int64_t)b * S + kg) * lddk + (int64_t)hkv * D;
        bf16_t* vrow = dv + ((int64_t)b * S + kg) * lddv + (int64_t)hkv * D;
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d = dt * 32 + 8 * g4 + 4 * (lane >> 5);
                u32x2 a = {pack_bf2(dkacc[dt][4 * g4], dkacc[dt][4 * g4 + 1]), pack_bf2(dkacc[dt][4 * g4 + 2], dkacc[dt][4 * g4 + 3])};
                u32x2 c = {pack_bf2(dvacc[dt][4 * g4], dvacc[dt][4 * g4 + 1]), pack_bf2(dvacc[dt][4 * g4 + 2], dvacc[dt][4 * g4 + 3])};
                *reinterpret_cast<u32x2*>(krow + d) = a;
                *reinterpret_cast<u32x2*>(vrow + d) = c;
            }
    }
}

int check_common(const char* name, int B, int S, int Hq, int Hkv, int D) {
    MI355_REQUIRE(D == 64 || D == 128, "%s: head_dim must be 64 or 128 (got %d)", name, D);
    MI355_REQUIRE(B > 0 && S > 0 && Hq > 0 && Hkv > 0 && Hq % Hkv == 0, "%s: bad shape B=%d S=%d Hq=%d Hkv=%d", name, B, S, Hq, Hkv);
    return 0;
}

}  // namespace

extern "C" int mi355_attn_fwd(int B, int S, int Hq, int Hkv, int D, const void* q, int64_t ldq, const void* k, int64_t ldk,
                              const void* v, int64_t ldv, void* o, int64_t ldo, float* lse, const uint8_t* key_mask,
                              int causal, float scale, void* stream) {
    if (check_common("mi355_attn_fwd", B, S, Hq, Hkv, D)) return 1;
    MI355_REQUIRE(q && k && v && o && lse, "mi355_attn_fwd: null pointer");
    MI355_REQUIRE(((ldq | ldk | ldv | ldo) & 7) == 0, "mi355_attn_fwd: leading dimensions must be multiples of 8");
    MI355_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o) & 15) == 0, "mi355_attn_fwd: operands must be 16-byte aligned");
    const int64_t grid = (int64_t)B * Hq * ((S + 127) / 128);
    MI355_REQUIRE(grid < 0x7fffffffLL, "mi355_attn_fwd: grid too large");
    hipStream_t s = (hipStream_t)stream;
    const float sl2 = scale * LOG2E;
    if (D == 128)
        hipLaunchKernelGGL(attn_fwd_kernel<128>, dim3((unsigned)grid), dim3(256), 0, s, B, S, Hq, Hkv, (const bf16_t*)q, ldq, (const bf16_t*)k, ldk, (const bf16_t*)v, ldv, (bf16_t*)o, ldo, lse, key_mask, causal, sl2);
    else
        hipLaunchKernelGGL(attn_fwd_kernel<64>, dim3((unsigned)grid), dim3(256), 0, s, B, S, Hq, Hkv, (const bf16_t*)q, ldq, (const bf16_t*)k, ldk, (const bf16_t*)v, ldv, (bf16_t*)o, ldo, lse, key_mask, causal, sl2);
    MI355_LAUNCH_CHECK("mi355_attn_fwd");
    return 0;
}

extern "C" int mi355_attn_bwd(int B, int S, int Hq, int Hkv, int D, const void* q, int64_t ldq, const void* k, int64_t ldk,
                              const void* v, int64_t ldv, const void* o, int64_t ldo, const void* d_o, int64_t lddo,
                              const float* lse, float* delta, void* dq, int64_t lddq, void* dk, int64_t lddk, void* dv,
                              int64_t lddv, const uint8_t* key_mask, int causal, float scale, void* stream) {
    if (check_common("mi355_attn_bwd", B, S, Hq, Hkv, D)) return 1;
    MI355_REQUIRE(q && k && v && o && d_o && lse && delta && dq && dk && dv, "mi355_attn_bwd: null pointer");
    MI355_REQUIRE(((ldq | ldk | ldv | ldo | lddo | lddq | lddk | lddv) & 7) == 0, "mi355_attn_bwd: leading dimensions must be multiples of 8");
    hipStream_t s = (hipStream_t)stream;
    const float sl2 = scale * LOG2E;
    const int64_t items = (int64_t)B * S * Hq;
    const int dgrid = (int)((items + 3) / 4 > 2048 ? 2048 : (items + 3) / 4);
    const int64_t gq = (int64_t)B * Hq * ((S + 127) / 128), gk = (int64_t)B * Hkv * ((S + 127) / 128);
    MI355_REQUIRE(gq < 0x7fffffffLL, "mi355_attn_bwd: grid too large");
#define BWD_LAUNCH(DD)                                                                                                              \
    hipLaunchKernelGGL(attn_delta_kernel<DD>, dim3(dgrid), dim3(256), 0, s, B, S, Hq, (const bf16_t*)o, ldo, (const bf16_t*)d_o, lddo, delta); \
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<DD>, dim3((unsigned)gk), dim3(256), 0, s, B, S, Hq, Hkv, (const bf16_t*)q, ldq, (const bf16_t*)k, ldk, \
                       (const bf16_t*)v, ldv, (const bf16_t*)d_o, lddo, lse, delta, (bf16_t*)dk, lddk, (bf16_t*)dv, lddv, key_mask, causal, scale, sl2); \
    hipLaunchKernelGGL(attn_bwd_dq_kernel<DD>, dim3((unsigned)gq), dim3(256), 0, s, B, S, Hq, Hkv, (const bf16_t*)q, ldq, (const bf16_t*)k, ldk,  \
                       (const bf16_t*)v, ldv, (const bf16_t*)d_o, lddo, lse, delta, (bf16_t*)dq, lddq, key_mask, causal, scale, sl2);
    if (D == 128) {
        BWD_LAUNCH(128)
    } else {
        BWD_LAUNCH(64)
    }
#undef BWD_LAUNCH
    MI355_LAUNCH_CHECK("mi355_attn_bwd");
    return 0;
}
