// Attention for head dims the tuned kernels of attention.hip do not cover (d_h = 256 of the Qwen3.5 gated-attention layers;
// 32 for the tiny fixtures), with the mask semantics of F.scaled_dot_product_attention as the reference calls it
// (qwen3_next_attention.py:238-254, qwen3_5_text_model.py:244-259): boolean mask, True = attend, everything else -inf.
//   allowed(i, j) = j <= i  OR  key j is padding        <- upstream ORs the inverted padding mask into the ALLOW mask,
//                                                           which un-masks padded keys for every query; reproduced.
// Same data flow as the tuned forward: S^T = K Q^T with the query on the MFMA lane (mfma_f32_32x32x16_bf16, A = K rows from
// LDS, B = Q rows in registers), lane-local online softmax, P^T packed from the accumulators as the B operand of
// O^T += V^T P^T, V consumed K-strided through ds_read_b64_tr_b16.  Plain (padded, unswizzled) LDS images, one 32-key tile per
// barrier pair, no pipelining: these layers are 1.2 % of config 5's FLOPs -- correctness and determinism first.
// Backward: delta = rowsum(dO * O); dQ pass query-major; dK/dV pass key-major (no atomics), with the d_h = 256 accumulators
// split into two 128-wide passes so a wave's dK^T / dV^T tiles fit the register file.
//
// PLAIN = true is the second semantics built on the same kernels (mi355_attn_dropout_fwd / _bwd): softmax over all keys or
// plain causal (no key mask, no quirk), with nn.Dropout(p) on the NORMALISED weights as ViTMultiHeadAttention applies it
// (vit_attention.py:79): O = ((P o M) / (1 - p)) V / l with l from the un-dropped P.  The mask is never stored: element
// (b, h, query, key) keeps iff Philox4x32-10(seed; key / 4, (b Hq + h) S + query, offset)[key % 4] >= round(p 2^32), and the
// backward regenerates it (dV += (P o M / keep)^T dO, dP = (dO V^T) o M / keep, dS = P (dP - delta), delta = rowsum(dO o O)).
#include "common.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
#define NEG_INF (-__builtin_huge_valf())

struct DropArgs {
    int causal;       // PLAIN only: 1 = keys <= query, 0 = every key
    unsigned thresh;  // 0 = no dropout
    float inv_keep;
    unsigned k0, k1, o0, o1;
};
// keep multipliers (0 or 1 / (1 - p)) of the four consecutive keys key4 .. key4+3 (key4 % 4 == 0) of attention row `row`
__device__ __forceinline__ void drop4(const DropArgs& da, unsigned row, int key4, float (&mul)[4]) {
    unsigned bits[4];
    philox4x32_10((unsigned)key4 >> 2, row, da.o0, da.o1, da.k0, da.k1, bits);
#pragma unroll
    for (int e = 0; e < 4; ++e) mul[e] = bits[e] >= da.thresh ? da.inv_keep : 0.f;
}

template <int D>
struct GA {
    static constexpr int PITCH = D * 2 + 16;  // bytes per LDS row
    static constexpr int KS = D / 16;         // k-steps over d
    static constexpr int DT = D / 32;         // 32-row tiles of a transposed [d x 32] accumulator
    static constexpr int IMG = 32 * PITCH;    // one 32-row image
};

// cooperative load of 32 rows x D bf16 (token rows tok0.., column col0) into a padded LDS image; rows >= rows_valid are zero
template <int D, int NT>
__device__ __forceinline__ void load_tile(char* img, const bf16_t* base, int64_t ld, int rows_valid, int tid) {
    constexpr int CH = D / 8;
    for (int c = tid; c < 32 * CH; c += NT) {
        const int row = c / CH, ch = c % CH;
        u32x4 v = {0, 0, 0, 0};
        if (row < rows_valid) v = *reinterpret_cast<const u32x4*>(base + (int64_t)row * ld + ch * 8);
        *reinterpret_cast<u32x4*>(img + row * GA<D>::PITCH + ch * 16) = v;
    }
}
// A operand (32 rows x 16 k) from a row image: row = lane & 31, k = 16 ks + 8 (lane >> 5) ..
template <int D>
__device__ __forceinline__ bf16x8 frag_rows(const char* img, int ks, int lane) {
    return *reinterpret_cast<const bf16x8*>(img + (lane & 31) * GA<D>::PITCH + (2 * ks + (lane >> 5)) * 16);
}
// A operand of the TRANSPOSE of a row image: rows of A = image columns c0 .. c0+31, k = image rows in the order in which an
// accumulator tile packs into a B operand: element j <-> image row k0 + 8 (j >> 2) + 4 (lane >> 5) + (j & 3)
template <int D>
__device__ __forceinline__ bf16x8 frag_cols(const char* img, int c0, int k0, int lane) {
    const int g = lane >> 4, q4 = (lane >> 2) & 3, p = lane & 3;
    const int row = k0 + 4 * (g >> 1) + q4;
    const int col = c0 + 16 * (g & 1) + 4 * p;
    const char* a = img + row * GA<D>::PITCH + col * 2;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(a));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(a + 8 * GA<D>::PITCH));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ bf16x8 pack_frag(const f32x16& x, int s) {
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack_bf2(x[8 * s + 2 * e], x[8 * s + 2 * e + 1]);
    return __builtin_bit_cast(bf16x8, o);
}
// B operand fragments of a row held on the lane (row = lane & 31 of the wave's 32 rows)
template <int D>
__device__ __forceinline__ void load_row_frags(const bf16_t* rowptr, bool valid, int lane, bf16x8 (&f)[GA<D>::KS]) {
#pragma unroll
    for (int ks = 0; ks < GA<D>::KS; ++ks) {
        u32x4 v = {0, 0, 0, 0};
        if (valid) v = *reinterpret_cast<const u32x4*>(rowptr + 16 * ks + 8 * (lane >> 5));
        f[ks] = __builtin_bit_cast(bf16x8, v);
    }
}
// accumulator tile [32 d x 32 rows-on-lane] -> token-major bf16 rows (4 consecutive d per 8-byte store)
template <int NDT>
__device__ __forceinline__ void store_t_tiles(const f32x16 (&acc)[NDT], float mul, bf16_t* rowptr, bool valid, int lane) {
    if (!valid) return;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) {
            u32x2 w;
            w[0] = pack_bf2(acc[dt][4 * i4] * mul, acc[dt][4 * i4 + 1] * mul);
            w[1] = pack_bf2(acc[dt][4 * i4 + 2] * mul, acc[dt][4 * i4 + 3] * mul);
            *reinterpret_cast<u32x2*>(rowptr + 32 * dt + 8 * i4 + 4 * (lane >> 5)) = w;
        }
}

template <int D, bool PLAIN, bool DROP>
__global__ __launch_bounds__(256) void ga_fwd_kernel(int B, int S, int Hq, int Hkv, const bf16_t* __restrict__ q, int64_t ldq,
                                                     const bf16_t* __restrict__ k, int64_t ldk, const bf16_t* __restrict__ v, int64_t ldv,
                                                     bf16_t* __restrict__ o, int64_t ldo, float* __restrict__ lse,
                                                     const uint8_t* __restrict__ key_mask, float scale_log2, DropArgs da) {
    using C = GA<D>;
    __shared__ __attribute__((aligned(16))) char smem[2 * C::IMG];
    char* kimg = smem;
    char* vimg = smem + C::IMG;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.z, h = blockIdx.y, hkv = h / (Hq / Hkv);
    const int q0 = blockIdx.x * 128 + wave * 32;
    const int query = q0 + (lane & 31);
    const bool qvalid = query < S;
    const int64_t tok0 = (int64_t)b * S;
    bf16x8 qf[C::KS];
    load_row_frags<D>(q + (tok0 + query) * ldq + (int64_t)h * D, qvalid, lane, qf);
    f32x16 acc[C::DT];
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[dt][i] = 0.f;
    float m = NEG_INF, l = 0.f;
    const int qlast = (blockIdx.x * 128 + 127 < S ? blockIdx.x * 128 + 127 : S - 1);
    const bool all_keys = PLAIN ? !da.causal : key_mask != nullptr;  // every key tile is walked
    const int ntiles = all_keys ? (S + 31) / 32 : qlast / 32 + 1;
    const uint8_t* km = (!PLAIN && key_mask) ? key_mask + tok0 : nullptr;
    const unsigned drow = (unsigned)(((int64_t)b * Hq + h) * S + query);
    for (int kt = 0; kt < ntiles; ++kt) {
        const int key0 = kt * 32;
        __syncthreads();
        load_tile<D, 256>(kimg, k + (tok0 + key0) * ldk + (int64_t)hkv * D, ldk, S - key0, threadIdx.x);
        load_tile<D, 256>(vimg, v + (tok0 + key0) * ldv + (int64_t)hkv * D, ldv, S - key0, threadIdx.x);
        __syncthreads();
        if (!all_keys && key0 > q0 + 31) continue;  // tile entirely above this wave's diagonal
        f32x16 s;
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = 0.f;
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<D>(kimg, ks, lane), qf[ks], s, 0, 0, 0);
        float mx = NEG_INF;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int key = key0 + 8 * (i >> 2) + 4 * (lane >> 5) + (i & 3);
            const bool ok = PLAIN ? (key < S && (!da.causal || key <= query)) : (key < S && (key <= query || (km && !km[key])));
            s[i] = ok ? s[i] * scale_log2 : NEG_INF;
            mx = fmaxf(mx, s[i]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m, mx);
        const float m_use = m_new == NEG_INF ? 0.f : m_new;
        const float alpha = exp2f(m - m_use);
        float rs = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            s[i] = exp2f(s[i] - m_use);
            rs += s[i];
        }
        rs += __shfl_xor(rs, 32, 64);
        l = l * alpha + rs;
        m = m_new;
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[dt][i] *= alpha;
        if (DROP && da.thresh) {  // (DROP is a template parameter: the p = 0 kernels of the Qwen3.5 training path carry no Philox code)  // dropout on the weights that multiply V; the row sum above keeps every key
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float mul[4];
                drop4(da, drow, key0 + 8 * g + 4 * (lane >> 5), mul);
#pragma unroll
                for (int e = 0; e < 4; ++e) s[4 * g + e] *= mul[e];
            }
        }
        const bf16x8 p0 = pack_frag(s, 0), p1 = pack_frag(s, 1);
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) {
            acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols<D>(vimg, 32 * dt, 0, lane), p0, acc[dt], 0, 0, 0);
            acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols<D>(vimg, 32 * dt, 16, lane), p1, acc[dt], 0, 0, 0);
        }
    }
    store_t_tiles<C::DT>(acc, 1.f / l, o + (tok0 + query) * ldo + (int64_t)h * D, qvalid, lane);
    if (lane < 32 && qvalid) lse[((int64_t)b * Hq + h) * S + query] = (m + log2f(l)) * LN2;
}

// delta[b, h, s] = sum_d dO * O
__global__ __launch_bounds__(256) void ga_delta_kernel(int64_t tokens, int S, int Hq, int D, const bf16_t* __restrict__ o, int64_t ldo,
                                                       const bf16_t* __restrict__ d_o, int64_t lddo, float* __restrict__ delta) {
    const int lane = threadIdx.x & 63;
    const int64_t total = tokens * Hq;
    for (int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); item < total; item += (int64_t)gridDim.x * 4) {
        const int64_t t = item / Hq;
        const int h = (int)(item % Hq);
        float acc = 0.f;
        for (int i = lane; i < D; i += 64) acc += bf2f(o[t * ldo + (int64_t)h * D + i]) * bf2f(d_o[t * lddo + (int64_t)h * D + i]);
        acc = wave_sum(acc);
        if (lane == 0) delta[((t / S) * Hq + h) * S + (t % S)] = acc;
    }
}

template <int D, bool PLAIN, bool DROP>
__global__ __launch_bounds__(256) void ga_bwd_dq_kernel(int B, int S, int Hq, int Hkv, const bf16_t* __restrict__ q, int64_t ldq,
                                                        const bf16_t* __restrict__ k, int64_t ldk, const bf16_t* __restrict__ v, int64_t ldv,
                                                        const bf16_t* __restrict__ d_o, int64_t lddo, const float* __restrict__ lse,
                                                        const float* __restrict__ delta, bf16_t* __restrict__ dq, int64_t lddq,
                                                        const uint8_t* __restrict__ key_mask, float scale, DropArgs da) {
    using C = GA<D>;
    __shared__ __attribute__((aligned(16))) char smem[2 * C::IMG];
    char* kimg = smem;
    char* vimg = smem + C::IMG;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.z, h = blockIdx.y, hkv = h / (Hq / Hkv);
    const int q0 = blockIdx.x * 128 + wave * 32;
    const int query = q0 + (lane & 31);
    const bool qvalid = query < S;
    const int64_t tok0 = (int64_t)b * S;
    bf16x8 qf[C::KS], gf[C::KS];
    load_row_frags<D>(q + (tok0 + query) * ldq + (int64_t)h * D, qvalid, lane, qf);
    load_row_frags<D>(d_o + (tok0 + query) * lddo + (int64_t)h * D, qvalid, lane, gf);
    const float lse_q = qvalid ? lse[((int64_t)b * Hq + h) * S + query] * LOG2E : 0.f;
    const float delta_q = qvalid ? delta[((int64_t)b * Hq + h) * S + query] : 0.f;
    const float scale_log2 = scale * LOG2E;
    f32x16 acc[C::DT];
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[dt][i] = 0.f;
    const int qlast = (blockIdx.x * 128 + 127 < S ? blockIdx.x * 128 + 127 : S - 1);
    const bool all_keys = PLAIN ? !da.causal : key_mask != nullptr;
    const int ntiles = all_keys ? (S + 31) / 32 : qlast / 32 + 1;
    const uint8_t* km = (!PLAIN && key_mask) ? key_mask + tok0 : nullptr;
    const unsigned drow = (unsigned)(((int64_t)b * Hq + h) * S + query);
    for (int kt = 0; kt < ntiles; ++kt) {
        const int key0 = kt * 32;
        __syncthreads();
        load_tile<D, 256>(kimg, k + (tok0 + key0) * ldk + (int64_t)hkv * D, ldk, S - key0, threadIdx.x);
        load_tile<D, 256>(vimg, v + (tok0 + key0) * ldv + (int64_t)hkv * D, ldv, S - key0, threadIdx.x);
        __syncthreads();
        if (!all_keys && key0 > q0 + 31) continue;
        f32x16 s, dp;
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = dp[i] = 0.f;
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) {
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<D>(kimg, ks, lane), qf[ks], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<D>(vimg, ks, lane), gf[ks], dp, 0, 0, 0);
        }
        if (DROP && da.thresh) {  // (DROP is a template parameter: the p = 0 kernels of the Qwen3.5 training path carry no Philox code)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float mul[4];
                drop4(da, drow, key0 + 8 * g + 4 * (lane >> 5), mul);
#pragma unroll
                for (int e = 0; e < 4; ++e) dp[4 * g + e] *= mul[e];
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int key = key0 + 8 * (i >> 2) + 4 * (lane >> 5) + (i & 3);
            const bool ok = PLAIN ? (qvalid && key < S && (!da.causal || key <= query)) : (qvalid && key < S && (key <= query || (km && !km[key])));
            const float p = ok ? exp2f(s[i] * scale_log2 - lse_q) : 0.f;
            s[i] = p * (dp[i] - delta_q) * scale;
        }
        const bf16x8 d0 = pack_frag(s, 0), d1 = pack_frag(s, 1);
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) {
            acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols<D>(kimg, 32 * dt, 0, lane), d0, acc[dt], 0, 0, 0);
            acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols<D>(kimg, 32 * dt, 16, lane), d1, acc[dt], 0, 0, 0);
        }
    }
    store_t_tiles<C::DT>(acc, 1.f, dq + (tok0 + query) * lddq + (int64_t)h * D, qvalid, lane);
}

// key-major pass: a wave owns 32 keys of one kv head; NP = number of d-slices the dK^T / dV^T accumulators are split into
// (blockIdx.x = key_block * NP + slice); all q heads of the group and all query tiles from the diagonal on are walked.
template <int D, int NP, bool PLAIN, bool DROP>
__global__ __launch_bounds__(256) void ga_bwd_dkv_kernel(int B, int S, int Hq, int Hkv, const bf16_t* __restrict__ q, int64_t ldq,
                                                         const bf16_t* __restrict__ k, int64_t ldk, const bf16_t* __restrict__ v, int64_t ldv,
                                                         const bf16_t* __restrict__ d_o, int64_t lddo, const float* __restrict__ lse,
                                                         const float* __restrict__ delta, bf16_t* __restrict__ dk, int64_t lddk,
                                                         bf16_t* __restrict__ dv, int64_t lddv, const uint8_t* __restrict__ key_mask, float scale,
                                                         DropArgs da) {
    using C = GA<D>;
    constexpr int NDT = C::DT / NP;
    __shared__ __attribute__((aligned(16))) char smem[2 * C::IMG];
    __shared__ float stat[2][32];
    char* qimg = smem;
    char* gimg = smem + C::IMG;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.z, hkv = blockIdx.y, rep = Hq / Hkv;
    const int kb = blockIdx.x / NP, slice = blockIdx.x % NP;
    const int k0 = kb * 128 + wave * 32;
    const int key = k0 + (lane & 31);
    const bool kvalid = key < S;
    const int64_t tok0 = (int64_t)b * S;
    bf16x8 kf[C::KS], vf[C::KS];
    load_row_frags<D>(k + (tok0 + key) * ldk + (int64_t)hkv * D, kvalid, lane, kf);
    load_row_frags<D>(v + (tok0 + key) * ldv + (int64_t)hkv * D, kvalid, lane, vf);
    const bool padded = !PLAIN && key_mask && kvalid && !key_mask[tok0 + key];
    const bool all_queries = PLAIN ? !da.causal : key_mask != nullptr;  // every query tile is walked
    const float scale_log2 = scale * LOG2E;
    f32x16 adk[NDT], adv[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) adk[dt][i] = adv[dt][i] = 0.f;
    const int qt0 = all_queries ? 0 : (kb * 128) / 32;
    const int nqt = (S + 31) / 32;
    for (int hq = hkv * rep; hq < (hkv + 1) * rep; ++hq) {
        for (int qt = qt0; qt < nqt; ++qt) {
            const int qs = qt * 32;
            __syncthreads();
            load_tile<D, 256>(qimg, q + (tok0 + qs) * ldq + (int64_t)hq * D, ldq, S - qs, threadIdx.x);
            load_tile<D, 256>(gimg, d_o + (tok0 + qs) * lddo + (int64_t)hq * D, lddo, S - qs, threadIdx.x);
            if (threadIdx.x < 32) {
                const bool okq = qs + threadIdx.x < S;
                stat[0][threadIdx.x] = okq ? lse[((int64_t)b * Hq + hq) * S + qs + threadIdx.x] * LOG2E : 0.f;
                stat[1][threadIdx.x] = okq ? delta[((int64_t)b * Hq + hq) * S + qs + threadIdx.x] : 0.f;
            }
            __syncthreads();
            if (!all_queries && qs + 31 < k0) continue;  // every query of the tile precedes this wave's keys
            f32x16 s, dp;
#pragma unroll
            for (int i = 0; i < 16; ++i) s[i] = dp[i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < C::KS; ++ks) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<D>(qimg, ks, lane), kf[ks], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<D>(gimg, ks, lane), vf[ks], dp, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int r = 8 * (i >> 2) + 4 * (lane >> 5) + (i & 3);
                const int qi = qs + r;
                const bool ok = PLAIN ? (kvalid && qi < S && (!da.causal || key <= qi)) : (kvalid && qi < S && (key <= qi || padded));
                const float p = ok ? exp2f(s[i] * scale_log2 - stat[0][r]) : 0.f;
                float keep = 1.f;
                if (DROP && da.thresh) {  // (DROP is a template parameter: the p = 0 kernels of the Qwen3.5 training path carry no Philox code)  // this lane's key inside its group of four, for attention row (b, hq, qi)
                    unsigned bits[4];
                    philox4x32_10((unsigned)key >> 2, (unsigned)(((int64_t)b * Hq + hq) * S + qi), da.o0, da.o1, da.k0, da.k1, bits);
                    const unsigned bsel = (key & 3) == 0 ? bits[0] : (key & 3) == 1 ? bits[1] : (key & 3) == 2 ? bits[2] : bits[3];
                    keep = bsel >= da.thresh ? da.inv_keep : 0.f;
                }
                s[i] = p * keep;
                dp[i] = p * (dp[i] * keep - stat[1][r]) * scale;
            }
            const bf16x8 p0 = pack_frag(s, 0), p1 = pack_frag(s, 1), d0 = pack_frag(dp, 0), d1 = pack_frag(dp, 1);
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                const int c0 = 32 * (slice * NDT + dt);
                adv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols<D>(gimg, c0, 0, lane), p0, adv[dt], 0, 0, 0);
                adv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols<D>(gimg, c0, 16, lane), p1, adv[dt], 0, 0, 0);
                adk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols<D>(qimg, c0, 0, lane), d0, adk[dt], 0, 0, 0);
                adk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols<D>(qimg, c0, 16, lane), d1, adk[dt], 0, 0, 0);
            }
        }
    }
    const int dofs = 32 * slice * NDT;
    store_t_tiles<NDT>(adk, 1.f, dk + (tok0 + key) * lddk + (int64_t)hkv * D + dofs, kvalid, lane);
    store_t_tiles<NDT>(adv, 1.f, dv + (tok0 + key) * lddv + (int64_t)hkv * D + dofs, kvalid, lane);
}

int check_ga(int B, int S, int Hq, int Hkv, int D, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo) {
    MI355_REQUIRE(B > 0 && S > 0 && Hq > 0 && Hkv > 0 && Hq % Hkv == 0, "attn_generic: query heads (%d) must be a multiple of kv heads (%d)", Hq, Hkv);
    MI355_REQUIRE(D == 32 || D == 64 || D == 128 || D == 256, "attn_generic: head_dim %d not built (32, 64, 128, 256)", D);
    MI355_REQUIRE(ldq >= (int64_t)Hq * D && ldo >= (int64_t)Hq * D && ldk >= (int64_t)Hkv * D && ldv >= (int64_t)Hkv * D, "attn_generic: leading dimension smaller than heads*head_dim");
    MI355_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 8 == 0, "attn_generic: leading dimensions must be multiples of 8 elements");
    MI355_REQUIRE(B <= 65535 && Hq <= 65535, "attn_generic: grid limits");
    return 0;
}

}  // namespace

#define ST(s) ((hipStream_t)(s))

extern "C" int mi355_attn_generic_fwd(int B, int S, int Hq, int Hkv, int D, const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v,
                                      int64_t ldv, void* o, int64_t ldo, float* lse, const uint8_t* key_mask, float scale, void* stream) {
#define GA_DROP false
    if (check_ga(B, S, Hq, Hkv, D, ldq, ldk, ldv, ldo)) return 1;
    MI355_REQUIRE(q && k && v && o && lse, "attn_generic_fwd: null pointer");
    dim3 grid((S + 127) / 128, Hq, B);
    const DropArgs da = {};
#define LAUNCH(DD) ga_fwd_kernel<DD, false, GA_DROP><<<grid, 256, 0, ST(stream)>>>(B, S, Hq, Hkv, (const bf16_t*)q, ldq, (const bf16_t*)k, ldk, (const bf16_t*)v, ldv, (bf16_t*)o, ldo, lse, key_mask, scale * LOG2E, da)
    switch (D) {
        case 32: LAUNCH(32); break;
        case 64: LAUNCH(64); break;
        case 128: LAUNCH(128); break;
        default: LAUNCH(256); break;
    }
#undef LAUNCH
    MI355_LAUNCH_CHECK("attn_generic_fwd");
#undef GA_DROP
    return 0;
}

extern "C" int mi355_attn_generic_bwd(int B, int S, int Hq, int Hkv, int D, const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v,
                                      int64_t ldv, const void* o, int64_t ldo, const void* d_o, int64_t lddo, const float* lse, float* delta,
                                      void* dq, int64_t lddq, void* dk, int64_t lddk, void* dv, int64_t lddv, const uint8_t* key_mask,
                                      float scale, void* stream) {
#define GA_DROP false
    if (check_ga(B, S, Hq, Hkv, D, ldq, ldk, ldv, ldo)) return 1;
    if (check_ga(B, S, Hq, Hkv, D, lddq, lddk, lddv, lddo)) return 1;
    MI355_REQUIRE(q && k && v && o && d_o && lse && delta && dq && dk && dv, "attn_generic_bwd: null pointer");
    const int64_t tokens = (int64_t)B * S;
    int64_t dg = (tokens * Hq + 3) / 4;
    ga_delta_kernel<<<(int)(dg > 8192 ? 8192 : dg), 256, 0, ST(stream)>>>(tokens, S, Hq, D, (const bf16_t*)o, ldo, (const bf16_t*)d_o, lddo, delta);
    MI355_LAUNCH_CHECK("attn_generic_bwd(delta)");
    dim3 gq((S + 127) / 128, Hq, B);
    const DropArgs da = {};
#define LAUNCH_DQ(DD) ga_bwd_dq_kernel<DD, false, GA_DROP><<<gq, 256, 0, ST(stream)>>>(B, S, Hq, Hkv, (const bf16_t*)q, ldq, (const bf16_t*)k, ldk, (const bf16_t*)v, ldv, (const bf16_t*)d_o, lddo, lse, delta, (bf16_t*)dq, lddq, key_mask, scale, da)
#define LAUNCH_DKV(DD, NP) ga_bwd_dkv_kernel<DD, NP, false, GA_DROP><<<dim3(((S + 127) / 128) * NP, Hkv, B), 256, 0, ST(stream)>>>(B, S, Hq, Hkv, (const bf16_t*)q, ldq, (const bf16_t*)k, ldk, (const bf16_t*)v, ldv, (const bf16_t*)d_o, lddo, lse, delta, (bf16_t*)dk, lddk, (bf16_t*)dv, lddv, key_mask, scale, da)
    switch (D) {
        case 32: LAUNCH_DQ(32); LAUNCH_DKV(32, 1); break;
        case 64: LAUNCH_DQ(64); LAUNCH_DKV(64, 1); break;
        case 128: LAUNCH_DQ(128); LAUNCH_DKV(128, 1); break;
        default: LAUNCH_DQ(256); LAUNCH_DKV(256, 2); break;
    }
#undef LAUNCH_DQ
#undef LAUNCH_DKV
    MI355_LAUNCH_CHECK("attn_generic_bwd");
#undef GA_DROP
    return 0;
}

// ---- PLAIN semantics: full or plain-causal softmax with dropout on the weights (ViTMultiHeadAttention in train mode) -------------
static int make_drop_args(const char* who, int B, int S, int Hq, int causal, float p, uint64_t seed, uint64_t offset, DropArgs* da) {
    MI355_REQUIRE(p >= 0.f && p < 1.f, "%s: p must be in [0, 1) (got %f)", who, (double)p);
    MI355_REQUIRE((int64_t)B * Hq * S < (int64_t)1 << 32, "%s: B * Hq * S must fit 32 bits (dropout counter)", who);
    da->causal = causal != 0;
    da->thresh = p > 0.f ? mi355_dropout_threshold(p) : 0u;
    da->inv_keep = 1.0f / (1.0f - p);
    da->k0 = (unsigned)seed, da->k1 = (unsigned)(seed >> 32), da->o0 = (unsigned)offset, da->o1 = (unsigned)(offset >> 32);
    return 0;
}

extern "C" int mi355_attn_dropout_fwd(int B, int S, int Hq, int Hkv, int D, const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v,
                                      int64_t ldv, void* o, int64_t ldo, float* lse, int causal, float scale, float p, uint64_t seed, uint64_t offset,
                                      void* stream) {
    if (check_ga(B, S, Hq, Hkv, D, ldq, ldk, ldv, ldo)) return 1;
    MI355_REQUIRE(q && k && v && o && lse, "attn_dropout_fwd: null pointer");
    DropArgs da;
    if (make_drop_args("attn_dropout_fwd", B, S, Hq, causal, p, seed, offset, &da)) return 1;
    dim3 grid((S + 127) / 128, Hq, B);
#define LAUNCH(DD) ga_fwd_kernel<DD, true, true><<<grid, 256, 0, ST(stream)>>>(B, S, Hq, Hkv, (const bf16_t*)q, ldq, (const bf16_t*)k, ldk, (const bf16_t*)v, ldv, (bf16_t*)o, ldo, lse, nullptr, scale * LOG2E, da)
    switch (D) {
        case 32: LAUNCH(32); break;
        case 64: LAUNCH(64); break;
        case 128: LAUNCH(128); break;
        default: LAUNCH(256); break;
    }
#undef LAUNCH
    MI355_LAUNCH_CHECK("attn_dropout_fwd");
    return 0;
}

extern "C" int mi355_attn_dropout_bwd(int B, int S, int Hq, int Hkv, int D, const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v,
                                      int64_t ldv, const void* o, int64_t ldo, const void* d_o, int64_t lddo, const float* lse, float* delta,
                                      void* dq, int64_t lddq, void* dk, int64_t lddk, void* dv, int64_t lddv, int causal, float scale, float p,
                                      uint64_t seed, uint64_t offset, void* stream) {
    if (check_ga(B, S, Hq, Hkv, D, ldq, ldk, ldv, ldo)) return 1;
    if (check_ga(B, S, Hq, Hkv, D, lddq, lddk, lddv, lddo)) return 1;
    MI355_REQUIRE(q && k && v && o && d_o && lse && delta && dq && dk && dv, "attn_dropout_bwd: null pointer");
    DropArgs da;
    if (make_drop_args("attn_dropout_bwd", B, S, Hq, causal, p, seed, offset, &da)) return 1;
    const int64_t tokens = (int64_t)B * S;
    int64_t dg = (tokens * Hq + 3) / 4;
    ga_delta_kernel<<<(int)(dg > 8192 ? 8192 : dg), 256, 0, ST(stream)>>>(tokens, S, Hq, D, (const bf16_t*)o, ldo, (const bf16_t*)d_o, lddo, delta);
    MI355_LAUNCH_CHECK("attn_dropout_bwd(delta)");
    dim3 gq((S + 127) / 128, Hq, B);
#define LAUNCH_DQ(DD) ga_bwd_dq_kernel<DD, true, true><<<gq, 256, 0, ST(stream)>>>(B, S, Hq, Hkv, (const bf16_t*)q, ldq, (const bf16_t*)k, ldk, (const bf16_t*)v, ldv, (const bf16_t*)d_o, lddo, lse, delta, (bf16_t*)dq, lddq, nullptr, scale, da)
#define LAUNCH_DKV(DD, NP) ga_bwd_dkv_kernel<DD, NP, true, true><<<dim3(((S + 127) / 128) * NP, Hkv, B), 256, 0, ST(stream)>>>(B, S, Hq, Hkv, (const bf16_t*)q, ldq, (const bf16_t*)k, ldk, (const bf16_t*)v, ldv, (const bf16_t*)d_o, lddo, lse, delta, (bf16_t*)dk, lddk, (bf16_t*)dv, lddv, nullptr, scale, da)
    switch (D) {
        case 32: LAUNCH_DQ(32); LAUNCH_DKV(32, 1); break;
        case 64: LAUNCH_DQ(64); LAUNCH_DKV(64, 1); break;
        case 128: LAUNCH_DQ(128); LAUNCH_DKV(128, 1); break;
        default: LAUNCH_DQ(256); LAUNCH_DKV(256, 2); break;
    }
#undef LAUNCH_DQ
#undef LAUNCH_DKV
    MI355_LAUNCH_CHECK("attn_dropout_bwd");
    return 0;
}

// ---- the reference's SDPA call WITH dropout_p and a padding mask (GatedAttention in training mode on padded batches, qwen3_next_attention.py:240-253):
// the quirk-mask kernels (PLAIN = false) with the dropout arguments set -- Philox masks on the normalised weights, regenerated by the backward.
extern "C" int mi355_attn_generic_dropout_fwd(int B, int S, int Hq, int Hkv, int D, const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v,
                                              int64_t ldv, void* o, int64_t ldo, float* lse, const uint8_t* key_mask, float scale, float p, uint64_t seed,
                                              uint64_t offset, void* stream) {
#define GA_DROP true
    if (check_ga(B, S, Hq, Hkv, D, ldq, ldk, ldv, ldo)) return 1;
    MI355_REQUIRE(q && k && v && o && lse, "attn_generic_dropout_fwd: null pointer");
    DropArgs da;
    if (make_drop_args("attn_generic_dropout_fwd", B, S, Hq, 1, p, seed, offset, &da)) return 1;
    dim3 grid((S + 127) / 128, Hq, B);
#define LAUNCH(DD) ga_fwd_kernel<DD, false, GA_DROP><<<grid, 256, 0, ST(stream)>>>(B, S, Hq, Hkv, (const bf16_t*)q, ldq, (const bf16_t*)k, ldk, (const bf16_t*)v, ldv, (bf16_t*)o, ldo, lse, key_mask, scale * LOG2E, da)
    switch (D) {
        case 32: LAUNCH(32); break;
        case 64: LAUNCH(64); break;
        case 128: LAUNCH(128); break;
        default: LAUNCH(256); break;
    }
#undef LAUNCH
    MI355_LAUNCH_CHECK("attn_generic_dropout_fwd");
#undef GA_DROP
    return 0;
}

extern "C" int mi355_attn_generic_dropout_bwd(int B, int S, int Hq, int Hkv, int D, const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v,
                                              int64_t ldv, const void* o, int64_t ldo, const void* d_o, int64_t lddo, const float* lse, float* delta,
                                              void* dq, int64_t lddq, void* dk, int64_t lddk, void* dv, int64_t lddv, const uint8_t* key_mask, float scale,
                                              float p, uint64_t seed, uint64_t offset, void* stream) {
#define GA_DROP true
    if (check_ga(B, S, Hq, Hkv, D, ldq, ldk, ldv, ldo)) return 1;
    if (check_ga(B, S, Hq, Hkv, D, lddq, lddk, lddv, lddo)) return 1;
    MI355_REQUIRE(q && k && v && o && d_o && lse && delta && dq && dk && dv, "attn_generic_dropout_bwd: null pointer");
    DropArgs da;
    if (make_drop_args("attn_generic_dropout_bwd", B, S, Hq, 1, p, seed, offset, &da)) return 1;
    const int64_t tokens = (int64_t)B * S;
    int64_t dg = (tokens * Hq + 3) / 4;
    ga_delta_kernel<<<(int)(dg > 8192 ? 8192 : dg), 256, 0, ST(stream)>>>(tokens, S, Hq, D, (const bf16_t*)o, ldo, (const bf16_t*)d_o, lddo, delta);
    MI355_LAUNCH_CHECK("attn_generic_dropout_bwd(delta)");
    dim3 gq((S + 127) / 128, Hq, B);
#define LAUNCH_DQ(DD) ga_bwd_dq_kernel<DD, false, GA_DROP><<<gq, 256, 0, ST(stream)>>>(B, S, Hq, Hkv, (const bf16_t*)q, ldq, (const bf16_t*)k, ldk, (const bf16_t*)v, ldv, (const bf16_t*)d_o, lddo, lse, delta, (bf16_t*)dq, lddq, key_mask, scale, da)
#define LAUNCH_DKV(DD, NP) ga_bwd_dkv_kernel<DD, NP, false, GA_DROP><<<dim3(((S + 127) / 128) * NP, Hkv, B), 256, 0, ST(stream)>>>(B, S, Hq, Hkv, (const bf16_t*)q, ldq, (const bf16_t*)k, ldk, (const bf16_t*)v, ldv, (const bf16_t*)d_o, lddo, lse, delta, (bf16_t*)dk, lddk, (bf16_t*)dv, lddv, key_mask, scale, da)
    switch (D) {
        case 32: LAUNCH_DQ(32); LAUNCH_DKV(32, 1); break;
        case 64: LAUNCH_DQ(64); LAUNCH_DKV(64, 1); break;
        case 128: LAUNCH_DQ(128); LAUNCH_DKV(128, 1); break;
        default: LAUNCH_DQ(256); LAUNCH_DKV(256, 2); break;
    }
#undef LAUNCH_DQ
#undef LAUNCH_DKV
    MI355_LAUNCH_CHECK("attn_generic_dropout_bwd");
#undef GA_DROP
    return 0;
}
