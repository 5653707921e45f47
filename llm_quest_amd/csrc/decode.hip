// KV-cache decoding on the step's right edge: the per-token path of generate_loop_kv_cache
// (llm_quest/generate.py:97-151) through Qwen3 with utils.KVCache (utils.py:409-531).  One new token per sequence makes every
// linear layer a matrix-VECTOR product and attention one query row against the cache: both are HBM-bound byte streams (weights,
// resp. cached K/V read exactly once per token), so the kernels here are bandwidth kernels, not MFMA kernels.
//   gemv            y[m, :] = x[m, :] W^T (+ residual), m < 8 rows: one wave per output column, 16-byte weight loads, fp32 accumulate
//   attn_decode     one query per (batch, head) over `len` cached keys: 4 waves split the keys (flash-decoding), scores with the
//                   key on the lane, P V with the feature on the lane, combined through LDS; reference mask semantics (finite fill)
//   gemv_pro        the same weight stream with the row operation that feeds it folded in (PRO 1: RMSNorm of x, PRO 2: SwiGLU of a gate-up row): every
//                   workgroup rebuilds the <= 8 operand rows in LDS (a few KB) instead of a launch of its own writing them -- a decode step is launch-bound
//   attn_decode_qkv the same attention started from the fused QKV stream's raw row: QK-RMSNorm + RoPE of the head's query and of its kv head's new key inside the launch,
//                   key and value head written to the cache row at the device-side position (qknorm_rope_fwd + kv_append + attn_decode in one launch)
//   argmax_rows     greedy sampling: first index of the row maximum (torch.argmax tie rule on ties is unspecified; first is used)
#include "common.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr float MASK_T = -2.0e38f;  // finite "masked" score in the log2 domain (finfo(bf16).min / 2 semantics: exp -> 0, uniform if all masked)

__device__ __forceinline__ void unpack8(const u32x4 v, float (&f)[8]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        f[2 * e] = __uint_as_float(v[e] << 16);
        f[2 * e + 1] = __uint_as_float(v[e] & 0xffff0000u);
    }
}

// ------------------------------------------------------------------------------------------- skinny GEMM (NT), M <= 8
template <int MT>
__global__ __launch_bounds__(256) void gemv_kernel(int M, int64_t N, int K, const bf16_t* __restrict__ x, int64_t ldx, const bf16_t* __restrict__ W,
                                                   int64_t ldw, bf16_t* __restrict__ y, int64_t ldy, const bf16_t* __restrict__ res, int64_t ldr) {
    const int lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const bf16_t* w = W + n * ldw;
    float acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = 0.f;
    for (int k0 = lane * 8; k0 < K; k0 += 512) {
        float wf[8];
        unpack8(*reinterpret_cast<const u32x4*>(w + k0), wf);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            if (m >= M) break;  // MT is the next power of two: rows M..MT-1 do not exist
            float xf[8];
            unpack8(*reinterpret_cast<const u32x4*>(x + m * ldx + k0), xf);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[m] = fmaf(wf[e], xf[e], acc[m]);
        }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        if (m >= M) break;
        const float s = wave_sum(acc[m]);
        if (lane == 0) y[m * ldy + n] = f2bf(res ? s + bf2f(res[m * ldr + n]) : s);
    }
}

__device__ __forceinline__ u32x4 pack8(const float (&f)[8]) {
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack_bf2(f[2 * e], f[2 * e + 1]);
    return o;
}
__device__ __forceinline__ float rbf(float x) { return bf2f(f2bf(x)); }

// The weight stream with its operand rows built in LDS first.  PRO 1: xs[m] = RMSNorm(x[m]) * nw, the arithmetic and summation order of rmsnorm_fwd_generic
// (norm_rope.hip; == rmsnorm_fwd_kernel<2> at width 1024), so fused == the two launches bit for bit.  PRO 2: x holds gate-up rows [M, 2K] (up | gate, as the fused
// lin1 | lin_gate projection writes them), xs[m] = swiglu_act(up, gate) as swiglu_fwd_kernel rounds it.  Each wave then walks output columns wave, wave + 4 * grid, ...
template <int MT, int PRO>
__global__ __launch_bounds__(256) void gemv_pro_kernel(int M, int64_t N, int K, const bf16_t* __restrict__ x, int64_t ldx, const bf16_t* __restrict__ nw, float eps,
                                                       const bf16_t* __restrict__ W, int64_t ldw, bf16_t* __restrict__ y, int64_t ldy,
                                                       const bf16_t* __restrict__ res, int64_t ldr) {
    extern __shared__ __attribute__((aligned(16))) bf16_t xs[];  // [M][K]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nvec = K >> 3;
    if constexpr (PRO == 1) {
        __shared__ float rs[8];
        {  // the row kernel's sum, in its order: one wave per row, rows wave and wave + 4 side by side
            const int m0 = wave, m1 = wave + 4;
            float ss0 = 0.f, ss1 = 0.f;
            for (int i = lane; i < nvec; i += 64) {
                float v0[8], v1[8];
                const u32x4 r0 = m0 < M ? *reinterpret_cast<const u32x4*>(x + m0 * ldx + i * 8) : (u32x4){0u, 0u, 0u, 0u};
                const u32x4 r1 = m1 < M ? *reinterpret_cast<const u32x4*>(x + m1 * ldx + i * 8) : (u32x4){0u, 0u, 0u, 0u};
                unpack8(r0, v0);
                unpack8(r1, v1);
#pragma unroll
                for (int e = 0; e < 8; ++e) { ss0 += v0[e] * v0[e]; ss1 += v1[e] * v1[e]; }
            }
            ss0 = wave_sum(ss0);
            ss1 = wave_sum(ss1);
            if (lane == 0 && m0 < M) rs[m0] = rsqrtf(ss0 / (float)K + eps);
            if (lane == 0 && m1 < M) rs[m1] = rsqrtf(ss1 / (float)K + eps);
        }
        __syncthreads();
        for (int idx0 = threadIdx.x; idx0 < M * nvec; idx0 += 1024) {  // every thread a vector of some row; four requested together (eight rows: a chain of four round trips otherwise)
            u32x4 xv[4], nv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = idx0 + u * 256;
                if (idx < M * nvec) {
                    const int m = idx / nvec, i = idx - m * nvec;
                    xv[u] = *reinterpret_cast<const u32x4*>(x + m * ldx + i * 8);
                    nv[u] = *reinterpret_cast<const u32x4*>(nw + i * 8);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = idx0 + u * 256;
                if (idx < M * nvec) {
                    const int m = idx / nvec, i = idx - m * nvec;
                    const float r = rs[m];
                    float v[8], wv[8], o[8];
                    unpack8(xv[u], v);
                    unpack8(nv[u], wv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = v[e] * r * wv[e];
                    *reinterpret_cast<u32x4*>(xs + m * K + i * 8) = pack8(o);
                }
            }
        }
    } else {
        for (int idx0 = threadIdx.x; idx0 < M * nvec; idx0 += 1024) {
            u32x4 uv[4], gv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = idx0 + u * 256;
                if (idx < M * nvec) {
                    const int m = idx / nvec, i = idx - m * nvec;
                    uv[u] = *reinterpret_cast<const u32x4*>(x + m * ldx + i * 8);
                    gv[u] = *reinterpret_cast<const u32x4*>(x + m * ldx + K + i * 8);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = idx0 + u * 256;
                if (idx < M * nvec) {
                    const int m = idx / nvec, i = idx - m * nvec;
                    float uf[8], g[8], o[8];
                    unpack8(uv[u], uf);
                    unpack8(gv[u], g);
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = swiglu_act(uf[e], g[e]);
                    *reinterpret_cast<u32x4*>(xs + m * K + i * 8) = pack8(o);
                }
            }
        }
    }
    __syncthreads();
    for (int64_t n = (int64_t)blockIdx.x * 4 + wave; n < N; n += (int64_t)gridDim.x * 4) {
        const bf16_t* w = W + n * ldw;
        float acc[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = 0.f;
        for (int k0 = lane * 8; k0 < K; k0 += 512) {
            float wf[8];
            unpack8(*reinterpret_cast<const u32x4*>(w + k0), wf);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                if (m >= M) break;
                float xf[8];
                unpack8(*reinterpret_cast<const u32x4*>(xs + m * K + k0), xf);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[m] = fmaf(wf[e], xf[e], acc[m]);
            }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            if (m >= M) break;
            const float s = wave_sum(acc[m]);
            if (lane == 0) y[m * ldy + n] = f2bf(res ? s + bf2f(res[m * ldr + n]) : s);
        }
    }
}

// ------------------------------------------------------------------------------------------- decode attention
// grid (Hq, B), WAVES waves; chunks of CK keys go round the waves (chunk c to wave c % WAVES).  D = 32 .. 256.
// POST: the launch starts from the fused QKV stream's raw row instead of finished q rows and a finished cache row -- QK-RMSNorm + RoPE of this head's query (to LDS, never to
// memory) and of its kv head's new key, which goes with the value head into cache row *write_pos, exactly as qknorm_rope_fwd_kernel (norm_rope.hip) computes them: fp32 norm
// rounded to bf16, bf16-rounded cos / sin, every product rounded.  The query heads of one kv head all write the same bytes to the same row; each workgroup reads back its own.
struct QkvPost {
    const bf16_t* qkv; int64_t ldqkv; const bf16_t* qw; const bf16_t* kw; const float* cosT; const float* sinT; const int32_t* pos; const int32_t* write_pos; int capacity; float eps;
};

// QK-RMSNorm + RoPE of one head in two stages (request the operands; later, finish): lanes 0 .. LPH-1 hold the head, 8 features of each half per lane.
struct HeadOperands {
    u32x4 w1, w2, x1, x2;
    f32x4 c1a, c1b, s1a, s1b, c2a, c2b, s2a, s2b;
    template <int D>
    __device__ __forceinline__ void load(const bf16_t* __restrict__ src, const bf16_t* __restrict__ w, const float* __restrict__ cosT, const float* __restrict__ sinT, int64_t p, int lane) {
        constexpr int HALF = D / 2, LPH = HALF / 8;
        const int i = (lane % LPH) * 8;
        x1 = *reinterpret_cast<const u32x4*>(src + i);
        x2 = *reinterpret_cast<const u32x4*>(src + HALF + i);
        w1 = *reinterpret_cast<const u32x4*>(w + i);
        w2 = *reinterpret_cast<const u32x4*>(w + HALF + i);
        const f32x4 *c1 = reinterpret_cast<const f32x4*>(cosT + p * D + i), *s1 = reinterpret_cast<const f32x4*>(sinT + p * D + i);
        const f32x4 *c2 = reinterpret_cast<const f32x4*>(cosT + p * D + HALF + i), *s2 = reinterpret_cast<const f32x4*>(sinT + p * D + HALF + i);
        c1a = c1[0]; c1b = c1[1]; s1a = s1[0]; s1b = s1[1]; c2a = c2[0]; c2b = c2[1]; s2a = s2[0]; s2b = s2[1];
    }
    template <int D>
    __device__ __forceinline__ void finish(float eps, float (&y1)[8], float (&y2)[8]) const {
        constexpr int HALF = D / 2, LPH = HALF / 8;
        float wf1[8], wf2[8], xf1[8], xf2[8], cb1[8], sb1[8], cb2[8], sb2[8];
        unpack8(w1, wf1); unpack8(w2, wf2); unpack8(x1, xf1); unpack8(x2, xf2);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            cb1[e] = rbf(c1a[e]); cb1[4 + e] = rbf(c1b[e]); sb1[e] = rbf(s1a[e]); sb1[4 + e] = rbf(s1b[e]);
            cb2[e] = rbf(c2a[e]); cb2[4 + e] = rbf(c2b[e]); sb2[e] = rbf(s2a[e]); sb2[4 + e] = rbf(s2b[e]);
        }
        float ss = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) ss += xf1[e] * xf1[e] + xf2[e] * xf2[e];
#pragma unroll
        for (int o = LPH / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
        const float r = rsqrtf(ss / (float)D + eps);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float n1 = rbf(xf1[e] * r * wf1[e]), n2 = rbf(xf2[e] * r * wf2[e]);
            y1[e] = rbf(rbf(cb1[e] * n1) + rbf(sb1[e] * (-n2)));
            y2[e] = rbf(rbf(cb2[e] * n2) + rbf(sb2[e] * n1));
        }
    }
};

template <int D, int WAVES, bool POST>
__global__ __launch_bounds__(64 * WAVES) void attn_decode_kernel(int Hq, int Hkv, const bf16_t* __restrict__ q, const bf16_t* kc, const bf16_t* vc, int64_t batch_stride,
                                                                 int64_t ld, int len, const int32_t* __restrict__ len_dev, const uint8_t* __restrict__ key_mask, int64_t ldm,
                                                                 bf16_t* __restrict__ o, float scale_log2, QkvPost post, float* __restrict__ ws) {
    // One mapping for both products: lane = (key group kg, 16-byte feature chunk ch); a load instruction covers KG whole key rows, coalesced.  Scores: 8 features per lane,
    // folded over the CH lanes of a key by a butterfly, so every lane of a key group holds the scores of its NT keys and the probabilities never leave registers.
    // A one-token step is a chain of memory round trips, so the chain is kept at two: (1) the device-side length / positions, (2) EVERYTHING else at once -- the wave's
    // first chunk of key AND value rows is requested before the query exists (and before the QKV post-processing of POST runs); the new token's own key / value never
    // come back from memory (POST: they are used from LDS, the cache row is written for the later steps only).
    constexpr int CH = D / 8, KG = 64 / CH;
    constexpr int NT = CH < 8 ? CH : 8;    // keys per lane and chunk: 2 x NT 16-byte rows in flight per lane
    constexpr int CK = NT * KG;           // keys per chunk; chunk c belongs to split c % S (blockIdx.z of S = gridDim.z), there to wave (c / S) % WAVES
    // S > 1 (long caches; a workgroup streams ~ 256 keys per memory round trip): every split leaves its un-normalised context, maximum and sum in `ws`
    // ([B, Hq, S, D + 2] floats), attn_decode_combine_kernel folds them in split order.
    __shared__ float qs[D];
    __shared__ float ksn[POST ? D : 1], vsn[POST ? D : 1];
    __shared__ float part[WAVES][D + 2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kg = lane / CH, ch = lane % CH;
    const int h = blockIdx.x, b = blockIdx.y, hk = h / (Hq / Hkv);
    const int S = gridDim.z, sp = blockIdx.z;
    const int c0 = sp + S * wave, cstep = S * WAVES;
    if (len_dev) len = min(len, *len_dev);  // graph replay: the current length lives on the device, `len` is the capacity bound
    int wp = -1;                            // POST: the cache row of the new token (its key / value are taken from LDS)
    int64_t p = 0;
    if constexpr (POST) {
        wp = *post.write_pos;
        p = post.pos[b];
    }
    const bf16_t* kb = kc + b * batch_stride + (int64_t)hk * D + ch * 8;
    const bf16_t* vb = vc + b * batch_stride + (int64_t)hk * D + ch * 8;
    u32x4 kraw[NT], vraw[NT];
    auto request_k = [&](int base) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int j = base + t * KG + kg;
            kraw[t] = (j < len && j != wp) ? *reinterpret_cast<const u32x4*>(kb + (int64_t)j * ld) : (u32x4){0u, 0u, 0u, 0u};
        }
    };
    auto request_v = [&](int base) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int j = base + t * KG + kg;
            vraw[t] = (j < len && j != wp) ? *reinterpret_cast<const u32x4*>(vb + (int64_t)j * ld) : (u32x4){0u, 0u, 0u, 0u};
        }
    };
    if constexpr (POST) {
        static_assert(D == 64 || D == 128, "QK-norm + RoPE geometry: 8 features per lane and half");
        constexpr int HALF = D / 2, LPH = HALF / 8;
        const bool room = wp >= 0 && wp < post.capacity;  // never write outside the cache
        const bf16_t* row = post.qkv + b * post.ldqkv;
        const int i = (lane % LPH) * 8;
        // loads return in order: the post-processing operands (hot in L2) are requested in front of the cache rows (HBM), their arithmetic runs under the rows' flight
        HeadOperands hop;
        u32x4 vnew = {0u, 0u, 0u, 0u};
        if (wave == 0) hop.load<D>(row + (int64_t)h * D, post.qw, post.cosT, post.sinT, p, lane);
        else if (wave == 1) hop.load<D>(row + (int64_t)(Hq + hk) * D, post.kw, post.cosT, post.sinT, p, lane);
        else if (wave == 2 && lane < D / 8) vnew = *reinterpret_cast<const u32x4*>(row + (int64_t)(Hq + Hkv + hk) * D + lane * 8);
        request_k(c0 * CK);
        request_v(c0 * CK);
        if (wave == 0) {
            float y1[8], y2[8];
            hop.finish<D>(post.eps, y1, y2);
            if (lane < LPH) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { qs[i + e] = y1[e]; qs[HALF + i + e] = y2[e]; }
            }
        } else if (wave == 1) {
            float y1[8], y2[8];
            hop.finish<D>(post.eps, y1, y2);
            if (lane < LPH) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { ksn[i + e] = y1[e]; ksn[HALF + i + e] = y2[e]; }
                if (room) {
                    bf16_t* dst = const_cast<bf16_t*>(kc) + b * batch_stride + (int64_t)wp * ld + (int64_t)hk * D;
                    *reinterpret_cast<u32x4*>(dst + i) = pack8(y1);
                    *reinterpret_cast<u32x4*>(dst + HALF + i) = pack8(y2);
                }
            }
        } else if (wave == 2) {
            if (lane < D / 8) {
                float vf[8];
                unpack8(vnew, vf);
#pragma unroll
                for (int e = 0; e < 8; ++e) vsn[lane * 8 + e] = vf[e];
                if (room) *reinterpret_cast<u32x4*>(const_cast<bf16_t*>(vc) + b * batch_stride + (int64_t)wp * ld + (int64_t)hk * D + lane * 8) = vnew;
            }
        }
    } else {
        for (int i = threadIdx.x; i < D; i += 64 * WAVES) qs[i] = bf2f(q[((int64_t)b * Hq + h) * D + i]);
        request_k(c0 * CK);
        request_v(c0 * CK);
    }
    __syncthreads();
    const uint8_t* km = key_mask ? key_mask + b * ldm : nullptr;
    float qf[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) qf[e] = qs[ch * 8 + e];
    float m = -__builtin_huge_valf(), l = 0.f, acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int base = c0 * CK; base < len; base += cstep * CK) {
        const int nb = base + cstep * CK;
        float sc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int j = base + t * KG + kg;
            float kf[8];
            unpack8(kraw[t], kf);
            if constexpr (POST) {
                if (j == wp) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) kf[e] = ksn[ch * 8 + e];
                }
            }
            float d = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) d = fmaf(kf[e], qf[e], d);
#pragma unroll
            for (int o = 1; o < CH; o <<= 1) d += __shfl_xor(d, o, 64);
            sc[t] = j < len ? ((km && !km[j]) ? MASK_T : d * scale_log2) : -__builtin_huge_valf();  // keys beyond the cache do not exist
        }
        if (nb < len) request_k(nb);  // the next round's rows travel under this round's arithmetic
        float mc = sc[0];
#pragma unroll
        for (int t = 1; t < NT; ++t) mc = fmaxf(mc, sc[t]);
#pragma unroll
        for (int o = CH; o < 64; o <<= 1) mc = fmaxf(mc, __shfl_xor(mc, o, 64));
        const float mn = fmaxf(m, mc);
        const float alpha = exp2f(m - mn);  // m = -inf on the first chunk -> 0
        float psum = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            sc[t] = exp2f(sc[t] - mn);  // -inf -> 0
            psum += sc[t];
        }
#pragma unroll
        for (int o = CH; o < 64; o <<= 1) psum += __shfl_xor(psum, o, 64);
        l = l * alpha + psum;
        m = mn;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] *= alpha;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int j = base + t * KG + kg;
            float vf[8];
            unpack8(vraw[t], vf);
            if constexpr (POST) {
                if (j == wp) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) vf[e] = vsn[ch * 8 + e];
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = fmaf(sc[t], vf[e], acc[e]);  // rows beyond the cache: probability 0 times a zero row
        }
        if (nb < len) request_v(nb);
    }
    // fold the key groups of the wave (lanes with the same chunk), then combine the waves' key ranges
#pragma unroll
    for (int e = 0; e < 8; ++e)
        for (int o = CH; o < 64; o <<= 1) acc[e] += __shfl_xor(acc[e], o, 64);
    if (kg == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) part[wave][ch * 8 + e] = acc[e];
    }
    if (lane == 0) {
        part[wave][D] = m;
        part[wave][D + 1] = l;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < D; i += 64 * WAVES) {
        float mg = -__builtin_huge_valf();
        for (int w = 0; w < WAVES; ++w) mg = fmaxf(mg, part[w][D]);
        float lg = 0.f, out = 0.f;
        for (int w = 0; w < WAVES; ++w) {
            const float mw = part[w][D];
            const float f = mw == -__builtin_huge_valf() ? 0.f : exp2f(mw - mg);
            lg += part[w][D + 1] * f;
            out += part[w][i] * f;
        }
        if (S == 1) {
            o[((int64_t)b * Hq + h) * D + i] = f2bf(out / lg);
        } else {
            float* dst = ws + (((int64_t)b * Hq + h) * S + sp) * (D + 2);
            dst[i] = out;
            if (i == 0) { dst[D] = mg; dst[D + 1] = lg; }
        }
    }
}

// The splits of one (sequence, head) folded in split order: grid (Hq, B), D threads.
__global__ void attn_decode_combine_kernel(int Hq, int D, int S, const float* __restrict__ ws, bf16_t* __restrict__ o) {
    const int h = blockIdx.x, b = blockIdx.y, i = threadIdx.x;
    const float* src = ws + ((int64_t)b * Hq + h) * S * (D + 2);
    float mg = -__builtin_huge_valf();
    for (int s = 0; s < S; ++s) mg = fmaxf(mg, src[s * (D + 2) + D]);
    float lg = 0.f, out = 0.f;
    for (int s = 0; s < S; ++s) {
        const float ms = src[s * (D + 2) + D];
        const float f = ms == -__builtin_huge_valf() ? 0.f : exp2f(ms - mg);  // a split without keys
        lg += src[s * (D + 2) + D + 1] * f;
        out += src[s * (D + 2) + i] * f;
    }
    o[((int64_t)b * Hq + h) * D + i] = f2bf(out / lg);
}

// The tail of a greedy step: the chosen ids become the next input ids, every counter of the device-side bookkeeping moves on by one (one launch instead of four).
__global__ void decode_advance_kernel(int B, const int64_t* __restrict__ next, int64_t* __restrict__ tok, int32_t* __restrict__ rope_pos, int32_t* __restrict__ write_pos,
                                      int32_t* __restrict__ length) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i < B) {
        tok[i] = next[i];
        rope_pos[i] += 1;
    }
    if (i == 0) {
        *write_pos += 1;
        *length += 1;
    }
}

// cache[b, pos[0], :] = rows[b, :] for K and V: the KVCache append of one decoded token with the position read from the device
__global__ void kv_append_kernel(int B, int width, const bf16_t* __restrict__ k_rows, int64_t ldk, const bf16_t* __restrict__ v_rows, int64_t ldv,
                                 bf16_t* __restrict__ kc, bf16_t* __restrict__ vc, int64_t batch_stride, int64_t ld, int capacity,
                                 const int32_t* __restrict__ pos) {
    const int vec = width >> 3;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)B * vec) return;
    const int b = (int)(idx / vec), c = (int)(idx % vec);
    const int p = *pos;
    if (p < 0 || p >= capacity) return;  // never write outside the cache
    *reinterpret_cast<u32x4*>(kc + b * batch_stride + (int64_t)p * ld + c * 8) = *reinterpret_cast<const u32x4*>(k_rows + b * ldk + c * 8);
    *reinterpret_cast<u32x4*>(vc + b * batch_stride + (int64_t)p * ld + c * 8) = *reinterpret_cast<const u32x4*>(v_rows + b * ldv + c * 8);
}

// two stages so that a handful of rows still fills the chip: stage 1 = ARG_SPLIT column ranges per row, stage 2 = merge
constexpr int ARG_SPLIT = 64;
__device__ __forceinline__ void argmax_merge(float& best, int64_t& idx, float ov, int64_t oi) {
    if (ov > best || (ov == best && oi < idx)) {
        best = ov;
        idx = oi;
    }
}
__global__ __launch_bounds__(256) void argmax_part_kernel(int64_t V, const bf16_t* __restrict__ x, int64_t ld, float* __restrict__ pv, int64_t* __restrict__ pi) {
    __shared__ float bv[4];
    __shared__ int64_t bi[4];
    const bf16_t* r = x + (int64_t)blockIdx.y * ld;
    const int64_t span = (V + ARG_SPLIT - 1) / ARG_SPLIT, c0 = blockIdx.x * span, c1 = c0 + span < V ? c0 + span : V;
    float best = -__builtin_huge_valf();
    int64_t idx = 0x7fffffffffffffffll;
    for (int64_t i = c0 + threadIdx.x; i < c1; i += 256) {
        const float v = bf2f(r[i]);
        if (v > best) {
            best = v;
            idx = i;
        }
    }
    for (int o = 32; o > 0; o >>= 1) argmax_merge(best, idx, __shfl_xor(best, o, 64), __shfl_xor(idx, o, 64));
    if ((threadIdx.x & 63) == 0) {
        bv[threadIdx.x >> 6] = best;
        bi[threadIdx.x >> 6] = idx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) argmax_merge(best, idx, bv[w], bi[w]);
        pv[(int64_t)blockIdx.y * ARG_SPLIT + blockIdx.x] = best;
        pi[(int64_t)blockIdx.y * ARG_SPLIT + blockIdx.x] = idx;
    }
}
__global__ __launch_bounds__(64) void argmax_merge_kernel(const float* __restrict__ pv, const int64_t* __restrict__ pi, int64_t* __restrict__ out) {
    float best = pv[(int64_t)blockIdx.x * ARG_SPLIT + threadIdx.x];
    int64_t idx = pi[(int64_t)blockIdx.x * ARG_SPLIT + threadIdx.x];
    for (int o = 32; o > 0; o >>= 1) argmax_merge(best, idx, __shfl_xor(best, o, 64), __shfl_xor(idx, o, 64));
    if (threadIdx.x == 0) out[blockIdx.x] = idx;
}

}  // namespace

#define ST(s) ((hipStream_t)(s))

extern "C" int mi355_gemv_bf16(int M, int64_t N, int K, const void* x, int64_t ldx, const void* W, int64_t ldw, void* y, int64_t ldy,
                               const void* residual, int64_t ldr, void* stream) {
    MI355_REQUIRE(M >= 1 && M <= 8, "gemv_bf16: 1..8 rows (got %d); larger batches go through mi355_gemm_bf16", M);
    MI355_REQUIRE(N > 0 && K > 0 && K % 8 == 0 && ldx % 8 == 0 && ldw % 8 == 0 && ldx >= K && ldw >= K && ldy >= N, "gemv_bf16: K and the leading dimensions of x / W must be multiples of 8");
    MI355_REQUIRE(x && W && y && (!residual || ldr >= N), "gemv_bf16: null pointer or residual pitch too small");
    const int grid = (int)((N + 3) / 4);
#define LAUNCH(MT) gemv_kernel<MT><<<grid, 256, 0, ST(stream)>>>(M, N, K, (const bf16_t*)x, ldx, (const bf16_t*)W, ldw, (bf16_t*)y, ldy, (const bf16_t*)residual, ldr)
    if (M == 1) LAUNCH(1); else if (M == 2) LAUNCH(2); else if (M <= 4) LAUNCH(4); else LAUNCH(8);
#undef LAUNCH
    MI355_LAUNCH_CHECK("gemv_bf16");
    return 0;
}

extern "C" int mi355_gemv_bf16_pro(int M, int64_t N, int K, const void* x, int64_t ldx, int prologue, const void* norm_w, float eps, const void* W, int64_t ldw,
                                   void* y, int64_t ldy, const void* residual, int64_t ldr, void* stream) {
    MI355_REQUIRE(M >= 1 && M <= 8, "gemv_bf16_pro: 1..8 rows (got %d)", M);
    MI355_REQUIRE(prologue == 1 || prologue == 2, "gemv_bf16_pro: prologue 1 (RMSNorm of x) or 2 (SwiGLU of gate-up rows), got %d", prologue);
    MI355_REQUIRE(N > 0 && K > 0 && K % 8 == 0 && ldx % 8 == 0 && ldw % 8 == 0 && ldx >= (prologue == 2 ? 2 * K : K) && ldw >= K && ldy >= N,
                  "gemv_bf16_pro: K and the leading dimensions must be multiples of 8 (x rows hold 2K elements under prologue 2)");
    // (the RMSNorm prologue keeps eight row sums in 32 bytes of static LDS beside the dynamic image: together they must stay within the 64 KiB a launch gets by default)
    MI355_REQUIRE((int64_t)M * K * 2 + 32 <= 65536, "gemv_bf16_pro: the operand rows (%d x %d bf16) plus 32 bytes of row sums exceed the 64 KiB LDS image", M, K);
    MI355_REQUIRE(x && W && y && (!residual || ldr >= N) && (prologue != 1 || norm_w), "gemv_bf16_pro: null pointer or residual pitch too small");
    const int grid = (int)((N + 3) / 4 < 2048 ? (N + 3) / 4 : 2048);
    const size_t lds = (size_t)M * K * 2;
#define LAUNCH(MT, PRO) gemv_pro_kernel<MT, PRO><<<grid, 256, lds, ST(stream)>>>(M, N, K, (const bf16_t*)x, ldx, (const bf16_t*)norm_w, eps, (const bf16_t*)W, ldw, (bf16_t*)y, ldy, (const bf16_t*)residual, ldr)
    if (prologue == 1) { if (M == 1) LAUNCH(1, 1); else if (M == 2) LAUNCH(2, 1); else if (M <= 4) LAUNCH(4, 1); else LAUNCH(8, 1); }
    else { if (M == 1) LAUNCH(1, 2); else if (M == 2) LAUNCH(2, 2); else if (M <= 4) LAUNCH(4, 2); else LAUNCH(8, 2); }
#undef LAUNCH
    MI355_LAUNCH_CHECK("gemv_bf16_pro");
    return 0;
}

extern "C" int mi355_kv_append(int B, int width, const void* k_rows, int64_t ldk, const void* v_rows, int64_t ldv, void* k_cache, void* v_cache,
                               int64_t batch_stride, int64_t ld, int capacity, const int32_t* pos, void* stream) {
    MI355_REQUIRE(B > 0 && width > 0 && width % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ld % 8 == 0, "kv_append: width and pitches must be multiples of 8 elements");
    MI355_REQUIRE(k_rows && v_rows && k_cache && v_cache && pos && capacity > 0 && ld >= width && batch_stride >= (int64_t)capacity * ld, "kv_append: bad arguments");
    kv_append_kernel<<<(int)(((int64_t)B * (width / 8) + 255) / 256), 256, 0, ST(stream)>>>(B, width, (const bf16_t*)k_rows, ldk, (const bf16_t*)v_rows, ldv, (bf16_t*)k_cache, (bf16_t*)v_cache, batch_stride, ld, capacity, pos);
    MI355_LAUNCH_CHECK("kv_append");
    return 0;
}

static int attn_decode_launch(const char* who, int B, int Hq, int Hkv, int D, const void* q, const void* k_cache, const void* v_cache, int64_t batch_stride, int64_t ld, int len,
                              const int32_t* len_dev, const uint8_t* key_mask, int64_t ldm, void* o, float scale, const QkvPost* post, int splits, float* workspace,
                              void* stream) {
    MI355_REQUIRE(B > 0 && Hq > 0 && Hkv > 0 && Hq % Hkv == 0 && len > 0, "%s: bad sizes", who);
    MI355_REQUIRE(k_cache && v_cache && o && ld >= (int64_t)Hkv * D && ld % 8 == 0 && batch_stride >= (int64_t)len * ld, "%s: cache pitch / stride too small", who);
    MI355_REQUIRE(!key_mask || ldm >= len, "%s: key mask pitch smaller than the cache length", who);
    MI355_REQUIRE(B <= 65535, "%s: grid limits", who);
    MI355_REQUIRE(splits >= 1 && splits <= 64 && (splits == 1 || workspace), "%s: 1..64 splits, and a workspace of B * Hq * splits * (D + 2) floats for more than one", who);
    dim3 grid(Hq, B, splits);
    const QkvPost none = {};
#define LAUNCH(DD, WV, PP) attn_decode_kernel<DD, WV, PP><<<grid, 64 * WV, 0, ST(stream)>>>(Hq, Hkv, (const bf16_t*)q, (const bf16_t*)k_cache, (const bf16_t*)v_cache, batch_stride, ld, len, len_dev, key_mask, ldm, (bf16_t*)o, scale * LOG2E, PP ? *post : none, workspace)
    if (post) {
        if (D == 64) LAUNCH(64, 8, true); else LAUNCH(128, 8, true);
    } else {
        if (D == 32) LAUNCH(32, 8, false); else if (D == 64) LAUNCH(64, 8, false); else if (D == 128) LAUNCH(128, 8, false); else LAUNCH(256, 8, false);
    }
#undef LAUNCH
    if (splits > 1) attn_decode_combine_kernel<<<dim3(Hq, B), D, 0, ST(stream)>>>(Hq, D, splits, workspace, (bf16_t*)o);
    return 0;
}

extern "C" int mi355_attn_decode(int B, int Hq, int Hkv, int D, const void* q, const void* k_cache, const void* v_cache, int64_t batch_stride,
                                 int64_t ld, int len, const int32_t* len_dev, const uint8_t* key_mask, int64_t ldm, void* o, float scale, int splits, float* workspace,
                                 void* stream) {
    MI355_REQUIRE(D == 32 || D == 64 || D == 128 || D == 256, "attn_decode: head_dim %d not built (32, 64, 128, 256)", D);
    MI355_REQUIRE(q != nullptr, "attn_decode: null query");
    if (int rc = attn_decode_launch("attn_decode", B, Hq, Hkv, D, q, k_cache, v_cache, batch_stride, ld, len, len_dev, key_mask, ldm, o, scale, nullptr, splits, workspace, stream)) return rc;
    MI355_LAUNCH_CHECK("attn_decode");
    return 0;
}

extern "C" int mi355_attn_decode_qkv(int B, int Hq, int Hkv, int D, const void* qkv, int64_t ldqkv, const void* q_norm_w, const void* k_norm_w, const float* cos,
                                     const float* sin, const int32_t* pos, void* k_cache, void* v_cache, int64_t batch_stride, int64_t ld, int capacity,
                                     const int32_t* write_pos, const int32_t* len_dev, const uint8_t* key_mask, int64_t ldm, void* o, float scale, float eps, int splits,
                                     float* workspace, void* stream) {
    MI355_REQUIRE(D == 128 || D == 64, "attn_decode_qkv: head_dim must be 64 or 128 (got %d)", D);
    MI355_REQUIRE(Hq > 0 && Hkv > 0 && ldqkv >= (int64_t)(Hq + 2 * Hkv) * D && ldqkv % 8 == 0, "attn_decode_qkv: the qkv row holds (Hq + 2 Hkv) * D elements, pitch a multiple of 8");
    MI355_REQUIRE(qkv && q_norm_w && k_norm_w && cos && sin && pos && write_pos && len_dev, "attn_decode_qkv: null pointer (the write position and the length live on the device)");
    const QkvPost post = {(const bf16_t*)qkv, ldqkv, (const bf16_t*)q_norm_w, (const bf16_t*)k_norm_w, cos, sin, pos, write_pos, capacity, eps};
    if (int rc = attn_decode_launch("attn_decode_qkv", B, Hq, Hkv, D, nullptr, k_cache, v_cache, batch_stride, ld, capacity, len_dev, key_mask, ldm, o, scale, &post, splits, workspace, stream)) return rc;
    MI355_LAUNCH_CHECK("attn_decode_qkv");
    return 0;
}

extern "C" int mi355_decode_advance(int B, const int64_t* next_ids, int64_t* tok, int32_t* rope_pos, int32_t* write_pos, int32_t* length, void* stream) {
    MI355_REQUIRE(B > 0 && next_ids && tok && rope_pos && write_pos && length, "decode_advance: bad arguments");
    decode_advance_kernel<<<(B + 63) / 64, 64, 0, ST(stream)>>>(B, next_ids, tok, rope_pos, write_pos, length);
    MI355_LAUNCH_CHECK("decode_advance");
    return 0;
}

extern "C" int mi355_argmax_rows(int64_t rows, int64_t V, const void* logits, int64_t ld, int64_t* out, void* workspace, void* stream) {
    MI355_REQUIRE(rows > 0 && rows <= 65535 && V > 0 && logits && out && workspace && ld >= V, "argmax_rows: bad arguments (workspace: rows * 64 * 12 bytes)");
    static_assert(ARG_SPLIT == 64, "the merge kernel is one wave");
    float* pv = (float*)workspace;
    int64_t* pi = (int64_t*)((char*)workspace + ((rows * ARG_SPLIT * 4 + 7) / 8) * 8);
    argmax_part_kernel<<<dim3(ARG_SPLIT, (unsigned)rows), 256, 0, ST(stream)>>>(V, (const bf16_t*)logits, ld, pv, pi);
    MI355_LAUNCH_CHECK("argmax_rows(part)");
    argmax_merge_kernel<<<(int)rows, 64, 0, ST(stream)>>>(pv, pi, out);
    MI355_LAUNCH_CHECK("argmax_rows");
    return 0;
}
