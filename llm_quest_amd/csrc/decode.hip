// KV-cache decoding on the step's right edge: the per-token path of generate_loop_kv_cache
// (llm_quest/generate.py:97-151) through Qwen3 with utils.KVCache (utils.py:409-531).  One new token per sequence makes every
// linear layer a matrix-VECTOR product and attention one query row against the cache: both are HBM-bound byte streams (weights,
// resp. cached K/V read exactly once per token), so the kernels here are bandwidth kernels, not MFMA kernels.
//   gemv            y[m, :] = x[m, :] W^T (+ residual), m < 8 rows: one wave per output column, 16-byte weight loads, fp32 accumulate
//   attn_decode     one query per (batch, head) over `len` cached keys: 4 waves split the keys (flash-decoding), scores with the
//                   key on the lane, P V with the feature on the lane, combined through LDS; reference mask semantics (finite fill)
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
__global__ __launch_bounds__(256) void gemv_kernel(int64_t N, int K, const bf16_t* __restrict__ x, int64_t ldx, const bf16_t* __restrict__ W,
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
            float xf[8];
            unpack8(*reinterpret_cast<const u32x4*>(x + m * ldx + k0), xf);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[m] = fmaf(wf[e], xf[e], acc[m]);
        }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const float s = wave_sum(acc[m]);
        if (lane == 0) y[m * ldy + n] = f2bf(res ? s + bf2f(res[m * ldr + n]) : s);
    }
}

// ------------------------------------------------------------------------------------------- decode attention
// grid (Hq, B), 256 threads.  Wave w owns keys [w * span, (w+1) * span); per chunk of 64 keys: lane = key for the scores (the
// query comes from LDS, broadcast), then lane = feature pair for P V.  D <= 256, D % 64 == 0... (D = 64, 128, 256).
template <int D>
__global__ __launch_bounds__(256) void attn_decode_kernel(int Hq, int Hkv, const bf16_t* __restrict__ q, const bf16_t* __restrict__ kc,
                                                          const bf16_t* __restrict__ vc, int64_t batch_stride, int64_t ld, int len,
                                                          const uint8_t* __restrict__ key_mask, int64_t ldm, bf16_t* __restrict__ o, float scale_log2) {
    constexpr int EPL = D / 64;  // features per lane in the P V phase
    __shared__ float qs[D];
    __shared__ float ps[4][64];
    __shared__ float part[4][D + 2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = blockIdx.x, b = blockIdx.y, hk = h / (Hq / Hkv);
    for (int i = threadIdx.x; i < D; i += 256) qs[i] = bf2f(q[((int64_t)b * Hq + h) * D + i]);
    __syncthreads();
    const bf16_t* kb = kc + b * batch_stride + (int64_t)hk * D;
    const bf16_t* vb = vc + b * batch_stride + (int64_t)hk * D;
    const uint8_t* km = key_mask ? key_mask + b * ldm : nullptr;
    const int span = ((len + 3) / 4 + 63) / 64 * 64;
    const int j0 = wave * span, j1 = min(len, j0 + span);
    float m = -__builtin_huge_valf(), l = 0.f, acc[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) acc[e] = 0.f;
    for (int base = j0; base < j1; base += 64) {
        const int j = base + lane;
        float s = -__builtin_huge_valf();  // keys beyond the cache do not exist
        if (j < j1) {
            const bf16_t* kr = kb + (int64_t)j * ld;
            float d = 0.f;
#pragma unroll
            for (int c = 0; c < D / 8; ++c) {
                float kf[8];
                unpack8(*reinterpret_cast<const u32x4*>(kr + 8 * c), kf);
#pragma unroll
                for (int e = 0; e < 8; ++e) d = fmaf(kf[e], qs[8 * c + e], d);
            }
            s = (km && !km[j]) ? MASK_T : d * scale_log2;
        }
        const float mc = wave_max(s);
        const float mn = fmaxf(m, mc);
        const float alpha = exp2f(m - mn);  // m = -inf on the first chunk -> 0
        const float p = j < j1 ? exp2f(s - mn) : 0.f;
        l = l * alpha + wave_sum(p);
        m = mn;
        ps[wave][lane] = p;
#pragma unroll
        for (int e = 0; e < EPL; ++e) acc[e] *= alpha;
        __builtin_amdgcn_wave_barrier();
        const int nk = min(64, j1 - base);
        for (int t = 0; t < nk; ++t) {
            const float pt = ps[wave][t];
            const bf16_t* vr = vb + (int64_t)(base + t) * ld;
#pragma unroll
            for (int e = 0; e < EPL; ++e) acc[e] = fmaf(pt, bf2f(vr[lane + 64 * e]), acc[e]);
        }
        __builtin_amdgcn_wave_barrier();
    }
    // combine the four key ranges
#pragma unroll
    for (int e = 0; e < EPL; ++e) part[wave][lane + 64 * e] = acc[e];
    if (lane == 0) {
        part[wave][D] = m;
        part[wave][D + 1] = l;
    }
    __syncthreads();
    if (wave == 0) {
        float mg = -__builtin_huge_valf();
        for (int w = 0; w < 4; ++w) mg = fmaxf(mg, part[w][D]);
        float lg = 0.f, out[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) out[e] = 0.f;
        for (int w = 0; w < 4; ++w) {
            const float mw = part[w][D];
            const float f = mw == -__builtin_huge_valf() ? 0.f : exp2f(mw - mg);
            lg += part[w][D + 1] * f;
#pragma unroll
            for (int e = 0; e < EPL; ++e) out[e] += part[w][lane + 64 * e] * f;
        }
#pragma unroll
        for (int e = 0; e < EPL; ++e) o[((int64_t)b * Hq + h) * D + lane + 64 * e] = f2bf(out[e] / lg);
    }
}

__global__ __launch_bounds__(256) void argmax_rows_kernel(int64_t V, const bf16_t* __restrict__ x, int64_t ld, int64_t* __restrict__ out) {
    __shared__ float bv[4];
    __shared__ int64_t bi[4];
    const bf16_t* r = x + (int64_t)blockIdx.x * ld;
    float best = -__builtin_huge_valf();
    int64_t idx = 0x7fffffffffffffffll;
    for (int64_t i = threadIdx.x; i < V; i += 256) {
        const float v = bf2f(r[i]);
        if (v > best) {
            best = v;
            idx = i;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int64_t oi = __shfl_xor(idx, o, 64);
        if (ov > best || (ov == best && oi < idx)) {
            best = ov;
            idx = oi;
        }
    }
    if ((threadIdx.x & 63) == 0) {
        bv[threadIdx.x >> 6] = best;
        bi[threadIdx.x >> 6] = idx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w)
            if (bv[w] > best || (bv[w] == best && bi[w] < idx)) {
                best = bv[w];
                idx = bi[w];
            }
        out[blockIdx.x] = idx;
    }
}

}  // namespace

#define ST(s) ((hipStream_t)(s))

extern "C" int mi355_gemv_bf16(int M, int64_t N, int K, const void* x, int64_t ldx, const void* W, int64_t ldw, void* y, int64_t ldy,
                               const void* residual, int64_t ldr, void* stream) {
    MI355_REQUIRE(M >= 1 && M <= 8, "gemv_bf16: 1..8 rows (got %d); larger batches go through mi355_gemm_bf16", M);
    MI355_REQUIRE(N > 0 && K > 0 && K % 8 == 0 && ldx % 8 == 0 && ldw % 8 == 0 && ldx >= K && ldw >= K && ldy >= N, "gemv_bf16: K and the leading dimensions of x / W must be multiples of 8");
    MI355_REQUIRE(x && W && y && (!residual || ldr >= N), "gemv_bf16: null pointer or residual pitch too small");
    const int grid = (int)((N + 3) / 4);
#define LAUNCH(MT) gemv_kernel<MT><<<grid, 256, 0, ST(stream)>>>(N, K, (const bf16_t*)x, ldx, (const bf16_t*)W, ldw, (bf16_t*)y, ldy, (const bf16_t*)residual, ldr)
    if (M == 1) LAUNCH(1); else if (M == 2) LAUNCH(2); else if (M <= 4) LAUNCH(4); else LAUNCH(8);
#undef LAUNCH
    MI355_LAUNCH_CHECK("gemv_bf16");
    return 0;
}

extern "C" int mi355_attn_decode(int B, int Hq, int Hkv, int D, const void* q, const void* k_cache, const void* v_cache, int64_t batch_stride,
                                 int64_t ld, int len, const uint8_t* key_mask, int64_t ldm, void* o, float scale, void* stream) {
    MI355_REQUIRE(B > 0 && Hq > 0 && Hkv > 0 && Hq % Hkv == 0 && len > 0, "attn_decode: bad sizes");
    MI355_REQUIRE(D == 64 || D == 128 || D == 256, "attn_decode: head_dim %d not built (64, 128, 256)", D);
    MI355_REQUIRE(q && k_cache && v_cache && o && ld >= (int64_t)Hkv * D && ld % 8 == 0 && batch_stride >= (int64_t)len * ld, "attn_decode: cache pitch / stride too small");
    MI355_REQUIRE(!key_mask || ldm >= len, "attn_decode: key mask pitch smaller than the cache length");
    MI355_REQUIRE(B <= 65535, "attn_decode: grid limits");
    dim3 grid(Hq, B);
#define LAUNCH(DD) attn_decode_kernel<DD><<<grid, 256, 0, ST(stream)>>>(Hq, Hkv, (const bf16_t*)q, (const bf16_t*)k_cache, (const bf16_t*)v_cache, batch_stride, ld, len, key_mask, ldm, (bf16_t*)o, scale * LOG2E)
    if (D == 64) LAUNCH(64); else if (D == 128) LAUNCH(128); else LAUNCH(256);
#undef LAUNCH
    MI355_LAUNCH_CHECK("attn_decode");
    return 0;
}

extern "C" int mi355_argmax_rows(int64_t rows, int64_t V, const void* logits, int64_t ld, int64_t* out, void* stream) {
    MI355_REQUIRE(rows > 0 && rows <= 0x7fffffff && V > 0 && logits && out && ld >= V, "argmax_rows: bad arguments");
    argmax_rows_kernel<<<(int)rows, 256, 0, ST(stream)>>>(V, (const bf16_t*)logits, ld, out);
    MI355_LAUNCH_CHECK("argmax_rows");
    return 0;
}
