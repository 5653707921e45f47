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
constexpr int DEC_WAVES = 8;   // waves (key ranges) per (batch, head) of the decode attention
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

// ------------------------------------------------------------------------------------------- decode attention
// grid (Hq, B), 8 waves (512 threads leave 256 registers per lane: the unrolled score loop holds a 256-byte key row + the query).  Wave w owns keys [w * span, (w+1) * span); per chunk of 64 keys: lane = key for the scores (the
// query comes from LDS, broadcast), then lane = (key group, 16-byte feature chunk) for P V.  D = 64, 128, 256.
template <int D>
__global__ __launch_bounds__(64 * DEC_WAVES) void attn_decode_kernel(int Hq, int Hkv, const bf16_t* __restrict__ q, const bf16_t* __restrict__ kc,
                                                          const bf16_t* __restrict__ vc, int64_t batch_stride, int64_t ld, int len,
                                                          const int32_t* __restrict__ len_dev, const uint8_t* __restrict__ key_mask, int64_t ldm,
                                                          bf16_t* __restrict__ o, float scale_log2) {
    constexpr int CH = D / 8, KG = 64 / CH;  // P V phase: lane = (key group, 16-byte feature chunk); KG keys per load instruction
    __shared__ float qs[D];
    __shared__ float ps[DEC_WAVES][64];
    __shared__ float part[DEC_WAVES][D + 2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kg = lane / CH, ch = lane % CH;
    const int h = blockIdx.x, b = blockIdx.y, hk = h / (Hq / Hkv);
    if (len_dev) len = min(len, *len_dev);  // graph replay: the current length lives on the device, `len` is the capacity bound
    for (int i = threadIdx.x; i < D; i += 64 * DEC_WAVES) qs[i] = bf2f(q[((int64_t)b * Hq + h) * D + i]);
    __syncthreads();
    const bf16_t* kb = kc + b * batch_stride + (int64_t)hk * D;
    const bf16_t* vb = vc + b * batch_stride + (int64_t)hk * D + ch * 8;
    const uint8_t* km = key_mask ? key_mask + b * ldm : nullptr;
    const int span = ((len + DEC_WAVES - 1) / DEC_WAVES + 63) / 64 * 64;
    const int j0 = wave * span, j1 = min(len, j0 + span);
    float m = -__builtin_huge_valf(), l = 0.f, acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int base = j0; base < j1; base += 64) {
        const int j = base + lane;
        float s = -__builtin_huge_valf();  // keys beyond the cache do not exist
        if (j < j1) {
            const bf16_t* kr = kb + (int64_t)j * ld;
            float d = 0.f;
            constexpr int UNR = D > 128 ? 8 : D / 8;  // a fully unrolled 512-byte row + the query would not fit the register file
#pragma unroll UNR
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
        for (int e = 0; e < 8; ++e) acc[e] *= alpha;
        __builtin_amdgcn_wave_barrier();
        const int nk = min(64, j1 - base);
#pragma unroll 4
        for (int t = kg; t < nk; t += KG) {  // independent 16-byte loads: several keys in flight per lane
            const float pt = ps[wave][t];
            float vf[8];
            unpack8(*reinterpret_cast<const u32x4*>(vb + (int64_t)(base + t) * ld), vf);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = fmaf(pt, vf[e], acc[e]);
        }
        __builtin_amdgcn_wave_barrier();
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
    for (int i = threadIdx.x; i < D; i += 64 * DEC_WAVES) {
        float mg = -__builtin_huge_valf();
        for (int w = 0; w < DEC_WAVES; ++w) mg = fmaxf(mg, part[w][D]);
        float lg = 0.f, out = 0.f;
        for (int w = 0; w < DEC_WAVES; ++w) {
            const float mw = part[w][D];
            const float f = mw == -__builtin_huge_valf() ? 0.f : exp2f(mw - mg);
            lg += part[w][D + 1] * f;
            out += part[w][i] * f;
        }
        o[((int64_t)b * Hq + h) * D + i] = f2bf(out / lg);
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

extern "C" int mi355_kv_append(int B, int width, const void* k_rows, int64_t ldk, const void* v_rows, int64_t ldv, void* k_cache, void* v_cache,
                               int64_t batch_stride, int64_t ld, int capacity, const int32_t* pos, void* stream) {
    MI355_REQUIRE(B > 0 && width > 0 && width % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ld % 8 == 0, "kv_append: width and pitches must be multiples of 8 elements");
    MI355_REQUIRE(k_rows && v_rows && k_cache && v_cache && pos && capacity > 0 && ld >= width && batch_stride >= (int64_t)capacity * ld, "kv_append: bad arguments");
    kv_append_kernel<<<(int)(((int64_t)B * (width / 8) + 255) / 256), 256, 0, ST(stream)>>>(B, width, (const bf16_t*)k_rows, ldk, (const bf16_t*)v_rows, ldv, (bf16_t*)k_cache, (bf16_t*)v_cache, batch_stride, ld, capacity, pos);
    MI355_LAUNCH_CHECK("kv_append");
    return 0;
}

extern "C" int mi355_attn_decode(int B, int Hq, int Hkv, int D, const void* q, const void* k_cache, const void* v_cache, int64_t batch_stride,
                                 int64_t ld, int len, const int32_t* len_dev, const uint8_t* key_mask, int64_t ldm, void* o, float scale, void* stream) {
    MI355_REQUIRE(B > 0 && Hq > 0 && Hkv > 0 && Hq % Hkv == 0 && len > 0, "attn_decode: bad sizes");
    MI355_REQUIRE(D == 32 || D == 64 || D == 128 || D == 256, "attn_decode: head_dim %d not built (32, 64, 128, 256)", D);
    MI355_REQUIRE(q && k_cache && v_cache && o && ld >= (int64_t)Hkv * D && ld % 8 == 0 && batch_stride >= (int64_t)len * ld, "attn_decode: cache pitch / stride too small");
    MI355_REQUIRE(!key_mask || ldm >= len, "attn_decode: key mask pitch smaller than the cache length");
    MI355_REQUIRE(B <= 65535, "attn_decode: grid limits");
    dim3 grid(Hq, B);
#define LAUNCH(DD) attn_decode_kernel<DD><<<grid, 64 * DEC_WAVES, 0, ST(stream)>>>(Hq, Hkv, (const bf16_t*)q, (const bf16_t*)k_cache, (const bf16_t*)v_cache, batch_stride, ld, len, len_dev, key_mask, ldm, (bf16_t*)o, scale * LOG2E)
    if (D == 32) LAUNCH(32); else if (D == 64) LAUNCH(64); else if (D == 128) LAUNCH(128); else LAUNCH(256);
#undef LAUNCH
    MI355_LAUNCH_CHECK("attn_decode");
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
