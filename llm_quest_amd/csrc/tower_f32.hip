// The frozen vision tower at the REFERENCE's precision (multimodal/vlm_engine.py:99-104 runs the ViT without autocast: fp32 tensors end to end,
// vit_attention.py:58-91, vit_transformer_block.py:106-127).  Two pieces the bf16 tower does not have:
//
//   * split3: an fp32 matrix as THREE bf16 column blocks, so that the bf16 MFMA GEMMs of gemm.hip compute an fp32-grade product without a new
//     kernel.  x = hi + lo + O(2^-17 |x|) with hi = bf16(x), lo = bf16(x - hi); activations are laid out [hi | lo | hi], weights [hi | hi | lo]
//     along K, so ONE NT GEMM with K' = 3 K accumulates a_hi w_hi + a_lo w_hi + a_hi w_lo in fp32 (the dropped a_lo w_lo term is 2^-18 relative).
//     HBM-bound: 4 bytes read, 6 written per element.
//   * attn_f32: softmax(Q K^T * scale) V on fp32 tensors with the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32: the fp32 vector rate, 1/16 of the bf16
//     rate -- the tower's attention is 4 % of its FLOPs).  S <= 288 keys, head_dim 64, no mask (ViT attends everywhere): a head's K and V sit
//     whole in LDS, a wave owns 32 queries and keeps its whole score block S^T [keys x 32 queries] in accumulators (query on the lane: the
//     softmax is lane-local plus ONE exchange between the two half-waves), and the accumulators are the B operands of O^T += V^T P^T as they
//     stand -- the K-steps of that product pair key a (half-wave 0) with key a + 4 (half-wave 1), which is exactly where the 32x32 accumulator
//     layout leaves them.
#include "common.h"

namespace {

inline int grid_for(int64_t work_items, int per_block) {
    int64_t g = (work_items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > 4096) g = 4096;
    return (int)g;
}

// rows x K fp32 (row pitch ldx) -> rows x 3K bf16 (dense).  WEIGHT = false: [hi | lo | hi]; true: [hi | hi | lo].  8 elements per thread and step.
template <bool WEIGHT>
__global__ __launch_bounds__(256) void split3_kernel(int64_t rows, int K, const float* __restrict__ x, int64_t ldx, bf16_t* __restrict__ y) {
    const int kv = K >> 3;
    const int64_t n = rows * kv;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / kv;
        const int c = (int)(i % kv) * 8;
        const float* s = x + r * ldx + c;
        const f32x4 a = *reinterpret_cast<const f32x4*>(s), b = *reinterpret_cast<const f32x4*>(s + 4);
        const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        float lo[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) lo[e] = v[e] - bf2f(f2bf(v[e]));  // exact in fp32 (hi keeps the leading 8 bits of v)
        const u32x4 h = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
        const u32x4 l = {pack_bf2(lo[0], lo[1]), pack_bf2(lo[2], lo[3]), pack_bf2(lo[4], lo[5]), pack_bf2(lo[6], lo[7])};
        bf16_t* d = y + r * 3 * (int64_t)K + c;
        *reinterpret_cast<u32x4*>(d) = h;
        *reinterpret_cast<u32x4*>(d + K) = WEIGHT ? h : l;
        *reinterpret_cast<u32x4*>(d + 2 * K) = WEIGHT ? l : h;
    }
}

constexpr float LOG2E = 1.4426950408889634f;
constexpr int AD = 64;        // head_dim
constexpr int KP = AD + 1;    // K row pitch in floats: the 32 lanes of a half-wave read one feature of 32 consecutive keys
__device__ __forceinline__ int acc_row32(int e, int half) { return (e & 3) + 8 * (e >> 2) + 4 * half; }

template <int NT>  // key tiles of 32: S <= 32 NT
__global__ __launch_bounds__(256, 1) void attn_f32_kernel(int B, int S, int H, const float* __restrict__ q, int64_t ldq, const float* __restrict__ k, int64_t ldk,
                                                         const float* __restrict__ v, int64_t ldv, float* __restrict__ o, int64_t ldo, float scale) {
    __shared__ __attribute__((aligned(16))) float sm[NT * 32 * (KP + AD)];  // static: up to 149 KB at NT = 9, one workgroup per CU
    float* Ks = sm;                  // [NT * 32][KP]
    float* Vs = sm + NT * 32 * KP;   // [NT * 32][AD]   (NT * 32 * KP is a multiple of 4 floats: 32 * 65 = 2080)
    const int nqb = (S + 127) / 128;
    const int qb = (int)blockIdx.x % nqb, bh = (int)blockIdx.x / nqb;
    const int h = bh % H, b = bh / H;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l32 = lane & 31;

    // the head's keys and values -> LDS (rows beyond S: zeros), 16 bytes per lane
    {
        const float* kb = k + (int64_t)b * S * ldk + (int64_t)h * AD;
        const float* vb = v + (int64_t)b * S * ldv + (int64_t)h * AD;
        for (int i = tid; i < NT * 32 * 16; i += 256) {
            const int row = i >> 4, ch = (i & 15) * 4;
            f32x4 kk = {0.f, 0.f, 0.f, 0.f}, vv = kk;
            if (row < S) {
                kk = *reinterpret_cast<const f32x4*>(kb + (int64_t)row * ldk + ch);
                vv = *reinterpret_cast<const f32x4*>(vb + (int64_t)row * ldv + ch);
            }
            float* kd = Ks + row * KP + ch;
            kd[0] = kk[0], kd[1] = kk[1], kd[2] = kk[2], kd[3] = kk[3];
            *reinterpret_cast<f32x4*>(Vs + row * AD + ch) = vv;
        }
    }
    const int qw = qb * 128 + wave * 32, qi = qw + l32;
    // the contraction over the 64 features runs in the order (kk, half) -> feature 32 half + kk on BOTH operands: a lane's 32 query values are contiguous
    float qreg[32];
    {
        const float* qp = q + ((int64_t)b * S + min(qi, S - 1)) * ldq + (int64_t)h * AD + half * 32;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(qp + c * 4);
            qreg[4 * c] = t[0], qreg[4 * c + 1] = t[1], qreg[4 * c + 2] = t[2], qreg[4 * c + 3] = t[3];
        }
    }
    __syncthreads();
    if (qw >= S) return;  // a wave without queries (after the barrier: it helped to load)

    f32x16 sacc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int e = 0; e < 16; ++e) sacc[t][e] = 0.f;
        const float* kr = Ks + (t * 32 + l32) * KP + half * 32;
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) sacc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(kr[kk], qreg[kk], sacc[t], 0, 0, 0);
    }
    // softmax over the keys of this lane's query: lane-local, then the two half-waves meet once
    const float c2 = scale * LOG2E;
    float m = -3.0e38f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            if (t * 32 + acc_row32(e, half) >= S) sacc[t][e] = -3.0e38f;
            m = fmaxf(m, sacc[t][e]);
        }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    const float mc = m * c2;
    float l = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float p = (t * 32 + acc_row32(e, half) >= S) ? 0.f : exp2f(fmaf(sacc[t][e], c2, -mc));
            sacc[t][e] = p;
            l += p;
        }
    l += __shfl_xor(l, 32, 64);

    f32x16 oacc[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[dt][e] = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float* vr = Vs + (t * 32 + acc_row32(e, half)) * AD + l32;  // this half-wave's key of the pair (a, a + 4)
            oacc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[0], sacc[t][e], oacc[0], 0, 0, 0);
            oacc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[32], sacc[t][e], oacc[1], 0, 0, 0);
        }
    if (qi < S) {
        const float inv = 1.0f / l;
        float* op = o + ((int64_t)b * S + qi) * ldo + (int64_t)h * AD + 4 * half;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 w = {oacc[dt][4 * g] * inv, oacc[dt][4 * g + 1] * inv, oacc[dt][4 * g + 2] * inv, oacc[dt][4 * g + 3] * inv};
                *reinterpret_cast<f32x4*>(op + dt * 32 + 8 * g) = w;
            }
    }
}

}  // namespace

#define STREAM ((hipStream_t)stream)

extern "C" int mi355_split3_bf16(int64_t rows, int K, const float* x, int64_t ldx, void* y, int weight_order, void* stream) {
    MI355_REQUIRE(rows > 0 && K > 0 && (K & 7) == 0 && x && y && ldx >= K && (ldx & 3) == 0 && (weight_order == 0 || weight_order == 1),
                  "mi355_split3_bf16: rows, K must be positive, K a multiple of 8, ldx >= K and a multiple of 4, weight_order 0 or 1");
    const int grid = grid_for(rows * (K >> 3), 256);
    if (weight_order) hipLaunchKernelGGL(split3_kernel<true>, dim3(grid), dim3(256), 0, STREAM, rows, K, x, ldx, (bf16_t*)y);
    else hipLaunchKernelGGL(split3_kernel<false>, dim3(grid), dim3(256), 0, STREAM, rows, K, x, ldx, (bf16_t*)y);
    MI355_LAUNCH_CHECK("mi355_split3_bf16");
    return 0;
}

extern "C" int mi355_attn_f32_fwd(int B, int S, int H, int D, const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                                  float* o, int64_t ldo, float scale, void* stream) {
    MI355_REQUIRE(B > 0 && S > 0 && H > 0 && q && k && v && o, "mi355_attn_f32_fwd: bad arguments");
    MI355_REQUIRE(D == 64, "mi355_attn_f32_fwd: head_dim must be 64 (got %d)", D);
    MI355_REQUIRE(S <= 288, "mi355_attn_f32_fwd: at most 288 keys (got %d): a head's K and V sit whole in LDS", S);
    MI355_REQUIRE((ldq & 3) == 0 && (ldk & 3) == 0 && (ldv & 3) == 0 && (ldo & 3) == 0 && ldq >= (int64_t)H * D && ldk >= (int64_t)H * D && ldv >= (int64_t)H * D &&
                      ldo >= (int64_t)H * D, "mi355_attn_f32_fwd: row pitches must be multiples of 4 floats and cover H * D");
    MI355_REQUIRE((int64_t)B * H * ((S + 127) / 128) < (1ll << 31), "mi355_attn_f32_fwd: grid too large");
    const int nt = (S + 31) / 32;
    const dim3 grid((unsigned)((int64_t)B * H * ((S + 127) / 128)));
#define LAUNCH_NT(N) \
    case N: hipLaunchKernelGGL(attn_f32_kernel<N>, grid, dim3(256), 0, STREAM, B, S, H, q, ldq, k, ldk, v, ldv, o, ldo, scale); break;
    switch (nt) {
        LAUNCH_NT(1)
        LAUNCH_NT(2)
        LAUNCH_NT(3)
        LAUNCH_NT(4)
        LAUNCH_NT(5)
        LAUNCH_NT(6)
        LAUNCH_NT(7)
        LAUNCH_NT(8)
        LAUNCH_NT(9)
    }
#undef LAUNCH_NT
    MI355_LAUNCH_CHECK("mi355_attn_f32_fwd");
    return 0;
}
