// Shared device/host helpers for the gfx950 kernels.  gfx950 only: wave = 64 lanes, no portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/mi355_vlm.h"

typedef unsigned short bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ float bf2f(bf16_t b) { return __uint_as_float(((unsigned)b) << 16); }

// round-to-nearest-even, NaN preserving (hipcc lowers the cast to v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 h = (__bf16)f;
    return __builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
    typedef __attribute__((ext_vector_type(2))) float f2;
    f2 v = {lo, hi};
    bf2 r = __builtin_convertvector(v, bf2);
    return __builtin_bit_cast(unsigned, r);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- SwiGLU: one definition for the elementwise kernels and the GEMM epilogues, so fused == separate bit for bit.  sigmoid through v_exp + v_rcp
// (1 ulp each; an IEEE division costs ten instructions per element, and the epilogues are bound by what their lanes issue); operands and results
// are rounded to bf16 where the reference's bf16 tensors are (silu(g) is a bf16 tensor before it multiplies u, qwen3_transformer_block.py:48-53).
__device__ __forceinline__ float sigmoid_fast(float g) { return __builtin_amdgcn_rcpf(1.0f + __expf(-g)); }
__device__ __forceinline__ float swiglu_act(float u, float g) { return u * bf2f(f2bf(g * sigmoid_fast(g))); }
__device__ __forceinline__ void swiglu_grads(float d, float u, float g, float& du, float& dg) {
    const float sg = sigmoid_fast(g);
    du = d * g * sg;
    dg = d * u * sg * (1.0f + g * (1.0f - sg));
}

// ---- GELU: KIND 0 exact erf (nn.GELU()), KIND 1 tanh approximation; one definition for the elementwise kernels and the GEMM epilogues
template <int KIND>
__device__ __forceinline__ float gelu_val(float x) {
    if (KIND == 0) return x * 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
    return 0.5f * x * (1.0f + tanhf(u));
}
template <int KIND>
__device__ __forceinline__ float gelu_grad(float x) {
    if (KIND == 0) {
        const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
        return cdf + x * 0.3989422804014327f * __expf(-0.5f * x * x);
    }
    const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
    const float th = tanhf(u);
    return 0.5f * (1.0f + th) + 0.5f * x * (1.0f - th * th) * 0.7978845608028654f * (1.0f + 3.0f * 0.044715f * x * x);
}

// ---- counter-based random bits for dropout: Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as
// 1, 2, 3", SC'11).  One call = 4 x 32 bits for counter (c0..c3) under key (k0, k1); no state, so the backward regenerates the
// forward's mask from (seed, offset) alone.  The oracle restates the same function in numpy (oracle/dropout.py).
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0;
        c1 = lo1;
        c2 = hi0 ^ c3 ^ k1;
        c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0, out[1] = c1, out[2] = c2, out[3] = c3;
}
// Dropout keep rule shared by every site: element e of a group of four keeps its value iff bits[e] >= thresh, thresh = round(p * 2^32).

static inline unsigned mi355_dropout_threshold(float p) {
    double t = (double)p * 4294967296.0 + 0.5;
    if (t < 0.0) t = 0.0;
    if (t > 4294967295.0) t = 4294967295.0;
    return (unsigned)t;
}

// ---- host side error plumbing -------------------------------------------------------------
void mi355_set_error(const char* fmt, ...);
#define MI355_REQUIRE(cond, ...)          \
    do {                                  \
        if (!(cond)) {                    \
            mi355_set_error(__VA_ARGS__); \
            return 1;                     \
        }                                 \
    } while (0)
#define MI355_LAUNCH_CHECK(name)                                                      \
    do {                                                                              \
        hipError_t e_ = hipGetLastError();                                            \
        if (e_ != hipSuccess) {                                                       \
            mi355_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));    \
            return 2;                                                                 \
        }                                                                             \
    } while (0)
