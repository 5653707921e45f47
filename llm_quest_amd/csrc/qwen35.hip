// Qwen3.5 hybrid text stack (BASELINE config 5): the row / elementwise / recurrence kernels of
// FusedGatedDeltaNet and MRoPEGatedAttention.  All HBM- or latency-bound; the dense contractions of these layers run on
// gemm.hip and attention_generic.hip.
//
//   zc_weight            (1 + scale) of ZeroCenteredRMSNorm, rounded to bf16 exactly where the reference rounds it
//   mrope_table          per-token interleaved MRoPE cos / sin rows (integer-exact gather)
//   headnorm_rope        per-head ZC-RMSNorm + partial rotation on strided heads of a fused projection
//   sigmoid_gate         ctx * sigmoid(gate) of GatedAttention
//   gdn_gates            beta = sigmoid(.), alpha = exp(-exp(log_A) * softplus(. + dt_bias))
//   causal_conv_silu     depthwise causal Conv1d(k) + SiLU on the token-major fused QKV projection
//   l2norm               q / k heads scaled to unit length (clamped norm)
//   gated_delta_rule     S_t = a_t S_{t-1} + b_t (v_t - a_t S_{t-1} k_t) k_t^T, o_t = S_t q_t/sqrt(dk): every state ROW is an
//                        independent recurrence, so rows are spread over lanes and the time loop runs inside the kernel
//   gated_rmsnorm        RMSNorm(fp32) * silu(gate) -> bf16
//   rowmask              x * padding_mask
#include "common.h"

namespace {

__device__ __forceinline__ float rbf(float x) { return bf2f(f2bf(x)); }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float silu_f(float x) { return x / (1.f + expf(-x)); }
__device__ __forceinline__ float dsilu_f(float x) {
    const float s = sigmoid_f(x);
    return s * (1.f + x * (1.f - s));
}
__device__ __forceinline__ void unpack8(const u32x4 v, float (&f)[8]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        f[2 * e] = __uint_as_float(v[e] << 16);
        f[2 * e + 1] = __uint_as_float(v[e] & 0xffff0000u);
    }
}
__device__ __forceinline__ u32x4 pack8(const float (&f)[8]) {
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack_bf2(f[2 * e], f[2 * e + 1]);
    return o;
}
// Cross-lane sums on the VALU (DPP / permlane swaps), not through the LDS crossbar: ds_bpermute (what __shfl_xor compiles to)
// costs an LDS round trip per step of every reduction, and the recurrence kernels below are one dependent chain per time step.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int W>
__device__ __forceinline__ float lanes_sum(float v) {  // all-reduce over aligned groups of W (4, 8 or 16) lanes
    static_assert(W == 4 || W == 8 || W == 16, "lanes_sum: 4, 8 or 16");
    if constexpr (W == 16) {
        v += dpp_f<0x128>(v);  // row_ror:8
        v += dpp_f<0x124>(v);  // row_ror:4
        v += dpp_f<0x122>(v);  // row_ror:2
        v += dpp_f<0x121>(v);  // row_ror:1
    } else {
        v += dpp_f<0xB1>(v);  // quad_perm:[1,0,3,2]
        v += dpp_f<0x4E>(v);  // quad_perm:[2,3,0,1]
        if constexpr (W == 8) v += dpp_f<0x141>(v);  // row_half_mirror: the other quad of the 8-lane group
    }
    return v;
}
// v[lane] + v[lane ^ 16] + v[lane ^ 32] + v[lane ^ 48]: the same element of the wave's four 16-lane rows
__device__ __forceinline__ float rows_sum(float v) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// Reduce-scatter over the wave's four 16-lane rows: ONE swap + ONE add folds TWO values across a row pair (v_permlane16_swap
// exchanges the odd rows of its first operand with the even rows of its second): afterwards even rows hold sum(a), odd rows sum(b)
// of their pair.  rows_fold32 does the same across the lower / upper half of the wave.
__device__ __forceinline__ float rows_fold16(float a, float b) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float rows_fold32(float a, float b) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// dot product of two register vectors on two partial sums (packed f32 FMAs instead of one serial chain)
template <int N>
__device__ __forceinline__ float dot_pk(const float* __restrict__ x, const float (&y)[N]) {
    if constexpr (N % 2 == 0) {
        f32x2 acc = {0.f, 0.f};
#pragma unroll
        for (int j = 0; j < N; j += 2) acc = __builtin_elementwise_fma(f32x2{x[j], x[j + 1]}, f32x2{y[j], y[j + 1]}, acc);
        return acc[0] + acc[1];
    } else {
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < N; ++j) acc = fmaf(x[j], y[j], acc);
        return acc;
    }
}

// ------------------------------------------------------------------------------------------- small helpers
__global__ void zc_weight_kernel(int64_t n, const bf16_t* __restrict__ s, bf16_t* __restrict__ w) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) w[i] = f2bf(1.0f + bf2f(s[i]));
}

// cos_t[t, j] = cos_t[t, j + half] = cos[pid[axis(j)][t], j]   (rope.py:246-294, 297-343)
__global__ void mrope_table_kernel(int64_t tokens, int R, int64_t ctx, const float* __restrict__ cosv, const float* __restrict__ sinv,
                                   const int64_t* __restrict__ pid, int sec_h, int sec_w, float* __restrict__ cos_t,
                                   float* __restrict__ sin_t) {
    const int half = R >> 1;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= tokens * half) return;
    const int64_t t = idx / half;
    const int j = (int)(idx % half);
    const int axis = (j % 3 == 1 && j < 3 * sec_h) ? 1 : (j % 3 == 2 && j < 3 * sec_w) ? 2 : 0;
    int64_t p = pid[axis * tokens + t];
    p = p < 0 ? 0 : (p >= ctx ? ctx - 1 : p);  // never read outside the table
    const float c = cosv[p * R + j], s = sinv[p * R + j];
    cos_t[t * R + j] = c;
    cos_t[t * R + j + half] = c;
    sin_t[t * R + j] = s;
    sin_t[t * R + j + half] = s;
}

__global__ void rowmask_kernel(int64_t rows, int width, const bf16_t* __restrict__ x, const uint8_t* __restrict__ mask,
                               bf16_t* __restrict__ y) {
    const int nvec = width >> 3;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * nvec) return;
    const int64_t r = idx / nvec;
    u32x4 v = *reinterpret_cast<const u32x4*>(x + idx * 8);
    if (!mask[r]) v = u32x4{0, 0, 0, 0};
    *reinterpret_cast<u32x4*>(y + idx * 8) = v;
}

// ------------------------------------------------------------------------------------------- head norm + partial RoPE
// One wave per (token, head); lane owns elements lane + 64 j.  The rotated prefix R <= 64 lives in element j = 0, the
// rotation partner is lane ^ (R/2).  Rounding points as upstream: norm result -> bf16, cos/sin -> bf16, each product and the
// sum -> bf16 (qwen3_next_attention.py:38-46, rope.py:226-243, 345-358).
template <int EPL>
__global__ __launch_bounds__(256) void headnorm_rope_fwd_kernel(int64_t tokens, int H, int D, int R, const bf16_t* __restrict__ src,
                                                                int64_t ld, int64_t hstride, const bf16_t* __restrict__ w,
                                                                const float* __restrict__ cos_t, const float* __restrict__ sin_t,
                                                                const int32_t* __restrict__ pos, bf16_t* __restrict__ out,
                                                                float* __restrict__ rstd, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t total = tokens * H;
    for (int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); item < total; item += (int64_t)gridDim.x * 4) {
        const int64_t t = item / H;
        const int h = (int)(item % H);
        const bf16_t* x = src + t * ld + h * hstride;
        float v[EPL], ss = 0.f;
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            const int i = lane + 64 * j;
            v[j] = i < D ? bf2f(x[i]) : 0.f;
            ss += v[j] * v[j];
        }
        ss = wave_sum(ss);
        const float r = rsqrtf(ss / (float)D + eps);
        if (lane == 0) rstd[item] = r;
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            const int i = lane + 64 * j;
            v[j] = i < D ? rbf(v[j] * r * bf2f(w[i])) : 0.f;
        }
        if (R > 0) {
            const float partner = __shfl_xor(v[0], R >> 1, 64);
            if (lane < R) {
                const int64_t p = pos[t];
                const float c = rbf(cos_t[p * R + lane]), s = rbf(sin_t[p * R + lane]);
                const float rot = lane < (R >> 1) ? -partner : partner;
                v[0] = rbf(rbf(c * v[0]) + rbf(s * rot));
            }
        }
        bf16_t* o = out + item * D;
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            const int i = lane + 64 * j;
            if (i < D) o[i] = f2bf(v[j]);
        }
    }
}

template <int EPL>
__global__ __launch_bounds__(256) void headnorm_rope_bwd_kernel(int64_t tokens, int H, int D, int R, const bf16_t* __restrict__ src,
                                                                int64_t ld, int64_t hstride, const bf16_t* __restrict__ w,
                                                                const float* __restrict__ cos_t, const float* __restrict__ sin_t,
                                                                const int32_t* __restrict__ pos, const float* __restrict__ rstd,
                                                                const bf16_t* __restrict__ dout, bf16_t* __restrict__ dsrc, int64_t ldd,
                                                                int64_t dhstride, float* __restrict__ dw_partial) {
    __shared__ float dw_lds[4][64 * EPL];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t total = tokens * H;
    float dwacc[EPL];
#pragma unroll
    for (int j = 0; j < EPL; ++j) dwacc[j] = 0.f;
    for (int64_t item = (int64_t)blockIdx.x * 4 + wave; item < total; item += (int64_t)gridDim.x * 4) {
        const int64_t t = item / H;
        const int h = (int)(item % H);
        const bf16_t* x = src + t * ld + h * hstride;
        const bf16_t* gptr = dout + item * D;
        float g[EPL], xv[EPL];
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            const int i = lane + 64 * j;
            g[j] = i < D ? bf2f(gptr[i]) : 0.f;
            xv[j] = i < D ? bf2f(x[i]) : 0.f;
        }
        if (R > 0) {  // transpose of the rotation: dn_j = c_j g_j + (j < R/2 ? +1 : -1) s_j g_partner
            const float gp = __shfl_xor(g[0], R >> 1, 64);
            if (lane < R) {
                const int64_t p = pos[t];
                const float c = rbf(cos_t[p * R + lane]), s = rbf(sin_t[p * R + lane]);
                g[0] = c * g[0] + (lane < (R >> 1) ? s * gp : -(s * gp));
            }
        }
        const float r = rstd[item];
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            const int i = lane + 64 * j;
            const float xh = xv[j] * r;
            dwacc[j] += g[j] * xh;
            g[j] *= i < D ? bf2f(w[i]) : 0.f;
            dot += g[j] * xh;
            xv[j] = xh;
        }
        dot = wave_sum(dot) / (float)D;
        bf16_t* d = dsrc + t * ldd + h * dhstride;
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            const int i = lane + 64 * j;
            if (i < D) d[i] = f2bf(r * (g[j] - xv[j] * dot));
        }
    }
#pragma unroll
    for (int j = 0; j < EPL; ++j) dw_lds[wave][lane + 64 * j] = dwacc[j];
    __syncthreads();
    for (int i = threadIdx.x; i < D; i += 256)
        dw_partial[(int64_t)blockIdx.x * D + i] = (dw_lds[0][i] + dw_lds[1][i]) + (dw_lds[2][i] + dw_lds[3][i]);
}

// ------------------------------------------------------------------------------------------- sigmoid output gate
// out = bf16(ctx * bf16(sigmoid(gate)))   (qwen3_next_attention.py:221, 257)
__global__ void sigmoid_gate_fwd_kernel(int64_t tokens, int H, int D, const bf16_t* __restrict__ ctx, const bf16_t* __restrict__ gate,
                                        int64_t ldg, int64_t ghs, bf16_t* __restrict__ out) {
    const int vph = D >> 3;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= tokens * H * vph) return;
    const int64_t t = idx / (H * vph);
    const int rem = (int)(idx % (H * vph));
    const int h = rem / vph, c = rem % vph;
    float cv[8], gv[8], o[8];
    unpack8(*reinterpret_cast<const u32x4*>(ctx + idx * 8), cv);
    unpack8(*reinterpret_cast<const u32x4*>(gate + t * ldg + h * ghs + c * 8), gv);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = cv[e] * rbf(sigmoid_f(gv[e]));
    *reinterpret_cast<u32x4*>(out + idx * 8) = pack8(o);
}
__global__ void sigmoid_gate_bwd_kernel(int64_t tokens, int H, int D, const bf16_t* __restrict__ ctx, const bf16_t* __restrict__ gate,
                                        int64_t ldg, int64_t ghs, const bf16_t* __restrict__ dout, bf16_t* __restrict__ dctx,
                                        bf16_t* __restrict__ dgate, int64_t lddg, int64_t dghs) {
    const int vph = D >> 3;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= tokens * H * vph) return;
    const int64_t t = idx / (H * vph);
    const int rem = (int)(idx % (H * vph));
    const int h = rem / vph, c = rem % vph;
    float cv[8], gv[8], dv[8], dc[8], dg[8];
    unpack8(*reinterpret_cast<const u32x4*>(ctx + idx * 8), cv);
    unpack8(*reinterpret_cast<const u32x4*>(gate + t * ldg + h * ghs + c * 8), gv);
    unpack8(*reinterpret_cast<const u32x4*>(dout + idx * 8), dv);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float s = sigmoid_f(gv[e]);
        dc[e] = dv[e] * rbf(s);
        dg[e] = dv[e] * cv[e] * s * (1.f - s);
    }
    *reinterpret_cast<u32x4*>(dctx + idx * 8) = pack8(dc);
    *reinterpret_cast<u32x4*>(dgate + t * lddg + h * dghs + c * 8) = pack8(dg);
}

// ------------------------------------------------------------------------------------------- GDN gates
// beta = float(bf16(sigmoid(b)));  alpha = exp(-exp(log_A) * float(bf16(softplus(bf16(a + dt_bias)))))   fp32
// (qwen3_5_text_model.py:115-117, qwen3_next_attention.py:71-100)
__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }

__global__ void gdn_gates_fwd_kernel(int64_t tokens, int Hv, const bf16_t* __restrict__ b_lin, const bf16_t* __restrict__ a_lin,
                                     int64_t ld, const float* __restrict__ log_A, const bf16_t* __restrict__ dt_bias,
                                     float* __restrict__ beta, float* __restrict__ alpha) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= tokens * Hv) return;
    const int64_t t = idx / Hv;
    const int h = (int)(idx % Hv);
    beta[idx] = rbf(sigmoid_f(bf2f(b_lin[t * ld + h])));
    const float z = rbf(bf2f(a_lin[t * ld + h]) + bf2f(dt_bias[h]));
    const float sp = rbf(softplus_f(z));
    alpha[idx] = expf(-expf(log_A[h]) * sp);
}
// d b_lin = dbeta * s(1-s);  d a_lin = dalpha * alpha * (-A) * sigmoid(z);  dlog_A += dalpha * alpha * (-A sp);  ddt_bias += d a_lin
// partial[block][0:Hv] = dlog_A, [Hv:2Hv] = ddt_bias (a block walks a fixed token range; 256 / Hv tokens per pass)
__global__ __launch_bounds__(256) void gdn_gates_bwd_kernel(int64_t tokens, int Hv, const bf16_t* __restrict__ b_lin,
                                                            const bf16_t* __restrict__ a_lin, int64_t ld, const float* __restrict__ log_A,
                                                            const bf16_t* __restrict__ dt_bias, const float* __restrict__ dbeta,
                                                            const float* __restrict__ dalpha, bf16_t* __restrict__ db_lin,
                                                            bf16_t* __restrict__ da_lin, int64_t ldd, float* __restrict__ partial) {
    __shared__ float red[2][256];
    const int tpp = 256 / Hv;  // tokens per pass (Hv divides 256)
    const int h = threadIdx.x % Hv, tl = threadIdx.x / Hv;
    float accA = 0.f, accB = 0.f;
    if (tl < tpp) {
        const float A = expf(log_A[h]);
        const float bias = bf2f(dt_bias[h]);
        for (int64_t t = (int64_t)blockIdx.x * tpp + tl; t < tokens; t += (int64_t)gridDim.x * tpp) {
            const float s = sigmoid_f(bf2f(b_lin[t * ld + h]));
            db_lin[t * ldd + h] = f2bf(dbeta[t * Hv + h] * s * (1.f - s));
            const float z = rbf(bf2f(a_lin[t * ld + h]) + bias);
            const float sp = rbf(softplus_f(z));
            const float al = expf(-A * sp);
            const float g = dalpha[t * Hv + h] * al;  // d(-A sp)
            const float da = g * (-A) * (z > 20.f ? 1.f : sigmoid_f(z));
            da_lin[t * ldd + h] = f2bf(da);
            accA += g * (-A * sp);
            accB += da;
        }
    }
    red[0][threadIdx.x] = accA;
    red[1][threadIdx.x] = accB;
    __syncthreads();
    if (threadIdx.x < Hv) {
        float sa = 0.f, sb = 0.f;
        for (int i = 0; i < tpp; ++i) {
            sa += red[0][i * Hv + threadIdx.x];
            sb += red[1][i * Hv + threadIdx.x];
        }
        partial[(int64_t)blockIdx.x * 2 * Hv + threadIdx.x] = sa;
        partial[(int64_t)blockIdx.x * 2 * Hv + Hv + threadIdx.x] = sb;
    }
}

// ------------------------------------------------------------------------------------------- causal depthwise conv + SiLU
// y[t, c] = bf16(silu(bf16(sum_j w[c, j] x[t - (k-1) + j, c])))  within each sequence (qwen3_5_text_model.py:81-90, 138-140).
// A thread owns 8 adjacent channels (16-byte loads) of one token.
template <int KS>
__global__ __launch_bounds__(256) void conv_silu_fwd_kernel(int B, int S, int C, const bf16_t* __restrict__ x, int64_t ldx,
                                                            const bf16_t* __restrict__ w, bf16_t* __restrict__ y) {
    const int cvec = C >> 3;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)B * S * cvec) return;
    const int64_t tok = idx / cvec;
    const int c0 = (int)(idx % cvec) * 8;
    const int s = (int)(tok % S);
    float wv[8][KS];
    {
        float tmp[8 * KS];
#pragma unroll
        for (int i = 0; i < KS; ++i) unpack8(*reinterpret_cast<const u32x4*>(w + (int64_t)c0 * KS + i * 8), *reinterpret_cast<float(*)[8]>(tmp + 8 * i));
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
            for (int j = 0; j < KS; ++j) wv[c][j] = tmp[c * KS + j];
    }
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < KS; ++j) {
        const int sj = s - (KS - 1) + j;
        if (sj >= 0) {
            float xv[8];
            unpack8(*reinterpret_cast<const u32x4*>(x + (tok - (KS - 1) + j) * ldx + c0), xv);
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c] = fmaf(wv[c][j], xv[c], acc[c]);
        }
    }
    float o[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) o[c] = silu_f(rbf(acc[c]));
    *reinterpret_cast<u32x4*>(y + tok * C + c0) = pack8(o);
}

// pass 1: a thread owns 8 channels and walks a chunk of TCH tokens of one sequence: dc = bf16(dy * silu'(c)) -> dcws, and the
// filter gradient of its channels summed over the chunk -> dw_partial[chunk][C*KS]
template <int KS>
__global__ __launch_bounds__(256) void conv_silu_bwd_dc_kernel(int B, int S, int C, int TCH, int chunks_per_seq, const bf16_t* __restrict__ x,
                                                               int64_t ldx, const bf16_t* __restrict__ w, const bf16_t* __restrict__ dy,
                                                               bf16_t* __restrict__ dcws, float* __restrict__ dw_partial) {
    const int cvec = C >> 3;
    const int cv = blockIdx.x * 256 + threadIdx.x;
    if (cv >= cvec) return;
    const int c0 = cv * 8;
    const int chunk = blockIdx.y;  // over B * chunks_per_seq
    const int b = chunk / chunks_per_seq, s0 = (chunk % chunks_per_seq) * TCH;
    const int s1 = s0 + TCH < S ? s0 + TCH : S;
    float wv[8][KS], dw[8][KS];
    {
        float tmp[8 * KS];
#pragma unroll
        for (int i = 0; i < KS; ++i) unpack8(*reinterpret_cast<const u32x4*>(w + (int64_t)c0 * KS + i * 8), *reinterpret_cast<float(*)[8]>(tmp + 8 * i));
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
            for (int j = 0; j < KS; ++j) {
                wv[c][j] = tmp[c * KS + j];
                dw[c][j] = 0.f;
            }
    }
    float win[KS][8];  // sliding window: win[j] = x[s - (KS-1) + j]
#pragma unroll
    for (int j = 0; j < KS - 1; ++j) {
        const int sj = s0 - (KS - 1) + j;
        if (sj >= 0)
            unpack8(*reinterpret_cast<const u32x4*>(x + ((int64_t)b * S + sj) * ldx + c0), win[j + 1]);
        else
#pragma unroll
            for (int c = 0; c < 8; ++c) win[j + 1][c] = 0.f;
    }
    for (int s = s0; s < s1; ++s) {
        const int64_t tok = (int64_t)b * S + s;
#pragma unroll
        for (int j = 0; j < KS - 1; ++j)
#pragma unroll
            for (int c = 0; c < 8; ++c) win[j][c] = win[j + 1][c];
        unpack8(*reinterpret_cast<const u32x4*>(x + tok * ldx + c0), win[KS - 1]);
        float g[8], dc[8];
        unpack8(*reinterpret_cast<const u32x4*>(dy + tok * C + c0), g);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < KS; ++j) acc = fmaf(wv[c][j], win[j][c], acc);
            dc[c] = rbf(g[c] * dsilu_f(rbf(acc)));
#pragma unroll
            for (int j = 0; j < KS; ++j) dw[c][j] = fmaf(dc[c], win[j][c], dw[c][j]);
        }
        *reinterpret_cast<u32x4*>(dcws + tok * C + c0) = pack8(dc);
    }
    float* dst = dw_partial + (int64_t)chunk * C * KS + (int64_t)c0 * KS;
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int j = 0; j < KS; ++j) dst[c * KS + j] = dw[c][j];
}
// pass 2: dx[t, c] = sum_j w[c, j] dc[t + (k-1) - j, c]
template <int KS>
__global__ __launch_bounds__(256) void conv_silu_bwd_dx_kernel(int B, int S, int C, const bf16_t* __restrict__ w, const bf16_t* __restrict__ dcws,
                                                               bf16_t* __restrict__ dx, int64_t lddx) {
    const int cvec = C >> 3;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)B * S * cvec) return;
    const int64_t tok = idx / cvec;
    const int c0 = (int)(idx % cvec) * 8;
    const int s = (int)(tok % S);
    float tmp[8 * KS];
#pragma unroll
    for (int i = 0; i < KS; ++i) unpack8(*reinterpret_cast<const u32x4*>(w + (int64_t)c0 * KS + i * 8), *reinterpret_cast<float(*)[8]>(tmp + 8 * i));
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < KS; ++j) {
        const int sj = s + (KS - 1) - j;
        if (sj < S) {
            float g[8];
            unpack8(*reinterpret_cast<const u32x4*>(dcws + (tok + (KS - 1) - j) * C + c0), g);
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c] = fmaf(tmp[c * KS + j], g[c], acc[c]);
        }
    }
    *reinterpret_cast<u32x4*>(dx + tok * lddx + c0) = pack8(acc);
}

// one decoded token (torch_causal_conv1d_update of the reference, qwen3_5_text_model.py:425-456 + SiLU): the state holds the last
// KS pre-conv inputs token-major [B, KS, C]; shift it by one, append x_new, y = silu(bf16(sum_j w[c, j] state[j, c])).  A thread owns
// 8 channels of one sequence and all KS rows of them, so the in-place update has no race.
template <int KS>
__global__ __launch_bounds__(256) void conv_silu_step_kernel(int B, int C, const bf16_t* __restrict__ x, int64_t ldx, bf16_t* __restrict__ state,
                                                             const bf16_t* __restrict__ w, bf16_t* __restrict__ y) {
    const int cvec = C >> 3;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)B * cvec) return;
    const int b = (int)(idx / cvec), c0 = (int)(idx % cvec) * 8;
    float tmp[8 * KS];
#pragma unroll
    for (int i = 0; i < KS; ++i) unpack8(*reinterpret_cast<const u32x4*>(w + (int64_t)c0 * KS + i * 8), *reinterpret_cast<float(*)[8]>(tmp + 8 * i));
    bf16_t* st = state + (int64_t)b * KS * C + c0;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < KS; ++j) {
        const u32x4 raw = j + 1 < KS ? *reinterpret_cast<const u32x4*>(st + (int64_t)(j + 1) * C) : *reinterpret_cast<const u32x4*>(x + (int64_t)b * ldx + c0);
        float xv[8];
        unpack8(raw, xv);
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = fmaf(tmp[c * KS + j], xv[c], acc[c]);
        *reinterpret_cast<u32x4*>(st + (int64_t)j * C) = raw;
    }
    float o[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) o[c] = silu_f(rbf(acc[c]));
    *reinterpret_cast<u32x4*>(y + (int64_t)b * C + c0) = pack8(o);
}

// ------------------------------------------------------------------------------------------- l2 norm of q / k heads
// y = bf16(x * bf16(1 / max(bf16(||x||), 1e-6)))   (qwen3_next_attention.py:51-60 on bf16 tensors)
template <int EPL>
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(int64_t tokens, int H, int D, const bf16_t* __restrict__ x, int64_t ldx,
                                                         bf16_t* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int64_t total = tokens * H;
    const float floor_ = rbf(1e-6f);
    for (int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); item < total; item += (int64_t)gridDim.x * 4) {
        const int64_t t = item / H;
        const int h = (int)(item % H);
        const bf16_t* xp = x + t * ldx + (int64_t)h * D;
        float v[EPL], ss = 0.f;
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            const int i = lane + 64 * j;
            v[j] = i < D ? bf2f(xp[i]) : 0.f;
            ss += v[j] * v[j];
        }
        ss = wave_sum(ss);
        const float n = fmaxf(rbf(sqrtf(ss)), floor_);
        const float inv = rbf(1.f / n);
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            const int i = lane + 64 * j;
            if (i < D) y[item * D + i] = f2bf(v[j] * inv);
        }
    }
}
template <int EPL>
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(int64_t tokens, int H, int D, const bf16_t* __restrict__ x, int64_t ldx,
                                                         const bf16_t* __restrict__ dy, bf16_t* __restrict__ dx, int64_t lddx) {
    const int lane = threadIdx.x & 63;
    const int64_t total = tokens * H;
    for (int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); item < total; item += (int64_t)gridDim.x * 4) {
        const int64_t t = item / H;
        const int h = (int)(item % H);
        const bf16_t* xp = x + t * ldx + (int64_t)h * D;
        float v[EPL], g[EPL], ss = 0.f, dot = 0.f;
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            const int i = lane + 64 * j;
            v[j] = i < D ? bf2f(xp[i]) : 0.f;
            g[j] = i < D ? bf2f(dy[item * D + i]) : 0.f;
            ss += v[j] * v[j];
            dot += v[j] * g[j];
        }
        ss = wave_sum(ss);
        dot = wave_sum(dot);
        const float n = sqrtf(ss);
        const bool clamped = n < 1e-6f;
        const float inv = 1.f / fmaxf(n, 1e-6f);
        const float k3 = clamped ? 0.f : dot * inv * inv * inv;
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            const int i = lane + 64 * j;
            if (i < D) dx[t * lddx + (int64_t)h * D + i] = f2bf(g[j] * inv - v[j] * k3);
        }
    }
}

// ------------------------------------------------------------------------------------------- gated delta rule
// Reference: qwen3_next_attention.py:103-159 (fp32 recurrence on bf16 operands).  Row i of the state obeys
//     G_i = a S_i ;  u_i = G_i . k ;  c_i = b (v_i - u_i) ;  S_i <- G_i + c_i k ;  o_i = S_i . (q / sqrt(dk))
// independently of every other row, so the kernels spread rows over lanes and keep the time loop inside.  A time step is one
// dependent chain; what hides its latencies is (1) operands of step t+1 requested as RAW bf16 words before step t's arithmetic
// and unpacked only when used, (2) cross-lane sums on DPP / permlane, (3) in the forward, two waves per SIMD.
template <int N>
struct RawVec {  // N bf16 values as they come from memory
    unsigned w[(N + 1) / 2];
};
template <int N>
__device__ __forceinline__ void raw_load(RawVec<N>& r, const bf16_t* p) {
    if constexpr (N % 8 == 0) {
#pragma unroll
        for (int i = 0; i < N / 8; ++i) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(p + 8 * i);
            r.w[4 * i] = v[0];
            r.w[4 * i + 1] = v[1];
            r.w[4 * i + 2] = v[2];
            r.w[4 * i + 3] = v[3];
        }
    } else if constexpr (N % 2 == 0) {
#pragma unroll
        for (int i = 0; i < N / 2; ++i) r.w[i] = *reinterpret_cast<const unsigned*>(p + 2 * i);
    } else {
        static_assert(N == 1, "odd vectors: one element only");
        r.w[0] = *p;
    }
}
template <int N>
__device__ __forceinline__ float raw_get(const RawVec<N>& r, int i) {
    return (i & 1) ? __uint_as_float(r.w[i >> 1] & 0xffff0000u) : __uint_as_float(r.w[i >> 1] << 16);
}

// Forward layout: LPR lanes per row (CPL = DK / LPR columns each), 64 / LPR rows per wave; the two dot products of a step close
// with log2(LPR) DPP adds.  Checkpoints of S every CH steps feed the backward pass.
template <int DK, int LPR>
__global__ __launch_bounds__(256) void gdr_fwd_kernel(int B, int S, int Hqk, int Hv, int Dv, const bf16_t* __restrict__ q,
                                                      const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, int64_t ldv,
                                                      const float* __restrict__ beta, const float* __restrict__ alpha,
                                                      bf16_t* __restrict__ o, float* __restrict__ ckpt, int CH, int nchunk,
                                                      const float* __restrict__ initial_state, float* __restrict__ final_state, float qscale) {
    constexpr int CPL = DK / LPR, RPW = 64 / LPR;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.z, h = blockIdx.y;
    const int row0 = (blockIdx.x * 4 + wave) * RPW;
    if (row0 >= Dv) return;  // whole wave out of range (Dv is a multiple of 16 >= RPW); no block-level synchronisation below
    const int row = row0 + lane / LPR, cg = lane % LPR;
    const int hq = h / (Hv / Hqk);
    const int64_t tok0 = (int64_t)b * S;
    const int64_t ldqk = (int64_t)Hqk * DK;
    const bf16_t* kp = k + tok0 * ldqk + hq * DK + cg * CPL;
    const bf16_t* qp = q + tok0 * ldqk + hq * DK + cg * CPL;
    const bf16_t* vp = v + tok0 * ldv + (int64_t)h * Dv + row;
    const float* ap = alpha + tok0 * Hv + h;
    const float* bp = beta + tok0 * Hv + h;
    bf16_t* op = o + tok0 * Hv * Dv + (int64_t)h * Dv + row;
    static_assert(CPL % 2 == 0, "forward: an even number of columns per lane (packed f32 math)");
    constexpr int CP2 = CPL / 2;
    f32x2 st[CP2];
#pragma unroll
    for (int j = 0; j < CP2; ++j) st[j] = f32x2{0.f, 0.f};
    if (initial_state) {  // decode / continued prefill: the recurrent state of Qwen3_5Cache (may alias final_state: each lane reads, then writes, its own slice)
        const float* c = initial_state + (((int64_t)b * Hv + h) * Dv + row) * DK + cg * CPL;
#pragma unroll
        for (int j = 0; j < CP2; ++j) st[j] = f32x2{c[2 * j], c[2 * j + 1]};
    }
    struct In {
        RawVec<CPL> k, q;
        float a, b;
        bf16_t v;
    };
    auto load_in = [&](In& in, int t) {
        raw_load<CPL>(in.k, kp + t * ldqk);
        raw_load<CPL>(in.q, qp + t * ldqk);
        in.a = ap[(int64_t)t * Hv];
        in.b = bp[(int64_t)t * Hv];
        in.v = vp[t * ldv];
    };
    // one time step on packed pairs (v_pk_mul / v_pk_fma): both dot products run on two partial sums, one per half of a bf16 pair
    auto step = [&](const In& in, int t) {
        if (ckpt && t % CH == 0) {
            float* c = ckpt + ((((int64_t)b * Hv + h) * nchunk + t / CH) * Dv + row) * DK + cg * CPL;
#pragma unroll
            for (int j = 0; j < CP2; ++j) {
                c[2 * j] = st[j][0];
                c[2 * j + 1] = st[j][1];
            }
        }
        const f32x2 a2 = {in.a, in.a};
        f32x2 u2 = {0.f, 0.f};
#pragma unroll
        for (int j = 0; j < CP2; ++j) {
            const f32x2 k2 = {__uint_as_float(in.k.w[j] << 16), __uint_as_float(in.k.w[j] & 0xffff0000u)};
            st[j] = st[j] * a2;
            u2 = __builtin_elementwise_fma(st[j], k2, u2);
        }
        const float u = lanes_sum<LPR>(u2[0] + u2[1]);
        const float c = in.b * (bf2f(in.v) - u);
        const f32x2 c2 = {c, c};
        f32x2 o2 = {0.f, 0.f};
#pragma unroll
        for (int j = 0; j < CP2; ++j) {
            const f32x2 k2 = {__uint_as_float(in.k.w[j] << 16), __uint_as_float(in.k.w[j] & 0xffff0000u)};
            const f32x2 q2 = {__uint_as_float(in.q.w[j] << 16), __uint_as_float(in.q.w[j] & 0xffff0000u)};
            st[j] = __builtin_elementwise_fma(c2, k2, st[j]);
            o2 = __builtin_elementwise_fma(st[j], q2, o2);
        }
        const float oo = lanes_sum<LPR>(o2[0] + o2[1]);
        if (cg == 0) op[(int64_t)t * Hv * Dv] = f2bf(oo * qscale);
    };
    // two operand sets in ping-pong (the loop is unrolled by two, so no set is ever copied): the operands of step t+1 are requested
    // as RAW bf16 words before step t's arithmetic and unpacked only when used
    In inA, inB;
    load_in(inA, 0);
    for (int t = 0; t < S; t += 2) {
        if (t + 1 < S) load_in(inB, t + 1);
        step(inA, t);
        if (t + 1 < S) {
            if (t + 2 < S) load_in(inA, t + 2);
            step(inB, t + 1);
        }
    }
    if (final_state) {
        float* c = final_state + (((int64_t)b * Hv + h) * Dv + row) * DK + cg * CPL;
#pragma unroll
        for (int j = 0; j < CP2; ++j) {
            c[2 * j] = st[j][0];
            c[2 * j + 1] = st[j][1];
        }
    }
}

// Backward layout: a lane owns 4 rows x CPL columns, 16 lanes span Dk = 16*CPL, a wave owns 16 rows (4 row quads).  Row dot
// products close inside 16-lane groups, the sums over rows (dq, dk, dbeta, dalpha) inside the lane and across the 4 quads;
// each wave writes its partial sums for its 16 rows, a second kernel adds the row groups (and the value heads that share a
// q/k head).
constexpr int GDR_CH = 8;   // spacing of the HBM checkpoints the forward leaves (measured, B = 8, S = 708, fwd + bwd per layer: 4 -> 2.29 ms, 8 -> 2.22 ms, 16 -> 2.37 ms)
constexpr int GDR_SUB = 4;  // steps whose S_{t-1} are parked in LDS at a time
constexpr int GDR_NSUB = GDR_CH / GDR_SUB;
template <int CPL>
struct GdrOps {
    RawVec<CPL> k, q;
    float a, b;
    bf16_t v[4], g[4];
};
template <int CPL>
__device__ __forceinline__ void gdr_load_ops(GdrOps<CPL>& s, const bf16_t* kp, const bf16_t* qp, const bf16_t* vp, const bf16_t* gp,
                                             const float* ap, const float* bp) {
    raw_load<CPL>(s.k, kp);
    raw_load<CPL>(s.q, qp);
    s.a = *ap;
    s.b = *bp;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        s.v[r] = vp[r];
        s.g[r] = gp[r];
    }
}
// one forward step on a lane's 4 x CPL state slice (rows x columns), row dots over the 16 column lanes
template <int CPL>
__device__ __forceinline__ void gdr_replay_step(float (&st)[4 * CPL], const GdrOps<CPL>& op) {
    float kf[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) kf[j] = raw_get<CPL>(op.k, j);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int j = 0; j < CPL; ++j) st[r * CPL + j] *= op.a;
        const float u = lanes_sum<16>(dot_pk<CPL>(&st[r * CPL], kf));
        const float c = op.b * (bf2f(op.v[r]) - u);
#pragma unroll
        for (int j = 0; j < CPL; ++j) st[r * CPL + j] = fmaf(c, kf[j], st[r * CPL + j]);
    }
}

// Two-level recomputation.  The reverse step t needs S_{t-1}.  Version 1 parked every S_{t-1} in HBM (2 x 5.9 GB per layer at
// B = 8: purely bandwidth-bound, 3.1 ms); version 2 took 4-step checkpoints from the forward (1.45 GB per layer each way) and
// parked 4 states in LDS; now the forward leaves a checkpoint every 16 steps (0.37 GB per layer) and, per 16-step chunk (last
// first), this kernel (1) replays steps 0..11 keeping S_4, S_8, S_12 in registers (the accumulator file takes them), then for
// the 4-step groups 3, 2, 1, 0: (2) phase A replays the group from its sub-checkpoint parking S_{t-1} in LDS -- each lane's own
// 16-byte slots [step][vector][lane]: conflict-free, no barrier, nobody else reads them -- and (3) phase B walks the group
// backwards.  The operands of the NEXT group in this order (and the next chunk's checkpoint) are requested before the current
// group's arithmetic.
template <int CPL>
__global__ __launch_bounds__(256) void gdr_bwd_kernel(int B, int S, int Hqk, int Hv, int Dv, const bf16_t* __restrict__ q,
                                                      const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, int64_t ldv,
                                                      const float* __restrict__ beta, const float* __restrict__ alpha,
                                                      const float* __restrict__ ckpt, int nchunk, const bf16_t* __restrict__ d_o,
                                                      bf16_t* __restrict__ dv, int64_t lddv, float* __restrict__ pdq,
                                                      float* __restrict__ pdk, float* __restrict__ pdb, float* __restrict__ pda, float qscale,
                                                      const float* d_final_state, float* d_initial_state) {
    // d_final_state (optional): gradient arriving at the returned last state -- the starting value of the state gradient; d_initial_state (optional):
    // where the state gradient is left after step 0, i.e. d(loss) / d(carried-in state).  Both fp32 [B, Hv, Dv, DK]; they may be the same buffer.
    constexpr int DK = 16 * CPL, NS = 4 * CPL;
    __shared__ f32x4 park[4][GDR_SUB][CPL][64];  // [wave][step][vector of the lane's 4*CPL state floats][lane]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.z, h = blockIdx.y;
    const int rg = blockIdx.x * 4 + wave, RG = Dv / 16;
    if (rg >= RG) return;  // whole wave out of range; no block-level synchronisation below
    const int rq = lane >> 4, cl = lane & 15;
    const int row = rg * 16 + rq * 4;  // first of this lane's 4 rows
    const int col = cl * CPL;
    const int hq = h / (Hv / Hqk);
    const int64_t tok0 = (int64_t)b * S;
    const int64_t ldqk = (int64_t)Hqk * DK;
    const bf16_t* kp = k + tok0 * ldqk + hq * DK + col;
    const bf16_t* qp = q + tok0 * ldqk + hq * DK + col;
    const bf16_t* vp = v + tok0 * ldv + (int64_t)h * Dv + row;
    const bf16_t* gp = d_o + tok0 * Hv * Dv + (int64_t)h * Dv + row;
    const float* ap = alpha + tok0 * Hv + h;
    const float* bp = beta + tok0 * Hv + h;
    const float* ckp = ckpt + ((((int64_t)b * Hv + h) * nchunk) * Dv + row) * DK + col;  // + chunk*Dv*DK + r*DK + j
    const int64_t slot = ((int64_t)b * Hv + h) * RG + rg;
    float* wq = pdq + slot * (int64_t)S * DK + col;
    float* wk = pdk + slot * (int64_t)S * DK + col;
    float dS[NS];
#pragma unroll
    for (int e = 0; e < NS; ++e) dS[e] = 0.f;
    const int64_t st_off = (((int64_t)b * Hv + h) * Dv + row) * DK + col;  // this lane's 4 x CPL slice of a [B, Hv, Dv, DK] state
    if (d_final_state) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < CPL; ++j) dS[r * CPL + j] = d_final_state[st_off + r * DK + j];
    }

    GdrOps<CPL> cur[GDR_SUB], nxt[GDR_SUB];
    float sub[GDR_NSUB][NS], ckN[NS];
    auto load_group = [&](GdrOps<CPL>(&ops)[GDR_SUB], int tb) {
#pragma unroll
        for (int i = 0; i < GDR_SUB; ++i) {
            const int t = tb + i < S ? tb + i : S - 1;  // a ragged last group re-reads the last step; the copies are not used
            gdr_load_ops<CPL>(ops[i], kp + t * ldqk, qp + t * ldqk, vp + t * ldv, gp + (int64_t)t * Hv * Dv, ap + (int64_t)t * Hv, bp + (int64_t)t * Hv);
        }
    };
    auto load_ckpt = [&](float(&c)[NS], int chunk) {
        const float* cp = ckp + (int64_t)chunk * Dv * DK;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < CPL; ++j) c[r * CPL + j] = cp[r * DK + j];
    };
    load_group(cur, (nchunk - 1) * GDR_CH);
    load_ckpt(sub[0], nchunk - 1);

    for (int chunk = nchunk - 1; chunk >= 0; --chunk) {
        const int c0 = chunk * GDR_CH;
        const int n = S - c0 < GDR_CH ? S - c0 : GDR_CH;
        const int nsub = (n + GDR_SUB - 1) / GDR_SUB;
        if (chunk > 0) load_ckpt(ckN, chunk - 1);
        // ---- (1) sub-checkpoints S_4, S_8, S_12: groups 0 .. nsub-2 are complete groups
#pragma unroll
        for (int s_ = 0; s_ + 1 < GDR_NSUB; ++s_) {
            if (s_ + 1 < nsub) {
                load_group(nxt, c0 + (s_ + 1) * GDR_SUB);
#pragma unroll
                for (int e = 0; e < NS; ++e) sub[s_ + 1][e] = sub[s_][e];
#pragma unroll
                for (int i = 0; i < GDR_SUB; ++i) gdr_replay_step<CPL>(sub[s_ + 1], cur[i]);
#pragma unroll
                for (int i = 0; i < GDR_SUB; ++i) cur[i] = nxt[i];
            }
        }
        // ---- groups nsub-1 .. 0: (2) park, (3) reverse
#pragma unroll
        for (int s_ = GDR_NSUB - 1; s_ >= 0; --s_) {
            if (s_ < nsub) {
                const int t0 = c0 + s_ * GDR_SUB;
                const int ng = S - t0 < GDR_SUB ? S - t0 : GDR_SUB;
                if (s_ > 0) load_group(nxt, t0 - GDR_SUB);
                else if (chunk > 0) load_group(nxt, c0 - GDR_CH);
                float work[NS];
#pragma unroll
                for (int e = 0; e < NS; ++e) work[e] = sub[s_][e];
#pragma unroll
                for (int i = 0; i < GDR_SUB; ++i) {
                    if (i < ng) {
#pragma unroll
                        for (int vi = 0; vi < CPL; ++vi) park[wave][i][vi][lane] = f32x4{work[4 * vi], work[4 * vi + 1], work[4 * vi + 2], work[4 * vi + 3]};
                        if (i + 1 < ng) gdr_replay_step<CPL>(work, cur[i]);
                    }
                }
#pragma unroll
                for (int i = GDR_SUB - 1; i >= 0; --i) {
                    if (i < ng) {
                        const int t = t0 + i;
                        float Sp[NS];
#pragma unroll
                        for (int vi = 0; vi < CPL; ++vi) {
                            const f32x4 x = park[wave][i][vi][lane];
                            Sp[4 * vi] = x[0];
                            Sp[4 * vi + 1] = x[1];
                            Sp[4 * vi + 2] = x[2];
                            Sp[4 * vi + 3] = x[3];
                        }
                        const GdrOps<CPL>& op = cur[i];
                        float kf[CPL], qf[CPL], pq[CPL], pk[CPL];
#pragma unroll
                        for (int j = 0; j < CPL; ++j) {
                            kf[j] = raw_get<CPL>(op.k, j);
                            qf[j] = raw_get<CPL>(op.q, j);
                            pq[j] = pk[j] = 0.f;
                        }
                        float dbp = 0.f, dap = 0.f;
                        float dvr[4];
                        if constexpr (CPL % 2 == 0) {  // packed pairs over the columns (v_pk_mul / v_pk_fma), explicit: the SLP vectoriser packs only part of it
                            constexpr int P = CPL / 2;
                            f32x2 k2[P], q2[P], pq2[P], pk2[P];
#pragma unroll
                            for (int j = 0; j < P; ++j) {
                                k2[j] = f32x2{kf[2 * j], kf[2 * j + 1]};
                                q2[j] = f32x2{qf[2 * j], qf[2 * j + 1]};
                                pq2[j] = pk2[j] = f32x2{0.f, 0.f};
                            }
                            const f32x2 a2 = {op.a, op.a};
                            f32x2 dap2 = {0.f, 0.f};
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                f32x2 G2[P], Sp2[P], d2[P];
                                const float gr = bf2f(op.g[r]) * qscale;  // d(o_i)/d(S_ij) = q_j / sqrt(dk)
                                const f32x2 gr2 = {gr, gr};
                                f32x2 acc = {0.f, 0.f};
#pragma unroll
                                for (int j = 0; j < P; ++j) {
                                    Sp2[j] = f32x2{Sp[r * CPL + 2 * j], Sp[r * CPL + 2 * j + 1]};
                                    d2[j] = f32x2{dS[r * CPL + 2 * j], dS[r * CPL + 2 * j + 1]};
                                    G2[j] = Sp2[j] * a2;
                                    acc = __builtin_elementwise_fma(G2[j], k2[j], acc);
                                }
                                const float u = lanes_sum<16>(acc[0] + acc[1]);
                                const float resid = bf2f(op.v[r]) - u;
                                const float c = op.b * resid;
                                const f32x2 c2 = {c, c};
                                acc = f32x2{0.f, 0.f};
#pragma unroll
                                for (int j = 0; j < P; ++j) {
                                    const f32x2 Sn = __builtin_elementwise_fma(c2, k2[j], G2[j]);   // S_t
                                    d2[j] = __builtin_elementwise_fma(gr2, q2[j], d2[j]);           // dS += do q~^T
                                    pq2[j] = __builtin_elementwise_fma(gr2, Sn, pq2[j]);            // dq = S^T do / sqrt(dk)
                                    acc = __builtin_elementwise_fma(d2[j], k2[j], acc);
                                }
                                const float dc = lanes_sum<16>(acc[0] + acc[1]);
                                const float du = -op.b * dc;
                                const f32x2 du2 = {du, du};
                                dbp += dc * resid;
                                dvr[r] = op.b * dc;
#pragma unroll
                                for (int j = 0; j < P; ++j) {
                                    pk2[j] = __builtin_elementwise_fma(c2, d2[j], __builtin_elementwise_fma(du2, G2[j], pk2[j]));
                                    const f32x2 dG = __builtin_elementwise_fma(du2, k2[j], d2[j]);
                                    dap2 = __builtin_elementwise_fma(dG, Sp2[j], dap2);
                                    const f32x2 nd = a2 * dG;
                                    dS[r * CPL + 2 * j] = nd[0];
                                    dS[r * CPL + 2 * j + 1] = nd[1];
                                }
                            }
                            dap = dap2[0] + dap2[1];
#pragma unroll
                            for (int j = 0; j < P; ++j) {
                                pq[2 * j] = pq2[j][0];
                                pq[2 * j + 1] = pq2[j][1];
                                pk[2 * j] = pk2[j][0];
                                pk[2 * j + 1] = pk2[j][1];
                            }
                        } else {
    #pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                float G[CPL], dGv[CPL];
                                const float gr = bf2f(op.g[r]) * qscale;  // d(o_i)/d(S_ij) = q_j / sqrt(dk)
    #pragma unroll
                                for (int j = 0; j < CPL; ++j) G[j] = Sp[r * CPL + j] * op.a;
                                const float u = lanes_sum<16>(dot_pk<CPL>(G, kf));
                                const float resid = bf2f(op.v[r]) - u;
                                const float c = op.b * resid;
    #pragma unroll
                                for (int j = 0; j < CPL; ++j) {
                                    const float Sn = fmaf(c, kf[j], G[j]);                      // S_t
                                    dS[r * CPL + j] = fmaf(gr, qf[j], dS[r * CPL + j]);         // dS += do q~^T
                                    pq[j] = fmaf(gr, Sn, pq[j]);                                // dq = S^T do / sqrt(dk)
                                }
                                const float dc = lanes_sum<16>(dot_pk<CPL>(&dS[r * CPL], kf));
                                const float du = -op.b * dc;
                                dbp += dc * resid;
                                dvr[r] = op.b * dc;
    #pragma unroll
                                for (int j = 0; j < CPL; ++j) {
                                    pk[j] = fmaf(c, dS[r * CPL + j], fmaf(du, G[j], pk[j]));
                                    dGv[j] = fmaf(du, kf[j], dS[r * CPL + j]);
                                    dS[r * CPL + j] = op.a * dGv[j];
                                }
                                dap += dot_pk<CPL>(&Sp[r * CPL], dGv);
                            }
                        }
                        // sums over this wave's 16 rows (4 row quads) as a reduce-scatter: pq_j pairs with pk_j across the row pairs, then the
                        // halves pair up: quad 0 ends with the totals of pq[0 .. CPL/2), quad 1 pk[0 .. CPL/2), quad 2 pq[CPL/2 ..), quad 3 pk[CPL/2 ..)
                        dap = lanes_sum<16>(dap);
                        {
                            float z[CPL];
#pragma unroll
                            for (int j = 0; j < CPL; ++j) z[j] = rows_fold16(pq[j], pk[j]);
                            float* dst = (rq & 1) ? wk : wq;
                            if constexpr (CPL >= 2) {
                                constexpr int H = CPL / 2;
                                float w_[H];
#pragma unroll
                                for (int j = 0; j < H; ++j) w_[j] = rows_fold32(z[j], z[j + H]);
                                const int off = (rq >> 1) * H;
#pragma unroll
                                for (int j = 0; j < H; ++j) dst[(int64_t)t * DK + off + j] = w_[j];
                            } else {
                                const float w_ = rows_fold32(z[0], z[0]);
                                if (rq < 2) dst[(int64_t)t * DK] = w_;
                            }
                            const float sc = rows_fold16(dbp, dap);   // even quads: dbp of the pair, odd quads: dap
                            const float tot = rows_fold32(sc, sc);
                            if (lane == 0) pdb[slot * S + t] = tot;
                            if (lane == 16) pda[slot * S + t] = tot;
                        }
                        if (cl == 0) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) dv[(tok0 + t) * lddv + (int64_t)h * Dv + row + r] = f2bf(dvr[r]);
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < GDR_SUB; ++i) cur[i] = nxt[i];
            }
        }
#pragma unroll
        for (int e = 0; e < NS; ++e) sub[0][e] = ckN[e];
    }
    if (d_initial_state) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < CPL; ++j) d_initial_state[st_off + r * DK + j] = dS[r * CPL + j];
    }
}

// dq[b,t,hq,:] = sum over the value heads of hq and their row groups of pdq (the 1/sqrt(dk) is already in);  dk likewise;
// dbeta / dalpha [tokens, Hv] = sum over row groups.
__global__ __launch_bounds__(256) void gdr_bwd_reduce_kernel(int B, int S, int Hqk, int Hv, int RG, int DK, const float* __restrict__ pdq,
                                                             const float* __restrict__ pdk, const float* __restrict__ pdb,
                                                             const float* __restrict__ pda, bf16_t* __restrict__ dq, bf16_t* __restrict__ dk,
                                                             float* __restrict__ dbeta, float* __restrict__ dalpha, float qscale) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t nqk = (int64_t)B * S * Hqk * DK;
    const int rep = Hv / Hqk;
    if (idx < nqk) {
        const int j = (int)(idx % DK);
        const int hq = (int)((idx / DK) % Hqk);
        const int64_t tok = idx / ((int64_t)DK * Hqk);
        const int b = (int)(tok / S), t = (int)(tok % S);
        float sq = 0.f, sk = 0.f;
        for (int h = hq * rep; h < (hq + 1) * rep; ++h)
            for (int g = 0; g < RG; ++g) {
                const int64_t o = ((((int64_t)b * Hv + h) * RG + g) * S + t) * DK + j;
                sq += pdq[o];
                sk += pdk[o];
            }
        dq[idx] = f2bf(sq);
        dk[idx] = f2bf(sk);
    } else if (idx < nqk + (int64_t)B * S * Hv) {
        const int64_t e = idx - nqk;
        const int h = (int)(e % Hv);
        const int64_t tok = e / Hv;
        const int b = (int)(tok / S), t = (int)(tok % S);
        float sb = 0.f, sa = 0.f;
        for (int g = 0; g < RG; ++g) {
            const int64_t o = (((int64_t)b * Hv + h) * RG + g) * S + t;
            sb += pdb[o];
            sa += pda[o];
        }
        dbeta[e] = sb;
        dalpha[e] = sa;
    }
}

// ------------------------------------------------------------------------------------------- gated RMSNorm
// out = bf16(silu(float(gate)) * (float(o) * rsqrt(mean o^2 + eps) * w))   fp32 weight (qwen3_5_text_model.py:181-187)
template <int EPL>
__global__ __launch_bounds__(256) void gated_rmsnorm_fwd_kernel(int64_t tokens, int H, int D, const bf16_t* __restrict__ o,
                                                                const float* __restrict__ w, const bf16_t* __restrict__ gate, int64_t ldg,
                                                                bf16_t* __restrict__ out, float* __restrict__ rstd, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t total = tokens * H;
    for (int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); item < total; item += (int64_t)gridDim.x * 4) {
        const int64_t t = item / H;
        const int h = (int)(item % H);
        float v[EPL], ss = 0.f;
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            const int i = lane + 64 * j;
            v[j] = i < D ? bf2f(o[item * D + i]) : 0.f;
            ss += v[j] * v[j];
        }
        ss = wave_sum(ss);
        const float r = rsqrtf(ss / (float)D + eps);
        if (lane == 0) rstd[item] = r;
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            const int i = lane + 64 * j;
            if (i < D) {
                const float g = bf2f(gate[t * ldg + (int64_t)h * D + i]);
                out[item * D + i] = f2bf(silu_f(g) * (v[j] * r * w[i]));
            }
        }
    }
}
template <int EPL>
__global__ __launch_bounds__(256) void gated_rmsnorm_bwd_kernel(int64_t tokens, int H, int D, const bf16_t* __restrict__ o,
                                                                const float* __restrict__ w, const bf16_t* __restrict__ gate, int64_t ldg,
                                                                const float* __restrict__ rstd, const bf16_t* __restrict__ dout,
                                                                bf16_t* __restrict__ d_o, bf16_t* __restrict__ dgate, int64_t lddg,
                                                                float* __restrict__ dw_partial) {
    __shared__ float dw_lds[4][64 * EPL];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t total = tokens * H;
    float dwacc[EPL];
#pragma unroll
    for (int j = 0; j < EPL; ++j) dwacc[j] = 0.f;
    for (int64_t item = (int64_t)blockIdx.x * 4 + wave; item < total; item += (int64_t)gridDim.x * 4) {
        const int64_t t = item / H;
        const int h = (int)(item % H);
        const float r = rstd[item];
        float xh[EPL], gw[EPL], dot = 0.f;
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            const int i = lane + 64 * j;
            xh[j] = gw[j] = 0.f;
            if (i < D) {
                const float g = bf2f(gate[t * ldg + (int64_t)h * D + i]);
                const float dy = bf2f(dout[item * D + i]);
                xh[j] = bf2f(o[item * D + i]) * r;
                const float normed = xh[j] * w[i];
                dgate[t * lddg + (int64_t)h * D + i] = f2bf(dy * normed * dsilu_f(g));
                const float dn = dy * silu_f(g);
                dwacc[j] += dn * xh[j];
                gw[j] = dn * w[i];
                dot += gw[j] * xh[j];
            }
        }
        dot = wave_sum(dot) / (float)D;
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            const int i = lane + 64 * j;
            if (i < D) d_o[item * D + i] = f2bf(r * (gw[j] - xh[j] * dot));
        }
    }
#pragma unroll
    for (int j = 0; j < EPL; ++j) dw_lds[wave][lane + 64 * j] = dwacc[j];
    __syncthreads();
    for (int i = threadIdx.x; i < D; i += 256)
        dw_partial[(int64_t)blockIdx.x * D + i] = (dw_lds[0][i] + dw_lds[1][i]) + (dw_lds[2][i] + dw_lds[3][i]);
}

inline int grid1d(int64_t n, int per_block = 256) { return (int)((n + per_block - 1) / per_block); }
inline int wave_grid(int64_t items, int cap) {
    int64_t g = (items + 3) / 4;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

#define ST(s) ((hipStream_t)(s))

extern "C" int mi355_zc_weight(int64_t n, const void* scale, void* w_eff, void* stream) {
    MI355_REQUIRE(n > 0 && scale && w_eff, "zc_weight: bad arguments");
    zc_weight_kernel<<<grid1d(n), 256, 0, ST(stream)>>>(n, (const bf16_t*)scale, (bf16_t*)w_eff);
    MI355_LAUNCH_CHECK("zc_weight");
    return 0;
}

extern "C" int mi355_mrope_table(int64_t tokens, int R, int64_t ctx, const float* cosv, const float* sinv, const int64_t* position_ids,
                                 int sec_h, int sec_w, float* cos_t, float* sin_t, void* stream) {
    MI355_REQUIRE(tokens > 0 && R > 0 && R % 2 == 0 && ctx > 0 && cosv && sinv && position_ids && cos_t && sin_t, "mrope_table: bad arguments");
    MI355_REQUIRE(sec_h >= 0 && sec_w >= 0, "mrope_table: negative section");
    mrope_table_kernel<<<grid1d(tokens * (R / 2)), 256, 0, ST(stream)>>>(tokens, R, ctx, cosv, sinv, position_ids, sec_h, sec_w, cos_t, sin_t);
    MI355_LAUNCH_CHECK("mrope_table");
    return 0;
}

extern "C" int mi355_rowmask(int64_t rows, int width, const void* x, const uint8_t* mask, void* y, void* stream) {
    MI355_REQUIRE(rows > 0 && width > 0 && width % 8 == 0 && x && mask && y, "rowmask: width must be a multiple of 8");
    rowmask_kernel<<<grid1d(rows * (width / 8)), 256, 0, ST(stream)>>>(rows, width, (const bf16_t*)x, mask, (bf16_t*)y);
    MI355_LAUNCH_CHECK("rowmask");
    return 0;
}

static int check_headnorm(int64_t tokens, int H, int D, int R) {
    MI355_REQUIRE(tokens > 0 && H > 0 && D > 0 && D <= 256, "headnorm_rope: head_dim must be in 1..256, got %d", D);
    MI355_REQUIRE(R >= 0 && R <= 64 && R <= D && (R == 0 || ((R / 2) & (R / 2 - 1)) == 0), "headnorm_rope: rotation dim %d must be <= 64 with a power-of-two half", R);
    return 0;
}

extern "C" int mi355_headnorm_rope_fwd(int64_t tokens, int H, int D, int R, const void* src, int64_t ld, int64_t head_stride, const void* w,
                                       const float* cos_t, const float* sin_t, const int32_t* pos, void* out, float* rstd, float eps,
                                       void* stream) {
    if (check_headnorm(tokens, H, D, R)) return 1;
    MI355_REQUIRE(src && w && out && rstd && (R == 0 || (cos_t && sin_t && pos)), "headnorm_rope_fwd: null pointer");
    MI355_REQUIRE(head_stride >= D && ld >= (int64_t)(H - 1) * head_stride + D, "headnorm_rope_fwd: ld / head_stride do not cover the heads");
    const int grid = wave_grid(tokens * H, 4096);
#define LAUNCH(E) headnorm_rope_fwd_kernel<E><<<grid, 256, 0, ST(stream)>>>(tokens, H, D, R, (const bf16_t*)src, ld, head_stride, (const bf16_t*)w, cos_t, sin_t, pos, (bf16_t*)out, rstd, eps)
    if (D <= 64) LAUNCH(1); else if (D <= 128) LAUNCH(2); else LAUNCH(4);
#undef LAUNCH
    MI355_LAUNCH_CHECK("headnorm_rope_fwd");
    return 0;
}

extern "C" int mi355_headnorm_rope_bwd(int64_t tokens, int H, int D, int R, const void* src, int64_t ld, int64_t head_stride, const void* w,
                                       const float* cos_t, const float* sin_t, const int32_t* pos, const float* rstd, const void* dout,
                                       void* dsrc, int64_t ldd, int64_t dhead_stride, float* dw_partial, int parts, void* stream) {
    if (check_headnorm(tokens, H, D, R)) return 1;
    MI355_REQUIRE(src && w && rstd && dout && dsrc && dw_partial && (R == 0 || (cos_t && sin_t && pos)), "headnorm_rope_bwd: null pointer");
    MI355_REQUIRE(parts >= 1 && parts <= 4096, "headnorm_rope_bwd: parts out of range");
    MI355_REQUIRE(head_stride >= D && ld >= (int64_t)(H - 1) * head_stride + D && dhead_stride >= D && ldd >= (int64_t)(H - 1) * dhead_stride + D,
                  "headnorm_rope_bwd: ld / head_stride do not cover the heads");
#define LAUNCH(E) headnorm_rope_bwd_kernel<E><<<parts, 256, 0, ST(stream)>>>(tokens, H, D, R, (const bf16_t*)src, ld, head_stride, (const bf16_t*)w, cos_t, sin_t, pos, rstd, (const bf16_t*)dout, (bf16_t*)dsrc, ldd, dhead_stride, dw_partial)
    if (D <= 64) LAUNCH(1); else if (D <= 128) LAUNCH(2); else LAUNCH(4);
#undef LAUNCH
    MI355_LAUNCH_CHECK("headnorm_rope_bwd");
    return 0;
}

extern "C" int mi355_sigmoid_gate_fwd(int64_t tokens, int H, int D, const void* ctx, const void* gate, int64_t ldg, int64_t gate_head_stride,
                                      void* out, void* stream) {
    MI355_REQUIRE(tokens > 0 && H > 0 && D > 0 && D % 8 == 0 && ldg % 8 == 0 && gate_head_stride % 8 == 0, "sigmoid_gate: D, ld and head stride must be multiples of 8");
    MI355_REQUIRE(ctx && gate && out, "sigmoid_gate_fwd: null pointer");
    sigmoid_gate_fwd_kernel<<<grid1d(tokens * H * (D / 8)), 256, 0, ST(stream)>>>(tokens, H, D, (const bf16_t*)ctx, (const bf16_t*)gate, ldg, gate_head_stride, (bf16_t*)out);
    MI355_LAUNCH_CHECK("sigmoid_gate_fwd");
    return 0;
}
extern "C" int mi355_sigmoid_gate_bwd(int64_t tokens, int H, int D, const void* ctx, const void* gate, int64_t ldg, int64_t gate_head_stride,
                                      const void* dout, void* dctx, void* dgate, int64_t lddg, int64_t dgate_head_stride, void* stream) {
    MI355_REQUIRE(tokens > 0 && H > 0 && D > 0 && D % 8 == 0 && ldg % 8 == 0 && gate_head_stride % 8 == 0 && lddg % 8 == 0 && dgate_head_stride % 8 == 0,
                  "sigmoid_gate: D, ld and head stride must be multiples of 8");
    MI355_REQUIRE(ctx && gate && dout && dctx && dgate, "sigmoid_gate_bwd: null pointer");
    sigmoid_gate_bwd_kernel<<<grid1d(tokens * H * (D / 8)), 256, 0, ST(stream)>>>(tokens, H, D, (const bf16_t*)ctx, (const bf16_t*)gate, ldg, gate_head_stride, (const bf16_t*)dout, (bf16_t*)dctx, (bf16_t*)dgate, lddg, dgate_head_stride);
    MI355_LAUNCH_CHECK("sigmoid_gate_bwd");
    return 0;
}

extern "C" int mi355_gdn_gates_fwd(int64_t tokens, int Hv, const void* b_lin, const void* a_lin, int64_t ld, const float* log_A,
                                   const void* dt_bias, float* beta, float* alpha, void* stream) {
    MI355_REQUIRE(tokens > 0 && Hv > 0 && b_lin && a_lin && log_A && dt_bias && beta && alpha && ld >= Hv, "gdn_gates_fwd: bad arguments");
    gdn_gates_fwd_kernel<<<grid1d(tokens * Hv), 256, 0, ST(stream)>>>(tokens, Hv, (const bf16_t*)b_lin, (const bf16_t*)a_lin, ld, log_A, (const bf16_t*)dt_bias, beta, alpha);
    MI355_LAUNCH_CHECK("gdn_gates_fwd");
    return 0;
}
extern "C" int mi355_gdn_gates_bwd(int64_t tokens, int Hv, const void* b_lin, const void* a_lin, int64_t ld, const float* log_A,
                                   const void* dt_bias, const float* dbeta, const float* dalpha, void* db_lin, void* da_lin, int64_t ldd,
                                   float* dparam_partial, int parts, void* stream) {
    MI355_REQUIRE(tokens > 0 && Hv > 0 && Hv <= 256 && 256 % Hv == 0, "gdn_gates_bwd: the number of value heads must divide 256, got %d", Hv);
    MI355_REQUIRE(b_lin && a_lin && log_A && dt_bias && dbeta && dalpha && db_lin && da_lin && dparam_partial && parts >= 1 && parts <= 4096, "gdn_gates_bwd: bad arguments");
    gdn_gates_bwd_kernel<<<parts, 256, 0, ST(stream)>>>(tokens, Hv, (const bf16_t*)b_lin, (const bf16_t*)a_lin, ld, log_A, (const bf16_t*)dt_bias, dbeta, dalpha, (bf16_t*)db_lin, (bf16_t*)da_lin, ldd, dparam_partial);
    MI355_LAUNCH_CHECK("gdn_gates_bwd");
    return 0;
}

extern "C" int mi355_causal_conv_silu_fwd(int B, int S, int C, int ksize, const void* x, int64_t ldx, const void* w, void* y, void* stream) {
    MI355_REQUIRE(B > 0 && S > 0 && C > 0 && C % 8 == 0 && ldx % 8 == 0 && ldx >= C, "causal_conv_silu: channels and ld must be multiples of 8");
    MI355_REQUIRE(ksize == 4, "causal_conv_silu: kernel size %d not built (4 only: linear_conv_kernel_size of every Qwen3.5 config)", ksize);
    MI355_REQUIRE(x && w && y, "causal_conv_silu_fwd: null pointer");
    conv_silu_fwd_kernel<4><<<grid1d((int64_t)B * S * (C / 8)), 256, 0, ST(stream)>>>(B, S, C, (const bf16_t*)x, ldx, (const bf16_t*)w, (bf16_t*)y);
    MI355_LAUNCH_CHECK("causal_conv_silu_fwd");
    return 0;
}
extern "C" int mi355_causal_conv_silu_bwd(int B, int S, int C, int ksize, const void* x, int64_t ldx, const void* w, const void* dy, void* dc_ws,
                                          void* dx, int64_t lddx, float* dw_partial, int token_chunk, void* stream) {
    MI355_REQUIRE(B > 0 && S > 0 && C > 0 && C % 8 == 0 && ldx % 8 == 0 && lddx % 8 == 0 && ldx >= C && lddx >= C, "causal_conv_silu: channels and ld must be multiples of 8");
    MI355_REQUIRE(ksize == 4, "causal_conv_silu: kernel size %d not built (4 only)", ksize);
    MI355_REQUIRE(x && w && dy && dc_ws && dx && dw_partial && token_chunk >= 1, "causal_conv_silu_bwd: bad arguments");
    const int cps = (S + token_chunk - 1) / token_chunk;
    dim3 g1((C / 8 + 255) / 256, B * cps);
    conv_silu_bwd_dc_kernel<4><<<g1, 256, 0, ST(stream)>>>(B, S, C, token_chunk, cps, (const bf16_t*)x, ldx, (const bf16_t*)w, (const bf16_t*)dy, (bf16_t*)dc_ws, dw_partial);
    MI355_LAUNCH_CHECK("causal_conv_silu_bwd(dc)");
    conv_silu_bwd_dx_kernel<4><<<grid1d((int64_t)B * S * (C / 8)), 256, 0, ST(stream)>>>(B, S, C, (const bf16_t*)w, (const bf16_t*)dc_ws, (bf16_t*)dx, lddx);
    MI355_LAUNCH_CHECK("causal_conv_silu_bwd(dx)");
    return 0;
}

extern "C" int mi355_causal_conv_silu_step(int B, int C, int ksize, const void* x_new, int64_t ldx, void* conv_state, const void* w, void* y,
                                           void* stream) {
    MI355_REQUIRE(B > 0 && C > 0 && C % 8 == 0 && ldx % 8 == 0 && ldx >= C, "causal_conv_silu_step: channels and ld must be multiples of 8");
    MI355_REQUIRE(ksize == 4, "causal_conv_silu_step: kernel size %d not built (4 only)", ksize);
    MI355_REQUIRE(x_new && conv_state && w && y, "causal_conv_silu_step: null pointer");
    conv_silu_step_kernel<4><<<grid1d((int64_t)B * (C / 8)), 256, 0, ST(stream)>>>(B, C, (const bf16_t*)x_new, ldx, (bf16_t*)conv_state, (const bf16_t*)w, (bf16_t*)y);
    MI355_LAUNCH_CHECK("causal_conv_silu_step");
    return 0;
}

extern "C" int mi355_l2norm_fwd(int64_t tokens, int H, int D, const void* x, int64_t ldx, void* y, void* stream) {
    MI355_REQUIRE(tokens > 0 && H > 0 && D > 0 && D <= 256 && x && y && ldx >= (int64_t)H * D, "l2norm_fwd: bad arguments (head_dim <= 256)");
    const int grid = wave_grid(tokens * H, 8192);
    if (D <= 64) l2norm_fwd_kernel<1><<<grid, 256, 0, ST(stream)>>>(tokens, H, D, (const bf16_t*)x, ldx, (bf16_t*)y);
    else if (D <= 128) l2norm_fwd_kernel<2><<<grid, 256, 0, ST(stream)>>>(tokens, H, D, (const bf16_t*)x, ldx, (bf16_t*)y);
    else l2norm_fwd_kernel<4><<<grid, 256, 0, ST(stream)>>>(tokens, H, D, (const bf16_t*)x, ldx, (bf16_t*)y);
    MI355_LAUNCH_CHECK("l2norm_fwd");
    return 0;
}
extern "C" int mi355_l2norm_bwd(int64_t tokens, int H, int D, const void* x, int64_t ldx, const void* dy, void* dx, int64_t lddx, void* stream) {
    MI355_REQUIRE(tokens > 0 && H > 0 && D > 0 && D <= 256 && x && dy && dx && ldx >= (int64_t)H * D && lddx >= (int64_t)H * D, "l2norm_bwd: bad arguments (head_dim <= 256)");
    const int grid = wave_grid(tokens * H, 8192);
    if (D <= 64) l2norm_bwd_kernel<1><<<grid, 256, 0, ST(stream)>>>(tokens, H, D, (const bf16_t*)x, ldx, (const bf16_t*)dy, (bf16_t*)dx, lddx);
    else if (D <= 128) l2norm_bwd_kernel<2><<<grid, 256, 0, ST(stream)>>>(tokens, H, D, (const bf16_t*)x, ldx, (const bf16_t*)dy, (bf16_t*)dx, lddx);
    else l2norm_bwd_kernel<4><<<grid, 256, 0, ST(stream)>>>(tokens, H, D, (const bf16_t*)x, ldx, (const bf16_t*)dy, (bf16_t*)dx, lddx);
    MI355_LAUNCH_CHECK("l2norm_bwd");
    return 0;
}

static int check_gdr(int B, int S, int Hqk, int Hv, int Dk, int Dv) {
    MI355_REQUIRE(B > 0 && S > 0 && Hqk > 0 && Hv > 0 && Hv % Hqk == 0, "gated_delta_rule: value heads (%d) must be a multiple of q/k heads (%d)", Hv, Hqk);
    MI355_REQUIRE(Dk == 16 || Dk == 128, "gated_delta_rule: qk head dim %d not built (16, 128)", Dk);
    MI355_REQUIRE(Dv > 0 && Dv % 16 == 0, "gated_delta_rule: value head dim %d must be a multiple of 16", Dv);
    MI355_REQUIRE(Hv <= 65535 && B <= 65535, "gated_delta_rule: grid limits");
    return 0;
}
extern "C" int mi355_gated_delta_rule_chunk(void) { return GDR_CH; }

extern "C" int mi355_gated_delta_rule_fwd(int B, int S, int Hqk, int Hv, int Dk, int Dv, const void* q, const void* k, const void* v, int64_t ldv,
                                          const float* beta, const float* alpha, void* o, float* checkpoints, const float* initial_state,
                                          float* final_state, void* stream) {
    if (check_gdr(B, S, Hqk, Hv, Dk, Dv)) return 1;
    MI355_REQUIRE(q && k && v && beta && alpha && o && ldv >= (int64_t)Hv * Dv, "gated_delta_rule_fwd: bad arguments");
    const int CH = mi355_gated_delta_rule_chunk(), nchunk = (S + CH - 1) / CH;
    const float qs = 1.0f / sqrtf((float)Dk);
    const int rows_per_block = Dk == 128 ? 4 * (64 / 8) : 4 * (64 / 4);
    dim3 grid((Dv + rows_per_block - 1) / rows_per_block, Hv, B);
    if (Dk == 128)
        gdr_fwd_kernel<128, 8><<<grid, 256, 0, ST(stream)>>>(B, S, Hqk, Hv, Dv, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, ldv, beta, alpha, (bf16_t*)o, checkpoints, CH, nchunk, initial_state, final_state, qs);
    else
        gdr_fwd_kernel<16, 4><<<grid, 256, 0, ST(stream)>>>(B, S, Hqk, Hv, Dv, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, ldv, beta, alpha, (bf16_t*)o, checkpoints, CH, nchunk, initial_state, final_state, qs);
    MI355_LAUNCH_CHECK("gated_delta_rule_fwd");
    return 0;
}

extern "C" int64_t mi355_gated_delta_rule_bwd_workspace_bytes(int B, int S, int Hv, int Dk, int Dv) {
    const int64_t RG = Dv / 16, slots = (int64_t)B * Hv * RG;
    return 4 * (2 * slots * S * Dk + 2 * slots * S);
}

extern "C" int mi355_gated_delta_rule_bwd(int B, int S, int Hqk, int Hv, int Dk, int Dv, const void* q, const void* k, const void* v, int64_t ldv,
                                          const float* beta, const float* alpha, const float* checkpoints, const void* d_o, void* dq, void* dk,
                                          void* dv, int64_t lddv, float* dbeta, float* dalpha, void* workspace, int64_t workspace_bytes,
                                          const float* d_final_state, float* d_initial_state, void* stream) {
    if (check_gdr(B, S, Hqk, Hv, Dk, Dv)) return 1;
    MI355_REQUIRE(q && k && v && beta && alpha && checkpoints && d_o && dq && dk && dv && dbeta && dalpha && workspace, "gated_delta_rule_bwd: null pointer");
    MI355_REQUIRE(ldv >= (int64_t)Hv * Dv && lddv >= (int64_t)Hv * Dv, "gated_delta_rule_bwd: ld too small");
    MI355_REQUIRE(workspace_bytes >= mi355_gated_delta_rule_bwd_workspace_bytes(B, S, Hv, Dk, Dv), "gated_delta_rule_bwd: workspace too small (%lld bytes)", (long long)workspace_bytes);
    const int nchunk = (S + GDR_CH - 1) / GDR_CH, RG = Dv / 16;
    const int64_t slots = (int64_t)B * Hv * RG;
    float* pdq = (float*)workspace;
    float* pdk = pdq + slots * S * Dk;
    float* pdb = pdk + slots * S * Dk;
    float* pda = pdb + slots * S;
    dim3 grid((RG + 3) / 4, Hv, B);
    const float qs = 1.0f / sqrtf((float)Dk);
    if (Dk == 128)
        gdr_bwd_kernel<8><<<grid, 256, 0, ST(stream)>>>(B, S, Hqk, Hv, Dv, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, ldv, beta, alpha, checkpoints, nchunk, (const bf16_t*)d_o, (bf16_t*)dv, lddv, pdq, pdk, pdb, pda, qs, d_final_state, d_initial_state);
    else
        gdr_bwd_kernel<1><<<grid, 256, 0, ST(stream)>>>(B, S, Hqk, Hv, Dv, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, ldv, beta, alpha, checkpoints, nchunk, (const bf16_t*)d_o, (bf16_t*)dv, lddv, pdq, pdk, pdb, pda, qs, d_final_state, d_initial_state);
    MI355_LAUNCH_CHECK("gated_delta_rule_bwd");
    const int64_t n = (int64_t)B * S * Hqk * Dk + (int64_t)B * S * Hv;
    gdr_bwd_reduce_kernel<<<grid1d(n), 256, 0, ST(stream)>>>(B, S, Hqk, Hv, RG, Dk, pdq, pdk, pdb, pda, (bf16_t*)dq, (bf16_t*)dk, dbeta, dalpha, qs);
    MI355_LAUNCH_CHECK("gated_delta_rule_bwd(reduce)");
    return 0;
}

extern "C" int mi355_gated_rmsnorm_fwd(int64_t tokens, int H, int D, const void* o, const float* w, const void* gate, int64_t ldg, void* out,
                                       float* rstd, float eps, void* stream) {
    MI355_REQUIRE(tokens > 0 && H > 0 && D > 0 && D <= 256 && o && w && gate && out && rstd && ldg >= (int64_t)H * D, "gated_rmsnorm_fwd: bad arguments (head_dim <= 256)");
    const int grid = wave_grid(tokens * H, 8192);
    if (D <= 64) gated_rmsnorm_fwd_kernel<1><<<grid, 256, 0, ST(stream)>>>(tokens, H, D, (const bf16_t*)o, w, (const bf16_t*)gate, ldg, (bf16_t*)out, rstd, eps);
    else if (D <= 128) gated_rmsnorm_fwd_kernel<2><<<grid, 256, 0, ST(stream)>>>(tokens, H, D, (const bf16_t*)o, w, (const bf16_t*)gate, ldg, (bf16_t*)out, rstd, eps);
    else gated_rmsnorm_fwd_kernel<4><<<grid, 256, 0, ST(stream)>>>(tokens, H, D, (const bf16_t*)o, w, (const bf16_t*)gate, ldg, (bf16_t*)out, rstd, eps);
    MI355_LAUNCH_CHECK("gated_rmsnorm_fwd");
    return 0;
}
extern "C" int mi355_gated_rmsnorm_bwd(int64_t tokens, int H, int D, const void* o, const float* w, const void* gate, int64_t ldg, const float* rstd,
                                       const void* dout, void* d_o, void* dgate, int64_t lddg, float* dw_partial, int parts, void* stream) {
    MI355_REQUIRE(tokens > 0 && H > 0 && D > 0 && D <= 256 && o && w && gate && rstd && dout && d_o && dgate && dw_partial, "gated_rmsnorm_bwd: bad arguments (head_dim <= 256)");
    MI355_REQUIRE(ldg >= (int64_t)H * D && lddg >= (int64_t)H * D && parts >= 1 && parts <= 8192, "gated_rmsnorm_bwd: ld / parts out of range");
    if (D <= 64) gated_rmsnorm_bwd_kernel<1><<<parts, 256, 0, ST(stream)>>>(tokens, H, D, (const bf16_t*)o, w, (const bf16_t*)gate, ldg, rstd, (const bf16_t*)dout, (bf16_t*)d_o, (bf16_t*)dgate, lddg, dw_partial);
    else if (D <= 128) gated_rmsnorm_bwd_kernel<2><<<parts, 256, 0, ST(stream)>>>(tokens, H, D, (const bf16_t*)o, w, (const bf16_t*)gate, ldg, rstd, (const bf16_t*)dout, (bf16_t*)d_o, (bf16_t*)dgate, lddg, dw_partial);
    else gated_rmsnorm_bwd_kernel<4><<<parts, 256, 0, ST(stream)>>>(tokens, H, D, (const bf16_t*)o, w, (const bf16_t*)gate, ldg, rstd, (const bf16_t*)dout, (bf16_t*)d_o, (bf16_t*)dgate, lddg, dw_partial);
    MI355_LAUNCH_CHECK("gated_rmsnorm_bwd");
    return 0;
}
