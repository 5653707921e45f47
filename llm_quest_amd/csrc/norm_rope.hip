// HBM-bound row kernels: RMSNorm fwd/bwd, fused QK-RMSNorm + RoPE fwd/bwd, ViT LayerNorm (sigma + eps).
// One wave (64 lanes) per row, 16-byte vector loads, fp32 math, wavefront-shuffle reductions; rounding
// points follow the reference (RMSNorm fully in fp32, RoPE multiplies in bf16 with bf16-rounded cos/sin).
#include "common.h"

namespace {

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
__device__ __forceinline__ float rbf(float x) { return bf2f(f2bf(x)); }  // round through bf16

// ------------------------------------------------------------------------------------------- RMSNorm
// width = 64 * 8 * VPL elements per row (VPL 16-byte vectors per lane); width 1024 -> VPL 2, 128 handled apart.
template <int VPL>
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(int64_t rows, int width, const bf16_t* __restrict__ x,
                                                          const bf16_t* __restrict__ w, bf16_t* __restrict__ y,
                                                          float* __restrict__ rstd, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    for (int64_t row = row0; row < rows; row += (int64_t)gridDim.x * 4) {
        const bf16_t* xr = x + row * width;
        float v[VPL][8];
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            unpack8(*reinterpret_cast<const u32x4*>(xr + (i * 64 + lane) * 8), v[i]);
#pragma unroll
            for (int e = 0; e < 8; ++e) ss += v[i][e] * v[i][e];
        }
        ss = wave_sum(ss);
        const float r = rsqrtf(ss / (float)width + eps);
        if (lane == 0 && rstd) rstd[row] = r;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            float wv[8], o[8];
            unpack8(*reinterpret_cast<const u32x4*>(w + (i * 64 + lane) * 8), wv);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = v[i][e] * r * wv[e];
            *reinterpret_cast<u32x4*>(y + row * width + (i * 64 + lane) * 8) = pack8(o);
        }
    }
}

// generic-width fallback (width multiple of 8, <= 64*8*8): lanes stride over vectors
__global__ __launch_bounds__(256) void rmsnorm_fwd_generic(int64_t rows, int width, const bf16_t* __restrict__ x,
                                                           const bf16_t* __restrict__ w, bf16_t* __restrict__ y,
                                                           float* __restrict__ rstd, float eps) {
    const int lane = threadIdx.x & 63;
    const int nvec = width >> 3;
    const int64_t row0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    for (int64_t row = row0; row < rows; row += (int64_t)gridDim.x * 4) {
        const bf16_t* xr = x + row * width;
        float ss = 0.f;
        for (int i = lane; i < nvec; i += 64) {
            float v[8];
            unpack8(*reinterpret_cast<const u32x4*>(xr + i * 8), v);
#pragma unroll
            for (int e = 0; e < 8; ++e) ss += v[e] * v[e];
        }
        ss = wave_sum(ss);
        const float r = rsqrtf(ss / (float)width + eps);
        if (lane == 0 && rstd) rstd[row] = r;
        for (int i = lane; i < nvec; i += 64) {
            float v[8], wv[8], o[8];
            unpack8(*reinterpret_cast<const u32x4*>(xr + i * 8), v);
            unpack8(*reinterpret_cast<const u32x4*>(w + i * 8), wv);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = v[e] * r * wv[e];
            *reinterpret_cast<u32x4*>(y + row * width + i * 8) = pack8(o);
        }
    }
}

// dx = rstd * (g - xhat * mean(g * xhat)) with g = w*dy, xhat = x*rstd;  dx += dres (residual-stream grad);
// dw partial[block][c] = sum over the block's rows of dy*xhat.
// Fast path (width == 512*VPL): one wave per row, the row stays in registers (x, dy read ONCE), dw accumulated in
// registers across the rows a wave owns and merged through LDS at the end.
template <int VPL>
__global__ __launch_bounds__(256) void rmsnorm_bwd_rowreg_kernel(int64_t rows, int width, const bf16_t* __restrict__ x,
                                                                 const bf16_t* __restrict__ w, const float* __restrict__ rstd,
                                                                 const bf16_t* __restrict__ dy, const bf16_t* __restrict__ dres,
                                                                 bf16_t* __restrict__ dx, float* __restrict__ dw_partial) {
    __shared__ float dw_lds[4][512 * VPL];  // one region per wave: the block's sum is taken in a FIXED order (bit-reproducible)
    const int lane = threadIdx.x & 63, wv_id = threadIdx.x >> 6;
    float wf[VPL][8], dwacc[VPL][8];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        unpack8(*reinterpret_cast<const u32x4*>(w + (i * 64 + lane) * 8), wf[i]);
#pragma unroll
        for (int e = 0; e < 8; ++e) dwacc[i][e] = 0.f;
    }
    for (int64_t row = (int64_t)blockIdx.x * 4 + wv_id; row < rows; row += (int64_t)gridDim.x * 4) {
        float xv[VPL][8], gv[VPL][8];
        float dot = 0.f;
        const float r = rstd[row];
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            unpack8(*reinterpret_cast<const u32x4*>(x + row * width + (i * 64 + lane) * 8), xv[i]);
            unpack8(*reinterpret_cast<const u32x4*>(dy + row * width + (i * 64 + lane) * 8), gv[i]);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                dwacc[i][e] += gv[i][e] * xv[i][e] * r;
                gv[i][e] *= wf[i][e];
                dot += gv[i][e] * xv[i][e];
            }
        }
        dot = wave_sum(dot) * r * r / (float)width;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            float rs[8] = {0, 0, 0, 0, 0, 0, 0, 0}, o[8];
            if (dres) unpack8(*reinterpret_cast<const u32x4*>(dres + row * width + (i * 64 + lane) * 8), rs);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = r * (gv[i][e] - xv[i][e] * dot) + rs[e];
            *reinterpret_cast<u32x4*>(dx + row * width + (i * 64 + lane) * 8) = pack8(o);
        }
    }
#pragma unroll
    for (int i = 0; i < VPL; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) dw_lds[wv_id][(i * 64 + lane) * 8 + e] = dwacc[i][e];
    __syncthreads();
    for (int i = threadIdx.x; i < width; i += 256)
        dw_partial[(int64_t)blockIdx.x * width + i] = ((dw_lds[0][i] + dw_lds[1][i]) + dw_lds[2][i]) + dw_lds[3][i];
}

// generic width (multiple of 8): three sweeps over the row
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(int64_t rows, int width, const bf16_t* __restrict__ x,
                                                          const bf16_t* __restrict__ w, const float* __restrict__ rstd,
                                                          const bf16_t* __restrict__ dy, const bf16_t* __restrict__ dres,
                                                          bf16_t* __restrict__ dx, float* __restrict__ dw_partial) {
    extern __shared__ __attribute__((aligned(16))) float dw_lds_dyn[];  // [4][width]: one region per wave, summed in a fixed order
    const int lane = threadIdx.x & 63, wv_id = threadIdx.x >> 6;
    const int nvec = width >> 3;
    float* dw_mine = dw_lds_dyn + wv_id * width;  // element i*8+e is touched by lane i % 64 of this wave only: plain adds
    for (int i = threadIdx.x; i < 4 * width; i += 256) dw_lds_dyn[i] = 0.f;
    __syncthreads();
    for (int64_t row = (int64_t)blockIdx.x * 4 + wv_id; row < rows; row += (int64_t)gridDim.x * 4) {
        const float r = rstd[row];
        float dot = 0.f;
        for (int i = lane; i < nvec; i += 64) {
            float xv[8], dyv[8], wf[8];
            unpack8(*reinterpret_cast<const u32x4*>(x + row * width + i * 8), xv);
            unpack8(*reinterpret_cast<const u32x4*>(dy + row * width + i * 8), dyv);
            unpack8(*reinterpret_cast<const u32x4*>(w + i * 8), wf);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                dot += dyv[e] * wf[e] * xv[e];
                dw_mine[i * 8 + e] += dyv[e] * xv[e] * r;
            }
        }
        dot = wave_sum(dot) * r * r / (float)width;
        for (int i = lane; i < nvec; i += 64) {
            float xv[8], dyv[8], wf[8], o[8];
            unpack8(*reinterpret_cast<const u32x4*>(x + row * width + i * 8), xv);
            unpack8(*reinterpret_cast<const u32x4*>(dy + row * width + i * 8), dyv);
            unpack8(*reinterpret_cast<const u32x4*>(w + i * 8), wf);
            float rs[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (dres) unpack8(*reinterpret_cast<const u32x4*>(dres + row * width + i * 8), rs);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = r * (dyv[e] * wf[e] - xv[e] * dot) + rs[e];
            *reinterpret_cast<u32x4*>(dx + row * width + i * 8) = pack8(o);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < width; i += 256)
        dw_partial[(int64_t)blockIdx.x * width + i] = ((dw_lds_dyn[i] + dw_lds_dyn[width + i]) + dw_lds_dyn[2 * width + i]) + dw_lds_dyn[3 * width + i];
}

// out[n] (+)= sum_p partial[p][n]: 64 columns per block, 16 waves split the parts (the loads are the latency: keep many in flight)
__global__ __launch_bounds__(1024) void reduce_rows_kernel(int parts, int64_t n, const float* __restrict__ partial,
                                                           void* __restrict__ out, int out_dtype, int accumulate) {
    __shared__ float red[16][64];
    const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + c;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < n) {
        int p = rl;
        for (; p + 48 < parts; p += 64) {
            s0 += partial[(int64_t)p * n + i];
            s1 += partial[(int64_t)(p + 16) * n + i];
            s2 += partial[(int64_t)(p + 32) * n + i];
            s3 += partial[(int64_t)(p + 48) * n + i];
        }
        for (; p < parts; p += 16) s0 += partial[(int64_t)p * n + i];
    }
    red[rl][c] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rl != 0 || i >= n) return;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += red[k][c];
    if (out_dtype == MI355_DT_BF16) {
        bf16_t* o = reinterpret_cast<bf16_t*>(out);
        if (accumulate) s += bf2f(o[i]);
        o[i] = f2bf(s);
    } else {
        float* o = reinterpret_cast<float*>(out);
        if (accumulate) s += o[i];
        o[i] = s;
    }
}

// --------------------------------------------------------------------------- fused QK-RMSNorm + RoPE
// RoPE pairs element i with i + D/2, so a lane owns 8 consecutive elements of the first half and the matching 8 of the
// second half (two 16-byte loads); D/16 lanes cover a head and a wave processes 64/(D/16) heads of one token per pass
// (8 for D=128, 16 for D=64); a wave walks all heads of its token, so the rotary coefficients are fetched once per token.  Reference rounding points:
//   n  = bf16( float(x) * rstd * float(w) )                                                     (PytorchRMSNorm)
//   y1 = bf16( bf16(cos_b*n1) + bf16(sin_b*(-n2)) ),  y2 = bf16( bf16(cos_b*n2) + bf16(sin_b*n1) )  (RoPE.apply in bf16)
constexpr int QV = 8;  // features per lane per half: one 16-byte load
__device__ __forceinline__ void load8f(const float* p, float (&f)[QV]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { f[e] = a[e]; f[4 + e] = b[e]; }
}

template <int D>
struct QkGeom {
    static constexpr int HALF = D / 2, LPH = HALF / QV, HPW = 64 / LPH;
};

template <int LPH>
__device__ __forceinline__ float head_sum(float v) {
#pragma unroll
    for (int o = LPH / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <int D>
__global__ __launch_bounds__(256) void qknorm_rope_fwd_kernel(int64_t tokens, int Hq, int Hkv, const bf16_t* __restrict__ qkv,
                                                              const bf16_t* __restrict__ qw, const bf16_t* __restrict__ kw,
                                                              const float* __restrict__ cosT, const float* __restrict__ sinT,
                                                              const int32_t* __restrict__ pos, bf16_t* __restrict__ qo,
                                                              bf16_t* __restrict__ ko, float* __restrict__ rstd, float eps) {
    using G = QkGeom<D>;
    const int lane = threadIdx.x & 63;
    const int sub = lane / G::LPH, i = (lane % G::LPH) * QV;
    const int H = Hq + Hkv;
    const int groups = (H + G::HPW - 1) / G::HPW;
    const int64_t ld = (int64_t)(Hq + 2 * Hkv) * D;
    const bool norm = qw != nullptr;  // qw == kw == NULL: RoPE only (Qwen3.5 vision attention, qwen3_5_vision_model.py:176-177)
    // per-lane constants of the whole launch: the norm weights of this lane's 2 x 4 features, for q heads and for k heads
    float wq1[QV] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f}, wq2[QV] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f}, wk1[QV] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f}, wk2[QV] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
    if (norm) {
        unpack8(*reinterpret_cast<const u32x4*>(qw + i), wq1);
        unpack8(*reinterpret_cast<const u32x4*>(qw + G::HALF + i), wq2);
        unpack8(*reinterpret_cast<const u32x4*>(kw + i), wk1);
        unpack8(*reinterpret_cast<const u32x4*>(kw + G::HALF + i), wk2);
    }
    // one wave = one token at a time: its rotary coefficients are fetched once and reused by every head of the token
    for (int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); t < tokens; t += (int64_t)gridDim.x * 4) {
        const int64_t p = pos[t];
        float cb1[QV], sb1[QV], cb2[QV], sb2[QV];
        load8f(cosT + p * D + i, cb1);
        load8f(sinT + p * D + i, sb1);
        load8f(cosT + p * D + G::HALF + i, cb2);
        load8f(sinT + p * D + G::HALF + i, sb2);
#pragma unroll
        for (int e = 0; e < QV; ++e) { cb1[e] = rbf(cb1[e]); sb1[e] = rbf(sb1[e]); cb2[e] = rbf(cb2[e]); sb2[e] = rbf(sb2[e]); }
        for (int grp = 0; grp < groups; ++grp) {
            const int h = grp * G::HPW + sub;
            const bool valid = h < H;
            const int hh = valid ? h : 0;
            const bool isq = hh < Hq;
            const bf16_t* src = qkv + t * ld + (int64_t)hh * D;
            float x1[QV], x2[QV];
            unpack8(*reinterpret_cast<const u32x4*>(src + i), x1);
            unpack8(*reinterpret_cast<const u32x4*>(src + G::HALF + i), x2);
            float r = 1.0f;
            if (norm) {
                float ss = 0.f;
#pragma unroll
                for (int e = 0; e < QV; ++e) ss += x1[e] * x1[e] + x2[e] * x2[e];
                ss = head_sum<G::LPH>(ss);
                r = rsqrtf(ss / (float)D + eps);
            }
            float y1[QV], y2[QV];
#pragma unroll
            for (int e = 0; e < QV; ++e) {
                const float n1 = rbf(x1[e] * r * (isq ? wq1[e] : wk1[e])), n2 = rbf(x2[e] * r * (isq ? wq2[e] : wk2[e]));
                y1[e] = rbf(cb1[e] * n1) + rbf(sb1[e] * (-n2));
                y2[e] = rbf(cb2[e] * n2) + rbf(sb2[e] * n1);
            }
            if (valid) {
                bf16_t* dst = isq ? qo + t * (int64_t)Hq * D + (int64_t)h * D : ko + t * (int64_t)Hkv * D + (int64_t)(h - Hq) * D;
                *reinterpret_cast<u32x4*>(dst + i) = pack8(y1);
                *reinterpret_cast<u32x4*>(dst + G::HALF + i) = pack8(y2);
                if (i == 0 && rstd) rstd[t * H + h] = r;
            }
        }
    }
}

// backward of the same: dn = RoPE^T dy (fp32), then RMSNorm backward per head; dw partials per block.
// KONLY: the key heads only (dq == NULL: the training step, where the query heads' share is the write-out of the attention backward's dQ pass) -- no query-side weights,
// sums or selects: 196 -> fewer registers, a third wave per SIMD for a kernel that waits on its per-token loads
template <int D, bool KONLY>
__global__ __launch_bounds__(256, KONLY ? 3 : 2) void qknorm_rope_bwd_kernel(int64_t tokens, int Hq, int Hkv, const bf16_t* __restrict__ qkv,
                                                              const bf16_t* __restrict__ qw, const bf16_t* __restrict__ kw,
                                                              const float* __restrict__ cosT, const float* __restrict__ sinT,
                                                              const int32_t* __restrict__ pos, const float* __restrict__ rstd,
                                                              const bf16_t* __restrict__ dq, const bf16_t* __restrict__ dk,
                                                              bf16_t* __restrict__ dqkv, float* __restrict__ dw_partial) {
    using G = QkGeom<D>;
    __shared__ float dw_lds[4 * G::HPW][2 * D];  // one region per (wave, head slot): summed in a fixed order (bit-reproducible)
    const int lane = threadIdx.x & 63;
    const int sub = lane / G::LPH, i = (lane % G::LPH) * QV;
    const int H = Hq + Hkv;
    const int h0 = KONLY ? Hq : (dq ? 0 : Hq);  // dq == NULL: the key heads only (the query heads' share ran as the write-out of the attention backward's dQ pass)
    const int groups = (H - h0 + G::HPW - 1) / G::HPW;
    const int64_t ld = (int64_t)(Hq + 2 * Hkv) * D;
    float dwq1[QV] = {0, 0, 0, 0, 0, 0, 0, 0}, dwq2[QV] = {0, 0, 0, 0, 0, 0, 0, 0}, dwk1[QV] = {0, 0, 0, 0, 0, 0, 0, 0}, dwk2[QV] = {0, 0, 0, 0, 0, 0, 0, 0};
    const bool norm = qw != nullptr;
    float wq1[QV] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f}, wq2[QV] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f}, wk1[QV] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f}, wk2[QV] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
    if (norm) {
        if constexpr (!KONLY) {
            unpack8(*reinterpret_cast<const u32x4*>(qw + i), wq1);
            unpack8(*reinterpret_cast<const u32x4*>(qw + G::HALF + i), wq2);
        }
        unpack8(*reinterpret_cast<const u32x4*>(kw + i), wk1);
        unpack8(*reinterpret_cast<const u32x4*>(kw + G::HALF + i), wk2);
    }
    // one wave = one token at a time (rotary coefficients fetched once per token, reused by all its heads)
    for (int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); t < tokens; t += (int64_t)gridDim.x * 4) {
        const int64_t p = pos[t];
        float cb1[QV], sb1[QV], cb2[QV], sb2[QV];
        load8f(cosT + p * D + i, cb1);
        load8f(sinT + p * D + i, sb1);
        load8f(cosT + p * D + G::HALF + i, cb2);
        load8f(sinT + p * D + G::HALF + i, sb2);
#pragma unroll
        for (int e = 0; e < QV; ++e) { cb1[e] = rbf(cb1[e]); sb1[e] = rbf(sb1[e]); cb2[e] = rbf(cb2[e]); sb2[e] = rbf(sb2[e]); }
        for (int grp = 0; grp < groups; ++grp) {
            const int h = h0 + grp * G::HPW + sub;
            const bool valid = h < H;
            const int hh = valid ? h : h0;
            const bool isq = !KONLY && hh < Hq;
            const bf16_t* src = qkv + t * ld + (int64_t)hh * D;
            const bf16_t* g = isq ? dq + t * (int64_t)Hq * D + (int64_t)hh * D : dk + t * (int64_t)Hkv * D + (int64_t)(hh - Hq) * D;
            float x1[QV] = {0, 0, 0, 0, 0, 0, 0, 0}, x2[QV] = {0, 0, 0, 0, 0, 0, 0, 0}, g1[QV], g2[QV];
            if (norm) {
                unpack8(*reinterpret_cast<const u32x4*>(src + i), x1);
                unpack8(*reinterpret_cast<const u32x4*>(src + G::HALF + i), x2);
            }
            unpack8(*reinterpret_cast<const u32x4*>(g + i), g1);
            unpack8(*reinterpret_cast<const u32x4*>(g + G::HALF + i), g2);
            const float r = norm ? rstd[t * H + hh] : 1.0f;
            float dn1[QV], dn2[QV], xh1[QV], xh2[QV], w1[QV], w2[QV];
            float dot = 0.f;
#pragma unroll
            for (int e = 0; e < QV; ++e) {
                // y1 = c1*n1 - s1*n2 ; y2 = c2*n2 + s2*n1
                w1[e] = isq ? wq1[e] : wk1[e];
                w2[e] = isq ? wq2[e] : wk2[e];
                dn1[e] = cb1[e] * g1[e] + sb2[e] * g2[e];
                dn2[e] = cb2[e] * g2[e] - sb1[e] * g1[e];
                xh1[e] = x1[e] * r;
                xh2[e] = x2[e] * r;
                dot += dn1[e] * w1[e] * xh1[e] + dn2[e] * w2[e] * xh2[e];
            }
            dot = head_sum<G::LPH>(dot) / (float)D;
            if (valid) {
                float d1[QV], d2[QV];
#pragma unroll
                for (int e = 0; e < QV; ++e) {
                    d1[e] = norm ? r * (dn1[e] * w1[e] - xh1[e] * dot) : dn1[e];
                    d2[e] = norm ? r * (dn2[e] * w2[e] - xh2[e] * dot) : dn2[e];
                    if (isq) { dwq1[e] += dn1[e] * xh1[e]; dwq2[e] += dn2[e] * xh2[e]; } else { dwk1[e] += dn1[e] * xh1[e]; dwk2[e] += dn2[e] * xh2[e]; }
                }
                bf16_t* dst = dqkv + t * ld + (int64_t)h * D;
                *reinterpret_cast<u32x4*>(dst + i) = pack8(d1);
                *reinterpret_cast<u32x4*>(dst + G::HALF + i) = pack8(d2);
            }
        }
    }
    float* mine = dw_lds[(threadIdx.x >> 6) * G::HPW + sub];
#pragma unroll
    for (int e = 0; e < QV; ++e) {
        mine[i + e] = dwq1[e];
        mine[G::HALF + i + e] = dwq2[e];
        mine[D + i + e] = dwk1[e];
        mine[D + G::HALF + i + e] = dwk2[e];
    }
    __syncthreads();
    for (int j = threadIdx.x; j < 2 * D; j += 256) {
        float acc = 0.f;
#pragma unroll
        for (int rgn = 0; rgn < 4 * G::HPW; ++rgn) acc += dw_lds[rgn][j];
        dw_partial[(int64_t)blockIdx.x * 2 * D + j] = acc;
    }
}

// ------------------------------------------------------------------------ ViT LayerNorm (sigma + eps)
// x fp32 rows; y = scale*(x-mean)/(sqrt(mean((x-mean)^2)) + eps) + shift
template <int OUT_DT>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(int64_t rows, int width, const float* __restrict__ x,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            void* __restrict__ y, float* __restrict__ mean_out,
                                                            float* __restrict__ rsig_out, float eps, int mode) {
    const int lane = threadIdx.x & 63;
    const int nvec = width >> 2;
    const int64_t row0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    for (int64_t row = row0; row < rows; row += (int64_t)gridDim.x * 4) {
        const float* xr = x + row * width;
        float s = 0.f;
        for (int i = lane; i < nvec; i += 64) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(xr + i * 4);
            s += v[0] + v[1] + v[2] + v[3];
        }
        const float mu = wave_sum(s) / (float)width;
        float ss = 0.f;
        for (int i = lane; i < nvec; i += 64) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(xr + i * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) ss += (v[e] - mu) * (v[e] - mu);
        }
        const float var = wave_sum(ss) / (float)width;
        // mode 0: ViT/GPT LayerNorm of the reference, eps added to sigma; mode 1: nn.LayerNorm, eps inside the square root
        const float inv = mode == 0 ? 1.0f / (sqrtf(var) + eps) : rsqrtf(var + eps);
        if (lane == 0) {
            if (mean_out) mean_out[row] = mu;
            if (rsig_out) rsig_out[row] = inv;
        }
        for (int i = lane; i < nvec; i += 64) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(xr + i * 4);
            const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + i * 4);
            const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + i * 4);
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = sc[e] * ((v[e] - mu) * inv) + sh[e];
            if constexpr (OUT_DT == MI355_DT_BF16) {
                u32x2 pk = {pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3])};
                *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(y) + row * width + i * 4) = pk;
            } else if constexpr (OUT_DT == MI355_DT_SPLIT3) {  // [hi | lo | hi]: what mi355_split3_bf16 makes of the fp32 row, without the round trip
                float lo[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) lo[e] = o[e] - bf2f(f2bf(o[e]));
                const u32x2 hi = {pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3])}, lw = {pack_bf2(lo[0], lo[1]), pack_bf2(lo[2], lo[3])};
                bf16_t* d = reinterpret_cast<bf16_t*>(y) + row * 3 * (int64_t)width + i * 4;
                *reinterpret_cast<u32x2*>(d) = hi;
                *reinterpret_cast<u32x2*>(d + width) = lw;
                *reinterpret_cast<u32x2*>(d + 2 * width) = hi;
            } else {
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(y) + row * width + i * 4) = (f32x4){o[0], o[1], o[2], o[3]};
            }
        }
    }
}

// The same with the row held in registers (width = 256 NV: 768 and 1024): ONE read of x instead of three passes over it.  Same per-lane accumulation order and the
// same cross-lane sums as layernorm_fwd_kernel, so the same bits.
template <int OUT_DT, int NV>
__global__ __launch_bounds__(256) void layernorm_fwd_rowreg_kernel(int64_t rows, const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                                                                   void* __restrict__ y, float* __restrict__ mean_out, float* __restrict__ rsig_out, float eps, int mode) {
    constexpr int width = 256 * NV;
    const int lane = threadIdx.x & 63;
    f32x4 sc[NV], sh[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        sc[j] = *reinterpret_cast<const f32x4*>(scale + (lane + 64 * j) * 4);
        sh[j] = *reinterpret_cast<const f32x4*>(shift + (lane + 64 * j) * 4);
    }
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (int64_t)gridDim.x * 4) {
        const float* xr = x + row * width;
        f32x4 v[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) v[j] = *reinterpret_cast<const f32x4*>(xr + (lane + 64 * j) * 4);
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) s += v[j][0] + v[j][1] + v[j][2] + v[j][3];
        const float mu = wave_sum(s) / (float)width;
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) ss += (v[j][e] - mu) * (v[j][e] - mu);
        const float var = wave_sum(ss) / (float)width;
        const float inv = mode == 0 ? 1.0f / (sqrtf(var) + eps) : rsqrtf(var + eps);
        if (lane == 0) {
            if (mean_out) mean_out[row] = mu;
            if (rsig_out) rsig_out[row] = inv;
        }
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = lane + 64 * j;
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = sc[j][e] * ((v[j][e] - mu) * inv) + sh[j][e];
            if constexpr (OUT_DT == MI355_DT_BF16) {
                u32x2 pk = {pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3])};
                *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(y) + row * width + i * 4) = pk;
            } else if constexpr (OUT_DT == MI355_DT_SPLIT3) {
                float lo[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) lo[e] = o[e] - bf2f(f2bf(o[e]));
                const u32x2 hi = {pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3])}, lw = {pack_bf2(lo[0], lo[1]), pack_bf2(lo[2], lo[3])};
                bf16_t* d = reinterpret_cast<bf16_t*>(y) + row * 3 * (int64_t)width + i * 4;
                *reinterpret_cast<u32x2*>(d) = hi;
                *reinterpret_cast<u32x2*>(d + width) = lw;
                *reinterpret_cast<u32x2*>(d + 2 * width) = hi;
            } else {
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(y) + row * width + i * 4) = (f32x4){o[0], o[1], o[2], o[3]};
            }
        }
    }
}

// backward of y = scale*n + shift, n = (x-mean)*rsig, rsig = 1/(sigma+eps):
//   dn = dy*scale;  dx = rsig * (dn - mean(dn) - n * mean(dn*n) * (1/rsig)/(1/rsig - eps)) (+ dres)
//   dscale = sum_rows dy*n, dshift = sum_rows dy   (per-block partials [parts][2*width], LDS-merged)
template <int DY_DT>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(int64_t rows, int width, const float* __restrict__ x,
                                                            const float* __restrict__ scale, const float* __restrict__ mean,
                                                            const float* __restrict__ rsig, const void* __restrict__ dy,
                                                            const float* __restrict__ dres, float* __restrict__ dx,
                                                            float* __restrict__ dparam_partial, float eps, int mode) {
    extern __shared__ __attribute__((aligned(16))) float dp_lds[];  // [4][2*width]: one region per wave, summed in a fixed order
    const int lane = threadIdx.x & 63;
    float* dp_mine = dp_lds + (threadIdx.x >> 6) * 2 * width;  // column c belongs to lane c % 64 of this wave only: plain adds
    for (int i = threadIdx.x; i < 8 * width; i += 256) dp_lds[i] = 0.f;
    __syncthreads();
    auto load_dy = [&](int64_t row, int c) -> float {
        if (DY_DT == MI355_DT_BF16) return bf2f(reinterpret_cast<const bf16_t*>(dy)[row * width + c]);
        return reinterpret_cast<const float*>(dy)[row * width + c];
    };
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (int64_t)gridDim.x * 4) {
        const float mu = mean[row], rs = rsig[row];
        float s1 = 0.f, s2 = 0.f;
        for (int c = lane; c < width; c += 64) {
            const float n = (x[row * width + c] - mu) * rs;
            const float g = load_dy(row, c);
            const float dn = g * scale[c];
            s1 += dn;
            s2 += dn * n;
            dp_mine[c] += g * n;
            dp_mine[width + c] += g;
        }
        s1 = wave_sum(s1) / (float)width;
        s2 = wave_sum(s2) / (float)width;
        const float s_over_sigma = mode == 0 ? (1.0f / rs) / (1.0f / rs - eps) : 1.0f;
        for (int c = lane; c < width; c += 64) {
            const float n = (x[row * width + c] - mu) * rs;
            const float dn = load_dy(row, c) * scale[c];
            dx[row * width + c] = rs * (dn - s1 - n * s2 * s_over_sigma) + (dres ? dres[row * width + c] : 0.f);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * width; i += 256)
        dparam_partial[(int64_t)blockIdx.x * 2 * width + i] = ((dp_lds[i] + dp_lds[2 * width + i]) + dp_lds[4 * width + i]) + dp_lds[6 * width + i];
}

// Same, for widths that are multiples of 256 (ViT 768): the row lives in registers (VPL float4 per lane), the parameter
// gradients accumulate in registers across the rows of a wave and meet in LDS once per block -- no per-element LDS atomics.
template <int DY_DT, int VPL>
__global__ __launch_bounds__(256) void layernorm_bwd_rowreg_kernel(int64_t rows, int width, const float* __restrict__ x,
                                                                   const float* __restrict__ scale, const float* __restrict__ mean,
                                                                   const float* __restrict__ rsig, const void* __restrict__ dy,
                                                                   const float* __restrict__ dres, float* __restrict__ dx,
                                                                   float* __restrict__ dparam_partial, float eps, int mode) {
    extern __shared__ __attribute__((aligned(16))) float dp_lds[];  // [4][2*width]: one region per wave, summed in a fixed order
    const int lane = threadIdx.x & 63;
    float* dp_mine = dp_lds + (threadIdx.x >> 6) * 2 * width;
    f32x4 sc[VPL], dsc[VPL], dsh[VPL];
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
        sc[v] = *reinterpret_cast<const f32x4*>(scale + (v * 64 + lane) * 4);
        dsc[v] = (f32x4){0.f, 0.f, 0.f, 0.f};
        dsh[v] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (int64_t)gridDim.x * 4) {
        const float mu = mean[row], rs = rsig[row];
        f32x4 n[VPL], dn[VPL];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
            const int c = (v * 64 + lane) * 4;
            const f32x4 xv = *reinterpret_cast<const f32x4*>(x + row * width + c);
            f32x4 g;
            if constexpr (DY_DT == MI355_DT_BF16) {
                const u32x2 gv = *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(dy) + row * width + c);
                g = (f32x4){__uint_as_float(gv[0] << 16), __uint_as_float(gv[0] & 0xffff0000u), __uint_as_float(gv[1] << 16), __uint_as_float(gv[1] & 0xffff0000u)};
            } else {
                g = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(dy) + row * width + c);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                n[v][e] = (xv[e] - mu) * rs;
                dn[v][e] = g[e] * sc[v][e];
                s1 += dn[v][e];
                s2 += dn[v][e] * n[v][e];
                dsc[v][e] += g[e] * n[v][e];
                dsh[v][e] += g[e];
            }
        }
        s1 = wave_sum(s1) / (float)width;
        s2 = wave_sum(s2) / (float)width;
        const float s_over_sigma = mode == 0 ? (1.0f / rs) / (1.0f / rs - eps) : 1.0f;
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
            const int c = (v * 64 + lane) * 4;
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = rs * (dn[v][e] - s1 - n[v][e] * s2 * s_over_sigma);
            if (dres) o += *reinterpret_cast<const f32x4*>(dres + row * width + c);
            *reinterpret_cast<f32x4*>(dx + row * width + c) = o;
        }
    }
#pragma unroll
    for (int v = 0; v < VPL; ++v)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            dp_mine[(v * 64 + lane) * 4 + e] = dsc[v][e];
            dp_mine[width + (v * 64 + lane) * 4 + e] = dsh[v][e];
        }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * width; i += 256)
        dparam_partial[(int64_t)blockIdx.x * 2 * width + i] = ((dp_lds[i] + dp_lds[2 * width + i]) + dp_lds[4 * width + i]) + dp_lds[6 * width + i];
}

inline int row_grid(int64_t rows) {
    int64_t g = (rows + 3) / 4;
    return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

}  // namespace

extern "C" int mi355_rmsnorm_fwd(int64_t rows, int width, const void* x, const void* w, void* y, float* rstd, float eps,
                                 void* stream) {
    MI355_REQUIRE(rows > 0 && width > 0 && (width & 7) == 0 && width <= 8192, "mi355_rmsnorm_fwd: width must be a multiple of 8 and <= 8192 (got %d)", width);
    MI355_REQUIRE(x && w && y, "mi355_rmsnorm_fwd: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const int grid = row_grid(rows);
    if (width == 1024)
        hipLaunchKernelGGL(rmsnorm_fwd_kernel<2>, dim3(grid), dim3(256), 0, s, rows, width, (const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y, rstd, eps);
    else
        hipLaunchKernelGGL(rmsnorm_fwd_generic, dim3(grid), dim3(256), 0, s, rows, width, (const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y, rstd, eps);
    MI355_LAUNCH_CHECK("mi355_rmsnorm_fwd");
    return 0;
}

extern "C" int mi355_rmsnorm_bwd(int64_t rows, int width, const void* x, const void* w, const float* rstd, const void* dy,
                                 const void* dres, void* dx, float* dw_partial, int parts, void* stream) {
    MI355_REQUIRE(rows > 0 && (width & 7) == 0 && width <= 4096, "mi355_rmsnorm_bwd: bad width %d (multiple of 8, <= 4096: four per-wave LDS regions of width floats)", width);
    MI355_REQUIRE(x && w && rstd && dy && dx && dw_partial && parts > 0, "mi355_rmsnorm_bwd: null pointer / parts");
    hipStream_t s = (hipStream_t)stream;
    if (width == 1024)
        hipLaunchKernelGGL(rmsnorm_bwd_rowreg_kernel<2>, dim3(parts), dim3(256), 0, s, rows, width, (const bf16_t*)x,
                           (const bf16_t*)w, rstd, (const bf16_t*)dy, (const bf16_t*)dres, (bf16_t*)dx, dw_partial);
    else if (width == 512)
        hipLaunchKernelGGL(rmsnorm_bwd_rowreg_kernel<1>, dim3(parts), dim3(256), 0, s, rows, width, (const bf16_t*)x,
                           (const bf16_t*)w, rstd, (const bf16_t*)dy, (const bf16_t*)dres, (bf16_t*)dx, dw_partial);
    else
        hipLaunchKernelGGL(rmsnorm_bwd_kernel, dim3(parts), dim3(256), 4 * width * sizeof(float), s, rows, width, (const bf16_t*)x,
                           (const bf16_t*)w, rstd, (const bf16_t*)dy, (const bf16_t*)dres, (bf16_t*)dx, dw_partial);
    MI355_LAUNCH_CHECK("mi355_rmsnorm_bwd");
    return 0;
}

extern "C" int mi355_reduce_rows_f32(int parts, int64_t n, const float* partial, void* out, int out_dtype, int accumulate,
                                     void* stream) {
    MI355_REQUIRE(parts > 0 && n > 0 && partial && out, "mi355_reduce_rows_f32: bad arguments");
    hipLaunchKernelGGL(reduce_rows_kernel, dim3((unsigned)((n + 63) / 64)), dim3(1024), 0, (hipStream_t)stream, parts, n,
                       partial, out, out_dtype, accumulate);
    MI355_LAUNCH_CHECK("mi355_reduce_rows_f32");
    return 0;
}

extern "C" int mi355_qknorm_rope_fwd(int64_t tokens, int Hq, int Hkv, int D, const void* qkv, const void* qw, const void* kw,
                                     const float* cos, const float* sin, const int32_t* pos, void* q_out, void* k_out,
                                     float* rstd, float eps, void* stream) {
    MI355_REQUIRE(D == 128 || D == 64, "mi355_qknorm_rope_fwd: head_dim must be 64 or 128 (got %d)", D);
    MI355_REQUIRE(tokens > 0 && Hq > 0 && Hkv > 0 && qkv && cos && sin && pos && q_out && k_out, "mi355_qknorm_rope_fwd: bad arguments");
    MI355_REQUIRE((qw == nullptr) == (kw == nullptr) && (qw == nullptr || rstd != nullptr), "mi355_qknorm_rope_fwd: pass both norm weights (and rstd) or neither (RoPE only)");
    const int grid = row_grid(tokens);
    hipStream_t s = (hipStream_t)stream;
    if (D == 128)
        hipLaunchKernelGGL(qknorm_rope_fwd_kernel<128>, dim3(grid), dim3(256), 0, s, tokens, Hq, Hkv, (const bf16_t*)qkv, (const bf16_t*)qw, (const bf16_t*)kw, cos, sin, pos, (bf16_t*)q_out, (bf16_t*)k_out, rstd, eps);
    else
        hipLaunchKernelGGL(qknorm_rope_fwd_kernel<64>, dim3(grid), dim3(256), 0, s, tokens, Hq, Hkv, (const bf16_t*)qkv, (const bf16_t*)qw, (const bf16_t*)kw, cos, sin, pos, (bf16_t*)q_out, (bf16_t*)k_out, rstd, eps);
    MI355_LAUNCH_CHECK("mi355_qknorm_rope_fwd");
    return 0;
}

extern "C" int mi355_qknorm_rope_bwd(int64_t tokens, int Hq, int Hkv, int D, const void* qkv, const void* qw, const void* kw,
                                     const float* cos, const float* sin, const int32_t* pos, const float* rstd,
                                     const void* dq, const void* dk, void* dqkv, float* dw_partial, int parts, void* stream) {
    MI355_REQUIRE(D == 128 || D == 64, "mi355_qknorm_rope_bwd: head_dim must be 64 or 128 (got %d)", D);
    MI355_REQUIRE(tokens > 0 && parts > 0 && qkv && cos && sin && pos && dk && dqkv && dw_partial, "mi355_qknorm_rope_bwd: bad arguments");  // dq may be NULL: key heads only
    MI355_REQUIRE((qw == nullptr) == (kw == nullptr) && (qw == nullptr || rstd != nullptr), "mi355_qknorm_rope_bwd: pass both norm weights (and rstd) or neither (RoPE only)");
    hipStream_t s = (hipStream_t)stream;
#define QKB(DD, KO) hipLaunchKernelGGL((qknorm_rope_bwd_kernel<DD, KO>), dim3(parts), dim3(256), 0, s, tokens, Hq, Hkv, (const bf16_t*)qkv, (const bf16_t*)qw, (const bf16_t*)kw, cos, sin, pos, rstd, (const bf16_t*)dq, (const bf16_t*)dk, (bf16_t*)dqkv, dw_partial)
    if (D == 128) {
        if (dq) QKB(128, false); else QKB(128, true);
    } else {
        if (dq) QKB(64, false); else QKB(64, true);
    }
#undef QKB
    MI355_LAUNCH_CHECK("mi355_qknorm_rope_bwd");
    return 0;
}

extern "C" int mi355_layernorm_fwd(int64_t rows, int width, const float* x, const float* scale, const float* shift, void* y,
                                   int y_dtype, float* mean, float* rsig, float eps, int mode, void* stream) {
    MI355_REQUIRE(rows > 0 && width > 0 && (width & 3) == 0, "mi355_layernorm_fwd: width must be a multiple of 4 (got %d)", width);
    MI355_REQUIRE(x && scale && shift && y, "mi355_layernorm_fwd: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const int grid = row_grid(rows);
    const bool al16 = (((uintptr_t)x | (uintptr_t)scale | (uintptr_t)shift | (uintptr_t)y) & 15) == 0;
    if ((width == 768 || width == 1024) && al16) {  // the row in registers: one read of x
#define LN_ROWREG(DT)                                                                                                                                         \
    do {                                                                                                                                                      \
        if (width == 768) hipLaunchKernelGGL((layernorm_fwd_rowreg_kernel<DT, 3>), dim3(grid), dim3(256), 0, s, rows, x, scale, shift, y, mean, rsig, eps, mode); \
        else hipLaunchKernelGGL((layernorm_fwd_rowreg_kernel<DT, 4>), dim3(grid), dim3(256), 0, s, rows, x, scale, shift, y, mean, rsig, eps, mode);              \
    } while (0)
        if (y_dtype == MI355_DT_BF16) LN_ROWREG(MI355_DT_BF16);
        else if (y_dtype == MI355_DT_SPLIT3) LN_ROWREG(MI355_DT_SPLIT3);
        else LN_ROWREG(MI355_DT_F32);
#undef LN_ROWREG
        MI355_LAUNCH_CHECK("mi355_layernorm_fwd");
        return 0;
    }
    if (y_dtype == MI355_DT_BF16)
        hipLaunchKernelGGL(layernorm_fwd_kernel<MI355_DT_BF16>, dim3(grid), dim3(256), 0, s, rows, width, x, scale, shift, y, mean, rsig, eps, mode);
    else if (y_dtype == MI355_DT_SPLIT3)
        hipLaunchKernelGGL(layernorm_fwd_kernel<MI355_DT_SPLIT3>, dim3(grid), dim3(256), 0, s, rows, width, x, scale, shift, y, mean, rsig, eps, mode);
    else
        hipLaunchKernelGGL(layernorm_fwd_kernel<MI355_DT_F32>, dim3(grid), dim3(256), 0, s, rows, width, x, scale, shift, y, mean, rsig, eps, mode);
    MI355_LAUNCH_CHECK("mi355_layernorm_fwd");
    return 0;
}

extern "C" int mi355_layernorm_bwd(int64_t rows, int width, const float* x, const float* scale, const float* mean, const float* rsig,
                                   const void* dy, int dy_dtype, const float* dres, float* dx, float* dparam_partial, int parts,
                                   float eps, int mode, void* stream) {
    MI355_REQUIRE(rows > 0 && width > 0 && width <= 2048 && parts > 0, "mi355_layernorm_bwd: bad shape (width %d: 1 .. 2048, four per-wave LDS regions of 2 * width floats)", width);
    MI355_REQUIRE(x && scale && mean && rsig && dy && dx && dparam_partial, "mi355_layernorm_bwd: null pointer");
    hipStream_t s = (hipStream_t)stream;
#define LN_BWD(KERNEL) \
    hipLaunchKernelGGL(KERNEL, dim3(parts), dim3(256), 8 * width * sizeof(float), s, rows, width, x, scale, mean, rsig, dy, dres, dx, dparam_partial, eps, mode)
    const bool aligned = (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx | (uintptr_t)dres | (uintptr_t)scale) & 15) == 0;
    if (width == 768 && aligned) {
        if (dy_dtype == MI355_DT_BF16) LN_BWD((layernorm_bwd_rowreg_kernel<MI355_DT_BF16, 3>));
        else LN_BWD((layernorm_bwd_rowreg_kernel<MI355_DT_F32, 3>));
    } else if (width == 1024 && aligned) {
        if (dy_dtype == MI355_DT_BF16) LN_BWD((layernorm_bwd_rowreg_kernel<MI355_DT_BF16, 4>));
        else LN_BWD((layernorm_bwd_rowreg_kernel<MI355_DT_F32, 4>));
    } else if (dy_dtype == MI355_DT_BF16) {
        LN_BWD(layernorm_bwd_kernel<MI355_DT_BF16>);
    } else {
        LN_BWD(layernorm_bwd_kernel<MI355_DT_F32>);
    }
#undef LN_BWD
    MI355_LAUNCH_CHECK("mi355_layernorm_bwd");
    return 0;
}
