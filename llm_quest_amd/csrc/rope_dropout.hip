// Stand-alone rotary embedding (the Python boundary's RoPE.apply / RoPE.apply_mrope / VisionRoPE.apply on device tensors,
// reference common/rope.py:180-243, 297-358, 484-500) and counter-based dropout (nn.Dropout sites of the ViT path,
// reference vit_model.py:146, vit_transformer_block.py:117,124, vit_engine.py:51).  Both are HBM-bound element-wise
// passes: 16-byte accesses, one pass over the tensor, nothing staged.
#include "common.h"

namespace {

inline int grid_for(int64_t work_items, int per_block) {
    int64_t g = (work_items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > 16384) g = 16384;
    return (int)g;
}

template <typename T>
struct Elem;
template <>
struct Elem<bf16_t> {
    static __device__ __forceinline__ float ld(bf16_t v) { return bf2f(v); }
    static __device__ __forceinline__ bf16_t st(float v) { return f2bf(v); }
    // the reference multiplies in the tensor's dtype: cos/sin are cast to bf16 first, every product and the sum round to bf16
    static __device__ __forceinline__ float rnd(float v) { return bf2f(f2bf(v)); }
};
template <>
struct Elem<float> {
    static __device__ __forceinline__ float ld(float v) { return v; }
    static __device__ __forceinline__ float st(float v) { return v; }
    static __device__ __forceinline__ float rnd(float v) { return v; }
};

// x viewed as (b, h, s, D) with element strides (sb, sh, ss, 1); out likewise.  Features [0, R) rotate as two halves, [R, D)
// pass through.  Coefficient row of (b, s): idx ? idx[b * S + s] : s.  TRANSPOSE = the backward map (dx from dy).
// One thread = V consecutive features of the first half AND their partners in the second half, or V pass-through features.
template <typename T, int V, bool TRANSPOSE>
__global__ __launch_bounds__(256) void rope_apply_kernel(int64_t rows, int H, int S, int D, int R, const T* __restrict__ x, int64_t sb, int64_t sh,
                                                         int64_t ss, const float* __restrict__ cosr, const float* __restrict__ sinr,
                                                         const int32_t* __restrict__ idx, int64_t table_rows, T* __restrict__ out, int64_t ob, int64_t oh, int64_t os) {
    const int half = R >> 1;
    const int cpr = half / V + (D - R) / V;  // chunks per row
    const int64_t total = rows * cpr;
    for (int64_t item = (int64_t)blockIdx.x * 256 + threadIdx.x; item < total; item += (int64_t)gridDim.x * 256) {
        const int64_t row = item / cpr;
        const int c = (int)(item % cpr);
        const int s = (int)(row % S);
        const int h = (int)((row / S) % H);
        const int64_t b = row / ((int64_t)S * H);
        const T* xr = x + b * sb + (int64_t)h * sh + (int64_t)s * ss;
        T* orow = out + b * ob + (int64_t)h * oh + (int64_t)s * os;
        if (c >= half / V) {  // pass-through features
            const int j = R + (c - half / V) * V;
#pragma unroll
            for (int e = 0; e < V; ++e) orow[j + e] = xr[j + e];
            continue;
        }
        const int j = c * V;
        // a position outside the coefficient table (the reference would raise an index error) never reads out of bounds: its row comes out as NaN
        int64_t crow = idx ? (int64_t)idx[b * S + s] : (int64_t)s;
        const bool bad_row = crow < 0 || crow >= table_rows;
        crow = bad_row ? 0 : crow;
        const float* cr = cosr + crow * R;
        const float* sr = sinr + crow * R;
        T o1[V], o2[V];
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const float x1 = Elem<T>::ld(xr[j + e]), x2 = Elem<T>::ld(xr[j + half + e]);
            const float c1 = Elem<T>::rnd(cr[j + e]), c2 = Elem<T>::rnd(cr[j + half + e]);
            const float s1 = Elem<T>::rnd(sr[j + e]), s2 = Elem<T>::rnd(sr[j + half + e]);
            float r1, r2;
            if (!TRANSPOSE) {  // y = cos * x + sin * cat(-x2, x1)
                r1 = Elem<T>::rnd(Elem<T>::rnd(c1 * x1) + Elem<T>::rnd(s1 * (-x2)));
                r2 = Elem<T>::rnd(Elem<T>::rnd(c2 * x2) + Elem<T>::rnd(s2 * x1));
            } else {  // dx = cos * dy + (z2, -z1), z = sin * dy
                r1 = Elem<T>::rnd(Elem<T>::rnd(c1 * x1) + Elem<T>::rnd(s2 * x2));
                r2 = Elem<T>::rnd(Elem<T>::rnd(c2 * x2) + (-Elem<T>::rnd(s1 * x1)));
            }
            o1[e] = Elem<T>::st(bad_row ? __builtin_nanf("") : r1);
            o2[e] = Elem<T>::st(bad_row ? __builtin_nanf("") : r2);
        }
#pragma unroll
        for (int e = 0; e < V; ++e) {
            orow[j + e] = o1[e];
            orow[j + half + e] = o2[e];
        }
    }
}

// y[i] = (res ? res[i] : 0) + (keep(i) ? x[i] / (1 - p) : 0); keep(i) = Philox(seed; i / 4, offset)[i % 4] >= thresh.
template <typename TX, typename TY>
__global__ __launch_bounds__(256) void dropout_kernel(int64_t n, const TX* __restrict__ x, const TY* __restrict__ res, TY* __restrict__ y,
                                                      unsigned thresh, float inv_keep, unsigned k0, unsigned k1, unsigned off_lo, unsigned off_hi) {
    const int64_t groups = (n + 3) >> 2;
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < groups; g += (int64_t)gridDim.x * 256) {
        unsigned bits[4];
        philox4x32_10((unsigned)g, (unsigned)(g >> 32), off_lo, off_hi, k0, k1, bits);
        const int64_t i0 = g << 2;
        if (i0 + 3 < n) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = bits[e] >= thresh ? Elem<TX>::ld(x[i0 + e]) * inv_keep : 0.f;
            if (res) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = Elem<TY>::ld(res[i0 + e]) + Elem<TY>::rnd(v[e]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) y[i0 + e] = Elem<TY>::st(v[e]);
        } else {
            for (int e = 0; e < 4 && i0 + e < n; ++e) {
                float v = bits[e] >= thresh ? Elem<TX>::ld(x[i0 + e]) * inv_keep : 0.f;
                if (res) v = Elem<TY>::ld(res[i0 + e]) + Elem<TY>::rnd(v);
                y[i0 + e] = Elem<TY>::st(v);
            }
        }
    }
}

}  // namespace

#define STREAM ((hipStream_t)stream)

extern "C" int mi355_rope_apply(int B, int H, int S, int D, int R, const void* x, int dtype, int64_t sb, int64_t sh, int64_t ss, const float* cos_t,
                                const float* sin_t, int64_t table_rows, const int32_t* idx, void* out, int64_t ob, int64_t oh, int64_t os,
                                int transpose, void* stream) {
    MI355_REQUIRE(B > 0 && H > 0 && S > 0 && D > 0 && x && cos_t && sin_t && out, "mi355_rope_apply: bad arguments");
    MI355_REQUIRE(R > 0 && R <= D && (R & 1) == 0, "mi355_rope_apply: rotation width %d must be even and <= head_dim %d", R, D);
    MI355_REQUIRE(dtype == MI355_DT_BF16 || dtype == MI355_DT_F32, "mi355_rope_apply: dtype must be bf16 or fp32");
    MI355_REQUIRE(table_rows > 0 && (idx || table_rows >= S), "mi355_rope_apply: coefficient table has %lld rows, sequence length is %d", (long long)table_rows, S);
    const int64_t rows = (int64_t)B * H * S;
    const int half = R / 2;
    const int vw = dtype == MI355_DT_BF16 ? 8 : 4;
    const uintptr_t al = (uintptr_t)x | (uintptr_t)out;
    const bool vec = half % vw == 0 && (D - R) % vw == 0 && (sb | sh | ss | ob | oh | os) % vw == 0 && al % 16 == 0;
    const int cpr = vec ? half / vw + (D - R) / vw : half + (D - R);
    const int grid = grid_for(rows * cpr, 256);
#define RL(T, V, TR) \
    hipLaunchKernelGGL((rope_apply_kernel<T, V, TR>), dim3(grid), dim3(256), 0, STREAM, rows, H, S, D, R, (const T*)x, sb, sh, ss, cos_t, sin_t, idx, table_rows, (T*)out, ob, oh, os)
    if (dtype == MI355_DT_BF16) {
        if (vec) { if (transpose) RL(bf16_t, 8, true); else RL(bf16_t, 8, false); }
        else { if (transpose) RL(bf16_t, 1, true); else RL(bf16_t, 1, false); }
    } else {
        if (vec) { if (transpose) RL(float, 4, true); else RL(float, 4, false); }
        else { if (transpose) RL(float, 1, true); else RL(float, 1, false); }
    }
#undef RL
    MI355_LAUNCH_CHECK("mi355_rope_apply");
    return 0;
}

extern "C" int mi355_dropout(int64_t n, const void* x, int x_dtype, const void* residual, void* y, int y_dtype, float p, uint64_t seed,
                             uint64_t offset, void* stream) {
    MI355_REQUIRE(n > 0 && x && y, "mi355_dropout: bad arguments");
    MI355_REQUIRE(p >= 0.f && p < 1.f, "mi355_dropout: p must be in [0, 1) (got %f)", (double)p);
    MI355_REQUIRE((x_dtype == MI355_DT_BF16 || x_dtype == MI355_DT_F32) && (y_dtype == MI355_DT_BF16 || y_dtype == MI355_DT_F32), "mi355_dropout: dtypes must be bf16 or fp32");
    const unsigned thresh = mi355_dropout_threshold(p);
    const float inv = 1.0f / (1.0f - p);
    const unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32), o0 = (unsigned)offset, o1 = (unsigned)(offset >> 32);
    const int grid = grid_for((n + 3) / 4, 256);
#define DL(TX, TY) hipLaunchKernelGGL((dropout_kernel<TX, TY>), dim3(grid), dim3(256), 0, STREAM, n, (const TX*)x, (const TY*)residual, (TY*)y, thresh, inv, k0, k1, o0, o1)
    if (x_dtype == MI355_DT_BF16 && y_dtype == MI355_DT_BF16) DL(bf16_t, bf16_t);
    else if (x_dtype == MI355_DT_BF16) DL(bf16_t, float);
    else if (y_dtype == MI355_DT_BF16) DL(float, bf16_t);
    else DL(float, float);
#undef DL
    MI355_LAUNCH_CHECK("mi355_dropout");
    return 0;
}
