// Input pipeline on the step's left edge: the per-sample work of MultimodalDataset
// (llm_quest/dataset.py:295-383) as HBM-bound byte kernels -- Pillow's two-pass fixed-point bilinear resize
// (transforms.Resize on a PIL image), ToTensor + Normalize, and the pad / truncate / mask of the tokenised caption.
// Integer arithmetic end to end up to the final float conversion: bit-exact against Pillow 12.2 (tests/test_pipeline_*.py).
#include "common.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;  // Pillow Resample.c

__device__ __forceinline__ int clip8(int acc) {
    const int v = acc >> PRECISION_BITS;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// horizontal pass: dst[y, xx, c] = clip8(2^21 + sum_x kk[xx, x] * src[y, x0 + x, c]); a thread owns one output pixel (C <= 4)
__global__ __launch_bounds__(256) void resize_h_kernel(int H, int W_out, int C, const uint8_t* __restrict__ src, int64_t src_pitch,
                                                       const int32_t* __restrict__ bounds, const int32_t* __restrict__ kk, int ksize,
                                                       uint8_t* __restrict__ dst) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)H * W_out) return;
    const int y = (int)(idx / W_out), xx = (int)(idx % W_out);
    const int x0 = bounds[2 * xx], n = bounds[2 * xx + 1];
    const int32_t* k = kk + (int64_t)xx * ksize;
    const uint8_t* row = src + y * src_pitch + (int64_t)x0 * C;
    int acc[4] = {1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1)};
    for (int x = 0; x < n; ++x) {
        const int w = k[x];
        for (int c = 0; c < C; ++c) acc[c] += (int)row[x * C + c] * w;
    }
    for (int c = 0; c < C; ++c) dst[idx * C + c] = (uint8_t)clip8(acc[c]);
}

// vertical pass fused with ToTensor (+ Normalize): dst[c, yy, xx] = ((clip8(...) / 255) - mean[c]) / std[c]   fp32, CHW
__global__ __launch_bounds__(256) void resize_v_normalize_kernel(int H_out, int W, int C, const uint8_t* __restrict__ src,
                                                                 const int32_t* __restrict__ bounds, const int32_t* __restrict__ kk, int ksize,
                                                                 const float* __restrict__ mean, const float* __restrict__ stdv,
                                                                 float* __restrict__ dst) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)H_out * W) return;
    const int yy = (int)(idx / W), xx = (int)(idx % W);
    const int y0 = bounds[2 * yy], n = bounds[2 * yy + 1];
    const int32_t* k = kk + (int64_t)yy * ksize;
    int acc[4] = {1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1)};
    for (int y = 0; y < n; ++y) {
        const uint8_t* p = src + ((int64_t)(y0 + y) * W + xx) * C;
        const int w = k[y];
        for (int c = 0; c < C; ++c) acc[c] += (int)p[c] * w;
    }
    for (int c = 0; c < C; ++c) {
        float v = (float)clip8(acc[c]) / 255.0f;
        if (mean) v = (v - mean[c]) / stdv[c];
        dst[((int64_t)c * H_out + yy) * W + xx] = v;
    }
}

// ids_out[b, i] = i < min(len_b, L) ? flat[offsets[b] + i] : pad;  mask_out = (i < min(len_b, L))
__global__ void pad_tokens_kernel(int B, int L, const int64_t* __restrict__ flat, const int64_t* __restrict__ offsets, int64_t pad_id,
                                  int64_t* __restrict__ ids_out, uint8_t* __restrict__ mask_out) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)B * L) return;
    const int b = (int)(idx / L), i = (int)(idx % L);
    const int64_t o = offsets[b], len = offsets[b + 1] - o;
    const bool real = i < len;
    ids_out[idx] = real ? flat[o + i] : pad_id;
    mask_out[idx] = real ? 1 : 0;
}

}  // namespace

#define ST(s) ((hipStream_t)(s))

extern "C" int mi355_resize_h_u8(int H, int W_in, int W_out, int C, const uint8_t* src, int64_t src_pitch, const int32_t* bounds, const int32_t* kk,
                                 int ksize, uint8_t* dst, void* stream) {
    MI355_REQUIRE(H > 0 && W_in > 0 && W_out > 0 && C >= 1 && C <= 4 && ksize >= 1, "resize_h_u8: bad sizes (1..4 channels)");
    MI355_REQUIRE(src && bounds && kk && dst && src_pitch >= (int64_t)W_in * C, "resize_h_u8: null pointer or pitch smaller than a row");
    resize_h_kernel<<<(int)(((int64_t)H * W_out + 255) / 256), 256, 0, ST(stream)>>>(H, W_out, C, src, src_pitch, bounds, kk, ksize, dst);
    MI355_LAUNCH_CHECK("resize_h_u8");
    return 0;
}

extern "C" int mi355_resize_v_normalize(int H_in, int H_out, int W, int C, const uint8_t* src, const int32_t* bounds, const int32_t* kk, int ksize,
                                        const float* mean, const float* stdv, float* dst, void* stream) {
    MI355_REQUIRE(H_in > 0 && H_out > 0 && W > 0 && C >= 1 && C <= 4 && ksize >= 1, "resize_v_normalize: bad sizes (1..4 channels)");
    MI355_REQUIRE(src && bounds && kk && dst && ((mean == nullptr) == (stdv == nullptr)), "resize_v_normalize: null pointer (mean and std come together)");
    resize_v_normalize_kernel<<<(int)(((int64_t)H_out * W + 255) / 256), 256, 0, ST(stream)>>>(H_out, W, C, src, bounds, kk, ksize, mean, stdv, dst);
    MI355_LAUNCH_CHECK("resize_v_normalize");
    return 0;
}

extern "C" int mi355_pad_tokens(int B, int L, const int64_t* flat_ids, const int64_t* offsets, int64_t pad_id, int64_t* ids_out, uint8_t* mask_out,
                                void* stream) {
    MI355_REQUIRE(B > 0 && L > 0 && flat_ids && offsets && ids_out && mask_out, "pad_tokens: bad arguments");
    pad_tokens_kernel<<<(int)(((int64_t)B * L + 255) / 256), 256, 0, ST(stream)>>>(B, L, flat_ids, offsets, pad_id, ids_out, mask_out);
    MI355_LAUNCH_CHECK("pad_tokens");
    return 0;
}
