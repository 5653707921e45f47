// Measurement aid, not part of the training path: what the matrix pipe ALONE sustains on this board -- v_mfma_f32_32x32x16_bf16 on every SIMD of the launch's workgroups, random bf16
// operands held in registers, no memory traffic.  On random data the board's power cap, not the 2.4-GHz slot count behind the 2.5-PFLOP/s dense peak, sets this rate
// (tools/microbench/mfma_power.hip is the stand-alone form with the all-zero and LDS-fed variants).  bench.py times it beside the step so that the
// bench line carries the ceiling of the box it ran on.
#include "common.h"

namespace {

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

__device__ __forceinline__ unsigned mix(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// a bf16 pair with random sign and mantissa, exponents 2^-3 .. 2^0
__device__ __forceinline__ unsigned rnd_pair(unsigned s) {
    const unsigned r = mix(s);
    const unsigned lo = (r & 0x807fu) | ((124u + ((r >> 8) & 3u)) << 7);
    const unsigned hi = ((r >> 16) & 0x807fu) | ((124u + ((r >> 24) & 3u)) << 7);
    return lo | (hi << 16);
}

__global__ __launch_bounds__(256, 1) void mfma_pipe_kernel(float* __restrict__ out, int reps) {
    u32x4 fr[8];
    for (int j = 0; j < 8; ++j)
        for (int e = 0; e < 4; ++e) fr[j][e] = rnd_pair((blockIdx.x * 256 + threadIdx.x) * 64 + j * 4 + e);
    f32x16 acc[16];
    for (int t = 0; t < 16; ++t)
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    for (int r = 0; r < reps; ++r) {  // one K step of a 128 x 128 wave tile: sixteen products on four + four fragments
        asm volatile("" : "+v"(fr[0]), "+v"(fr[1]), "+v"(fr[2]), "+v"(fr[3]), "+v"(fr[4]), "+v"(fr[5]), "+v"(fr[6]), "+v"(fr[7]));
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fr[i]), __builtin_bit_cast(bf16x8, fr[4 + j]), acc[i * 4 + j], 0, 0, 0);
    }
    float s = 0.f;
    for (int t = 0; t < 16; ++t)
        for (int e = 0; e < 16; ++e) s += acc[t][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// The shape the GEMMs issue (gemm.hip: v_mfma_f32_16x16x32_bf16 on a 128 x 64 wave tile): one K step = 32 products on eight + four fragments, 32 independent
// accumulators of four registers.  Round 5's attempt at this loop came out issue-bound as compiled (63 % of the slots: 64 accumulators, the fragments re-materialised);
// here the twelve fragments stay pinned across the loop and a repetition is exactly 32 back-to-back MFMAs.
__global__ __launch_bounds__(256, 1) void mfma_pipe16_kernel(float* __restrict__ out, int reps) {
    u32x4 fr[12];
    for (int j = 0; j < 12; ++j)
        for (int e = 0; e < 4; ++e) fr[j][e] = rnd_pair((blockIdx.x * 256 + threadIdx.x) * 64 + j * 4 + e);
    f32x4 acc[8][4];
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < reps; ++r) {
        asm volatile("" : "+v"(fr[0]), "+v"(fr[1]), "+v"(fr[2]), "+v"(fr[3]), "+v"(fr[4]), "+v"(fr[5]), "+v"(fr[6]), "+v"(fr[7]), "+v"(fr[8]), "+v"(fr[9]), "+v"(fr[10]), "+v"(fr[11]));
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(fr[8 + j]), "v"(fr[i]));  // in place (the intrinsic form compiles to an accumulator copy per product)
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

}  // namespace

extern "C" int mi355_mfma_pipe_probe(int blocks, int reps, float* out, void* stream) {
    MI355_REQUIRE(blocks > 0 && blocks <= 65535 && reps != 0 && out, "mfma_pipe_probe: blocks 1..65535, reps != 0, out = blocks * 256 floats");
    if (reps < 0) {  // the 16x16x32 shape: -reps repetitions of 32 products
        hipLaunchKernelGGL(mfma_pipe16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, -reps);
        MI355_LAUNCH_CHECK("mfma_pipe_probe(16x16x32)");
        return 0;
    }
    hipLaunchKernelGGL(mfma_pipe_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, reps);
    MI355_LAUNCH_CHECK("mfma_pipe_probe");
    return 0;
}
