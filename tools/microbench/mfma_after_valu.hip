// Does an MFMA run slower when its B operand was produced by VALU (v_exp / v_cvt_pk_bf16_f32) earlier in the same loop trip?
// One wave per SIMD, in-place AGPR accumulators (the C phases of the attention backward).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned u32x4;
#define CL "a128","a129","a130","a131","a132","a133","a134","a135","a136","a137","a138","a139","a140","a141","a142","a143","a144","a145","a146","a147","a148","a149","a150","a151","a152","a153","a154","a155","a156","a157","a158","a159","a160","a161","a162","a163","a164","a165","a166","a167","a168","a169","a170","a171","a172","a173","a174","a175","a176","a177","a178","a179","a180","a181","a182","a183","a184","a185","a186","a187","a188","a189","a190","a191","a192","a193","a194","a195","a196","a197","a198","a199","a200","a201","a202","a203","a204","a205","a206","a207","a208","a209","a210","a211","a212","a213","a214","a215","a216","a217","a218","a219","a220","a221","a222","a223","a224","a225","a226","a227","a228","a229","a230","a231","a232","a233","a234","a235","a236","a237","a238","a239","a240","a241","a242","a243","a244","a245","a246","a247","a248","a249","a250","a251","a252","a253","a254","a255"
template <int R0>
__device__ __forceinline__ void mfma_inplace(const bf16x8& a, const bf16x8& b) {
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(a), "v"(b), "i"(R0), "i"(R0 + 15) : CL, "memory");
}
__device__ __forceinline__ unsigned pack(float lo, float hi) {
    unsigned r;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
// MODE 0: B words recomputed every trip (exp + pack), MFMAs use them.  MODE 1: the same VALU work, MFMAs use loop-invariant words.
// MODE 2: no VALU work, loop-invariant words.  MODE 3: like 0 but NaN-free small data (x = 0).
template <int MODE>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* cyc, int reps, float x0) {
    bf16x8 av[4];
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 8; ++i) av[j][i] = (__bf16)(float)(threadIdx.x % (7 + j) + i);
    asm volatile("" : "+v"(av[0]), "+v"(av[1]), "+v"(av[2]), "+v"(av[3]));
    u32x4 fixed[2] = {{0x3f803f80u, 0x3f003f80u, 0x3e803f80u, 0x3f803e00u}, {0x3f803f00u, 0x3f003f00u, 0x3e803f00u, 0x3f803e80u}};
    asm volatile("" : "+v"(fixed[0]), "+v"(fixed[1]));
    float x[16];
    for (int e = 0; e < 16; ++e) x[e] = x0 * (float)(threadIdx.x + e);
    unsigned long long t_m = 0, t_all0 = __builtin_readcyclecounter();
    asm volatile("s_nop 0" ::: CL);
    for (int r = 0; r < reps; ++r) {
        u32x4 w[2];
        if constexpr (MODE != 2) {
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
                float p0, p1;
                asm volatile("v_exp_f32 %0, %1" : "=v"(p0) : "v"(x[e]));
                asm volatile("v_exp_f32 %0, %1" : "=v"(p1) : "v"(x[e + 1]));
                w[e / 8][(e % 8) / 2] = pack(p0, p1);
                x[e] += 1e-6f;
            }
        }
        const unsigned long long t0 = __builtin_readcyclecounter();
        const bf16x8 b0 = __builtin_bit_cast(bf16x8, MODE == 0 || MODE == 3 ? w[0] : fixed[0]), b1 = __builtin_bit_cast(bf16x8, MODE == 0 || MODE == 3 ? w[1] : fixed[1]);
        if constexpr (MODE == 1) asm volatile("" ::"v"(w[0]), "v"(w[1]));
        mfma_inplace<128>(av[0], b0); mfma_inplace<144>(av[1], b1); mfma_inplace<160>(av[2], b0); mfma_inplace<176>(av[3], b1);
        mfma_inplace<192>(av[0], b0); mfma_inplace<208>(av[1], b1); mfma_inplace<224>(av[2], b0); mfma_inplace<240>(av[3], b1);
        mfma_inplace<128>(av[0], b0); mfma_inplace<144>(av[1], b1); mfma_inplace<160>(av[2], b0); mfma_inplace<176>(av[3], b1);
        mfma_inplace<192>(av[0], b0); mfma_inplace<208>(av[1], b1); mfma_inplace<224>(av[2], b0); mfma_inplace<240>(av[3], b1);
        t_m += __builtin_readcyclecounter() - t0;
    }
    const unsigned long long t_all = __builtin_readcyclecounter() - t_all0;
    float s = 0.f;
    for (int e = 0; e < 16; ++e) s += x[e];
    if (reps == 12345) out[threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t_m; cyc[1] = t_all; }
}
template <int MODE>
void run(const char* name, float x0) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 4096); hipMalloc(&cyc, 16);
    const int reps = 2000;
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, out, cyc, reps, x0);
    hipDeviceSynchronize();
    unsigned long long c[2]; hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
    printf("%-72s %6.1f cycles / MFMA inside the MFMA segment, %7.1f cycles / trip\n", name, (double)c[0] / (16.0 * reps), (double)c[1] / reps);
    hipFree(out); hipFree(cyc);
}
int main() {
    run<2>("no VALU, loop-invariant B words", 0.01f);
    run<1>("exp + pack every trip, MFMAs use loop-invariant B words", 0.01f);
    run<0>("exp + pack every trip, MFMAs use those words (finite data)", -0.001f);
    run<0>("exp + pack every trip, MFMAs use those words (x grows: inf)", 0.5f);
    run<3>("exp + pack every trip, MFMAs use those words (x = 0)", 0.f);
    return 0;
}
