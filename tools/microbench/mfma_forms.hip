// Issue rate of v_mfma_f32_32x32x16_bf16 on gfx950 by operand placement, one wave per SIMD (what the attention backward runs at).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_forms tools/microbench/mfma_forms.hip && /tmp/mfma_forms
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
#define CL "a128","a129","a130","a131","a132","a133","a134","a135","a136","a137","a138","a139","a140","a141","a142","a143","a144","a145","a146","a147","a148","a149","a150","a151","a152","a153","a154","a155","a156","a157","a158","a159","a160","a161","a162","a163","a164","a165","a166","a167","a168","a169","a170","a171","a172","a173","a174","a175","a176","a177","a178","a179","a180","a181","a182","a183","a184","a185","a186","a187","a188","a189","a190","a191","a192","a193","a194","a195","a196","a197","a198","a199","a200","a201","a202","a203","a204","a205","a206","a207","a208","a209","a210","a211","a212","a213","a214","a215","a216","a217","a218","a219","a220","a221","a222","a223","a224","a225","a226","a227","a228","a229","a230","a231","a232","a233","a234","a235","a236","a237","a238","a239","a240","a241","a242","a243","a244","a245","a246","a247","a248","a249","a250","a251","a252","a253","a254","a255"
template <int R0, bool NOP>
__device__ __forceinline__ void mfma_inplace(const bf16x8& a, const bf16x8& b) {
    if constexpr (NOP) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(a), "v"(b), "i"(R0), "i"(R0 + 15) : CL, "memory");
    else asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(a), "v"(b), "i"(R0), "i"(R0 + 15) : CL, "memory");
}
template <int R0>
__device__ __forceinline__ void mfma_vacc_aB(f32x16& acc, const bf16x8& a) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], %0" : "+v"(acc) : "v"(a), "i"(R0), "i"(R0 + 3) : CL, "memory");
}
template <int MODE>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* cyc, int reps, int nacc) {
    bf16x8 a, b, av[4], bv[4];
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x % 7 + i); b[i] = (__bf16)(float)(threadIdx.x % 5 - i); }
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 8; ++i) { av[j][i] = (__bf16)(float)(threadIdx.x % (7 + j) + i); bv[j][i] = (__bf16)(float)(threadIdx.x % (5 + j) - i); }
    asm volatile("" : "+v"(av[0]), "+v"(av[1]), "+v"(av[2]), "+v"(av[3]), "+v"(bv[0]), "+v"(bv[1]), "+v"(bv[2]), "+v"(bv[3]));
    f32x16 acc[8];
    for (int t = 0; t < 8; ++t) for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    if constexpr (MODE == 1 || MODE == 2 || MODE == 3 || MODE == 5)
        for (int r = 128; r < 256; ++r) {}
    asm volatile("s_nop 0" ::: CL);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
        if constexpr (MODE == 0) {
#pragma unroll
            for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
        } else if constexpr (MODE == 1) {
            mfma_inplace<128, true>(a, b); mfma_inplace<144, true>(a, b); mfma_inplace<160, true>(a, b); mfma_inplace<176, true>(a, b);
            mfma_inplace<192, true>(a, b); mfma_inplace<208, true>(a, b); mfma_inplace<224, true>(a, b); mfma_inplace<240, true>(a, b);
        } else if constexpr (MODE == 2) {
            mfma_inplace<128, false>(a, b); mfma_inplace<144, false>(a, b); mfma_inplace<160, false>(a, b); mfma_inplace<176, false>(a, b);
            mfma_inplace<192, false>(a, b); mfma_inplace<208, false>(a, b); mfma_inplace<224, false>(a, b); mfma_inplace<240, false>(a, b);
        } else if constexpr (MODE == 3) {  // two chains, VGPR accumulators, B in AGPRs (the A phases)
            mfma_vacc_aB<128>(acc[0], a); mfma_vacc_aB<132>(acc[1], a); mfma_vacc_aB<136>(acc[0], a); mfma_vacc_aB<140>(acc[1], a);
            mfma_vacc_aB<144>(acc[0], a); mfma_vacc_aB<148>(acc[1], a); mfma_vacc_aB<152>(acc[0], a); mfma_vacc_aB<156>(acc[1], a);
        } else if constexpr (MODE == 4) {  // 8 independent VGPR accumulators via asm, operands in VGPRs
#pragma unroll
            for (int t = 0; t < 8; ++t) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[t]) : "v"(a), "v"(b));
        } else if constexpr (MODE == 6) {  // B alternates between two tuples
            mfma_inplace<128, false>(a, bv[0]); mfma_inplace<144, false>(a, bv[1]); mfma_inplace<160, false>(a, bv[0]); mfma_inplace<176, false>(a, bv[1]);
            mfma_inplace<192, false>(a, bv[0]); mfma_inplace<208, false>(a, bv[1]); mfma_inplace<224, false>(a, bv[0]); mfma_inplace<240, false>(a, bv[1]);
        } else if constexpr (MODE == 7) {  // A rotates over four tuples
            mfma_inplace<128, false>(av[0], b); mfma_inplace<144, false>(av[1], b); mfma_inplace<160, false>(av[2], b); mfma_inplace<176, false>(av[3], b);
            mfma_inplace<192, false>(av[0], b); mfma_inplace<208, false>(av[1], b); mfma_inplace<224, false>(av[2], b); mfma_inplace<240, false>(av[3], b);
        } else if constexpr (MODE == 8) {  // both
            mfma_inplace<128, false>(av[0], bv[0]); mfma_inplace<144, false>(av[1], bv[1]); mfma_inplace<160, false>(av[2], bv[0]); mfma_inplace<176, false>(av[3], bv[1]);
            mfma_inplace<192, false>(av[0], bv[0]); mfma_inplace<208, false>(av[1], bv[1]); mfma_inplace<224, false>(av[2], bv[0]); mfma_inplace<240, false>(av[3], bv[1]);
        } else if constexpr (MODE == 9) {  // both, with the s_nop 1 and a counted lgkmcnt wait in front (the C phases of the attention backward)
            asm volatile("s_waitcnt lgkmcnt(12)"); mfma_inplace<128, true>(av[0], bv[0]); asm volatile("s_waitcnt lgkmcnt(12)"); mfma_inplace<144, true>(av[1], bv[1]);
            asm volatile("s_waitcnt lgkmcnt(12)"); mfma_inplace<160, true>(av[2], bv[0]); asm volatile("s_waitcnt lgkmcnt(12)"); mfma_inplace<176, true>(av[3], bv[1]);
            asm volatile("s_waitcnt lgkmcnt(12)"); mfma_inplace<192, true>(av[0], bv[0]); asm volatile("s_waitcnt lgkmcnt(12)"); mfma_inplace<208, true>(av[1], bv[1]);
            asm volatile("s_waitcnt lgkmcnt(12)"); mfma_inplace<224, true>(av[2], bv[0]); asm volatile("s_waitcnt lgkmcnt(12)"); mfma_inplace<240, true>(av[3], bv[1]);
        } else if constexpr (MODE == 5) {  // in place in AGPRs, only 4 tiles in rotation
            mfma_inplace<128, false>(a, b); mfma_inplace<144, false>(a, b); mfma_inplace<160, false>(a, b); mfma_inplace<176, false>(a, b);
            mfma_inplace<128, false>(a, b); mfma_inplace<144, false>(a, b); mfma_inplace<160, false>(a, b); mfma_inplace<176, false>(a, b);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int t = 0; t < 8; ++t) for (int e = 0; e < 16; ++e) s += acc[t][e];
    if (nacc == 12345) out[threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE>
void run(const char* name, int blocks) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 4096); hipMalloc(&cyc, 8);
    const int reps = 2000;
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, cyc, reps, 0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, cyc, reps, 0); hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-64s blocks %4d: %6.1f counter ticks / MFMA, %7.1f ns / MFMA by the event clock, %6.1f TFLOP/s\n", name, blocks, (double)c / (8.0 * reps), ms * 1e6 / (8.0 * reps),
           2.0 * 32 * 32 * 16 * 8.0 * reps * 4 * blocks / (ms * 1e-3) / 1e12);
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int blocks : {1, 256}) {
        run<0>("compiler-managed, 8 tiles", blocks);
        run<4>("asm, VGPR accumulators, 8 tiles", blocks);
        run<1>("asm, AGPR in place, 8 tiles, s_nop 1", blocks);
        run<2>("asm, AGPR in place, 8 tiles", blocks);
        run<5>("asm, AGPR in place, 4 tiles", blocks);
        run<3>("asm, 2 VGPR chains, B operand in AGPRs", blocks);
        run<6>("asm, AGPR in place, B alternates over 2 tuples", blocks);
        run<7>("asm, AGPR in place, A rotates over 4 tuples", blocks);
        run<8>("asm, AGPR in place, A rotates, B alternates", blocks);
        run<9>("asm, AGPR in place, A rotates, B alternates, s_nop + lgkmcnt", blocks);
    }
    return 0;
}
