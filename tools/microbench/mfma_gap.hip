// What one wave per SIMD pays per v_mfma_f32_32x32x16_bf16 gap for the fillers of the attention backward (gfx950).
// Each variant is a loop of 16 gaps; the fillers are asm statements in a fixed order.  Prints cycles per gap.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
#define CL "a128","a129","a130","a131","a132","a133","a134","a135","a136","a137","a138","a139","a140","a141","a142","a143","a144","a145","a146","a147","a148","a149","a150","a151","a152","a153","a154","a155","a156","a157","a158","a159","a160","a161","a162","a163","a164","a165","a166","a167","a168","a169","a170","a171","a172","a173","a174","a175","a176","a177","a178","a179","a180","a181","a182","a183","a184","a185","a186","a187","a188","a189","a190","a191","a192","a193","a194","a195","a196","a197","a198","a199","a200","a201","a202","a203","a204","a205","a206","a207","a208","a209","a210","a211","a212","a213","a214","a215","a216","a217","a218","a219","a220","a221","a222","a223","a224","a225","a226","a227","a228","a229","a230","a231","a232","a233","a234","a235","a236","a237","a238","a239","a240","a241","a242","a243","a244","a245","a246","a247","a248","a249","a250","a251","a252","a253","a254","a255"
template <int R0>
__device__ __forceinline__ void mfma_inplace(const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(a), "v"(b), "i"(R0), "i"(R0 + 15) : CL, "memory");
}
__device__ __forceinline__ void mfma_vacc(f32x16& acc, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "memory");
}
// bits of F: 1 = v_mul + v_fmac, 2 = v_exp, 4 = v_sub + v_mul, 8 = two v_cvt_pk, 16 = one ds_read_b128, 32 = s_nop 1, 64 = second exp chain (2 elements / gap),
// 128 = VGPR accumulators (two chains) instead of in-place AGPR tiles, 256 = two ds_read_b64_tr_b16 instead of the b128, 512 = counted lgkmcnt wait
template <int F>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* cyc, int reps, float x0) {
    __shared__ __attribute__((aligned(16))) char smem[65536];
    for (int i = threadIdx.x; i < 16384; i += 256) reinterpret_cast<float*>(smem)[i] = (float)i;
    __syncthreads();
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x % 7 + i); b[i] = (__bf16)(float)(threadIdx.x % 5 - i); }
    asm volatile("" : "+v"(a), "+v"(b));
    f32x16 acc0, acc1;
    for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
    float x = x0 * threadIdx.x, y = 0.5f, c = 1.25f, l = 0.75f, p = 0.f, d = 0.f, p2 = 0.f, d2 = 0.f;
    unsigned w0 = 0, w1 = 0;
    f32x4 frag = {0, 0, 0, 0};
    const unsigned lds = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem + (threadIdx.x & 63) * 16;
    asm volatile("s_nop 0" ::: CL);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            if constexpr (F & 512) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
            if constexpr (F & 32) asm volatile("s_nop 1");
            if constexpr (F & 128) { if (g & 1) mfma_vacc(acc1, a, b); else mfma_vacc(acc0, a, b); }
            else {
                switch (g & 7) {
                    case 0: mfma_inplace<128>(a, b); break; case 1: mfma_inplace<144>(a, b); break; case 2: mfma_inplace<160>(a, b); break; case 3: mfma_inplace<176>(a, b); break;
                    case 4: mfma_inplace<192>(a, b); break; case 5: mfma_inplace<208>(a, b); break; case 6: mfma_inplace<224>(a, b); break; default: mfma_inplace<240>(a, b); break;
                }
            }
            if constexpr (F & 4) { asm volatile("v_sub_f32 %0, %1, %2" : "=v"(d) : "v"(y), "v"(c)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(d) : "v"(p)); }
            if constexpr (F & 8) { asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w0) : "v"(p), "v"(p2)); asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w1) : "v"(d), "v"(d2)); }
            if constexpr (F & 1) { asm volatile("v_mul_f32 %0, 0xbfb8aa3b, %1" : "=v"(p) : "v"(l)); asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(p) : "v"(x), "v"(c)); }
            if constexpr (F & 2) asm volatile("v_exp_f32 %0, %0" : "+v"(p));
            if constexpr (F & 64) {
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(d2) : "v"(y), "v"(c)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(d2) : "v"(p2));
                asm volatile("v_mul_f32 %0, 0xbfb8aa3b, %1" : "=v"(p2) : "v"(l)); asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(p2) : "v"(x), "v"(c));
                asm volatile("v_exp_f32 %0, %0" : "+v"(p2));
            }
            if constexpr (F & 16) asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(frag) : "v"(lds), "i"((g & 7) * 1024) : "memory");
            if constexpr (F & 256) {
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%c2" : "=v"(w0) , "=v"(w1) : "v"(lds), "i"((g & 7) * 1024) : "memory");
            }
        }
        if constexpr (F & (16 | 256)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = p + d + p2 + d2 + frag[0] + acc0[0] + acc1[3] + (float)(w0 + w1);
    if (reps == 12345) out[threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int F>
void run(const char* name) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 4096); hipMalloc(&cyc, 8);
    const int reps = 500;
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<F>, dim3(256), dim3(256), 0, 0, out, cyc, reps, 0.001f);
    hipDeviceSynchronize();
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-86s %6.1f cycles / gap\n", name, (double)c / (16.0 * reps));
    hipFree(out); hipFree(cyc);
}
int main() {
    run<0>("bare in-place MFMAs");
    run<128>("bare MFMAs, two VGPR accumulator chains");
    run<1>("+ mul, fmac");
    run<1 | 2>("+ mul, fmac, exp");
    run<1 | 2 | 4>("+ mul, fmac, exp, sub, mul");
    run<1 | 2 | 4 | 8>("+ mul, fmac, exp, sub, mul, 2 cvt_pk  (7 VALU)");
    run<1 | 2 | 4 | 8 | 16>("+ 7 VALU + ds_read_b128");
    run<1 | 2 | 4 | 8 | 16 | 512>("+ 7 VALU + ds_read_b128 + lgkmcnt(6)");
    run<1 | 2 | 4 | 8 | 16 | 512 | 128>("+ 7 VALU + ds_read_b128 + lgkmcnt(6), VGPR accumulator chains (an A(1) gap)");
    run<1 | 4 | 8 | 16 | 512 | 128>("  the same without the exp");
    run<1 | 2 | 4 | 16 | 512 | 128>("  the same without the cvt_pk");
    run<16 | 512 | 128>("  ds_read_b128 + lgkmcnt(6) only, VGPR accumulator chains");
    run<1 | 2 | 4 | 8 | 32>("+ 7 VALU + s_nop 1");
    run<1 | 2 | 4 | 8 | 64>("+ 12 VALU (two elements)");
    return 0;
}
