// Issue cost of INDEPENDENT fillers in a v_mfma_f32_32x32x16_bf16 gap, one wave per SIMD (gfx950).  Cycles per gap.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
__device__ __forceinline__ void mfma_vacc(f32x16& acc, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "memory");
}
// NF plain v_fma (independent, 8 rotating destinations), NE v_exp, NC v_cvt_pk, NL ds_read_b128, NT ds_read_b64_tr_b16, W: counted wait, DEP: fma chain on ONE register
template <int NF, int NE, int NC, int NL, int NT, int W, int DEP>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* cyc, int reps, float x0) {
    __shared__ __attribute__((aligned(16))) char smem[65536];
    for (int i = threadIdx.x; i < 16384; i += 256) reinterpret_cast<float*>(smem)[i] = (float)i;
    __syncthreads();
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x % 7 + i); b[i] = (__bf16)(float)(threadIdx.x % 5 - i); }
    asm volatile("" : "+v"(a), "+v"(b));
    f32x16 acc0, acc1;
    for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
    float r[8], x = x0 * threadIdx.x, y = 0.5f;
    for (int i = 0; i < 8; ++i) r[i] = x0 * i;
    unsigned w[4] = {0, 0, 0, 0};
    f32x4 frag[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    unsigned long long t2[2] = {0, 0};
    const unsigned lds = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem + (threadIdx.x & 63) * 16;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int rr = 0; rr < reps; ++rr) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            if constexpr (W) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(W) : "memory");
            if (g & 1) mfma_vacc(acc1, a, b); else mfma_vacc(acc0, a, b);
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                if constexpr (DEP) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[0]) : "v"(x), "v"(y));
                else asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r[i % 8]) : "v"(x), "v"(y), "v"(r[(i + 4) % 8]));
            }
#pragma unroll
            for (int i = 0; i < NE; ++i) asm volatile("v_exp_f32 %0, %1" : "=v"(r[(i + 5) % 8]) : "v"(x));
#pragma unroll
            for (int i = 0; i < NC; ++i) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w[i % 4]) : "v"(x), "v"(y));
#pragma unroll
            for (int i = 0; i < NL; ++i) asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(frag[i % 2]) : "v"(lds), "i"(((g + i) & 7) * 1024) : "memory");
#pragma unroll
            for (int i = 0; i < NT; ++i) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%c2" : "=v"(t2[i % 2]) : "v"(lds), "i"(((g + i) & 7) * 1024) : "memory");
        }
        if constexpr (NL + NT > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = frag[0][0] + frag[1][1] + acc0[0] + acc1[3] + (float)(w[0] + w[1] + w[2] + w[3]) + (float)(t2[0] + t2[1]);
    for (int i = 0; i < 8; ++i) s += r[i];
    if (reps == 12345) out[threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NF, int NE, int NC, int NL, int NT, int W, int DEP>
void run() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 4096); hipMalloc(&cyc, 8);
    const int reps = 500;
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<NF, NE, NC, NL, NT, W, DEP>), dim3(256), dim3(256), 0, 0, out, cyc, reps, 0.001f);
    hipDeviceSynchronize();
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("fma %d%s exp %d cvt_pk %d ds_read_b128 %d ds_read_b64_tr %d wait %d: %6.1f cycles / gap\n", NF, DEP ? " (one chain)" : "", NE, NC, NL, NT, W, (double)c / (16.0 * reps));
    hipFree(out); hipFree(cyc);
}
int main() {
    run<0, 0, 0, 0, 0, 0, 0>(); run<2, 0, 0, 0, 0, 0, 0>(); run<4, 0, 0, 0, 0, 0, 0>(); run<5, 0, 0, 0, 0, 0, 0>(); run<6, 0, 0, 0, 0, 0, 0>(); run<7, 0, 0, 0, 0, 0, 0>(); run<8, 0, 0, 0, 0, 0, 0>();
    run<4, 0, 0, 0, 0, 0, 1>(); run<6, 0, 0, 0, 0, 0, 1>();
    run<0, 1, 0, 0, 0, 0, 0>(); run<0, 2, 0, 0, 0, 0, 0>(); run<0, 3, 0, 0, 0, 0, 0>();
    run<0, 0, 2, 0, 0, 0, 0>(); run<0, 0, 4, 0, 0, 0, 0>(); run<0, 0, 6, 0, 0, 0, 0>();
    run<4, 1, 2, 0, 0, 0, 0>(); run<4, 1, 0, 0, 0, 0, 0>(); run<3, 1, 1, 0, 0, 0, 0>();
    run<0, 0, 0, 1, 0, 0, 0>(); run<0, 0, 0, 2, 0, 0, 0>(); run<0, 0, 0, 0, 2, 0, 0>(); run<0, 0, 0, 1, 0, 6, 0>();
    run<4, 1, 2, 1, 0, 0, 0>(); run<4, 1, 2, 1, 0, 6, 0>(); run<4, 1, 2, 0, 2, 0, 0>(); run<4, 1, 2, 0, 2, 12, 0>();
    run<3, 1, 1, 1, 0, 6, 0>(); run<3, 1, 1, 0, 2, 12, 0>(); run<2, 1, 1, 0, 2, 12, 0>(); run<2, 0, 1, 0, 2, 12, 0>();
    return 0;
}
