// Microbenchmark behind DESIGN section 5's "what does a GEMM write-out cost": how fast a CU moves 16-byte stores, alone and with the whole
// chip storing, and whether stores trickled between the MFMAs (and LDS-DMA pieces) of a main loop are free.  Stand-alone:
//   hipcc -O3 --offload-arch=gfx950 -o store_rate store_rate.hip && ./store_rate
// One workgroup = 4 waves (one per SIMD, 128 KiB of LDS so that a CU holds exactly one), a "tile" = 128 KiB of output per workgroup = 32 stores of
// 1 KiB per wave; between two stores a wave issues MPS 16x16x32 bf16 MFMAs (64 per K-slice in the 4-wave GEMM loop) and, with LOADS, 8 LDS-DMA
// pieces of 1 KiB per 64 MFMAs (that loop's operand stream, here from an L2-resident buffer).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MPS, bool STORE, bool LOADS, bool BURST = false>
__global__ __launch_bounds__(256, 1) void k(char* __restrict__ out, const char* __restrict__ in, int tiles, float* sink) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {8, 7, 6, 5, 4, 3, 2, 1};
    a[0] += (short)lane;
    u32x4 payload = {(unsigned)lane, (unsigned)wave, (unsigned)blockIdx.x, 7u};
    auto rsrc_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(in), 0, 0x7fffffff, 0x00020000);
    auto rsrc_out = __builtin_amdgcn_make_buffer_rsrc(out, 0, 0x7fffffff, 0x00020000);
    int mf = 0;
    for (int t = 0; t < tiles; ++t) {
        const unsigned tile_off = (unsigned)(((size_t)blockIdx.x * tiles + t) * 131072u + wave * 32768u);  // < 4 GiB by construction (host checks)
#pragma unroll 1
        for (int s = 0; s < 32; ++s) {
#pragma unroll
            for (int m = 0; m < MPS; ++m) {
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[m & 7]) : "v"(a), "v"(b));
                if (LOADS && (m & 7) == 7) {  // 8 pieces per 64 MFMAs
                    const int piece = (mf++) & 127;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_in, LDS_PTR(smem + wave * 32768 + (piece & 31) * 1024), 16,
                                                             (unsigned)((blockIdx.x & 63) * 131072 + piece * 1024 + lane * 16), 0, 0, 0);
                }
            }
            if (STORE && !BURST) __builtin_amdgcn_raw_buffer_store_b128(payload, rsrc_out, tile_off + s * 1024 + lane * 16, 0, 0);
            if (LOADS) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");  // the ring never drains, but does not run away either
        }
        if (STORE && BURST) {  // what the GEMM does today: the whole tile at the end of its main loop
#pragma unroll
            for (int s = 0; s < 32; ++s) __builtin_amdgcn_raw_buffer_store_b128(payload, rsrc_out, tile_off + s * 1024 + lane * 16, 0, 0);
        }
    }
    float keep = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) keep += acc[i][0] + acc[i][3];
    if (keep == 1234.5f) sink[0] = keep + smem[lane];
}

template <int MPS, bool STORE, bool LOADS, bool BURST = false>
static void run(const char* name, int grid, int tiles, char* out, const char* in, float* sink) {
    if ((size_t)grid * tiles * 131072u >= (1ull << 32)) { fprintf(stderr, "grid * tiles too large for 32-bit offsets\n"); exit(1); }
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    auto kern = k<MPS, STORE, LOADS, BURST>;
    CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 131072, 0, out, in, tiles, sink);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    const double us = best * 1e3, per_tile = us / tiles;
    const double bytes = STORE ? (double)grid * tiles * 131072.0 : 0.0;
    const double mfma = (double)grid * tiles * 4 * 32 * MPS * 16384.0;
    printf("%-44s grid %4d tiles %3d  %9.1f us  %7.2f us/tile  store %7.1f GB/s (%5.1f B/us/CU /1000)  mfma %7.1f TFLOP/s\n", name, grid, tiles, us, per_tile,
           bytes / us * 1e-3, grid ? bytes / us / grid * 1e-3 : 0.0, mfma / us * 1e-6);
    fflush(stdout);
}

int main() {
    char *out, *in;
    float* sink;
    CHECK(hipMalloc(&out, (size_t)3 << 30));
    CHECK(hipMalloc(&in, 64 * 131072));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(in, 1, 64 * 131072));
    for (int grid : {1, 8, 64, 256}) {
        const int tiles = 32;
        run<0, true, false>("stores only", grid, tiles, out, in, sink);
        run<64, false, false>("MFMAs only (64 per slot)", grid, tiles, out, in, sink);
        run<64, true, false>("64 MFMAs per store", grid, tiles, out, in, sink);
        run<32, true, false>("32 MFMAs per store", grid, tiles, out, in, sink);
        run<16, true, false>("16 MFMAs per store", grid, tiles, out, in, sink);
        run<64, false, true>("64 MFMAs + 8 DMA pieces", grid, tiles, out, in, sink);
        run<64, true, true>("64 MFMAs + 8 DMA pieces + 1 store", grid, tiles, out, in, sink);
        run<32, true, true>("32 MFMAs + 4 DMA pieces + 1 store", grid, tiles, out, in, sink);
        run<64, true, true, true>("64 MFMAs + 8 DMA pieces, 32 stores at the end", grid, tiles, out, in, sink);
        run<64, true, false, true>("64 MFMAs, 32 stores at the end", grid, tiles, out, in, sink);
        run<0, true, true>("stores only (LOADS flag, no MFMAs)", grid, tiles, out, in, sink);
    }
    return 0;
}
