// bf16 MFMA GEMM for gfx950 with fused epilogue (bias / exact GELU / residual-or-accumulate).
//
// One kernel template covers the three operand forms a Linear layer needs (include/mi355_vlm.h):
//   NT  y  = x W^T      both operands K-contiguous            -> ds_read_b128 fragments
//   NN  dx = dy W       B is K-strided ([K][N] row-major)     -> ds_read_b64_tr_b16 fragments for B
//   TN  dW = dy^T x     A and B K-strided                     -> transposing reads for both
// so no operand is ever transposed in HBM.
//
// Structure (per 256-thread workgroup = 4 waves as 2x2, 128x128 output tile, K-step 64):
//   * HBM -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds, 1 KiB per wave-instruction), two LDS stages, the
//     next K-tile's DMA in flight while the current one feeds the MFMAs; one barrier per K-tile.
//   * The DMA destination is lane-linear, so the bank-conflict swizzle is applied on the per-lane SOURCE
//     address and again on the fragment read (both-sides rule): row-major-K tiles use 128-B rows with
//     chunk' = chunk ^ ((row>>1)&7); K-strided tiles use 256-B rows with chunk' = chunk ^ (f(k)<<1).
//   * Out-of-range rows/cols/K are zero-filled by the buffer range check (offset 0x80000000 > num_records),
//     so any M and any K,N multiple of 8 work without a tail path in the main loop.
//   * Every wave owns a 64x64 sub-tile = 4x4 mfma_f32_16x16x32_bf16 accumulators (64 VGPRs).
//   * Epilogue: accumulators -> LDS (fp32) -> row-contiguous 16-B global stores with bias/GELU/residual fused.
//   * Workgroup -> tile map: XCD-aware (blocks b and b+8 share an L2) then 8-row super-groups.
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int NTHREADS = 256;
constexpr int TILE_BYTES = 128 * 64 * 2;        // one operand tile, either orientation: 16 KiB
constexpr int STAGE_BYTES = 2 * TILE_BYTES;     // A + B
constexpr int EPI_LD = 68;                      // fp32 row pitch of the epilogue staging (bank-spread, 16-B aligned)
constexpr int SMEM_BYTES = 4 * 64 * EPI_LD * 4; // 69632 >= 2 stages (65536)
constexpr unsigned OOB = 0x80000000u;           // beyond num_records (0x7fffffff): load returns zeros

struct GemmParams {
    const bf16_t* A;
    const bf16_t* B;
    void* C;
    const float* bias;
    const void* R;
    int64_t M, N, K, lda, ldb, ldc, ldr;
    int tiles_m, tiles_n, epilogue;
};

__device__ __forceinline__ float gelu_erf(float x) { return x * 0.5f * (1.0f + erff(x * 0.70710678118654752440f)); }

__device__ __forceinline__ int tr_f(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }

// Per-lane byte offsets (relative to the tile's (row0,k0) corner) of this wave's 4 DMA pieces of one operand tile,
// plus the K-extent each piece needs for validity.  Non-TR: tile [128 rows][64 k], piece = 8 rows.
template <bool TR>
__device__ __forceinline__ void piece_offsets(int wave, int lane, int64_t ld, int64_t rows_left, unsigned (&voff)[4],
                                              int (&kneed)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int pi = wave * 4 + j;
        if constexpr (!TR) {
            const int r = 8 * pi + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            voff[j] = (r < rows_left) ? (unsigned)(r * ld * 2 + c * 16) : OOB;
            kneed[j] = c * 8;  // valid iff kneed < K - k0
        } else {
            const int kr = 4 * pi + (lane >> 4);
            const int c = (lane & 15) ^ (tr_f(kr) << 1);
            voff[j] = (c * 8 < rows_left) ? (unsigned)(kr * ld * 2 + c * 16) : OOB;  // rows_left = cols left here
            kneed[j] = kr;
        }
    }
}

__device__ __forceinline__ void dma_piece(const void* base, unsigned voff, char* lds_dst) {
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(lds_dst), 16, voff, 0, 0, 0);
}

// fragment of a row-major-K tile: 16 rows starting at r0, k-step kk (32 wide)
__device__ __forceinline__ bf16x8 frag_rowk(const char* tile, int r0, int kk, int lane) {
    const int r = r0 + (lane & 15);
    const int c = kk * 4 + (lane >> 4);
    return *reinterpret_cast<const bf16x8*>(tile + r * 128 + ((c ^ ((r >> 1) & 7)) << 4));
}

// fragment of a K-strided tile [64 k][128 cols]: 16 cols starting at c0, k-step kk, via two transposing reads
__device__ __forceinline__ bf16x8 frag_tr(const char* tile, int c0, int kk, int lane) {
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int f = q | ((g & 1) << 2);
    const int chunk = ((c0 >> 3) + (p >> 1)) ^ (f << 1);
    const int row = kk * 32 + 8 * g + q;
    const char* a0 = tile + row * 256 + (chunk << 4) + (p & 1) * 8;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(a0));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(a0 + 4 * 256));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

template <bool A_TR, bool B_TR, int OUT_DT>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_bf16_kernel(GemmParams p) {
    __shared__ __attribute__((aligned(16))) char smem[SMEM_BYTES];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    // ---- workgroup -> tile: XCD chunking, then 8-row super-groups --------------------------------------
    const int nwg = p.tiles_m * p.tiles_n;
    int pid;
    {
        const int b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
        pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    const int GROUP_M = 8;
    const int in_group = GROUP_M * p.tiles_n;
    const int first_m = (pid / in_group) * GROUP_M;
    const int gsz = min(p.tiles_m - first_m, GROUP_M);
    const int tm = first_m + (pid % in_group) % gsz;
    const int tn = (pid % in_group) / gsz;
    const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;

    // ---- DMA plan -------------------------------------------------------------------------------------
    unsigned voffA[4], voffB[4];
    int kneedA[4], kneedB[4];
    piece_offsets<A_TR>(wave, lane, p.lda, p.M - m0, voffA, kneedA);
    piece_offsets<B_TR>(wave, lane, p.ldb, p.N - n0, voffB, kneedB);
    const bf16_t* baseA = A_TR ? p.A + m0 : p.A + m0 * p.lda;
    const bf16_t* baseB = B_TR ? p.B + n0 : p.B + n0 * p.ldb;
    const int64_t stepA = A_TR ? (int64_t)BK * p.lda : BK;
    const int64_t stepB = B_TR ? (int64_t)BK * p.ldb : BK;

    auto issue_tile = [&](int t, int stage) {
        const int64_t krem = p.K - (int64_t)t * BK;
        const bf16_t* pa = baseA + t * stepA;
        const bf16_t* pb = baseB + t * stepB;
        char* dst = smem + stage * STAGE_BYTES + wave * 4096;
#pragma unroll
        for (int j = 0; j < 4; ++j) dma_piece(pa, kneedA[j] < krem ? voffA[j] : OOB, dst + j * 1024);
#pragma unroll
        for (int j = 0; j < 4; ++j) dma_piece(pb, kneedB[j] < krem ? voffB[j] : OOB, dst + TILE_BYTES + j * 1024);
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int wr0 = (wave >> 1) * 64, wc0 = (wave & 1) * 64;
    const int nt = (int)((p.K + BK - 1) / BK);

    issue_tile(0, 0);
    for (int t = 0; t < nt; ++t) {
        // tile t landed (every wave drains its own DMA) and everyone is done reading the other stage
        __syncthreads();
        if (t + 1 < nt) issue_tile(t + 1, (t + 1) & 1);
        const char* sA = smem + (t & 1) * STAGE_BYTES;
        const char* sB = sA + TILE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = A_TR ? frag_tr(sA, wr0 + i * 16, kk, lane) : frag_rowk(sA, wr0 + i * 16, kk, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = B_TR ? frag_tr(sB, wc0 + j * 16, kk, lane) : frag_rowk(sB, wc0 + j * 16, kk, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }

    // ---- epilogue: acc -> LDS (fp32) -> coalesced rows ---------------------------------------------------
    __syncthreads();
    float* stg = reinterpret_cast<float*>(smem) + wave * 64 * EPI_LD;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) stg[(i * 16 + (lane >> 4) * 4 + e) * EPI_LD + j * 16 + (lane & 15)] = acc[i][j][e];
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's LDS writes are done before it reads them back

    const int64_t gn = n0 + wc0 + (lane & 7) * 8;
    const bool vec_ok = (gn + 8 <= p.N) && ((p.ldc & 7) == 0) && ((p.ldr & 7) == 0 || p.R == nullptr);
#pragma unroll
    for (int tpass = 0; tpass < 8; ++tpass) {
        const int row = tpass * 8 + (lane >> 3);
        const int64_t gm = m0 + wr0 + row;
        if (gm >= p.M || gn >= p.N) continue;
        float v[8];
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(stg + row * EPI_LD + (lane & 7) * 8);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(stg + row * EPI_LD + (lane & 7) * 8 + 4);
        v[0] = v0[0]; v[1] = v0[1]; v[2] = v0[2]; v[3] = v0[3];
        v[4] = v1[0]; v[5] = v1[1]; v[6] = v1[2]; v[7] = v1[3];
        const int nvalid = (int)min((int64_t)8, p.N - gn);
        if (p.bias) {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (e < nvalid) v[e] += p.bias[gn + e];
        }
        if (p.epilogue == MI355_EPI_GELU_ERF) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = gelu_erf(v[e]);
        }
        if constexpr (OUT_DT == MI355_DT_BF16) {
            bf16_t* c = reinterpret_cast<bf16_t*>(p.C) + gm * p.ldc + gn;
            const bf16_t* r = p.R ? reinterpret_cast<const bf16_t*>(p.R) + gm * p.ldr + gn : nullptr;
            if (vec_ok) {
                if (r) {
                    const u32x4 rv = *reinterpret_cast<const u32x4*>(r);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[2 * e] += __uint_as_float(rv[e] << 16);
                        v[2 * e + 1] += __uint_as_float(rv[e] & 0xffff0000u);
                    }
                }
                u32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = pack_bf2(v[2 * e], v[2 * e + 1]);
                *reinterpret_cast<u32x4*>(c) = o;
            } else {
                for (int e = 0; e < nvalid; ++e) c[e] = f2bf(v[e] + (r ? bf2f(r[e]) : 0.f));
            }
        } else {
            float* c = reinterpret_cast<float*>(p.C) + gm * p.ldc + gn;
            const float* r = p.R ? reinterpret_cast<const float*>(p.R) + gm * p.ldr + gn : nullptr;
            if (vec_ok) {
                f32x4 o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
                if (r) {
                    o0 += *reinterpret_cast<const f32x4*>(r);
                    o1 += *reinterpret_cast<const f32x4*>(r + 4);
                }
                *reinterpret_cast<f32x4*>(c) = o0;
                *reinterpret_cast<f32x4*>(c + 4) = o1;
            } else {
                for (int e = 0; e < nvalid; ++e) c[e] = v[e] + (r ? r[e] : 0.f);
            }
        }
    }
}

template <bool A_TR, bool B_TR>
int launch(const GemmParams& p, int out_dtype, hipStream_t s) {
    const int grid = p.tiles_m * p.tiles_n;
    if (out_dtype == MI355_DT_BF16)
        hipLaunchKernelGGL((gemm_bf16_kernel<A_TR, B_TR, MI355_DT_BF16>), dim3(grid), dim3(NTHREADS), 0, s, p);
    else
        hipLaunchKernelGGL((gemm_bf16_kernel<A_TR, B_TR, MI355_DT_F32>), dim3(grid), dim3(NTHREADS), 0, s, p);
    MI355_LAUNCH_CHECK("mi355_gemm_bf16");
    return 0;
}

// ---------------------------------------------------------------------------------------------- colsum
__global__ __launch_bounds__(256) void colsum_kernel(int64_t M, int64_t N, const bf16_t* X, int64_t ldx, float* out,
                                                     int accumulate, int rows_per_block) {
    // block = 256 threads = 64 columns x 4 row-lanes; grid.x over column groups, grid.y over row slabs
    __shared__ float red[4][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    const int64_t r1 = min(M, r0 + rows_per_block);
    float s = 0.f;
    if (col < N)
        for (int64_t r = r0 + rl; r < r1; r += 4) s += bf2f(X[r * ldx + col]);
    red[rl][threadIdx.x & 63] = s;
    __syncthreads();
    if (rl == 0 && col < N) {
        s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        atomicAdd(out + col, s);
    }
    (void)accumulate;
}

}  // namespace

extern "C" int mi355_gemm_bf16(int form, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B,
                               int64_t ldb, void* C, int64_t ldc, int out_dtype, const float* bias,
                               const void* residual, int64_t ldr, int epilogue, void* stream) {
    MI355_REQUIRE(form >= 0 && form <= 2, "mi355_gemm_bf16: bad form %d", form);
    MI355_REQUIRE(M > 0 && N > 0 && K > 0, "mi355_gemm_bf16: empty problem M=%ld N=%ld K=%ld", (long)M, (long)N, (long)K);
    MI355_REQUIRE(out_dtype == MI355_DT_BF16 || out_dtype == MI355_DT_F32, "mi355_gemm_bf16: bad out_dtype");
    MI355_REQUIRE(A && B && C, "mi355_gemm_bf16: null operand");
    MI355_REQUIRE((lda & 7) == 0 && (ldb & 7) == 0, "mi355_gemm_bf16: lda/ldb must be multiples of 8 (16-byte rows)");
    MI355_REQUIRE(((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0, "mi355_gemm_bf16: A/B must be 16-byte aligned");
    if (form == MI355_GEMM_NT) MI355_REQUIRE((K & 7) == 0, "mi355_gemm_bf16(NT): K must be a multiple of 8");
    if (form == MI355_GEMM_NN) MI355_REQUIRE((K & 7) == 0 && (N & 7) == 0, "mi355_gemm_bf16(NN): K,N must be multiples of 8");
    if (form == MI355_GEMM_TN) MI355_REQUIRE((M & 7) == 0 && (N & 7) == 0, "mi355_gemm_bf16(TN): M,N must be multiples of 8");
    // a tile's DMA offsets are 31-bit: 128 rows (or 64 k-rows) of one operand must span < 2 GiB
    MI355_REQUIRE(lda * 2 * 128 < 0x7fffffffLL && ldb * 2 * 128 < 0x7fffffffLL, "mi355_gemm_bf16: leading dimension too large");
    GemmParams p;
    p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = C; p.bias = bias; p.R = residual;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldr = ldr;
    p.tiles_m = (int)((M + BM - 1) / BM); p.tiles_n = (int)((N + BN - 1) / BN); p.epilogue = epilogue;
    MI355_REQUIRE((int64_t)p.tiles_m * p.tiles_n < 0x7fffffffLL, "mi355_gemm_bf16: grid too large");
    hipStream_t s = (hipStream_t)stream;
    switch (form) {
        case MI355_GEMM_NT: return launch<false, false>(p, out_dtype, s);
        case MI355_GEMM_NN: return launch<false, true>(p, out_dtype, s);
        default: return launch<true, true>(p, out_dtype, s);
    }
}

extern "C" int mi355_colsum_bf16(int64_t M, int64_t N, const void* X, int64_t ldx, float* out, int accumulate,
                                 void* stream) {
    MI355_REQUIRE(M > 0 && N > 0 && X && out, "mi355_colsum_bf16: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (!accumulate) {
        if (hipMemsetAsync(out, 0, N * sizeof(float), s) != hipSuccess) {
            mi355_set_error("mi355_colsum_bf16: memset failed");
            return 2;
        }
    }
    const int rows_per_block = 512;
    dim3 grid((unsigned)((N + 63) / 64), (unsigned)((M + rows_per_block - 1) / rows_per_block));
    hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, s, M, N, (const bf16_t*)X, ldx, out, accumulate, rows_per_block);
    MI355_LAUNCH_CHECK("mi355_colsum_bf16");
    return 0;
}
