// Chunked (WY / UT-transform) FORWARD of the gated delta rule on fp32-input MFMA (v_mfma_f32_16x16x4_f32), head dims 128 x 128:
// the no-grad / prefill path of gated_delta_rule (reference llm_quest/qwen/qwen3_next/qwen3_next_attention.py:103-159).  The training path keeps the
// sequential kernels of qwen35.hip (their backward replays from 8-step checkpoints this form does not produce); algorithm, its hand-written backward and
// what both cost on this chip: tools/gdr_chunk_proto.py and DESIGN.md section 5.
//
// Per (batch row, value head) and chunk of C = 64 tokens with incoming state S0 [Dv, Dk] (all fp32):
//     g = cumsum(log alpha), gam = exp(g), D[i, j] = exp(g_i - g_j) (i >= j), Gp_j = exp(g_C - g_j)
//     M = diag(beta) tril(K K^T * D, -1),  R = (I + M)^-1,  T = R diag(beta),  P' = scale tril(Q K^T * D)
//     Uv = T V,  nKw = -(T diag(gam)) K,  Q' = scale diag(gam) Q,  K'T = (diag(Gp) K)^T                  -- gdr_chunk_prep_kernel, one workgroup per chunk
//     U = Uv + nKw S0^T,  O = Q' S0^T + P' U,  S_C = gam_C S0 + (K'T U)^T                               -- gdr_chunk_scan_kernel, chunks in sequence
// The scan keeps S^T and U as MFMA accumulators and feeds them back as B operands without moving them: a 16x16 accumulator tile holds rows 4 (lane >> 4) + e
// (e = 0..3) of column lane & 15 -- exactly the four k values a lane supplies to four consecutive 16x16x4 steps when the contraction index is permuted
// as k = 16 kb + 4 (lane >> 4) + e, which the A operand (read from LDS as one 16-byte piece per lane) follows for free.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int GC = 64;         // chunk length
constexpr int GD = 128;        // Dk = Dv
constexpr int LDW = GD + 4;    // row pitch (floats) of a [rows][128] LDS image: 16-byte aligned rows, bank-spread
constexpr int LDC = GC + 4;    // row pitch of a [rows][64] LDS image
// workspace per (chunk, head), in floats
constexpr int W_NKW = 0, W_UV = W_NKW + GC * GD, W_P = W_UV + GC * GD, W_Q = W_P + GC * GC, W_KT = W_Q + GC * GD, W_GC = W_KT + GD * GC, W_PER = W_GC + 32;

__device__ __forceinline__ f32x4 mma4(f32x4 acc, const f32x4 a, const f32x4 b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], acc, 0, 0, 0);
    return acc;
}
__device__ __forceinline__ f32x4 ldsv(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// ---------------------------------------------------------------------------------------------------------------- the part that needs no state
__global__ __launch_bounds__(256) void gdr_chunk_prep_kernel(int B, int S, int Hqk, int Hv, const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                             const bf16_t* __restrict__ v, int64_t ldv, const float* __restrict__ beta,
                                                             const float* __restrict__ alpha, float* __restrict__ ws, int nchunk, float scale, int abl) {
    __shared__ __attribute__((aligned(16))) float Ks[GC * LDW];   // K rows
    __shared__ __attribute__((aligned(16))) float Xs[GD * LDC];   // Q rows ([GC][LDW] fits), later V^T
    __shared__ __attribute__((aligned(16))) float KT[GD * LDC];   // K^T
    __shared__ __attribute__((aligned(16))) float Ms[GC * LDC];   // diag(beta) tril(K K^T * D, -1)
    __shared__ __attribute__((aligned(16))) float Ts[GC * LDC];   // T = (I + M)^-1 diag(beta)
    __shared__ __attribute__((aligned(16))) float gs[GC], gam[GC], gp[GC], bet[GC];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, g = lane >> 4;
    const int c = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int hq = h / (Hv / Hqk);
    const int t0 = c * GC, nv = min(GC, S - t0);
    const int64_t tok0 = (int64_t)b * S + t0;
    const int64_t ldqk = (int64_t)Hqk * GD;
    float* w = ws + (((int64_t)b * Hv + h) * nchunk + c) * W_PER;

    // ---- a. K and Q rows to LDS (fp32), the decay vectors
    {
        const int row = tid >> 2, part = (tid & 3) * 32;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            u32x4 kv = {0u, 0u, 0u, 0u}, qv = {0u, 0u, 0u, 0u};
            if (row < nv) {
                kv = *reinterpret_cast<const u32x4*>(k + (tok0 + row) * ldqk + hq * GD + part + 8 * u);
                qv = *reinterpret_cast<const u32x4*>(q + (tok0 + row) * ldqk + hq * GD + part + 8 * u);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                Ks[row * LDW + part + 8 * u + 2 * e] = __uint_as_float(kv[e] << 16);
                Ks[row * LDW + part + 8 * u + 2 * e + 1] = __uint_as_float(kv[e] & 0xffff0000u);
                Xs[row * LDW + part + 8 * u + 2 * e] = __uint_as_float(qv[e] << 16);
                Xs[row * LDW + part + 8 * u + 2 * e + 1] = __uint_as_float(qv[e] & 0xffff0000u);
            }
        }
    }
    if (wave == 0) {  // padded rows of the last chunk are identity steps: alpha = 1, beta = 0, k = q = v = 0
        const float a = lane < nv ? alpha[(tok0 + lane) * Hv + h] : 1.0f;
        float cs = logf(a);
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const float up = __shfl_up(cs, o, 64);
            if (lane >= o) cs += up;
        }
        const float last = __shfl(cs, 63, 64);
        gs[lane] = cs;
        gam[lane] = expf(cs);
        gp[lane] = expf(last - cs);
        bet[lane] = lane < nv ? beta[(tok0 + lane) * Hv + h] : 0.0f;
    }
    __syncthreads();

    // ---- b. K K^T and Q K^T (wave = 16-row block), masked and decayed
#pragma unroll 1
    for (int nt = 0; nt < ((abl & 2) ? 0 : 4); ++nt) {
        f32x4 akk = {0.f, 0.f, 0.f, 0.f}, aqk = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < GD / 16; ++kb) {
            const f32x4 bk = ldsv(Ks + (16 * nt + r) * LDW + 16 * kb + 4 * g);
            akk = mma4(akk, ldsv(Ks + (16 * wave + r) * LDW + 16 * kb + 4 * g), bk);
            aqk = mma4(aqk, ldsv(Xs + (16 * wave + r) * LDW + 16 * kb + 4 * g), bk);
        }
        const int j = 16 * nt + r;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = 16 * wave + 4 * g + e;
            const float d = j <= i ? expf(gs[i] - gs[j]) : 0.0f;
            Ms[i * LDC + j] = j < i ? bet[i] * d * akk[e] : 0.0f;
            w[W_P + i * GC + j] = scale * d * aqk[e];
        }
    }
    __syncthreads();

    // ---- c. wave 0: R = (I + M)^-1 by forward substitution, a column per lane (row i of R = e_i - sum_{k<i} M[i][k] R[k]);  waves 1-3: Q', then K^T, V^T, K'T
    if (wave == 0) {
        float R[GC];
        if (abl & 1) {  // profiling only: T = diag(beta)
#pragma unroll
            for (int i = 0; i < GC; ++i) Ts[i * LDC + lane] = i == lane ? bet[lane] : 0.0f;
        } else {
#pragma unroll
        for (int i = 0; i < GC; ++i) {
            float s = i == lane ? 1.0f : 0.0f;
#pragma unroll
            for (int kk = 0; kk < i; ++kk) s -= Ms[i * LDC + kk] * R[kk];
            R[i] = s;
        }
        const float bc = bet[lane];
#pragma unroll
        for (int i = 0; i < GC; ++i) Ts[i * LDC + lane] = R[i] * bc;
        }
    } else {
        const int t3 = tid - 64;  // 0..191
        for (int idx = t3; idx < GC * GD / 4; idx += 192) {  // Q' = scale gam_i Q, row-major [GC][GD]
            const int i = idx / (GD / 4), d4 = (idx % (GD / 4)) * 4;
            f32x4 x = ldsv(Xs + i * LDW + d4);
            const float f = scale * gam[i];
            x *= f;
            *reinterpret_cast<f32x4*>(w + W_Q + i * GD + d4) = x;
        }
        for (int idx = t3; idx < GC * GD; idx += 192) {  // K^T in LDS
            const int j = idx % GC, d = idx / GC;
            KT[d * LDC + j] = Ks[j * LDW + d];
        }
    }
    __syncthreads();
    // V^T over the Q rows' LDS, K'T out
    for (int idx = tid; idx < GC * GD / 8; idx += 256) {
        const int j = idx / (GD / 8), d8 = (idx % (GD / 8)) * 8;
        u32x4 vv = {0u, 0u, 0u, 0u};
        if (j < nv) vv = *reinterpret_cast<const u32x4*>(v + (tok0 + j) * ldv + (int64_t)h * GD + d8);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            Xs[(d8 + 2 * e) * LDC + j] = __uint_as_float(vv[e] << 16);
            Xs[(d8 + 2 * e + 1) * LDC + j] = __uint_as_float(vv[e] & 0xffff0000u);
        }
    }
    for (int idx = tid; idx < GD * GC / 4; idx += 256) {
        const int d = idx / (GC / 4), j4 = (idx % (GC / 4)) * 4;
        f32x4 x = ldsv(KT + d * LDC + j4);
        x *= ldsv(gp + j4);
        *reinterpret_cast<f32x4*>(w + W_KT + d * GC + j4) = x;
    }
    if (tid == 0) w[W_GC] = gam[GC - 1];
    __syncthreads();

    // ---- d / e. Uv = T V (accumulator-tile layout of the scan), nKw = -(T diag(gam)) K (row-major)
#pragma unroll 1
    for (int nt = 0; nt < ((abl & 4) ? 0 : GD / 16); ++nt) {
        f32x4 au = {0.f, 0.f, 0.f, 0.f}, ak = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < GC / 16; ++kb) {
            const f32x4 t4 = ldsv(Ts + (16 * wave + r) * LDC + 16 * kb + 4 * g);
            au = mma4(au, t4, ldsv(Xs + (16 * nt + r) * LDC + 16 * kb + 4 * g));
            ak = mma4(ak, t4, ldsv(KT + (16 * nt + r) * LDC + 16 * kb + 4 * g) * ldsv(gam + 16 * kb + 4 * g));
        }
        *reinterpret_cast<f32x4*>(w + W_UV + ((wave * (GD / 16) + nt) * 64 + lane) * 4) = au;
#pragma unroll
        for (int e = 0; e < 4; ++e) w[W_NKW + (16 * wave + 4 * g + e) * GD + 16 * nt + r] = -ak[e];
    }
}

// ---------------------------------------------------------------------------------------------------------------- the scan over chunks
// NSPLIT workgroups per (batch row, head) split the Dv columns (state rows are independent); a wave owns NTW 16-column tiles of them.
template <int NSPLIT>
__global__ __launch_bounds__(256) void gdr_chunk_scan_kernel(int B, int S, int Hv, const float* __restrict__ ws, int nchunk, bf16_t* __restrict__ o,
                                                             const float* initial_state, float* final_state) {  // the two may alias (state updated in place): no __restrict__
    constexpr int NTW = (GD / 16) / (4 * NSPLIT);
    __shared__ __attribute__((aligned(16))) float As[GC * LDW];   // nKw
    __shared__ __attribute__((aligned(16))) float Bs[GC * LDW];   // Q'
    __shared__ __attribute__((aligned(16))) float Ps[GC * LDC];   // P'
    __shared__ __attribute__((aligned(16))) float KTs[GD * LDC];  // K'T
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, g = lane >> 4;
    const int h = blockIdx.x / NSPLIT, part = blockIdx.x % NSPLIT, b = blockIdx.y;
    const int nt0 = part * (GD / 16 / NSPLIT) + wave * NTW;  // first 16-column tile of this wave
    f32x4 St[GD / 16][NTW];  // S^T tiles: rows d = 16 dt + 4 g + e, column n = 16 (nt0 + t) + r
#pragma unroll
    for (int dt = 0; dt < GD / 16; ++dt)
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            St[dt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (initial_state)
                St[dt][t] = *reinterpret_cast<const f32x4*>(initial_state + (((int64_t)b * Hv + h) * GD + 16 * (nt0 + t) + r) * GD + 16 * dt + 4 * g);
        }
    for (int c = 0; c < nchunk; ++c) {
        const float* w = ws + (((int64_t)b * Hv + h) * nchunk + c) * W_PER;
        __syncthreads();  // the previous chunk's reads of the images are complete
        for (int idx = tid; idx < GC * GD / 4; idx += 256) {
            const int i = idx / (GD / 4), d4 = (idx % (GD / 4)) * 4;
            *reinterpret_cast<f32x4*>(As + i * LDW + d4) = *reinterpret_cast<const f32x4*>(w + W_NKW + i * GD + d4);
            *reinterpret_cast<f32x4*>(Bs + i * LDW + d4) = *reinterpret_cast<const f32x4*>(w + W_Q + i * GD + d4);
        }
        for (int idx = tid; idx < GC * GC / 4; idx += 256) {
            const int i = idx / (GC / 4), j4 = (idx % (GC / 4)) * 4;
            *reinterpret_cast<f32x4*>(Ps + i * LDC + j4) = *reinterpret_cast<const f32x4*>(w + W_P + i * GC + j4);
        }
        for (int idx = tid; idx < GD * GC / 4; idx += 256) {
            const int d = idx / (GC / 4), j4 = (idx % (GC / 4)) * 4;
            *reinterpret_cast<f32x4*>(KTs + d * LDC + j4) = *reinterpret_cast<const f32x4*>(w + W_KT + d * GC + j4);
        }
        const float gC = w[W_GC];
        __syncthreads();
        const int nv = min(GC, S - c * GC);
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            f32x4 U[GC / 16];
            // U = Uv + nKw S0^T
#pragma unroll
            for (int mt = 0; mt < GC / 16; ++mt) {
                f32x4 acc = *reinterpret_cast<const f32x4*>(w + W_UV + ((mt * (GD / 16) + nt0 + t) * 64 + lane) * 4);
#pragma unroll
                for (int kb = 0; kb < GD / 16; ++kb) acc = mma4(acc, ldsv(As + (16 * mt + r) * LDW + 16 * kb + 4 * g), St[kb][t]);
                U[mt] = acc;
            }
            // O = Q' S0^T + P' U
#pragma unroll
            for (int mt = 0; mt < GC / 16; ++mt) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kb = 0; kb < GD / 16; ++kb) acc = mma4(acc, ldsv(Bs + (16 * mt + r) * LDW + 16 * kb + 4 * g), St[kb][t]);
#pragma unroll
                for (int kb = 0; kb < GC / 16; ++kb) acc = mma4(acc, ldsv(Ps + (16 * mt + r) * LDC + 16 * kb + 4 * g), U[kb]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = 16 * mt + 4 * g + e;
                    if (i < nv) o[((int64_t)b * S + c * GC + i) * Hv * GD + (int64_t)h * GD + 16 * (nt0 + t) + r] = f2bf(acc[e]);
                }
            }
            // S_C^T = gam_C S0^T + K'T U
#pragma unroll
            for (int dt = 0; dt < GD / 16; ++dt) {
                f32x4 acc = St[dt][t] * gC;
#pragma unroll
                for (int kb = 0; kb < GC / 16; ++kb) acc = mma4(acc, ldsv(KTs + (16 * dt + r) * LDC + 16 * kb + 4 * g), U[kb]);
                St[dt][t] = acc;
            }
        }
    }
    if (final_state)
#pragma unroll
        for (int dt = 0; dt < GD / 16; ++dt)
#pragma unroll
            for (int t = 0; t < NTW; ++t)
                *reinterpret_cast<f32x4*>(final_state + (((int64_t)b * Hv + h) * GD + 16 * (nt0 + t) + r) * GD + 16 * dt + 4 * g) = St[dt][t];
}

}  // namespace

extern "C" int64_t mi355_gated_delta_rule_chunked_workspace_bytes(int B, int S, int Hv) {
    if (B <= 0 || S <= 0 || Hv <= 0) return 0;
    return (int64_t)B * Hv * ((S + GC - 1) / GC) * W_PER * 4;
}

extern "C" int mi355_gated_delta_rule_chunked_fwd(int B, int S, int Hqk, int Hv, int Dk, int Dv, const void* q, const void* k, const void* v, int64_t ldv,
                                                  const float* beta, const float* alpha, void* o, const float* initial_state, float* final_state,
                                                  float* workspace, int64_t workspace_bytes, void* stream) {
    MI355_REQUIRE(B > 0 && S > 0 && Hqk > 0 && Hv > 0 && Hv % Hqk == 0, "gated_delta_rule_chunked_fwd: value heads (%d) must be a multiple of q/k heads (%d)", Hv, Hqk);
    MI355_REQUIRE(Dk == GD && Dv == GD, "gated_delta_rule_chunked_fwd: head dims %d x %d not built (128 x 128 only; the sequential kernels take the rest)", Dk, Dv);
    MI355_REQUIRE(q && k && v && beta && alpha && o && workspace && ldv >= (int64_t)Hv * Dv && (ldv & 7) == 0, "gated_delta_rule_chunked_fwd: bad arguments");
    MI355_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)workspace) & 15) == 0 && (!initial_state || ((uintptr_t)initial_state & 15) == 0) &&
                      (!final_state || ((uintptr_t)final_state & 15) == 0),
                  "gated_delta_rule_chunked_fwd: operands, states and workspace must be 16-byte aligned");
    MI355_REQUIRE(workspace_bytes >= mi355_gated_delta_rule_chunked_workspace_bytes(B, S, Hv), "gated_delta_rule_chunked_fwd: workspace of %lld bytes, %lld needed",
                  (long long)workspace_bytes, (long long)mi355_gated_delta_rule_chunked_workspace_bytes(B, S, Hv));
    MI355_REQUIRE(Hv <= 65535 && B <= 65535, "gated_delta_rule_chunked_fwd: grid limits");
    const int nchunk = (S + GC - 1) / GC;
    hipStream_t s = (hipStream_t)stream;
    const float scale = 1.0f / sqrtf((float)Dk);
#ifndef GDR_ABL
#define GDR_ABL 0  // profiling builds only (make FLAGS_gdr_chunk=-DGDR_ABL=n): 1 no inverse, 2 no K K^T / Q K^T, 4 no T V / T K, 8 no scan (outputs then unwritten)
#endif
    constexpr int abl = GDR_ABL;
    hipLaunchKernelGGL(gdr_chunk_prep_kernel, dim3((unsigned)nchunk, (unsigned)Hv, (unsigned)B), dim3(256), 0, s, B, S, Hqk, Hv, (const bf16_t*)q, (const bf16_t*)k,
                       (const bf16_t*)v, ldv, beta, alpha, workspace, nchunk, scale, abl);
    if (abl & 8) {
    } else if ((int64_t)B * Hv >= 256)
        hipLaunchKernelGGL(gdr_chunk_scan_kernel<1>, dim3((unsigned)Hv, (unsigned)B), dim3(256), 0, s, B, S, Hv, (const float*)workspace, nchunk, (bf16_t*)o, initial_state, final_state);
    else
        hipLaunchKernelGGL(gdr_chunk_scan_kernel<2>, dim3((unsigned)Hv * 2, (unsigned)B), dim3(256), 0, s, B, S, Hv, (const float*)workspace, nchunk, (bf16_t*)o, initial_state, final_state);
    MI355_LAUNCH_CHECK("gated_delta_rule_chunked_fwd");
    return 0;
}
