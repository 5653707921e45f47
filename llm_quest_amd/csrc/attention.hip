// Flash-style attention forward + backward for gfx950, token-major operands (no head transposes in HBM).
//
// Forward (and the dQ pass of backward) put the QUERY on the MFMA lane:  S^T = K Q^T  via
// mfma_f32_32x32x16_bf16(A = K rows from LDS, B = Q rows held in registers), so each lane owns one query column,
// the online-softmax row statistics are lane-local (one cross-half exchange), and the fp32 accumulator tile of
// P^T is, after a pairwise bf16 pack, directly the B operand of  O^T += V^T P^T  (no LDS round trip for P).
// V (and K in the dQ pass) is consumed K-strided through ds_read_b64_tr_b16 from a row-major LDS image.
// K/V tiles (64 keys) arrive by LDS-DMA (buffer_load ... lds) with the bank swizzle on the source address,
// double-buffered: the next tile's DMA is in flight under the current tile's MFMAs.
//
// Masking follows the reference (qwen3_attention.py:130-142): masked scores take a FINITE fill value, so a row
// whose visible keys are all masked degenerates to uniform attention over all S keys exactly as upstream; keys
// beyond S do not exist and get -inf.  Scores stay in fp32 (the reference rounds them to bf16 twice).
#include "attn_common.h"

namespace {

template <int D>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(int B, int S, int Hq, int Hkv, const bf16_t* __restrict__ q, int64_t ldq,
                                                          const bf16_t* __restrict__ k, int64_t ldk, const bf16_t* __restrict__ v,
                                                          int64_t ldv, bf16_t* __restrict__ o, int64_t ldo, float* __restrict__ lse,
                                                          const uint8_t* __restrict__ key_mask, int causal, float scale_log2) {
    using C = Cfg<D>;
    const int abl = causal >> 8;  // profiling only: 1 = no DMA after the first tile, 2 = no exp
    causal &= 0xff;
    __shared__ __attribute__((aligned(16))) char smem[4 * C::TILE];  // 2 stages x (K row image, V tr image)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nqb = (S + 127) / 128;
    // heaviest (latest) query blocks first under the causal mask
    const int vid = (abl >> 8) & 1 ? (int)blockIdx.x : xcd_chunked((int)blockIdx.x, (int)gridDim.x);  // ablation bit 8: plain order
    const int qb = nqb - 1 - (vid % nqb);
    const int bh = vid / nqb;
    const int hq = bh % Hq, b = bh / Hq;
    const int hkv = hq / (Hq / Hkv);
    const int q0 = qb * 128;
    const int qg = q0 + wave * 32 + (lane & 31);
    const bool qvalid = qg < S;

    const bf16_t* kbase = k + (int64_t)b * S * ldk + (int64_t)hkv * D;
    const bf16_t* vbase = v + (int64_t)b * S * ldv + (int64_t)hkv * D;
    const int ntiles_all = (S + 63) / 64;
    int ntiles = causal ? min(ntiles_all, (q0 + 127) / 64 + 1) : ntiles_all;

    f32x16 oacc[C::DT];
#pragma unroll
    for (int i = 0; i < C::DT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[i][e] = 0.f;
    float m = -INFINITY, l = 0.f;

    auto issue = [&](int kt, int stage) {
        char* ks_ = smem + stage * 2 * C::TILE;
        dma_tile<D, IMG_ROW>(kbase + (int64_t)kt * 64 * ldk, ldk, S - kt * 64, ks_, wave, lane);
        dma_tile<D, IMG_TR>(vbase + (int64_t)kt * 64 * ldv, ldv, S - kt * 64, ks_ + C::TILE, wave, lane);
    };

    const LaneOff<D> lo = lane_offsets<D>(lane);
    issue(0, 0);  // first tile first, then the query rows: their latencies overlap
    bf16x8 qf[C::KS];
    load_rows_frag<D>(q + (int64_t)b * S * ldq + (int64_t)hq * D, ldq, qg, qvalid, lane, qf);
    // key-padding byte of this lane's key in the NEXT tile: loaded one tile ahead so its latency hides behind a whole tile
    uint8_t mk = (key_mask && lane < S) ? key_mask[(int64_t)b * S + lane] : (uint8_t)0;
    [[maybe_unused]] const bool prof_on = threadIdx.x == 0;
    [[maybe_unused]] unsigned long long prof_acc[16] = {};
    [[maybe_unused]] const unsigned long long t_wg = PROF_T();
    for (int kt = 0; kt < ntiles; ++kt) {
        [[maybe_unused]] const unsigned long long t_sync = PROF_T();
        __syncthreads();
        PROF_ADD(1, t_sync);
        [[maybe_unused]] const unsigned long long t_iss = PROF_T();
        if (kt + 1 < ntiles && !(abl & 1)) issue(kt + 1, (kt + 1) & 1);
        PROF_ADD(2, t_iss);
        [[maybe_unused]] unsigned long long t_seg = PROF_T();
        if (prof_on) prof_acc[7] += 1;
        const int koff = (kt & 1) * 2 * C::TILE, voff = koff + C::TILE;
        // key-padding bits of this tile (1 = real token); keys beyond S read as padding here and are removed below
        unsigned long long kbits = ~0ull;
        if (key_mask) {
            kbits = __ballot(mk != 0);
            const int kn = (kt + 1) * 64 + lane;
            mk = kn < S ? key_mask[(int64_t)b * S + kn] : (uint8_t)0;
        }
        // a wave whose 32 queries all precede this tile has nothing visible here (unless a row is still fully masked)
        const bool wave_active = !causal || (kt * 64 <= q0 + wave * 32 + 31) || __any(m == MASK_T);
        if (wave_active) {
            int vkx[C::KS], vv[C::DT];
#pragma unroll
            for (int ks = 0; ks < C::KS; ++ks) vkx[ks] = (lo.row + koff) ^ (ks << 5);
#pragma unroll
            for (int dt = 0; dt < C::DT; ++dt) vv[dt] = lo.col[dt] + voff;
            f32x16 sacc[2];
#pragma unroll
            for (int st = 0; st < 2; ++st) {
#pragma unroll
                for (int e = 0; e < 16; ++e) sacc[st][e] = 0.f;
#pragma unroll
                for (int ks = 0; ks < C::KS; ++ks)
                    sacc[st] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_frag(smem, vkx[ks], st * 32 * C::ROWB), qf[ks], sacc[st], 0, 0, 0);
            }
#if ATTN_ABL & 16
            asm volatile("" : "+v"(sacc[0]), "+v"(sacc[1]));
            if (prof_on) { const unsigned long long now = __builtin_readcyclecounter(); prof_acc[3] += now - t_seg; t_seg = now; }
#endif
            // masks are needed only on the diagonal / tail / padded tiles (wave-uniform test)
            const bool boundary = (kbits != ~0ull) || (kt * 64 + 64 > S) || (causal && kt * 64 + 63 > q0 + wave * 32);
            float mnew;
            if (boundary) {
                float tmax = -INFINITY;
#pragma unroll
                for (int st = 0; st < 2; ++st)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int kl = st * 32 + acc_row(e, lane);
                        const int kg = kt * 64 + kl;
                        float t = sacc[st][e] * scale_log2;
                        const bool masked = (causal && kg > qg) || !((kbits >> kl) & 1ull);
                        t = masked ? MASK_T : t;
                        t = kg < S ? t : -INFINITY;
                        sacc[st][e] = t;
                        tmax = fmaxf(tmax, t);
                    }
                tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
                mnew = fmaxf(m, tmax);
#pragma unroll
                for (int st = 0; st < 2; ++st)
#pragma unroll
                    for (int e = 0; e < 16; ++e) sacc[st][e] = __builtin_amdgcn_exp2f(sacc[st][e] - mnew);
            } else {
                float rmax = sacc[0][0];
#pragma unroll
                for (int st = 0; st < 2; ++st)
#pragma unroll
                    for (int e = 0; e < 16; ++e) rmax = fmaxf(rmax, sacc[st][e]);
                rmax = fmaxf(rmax, __shfl_xor(rmax, 32, 64));
                mnew = fmaxf(m, rmax * scale_log2);
#pragma unroll
                for (int st = 0; st < 2; ++st)
#pragma unroll
                    for (int e = 0; e < 16; ++e) sacc[st][e] = __builtin_amdgcn_exp2f(fmaf(sacc[st][e], scale_log2, -mnew));
            }
            float psum = 0.f;
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int e = 0; e < 16; ++e) psum += sacc[st][e];
            if (__any(mnew != m)) {  // the running max moved for some row of this wave: rescale (else alpha == 1 exactly)
                const float alpha = __builtin_amdgcn_exp2f(m - mnew);
                l *= alpha;
#pragma unroll
                for (int i = 0; i < C::DT; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e) oacc[i][e] *= alpha;
                m = mnew;
            }
            l += psum;
#if ATTN_ABL & 16
            asm volatile("" : "+v"(sacc[0]), "+v"(sacc[1]), "+v"(l));
            if (prof_on) { const unsigned long long now = __builtin_readcyclecounter(); prof_acc[4] += now - t_seg; t_seg = now; }
#endif
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const bf16x8 pf = pack_frag(sacc[st], s);
#pragma unroll
                    for (int dt = 0; dt < C::DT; ++dt)
                        oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_frag_tr<D>(smem, vv[dt], (st * 32 + 16 * s) * C::ROWB), pf, oacc[dt], 0, 0, 0);
                }
        }
#if ATTN_ABL & 16
        if (wave_active) { asm volatile("" : "+v"(oacc[0]), "+v"(oacc[1])); PROF_ADD(5, t_seg); if (prof_on) prof_acc[6] += 1; }
#endif
        // reference semantics for rows whose visible keys are all padding: keep going over the causally hidden tiles
        if (key_mask && causal && kt + 1 == ntiles && ntiles < ntiles_all) {
            if (__syncthreads_or(m == MASK_T)) {
                ntiles = ntiles_all;
                issue(kt + 1, (kt + 1) & 1);  // that stage was last read one iteration ago: free
            }
        }
    }

#if ATTN_ABL & 16
    PROF_ADD(0, t_wg);
    if (prof_on && (abl & 4096) && (blockIdx.x & 63) == 5)  // one workgroup in 64 reports: 16 contended atomics from each of 6 144 would be the whole run time
        for (int i = 0; i < 16; ++i) atomicAdd(&g_prof[i], prof_acc[i]);
#endif
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    if (qvalid) {
        bf16_t* orow = o + ((int64_t)b * S + qg) * ldo + (int64_t)hq * D;
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d = dt * 32 + 8 * g4 + 4 * (lane >> 5);
                u32x2 pk = {pack_bf2(oacc[dt][4 * g4] * inv, oacc[dt][4 * g4 + 1] * inv),
                            pack_bf2(oacc[dt][4 * g4 + 2] * inv, oacc[dt][4 * g4 + 3] * inv)};
                *reinterpret_cast<u32x2*>(orow + d) = pk;
            }
        if (lane < 32) lse[((int64_t)b * Hq + hq) * S + qg] = (m + __builtin_amdgcn_logf(l)) * LN2;
    }
}

// ================================================================================================ forward, lean softmax
// The kernel above spends 10.4 vector instructions per MFMA (profiles/r02_pmc_sq_counters.json) and is bound by what its two waves per SIMD can
// ISSUE, not by the matrix pipe.  This one keeps its shape (4 waves x 32 queries of one head, K / V tiles by LDS-DMA, two workgroups per CU so
// that one wave's vector work runs under the other's MFMAs) and cuts the softmax to exp2 + add + half a pack + half a max3 per score:
//   * the scale log2(e) / sqrt(d) is folded into the Q rows once (bf16(q * c): one more rounding of a bf16 operand; the reference rounds the
//     scores themselves to bf16, twice);
//   * a row's reference m is the INITIAL ACCUMULATOR of its score products (a 16-register tile holding -m): the MFMA chain delivers s * c - m;
//   * the reference moves only when a row's scores outgrow it by 2^8 (before anything is accumulated: in either direction); O, l, the tile at
//     hand and the initial-accumulator tile are then re-based by one factor -- rare, data-dependent, forced explicitly by a test;
//   * maxima with v_max3 from asm (hipcc canonicalises both operands of every fmaxf on MFMA outputs: three instructions per maximum);
//   * masks are a 32-bit word per lane and tile applied with v_bfe_i32 + v_bfi_b32, on boundary tiles only;
//   * rows whose visible keys are ALL padding (reference semantics: uniform attention over all S keys) are known before the first tile -- under the
//     causal mask they are the queries in front of the first real key of a left-padded batch row -- and get score 0 for every existing key; their
//     batch rows walk every tile.
// (An experiment with ONE wave per SIMD, both heads of a kv pair per wave and a hand-placed schedule lives outside the product library in
// tools/experimental/attention_fwd2.hip: its MFMAs hide completely, but a lone wave pays ~10 cycles for every dependent vector instruction and
// nothing covers its per-block prologue / epilogue: 322 us against this kernel's time at the headline shape)
constexpr float LEAN_THR = 8.0f;
constexpr unsigned LEAN_FILL = __builtin_bit_cast(unsigned, MASK_T);
__device__ __forceinline__ float lean_max3(float m, float x, float y) {
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(m), "v"(x), "v"(y));
    return m;
}
__device__ __forceinline__ float lean_set_if(float x, unsigned word, int e, unsigned fill_bits) {
    const unsigned sel = (unsigned)__builtin_amdgcn_sbfe((int)word, (unsigned)e, 1u);  // 0 or ~0
    return __uint_as_float((__float_as_uint(x) & ~sel) | (fill_bits & sel));
}
// the 16 accumulator rows of a lane (bits 0-3, 8-11, 16-19, 24-27 of a 32-key word already shifted by 4 * half-wave) gathered into 16 bits
__device__ __forceinline__ unsigned lean_gather16(unsigned w) { return (w & 0xFu) | ((w >> 4) & 0xF0u) | ((w >> 8) & 0xF00u) | ((w >> 12) & 0xF000u); }

template <int D>
__global__ __launch_bounds__(256, 2) void attn_fwd_lean_kernel(int B, int S, int Hq, int Hkv, const bf16_t* __restrict__ q, int64_t ldq,
                                                               const bf16_t* __restrict__ k, int64_t ldk, const bf16_t* __restrict__ v,
                                                               int64_t ldv, bf16_t* __restrict__ o, int64_t ldo, float* __restrict__ lse,
                                                               const uint8_t* __restrict__ key_mask, int causal, float scale_log2) {
    using C = Cfg<D>;
    [[maybe_unused]] const int abl = causal >> 8;
    causal &= 0xff;
    __shared__ __attribute__((aligned(16))) char smem[4 * C::TILE];  // 2 stages x (K row image, V tr image)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nqb = (S + 127) / 128;
    const int vid = xcd_chunked((int)blockIdx.x, (int)gridDim.x);
    int qb, bh;
    heavy_first(vid, nqb, B * Hq, (abl & 256) ? 1 : (abl & 1024) ? 8 : (abl & 128) ? 32 : (abl & 64) ? 64 : ATTN_HEAD_GROUP, qb, bh);  // ablation bits 8 / 10 / 7 / 6: head by head, groups of 8 / 32 / 64
    const int hq = bh % Hq, b = bh / Hq;
    const int hkv = hq / (Hq / Hkv);
    const int q0 = qb * 128;
    const int qw = q0 + wave * 32;
    const int qg = qw + (lane & 31);
    const bool qvalid = qg < S;
    const bf16_t* kbase = k + (int64_t)b * S * ldk + (int64_t)hkv * D;
    const bf16_t* vbase = v + (int64_t)b * S * ldv + (int64_t)hkv * D;
    const int ntiles_all = (S + 63) / 64;

    // a batch row whose key 0 is padding may hold rows with no visible real key: find its first real key (S: none) and walk every tile
    bool allt = false;
    int first_real = 0;
    if (key_mask) {
        allt = __builtin_amdgcn_readfirstlane((int)key_mask[(int64_t)b * S]) == 0;
        if (allt) {
            first_real = S;
            for (int t = 0; t < ntiles_all; ++t) {
                const int kgl = t * 64 + lane;
                const unsigned long long bits = __ballot(kgl < S && key_mask[(int64_t)b * S + kgl] != 0);
                if (bits) {
                    first_real = t * 64 + (int)__builtin_ctzll(bits);
                    break;
                }
            }
        }
    }
    const bool qrow = allt && (causal ? qg < first_real : first_real >= S);
    const int ntiles = (causal && !allt) ? min(ntiles_all, (q0 + 127) / 64 + 1) : ntiles_all;

    // per-lane parts of the DMA source offsets.  A wave's piece j of an image covers rows RPP (PPW wave + j) + lane / CH, chunk lane % CH; the chunk
    // swizzle of the K row image depends on the row's low bits and so on j (an XOR on the chunk), that of the V image does not -- a whole tile
    // (no row beyond S) then needs one XOR + one add per K piece, nothing per V piece (the rows of piece j go into the scalar offset)
    constexpr int RPW = C::RPP * C::PPW;  // rows per wave
    const int prow = lane / C::CH, pch = lane % C::CH;
    const unsigned dk_row = (unsigned)((wave * RPW + prow) * (int)ldk * 2), dv_row = (unsigned)((wave * RPW + prow) * (int)ldv * 2);
    auto issue = [&](int kt, int stage) {
        char* ks_ = smem + stage * 2 * C::TILE;
        const bf16_t* kp = kbase + (int64_t)kt * 64 * ldk;
        const bf16_t* vp = vbase + (int64_t)kt * 64 * ldv;
        if (S - kt * 64 >= 64) {
            auto kr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(kp), 0, 0x7fffffff, 0x00020000);
            auto vr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(vp), 0, 0x7fffffff, 0x00020000);
#pragma unroll
            for (int j = 0; j < C::PPW; ++j) {
                const int row = (wave * C::PPW + j) * C::RPP + prow;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(kr, LDS_PTR(ks_ + (wave * C::PPW + j) * 1024), 16, dk_row + (unsigned)(swz_row<D>(pch, row) << 4),
                                                         (int)(j * C::RPP * ldk * 2), 0, 0);
            }
#pragma unroll
            for (int j = 0; j < C::PPW; ++j) {
                const int row = (wave * C::PPW + j) * C::RPP + prow;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(vr, LDS_PTR(ks_ + C::TILE + (wave * C::PPW + j) * 1024), 16, dv_row + (unsigned)(swz_tr<D>(pch, row) << 4),
                                                         (int)(j * C::RPP * ldv * 2), 0, 0);
            }
        } else {
            dma_tile<D, IMG_ROW>(kp, ldk, S - kt * 64, ks_, wave, lane);
            dma_tile<D, IMG_TR>(vp, ldv, S - kt * 64, ks_ + C::TILE, wave, lane);
        }
    };
    const LaneOff<D> lo = lane_offsets<D>(lane);
    const unsigned lds0 = (unsigned)(unsigned long long)LDS_PTR(smem);
    issue(0, 0);  // first tile first, then the query rows: their latencies overlap
    bf16x8 qf[C::KS];
    if (abl & 32) {  // ablation: a query row per lane straight from memory (64 lanes in 64 rows: the slow access shape)
        load_rows_frag<D>(q + (int64_t)b * S * ldq + (int64_t)hq * D, ldq, qg, qvalid, lane, qf);
    } else {
        // the 128 query rows as two 64-row row images in stage 1 (free until tile 1 is requested behind the loop's first barrier), by LDS-DMA: whole
        // rows per request instead of a row per lane; the fragments are then A-shaped reads of a row image, which is what a B operand of S^T = K Q^T is
        const bf16_t* qblk = q + ((int64_t)b * S + q0) * ldq + (int64_t)hq * D;
        dma_tile<D, IMG_ROW>(qblk, ldq, S - q0, smem + 2 * C::TILE, wave, lane);
        dma_tile<D, IMG_ROW>(qblk + 64 * ldq, ldq, S - q0 - 64, smem + 3 * C::TILE, wave, lane);
        wait_vmcnt<0>();
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) qf[ks] = frag_rows<D>(smem + (2 + (wave >> 1)) * C::TILE, (wave & 1) * 32, ks, lane);
    }
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {  // q * log2(e) / sqrt(d), rounded to bf16 again
        u32x4 w = __builtin_bit_cast(u32x4, qf[ks]);
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = pack_bf2(__uint_as_float(w[e] << 16) * scale_log2, __uint_as_float(w[e] & 0xffff0000u) * scale_log2);
        qf[ks] = __builtin_bit_cast(bf16x8, w);
    }
    // key-padding byte of this lane's key in the NEXT tile: loaded one tile ahead so its latency hides behind a whole tile
    // (unconditional, address clamped to the row: a guarded load becomes a branch whose join the compiler serves with `s_waitcnt vmcnt(0)`
    // right behind the load, i.e. behind the tile request in front of it; bits of keys beyond S are cleared by `kbits` below)
    const uint8_t* mrow = key_mask ? key_mask + (int64_t)b * S : nullptr;
    uint8_t mk = mrow ? mrow[min(lane, S - 1)] : (uint8_t)0;
    unsigned tri16 = 0;  // causal mask of a diagonal 32 x 32 sub-tile: bit ee set <=> key row acc_row(ee) lies behind this lane's query
#pragma unroll
    for (int ee = 0; ee < 16; ++ee) tri16 |= (acc_row(ee, lane) > (lane & 31) ? 1u : 0u) << ee;

    f32x16 oacc[C::DT], ninit;
#pragma unroll
    for (int i = 0; i < C::DT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[i][e] = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) ninit[e] = 0.f;
    float mref = 0.f, l = 0.f;
    [[maybe_unused]] const bool prof_on = threadIdx.x == 0;
    [[maybe_unused]] unsigned long long prof_acc[16] = {};
    [[maybe_unused]] const unsigned long long t_wg = PROF_T();

    for (int kt = 0; kt < ntiles; ++kt) {
        [[maybe_unused]] const unsigned long long t_sync = PROF_T();
        // this wave's share of tile kt (requested a whole tile ago) and the mask byte have landed.  Explicit: the compiler orders LDS-DMA only against
        // ds_read_tr INTRINSICS, and the V fragments below are read from asm statements
        wait_vmcnt<0>();
        __syncthreads();
        PROF_ADD(1, t_sync);
        // bit per key of this tile: 1 = a real token that exists (consumed before the next request, so its wait costs nothing)
        const int nv = S - kt * 64;
        unsigned long long kbits = nv >= 64 ? ~0ull : ((1ull << nv) - 1ull);
        if (mrow) kbits &= __ballot(mk != 0);
        [[maybe_unused]] const unsigned long long t_iss = PROF_T();
        // the next tile's 2 PPW pieces: a whole tile's are spread over the S product below, one behind every second MFMA (requested in one burst, the
        // four waves' 32 pieces queue at the CU's address unit and each wave sits ~75 cycles per piece in front of it); a ragged last tile keeps the burst
        const bool more = kt + 1 < ntiles;
        const bool spread = more && S - (kt + 1) * 64 >= 64 && !(abl & 512);
        if (more && !spread) issue(kt + 1, (kt + 1) & 1);
        const auto nkr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(kbase + (int64_t)(kt + 1) * 64 * ldk), 0, 0x7fffffff, 0x00020000);
        const auto nvr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(vbase + (int64_t)(kt + 1) * 64 * ldv), 0, 0x7fffffff, 0x00020000);
        char* const nst = smem + ((kt + 1) & 1) * 2 * C::TILE;
        auto piece = [&](auto i_) {
            constexpr int i = i_.value;
            if constexpr (i < C::PPW) {
                const int row = (wave * C::PPW + i) * C::RPP + prow;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(nkr, LDS_PTR(nst + (wave * C::PPW + i) * 1024), 16, dk_row + (unsigned)(swz_row<D>(pch, row) << 4), (int)(i * C::RPP * ldk * 2), 0, 0);
            } else {
                constexpr int j = i - C::PPW;
                const int row = (wave * C::PPW + j) * C::RPP + prow;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(nvr, LDS_PTR(nst + C::TILE + (wave * C::PPW + j) * 1024), 16, dv_row + (unsigned)(swz_tr<D>(pch, row) << 4), (int)(j * C::RPP * ldv * 2), 0, 0);
            }
        };
        PROF_ADD(2, t_iss);
        if (mrow) mk = mrow[min((kt + 1) * 64 + lane, S - 1)];
        [[maybe_unused]] unsigned long long t_seg = PROF_T();
        if (prof_on) prof_acc[7] += 1;
        const int koff = (kt & 1) * 2 * C::TILE, voff = koff + C::TILE;
        // a wave whose 32 queries all precede this tile has nothing visible here
        if (causal && !allt && kt * 64 > qw + 31) continue;
        int vkx[C::KS], vv[C::DT];
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) vkx[ks] = (lo.row + koff) ^ (ks << 5);
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) vv[dt] = lo.col[dt] + voff;
        f32x16 sacc[2];
        static_assert(2 * C::PPW == C::KS, "one next-tile piece behind every second MFMA of the S product");
        // two straight-line copies: a branch per piece would keep hipcc from requesting the K fragments ahead.  (K fragments from asm statements, four
        // products ahead of their MFMA -- hipcc does not move a visible LDS read across an LDS-DMA request, so the spread copy reads only one pair
        // ahead -- measured 254 us against 245 for this form: the compiler's own placement of the reads is the better one; attn_common.h keeps the helper.)
        auto s_product = [&](auto spread_c) {
            static_for<2>([&](auto st_) {
                constexpr int st = st_.value;
                static_for<C::KS>([&](auto ks_) {
                    constexpr int ks = ks_.value, m = st * C::KS + ks;
                    sacc[st] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_frag(smem, vkx[ks], st * 32 * C::ROWB), qf[ks], ks == 0 ? ninit : sacc[st], 0, 0, 0);
                    if constexpr (decltype(spread_c)::value && m % 2 == 1) piece(std::integral_constant<int, m / 2>{});
                });
            });
        };
        if (spread) s_product(std::true_type{});
        else s_product(std::false_type{});
        TrHalves ft[2][C::DT];
        auto issue_v = [&](auto g_) {
            constexpr int g = g_.value, imm = 16 * g * C::ROWB;
            static_for<C::DT>([&](auto dt_) { tr_issue<imm, imm + 8 * C::ROWB>(ft[g & 1][dt_.value], lds0 + vv[dt_.value], lds0 + vv[dt_.value]); });
        };
        issue_v(std::integral_constant<int, 0>{});  // the first 16 keys' V fragments travel behind the softmax
#if ATTN_ABL & 16
        asm volatile("" : "+v"(sacc[0]), "+v"(sacc[1]));
        if (prof_on) { const unsigned long long now = __builtin_readcyclecounter(); prof_acc[3] += now - t_seg; t_seg = now; }
#endif
        // masks only on the diagonal / tail / padded tiles (wave-uniform test)
        if ((kbits != ~0ull) || allt || (causal && kt * 64 + 63 > qw)) {
            const unsigned long long exist = nv >= 64 ? ~0ull : ((1ull << nv) - 1ull);
            unsigned m16[2], x16[2];
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const unsigned sh = 4u * (unsigned)(lane >> 5);
                const unsigned vis = lean_gather16((st ? (unsigned)(kbits >> 32) : (unsigned)kbits) >> sh);
                x16[st] = lean_gather16((st ? (unsigned)(exist >> 32) : (unsigned)exist) >> sh);
                const int x = qw - kt * 64 - st * 32;  // the wave's first query against the sub-tile's first key (a multiple of 32): 0 = diagonal
                const unsigned cz = (!causal || x >= 32) ? 0u : (x == 0 ? tri16 : 0xFFFFu);
                m16[st] = (~vis & 0xFFFFu) | cz;
            }
            const unsigned mall = m16[0] | (m16[1] << 16), eall = x16[0] | (x16[1] << 16);
            const unsigned mset = qrow ? ~eall : mall;  // -> the fill value
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int ee = 0; ee < 16; ++ee) sacc[st][ee] = lean_set_if(sacc[st][ee], mset, 16 * st + ee, LEAN_FILL);
            if (allt) {  // rows whose visible keys are all padding: every existing key counts alike
                const unsigned zset = qrow ? eall : 0u;
#pragma unroll
                for (int st = 0; st < 2; ++st)
#pragma unroll
                    for (int ee = 0; ee < 16; ++ee) sacc[st][ee] = lean_set_if(sacc[st][ee], zset, 16 * st + ee, 0u);
            }
        }
        // row maxima of the tile (the two half-waves hold the two halves of a query's keys)
        float t0 = lean_max3(sacc[0][0], sacc[0][1], sacc[0][2]), t1 = lean_max3(sacc[1][0], sacc[1][1], sacc[1][2]);
#pragma unroll
        for (int e = 3; e < 15; e += 2) {
            t0 = lean_max3(t0, sacc[0][e], sacc[0][e + 1]);
            t1 = lean_max3(t1, sacc[1][e], sacc[1][e + 1]);
        }
        float t = lean_max3(t0, t1, sacc[0][15]);
        t = lean_max3(t, sacc[1][15], __shfl_xor(lean_max3(t, sacc[1][15], sacc[1][15]), 32, 64));
        // up: always.  Down: only while nothing is accumulated, and never onto the fill value
        const bool want = t > LEAN_THR || (l == 0.f && t < -LEAN_THR && t > 0.5f * MASK_T);
        if (__builtin_expect(__any(want), 0)) {
            const float delta = want ? t : 0.f;
            const float alpha = __builtin_amdgcn_exp2f(-delta);
            mref += delta;
            l *= alpha;
#pragma unroll
            for (int i = 0; i < C::DT; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) oacc[i][e] *= alpha;
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float x = sacc[st][e];  // a masked score stays AT the fill value
                    sacc[st][e] = x < 0.5f * MASK_T ? x : x - delta;
                }
            const float nm = -mref;
#pragma unroll
            for (int e = 0; e < 16; ++e) ninit[e] = nm;
        }
        float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
                sacc[st][e] = __builtin_amdgcn_exp2f(sacc[st][e]);
                sacc[st][e + 1] = __builtin_amdgcn_exp2f(sacc[st][e + 1]);
                ps0 += sacc[st][e];
                ps1 += sacc[st][e + 1];
            }
        l += ps0 + ps1;
#if ATTN_ABL & 16
        asm volatile("" : "+v"(sacc[0]), "+v"(sacc[1]), "+v"(l));
        if (prof_on) { const unsigned long long now = __builtin_readcyclecounter(); prof_acc[4] += now - t_seg; t_seg = now; }
#endif
        // V fragments from asm statements (attn_common.h: an intrinsic read would draw `s_waitcnt vmcnt(0)` -- the whole latency of the tile
        // just requested -- into the middle of this one); group g = 16 keys, its successor is requested before it is consumed
        static_for<4>([&](auto g_) {
            constexpr int g = g_.value;
            if constexpr (g + 1 < 4) issue_v(std::integral_constant<int, g + 1>{});
            const bf16x8 pf = pack_frag(sacc[g >> 1], g & 1);
            static_for<C::DT>([&](auto dt_) {
                constexpr int dt = dt_.value;
                constexpr int younger = 2 * (C::DT - 1 - dt) + (g + 1 < 4 ? 2 * C::DT : 0);
                oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_wait<younger>(ft[g & 1][dt]), pf, oacc[dt], 0, 0, 0);
            });
        });
#if ATTN_ABL & 16
        asm volatile("" : "+v"(oacc[0]), "+v"(oacc[1]));
        PROF_ADD(5, t_seg);
        if (prof_on) prof_acc[6] += 1;
#endif
    }
#if ATTN_ABL & 16
    PROF_ADD(0, t_wg);
    if (prof_on && (abl & 4096) && (blockIdx.x & 63) == 5)  // one workgroup in 64 reports
        for (int i = 0; i < 16; ++i) atomicAdd(&g_prof[i], prof_acc[i]);
#endif

    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    if (abl & 32) {
        // whole 16-byte pieces: one v_permlane32_swap per word pairs the half-waves' 8-byte pieces of a row
        bf16_t* orow = o + ((int64_t)b * S + qg) * ldo + (int64_t)hq * D + 8 * (lane >> 5);
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                const int r0 = 8 * gp;
                const unsigned a0 = pack_bf2(oacc[dt][r0] * inv, oacc[dt][r0 + 1] * inv), a1 = pack_bf2(oacc[dt][r0 + 2] * inv, oacc[dt][r0 + 3] * inv);
                const unsigned b0 = pack_bf2(oacc[dt][r0 + 4] * inv, oacc[dt][r0 + 5] * inv), b1 = pack_bf2(oacc[dt][r0 + 6] * inv, oacc[dt][r0 + 7] * inv);
                const auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false), s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                const u32x4 w = {s0[0], s1[0], s0[1], s1[1]};
                if (qvalid) *reinterpret_cast<u32x4*>(orow + dt * 32 + 16 * gp) = w;
            }
    } else {
        // through LDS (the tile stages are dead): [128 rows][2 D bytes], the 16-byte piece index XORed with the row so that the 8-byte writes of a
        // half-wave's 32 rows spread over the banks; then every store instruction of a wave covers whole rows (a row per lane is the 17x slower shape)
        constexpr int RB = 2 * D, PCS = RB / 16;  // row bytes, 16-byte pieces per row
        __syncthreads();
        {
            const int row = wave * 32 + (lane & 31), hb = lane >> 5;
#pragma unroll
            for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int chunk = dt * 8 + 2 * g4 + hb;  // features 4 chunk .. 4 chunk + 3 (8 bytes)
                    const u32x2 pk = {pack_bf2(oacc[dt][4 * g4] * inv, oacc[dt][4 * g4 + 1] * inv), pack_bf2(oacc[dt][4 * g4 + 2] * inv, oacc[dt][4 * g4 + 3] * inv)};
                    *reinterpret_cast<u32x2*>(smem + row * RB + (((chunk >> 1) ^ (row & (PCS - 1))) << 4) + (chunk & 1) * 8) = pk;
                }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 128 * PCS / 256; ++it) {
            const int i = it * 256 + (int)threadIdx.x, row = i / PCS, pc = i % PCS;
            const u32x4 w = *reinterpret_cast<const u32x4*>(smem + row * RB + ((pc ^ (row & (PCS - 1))) << 4));
            if (q0 + row < S) *reinterpret_cast<u32x4*>(o + ((int64_t)b * S + q0 + row) * ldo + (int64_t)hq * D + pc * 8) = w;
        }
    }
    // a row whose visible keys are all padding reports the fill value as its maximum, as the first-generation kernel does
    if (qvalid && lane < 32) lse[((int64_t)b * Hq + hq) * S + qg] = ((qrow ? MASK_T : mref) + __builtin_amdgcn_logf(l)) * LN2;
}

// ================================================================================================ backward
// delta[b,h,q] = sum_d dO[q,d] * O[q,d].  HBM-bound: 16-byte loads, D/8 lanes per (token, head) row, so one wave covers
// 512/D adjacent heads of a token (contiguous in the token-major operands) per 1-KiB load instruction.
template <int D>
__global__ __launch_bounds__(256) void attn_delta_kernel(int B, int S, int Hq, const bf16_t* __restrict__ o, int64_t ldo,
                                                         const bf16_t* __restrict__ d_o, int64_t lddo, float* __restrict__ delta,
                                                         const float* __restrict__ lse, float* __restrict__ nl2, float* __restrict__ ndl) {
    // nl2 / ndl (optional): -lse * log2(e) and -delta, the INITIAL ACCUMULATORS of the dK/dV pass's S and dP products (its lean form)
    constexpr int LPR = D / 8, RPW = 64 / LPR;  // lanes per row, rows (heads) per wave-instruction
    const int lane = threadIdx.x & 63, sub = lane / LPR, li = lane % LPR;
    const int hgroups = (Hq + RPW - 1) / RPW;
    const int64_t total = (int64_t)B * S * hgroups;
    for (int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); item < total; item += (int64_t)gridDim.x * 4) {
        const int64_t tok = item / hgroups;
        const int h = (int)(item - tok * hgroups) * RPW + sub;
        float s = 0.f;
        if (h < Hq) {
            const u32x4 a = *reinterpret_cast<const u32x4*>(o + tok * ldo + (int64_t)h * D + li * 8);
            const u32x4 g = *reinterpret_cast<const u32x4*>(d_o + tok * lddo + (int64_t)h * D + li * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                s += __uint_as_float(a[e] << 16) * __uint_as_float(g[e] << 16) + __uint_as_float(a[e] & 0xffff0000u) * __uint_as_float(g[e] & 0xffff0000u);
        }
#pragma unroll
        for (int off = LPR / 2; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
        if (li == 0 && h < Hq) {
            const int64_t bb = tok / S, sq = tok - bb * S;
            const int64_t di = (bb * Hq + h) * S + sq;
            delta[di] = s;
            if (nl2) {
                nl2[di] = -lse[di] * LOG2E;
                ndl[di] = -s;
            }
        }
    }
}

// ---- dQ pass: query on the lane (the forward's orientation).  Per 64-key tile, for each 32-key sub-tile st:
//   A(st):  S^T = K Q^T ;  dP^T = V dO^T            (A operand: K / V rows from LDS, B operand: Q / dO rows, owned AGPRs)
//   B(st):  P^T = exp2(S^T*c - lse2[q]) ;  dS^T = P^T * (dP^T - delta[q]) * scale        (VALU, lane-local row constants)
//   C(st):  dQ^T += K^T dS^T                        (A operand: transposed reads of the K tile, B operand: packed dS^T)
// One wave per SIMD, so the overlap is written out: the MFMAs run in the order A(0) A(1) C(0) C(1), B(0) is spread over the
// MFMAs of A(1) and B(1) over those of C(0), and fragment reads run PD MFMAs ahead of their consumer.
template <int D>
__global__ __launch_bounds__(256, 1) void attn_bwd_dq_kernel(int B, int S, int Hq, int Hkv, const bf16_t* __restrict__ q, int64_t ldq,
                                                             const bf16_t* __restrict__ k, int64_t ldk, const bf16_t* __restrict__ v,
                                                             int64_t ldv, const bf16_t* __restrict__ d_o, int64_t lddo,
                                                             const float* __restrict__ lse, const float* __restrict__ delta,
                                                             bf16_t* __restrict__ dq, int64_t lddq, const uint8_t* __restrict__ key_mask,
                                                             int causal, float scale, float scale_log2, int bpw) {
    using C = Cfg<D>;
    constexpr int KS = C::KS, DT = C::DT;
    constexpr int NA = 2 * KS, NC = 2 * DT, NG = 2 * NA + 2 * NC;  // MFMAs of one A part, one C part, one key tile
    constexpr int OWNED = 16 * DT + 8 * KS, QF0 = 16 * DT, OF0 = QF0 + 4 * KS;  // dQ^T tiles | Q rows | dO rows
    constexpr int RING = ATTN_RING, PD = ATTN_PD;
    // LDS: NST stages x (K [unified] | V rows)  or, D = 64,  (K rows | K transposed | V rows);  then the key-mask words
    constexpr bool UNI = C::UNI;
    constexpr int NIMG = UNI ? 2 : 3, NST = 3, STAGE = NIMG * C::TILE, PIECES = NIMG * C::PPW, VIMG = (NIMG - 1) * C::TILE;
    const int abl = causal >> 8;  // profiling only: 1 = no DMA after the first tiles
    causal &= 0xff;
    __shared__ __attribute__((aligned(16))) char smem[NST * STAGE + ATTN_MAX_TILES * 8];
    unsigned long long* kmw = reinterpret_cast<unsigned long long*>(smem + NST * STAGE);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // one workgroup = `bpw` consecutive 128-query blocks of one (batch, head), heaviest (latest) first; adjacent virtual ids (the
    // chunks of a head, the two query heads of a kv head) share an XCD
    const int nqb = (S + 127) / 128, nchunk = (nqb + bpw - 1) / bpw;
    const int vid = (abl >> 8) & 1 ? (int)blockIdx.x : xcd_chunked((int)blockIdx.x, (int)gridDim.x);
    const int chunk = vid % nchunk, bh = vid / nchunk;
    const int hq = bh % Hq, b = bh / Hq;
    const int hkv = hq / (Hq / Hkv);
    const int qb_hi = nqb - chunk * bpw, qb_lo = max(0, qb_hi - bpw);
    const int ntiles_all = (S + 63) / 64;
    auto tiles_of = [&](int qb) { return causal ? min(ntiles_all, (qb * 128 + 127) / 64 + 1) : ntiles_all; };
    const bf16_t* kbase = k + (int64_t)b * S * ldk + (int64_t)hkv * D;
    const bf16_t* vbase = v + (int64_t)b * S * ldv + (int64_t)hkv * D;

    // ---- K/V tile stream over (block, tile), NST-deep: the tile two ahead is in flight while one is being consumed
    int iqb = qb_hi - 1, ikt = 0, istage = 0;
    auto issue_next = [&]() -> bool {
        if (iqb < qb_lo) return false;
        char* st_ = smem + istage * STAGE;
        const bf16_t* kp = kbase + (int64_t)ikt * 64 * ldk;
        if constexpr (UNI) {
            dma_tile<D, IMG_UNI>(kp, ldk, S - ikt * 64, st_, wave, lane);
        } else {
            dma_tile<D, IMG_ROW>(kp, ldk, S - ikt * 64, st_, wave, lane);
            dma_tile<D, IMG_TR>(kp, ldk, S - ikt * 64, st_ + C::TILE, wave, lane);
        }
        dma_tile<D, IMG_ROW>(vbase + (int64_t)ikt * 64 * ldv, ldv, S - ikt * 64, st_ + VIMG, wave, lane);
        if (++ikt == tiles_of(iqb)) { ikt = 0; --iqb; }
        istage = istage == NST - 1 ? 0 : istage + 1;
        return true;
    };
    int inflight = 0;
    inflight += issue_next();
    inflight += issue_next();

    // key-padding bits of every 64-key tile of this batch row (1 = real token), once per workgroup
    if (key_mask)
        for (int t = wave; t < ntiles_all; t += 4) {
            const int kgl = t * 64 + lane;
            const unsigned long long bits = __ballot(kgl < S && key_mask[(int64_t)b * S + kgl] != 0);
            if (lane == 0) kmw[t] = bits;
        }

    const LaneOff<D> lo = lane_offsets<D>(lane);
    const unsigned lds0 = (unsigned)(unsigned long long)LDS_PTR(smem);
    int cstage = 0;
    for (int qb = qb_hi - 1; qb >= qb_lo; --qb) {
        const int q0 = qb * 128;
        const int qg = q0 + wave * 32 + (lane & 31);
        const bool qvalid = qg < S;
        {
            bf16x8 tq[KS], to[KS];
            load_rows_frag<D>(q + (int64_t)b * S * ldq + (int64_t)hq * D, ldq, qg, qvalid, lane, tq);
            load_rows_frag<D>(d_o + (int64_t)b * S * lddo + (int64_t)hq * D, lddo, qg, qvalid, lane, to);
            owned_zero<OWNED, 0, 16 * DT>();
            static_for<KS>([&](auto ks) { owned_write4<OWNED, QF0 + 4 * ks.value>(tq[ks.value]); });
            static_for<KS>([&](auto ks) { owned_write4<OWNED, OF0 + 4 * ks.value>(to[ks.value]); });
        }
        const float lse2 = qvalid ? lse[((int64_t)b * Hq + hq) * S + qg] * LOG2E : 0.f;
        const float dlt = qvalid ? delta[((int64_t)b * Hq + hq) * S + qg] : 0.f;
        const int ntiles = tiles_of(qb);
        for (int kt = 0; kt < ntiles; ++kt) {
            // this wave's pieces of the tile have landed (a younger tile may stay in flight), then everybody's
            if (inflight >= 2) wait_vmcnt<PIECES>(); else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            --inflight;
            if (!(abl & 1)) inflight += issue_next();  // into the stage that was consumed before this barrier
            const int kroff = cstage * STAGE, ktoff = UNI ? kroff : kroff + C::TILE, vroff = kroff + VIMG;
            cstage = cstage == NST - 1 ? 0 : cstage + 1;
            unsigned long long kbits = ~0ull;
            if (key_mask) {
                const unsigned long long w = kmw[kt];
                kbits = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(w >> 32)) << 32) | (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)w);
            }
            const bool wave_active = !causal || (kt * 64 <= q0 + wave * 32 + 31);
            if (!wave_active) continue;
            const bool boundary = (kbits != ~0ull) || (kt * 64 + 64 > S) || (causal && kt * 64 + 63 > q0 + wave * 32);
            int vkx[KS], vvx[KS], vkt[DT];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                vkx[ks] = ((UNI ? lo.rowu : lo.row) + kroff) ^ (ks << 5);
                vvx[ks] = (lo.row + vroff) ^ (ks << 5);
            }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) vkt[dt] = (UNI ? lo.colu[dt] : lo.col[dt]) + ktoff;

            auto tile_body = [&](auto bc) {
                constexpr bool BOUNDARY = decltype(bc)::value;
                bf16x8 f[RING] = {};
                TrHalves ft[RING] = {};
                f32x16 sacc[2], pacc[2];
                unsigned dsw[2][8] = {};  // packed dS^T: words 4s..4s+3 of sub-tile st are the B operand of k-step s
                float dsv[16];
                auto load = [&](auto gc) {
                    constexpr int g = gc.value;
                    if constexpr (ATTN_ABL & 1) return;
                    if constexpr (g < 2 * NA) {
                        constexpr int st = g / NA, ks = (g % NA) / 2, which = g % 2;
                        f[g % RING] = lds_frag(smem, which ? vvx[ks] : vkx[ks], st * 32 * C::ROWB);
                    } else if constexpr (g < NG) {
                        constexpr int h = g - 2 * NA, st = h / NC, sd = (h % NC) / DT, dt = h % DT, imm = (st * 32 + 16 * sd) * C::ROWB;
                        tr_issue<imm, imm + 8 * C::ROWB>(ft[g % RING], lds0 + vkt[dt], lds0 + (UNI ? vkt[dt] ^ 0x20 : vkt[dt]));
                    }
                };
                auto element = [&](auto stc, auto ec) {
                    constexpr int st = stc.value, e = ec.value;
                    if constexpr (ATTN_ABL & 2) return;
                    float t = fmaf(sacc[st][e], scale_log2, -lse2);
                    if constexpr (BOUNDARY) {
                        const int kl = st * 32 + acc_row(e, lane);
                        const int kg = kt * 64 + kl;
                        const bool masked = (causal && kg > qg) || !((kbits >> kl) & 1ull) || kg >= S;
                        t = masked ? -INFINITY : t;  // exp2(-inf) = 0: no branch around the exp
                    }
                    dsv[e] = __builtin_amdgcn_exp2f(t) * (pacc[st][e] - dlt);  // dS^T / scale (the factor is applied to dQ once)
                    if constexpr (e % 2 == 1) dsw[st][e / 2] = pack_bf2(dsv[e - 1], dsv[e]);
                };
                static_for<PD>([&](auto g) { load(g); });
                static_for<NG>([&](auto gc) {
                    constexpr int g = gc.value;
                    load(std::integral_constant<int, g + PD>{});
                    if constexpr (g < 2 * NA) {
                        constexpr int st = g / NA, ks = (g % NA) / 2, which = g % 2;
                        if constexpr ((ATTN_ABL & 8) && ks > 0) {
                        } else if constexpr (which == 0) mfma_ownedB<OWNED, QF0 + 4 * ks, ks == 0>(sacc[st], f[g % RING]);
                        else mfma_ownedB<OWNED, OF0 + 4 * ks, ks == 0>(pacc[st], f[g % RING]);
                    } else if constexpr (!(ATTN_ABL & 4)) {
                        constexpr int h = g - 2 * NA, st = h / NC, sd = (h % NC) / DT, dt = h % DT;
                        const u32x4 w = {dsw[st][4 * sd], dsw[st][4 * sd + 1], dsw[st][4 * sd + 2], dsw[st][4 * sd + 3]};
                        constexpr int younger = NG - 1 - g < PD ? NG - 1 - g : PD;
                        mfma_owned<OWNED, 16 * dt>(tr_wait<2 * younger>(ft[g % RING]), __builtin_bit_cast(bf16x8, w));
                    }
                    // B(0) rides on the MFMAs of A(1), B(1) on those of C(0)
                    if constexpr (g == NA) tiles_settle(sacc[0], pacc[0]);
                    if constexpr (g == 2 * NA) tiles_settle(sacc[1], pacc[1]);
                    if constexpr (g >= NA && g < 2 * NA) {
                        constexpr int m = g - NA, per = 16 / NA;
                        static_for<per>([&](auto i) { element(std::integral_constant<int, 0>{}, std::integral_constant<int, m * per + i.value>{}); });
                    } else if constexpr (g >= 2 * NA && g < 2 * NA + NC) {
                        constexpr int m = g - 2 * NA, per = 16 / NC;
                        static_for<per>([&](auto i) { element(std::integral_constant<int, 1>{}, std::integral_constant<int, m * per + i.value>{}); });
                    }
                    // keep this step's VALU where it is written: left alone, hipcc gathers a whole B phase into one MFMA gap
                    __builtin_amdgcn_sched_barrier(0);
                });
            };
            if (boundary) tile_body(std::true_type{});
            else tile_body(std::false_type{});
        }
        owned_settle<OWNED>();
        bf16_t* row = dq + ((int64_t)b * S + qg) * lddq + (int64_t)hq * D;
        static_for<DT * 4>([&](auto i) {
            constexpr int dt = i.value / 4, g4 = i.value % 4, r = 16 * dt + 4 * g4;
            const int d = dt * 32 + 8 * g4 + 4 * (lane >> 5);
            const u32x2 pk = {pack_bf2(owned_read<OWNED, r>() * scale, owned_read<OWNED, r + 1>() * scale),
                              pack_bf2(owned_read<OWNED, r + 2>() * scale, owned_read<OWNED, r + 3>() * scale)};
            if (qvalid && !(abl & 2)) *reinterpret_cast<u32x2*>(row + d) = pk;
        });
    }
}

// ---- dK/dV pass: KEY on the lane.  One workgroup = 128 keys of one (batch, kv head); each wave owns 32 keys and keeps
// dK^T, dV^T for them in (owned) accumulators while the workgroup sweeps the group's query heads x 64-query tiles.  Per
// 32-query sub-tile st:
//   A(st):  S = Q K^T ;  dP = dO V^T                (A operand: Q / dO rows from LDS, B operand: K / V rows, owned AGPRs)
//   B(st):  P = exp2(S*c - lse2[q]) ;  dS = P*(dP - delta[q])*scale                      (row constants come from LDS)
//   C(st):  dV^T += dO^T P ;  dK^T += Q^T dS        (A operand: transposed reads of the dO / Q tile, B: packed P / dS)
// Same schedule as the dQ pass: MFMAs in the order A(0) A(1) C(0) C(1), B(0) under A(1), B(1) under C(0), reads PD ahead.
// LEAN: the row constants are INITIAL ACCUMULATORS (`lse` / `delta` then point at -lse * log2(e) and -delta, written by the delta kernel) and the
// scale log2(e) / sqrt(d) is folded into this wave's K rows once per key block (bf16(k * c): one more rounding of a bf16 operand), so the S chain
// delivers the exponent of P and the dP chain dP - delta: the B phase shrinks from six vector instructions per score (scale the constant, fma, exp2,
// subtract, multiply, pack) to three (exp2, multiply, pack) -- the pass is bound by what its lone wave has to issue.
template <int D, bool LEAN>
__global__ __launch_bounds__(256, 1) void attn_bwd_dkv_kernel(int B, int S, int Hq, int Hkv, const bf16_t* __restrict__ q, int64_t ldq,
                                                              const bf16_t* __restrict__ k, int64_t ldk, const bf16_t* __restrict__ v,
                                                              int64_t ldv, const bf16_t* __restrict__ d_o, int64_t lddo,
                                                              const float* __restrict__ lse, const float* __restrict__ delta,
                                                              bf16_t* __restrict__ dk, int64_t lddk, bf16_t* __restrict__ dv, int64_t lddv,
                                                              const uint8_t* __restrict__ key_mask, int causal, float scale, float scale_log2,
                                                              int bpw, bf16_t* __restrict__ ds_out) {
    // ds_out != nullptr: this pass also leaves dS / scale (bf16, the values its own dK^T product consumes) in a scratch buffer, so that
    // dQ = scale * dS K becomes ONE product (attn_bwd_dq_spill_kernel) instead of the dQ pass's three (it recomputes S and dP):
    // 5 products for the whole backward instead of 7.
    using C = Cfg<D>;
    constexpr int KS = C::KS, DT = C::DT;
    constexpr int NA = 2 * KS, NC = 4 * DT, NG = 2 * NA + 2 * NC;
    constexpr int OWNED = 32 * DT + 8 * KS, DK0 = 16 * DT, KF0 = 32 * DT, VF0 = KF0 + 4 * KS;  // dV^T | dK^T | K rows | V rows
    constexpr int RING = ATTN_RING, PD = ATTN_PD;
    // LDS: NST stages x (Q | dO [unified images])  or, D = 64,  (Q rows | Q transposed | dO rows | dO transposed);
    // then NST x 64 x (lse, delta)
    constexpr bool UNI = C::UNI;
    constexpr int NIMG = UNI ? 2 : 4, NST = 3, STAGE = NIMG * C::TILE, PIECES = NIMG * C::PPW;
    constexpr int OIMG = (UNI ? 1 : 2) * C::TILE;  // offset of the dO image(s) inside a stage
    const int abl = causal >> 8;  // profiling only: 1 = no DMA after the first tiles
    causal &= 0xff;
    __shared__ __attribute__((aligned(16))) char smem[NST * STAGE + NST * 512];
    char* rowc = smem + NST * STAGE;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // one workgroup = `bpw` consecutive 128-key blocks of one (batch, kv head)
    const int nkb = (S + 127) / 128, nchunk = (nkb + bpw - 1) / bpw;
    const int chunk = blockIdx.x % nchunk, bh = blockIdx.x / nchunk;
    const int hkv = bh % Hkv, b = bh / Hkv;
    const int rep = Hq / Hkv;
    const int kb_lo = chunk * bpw, kb_hi = min(nkb, kb_lo + bpw);
    const int nqt_all = (S + 63) / 64;
    auto qt0_of = [&](int kb) { return causal ? kb * 2 : 0; };  // first query tile that can see any key of the block

    // ---- Q / dO tile stream over (key block, head, query tile), NST-deep
    // (head and query tile advance as counters: a scalar division per tile was a tenth of the tile's time)
    int ikb = kb_lo, ih = 0, iq = 0, istage = 0;  // next tile to request: key block, query head within the kv group, query tile counted from the block's first
    auto issue_next = [&]() -> bool {
        if (ikb < kb_hi && iq >= nqt_all - qt0_of(ikb)) { iq = 0; ++ih; }
        if (ikb < kb_hi && ih >= rep) { ih = 0; ++ikb; }
        if (ikb >= kb_hi) return false;
        const int hq = hkv * rep + ih;
        const int qt = qt0_of(ikb) + iq;
        char* st_ = smem + istage * STAGE;
        const bf16_t* qb_ = q + ((int64_t)b * S + (int64_t)qt * 64) * ldq + (int64_t)hq * D;
        const bf16_t* ob_ = d_o + ((int64_t)b * S + (int64_t)qt * 64) * lddo + (int64_t)hq * D;
        if constexpr (UNI) {
            dma_tile<D, IMG_UNI>(qb_, ldq, S - qt * 64, st_, wave, lane);
            dma_tile<D, IMG_UNI>(ob_, lddo, S - qt * 64, st_ + OIMG, wave, lane);
        } else {
            dma_tile<D, IMG_ROW>(qb_, ldq, S - qt * 64, st_, wave, lane);
            dma_tile<D, IMG_TR>(qb_, ldq, S - qt * 64, st_ + C::TILE, wave, lane);
            dma_tile<D, IMG_ROW>(ob_, lddo, S - qt * 64, st_ + OIMG, wave, lane);
            dma_tile<D, IMG_TR>(ob_, lddo, S - qt * 64, st_ + OIMG + C::TILE, wave, lane);
        }
        {   // the tile's 64 lse / delta values: 4-byte-per-lane pieces, rows >= S read 0.  Every wave issues one (waves 2, 3
            // duplicate 0, 1) so that all waves have the same number of pieces in flight for the counted wait.
            const float* src = ((wave & 1) == 0 ? lse : delta) + ((int64_t)b * Hq + hq) * S + (int64_t)qt * 64;
            auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 0x7fffffff, 0x00020000);
            const unsigned voff = qt * 64 + lane < S ? (unsigned)(lane * 4) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(rowc + istage * 512 + (wave & 1) * 256), 4, voff, 0, 0, 0);
        }
        ++iq;
        istage = istage == NST - 1 ? 0 : istage + 1;
        return true;
    };
    [[maybe_unused]] const bool prof_on = threadIdx.x == 0;
    [[maybe_unused]] unsigned long long prof_acc[16] = {};
    [[maybe_unused]] const unsigned long long t_wg = PROF_T();
    int inflight = 0;
    inflight += issue_next();
    inflight += issue_next();

    const LaneOff<D> lo = lane_offsets<D>(lane);
    const unsigned lds0 = (unsigned)(unsigned long long)LDS_PTR(smem);
    int cstage = 0;
    int st1 = 0, st2 = 0;  // dS stores issued by the previous trip and by the one before it
    // the K / V rows of a key block are requested while the PREVIOUS block's dK / dV are written out (the tile loop's registers are free there): their
    // latency, 3-4k cycles per block in front of the first tile before, hides behind the epilogue's stores
    bf16x8 tk[KS], tv[KS];
    auto request_kv = [&](int kb_) {
        const int kg_ = kb_ * 128 + wave * 32 + (lane & 31);
        load_rows_frag<D>(k + (int64_t)b * S * ldk + (int64_t)hkv * D, ldk, kg_, kg_ < S, lane, tk);
        load_rows_frag<D>(v + (int64_t)b * S * ldv + (int64_t)hkv * D, ldv, kg_, kg_ < S, lane, tv);
    };
    if (kb_lo < kb_hi) request_kv(kb_lo);
    for (int kb = kb_lo; kb < kb_hi; ++kb) {
        [[maybe_unused]] const unsigned long long t_pro = PROF_T();
        const int k0 = kb * 128;
        const int kg = k0 + wave * 32 + (lane & 31);
        const bool kvalid = kg < S;
        const bool kreal = kvalid && (key_mask == nullptr || key_mask[(int64_t)b * S + kg] != 0);
        owned_zero<OWNED, 0, 32 * DT>();
        if constexpr (LEAN) {  // k * log2(e) / sqrt(d), rounded to bf16 again: the S chain then delivers the exponent of P
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                u32x4 w = __builtin_bit_cast(u32x4, tk[ks]);
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = pack_bf2(__uint_as_float(w[e] << 16) * scale_log2, __uint_as_float(w[e] & 0xffff0000u) * scale_log2);
                tk[ks] = __builtin_bit_cast(bf16x8, w);
            }
        }
        static_for<KS>([&](auto ks) { owned_write4<OWNED, KF0 + 4 * ks.value>(tk[ks.value]); });
        static_for<KS>([&](auto ks) { owned_write4<OWNED, VF0 + 4 * ks.value>(tv[ks.value]); });
        const int qt0 = qt0_of(kb);
        const int per_head = nqt_all - qt0;
        const int nit = per_head * rep;
        PROF_ADD(3, t_pro);
        int ch = 0, cq = -1;  // the tile consumed: query head within the kv group, query tile counted from qt0
        for (int it = 0; it < nit; ++it) {
            if (++cq == per_head) { cq = 0; ++ch; }
            // this wave's pieces of the tile have landed (a younger tile may stay in flight), then everybody's
            [[maybe_unused]] const unsigned long long t_w = PROF_T();
            // (gfx9 counts stores in vmcnt too, in issue order with the loads: the dS stores of the last two trips, 4 each, are younger than the tile awaited)
            // order of issue behind the awaited tile: the stores of trip m-2, the next tile's pieces (+ its row constants), the stores of trip m-1
            if (inflight >= 2) {
                if (st1 + st2 == 8) wait_vmcnt<PIECES + 1 + 8>();
                else if (st1 + st2 == 4) wait_vmcnt<PIECES + 1 + 4>();
                else wait_vmcnt<PIECES + 1>();
            } else wait_vmcnt<0>();
            st2 = st1;
            st1 = 0;
            PROF_ADD(1, t_w);
            [[maybe_unused]] const unsigned long long t_b = PROF_T();
            __builtin_amdgcn_s_barrier();
            PROF_ADD(6, t_b);
            --inflight;
            [[maybe_unused]] const unsigned long long t_i = PROF_T();
            if (!(abl & 1)) inflight += issue_next();  // into the stage that was consumed before this barrier
            PROF_ADD(7, t_i);
#if ATTN_ABL & 16
            struct BodyTimer {
                unsigned long long t0; bool on; unsigned long long* acc;
                __device__ ~BodyTimer() { if (on) { acc[2] += __builtin_readcyclecounter() - t0; acc[5] += 1; } }
            } body_timer{PROF_T(), prof_on, prof_acc};
#endif
            const int qt = qt0 + cq;
            const int qroff = cstage * STAGE, qtoff = UNI ? qroff : qroff + C::TILE, oroff = qroff + OIMG, otoff = UNI ? oroff : oroff + C::TILE;
            const float* rc = reinterpret_cast<const float*>(rowc + cstage * 512) + 4 * (lane >> 5);
            cstage = cstage == NST - 1 ? 0 : cstage + 1;
            // queries of this tile all precede this wave's keys -> nothing visible
            if (causal && qt * 64 + 63 < k0 + wave * 32) continue;
            // masks only where the tile touches the diagonal, the sequence end, or this wave holds padded / out-of-range keys
            const bool boundary = (causal && qt * 64 < k0 + wave * 32 + 31) || (qt * 64 + 64 > S) || __any(!kreal);
            int vqx[KS], vox[KS], vqt[DT], vot[DT];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                vqx[ks] = ((UNI ? lo.rowu : lo.row) + qroff) ^ (ks << 5);
                vox[ks] = ((UNI ? lo.rowu : lo.row) + oroff) ^ (ks << 5);
            }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                vqt[dt] = (UNI ? lo.colu[dt] : lo.col[dt]) + qtoff;
                vot[dt] = (UNI ? lo.colu[dt] : lo.col[dt]) + otoff;
            }

            auto tile_body = [&](auto bc) {
                constexpr bool BOUNDARY = decltype(bc)::value;
                bf16x8 f[RING] = {};
                TrHalves ft[RING] = {};
                f32x16 sacc[2], pacc[2];
                unsigned pw[2][8] = {}, dsw[2][8] = {};  // packed P / dS: words 4s..4s+3 of sub-tile st are the B operand of k-step s
                float lsr[2][16], dlr[2][16], tt[2][16], pv[2][16], dsv[2][16];
                [[maybe_unused]] f32x16 linit[2], dinit[2];  // LEAN: -lse * log2(e) and -delta of the accumulator rows = the chains' initial values
                auto load = [&](auto gc) {
                    constexpr int g = gc.value;
                    if constexpr (ATTN_ABL & 1) return;
                    if constexpr (g < 2 * NA) {
                        constexpr int st = g / NA, ks = (g % NA) / 2, which = g % 2;
                        f[g % RING] = lds_frag(smem, which ? vox[ks] : vqx[ks], st * 32 * C::ROWB);
                    } else if constexpr (g < NG) {
                        constexpr int h = g - 2 * NA, st = h / NC, sd = (h % NC) / (2 * DT), dt = (h % (2 * DT)) / 2, which = h % 2;
                        constexpr int imm = (st * 32 + 16 * sd) * C::ROWB;
                        const int va = which ? vqt[dt] : vot[dt];
                        if constexpr (!(ATTN_ABL & 32)) tr_issue<imm, imm + 8 * C::ROWB>(ft[g % RING], lds0 + va, lds0 + (UNI ? va ^ 0x20 : va));
                    }
                };
                // per-query constants of the 16 accumulator rows of sub-tile st: rows 8*g4 + 4*h + 0..3 (lse, then delta)
                auto row_constants = [&](auto stc) {
                    constexpr int st = stc.value;
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const f32x4 a = *reinterpret_cast<const f32x4*>(rc + st * 32 + 8 * g4);
                        const f32x4 c = *reinterpret_cast<const f32x4*>(rc + 64 + st * 32 + 8 * g4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if constexpr (LEAN) {
                                linit[st][4 * g4 + e] = a[e];
                                dinit[st][4 * g4 + e] = c[e];
                            } else {
                                lsr[st][4 * g4 + e] = a[e];
                                dlr[st][4 * g4 + e] = c[e];
                            }
                        }
                    }
                };
                // One element (accumulator register e of sub-tile st) in four stages.  A wave alone on its SIMD pays every dependent pair of
                // vector instructions in full (tools/microbench/mfma_gap.hip: seven instructions of ONE element's chain beside an MFMA make the gap
                // 62 cycles, three of them 33), so a gap never holds two stages of the same element: b_step(m) runs stage k of element m - k.
                // The empty statements pin each stage's result where it is written (LLVM otherwise sinks a whole chain to its consumer).
                auto stage = [&](auto stc, auto ec, auto kc) {
                    constexpr int st = stc.value, e = ec.value, k = kc.value;
                    if constexpr (e < 0 || e >= 16 || (ATTN_ABL & 2)) return;
                    else if constexpr (LEAN) {
                        // sacc = the exponent of P already, pacc = dP - delta already: exp2, multiply, pack
                        if constexpr (k == 1 && BOUNDARY) {
                            const int qq = qt * 64 + st * 32 + acc_row(e, lane);
                            const bool masked = (causal && kg > qq) || !kreal || qq >= S;
                            sacc[st][e] = masked ? -INFINITY : sacc[st][e];  // exp2(-inf) = 0: no branch around the exp
                            asm volatile("" : "+v"(sacc[st][e]));
                        } else if constexpr (k == 2) {
                            pv[st][e] = __builtin_amdgcn_exp2f(sacc[st][e]);
                            asm volatile("" : "+v"(pv[st][e]));
                        } else if constexpr (k == 3) {
                            dsv[st][e] = pacc[st][e] * pv[st][e];  // dS / scale (the factor is applied to dK once)
                            if constexpr (e % 2 == 1) {
                                pw[st][e / 2] = pack_bf2(pv[st][e - 1], pv[st][e]);
                                dsw[st][e / 2] = pack_bf2(dsv[st][e - 1], dsv[st][e]);
                                asm volatile("" : "+v"(pw[st][e / 2]), "+v"(dsw[st][e / 2]));
                            } else {
                                asm volatile("" : "+v"(dsv[st][e]));
                            }
                        }
                    } else if constexpr (k == 0) {
                        lsr[st][e] *= LOG2E;
                        asm volatile("" : "+v"(lsr[st][e]));
                    } else if constexpr (k == 1) {
                        float t = fmaf(sacc[st][e], scale_log2, -lsr[st][e]);
                        if constexpr (BOUNDARY) {
                            const int qq = qt * 64 + st * 32 + acc_row(e, lane);
                            const bool masked = (causal && kg > qq) || !kreal || qq >= S;
                            t = masked ? -INFINITY : t;  // exp2(-inf) = 0: no branch around the exp
                        }
                        tt[st][e] = t;
                        asm volatile("" : "+v"(tt[st][e]));
                    } else if constexpr (k == 2) {
                        pv[st][e] = __builtin_amdgcn_exp2f(tt[st][e]);
                        dsv[st][e] = pacc[st][e] - dlr[st][e];
                        asm volatile("" : "+v"(pv[st][e]), "+v"(dsv[st][e]));
                    } else {
                        dsv[st][e] *= pv[st][e];  // dS / scale (the factor is applied to dK once)
                        if constexpr (e % 2 == 1) {
                            pw[st][e / 2] = pack_bf2(pv[st][e - 1], pv[st][e]);
                            dsw[st][e / 2] = pack_bf2(dsv[st][e - 1], dsv[st][e]);
                            asm volatile("" : "+v"(pw[st][e / 2]), "+v"(dsw[st][e / 2]));
                        } else {
                            asm volatile("" : "+v"(dsv[st][e]));
                        }
                    }
                };
                // virtual step m of the B phase of sub-tile st, PER elements entering per step: steps 0 .. 16 / PER + 2
                auto b_step = [&](auto stc, auto mc, auto perc) {
                    constexpr int m = mc.value, PER = perc.value;
                    if constexpr (m >= 0 && m < 16 / PER + 3)
                        static_for<4>([&](auto kc) {
                            static_for<PER>([&](auto i) { stage(stc, std::integral_constant<int, (m - kc.value) * PER + i.value>{}, kc); });
                        });
                };
                if constexpr (LEAN) row_constants(std::integral_constant<int, 0>{});  // the first chains start from them
                static_for<PD>([&](auto g) { load(g); });
                [[maybe_unused]] unsigned long long t_seg = PROF_T();
                static_for<NG>([&](auto gc) {
                    constexpr int g = gc.value;
#if ATTN_ABL & 16
                    if constexpr (g == NA || g == 2 * NA || g == 2 * NA + NC) {
                        if (prof_on) { const unsigned long long now = __builtin_readcyclecounter(); prof_acc[8 + (g == NA ? 0 : g == 2 * NA ? 1 : 2)] += now - t_seg; t_seg = now; }
                    }
#endif
                    if constexpr (!LEAN && g == NA - 1) row_constants(std::integral_constant<int, 0>{});
                    if constexpr (LEAN && g == NA - 4) row_constants(std::integral_constant<int, 1>{});  // a few products ahead of the chains they start
                    load(std::integral_constant<int, g + PD>{});
                    if constexpr (g < 2 * NA) {
                        constexpr int st = g / NA, ks = (g % NA) / 2, which = g % 2;
                        if constexpr ((ATTN_ABL & 8) && ks > 0) {
                        } else if constexpr (LEAN && ks == 0) {
                            if constexpr (which == 0) mfma_ownedB_init<OWNED, KF0>(sacc[st], f[g % RING], linit[st]);
                            else mfma_ownedB_init<OWNED, VF0>(pacc[st], f[g % RING], dinit[st]);
                        } else if constexpr (which == 0) mfma_ownedB<OWNED, KF0 + 4 * ks, ks == 0>(sacc[st], f[g % RING]);
                        else mfma_ownedB<OWNED, VF0 + 4 * ks, ks == 0>(pacc[st], f[g % RING]);
                    } else if constexpr (!(ATTN_ABL & 4)) {
                        constexpr int h = g - 2 * NA, st = h / NC, sd = (h % NC) / (2 * DT), dt = (h % (2 * DT)) / 2, which = h % 2;
                        constexpr int younger = NG - 1 - g < PD ? NG - 1 - g : PD;
                        const bf16x8 fa = tr_wait<2 * younger>(ft[g % RING]);
                        if constexpr (which == 0) {
                            const u32x4 w = {pw[st][4 * sd], pw[st][4 * sd + 1], pw[st][4 * sd + 2], pw[st][4 * sd + 3]};
                            mfma_owned<OWNED, 16 * dt, false>(fa, __builtin_bit_cast(bf16x8, w));
                        } else {
                            const u32x4 w = {dsw[st][4 * sd], dsw[st][4 * sd + 1], dsw[st][4 * sd + 2], dsw[st][4 * sd + 3]};
                            mfma_owned<OWNED, DK0 + 16 * dt, false>(fa, __builtin_bit_cast(bf16x8, w));
                        }
                    }
                    // B(0) rides on the MFMAs of A(1) and the first three of C(0), B(1) on those of C(0) and the first three of C(1): the k-step
                    // sd = 1 products that need the last elements start later than that in both C phases
                    if constexpr (g == NA) tiles_settle(sacc[0], pacc[0]);
                    if constexpr (g == 2 * NA) tiles_settle(sacc[1], pacc[1]);
                    static_assert(16 % NA == 0 && 16 % NC == 0 && 16 / (16 / NA) + 3 <= NA + NC / 2 && 16 / (16 / NC) + 3 <= NC + NC / 2, "B phase does not fit its gaps");
                    b_step(std::integral_constant<int, 0>{}, std::integral_constant<int, g - NA>{}, std::integral_constant<int, 16 / NA>{});
                    b_step(std::integral_constant<int, 1>{}, std::integral_constant<int, g - 2 * NA>{}, std::integral_constant<int, 16 / NC>{});
                    if constexpr (!LEAN && g == 2 * NA - 1) row_constants(std::integral_constant<int, 1>{});
                    // keep this step's VALU where it is written: left alone, hipcc gathers a whole B phase into one MFMA gap
                    __builtin_amdgcn_sched_barrier(0);
                });
                PROF_ADD(11, t_seg);
                if (ds_out) {
                    // 16-byte units in the order the registers hold them: store (st, c2) = words 4 c2 .. 4 c2 + 3 of sub-tile st = this lane's key
                    // against queries 32 st + 16 c2 + 4 (lane >> 5) + {0..3, 8..11}; unit index = lane, so every store is one contiguous KiB.
                    // Blocks of 2 KiB: [b][hq][query tile][st][32-key group] -- what one wave of the dQ kernel streams through.
                    const int hq_ = hkv * rep + ch;
                    const int nqb_ = (S + 127) / 128;
                    char* blk = reinterpret_cast<char*>(ds_out) + ((((int64_t)b * Hq + hq_) * (2 * nqb_) + qt) * 2 * (4 * nqb_) + (kb * 4 + wave)) * 2048 + lane * 16;
                    static_for<4>([&](auto i) {
                        constexpr int st = i.value / 2, c2 = i.value % 2;
                        const u32x4 w4 = {dsw[st][4 * c2], dsw[st][4 * c2 + 1], dsw[st][4 * c2 + 2], dsw[st][4 * c2 + 3]};
                        *reinterpret_cast<u32x4*>(blk + (int64_t)st * (4 * nqb_) * 2048 + c2 * 1024) = w4;
                    });
                }
            };
            if (boundary) tile_body(std::true_type{});
            else tile_body(std::false_type{});
            st1 = ds_out != nullptr ? 4 : 0;
        }
        [[maybe_unused]] const unsigned long long t_epi = PROF_T();
        owned_settle<OWNED>();
        if (kb + 1 < kb_hi) request_kv(kb + 1);
        // 16-byte stores: the two half-waves hold adjacent 8-byte pieces of a key's row (d = 8 g4 + 4 (lane >> 5) + 0..3); one v_permlane32_swap per word
        // gives the lower half-wave both pieces of an even g4 and the upper half-wave both pieces of the odd one -- 8 stores per tensor instead of 16
        // (the write-out is bound by what it costs to issue a store)
        bf16_t* krow = dk + ((int64_t)b * S + kg) * lddk + (int64_t)hkv * D + 8 * (lane >> 5);
        bf16_t* vrow = dv + ((int64_t)b * S + kg) * lddv + (int64_t)hkv * D + 8 * (lane >> 5);
        static_for<DT * 2>([&](auto i) {
            constexpr int dt = i.value / 2, gp = i.value % 2, rv = 16 * dt + 8 * gp, rk = DK0 + rv;
            const unsigned ka0 = pack_bf2(owned_read<OWNED, rk>() * scale, owned_read<OWNED, rk + 1>() * scale), ka1 = pack_bf2(owned_read<OWNED, rk + 2>() * scale, owned_read<OWNED, rk + 3>() * scale);
            const unsigned kb0 = pack_bf2(owned_read<OWNED, rk + 4>() * scale, owned_read<OWNED, rk + 5>() * scale), kb1 = pack_bf2(owned_read<OWNED, rk + 6>() * scale, owned_read<OWNED, rk + 7>() * scale);
            const unsigned va0 = pack_bf2(owned_read<OWNED, rv>(), owned_read<OWNED, rv + 1>()), va1 = pack_bf2(owned_read<OWNED, rv + 2>(), owned_read<OWNED, rv + 3>());
            const unsigned vb0 = pack_bf2(owned_read<OWNED, rv + 4>(), owned_read<OWNED, rv + 5>()), vb1 = pack_bf2(owned_read<OWNED, rv + 6>(), owned_read<OWNED, rv + 7>());
            const auto k0s = __builtin_amdgcn_permlane32_swap(ka0, kb0, false, false), k1s = __builtin_amdgcn_permlane32_swap(ka1, kb1, false, false);
            const auto v0s = __builtin_amdgcn_permlane32_swap(va0, vb0, false, false), v1s = __builtin_amdgcn_permlane32_swap(va1, vb1, false, false);
            const u32x4 wk = {k0s[0], k1s[0], k0s[1], k1s[1]}, wv = {v0s[0], v1s[0], v0s[1], v1s[1]};
            if (kvalid && !(abl & 2)) {
                *reinterpret_cast<u32x4*>(krow + dt * 32 + 16 * gp) = wk;
                *reinterpret_cast<u32x4*>(vrow + dt * 32 + 16 * gp) = wv;
            }
        });
        PROF_ADD(4, t_epi);
    }
    PROF_ADD(0, t_wg);
#if ATTN_ABL & 16
    if (prof_on)
        for (int i = 0; i < 16; ++i) atomicAdd(&g_prof[i], prof_acc[i]);
#endif
}

// ---- dQ from the spilled dS:  dQ^T[d x q] = scale * K^T[d x key] dS^T[key x q].  One workgroup = 128 queries of one (batch, head), a
// wave = 32 of them = one (query tile, st) stream of the scratch buffer.  Per 64-key tile the K tile (transposed-read image, shared) and
// each wave's own two 2-KiB dS blocks arrive by LDS-DMA, double-buffered; both MFMA operands are transposing reads in the same k order,
// so the loop is 16 MFMAs and 40 LDS reads per tile and wave with no vector arithmetic at all.  The causal bound mirrors the dK/dV pass:
// a (wave, 32-key group) is fetched iff that pass wrote it.
// dS block in LDS: 2 pieces (c2) of 64 16-byte units; unit (h, key) sits in slot (32 h + key) ^ (h << 2) (the DMA lane for slot i fetches unit
// i ^ ((i >> 5) << 2)), so the 16 lanes of a transposing read -- 4 keys x {h = 0, 1} x two 8-byte halves -- cover 16 distinct 8-byte bank slots.
// Unit (c2, h, key) holds queries 16 c2 + 4 h + {0..3} and + {8..11}: query quad qd (of the wave's 8) is half (qd >> 1) & 1 of unit (c2 = qd >> 2, h = qd & 1).
// FUSE: the write-out is the backward of the projection's QK-norm + RoPE for the query heads (norm_rope.hip's qknorm_rope_bwd_kernel, same
// arithmetic, on the fp32 accumulators instead of a bf16 dQ read back from memory): a lane holds 64 of its query row's 128 features, and
// feature d's rotary partner d + 64 is in the same lane; d(qkv) rows go out directly, the norm-weight gradient as one partial row per workgroup
// (summed in a fixed order: bit-reproducible).
struct QkFuse {
    const bf16_t* qkv;   // pre-norm projections [tokens, ldqkv]: query head h at column h * D
    int64_t ldqkv;
    const bf16_t* qw;    // RMSNorm weight of the query heads [D]
    const float* cosT;   // [positions, D]
    const float* sinT;
    const int32_t* pos;  // [tokens]
    const float* rstd;   // [tokens, rstd_heads]
    int rstd_heads;
    bf16_t* dqkv;        // d(qkv) [tokens, lddqkv]
    int64_t lddqkv;
    float* dwp;          // [gridDim.x, D]
    const bf16_t* cs16;  // optional: [positions][cos (D/2) | sin (D/2)] bf16, valid when both halves of the fp32 tables are equal (plain RoPE): the
                         // coefficients are rounded to bf16 before use anyway, so the results are the same bits from a quarter of the loads
};
__device__ __forceinline__ void unpack4(const u32x2 v, float (&o)[4]) {
    o[0] = __uint_as_float(v[0] << 16);
    o[1] = __uint_as_float(v[0] & 0xffff0000u);
    o[2] = __uint_as_float(v[1] << 16);
    o[3] = __uint_as_float(v[1] & 0xffff0000u);
}
__device__ __forceinline__ float round_bf(float x) { return bf2f(f2bf(x)); }

#ifndef DQ_ABL
#define DQ_ABL 0  // profiling builds only: 1 = dS pieces fetched for the first tile only, 2 = K tiles fetched for the first tile only, 4 = no MFMAs / fragment reads, 8 = no write-out
#endif
#if DQ_ABL & 16
__device__ unsigned long long g_dq_prof[32];
#define DQ_T() __builtin_readcyclecounter()
#define DQ_MARK(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); dqp[i] += now_ - dq_t; dq_t = now_; } while (0)
#else
#define DQ_T() 0ull
#define DQ_MARK(i) do { } while (0)
#endif
template <int D, bool FUSE>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_spill_kernel(int B, int S, int Hq, int Hkv, const bf16_t* __restrict__ k, int64_t ldk,
                                                                   const bf16_t* __restrict__ ds, bf16_t* __restrict__ dq, int64_t lddq,
                                                                   int causal, float scale, QkFuse f) {
    using C = Cfg<D>;
    constexpr int DT = C::DT, DSW = 4096;  // a wave's dS bytes per 64-key tile: 2 key groups x 2 KiB
    // Two workgroups per CU (one's LDS-DMA issue and write-out run under the other's tiles) and TWO rings: the K tiles (shared by the four waves, served
    // by L2: the workgroups of a head read the same rows) two stages deep, the dS blocks (private per wave, straight from HBM, written once by the
    // dK/dV pass) THREE.  Round-4 stamps (tools/ablate_dq.py on a -DDQ_ABL=16 build) of the two-stage form: a third of a workgroup's cycles waiting
    // for the dS piece requested one tile (1.8k cycles) earlier -- the loaded HBM latency is twice that.  With two dS tiles in flight the wait is
    // covered; 2 x 16 + 3 x 16 KiB = 80 KiB per workgroup, i.e. all 160 KiB for the two.  (Four whole stages with one workgroup per CU: 249 us against
    // 205 at the round-3 headline shape -- the second workgroup is worth more than the depth.)
    constexpr int KST = 2, DST = 3, DS_BASE = KST * C::TILE, DS_STAGE = 4 * DSW;
    __shared__ __attribute__((aligned(16))) char smem[KST * C::TILE + DST * DS_STAGE];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nqb = (S + 127) / 128;
    const int vid = xcd_chunked((int)blockIdx.x, (int)gridDim.x);
    int qb, bh;
    heavy_first(vid, nqb, B * Hq, ATTN_HEAD_GROUP, qb, bh);
    const int hq = bh % Hq, b = bh / Hq;
    const int hkv = hq / (Hq / Hkv);
    const int q0 = qb * 128;
    const int qw = q0 + wave * 32;
    const int qg = qw + (lane & 31);
    const int ntiles_all = (S + 63) / 64;
    const int ntiles = causal ? min(ntiles_all, (q0 + 127) / 64 + 1) : ntiles_all;
    const bf16_t* kbase = k + (int64_t)b * S * ldk + (int64_t)hkv * D;
    // this wave's stream: blocks [key group] of (b, hq, query tile 2 qb + wave / 2, st = wave % 2)
    const char* sblk = reinterpret_cast<const char*>(ds) + (((((int64_t)b * Hq + hq) * (2 * nqb) + 2 * qb + (wave >> 1)) * 2 + (wave & 1)) * (4 * nqb)) * 2048;
    auto srsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(sblk), 0, 0x7fffffff, 0x00020000);
    const unsigned unit = (unsigned)(lane ^ ((lane >> 5) << 2)) * 16;
    auto fetched = [&](int kgrp) { return !(causal && kgrp * 32 > qw + 31) && kgrp * 32 < S; };
    auto issue_k = [&](int kt) {  // 4 pieces per wave
        if (!((DQ_ABL & 2) && kt > 0)) dma_tile<D, IMG_TR>(kbase + (int64_t)kt * 64 * ldk, ldk, S - kt * 64, smem + (kt % KST) * C::TILE, wave, lane);
    };
    auto issue_ds = [&](int kt) {  // ALWAYS 4 pieces per wave (a key group the dK/dV pass did not write is requested out of range: zero fill, no traffic), so the
        char* mine = smem + DS_BASE + (kt % DST) * DS_STAGE + wave * DSW;  // counted vmcnt in front of the tile is one constant
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const bool live = fetched(2 * kt + half) && kt < ntiles && !((DQ_ABL & 1) && kt > 0);
            const unsigned o0 = live ? (unsigned)((2 * kt + half) * 2048) + unit : OOB;
            const unsigned o1 = live ? (unsigned)((2 * kt + half) * 2048 + 1024) + unit : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srsrc, LDS_PTR(mine + half * 2048), 16, o0, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srsrc, LDS_PTR(mine + half * 2048 + 1024), 16, o1, 0, 0, 0);
        }
    };
    const LaneOff<D> lok = lane_offsets<D>(lane);
    const int g = lane >> 4, q4 = (lane >> 2) & 3, p = lane & 3;
    const unsigned scol = (unsigned)((g & 1) * 1024 + ((((p & 1) * 32 + 4 * (g >> 1) + q4) ^ ((p & 1) << 2)) * 16) + (p >> 1) * 8) + DS_BASE + wave * DSW;
    const unsigned lds0 = (unsigned)(unsigned long long)LDS_PTR(smem);
    f32x16 acc[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    [[maybe_unused]] unsigned long long dqp[8] = {};
    [[maybe_unused]] unsigned long long dq_t = DQ_T();
    [[maybe_unused]] const unsigned long long dq_t0 = dq_t;
    // the write-out's coefficient rows are addressed by pos[token]: requested here, a whole main loop ahead (the stamps charged the dependent load ~2.5k cycles)
    [[maybe_unused]] int pos_pre[4] = {0, 0, 0, 0};
    if constexpr (FUSE) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int qrow = q0 + (int)(threadIdx.x >> 3) + 32 * it;
            pos_pre[it] = f.pos[(int64_t)b * S + (qrow < S ? qrow : S - 1)];
        }
    }
    // request order: dS(0) K(0) dS(1) | per tile kt: K(kt+1) dS(kt+2).  In front of tile kt the youngest four requests are dS(kt+1): vmcnt(4) leaves
    // exactly them in flight (dS(kt) and K(kt) are older and have landed); the last tile has nothing younger
    issue_ds(0);
    issue_k(0);
    issue_ds(1);
    for (int kt = 0; kt < ntiles; ++kt) {
        wait_vmcnt<4>();               // this wave's pieces of tile kt; the next dS block stays in flight ...
        DQ_MARK(0);                    // [0] waiting for this wave's DMA
        __builtin_amdgcn_s_barrier();  // ... and everybody's K pieces; the K stage refilled next was read one iteration ago
        DQ_MARK(1);                    // [1] per-tile barrier
        if (kt + 1 < ntiles) issue_k(kt + 1);
        else issue_ds(ntiles);         // keeps the request count per iteration constant (out of range: zero fill into a dS stage nobody reads again)
        issue_ds(kt + 2);
        DQ_MARK(2);                    // [2] DMA issue
        const unsigned kimg = lds0 + (kt % KST) * C::TILE, simg = lds0 + (kt % DST) * DS_STAGE + scol;
        static_for<2>([&](auto hc) {
            constexpr int half = hc.value;
            if (!fetched(2 * kt + half)) return;
            if constexpr (DQ_ABL & 4) return;
            // the half's two 16-key k-steps: B = dS^T fragment of this wave's 32 queries, A = K^T fragments of the DT column groups;
            // all reads from asm (a compiler-visible LDS read behind the DMA above would be guarded by vmcnt(0))
            TrHalves bs[2], ak[2][DT];
            static_for<2>([&](auto sc) {
                constexpr int sd = sc.value, imm_s = half * 2048 + sd * 256, imm_k = (half * 32 + 16 * sd) * C::ROWB;
                tr_issue<imm_s, imm_s + 128>(bs[sd], simg, simg);
                static_for<DT>([&](auto dt) { tr_issue<imm_k, imm_k + 8 * C::ROWB>(ak[sd][dt.value], kimg + lok.col[dt.value], kimg + lok.col[dt.value]); });
            });
            static_for<2>([&](auto sc) {
                constexpr int sd = sc.value;
                // LDS reads issued after bs[sd]: its own DT K fragments and, for sd = 0, the whole second k-step: 2 (DT + (1 - sd) (1 + DT))
                constexpr int after_b = 2 * DT + (sd == 0 ? 2 * (1 + DT) : 0);
                const bf16x8 bf = tr_wait<(after_b < 15 ? after_b : 15)>(bs[sd]);
                static_for<DT>([&](auto dt) {
                    constexpr int after_a = 2 * (DT - 1 - dt.value) + (sd == 0 ? 2 * (1 + DT) : 0);
                    acc[dt.value] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_wait<(after_a < 15 ? after_a : 15)>(ak[sd][dt.value]), bf, acc[dt.value], 0, 0, 0);
                });
            });
        });
        DQ_MARK(3);                    // [3] fragment reads + MFMAs
    }
    if constexpr (DQ_ABL & 8) {
        if (acc[0][0] == 12345.678f) f.dwp[0] = acc[1][1] + acc[2][2] + acc[3][3];  // keeps the loop alive
        return;
    }
    if constexpr (!FUSE) {
        if (qg < S) {
            bf16_t* row = dq + ((int64_t)b * S + qg) * lddq + (int64_t)hq * D;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int d = dt * 32 + 8 * g4 + 4 * (lane >> 5);
                    const u32x2 pk = {pack_bf2(acc[dt][4 * g4] * scale, acc[dt][4 * g4 + 1] * scale), pack_bf2(acc[dt][4 * g4 + 2] * scale, acc[dt][4 * g4 + 3] * scale)};
                    *reinterpret_cast<u32x2*>(row + d) = pk;
                }
        }
    } else {
        static_assert(D == 128 && KST * C::TILE + DST * DS_STAGE >= 128 * D * 4, "fused write-out: head_dim 128, the rings hold the workgroup's [128][D] fp32 block");
        wait_vmcnt<0>();  // the zero-fill requests behind the last tile must have landed before the rings become the transposition buffer
        // the write-out's own operands (four rows per thread: row slot + 32 it) are requested first -- rows 0 and 1 here, under the transposition,
        // rows 2 and 3 under the arithmetic of rows 0 and 1 -- so the pass pays two memory latencies, not eight
        const int sub = threadIdx.x >> 3, i = (threadIdx.x & 7) * 8;
        if (f.cs16 != nullptr) {
            // ---- compact coefficient table: five loads per row (two of the pre-norm row, cos, sin, rstd), so all four rows of a thread are requested
            // here, in front of the transposition, and the pass pays ONE memory latency
            struct RowC {
                u32x4 xa, xb, c, sn;
                float r;
                int64_t tok;
                bool live;
            };
            RowC in[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int qrow = q0 + sub + 32 * it;
                in[it].live = qrow < S;
                in[it].tok = (int64_t)b * S + (in[it].live ? qrow : S - 1);
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const bf16_t* xr = f.qkv + in[it].tok * f.ldqkv + (int64_t)hq * D + i;
                in[it].xa = *reinterpret_cast<const u32x4*>(xr);
                in[it].xb = *reinterpret_cast<const u32x4*>(xr + D / 2);
                in[it].r = f.rstd[in[it].tok * f.rstd_heads + hq];
                const bf16_t* cr = f.cs16 + (int64_t)pos_pre[it] * D + i;
                in[it].c = *reinterpret_cast<const u32x4*>(cr);
                in[it].sn = *reinterpret_cast<const u32x4*>(cr + D / 2);
            }
            DQ_MARK(4);       // [4] write-out operand requests
            __syncthreads();  // every wave is past its last tile: the stages become the transposition buffer
            DQ_MARK(5);       // [5] waiting for the slowest wave
            float* xs = reinterpret_cast<float*>(smem);
            {
                const int row = wave * 32 + (lane & 31), hb = lane >> 5;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const int chunk = dt * 8 + 2 * g4 + hb;
                        const f32x4 val = {acc[dt][4 * g4] * scale, acc[dt][4 * g4 + 1] * scale, acc[dt][4 * g4 + 2] * scale, acc[dt][4 * g4 + 3] * scale};
                        *reinterpret_cast<f32x4*>(xs + row * D + 4 * (chunk ^ (row & 31))) = val;
                    }
            }
            __syncthreads();
            DQ_MARK(6);       // [6] transposition through LDS
            float w1[8], w2[8], dw1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dw2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            {
                const u32x4 a = *reinterpret_cast<const u32x4*>(f.qw + i), bq = *reinterpret_cast<const u32x4*>(f.qw + D / 2 + i);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    w1[2 * e] = __uint_as_float(a[e] << 16);
                    w1[2 * e + 1] = __uint_as_float(a[e] & 0xffff0000u);
                    w2[2 * e] = __uint_as_float(bq[e] << 16);
                    w2[2 * e + 1] = __uint_as_float(bq[e] & 0xffff0000u);
                }
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = sub + 32 * it;
                const float r = in[it].r;
                float g1[8], g2[8];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int ch = (i >> 2) + hh;
                    const f32x4 ga = *reinterpret_cast<const f32x4*>(xs + row * D + 4 * (ch ^ (row & 31)));
                    const f32x4 gb = *reinterpret_cast<const f32x4*>(xs + row * D + 4 * ((ch + 16) ^ (row & 31)));
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        g1[4 * hh + e] = ga[e];
                        g2[4 * hh + e] = gb[e];
                    }
                }
                float dn1[8], dn2[8], xh1[8], xh2[8];
                float dot = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float x1 = (e & 1) ? __uint_as_float(in[it].xa[e >> 1] & 0xffff0000u) : __uint_as_float(in[it].xa[e >> 1] << 16);
                    const float x2 = (e & 1) ? __uint_as_float(in[it].xb[e >> 1] & 0xffff0000u) : __uint_as_float(in[it].xb[e >> 1] << 16);
                    const float cc = (e & 1) ? __uint_as_float(in[it].c[e >> 1] & 0xffff0000u) : __uint_as_float(in[it].c[e >> 1] << 16);   // cos of d and of d + 64
                    const float ss = (e & 1) ? __uint_as_float(in[it].sn[e >> 1] & 0xffff0000u) : __uint_as_float(in[it].sn[e >> 1] << 16);
                    // y1 = c n1 - s n2, y2 = c n2 + s n1
                    dn1[e] = cc * g1[e] + ss * g2[e];
                    dn2[e] = cc * g2[e] - ss * g1[e];
                    xh1[e] = x1 * r;
                    xh2[e] = x2 * r;
                    dot += dn1[e] * w1[e] * xh1[e] + dn2[e] * w2[e] * xh2[e];
                }
                dot += __shfl_xor(dot, 4, 64);
                dot += __shfl_xor(dot, 2, 64);
                dot += __shfl_xor(dot, 1, 64);
                dot *= 1.0f / (float)D;
                if (in[it].live) {
                    u32x4 o1, o2;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        o1[e] = pack_bf2(r * (dn1[2 * e] * w1[2 * e] - xh1[2 * e] * dot), r * (dn1[2 * e + 1] * w1[2 * e + 1] - xh1[2 * e + 1] * dot));
                        o2[e] = pack_bf2(r * (dn2[2 * e] * w2[2 * e] - xh2[2 * e] * dot), r * (dn2[2 * e + 1] * w2[2 * e + 1] - xh2[2 * e + 1] * dot));
                    }
                    bf16_t* orow = f.dqkv + in[it].tok * f.lddqkv + (int64_t)hq * D + i;
                    *reinterpret_cast<u32x4*>(orow) = o1;
                    *reinterpret_cast<u32x4*>(orow + D / 2) = o2;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        dw1[e] += dn1[e] * xh1[e];
                        dw2[e] += dn2[e] * xh2[e];
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                xs[sub * D + i + e] = dw1[e];
                xs[sub * D + D / 2 + i + e] = dw2[e];
            }
            __syncthreads();
            if (threadIdx.x < D) {
                float sum = 0.f;
#pragma unroll
                for (int rgn = 0; rgn < 32; ++rgn) sum += xs[rgn * D + threadIdx.x];
                f.dwp[(int64_t)blockIdx.x * D + threadIdx.x] = sum;
            }
#if DQ_ABL & 16
            DQ_MARK(7);       // [7] arithmetic, stores, weight-gradient partial
            if ((threadIdx.x & 63) == 0 && (wave == 0 || wave == 3) && (blockIdx.x & 15) == 5) {
                const int o = wave == 0 ? 0 : 16;
                for (int k = 0; k < 8; ++k) atomicAdd(&g_dq_prof[o + k], dqp[k]);
                atomicAdd(&g_dq_prof[o + 8], __builtin_readcyclecounter() - dq_t0);
                atomicAdd(&g_dq_prof[o + 9], 1ull);
                atomicAdd(&g_dq_prof[o + 10], (unsigned long long)ntiles);
            }
#endif
            return;
        }
        struct RowIn {
            u32x4 xa, xb;
            f32x4 c[4], sn[4];  // cos / sin of features i..i+3, i+4..i+7, 64+i.., 64+i+4..
            float r;
            int64_t tok;
            bool live;
        };
        RowIn in[4];
        int64_t pp[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int qrow = q0 + sub + 32 * it;
            in[it].live = qrow < S;
            in[it].tok = (int64_t)b * S + (in[it].live ? qrow : S - 1);
            pp[it] = pos_pre[it];
        }
        auto fetch = [&](int it) {
            const bf16_t* xr = f.qkv + in[it].tok * f.ldqkv + (int64_t)hq * D + i;
            in[it].xa = *reinterpret_cast<const u32x4*>(xr);
            in[it].xb = *reinterpret_cast<const u32x4*>(xr + D / 2);
            in[it].r = f.rstd[in[it].tok * f.rstd_heads + hq];
            const float *cr = f.cosT + pp[it] * D + i, *sr = f.sinT + pp[it] * D + i;
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                in[it].c[hh] = *reinterpret_cast<const f32x4*>(cr + 4 * hh);
                in[it].sn[hh] = *reinterpret_cast<const f32x4*>(sr + 4 * hh);
                in[it].c[2 + hh] = *reinterpret_cast<const f32x4*>(cr + D / 2 + 4 * hh);
                in[it].sn[2 + hh] = *reinterpret_cast<const f32x4*>(sr + D / 2 + 4 * hh);
            }
        };
        fetch(0);
        fetch(1);
        __syncthreads();  // every wave is past its last tile: the stages become the transposition buffer
        // scale * dQ to LDS, row-major with the 16-byte chunk index XORed by the row (a lane holds 4-feature runs of ONE row; the rows of a half-wave
        // are 512 B apart): the stores of 8 consecutive rows then fall on 8 different chunk columns, and so do the loads of a row's 8 lanes below
        float* xs = reinterpret_cast<float*>(smem);
        {
            const int row = wave * 32 + (lane & 31), hb = lane >> 5;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int chunk = dt * 8 + 2 * g4 + hb;  // features 4 chunk .. 4 chunk + 3
                    const f32x4 val = {acc[dt][4 * g4] * scale, acc[dt][4 * g4 + 1] * scale, acc[dt][4 * g4 + 2] * scale, acc[dt][4 * g4 + 3] * scale};
                    *reinterpret_cast<f32x4*>(xs + row * D + 4 * (chunk ^ (row & 31))) = val;
                }
        }
        __syncthreads();
        fetch(2);
        fetch(3);
        // the arithmetic of qknorm_rope_bwd_kernel: 8 lanes per row, a lane owns features i..i+7 and their rotary partners 64+i..64+i+7
        float w1[8], w2[8], dw1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dw2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        {
            const u32x4 a = *reinterpret_cast<const u32x4*>(f.qw + i), bq = *reinterpret_cast<const u32x4*>(f.qw + D / 2 + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                w1[2 * e] = __uint_as_float(a[e] << 16);
                w1[2 * e + 1] = __uint_as_float(a[e] & 0xffff0000u);
                w2[2 * e] = __uint_as_float(bq[e] << 16);
                w2[2 * e + 1] = __uint_as_float(bq[e] & 0xffff0000u);
            }
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = sub + 32 * it;
            const float r = in[it].r;
            float g1[8], g2[8], c1[8], s1[8], c2[8], s2[8];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int ch = (i >> 2) + hh;
                const f32x4 ga = *reinterpret_cast<const f32x4*>(xs + row * D + 4 * (ch ^ (row & 31)));
                const f32x4 gb = *reinterpret_cast<const f32x4*>(xs + row * D + 4 * ((ch + 16) ^ (row & 31)));
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    g1[4 * hh + e] = ga[e];
                    g2[4 * hh + e] = gb[e];
                    c1[4 * hh + e] = round_bf(in[it].c[hh][e]);  // coefficients rounded to bf16, as the forward applies them
                    s1[4 * hh + e] = round_bf(in[it].sn[hh][e]);
                    c2[4 * hh + e] = round_bf(in[it].c[2 + hh][e]);
                    s2[4 * hh + e] = round_bf(in[it].sn[2 + hh][e]);
                }
            }
            float dn1[8], dn2[8], xh1[8], xh2[8];
            float dot = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float x1 = (e & 1) ? __uint_as_float(in[it].xa[e >> 1] & 0xffff0000u) : __uint_as_float(in[it].xa[e >> 1] << 16);
                const float x2 = (e & 1) ? __uint_as_float(in[it].xb[e >> 1] & 0xffff0000u) : __uint_as_float(in[it].xb[e >> 1] << 16);
                // y1 = c1 n1 - s1 n2, y2 = c2 n2 + s2 n1
                dn1[e] = c1[e] * g1[e] + s2[e] * g2[e];
                dn2[e] = c2[e] * g2[e] - s1[e] * g1[e];
                xh1[e] = x1 * r;
                xh2[e] = x2 * r;
                dot += dn1[e] * w1[e] * xh1[e] + dn2[e] * w2[e] * xh2[e];
            }
            dot += __shfl_xor(dot, 4, 64);
            dot += __shfl_xor(dot, 2, 64);
            dot += __shfl_xor(dot, 1, 64);
            dot *= 1.0f / (float)D;
            if (in[it].live) {
                u32x4 o1, o2;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o1[e] = pack_bf2(r * (dn1[2 * e] * w1[2 * e] - xh1[2 * e] * dot), r * (dn1[2 * e + 1] * w1[2 * e + 1] - xh1[2 * e + 1] * dot));
                    o2[e] = pack_bf2(r * (dn2[2 * e] * w2[2 * e] - xh2[2 * e] * dot), r * (dn2[2 * e + 1] * w2[2 * e + 1] - xh2[2 * e + 1] * dot));
                }
                bf16_t* orow = f.dqkv + in[it].tok * f.lddqkv + (int64_t)hq * D + i;
                *reinterpret_cast<u32x4*>(orow) = o1;
                *reinterpret_cast<u32x4*>(orow + D / 2) = o2;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    dw1[e] += dn1[e] * xh1[e];
                    dw2[e] += dn2[e] * xh2[e];
                }
            }
        }
        __syncthreads();  // the transposition buffer is dead: its first 16 KB take one [D] row per row slot, summed in a fixed order
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            xs[sub * D + i + e] = dw1[e];
            xs[sub * D + D / 2 + i + e] = dw2[e];
        }
        __syncthreads();
        if (threadIdx.x < D) {
            float sum = 0.f;
#pragma unroll
            for (int rgn = 0; rgn < 32; ++rgn) sum += xs[rgn * D + threadIdx.x];
            f.dwp[(int64_t)blockIdx.x * D + threadIdx.x] = sum;
        }
    }
}


int check_common(const char* name, int B, int S, int Hq, int Hkv, int D) {
    MI355_REQUIRE(D == 64 || D == 128, "%s: head_dim must be 64 or 128 (got %d)", name, D);
    MI355_REQUIRE(B > 0 && S > 0 && Hq > 0 && Hkv > 0 && Hq % Hkv == 0, "%s: bad shape B=%d S=%d Hq=%d Hkv=%d", name, B, S, Hq, Hkv);
    return 0;
}

}  // namespace

#if DQ_ABL & 16
extern "C" int mi355_debug_dq_prof(unsigned long long* out, int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dq_prof), sizeof(unsigned long long) * 32) != hipSuccess) return 2;
    if (reset) {
        unsigned long long z[32] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_dq_prof), z, sizeof(z)) != hipSuccess) return 3;
    }
    return 0;
}
#endif
#if ATTN_ABL & 16
extern "C" int mi355_debug_prof(unsigned long long* out, int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof(unsigned long long) * 16) != hipSuccess) return 2;
    if (reset) {
        unsigned long long z[16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof(z)) != hipSuccess) return 3;
    }
    return 0;
}
#endif

extern "C" int mi355_attn_fwd(int B, int S, int Hq, int Hkv, int D, const void* q, int64_t ldq, const void* k, int64_t ldk,
                              const void* v, int64_t ldv, void* o, int64_t ldo, float* lse, const uint8_t* key_mask,
                              int causal, float scale, void* stream) {
    if (check_common("mi355_attn_fwd", B, S, Hq, Hkv, D)) return 1;
    MI355_REQUIRE(q && k && v && o && lse, "mi355_attn_fwd: null pointer");
    MI355_REQUIRE(((ldq | ldk | ldv | ldo) & 7) == 0, "mi355_attn_fwd: leading dimensions must be multiples of 8");
    MI355_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o) & 15) == 0, "mi355_attn_fwd: operands must be 16-byte aligned");
    const int64_t grid = (int64_t)B * Hq * ((S + 127) / 128);
    MI355_REQUIRE(grid < 0x7fffffffLL, "mi355_attn_fwd: grid too large");
    hipStream_t s = (hipStream_t)stream;
    const float sl2 = scale * LOG2E;
    if (!((causal >> 8) & 2048)) {  // the lean-softmax kernel; ablation bit 11 keeps the first-generation one
        if (D == 128)
            hipLaunchKernelGGL(attn_fwd_lean_kernel<128>, dim3((unsigned)grid), dim3(256), 0, s, B, S, Hq, Hkv, (const bf16_t*)q, ldq, (const bf16_t*)k, ldk, (const bf16_t*)v, ldv, (bf16_t*)o, ldo, lse, key_mask, causal, sl2);
        else
            hipLaunchKernelGGL(attn_fwd_lean_kernel<64>, dim3((unsigned)grid), dim3(256), 0, s, B, S, Hq, Hkv, (const bf16_t*)q, ldq, (const bf16_t*)k, ldk, (const bf16_t*)v, ldv, (bf16_t*)o, ldo, lse, key_mask, causal, sl2);
        MI355_LAUNCH_CHECK("mi355_attn_fwd");
        return 0;
    }
    if (D == 128)
        hipLaunchKernelGGL(attn_fwd_kernel<128>, dim3((unsigned)grid), dim3(256), 0, s, B, S, Hq, Hkv, (const bf16_t*)q, ldq, (const bf16_t*)k, ldk, (const bf16_t*)v, ldv, (bf16_t*)o, ldo, lse, key_mask, causal, sl2);
    else
        hipLaunchKernelGGL(attn_fwd_kernel<64>, dim3((unsigned)grid), dim3(256), 0, s, B, S, Hq, Hkv, (const bf16_t*)q, ldq, (const bf16_t*)k, ldk, (const bf16_t*)v, ldv, (bf16_t*)o, ldo, lse, key_mask, causal, sl2);
    MI355_LAUNCH_CHECK("mi355_attn_fwd");
    return 0;
}

// the dS scratch: one bf16 per (query, key) of the padded square
static int64_t attn_bwd_ds_bytes(int B, int S, int Hq) {
    const int64_t nqb = (S + 127) / 128;
    return (int64_t)B * Hq * nqb * nqb * 128 * 128 * 2;
}
extern "C" int64_t mi355_attn_bwd_workspace_bytes(int B, int S, int Hq, int D) {
    if (D != 128 || B <= 0 || S <= 0 || Hq <= 0) return 0;  // the spilled form is built for head_dim 128
    // dS scratch, then -lse * log2(e) and -delta (fp32 [B, Hq, S] each): the initial accumulators of the dK/dV pass
    return attn_bwd_ds_bytes(B, S, Hq) + 2 * (((int64_t)B * Hq * S * 4 + 15) & ~(int64_t)15);
}

extern "C" int mi355_attn_bwd_ws(int B, int S, int Hq, int Hkv, int D, const void* q, int64_t ldq, const void* k, int64_t ldk,
                                 const void* v, int64_t ldv, const void* o, int64_t ldo, const void* d_o, int64_t lddo,
                                 const float* lse, float* delta, void* dq, int64_t lddq, void* dk, int64_t lddk, void* dv,
                                 int64_t lddv, const uint8_t* key_mask, int causal, float scale, void* workspace, int64_t workspace_bytes,
                                 void* stream);

extern "C" int mi355_attn_bwd(int B, int S, int Hq, int Hkv, int D, const void* q, int64_t ldq, const void* k, int64_t ldk,
                              const void* v, int64_t ldv, const void* o, int64_t ldo, const void* d_o, int64_t lddo,
                              const float* lse, float* delta, void* dq, int64_t lddq, void* dk, int64_t lddk, void* dv,
                              int64_t lddv, const uint8_t* key_mask, int causal, float scale, void* stream) {
    return mi355_attn_bwd_ws(B, S, Hq, Hkv, D, q, ldq, k, ldk, v, ldv, o, ldo, d_o, lddo, lse, delta, dq, lddq, dk, lddk, dv, lddv, key_mask, causal, scale,
                             nullptr, 0, stream);
}

static int attn_bwd_impl(int B, int S, int Hq, int Hkv, int D, const void* q, int64_t ldq, const void* k, int64_t ldk,
                         const void* v, int64_t ldv, const void* o, int64_t ldo, const void* d_o, int64_t lddo,
                         const float* lse, float* delta, void* dq, int64_t lddq, void* dk, int64_t lddk, void* dv,
                         int64_t lddv, const uint8_t* key_mask, int causal, float scale, void* workspace, int64_t workspace_bytes,
                         void* stream, const QkFuse* fuse) {
    if (check_common("mi355_attn_bwd", B, S, Hq, Hkv, D)) return 1;
    // MI355_ATTN_DELTA_READY: delta and the two row-constant arrays of the workspace were written by mi355_gemm_bf16_attn_delta (the epilogue of the
    // out-projection's dgrad): the delta kernel is skipped
    const bool delta_ready = (causal & MI355_ATTN_DELTA_READY) != 0;
    causal &= ~MI355_ATTN_DELTA_READY;
    if (fuse) {  // dQ never exists as a matrix: the dQ pass writes d(qkv) rows
        dq = const_cast<void*>(q);
        lddq = ldq;
    }
    MI355_REQUIRE(q && k && v && o && d_o && lse && delta && dq && dk && dv, "mi355_attn_bwd: null pointer");
    MI355_REQUIRE(((ldq | ldk | ldv | ldo | lddo | lddq | lddk | lddv) & 7) == 0, "mi355_attn_bwd: leading dimensions must be multiples of 8");
    MI355_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o | (uintptr_t)d_o | (uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) & 15) == 0,
                  "mi355_attn_bwd: operands and gradients must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const float sl2 = scale * LOG2E;
    const int64_t items = (int64_t)B * S * ((Hq + 512 / D - 1) / (512 / D));
    const int dgrid = (int)((items + 3) / 4 > 8192 ? 8192 : (items + 3) / 4);
    // enough (batch, head) pairs to fill the chip twice over: one workgroup walks all blocks of its pair (no per-block launch,
    // prologue and first-tile latency); otherwise one block per workgroup for parallelism
    const int nblk = (S + 127) / 128;
    const int bpw_q = (int64_t)B * Hq >= 512 ? nblk : 1;
    const int bpw_k = (int64_t)B * Hkv >= 512 ? nblk : 1;
    const int64_t gq = (int64_t)B * Hq * ((nblk + bpw_q - 1) / bpw_q), gk = (int64_t)B * Hkv * ((nblk + bpw_k - 1) / bpw_k);
    MI355_REQUIRE(S <= 64 * ATTN_MAX_TILES, "mi355_attn_bwd: S must be <= %d", 64 * ATTN_MAX_TILES);
    MI355_REQUIRE(gq < 0x7fffffffLL, "mi355_attn_bwd: grid too large");
    // workspace given (head_dim 128): the dK/dV pass spills dS and dQ is one product over it; otherwise the dQ pass recomputes S and dP
    const bool spill = D == 128 && workspace != nullptr && !((causal >> 8) & 1024);  // ablation bit 10: the three-product dQ pass
    MI355_REQUIRE(!fuse || spill, "mi355_attn_bwd_qnorm: head_dim 128 and a workspace (the one-product dQ pass) are required");
    if (spill) {
        MI355_REQUIRE(workspace_bytes >= mi355_attn_bwd_workspace_bytes(B, S, Hq, D), "mi355_attn_bwd_ws: workspace of %lld bytes, %lld needed",
                      (long long)workspace_bytes, (long long)mi355_attn_bwd_workspace_bytes(B, S, Hq, D));
        MI355_REQUIRE(((uintptr_t)workspace & 15) == 0, "mi355_attn_bwd_ws: workspace must be 16-byte aligned");
    }
    MI355_REQUIRE(!delta_ready || spill, "mi355_attn_bwd: MI355_ATTN_DELTA_READY needs the workspace form (head_dim 128)");
    bf16_t* ds_ws = spill ? (bf16_t*)workspace : nullptr;
    float* nl2 = spill ? (float*)((char*)workspace + attn_bwd_ds_bytes(B, S, Hq)) : nullptr;
    float* ndl = spill ? (float*)((char*)nl2 + (((int64_t)B * Hq * S * 4 + 15) & ~(int64_t)15)) : nullptr;
#define BWD_LAUNCH(DD)                                                                                                              \
    if (!delta_ready)                                                                                                               \
        hipLaunchKernelGGL(attn_delta_kernel<DD>, dim3(dgrid), dim3(256), 0, s, B, S, Hq, (const bf16_t*)o, ldo, (const bf16_t*)d_o, lddo, delta, lse, nl2, ndl); \
    if (spill && DD == 128)                                                                                                         \
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<128, true>), dim3((unsigned)gk), dim3(256), 0, s, B, S, Hq, Hkv, (const bf16_t*)q, ldq, (const bf16_t*)k, ldk, \
                           (const bf16_t*)v, ldv, (const bf16_t*)d_o, lddo, nl2, ndl, (bf16_t*)dk, lddk, (bf16_t*)dv, lddv, key_mask, causal, scale, sl2, bpw_k, ds_ws); \
    else                                                                                                                             \
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<DD, false>), dim3((unsigned)gk), dim3(256), 0, s, B, S, Hq, Hkv, (const bf16_t*)q, ldq, (const bf16_t*)k, ldk, \
                           (const bf16_t*)v, ldv, (const bf16_t*)d_o, lddo, lse, delta, (bf16_t*)dk, lddk, (bf16_t*)dv, lddv, key_mask, causal, scale, sl2, bpw_k, ds_ws); \
    if (!spill)                                                                                                                      \
        hipLaunchKernelGGL(attn_bwd_dq_kernel<DD>, dim3((unsigned)gq), dim3(256), 0, s, B, S, Hq, Hkv, (const bf16_t*)q, ldq, (const bf16_t*)k, ldk,  \
                           (const bf16_t*)v, ldv, (const bf16_t*)d_o, lddo, lse, delta, (bf16_t*)dq, lddq, key_mask, causal, scale, sl2, bpw_q);
    if (D == 128) {
        BWD_LAUNCH(128)
        if (spill && fuse)
            hipLaunchKernelGGL((attn_bwd_dq_spill_kernel<128, true>), dim3((unsigned)((int64_t)B * Hq * nblk)), dim3(256), 0, s, B, S, Hq, Hkv, (const bf16_t*)k, ldk,
                               (const bf16_t*)ds_ws, (bf16_t*)nullptr, (int64_t)0, causal & 0xff, scale, *fuse);
        else if (spill)
            hipLaunchKernelGGL((attn_bwd_dq_spill_kernel<128, false>), dim3((unsigned)((int64_t)B * Hq * nblk)), dim3(256), 0, s, B, S, Hq, Hkv, (const bf16_t*)k, ldk,
                               (const bf16_t*)ds_ws, (bf16_t*)dq, lddq, causal & 0xff, scale, QkFuse{});
    } else {
        BWD_LAUNCH(64)
    }
#undef BWD_LAUNCH
    MI355_LAUNCH_CHECK("mi355_attn_bwd");
    return 0;
}

extern "C" int64_t mi355_attn_bwd_workspace_rowconst_offset(int B, int S, int Hq, int D, int which) {
    if (D != 128 || B <= 0 || S <= 0 || Hq <= 0 || which < 0 || which > 1) return -1;
    const int64_t first = attn_bwd_ds_bytes(B, S, Hq);
    return which == 0 ? first : first + (((int64_t)B * Hq * S * 4 + 15) & ~(int64_t)15);
}

extern "C" int mi355_attn_bwd_ws(int B, int S, int Hq, int Hkv, int D, const void* q, int64_t ldq, const void* k, int64_t ldk,
                                 const void* v, int64_t ldv, const void* o, int64_t ldo, const void* d_o, int64_t lddo,
                                 const float* lse, float* delta, void* dq, int64_t lddq, void* dk, int64_t lddk, void* dv,
                                 int64_t lddv, const uint8_t* key_mask, int causal, float scale, void* workspace, int64_t workspace_bytes,
                                 void* stream) {
    return attn_bwd_impl(B, S, Hq, Hkv, D, q, ldq, k, ldk, v, ldv, o, ldo, d_o, lddo, lse, delta, dq, lddq, dk, lddk, dv, lddv, key_mask, causal, scale, workspace,
                         workspace_bytes, stream, nullptr);
}

extern "C" int64_t mi355_attn_bwd_qnorm_partials(int B, int S, int Hq) { return (int64_t)B * Hq * ((S + 127) / 128); }

extern "C" int mi355_attn_bwd_qnorm(int B, int S, int Hq, int Hkv, int D, const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                                    const void* o, int64_t ldo, const void* d_o, int64_t lddo, const float* lse, float* delta, void* dk, int64_t lddk,
                                    void* dv, int64_t lddv, const uint8_t* key_mask, int causal, float scale, void* workspace, int64_t workspace_bytes,
                                    const void* qkv, int64_t ldqkv, const void* q_weight, const float* cos, const float* sin, const int32_t* pos,
                                    const float* rstd, int rstd_heads, void* dqkv, int64_t lddqkv, float* dqw_partial, const void* rope_cs16, void* stream) {
    MI355_REQUIRE(D == 128, "mi355_attn_bwd_qnorm: head_dim must be 128 (got %d)", D);
    MI355_REQUIRE(qkv && q_weight && cos && sin && pos && rstd && dqkv && dqw_partial && workspace, "mi355_attn_bwd_qnorm: null pointer");
    MI355_REQUIRE(rstd_heads >= Hq && ldqkv >= (int64_t)Hq * D && lddqkv >= (int64_t)Hq * D && ((ldqkv | lddqkv) & 3) == 0,
                  "mi355_attn_bwd_qnorm: qkv / dqkv rows must hold the query heads (8-byte aligned pitches), rstd one column per head");
    MI355_REQUIRE((((uintptr_t)qkv | (uintptr_t)dqkv | (uintptr_t)q_weight) & 7) == 0 && (((uintptr_t)cos | (uintptr_t)sin) & 15) == 0,
                  "mi355_attn_bwd_qnorm: qkv / dqkv / weight 8-byte aligned, cos / sin 16-byte aligned");
    MI355_REQUIRE(!rope_cs16 || ((uintptr_t)rope_cs16 & 15) == 0, "mi355_attn_bwd_qnorm: rope_cs16 must be 16-byte aligned");
    const QkFuse f{(const bf16_t*)qkv, ldqkv, (const bf16_t*)q_weight, cos, sin, pos, rstd, rstd_heads, (bf16_t*)dqkv, lddqkv, dqw_partial, (const bf16_t*)rope_cs16};
    return attn_bwd_impl(B, S, Hq, Hkv, D, q, ldq, k, ldk, v, ldv, o, ldo, d_o, lddo, lse, delta, nullptr, 0, dk, lddk, dv, lddv, key_mask, causal, scale, workspace,
                         workspace_bytes, stream, &f);
}
