// HBM-bound elementwise / gather kernels: SwiGLU, cross-entropy rows, embedding gather + scatter-add,
// strided copy (early-fusion concat), im2row patch gather, casts, grad-norm helpers.
// All use 16-byte per-lane accesses where the layout allows and grid-stride over <= 2048 workgroups.
#include <stdarg.h>

#include "common.h"

// ------------------------------------------------------------------------------------------ error state
static thread_local char g_err[512] = "";
void mi355_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* mi355_last_error(void) { return g_err; }
extern "C" int mi355_abi_version(void) { return 1; }

namespace {

__device__ __forceinline__ void unpack8(const u32x4 v, float (&f)[8]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        f[2 * e] = __uint_as_float(v[e] << 16);
        f[2 * e + 1] = __uint_as_float(v[e] & 0xffff0000u);
    }
}
__device__ __forceinline__ u32x4 pack8(const float (&f)[8]) {
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack_bf2(f[2 * e], f[2 * e + 1]);
    return o;
}
__device__ __forceinline__ float rbf(float x) { return bf2f(f2bf(x)); }

inline int grid_for(int64_t work_items, int per_block) {
    int64_t g = (work_items + per_block - 1) / per_block;
    return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

// ------------------------------------------------------------------------------------------------ SwiGLU
// gu [tokens, 2F] = [u | g];  a = bf16( u * bf16(silu(g)) )   (reference: silu output is a bf16 tensor, then *)
__global__ __launch_bounds__(256) void swiglu_fwd_kernel(int64_t tokens, int F, const bf16_t* __restrict__ gu, bf16_t* __restrict__ a) {
    const int fv = F >> 3;
    const int64_t total = tokens * fv;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t t = i / fv;
        const int c = (int)(i - t * fv) * 8;
        float u[8], g[8], o[8];
        unpack8(*reinterpret_cast<const u32x4*>(gu + t * 2 * F + c), u);
        unpack8(*reinterpret_cast<const u32x4*>(gu + t * 2 * F + F + c), g);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = swiglu_act(u[e], g[e]);
        *reinterpret_cast<u32x4*>(a + t * F + c) = pack8(o);
    }
}
// du = da * silu(g);  dg = da * u * sig(g) * (1 + g*(1-sig(g)))
__global__ __launch_bounds__(256) void swiglu_bwd_kernel(int64_t tokens, int F, const bf16_t* __restrict__ gu,
                                                         const bf16_t* __restrict__ da, bf16_t* __restrict__ dgu) {
    const int fv = F >> 3;
    const int64_t total = tokens * fv;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t t = i / fv;
        const int c = (int)(i - t * fv) * 8;
        float u[8], g[8], d[8], du[8], dg[8];
        unpack8(*reinterpret_cast<const u32x4*>(gu + t * 2 * F + c), u);
        unpack8(*reinterpret_cast<const u32x4*>(gu + t * 2 * F + F + c), g);
        unpack8(*reinterpret_cast<const u32x4*>(da + t * F + c), d);
#pragma unroll
        for (int e = 0; e < 8; ++e) swiglu_grads(d[e], u[e], g[e], du[e], dg[e]);
        *reinterpret_cast<u32x4*>(dgu + t * 2 * F + c) = pack8(du);
        *reinterpret_cast<u32x4*>(dgu + t * 2 * F + F + c) = pack8(dg);
    }
}

// ------------------------------------------------------------------------------------------------ GELU (erf)
// nn.GELU() on a bf16 tensor (ViTAdapter, vit_engine.py:50): a = bf16(gelu(x));  backward dx = da * gelu'(x)
// KIND 0: exact erf GELU (nn.GELU());  KIND 1: tanh approximation (nn.GELU(approximate="tanh"), qwen3_5_vision_model.py:122)
// gelu_val<KIND> / gelu_grad<KIND> live in common.h (shared with the GEMM epilogues, so fused == separate bit for bit)
template <int KIND>
__global__ __launch_bounds__(256) void gelu_fwd_kernel(int64_t n, const bf16_t* __restrict__ x, bf16_t* __restrict__ y) {
    const int64_t nv = n >> 3;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (int64_t)gridDim.x * 256) {
        float v[8];
        unpack8(*reinterpret_cast<const u32x4*>(x + i * 8), v);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = gelu_val<KIND>(v[e]);
        *reinterpret_cast<u32x4*>(y + i * 8) = pack8(v);
    }
}
template <int KIND>
__global__ __launch_bounds__(256) void gelu_bwd_kernel(int64_t n, const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, bf16_t* __restrict__ dx) {
    const int64_t nv = n >> 3;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (int64_t)gridDim.x * 256) {
        float v[8], g[8];
        unpack8(*reinterpret_cast<const u32x4*>(x + i * 8), v);
        unpack8(*reinterpret_cast<const u32x4*>(dy + i * 8), g);
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] *= gelu_grad<KIND>(v[e]);
        *reinterpret_cast<u32x4*>(dx + i * 8) = pack8(g);
    }
}

// 3-D patch gather for Conv3d(k = s = (TP, P, P)) (qwen3_5_vision_model.py:79-107): tokens ordered (t', ph, pw), K ordered (c, dt, i, j)
template <int OUT_DT>
__global__ __launch_bounds__(256) void patchify3d_kernel(int B, int C, int T, int H, int W, int P, int TP, const float* __restrict__ img, void* __restrict__ rows) {
    const int gh = H / P, gw = W / P, gt = T / TP, K = C * TP * P * P, pv = P >> 2;
    const int64_t total = (int64_t)B * C * T * H * gw * pv;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        int64_t r = i;  // (b, c, t, y, pw, jv): consecutive threads read consecutive image addresses
        const int jv = (int)(r % pv); r /= pv;
        const int pw = (int)(r % gw); r /= gw;
        const int y = (int)(r % H); r /= H;
        const int t = (int)(r % T); r /= T;
        const int c = (int)(r % C); r /= C;
        const int b = (int)r;
        const int ph = y / P, ii = y - ph * P, tq = t / TP, dt = t - tq * TP;
        const f32x4 v = *reinterpret_cast<const f32x4*>(img + ((((int64_t)b * C + c) * T + t) * H + y) * W + pw * P + jv * 4);
        const int64_t orow = (((int64_t)b * gt + tq) * gh + ph) * gw + pw;
        const int k = ((c * TP + dt) * P + ii) * P + jv * 4;
        if constexpr (OUT_DT == MI355_DT_BF16) {
            u32x2 pk = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
            *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(rows) + orow * K + k) = pk;
        } else {
            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(rows) + orow * K + k) = v;
        }
    }
}

// ViTMergeAdapter permute (qwen3_5_vision_model.py:421-424): merged[(b,t,bh,bw)][(i,j,:)] <-> x[(b,t,bh*m+i,bw*m+j)][:]
// 16-byte chunks; inverse = the same map with src/dst swapped (backward).
__global__ __launch_bounds__(256) void merge_patches_kernel(int64_t frames, int gh, int gw, int m, int row_bytes, const char* __restrict__ src,
                                                            char* __restrict__ dst, int inverse) {
    const int cv = row_bytes >> 4;
    const int64_t total = frames * gh * gw * cv;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % cv);
        int64_t r = i / cv;  // index of the un-merged patch row (frame, y, x)
        const int x = (int)(r % gw); const int y = (int)((r / gw) % gh); const int64_t f = r / ((int64_t)gh * gw);
        const int bh = y / m, ii = y - bh * m, bw = x / m, jj = x - bw * m;
        const int64_t mrow = (f * (gh / m) + bh) * (gw / m) + bw;          // merged token
        const int64_t moff = (mrow * m * m + ii * m + jj) * (int64_t)row_bytes + c * 16;  // inside it: (i, j, :)
        const int64_t xoff = r * (int64_t)row_bytes + c * 16;
        if (inverse) *reinterpret_cast<u32x4*>(dst + xoff) = *reinterpret_cast<const u32x4*>(src + moff);
        else *reinterpret_cast<u32x4*>(dst + moff) = *reinterpret_cast<const u32x4*>(src + xoff);
    }
}

// masked_scatter of vision rows into the embedded sequence (qwen3_5_vlm_model.py:209-211), row-major fill order:
//   forward : out[t] = mask[t] ? vis[slot[t]] : emb[t]
//   backward: d_emb[t] = mask[t] ? 0 : g[t];   d_vis[slot[t]] = g[t] where mask[t]
__global__ __launch_bounds__(256) void scatter_rows_kernel(int64_t tokens, int row_bytes, const uint8_t* __restrict__ mask, const int32_t* __restrict__ slot,
                                                           const char* __restrict__ a, const char* __restrict__ b, char* __restrict__ out_a, char* __restrict__ out_b, int backward) {
    const int cv = row_bytes >> 4;
    const int64_t total = tokens * cv;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t t = i / cv;
        const int64_t c = (i - t * cv) * 16;
        const bool mk = mask[t] != 0;
        if (!backward) {  // a = emb, b = vis, out_a = fused
            const char* src = mk ? b + (int64_t)slot[t] * row_bytes : a + t * row_bytes;
            *reinterpret_cast<u32x4*>(out_a + t * row_bytes + c) = *reinterpret_cast<const u32x4*>(src + c);
        } else {          // a = grad of fused, out_a = d_emb, out_b = d_vis
            const u32x4 g = *reinterpret_cast<const u32x4*>(a + t * row_bytes + c);
            *reinterpret_cast<u32x4*>(out_a + t * row_bytes + c) = mk ? (u32x4){0, 0, 0, 0} : g;
            if (mk) *reinterpret_cast<u32x4*>(out_b + (int64_t)slot[t] * row_bytes + c) = g;
        }
    }
}

// ----------------------------------------------------------------------------------------- cross entropy
// One 256-thread block per row.  Pass 1: online max / sum-exp over V (16-byte loads).  Pass 2 (optional):
// dlogits = (exp(l - lse) - onehot) * scale, written bf16 (in place allowed).  The row (V*2 bytes ~ 300 KB)
// is re-read from L2/Infinity Cache in pass 2.  Targets: -100 = ignore_index (zero loss, zero gradient); anything else
// outside [0, V) is an error -- torch raises a device assert there; here the row's loss is NaN (which poisons the mean:
// loud, not silent) and its gradient row is zero, and nothing is read out of bounds.
__global__ __launch_bounds__(256) void ce_rows_kernel(int64_t rows, int64_t V, const bf16_t* __restrict__ logits, int64_t ldl,
                                                      const int64_t* __restrict__ targets, float* __restrict__ loss_rows,
                                                      bf16_t* dlogits, const float* __restrict__ grad_scale) {
    __shared__ float red_m[4], red_s[4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
        const int64_t tgt = targets[row];
        const bf16_t* lr = logits + row * ldl;
        bf16_t* dr = dlogits ? dlogits + row * ldl : nullptr;
        if (tgt < 0 || tgt >= V) {  // ignore_index (-100): zero loss, zero gradient; any other value out of range: NaN loss
            if (threadIdx.x == 0) loss_rows[row] = tgt == -100 ? 0.f : __builtin_nanf("");
            if (dr) {
                const int64_t nv = V >> 3;
                for (int64_t i = threadIdx.x; i < nv; i += 256) *reinterpret_cast<u32x4*>(dr + i * 8) = (u32x4){0, 0, 0, 0};
                for (int64_t i = (nv << 3) + threadIdx.x; i < V; i += 256) dr[i] = 0;
            }
            continue;
        }
        // the target logit is read BEFORE the barriers below: in place (dr == lr) the other waves start overwriting the row with
        // the gradient as soon as they pass the last barrier
        const float tgt_logit = threadIdx.x == 0 ? bf2f(lr[tgt]) : 0.f;
        float m = -INFINITY, s = 0.f;
        const int64_t nv = V >> 3;
        for (int64_t i = threadIdx.x; i < nv; i += 256) {
            float v[8];
            unpack8(*reinterpret_cast<const u32x4*>(lr + i * 8), v);
            float vm = v[0];
#pragma unroll
            for (int e = 1; e < 8; ++e) vm = fmaxf(vm, v[e]);
            if (vm > m) {
                s *= __expf(m - vm);
                m = vm;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) s += __expf(v[e] - m);
        }
        for (int64_t i = (nv << 3) + threadIdx.x; i < V; i += 256) {
            const float v = bf2f(lr[i]);
            if (v > m) {
                s *= __expf(m - v);
                m = v;
            }
            s += __expf(v - m);
        }
        const float wm = wave_max(m);
        s = wave_sum(m == -INFINITY ? 0.f : s * __expf(m - wm));  // lanes/waves that saw no element carry (-inf, 0)
        __syncthreads();  // protects red_* reuse across rows
        if (lane == 0) {
            red_m[wid] = wm;
            red_s[wid] = s;
        }
        __syncthreads();
        const float bm = fmaxf(fmaxf(red_m[0], red_m[1]), fmaxf(red_m[2], red_m[3]));
        float bs = 0.f;
#pragma unroll
        for (int wv = 0; wv < 4; ++wv) bs += red_m[wv] == -INFINITY ? 0.f : red_s[wv] * __expf(red_m[wv] - bm);
        const float lse = bm + __logf(bs);
        if (threadIdx.x == 0) loss_rows[row] = lse - tgt_logit;
        if (dr) {
            const float sc = *grad_scale;
            for (int64_t i = threadIdx.x; i < nv; i += 256) {
                float v[8], o[8];
                unpack8(*reinterpret_cast<const u32x4*>(lr + i * 8), v);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (__expf(v[e] - lse) - ((i * 8 + e) == tgt ? 1.0f : 0.0f)) * sc;
                *reinterpret_cast<u32x4*>(dr + i * 8) = pack8(o);
            }
            for (int64_t i = (nv << 3) + threadIdx.x; i < V; i += 256)
                dr[i] = f2bf((__expf(bf2f(lr[i]) - lse) - (i == tgt ? 1.0f : 0.0f)) * sc);
        }
    }
}

// The same row pass with the row kept in registers: one 1024-thread block per row, thread t holds the 16-byte chunks t, t + 1024, ... (NCH of
// them: 76 registers for a vocabulary up to 155 648), so the logits are read from memory ONCE -- maximum, sum of exponentials and the gradient all come
// from the registers -- and a training step moves 20 GB through this kernel instead of 30.  Needs V % 8 == 0; anything else takes ce_rows_kernel.
// Every index into the chunk array is a compile-time constant (dynamic indexing would put the array in scratch).
template <int NCH>
__global__ __launch_bounds__(1024) void ce_rows_regs_kernel(int64_t rows, int64_t V, const bf16_t* __restrict__ logits, int64_t ldl,
                                                            const int64_t* __restrict__ targets, float* __restrict__ loss_rows,
                                                            bf16_t* dlogits, const float* __restrict__ grad_scale) {
    __shared__ float red_m[16], red_s[16];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int nv = (int)(V >> 3);
    const unsigned tid8 = threadIdx.x * 8;
    for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
        const int64_t tgt = targets[row];
        const bf16_t* lr = logits + row * ldl;
        bf16_t* dr = dlogits ? dlogits + row * ldl : nullptr;
        if (tgt < 0 || tgt >= V) {  // ignore_index (-100): zero loss, zero gradient; any other value out of range: NaN loss
            if (threadIdx.x == 0) loss_rows[row] = tgt == -100 ? 0.f : __builtin_nanf("");
            if (dr)
                for (int i = threadIdx.x; i < nv; i += 1024) *reinterpret_cast<u32x4*>(dr + (int64_t)i * 8) = (u32x4){0, 0, 0, 0};
            continue;
        }
        const float tgt_logit = threadIdx.x == 0 ? bf2f(lr[tgt]) : 0.f;  // before the barriers: in place, the row is overwritten behind them
        u32x4 r[NCH];
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int i = threadIdx.x + 1024 * j;
            // (uniform row pointer + j * 16 KiB) + one 32-bit lane offset: 19 scalar bases and ONE offset register instead of 19 64-bit lane addresses
            r[j] = i < nv ? *reinterpret_cast<const u32x4*>(lr + (int64_t)j * 8192 + tid8) : (u32x4){0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u};  // -inf pairs
        }
        float m = -INFINITY;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            float v[8];
            asm volatile("" : "+v"(r[j]));  // opaque: without it the 152 unpacked values of the first pass are kept for the other two (and spilled)
            unpack8(r[j], v);
#pragma unroll
            for (int e = 0; e < 8; ++e) m = fmaxf(m, v[e]);
            __builtin_amdgcn_sched_barrier(0);  // one chunk at a time: left alone, the scheduler unpacks all 152 values first and spills the row
        }
        m = wave_max(m);
        __syncthreads();  // protects red_* reuse across rows
        if (lane == 0) red_m[wid] = m;
        __syncthreads();
        float bm = red_m[0];
#pragma unroll
        for (int wv = 1; wv < 16; ++wv) bm = fmaxf(bm, red_m[wv]);
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            float v[8];
            asm volatile("" : "+v"(r[j]));  // opaque: without it the 152 unpacked values of the first pass are kept for the other two (and spilled)
            unpack8(r[j], v);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += __expf(v[e] - bm);  // exp(-inf) = 0 for the padding of the last chunks
            __builtin_amdgcn_sched_barrier(0);
        }
        s = wave_sum(s);
        if (lane == 0) red_s[wid] = s;
        __syncthreads();
        float bs = 0.f;
#pragma unroll
        for (int wv = 0; wv < 16; ++wv) bs += red_s[wv];
        const float lse = bm + __logf(bs);
        if (threadIdx.x == 0) loss_rows[row] = lse - tgt_logit;
        if (dr) {
            const float sc = *grad_scale;
            const int tch = (int)(tgt >> 3), te = (int)(tgt & 7);  // 32-bit chunk / element of the target (64-bit per-element indices cost two registers each)
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const int i = threadIdx.x + 1024 * j;
                float v[8], o[8];
                asm volatile("" : "+v"(r[j]));
                unpack8(r[j], v);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (__expf(v[e] - lse) - ((i == tch && e == te) ? 1.0f : 0.0f)) * sc;
                if (i < nv) *reinterpret_cast<u32x4*>(dr + (int64_t)j * 8192 + tid8) = pack8(o);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

// One block of 1024 threads, four independent row loads in flight per thread (round 6: 256 threads walking 320 dependent iterations took 233 us -- on the step's critical
// path between the loss rows and the backward -- this form ~30); a fixed summation order: per thread, per wave, then the sixteen waves in order.
__global__ __launch_bounds__(1024) void ce_finalize_kernel(int64_t rows, const float* __restrict__ loss_rows,
                                                           const int64_t* __restrict__ targets, float* __restrict__ out3) {
    __shared__ float rs[16], rc[16];
    float s4[4] = {0.f, 0.f, 0.f, 0.f}, c4[4] = {0.f, 0.f, 0.f, 0.f};
    for (int64_t i0 = threadIdx.x; i0 < rows; i0 += 4096) {
        int64_t tg[4];
        float lv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t i = i0 + 1024 * u;
            tg[u] = i < rows ? targets[i] : -100;
            lv[u] = i < rows ? loss_rows[i] : 0.f;  // (read unconditionally: an ignored row's loss is a finite number or NaN that is not added)
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (tg[u] != -100) {  // an out-of-range target carries a NaN row loss: the mean shows it
                s4[u] += lv[u];
                c4[u] += 1.f;
            }
    }
    float s = (s4[0] + s4[1]) + (s4[2] + s4[3]), c = (c4[0] + c4[1]) + (c4[2] + c4[3]);
    s = wave_sum(s);
    c = wave_sum(c);
    if ((threadIdx.x & 63) == 0) {
        rs[threadIdx.x >> 6] = s;
        rc[threadIdx.x >> 6] = c;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float ts = 0.f, tc = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) ts += rs[w], tc += rc[w];
        out3[0] = ts / tc;  // NaN when every target is ignored, like F.cross_entropy
        out3[1] = tc;
        out3[2] = 1.0f / tc;
    }
}

// --------------------------------------------------------------------------------------------- embedding
__global__ __launch_bounds__(256) void embedding_fwd_kernel(int64_t tokens, int width, int64_t vocab, const int64_t* __restrict__ ids,
                                                            const bf16_t* __restrict__ table, bf16_t* __restrict__ out, int64_t ldo) {
    const int wv = width >> 3;
    const int64_t total = tokens * wv;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t t = i / wv;
        const int c = (int)(i - t * wv) * 8;
        int64_t id = ids[t];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);  // host validates; clamp keeps a bad id from faulting
        *reinterpret_cast<u32x4*>(out + t * ldo + c) = *reinterpret_cast<const u32x4*>(table + id * width + c);
    }
}
// one wave per token; fp32 atomics, 256 contiguous bytes per wave-instruction (the shape the atomic units like)
__global__ __launch_bounds__(256) void embedding_bwd_kernel(int64_t tokens, int width, int64_t vocab, const int64_t* __restrict__ ids,
                                                            const bf16_t* __restrict__ dout, int64_t ldd, float* __restrict__ dtable) {
    const int lane = threadIdx.x & 63;
    for (int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); t < tokens; t += (int64_t)gridDim.x * 4) {
        const int64_t id = ids[t];
        if (id < 0 || id >= vocab) continue;
        for (int c = lane; c < width; c += 64) atomicAdd(dtable + id * width + c, bf2f(dout[t * ldd + c]));
    }
}

// Deterministic form: the ids arrive SORTED (stably: equal ids keep the order of their token positions) together with the permutation that
// sorted them.  Every vocabulary row that occurs is written once with the fp32 sum of its tokens' rows: no atomics, the same bits every time
// whatever the ids repeat, and only the touched rows move (the atomic form walks a dense fp32 copy of the table).
// A run of equal ids can be long (padding or placeholder tokens: thousands of positions), so the sum is a fixed two-level tree over blocks of
// EMB_BLK sorted positions, whatever the launch geometry: pass 1, one wave per block, sums each run's part inside the block in position order --
// a run that lies inside one block is finished there; a part that began before the block goes to scratch slot 0 of the block, a part that goes on
// past it to slot 1 -- and pass 2, one wave per block whose slot 1 is in use (the block that holds the run's first position), adds the slot-0
// parts of the following blocks in block order and writes the row.  scale: 1 / world under data parallelism.
constexpr int EMB_BLK = 32;
__device__ __forceinline__ void emb_store_row(bf16_t* row, const float (&acc)[8], float scale, int accumulate) {
    u32x4 o;
    if (accumulate) {
        const u32x4 old = *reinterpret_cast<const u32x4*>(row);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = pack_bf2(__uint_as_float(old[e] << 16) + acc[2 * e] * scale, __uint_as_float(old[e] & 0xffff0000u) + acc[2 * e + 1] * scale);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = pack_bf2(acc[2 * e] * scale, acc[2 * e + 1] * scale);
    }
    *reinterpret_cast<u32x4*>(row) = o;
}

__global__ __launch_bounds__(256) void embedding_bwd_sorted_kernel(int64_t tokens, int width, int64_t vocab, const int64_t* __restrict__ sid,
                                                                   const int64_t* __restrict__ perm, const bf16_t* __restrict__ dout, int64_t ldd,
                                                                   float scale, bf16_t* __restrict__ table, int64_t ldt, int accumulate,
                                                                   float* __restrict__ part) {
    const int lane = threadIdx.x & 63;
    const int64_t nblk = (tokens + EMB_BLK - 1) / EMB_BLK;
    for (int64_t blk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); blk < nblk; blk += (int64_t)gridDim.x * 4) {
        const int64_t lo = blk * EMB_BLK, hi = min(tokens, lo + EMB_BLK);
        int64_t j = lo;
        while (j < hi) {
            const int64_t id = sid[j];
            int64_t e = j + 1;
            while (e < hi && sid[e] == id) ++e;
            const bool before = j == lo && lo > 0 && sid[lo - 1] == id, after = e == hi && hi < tokens && sid[hi] == id;
            if (id >= 0 && id < vocab) {
                for (int c0 = 0; c0 < width; c0 += 512) {  // 8 columns per lane and sweep
                    const int c = c0 + lane * 8;
                    if (c >= width) continue;
                    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    for (int64_t jj = j; jj < e; jj += 8) {  // eight rows requested together, added in position order
                        u32x4 v[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) v[u] = jj + u < e ? *reinterpret_cast<const u32x4*>(dout + perm[jj + u] * ldd + c) : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
                        for (int u = 0; u < 8; ++u)
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                acc[2 * q] += __uint_as_float(v[u][q] << 16);
                                acc[2 * q + 1] += __uint_as_float(v[u][q] & 0xffff0000u);
                            }
                    }
                    if (!before && !after) {
                        emb_store_row(table + id * ldt + c, acc, scale, accumulate);
                    } else {
                        float* dst = part + ((blk * 2 + (before ? 0 : 1)) * (int64_t)width + c);
                        *reinterpret_cast<f32x4*>(dst) = f32x4{acc[0], acc[1], acc[2], acc[3]};
                        *reinterpret_cast<f32x4*>(dst + 4) = f32x4{acc[4], acc[5], acc[6], acc[7]};
                    }
                }
            }
            j = e;
        }
    }
}

__global__ __launch_bounds__(256) void embedding_bwd_join_kernel(int64_t tokens, int width, int64_t vocab, const int64_t* __restrict__ sid, float scale,
                                                                 bf16_t* __restrict__ table, int64_t ldt, int accumulate, const float* __restrict__ part) {
    const int lane = threadIdx.x & 63;
    const int64_t nblk = (tokens + EMB_BLK - 1) / EMB_BLK;
    for (int64_t blk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); blk + 1 < nblk; blk += (int64_t)gridDim.x * 4) {
        const int64_t lo = blk * EMB_BLK, hi = lo + EMB_BLK;  // a block with a successor is full
        const int64_t id = sid[hi - 1];
        if (sid[hi] != id || id < 0 || id >= vocab) continue;             // the last run of the block does not go on
        if (sid[lo] == id && lo > 0 && sid[lo - 1] == id) continue;      // ... or it began before the block: an earlier block owns it
        // blocks blk + 1 .. last hold the run's further parts in their slot 0; the run ends in the first block it does not fill to the end
        int64_t last = blk + 1;
        while (last + 1 < nblk && sid[(last + 1) * EMB_BLK - 1] == id && sid[(last + 1) * EMB_BLK] == id) ++last;
        for (int c0 = 0; c0 < width; c0 += 512) {
            const int c = c0 + lane * 8;
            if (c >= width) continue;
            float acc[8];
            {
                const float* src = part + ((blk * 2 + 1) * (int64_t)width + c);
                const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
                acc[0] = a[0]; acc[1] = a[1]; acc[2] = a[2]; acc[3] = a[3]; acc[4] = b[0]; acc[5] = b[1]; acc[6] = b[2]; acc[7] = b[3];
            }
            for (int64_t nb = blk + 1; nb <= last; nb += 4) {  // four parts requested together, added in block order
                f32x4 a[4], b[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float* src = part + (((nb + u) * 2) * (int64_t)width + c);
                    a[u] = nb + u <= last ? *reinterpret_cast<const f32x4*>(src) : f32x4{0.f, 0.f, 0.f, 0.f};
                    b[u] = nb + u <= last ? *reinterpret_cast<const f32x4*>(src + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        acc[q] += a[u][q];
                        acc[4 + q] += b[u][q];
                    }
            }
            emb_store_row(table + id * ldt + c, acc, scale, accumulate);
        }
    }
}

// ------------------------------------------------------------------------------------------- strided copy
// y[c][r] = x[r][c] for a bf16 matrix, 64x64 tiles through LDS: 16-byte loads along x's rows, 16-byte stores along y's rows.
// Used once per weight and backward pass: dgrad GEMMs then read W^T with K contiguous (NT form) instead of the K-strided NN form.
__global__ __launch_bounds__(256) void transpose_bf16_kernel(int64_t rows, int64_t cols, const unsigned short* __restrict__ x, int64_t ldx,
                                                             unsigned short* __restrict__ y, int64_t ldy) {
    __shared__ unsigned short t[64][72];  // [c][r], row pitch 144 B: 16-byte aligned rows, bank-spread columns
    const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
    const int tid = threadIdx.x;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int r = (tid >> 3) + h * 32, ch = tid & 7;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (r0 + r < rows && c0 + ch * 8 < cols) v = *reinterpret_cast<const u32x4*>(x + (r0 + r) * ldx + c0 + ch * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            t[ch * 8 + 2 * e][r] = (unsigned short)(v[e] & 0xffffu);
            t[ch * 8 + 2 * e + 1][r] = (unsigned short)(v[e] >> 16);
        }
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int c = (tid >> 3) + h * 32, rh = tid & 7;
        if (c0 + c < cols && r0 + rh * 8 < rows) *reinterpret_cast<u32x4*>(y + (c0 + c) * ldy + r0 + rh * 8) = *reinterpret_cast<const u32x4*>(&t[c][rh * 8]);
    }
}

__global__ __launch_bounds__(256) void copy2d_kernel(int64_t rows, int64_t wvec, const char* __restrict__ src, int64_t sp,
                                                     char* __restrict__ dst, int64_t dp) {
    const int64_t total = rows * wvec;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / wvec, c = (i - r * wvec) * 16;
        *reinterpret_cast<u32x4*>(dst + r * dp + c) = *reinterpret_cast<const u32x4*>(src + r * sp + c);
    }
}
__global__ __launch_bounds__(256) void copy2d_bytes_kernel(int64_t rows, int64_t w, const char* __restrict__ src, int64_t sp,
                                                           char* __restrict__ dst, int64_t dp) {
    const int64_t total = rows * w;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / w, c = i - r * w;
        dst[r * dp + c] = src[r * sp + c];
    }
}

// ------------------------------------------------------------------------------------------- patch gather
// rows[(b*gh+ph)*gw+pw][(c*P+i)*P+j] = img[b][c][ph*P+i][pw*P+j].  One thread moves 4 consecutive j (16-B read).
template <int OUT_DT>
__global__ __launch_bounds__(256) void patchify_kernel(int B, int C, int H, int W, int P, const float* __restrict__ img, void* __restrict__ rows) {
    const int gh = H / P, gw = W / P, K = C * P * P, pv = P >> 2;
    const int64_t total = (int64_t)B * gh * gw * C * P * pv;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        // order the work so consecutive threads read consecutive image addresses: (b, c, y, pw, jv)
        int64_t r = i;
        const int jv = (int)(r % pv); r /= pv;
        const int pw = (int)(r % gw); r /= gw;
        const int y = (int)(r % H); r /= H;
        const int c = (int)(r % C); r /= C;
        const int b = (int)r;
        const int ph = y / P, ii = y - ph * P;
        const f32x4 v = *reinterpret_cast<const f32x4*>(img + (((int64_t)b * C + c) * H + y) * W + pw * P + jv * 4);
        const int64_t orow = ((int64_t)b * gh + ph) * gw + pw;
        const int k = (c * P + ii) * P + jv * 4;
        if constexpr (OUT_DT == MI355_DT_BF16) {
            u32x2 pk = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
            *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(rows) + orow * K + k) = pk;
        } else {
            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(rows) + orow * K + k) = v;
        }
    }
}

// out[b, 0, :] = cls + pos[0];  out[b, 1+p, :] = proj[b*Np + p, :] + pos[1+p]
__global__ __launch_bounds__(256) void vit_embed_assemble_kernel(int B, int S, int width, const float* __restrict__ proj,
                                                                 const float* __restrict__ cls, const float* __restrict__ pos,
                                                                 float* __restrict__ out) {
    const int wv = width >> 2;
    const int64_t total = (int64_t)B * S * wv;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % wv) * 4;
        const int64_t bs = i / wv;
        const int s = (int)(bs % S);
        const int64_t b = bs / S;
        const f32x4 pe = *reinterpret_cast<const f32x4*>(pos + (int64_t)s * width + c);
        const f32x4 base = s == 0 ? *reinterpret_cast<const f32x4*>(cls + c)
                                  : *reinterpret_cast<const f32x4*>(proj + (b * (S - 1) + (s - 1)) * width + c);
        *reinterpret_cast<f32x4*>(out + bs * width + c) = base + pe;
    }
}

// --------------------------------------------------------------------------------------------------- casts
__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(int64_t n, const float* __restrict__ s, bf16_t* __restrict__ d) {
    const int64_t nv = n >> 3;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (int64_t)gridDim.x * 256) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(s + i * 8), b = *reinterpret_cast<const f32x4*>(s + i * 8 + 4);
        *reinterpret_cast<u32x4*>(d + i * 8) = (u32x4){pack_bf2(a[0], a[1]), pack_bf2(a[2], a[3]), pack_bf2(b[0], b[1]), pack_bf2(b[2], b[3])};
    }
    for (int64_t i = (nv << 3) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) d[i] = f2bf(s[i]);
}
__global__ __launch_bounds__(256) void cast_bf16_f32_kernel(int64_t n, const bf16_t* __restrict__ s, float* __restrict__ d) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) d[i] = bf2f(s[i]);
}
__global__ __launch_bounds__(256) void add_f32_to_bf16_kernel(int64_t n, const float* __restrict__ a, const bf16_t* __restrict__ b, bf16_t* __restrict__ d) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        d[i] = f2bf(a[i] + (b ? bf2f(b[i]) : 0.f));
}

__global__ __launch_bounds__(256) void scale_bf16_kernel(int64_t n, const bf16_t* __restrict__ x, const float* __restrict__ scale, bf16_t* __restrict__ y) {
    const float sc = *scale;
    const int64_t nv = n >> 3;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (int64_t)gridDim.x * 256) {
        float v[8];
        unpack8(*reinterpret_cast<const u32x4*>(x + i * 8), v);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= sc;
        *reinterpret_cast<u32x4*>(y + i * 8) = pack8(v);
    }
    for (int64_t i = (nv << 3) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[i] = f2bf(bf2f(x[i]) * sc);
}

// Global squared norm, bit-reproducible: every block leaves ONE partial in a per-device scratch row, a second one-wave kernel adds the
// partials in a fixed order and accumulates into *out (no float atomics: their arrival order would move the last bits of the clip
// factor from run to run).  The partials live in a caller-provided buffer (MI355_SUMSQ_PARTS floats): one per stream, so that calls on
// different streams cannot race for it.
constexpr int SUMSQ_MAX_PARTS = MI355_SUMSQ_PARTS;

__global__ __launch_bounds__(64) void sumsq_finish_kernel(int parts, const float* __restrict__ part, float* __restrict__ out) {
    float s = 0.f;
    for (int i = threadIdx.x; i < parts; i += 64) s += part[i];
    s = wave_sum(s);
    if (threadIdx.x == 0) *out += s;
}

template <int DT>
__global__ __launch_bounds__(256) void sumsq_kernel(int64_t n, const void* __restrict__ x, float* __restrict__ part) {
    __shared__ float red[4];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float v = DT == MI355_DT_BF16 ? bf2f(reinterpret_cast<const bf16_t*>(x)[i]) : reinterpret_cast<const float*>(x)[i];
        s += v * v;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
template <int DT>
__global__ __launch_bounds__(256) void clip_scale_kernel(int64_t n, void* __restrict__ x, const float* __restrict__ sumsq, float max_norm) {
    const float coef = fminf(1.0f, max_norm / (sqrtf(*sumsq) + 1e-6f));
    if (coef >= 1.0f) return;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        if (DT == MI355_DT_BF16) {
            bf16_t* p = reinterpret_cast<bf16_t*>(x);
            p[i] = f2bf(bf2f(p[i]) * coef);
        } else {
            reinterpret_cast<float*>(x)[i] *= coef;
        }
    }
}

// AdamW over one flat buffer (a parameter arena): decoupled weight decay, fp32 moments, parameter and gradient in their own
// dtype (bf16 or fp32); the optional device scalar `sumsq` applies the global-norm clip coefficient to the gradient on the
// fly (torch.nn.utils.clip_grad_norm_ semantics) so the clip never costs a pass of its own.
template <int P_DT, int G_DT>
__global__ __launch_bounds__(256) void adamw_kernel(int64_t n, void* __restrict__ param, const void* __restrict__ grad,
                                                    float* __restrict__ exp_avg, float* __restrict__ exp_avg_sq, float lr, float beta1,
                                                    float beta2, float eps, float weight_decay, float bc1, float bc2_sqrt,
                                                    const float* __restrict__ sumsq, float max_norm) {
    const float coef = sumsq ? fminf(1.0f, max_norm / (sqrtf(*sumsq) + 1e-6f)) : 1.0f;
    const float step = lr / bc1, decay = 1.0f - lr * weight_decay;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float g = (G_DT == MI355_DT_BF16 ? bf2f(reinterpret_cast<const bf16_t*>(grad)[i]) : reinterpret_cast<const float*>(grad)[i]) * coef;
        float p = P_DT == MI355_DT_BF16 ? bf2f(reinterpret_cast<bf16_t*>(param)[i]) : reinterpret_cast<float*>(param)[i];
        const float m = beta1 * exp_avg[i] + (1.0f - beta1) * g;
        const float v = beta2 * exp_avg_sq[i] + (1.0f - beta2) * g * g;
        exp_avg[i] = m;
        exp_avg_sq[i] = v;
        p = p * decay - step * m / (sqrtf(v) / bc2_sqrt + eps);
        if (P_DT == MI355_DT_BF16) reinterpret_cast<bf16_t*>(param)[i] = f2bf(p);
        else reinterpret_cast<float*>(param)[i] = p;
    }
}

}  // namespace

#define STREAM ((hipStream_t)stream)

extern "C" int mi355_swiglu_fwd(int64_t tokens, int F, const void* gu, void* a, void* stream) {
    MI355_REQUIRE(tokens > 0 && F > 0 && (F & 7) == 0 && gu && a, "mi355_swiglu_fwd: F must be a multiple of 8 and pointers non-null");
    hipLaunchKernelGGL(swiglu_fwd_kernel, dim3(grid_for(tokens * (F >> 3), 256)), dim3(256), 0, STREAM, tokens, F, (const bf16_t*)gu, (bf16_t*)a);
    MI355_LAUNCH_CHECK("mi355_swiglu_fwd");
    return 0;
}
extern "C" int mi355_swiglu_bwd(int64_t tokens, int F, const void* gu, const void* da, void* dgu, void* stream) {
    MI355_REQUIRE(tokens > 0 && F > 0 && (F & 7) == 0 && gu && da && dgu, "mi355_swiglu_bwd: bad arguments");
    hipLaunchKernelGGL(swiglu_bwd_kernel, dim3(grid_for(tokens * (F >> 3), 256)), dim3(256), 0, STREAM, tokens, F, (const bf16_t*)gu, (const bf16_t*)da, (bf16_t*)dgu);
    MI355_LAUNCH_CHECK("mi355_swiglu_bwd");
    return 0;
}

extern "C" int mi355_gelu_fwd(int64_t n, const void* x, void* y, int kind, void* stream) {
    MI355_REQUIRE(n > 0 && (n & 7) == 0 && x && y && (kind == 0 || kind == 1), "mi355_gelu_fwd: n must be a positive multiple of 8, kind 0 (erf) or 1 (tanh)");
    if (kind == 0) hipLaunchKernelGGL(gelu_fwd_kernel<0>, dim3(grid_for(n >> 3, 256)), dim3(256), 0, STREAM, n, (const bf16_t*)x, (bf16_t*)y);
    else hipLaunchKernelGGL(gelu_fwd_kernel<1>, dim3(grid_for(n >> 3, 256)), dim3(256), 0, STREAM, n, (const bf16_t*)x, (bf16_t*)y);
    MI355_LAUNCH_CHECK("mi355_gelu_fwd");
    return 0;
}
extern "C" int mi355_gelu_bwd(int64_t n, const void* x, const void* dy, void* dx, int kind, void* stream) {
    MI355_REQUIRE(n > 0 && (n & 7) == 0 && x && dy && dx && (kind == 0 || kind == 1), "mi355_gelu_bwd: n must be a positive multiple of 8, kind 0 or 1");
    if (kind == 0) hipLaunchKernelGGL(gelu_bwd_kernel<0>, dim3(grid_for(n >> 3, 256)), dim3(256), 0, STREAM, n, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx);
    else hipLaunchKernelGGL(gelu_bwd_kernel<1>, dim3(grid_for(n >> 3, 256)), dim3(256), 0, STREAM, n, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx);
    MI355_LAUNCH_CHECK("mi355_gelu_bwd");
    return 0;
}

extern "C" int mi355_patchify3d(int B, int C, int T, int H, int W, int P, int TP, const float* img, void* rows, int out_dtype, void* stream) {
    MI355_REQUIRE(B > 0 && C > 0 && T > 0 && P > 0 && TP > 0 && (P & 3) == 0 && H % P == 0 && W % P == 0 && T % TP == 0 && img && rows,
                  "mi355_patchify3d: patch size must be a multiple of 4 and divide H, W; TP must divide T");
    MI355_REQUIRE(((uintptr_t)img & 15) == 0, "mi355_patchify3d: image must be 16-byte aligned");
    const int64_t work = (int64_t)B * C * T * H * (W / 4);
    if (out_dtype == MI355_DT_BF16)
        hipLaunchKernelGGL(patchify3d_kernel<MI355_DT_BF16>, dim3(grid_for(work, 256)), dim3(256), 0, STREAM, B, C, T, H, W, P, TP, img, rows);
    else
        hipLaunchKernelGGL(patchify3d_kernel<MI355_DT_F32>, dim3(grid_for(work, 256)), dim3(256), 0, STREAM, B, C, T, H, W, P, TP, img, rows);
    MI355_LAUNCH_CHECK("mi355_patchify3d");
    return 0;
}

extern "C" int mi355_merge_patches(int64_t frames, int gh, int gw, int m, int64_t row_bytes, const void* src, void* dst, int inverse, void* stream) {
    MI355_REQUIRE(frames > 0 && gh > 0 && gw > 0 && m > 0 && gh % m == 0 && gw % m == 0 && row_bytes > 0 && (row_bytes & 15) == 0 && src && dst,
                  "mi355_merge_patches: merge size must divide the patch grid; rows must be multiples of 16 bytes");
    hipLaunchKernelGGL(merge_patches_kernel, dim3(grid_for(frames * gh * gw * (row_bytes >> 4), 256)), dim3(256), 0, STREAM, frames, gh, gw, m, (int)row_bytes, (const char*)src, (char*)dst, inverse);
    MI355_LAUNCH_CHECK("mi355_merge_patches");
    return 0;
}

extern "C" int mi355_scatter_rows(int64_t tokens, int64_t row_bytes, const uint8_t* mask, const int32_t* slot, const void* a, const void* b,
                                  void* out_a, void* out_b, int backward, void* stream) {
    MI355_REQUIRE(tokens > 0 && row_bytes > 0 && (row_bytes & 15) == 0 && mask && slot && a && out_a, "mi355_scatter_rows: bad arguments (rows must be multiples of 16 bytes)");
    MI355_REQUIRE(backward ? out_b != nullptr : b != nullptr, "mi355_scatter_rows: missing vision-row buffer");
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(grid_for(tokens * (row_bytes >> 4), 256)), dim3(256), 0, STREAM, tokens, (int)row_bytes, mask, slot,
                       (const char*)a, (const char*)b, (char*)out_a, (char*)out_b, backward);
    MI355_LAUNCH_CHECK("mi355_scatter_rows");
    return 0;
}

extern "C" int mi355_cross_entropy(int64_t rows, int64_t V, const void* logits, int64_t ldl, const int64_t* targets,
                                   float* loss_rows, void* dlogits, const float* grad_scale, void* stream) {
    MI355_REQUIRE(rows > 0 && V > 0 && logits && targets && loss_rows, "mi355_cross_entropy: bad arguments");
    MI355_REQUIRE((ldl & 7) == 0 && ((uintptr_t)logits & 15) == 0, "mi355_cross_entropy: logits rows must be 16-byte aligned");
    MI355_REQUIRE(dlogits == nullptr || grad_scale != nullptr, "mi355_cross_entropy: dlogits needs grad_scale");
    const int64_t chunks = V >> 3;
    const unsigned grid = (unsigned)(rows < 65535 ? rows : 65535);
    static const bool regs_form = [] { const char* e = getenv("MI355_CE_ROW_IN_REGISTERS"); return !(e && e[0] == '0'); }();  // 0: the two-read kernel (A/B)
    if ((V & 7) == 0 && regs_form && chunks > 4 * 1024 && chunks <= 19 * 1024)  // large vocabularies: the row stays in registers, one read
        hipLaunchKernelGGL(ce_rows_regs_kernel<19>, dim3(grid), dim3(1024), 0, STREAM, rows, V, (const bf16_t*)logits, ldl, targets, loss_rows, (bf16_t*)dlogits, grad_scale);
    else if ((V & 7) == 0 && regs_form && chunks > 1024 && chunks <= 4 * 1024)
        hipLaunchKernelGGL(ce_rows_regs_kernel<4>, dim3(grid), dim3(1024), 0, STREAM, rows, V, (const bf16_t*)logits, ldl, targets, loss_rows, (bf16_t*)dlogits, grad_scale);
    else
        hipLaunchKernelGGL(ce_rows_kernel, dim3(grid), dim3(256), 0, STREAM, rows, V, (const bf16_t*)logits, ldl, targets, loss_rows, (bf16_t*)dlogits, grad_scale);
    MI355_LAUNCH_CHECK("mi355_cross_entropy");
    return 0;
}
extern "C" int mi355_ce_finalize(int64_t rows, const float* loss_rows, const int64_t* targets, float* out3, void* stream) {
    MI355_REQUIRE(rows > 0 && loss_rows && targets && out3, "mi355_ce_finalize: bad arguments");
    hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(1024), 0, STREAM, rows, loss_rows, targets, out3);
    MI355_LAUNCH_CHECK("mi355_ce_finalize");
    return 0;
}

extern "C" int mi355_embedding_fwd(int64_t tokens, int width, int64_t vocab, const int64_t* ids, const void* table, void* out,
                                   int64_t ldo, void* stream) {
    MI355_REQUIRE(tokens > 0 && width > 0 && (width & 7) == 0 && (ldo & 7) == 0 && ids && table && out, "mi355_embedding_fwd: bad arguments (width/ldo multiple of 8)");
    hipLaunchKernelGGL(embedding_fwd_kernel, dim3(grid_for(tokens * (width >> 3), 256)), dim3(256), 0, STREAM, tokens, width, vocab, ids, (const bf16_t*)table, (bf16_t*)out, ldo);
    MI355_LAUNCH_CHECK("mi355_embedding_fwd");
    return 0;
}
extern "C" int mi355_embedding_bwd(int64_t tokens, int width, int64_t vocab, const int64_t* ids, const void* dout, int64_t ldd,
                                   float* dtable_f32, void* stream) {
    MI355_REQUIRE(tokens > 0 && width > 0 && ids && dout && dtable_f32, "mi355_embedding_bwd: bad arguments");
    hipLaunchKernelGGL(embedding_bwd_kernel, dim3(grid_for(tokens, 4)), dim3(256), 0, STREAM, tokens, width, vocab, ids, (const bf16_t*)dout, ldd, dtable_f32);
    MI355_LAUNCH_CHECK("mi355_embedding_bwd");
    return 0;
}

extern "C" int64_t mi355_embedding_bwd_sorted_workspace_bytes(int64_t tokens, int width) {
    return tokens > 0 && width > 0 ? ((tokens + EMB_BLK - 1) / EMB_BLK) * 2 * (int64_t)width * 4 : 0;
}

extern "C" int mi355_embedding_bwd_sorted(int64_t tokens, int width, int64_t vocab, const int64_t* sorted_ids, const int64_t* perm, const void* dout,
                                          int64_t ldd, float scale, void* table, int64_t ldt, int accumulate, float* workspace, int64_t workspace_bytes,
                                          void* stream) {
    MI355_REQUIRE(tokens > 0 && width > 0 && vocab > 0 && sorted_ids && perm && dout && table, "mi355_embedding_bwd_sorted: bad arguments");
    MI355_REQUIRE((width & 7) == 0 && (ldd & 7) == 0 && (ldt & 7) == 0 && (((uintptr_t)dout | (uintptr_t)table) & 15) == 0,
                  "mi355_embedding_bwd_sorted: width and both row pitches must be multiples of 8 elements, pointers 16-byte aligned");
    MI355_REQUIRE(workspace && ((uintptr_t)workspace & 15) == 0 && workspace_bytes >= mi355_embedding_bwd_sorted_workspace_bytes(tokens, width),
                  "mi355_embedding_bwd_sorted: a 16-byte aligned workspace of %lld bytes is needed (mi355_embedding_bwd_sorted_workspace_bytes)",
                  (long long)mi355_embedding_bwd_sorted_workspace_bytes(tokens, width));
    const int64_t nblk = (tokens + EMB_BLK - 1) / EMB_BLK;
    hipLaunchKernelGGL(embedding_bwd_sorted_kernel, dim3(grid_for(nblk, 4)), dim3(256), 0, STREAM, tokens, width, vocab, sorted_ids, perm, (const bf16_t*)dout, ldd,
                       scale, (bf16_t*)table, ldt, accumulate, workspace);
    if (nblk > 1)
        hipLaunchKernelGGL(embedding_bwd_join_kernel, dim3(grid_for(nblk - 1, 4)), dim3(256), 0, STREAM, tokens, width, vocab, sorted_ids, scale, (bf16_t*)table, ldt,
                           accumulate, (const float*)workspace);
    MI355_LAUNCH_CHECK("mi355_embedding_bwd_sorted");
    return 0;
}

extern "C" int mi355_transpose_bf16(int64_t rows, int64_t cols, const void* x, int64_t ldx, void* y, int64_t ldy, void* stream) {
    MI355_REQUIRE(rows > 0 && cols > 0 && x && y, "mi355_transpose_bf16: bad arguments");
    MI355_REQUIRE(((rows | cols | ldx | ldy) & 7) == 0 && ldx >= cols && ldy >= rows && (((uintptr_t)x | (uintptr_t)y) & 15) == 0,
                  "mi355_transpose_bf16: rows, cols and both pitches must be multiples of 8 elements, pointers 16-byte aligned");
    MI355_REQUIRE((rows + 63) / 64 <= 65535, "mi355_transpose_bf16: too many rows");
    hipLaunchKernelGGL(transpose_bf16_kernel, dim3((unsigned)((cols + 63) / 64), (unsigned)((rows + 63) / 64)), dim3(256), 0, STREAM, rows, cols,
                       (const unsigned short*)x, ldx, (unsigned short*)y, ldy);
    MI355_LAUNCH_CHECK("mi355_transpose_bf16");
    return 0;
}

extern "C" int mi355_copy2d(int64_t rows, int64_t width_bytes, const void* src, int64_t src_pitch_bytes, void* dst,
                            int64_t dst_pitch_bytes, void* stream) {
    MI355_REQUIRE(rows > 0 && width_bytes > 0 && src && dst, "mi355_copy2d: bad arguments");
    const bool vec = ((width_bytes | src_pitch_bytes | dst_pitch_bytes) & 15) == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0;
    if (vec)
        hipLaunchKernelGGL(copy2d_kernel, dim3(grid_for(rows * (width_bytes >> 4), 256)), dim3(256), 0, STREAM, rows, width_bytes >> 4, (const char*)src, src_pitch_bytes, (char*)dst, dst_pitch_bytes);
    else
        hipLaunchKernelGGL(copy2d_bytes_kernel, dim3(grid_for(rows * width_bytes, 256)), dim3(256), 0, STREAM, rows, width_bytes, (const char*)src, src_pitch_bytes, (char*)dst, dst_pitch_bytes);
    MI355_LAUNCH_CHECK("mi355_copy2d");
    return 0;
}

extern "C" int mi355_patchify(int B, int C, int H, int W, int P, const float* img, void* rows, int out_dtype, void* stream) {
    MI355_REQUIRE(B > 0 && C > 0 && P > 0 && (P & 3) == 0 && H % P == 0 && W % P == 0 && img && rows, "mi355_patchify: patch size must be a multiple of 4 and divide H and W");
    MI355_REQUIRE(((uintptr_t)img & 15) == 0, "mi355_patchify: image must be 16-byte aligned");
    const int64_t work = (int64_t)B * C * H * (W / 4);
    if (out_dtype == MI355_DT_BF16)
        hipLaunchKernelGGL(patchify_kernel<MI355_DT_BF16>, dim3(grid_for(work, 256)), dim3(256), 0, STREAM, B, C, H, W, P, img, rows);
    else
        hipLaunchKernelGGL(patchify_kernel<MI355_DT_F32>, dim3(grid_for(work, 256)), dim3(256), 0, STREAM, B, C, H, W, P, img, rows);
    MI355_LAUNCH_CHECK("mi355_patchify");
    return 0;
}

extern "C" int mi355_vit_embed_assemble(int B, int S, int width, const float* patch_proj, const float* cls, const float* pos,
                                        float* out, void* stream) {
    MI355_REQUIRE(B > 0 && S > 1 && width > 0 && (width & 3) == 0 && patch_proj && cls && pos && out, "mi355_vit_embed_assemble: bad arguments");
    hipLaunchKernelGGL(vit_embed_assemble_kernel, dim3(grid_for((int64_t)B * S * (width >> 2), 256)), dim3(256), 0, STREAM, B, S, width, patch_proj, cls, pos, out);
    MI355_LAUNCH_CHECK("mi355_vit_embed_assemble");
    return 0;
}

extern "C" int mi355_cast(int64_t n, const void* src, int src_dtype, void* dst, int dst_dtype, void* stream) {
    MI355_REQUIRE(n > 0 && src && dst && src_dtype != dst_dtype, "mi355_cast: bad arguments");
    if (src_dtype == MI355_DT_F32)
        hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid_for(n >> 3, 256)), dim3(256), 0, STREAM, n, (const float*)src, (bf16_t*)dst);
    else
        hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(grid_for(n, 256)), dim3(256), 0, STREAM, n, (const bf16_t*)src, (float*)dst);
    MI355_LAUNCH_CHECK("mi355_cast");
    return 0;
}

extern "C" int mi355_add_f32_to_bf16(int64_t n, const float* a, const void* b_bf16, void* dst_bf16, void* stream) {
    MI355_REQUIRE(n > 0 && a && dst_bf16, "mi355_add_f32_to_bf16: bad arguments");
    hipLaunchKernelGGL(add_f32_to_bf16_kernel, dim3(grid_for(n, 256)), dim3(256), 0, STREAM, n, a, (const bf16_t*)b_bf16, (bf16_t*)dst_bf16);
    MI355_LAUNCH_CHECK("mi355_add_f32_to_bf16");
    return 0;
}

extern "C" int mi355_scale_bf16(int64_t n, const void* x, const float* scale, void* y, void* stream) {
    MI355_REQUIRE(n > 0 && x && scale && y, "mi355_scale_bf16: bad arguments");
    hipLaunchKernelGGL(scale_bf16_kernel, dim3(grid_for(n >> 3, 256)), dim3(256), 0, STREAM, n, (const bf16_t*)x, scale, (bf16_t*)y);
    MI355_LAUNCH_CHECK("mi355_scale_bf16");
    return 0;
}

extern "C" int mi355_sumsq(int64_t n, const void* x, int dtype, float* out, float* partials, void* stream) {
    MI355_REQUIRE(n > 0 && x && out && partials, "mi355_sumsq: bad arguments");
    int parts = grid_for(n, 256 * 8);
    if (parts > SUMSQ_MAX_PARTS) parts = SUMSQ_MAX_PARTS;
    if (dtype == MI355_DT_BF16)
        hipLaunchKernelGGL(sumsq_kernel<MI355_DT_BF16>, dim3(parts), dim3(256), 0, STREAM, n, x, partials);
    else
        hipLaunchKernelGGL(sumsq_kernel<MI355_DT_F32>, dim3(parts), dim3(256), 0, STREAM, n, x, partials);
    hipLaunchKernelGGL(sumsq_finish_kernel, dim3(1), dim3(64), 0, STREAM, parts, partials, out);
    MI355_LAUNCH_CHECK("mi355_sumsq");
    return 0;
}
extern "C" int mi355_clip_scale(int64_t n, void* x, int dtype, const float* sumsq, float max_norm, void* stream) {
    MI355_REQUIRE(n > 0 && x && sumsq, "mi355_clip_scale: bad arguments");
    if (dtype == MI355_DT_BF16)
        hipLaunchKernelGGL(clip_scale_kernel<MI355_DT_BF16>, dim3(grid_for(n, 256 * 4)), dim3(256), 0, STREAM, n, x, sumsq, max_norm);
    else
        hipLaunchKernelGGL(clip_scale_kernel<MI355_DT_F32>, dim3(grid_for(n, 256 * 4)), dim3(256), 0, STREAM, n, x, sumsq, max_norm);
    MI355_LAUNCH_CHECK("mi355_clip_scale");
    return 0;
}

extern "C" int mi355_adamw(int64_t n, void* param, int p_dtype, const void* grad, int g_dtype, float* exp_avg, float* exp_avg_sq, float lr,
                           float beta1, float beta2, float eps, float weight_decay, int step, const float* sumsq, float max_norm,
                           void* stream) {
    MI355_REQUIRE(n > 0 && param && grad && exp_avg && exp_avg_sq && step >= 1, "mi355_adamw: bad arguments");
    MI355_REQUIRE((p_dtype == MI355_DT_BF16 || p_dtype == MI355_DT_F32) && (g_dtype == MI355_DT_BF16 || g_dtype == MI355_DT_F32), "mi355_adamw: bad dtype");
    const float bc1 = 1.0f - powf(beta1, (float)step), bc2_sqrt = sqrtf(1.0f - powf(beta2, (float)step));
    const dim3 grid(grid_for(n, 256 * 4));
#define ADAMW(P, G) hipLaunchKernelGGL((adamw_kernel<P, G>), grid, dim3(256), 0, STREAM, n, param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, bc1, bc2_sqrt, sumsq, max_norm)
    if (p_dtype == MI355_DT_BF16 && g_dtype == MI355_DT_BF16) ADAMW(MI355_DT_BF16, MI355_DT_BF16);
    else if (p_dtype == MI355_DT_BF16) ADAMW(MI355_DT_BF16, MI355_DT_F32);
    else if (g_dtype == MI355_DT_BF16) ADAMW(MI355_DT_F32, MI355_DT_BF16);
    else ADAMW(MI355_DT_F32, MI355_DT_F32);
#undef ADAMW
    MI355_LAUNCH_CHECK("mi355_adamw");
    return 0;
}
