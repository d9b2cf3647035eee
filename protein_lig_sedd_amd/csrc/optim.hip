// Optimiser pass over the flat fp32 parameter buffer: clip -> AdamW -> EMA -> bf16 compute copy, and the
// deterministic gradient-norm reduction that feeds the clip coefficient.
// Reference: protlig_dd/processing/losses.py:35-69 (AdamW, warm-up, clip_grad_norm_), protlig_dd/model/ema.py:32-52.
// HBM-bound: reads p,g,m,v,ema (20 B/param), writes p,m,v,ema (+2 B bf16 copy).
#include "common.h"

#define TPB 256

// g * cs + l2 * p with the rounding points fixed (the product g * cs rounded, then one fma), so that the one- and the
// two-step instantiations below cannot contract it differently
__device__ __forceinline__ float l2_grad(float g, float cs, float l2, float p) {
#pragma clang fp contract(off)
    const float gs = g * cs;
    return __builtin_fmaf(l2, p, gs);
}

// NS = 2: the step is applied twice on the same gradients with the bias corrections of `step` and `step + 1`, the EMA
// taking the weights between the two (the dual-track trainer's optimize_fn + optimizer.step(), quirk Q2) — one pass over
// the buffers instead of two, the same arithmetic per element in the same order.
template <int NS>
__global__ void __launch_bounds__(TPB) adamw_ema_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v,
                                                        float* __restrict__ ema, bf16_t* __restrict__ pb,
                                                        const float* __restrict__ clip, float gscale, float decay_mul,
                                                        float l2, float beta1, float beta2, float eps, float step_size,
                                                        float inv_bc2_sqrt, float one_m_ema, int64_t n,
                                                        float step_size2, float inv_bc2_sqrt2,
                                                        const float* __restrict__ hyper) {
    // hyper (optional, device memory): the per-step scalars of a step whose launch arguments were frozen by a HIP-graph
    // capture — {decay_mul, step_size, inv_bc2_sqrt, one_m_ema, step_size2, inv_bc2_sqrt2}, written by the host before
    // each replay
    if (hyper) {
        decay_mul = hyper[0]; step_size = hyper[1]; inv_bc2_sqrt = hyper[2]; one_m_ema = hyper[3];
        step_size2 = hyper[4]; inv_bc2_sqrt2 = hyper[5];
    }
    const float cs = (clip ? clip[0] : 1.0f) * gscale;
    const int64_t nv = n / 4;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < nv; i += (int64_t)gridDim.x * TPB) {
        float4 pp = reinterpret_cast<float4*>(p)[i];
        const float4 gg = reinterpret_cast<const float4*>(g)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
        float4 ee = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ema) ee = reinterpret_cast<float4*>(ema)[i];
        float* P = &pp.x; const float* G = &gg.x; float* Mm = &mm.x; float* Vv = &vv.x; float* E = &ee.x;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const float ss = s == 0 ? step_size : step_size2, ib = s == 0 ? inv_bc2_sqrt : inv_bc2_sqrt2;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float gk = l2_grad(G[k], cs, l2, P[k]);  // l2 != 0: torch.optim.Adam's weight decay (decay_mul is then 1)
                P[k] *= decay_mul;
                Mm[k] = Mm[k] * beta1 + gk * (1.0f - beta1);
                Vv[k] = Vv[k] * beta2 + gk * gk * (1.0f - beta2);
                const float denom = sqrtf(Vv[k]) * ib + eps;
                P[k] -= ss * (Mm[k] / denom);
                if (s == 0) E[k] -= one_m_ema * (E[k] - P[k]);
            }
        }
        reinterpret_cast<float4*>(p)[i] = pp;
        reinterpret_cast<float4*>(m)[i] = mm;
        reinterpret_cast<float4*>(v)[i] = vv;
        if (ema) reinterpret_cast<float4*>(ema)[i] = ee;
        if (pb) {
            union { bf16_t h[4]; uint2 u; } c;
            c.h[0] = (bf16_t)pp.x; c.h[1] = (bf16_t)pp.y; c.h[2] = (bf16_t)pp.z; c.h[3] = (bf16_t)pp.w;
            reinterpret_cast<uint2*>(pb)[i] = c.u;
        }
    }
}

static void adamw_scalars(float lr, float beta1, float beta2, float weight_decay, int decoupled, int step,
                          float ema_decay, float out[6]) {
    for (int s = 0; s < 2; ++s) {
        const double bc1 = 1.0 - pow((double)beta1, step + s), bc2 = 1.0 - pow((double)beta2, step + s);
        out[1 + 3 * s] = (float)((double)lr / bc1);
        out[2 + 3 * s] = (float)(1.0 / sqrt(bc2));
    }
    out[0] = decoupled ? (float)(1.0 - (double)lr * (double)weight_decay) : 1.0f;
    out[3] = 1.0f - ema_decay;
}

// the six per-step scalars of sedd_adamw_ema_step / _double_step as the kernel consumes them (host arithmetic only):
// what a caller writes into the `hyper` device buffer of sedd_adamw_ema_step_dev before each replay of a captured step
extern "C" int sedd_adamw_step_scalars(float lr, float beta1, float beta2, float weight_decay, int decoupled, int step,
                                       float ema_decay, float* out6_host) {
    SEDD_CHECK_ARG(out6_host && step >= 1, "adamw_step_scalars: bad arguments");
    adamw_scalars(lr, beta1, beta2, weight_decay, decoupled, step, ema_decay, out6_host);
    return SEDD_OK;
}

static int adamw_ema_launch(int ns, float* p, const float* g, float* m, float* v, float* ema, void* p_bf16,
                            const float* clip_coef, float grad_scale, float lr, float beta1, float beta2, float eps,
                            float weight_decay, int decoupled, int step, float ema_decay, int64_t n,
                            sedd_stream_t stream, const float* hyper = nullptr) {
    SEDD_CHECK_ARG(n >= 0 && step >= 1, "adamw_ema_step: bad n=%ld step=%d", (long)n, step);
    if (n % 4 != 0) {
        sedd_set_error("adamw_ema_step: n=%ld must be a multiple of 4 (pad the flat buffer)", (long)n);
        return SEDD_ERR_UNSUPPORTED_SHAPE;
    }
    if (n == 0) return SEDD_OK;
    SEDD_CHECK_ARG(p && g && m && v, "adamw_ema_step: null pointer");
    float sc[6];
    adamw_scalars(lr, beta1, beta2, weight_decay, decoupled, step, ema_decay, sc);
    const float step_size[2] = {sc[1], sc[4]}, inv_bc2_sqrt[2] = {sc[2], sc[5]};
    const float decay_mul = sc[0];
    const float l2 = decoupled ? 0.0f : weight_decay;
    const dim3 grid(sedd_grid_for(n / 4, TPB, 256 * 16));
    if (ns == 1)
        hipLaunchKernelGGL(adamw_ema_kernel<1>, grid, dim3(TPB), 0, (hipStream_t)stream, p, g, m, v, ema, (bf16_t*)p_bf16,
                           clip_coef, grad_scale, decay_mul, l2, beta1, beta2, eps, step_size[0], inv_bc2_sqrt[0],
                           1.0f - ema_decay, n, 0.f, 0.f, hyper);
    else
        hipLaunchKernelGGL(adamw_ema_kernel<2>, grid, dim3(TPB), 0, (hipStream_t)stream, p, g, m, v, ema, (bf16_t*)p_bf16,
                           clip_coef, grad_scale, decay_mul, l2, beta1, beta2, eps, step_size[0], inv_bc2_sqrt[0],
                           1.0f - ema_decay, n, step_size[1], inv_bc2_sqrt[1], hyper);
    SEDD_LAUNCH_CHECK();
    return SEDD_OK;
}

extern "C" int sedd_adamw_ema_step(float* p, const float* g, float* m, float* v, float* ema, void* p_bf16,
                                   const float* clip_coef, float grad_scale, float lr, float beta1, float beta2,
                                   float eps, float weight_decay, int decoupled, int step, float ema_decay, int64_t n,
                                   sedd_stream_t stream) {
    return adamw_ema_launch(1, p, g, m, v, ema, p_bf16, clip_coef, grad_scale, lr, beta1, beta2, eps, weight_decay,
                            decoupled, step, ema_decay, n, stream);
}

// two optimiser steps (`step`, `step + 1`) on the same clipped gradients in one pass, the EMA updated between them:
// bit-identical to sedd_adamw_ema_step(step, ema_decay) followed by sedd_adamw_ema_step(step + 1, ema_decay = 1)
extern "C" int sedd_adamw_ema_double_step(float* p, const float* g, float* m, float* v, float* ema, void* p_bf16,
                                          const float* clip_coef, float grad_scale, float lr, float beta1, float beta2,
                                          float eps, float weight_decay, int decoupled, int step, float ema_decay,
                                          int64_t n, sedd_stream_t stream) {
    return adamw_ema_launch(2, p, g, m, v, ema, p_bf16, clip_coef, grad_scale, lr, beta1, beta2, eps, weight_decay,
                            decoupled, step, ema_decay, n, stream);
}

// sedd_adamw_ema_step / _double_step (steps = 1 | 2) with the per-step scalars read from DEVICE memory (`hyper`: six
// floats, see sedd_adamw_step_scalars) instead of the launch arguments: for callers that capture the whole train step in
// a HIP graph, where launch arguments are frozen at capture (lr / bias corrections / EMA decay change every step).
extern "C" int sedd_adamw_ema_step_dev(float* p, const float* g, float* m, float* v, float* ema, void* p_bf16,
                                       const float* clip_coef, const float* hyper, float grad_scale, float beta1,
                                       float beta2, float eps, float weight_decay, int decoupled, int steps, int64_t n,
                                       sedd_stream_t stream) {
    SEDD_CHECK_ARG(hyper, "adamw_ema_step_dev: hyper is null");
    SEDD_CHECK_ARG(steps == 1 || steps == 2, "adamw_ema_step_dev: steps must be 1 or 2");
    return adamw_ema_launch(steps, p, g, m, v, ema, p_bf16, clip_coef, grad_scale, 0.f, beta1, beta2, eps, weight_decay,
                            decoupled, 1, 0.f, n, stream, hyper);
}

// The per-step device scalars of a captured train step, written by a one-thread launch whose values travel as launch
// ARGUMENTS (no host buffer has to outlive the call, nothing synchronises): hyper[0..5] for sedd_adamw_ema_step_dev and
// the dropout salt of the indirect seeds (sedd_gate_residual_fwd).  Either destination may be NULL.
__global__ void write_step_scalars_kernel(float* hyper, uint64_t* salt_dst, float v0, float v1, float v2, float v3,
                                          float v4, float v5, uint64_t salt) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        if (hyper) { hyper[0] = v0; hyper[1] = v1; hyper[2] = v2; hyper[3] = v3; hyper[4] = v4; hyper[5] = v5; }
        if (salt_dst) *salt_dst = salt;
    }
}
extern "C" int sedd_write_step_scalars(float* hyper_dev, uint64_t* salt_dev, float v0, float v1, float v2, float v3,
                                       float v4, float v5, uint64_t salt, sedd_stream_t stream) {
    hipLaunchKernelGGL(write_step_scalars_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, hyper_dev, salt_dev, v0, v1,
                       v2, v3, v4, v5, salt);
    SEDD_LAUNCH_CHECK();
    return SEDD_OK;
}

__global__ void __launch_bounds__(TPB) ema_update_kernel(float* __restrict__ s, const float* __restrict__ p, float omd,
                                                         int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB)
        s[i] -= omd * (s[i] - p[i]);
}
extern "C" int sedd_ema_update(float* shadow, const float* p, float one_minus_decay, int64_t n, sedd_stream_t stream) {
    SEDD_CHECK_ARG(n >= 0, "ema_update: bad n");
    if (n == 0) return SEDD_OK;
    SEDD_CHECK_ARG(shadow && p, "ema_update: null pointer");
    hipLaunchKernelGGL(ema_update_kernel, dim3(sedd_grid_for(n, TPB, 256 * 16)), dim3(TPB), 0, (hipStream_t)stream, shadow,
                       p, one_minus_decay, n);
    SEDD_LAUNCH_CHECK();
    return SEDD_OK;
}

// ---- sum of squares, two deterministic stages ---------------------------------------------------------------
#define SUMSQ_MAX_BLOCKS 1024
__global__ void __launch_bounds__(TPB) sumsq_stage1(const float* __restrict__ g, int64_t n, float* __restrict__ ws) {
    __shared__ float red[TPB / 64];
    float s = 0.f;
    const int64_t nv = n / 4;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < nv; i += (int64_t)gridDim.x * TPB) {
        const float4 x = reinterpret_cast<const float4*>(g)[i];
        s += (x.x * x.x + x.y * x.y) + (x.z * x.z + x.w * x.w);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float x = g[nv * 4 + threadIdx.x]; s += x * x; }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) ws[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void __launch_bounds__(TPB) sumsq_stage2(const float* __restrict__ ws, int nb, float gscale, float max_norm,
                                                    float* __restrict__ out) {
    __shared__ float red[TPB / 64];
    float s = 0.f;
    for (int i = threadIdx.x; i < nb; i += TPB) s += ws[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float t = (red[0] + red[1]) + (red[2] + red[3]);
        out[0] = t;
        // clip_grad_norm_(max_norm): max_norm == 0 zeroes the gradients (losses.py:65-66 clips whenever grad_clip >= 0);
        // a negative max_norm means "no clipping" and leaves out[1] alone
        if (max_norm >= 0.f) out[1] = fminf(1.0f, max_norm / (sqrtf(t) * gscale + 1e-6f));
    }
}

extern "C" int sedd_sumsq_clip(const float* g, int64_t n, float gscale, float max_norm, float* ws, float* out,
                               sedd_stream_t stream) {
    SEDD_CHECK_ARG(n >= 0, "sumsq_clip: bad n");
    SEDD_CHECK_ARG(ws && out && (n == 0 || g), "sumsq_clip: null pointer");
    const int nb = n == 0 ? 1 : sedd_grid_for(n / 4 + 1, TPB * 4, SUMSQ_MAX_BLOCKS);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(sumsq_stage1, dim3(nb), dim3(TPB), 0, st, g, n, ws);
    hipLaunchKernelGGL(sumsq_stage2, dim3(1), dim3(TPB), 0, st, ws, nb, gscale, max_norm, out);
    SEDD_LAUNCH_CHECK();
    return SEDD_OK;
}

// clip coefficient from an already-reduced sum of squares: out[1] = min(1, max_norm / (sqrt(out[0]) * gscale + 1e-6)) —
// sedd_sumsq_clip's last step on its own, for the sharded optimiser (each rank reduces its shard with max_norm < 0, the
// partial sums are all-reduced into out[0], then this).  max_norm < 0 leaves out[1] alone.
__global__ void clip_from_sumsq_kernel(float* out, float gscale, float max_norm) {
    if (threadIdx.x == 0 && blockIdx.x == 0 && max_norm >= 0.f)
        out[1] = fminf(1.0f, max_norm / (sqrtf(out[0]) * gscale + 1e-6f));
}
extern "C" int sedd_clip_from_sumsq(float* out, float gscale, float max_norm, sedd_stream_t stream) {
    SEDD_CHECK_ARG(out, "clip_from_sumsq: null pointer");
    hipLaunchKernelGGL(clip_from_sumsq_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out, gscale, max_norm);
    SEDD_LAUNCH_CHECK();
    return SEDD_OK;
}

// ---- loss reduction: sum_r w[r / L] * x[r] --------------------------------------------------------------------
__global__ void __launch_bounds__(TPB) wsum_stage1(const float* __restrict__ x, const float* __restrict__ w, int64_t M,
                                                   int64_t L, float* __restrict__ ws) {
    __shared__ float red[TPB / 64];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < M; i += (int64_t)gridDim.x * TPB) s += w[i / L] * x[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) ws[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
extern "C" int sedd_batch_weighted_sum(const float* x, const float* w, int64_t M, int64_t L, float* ws, float* out,
                                       sedd_stream_t stream) {
    SEDD_CHECK_ARG(M >= 0 && L > 0 && M % L == 0, "batch_weighted_sum: bad sizes");
    SEDD_CHECK_ARG(ws && out && (M == 0 || (x && w)), "batch_weighted_sum: null pointer");
    const int nb = M == 0 ? 1 : sedd_grid_for(M, TPB * 4, SUMSQ_MAX_BLOCKS);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(wsum_stage1, dim3(nb), dim3(TPB), 0, st, x, w, M, L, ws);
    hipLaunchKernelGGL(sumsq_stage2, dim3(1), dim3(TPB), 0, st, ws, nb, 1.0f, -1.0f, out);
    SEDD_LAUNCH_CHECK();
    return SEDD_OK;
}
