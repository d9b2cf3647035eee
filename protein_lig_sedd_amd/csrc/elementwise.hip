// HBM-bound model kernels: LayerNorm+adaLN modulate (fwd/bwd), gated residual with dropout (fwd/bwd), rotary,
// GELU, SiLU, timestep embedding, embedding gather / scatter-add, logits epilogue, column sums, transpose, cast.
// Reference call sites are cited per kernel (protlig_dd/model/*.py).  All are one-pass over their operands with
// 16-byte (fp32) / 8-byte (bf16) per-lane vector accesses; row statistics are wave-level xor-shuffle reductions.
#include "common.h"

#define TPB 256
// rows per backward chunk (never straddles a batch element): 32, or 64 once there are enough rows to fill the chip
// with the longer chunks (fewer block prologues / partial-sum folds per byte)
__host__ __device__ static inline int chunk_rows(int64_t M) { return M >= 16384 ? 64 : 32; }

// ---- 4-wide vector access helpers -------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ float4 load4(const T* p);
template <> __device__ __forceinline__ float4 load4<float>(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <> __device__ __forceinline__ float4 load4<bf16_t>(const bf16_t* p) {
    const uint2 r = *reinterpret_cast<const uint2*>(p);
    float4 o;
    o.x = __uint_as_float(r.x << 16);
    o.y = __uint_as_float(r.x & 0xffff0000u);
    o.z = __uint_as_float(r.y << 16);
    o.w = __uint_as_float(r.y & 0xffff0000u);
    return o;
}
template <typename T> __device__ __forceinline__ void store4(T* p, float4 v);
template <> __device__ __forceinline__ void store4<float>(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, float4 v) {
    union { bf16_t h[4]; uint2 u; } c;
    c.h[0] = (bf16_t)v.x; c.h[1] = (bf16_t)v.y; c.h[2] = (bf16_t)v.z; c.h[3] = (bf16_t)v.w;
    *reinterpret_cast<uint2*>(p) = c.u;
}
__device__ __forceinline__ float4 f4(float a) { return make_float4(a, a, a, a); }
// v as it reads back after a store in type T
template <typename T> __device__ __forceinline__ float4 round4(float4 v) {
    if constexpr (sizeof(T) == 2)
        return make_float4((float)(bf16_t)v.x, (float)(bf16_t)v.y, (float)(bf16_t)v.z, (float)(bf16_t)v.w);
    else return v;
}

#define DISPATCH_DT(code, TT, ...)                              \
    do {                                                        \
        if ((code) == SEDD_F32) { using TT = float; __VA_ARGS__; } \
        else { using TT = bf16_t; __VA_ARGS__; }                \
    } while (0)

static bool dt_ok(int d) { return d == SEDD_F32 || d == SEDD_BF16; }

// =================================================================================== LayerNorm + modulate fwd
// transformer.py:30-38 + fused_add_dropout_scale.py:33-34.  One wave per row, row held in registers.
template <typename TM, typename TO, int NV>
__global__ void __launch_bounds__(TPB) ln_mod_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, const TM* __restrict__ shift,
                                                         const TM* __restrict__ scale, int64_t mod_stride,
                                                         TO* __restrict__ y, float* __restrict__ mean,
                                                         float* __restrict__ rstd, int64_t M, int64_t L, int D) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (TPB / 64) + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + row * D;
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int c = k * 256 + lane * 4;
        v[k] = c < D ? load4<float>(xr + c) : f4(0.f);
        s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    }
    const float mu = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int c = k * 256 + lane * 4;
        if (c < D) {
            const float a = v[k].x - mu, b = v[k].y - mu, cc = v[k].z - mu, d = v[k].w - mu;
            q += (a * a + b * b) + (cc * cc + d * d);
        }
    }
    const float rs = 1.0f / sqrtf(wave_sum(q) / (float)D + 1e-5f);
    if (lane == 0) {
        if (mean) mean[row] = mu;
        if (rstd) rstd[row] = rs;
    }
    const int64_t b = row / L;
    TO* yr = y + row * D;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int c = k * 256 + lane * 4;
        if (c < D) {
            const float4 ww = load4<float>(w + c);
            float4 o;
            o.x = (v[k].x - mu) * rs * ww.x; o.y = (v[k].y - mu) * rs * ww.y;
            o.z = (v[k].z - mu) * rs * ww.z; o.w = (v[k].w - mu) * rs * ww.w;
            if (scale) {
                const float4 sc = load4<TM>(scale + b * mod_stride + c);
                const float4 sh = load4<TM>(shift + b * mod_stride + c);
                o.x = o.x * (1.0f + sc.x) + sh.x; o.y = o.y * (1.0f + sc.y) + sh.y;
                o.z = o.z * (1.0f + sc.z) + sh.z; o.w = o.w * (1.0f + sc.w) + sh.w;
            } else if (bias) {
                const float4 bb = load4<float>(bias + c);
                o.x += bb.x; o.y += bb.y; o.z += bb.z; o.w += bb.w;
            }
            store4<TO>(yr + c, o);
        }
    }
}

extern "C" int sedd_ln_modulate_fwd(const float* x, const float* w, const float* bias, const void* shift,
                                    const void* scale, int mod_dtype, int64_t mod_stride, void* y, int out_dtype,
                                    float* mean, float* rstd, int64_t M, int64_t L, int D, sedd_stream_t stream) {
    SEDD_CHECK_ARG(dt_ok(mod_dtype) && dt_ok(out_dtype), "ln_modulate_fwd: bad dtype");
    SEDD_CHECK_ARG(M >= 0 && L > 0 && M % L == 0, "ln_modulate_fwd: bad sizes M=%ld L=%ld", (long)M, (long)L);
    if (D <= 0 || D % 4 != 0 || D > 2048 || mod_stride % 4 != 0) {
        sedd_set_error("ln_modulate_fwd: D=%d mod_stride=%ld unsupported (need D %% 4 == 0, D <= 2048)", D, (long)mod_stride);
        return SEDD_ERR_UNSUPPORTED_SHAPE;
    }
    if (M == 0) return SEDD_OK;
    SEDD_CHECK_ARG(x && w && y, "ln_modulate_fwd: null pointer");
    SEDD_CHECK_ARG((shift == nullptr) == (scale == nullptr), "ln_modulate_fwd: shift and scale go together");
    const dim3 grid((unsigned)((M + 3) / 4));
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_DT(mod_dtype, TM, DISPATCH_DT(out_dtype, TO, {
        if (D <= 768)
            hipLaunchKernelGGL((ln_mod_fwd_kernel<TM, TO, 3>), grid, dim3(TPB), 0, st, x, w, bias, (const TM*)shift,
                               (const TM*)scale, mod_stride, (TO*)y, mean, rstd, M, L, D);
        else if (D <= 1024)
            hipLaunchKernelGGL((ln_mod_fwd_kernel<TM, TO, 4>), grid, dim3(TPB), 0, st, x, w, bias, (const TM*)shift,
                               (const TM*)scale, mod_stride, (TO*)y, mean, rstd, M, L, D);
        else
            hipLaunchKernelGGL((ln_mod_fwd_kernel<TM, TO, 8>), grid, dim3(TPB), 0, st, x, w, bias, (const TM*)shift,
                               (const TM*)scale, mod_stride, (TO*)y, mean, rstd, M, L, D);
    }));
    SEDD_LAUNCH_CHECK();
    return SEDD_OK;
}

// Counter-based dropout: the keep decision of element e is a pure function of (seed, e), so forward and backward
// regenerate the same mask instead of writing / reading a [M, D] byte mask (and no torch RNG kernel runs per block).
// Two rounds of a 32-bit avalanche hash (lowbias32) per group of 4 consecutive elements give four 16-bit uniforms;
// keep <=> u16 >= thr16 with thr16 = round(p * 65536).
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
// A seed with bit 63 set is INDIRECT: bits 47..0 hold the device address of a uint64 salt, bits 62..48 a site number, and
// the effective seed is salt + site * odd constant — a step captured in a HIP graph replays with fresh masks when the host
// (or a node of the graph) changes the salt in device memory, while the launch arguments stay what they were at capture.
__device__ __forceinline__ uint64_t resolve_seed(uint64_t seed) {
    if ((seed >> 63) == 0) return seed;
    const uint64_t salt = *reinterpret_cast<const uint64_t*>(seed & 0x0000FFFFFFFFFFFFull);
    return (salt + ((seed >> 48) & 0x7FFFull) * 0xD1B54A32D192ED03ull) & 0x7FFFFFFFFFFFFFFFull;
}
__device__ __forceinline__ float4 keep4_hashed(uint64_t seed, int64_t off, uint32_t thr16, float inv_keep) {
    const uint64_t q = (uint64_t)off >> 2;
    const uint32_t h0 = mix32(mix32((uint32_t)q ^ (uint32_t)seed) + (uint32_t)(q >> 32) + (uint32_t)(seed >> 32));
    const uint32_t h1 = mix32(h0 ^ 0x9e3779b9U);
    return make_float4((h0 & 0xffffU) >= thr16 ? inv_keep : 0.f, (h0 >> 16) >= thr16 ? inv_keep : 0.f,
                       (h1 & 0xffffU) >= thr16 ? inv_keep : 0.f, (h1 >> 16) >= thr16 ? inv_keep : 0.f);
}

// =================================================================================== LayerNorm + modulate bwd
// y = xhat*w*(1+sc)+sh, xhat = (x-mu)*rs.  g = dy*w*(1+sc);  dx = rs*(g - mean(g) - xhat*mean(g*xhat));
// dw += dy*(1+sc)*xhat ; dsh += dy ; dsc += dy*xhat*w.  One block per chunk of <= 32 rows of one batch element.
extern "C" int64_t sedd_ln_bwd_chunks(int64_t M, int64_t L) {
    if (L <= 0 || M <= 0) return 0;
    return (M / L) * ((L + chunk_rows(M) - 1) / chunk_rows(M));
}

// GATE: the finished dx row additionally goes through the backward of the gated residual that FEEDS this LayerNorm
// (x = residual + gate_b * dropout(a), fused_add_dropout_scale.py:13-48): da = dx * gate_b * keep (dtype of dy) and
// the per-chunk partials of dgate = sum_l dx * a * keep go to ws_gate — what gate_res_bwd_kernel would compute from a
// second pass over dx.  ws_bias (optional): per-chunk column sums of da AS STORED (the bias gradient of the Linear that
// produced a), gate_res_bwd_kernel's ws2.
struct GateBwdArgs {
    const void* a; const void* gate; void* da; float* ws_gate; float inv_keep; uint64_t seed; uint32_t thr16;
    float* ws_bias;
};
template <typename TD, typename TM, int NV, bool GATE>
__global__ void __launch_bounds__(TPB) ln_mod_bwd_kernel(const TD* __restrict__ dy, const float* __restrict__ x,
                                                         const float* __restrict__ w, const TM* __restrict__ scale,
                                                         int64_t mod_stride, const float* __restrict__ mean,
                                                         const float* __restrict__ rstd, float* dx,
                                                         const float* dx_add, float* __restrict__ ws, int64_t nchunk,
                                                         int64_t L, int D, int CR, GateBwdArgs ga) {
    __shared__ float red[3 * NV * 256];
    if (GATE && ga.thr16) ga.seed = resolve_seed(ga.seed);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t cpb = (L + CR - 1) / CR;
    const int64_t chunk = blockIdx.x, b = chunk / cpb, c0 = (chunk % cpb) * CR;
    const int nrows = (int)((L - c0) < CR ? (L - c0) : CR);
    float4 ww[NV], sc1[NV], aw[NV], ash[NV], asc[NV];
    float4 gt[GATE ? NV : 1], agt[GATE ? NV : 1], abs_[GATE ? NV : 1];
    const TD* ga_a = reinterpret_cast<const TD*>(ga.a);
    TD* ga_da = reinterpret_cast<TD*>(ga.da);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int c = k * 256 + lane * 4;
        if constexpr (GATE) {
            gt[k] = c < D ? load4<TM>(reinterpret_cast<const TM*>(ga.gate) + b * mod_stride + c) : f4(0.f);
            agt[k] = abs_[k] = f4(0.f);
        }
        ww[k] = c < D ? load4<float>(w + c) : f4(0.f);
        float4 s = f4(0.f);
        if (scale && c < D) s = load4<TM>(scale + b * mod_stride + c);
        sc1[k] = make_float4(1.f + s.x, 1.f + s.y, 1.f + s.z, 1.f + s.w);
        aw[k] = ash[k] = asc[k] = f4(0.f);
    }
    // Rows are software-pipelined: the loads of this wave's next row (dy, x, the residual gradient, mean / rstd) are
    // issued before the current row's two wave reductions, so a wave keeps two rows of HBM traffic in flight.
    // (dx_add may alias dx: a row is read and then written by the same wave, and rows are disjoint between waves.)
    float4 dA[NV], xA[NV], pA[NV], gA[GATE ? NV : 1];
    float muA = 0.f, rsA = 0.f;
#define LN_BWD_LOAD(r_, d_, x_, p_, mu_, rs_, g_)                                                    \
    do {                                                                                             \
        const int64_t row_ = b * L + c0 + (r_);                                                      \
        mu_ = mean[row_]; rs_ = rstd[row_];                                                          \
        _Pragma("unroll") for (int k = 0; k < NV; ++k) {                                             \
            const int c = k * 256 + lane * 4;                                                        \
            d_[k] = f4(0.f); x_[k] = f4(mu_); p_[k] = f4(0.f);                                       \
            if (c < D) {                                                                             \
                d_[k] = load4<TD>(dy + row_ * D + c);                                                \
                x_[k] = load4<float>(x + row_ * D + c);                                              \
                if (dx_add) p_[k] = load4<float>(dx_add + row_ * D + c);                             \
                if constexpr (GATE) g_[k] = load4<TD>(ga_a + row_ * D + c);                          \
            }                                                                                        \
        }                                                                                            \
    } while (0)
    if (wave < nrows) LN_BWD_LOAD(wave, dA, xA, pA, muA, rsA, gA);
    for (int r = wave; r < nrows; r += TPB / 64) {
        const int64_t row = b * L + c0 + r;
        float4 dB[NV], xB[NV], pB[NV], gB[GATE ? NV : 1];
        float muB = 0.f, rsB = 0.f;
        if (r + TPB / 64 < nrows) LN_BWD_LOAD(r + TPB / 64, dB, xB, pB, muB, rsB, gB);
        const float mu = muA, rs = rsA;
        float4 g[NV], xh[NV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const float4 d = dA[k], xv = xA[k];
            xh[k] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
            g[k] = make_float4(d.x * ww[k].x * sc1[k].x, d.y * ww[k].y * sc1[k].y, d.z * ww[k].z * sc1[k].z,
                               d.w * ww[k].w * sc1[k].w);
            s1 += (g[k].x + g[k].y) + (g[k].z + g[k].w);
            s2 += (g[k].x * xh[k].x + g[k].y * xh[k].y) + (g[k].z * xh[k].z + g[k].w * xh[k].w);
            aw[k].x += d.x * sc1[k].x * xh[k].x; aw[k].y += d.y * sc1[k].y * xh[k].y;
            aw[k].z += d.z * sc1[k].z * xh[k].z; aw[k].w += d.w * sc1[k].w * xh[k].w;
            ash[k].x += d.x; ash[k].y += d.y; ash[k].z += d.z; ash[k].w += d.w;
            asc[k].x += d.x * xh[k].x * ww[k].x; asc[k].y += d.y * xh[k].y * ww[k].y;
            asc[k].z += d.z * xh[k].z * ww[k].z; asc[k].w += d.w * xh[k].w * ww[k].w;
        }
        const float c1 = wave_sum(s1) / (float)D, c2 = wave_sum(s2) / (float)D;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int c = k * 256 + lane * 4;
            if (c < D) {
                float4 o = make_float4(rs * (g[k].x - c1 - xh[k].x * c2), rs * (g[k].y - c1 - xh[k].y * c2),
                                       rs * (g[k].z - c1 - xh[k].z * c2), rs * (g[k].w - c1 - xh[k].w * c2));
                if (dx_add) { o.x += pA[k].x; o.y += pA[k].y; o.z += pA[k].z; o.w += pA[k].w; }
                store4<float>(dx + row * D + c, o);
                if constexpr (GATE) {
                    float4 kk = f4(1.f);
                    if (ga.thr16) kk = keep4_hashed(ga.seed, row * D + c, ga.thr16, ga.inv_keep);
                    const float4 dav = make_float4(o.x * gt[k].x * kk.x, o.y * gt[k].y * kk.y, o.z * gt[k].z * kk.z,
                                                   o.w * gt[k].w * kk.w);
                    store4<TD>(ga_da + row * D + c, dav);
                    if (ga.ws_bias) {
                        const float4 rr = round4<TD>(dav);
                        abs_[k].x += rr.x; abs_[k].y += rr.y; abs_[k].z += rr.z; abs_[k].w += rr.w;
                    }
                    agt[k].x += o.x * gA[k].x * kk.x; agt[k].y += o.y * gA[k].y * kk.y;
                    agt[k].z += o.z * gA[k].z * kk.z; agt[k].w += o.w * gA[k].w * kk.w;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < NV; ++k) { dA[k] = dB[k]; xA[k] = xB[k]; pA[k] = pB[k]; }
        if constexpr (GATE) {
#pragma unroll
            for (int k = 0; k < NV; ++k) gA[k] = gB[k];
        }
        muA = muB; rsA = rsB;
    }
#undef LN_BWD_LOAD
    // fold the 4 waves' column partials through LDS (wave by wave: deterministic order)
    for (int wv = 0; wv < TPB / 64; ++wv) {
        if (wave == wv) {
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                float* r0 = red + (0 * NV + k) * 256 + lane * 4;
                float* r1 = red + (1 * NV + k) * 256 + lane * 4;
                float* r2 = red + (2 * NV + k) * 256 + lane * 4;
                if (wv == 0) {
                    store4<float>(r0, aw[k]); store4<float>(r1, ash[k]); store4<float>(r2, asc[k]);
                } else {
                    float4 p = load4<float>(r0);
                    store4<float>(r0, make_float4(p.x + aw[k].x, p.y + aw[k].y, p.z + aw[k].z, p.w + aw[k].w));
                    p = load4<float>(r1);
                    store4<float>(r1, make_float4(p.x + ash[k].x, p.y + ash[k].y, p.z + ash[k].z, p.w + ash[k].w));
                    p = load4<float>(r2);
                    store4<float>(r2, make_float4(p.x + asc[k].x, p.y + asc[k].y, p.z + asc[k].z, p.w + asc[k].w));
                }
            }
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < 3 * NV * 64; i += TPB) {
        const int plane = i / (NV * 64), rem = i % (NV * 64);
        const int c = rem * 4;
        if (c < D) store4<float>(ws + ((int64_t)plane * nchunk + chunk) * D + c, load4<float>(red + plane * NV * 256 + c));
    }
    if constexpr (GATE) {  // the gate-gradient partials take the same wave-by-wave fold through plane 0 of `red`
        __syncthreads();
        for (int wv = 0; wv < TPB / 64; ++wv) {
            if (wave == wv) {
#pragma unroll
                for (int k = 0; k < NV; ++k) {
                    float* r0 = red + k * 256 + lane * 4;
                    if (wv == 0) store4<float>(r0, agt[k]);
                    else {
                        const float4 p = load4<float>(r0);
                        store4<float>(r0, make_float4(p.x + agt[k].x, p.y + agt[k].y, p.z + agt[k].z, p.w + agt[k].w));
                    }
                }
            }
            __syncthreads();
        }
        for (int i = threadIdx.x; i < NV * 64; i += TPB) {
            const int c = i * 4;
            if (c < D) store4<float>(ga.ws_gate + chunk * D + c, load4<float>(red + c));
        }
        if (ga.ws_bias) {  // (block-uniform) the same fold for the column sums of da
            __syncthreads();
            for (int wv = 0; wv < TPB / 64; ++wv) {
                if (wave == wv) {
#pragma unroll
                    for (int k = 0; k < NV; ++k) {
                        float* r0 = red + k * 256 + lane * 4;
                        if (wv == 0) store4<float>(r0, abs_[k]);
                        else {
                            const float4 p = load4<float>(r0);
                            store4<float>(r0, make_float4(p.x + abs_[k].x, p.y + abs_[k].y, p.z + abs_[k].z, p.w + abs_[k].w));
                        }
                    }
                }
                __syncthreads();
            }
            for (int i = threadIdx.x; i < NV * 64; i += TPB) {
                const int c = i * 4;
                if (c < D) store4<float>(ga.ws_bias + chunk * D + c, load4<float>(red + c));
            }
        }
    }
}

static int ln_modulate_bwd_impl(const char* nm, const void* dy, int dy_dtype, const float* x, const float* w,
                                const void* scale, int mod_dtype, int64_t mod_stride, const float* mean,
                                const float* rstd, float* dx, const float* dx_add, float* ws, int64_t M, int64_t L, int D,
                                const GateBwdArgs* ga, sedd_stream_t stream) {
    SEDD_CHECK_ARG(dt_ok(dy_dtype) && dt_ok(mod_dtype), "%s: bad dtype", nm);
    SEDD_CHECK_ARG(M >= 0 && L > 0 && M % L == 0, "%s: bad sizes M=%ld L=%ld", nm, (long)M, (long)L);
    if (D <= 0 || D % 4 != 0 || D > (ga ? 1024 : 2048) || mod_stride % 4 != 0) {
        sedd_set_error("%s: D=%d unsupported (need D %% 4 == 0, D <= %d)", nm, D, ga ? 1024 : 2048);
        return SEDD_ERR_UNSUPPORTED_SHAPE;
    }
    if (M == 0) return SEDD_OK;
    SEDD_CHECK_ARG(dy && x && w && mean && rstd && dx && ws, "%s: null pointer", nm);
    SEDD_CHECK_ARG(!ga || (ga->a && ga->gate && ga->da && ga->ws_gate), "%s: null pointer", nm);
    const int64_t nchunk = sedd_ln_bwd_chunks(M, L);
    hipStream_t st = (hipStream_t)stream;
    const GateBwdArgs g0 = ga ? *ga : GateBwdArgs{nullptr, nullptr, nullptr, nullptr, 1.f, 0, 0, nullptr};
#define LN_BWD_LAUNCH(NV_, G_)                                                                                        \
    hipLaunchKernelGGL((ln_mod_bwd_kernel<TD, TM, NV_, G_>), dim3((unsigned)nchunk), dim3(TPB), 0, st, (const TD*)dy, x, \
                       w, (const TM*)scale, mod_stride, mean, rstd, dx, dx_add, ws, nchunk, L, D, chunk_rows(M), g0)
    DISPATCH_DT(dy_dtype, TD, DISPATCH_DT(mod_dtype, TM, {
        if (ga) {
            if (D <= 768) LN_BWD_LAUNCH(3, true);
            else LN_BWD_LAUNCH(4, true);
        } else if (D <= 768) LN_BWD_LAUNCH(3, false);
        else if (D <= 1024) LN_BWD_LAUNCH(4, false);
        else LN_BWD_LAUNCH(8, false);
    }));
#undef LN_BWD_LAUNCH
    SEDD_LAUNCH_CHECK();
    return SEDD_OK;
}

extern "C" int sedd_ln_modulate_bwd(const void* dy, int dy_dtype, const float* x, const float* w, const void* scale,
                                    int mod_dtype, int64_t mod_stride, const float* mean, const float* rstd,
                                    float* dx, const float* dx_add, float* ws, int64_t M, int64_t L, int D,
                                    sedd_stream_t stream) {
    return ln_modulate_bwd_impl("ln_modulate_bwd", dy, dy_dtype, x, w, scale, mod_dtype, mod_stride, mean, rstd, dx,
                                dx_add, ws, M, L, D, nullptr, stream);
}

// sedd_ln_modulate_bwd + the sedd_gate_residual_bwd of the gated residual that produced x, in one pass over the rows:
// da (dy's dtype) = dx * gate_b * keep, ws_gate fp32 [nchunk, D] = partials of dgate (fold with
// sedd_gate_residual_bwd_reduce).  a in dy's dtype; gate in mod_dtype with row stride mod_stride.
extern "C" int sedd_ln_modulate_bwd_gate(const void* dy, int dy_dtype, const float* x, const float* w, const void* scale,
                                         int mod_dtype, int64_t mod_stride, const float* mean, const float* rstd,
                                         float* dx, const float* dx_add, float* ws, const void* a, const void* gate,
                                         float inv_keep, uint64_t seed, float p_drop, void* da, float* ws_gate,
                                         int64_t M, int64_t L, int D, sedd_stream_t stream) {
    SEDD_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "ln_modulate_bwd_gate: p_drop %f", (double)p_drop);
    const GateBwdArgs ga{a, gate, da, ws_gate, inv_keep, seed, (uint32_t)(p_drop * 65536.0f + 0.5f), nullptr};
    return ln_modulate_bwd_impl("ln_modulate_bwd_gate", dy, dy_dtype, x, w, scale, mod_dtype, mod_stride, mean, rstd, dx,
                                dx_add, ws, M, L, D, &ga, stream);
}

// ... and with the column sums of da as stored (sedd_gate_residual_bwd's want_colsum): ws_gate then holds TWO planes of
// [nchunk, D] — the gate-gradient partials and the bias-gradient partials — exactly what sedd_gate_residual_bwd_reduce
// folds.  This is the form a block BOUNDARY takes: the LayerNorm is the next block's norm1, the gated residual the
// previous block's MLP branch, whose second Linear's bias gradient is the column sum.
extern "C" int sedd_ln_modulate_bwd_gate_colsum(const void* dy, int dy_dtype, const float* x, const float* w,
                                                const void* scale, int mod_dtype, int64_t mod_stride, const float* mean,
                                                const float* rstd, float* dx, const float* dx_add, float* ws,
                                                const void* a, const void* gate, float inv_keep, uint64_t seed,
                                                float p_drop, void* da, float* ws_gate, int64_t M, int64_t L, int D,
                                                sedd_stream_t stream) {
    SEDD_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "ln_modulate_bwd_gate_colsum: p_drop %f", (double)p_drop);
    SEDD_CHECK_ARG(ws_gate != nullptr, "ln_modulate_bwd_gate_colsum: null pointer");
    const GateBwdArgs ga{a, gate, da, ws_gate, inv_keep, seed, (uint32_t)(p_drop * 65536.0f + 0.5f),
                         ws_gate + sedd_ln_bwd_chunks(M, L) * D};
    return ln_modulate_bwd_impl("ln_modulate_bwd_gate_colsum", dy, dy_dtype, x, w, scale, mod_dtype, mod_stride, mean, rstd,
                                dx, dx_add, ws, M, L, D, &ga, stream);
}

// out_k[b, j] = sum over the chunks of batch b of plane k (blocks [0, nxb * B));  planes in vec_mask are instead summed
// over ALL chunks into a [D] vector (+=) by the trailing "fold" blocks: 64 columns per block, 16 row groups of 16 column
// quads each walk the chunks with stride 16 and meet in LDS in a fixed order — one writer per column, no atomics, so
// the bias / LayerNorm-weight gradients come out bit-identical from run to run.
__global__ void __launch_bounds__(TPB) chunk_reduce_kernel(const float* __restrict__ ws, int nplanes,
                                                           float* __restrict__ o0, float* __restrict__ o1,
                                                           float* __restrict__ o2, int64_t ostride, int vec_mask,
                                                           int64_t nchunk, int64_t cpb, int D, int nxb, int64_t B) {
    float* outs[3] = {o0, o1, o2};
    const int64_t nperb = (int64_t)nxb * B;
    if ((int64_t)blockIdx.x < nperb) {
        const int j = (int)(blockIdx.x % nxb) * TPB + threadIdx.x;
        const int64_t b = blockIdx.x / nxb;
        if (j >= D) return;
        for (int k = 0; k < nplanes; ++k) {
            if (!outs[k] || ((vec_mask >> k) & 1)) continue;
            float s = 0.f;
            for (int64_t c = 0; c < cpb; ++c) s += ws[((int64_t)k * nchunk + b * cpb + c) * D + j];
            outs[k][b * ostride + j] = s;
        }
        return;
    }
    __shared__ float4 red[16][16];
    const int col = (int)(blockIdx.x - nperb) * 64 + (threadIdx.x & 15) * 4, rg = threadIdx.x >> 4;
    for (int k = 0; k < nplanes; ++k) {
        if (!outs[k] || !((vec_mask >> k) & 1)) continue;
        float4 a = f4(0.f);
        if (col < D) {
            // four running sums per row group (chunks c, c+16, c+32, c+48 in flight together: the walk is a chain of
            // dependent load latencies otherwise), combined in a fixed order
            float4 a1 = f4(0.f), a2 = f4(0.f), a3 = f4(0.f);
            const float* base = ws + (int64_t)k * nchunk * D + col;
            int64_t c = rg;
            for (; c + 48 < nchunk; c += 64) {
                const float4 v0 = load4<float>(base + c * D), v1 = load4<float>(base + (c + 16) * D);
                const float4 v2 = load4<float>(base + (c + 32) * D), v3 = load4<float>(base + (c + 48) * D);
                a.x += v0.x; a.y += v0.y; a.z += v0.z; a.w += v0.w;
                a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
                a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
                a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
            }
            for (; c < nchunk; c += 16) {
                const float4 v = load4<float>(base + c * D);
                a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
            }
            a.x = (a.x + a1.x) + (a2.x + a3.x); a.y = (a.y + a1.y) + (a2.y + a3.y);
            a.z = (a.z + a1.z) + (a2.z + a3.z); a.w = (a.w + a1.w) + (a2.w + a3.w);
        }
        red[rg][threadIdx.x & 15] = a;
        __syncthreads();
        if (threadIdx.x < 16 && col < D) {
            float4 t = red[0][threadIdx.x];
            for (int r = 1; r < 16; ++r) {
                const float4 v = red[r][threadIdx.x];
                t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
            }
            float* o = outs[k] + col;  // (scalar accesses: the gradient vector need not be 16-byte aligned)
            o[0] += t.x; o[1] += t.y; o[2] += t.z; o[3] += t.w;
        }
        __syncthreads();
    }
}
static void launch_chunk_reduce(const float* ws, int nplanes, float* o0, float* o1, float* o2, int64_t ostride,
                                int vec_mask, int64_t nchunk, int64_t B, int D, hipStream_t st) {
    const int nxb = (D + TPB - 1) / TPB;
    bool vec = false;
    float* outs[3] = {o0, o1, o2};
    for (int k = 0; k < nplanes; ++k) vec = vec || (outs[k] && ((vec_mask >> k) & 1));
    const unsigned nblk = (unsigned)(nxb * B + (vec ? (D + 63) / 64 : 0));
    hipLaunchKernelGGL(chunk_reduce_kernel, dim3(nblk), dim3(TPB), 0, st, ws, nplanes, o0, o1, o2, ostride, vec_mask,
                       nchunk, nchunk / B, D, nxb, B);
}

extern "C" int sedd_ln_modulate_bwd_reduce(const float* ws, float* dw, float* dshift, float* dscale,
                                           int64_t dmod_stride, int64_t M, int64_t L, int D, sedd_stream_t stream) {
    SEDD_CHECK_ARG(M >= 0 && L > 0 && M % L == 0 && D > 0, "ln_modulate_bwd_reduce: bad sizes");
    if (M == 0) return SEDD_OK;
    SEDD_CHECK_ARG(ws, "ln_modulate_bwd_reduce: ws is null");
    const int64_t nchunk = sedd_ln_bwd_chunks(M, L), B = M / L;
    SEDD_CHECK_ARG(D % 4 == 0, "ln_modulate_bwd_reduce: D %% 4 != 0");
    launch_chunk_reduce(ws, 3, dw, dshift, dscale, dmod_stride, 1, nchunk, B, D, (hipStream_t)stream);
    SEDD_LAUNCH_CHECK();
    return SEDD_OK;
}

// ---------------------------------------------------------------------------------- entry point that allocates nothing
// sedd_ln_modulate_bwd + sedd_ln_modulate_bwd_reduce in one call, with the [3, nchunk, D] partial-sum workspace in a
// file-scope device array instead of caller memory: for callers that cannot allocate (HIP graph capture, serving loops,
// harnesses that forbid hipMalloc).  Each GPU holds its own copy of the array at its own address; the address is looked
// up once per device with hipGetSymbolAddress and only that pointer is ever handed to kernels.
#define SEDD_LN_BWD_SCRATCH_FLOATS SEDD_LN_BWD_NOALLOC_MAX_FLOATS /* 16 MiB per GPU, held until the process ends */
#define SEDD_MAX_DEVICES 64
static __device__ __align__(256) float ln_bwd_scratch[SEDD_LN_BWD_SCRATCH_FLOATS];
static float* volatile ln_bwd_scratch_ptr[SEDD_MAX_DEVICES];  // per device; written once (idempotent value)

static int ln_bwd_scratch_for_current_device(float** out) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= SEDD_MAX_DEVICES) {
        sedd_set_error("ln_modulate_bwd_noalloc: no current device (or device index %d >= %d)", dev, SEDD_MAX_DEVICES);
        return SEDD_ERR_LAUNCH;
    }
    float* p = ln_bwd_scratch_ptr[dev];
    if (!p) {  // first call on this device (racing first calls resolve the same address)
        const hipError_t e = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(ln_bwd_scratch));
        if (e != hipSuccess || !p) {
            sedd_set_error("ln_modulate_bwd_noalloc: hipGetSymbolAddress failed on device %d: %s", dev, hipGetErrorString(e));
            return SEDD_ERR_LAUNCH;
        }
        ln_bwd_scratch_ptr[dev] = p;
    }
    *out = p;
    return SEDD_OK;
}

extern "C" int sedd_ln_modulate_bwd_noalloc(const void* dy, int dy_dtype, const float* x, const float* w,
                                            const void* scale, int mod_dtype, int64_t mod_stride, const float* mean,
                                            const float* rstd, float* dx, const float* dx_add, float* dw, float* dshift,
                                            float* dscale, int64_t dmod_stride, int64_t M, int64_t L, int D,
                                            sedd_stream_t stream) {
    SEDD_CHECK_ARG(M >= 0 && L > 0 && M % L == 0 && D > 0, "ln_modulate_bwd_noalloc: bad sizes M=%ld L=%ld D=%d", (long)M,
                   (long)L, D);
    if (M == 0) return SEDD_OK;
    const int64_t need = 3 * sedd_ln_bwd_chunks(M, L) * (int64_t)D;
    if (need > (int64_t)SEDD_LN_BWD_SCRATCH_FLOATS) {
        sedd_set_error("ln_modulate_bwd_noalloc: M=%ld L=%ld D=%d needs %ld workspace floats, the static scratch holds %ld "
                       "(16 MiB): use sedd_ln_modulate_bwd with a caller workspace", (long)M, (long)L, D, (long)need,
                       (long)SEDD_LN_BWD_SCRATCH_FLOATS);
        return SEDD_ERR_UNSUPPORTED_SHAPE;
    }
    float* ws = nullptr;
    int rc = ln_bwd_scratch_for_current_device(&ws);
    if (rc != SEDD_OK) return rc;
    rc = ln_modulate_bwd_impl("ln_modulate_bwd_noalloc", dy, dy_dtype, x, w, scale, mod_dtype, mod_stride, mean, rstd, dx,
                              dx_add, ws, M, L, D, nullptr, stream);
    if (rc != SEDD_OK) return rc;
    return sedd_ln_modulate_bwd_reduce(ws, dw, dshift, dscale, dmod_stride, M, L, D, stream);
}

// ====================================================================================== gated residual fwd
// fused_add_dropout_scale.py:13-48 with bias == None: out = residual + gate_b * dropout(a)
template <typename TA, typename TM>
__global__ void __launch_bounds__(TPB) gate_res_fwd_kernel(const float* __restrict__ res, const TA* __restrict__ a,
                                                           const TM* __restrict__ gate, int64_t mod_stride,
                                                           const uint8_t* __restrict__ keep, float inv_keep,
                                                           uint64_t seed, uint32_t thr16,
                                                           float* __restrict__ out, int64_t M, int64_t L, int D) {
    if (thr16) seed = resolve_seed(seed);
    const int64_t nvec = M * (int64_t)(D / 4);
    const int dv = D / 4;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * TPB) {
        const int64_t row = i / dv;
        const int c = (int)(i % dv) * 4;
        const int64_t off = row * D + c;
        float4 av = load4<TA>(a + off);
        if (keep) {
            const uchar4 k = *reinterpret_cast<const uchar4*>(keep + off);
            av.x *= k.x ? inv_keep : 0.f; av.y *= k.y ? inv_keep : 0.f;
            av.z *= k.z ? inv_keep : 0.f; av.w *= k.w ? inv_keep : 0.f;
        } else if (thr16) {
            const float4 kk = keep4_hashed(seed, off, thr16, inv_keep);
            av.x *= kk.x; av.y *= kk.y; av.z *= kk.z; av.w *= kk.w;
        }
        const float4 g = load4<TM>(gate + (row / L) * mod_stride + c);
        const float4 r = load4<float>(res + off);
        store4<float>(out + off, make_float4(r.x + g.x * av.x, r.y + g.y * av.y, r.z + g.z * av.z, r.w + g.w * av.w));
    }
}

extern "C" int sedd_gate_residual_fwd(const float* residual, const void* a, int a_dtype, const void* gate,
                                      int mod_dtype, int64_t mod_stride, const uint8_t* keep, float inv_keep,
                                      uint64_t seed, float p_drop, float* out, int64_t M, int64_t L, int D,
                                      sedd_stream_t stream) {
    SEDD_CHECK_ARG(dt_ok(a_dtype) && dt_ok(mod_dtype), "gate_residual_fwd: bad dtype");
    SEDD_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "gate_residual_fwd: p_drop %f", (double)p_drop);
    const uint32_t thr16 = (uint32_t)(p_drop * 65536.0f + 0.5f);
    SEDD_CHECK_ARG(M >= 0 && L > 0 && M % L == 0, "gate_residual_fwd: bad sizes");
    if (D <= 0 || D % 4 != 0 || mod_stride % 4 != 0) {
        sedd_set_error("gate_residual_fwd: D=%d unsupported (need D %% 4 == 0)", D);
        return SEDD_ERR_UNSUPPORTED_SHAPE;
    }
    if (M == 0) return SEDD_OK;
    SEDD_CHECK_ARG(residual && a && gate && out, "gate_residual_fwd: null pointer");
    const dim3 grid(sedd_grid_for(M * (D / 4), TPB, 256 * 16));
    DISPATCH_DT(a_dtype, TA, DISPATCH_DT(mod_dtype, TM,
        hipLaunchKernelGGL((gate_res_fwd_kernel<TA, TM>), grid, dim3(TPB), 0, (hipStream_t)stream, residual,
                           (const TA*)a, (const TM*)gate, mod_stride, keep, inv_keep, seed, thr16, out, M, L, D)));
    SEDD_LAUNCH_CHECK();
    return SEDD_OK;
}

// ============================================================== gated residual + LayerNorm-modulate, one pass
// x1 = residual + gate_b * dropout(a) (fused_add_dropout_scale.py:13-48) followed by the next sub-layer's
// h = LN(x1) * w * (1 + scale_b) + shift_b (transformer.py:30-38, :186): the row of x1 never leaves the registers
// between the two, so x1 is written once and not read back (one wave per row, like ln_mod_fwd_kernel).
template <typename TA, typename TM, int NV>
__global__ void __launch_bounds__(TPB) gate_res_ln_mod_fwd_kernel(
    const float* __restrict__ res, const TA* __restrict__ a, const TM* __restrict__ gate, const TM* __restrict__ shift,
    const TM* __restrict__ scale, int64_t mod_stride, float inv_keep, uint64_t seed, uint32_t thr16,
    const float* __restrict__ w, float* __restrict__ x1, TA* __restrict__ y, float* __restrict__ mean,
    float* __restrict__ rstd, int64_t M, int64_t L, int D) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (TPB / 64) + (threadIdx.x >> 6);
    if (row >= M) return;
    if (thr16) seed = resolve_seed(seed);
    const int64_t b = row / L;
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int c = k * 256 + lane * 4;
        v[k] = f4(0.f);
        if (c < D) {
            const int64_t off = row * D + c;
            float4 av = load4<TA>(a + off);
            if (thr16) {
                const float4 kk = keep4_hashed(seed, off, thr16, inv_keep);
                av.x *= kk.x; av.y *= kk.y; av.z *= kk.z; av.w *= kk.w;
            }
            const float4 g = load4<TM>(gate + b * mod_stride + c);
            const float4 r = load4<float>(res + off);
            v[k] = make_float4(r.x + g.x * av.x, r.y + g.y * av.y, r.z + g.z * av.z, r.w + g.w * av.w);
            store4<float>(x1 + off, v[k]);
        }
        s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    }
    const float mu = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int c = k * 256 + lane * 4;
        if (c < D) {
            const float a0 = v[k].x - mu, a1 = v[k].y - mu, a2 = v[k].z - mu, a3 = v[k].w - mu;
            q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
        }
    }
    const float rs = 1.0f / sqrtf(wave_sum(q) / (float)D + 1e-5f);
    if (lane == 0) {
        mean[row] = mu;
        rstd[row] = rs;
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int c = k * 256 + lane * 4;
        if (c < D) {
            const float4 ww = load4<float>(w + c);
            const float4 sc = load4<TM>(scale + b * mod_stride + c);
            const float4 sh = load4<TM>(shift + b * mod_stride + c);
            float4 o;
            o.x = (v[k].x - mu) * rs * ww.x; o.y = (v[k].y - mu) * rs * ww.y;
            o.z = (v[k].z - mu) * rs * ww.z; o.w = (v[k].w - mu) * rs * ww.w;
            o.x = o.x * (1.0f + sc.x) + sh.x; o.y = o.y * (1.0f + sc.y) + sh.y;
            o.z = o.z * (1.0f + sc.z) + sh.z; o.w = o.w * (1.0f + sc.w) + sh.w;
            store4<TA>(y + row * D + c, o);
        }
    }
}

extern "C" int sedd_gate_residual_ln_modulate_fwd(const float* residual, const void* a, int a_dtype, const void* gate,
                                                  const void* shift, const void* scale, int mod_dtype,
                                                  int64_t mod_stride, float inv_keep, uint64_t seed, float p_drop,
                                                  const float* w, float* x_out, void* y, float* mean, float* rstd,
                                                  int64_t M, int64_t L, int D, sedd_stream_t stream) {
    SEDD_CHECK_ARG(dt_ok(a_dtype) && dt_ok(mod_dtype), "gate_residual_ln_modulate_fwd: bad dtype");
    SEDD_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "gate_residual_ln_modulate_fwd: p_drop %f", (double)p_drop);
    const uint32_t thr16 = (uint32_t)(p_drop * 65536.0f + 0.5f);
    SEDD_CHECK_ARG(M >= 0 && L > 0 && M % L == 0, "gate_residual_ln_modulate_fwd: bad sizes");
    if (D <= 0 || D % 4 != 0 || D > 1024 || mod_stride % 4 != 0) {
        sedd_set_error("gate_residual_ln_modulate_fwd: D=%d unsupported (need D %% 4 == 0, D <= 1024)", D);
        return SEDD_ERR_UNSUPPORTED_SHAPE;
    }
    if (M == 0) return SEDD_OK;
    SEDD_CHECK_ARG(residual && a && gate && shift && scale && w && x_out && y && mean && rstd,
                   "gate_residual_ln_modulate_fwd: null pointer");
    const dim3 grid((unsigned)((M + 3) / 4));
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_DT(a_dtype, TA, DISPATCH_DT(mod_dtype, TM, {
        if (D <= 768)
            hipLaunchKernelGGL((gate_res_ln_mod_fwd_kernel<TA, TM, 3>), grid, dim3(TPB), 0, st, residual, (const TA*)a,
                               (const TM*)gate, (const TM*)shift, (const TM*)scale, mod_stride, inv_keep, seed, thr16, w,
                               x_out, (TA*)y, mean, rstd, M, L, D);
        else
            hipLaunchKernelGGL((gate_res_ln_mod_fwd_kernel<TA, TM, 4>), grid, dim3(TPB), 0, st, residual, (const TA*)a,
                               (const TM*)gate, (const TM*)shift, (const TM*)scale, mod_stride, inv_keep, seed, thr16, w,
                               x_out, (TA*)y, mean, rstd, M, L, D);
    }));
    SEDD_LAUNCH_CHECK();
    return SEDD_OK;
}

// ====================================================================================== gated residual bwd
// da = dout*gate*keep/(1-p);  dgate[b] = sum_l dout*a*keep/(1-p)  (chunk partials, folded by the reduce kernel)
template <typename TA, typename TM>
__global__ void __launch_bounds__(TPB) gate_res_bwd_kernel(const float* __restrict__ dout, const TA* __restrict__ a,
                                                           const TM* __restrict__ gate, int64_t mod_stride,
                                                           const uint8_t* __restrict__ keep, float inv_keep,
                                                           uint64_t seed, uint32_t thr16,
                                                           TA* __restrict__ da, float* __restrict__ ws,
                                                           float* __restrict__ ws2, int64_t L, int D, int CR) {
    __shared__ float red[(TPB / 64) * 256];
    if (thr16) seed = resolve_seed(seed);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t cpb = (L + CR - 1) / CR;
    const int64_t chunk = blockIdx.x, b = chunk / cpb, c0 = (chunk % cpb) * CR;
    const int nrows = (int)((L - c0) < CR ? (L - c0) : CR);
    for (int cb = 0; cb < D; cb += 256) {  // 256-column slabs
        const int c = cb + lane * 4;
        float4 acc = f4(0.f), acc2 = f4(0.f);
        if (c < D) {
            const float4 g = load4<TM>(gate + b * mod_stride + c);
            for (int r = wave; r < nrows; r += TPB / 64) {
                const int64_t off = (b * L + c0 + r) * D + c;
                const float4 d = load4<float>(dout + off);
                float4 av = load4<TA>(a + off);
                float4 kk = f4(1.f);
                if (keep) {
                    const uchar4 k = *reinterpret_cast<const uchar4*>(keep + off);
                    kk = make_float4(k.x ? inv_keep : 0.f, k.y ? inv_keep : 0.f, k.z ? inv_keep : 0.f, k.w ? inv_keep : 0.f);
                } else if (thr16) {
                    kk = keep4_hashed(seed, off, thr16, inv_keep);
                }
                const float4 dav = make_float4(d.x * g.x * kk.x, d.y * g.y * kk.y, d.z * g.z * kk.z, d.w * g.w * kk.w);
                store4<TA>(da + off, dav);
                acc.x += d.x * av.x * kk.x; acc.y += d.y * av.y * kk.y;
                acc.z += d.z * av.z * kk.z; acc.w += d.w * av.w * kk.w;
                if (ws2) {  // column sums of da as stored (the bias gradient of the Linear that produced a)
                    const float4 r = round4<TA>(dav);
                    acc2.x += r.x; acc2.y += r.y; acc2.z += r.z; acc2.w += r.w;
                }
            }
        }
        store4<float>(red + wave * 256 + lane * 4, acc);
        __syncthreads();
        if (wave == 0 && c < D) {
            float4 s = load4<float>(red + lane * 4);
            for (int wv = 1; wv < TPB / 64; ++wv) {
                const float4 p = load4<float>(red + wv * 256 + lane * 4);
                s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
            }
            store4<float>(ws + chunk * D + c, s);
        }
        __syncthreads();
        if (ws2) {
            store4<float>(red + wave * 256 + lane * 4, acc2);
            __syncthreads();
            if (wave == 0 && c < D) {
                float4 s = load4<float>(red + lane * 4);
                for (int wv = 1; wv < TPB / 64; ++wv) {
                    const float4 p = load4<float>(red + wv * 256 + lane * 4);
                    s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
                }
                store4<float>(ws2 + chunk * D + c, s);
            }
            __syncthreads();
        }
    }
}

extern "C" int sedd_gate_residual_bwd(const float* dout, const void* a, int a_dtype, const void* gate, int mod_dtype,
                                      int64_t mod_stride, const uint8_t* keep, float inv_keep, uint64_t seed,
                                      float p_drop, void* da, float* ws, int want_colsum, int64_t M, int64_t L, int D,
                                      sedd_stream_t stream) {
    SEDD_CHECK_ARG(dt_ok(a_dtype) && dt_ok(mod_dtype), "gate_residual_bwd: bad dtype");
    SEDD_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "gate_residual_bwd: p_drop %f", (double)p_drop);
    const uint32_t thr16 = (uint32_t)(p_drop * 65536.0f + 0.5f);
    SEDD_CHECK_ARG(M >= 0 && L > 0 && M % L == 0, "gate_residual_bwd: bad sizes");
    if (D <= 0 || D % 4 != 0 || mod_stride % 4 != 0) {
        sedd_set_error("gate_residual_bwd: D=%d unsupported (need D %% 4 == 0)", D);
        return SEDD_ERR_UNSUPPORTED_SHAPE;
    }
    if (M == 0) return SEDD_OK;
    SEDD_CHECK_ARG(dout && a && gate && da && ws, "gate_residual_bwd: null pointer");
    const int64_t nchunk = sedd_ln_bwd_chunks(M, L);
    DISPATCH_DT(a_dtype, TA, DISPATCH_DT(mod_dtype, TM,
        hipLaunchKernelGGL((gate_res_bwd_kernel<TA, TM>), dim3((unsigned)nchunk), dim3(TPB), 0, (hipStream_t)stream,
                           dout, (const TA*)a, (const TM*)gate, mod_stride, keep, inv_keep, seed, thr16, (TA*)da, ws,
                           want_colsum ? ws + nchunk * D : (float*)nullptr, L, D,
                           chunk_rows(M))));
    SEDD_LAUNCH_CHECK();
    return SEDD_OK;
}

extern "C" int sedd_gate_residual_bwd_reduce(const float* ws, float* dgate, int64_t dmod_stride, float* dbias,
                                             int64_t M, int64_t L, int D, sedd_stream_t stream) {
    SEDD_CHECK_ARG(M >= 0 && L > 0 && M % L == 0 && D > 0, "gate_residual_bwd_reduce: bad sizes");
    if (M == 0) return SEDD_OK;
    SEDD_CHECK_ARG(ws && dgate, "gate_residual_bwd_reduce: null pointer");
    const int64_t nchunk = sedd_ln_bwd_chunks(M, L), B = M / L;
    SEDD_CHECK_ARG(D % 4 == 0, "gate_residual_bwd_reduce: D %% 4 != 0");
    launch_chunk_reduce(ws, dbias ? 2 : 1, dgate, dbias, (float*)nullptr, dmod_stride, dbias ? 2 : 0, nchunk, B, D,
                        (hipStream_t)stream);
    SEDD_LAUNCH_CHECK();
    return SEDD_OK;
}

// ================================================================================================== rotary
// rotary.py:30-39 on the Q and K slots of [B, L, 3, H, d]; one thread per 4 (first-half, second-half) pairs.
template <typename T>
__global__ void __launch_bounds__(TPB) rotary_kernel(T* __restrict__ qkv, const float* __restrict__ cs,
                                                     const float* __restrict__ sn, int64_t M, int L, int H, int d,
                                                     int conj) {
    const int hv = d / 8;  // float4 groups per half
    const int64_t total = M * 2 * H * hv;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < total; i += (int64_t)gridDim.x * TPB) {
        const int j = (int)(i % hv) * 4;
        int64_t t = i / hv;
        const int h = (int)(t % H); t /= H;
        const int slot = (int)(t % 2);
        const int64_t row = t / 2;
        const int l = (int)(row % L);
        T* p = qkv + ((row * 3 + slot) * H + h) * d + j;
        const float4 x1 = load4<T>(p), x2 = load4<T>(p + d / 2);
        const float4 c = load4<float>(cs + (int64_t)l * (d / 2) + j);
        float4 s = load4<float>(sn + (int64_t)l * (d / 2) + j);
        if (conj) s = make_float4(-s.x, -s.y, -s.z, -s.w);
        store4<T>(p, make_float4(x1.x * c.x - x2.x * s.x, x1.y * c.y - x2.y * s.y, x1.z * c.z - x2.z * s.z,
                                 x1.w * c.w - x2.w * s.w));
        store4<T>(p + d / 2, make_float4(x2.x * c.x + x1.x * s.x, x2.y * c.y + x1.y * s.y, x2.z * c.z + x1.z * s.z,
                                         x2.w * c.w + x1.w * s.w));
    }
}

extern "C" int sedd_rotary_qkv(void* qkv, int dtype, const float* cos, const float* sin, int B, int L, int H, int d,
                               int conj, sedd_stream_t stream) {
    SEDD_CHECK_ARG(dt_ok(dtype), "rotary_qkv: bad dtype");
    SEDD_CHECK_ARG(B >= 0 && L > 0 && H > 0, "rotary_qkv: bad sizes");
    if (d <= 0 || d % 8 != 0) {
        sedd_set_error("rotary_qkv: head dim %d unsupported (need d %% 8 == 0)", d);
        return SEDD_ERR_UNSUPPORTED_SHAPE;
    }
    if (B == 0) return SEDD_OK;
    SEDD_CHECK_ARG(qkv && cos && sin, "rotary_qkv: null pointer");
    const int64_t M = (int64_t)B * L;
    const dim3 grid(sedd_grid_for(M * 2 * H * (d / 8), TPB, 256 * 16));
    DISPATCH_DT(dtype, T, hipLaunchKernelGGL((rotary_kernel<T>), grid, dim3(TPB), 0, (hipStream_t)stream, (T*)qkv, cos,
                                             sin, M, L, H, d, conj));
    SEDD_LAUNCH_CHECK();
    return SEDD_OK;
}

// =========================================================================================== GELU / SiLU
__device__ __forceinline__ float gelu_tanh(float u) {
    const float t = tanhf(0.7978845608028654f * (u + 0.044715f * u * u * u));
    return 0.5f * u * (1.0f + t);
}
__device__ __forceinline__ float gelu_tanh_grad(float u) {
    const float t = tanhf(0.7978845608028654f * (u + 0.044715f * u * u * u));
    return 0.5f * (1.0f + t) + 0.5f * u * (1.0f - t * t) * 0.7978845608028654f * (1.0f + 3.0f * 0.044715f * u * u);
}
__device__ __forceinline__ float silu(float x) { return x / (1.0f + expf(-x)); }
__device__ __forceinline__ float silu_grad(float x) {
    const float s = 1.0f / (1.0f + expf(-x));
    return s * (1.0f + x * (1.0f - s));
}

// mode 0: gelu fwd (b unused) 1: gelu bwd (a=u, b=dg) 2: silu fwd 3: silu bwd (a=x, b=dy)
template <typename T, int MODE>
__global__ void __launch_bounds__(TPB) act_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ o,
                                                  int64_t n) {
    const int64_t nv = n / 4;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < nv; i += (int64_t)gridDim.x * TPB) {
        const float4 x = load4<T>(a + i * 4);
        float4 y = f4(0.f);
        if (MODE == 1 || MODE == 3) y = load4<T>(b + i * 4);
        float4 r;
        if (MODE == 0) r = make_float4(gelu_tanh(x.x), gelu_tanh(x.y), gelu_tanh(x.z), gelu_tanh(x.w));
        else if (MODE == 1) r = make_float4(y.x * gelu_tanh_grad(x.x), y.y * gelu_tanh_grad(x.y), y.z * gelu_tanh_grad(x.z), y.w * gelu_tanh_grad(x.w));
        else if (MODE == 2) r = make_float4(silu(x.x), silu(x.y), silu(x.z), silu(x.w));
        else r = make_float4(y.x * silu_grad(x.x), y.y * silu_grad(x.y), y.z * silu_grad(x.z), y.w * silu_grad(x.w));
        store4<T>(o + i * 4, r);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {  // tail
        const int64_t i = nv * 4 + threadIdx.x;
        const float x = to_f32<T>(a[i]);
        const float y = (MODE == 1 || MODE == 3) ? to_f32<T>(b[i]) : 0.f;
        float r;
        if (MODE == 0) r = gelu_tanh(x);
        else if (MODE == 1) r = y * gelu_tanh_grad(x);
        else if (MODE == 2) r = silu(x);
        else r = y * silu_grad(x);
        o[i] = from_f32<T>(r);
    }
}

template <int MODE>
static int launch_act(const void* a, const void* b, void* o, int dtype, int64_t n, sedd_stream_t stream, const char* nm) {
    SEDD_CHECK_ARG(dt_ok(dtype), "%s: bad dtype", nm);
    SEDD_CHECK_ARG(n >= 0, "%s: bad size", nm);
    if (n == 0) return SEDD_OK;
    SEDD_CHECK_ARG(a && o && (MODE == 0 || MODE == 2 || b), "%s: null pointer", nm);
    const dim3 grid(sedd_grid_for(n / 4 + 1, TPB, 256 * 16));
    DISPATCH_DT(dtype, T, hipLaunchKernelGGL((act_kernel<T, MODE>), grid, dim3(TPB), 0, (hipStream_t)stream,
                                             (const T*)a, (const T*)b, (T*)o, n));
    SEDD_LAUNCH_CHECK();
    return SEDD_OK;
}
extern "C" int sedd_gelu_fwd(const void* u, void* g, int dtype, int64_t n, sedd_stream_t stream) {
    return launch_act<0>(u, nullptr, g, dtype, n, stream, "gelu_fwd");
}
extern "C" int sedd_gelu_bwd(const void* u, const void* dg, void* du, int dtype, int64_t n, sedd_stream_t stream) {
    return launch_act<1>(u, dg, du, dtype, n, stream, "gelu_bwd");
}
extern "C" int sedd_silu_fwd(const float* x, float* y, int64_t n, sedd_stream_t stream) {
    return launch_act<2>(x, nullptr, y, SEDD_F32, n, stream, "silu_fwd");
}
extern "C" int sedd_silu_bwd(const float* x, const float* dy, float* dx, int64_t n, sedd_stream_t stream) {
    return launch_act<3>(x, dy, dx, SEDD_F32, n, stream, "silu_bwd");
}

// ============================================================================== pooled conditioners of SEDD
// transformer.py:289-301: c' = mean([c, BN(esm_proj(esm)), BN(mol_proj(mol))]) over the terms that are given, BN =
// nn.BatchNorm1d over the batch axis (:259-260).  [B, C] with C = cond_dim: latency-sized, one thread per column.
struct CondTerm {
    const float* x;      // [B, C] projection output (NULL: term absent)
    const float* w;      // [C] BatchNorm weight
    const float* b;      // [C] BatchNorm bias
    float* rmean;        // [C] running mean (updated in training)
    float* rvar;         // [C] running variance
    float* mean;         // [C] statistics used (saved for the backward)
    float* rstd;
    float* dx;           // backward: [B, C]
    float* dw;           // backward: [C]
    float* db;
};

__global__ void __launch_bounds__(64) cond_combine_fwd_kernel(const float* __restrict__ c, CondTerm t0, CondTerm t1,
                                                              float* __restrict__ out, int training, float momentum,
                                                              float eps, int B, int C) {
    const int col = blockIdx.x * 64 + threadIdx.x;
    if (col >= C) return;
    const CondTerm ts[2] = {t0, t1};
    float mean[2], rstd[2], w[2], b[2];
    int n = 1;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const CondTerm& t = ts[k];
        mean[k] = 0.f; rstd[k] = 0.f; w[k] = 0.f; b[k] = 0.f;
        if (!t.x) continue;
        ++n;
        if (training) {
            float s = 0.f;
            for (int r = 0; r < B; ++r) s += t.x[(int64_t)r * C + col];
            const float m = s / (float)B;
            float v = 0.f;
            for (int r = 0; r < B; ++r) { const float d = t.x[(int64_t)r * C + col] - m; v += d * d; }
            const float var = v / (float)B;
            mean[k] = m;
            rstd[k] = 1.0f / sqrtf(var + eps);
            t.rmean[col] = (1.0f - momentum) * t.rmean[col] + momentum * m;
            t.rvar[col] = (1.0f - momentum) * t.rvar[col] + momentum * (v / (float)(B - 1));
        } else {
            mean[k] = t.rmean[col];
            rstd[k] = 1.0f / sqrtf(t.rvar[col] + eps);
        }
        t.mean[col] = mean[k];
        t.rstd[col] = rstd[k];
        w[k] = t.w[col];
        b[k] = t.b[col];
    }
    const float inv_n = 1.0f / (float)n;
    for (int r = 0; r < B; ++r) {
        const int64_t i = (int64_t)r * C + col;
        float acc = c[i];
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (ts[k].x) acc += (ts[k].x[i] - mean[k]) * rstd[k] * w[k] + b[k];
        out[i] = acc * inv_n;
    }
}

__global__ void __launch_bounds__(64) cond_combine_bwd_kernel(const float* __restrict__ dout, CondTerm t0, CondTerm t1,
                                                              float* __restrict__ dc, int training, int accumulate,
                                                              int B, int C) {
    const int col = blockIdx.x * 64 + threadIdx.x;
    if (col >= C) return;
    const CondTerm ts[2] = {t0, t1};
    const float inv_n = 1.0f / (float)(1 + (t0.x != nullptr) + (t1.x != nullptr));
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const CondTerm& t = ts[k];
        if (!t.x) continue;
        const float m = t.mean[col], rs = t.rstd[col], w = t.w[col];
        float sdy = 0.f, sdyx = 0.f;
        for (int r = 0; r < B; ++r) {
            const int64_t i = (int64_t)r * C + col;
            const float dy = dout[i] * inv_n;
            sdy += dy;
            sdyx += dy * ((t.x[i] - m) * rs);
        }
        t.dw[col] = (accumulate ? t.dw[col] : 0.f) + sdyx;
        t.db[col] = (accumulate ? t.db[col] : 0.f) + sdy;
        if (t.dx) {
            const float invB = 1.0f / (float)B;
            for (int r = 0; r < B; ++r) {
                const int64_t i = (int64_t)r * C + col;
                const float dy = dout[i] * inv_n;
                t.dx[i] = training ? w * rs * (dy - invB * sdy - (t.x[i] - m) * rs * (invB * sdyx)) : dy * w * rs;
            }
        }
    }
    if (dc)
        for (int r = 0; r < B; ++r) dc[(int64_t)r * C + col] = dout[(int64_t)r * C + col] * inv_n;
}

static bool cond_term_ok(const CondTerm& t, bool bwd) {
    if (!t.x) return true;
    return bwd ? (t.w && t.mean && t.rstd && t.dw && t.db) : (t.w && t.b && t.rmean && t.rvar && t.mean && t.rstd);
}

extern "C" int sedd_cond_combine_fwd(const float* c, const float* e, const float* e_w, const float* e_b, float* e_rmean,
                                     float* e_rvar, float* e_mean, float* e_rstd, const float* m, const float* m_w,
                                     const float* m_b, float* m_rmean, float* m_rvar, float* m_mean, float* m_rstd,
                                     float* out, int training, float momentum, float eps, int B, int C,
                                     sedd_stream_t stream) {
    SEDD_CHECK_ARG(B >= 0 && C > 0, "cond_combine_fwd: bad sizes B=%d C=%d", B, C);
    if (B == 0) return SEDD_OK;
    SEDD_CHECK_ARG(!(training && B < 2 && (e || m)),
                   "cond_combine_fwd: Expected more than 1 value per channel when training, got input size [%d, %d]", B, C);
    SEDD_CHECK_ARG(momentum >= 0.f && momentum <= 1.f && eps > 0.f, "cond_combine_fwd: bad momentum / eps");
    CondTerm t0{e, e_w, e_b, e_rmean, e_rvar, e_mean, e_rstd, nullptr, nullptr, nullptr};
    CondTerm t1{m, m_w, m_b, m_rmean, m_rvar, m_mean, m_rstd, nullptr, nullptr, nullptr};
    SEDD_CHECK_ARG(c && out && cond_term_ok(t0, false) && cond_term_ok(t1, false), "cond_combine_fwd: null pointer");
    hipLaunchKernelGGL(cond_combine_fwd_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, c, t0, t1, out,
                       training, momentum, eps, B, C);
    SEDD_LAUNCH_CHECK();
    return SEDD_OK;
}

extern "C" int sedd_cond_combine_bwd(const float* dout, const float* e, const float* e_w, const float* e_mean,
                                     const float* e_rstd, float* de, float* e_dw, float* e_db, const float* m,
                                     const float* m_w, const float* m_mean, const float* m_rstd, float* dm, float* m_dw,
                                     float* m_db, float* dc, int training, int accumulate, int B, int C,
                                     sedd_stream_t stream) {
    SEDD_CHECK_ARG(B >= 0 && C > 0, "cond_combine_bwd: bad sizes B=%d C=%d", B, C);
    if (B == 0) return SEDD_OK;
    CondTerm t0{e, e_w, nullptr, nullptr, nullptr, (float*)e_mean, (float*)e_rstd, de, e_dw, e_db};
    CondTerm t1{m, m_w, nullptr, nullptr, nullptr, (float*)m_mean, (float*)m_rstd, dm, m_dw, m_db};
    SEDD_CHECK_ARG(dout && cond_term_ok(t0, true) && cond_term_ok(t1, true), "cond_combine_bwd: null pointer");
    hipLaunchKernelGGL(cond_combine_bwd_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, dout, t0, t1, dc,
                       training, accumulate, B, C);
    SEDD_LAUNCH_CHECK();
    return SEDD_OK;
}

// ======================================================================================= timestep embedding
// transformer.py:70-89
__global__ void __launch_bounds__(TPB) timestep_embedding_kernel(const float* __restrict__ sigma,
                                                                 float* __restrict__ out, int B, int dim) {
    const int half = dim / 2;
    for (int i = blockIdx.x * TPB + threadIdx.x; i < B * half; i += gridDim.x * TPB) {
        const int b = i / half, k = i % half;
        const float f = expf((-9.210340371976184f * (float)k) / (float)half);
        const float arg = sigma[b] * f;
        out[(int64_t)b * dim + k] = cosf(arg);
        out[(int64_t)b * dim + half + k] = sinf(arg);
    }
}
extern "C" int sedd_timestep_embedding(const float* sigma, float* out, int B, int dim, sedd_stream_t stream) {
    SEDD_CHECK_ARG(B >= 0 && dim > 0 && dim % 2 == 0, "timestep_embedding: bad sizes B=%d dim=%d", B, dim);
    if (B == 0) return SEDD_OK;
    SEDD_CHECK_ARG(sigma && out, "timestep_embedding: null pointer");
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3(sedd_grid_for((int64_t)B * dim / 2, TPB)), dim3(TPB), 0,
                       (hipStream_t)stream, sigma, out, B, dim);
    SEDD_LAUNCH_CHECK();
    return SEDD_OK;
}

// ================================================================================================ embedding
// transformer.py:193-218
__global__ void __launch_bounds__(TPB) embedding_fwd_kernel(const float* __restrict__ table,
                                                            const int64_t* __restrict__ idx, float* __restrict__ out,
                                                            int64_t M, int D) {
    const int dv = D / 4;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < M * dv; i += (int64_t)gridDim.x * TPB) {
        const int64_t row = i / dv;
        const int c = (int)(i % dv) * 4;
        store4<float>(out + row * D + c, load4<float>(table + idx[row] * D + c));
    }
}
extern "C" int sedd_embedding_fwd(const float* table, const int64_t* idx, float* out, int64_t M, int D, int V,
                                  sedd_stream_t stream) {
    SEDD_CHECK_ARG(M >= 0 && D > 0 && V > 0, "embedding_fwd: bad sizes");
    if (D % 4 != 0) { sedd_set_error("embedding_fwd: D=%d unsupported (need D %% 4 == 0)", D); return SEDD_ERR_UNSUPPORTED_SHAPE; }
    if (M == 0) return SEDD_OK;
    SEDD_CHECK_ARG(table && idx && out, "embedding_fwd: null pointer");
    hipLaunchKernelGGL(embedding_fwd_kernel, dim3(sedd_grid_for(M * (D / 4), TPB, 256 * 16)), dim3(TPB), 0,
                       (hipStream_t)stream, table, idx, out, M, D);
    SEDD_LAUNCH_CHECK();
    return SEDD_OK;
}

// scatter-add.  Small vocab (<= 64 rows): a block reduces 512 rows x 64 columns into LDS first, so the global
// float atomics drop from M*D to (M/512)*V*D; large vocab: direct global atomics (little contention).
#define EMB_ROWS 512
__global__ void __launch_bounds__(TPB) embedding_bwd_small_kernel(const float* __restrict__ dout,
                                                                  const int64_t* __restrict__ idx,
                                                                  float* __restrict__ dtable, int64_t M, int D, int V) {
    __shared__ float acc[64 * 64];
    const int col = threadIdx.x & 63, rr = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + col;
    const int64_t r0 = (int64_t)blockIdx.y * EMB_ROWS;
    for (int i = threadIdx.x; i < V * 64; i += TPB) acc[i] = 0.f;
    __syncthreads();
    if (c < D)
        for (int r = rr; r < EMB_ROWS && r0 + r < M; r += TPB / 64)
            atomicAdd(&acc[(int)idx[r0 + r] * 64 + col], dout[(r0 + r) * D + c]);
    __syncthreads();
    if (c < D)
        for (int v = rr; v < V; v += TPB / 64) {
            const float s = acc[v * 64 + col];
            if (s != 0.f) atomicAdd(dtable + (int64_t)v * D + c, s);
        }
}
// V <= 32 (the protein vocabulary): 512 rows x 256 columns per block, every wave into its OWN [V][256] LDS slice with
// plain read-add-write (a slice address is touched by one lane only, in program order: no LDS atomics, which ran the
// kernel at 0.8 TB/s), 16 bytes of a row per lane; the four slices are folded on the way out.
// ws != NULL: the block's [V, 256-column] partial table goes to ws[blockIdx.y] (every entry written, zeros included) and
// embedding_fold_kernel adds the row blocks up in order — bit-reproducible; ws == NULL: fp32 atomics into dtable.
__global__ void __launch_bounds__(TPB) embedding_bwd_v32_kernel(const float* __restrict__ dout,
                                                                const int64_t* __restrict__ idx,
                                                                float* __restrict__ dtable, int64_t M, int D, int V,
                                                                float* __restrict__ ws) {
    __shared__ __attribute__((aligned(16))) float acc[(TPB / 64) * 32 * 256];
    const int lane = threadIdx.x & 63, rr = threadIdx.x >> 6;
    const int c = blockIdx.x * 256 + lane * 4;
    const int64_t r0 = (int64_t)blockIdx.y * EMB_ROWS;
    for (int i = threadIdx.x; i < (TPB / 64) * V * 256; i += TPB) acc[i] = 0.f;
    __syncthreads();
    float* mine = acc + rr * V * 256 + lane * 4;
    if (c < D)  // (D % 4 == 0)
        for (int r = rr; r < EMB_ROWS && r0 + r < M; r += TPB / 64) {
            const float4 x = load4<float>(dout + (r0 + r) * D + c);
            float* a = mine + (int)idx[r0 + r] * 256;
            const float4 o = load4<float>(a);
            store4<float>(a, make_float4(o.x + x.x, o.y + x.y, o.z + x.z, o.w + x.w));
        }
    __syncthreads();
    for (int i = threadIdx.x; i < V * 256; i += TPB) {
        const int col = blockIdx.x * 256 + (i & 255);
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < TPB / 64; ++w) sum += acc[w * V * 256 + i];
        if (col >= D) continue;
        if (ws) ws[((int64_t)blockIdx.y * V + (i >> 8)) * D + col] = sum;
        else if (sum != 0.f) atomicAdd(dtable + (int64_t)(i >> 8) * D + col, sum);
    }
}
__global__ void __launch_bounds__(TPB) embedding_fold_kernel(const float* __restrict__ ws, int nblk, int64_t VD,
                                                             float* __restrict__ dtable) {
    const int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (e >= VD) return;
    float s = 0.f;
    for (int b = 0; b < nblk; ++b) s += ws[(int64_t)b * VD + e];
    dtable[e] += s;
}
__global__ void __launch_bounds__(TPB) embedding_bwd_large_kernel(const float* __restrict__ dout,
                                                                  const int64_t* __restrict__ idx,
                                                                  float* __restrict__ dtable, int64_t M, int D) {
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < M * D; i += (int64_t)gridDim.x * TPB) {
        const int64_t row = i / D;
        atomicAdd(dtable + idx[row] * D + (i % D), dout[i]);
    }
}
// "owner computes": block (64 columns, vocabulary row v) adds the rows of dout whose token is v — no atomics, no scratch,
// fixed order.  Each of the block's 4 or 16 waves scans its own 64-index segments of the token list (one index per lane,
// the matches collected with a ballot) and keeps a running sum per column, eight matched rows in flight at a time; the
// partial sums meet in LDS in wave order.  Under the absorbing graph half the tokens of a batch are the mask token, so
// ONE vocabulary row owns half the rows of dout: its chain is split over the waves and D / 64 column blocks share it (a
// form with one serial chain per 1024 columns took 1.2 ms on that row at M = 2048; the first form, which tested every
// index in every thread, 290-360 us on every shape).  Every column block scans the whole token list, V * M * D / 4096
// wave passes in all, so this serves the shapes where V * M is small (the cross model: 2365 x 512 and 37 x 2048, 27 and
// 15 us; config_large's 2397 x 16384 took 494 us and stays with the atomic kernel).
#define EMB_OWNER_FLIGHT 8
template <int EMB_OWNER_WAVES>
__global__ void __launch_bounds__(64 * EMB_OWNER_WAVES) embedding_bwd_owner_kernel(const float* __restrict__ dout,
                                                                                   const int64_t* __restrict__ idx,
                                                                                   float* __restrict__ dtable, int64_t M,
                                                                                   int D) {
    __shared__ float part[EMB_OWNER_WAVES][64];
    const int v = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + lane;
    const bool live = col < D;
    const float* src = dout + (live ? col : 0);
    float s = 0.f;
    int any = 0;
    for (int64_t r0 = (int64_t)wave * 64; r0 < M; r0 += 64 * EMB_OWNER_WAVES) {
        const int64_t r = r0 + lane;
        unsigned long long mk = __ballot(r < M && idx[r] == v);  // (wave-uniform)
        any |= mk != 0;
        while (mk) {
            int b[EMB_OWNER_FLIGHT];
#pragma unroll
            for (int q = 0; q < EMB_OWNER_FLIGHT; ++q) {
                b[q] = mk ? __builtin_ctzll(mk) : -1;
                mk &= mk - 1;  // (0 stays 0)
            }
            float x[EMB_OWNER_FLIGHT];
#pragma unroll
            for (int q = 0; q < EMB_OWNER_FLIGHT; ++q) x[q] = (b[q] >= 0 && live) ? src[(r0 + b[q]) * D] : 0.f;
#pragma unroll
            for (int q = 0; q < EMB_OWNER_FLIGHT; ++q)
                if (b[q] >= 0) s += x[q];
        }
    }
    part[wave][lane] = s;
    any = __syncthreads_or(any);
    if (wave == 0 && live && any) {
        float t = part[0][lane];
#pragma unroll
        for (int w = 1; w < EMB_OWNER_WAVES; ++w) t += part[w][lane];
        dtable[(int64_t)v * D + col] += t;
    }
}
extern "C" int64_t sedd_embedding_bwd_ws_floats(int64_t M, int D, int V) {
    return (V <= 32 && D % 4 == 0) ? ((M + EMB_ROWS - 1) / EMB_ROWS) * (int64_t)V * D : 0;
}
extern "C" int sedd_embedding_bwd(const float* dout, const int64_t* idx, float* dtable, float* ws, int64_t M, int D, int V,
                                  sedd_stream_t stream) {
    SEDD_CHECK_ARG(M >= 0 && D > 0 && V > 0, "embedding_bwd: bad sizes");
    if (M == 0) return SEDD_OK;
    SEDD_CHECK_ARG(dout && idx && dtable, "embedding_bwd: null pointer");
    if (!(V <= 32 && ws) && (int64_t)V * M <= (1 << 22) && V <= 65535) {
        // (a wave per 128 tokens, 4 or 16 to the block: idle waves of a 1024-thread block cost more than they scan)
        if (M <= 1024)
            hipLaunchKernelGGL(embedding_bwd_owner_kernel<4>, dim3((D + 63) / 64, (unsigned)V), dim3(256), 0,
                               (hipStream_t)stream, dout, idx, dtable, M, D);
        else
            hipLaunchKernelGGL(embedding_bwd_owner_kernel<16>, dim3((D + 63) / 64, (unsigned)V), dim3(1024), 0,
                               (hipStream_t)stream, dout, idx, dtable, M, D);
    } else if (V <= 32 && D % 4 == 0 && ((uintptr_t)dout & 15) == 0) {
        const int nblk = (int)((M + EMB_ROWS - 1) / EMB_ROWS);
        hipLaunchKernelGGL(embedding_bwd_v32_kernel, dim3((D + 255) / 256, (unsigned)nblk), dim3(TPB), 0,
                           (hipStream_t)stream, dout, idx, dtable, M, D, V, ws);
        if (ws) {
            SEDD_LAUNCH_CHECK();
            const int64_t VD = (int64_t)V * D;
            hipLaunchKernelGGL(embedding_fold_kernel, dim3((unsigned)((VD + TPB - 1) / TPB)), dim3(TPB), 0,
                               (hipStream_t)stream, ws, nblk, VD, dtable);
        }
    } else if (V <= 64)
        hipLaunchKernelGGL(embedding_bwd_small_kernel, dim3((D + 63) / 64, (unsigned)((M + EMB_ROWS - 1) / EMB_ROWS)),
                           dim3(TPB), 0, (hipStream_t)stream, dout, idx, dtable, M, D, V);
    else
        hipLaunchKernelGGL(embedding_bwd_large_kernel, dim3(sedd_grid_for(M * D, TPB, 256 * 16)), dim3(TPB), 0,
                           (hipStream_t)stream, dout, idx, dtable, M, D);
    SEDD_LAUNCH_CHECK();
    return SEDD_OK;
}

// ========================================================================================= logits epilogue
// transformer.py:313-322
// A thread owns one 16-byte chunk of the flat [M*V] array (VEC) or one element (!VEC: unaligned base); the
// (row, column, batch) coordinates are divided out once per chunk and then advanced incrementally, so the per-element
// cost is a compare/subtract and the traffic one vector load + one vector store.
template <typename T, bool VEC>
__global__ void __launch_bounds__(TPB) logits_epilogue_kernel(T* __restrict__ x, const float* __restrict__ sigma,
                                                              const int64_t* __restrict__ idx, int scale_by_sigma,
                                                              int64_t M, int64_t L, int V) {
    constexpr int EPT = VEC ? 16 / (int)sizeof(T) : 1;
    const float lv = logf((float)(V - 1));
    const int64_t total = M * V;
    const int64_t nchunks = (total + EPT - 1) / EPT;
    const bool small = total < ((int64_t)1 << 31);
    for (int64_t c = (int64_t)blockIdx.x * TPB + threadIdx.x; c < nchunks; c += (int64_t)gridDim.x * TPB) {
        const int64_t i0 = c * EPT;
        int64_t row, b;
        if (small) {
            row = (uint32_t)i0 / (uint32_t)V;
            b = (uint32_t)row / (uint32_t)L;
        } else {
            row = i0 / V;
            b = row / L;
        }
        int j = (int)(i0 - row * V);
        int64_t rl = row - b * L;
        float el = scale_by_sigma ? to_f32<T>(from_f32<T>(logf(esigm1f(sigma[b])))) : 0.f;  // .log().to(x.dtype)
        int64_t tgt = idx[row];
        T buf[EPT];
        const bool full = i0 + EPT <= total;
        if (VEC && full) {
            *reinterpret_cast<uint4*>(buf) = *reinterpret_cast<const uint4*>(x + i0);
        } else {
#pragma unroll
            for (int k = 0; k < EPT; ++k)
                if (i0 + k < total) buf[k] = x[i0 + k];
        }
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            float v = to_f32<T>(buf[k]);
            if (scale_by_sigma) {
                v = to_f32<T>(from_f32<T>(v - el));
                v = v - lv;
            }
            if (j == tgt) v = 0.f;
            buf[k] = from_f32<T>(v);
            if (k + 1 < EPT && ++j == V) {  // step to the next row (and batch element) inside the chunk
                j = 0;
                ++row;
                if (row < M) {
                    tgt = idx[row];
                    if (++rl == L) {
                        rl = 0;
                        ++b;
                        if (scale_by_sigma) el = to_f32<T>(from_f32<T>(logf(esigm1f(sigma[b]))));
                    }
                }
            }
        }
        if (VEC && full) {
            *reinterpret_cast<uint4*>(x + i0) = *reinterpret_cast<const uint4*>(buf);
        } else {
#pragma unroll
            for (int k = 0; k < EPT; ++k)
                if (i0 + k < total) x[i0 + k] = buf[k];
        }
    }
}
extern "C" int sedd_logits_epilogue(void* logits, int dtype, const float* sigma, const int64_t* idx,
                                    int scale_by_sigma, int64_t M, int64_t L, int V, sedd_stream_t stream) {
    SEDD_CHECK_ARG(dt_ok(dtype), "logits_epilogue: bad dtype");
    SEDD_CHECK_ARG(M >= 0 && L > 0 && V > 1 && M % L == 0, "logits_epilogue: bad sizes");
    if (M == 0) return SEDD_OK;
    SEDD_CHECK_ARG(logits && idx && (!scale_by_sigma || sigma), "logits_epilogue: null pointer");
    const int ept = dtype == SEDD_F32 ? 4 : 8;
    if (((uintptr_t)logits & 15) == 0) {
        DISPATCH_DT(dtype, T, hipLaunchKernelGGL((logits_epilogue_kernel<T, true>),
                                                 dim3(sedd_grid_for((M * V + ept - 1) / ept, TPB, 256 * 16)), dim3(TPB), 0,
                                                 (hipStream_t)stream, (T*)logits, sigma, idx, scale_by_sigma, M, L, V));
    } else {
        DISPATCH_DT(dtype, T, hipLaunchKernelGGL((logits_epilogue_kernel<T, false>),
                                                 dim3(sedd_grid_for(M * V, TPB, 256 * 16)), dim3(TPB), 0,
                                                 (hipStream_t)stream, (T*)logits, sigma, idx, scale_by_sigma, M, L, V));
    }
    SEDD_LAUNCH_CHECK();
    return SEDD_OK;
}
template <typename T>
__global__ void __launch_bounds__(TPB) logits_epilogue_bwd_kernel(T* __restrict__ dl, const int64_t* __restrict__ idx,
                                                                  int64_t M, int V) {
    for (int64_t r = (int64_t)blockIdx.x * TPB + threadIdx.x; r < M; r += (int64_t)gridDim.x * TPB) {
        const int64_t j = idx[r];
        if (j >= 0 && j < V) dl[r * V + j] = from_f32<T>(0.f);
    }
}
extern "C" int sedd_logits_epilogue_bwd(void* dlogits, int dtype, const int64_t* idx, int64_t M, int V,
                                        sedd_stream_t stream) {
    SEDD_CHECK_ARG(dt_ok(dtype), "logits_epilogue_bwd: bad dtype");
    SEDD_CHECK_ARG(M >= 0 && V > 0, "logits_epilogue_bwd: bad sizes");
    if (M == 0) return SEDD_OK;
    SEDD_CHECK_ARG(dlogits && idx, "logits_epilogue_bwd: null pointer");
    DISPATCH_DT(dtype, T, hipLaunchKernelGGL((logits_epilogue_bwd_kernel<T>), dim3(sedd_grid_for(M, TPB)), dim3(TPB),
                                             0, (hipStream_t)stream, (T*)dlogits, idx, M, V));
    SEDD_LAUNCH_CHECK();
    return SEDD_OK;
}

// ================================================================================================== colsum
#define CS_ROWS 128
template <typename T>
__global__ void __launch_bounds__(TPB) colsum_kernel(const T* __restrict__ a, int64_t lda, float* __restrict__ out,
                                                     int64_t M, int N, float* __restrict__ ws) {
    __shared__ float red[(TPB / 64) * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int64_t r0 = (int64_t)blockIdx.y * CS_ROWS;
    float s = 0.f;
    if (c < N)
        for (int r = wave; r < CS_ROWS && r0 + r < M; r += TPB / 64) s += to_f32<T>(a[(r0 + r) * lda + c]);
    red[wave * 64 + lane] = s;
    __syncthreads();
    if (wave == 0 && c < N) {
        for (int wv = 1; wv < TPB / 64; ++wv) s += red[wv * 64 + lane];
        if (ws) ws[(int64_t)blockIdx.y * N + c] = s;  // (folded in row-block order by colsum_rows_fold_kernel)
        else atomicAdd(out + c, s);
    }
}
// out[c] += sum over the row blocks b of ws[b][c], fixed order: one block per 4 columns, 64 contiguous ranges of row blocks
// per column (N is small where this runs — the [M, V] logits gradient — and the row blocks are many), xor-tree combine
__global__ void __launch_bounds__(TPB) colsum_rows_fold_kernel(const float* __restrict__ ws, int nblk, int N,
                                                               float* __restrict__ out) {
    static_assert(TPB == 256, "4 columns x 64 ranges");
    const int lane = threadIdx.x & 63, c = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int per = (nblk + 63) / 64, b0 = lane * per, b1 = b0 + per < nblk ? b0 + per : nblk;
    float s = 0.f;
    if (c < N)
        for (int b = b0; b < b1; ++b) s += ws[(int64_t)b * N + c];
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0 && c < N) out[c] += s;
}
// a few thousand rows (the dual-track model's batch of 4): ONE launch — no scratch, no fold launch, no memset.  The sum
// is latency-bound (megabytes, read once, cold): a block takes 8 columns x 32 row groups and every thread keeps 16
// independent loads in flight, so 2048 rows are 4 round trips to memory; the 32 partials of a column are combined
// through LDS in a fixed order.
#define CS_TALL_MAX 8192
template <typename T>
__global__ void __launch_bounds__(TPB) colsum_tall_kernel(const T* __restrict__ a, int64_t lda, float* __restrict__ out,
                                                          int64_t M, int N, int accumulate) {
    static_assert(TPB == 256, "32 row groups x 8 columns");
    __shared__ float red[32][8];
    const int cl = threadIdx.x & 7, rg = threadIdx.x >> 3, c = blockIdx.x * 8 + cl;
    float s[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) s[u] = 0.f;
    if (c < N) {
        const T* p = a + c;
        int64_t r = rg;
        for (; r + 15 * 32 < M; r += 16 * 32) {
#pragma unroll
            for (int u = 0; u < 16; ++u) s[u] += to_f32<T>(p[(r + u * 32) * lda]);
        }
        for (; r < M; r += 32) s[0] += to_f32<T>(p[r * lda]);
    }
#pragma unroll
    for (int w = 8; w > 0; w >>= 1)
#pragma unroll
        for (int u = 0; u < w; ++u) s[u] += s[u + w];
    red[rg][cl] = s[0];
    __syncthreads();
    if (rg == 0 && c < N) {
        float t = red[0][cl];
#pragma unroll
        for (int g = 1; g < 32; ++g) t += red[g][cl];
        out[c] = accumulate ? out[c] + t : t;
    }
}
extern "C" int sedd_colsum(const void* a, int dtype, int64_t lda, float* out, float* ws, int accumulate, int64_t M, int N,
                           sedd_stream_t stream) {
    SEDD_CHECK_ARG(dt_ok(dtype), "colsum: bad dtype");
    SEDD_CHECK_ARG(M >= 0 && N > 0 && lda >= N, "colsum: bad sizes");
    SEDD_CHECK_ARG(out, "colsum: out is null");
    hipStream_t st = (hipStream_t)stream;
    if (M > 0 && M <= CS_TALL_MAX && N >= 256) {
        SEDD_CHECK_ARG(a, "colsum: a is null");
        DISPATCH_DT(dtype, T, hipLaunchKernelGGL((colsum_tall_kernel<T>), dim3((N + 7) / 8), dim3(TPB), 0, st, (const T*)a,
                                                 lda, out, M, N, accumulate));
        SEDD_LAUNCH_CHECK();
        return SEDD_OK;
    }
    if (!accumulate) {
        if (hipMemsetAsync(out, 0, sizeof(float) * N, st) != hipSuccess) { sedd_set_error("colsum: memset failed"); return SEDD_ERR_LAUNCH; }
    }
    if (M == 0) return SEDD_OK;
    SEDD_CHECK_ARG(a, "colsum: a is null");
    const dim3 grid((N + 63) / 64, (unsigned)((M + CS_ROWS - 1) / CS_ROWS));
    if (grid.y == 1) ws = nullptr;  // one row block: nothing to order
    DISPATCH_DT(dtype, T, hipLaunchKernelGGL((colsum_kernel<T>), grid, dim3(TPB), 0, st, (const T*)a, lda, out, M, N, ws));
    if (ws) {
        SEDD_LAUNCH_CHECK();
        hipLaunchKernelGGL(colsum_rows_fold_kernel, dim3((N + 3) / 4), dim3(TPB), 0, st, ws, (int)grid.y, N, out);
    }
    SEDD_LAUNCH_CHECK();
    return SEDD_OK;
}

// =============================================================================================== transpose
template <typename TI, typename TO>
__global__ void __launch_bounds__(TPB) transpose_kernel(const TI* __restrict__ a, int64_t lda, TO* __restrict__ o,
                                                        int64_t ldo, int64_t M, int64_t N) {
    __shared__ float tile[64][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t m0 = (int64_t)blockIdx.y * 64, n0 = (int64_t)blockIdx.x * 64;
    for (int r = ty; r < 64; r += TPB / 64)
        if (m0 + r < M && n0 + tx < N) tile[r][tx] = to_f32<TI>(a[(m0 + r) * lda + n0 + tx]);
    __syncthreads();
    for (int r = ty; r < 64; r += TPB / 64)
        if (n0 + r < N && m0 + tx < M) o[(n0 + r) * ldo + m0 + tx] = from_f32<TO>(tile[tx][r]);
}
extern "C" int sedd_transpose(const void* a, int a_dtype, int64_t lda, void* out, int out_dtype, int64_t ldo,
                              int64_t M, int64_t N, sedd_stream_t stream) {
    SEDD_CHECK_ARG(dt_ok(a_dtype) && dt_ok(out_dtype), "transpose: bad dtype");
    SEDD_CHECK_ARG(M >= 0 && N >= 0 && lda >= N && ldo >= M, "transpose: bad sizes");
    if (M == 0 || N == 0) return SEDD_OK;
    SEDD_CHECK_ARG(a && out, "transpose: null pointer");
    const dim3 grid((unsigned)((N + 63) / 64), (unsigned)((M + 63) / 64));
    DISPATCH_DT(a_dtype, TI, DISPATCH_DT(out_dtype, TO,
        hipLaunchKernelGGL((transpose_kernel<TI, TO>), grid, dim3(TPB), 0, (hipStream_t)stream, (const TI*)a, lda,
                           (TO*)out, ldo, M, N)));
    SEDD_LAUNCH_CHECK();
    return SEDD_OK;
}

// ==================================================================================================== cast
template <typename TI, typename TO>
__global__ void __launch_bounds__(TPB) cast_kernel(const TI* __restrict__ a, TO* __restrict__ o, int64_t n) {
    const int64_t nv = n / 4;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < nv; i += (int64_t)gridDim.x * TPB)
        store4<TO>(o + i * 4, load4<TI>(a + i * 4));
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) o[nv * 4 + threadIdx.x] = from_f32<TO>(to_f32<TI>(a[nv * 4 + threadIdx.x]));
}
extern "C" int sedd_cast(const void* in, int in_dtype, void* out, int out_dtype, int64_t n, sedd_stream_t stream) {
    SEDD_CHECK_ARG(dt_ok(in_dtype) && dt_ok(out_dtype), "cast: bad dtype");
    SEDD_CHECK_ARG(n >= 0, "cast: bad size");
    if (n == 0) return SEDD_OK;
    SEDD_CHECK_ARG(in && out, "cast: null pointer");
    const dim3 grid(sedd_grid_for(n / 4 + 1, TPB, 256 * 16));
    DISPATCH_DT(in_dtype, TI, DISPATCH_DT(out_dtype, TO,
        hipLaunchKernelGGL((cast_kernel<TI, TO>), grid, dim3(TPB), 0, (hipStream_t)stream, (const TI*)in, (TO*)out, n)));
    SEDD_LAUNCH_CHECK();
    return SEDD_OK;
}

// ============================================================================================ scale by a device scalar
// x *= *s (fp32 product, rounded to x's type): the upstream gradient of the scalar loss applied to the gradient the
// fused loss kernel produced in its forward pass (losses.py:29's .mean() backward), without reading *s on the host.
template <typename T>
__global__ void __launch_bounds__(TPB) scale_kernel(T* __restrict__ x, const float* __restrict__ s, int64_t n) {
    const float f = s[0];
    const int64_t nv = n / 4;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < nv; i += (int64_t)gridDim.x * TPB) {
        const float4 v = load4<T>(x + i * 4);
        store4<T>(x + i * 4, make_float4(v.x * f, v.y * f, v.z * f, v.w * f));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3))
        x[nv * 4 + threadIdx.x] = from_f32<T>(to_f32<T>(x[nv * 4 + threadIdx.x]) * f);
}
extern "C" int sedd_scale_(void* x, int dtype, const float* scale, int64_t n, sedd_stream_t stream) {
    SEDD_CHECK_ARG(dt_ok(dtype), "scale_: bad dtype");
    SEDD_CHECK_ARG(n >= 0, "scale_: bad size");
    if (n == 0) return SEDD_OK;
    SEDD_CHECK_ARG(x && scale && ((uintptr_t)x & 15) == 0, "scale_: null or unaligned pointer");
    const dim3 grid(sedd_grid_for(n / 4 + 1, TPB, 256 * 16));
    DISPATCH_DT(dtype, T, hipLaunchKernelGGL((scale_kernel<T>), grid, dim3(TPB), 0, (hipStream_t)stream, (T*)x, scale, n));
    SEDD_LAUNCH_CHECK();
    return SEDD_OK;
}
