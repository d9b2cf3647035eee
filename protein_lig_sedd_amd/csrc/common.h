// Shared device/host helpers for libsedd_hip (gfx950 / CDNA4 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/sedd_hip.h"

typedef __bf16 bf16_t;

// ---- error plumbing (thread-local last-error string; see sedd_last_error in abi.hip) -------------------
void sedd_set_error(const char* fmt, ...);
#define SEDD_CHECK_ARG(cond, ...)                                   \
    do {                                                            \
        if (!(cond)) {                                              \
            sedd_set_error(__VA_ARGS__);                            \
            return SEDD_ERR_BAD_ARG;                                \
        }                                                           \
    } while (0)
#define SEDD_LAUNCH_CHECK()                                                          \
    do {                                                                             \
        hipError_t e__ = hipGetLastError();                                          \
        if (e__ != hipSuccess) {                                                     \
            sedd_set_error("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
            return SEDD_ERR_LAUNCH;                                                  \
        }                                                                            \
    } while (0)

// ---- dtype helpers --------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }

// ---- reductions inside a G-lane sub-group of a wave (G power of two <= 64) ------------------------------
template <int G> __device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
template <int G> __device__ __forceinline__ float group_max(float v) {
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
// argmax with torch semantics: first index among equal maxima; NaN counts as the maximum.
__device__ __forceinline__ bool argmax_better(float v, int i, float bv, int bi) {
    const bool vn = v != v, bn = bv != bv;
    if (vn || bn) return vn && (!bn || i < bi);
    return v > bv || (v == bv && i < bi);
}
template <int G> __device__ __forceinline__ void group_argmax(float& v, int& idx) {
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) {
        const float ov = __shfl_xor(v, off, 64);
        const int oi = __shfl_xor(idx, off, 64);
        if (argmax_better(ov, oi, v, idx)) { v = ov; idx = oi; }
    }
}
__device__ __forceinline__ float wave_sum(float v) { return group_sum<64>(v); }

// e^sigma - 1 as the reference computes it (graph_lib.py:279-283)
__device__ __forceinline__ float esigm1f(float s) { return s < 0.5f ? expm1f(s) : expf(s) - 1.0f; }

static inline int sedd_grid_for(long work_items, int per_block, int cap = 256 * 8) {
    long g = (work_items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}
static inline int sedd_group_for(int V) { return V <= 8 ? 8 : V <= 16 ? 16 : V <= 32 ? 32 : 64; }

#define SEDD_DISPATCH_GROUP(V, ...)                       \
    do {                                                  \
        switch (sedd_group_for(V)) {                      \
            case 8:  { constexpr int G = 8;  __VA_ARGS__; } break; \
            case 16: { constexpr int G = 16; __VA_ARGS__; } break; \
            case 32: { constexpr int G = 32; __VA_ARGS__; } break; \
            default: { constexpr int G = 64; __VA_ARGS__; } break; \
        }                                                 \
    } while (0)
