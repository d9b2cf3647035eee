// Softmax attention forward / backward on MFMA tiles — the flash-attn boundary of the reference
// (model/transformer.py:172-183, flash_attn_varlen_qkvpacked_func: non-causal, no dropout, scale d^-1/2) and the
// core of nn.MultiheadAttention (transformers_protlig_cross.py:135-138, :305), as one strided kernel family
// (Lq != Lk and a key mask allowed).  Written for 64-lane waves from the start:
//
//  * every product is computed TRANSPOSED (S^T = K Q^T, O^T = V^T P^T, dQ^T = K^T dS^T, ...) with
//    v_mfma_f32_16x16x32_bf16, so the softmax axis of a query lives in the registers of ONE lane column
//    (C/D map: col = lane & 15 = query, row = 4*(lane >> 4) + reg = key): row max / row sum are 16 local ops plus
//    two xor-shuffles, the running max / sum / LSE / delta of a query are lane-local scalars, and the fp32
//    probability tile is re-used in place as the next MFMA's B operand (accumulator-as-operand, no LDS round trip);
//  * the operand that must be read "column-wise" (V for O^T, K for dQ^T, dO/Q for dV^T/dK^T) is read from the same
//    row-major LDS tile with ds_read_b64_tr_b16 (hardware transposed read), so each K/V/Q/dO tile is staged once;
//  * K/V tiles are staged through LDS in 16-byte chunks with a 16-byte row pad (conflict-free ds_read_b128).
//
// The fp32 instantiation (exact v_mfma_f32_16x16x4_f32) exists for the fp32 parity configuration and the
// fp32-autocast cross model.  Backward is two deterministic kernels (dQ per query block, dK/dV per key block; no
// atomics); delta = rowsum(dO * O) is computed by the dQ kernel for its own queries and handed to the dK/dV kernel.
// MFMA-bound: forward 4*B*H*Lq*Lk*d flop, backward 14*B*H*Lq*Lk*d (7 products incl. the recomputed S and dP).
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// Occupancy targets (waves per SIMD) for the bf16 d <= 64 instantiations, where the register budget allows them
// without spilling (same-device A/B: forward 119 -> 96 us, backward 432 -> 351 us at B64 H12 L512 d64 together with
// -amdgpu-mfma-vgpr-form, which keeps the MFMA accumulators the softmax works on out of the AGPR half).
#define ATTN_WPE(n) __attribute__((amdgpu_waves_per_eu(n)))
#ifndef ATTN_FWD_WPE
#define ATTN_FWD_WPE 3
#endif
#ifndef ATTN_DQ_WPE
#define ATTN_DQ_WPE 1
#endif
#ifndef ATTN_DKV_WPE
#define ATTN_DKV_WPE 2
#endif
template <typename T, int D> constexpr int wpe_if_small(int n) { return sizeof(T) == 2 && D <= 64 ? n : 1; }
#define TILE 64          // rows of a staged LDS tile (keys in fwd / dQ, queries in dK/dV)
#define OWN_PER_WAVE 32  // rows owned by a wave (2 MFMA tiles)
#define OWN_PER_BLOCK 128
#define LOG2E 1.4426950408889634f
#define LN2 0.6931471805599453f

template <typename T> struct Op;
#ifndef ATTN_PAD
#define ATTN_PAD 16  // row pitch = 2(D + 16) B = 32 mod 64: ds_read_b128 row fragments and ds_read_b64_tr_b16 are conflict-free
#endif
template <> struct Op<bf16_t> { using type = bf16x8; static constexpr int KS = 32; static constexpr int PAD = ATTN_PAD; };
template <> struct Op<float> { using type = float; static constexpr int KS = 4; static constexpr int PAD = 4; };

template <typename T, int D> struct RowFrag { typename Op<T>::type v[D / Op<T>::KS]; };

// fragment of the tile row this lane owns (row = lane & 15 of the tile), g = lane >> 4
template <typename T, int D> __device__ __forceinline__ RowFrag<T, D> load_rowfrag(const T* row, int g) {
    RowFrag<T, D> f;
    if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int s = 0; s < D / 32; ++s) f.v[s] = *reinterpret_cast<const bf16x8*>(row + 32 * s + 8 * g);
    } else {
#pragma unroll
        for (int s = 0; s < D / 4; ++s) f.v[s] = row[4 * s + g];
    }
    return f;
}
template <typename T, int D> __device__ __forceinline__ RowFrag<T, D> zero_rowfrag() {
    RowFrag<T, D> f;
#pragma unroll
    for (int s = 0; s < D / Op<T>::KS; ++s) {
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int j = 0; j < 8; ++j) f.v[s][j] = (__bf16)0.f;
        } else f.v[s] = 0.f;
    }
    return f;
}

// acc[16x16] += A_rows x B_rows^T over the head dim (C/D: col = B's row (lane & 15), row = A's row 4g + reg)
template <typename T, int D>
__device__ __forceinline__ f32x4 mma_rows(const RowFrag<T, D>& a, const RowFrag<T, D>& b, f32x4 acc) {
#pragma unroll
    for (int s = 0; s < D / Op<T>::KS; ++s) {
        if constexpr (sizeof(T) == 2) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v[s], b.v[s], acc, 0, 0, 0);
        else acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[s], b.v[s], acc, 0, 0, 0);
    }
    return acc;
}

// accT[dt] (16 dims x 16 own rows) += X^T x W over the 64 rows of the LDS tile X[64][LS]; W[t][r] is this lane's
// fp32 weight for tile row 16t + 4g + r and its own column (lane & 15).  bf16: rows are consumed 32 at a time in the
// permuted order (4g+j of tile 2s, then 16+4g+j of tile 2s+1) on BOTH operands, which a sum over rows allows.
template <typename T, int D>
__device__ __forceinline__ void mma_tr(f32x4 (&accT)[D / 16], const T* X, const f32x4 (&W)[4], int lane) {
    constexpr int LS = D + Op<T>::PAD;
    const int g = lane >> 4, i = lane & 15;
    if constexpr (sizeof(T) == 2) {
        const int q = i >> 2, p = i & 3;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 b;
#pragma unroll
            for (int j = 0; j < 4; ++j) { b[j] = (__bf16)W[2 * s][j]; b[4 + j] = (__bf16)W[2 * s + 1][j]; }
            const T* r0 = X + (32 * s + 4 * g + q) * LS + 4 * p;
            const T* r1 = r0 + 16 * LS;
#pragma unroll
            for (int dt = 0; dt < D / 16; ++dt) {
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(r0 + dt * 16));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(r1 + dt * 16));
                union { struct { s16x4 lo, hi; } s; bf16x8 v; } a;
                a.s.lo = lo; a.s.hi = hi;
                accT[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b, accT[dt], 0, 0, 0);
            }
        }
    } else {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const T* row = X + (16 * t + 4 * g + r) * LS + i;
#pragma unroll
                for (int dt = 0; dt < D / 16; ++dt)
                    accT[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(row[dt * 16], W[t][r], accT[dt], 0, 0, 0);
            }
    }
}

// The same product for the wave's two own 16-row tiles at once and for HALF of the staged tile (rows 32h .. 32h+31,
// i.e. the score fragments t = 2h and 2h+1): the transposed X fragment is read once and feeds both own tiles, and the
// half-tile granularity lets the kernels start these MFMAs as soon as the first two score fragments are through the
// softmax arithmetic while the other two are still in the MFMA pipe / VALU.
template <typename T, int D>
__device__ __forceinline__ void mma_tr_pair(f32x4 (&acc0)[D / 16], f32x4 (&acc1)[D / 16], const T* X, int h,
                                            const f32x4& W0a, const f32x4& W0b, const f32x4& W1a, const f32x4& W1b,
                                            int lane) {
    constexpr int LS = D + Op<T>::PAD;
    const int g = lane >> 4, i = lane & 15;
    if constexpr (sizeof(T) == 2) {
        const int q = i >> 2, p = i & 3;
        bf16x8 b0, b1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            b0[j] = (__bf16)W0a[j]; b0[4 + j] = (__bf16)W0b[j];
            b1[j] = (__bf16)W1a[j]; b1[4 + j] = (__bf16)W1b[j];
        }
        const T* r0 = X + (32 * h + 4 * g + q) * LS + 4 * p;
        const T* r1 = r0 + 16 * LS;
#pragma unroll
        for (int dt = 0; dt < D / 16; ++dt) {
            union { struct { s16x4 lo, hi; } s; bf16x8 v; } a;
            a.s.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(r0 + dt * 16));
            a.s.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(r1 + dt * 16));
            acc0[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b0, acc0[dt], 0, 0, 0);
            acc1[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b1, acc1[dt], 0, 0, 0);
        }
    } else {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const T* row = X + (16 * (2 * h + tt) + 4 * g + r) * LS + i;
                const float w0 = tt ? W0b[r] : W0a[r], w1 = tt ? W1b[r] : W1a[r];
#pragma unroll
                for (int dt = 0; dt < D / 16; ++dt) {
                    const float x = row[dt * 16];
                    acc0[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, w0, acc0[dt], 0, 0, 0);
                    acc1[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, w1, acc1[dt], 0, 0, 0);
                }
            }
    }
}

// max over the four 16-lane rows of a wave (lanes l, l^16, l^32, l^48) without an LDS round trip: v_permlane16_swap /
// v_permlane32_swap exchange rows between two copies of the value
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
__device__ __forceinline__ float max_over_rows(float v) {
    u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// fmaxf on MFMA outputs makes the compiler canonicalise each operand first (a v_max_f32 x, x, x per element: 16 extra VALU
// instructions per score tile in a VALU-bound loop) unless it may assume the operands are not signalling NaNs: this file
// is built with -fno-honor-nans (Makefile).  (Inline-asm v_max3 on MFMA results is NOT an option: the compiler's hazard
// recogniser does not cover inline asm, and the MFMA -> VALU read wait states went missing: wrong maxima.)  Hardware
// behaviour on NaN inputs is unchanged (v_max returns the other operand, arithmetic propagates).
__device__ __forceinline__ float vmax3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
__device__ __forceinline__ float vmax2(float a, float b) { return fmaxf(a, b); }
// max of the 16 scores of one own tile column held by this lane (8 instructions)
__device__ __forceinline__ float vmax16(const f32x4 (&S)[4]) {
    float mx = vmax3(S[0][0], S[0][1], S[0][2]);
    mx = vmax3(mx, S[0][3], S[1][0]);
    mx = vmax3(mx, S[1][1], S[1][2]);
    mx = vmax3(mx, S[1][3], S[2][0]);
    mx = vmax3(mx, S[2][1], S[2][2]);
    mx = vmax3(mx, S[2][3], S[3][0]);
    mx = vmax3(mx, S[3][1], S[3][2]);
    return vmax2(mx, S[3][3]);
}
__device__ __forceinline__ float max_over_rows_raw(float v) {
    u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = vmax2(__uint_as_float(r[0]), __uint_as_float(r[1]));
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return vmax2(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// mma_tr_pair for the forward's P V product with the row sums of P taken on the matrix pipe as well: one more MFMA per
// own tile and half with an all-ones A operand leaves sum_k P[k][query] in every register of lsum (all four lane groups
// alike, so no cross-lane step at the end) — 21 VALU adds per tile traded for 4 MFMAs on a pipe that is 30 % busy.  The
// sum is taken over the bf16-rounded probabilities, the ones P V multiplies with.
template <int D>
__device__ __forceinline__ void mma_tr_pair_sum(f32x4 (&acc0)[D / 16], f32x4 (&acc1)[D / 16], f32x4& lsum0, f32x4& lsum1,
                                                const bf16_t* X, int h, const f32x4& W0a, const f32x4& W0b,
                                                const f32x4& W1a, const f32x4& W1b, int lane) {
    constexpr int LS = D + Op<bf16_t>::PAD;
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    bf16x8 b0, b1, ones;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        b0[j] = (__bf16)W0a[j]; b0[4 + j] = (__bf16)W0b[j];
        b1[j] = (__bf16)W1a[j]; b1[4 + j] = (__bf16)W1b[j];
        ones[j] = (__bf16)1.0f; ones[4 + j] = (__bf16)1.0f;
    }
    const bf16_t* r0 = X + (32 * h + 4 * g + q) * LS + 4 * p;
    const bf16_t* r1 = r0 + 16 * LS;
#pragma unroll
    for (int dt = 0; dt < D / 16; ++dt) {
        union { struct { s16x4 lo, hi; } s; bf16x8 v; } a;
        a.s.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(r0 + dt * 16));
        a.s.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(r1 + dt * 16));
        acc0[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b0, acc0[dt], 0, 0, 0);
        acc1[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b1, acc1[dt], 0, 0, 0);
    }
    lsum0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, b0, lsum0, 0, 0, 0);
    lsum1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, b1, lsum1, 0, 0, 0);
}

// Staging of rows [r0, r0+64) of a strided [rows, D] global matrix into LDS (zero beyond nrows), 16-byte chunks, split
// in two halves: stage_load puts the tile into registers, stage_store parks it in LDS.  The kernels issue the loads
// of tile t+1 right after tile t has been parked, so the global-memory latency runs under tile t's MFMA / softmax work.
template <typename T, int D> struct TileRegs {
    static constexpr int EPC = 16 / sizeof(T);   // elements per chunk
    static constexpr int CPR = D / EPC;          // chunks per row
    static constexpr int NCH = TILE * CPR / 256; // chunks per thread (256 threads)
    static_assert(TILE * CPR % 256 == 0, "tile chunks divide over the block");
    uint4 v[NCH];
};
// (An unguarded variant for the mask-free instantiations was tried: hipcc then keeps the tile registers in scratch memory
// — scratch_store right behind each global load — and the forward goes from 78 to 150 us.)
template <typename T, int D>
__device__ __forceinline__ void stage_load(TileRegs<T, D>& t, const T* g, int64_t ls, int r0, int nrows) {
    using R = TileRegs<T, D>;
#pragma unroll
    for (int n = 0; n < R::NCH; ++n) {
        const int c = threadIdx.x + 256 * n, row = c / R::CPR, cc = c % R::CPR;
        t.v[n] = make_uint4(0, 0, 0, 0);
        if (r0 + row < nrows) t.v[n] = *reinterpret_cast<const uint4*>(g + (int64_t)(r0 + row) * ls + cc * R::EPC);
    }
}
template <typename T, int D> __device__ __forceinline__ void stage_store(T* lds, const TileRegs<T, D>& t) {
    using R = TileRegs<T, D>;
    constexpr int LS = D + Op<T>::PAD;
#pragma unroll
    for (int n = 0; n < R::NCH; ++n) {
        const int c = threadIdx.x + 256 * n, row = c / R::CPR, cc = c % R::CPR;
        *reinterpret_cast<uint4*>(lds + row * LS + cc * R::EPC) = t.v[n];
    }
}
// 2^x on the hardware unit (v_exp_f32); probabilities below the normal range flush to zero
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

template <typename T> __device__ __forceinline__ void store4T(T* p, f32x4 v);
template <> __device__ __forceinline__ void store4T<float>(float* p, f32x4 v) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <> __device__ __forceinline__ void store4T<bf16_t>(bf16_t* p, f32x4 v) {
    union { bf16_t h[4]; uint2 u; } c;
    c.h[0] = (bf16_t)v[0]; c.h[1] = (bf16_t)v[1]; c.h[2] = (bf16_t)v[2]; c.h[3] = (bf16_t)v[3];
    *reinterpret_cast<uint2*>(p) = c.u;
}

// Write one own row's gradient (acc * mul, rounded to T like the unfused kernels' output), optionally through the
// CONJUGATE rotary rotation of rotary.py:42-51's backward (cs / sn = this position's [D/2] table rows, or null): the
// pair (j, j + D/2) of a NeoX-halves rotation lives in fragments dt and dt + D/32 of the same lane.
template <typename T, int D>
__device__ __forceinline__ void store_row_rot(T* row, const f32x4 (&acc)[D / 16], float mul, const float* cs,
                                              const float* sn, int g) {
    if (!cs) {
#pragma unroll
        for (int dt = 0; dt < D / 16; ++dt) store4T<T>(row + dt * 16 + 4 * g, acc[dt] * mul);
        return;
    }
#pragma unroll
    for (int dt = 0; dt < D / 32; ++dt) {
        const float4 c = *reinterpret_cast<const float4*>(cs + dt * 16 + 4 * g);
        const float4 sv = *reinterpret_cast<const float4*>(sn + dt * 16 + 4 * g);
        const float cc[4] = {c.x, c.y, c.z, c.w}, ss[4] = {-sv.x, -sv.y, -sv.z, -sv.w};
        f32x4 lo = acc[dt] * mul, hi = acc[dt + D / 32] * mul, olo, ohi;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float x1 = sizeof(T) == 2 ? (float)(T)lo[r] : lo[r], x2 = sizeof(T) == 2 ? (float)(T)hi[r] : hi[r];
            olo[r] = x1 * cc[r] - x2 * ss[r];
            ohi[r] = x2 * cc[r] + x1 * ss[r];
        }
        store4T<T>(row + dt * 16 + 4 * g, olo);
        store4T<T>(row + D / 2 + dt * 16 + 4 * g, ohi);
    }
}

// 1-D launch, XCD-aware: the blocks that share a (batch, head) — and so re-read the same K/V (or Q/dO) tiles — get
// consecutive slots of ONE XCD instead of being dealt round-robin over the eight L2s (PMC: the 4 query blocks of a
// head each pulled K/V through a different L2, 450-620 MB of fabric reads per launch against 150-250 MB of operands).
__device__ __forceinline__ int attn_xcd_remap(int bid, int nwg) {
    const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

struct AttnShape {
    int H, Lq, Lk;
    int64_t q_bs, q_ls, kv_bs, kv_ls, o_bs, o_ls;
    float scale;
    const float* rot_cos;  // backward only: conjugate rotary on dQ / dK ([L, D/2] fp32 tables), or null
    const float* rot_sin;
    // variable-length packing (flash_attn_varlen_qkvpacked_func's cu_seqlens, transformer.py:173-181): when cu != null the
    // launch's "batch" index is a segment of the packed token axis: segment s = tokens [cu[s], cu[s+1]) of `total`, q / k /
    // v / o addressed by token (x_ls), lse / delta laid out [H, total]; Lq = Lk = the launch's maximum segment length
    const int* cu;
    int total;
};

// Per-block view of a launch: base offsets of the block's batch element (or packed segment) and its lengths.
struct AttnView { int64_t qo, ko, oo, lb; int Lq, Lk; };
template <bool PLAIN>
__device__ __forceinline__ AttnView attn_view(const AttnShape& sh, int b, int h) {
    AttnView v{b * sh.q_bs, b * sh.kv_bs, b * sh.o_bs, ((int64_t)b * sh.H + h) * sh.Lq, sh.Lq, sh.Lk};
    if constexpr (!PLAIN) {
        if (sh.cu) {
            int s0 = sh.cu[b], s1 = sh.cu[b + 1];
            s0 = s0 < 0 ? 0 : (s0 > sh.total ? sh.total : s0);  // never address outside the packed buffer
            s1 = s1 < s0 ? s0 : (s1 > sh.total ? sh.total : s1);
            const int len = s1 - s0 < sh.Lq ? s1 - s0 : sh.Lq;  // (segments longer than the stated maximum are cut there)
            v = AttnView{s0 * sh.q_ls, s0 * sh.kv_ls, s0 * sh.o_ls, (int64_t)h * sh.total + s0, len, len};
        }
    }
    return v;
}

// ==================================================================================================== forward
// PLAIN: no key mask and Lk a multiple of TILE — every score is valid (the training shapes); the general
// instantiation carries the per-key validity selects.
template <typename T, int D, bool PLAIN>
__global__ void __launch_bounds__(256) ATTN_WPE((wpe_if_small<T, D>(ATTN_FWD_WPE)))
attn_fwd_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v, T* __restrict__ o,
                float* __restrict__ lse, const uint8_t* __restrict__ mask, AttnShape sh) {
    constexpr int LS = D + Op<T>::PAD;
    __shared__ __attribute__((aligned(16))) T Ks[TILE * LS];
    __shared__ __attribute__((aligned(16))) T Vs[TILE * LS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, i = lane & 15;
    const int nob = (sh.Lq + OWN_PER_BLOCK - 1) / OWN_PER_BLOCK, lin = attn_xcd_remap(blockIdx.x, gridDim.x);
    const int bx = lin % nob, by = lin / nob;
    const int b = by / sh.H, h = by % sh.H;
    const AttnView av = attn_view<PLAIN>(sh, b, h);
    if (bx * OWN_PER_BLOCK >= av.Lq) return;  // (block-uniform: a packed segment shorter than the launch's maximum)
    const T* qb = q + av.qo + (int64_t)h * D;
    const T* kb = k + av.ko + (int64_t)h * D;
    const T* vb = v + av.ko + (int64_t)h * D;
    const uint8_t* mb = !PLAIN && mask ? mask + (int64_t)b * sh.Lk : nullptr;
    const float sl2 = sh.scale * LOG2E;
    sh.Lq = av.Lq;
    sh.Lk = av.Lk;

    RowFrag<T, D> qf[2];
    int qrow[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        qrow[mt] = bx * OWN_PER_BLOCK + wave * OWN_PER_WAVE + mt * 16 + i;
        qf[mt] = qrow[mt] < sh.Lq ? load_rowfrag<T, D>(qb + (int64_t)qrow[mt] * sh.q_ls, g) : zero_rowfrag<T, D>();
    }
    float m[2] = {-INFINITY, -INFINITY}, lsum[2] = {0.f, 0.f};
    f32x4 lacc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};  // bf16: row sums on the matrix pipe
    f32x4 acc[2][D / 16];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int dt = 0; dt < D / 16; ++dt) acc[mt][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    TileRegs<T, D> rk, rv;
    stage_load<T, D>(rk, kb, sh.kv_ls, 0, sh.Lk);
    stage_load<T, D>(rv, vb, sh.kv_ls, 0, sh.Lk);
    for (int k0 = 0; k0 < sh.Lk; k0 += TILE) {
        __syncthreads();
        stage_store<T, D>(Ks, rk);
        stage_store<T, D>(Vs, rv);
        __syncthreads();
        if (k0 + TILE < sh.Lk) {
            stage_load<T, D>(rk, kb, sh.kv_ls, k0 + TILE, sh.Lk);
            stage_load<T, D>(rv, vb, sh.kv_ls, k0 + TILE, sh.Lk);
        }
        f32x4 S[2][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const RowFrag<T, D> kf = load_rowfrag<T, D>(Ks + (16 * t + i) * LS, g);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) S[mt][t] = mma_rows<T, D>(kf, qf[mt], (f32x4){0.f, 0.f, 0.f, 0.f});
        }
        if constexpr (!PLAIN) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = k0 + 16 * t + 4 * g + r;
                    const bool valid = key < sh.Lk && (!mb || mb[key < sh.Lk ? key : 0]);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) S[mt][t][r] = valid ? S[mt][t][r] : -INFINITY;
                }
        }
        // running max on the raw scores (the scale is positive); the scale is folded into the exponent's fma
        float ms[2];
        if constexpr (sizeof(T) == 2) {
            // bf16: the loop is VALU-issue-bound (SQ counters: vector issue 77 % busy, matrix pipe 30 %), so instructions
            // are what is saved here.  The accumulators are rescaled only when some query's maximum has grown by more
            // than 2^RESCALE_LOG2 since the last rescale (one wave-uniform branch; until then the exponent is taken against
            // the older, smaller maximum: probabilities up to 2^8, exact in fp32 and well inside bf16's range; the LSE
            // written at the end is m + log2(sum) whatever m was used) — on the training shapes that is the first tile only.
            constexpr float RESCALE_LOG2 = 8.0f;
            float mn[2];
            bool grow = false;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const float mx = max_over_rows_raw(vmax16(S[mt])) * sl2;
                mn[mt] = vmax2(m[mt], mx);
                grow |= mn[mt] > m[mt] + RESCALE_LOG2;
            }
            if (__builtin_amdgcn_ballot_w64(grow) != 0) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const float mref = mn[mt] == -INFINITY ? 0.f : mn[mt];
                    const float alpha = fast_exp2(m[mt] - mref);
                    m[mt] = mn[mt];
                    lacc[mt] *= alpha;
#pragma unroll
                    for (int dt = 0; dt < D / 16; ++dt) acc[mt][dt] *= alpha;
                }
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) ms[mt] = m[mt] == -INFINITY ? 0.f : m[mt];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            S[mt][2 * hh + tt][r] = fast_exp2(fmaf(S[mt][2 * hh + tt][r], sl2, -ms[mt]));
                mma_tr_pair_sum<D>(acc[0], acc[1], lacc[0], lacc[1], (const bf16_t*)Vs, hh, S[0][2 * hh], S[0][2 * hh + 1],
                                   S[1][2 * hh], S[1][2 * hh + 1], lane);
            }
            continue;
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            float mx = -INFINITY;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, S[mt][t][r]);
            mx = max_over_rows(mx) * sl2;
            const float mn = fmaxf(m[mt], mx);
            ms[mt] = mn == -INFINITY ? 0.f : mn;  // fully masked so far: keep everything at 0
            const float alpha = fast_exp2(m[mt] - ms[mt]);
            m[mt] = mn;
            lsum[mt] *= alpha;
#pragma unroll
            for (int dt = 0; dt < D / 16; ++dt) acc[mt][dt] *= alpha;
        }
        // probabilities and P V, half a tile at a time: the second half's exponentials run under the first half's MFMAs
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float p = fast_exp2(fmaf(S[mt][2 * hh + tt][r], sl2, -ms[mt]));
                        S[mt][2 * hh + tt][r] = p;
                        lsum[mt] += p;
                    }
            mma_tr_pair<T, D>(acc[0], acc[1], Vs, hh, S[0][2 * hh], S[0][2 * hh + 1], S[1][2 * hh], S[1][2 * hh + 1], lane);
        }
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        float l = lsum[mt];
        if constexpr (sizeof(T) == 2) {
            l = lacc[mt][0];
        } else {
            l += __shfl_xor(l, 16, 64);
            l += __shfl_xor(l, 32, 64);
        }
        if (qrow[mt] < sh.Lq) {
            const float inv = l > 0.f ? 1.0f / l : 0.f;
            T* orow = o + av.oo + (int64_t)qrow[mt] * sh.o_ls + (int64_t)h * D;
#pragma unroll
            for (int dt = 0; dt < D / 16; ++dt) store4T<T>(orow + dt * 16 + 4 * g, acc[mt][dt] * inv);
            if (g == 0) lse[av.lb + qrow[mt]] = (m[mt] + log2f(l)) * LN2;
        }
    }
}

// ================================================================================================ dQ kernel
// The softmax scale of dS = P (dP - delta) * scale is applied once to the finished dQ / dK accumulators.
template <typename T, int D, bool PLAIN>
__global__ void __launch_bounds__(256) ATTN_WPE((wpe_if_small<T, D>(ATTN_DQ_WPE)))
attn_bwd_dq_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                   const T* __restrict__ o, const T* __restrict__ dout, const float* __restrict__ lse,
                   float* __restrict__ delta, const uint8_t* __restrict__ mask, T* __restrict__ dq, AttnShape sh) {
    constexpr int LS = D + Op<T>::PAD;
    __shared__ __attribute__((aligned(16))) T Ks[TILE * LS];
    __shared__ __attribute__((aligned(16))) T Vs[TILE * LS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, i = lane & 15;
    const int nob = (sh.Lq + OWN_PER_BLOCK - 1) / OWN_PER_BLOCK, lin = attn_xcd_remap(blockIdx.x, gridDim.x);
    const int bx = lin % nob, by = lin / nob;
    const int b = by / sh.H, h = by % sh.H;
    const AttnView av = attn_view<PLAIN>(sh, b, h);
    if (bx * OWN_PER_BLOCK >= av.Lq) return;
    const T* qb = q + av.qo + (int64_t)h * D;
    const T* kb = k + av.ko + (int64_t)h * D;
    const T* vb = v + av.ko + (int64_t)h * D;
    const T* dob = dout + av.oo + (int64_t)h * D;
    const uint8_t* mb = !PLAIN && mask ? mask + (int64_t)b * sh.Lk : nullptr;
    const float sl2 = sh.scale * LOG2E;
    sh.Lq = av.Lq;
    sh.Lk = av.Lk;

    RowFrag<T, D> qf[2], dof[2];
    int qrow[2];
    float lq[2], dl[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        qrow[mt] = bx * OWN_PER_BLOCK + wave * OWN_PER_WAVE + mt * 16 + i;
        const bool ok = qrow[mt] < sh.Lq;
        qf[mt] = ok ? load_rowfrag<T, D>(qb + (int64_t)qrow[mt] * sh.q_ls, g) : zero_rowfrag<T, D>();
        dof[mt] = ok ? load_rowfrag<T, D>(dob + (int64_t)qrow[mt] * sh.o_ls, g) : zero_rowfrag<T, D>();
        const int64_t si = av.lb + (ok ? qrow[mt] : 0);
        lq[mt] = ok ? lse[si] * LOG2E : 0.f;
        // delta = rowsum(dO * O) of this query (flash-attention backward's D), from the dO fragment already in
        // registers and the matching O fragment; also left in `delta` for the dK/dV kernel, which runs next
        const RowFrag<T, D> of = ok ? load_rowfrag<T, D>(o + av.oo + (int64_t)h * D + (int64_t)qrow[mt] * sh.o_ls, g)
                                    : zero_rowfrag<T, D>();
        float dsum = 0.f;
#pragma unroll
        for (int s = 0; s < D / Op<T>::KS; ++s) {
            if constexpr (sizeof(T) == 2) {
#pragma unroll
                for (int j = 0; j < 8; ++j) dsum = fmaf((float)dof[mt].v[s][j], (float)of.v[s][j], dsum);
            } else dsum = fmaf(dof[mt].v[s], of.v[s], dsum);
        }
        dsum += __shfl_xor(dsum, 16, 64);
        dsum += __shfl_xor(dsum, 32, 64);
        dl[mt] = dsum;
        if (ok && g == 0) delta[si] = dsum;
    }
    f32x4 acc[2][D / 16];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int dt = 0; dt < D / 16; ++dt) acc[mt][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    TileRegs<T, D> rk, rv;
    stage_load<T, D>(rk, kb, sh.kv_ls, 0, sh.Lk);
    stage_load<T, D>(rv, vb, sh.kv_ls, 0, sh.Lk);
    for (int k0 = 0; k0 < sh.Lk; k0 += TILE) {
        __syncthreads();
        stage_store<T, D>(Ks, rk);
        stage_store<T, D>(Vs, rv);
        __syncthreads();
        if (k0 + TILE < sh.Lk) {
            stage_load<T, D>(rk, kb, sh.kv_ls, k0 + TILE, sh.Lk);
            stage_load<T, D>(rv, vb, sh.kv_ls, k0 + TILE, sh.Lk);
        }
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {  // half a tile at a time (see mma_tr_pair)
            f32x4 S[2][2], dP[2][2];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int t = 2 * hh + tt;
                const RowFrag<T, D> kf = load_rowfrag<T, D>(Ks + (16 * t + i) * LS, g);
                const RowFrag<T, D> vf = load_rowfrag<T, D>(Vs + (16 * t + i) * LS, g);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    S[mt][tt] = mma_rows<T, D>(kf, qf[mt], (f32x4){0.f, 0.f, 0.f, 0.f});
                    // the accumulator starts at -delta of the lane's query: dP - delta comes out of the MFMA
                    dP[mt][tt] = mma_rows<T, D>(vf, dof[mt], (f32x4){-dl[mt], -dl[mt], -dl[mt], -dl[mt]});
                }
            }
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    bool valid = true;
                    if constexpr (!PLAIN) {
                        const int key = k0 + 16 * (2 * hh + tt) + 4 * g + r;
                        valid = key < sh.Lk && (!mb || mb[key < sh.Lk ? key : 0]);
                    }
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        const float p = valid ? fast_exp2(fmaf(S[mt][tt][r], sl2, -lq[mt])) : 0.f;
                        S[mt][tt][r] = p * dP[mt][tt][r];  // dS^T / scale
                    }
                }
            mma_tr_pair<T, D>(acc[0], acc[1], Ks, hh, S[0][0], S[0][1], S[1][0], S[1][1], lane);
        }
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
        if (qrow[mt] < sh.Lq) {
            T* row = dq + av.qo + (int64_t)qrow[mt] * sh.q_ls + (int64_t)h * D;
            const int64_t ro = (int64_t)qrow[mt] * (D / 2);
            store_row_rot<T, D>(row, acc[mt], sh.scale, sh.rot_cos ? sh.rot_cos + ro : nullptr, sh.rot_sin + ro, g);
        }
}

// ============================================================================================ dK / dV kernel
// PLAIN: no key mask and Lk a multiple of the block's 128 own keys.
template <typename T, int D, bool PLAIN>
__global__ void __launch_bounds__(256) ATTN_WPE((wpe_if_small<T, D>(ATTN_DKV_WPE)))
attn_bwd_dkv_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                    const T* __restrict__ dout, const float* __restrict__ lse, const float* __restrict__ delta,
                    const uint8_t* __restrict__ mask, T* __restrict__ dk, T* __restrict__ dv, AttnShape sh) {
    constexpr int LS = D + Op<T>::PAD;
    __shared__ __attribute__((aligned(16))) T Qs[TILE * LS];
    __shared__ __attribute__((aligned(16))) T dOs[TILE * LS];
    __shared__ __attribute__((aligned(16))) float lse_s[TILE], del_s[TILE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, i = lane & 15;
    const int nob = (sh.Lk + OWN_PER_BLOCK - 1) / OWN_PER_BLOCK, lin = attn_xcd_remap(blockIdx.x, gridDim.x);
    const int bx = lin % nob, by = lin / nob;
    const int b = by / sh.H, h = by % sh.H;
    const AttnView av = attn_view<PLAIN>(sh, b, h);
    if (bx * OWN_PER_BLOCK >= av.Lk) return;
    const T* qb = q + av.qo + (int64_t)h * D;
    const T* kb = k + av.ko + (int64_t)h * D;
    const T* vb = v + av.ko + (int64_t)h * D;
    const T* dob = dout + av.oo + (int64_t)h * D;
    const float sl2 = sh.scale * LOG2E;
    const int64_t mask_b = (int64_t)b * sh.Lk;  // (the key mask is laid out with the launch's Lk)
    sh.Lq = av.Lq;
    sh.Lk = av.Lk;

    RowFrag<T, D> kf[2], vf[2];
    int krow[2];
    bool kvalid[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        krow[mt] = bx * OWN_PER_BLOCK + wave * OWN_PER_WAVE + mt * 16 + i;
        const bool ok = krow[mt] < sh.Lk;
        kf[mt] = ok ? load_rowfrag<T, D>(kb + (int64_t)krow[mt] * sh.kv_ls, g) : zero_rowfrag<T, D>();
        vf[mt] = ok ? load_rowfrag<T, D>(vb + (int64_t)krow[mt] * sh.kv_ls, g) : zero_rowfrag<T, D>();
        kvalid[mt] = PLAIN || (ok && (!mask || mask[mask_b + krow[mt]]));
    }
    f32x4 adk[2][D / 16], adv[2][D / 16];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int dt = 0; dt < D / 16; ++dt) adk[mt][dt] = adv[mt][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    TileRegs<T, D> rq, rdo;
    float r_lse = INFINITY, r_del = 0.f;  // +inf -> p = 0 for padded queries
#define DKV_LOAD(q0_)                                                                            \
    do {                                                                                         \
        stage_load<T, D>(rq, qb, sh.q_ls, (q0_), sh.Lq);                                         \
        stage_load<T, D>(rdo, dob, sh.o_ls, (q0_), sh.Lq);                                       \
        if (threadIdx.x < TILE) {                                                                \
            const int qi = (q0_) + threadIdx.x;                                                  \
            const int64_t si = av.lb + (qi < sh.Lq ? qi : 0);                                    \
            r_lse = qi < sh.Lq ? lse[si] * LOG2E : INFINITY;                                     \
            r_del = qi < sh.Lq ? delta[si] : 0.f;                                                \
        }                                                                                        \
    } while (0)
    DKV_LOAD(0);
    for (int q0 = 0; q0 < sh.Lq; q0 += TILE) {
        __syncthreads();
        stage_store<T, D>(Qs, rq);
        stage_store<T, D>(dOs, rdo);
        if (threadIdx.x < TILE) {
            lse_s[threadIdx.x] = r_lse;
            del_s[threadIdx.x] = r_del;
        }
        __syncthreads();
        if (q0 + TILE < sh.Lq) DKV_LOAD(q0 + TILE);
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {  // half a tile at a time (see mma_tr_pair)
            f32x4 S[2][2], dP[2][2];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int t = 2 * hh + tt;
                const RowFrag<T, D> qf = load_rowfrag<T, D>(Qs + (16 * t + i) * LS, g);
                const RowFrag<T, D> dof = load_rowfrag<T, D>(dOs + (16 * t + i) * LS, g);
                const f32x4 ndl4 = -*reinterpret_cast<const f32x4*>(&del_s[16 * t + 4 * g]);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {  // C/D: col = own key (lane & 15), row = query 16t + 4g + reg
                    S[mt][tt] = mma_rows<T, D>(qf, kf[mt], (f32x4){0.f, 0.f, 0.f, 0.f});
                    dP[mt][tt] = mma_rows<T, D>(dof, vf[mt], ndl4);  // starts at -delta: dP - delta comes out of the MFMA
                }
            }
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const f32x4 lq4 = *reinterpret_cast<const f32x4*>(&lse_s[16 * (2 * hh + tt) + 4 * g]);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        float p = fast_exp2(fmaf(S[mt][tt][r], sl2, -lq4[r]));
                        if constexpr (!PLAIN) p = kvalid[mt] ? p : 0.f;
                        S[mt][tt][r] = p;                            // P
                        dP[mt][tt][r] = p * dP[mt][tt][r];  // dS / scale
                    }
            }
            mma_tr_pair<T, D>(adv[0], adv[1], dOs, hh, S[0][0], S[0][1], S[1][0], S[1][1], lane);
            mma_tr_pair<T, D>(adk[0], adk[1], Qs, hh, dP[0][0], dP[0][1], dP[1][0], dP[1][1], lane);
        }
    }
#undef DKV_LOAD
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
        if (krow[mt] < sh.Lk) {
            const int64_t off = av.ko + (int64_t)krow[mt] * sh.kv_ls + (int64_t)h * D;
            const int64_t ro = (int64_t)krow[mt] * (D / 2);
            store_row_rot<T, D>(dk + off, adk[mt], sh.scale, sh.rot_cos ? sh.rot_cos + ro : nullptr, sh.rot_sin + ro, g);
#pragma unroll
            for (int dt = 0; dt < D / 16; ++dt) store4T<T>(dv + off + dt * 16 + 4 * g, adv[mt][dt]);
        }
}

// ============================================================================================ tiny sequences
// The dual-track model's self-attention as written (quirk Q1: nn.MultiheadAttention batch_first applied to [L, B, D], so
// the "sequence" is the batch index) attends over B = 4 keys in L * H = 16384 independent problems: a 128-query x
// 64-key tile per (batch, head) block is almost all padding there.  For Lq, Lk <= SMALL_L and d = 32 one THREAD owns one
// (batch, head, row): the row stays in registers, the other side's <= 16 rows stream through as 128-byte loads (adjacent
// threads = adjacent heads, then adjacent batches: consecutive addresses), all arithmetic fp32 on the vector ALU in the
// tile kernels' formulas (exp2 against the row maximum, LSE = (m + log2 l) ln 2, the scale applied once to dQ / dK).
constexpr int SMALL_L = 16;
template <typename T> __device__ __forceinline__ void small_load32(const T* p, float* r);
template <> __device__ __forceinline__ void small_load32<float>(const float* p, float* r) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float4 x = reinterpret_cast<const float4*>(p)[c];
        r[4 * c] = x.x; r[4 * c + 1] = x.y; r[4 * c + 2] = x.z; r[4 * c + 3] = x.w;
    }
}
template <> __device__ __forceinline__ void small_load32<bf16_t>(const bf16_t* p, float* r) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const bf16x8 x = reinterpret_cast<const bf16x8*>(p)[c];
#pragma unroll
        for (int e = 0; e < 8; ++e) r[8 * c + e] = (float)x[e];
    }
}
template <typename T> __device__ __forceinline__ void small_store32(T* p, const float* r, float mul);
template <> __device__ __forceinline__ void small_store32<float>(float* p, const float* r, float mul) {
#pragma unroll
    for (int c = 0; c < 8; ++c)
        reinterpret_cast<float4*>(p)[c] = make_float4(r[4 * c] * mul, r[4 * c + 1] * mul, r[4 * c + 2] * mul, r[4 * c + 3] * mul);
}
template <> __device__ __forceinline__ void small_store32<bf16_t>(bf16_t* p, const float* r, float mul) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        bf16x8 x;
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = (bf16_t)(r[8 * c + e] * mul);
        reinterpret_cast<bf16x8*>(p)[c] = x;
    }
}
__device__ __forceinline__ float small_dot32(const float* a, const float* b) {
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 32; ++e) s = fmaf(a[e], b[e], s);
    return s;
}
// thread -> (row, batch, head), heads fastest
#define SMALL_INDEX(L_)                                                         \
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;                 \
    if (idx >= (int64_t)B * sh.H * (L_)) return;                                 \
    const int h = (int)(idx % sh.H), b = (int)((idx / sh.H) % B), row = (int)(idx / ((int64_t)sh.H * B));

template <typename T>
__global__ void __launch_bounds__(256) attn_small_fwd_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                             const T* __restrict__ v, T* __restrict__ o,
                                                             float* __restrict__ lse, const uint8_t* __restrict__ mask,
                                                             AttnShape sh, int B) {
    SMALL_INDEX(sh.Lq)
    const float sl2 = sh.scale * LOG2E;
    float qr[32], s[SMALL_L];
    small_load32<T>(q + b * sh.q_bs + row * sh.q_ls + h * 32, qr);
    const T* kb = k + b * sh.kv_bs + h * 32;
    const T* vb = v + b * sh.kv_bs + h * 32;
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < SMALL_L; ++j) {
        s[j] = -INFINITY;
        if (j < sh.Lk && (!mask || mask[(int64_t)b * sh.Lk + j])) {
            float kr[32];
            small_load32<T>(kb + j * sh.kv_ls, kr);
            s[j] = small_dot32(qr, kr);
            mx = fmaxf(mx, s[j]);
        }
    }
    const float m2 = mx * sl2, ms = mx == -INFINITY ? 0.f : m2;
    float l = 0.f, acc[32];
#pragma unroll
    for (int e = 0; e < 32; ++e) acc[e] = 0.f;
#pragma unroll
    for (int j = 0; j < SMALL_L; ++j) {
        if (j < sh.Lk) {
            const float p = fast_exp2(fmaf(s[j], sl2, -ms));  // (masked: exp2(-inf) = 0)
            l += p;
            float vr[32];
            small_load32<T>(vb + j * sh.kv_ls, vr);
#pragma unroll
            for (int e = 0; e < 32; ++e) acc[e] = fmaf(p, vr[e], acc[e]);
        }
    }
    small_store32<T>(o + b * sh.o_bs + row * sh.o_ls + h * 32, acc, l > 0.f ? 1.0f / l : 0.f);
    lse[((int64_t)b * sh.H + h) * sh.Lq + row] = (m2 + log2f(l)) * LN2;
}

template <typename T>
__global__ void __launch_bounds__(256) attn_small_dq_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                            const T* __restrict__ v, const T* __restrict__ o,
                                                            const T* __restrict__ dout, const float* __restrict__ lse,
                                                            float* __restrict__ delta, const uint8_t* __restrict__ mask,
                                                            T* __restrict__ dq, AttnShape sh, int B) {
    SMALL_INDEX(sh.Lq)
    const float sl2 = sh.scale * LOG2E;
    float qr[32], dor[32], acc[32];
    small_load32<T>(q + b * sh.q_bs + row * sh.q_ls + h * 32, qr);
    small_load32<T>(dout + b * sh.o_bs + row * sh.o_ls + h * 32, dor);
    small_load32<T>(o + b * sh.o_bs + row * sh.o_ls + h * 32, acc);
    const float dl = small_dot32(acc, dor);
    const int64_t li = ((int64_t)b * sh.H + h) * sh.Lq + row;
    delta[li] = dl;
    const float l2 = lse[li] * LOG2E;
#pragma unroll
    for (int e = 0; e < 32; ++e) acc[e] = 0.f;
    const T* kb = k + b * sh.kv_bs + h * 32;
    const T* vb = v + b * sh.kv_bs + h * 32;
    for (int j = 0; j < sh.Lk; ++j) {
        if (mask && !mask[(int64_t)b * sh.Lk + j]) continue;
        float kr[32], vr[32];
        small_load32<T>(kb + j * sh.kv_ls, kr);
        small_load32<T>(vb + j * sh.kv_ls, vr);
        const float p = fast_exp2(fmaf(small_dot32(qr, kr), sl2, -l2));
        const float ds = p * (small_dot32(dor, vr) - dl);
#pragma unroll
        for (int e = 0; e < 32; ++e) acc[e] = fmaf(ds, kr[e], acc[e]);
    }
    small_store32<T>(dq + b * sh.q_bs + row * sh.q_ls + h * 32, acc, sh.scale);
}

template <typename T>
__global__ void __launch_bounds__(256) attn_small_dkv_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                             const T* __restrict__ v, const T* __restrict__ dout,
                                                             const float* __restrict__ lse,
                                                             const float* __restrict__ delta,
                                                             const uint8_t* __restrict__ mask, T* __restrict__ dk,
                                                             T* __restrict__ dv, AttnShape sh, int B) {
    SMALL_INDEX(sh.Lk)
    const float sl2 = sh.scale * LOG2E;
    float kr[32], vr[32], ak[32], av[32];
    small_load32<T>(k + b * sh.kv_bs + row * sh.kv_ls + h * 32, kr);
    small_load32<T>(v + b * sh.kv_bs + row * sh.kv_ls + h * 32, vr);
#pragma unroll
    for (int e = 0; e < 32; ++e) ak[e] = av[e] = 0.f;
    if (!mask || mask[(int64_t)b * sh.Lk + row]) {
        const T* qb = q + b * sh.q_bs + h * 32;
        const T* db = dout + b * sh.o_bs + h * 32;
        for (int i = 0; i < sh.Lq; ++i) {
            float qr[32], dor[32];
            small_load32<T>(qb + i * sh.q_ls, qr);
            small_load32<T>(db + i * sh.o_ls, dor);
            const int64_t li = ((int64_t)b * sh.H + h) * sh.Lq + i;
            const float p = fast_exp2(fmaf(small_dot32(qr, kr), sl2, -lse[li] * LOG2E));
            const float ds = p * (small_dot32(dor, vr) - delta[li]);
#pragma unroll
            for (int e = 0; e < 32; ++e) {
                av[e] = fmaf(p, dor[e], av[e]);
                ak[e] = fmaf(ds, qr[e], ak[e]);
            }
        }
    }
    small_store32<T>(dk + b * sh.kv_bs + row * sh.kv_ls + h * 32, ak, sh.scale);
    small_store32<T>(dv + b * sh.kv_bs + row * sh.kv_ls + h * 32, av, 1.0f);
}
#undef SMALL_INDEX
// (the one-thread-per-row kernels move o / dout as 16-byte vectors: their strides and bases must be 16-byte multiples —
// check_attn only asks o for 8 bytes in bf16 — else the tile kernels take the call as before)
static bool attn_small_ok(int Lq, int Lk, int d, const int* cu, int dtype, int64_t o_bs, int64_t o_ls, const void* o,
                          const void* dout) {
    static const bool off = getenv("SEDD_NO_SMALL_ATTN") != nullptr;  // A/B knob
    const int al = dtype == SEDD_BF16 ? 8 : 4;
    const bool aligned = o_bs % al == 0 && o_ls % al == 0 && ((uintptr_t)o & 15) == 0 && ((uintptr_t)dout & 15) == 0;
    return !off && !cu && d == 32 && Lq <= SMALL_L && Lk <= SMALL_L && aligned;
}

// ==================================================================================================== host
static int check_attn(const char* nm, int dtype, int B, int H, int Lq, int Lk, int d, int64_t q_bs, int64_t q_ls,
                      int64_t kv_bs, int64_t kv_ls, int64_t o_bs, int64_t o_ls) {
    SEDD_CHECK_ARG(dtype == SEDD_F32 || dtype == SEDD_BF16, "%s: bad dtype", nm);
    SEDD_CHECK_ARG(B >= 0 && H > 0 && Lq > 0 && Lk > 0, "%s: bad sizes B=%d H=%d Lq=%d Lk=%d", nm, B, H, Lq, Lk);
    const int al = dtype == SEDD_BF16 ? 8 : 4;
    if (!(d == 32 || d == 64 || d == 96 || d == 128) || q_bs % al || q_ls % al || kv_bs % al || kv_ls % al || o_bs % 4 ||
        o_ls % 4) {
        sedd_set_error("%s: head dim %d / strides unsupported (d in {32,64,96,128}, strides multiples of %d)", nm, d, al);
        return SEDD_ERR_UNSUPPORTED_SHAPE;
    }
    SEDD_CHECK_ARG((int64_t)B * H * ((Lq > Lk ? Lq : Lk) / OWN_PER_BLOCK + 1) < (1LL << 31), "%s: too many blocks", nm);
    return SEDD_OK;
}

#define DISPATCH_ATTN(dtype, d, ...)                                                          \
    do {                                                                                      \
        if ((dtype) == SEDD_BF16) {                                                           \
            using T = bf16_t;                                                                 \
            if ((d) == 32) { constexpr int D = 32; __VA_ARGS__; }                             \
            else if ((d) == 64) { constexpr int D = 64; __VA_ARGS__; }                        \
            else if ((d) == 96) { constexpr int D = 96; __VA_ARGS__; }                        \
            else { constexpr int D = 128; __VA_ARGS__; }                                      \
        } else {                                                                              \
            using T = float;                                                                  \
            if ((d) == 32) { constexpr int D = 32; __VA_ARGS__; }                             \
            else if ((d) == 64) { constexpr int D = 64; __VA_ARGS__; }                        \
            else if ((d) == 96) { constexpr int D = 96; __VA_ARGS__; }                        \
            else { constexpr int D = 128; __VA_ARGS__; }                                      \
        }                                                                                     \
    } while (0)

static int attn_fwd_impl(const void* q, const void* k, const void* v, void* o, float* lse, const uint8_t* key_mask,
                         int dtype, int B, int H, int Lq, int Lk, int d, int64_t q_bs, int64_t q_ls, int64_t kv_bs,
                         int64_t kv_ls, int64_t o_bs, int64_t o_ls, const int* cu, int total, sedd_stream_t stream) {
    int rc = check_attn("attn_fwd", dtype, B, H, Lq, Lk, d, q_bs, q_ls, kv_bs, kv_ls, o_bs, o_ls);
    if (rc) return rc;
    if (B == 0) return SEDD_OK;
    SEDD_CHECK_ARG(q && k && v && o && lse, "attn_fwd: null pointer");
    AttnShape sh{H, Lq, Lk, q_bs, q_ls, kv_bs, kv_ls, o_bs, o_ls, 1.0f / sqrtf((float)d), nullptr, nullptr, cu, total};
    if (attn_small_ok(Lq, Lk, d, cu, dtype, o_bs, o_ls, o, nullptr)) {
        const dim3 gs((unsigned)(((int64_t)B * H * Lq + 255) / 256));
        if (dtype == SEDD_BF16)
            hipLaunchKernelGGL(attn_small_fwd_kernel<bf16_t>, gs, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)q,
                               (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, lse, key_mask, sh, B);
        else
            hipLaunchKernelGGL(attn_small_fwd_kernel<float>, gs, dim3(256), 0, (hipStream_t)stream, (const float*)q,
                               (const float*)k, (const float*)v, (float*)o, lse, key_mask, sh, B);
        SEDD_LAUNCH_CHECK();
        return SEDD_OK;
    }
    const dim3 grid((unsigned)((int64_t)((Lq + OWN_PER_BLOCK - 1) / OWN_PER_BLOCK) * B * H));
    const bool plain = !key_mask && !cu && Lk % TILE == 0;  // (bf16 only: the fp32 instantiations are the parity path)
#define FWD_LAUNCH(PL) hipLaunchKernelGGL((attn_fwd_kernel<T, D, PL>), grid, dim3(256), 0, (hipStream_t)stream,       \
                                          (const T*)q, (const T*)k, (const T*)v, (T*)o, lse, key_mask, sh)
    DISPATCH_ATTN(dtype, d, {
        if constexpr (sizeof(T) == 2) {
            if (plain) FWD_LAUNCH(true);
            else FWD_LAUNCH(false);
        } else FWD_LAUNCH(false);
    });
#undef FWD_LAUNCH
    SEDD_LAUNCH_CHECK();
    return SEDD_OK;
}

static int attn_bwd_impl(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                         const uint8_t* key_mask, void* dq, void* dk, void* dv, float* delta, int dtype, int B, int H,
                         int Lq, int Lk, int d, int64_t q_bs, int64_t q_ls, int64_t kv_bs, int64_t kv_ls, int64_t o_bs,
                         int64_t o_ls, const float* rot_cos, const float* rot_sin, const int* cu, int total,
                         sedd_stream_t stream) {
    int rc = check_attn("attn_bwd", dtype, B, H, Lq, Lk, d, q_bs, q_ls, kv_bs, kv_ls, o_bs, o_ls);
    if (rc) return rc;
    if (B == 0) return SEDD_OK;
    SEDD_CHECK_ARG(q && k && v && o && dout && lse && dq && dk && dv && delta, "attn_bwd: null pointer");
    SEDD_CHECK_ARG((rot_cos == nullptr) == (rot_sin == nullptr), "attn_bwd: rot_cos and rot_sin go together");
    SEDD_CHECK_ARG(!rot_cos || Lq == Lk, "attn_bwd: the fused rotary gradient is for self-attention (Lq == Lk)");
    AttnShape sh{H, Lq, Lk, q_bs, q_ls, kv_bs, kv_ls, o_bs, o_ls, 1.0f / sqrtf((float)d), rot_cos, rot_sin, cu, total};
    hipStream_t st = (hipStream_t)stream;
    if (attn_small_ok(Lq, Lk, d, cu, dtype, o_bs, o_ls, o, dout) && !rot_cos) {
        const dim3 gsq((unsigned)(((int64_t)B * H * Lq + 255) / 256)), gsk((unsigned)(((int64_t)B * H * Lk + 255) / 256));
#define SMALL_BWD(T)                                                                                                   \
        hipLaunchKernelGGL(attn_small_dq_kernel<T>, gsq, dim3(256), 0, st, (const T*)q, (const T*)k, (const T*)v,      \
                           (const T*)o, (const T*)dout, lse, delta, key_mask, (T*)dq, sh, B);                          \
        hipLaunchKernelGGL(attn_small_dkv_kernel<T>, gsk, dim3(256), 0, st, (const T*)q, (const T*)k, (const T*)v,     \
                           (const T*)dout, lse, delta, key_mask, (T*)dk, (T*)dv, sh, B)
        if (dtype == SEDD_BF16) { SMALL_BWD(bf16_t); } else { SMALL_BWD(float); }
#undef SMALL_BWD
        SEDD_LAUNCH_CHECK();
        return SEDD_OK;
    }
    const dim3 gq((unsigned)((int64_t)((Lq + OWN_PER_BLOCK - 1) / OWN_PER_BLOCK) * B * H)),
        gk((unsigned)((int64_t)((Lk + OWN_PER_BLOCK - 1) / OWN_PER_BLOCK) * B * H));
    const bool plain_q = !key_mask && !cu && Lk % TILE == 0, plain_k = !key_mask && !cu && Lk % OWN_PER_BLOCK == 0;
#define DQ_LAUNCH(PL) hipLaunchKernelGGL((attn_bwd_dq_kernel<T, D, PL>), gq, dim3(256), 0, st, (const T*)q,           \
                                         (const T*)k, (const T*)v, (const T*)o, (const T*)dout, lse, delta, key_mask,  \
                                         (T*)dq, sh)
#define DKV_LAUNCH(PL) hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, D, PL>), gk, dim3(256), 0, st, (const T*)q,         \
                                          (const T*)k, (const T*)v, (const T*)dout, lse, delta, key_mask, (T*)dk,     \
                                          (T*)dv, sh)
    DISPATCH_ATTN(dtype, d, {
        if constexpr (sizeof(T) == 2) {
            if (plain_q) DQ_LAUNCH(true);
            else DQ_LAUNCH(false);
            if (plain_k) DKV_LAUNCH(true);
            else DKV_LAUNCH(false);
        } else {
            DQ_LAUNCH(false);
            DKV_LAUNCH(false);
        }
    });
#undef DQ_LAUNCH
#undef DKV_LAUNCH
    SEDD_LAUNCH_CHECK();
    return SEDD_OK;
}

extern "C" int sedd_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse,
                             const uint8_t* key_mask, int dtype, int B, int H, int Lq, int Lk, int d, int64_t q_bs,
                             int64_t q_ls, int64_t kv_bs, int64_t kv_ls, int64_t o_bs, int64_t o_ls,
                             sedd_stream_t stream) {
    return attn_fwd_impl(q, k, v, o, lse, key_mask, dtype, B, H, Lq, Lk, d, q_bs, q_ls, kv_bs, kv_ls, o_bs, o_ls, nullptr, 0,
                         stream);
}

extern "C" int sedd_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout,
                             const float* lse, const uint8_t* key_mask, void* dq, void* dk, void* dv, float* delta,
                             int dtype, int B, int H, int Lq, int Lk, int d, int64_t q_bs, int64_t q_ls, int64_t kv_bs,
                             int64_t kv_ls, int64_t o_bs, int64_t o_ls, const float* rot_cos, const float* rot_sin,
                             sedd_stream_t stream) {
    return attn_bwd_impl(q, k, v, o, dout, lse, key_mask, dq, dk, dv, delta, dtype, B, H, Lq, Lk, d, q_bs, q_ls, kv_bs, kv_ls,
                         o_bs, o_ls, rot_cos, rot_sin, nullptr, 0, stream);
}

// ---- variable-length packing: flash_attn_varlen_qkvpacked_func's general form (transformer.py:173-181)
static int check_varlen(const char* nm, const int* cu, int nseq, int total, int max_len) {
    SEDD_CHECK_ARG(nseq >= 0 && total >= 0 && max_len > 0, "%s: bad sizes nseq=%d total=%d max_len=%d", nm, nseq, total,
                   max_len);
    SEDD_CHECK_ARG(nseq == 0 || cu, "%s: cu_seqlens is null", nm);
    return SEDD_OK;
}
extern "C" int sedd_attn_varlen_fwd(const void* q, const void* k, const void* v, void* o, float* lse,
                                    const int32_t* cu_seqlens, int nseq, int total, int max_len, int dtype, int H, int d,
                                    int64_t q_ls, int64_t kv_ls, int64_t o_ls, sedd_stream_t stream) {
    int rc = check_varlen("attn_varlen_fwd", cu_seqlens, nseq, total, max_len);
    if (rc) return rc;
    return attn_fwd_impl(q, k, v, o, lse, nullptr, dtype, nseq, H, max_len, max_len, d, 0, q_ls, 0, kv_ls, 0, o_ls,
                         cu_seqlens, total, stream);
}
extern "C" int sedd_attn_varlen_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout,
                                    const float* lse, void* dq, void* dk, void* dv, float* delta,
                                    const int32_t* cu_seqlens, int nseq, int total, int max_len, int dtype, int H, int d,
                                    int64_t q_ls, int64_t kv_ls, int64_t o_ls, sedd_stream_t stream) {
    int rc = check_varlen("attn_varlen_bwd", cu_seqlens, nseq, total, max_len);
    if (rc) return rc;
    return attn_bwd_impl(q, k, v, o, dout, lse, nullptr, dq, dk, dv, delta, dtype, nseq, H, max_len, max_len, d, 0, q_ls, 0,
                         kv_ls, 0, o_ls, nullptr, nullptr, cu_seqlens, total, stream);
}
