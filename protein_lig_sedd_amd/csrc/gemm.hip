// MFMA GEMM for every nn.Linear on the hot path (transformer.py:125-134,140,225,229;
// transformers_protlig_cross.py:250-268): C[M,N] = A[M,K] * B[N,K]^T (+bias)(GELU-tanh), both operands
// K-contiguous.  Forward uses it directly (A = activations, B = weight [out,in]); the backward data/weight
// gradients are the same kernel on re-laid-out operands (sedd_transpose), weight gradients accumulate in fp32.
//
// bf16 fast path (K % 64 == 0): v_mfma_f32_16x16x32_bf16, 128x128x64 block tile, 4 waves as 2x2 (64x64 per wave =
//   4x4 MFMA tiles).  Operand tiles go global -> LDS directly (global_load_lds_dwordx4, no VGPR round trip) into a
//   double buffer; the next tile's 8 loads per wave are issued before the current tile's 32 MFMAs, one barrier per
//   K tile.  LDS rows are 128 B; the 16-byte chunk index is XOR-swizzled with (row & 7) — applied on the per-lane
//   SOURCE address (the LDS-DMA destination is lane-linear) and on the ds_read_b128 fragment reads.
//   The MFMA is issued with the operands swapped (D^T = B_frag x A_frag) so a lane holds 4 CONSECUTIVE COLUMNS of a C
//   row: 8-byte (bf16) / 16-byte (fp32) stores, float4 bias loads.  Blocks are ordered XCD-aware (neighbouring tiles
//   share an A panel in one XCD's L2).  Accumulating fp32 outputs (weight gradients: few tiles, huge K) are split
//   along K across blockIdx.y and combined with fp32 atomics.
// bf16 generic path: any K % 8 == 0, register-staged, same tile.
// fp32 path: exact-fp32 v_mfma_f32_32x32x2_f32 (k-ordered fmaf chain), 128x128x16 tile, for the fp32 parity
//   configuration and the fp32-autocast cross model.
// MFMA-bound: 2*M*N*K flop per call.
#include "common.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;

#define BM 128
#define BN 128
#define BK16 64

__device__ __forceinline__ float gelu_tanh_f(float u) {
    const float t = tanhf(0.7978845608028654f * (u + 0.044715f * u * u * u));
    return 0.5f * u * (1.0f + t);
}

// tanh on the hardware exp / rcp units: 1 - 2 / (1 + e^{2x}); absolute error ~1e-7, far inside bf16 rounding.  Used by
// the bf16 epilogues only (libm's tanhf costs ~40 VALU instructions per element, which a one-block-per-CU GEMM
// epilogue cannot hide); fp32 outputs keep tanhf.
__device__ __forceinline__ float tanh_fast(float x) {
    const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + e);
}
template <typename TO> __device__ __forceinline__ float gelu_tanh_t(float u) {
    const float a = 0.7978845608028654f * (u + 0.044715f * u * u * u);
    const float t = sizeof(TO) == 2 ? tanh_fast(a) : tanhf(a);
    return 0.5f * u * (1.0f + t);
}
template <typename TO> __device__ __forceinline__ float gelu_tanh_grad_t(float u) {
    const float a = 0.7978845608028654f * (u + 0.044715f * u * u * u);
    const float t = sizeof(TO) == 2 ? tanh_fast(a) : tanhf(a);
    return 0.5f * (1.0f + t) + 0.5f * u * (1.0f - t * t) * 0.7978845608028654f * (1.0f + 3.0f * 0.044715f * u * u);
}
__device__ __forceinline__ float gelu_tanh_grad_f(float u) {
    const float t = tanhf(0.7978845608028654f * (u + 0.044715f * u * u * u));
    return 0.5f * (1.0f + t) + 0.5f * u * (1.0f - t * t) * 0.7978845608028654f * (1.0f + 3.0f * 0.044715f * u * u);
}
// Four elements at a time in the bf16 epilogues, written on ext-vector types so the polynomial parts compile to packed
// fp32 ops (v_pk_mul/fma/add_f32, two elements per instruction): with s = sigmoid(2a) = 1 / (1 + 2^(u (A + B u^2))),
// gelu_tanh(u) = u s and gelu_tanh'(u) = s (1 + u (1 - s)(C + D u^2)) — 5 resp. 9 plain operations + exp2 + rcp per
// element instead of ~12 / ~22 through tanh (the fused dGELU epilogue is VALU-bound: every wave of the block is in it at
// the same time, nothing overlaps it).  fp32 outputs keep the exact tanhf forms.
template <typename TO> __device__ __forceinline__ f32x4 gelu4_t(f32x4 u) {
    if constexpr (sizeof(TO) == 2) {
        const f32x4 z = u * (-2.3022082f + -0.10294324f * (u * u));
        f32x4 s;
#pragma unroll
        for (int r = 0; r < 4; ++r) s[r] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z[r]));
        return u * s;
    } else {
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = gelu_tanh_t<TO>(u[r]);
        return o;
    }
}
template <typename TO> __device__ __forceinline__ f32x4 gelu_grad4_t(f32x4 u) {
    if constexpr (sizeof(TO) == 2) {
        const f32x4 u2 = u * u;
        const f32x4 z = u * (-2.3022082f + -0.10294324f * u2);
        f32x4 s;
#pragma unroll
        for (int r = 0; r < 4; ++r) s[r] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z[r]));
        const f32x4 w = u * (1.5957691216f + 0.21406445f * u2);
        return s + s * (w * (1.0f - s));
    } else {
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = gelu_tanh_grad_t<TO>(u[r]);
        return o;
    }
}
// gelu_tanh and its derivative from ONE sigmoid (the forward's SEDD_EPI_BIAS_GELU_GRAD epilogue hands the derivative to
// the backward, whose fused data-gradient epilogue is then one multiply: EPI_DMUL)
template <typename TO> __device__ __forceinline__ void gelu_both4_t(f32x4 u, f32x4& g, f32x4& gp) {
    if constexpr (sizeof(TO) == 2) {
        const f32x4 u2 = u * u;
        const f32x4 z = u * (-2.3022082f + -0.10294324f * u2);
        f32x4 s;
#pragma unroll
        for (int r = 0; r < 4; ++r) s[r] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z[r]));
        const f32x4 w = u * (1.5957691216f + 0.21406445f * u2);
        g = u * s;
        gp = s + s * (w * (1.0f - s));
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) { g[r] = gelu_tanh_t<TO>(u[r]); gp[r] = gelu_tanh_grad_t<TO>(u[r]); }
    }
}
// internal epilogues (sedd_gemm_nt_dgelu / _dmul): C = (A B^T) * gelu_tanh'(pre) resp. (A B^T) * pre; `pre` is an INPUT,
// column sums out
#define EPI_DGELU 16
#define EPI_DMUL 17

// bijective XCD-aware remap (8 XCDs): blocks that share an XCD get a contiguous range of tiles
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// Persistent tile t (t % 8 = the XCD that runs it under round-robin dispatch; speed only, never correctness) -> tile
// coordinates.  csplit == 1: every XCD owns a contiguous run of row panels, all column tiles of a panel back to back
// (the A row panel is fetched into ONE L2).  csplit = S > 1: the XCDs form an (8 / S) x S grid, XCD (xr, xc) owns the
// rows of group xr and the column tiles of slice xc, column index fastest: its slice of the weight (B / S bytes, chosen
// to fit the 4 MiB L2 beside the streams) stays resident for the whole launch instead of being re-fetched from the
// fabric every round, at the price of S XCDs reading each A panel (see choose_csplit).
__device__ __forceinline__ void big_tile_map(int tile, int ntiles, int ntn, int csplit, bool linear, int& tm, int& tn) {
    if (linear) { tm = tile / ntn; tn = tile % ntn; return; }
    if (csplit <= 1) {
        const int wg = xcd_remap(tile, ntiles);
        tm = wg / ntn; tn = wg % ntn;
        return;
    }
    const int xcd = tile & 7, idx = tile >> 3;
    const int cn = ntn / csplit, rn = (ntiles / ntn) / (8 / csplit);
    const int xr = xcd / csplit, xc = xcd % csplit;
    tm = xr * rn + idx / cn;
    tn = xc * cn + idx % cn;
}

template <typename TO> __device__ __forceinline__ void store_row4(TO* p, f32x4 v);
template <> __device__ __forceinline__ void store_row4<float>(float* p, f32x4 v) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <> __device__ __forceinline__ void store_row4<bf16_t>(bf16_t* p, f32x4 v) {
    union { bf16_t h[4]; uint2 u; } c;
    c.h[0] = (bf16_t)v[0]; c.h[1] = (bf16_t)v[1]; c.h[2] = (bf16_t)v[2]; c.h[3] = (bf16_t)v[3];
    *reinterpret_cast<uint2*>(p) = c.u;
}

// epilogue for 4 consecutive columns [col, col+4) of one C row held in one accumulator
// GGM = false: instantiations that never see SEDD_EPI_BIAS_GELU_GRAD (the register-bound big-tile kernels, which have a
// kernel variant of their own for it) do not carry its code
template <typename TO, bool GGM = true>
__device__ __forceinline__ void epilogue4(TO* __restrict__ C, int64_t ldc, TO* __restrict__ pre,
                                          const float* __restrict__ bias, int epi, int accumulate, int atomic,
                                          int64_t row, int64_t col, int64_t M, int64_t N, f32x4 v, bool vec_ok) {
    if (row >= M || col >= N) return;
    const bool full = vec_ok && col + 4 <= N;
    if (epi != SEDD_EPI_NONE) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += (col + r < N) ? bias[col + r] : 0.f;
    }
    TO* p = C + row * ldc + col;
    if (epi == SEDD_EPI_BIAS_GELU || (GGM && epi == SEDD_EPI_BIAS_GELU_GRAD)) {
        // the reference rounds the Linear output to the activation dtype before GELU
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = to_f32<TO>(from_f32<TO>(v[r]));
        f32x4 keep = v;  // what `pre` receives: the pre-activation, or (GELU_GRAD) gelu'(pre-activation)
        if (GGM && epi == SEDD_EPI_BIAS_GELU_GRAD) gelu_both4_t<TO>(v, v, keep);
        else v = gelu4_t<TO>(v);
        if (pre) {
            TO* pp = pre + row * ldc + col;
            if (full) store_row4<TO>(pp, keep);
            else
                for (int r = 0; r < 4 && col + r < N; ++r) pp[r] = from_f32<TO>(keep[r]);
        }
    }
    if (accumulate) {
        if constexpr (sizeof(TO) == 4) {
            if (atomic) {
                for (int r = 0; r < 4 && col + r < N; ++r) atomicAdd(reinterpret_cast<float*>(p) + r, v[r]);
                return;
            }
        }
        for (int r = 0; r < 4 && col + r < N; ++r) v[r] += to_f32<TO>(p[r]);
    }
    if (full) store_row4<TO>(p, v);
    else
        for (int r = 0; r < 4 && col + r < N; ++r) p[r] = from_f32<TO>(v[r]);
}

// 32 MFMAs on one staged 128x64 A tile and 128x64 B tile (operands swapped: acc[m][n] = C^T fragment)
__device__ __forceinline__ void mma_tile(const uint4* __restrict__ sA, const uint4* __restrict__ sB, int wr, int wc,
                                         int lane, f32x4 (&acc)[4][4]) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        bf16x8 a[4], b[4];
        const int kc = ks * 4 + (lane >> 4);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int row = wr * 64 + m * 16 + (lane & 15);
            a[m] = *reinterpret_cast<const bf16x8*>(&sA[row * 8 + (kc ^ (row & 7))]);
        }
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int row = wc * 64 + n * 16 + (lane & 15);
            b[n] = *reinterpret_cast<const bf16x8*>(&sB[row * 8 + (kc ^ (row & 7))]);
        }
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n)
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[n], a[m], acc[m][n], 0, 0, 0);
    }
}

template <typename TO>
__device__ __forceinline__ void store_tile(TO* __restrict__ C, int64_t ldc, TO* __restrict__ pre,
                                           const float* __restrict__ bias, int epi, int accumulate, int atomic,
                                           int64_t m0, int64_t n0, int wr, int wc, int lane, int64_t M, int64_t N,
                                           const f32x4 (&acc)[4][4]) {
    // swapped-operand C/D map: lane & 15 = row of C inside the 16x16 tile, 4*(lane >> 4) + reg = column
    const bool vec_ok = (ldc % 4 == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0) &&
                        (!pre || (reinterpret_cast<uintptr_t>(pre) & 15) == 0);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
            epilogue4<TO>(C, ldc, pre, bias, epi, accumulate, atomic, m0 + wr * 64 + m * 16 + (lane & 15),
                          n0 + wc * 64 + n * 16 + 4 * (lane >> 4), M, N, acc[m][n], vec_ok);
}

// ------------------------------------------------------------------------------------- bf16, LDS-DMA fast path
template <typename TO>
__global__ void __launch_bounds__(256) gemm_bf16_nt_glds_kernel(const bf16_t* __restrict__ A, int64_t lda,
                                                                const bf16_t* __restrict__ B, int64_t ldb,
                                                                TO* __restrict__ C, int64_t ldc,
                                                                const float* __restrict__ bias, int epi,
                                                                TO* __restrict__ pre, int accumulate, int64_t M,
                                                                int64_t N, int64_t Kper, int ntn,
                                                                float* __restrict__ slabs = nullptr) {
    __shared__ __attribute__((aligned(16))) uint4 smem[2][2][BM * 8];  // [buffer][A|B][row*8 + swizzled chunk] 64 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int64_t m0 = (int64_t)(wg / ntn) * BM, n0 = (int64_t)(wg % ntn) * BN;
    const int64_t kbeg = (int64_t)blockIdx.y * Kper;

    f32x4 acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // this lane's 4 source rows of A and of B (8 rows x 128 B per wave-instruction, 4 instructions per operand)
    const bf16_t* srcA[4];
    const bf16_t* srcB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wave * 4 + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ (r & 7);  // LDS slot (r, lane & 7) holds global chunk c (swizzle on the source)
        int64_t ar = m0 + r, br = n0 + r;
        ar = ar < M ? ar : M - 1;
        br = br < N ? br : N - 1;
        srcA[i] = A + ar * lda + kbeg + c * 8;
        srcB[i] = B + br * ldb + kbeg + c * 8;
    }
#define STAGE(buf, koff)                                                                                             \
    do {                                                                                                             \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                              \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcA[i] + (koff)),      \
                                             (__attribute__((address_space(3))) void*)&smem[buf][0][(wave * 4 + i) * 64], \
                                             16, 0, 0);                                                              \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcB[i] + (koff)),      \
                                             (__attribute__((address_space(3))) void*)&smem[buf][1][(wave * 4 + i) * 64], \
                                             16, 0, 0);                                                              \
        }                                                                                                            \
    } while (0)

    const int nkt = (int)(Kper / BK16);
    STAGE(0, 0);
    __syncthreads();  // (the compiler drains vmcnt before the barrier while an LDS-DMA is in flight)
    for (int kt = 0; kt < nkt; kt += 2) {
        if (kt + 1 < nkt) STAGE(1, (int64_t)(kt + 1) * BK16);
        mma_tile(smem[0][0], smem[0][1], wr, wc, lane, acc);
        __syncthreads();
        if (kt + 1 < nkt) {
            if (kt + 2 < nkt) STAGE(0, (int64_t)(kt + 2) * BK16);
            mma_tile(smem[1][0], smem[1][1], wr, wc, lane, acc);
            __syncthreads();
        }
    }
#undef STAGE
    if constexpr (sizeof(TO) == 4) {
        if (slabs) {  // split K with a workspace: this slice's partial tile, plain stores; splitk_fold_kernel adds the slices
            store_tile<float>(slabs + (int64_t)blockIdx.y * M * N, N, nullptr, nullptr, SEDD_EPI_NONE, 0, 0, m0, n0, wr,
                              wc, lane, M, N, acc);
            return;
        }
    }
    store_tile<TO>(C, ldc, pre, bias, epi, accumulate, gridDim.y > 1, m0, n0, wr, wc, lane, M, N, acc);
}
// C[m][n] += slab[0][m][n] + slab[1][m][n] + ... (slice order): the deterministic form of the split-K above
__global__ void __launch_bounds__(256) splitk_fold_kernel(const float* __restrict__ slabs, int nslices, int64_t M,
                                                          int64_t N, float* __restrict__ C, int64_t ldc,
                                                          int accumulate = 1, const float* __restrict__ bias = nullptr) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= M * N) return;
    float s = 0.f;
    for (int k = 0; k < nslices; ++k) s += slabs[(int64_t)k * M * N + e];
    if (bias) s += bias[e % N];
    float* c = C + (e / N) * ldc + e % N;
    *c = accumulate ? *c + s : s;
}


// ----------------------------------------------------------------- bf16, big-tile multi-stage LDS-DMA path
// 256 x BN x 32 block tile (BN = 256 or 128), 8 waves (2 per SIMD), NST-deep ring of LDS stages filled by LDS-DMA.
// A 128x128 tile moves 1 byte per 64 flop from L2 into LDS and can keep only one 64-wide K tile in flight per block,
// so every K step waits a full memory latency; the 256-wide tile halves the bytes per flop and the ring keeps NST-1
// K tiles (>= 1 us of MFMA work) in flight.  The ring needs counted waits — s_waitcnt vmcnt(N) leaves the N youngest
// DMA loads in flight — and a bare s_barrier (a __syncthreads() makes the compiler drain every outstanding LDS-DMA);
// the K loop is unrolled by NST so every stage is a compile-time LDS offset.
// LDS rows are 64 B (4 x 16-byte chunks); chunk index XOR ((row >> 1) & 3) makes the ds_read_b128 fragment reads
// conflict-free (swizzle applied on the DMA source address and on the reads).
#define BIG_BM 256
#define BIG_BK 32

// Output staging for the big-tile kernel.  In the MFMA C layout a lane owns 4 consecutive columns of one row, so a
// direct store writes 32-byte (bf16) row segments and the write path runs at ~1.2 TB/s.  Each wave instead parks 16
// rows x 64 columns at a time in a private LDS region outside the operand ring (row pitch padded by 16 B:
// conflict-free writes, 2-way reads) and stores them back out with 16 bytes per lane, 128 contiguous bytes per row.
// (fp32 outputs already write 64-byte segments from the accumulators and stay direct.)
template <typename TO> struct StageCfg {
    static constexpr int RPP = 16;                               // rows per pass (one MFMA fragment row block)
    static constexpr int PITCH = sizeof(TO) == 2 ? 144 : 272;    // bytes
    static constexpr int REGION = RPP * PITCH;                   // per-wave LDS bytes
    static constexpr int LPR = 64 * (int)sizeof(TO) / 16;        // lanes per 64-column row (8 / 16)
    static constexpr int EPC = 16 / (int)sizeof(TO);             // elements per 16-byte chunk
};
template <typename TO>
__device__ __forceinline__ void stage_put4(unsigned char* reg, int row_l, int col_l, f32x4 v) {
    store_row4<TO>(reinterpret_cast<TO*>(reg + row_l * StageCfg<TO>::PITCH) + col_l, v);
}
// flush one pass: rows [row_g, row_g + RPP) x columns [col_g, col_g + 64) of C; accumulate adds the old C (fp32/bf16)
// (Round 2 tried to take the per-chunk 64-bit row * ldc multiplies and bound compares out of the pass loops — a per-tile
// lane context with wave-uniform pass offsets, then with running pointers: the first form had the compiler hoist a
// (pass, group) table of 64-bit offsets into SGPRs and spill 136-152 of them to VGPR lanes, the second spilled 14-82
// VGPRs in the 256-wide kernels, which sit at the 256-register limit; both measured slower — fc1 epilogue 10.4 -> 12.6 us
// per tile, dGELU 19.4 -> 24 us.  tools/gemm_phase.py has the per-tile phase times: plain bf16 epilogue 4.0 us = the
// CU's store rate (128 KB at ~16 B/clk), GELU + pre 10.4 us, dGELU 19.4 us, rotary 5.3 us per 256 x 128 tile.
// Also tried, for the 256 x 128 plain / bias kernels: the finished tile parked as packed bf16 in 32 registers and sent out
// one 16-row pass per ring turn from inside the NEXT tile's K loop (LDS write, read-back and stores in three different
// K steps, vmcnt waits widened by the two stores in flight; 233 VGPRs, no spills, results bit-identical): -5 % at
// N = 768 for K = 768 and K = 3072 alike — the block-wide barrier after every K step turns whatever one wave adds
// to a step into a delay for all eight, and the added branches cost the loop more than the 2 us epilogue it hides.)
template <typename TO>
__device__ __forceinline__ void stage_flush(const unsigned char* reg, TO* __restrict__ C, int64_t ldc, int64_t row_g,
                                            int64_t col_g, int64_t M, int64_t N, int lane, int accumulate) {
    using SC = StageCfg<TO>;
    constexpr int RPI = 64 / SC::LPR;  // rows per wave-instruction
    const int rsub = lane / SC::LPR, chunk = lane % SC::LPR;
#pragma unroll
    for (int i = 0; i < SC::RPP / RPI; ++i) {
        const int row_l = i * RPI + rsub;
        const int64_t gr = row_g + row_l, gc = col_g + chunk * SC::EPC;
        if (gr >= M || gc >= N) continue;
        union { uint4 u; TO e[SC::EPC]; } v;
        v.u = *reinterpret_cast<const uint4*>(reg + row_l * SC::PITCH + chunk * 16);
        TO* p = C + gr * ldc + gc;
        if (gc + SC::EPC <= N) {
            if (accumulate) {
                union { uint4 u; TO e[SC::EPC]; } o;
                o.u = *reinterpret_cast<const uint4*>(p);
#pragma unroll
                for (int e = 0; e < SC::EPC; ++e) v.e[e] = from_f32<TO>(to_f32<TO>(v.e[e]) + to_f32<TO>(o.e[e]));
            }
            {  // non-temporal: the tile is not re-read by this kernel, and keeping it out of the L2's way leaves the
               // operand panels resident (+10-14 % on the K = 768 shapes, same-device A/B)
                typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
                const u32x4 w_ = {v.u.x, v.u.y, v.u.z, v.u.w};
#ifdef GEMM_PLAIN_STORES
                *reinterpret_cast<u32x4*>(p) = w_;
#else
                __builtin_nontemporal_store(w_, reinterpret_cast<u32x4*>(p));
#endif
            }
        } else {
            for (int e = 0; e < SC::EPC && gc + e < N; ++e)
                p[e] = accumulate ? from_f32<TO>(to_f32<TO>(v.e[e]) + to_f32<TO>(p[e])) : v.e[e];
        }
    }
}

// s_waitcnt vmcnt(N) only (gfx9 encoding: vmcnt = simm16[3:0] | simm16[15:14] << 4, expcnt [6:4] and lgkmcnt [11:8] left
// at "no wait").  The builtin, unlike inline asm, is visible to the compiler's own wait-count bookkeeping, which
// otherwise assumes the older LDS-DMA loads are still pending and drains vmcnt(0) before the next LDS read.
template <int N> __device__ __forceinline__ void vm_wait_le() {
    static_assert(N >= 0 && N < 64, "vmcnt is 6 bits");
    __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | (7 << 4) | (15 << 8));
}
// wait until at most `younger` K tiles of LPT loads each are still in flight (younger <= 4)
template <int LPT> __device__ __forceinline__ void wait_tiles(int younger) {
    switch (younger) {
        case 0: vm_wait_le<0>(); break;
        case 1: vm_wait_le<LPT>(); break;
        case 2: vm_wait_le<2 * LPT>(); break;
        case 3: vm_wait_le<3 * LPT>(); break;
        default: vm_wait_le<4 * LPT>(); break;
    }
}

// 16-byte-per-lane LDS-DMA (global -> LDS at lds_addr + 16 * lane, no VGPR round trip), issued from inline asm so
// that the ring's waits are ours alone: the compiler cannot tell the ring's stages apart and, when it sees the DMA
// through the builtin, drains vmcnt(0) before LDS reads of OTHER stages.  Unknown to its bookkeeping, these loads can
// only make the compiler's own vmcnt waits stricter (vmcnt counts in issue order), never too weak.
__device__ __forceinline__ void lds_dma16(const void* g, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_addr) : "memory", "m0");
}

// TN = false: C[M,N] = A[M,K] * B[N,K]^T, both operands K-contiguous (forward / data gradients).
// TN = true : C[M,N] = A[K,M]^T * B[K,N], both operands stored with the reduction index as the ROW (weight gradients:
//   A = dY [tokens, out], B = X [tokens, in]) — no transposed copies.  A stage holds 32 token rows of each operand as
//   they lie in memory (512-byte / 2*BN-byte rows, 16-byte chunk index XOR 2*(row & 7) against bank conflicts); the
//   k-contiguous MFMA fragments come out of LDS through the transposing read ds_read_b64_tr_b16, which hands lane
//   (g, i) the four rows 4g..4g+3 of column i: a fragment is two of them (rows 4g+j and 16+4g+j) — a permutation of the
//   32 k values that both operands share, which a sum over k allows.
// NW = 8: one 512-thread block per CU.  NW = 4 (BN_ = 128 only, experimental: SEDD_GEMM_NW4=1|2): 256-thread blocks
// with the same 128 x 64 wave tile as the 256-wide NW = 8 kernel and a 3-deep ring (72 KB), so TWO blocks share a CU
// and one block's LDS-DMA issue, barriers and epilogue run under the other block's MFMAs; the output staging region
// then aliases ring stage NST-1, which the next tile's prologue (stages 0..NST-2) leaves alone until after that tile's
// first two barriers.  Measured equal to the 8-wave kernels within noise (the 3-deep ring has one K step of
// look-ahead), so it is not the default.
// NNB = true (with TN = false): C[M,N] = A[M,K] * B[K,N] — A as in the NT form, B with the reduction index as the row
//   (a data gradient against the nn.Linear weight as stored, dX = dY W, no transposed weight copy).  B takes the TN
//   path's stage layout and transposing reads, but A's fragments hold k = 8g..8g+7 in natural order, so lane group g
//   addresses rows 8g+q and 8g+4+q instead, and the bank swizzle keys on row bits {0,1,3} (rows 8g+q, g = 0,1 of one
//   half-wave must land in 8 distinct 32-byte bank groups).
// ROT = true (sedd_gemm_nt_rotary): the output is a packed qkv [M, 3, H, 64] and the NeoX rotary rotation of
//   rotary.py:42-51 is applied to its q and k columns in the epilogue — a wave's 64 output columns are exactly one
//   head, and the pair (j, j + 32) sits in fragments n and n + 2 of the same lane.
struct RotArgs { const float* cos; const float* sin; int L; int64_t cols; };  // tables [L, 32]; cols = 2 * H * 64
#ifdef GEMM_CLOCK_STAMP
__device__ uint64_t sedd_gemm_stamps[2048];
extern "C" int sedd_debug_gemm_stamps(uint64_t* host_dst) {
    return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(sedd_gemm_stamps), sizeof(uint64_t) * 2048) == hipSuccess ? 0 : -3;
}
#endif
#ifdef GEMM_PHASE_STAMP  // diagnostic build (tools/gemm_phase.py): per block and tile round, 100 MHz stamps at loop start /
__device__ uint64_t sedd_gemm_phase[512 * 8 * 3];  // main loop end / epilogue issued
extern "C" int sedd_debug_gemm_phase(uint64_t* host_dst) {
    return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(sedd_gemm_phase), sizeof(sedd_gemm_phase)) == hipSuccess ? 0 : -3;
}
#define PHASE_STAMP(k)                                                                                               \
    do {                                                                                                             \
        if (tid == 0 && blockIdx.x < 512 && phase_round < 8)                                                         \
            sedd_gemm_phase[(blockIdx.x * 8 + phase_round) * 3 + (k)] = __builtin_amdgcn_s_memrealtime();            \
    } while (0)
#else
#define PHASE_STAMP(k) do {} while (0)
#endif
// DG: 0 = forward epilogues, 1 = EPI_DGELU (pre-activation in, gelu' computed here), 2 = EPI_DMUL (multiplier in),
//     3 = forward with SEDD_EPI_BIAS_GELU_GRAD (a kernel of its own: the 256-wide tiles sit at the register limit and a
//     run-time choice between the two GELU epilogues spilled in every instantiation)
//     4 = TN form that also emits the column sums of A over the reduction rows (the Linear's bias gradient next to its
//     weight gradient: colsum[m] (+)= sum_k A[k, m]) — one more MFMA per handled 16-row fragment and K tile against a
//     selector operand (row j of it all ones), so fragment j's sums land in column j of ONE extra accumulator; the
//     fragments of a row tile are dealt over the tile columns and their waves (fragment m to column-wave m mod count),
//     which keeps the extra matrix work to about one MFMA per wave and K tile
// BM_: rows of the output tile — 256, or 128 (NT / NN forms with the plain forward epilogues only) for outputs whose
//     256-row tiles would leave most of the chip idle (the dual-track model at batch 4: [2048, 1024] is 64 tiles of
//     256 x 128 = a quarter of the CUs, [512, 1024] sixteen).
template <typename TO, int BN_, int NST, bool TN, int DG, int NW = 8, bool NNB = false, bool ROT = false, int BM_ = 256>
__global__ void __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(2)))
gemm_bf16_big_kernel(const bf16_t* __restrict__ A, int64_t lda,
                                                            const bf16_t* __restrict__ B, int64_t ldb,
                                                            TO* __restrict__ C, int64_t ldc,
                                                            const float* __restrict__ bias, int epi,
                                                            TO* __restrict__ pre, int accumulate, int64_t M,
                                                            int64_t N, int64_t Kper, int64_t Ktot, int ntn_cs,
                                                            int ntiles, int splitk, float4* __restrict__ slabs,
                                                            float* __restrict__ colsum, RotArgs rot) {
    // ntn_cs = column tiles | (column split of the XCD map << 20), see big_tile_map / choose_csplit
    const int ntn = ntn_cs & 0xfffff, csplit = ntn_cs >> 20;
    static_assert(BM_ == 256 || (BM_ == 128 && !TN && DG == 0 && !ROT && NW == 8), "128-row tiles: plain NT / NN forms");
    static_assert(DG != 4 || (TN && sizeof(TO) == 4 && NW == 8), "column sums ride on the fp32 weight-gradient form");
    constexpr int WN = BN_ / 64, WM = NW / WN, TM = BM_ / WM / 16;  // waves along N / M, 16-row fragments per wave
    constexpr int LA = BM_ / (NW * 16), LB = BN_ / (NW * 16), LPT = LA + LB;  // DMA wave-instructions per stage
    constexpr int UNR = NST % 2 == 0 ? NST : 2 * NST;                 // K-loop unroll: whole ring turns, even
    static_assert(NW == 8 || (NW == 4 && BN_ == 128), "4-wave blocks carry 256 x 128 tiles");
    constexpr int ASLOTS = BM_ * 4, SLOTS = (BM_ + BN_) * 4;          // 16-byte slots per stage
    constexpr int TMH = TM / 2;
    constexpr bool STAGED = sizeof(TO) == 2;                          // bf16 outputs leave through the staging region
    static_assert(NST >= 3 && NST <= 6, "ring depth");
    static_assert(NW == 8 || SLOTS * 16 >= NW * StageCfg<TO>::REGION, "staging fits a ring stage");
    __shared__ __attribute__((aligned(16))) uint4 smem[NST][SLOTS];
    __shared__ __attribute__((aligned(16))) unsigned char stage_own[STAGED && NW == 8 ? 8 * StageCfg<TO>::REGION : 16];
    unsigned char* const stage_lds = NW == 8 ? stage_own : reinterpret_cast<unsigned char*>(&smem[NST - 1][0]);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    // Split-K launches are 1-D over (K slice, tile) with the slice as the slow index of the XCD-contiguous order: the
    // ~tiles*splitk/8 blocks that share an XCD's L2 then work on the same K slice, i.e. on the same operand panels.
    const int work = splitk > 1 ? xcd_remap(blockIdx.x, gridDim.x) : 0;
    const int64_t kbeg = (int64_t)(work / ntiles) * Kper;
    const int nkt = (int)((Ktot - kbeg < Kper ? Ktot - kbeg : Kper) / BIG_BK);
    constexpr bool TA = TN, TB = TN || NNB;  // which operand is stored with the reduction index as the row
    static_assert(!(TN && NNB), "NNB refines the NT form");
    const int64_t ksA = TA ? BIG_BK * lda : BIG_BK, ksB = TB ? BIG_BK * ldb : BIG_BK;  // operand advance per K tile
    const uint32_t ring = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)&smem[0][0];
    // fragment read offsets: lane -> row (lane & 15) of a 16-row fragment, k chunk (lane >> 4), swizzled
    const int r15 = lane & 15;
    const int fa = (wm * TM * 16 + r15) * 4 + ((lane >> 4) ^ ((r15 >> 1) & 3));
    const int fb = ASLOTS + (wn * 64 + r15) * 4 + ((lane >> 4) ^ ((r15 >> 1) & 3));
    // TN: transposing reads. lane = 16 g + 4 q + p supplies the address of row 4g+q, columns 4p..4p+3 of a 16-column block
    constexpr int PB = BN_ * 2;  // bytes per B row in a TN stage
    const int tg = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3, tyc = 4 * (tg & 1) + tq;
    const int ta = (4 * tg + tq) * 512 + (tp >> 1) * 16 + (tp & 1) * 8;
    const int tb = ASLOTS * 16 + ((NNB ? 8 : 4) * tg + tq) * PB + (tp >> 1) * 16 + (tp & 1) * 8;
    constexpr int TB_HI = (NNB ? 4 : 16) * PB;  // second transposing read of a B fragment: rows +4 (natural k) / +16
    const bool vec_ok = (ldc % 4 == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0) &&
                        (!pre || (reinterpret_cast<uintptr_t>(pre) & 15) == 0);
    const bool staged = STAGED && splitk == 1 && !accumulate && (ldc * sizeof(TO)) % 16 == 0 &&
                        ((reinterpret_cast<uintptr_t>(C) & 15) == 0) &&
                        (!pre || (reinterpret_cast<uintptr_t>(pre) & 15) == 0);

    // one DMA wave-instruction fills 16 rows x 64 B; lane -> (row, LDS chunk) = (lane >> 2, lane & 3)
    const bf16_t* srcA[LA];
    const bf16_t* srcB[LB];
    int64_t m0 = 0, n0 = 0;
#define BIG_TILE_SETUP(tile)                                                                                         \
    do {                                                                                                             \
        int tm_, tn_;                                                                                                \
        big_tile_map((int)(tile), ntiles, ntn, csplit, splitk > 1, tm_, tn_);                                        \
        m0 = (int64_t)tm_ * BM_;                                                                                     \
        n0 = (int64_t)tn_ * BN_;                                                                                     \
        _Pragma("unroll") for (int i = 0; i < LA; ++i) {                                                             \
            if constexpr (TA) { /* wave-instruction j = 2 token rows x 512 B */                                      \
                const int r = (i * NW + wave) * 2 + (lane >> 5);                                                      \
                int64_t col = m0 + (((lane & 31) ^ (2 * (r & 7))) << 3);                                             \
                col = col < M ? col : 0;                                                                             \
                srcA[i] = A + (kbeg + r) * lda + col;                                                                \
            } else {                                                                                                 \
                const int r = (i * NW + wave) * 16 + (lane >> 2);                                                     \
                const int c = (lane & 3) ^ ((r >> 1) & 3);                                                           \
                int64_t ar = m0 + r;                                                                                 \
                ar = ar < M ? ar : M - 1;                                                                            \
                srcA[i] = A + ar * lda + kbeg + c * 8;                                                               \
            }                                                                                                        \
        }                                                                                                            \
        _Pragma("unroll") for (int i = 0; i < LB; ++i) {                                                             \
            if constexpr (TB) {                                                                                      \
                constexpr int CPR = BN_ / 8; /* 16-byte chunks per B row */                                          \
                const int r = (i * NW + wave) * (64 / CPR) + lane / CPR;                                              \
                const int sw = NNB ? ((r & 3) | (((r >> 3) & 1) << 2)) : (r & 7);                                     \
                int64_t col = n0 + (((lane % CPR) ^ (2 * sw)) << 3);                                                 \
                col = col < N ? col : 0;                                                                             \
                srcB[i] = B + (kbeg + r) * ldb + col;                                                                \
            } else {                                                                                                 \
                const int r = (i * NW + wave) * 16 + (lane >> 2);                                                     \
                const int c = (lane & 3) ^ ((r >> 1) & 3);                                                           \
                int64_t br = n0 + r;                                                                                 \
                br = br < N ? br : N - 1;                                                                            \
                srcB[i] = B + br * ldb + kbeg + c * 8;                                                               \
            }                                                                                                        \
        }                                                                                                            \
    } while (0)
#define BIG_STAGE(st, kt)                                                                                            \
    do {                                                                                                             \
        _Pragma("unroll") for (int i = 0; i < LA; ++i)                                                               \
            lds_dma16(srcA[i] + (int64_t)(kt) * ksA, ring + ((st) * SLOTS + (i * NW + wave) * 64) * 16);              \
        _Pragma("unroll") for (int i = 0; i < LB; ++i)                                                               \
            lds_dma16(srcB[i] + (int64_t)(kt) * ksB, ring + ((st) * SLOTS + ASLOTS + (i * NW + wave) * 64) * 16);     \
    } while (0)
#define BIG_PROLOGUE()                                                                                               \
    do {                                                                                                             \
        _Pragma("unroll") for (int s = 0; s < NST - 1; ++s)                                                          \
            if (s < nkt) BIG_STAGE(s, s);                                                                            \
    } while (0)
#define BIG_TR(st, off) __builtin_amdgcn_ds_read_tr16_b64_v4i16(                                                     \
        (__attribute__((address_space(3))) s16x4*)(reinterpret_cast<unsigned char*>(&smem[st][0]) + (off)))
#define BIG_READ_A(dst, st, mbase)                                                                                   \
    do {                                                                                                             \
        _Pragma("unroll") for (int m = 0; m < TMH; ++m) {                                                            \
            if constexpr (TA) {                                                                                      \
                const int o_ = ta + (((wm * TM + (mbase) + m) ^ tyc) << 5);                                          \
                union { struct { s16x4 lo, hi; } s; bf16x8 v; } f_;                                                  \
                f_.s.lo = BIG_TR(st, o_);                                                                            \
                f_.s.hi = BIG_TR(st, o_ + 16 * 512);                                                                 \
                dst[m] = f_.v;                                                                                       \
            } else {                                                                                                 \
                dst[m] = *reinterpret_cast<const bf16x8*>(&smem[st][fa + ((mbase) + m) * 64]);                       \
            }                                                                                                        \
        }                                                                                                            \
    } while (0)
#define BIG_READ_B(set, st)                                                                                          \
    do {                                                                                                             \
        _Pragma("unroll") for (int n = 0; n < 4; ++n) {                                                              \
            if constexpr (TB) {                                                                                      \
                const int o_ = tb + (((wn * 4 + n) ^ tyc) << 5);                                                     \
                union { struct { s16x4 lo, hi; } s; bf16x8 v; } f_;                                                  \
                f_.s.lo = BIG_TR(st, o_);                                                                            \
                f_.s.hi = BIG_TR(st, o_ + TB_HI);                                                                    \
                b[set][n] = f_.v;                                                                                    \
            } else {                                                                                                 \
                b[set][n] = *reinterpret_cast<const bf16x8*>(&smem[st][fb + n * 64]);                                \
            }                                                                                                        \
        }                                                                                                            \
    } while (0)
#define BIG_MMA(asrc, mbase, set)                                                                                    \
    do {                                                                                                             \
        __builtin_amdgcn_s_setprio(1);                                                                               \
        _Pragma("unroll") for (int m = 0; m < TMH; ++m)                                                              \
            _Pragma("unroll") for (int n = 0; n < 4; ++n)                                                            \
                acc[(mbase) + m][n] =                                                                                \
                    __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[set][n], asrc[m], acc[(mbase) + m][n], 0, 0, 0);       \
        __builtin_amdgcn_s_setprio(0);                                                                               \
    } while (0)

#define BIG_CS(asrc, mbase)                                                                                          \
    do {                                                                                                             \
        if constexpr (DG == 4) {                                                                                     \
            _Pragma("unroll") for (int m = 0; m < TMH; ++m)                                                          \
                if ((cs_mask >> ((mbase) + m)) & 1) {                                                                \
                    union { uint32_t w[4]; bf16x8 v; } sel_;                                                         \
                    const uint32_t one2_ = r15 == (mbase) + m ? 0x3F803F80u : 0u;                                    \
                    sel_.w[0] = sel_.w[1] = sel_.w[2] = sel_.w[3] = one2_;                                           \
                    cs_acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sel_.v, asrc[m], cs_acc, 0, 0, 0);              \
                }                                                                                                    \
        }                                                                                                            \
    } while (0)

#ifdef GEMM_CLOCK_STAMP  // diagnostic build (tools/gemm_clock.py): shader clock held under load = d(memtime) / d(memrealtime) x 100 MHz
    const uint64_t stamp_c0 = __builtin_amdgcn_s_memtime(), stamp_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    // persistent over output tiles: the next tile's first K tiles are in flight while this tile's results leave
    int tile = splitk > 1 ? work % ntiles : (int)blockIdx.x;
    if (tile >= ntiles) return;
#ifdef GEMM_PHASE_STAMP
    int phase_round = 0;
#ifdef GEMM_SKEW_TICKS  // experiment: half of the blocks start late, so epilogues and main loops of neighbours overlap
    if (NW == 4 ? blockIdx.x >= gridDim.x / 2 : (blockIdx.x >> 3) & 1) {
        const uint64_t t0_ = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_s_memrealtime() - t0_ < GEMM_SKEW_TICKS) __builtin_amdgcn_s_sleep(32);
    }
#endif
#endif
    BIG_TILE_SETUP(tile);
    BIG_PROLOGUE();
    while (true) {
        PHASE_STAMP(0);
        f32x4 acc[TM][4];
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // Fragment registers are pipelined at half-stage granularity: while the lower half of the A fragments
        // multiplies, the upper half is read; while the upper half multiplies, the next K tile's lower half and B
        // fragments are read (B is double-buffered, A's halves alternate) — 64 fragment registers instead of 96.
        bf16x8 a_lo[TMH], a_hi[TMH], b[2][4];
        [[maybe_unused]] f32x4 cs_acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        [[maybe_unused]] int cs_mask = 0;
        if constexpr (DG == 4) {
            if (colsum) {
                const int q = (int)(n0 / BN_) * WN + wn, nq = ntn * WN;  // this wave among the row tile's column-waves
#pragma unroll
                for (int m = 0; m < TM; ++m) cs_mask |= (m % nq == q) << m;
            }
        }
        // first K tile of this output tile: everything older (the previous tile's stores included) must have drained
        wait_tiles<LPT>(nkt - 1 < NST - 2 ? nkt - 1 : NST - 2);
        __builtin_amdgcn_s_barrier();
        BIG_READ_A(a_lo, 0, 0);
        BIG_READ_B(0, 0);
        // (Tried and measured slower, same-device A/B on the cfg-2 shapes and 4096^3, results bit-identical: straight-line
        // steps with literal ring positions -2..-4 %; the next tile's fragment reads issued unconditionally so that the
        // second MFMA group's lgkmcnt need not cover them +2 % on top of that, still below this form; m0 as wave base +
        // literal with the K advance in the DMA's immediate offset -4 % — and note the instruction adds that immediate
        // to the LDS address as well as to the global one.  With all-zero operands the same binary runs 20 % faster
        // (1.26 -> 1.52 PFLOP/s at 4096^3): under random data the chip holds ~1.94 GHz, not 2.4, in this kernel.  Issuing
        // the DMA pieces one at a time inside the MFMA groups instead of in a burst: equal within noise on every shape.)
        for (int kt0 = 0; kt0 < nkt; kt0 += UNR) {
#pragma unroll
            for (int s = 0; s < UNR; ++s) {
                const int kt = kt0 + s;
                if (kt < nkt) {
                    BIG_READ_A(a_hi, s % NST, TMH);
                    BIG_MMA(a_lo, 0, s & 1);
                    BIG_CS(a_lo, 0);
                    if (kt + 1 < nkt) {
                        const int rem = nkt - 2 - kt;
                        wait_tiles<LPT>(rem < NST - 3 ? rem : NST - 3);  // K tile kt+1 has landed (this wave's share)
                        __builtin_amdgcn_s_barrier();  // ... every wave's share; the stage of tile kt-1 is consumed
#ifndef GEMM_NO_STAGGER
                        // the two waves of a SIMD run the same program in step: if both issue their LDS-DMA pieces right
                        // behind the barrier, neither feeds the matrix pipe meanwhile.  The second half of the block
                        // issues after its second MFMA group instead (same vmcnt bookkeeping: still before the next wait)
                        if (wave < NW / 2 && kt + NST - 1 < nkt) BIG_STAGE((s + NST - 1) % NST, kt + NST - 1);
#else
                        if (kt + NST - 1 < nkt) BIG_STAGE((s + NST - 1) % NST, kt + NST - 1);
#endif
                        BIG_READ_A(a_lo, (s + 1) % NST, 0);
                        BIG_READ_B((s + 1) & 1, (s + 1) % NST);
                    }
                    BIG_MMA(a_hi, TMH, s & 1);
                    BIG_CS(a_hi, TMH);
#ifndef GEMM_NO_STAGGER
                    if (wave >= NW / 2 && kt + 1 < nkt && kt + NST - 1 < nkt) BIG_STAGE((s + NST - 1) % NST, kt + NST - 1);
#endif
                }
            }
        }
        const int64_t row_t = m0 + wm * TM * 16, col_w = n0 + wn * 64;
        const int next = splitk > 1 ? ntiles : tile + (int)gridDim.x;
        // the ring is free once every wave has issued its last MFMAs: start the next tile's loads under the epilogue
        __builtin_amdgcn_s_barrier();
        PHASE_STAMP(1);
        if (next < ntiles) {
            BIG_TILE_SETUP(next);
            BIG_PROLOGUE();
        }
        if constexpr (DG == 4) {
            if (cs_mask) {
                // lane (i, g) holds, in element r, the sum of fragment 4g + r's row i
                float* cs_slab = reinterpret_cast<float*>(slabs + (int64_t)splitk * ntiles * (BM_ * BN_ / 4));
                const int64_t mpad = (int64_t)(ntiles / ntn) * BM_;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = 4 * (lane >> 4) + r;
                    const int64_t row = row_t + m * 16 + (lane & 15);
                    if (m < TM && ((cs_mask >> m) & 1) && row < M) {
                        if (splitk > 1 && slabs) cs_slab[(int64_t)(work / ntiles) * mpad + row] = cs_acc[r];
                        else if (splitk > 1) atomicAdd(colsum + row, cs_acc[r]);
                        else colsum[row] = accumulate ? colsum[row] + cs_acc[r] : cs_acc[r];
                    }
                }
            }
        }
        if (splitk > 1 && slabs) {
            // split-K partial tile: parked in the workspace in register order (every store a contiguous 1 KiB per wave);
            // splitk_reduce_kernel sums the slices and applies them to C
            float4* dst = slabs + ((int64_t)(work / ntiles) * ntiles + tile) * (BM_ * BN_ / 4) + wave * TM * 4 * 64 + lane;
#pragma unroll
            for (int m = 0; m < TM; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    dst[(m * 4 + n) * 64] = make_float4(acc[m][n][0], acc[m][n][1], acc[m][n][2], acc[m][n][3]);
        } else if (sizeof(TO) == 4 && accumulate && splitk == 1 && vec_ok && epi == SEDD_EPI_NONE) {
            // C += tile (weight gradients with a short reduction, e.g. adaLN: K = batch): the old values of a fragment
            // row are requested four fragments at a time — one load-add-store chain per fragment serialises 32 memory
            // round trips per lane, which at K = 64 was the whole kernel (29 us)
#pragma unroll
            for (int m = 0; m < TM; ++m) {
                const int64_t row = row_t + m * 16 + (lane & 15);
                float4 old[4];
                bool ok[4];
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    const int64_t col = col_w + n * 16 + 4 * (lane >> 4);
                    ok[n] = row < M && col + 4 <= N;
                    old[n] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (ok[n]) old[n] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(C) + row * ldc + col);
                }
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    const int64_t col = col_w + n * 16 + 4 * (lane >> 4);
                    if (ok[n])
                        *reinterpret_cast<float4*>(reinterpret_cast<float*>(C) + row * ldc + col) =
                            make_float4(old[n].x + acc[m][n][0], old[n].y + acc[m][n][1], old[n].z + acc[m][n][2],
                                        old[n].w + acc[m][n][3]);
                    else
                        epilogue4<TO, DG == 3>(C, ldc, pre, bias, epi, accumulate, 0, row, col, M, N, acc[m][n], vec_ok);
                }
            }
        } else if (!staged) {  // fp32 / split-K atomics / accumulate / unaligned C: straight from the accumulators
#pragma unroll
            for (int m = 0; m < TM; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    epilogue4<TO, DG == 3>(C, ldc, pre, bias, epi, accumulate, splitk > 1, row_t + m * 16 + (lane & 15),
                                  col_w + n * 16 + 4 * (lane >> 4), M, N, acc[m][n], vec_ok);
        } else if constexpr (STAGED) {
            using SC = StageCfg<TO>;
            constexpr int FPP = SC::RPP / 16;  // 16-row fragments per pass
            unsigned char* reg = stage_lds + wave * SC::REGION;
            float4 bv[4];
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                bv[n] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (epi == SEDD_EPI_BIAS || epi == SEDD_EPI_BIAS_GELU || DG == 3) {
                    const int64_t c = col_w + n * 16 + 4 * (lane >> 4);
                    bv[n].x = c + 0 < N ? bias[c + 0] : 0.f;
                    bv[n].y = c + 1 < N ? bias[c + 1] : 0.f;
                    bv[n].z = c + 2 < N ? bias[c + 2] : 0.f;
                    bv[n].w = c + 3 < N ? bias[c + 3] : 0.f;
                }
            }
            if constexpr (DG == 1 || DG == 2) {
                // data gradient through GELU: the pre-activation tile comes in with coalesced 16-byte loads (all
                // passes up front, under the next tile's prologue), is re-read from the staging region in the MFMA
                // layout, and the column sums of the result (the bias gradient of the Linear below) are reduced over
                // the wave's rows and added to colsum with one atomic per column and wave
                f32x4 cs[4];
#pragma unroll
                for (int n = 0; n < 4; ++n) cs[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if constexpr (DG == 2) {
                    // multiplier tile straight from global memory in the accumulator layout (lane: row lane & 15, the 4
                    // columns of fragment n): 8-byte loads, the four fragments of a row together cover its 128-byte line;
                    // no LDS round trip and no sigmoid — the pass is a multiply, the column sums and the store
                    constexpr int PD = TM;  // every pass in flight at once: the fragment registers are free here
                    uint2 mv[PD][4];
#define DMUL_LOAD(pass_)                                                                                             \
    _Pragma("unroll") for (int n = 0; n < 4; ++n) {                                                                  \
        const int64_t gr = row_t + (pass_) * 16 + (lane & 15), gc = col_w + n * 16 + 4 * (lane >> 4);                \
        mv[(pass_) % PD][n] = make_uint2(0, 0);                                                                      \
        if (gr < M && gc + 4 <= N) mv[(pass_) % PD][n] = *reinterpret_cast<const uint2*>(pre + gr * ldc + gc);       \
    }
#pragma unroll
                    for (int pass = 0; pass < PD; ++pass) DMUL_LOAD(pass)
#pragma unroll
                    for (int pass = 0; pass < TM; ++pass) {
#pragma unroll
                        for (int n = 0; n < 4; ++n) {
                            union { uint2 u; TO e[4]; } pv;
                            pv.u = mv[pass % PD][n];
                            f32x4& v = acc[pass][n];
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                // the reference rounds the Linear's data gradient to the activation dtype before GELU's
                                const float dy = to_f32<TO>(from_f32<TO>(v[r]));
                                v[r] = to_f32<TO>(from_f32<TO>(dy * to_f32<TO>(pv.e[r])));
                                cs[n][r] += v[r];  // (rows / columns beyond the matrix: multiplier 0, so they add 0)
                            }
                        }
                        if (pass + PD < TM) DMUL_LOAD(pass + PD)
#pragma unroll
                        for (int n = 0; n < 4; ++n)
                            stage_put4<TO>(reg, lane & 15, n * 16 + 4 * (lane >> 4), acc[pass][n]);
                        stage_flush<TO>(reg, C, ldc, row_t + pass * 16, col_w, M, N, lane, 0);
                    }
#undef DMUL_LOAD
                } else {
                constexpr int RPI = 64 / SC::LPR;
                const int rsub = lane / SC::LPR, chunk = lane % SC::LPR;
                constexpr int PD = TM > 4 ? 2 : 4;  // passes of `pre` kept in flight (register budget)
                uint4 pr[PD][SC::RPP / RPI];
#define DGELU_LOAD(pass_)                                                                                            \
    _Pragma("unroll") for (int i = 0; i < SC::RPP / RPI; ++i) {                                                      \
        const int64_t gr = row_t + (pass_) * 16 + i * RPI + rsub, gc = col_w + chunk * SC::EPC;                      \
        pr[(pass_) % PD][i] = make_uint4(0, 0, 0, 0);                                                                \
        if (gr < M && gc + SC::EPC <= N) pr[(pass_) % PD][i] = *reinterpret_cast<const uint4*>(pre + gr * ldc + gc); \
    }
#pragma unroll
                for (int pass = 0; pass < PD; ++pass) DGELU_LOAD(pass)
#pragma unroll
                for (int pass = 0; pass < TM; ++pass) {
#pragma unroll
                    for (int i = 0; i < SC::RPP / RPI; ++i)
                        *reinterpret_cast<uint4*>(reg + (i * RPI + rsub) * SC::PITCH + chunk * 16) = pr[pass % PD][i];
                    if (pass + PD < TM) DGELU_LOAD(pass + PD)
                    const bool rok = row_t + pass * 16 + (lane & 15) < M;
#pragma unroll
                    for (int n = 0; n < 4; ++n) {
                        union { uint2 u; TO e[4]; } pv;
                        pv.u = *reinterpret_cast<const uint2*>(reg + (lane & 15) * SC::PITCH + (n * 16 + 4 * (lane >> 4)) * sizeof(TO));
                        f32x4& v = acc[pass][n];
                        const f32x4 gp = gelu_grad4_t<TO>((f32x4){to_f32<TO>(pv.e[0]), to_f32<TO>(pv.e[1]),
                                                                  to_f32<TO>(pv.e[2]), to_f32<TO>(pv.e[3])});
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            // the reference rounds the Linear's data gradient to the activation dtype before GELU's
                            const float dy = to_f32<TO>(from_f32<TO>(v[r]));
                            v[r] = to_f32<TO>(from_f32<TO>(dy * gp[r]));
                            cs[n][r] += rok ? v[r] : 0.f;
                        }
                    }
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        stage_put4<TO>(reg, lane & 15, n * 16 + 4 * (lane >> 4), acc[pass][n]);
                    stage_flush<TO>(reg, C, ldc, row_t + pass * 16, col_w, M, N, lane, 0);
                }
#undef DGELU_LOAD
                }
                if (colsum) {
                    // the block's WM row groups meet in LDS (each wave's staging region is free after its last flush)
                    // and leave with ONE atomic per column and block, added in a fixed order: with a single row tile
                    // the sums are bit-reproducible, and the atomic count drops WM-fold
                    float* mine = reinterpret_cast<float*>(reg);
#pragma unroll
                    for (int n = 0; n < 4; ++n)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float t = cs[n][r];
                            t += __shfl_xor(t, 1, 64);
                            t += __shfl_xor(t, 2, 64);
                            t += __shfl_xor(t, 4, 64);
                            t += __shfl_xor(t, 8, 64);
                            if ((lane & 15) == 0) mine[n * 16 + 4 * (lane >> 4) + r] = t;
                        }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (not __syncthreads(): it would also wait for
                    __builtin_amdgcn_s_barrier();                       //  the tile's stores and the next tile's DMA)
                    if (wm == 0) {
                        float t = 0.f;
#pragma unroll
                        for (int w = 0; w < WM; ++w)
                            t += reinterpret_cast<const float*>(stage_lds + (w * WN + wn) * SC::REGION)[lane];
                        const int64_t c = col_w + lane;
                        // with a workspace (it rides in `slabs`, which split-K never uses here): this row tile's sums
                        // go to their own row of it and a fixed-order fold adds them up (bit-reproducible); else atomics
                        if (c < N) {
                            if (slabs) reinterpret_cast<float*>(slabs)[(row_t / BM_) * N + c] = t;
                            else atomicAdd(colsum + c, t);
                        }
                    }
                }
            } else {
            // ROT: the cos / sin rows of pass p+RPD are requested while pass p is rotated, staged and stored
            constexpr int RPD = 2;
            float4 rc[ROT ? RPD : 1][2], rs[ROT ? RPD : 1][2];
            const bool rot_on = ROT && col_w < rot.cols;  // (wave-uniform) a q or k head
#define ROT_LOAD(pass_)                                                                                              \
    do {                                                                                                             \
        const int64_t rr_ = row_t + (pass_) * 16 + (lane & 15);                                                      \
        const int l_ = (int)((rr_ < M ? rr_ : M - 1) % rot.L);                                                       \
        _Pragma("unroll") for (int nb = 0; nb < 2; ++nb) {                                                           \
            rc[(pass_) % RPD][nb] = *reinterpret_cast<const float4*>(rot.cos + (int64_t)l_ * 32 + 4 * (lane >> 4) + nb * 16); \
            rs[(pass_) % RPD][nb] = *reinterpret_cast<const float4*>(rot.sin + (int64_t)l_ * 32 + 4 * (lane >> 4) + nb * 16); \
        }                                                                                                            \
    } while (0)
            if constexpr (ROT) {
                if (rot_on) {
#pragma unroll
                    for (int p0 = 0; p0 < RPD && p0 < TM; ++p0) ROT_LOAD(p0);
                }
            }
#pragma unroll
            for (int pass = 0; pass < TM / FPP; ++pass) {
                const int64_t row_w = row_t + pass * SC::RPP;
                if (DG == 3 || epi == SEDD_EPI_BIAS_GELU) {
                    // the reference rounds the Linear output to the activation dtype before GELU; `pre` keeps it (or,
                    // GELU_GRAD, the derivative there: both come out of the one sigmoid)
#pragma unroll
                    for (int mm = 0; mm < FPP; ++mm)
#pragma unroll
                        for (int n = 0; n < 4; ++n) {
                            f32x4& v = acc[pass * FPP + mm][n];
                            v[0] = to_f32<TO>(from_f32<TO>(v[0] + bv[n].x));
                            v[1] = to_f32<TO>(from_f32<TO>(v[1] + bv[n].y));
                            v[2] = to_f32<TO>(from_f32<TO>(v[2] + bv[n].z));
                            v[3] = to_f32<TO>(from_f32<TO>(v[3] + bv[n].w));
                            if constexpr (DG == 3) {
                                f32x4 gp;
                                gelu_both4_t<TO>(v, v, gp);
                                if (pre) stage_put4<TO>(reg, mm * 16 + (lane & 15), n * 16 + 4 * (lane >> 4), gp);
                            } else {
                                if (pre) stage_put4<TO>(reg, mm * 16 + (lane & 15), n * 16 + 4 * (lane >> 4), v);
                                v = gelu4_t<TO>(v);
                            }
                        }
                    if (pre) stage_flush<TO>(reg, pre, ldc, row_w, col_w, M, N, lane, 0);
                } else if (epi == SEDD_EPI_BIAS) {
#pragma unroll
                    for (int mm = 0; mm < FPP; ++mm)
#pragma unroll
                        for (int n = 0; n < 4; ++n) {
                            f32x4& v = acc[pass * FPP + mm][n];
                            v[0] += bv[n].x; v[1] += bv[n].y; v[2] += bv[n].z; v[3] += bv[n].w;
                        }
                }
                if constexpr (ROT) {
                    static_assert(FPP == 1, "one 16-row fragment per pass");
                    if (rot_on) {
#pragma unroll
                        for (int nb = 0; nb < 2; ++nb) {
                            const float4 c4 = rc[pass % RPD][nb], s4 = rs[pass % RPD][nb];
                            const float cc[4] = {c4.x, c4.y, c4.z, c4.w}, ss[4] = {s4.x, s4.y, s4.z, s4.w};
                            f32x4& lo = acc[pass][nb];
                            f32x4& hi = acc[pass][nb + 2];
#pragma unroll
                            for (int r = 0; r < 4; ++r) {  // the Linear's output is rounded to the activation dtype first
                                const float x1 = to_f32<TO>(from_f32<TO>(lo[r])), x2 = to_f32<TO>(from_f32<TO>(hi[r]));
                                lo[r] = x1 * cc[r] - x2 * ss[r];
                                hi[r] = x2 * cc[r] + x1 * ss[r];
                            }
                        }
                        if (pass + RPD < TM) ROT_LOAD(pass + RPD);
                    }
                }
#pragma unroll
                for (int mm = 0; mm < FPP; ++mm)
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        stage_put4<TO>(reg, mm * 16 + (lane & 15), n * 16 + 4 * (lane >> 4), acc[pass * FPP + mm][n]);
                stage_flush<TO>(reg, C, ldc, row_w, col_w, M, N, lane, 0);
            }
#undef ROT_LOAD
            }
        }
        PHASE_STAMP(2);
#ifdef GEMM_PHASE_STAMP
        ++phase_round;
#endif
        if (next >= ntiles) break;
        tile = next;
    }
#ifdef GEMM_CLOCK_STAMP
    if (tid == 0 && blockIdx.x < 1024) {
        sedd_gemm_stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - stamp_c0;
        sedd_gemm_stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - stamp_r0;
    }
#endif
#undef BIG_MMA
#undef BIG_READ_B
#undef BIG_READ_A
#undef BIG_TR
#undef BIG_PROLOGUE
#undef BIG_STAGE
#undef BIG_TILE_SETUP
}

// ------------------------------------------------------------------------------- bf16, generic register-staged
template <typename TO>
__global__ void __launch_bounds__(256) gemm_bf16_nt_kernel(const bf16_t* __restrict__ A, int64_t lda,
                                                           const bf16_t* __restrict__ B, int64_t ldb,
                                                           TO* __restrict__ C, int64_t ldc,
                                                           const float* __restrict__ bias, int epi,
                                                           TO* __restrict__ pre, int accumulate, int64_t M, int64_t N,
                                                           int64_t K, int ntn) {
    __shared__ __attribute__((aligned(16))) uint4 sA[BM * 8];
    __shared__ __attribute__((aligned(16))) uint4 sB[BN * 8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int64_t m0 = (int64_t)(wg / ntn) * BM, n0 = (int64_t)(wg % ntn) * BN;

    f32x4 acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int64_t nkt = (K + BK16 - 1) / BK16;
    for (int64_t kt = 0; kt < nkt; ++kt) {
        uint4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
#define GLOAD(i, ra, rb)                                                                                   \
        {                                                                                                  \
            const int chunk = tid + 256 * (i), row = chunk >> 3, kc = chunk & 7;                           \
            const int64_t k = kt * BK16 + kc * 8;                                                          \
            int64_t ar = m0 + row, br = n0 + row;                                                          \
            ar = ar < M ? ar : M - 1;                                                                      \
            br = br < N ? br : N - 1;                                                                      \
            ra = make_uint4(0, 0, 0, 0);                                                                   \
            rb = make_uint4(0, 0, 0, 0);                                                                   \
            if (k < K) {                                                                                   \
                ra = *reinterpret_cast<const uint4*>(A + ar * lda + k);                                    \
                rb = *reinterpret_cast<const uint4*>(B + br * ldb + k);                                    \
            }                                                                                              \
        }
        GLOAD(0, ra0, rb0) GLOAD(1, ra1, rb1) GLOAD(2, ra2, rb2) GLOAD(3, ra3, rb3)
#undef GLOAD
        __syncthreads();
#define LSTORE(i, ra, rb)                                                       \
        {                                                                       \
            const int chunk = tid + 256 * (i), row = chunk >> 3, kc = chunk & 7; \
            sA[row * 8 + (kc ^ (row & 7))] = ra;                                \
            sB[row * 8 + (kc ^ (row & 7))] = rb;                                \
        }
        LSTORE(0, ra0, rb0) LSTORE(1, ra1, rb1) LSTORE(2, ra2, rb2) LSTORE(3, ra3, rb3)
#undef LSTORE
        __syncthreads();
        mma_tile(sA, sB, wr, wc, lane, acc);
    }
    store_tile<TO>(C, ldc, pre, bias, epi, accumulate, 0, m0, n0, wr, wc, lane, M, N, acc);
}

// ------------------------------------------------------------------------------------------------ fp32
template <typename TO>
__device__ __forceinline__ void epilogue1(TO* __restrict__ C, int64_t ldc, TO* __restrict__ pre,
                                          const float* __restrict__ bias, int epi, int accumulate, int64_t row,
                                          int64_t col, int64_t M, int64_t N, float v) {
    if (row >= M || col >= N) return;
    if (epi != SEDD_EPI_NONE) v += bias[col];
    if (epi == SEDD_EPI_BIAS_GELU || epi == SEDD_EPI_BIAS_GELU_GRAD) {
        const TO r = from_f32<TO>(v);
        if (pre) pre[row * ldc + col] = epi == SEDD_EPI_BIAS_GELU_GRAD ? from_f32<TO>(gelu_tanh_grad_f(to_f32<TO>(r))) : r;
        v = gelu_tanh_f(to_f32<TO>(r));
    }
    TO* p = C + row * ldc + col;
    if (accumulate) v += to_f32<TO>(*p);
    *p = from_f32<TO>(v);
}

// C[M,N] = A * B^T over exact-fp32 MFMA (32x32x2: the products of one output are summed in ascending k, an fmaf chain).
// Operand layouts are template flags, so the data-gradient (B stored [K, N]) and weight-gradient (A stored [K, M],
// B stored [K, N]) products of nn.Linear read the row-major activations and weights as they are, without transposed copies:
//   AKM = false: A[i][k] at A[i * lda + k]      AKM = true: at A[k * lda + i]   (likewise BKM for B[j][k])
// Block = 4 waves as 2 x 2, wave tile (TM*32) x (TN*32) -> block tile (TM*64) x (TN*64); the host picks 128x128, 128x64
// or 64x64 by how many tiles the output gives the 256 CUs.  K step BK: the next step's global loads are issued into
// registers before the MFMA loop of the current one and stored to LDS after it.  LDS holds a tile k-major
// ([BK][rows (+1)]): operand reads (lane -> row) are consecutive words; a row-major source is scattered in as scalars
// (the odd row stride spreads the 8 k-chunks of a wave over all banks), a k-major source as whole float4.
// gridDim.y > 1 splits K: slice y covers [y * kper, (y + 1) * kper) and parks its partial tile in `slabs`
// (splitk_fold_kernel adds the slices in order).
// Measured and not kept: two LDS buffers with the next tile stored half way through the current step's MFMAs and one
// barrier per step — 3-8 % slower on every shape (8192 x 4096 x 1024 NN: 586 -> 929 us); the doubled LDS halves the blocks
// a CU can hold, and co-resident blocks are what hides this kernel's store / barrier phases.
template <typename TO, int TM, int TN, bool AKM, bool BKM, int BK>
__global__ void __launch_bounds__(256) gemm_f32_kernel(const float* __restrict__ A, int64_t lda,
                                                       const float* __restrict__ B, int64_t ldb, TO* __restrict__ C,
                                                       int64_t ldc, const float* __restrict__ bias, int epi,
                                                       TO* __restrict__ pre, int accumulate, int64_t M, int64_t N,
                                                       int64_t K, int64_t kper, int ntn, float* __restrict__ slabs) {
    constexpr int RA = TM * 64, RB = TN * 64;                      // rows of the block's A / B tiles
    constexpr int SA = RA + (AKM ? 0 : 1), SB = RB + (BKM ? 0 : 1);  // LDS row strides (floats)
    constexpr int NA = RA * BK / 1024, NB = RB * BK / 1024;        // float4 chunks per thread and K step
    __shared__ __align__(16) float sA[BK * SA];
    __shared__ __align__(16) float sB[BK * SB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int64_t m0 = (int64_t)(wg / ntn) * RA, n0 = (int64_t)(wg % ntn) * RB;
    const int64_t kbeg = (int64_t)blockIdx.y * kper, kend = kbeg + kper < K ? kbeg + kper : K;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    float4 ra[NA], rb[NB];
    // one K step of an operand tile, global -> registers.  Rows past the edge are clamped (their products are never
    // stored), k past the end of the slice reads as zero.
#define F32_GLOAD(R_, KM_, NCH, P, ld, r0, ext, reg, k0)                                        \
    _Pragma("unroll") for (int i = 0; i < NCH; ++i) {                                           \
        const int c = tid + 256 * i;                                                            \
        reg[i] = make_float4(0.f, 0.f, 0.f, 0.f);                                               \
        if (KM_) {                                                                              \
            const int kk = c / (R_ / 4), r4 = c % (R_ / 4);                                     \
            int64_t row = r0 + r4 * 4;                                                          \
            row = row < ext ? row : ext - 4;                                                    \
            if (k0 + kk < kend) reg[i] = *reinterpret_cast<const float4*>(P + (k0 + kk) * ld + row); \
        } else {                                                                                \
            const int row_ = c / (BK / 4), kc = c % (BK / 4);                                                \
            int64_t row = r0 + row_;                                                            \
            row = row < ext ? row : ext - 1;                                                    \
            if (k0 + kc * 4 < kend) reg[i] = *reinterpret_cast<const float4*>(P + row * ld + k0 + kc * 4); \
        }                                                                                       \
    }
#define F32_LSTORE(R_, S_, KM_, NCH, sX, reg)                                                   \
    _Pragma("unroll") for (int i = 0; i < NCH; ++i) {                                           \
        const int c = tid + 256 * i;                                                            \
        if (KM_) {                                                                              \
            const int kk = c / (R_ / 4), r4 = c % (R_ / 4);                                     \
            *reinterpret_cast<float4*>(sX + kk * S_ + r4 * 4) = reg[i];                         \
        } else {                                                                                \
            const int row_ = c / (BK / 4), kc = c % (BK / 4);                                                \
            float* p = sX + (kc * 4) * S_ + row_;                                               \
            p[0] = reg[i].x; p[S_] = reg[i].y; p[2 * S_] = reg[i].z; p[3 * S_] = reg[i].w;      \
        }                                                                                       \
    }
    F32_GLOAD(RA, AKM, NA, A, lda, m0, M, ra, kbeg)
    F32_GLOAD(RB, BKM, NB, B, ldb, n0, N, rb, kbeg)
    for (int64_t k0 = kbeg; k0 < kend; k0 += BK) {
        F32_LSTORE(RA, SA, AKM, NA, sA, ra)
        F32_LSTORE(RB, SB, BKM, NB, sB, rb)
        __syncthreads();
        if (k0 + BK < kend) {
            F32_GLOAD(RA, AKM, NA, A, lda, m0, M, ra, k0 + BK)
            F32_GLOAD(RB, BKM, NB, B, ldb, n0, N, rb, k0 + BK)
        }
        // 32x32x2: lane l holds A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31].  Operands are read a group of
        // G k-pairs ahead of the MFMAs that use them (the scheduler, left alone, reads each pair right before its use
        // and every MFMA pair then waits a full LDS latency).
        constexpr int G = 4, NG = BK / 2 / G;
        float a[2][G][TM], b[2][G][TN];
#define F32_LREAD(g, buf)                                                                        \
        _Pragma("unroll") for (int q = 0; q < G; ++q) {                                          \
            const int k = ((g) * G + q) * 2 + (lane >> 5);                                       \
            _Pragma("unroll") for (int m = 0; m < TM; ++m)                                       \
                a[buf][q][m] = sA[k * SA + wr * (TM * 32) + m * 32 + (lane & 31)];               \
            _Pragma("unroll") for (int n = 0; n < TN; ++n)                                       \
                b[buf][q][n] = sB[k * SB + wc * (TN * 32) + n * 32 + (lane & 31)];               \
        }
        F32_LREAD(0, 0)
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g + 1 < NG) F32_LREAD(g + 1, (g + 1) & 1)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < G; ++q)
#pragma unroll
                for (int m = 0; m < TM; ++m)
#pragma unroll
                    for (int n = 0; n < TN; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g & 1][q][m], b[g & 1][q][n], acc[m][n], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#undef F32_LREAD
        __syncthreads();
    }
#undef F32_GLOAD
#undef F32_LSTORE
    // C/D map of the 32x32 tile: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = m0 + wr * (TM * 32) + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int64_t col = n0 + wc * (TN * 32) + n * 32 + (lane & 31);
                if (slabs) {
                    if (row < M && col < N) slabs[((int64_t)blockIdx.y * M + row) * N + col] = acc[m][n][r];
                } else {
                    epilogue1<TO>(C, ldc, pre, bias, epi, accumulate, row, col, M, N, acc[m][n][r]);
                }
            }
}


// second pass of a split-K product: C (+)= sum over K slices of the partial tiles parked by gemm_bf16_big_kernel.
// One thread per float4 of a tile in the kernel's register order: position = ((wave * TM + m) * 4 + n) * 64 + lane.
template <int BN_>
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const float4* __restrict__ slabs, float* __restrict__ C,
                                                            int64_t ldc, int64_t M, int64_t N, int ntn, int ntiles,
                                                            int splitk, int accumulate, float* __restrict__ colsum) {
    constexpr int WN = BN_ / 64, WM = 8 / WN, TM = BIG_BM / WM / 16, PPT = BIG_BM * BN_ / 4;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)ntiles * PPT) {
        // the column sums of A that rode along (DG = 4): one thread per row of C, the slices' partial sums lie behind
        // the partial tiles
        const int64_t row = idx - (int64_t)ntiles * PPT, mpad = (int64_t)(ntiles / ntn) * BIG_BM;
        if (!colsum || row >= M) return;
        const float* cs = reinterpret_cast<const float*>(slabs + (int64_t)splitk * ntiles * PPT);
        float t = cs[row];
        for (int k = 1; k < splitk; ++k) t += cs[(int64_t)k * mpad + row];
        colsum[row] = accumulate ? colsum[row] + t : t;
        return;
    }
    const int tile = (int)(idx / PPT), pos = (int)(idx % PPT);
    float4 s = slabs[(int64_t)tile * PPT + pos];
    for (int k = 1; k < splitk; ++k) {
        const float4 v = slabs[((int64_t)k * ntiles + tile) * PPT + pos];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const int lane = pos & 63, frag = pos >> 6, n = frag & 3, m = (frag >> 2) % TM, wave = frag / (4 * TM);
    const int64_t row = (int64_t)(tile / ntn) * BIG_BM + (wave / WN) * TM * 16 + m * 16 + (lane & 15);
    const int64_t col = (int64_t)(tile % ntn) * BN_ + (wave % WN) * 64 + n * 16 + 4 * (lane >> 4);
    if (row >= M || col >= N) return;
    float* p = C + row * ldc + col;
    const float v[4] = {s.x, s.y, s.z, s.w};
    if (col + 4 <= N && ldc % 4 == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0) {
        float4 o = accumulate ? *reinterpret_cast<float4*>(p) : make_float4(0.f, 0.f, 0.f, 0.f);
        o.x += s.x; o.y += s.y; o.z += s.z; o.w += s.w;
        *reinterpret_cast<float4*>(p) = o;
    } else {
        for (int r = 0; r < 4 && col + r < N; ++r) p[r] = accumulate ? p[r] + v[r] : v[r];
    }
}

// Column split of the XCD tile map (big_tile_map) for a multi-round NT / NN launch: modelled fabric reads =
// S x (A bytes: every A panel is read by S XCDs) + B traffic, where a B slice of <= 2.5 MB stays L2-resident (read once
// per XCD) and a larger one is streamed again every round.  fc1 forward (A 50 MB, B 4.7 MB, 6 rounds): S = 1 -> 276 MB
// (measured 294), S = 2 -> 119 MB.
// CUs the persistent big-tile launches may count on (default: all 256).  A data-parallel trainer that overlaps RCCL
// collectives with backward lowers it (multiples of 8: one CU fewer per XCD keeps the tile -> XCD map): a GEMM block needs a
// whole CU, so with a collective's workgroups resident a 256-block launch runs a second round for the blocks that found
// no CU (2x its time, tools/cu_share_probe.py); with the budget below the free CUs it stays one round.
static int g_cu_budget = []() {
    const char* e = getenv("SEDD_GEMM_CUS");
    const int v = e ? atoi(e) : 256;
    return v >= 8 && v <= 256 ? v / 8 * 8 : 256;
}();
static inline int cu_budget() { return g_cu_budget; }
extern "C" int sedd_gemm_set_cu_budget(int cus) {
    SEDD_CHECK_ARG(cus >= 8 && cus <= 256 && cus % 8 == 0, "gemm_set_cu_budget: %d is not a multiple of 8 in [8, 256]", cus);
    g_cu_budget = cus;
    return SEDD_OK;
}
extern "C" int sedd_gemm_get_cu_budget(void) { return g_cu_budget; }

static int choose_csplit(int64_t tm, int64_t tn, int64_t tiles, int64_t M, int64_t N, int64_t K) {
    static const char* ov = getenv("SEDD_GEMM_CSPLIT");
    if (tiles < 512 || tiles % 8) return 1;
    const double A = 2.0 * (double)M * (double)K, B = 2.0 * (double)N * (double)K, rounds = (double)tiles / cu_budget();
    int best = 1;
    double bc = 1e300;
    for (int S = 1; S <= 4; S *= 2) {
        if (tn % S || tm % (8 / S)) continue;
        if (ov && atoi(ov) != S) continue;
        const double bs = B / S, cost = S * A + (bs <= 2.5e6 ? 8.0 * bs : rounds * 8.0 * bs);
        if (cost < bc) { bc = cost; best = S; }
    }
    return best;
}

// split-K plan for accumulating fp32 outputs with too few tiles for the 256 CUs: as many K slices as keep
// tiles * slices <= 256 (one round of persistent blocks), every slice a whole number of K tiles, at least 8 of them
static void big_splitk(int64_t tiles, int64_t K, int* splitk, int64_t* kper) {
    const int64_t nkt = K / BIG_BK;
    static const int min_kt = getenv("SEDD_SPLITK_MIN_KT") ? atoi(getenv("SEDD_SPLITK_MIN_KT")) : 8;  // K tiles per slice at least
    // (A/B knob; 16 -> 8 in round 3: the dual-track model's weight gradients reduce over 2048 or 512 tokens only, cfg 3
    // bf16 batch 4 38.1 -> 36.7 ms; 4 is slower again — slab traffic; cfg 2 / 4 reduce over >= 16 k tokens: not binding)
    int64_t sk = tiles > 0 ? cu_budget() / tiles : 1;
    if (sk > nkt / min_kt) sk = nkt / min_kt;
    if (sk < 1) sk = 1;
    const int64_t per = (nkt + sk - 1) / sk;
    *splitk = (int)((nkt + per - 1) / per);
    *kper = per * BIG_BK;
}

static const RotArgs kNoRot{nullptr, nullptr, 1, 0};

// colsum[c] += sum over the row tiles t of ws[t][c] (the dGELU epilogue's per-row-tile column sums) in a FIXED order, no
// atomics: 64 columns x 4 contiguous ranges of row tiles per block, four independent running sums per thread (a serial
// loop over 128 dependent loads took 36 us), the ranges combined in order through LDS
__global__ void __launch_bounds__(256) colsum_fold_kernel(const float* __restrict__ ws, int ntm, int64_t N,
                                                          float* __restrict__ colsum) {
    __shared__ float part[4][64];
    const int cl = threadIdx.x & 63, p = threadIdx.x >> 6;
    const int64_t c = (int64_t)blockIdx.x * 64 + cl;
    const int per = (ntm + 3) / 4, t0 = p * per, t1 = t0 + per < ntm ? t0 + per : ntm;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < N) {
        int t = t0;
        for (; t + 4 <= t1; t += 4) {
            s0 += ws[(int64_t)t * N + c];
            s1 += ws[(int64_t)(t + 1) * N + c];
            s2 += ws[(int64_t)(t + 2) * N + c];
            s3 += ws[(int64_t)(t + 3) * N + c];
        }
        for (; t < t1; ++t) s0 += ws[(int64_t)t * N + c];
    }
    part[p][cl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (p == 0 && c < N) colsum[c] += (part[0][cl] + part[1][cl]) + (part[2][cl] + part[3][cl]);
}

// fp32 products.  layout 0: A [M,K], B [N,K] (NT)   1: A [M,K], B [K,N] (NN)   2: A [K,M], B [K,N] (TN).
// ws (optional; fp32 output, no epilogue or bias only): scratch for splitting K when even 64 x 64 tiles leave CUs idle
// (the fold adds the bias).
static int gemm_f32_launch(int layout, const float* A, int64_t lda, const float* B, int64_t ldb, void* C, int64_t ldc,
                           int out_dtype, const float* bias, int epi, void* pre, int accumulate, int64_t M, int64_t N,
                           int64_t K, float* ws, int64_t ws_bytes, hipStream_t st) {
    // tile shape: fewest chip-wide rounds x tile work, the smaller tiles priced a little above their flops
    static const int cand[3][2] = {{2, 2}, {2, 1}, {1, 1}};
    static const double price[3] = {1.0, 1.06, 1.15};
    int best = 0;
    double best_cost = 0;
    int64_t best_tiles = 0;
    for (int c = 0; c < 3; ++c) {
        const int64_t bm = cand[c][0] * 64, bn = cand[c][1] * 64;
        const int64_t t = ((M + bm - 1) / bm) * ((N + bn - 1) / bn);
        const double cost = (double)((t + 255) / 256) * (double)(bm * bn) * price[c];
        if (c == 0 || cost < best_cost) { best = c; best_cost = cost; best_tiles = t; }
    }
    static const int force = getenv("SEDD_F32_TILE") ? atoi(getenv("SEDD_F32_TILE")) : -1;  // tuning aid
    if (force >= 0 && force < 3) {
        best = force;
        best_tiles = ((M + cand[best][0] * 64 - 1) / (cand[best][0] * 64)) * ((N + cand[best][1] * 64 - 1) / (cand[best][1] * 64));
    }
    const int64_t bn = cand[best][1] * 64, ntn = (N + bn - 1) / bn;
    SEDD_CHECK_ARG(best_tiles < (1LL << 31), "gemm (fp32): too many tiles");
    int splitk = 1;
    // K step: 64 for the 64 x 64 tile (16 MFMAs per wave and step would not cover a global-load latency), else 32
    const int bk = best == 2 ? 64 : 32;
    int64_t kper = (K + bk - 1) / bk * bk;
    if (best == 2 && ws && out_dtype == SEDD_F32 && (epi == SEDD_EPI_NONE || epi == SEDD_EPI_BIAS)) {
        // every slice keeps >= 256 of K; slices x tiles stay within two blocks per CU
        while (best_tiles * splitk * 2 <= 512 && kper % (2 * bk) == 0 && kper / 2 >= 256 &&
               (int64_t)splitk * 2 * M * N * 4 <= ws_bytes)
            { splitk *= 2; kper /= 2; }
    }
    float* slabs = splitk > 1 ? ws : nullptr;
    const dim3 grid((unsigned)best_tiles, (unsigned)splitk);
#define F32_LAUNCH(TO, TM, TN, AKM, BKM, BK)                                                                          \
    hipLaunchKernelGGL((gemm_f32_kernel<TO, TM, TN, AKM, BKM, BK>), grid, dim3(256), 0, st, A, lda, B, ldb, (TO*)C, ldc,  \
                       bias, epi, (TO*)pre, accumulate, M, N, K, kper, (int)ntn, slabs)
#define F32_TILES(TO, AKM, BKM)                                                                                       \
    { if (best == 0) F32_LAUNCH(TO, 2, 2, AKM, BKM, 32); else if (best == 1) F32_LAUNCH(TO, 2, 1, AKM, BKM, 32);      \
      else F32_LAUNCH(TO, 1, 1, AKM, BKM, 64); }
    if (out_dtype == SEDD_BF16) {  // (fp32 operands, bf16 result: NT only)
        if (layout != 0) { sedd_set_error("gemm (fp32 operands): bf16 output only in the NT form"); return SEDD_ERR_UNSUPPORTED_SHAPE; }
        F32_TILES(bf16_t, false, false)
    } else if (layout == 0) F32_TILES(float, false, false)
    else if (layout == 1) F32_TILES(float, false, true)
    else F32_TILES(float, true, true)
#undef F32_TILES
#undef F32_LAUNCH
    SEDD_LAUNCH_CHECK();
    if (slabs) {
        hipLaunchKernelGGL(splitk_fold_kernel, dim3((unsigned)((M * N + 255) / 256)), dim3(256), 0, st, slabs, splitk, M,
                           N, (float*)C, ldc, accumulate, epi == SEDD_EPI_BIAS ? bias : (const float*)nullptr);
        SEDD_LAUNCH_CHECK();
    }
    return SEDD_OK;
}

static int gemm_nt_impl(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int in_dtype,
                        int out_dtype, const float* bias, int epilogue, void* pre, int accumulate, int64_t M, int64_t N,
                        int64_t K, float* colsum, sedd_stream_t stream, bool b_kn = false,
                        const RotArgs* rot = nullptr, float* colsum_ws = nullptr, float* sk_ws = nullptr,
                        int64_t sk_ws_bytes = 0) {
    SEDD_CHECK_ARG((in_dtype == SEDD_F32 || in_dtype == SEDD_BF16) && (out_dtype == SEDD_F32 || out_dtype == SEDD_BF16),
                   "gemm_nt: bad dtype");
    if (b_kn) {  // B given as [K, N] (NN form): bf16 big-tile kernel only
        if (in_dtype != SEDD_BF16 || out_dtype != SEDD_BF16 || accumulate || K % BIG_BK || M < 192 || N < 96 || N % 8 ||
            ldb % 8 || lda % 8 || ldb < N || lda < K || ldc < N || ((uintptr_t)A & 15) || ((uintptr_t)B & 15)) {
            sedd_set_error("gemm_nn: unsupported operands (bf16 in/out; M >= 192, N >= 96, K %% 32, N %% 8, lda %% 8, "
                           "ldb %% 8 must be 0, 16-byte aligned bases): M=%ld N=%ld K=%ld lda=%ld ldb=%ld", (long)M,
                           (long)N, (long)K, (long)lda, (long)ldb);
            return SEDD_ERR_UNSUPPORTED_SHAPE;
        }
    }
    const bool dg = epilogue == EPI_DGELU || epilogue == EPI_DMUL;
    SEDD_CHECK_ARG((epilogue >= SEDD_EPI_NONE && epilogue <= SEDD_EPI_BIAS_GELU_GRAD) || dg, "gemm_nt: bad epilogue %d", epilogue);
    SEDD_CHECK_ARG(M >= 0 && N >= 0 && K > 0, "gemm_nt: bad sizes M=%ld N=%ld K=%ld", (long)M, (long)N, (long)K);
    SEDD_CHECK_ARG(epilogue == SEDD_EPI_NONE || dg || bias, "gemm_nt: epilogue needs a bias");
    SEDD_CHECK_ARG(!accumulate || epilogue == SEDD_EPI_NONE, "gemm_nt: accumulate only with SEDD_EPI_NONE");
    const int kal = in_dtype == SEDD_BF16 ? 8 : 4;
    if (K % kal || lda % kal || ldb % kal || ((uintptr_t)A & 15) || ((uintptr_t)B & 15) || lda < K || (!b_kn && ldb < K) ||
        ldc < N) {
        sedd_set_error("gemm_nt: K=%ld lda=%ld ldb=%ld ldc=%ld unsupported (K, lda, ldb must be multiples of %d, "
                       "16-byte aligned bases)", (long)K, (long)lda, (long)ldb, (long)ldc, kal);
        return SEDD_ERR_UNSUPPORTED_SHAPE;
    }
    if (M == 0 || N == 0) return SEDD_OK;
    SEDD_CHECK_ARG(A && B && C, "gemm_nt: null pointer");
    const int64_t ntm = (M + BM - 1) / BM, ntn = (N + BN - 1) / BN;
    SEDD_CHECK_ARG(ntm * ntn < (1LL << 31), "gemm_nt: too many tiles");
    const dim3 grid((unsigned)(ntm * ntn));
    hipStream_t st = (hipStream_t)stream;
    if (in_dtype == SEDD_BF16) {
        // big-tile path: 256 x 256 or 256 x 128 tiles, whichever quantises better onto the 256 CUs
        const bool big_ok = K % BIG_BK == 0 && M >= 192 && N >= 96 && (b_kn || !getenv("SEDD_GEMM_SMALL_TILES"));
        if (dg) {
            // only the big-tile kernel's staged bf16 epilogue implements it
            if (!big_ok || out_dtype != SEDD_BF16 || accumulate || (ldc * 2) % 16 || ((uintptr_t)C & 15) ||
                ((uintptr_t)pre & 15) || N % 8) {
                sedd_set_error("gemm_nt_dgelu: unsupported shape/alignment (bf16, M >= 192, N >= 96, N %% 8 == 0, "
                               "K %% 32 == 0, 16-byte aligned C / pre rows)");
                return SEDD_ERR_UNSUPPORTED_SHAPE;
            }
        }
        if (big_ok) {
            const int64_t tm = (M + BIG_BM - 1) / BIG_BM;
            const int64_t t256 = tm * ((N + 255) / 256), t128 = tm * ((N + 127) / 128);
            // cost ~ (number of chip-wide rounds) x (tile work); the 256-wide tile does twice the work per round, at
            // ~13 % better efficiency per flop (measured: N = 2304 prefers 4.5 rounds of 256 over 9 rounds of 128)
            const int cus = cu_budget();
            const double c256 = (double)((t256 + cus - 1) / cus) * 2.0, c128 = (double)((t128 + cus - 1) / cus) * 1.15;
            bool wide = N > 128 && c256 <= c128;
            if (getenv("SEDD_GEMM_WIDE") && N > 128) wide = atoi(getenv("SEDD_GEMM_WIDE")) != 0;  // (A/B experiments)
            // rotary epilogue: the 128-wide tile has the registers to keep two passes of cos / sin rows in flight (the
            // 256-wide one spills with even one); same-box A/B at N = 2304: equal step time, so the narrow tile it is
            if (rot) wide = false;
            // 128-row tiles when even the narrow 256-row tiles would fill at most half of the CUs (plain forward epilogues,
            // bf16 out): twice the blocks, each with half the rows — the dual-track model's [2048 | 512, 1024] products
            static const bool no_bm128 = getenv("SEDD_GEMM_NO_BM128") != nullptr;  // (A/B knob)
            const bool bm128 = !no_bm128 && !dg && !rot && !accumulate && out_dtype == SEDD_BF16 &&
                               epilogue != SEDD_EPI_BIAS_GELU_GRAD && M >= 128 && t128 * 2 <= cus;
            if (bm128) wide = false;
            const int bn = wide ? 256 : 128;
            int64_t tn = (N + bn - 1) / bn;
            const int64_t tiles = (bm128 ? (M + 127) / 128 : tm) * tn;
            int64_t kper = K;
            int splitk = 1;
            if (accumulate && out_dtype == SEDD_F32) big_splitk(tiles, K, &splitk, &kper);
            SEDD_CHECK_ARG(tiles < (1LL << 30), "gemm_nt: too many tiles");
            // one persistent block per CU (the operand ring takes most of a CU's LDS)
            const dim3 g2((unsigned)(splitk > 1 ? tiles * splitk : (tiles < cus ? tiles : cus)));
            if (splitk == 1) tn |= (int64_t)choose_csplit(tm, tn, tiles, M, N, K) << 20;  // (tn is only passed on from here)
#define BIG_LAUNCH(TO, BN_, NST)                                                                                     \
    hipLaunchKernelGGL((gemm_bf16_big_kernel<TO, BN_, NST, false, 0>), g2, dim3(512), 0, st, (const bf16_t*)A, lda,     \
                       (const bf16_t*)B, ldb, (TO*)C, ldc, bias, epilogue, (TO*)pre, accumulate, M, N, kper, K,      \
                       (int)tn, (int)tiles, splitk, (float4*)nullptr, colsum, kNoRot)
#define BIG_LAUNCH_X(BN_, DG_, NNB_)                                                                                 \
    hipLaunchKernelGGL((gemm_bf16_big_kernel<bf16_t, BN_, 4, false, DG_, 8, NNB_>), g2, dim3(512), 0, st,            \
                       (const bf16_t*)A, lda, (const bf16_t*)B, ldb, (bf16_t*)C, ldc, bias, epilogue, (bf16_t*)pre,   \
                       accumulate, M, N, kper, K, (int)tn, (int)tiles, splitk,                                        \
                       (float4*)(DG_ == 1 || DG_ == 2 ? colsum_ws : nullptr), colsum, kNoRot)
            if (bm128) {
                if (b_kn)
                    hipLaunchKernelGGL((gemm_bf16_big_kernel<bf16_t, 128, 4, false, 0, 8, true, false, 128>), g2, dim3(512), 0,
                                       st, (const bf16_t*)A, lda, (const bf16_t*)B, ldb, (bf16_t*)C, ldc, bias, epilogue,
                                       (bf16_t*)pre, accumulate, M, N, kper, K, (int)tn, (int)tiles, splitk,
                                       (float4*)nullptr, colsum, kNoRot);
                else
                    hipLaunchKernelGGL((gemm_bf16_big_kernel<bf16_t, 128, 4, false, 0, 8, false, false, 128>), g2, dim3(512), 0,
                                       st, (const bf16_t*)A, lda, (const bf16_t*)B, ldb, (bf16_t*)C, ldc, bias, epilogue,
                                       (bf16_t*)pre, accumulate, M, N, kper, K, (int)tn, (int)tiles, splitk,
                                       (float4*)nullptr, colsum, kNoRot);
            } else if (rot) {
                    hipLaunchKernelGGL((gemm_bf16_big_kernel<bf16_t, 128, 4, false, 0, 8, false, true>), g2, dim3(512), 0,
                                       st, (const bf16_t*)A, lda, (const bf16_t*)B, ldb, (bf16_t*)C, ldc, bias, epilogue,
                                       (bf16_t*)pre, accumulate, M, N, kper, K, (int)tn, (int)tiles, splitk,
                                       (float4*)nullptr, colsum, *rot);
            } else if (dg) {
                const int dgk = epilogue == EPI_DMUL ? 2 : 1;
                if (b_kn) {
                    if (dgk == 2) { if (wide) BIG_LAUNCH_X(256, 2, true); else BIG_LAUNCH_X(128, 2, true); }
                    else { if (wide) BIG_LAUNCH_X(256, 1, true); else BIG_LAUNCH_X(128, 1, true); }
                } else {
                    if (dgk == 2) { if (wide) BIG_LAUNCH_X(256, 2, false); else BIG_LAUNCH_X(128, 2, false); }
                    else { if (wide) BIG_LAUNCH_X(256, 1, false); else BIG_LAUNCH_X(128, 1, false); }
                }
            } else if (b_kn) {
                if (wide) BIG_LAUNCH_X(256, 0, true); else BIG_LAUNCH_X(128, 0, true);
            } else if (out_dtype == SEDD_BF16 && epilogue == SEDD_EPI_BIAS_GELU_GRAD) {
                if (wide) BIG_LAUNCH_X(256, 3, false); else BIG_LAUNCH_X(128, 3, false);
            } else if (out_dtype == SEDD_BF16) {
                static const int nw4 = getenv("SEDD_GEMM_NW4") ? atoi(getenv("SEDD_GEMM_NW4")) : 0;
                if (nw4 && splitk == 1 && (nw4 == 2 || !wide)) {
                    // two 256-thread blocks per CU, 256 x 128 tiles
                    const int64_t tn4 = (N + 127) / 128, tiles4 = tm * tn4;
                    const dim3 g4((unsigned)(tiles4 < 512 ? tiles4 : 512));
                    hipLaunchKernelGGL((gemm_bf16_big_kernel<bf16_t, 128, 3, false, 0, 4>), g4, dim3(256), 0, st,
                                       (const bf16_t*)A, lda, (const bf16_t*)B, ldb, (bf16_t*)C, ldc, bias, epilogue,
                                       (bf16_t*)pre, accumulate, M, N, kper, K, (int)tn4, (int)tiles4, 1,
                                       (float4*)nullptr, colsum, kNoRot);
                } else if (wide) BIG_LAUNCH(bf16_t, 256, 4);
                else BIG_LAUNCH(bf16_t, 128, 4);
            } else {
                if (wide) BIG_LAUNCH(float, 256, 4);
                else BIG_LAUNCH(float, 128, 6);
            }
#undef BIG_LAUNCH
#undef BIG_LAUNCH_X
            if (dg && colsum && colsum_ws) {
                SEDD_LAUNCH_CHECK();
                hipLaunchKernelGGL(colsum_fold_kernel, dim3((unsigned)((N + 63) / 64)), dim3(256), 0, st, colsum_ws,
                                   (int)tm, N, colsum);
            }
        } else if (K % BK16 == 0) {
            // split K across blockIdx.y for accumulating fp32 outputs with too few tiles to fill 256 CUs
            int splitk = 1;
            if (accumulate && out_dtype == SEDD_F32) {
                const int64_t nkt = K / BK16;
                while (ntm * ntn * splitk < 512 && nkt % (splitk * 2) == 0 && nkt / (splitk * 2) >= 8) splitk *= 2;
            }
            const dim3 g2((unsigned)(ntm * ntn), (unsigned)splitk);
            if (out_dtype == SEDD_BF16)
                hipLaunchKernelGGL((gemm_bf16_nt_glds_kernel<bf16_t>), g2, dim3(256), 0, st, (const bf16_t*)A, lda,
                                   (const bf16_t*)B, ldb, (bf16_t*)C, ldc, bias, epilogue, (bf16_t*)pre, accumulate,
                                   M, N, K / splitk, (int)ntn);
            else {
                // with a large enough workspace the K slices park their partial tiles and a fixed-order fold adds them
                // to C (bit-reproducible); without one they add to C with fp32 atomics
                float* slabs = splitk > 1 && sk_ws && sk_ws_bytes >= (int64_t)splitk * M * N * 4 ? sk_ws : nullptr;
                hipLaunchKernelGGL((gemm_bf16_nt_glds_kernel<float>), g2, dim3(256), 0, st, (const bf16_t*)A, lda,
                                   (const bf16_t*)B, ldb, (float*)C, ldc, bias, epilogue, (float*)pre, accumulate, M,
                                   N, K / splitk, (int)ntn, slabs);
                if (slabs) {
                    SEDD_LAUNCH_CHECK();
                    hipLaunchKernelGGL(splitk_fold_kernel, dim3((unsigned)((M * N + 255) / 256)), dim3(256), 0, st, slabs,
                                       splitk, M, N, (float*)C, ldc);
                }
            }
        } else if (out_dtype == SEDD_BF16)
            hipLaunchKernelGGL((gemm_bf16_nt_kernel<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)A, lda,
                               (const bf16_t*)B, ldb, (bf16_t*)C, ldc, bias, epilogue, (bf16_t*)pre, accumulate, M, N,
                               K, (int)ntn);
        else
            hipLaunchKernelGGL((gemm_bf16_nt_kernel<float>), grid, dim3(256), 0, st, (const bf16_t*)A, lda,
                               (const bf16_t*)B, ldb, (float*)C, ldc, bias, epilogue, (float*)pre, accumulate, M, N, K,
                               (int)ntn);
    } else {
        return gemm_f32_launch(0, (const float*)A, lda, (const float*)B, ldb, C, ldc, out_dtype, bias, epilogue, pre,
                               accumulate, M, N, K, sk_ws, sk_ws_bytes, st);
    }
    SEDD_LAUNCH_CHECK();
    return SEDD_OK;
}

extern "C" int sedd_gemm_nt(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                            int in_dtype, int out_dtype, const float* bias, int epilogue, void* pre, int accumulate,
                            int64_t M, int64_t N, int64_t K, sedd_stream_t stream) {
    SEDD_CHECK_ARG(epilogue >= SEDD_EPI_NONE && epilogue <= SEDD_EPI_BIAS_GELU_GRAD, "gemm_nt: bad epilogue %d", epilogue);
    return gemm_nt_impl(A, lda, B, ldb, C, ldc, in_dtype, out_dtype, bias, epilogue, pre, accumulate, M, N, K, nullptr,
                        stream);
}

// sedd_gemm_nt with a caller-owned workspace for the launches that split K over blocks (accumulating fp32 outputs of
// small shapes: one or a few output tiles and a long reduction)
extern "C" int sedd_gemm_nt_ws(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                               int in_dtype, int out_dtype, const float* bias, int epilogue, void* pre, int accumulate,
                               int64_t M, int64_t N, int64_t K, void* workspace, int64_t workspace_bytes,
                               sedd_stream_t stream) {
    SEDD_CHECK_ARG(epilogue >= SEDD_EPI_NONE && epilogue <= SEDD_EPI_BIAS_GELU_GRAD, "gemm_nt_ws: bad epilogue %d", epilogue);
    SEDD_CHECK_ARG(workspace_bytes >= 0 && (!workspace || ((uintptr_t)workspace & 15) == 0), "gemm_nt_ws: bad workspace");
    return gemm_nt_impl(A, lda, B, ldb, C, ldc, in_dtype, out_dtype, bias, epilogue, pre, accumulate, M, N, K, nullptr,
                        stream, false, nullptr, nullptr, (float*)workspace, workspace_bytes);
}

// C[M,N] = (A[M,K] * B[N,K]^T) .* gelu_tanh'(pre[M,N]); colsum[N] += column sums of C (optional).  The data gradient of
// Linear -> GELU in one launch (autograd of nn.Sequential(Linear, GELU(tanh), Linear), transformer.py:129-133).
extern "C" int sedd_gemm_nt_dgelu(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                                  int in_dtype, int out_dtype, const void* pre, float* colsum, float* colsum_ws,
                                  int64_t M, int64_t N, int64_t K, sedd_stream_t stream) {
    SEDD_CHECK_ARG(pre, "gemm_nt_dgelu: pre is null");
    return gemm_nt_impl(A, lda, B, ldb, C, ldc, in_dtype, out_dtype, nullptr, EPI_DGELU, const_cast<void*>(pre), 0, M, N,
                        K, colsum, stream, false, nullptr, colsum_ws);
}

// C[M,N] = (A[M,K] * B[N,K]^T) .* mul[M,N]; colsum[N] += column sums of C (optional): sedd_gemm_nt_dgelu with the
// derivative already evaluated by the forward pass (sedd_gemm_nt's SEDD_EPI_BIAS_GELU_GRAD writes gelu'(pre-activation)
// where SEDD_EPI_BIAS_GELU writes the pre-activation) — the epilogue is a multiply instead of a sigmoid per element.
extern "C" int sedd_gemm_nt_dmul(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                                 int in_dtype, int out_dtype, const void* mul, float* colsum, float* colsum_ws,
                                 int64_t M, int64_t N, int64_t K, sedd_stream_t stream) {
    SEDD_CHECK_ARG(mul, "gemm_nt_dmul: mul is null");
    return gemm_nt_impl(A, lda, B, ldb, C, ldc, in_dtype, out_dtype, nullptr, EPI_DMUL, const_cast<void*>(mul), 0, M, N,
                        K, colsum, stream, false, nullptr, colsum_ws);
}

// The QKV projection with the rotary position embedding applied in the epilogue (transformer.py:165-171: attn_qkv,
// rearrange to [b, s, 3, h, d], rotary.apply_rotary_pos_emb): C = packed qkv [M, 3*H*64] bf16, rows are tokens of
// sequences of length L (position = row % L), cos / sin fp32 [L, 32].  The V columns pass through.
extern "C" int sedd_gemm_nt_rotary(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                                   int in_dtype, int out_dtype, const float* bias, int epilogue, const float* cos,
                                   const float* sin, int L, int H, int d, int64_t M, int64_t N, int64_t K,
                                   sedd_stream_t stream) {
    SEDD_CHECK_ARG(epilogue == SEDD_EPI_NONE || epilogue == SEDD_EPI_BIAS, "gemm_nt_rotary: bad epilogue %d", epilogue);
    SEDD_CHECK_ARG(cos && sin && L > 0 && H > 0, "gemm_nt_rotary: bad rotary arguments");
    if (in_dtype != SEDD_BF16 || out_dtype != SEDD_BF16 || d != 64 || N != (int64_t)3 * H * d || K % BIG_BK || M < 192 ||
        M % L || (ldc * 2) % 16 || ((uintptr_t)C & 15)) {
        sedd_set_error("gemm_nt_rotary: unsupported (bf16, head dim 64, N == 3*H*64, K %% 32 == 0, M >= 192, M %% L == 0, "
                       "16-byte aligned C rows); use sedd_gemm_nt + sedd_rotary_qkv");
        return SEDD_ERR_UNSUPPORTED_SHAPE;
    }
    const RotArgs rot{cos, sin, L, (int64_t)2 * H * d};
    return gemm_nt_impl(A, lda, B, ldb, C, ldc, in_dtype, out_dtype, bias, epilogue, nullptr, 0, M, N, K, nullptr, stream,
                        false, &rot);
}

// C[M,N] = A[M,K] * B[K,N] (bf16): the data gradient of nn.Linear against the weight as stored (dX = dY W, W [out, in]),
// no transposed weight copy.  epilogue: SEDD_EPI_NONE or SEDD_EPI_BIAS.
static int gemm_nn_impl(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int in_dtype,
                        int out_dtype, const float* bias, int epilogue, int64_t M, int64_t N, int64_t K, void* workspace,
                        int64_t workspace_bytes, sedd_stream_t stream) {
    SEDD_CHECK_ARG(epilogue == SEDD_EPI_NONE || epilogue == SEDD_EPI_BIAS, "gemm_nn: bad epilogue %d", epilogue);
    if (in_dtype == SEDD_F32) {  // exact-fp32 MFMA kernel reading B as stored
        SEDD_CHECK_ARG(M >= 0 && N >= 0 && K > 0, "gemm_nn: bad sizes M=%ld N=%ld K=%ld", (long)M, (long)N, (long)K);
        SEDD_CHECK_ARG(epilogue == SEDD_EPI_NONE || bias, "gemm_nn: epilogue needs a bias");
        if (out_dtype != SEDD_F32 || K % 4 || N % 4 || lda % 4 || ldb % 4 || ((uintptr_t)A & 15) || ((uintptr_t)B & 15) ||
            lda < K || ldb < N || ldc < N) {
            sedd_set_error("gemm_nn (fp32): unsupported operands (fp32 out; K %% 4, N %% 4, lda %% 4, ldb %% 4 must be 0, "
                           "16-byte aligned bases): M=%ld N=%ld K=%ld lda=%ld ldb=%ld", (long)M, (long)N, (long)K,
                           (long)lda, (long)ldb);
            return SEDD_ERR_UNSUPPORTED_SHAPE;
        }
        if (M == 0 || N == 0) return SEDD_OK;
        SEDD_CHECK_ARG(A && B && C, "gemm_nn: null pointer");
        const bool ws_ok = workspace && ((uintptr_t)workspace & 15) == 0;
        return gemm_f32_launch(1, (const float*)A, lda, (const float*)B, ldb, C, ldc, out_dtype, bias, epilogue, nullptr,
                               0, M, N, K, ws_ok ? (float*)workspace : nullptr, ws_ok ? workspace_bytes : 0,
                               (hipStream_t)stream);
    }
    return gemm_nt_impl(A, lda, B, ldb, C, ldc, in_dtype, out_dtype, bias, epilogue, nullptr, 0, M, N, K, nullptr, stream,
                        true);
}

extern "C" int sedd_gemm_nn(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int in_dtype,
                            int out_dtype, const float* bias, int epilogue, int64_t M, int64_t N, int64_t K,
                            sedd_stream_t stream) {
    return gemm_nn_impl(A, lda, B, ldb, C, ldc, in_dtype, out_dtype, bias, epilogue, M, N, K, nullptr, 0, stream);
}

// sedd_gemm_nn with a caller-owned workspace: fp32 products whose output has fewer than 128 tiles of 64 x 64 split K over
// blocks and fold the slices in a fixed order (bf16 products ignore it)
extern "C" int sedd_gemm_nn_ws(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int in_dtype,
                               int out_dtype, const float* bias, int epilogue, int64_t M, int64_t N, int64_t K,
                               void* workspace, int64_t workspace_bytes, sedd_stream_t stream) {
    SEDD_CHECK_ARG(workspace_bytes >= 0, "gemm_nn_ws: bad workspace");
    return gemm_nn_impl(A, lda, B, ldb, C, ldc, in_dtype, out_dtype, bias, epilogue, M, N, K, workspace, workspace_bytes,
                        stream);
}

// sedd_gemm_nt_dgelu with B given as [K, N]: dU = (dY W2) .* gelu'(U), colsum += column sums, W2 [out, hidden] as stored
extern "C" int sedd_gemm_nn_dgelu(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                                  int in_dtype, int out_dtype, const void* pre, float* colsum, float* colsum_ws,
                                  int64_t M, int64_t N, int64_t K, sedd_stream_t stream) {
    SEDD_CHECK_ARG(pre, "gemm_nn_dgelu: pre is null");
    return gemm_nt_impl(A, lda, B, ldb, C, ldc, in_dtype, out_dtype, nullptr, EPI_DGELU, const_cast<void*>(pre), 0, M, N,
                        K, colsum, stream, true, nullptr, colsum_ws);
}

// sedd_gemm_nt_dmul with B given as [K, N]
extern "C" int sedd_gemm_nn_dmul(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                                 int in_dtype, int out_dtype, const void* mul, float* colsum, float* colsum_ws,
                                 int64_t M, int64_t N, int64_t K, sedd_stream_t stream) {
    SEDD_CHECK_ARG(mul, "gemm_nn_dmul: mul is null");
    return gemm_nt_impl(A, lda, B, ldb, C, ldc, in_dtype, out_dtype, nullptr, EPI_DMUL, const_cast<void*>(mul), 0, M, N,
                        K, colsum, stream, true, nullptr, colsum_ws);
}

// C[M,N] (+)= A[K,M]^T * B[K,N]: weight gradient of nn.Linear straight from the row-major activations
// (dW[out,in] = dY[tokens,out]^T X[tokens,in]), no transposed copies.  bf16 operands, fp32 or bf16 C.
// colsum (optional; bf16 operands, fp32 C): colsum[M] (+)= sum_k A[k, m] out of the same launch — the Linear's bias
// gradient next to its weight gradient (sedd_gemm_tn_colsum)
static int gemm_tn_impl(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int in_dtype,
                        int out_dtype, int accumulate, int64_t M, int64_t N, int64_t K, void* workspace,
                        int64_t workspace_bytes, float* colsum, sedd_stream_t stream) {
    SEDD_CHECK_ARG(out_dtype == SEDD_F32 || out_dtype == SEDD_BF16, "gemm_tn: bad dtype");
    SEDD_CHECK_ARG(M >= 0 && N >= 0 && K > 0, "gemm_tn: bad sizes M=%ld N=%ld K=%ld", (long)M, (long)N, (long)K);
    SEDD_CHECK_ARG(!colsum || (in_dtype == SEDD_BF16 && out_dtype == SEDD_F32),
                   "gemm_tn_colsum: column sums ride on the bf16 -> fp32 form only");
    if (in_dtype == SEDD_F32) {  // exact-fp32 MFMA kernel, both operands read k-major as stored; any K
        if (out_dtype != SEDD_F32 || M % 4 || N % 4 || lda % 4 || ldb % 4 || ((uintptr_t)A & 15) || ((uintptr_t)B & 15) ||
            lda < M || ldb < N || ldc < N) {
            sedd_set_error("gemm_tn (fp32): unsupported operands (fp32 out; M %% 4, N %% 4, lda %% 4, ldb %% 4 must be 0, "
                           "16-byte aligned bases): M=%ld N=%ld K=%ld lda=%ld ldb=%ld", (long)M, (long)N, (long)K,
                           (long)lda, (long)ldb);
            return SEDD_ERR_UNSUPPORTED_SHAPE;
        }
        if (M == 0 || N == 0) return SEDD_OK;
        SEDD_CHECK_ARG(A && B && C, "gemm_tn: null pointer");
        const bool ws_ok = workspace && ((uintptr_t)workspace & 15) == 0;
        return gemm_f32_launch(2, (const float*)A, lda, (const float*)B, ldb, C, ldc, out_dtype, nullptr, SEDD_EPI_NONE,
                               nullptr, accumulate, M, N, K, ws_ok ? (float*)workspace : nullptr,
                               ws_ok ? workspace_bytes : 0, (hipStream_t)stream);
    }
    if (in_dtype != SEDD_BF16 || K % BIG_BK || M % 8 || N % 8 || lda % 8 || ldb % 8 || ((uintptr_t)A & 15) ||
        ((uintptr_t)B & 15) || lda < M || ldb < N || ldc < N) {
        sedd_set_error("gemm_tn: unsupported operands (bf16 only; K %% 32, M %% 8, N %% 8, lda %% 8, ldb %% 8 must be 0, "
                       "16-byte aligned bases): M=%ld N=%ld K=%ld lda=%ld ldb=%ld", (long)M, (long)N, (long)K, (long)lda,
                       (long)ldb);
        return SEDD_ERR_UNSUPPORTED_SHAPE;
    }
    if (M == 0 || N == 0) return SEDD_OK;
    SEDD_CHECK_ARG(A && B && C, "gemm_tn: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const int64_t tm = (M + BIG_BM - 1) / BIG_BM;
    const int64_t t256 = tm * ((N + 255) / 256), t128 = tm * ((N + 127) / 128);
    // few, long tiles: prefer the tile width that leaves the finer split-K granularity
    // A short reduction (the dual-track model at batch 4 reduces over 2048 or 512 tokens) is not split at all: 128-wide
    // tiles, one block per tile walking all of K and adding into C directly.  The same GPU time as split + second pass at
    // these sizes (1024 x 1024 over 2048 tokens: 17 + 10 us -> 27 us) and one launch fewer, which is what counts where
    // the step is bound by the host's enqueue rate.  SEDD_TN_NOSPLIT_KT=0 restores the split.
    static const int nosplit_kt = getenv("SEDD_TN_NOSPLIT_KT") ? atoi(getenv("SEDD_TN_NOSPLIT_KT")) : 64;
    const bool nosplit = out_dtype == SEDD_F32 && K / BIG_BK <= nosplit_kt && t128 >= 16;
    // (the 256-wide tile sits at 256 VGPRs: with the column sums' accumulator and selector it spills 71 -> 128-wide then)
    static const int force_wide = getenv("SEDD_TN_WIDE") ? atoi(getenv("SEDD_TN_WIDE")) : -1;  // A/B knob
    const bool wide = force_wide >= 0 ? (force_wide != 0 && N > 128 && !colsum)
                                      : !nosplit && !colsum && N > 128 && (t256 >= 128 || t128 > 256 || N % 256 == 0);
    const int bn = wide ? 256 : 128;
    const int64_t tn = (N + bn - 1) / bn, tiles = tm * tn;
    SEDD_CHECK_ARG(tiles < (1LL << 30), "gemm_tn: too many tiles");
    int64_t kper = K;
    int splitk = 1;
    // K is split when the output has too few tiles for the chip.  With a workspace the slices park their partial
    // tiles there and a second pass sums them (any fp32 C); without one they add into C with fp32 atomics, which
    // needs an accumulating call (C already holds valid numbers) and is several times slower at 7+ adders per address.
    const bool can_split = out_dtype == SEDD_F32 && (accumulate || workspace);
    if (can_split && !nosplit) big_splitk(tiles, K, &splitk, &kper);
    float4* slabs = nullptr;
    if (splitk > 1) {
        const int64_t need = (int64_t)splitk * tiles * BIG_BM * bn * 4 + (colsum ? (int64_t)splitk * tm * BIG_BM * 4 : 0);
        if (workspace && workspace_bytes >= need && ((uintptr_t)workspace & 15) == 0) slabs = (float4*)workspace;
        else if (!accumulate) { splitk = 1; kper = K; }
    }
    const dim3 g2((unsigned)(splitk > 1 ? tiles * splitk : (tiles < cu_budget() ? tiles : cu_budget())));
#define TN_LAUNCH(TO, BN_, DG_)                                                                                      \
    hipLaunchKernelGGL((gemm_bf16_big_kernel<TO, BN_, 4, true, DG_>), g2, dim3(512), 0, st, (const bf16_t*)A, lda,      \
                       (const bf16_t*)B, ldb, (TO*)C, ldc, (const float*)nullptr, (int)SEDD_EPI_NONE, (TO*)nullptr,  \
                       accumulate, M, N, kper, K, (int)tn, (int)tiles, splitk, slabs, colsum, kNoRot)
    if (out_dtype == SEDD_BF16) {
        if (wide) TN_LAUNCH(bf16_t, 256, 0);
        else TN_LAUNCH(bf16_t, 128, 0);
    } else if (colsum) {
        TN_LAUNCH(float, 128, 4);
    } else {
        if (wide) TN_LAUNCH(float, 256, 0);
        else TN_LAUNCH(float, 128, 0);
    }
#undef TN_LAUNCH
    SEDD_LAUNCH_CHECK();
    if (slabs) {
        const int64_t n4 = tiles * BIG_BM * bn / 4;  // (a multiple of 256: the column sums' threads start a new block)
        const dim3 gr((unsigned)((n4 + (colsum ? tm * BIG_BM : 0) + 255) / 256));
        if (wide)
            hipLaunchKernelGGL((splitk_reduce_kernel<256>), gr, dim3(256), 0, st, (const float4*)slabs, (float*)C, ldc, M,
                               N, (int)tn, (int)tiles, splitk, accumulate, colsum);
        else
            hipLaunchKernelGGL((splitk_reduce_kernel<128>), gr, dim3(256), 0, st, (const float4*)slabs, (float*)C, ldc, M,
                               N, (int)tn, (int)tiles, splitk, accumulate, colsum);
        SEDD_LAUNCH_CHECK();
    }
    return SEDD_OK;
}

extern "C" int sedd_gemm_tn(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int in_dtype,
                            int out_dtype, int accumulate, int64_t M, int64_t N, int64_t K, void* workspace,
                            int64_t workspace_bytes, sedd_stream_t stream) {
    return gemm_tn_impl(A, lda, B, ldb, C, ldc, in_dtype, out_dtype, accumulate, M, N, K, workspace, workspace_bytes,
                        nullptr, stream);
}

extern "C" int sedd_gemm_tn_colsum(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                                   void* colsum, int in_dtype, int accumulate, int64_t M, int64_t N, int64_t K,
                                   void* workspace, int64_t workspace_bytes, sedd_stream_t stream) {
    SEDD_CHECK_ARG(colsum != nullptr, "gemm_tn_colsum: null colsum");
    return gemm_tn_impl(A, lda, B, ldb, C, ldc, in_dtype, SEDD_F32, accumulate, M, N, K, workspace, workspace_bytes,
                        (float*)colsum, stream);
}
