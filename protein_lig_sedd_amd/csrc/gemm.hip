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

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define BM 128
#define BN 128
#define BK16 64

__device__ __forceinline__ float gelu_tanh_f(float u) {
    const float t = tanhf(0.7978845608028654f * (u + 0.044715f * u * u * u));
    return 0.5f * u * (1.0f + t);
}

// bijective XCD-aware remap (8 XCDs): blocks that share an XCD get a contiguous range of tiles
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
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
template <typename TO>
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
    if (epi == SEDD_EPI_BIAS_GELU) {
        // the reference rounds the Linear output to the activation dtype before GELU
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = to_f32<TO>(from_f32<TO>(v[r]));
        if (pre) {
            TO* pp = pre + row * ldc + col;
            if (full) store_row4<TO>(pp, v);
            else
                for (int r = 0; r < 4 && col + r < N; ++r) pp[r] = from_f32<TO>(v[r]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_tanh_f(v[r]);
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
                                                                int64_t N, int64_t Kper, int ntn) {
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
    store_tile<TO>(C, ldc, pre, bias, epi, accumulate, gridDim.y > 1, m0, n0, wr, wc, lane, M, N, acc);
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
#define BK32 16
#define LDK32 17  // padded row (floats): conflict-free ds_read_b32 column reads
template <typename TO>
__device__ __forceinline__ void epilogue1(TO* __restrict__ C, int64_t ldc, TO* __restrict__ pre,
                                          const float* __restrict__ bias, int epi, int accumulate, int64_t row,
                                          int64_t col, int64_t M, int64_t N, float v) {
    if (row >= M || col >= N) return;
    if (epi != SEDD_EPI_NONE) v += bias[col];
    if (epi == SEDD_EPI_BIAS_GELU) {
        const TO r = from_f32<TO>(v);
        if (pre) pre[row * ldc + col] = r;
        v = gelu_tanh_f(to_f32<TO>(r));
    }
    TO* p = C + row * ldc + col;
    if (accumulate) v += to_f32<TO>(*p);
    *p = from_f32<TO>(v);
}

template <typename TO>
__global__ void __launch_bounds__(256) gemm_f32_nt_kernel(const float* __restrict__ A, int64_t lda,
                                                          const float* __restrict__ B, int64_t ldb,
                                                          TO* __restrict__ C, int64_t ldc,
                                                          const float* __restrict__ bias, int epi,
                                                          TO* __restrict__ pre, int accumulate, int64_t M, int64_t N,
                                                          int64_t K, int ntn) {
    __shared__ float sA[BM * LDK32];
    __shared__ float sB[BN * LDK32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int64_t m0 = (int64_t)(wg / ntn) * BM, n0 = (int64_t)(wg % ntn) * BN;

    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    const int64_t nkt = (K + BK32 - 1) / BK32;
    for (int64_t kt = 0; kt < nkt; ++kt) {
        float4 ra0, ra1, rb0, rb1;
#define GLOAD(i, ra, rb)                                                                 \
        {                                                                                \
            const int chunk = tid + 256 * (i), row = chunk >> 2, kc = chunk & 3;         \
            const int64_t k = kt * BK32 + kc * 4;                                        \
            int64_t ar = m0 + row, br = n0 + row;                                        \
            ar = ar < M ? ar : M - 1;                                                    \
            br = br < N ? br : N - 1;                                                    \
            ra = make_float4(0.f, 0.f, 0.f, 0.f);                                        \
            rb = make_float4(0.f, 0.f, 0.f, 0.f);                                        \
            if (k < K) {                                                                 \
                ra = *reinterpret_cast<const float4*>(A + ar * lda + k);                 \
                rb = *reinterpret_cast<const float4*>(B + br * ldb + k);                 \
            }                                                                            \
        }
        GLOAD(0, ra0, rb0) GLOAD(1, ra1, rb1)
#undef GLOAD
        __syncthreads();
#define LSTORE(i, ra, rb)                                                          \
        {                                                                          \
            const int chunk = tid + 256 * (i), row = chunk >> 2, kc = chunk & 3;   \
            float* pa = sA + row * LDK32 + kc * 4;                                 \
            float* pb = sB + row * LDK32 + kc * 4;                                 \
            pa[0] = ra.x; pa[1] = ra.y; pa[2] = ra.z; pa[3] = ra.w;                \
            pb[0] = rb.x; pb[1] = rb.y; pb[2] = rb.z; pb[3] = rb.w;                \
        }
        LSTORE(0, ra0, rb0) LSTORE(1, ra1, rb1)
#undef LSTORE
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < BK32 / 2; ++ks) {
            // 32x32x2: lane l holds A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31]
            const int k = ks * 2 + (lane >> 5);
            float a[2], b[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) a[m] = sA[(wr * 64 + m * 32 + (lane & 31)) * LDK32 + k];
#pragma unroll
            for (int n = 0; n < 2; ++n) b[n] = sB[(wc * 64 + n * 32 + (lane & 31)) * LDK32 + k];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], b[n], acc[m][n], 0, 0, 0);
        }
    }
    // C/D map of the 32x32 tile: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                epilogue1<TO>(C, ldc, pre, bias, epi, accumulate,
                              m0 + wr * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5),
                              n0 + wc * 64 + n * 32 + (lane & 31), M, N, acc[m][n][r]);
}

extern "C" int sedd_gemm_nt(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                            int in_dtype, int out_dtype, const float* bias, int epilogue, void* pre, int accumulate,
                            int64_t M, int64_t N, int64_t K, sedd_stream_t stream) {
    SEDD_CHECK_ARG((in_dtype == SEDD_F32 || in_dtype == SEDD_BF16) && (out_dtype == SEDD_F32 || out_dtype == SEDD_BF16),
                   "gemm_nt: bad dtype");
    SEDD_CHECK_ARG(epilogue >= SEDD_EPI_NONE && epilogue <= SEDD_EPI_BIAS_GELU, "gemm_nt: bad epilogue %d", epilogue);
    SEDD_CHECK_ARG(M >= 0 && N >= 0 && K > 0, "gemm_nt: bad sizes M=%ld N=%ld K=%ld", (long)M, (long)N, (long)K);
    SEDD_CHECK_ARG(epilogue == SEDD_EPI_NONE || bias, "gemm_nt: epilogue needs a bias");
    SEDD_CHECK_ARG(!accumulate || epilogue == SEDD_EPI_NONE, "gemm_nt: accumulate only with SEDD_EPI_NONE");
    const int kal = in_dtype == SEDD_BF16 ? 8 : 4;
    if (K % kal || lda % kal || ldb % kal || ((uintptr_t)A & 15) || ((uintptr_t)B & 15) || lda < K || ldb < K || ldc < N) {
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
        if (K % BK16 == 0) {
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
            else
                hipLaunchKernelGGL((gemm_bf16_nt_glds_kernel<float>), g2, dim3(256), 0, st, (const bf16_t*)A, lda,
                                   (const bf16_t*)B, ldb, (float*)C, ldc, bias, epilogue, (float*)pre, accumulate, M,
                                   N, K / splitk, (int)ntn);
        } else if (out_dtype == SEDD_BF16)
            hipLaunchKernelGGL((gemm_bf16_nt_kernel<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)A, lda,
                               (const bf16_t*)B, ldb, (bf16_t*)C, ldc, bias, epilogue, (bf16_t*)pre, accumulate, M, N,
                               K, (int)ntn);
        else
            hipLaunchKernelGGL((gemm_bf16_nt_kernel<float>), grid, dim3(256), 0, st, (const bf16_t*)A, lda,
                               (const bf16_t*)B, ldb, (float*)C, ldc, bias, epilogue, (float*)pre, accumulate, M, N, K,
                               (int)ntn);
    } else {
        if (out_dtype == SEDD_BF16)
            hipLaunchKernelGGL((gemm_f32_nt_kernel<bf16_t>), grid, dim3(256), 0, st, (const float*)A, lda,
                               (const float*)B, ldb, (bf16_t*)C, ldc, bias, epilogue, (bf16_t*)pre, accumulate, M, N, K,
                               (int)ntn);
        else
            hipLaunchKernelGGL((gemm_f32_nt_kernel<float>), grid, dim3(256), 0, st, (const float*)A, lda,
                               (const float*)B, ldb, (float*)C, ldc, bias, epilogue, (float*)pre, accumulate, M, N, K,
                               (int)ntn);
    }
    SEDD_LAUNCH_CHECK();
    return SEDD_OK;
}
