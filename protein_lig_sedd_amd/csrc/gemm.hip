// MFMA GEMM for every nn.Linear on the hot path (transformer.py:125-134,140,225,229;
// transformers_protlig_cross.py:250-268): C[M,N] = A[M,K] * B[N,K]^T (+bias)(GELU-tanh), both operands
// K-contiguous.  Forward uses it directly (A = activations, B = weight [out,in]); the backward data/weight
// gradients are the same kernel on re-laid-out operands (sedd_transpose), weight gradients accumulate in fp32.
//
// bf16 path: v_mfma_f32_16x16x32_bf16, 128x128x64 block tile, 4 waves as 2x2 (64x64 per wave = 4x4 MFMA tiles),
//   operands staged global -> registers -> LDS with the next tile's global loads issued before the current tile's
//   MFMAs (async-stage split), 16-byte LDS chunks XOR-swizzled by row so ds_read_b128 fragment reads spread over
//   the banks, XCD-aware block order (neighbouring tiles that share an A panel stay on one XCD's L2).
// fp32 path: exact-fp32 v_mfma_f32_32x32x2_f32 (k-ordered fmaf chain), 128x128x16 tile, for the fp32 parity
//   configuration and the fp32-autocast cross model.
// MFMA-bound: 2*M*N*K flop per call.
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define BM 128
#define BN 128

__device__ __forceinline__ float gelu_tanh_f(float u) {
    const float t = tanhf(0.7978845608028654f * (u + 0.044715f * u * u * u));
    return 0.5f * u * (1.0f + t);
}

// bijective XCD-aware remap (8 XCDs): blocks that share an XCD get a contiguous range of tiles
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

template <typename TO>
__device__ __forceinline__ void epilogue_store(TO* __restrict__ C, int64_t ldc, TO* __restrict__ pre,
                                               const float* __restrict__ bias, int epi, int accumulate, int64_t row,
                                               int64_t col, int64_t M, int64_t N, float v) {
    if (row >= M || col >= N) return;
    if (epi != SEDD_EPI_NONE) v += bias[col];
    if (epi == SEDD_EPI_BIAS_GELU) {
        const TO r = from_f32<TO>(v);  // the reference rounds the Linear output to the activation dtype first
        if (pre) pre[row * ldc + col] = r;
        v = gelu_tanh_f(to_f32<TO>(r));
    }
    TO* p = C + row * ldc + col;
    if (accumulate) v += to_f32<TO>(*p);
    *p = from_f32<TO>(v);
}

// ------------------------------------------------------------------------------------------------ bf16
#define BK16 64
template <typename TO>
__global__ void __launch_bounds__(256) gemm_bf16_nt_kernel(const bf16_t* __restrict__ A, int64_t lda,
                                                           const bf16_t* __restrict__ B, int64_t ldb,
                                                           TO* __restrict__ C, int64_t ldc,
                                                           const float* __restrict__ bias, int epi,
                                                           TO* __restrict__ pre, int accumulate, int64_t M, int64_t N,
                                                           int64_t K, int ntn) {
    __shared__ __attribute__((aligned(16))) uint4 sA[BM * 8];  // [row][8 x 16-byte chunk], chunk index swizzled
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

    uint4 ra[4], rb[4];
    auto gload = [&](int64_t k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int chunk = tid + 256 * i, row = chunk >> 3, kc = chunk & 7;
            const int64_t k = k0 + kc * 8;
            int64_t ar = m0 + row, br = n0 + row;
            ar = ar < M ? ar : M - 1;
            br = br < N ? br : N - 1;
            const uint4 z = make_uint4(0, 0, 0, 0);
            ra[i] = k < K ? *reinterpret_cast<const uint4*>(A + ar * lda + k) : z;
            rb[i] = k < K ? *reinterpret_cast<const uint4*>(B + br * ldb + k) : z;
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int chunk = tid + 256 * i, row = chunk >> 3, kc = chunk & 7;
            sA[row * 8 + (kc ^ (row & 7))] = ra[i];
            sB[row * 8 + (kc ^ (row & 7))] = rb[i];
        }
    };

    const int64_t nkt = (K + BK16 - 1) / BK16;
    gload(0);
    lstore();
    __syncthreads();
    for (int64_t kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) gload((kt + 1) * BK16);
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
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m], b[n], acc[m][n], 0, 0, 0);
        }
        __syncthreads();
        if (kt + 1 < nkt) {
            lstore();
            __syncthreads();
        }
    }
    // C/D map of the 16x16 tile: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                epilogue_store<TO>(C, ldc, pre, bias, epi, accumulate, m0 + wr * 64 + m * 16 + (lane >> 4) * 4 + r,
                                   n0 + wc * 64 + n * 16 + (lane & 15), M, N, acc[m][n][r]);
}

// ------------------------------------------------------------------------------------------------ fp32
#define BK32 16
#define LDK32 17  // padded row (floats): conflict-free ds_read_b32 column reads
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

    float4 ra[2], rb[2];
    auto gload = [&](int64_t k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int chunk = tid + 256 * i, row = chunk >> 2, kc = chunk & 3;
            const int64_t k = k0 + kc * 4;
            int64_t ar = m0 + row, br = n0 + row;
            ar = ar < M ? ar : M - 1;
            br = br < N ? br : N - 1;
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            ra[i] = k < K ? *reinterpret_cast<const float4*>(A + ar * lda + k) : z;
            rb[i] = k < K ? *reinterpret_cast<const float4*>(B + br * ldb + k) : z;
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int chunk = tid + 256 * i, row = chunk >> 2, kc = chunk & 3;
            float* pa = sA + row * LDK32 + kc * 4;
            float* pb = sB + row * LDK32 + kc * 4;
            pa[0] = ra[i].x; pa[1] = ra[i].y; pa[2] = ra[i].z; pa[3] = ra[i].w;
            pb[0] = rb[i].x; pb[1] = rb[i].y; pb[2] = rb[i].z; pb[3] = rb[i].w;
        }
    };

    const int64_t nkt = (K + BK32 - 1) / BK32;
    gload(0);
    lstore();
    __syncthreads();
    for (int64_t kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) gload((kt + 1) * BK32);
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
        __syncthreads();
        if (kt + 1 < nkt) {
            lstore();
            __syncthreads();
        }
    }
    // C/D map of the 32x32 tile: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                epilogue_store<TO>(C, ldc, pre, bias, epi, accumulate,
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
        if (out_dtype == SEDD_BF16)
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
