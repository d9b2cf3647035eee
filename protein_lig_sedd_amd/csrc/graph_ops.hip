// Bandwidth-bound token-graph kernels: corruption, score-entropy loss (+gradient), Gumbel arg-max and the fused
// Euler / Analytic / Denoiser predictor updates.  Reference: protlig_dd/processing/graph_lib.py,
// protlig_dd/sampling/sampling.py, protlig_dd/utils/catsample.py (cited per kernel).
//
// Layout: score / log-score / noise rows are [M, V] row-major; a row is owned by a G-lane sub-group of a wave
// (G = 8/16/32/64 >= min(V, 64)), lanes stride over the vocab axis so consecutive lanes touch consecutive
// addresses and consecutive groups consecutive rows (fully coalesced for V <= 64); reductions over the vocab
// axis are xor-shuffles inside the group.  Everything here is HBM-bound: one read of every input byte, one
// write of every output byte, fp32 maths.
//
// Floating-point contraction is off in this file so that every product/sum rounds exactly like the reference's
// op-by-op torch arithmetic (needed for bit-exact sampled indices).
#include "common.h"
#include <stdlib.h>
#pragma clang fp contract(off)

#define TPB 256

template <int G> struct GroupIdx {
    int lane;
    int64_t first, stride;
    __device__ GroupIdx() {
        lane = threadIdx.x & (G - 1);
        first = (int64_t)blockIdx.x * (TPB / G) + (threadIdx.x / G);
        stride = (int64_t)gridDim.x * (TPB / G);
    }
};

__device__ __forceinline__ float gumbel_ratio(float p, float u) {
    // catsample.py:12-13: probs / (1e-10 - (u + 1e-10).log())
    return p / (1e-10f - logf(u + 1e-10f));
}


// logf for arguments that are never subnormal (u + 1e-10 with u >= 0): the library's algorithm — v_log_f32 followed by a
// compensated multiplication by ln 2 — without its subnormal pre-scaling, so the bits equal logf's on that domain.
__device__ __forceinline__ float logf_normal(float x) {
    const float r = __builtin_amdgcn_logf(x);
    const float hi = r * 0x1.62e42ep-1f;
    float lo = fmaf(r, 0x1.62e42ep-1f, -hi);
    lo = fmaf(r, 0x1.efa39ep-25f, lo);
    return fabsf(r) < INFINITY ? hi + lo : r;
}
__device__ __forceinline__ float gumbel_ratio_fast(float p, float u) {
    return p / (1e-10f - logf_normal(u + 1e-10f));
}
// arg-max over a row scanned in increasing index order, torch semantics (first maximum; NaN is the maximum)
struct ArgmaxScan {
    float bv;
    int bi, ni;
    __device__ __forceinline__ void first(float v) { bv = v; bi = 0; ni = v != v ? 0 : 0x7fffffff; }
    __device__ __forceinline__ void push(float v, int j) {
        if (v > bv) { bv = v; bi = j; }
        if (v != v && ni == 0x7fffffff) ni = j;
    }
    __device__ __forceinline__ int result() const { return ni != 0x7fffffff ? ni : bi; }
};
// Rows that need the full computation are compacted to the front of the block (wave ballot + one LDS atomic per wave)
// so that for an absorbing graph whole waves retire when most rows of a tile are unmasked fixed points.
__device__ __forceinline__ void rows_compact(bool active, int r, int* s_cnt, unsigned short* s_list) {
    const unsigned long long mask = __ballot(active);
    const int lane = threadIdx.x & 63;
    int base = 0;
    if (lane == 0 && mask) base = atomicAdd(s_cnt, __popcll(mask));
    base = __shfl(base, 0, 64);
    if (active) s_list[base + __popcll(mask & ((1ull << lane) - 1ull))] = (unsigned short)r;
}

// ===================================================================== small-vocab "row per lane" machinery
// For V <= 64 a G-lane group per row wastes most of its instructions on cross-lane shuffles, so the V <= 64 paths of
// score_entropy / gumbel_argmax / predictor_step use a second layout: a block stages ROWS consecutive rows — one
// contiguous, 16-byte aligned byte range of each [M, V] array — into LDS with 16-byte LDS-DMA loads (fully coalesced,
// no VGPR round trip), then every lane owns ONE row and walks its V entries out of LDS with compile-time unrolled
// loops (row stride V*esize bytes: conflict-free or 2-way for the vocabularies in use; strides that are a multiple of
// 32 bytes fall back to the group kernels).  Row sums are evaluated in the same association order as the group
// kernels' xor-shuffle tree, so both layouts return identical bits.
#ifndef SEDD_ROWS_PER_BLOCK
#define SEDD_ROWS_PER_BLOCK 256
#endif
template <int VP> struct RowsCfg { static constexpr int ROWS = VP == 64 ? (SEDD_ROWS_PER_BLOCK > 128 ? 128 : SEDD_ROWS_PER_BLOCK) : SEDD_ROWS_PER_BLOCK; };

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void rows_stage_in(const unsigned char* __restrict__ g, unsigned char* s, int nbytes,
                                              int nthreads) {
    const int nch = nbytes >> 4;
    const int lane = threadIdx.x & 63;
    for (int c0 = threadIdx.x - lane; c0 < nch; c0 += nthreads) {  // wave-uniform trip count and LDS base
        if (c0 + lane < nch)
            __builtin_amdgcn_global_load_lds((gptr_t)(g + (int64_t)(c0 + lane) * 16), (lptr_t)(s + c0 * 16), 16, 0, 0);
    }
    const int done = nch << 4;
    if ((int)threadIdx.x < nbytes - done) s[done + threadIdx.x] = g[done + threadIdx.x];
}
__device__ __forceinline__ void rows_stage_wait() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}
__device__ __forceinline__ void rows_stage_out(unsigned char* __restrict__ g, const unsigned char* s, int nbytes,
                                               int nthreads) {
    const int nch = nbytes >> 4;
    for (int c = threadIdx.x; c < nch; c += nthreads) ((uint4*)g)[c] = ((const uint4*)s)[c];
    const int done = nch << 4;
    if ((int)threadIdx.x < nbytes - done) g[done + threadIdx.x] = s[done + threadIdx.x];
}
// sum of v[0..VP) in the association order of group_sum<VP> (xor-shuffle tree, lane j holding v[j]); destroys v
template <int VP> __device__ __forceinline__ float tree_sum(float (&v)[VP]) {
#pragma unroll
    for (int off = VP / 2; off > 0; off >>= 1) {
#pragma unroll
        for (int i = 0; i < off; ++i) v[i] = v[i] + v[i + off];
    }
    return v[0];
}
// nu = number of fp32 noise columns staged beside the esize-byte score columns (0: none)
static inline bool rows_path_ok(int V, int esize, int nu, const void* a, const void* b, const void* c) {
    const uintptr_t al = (uintptr_t)a | (uintptr_t)b | (uintptr_t)c;
    const int rows = V > 32 ? 128 : 256;
    const size_t lds = ((size_t)rows * V * esize + 15) / 16 * 16 + (size_t)rows * nu * 4;
    return V <= 64 && (V * esize) % 32 != 0 && (nu == 0 || (nu * 4) % 32 != 0) && (al & 15) == 0 && lds <= 62 * 1024;
}
__host__ __device__ __host__ __device__ static inline size_t rows_lds(int rows, int rowbytes) { return ((size_t)rows * rowbytes + 15) & ~(size_t)15; }
#define SEDD_DISPATCH_VP(V, ...)                          \
    do {                                                  \
        switch (sedd_group_for(V)) {                      \
            case 8:  { constexpr int VP = 8;  __VA_ARGS__; } break; \
            case 16: { constexpr int VP = 16; __VA_ARGS__; } break; \
            case 32: { constexpr int VP = 32; __VA_ARGS__; } break; \
            default: { constexpr int VP = 64; __VA_ARGS__; } break; \
        }                                                 \
    } while (0)

// ------------------------------------------------------------------------------------- sample_transition
// graph_lib.py:258-262 (absorbing), :156-160 (uniform)
__global__ void __launch_bounds__(TPB) sample_transition_kernel(int graph, const int64_t* __restrict__ x0,
                                                               const float* __restrict__ sigma,
                                                               const float* __restrict__ u,
                                                               const int64_t* __restrict__ r,
                                                               int64_t* __restrict__ xt, int64_t M, int64_t L,
                                                               int V) {
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < M; i += (int64_t)gridDim.x * TPB) {
        const float move_chance = 1.0f - expf(-sigma[i / L]);
        const bool move = u[i] < move_chance;
        const int64_t tgt = graph == SEDD_ABSORB ? (int64_t)(V - 1) : r[i];
        xt[i] = move ? tgt : x0[i];
    }
}

extern "C" int sedd_sample_transition(int graph, const int64_t* x0, const float* sigma, const float* u,
                                      const int64_t* r, int64_t* xt, int64_t M, int64_t L, int V,
                                      sedd_stream_t stream) {
    SEDD_CHECK_ARG(graph == SEDD_ABSORB || graph == SEDD_UNIFORM, "sample_transition: unknown graph %d", graph);
    SEDD_CHECK_ARG(M == 0 || (x0 && sigma && u && xt), "sample_transition: null pointer");
    SEDD_CHECK_ARG(M == 0 || graph == SEDD_ABSORB || r, "sample_transition: uniform graph needs r");
    SEDD_CHECK_ARG(M >= 0 && L > 0 && V > 0 && M % L == 0, "sample_transition: bad sizes M=%ld L=%ld V=%d", (long)M,
                   (long)L, V);
    if (M == 0) return SEDD_OK;
    hipLaunchKernelGGL(sample_transition_kernel, dim3(sedd_grid_for(M, TPB)), dim3(TPB), 0, (hipStream_t)stream,
                       graph, x0, sigma, u, r, xt, M, L, V);
    SEDD_LAUNCH_CHECK();
    return SEDD_OK;
}

// ------------------------------------------------------------------------------------------ score entropy
// Absorbing: graph_lib.py:274-314.  Uniform: graph_lib.py:170-197.  gout == nullptr: gradient weight is gw[b]
// (fused loss path) ; dscore == nullptr: forward only ; out == nullptr: backward only.
template <typename T, int G>
__global__ void __launch_bounds__(TPB) score_entropy_kernel(int graph, const T* __restrict__ score,
                                                            const float* __restrict__ sigma,
                                                            const int64_t* __restrict__ x,
                                                            const int64_t* __restrict__ x0, float* __restrict__ out,
                                                            T* __restrict__ dscore, const float* __restrict__ gw,
                                                            const float* __restrict__ gout, int64_t M, int64_t L,
                                                            int V) {
    GroupIdx<G> g;
    for (int64_t row = g.first; row < M; row += g.stride) {
        const int64_t xi = x[row];
        const float sig = sigma[row / L];
        const float es = esigm1f(sig);
        const T* srow = score + row * V;
        T* drow = dscore ? dscore + row * V : nullptr;
        float w = 0.f;
        if (drow) w = gout ? gout[row] : gw[row / L];
        if (graph == SEDD_ABSORB) {
            if (xi != V - 1) {  // unmasked token: zero loss, zero gradient
                if (out && g.lane == 0) out[row] = 0.f;
                if (drow)
                    for (int j = g.lane; j < V; j += G) drow[j] = from_f32<T>(0.f);
                continue;
            }
            int64_t o = x0[row];
            o = o < 0 ? 0 : (o > V - 1 ? V - 1 : o);
            const float ratio = 1.0f / es;
            float pos = 0.f, so = 0.f;
            for (int j = g.lane; j < V; j += G) {
                const float s = to_f32<T>(srow[j]);
                const float e = j < V - 1 ? expf(s) : 0.f;
                pos += e;
                if (j == o) so = s;
                if (drow) drow[j] = from_f32<T>(w * (e - (j == o ? ratio : 0.f)));
            }
            pos = group_sum<G>(pos);
            so = group_sum<G>(so);
            if (out && g.lane == 0) out[row] = pos - ratio * so + ratio * (logf(ratio) - 1.0f);
        } else {
            const float Vf = (float)V;
            const int64_t o = x0[row];
            const float ratio = 1.0f - Vf / (es + Vf);
            const bool same = xi == o;
            float ssum = 0.f, esum = 0.f, sx = 0.f, so = 0.f;
            for (int j = g.lane; j < V; j += G) {
                const float s = to_f32<T>(srow[j]);
                const float e = expf(s);
                ssum += s;
                esum += e;
                if (j == xi) sx = s;
                if (j == o) so = s;
                if (drow) {
                    // d pos = (e^s_j - [j==x] e^s_x)/V ; d neg0 = (1 - [j==x])/V
                    const float notx = j == xi ? 0.f : 1.f;
                    const float dpos = notx * e / Vf;
                    const float dneg0 = notx / Vf;
                    const float dneg = same ? ratio * dneg0 : (j == o ? 1.0f / es : 0.f) + dneg0;
                    drow[j] = from_f32<T>(w * (dpos - dneg));
                }
            }
            ssum = group_sum<G>(ssum);
            esum = group_sum<G>(esum);
            sx = group_sum<G>(sx);
            so = group_sum<G>(so);
            if (out && g.lane == 0) {
                float neg = ssum / Vf - sx / Vf;
                neg = same ? ratio * neg : so / es + neg;
                const float lr = logf(ratio);
                const float cst = same ? (Vf - 1.0f) / Vf * ratio * (lr - 1.0f)
                                       : ((-lr - 1.0f) / ratio - (Vf - 2.0f)) / Vf;
                const float pos = esum / Vf - expf(sx) / Vf;
                out[row] = pos - neg + cst;
            }
        }
    }
}

// row-per-lane variant (V <= 64), same arithmetic; the gradient tile is built in place over the staged scores and
// leaves with 16-byte stores
template <typename T, int VP>
__global__ void __launch_bounds__(RowsCfg<VP>::ROWS)
    score_entropy_rows_kernel(int graph, const T* __restrict__ score, const float* __restrict__ sigma,
                              const int64_t* __restrict__ x, const int64_t* __restrict__ x0, float* __restrict__ out,
                              T* __restrict__ dscore, const float* __restrict__ gw, const float* __restrict__ gout,
                              int64_t M, int64_t L, int V) {
    constexpr int ROWS = RowsCfg<VP>::ROWS;
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ int s_cnt;
    __shared__ unsigned short s_list[ROWS];
    __shared__ int s_x[ROWS];
    __shared__ int64_t s_x0[ROWS];
    __shared__ float s_sig[ROWS], s_w[ROWS];
    const int64_t row0 = (int64_t)blockIdx.x * ROWS;
    const int nrows = (int)(M - row0 < ROWS ? M - row0 : ROWS);
    const int nbytes = nrows * V * (int)sizeof(T);
    if (threadIdx.x == 0) s_cnt = 0;
    rows_stage_in((const unsigned char*)(score + row0 * V), smem, nbytes, ROWS);
    int r = threadIdx.x;
    int64_t xr = 0, x0r = 0;
    float sig_r = 0.f, w_r = 0.f;
    if (r < nrows) {  // per-row scalars are fetched under the staging latency and handed over through LDS
        xr = x[row0 + r];
        x0r = x0[row0 + r];
        sig_r = sigma[(row0 + r) / L];
        if (dscore) w_r = gout ? gout[row0 + r] : gw[(row0 + r) / L];
    }
    rows_stage_wait();
    bool active = false;
    if (r < nrows) {
        if (graph == SEDD_ABSORB && xr != V - 1) {  // unmasked token: zero loss, zero gradient
            if (out) out[row0 + r] = 0.f;
            if (dscore) {
                T* srow = (T*)smem + r * V;
#pragma unroll
                for (int j = 0; j < VP; ++j)
                    if (j < V) srow[j] = from_f32<T>(0.f);
            }
        } else {
            active = true;
            s_x[r] = (int)xr;
            s_x0[r] = x0r;
            s_sig[r] = sig_r;
            s_w[r] = w_r;
        }
    }
    rows_compact(active, r, &s_cnt, s_list);
    __syncthreads();
    const int slot = (threadIdx.x + ROWS - 64 * (blockIdx.x % (ROWS / 64))) % ROWS;  // see predictor_step_rows_kernel
    if (slot < s_cnt) {
        r = s_list[slot];
        const int64_t row = row0 + r;
        const int xi = s_x[r];
        const float es = esigm1f(s_sig[r]);
        T* srow = (T*)smem + r * V;
        const float w = s_w[r];
        if (graph == SEDD_ABSORB) {
            int64_t o = s_x0[r];
            o = o < 0 ? 0 : (o > V - 1 ? V - 1 : o);
            const float ratio = 1.0f / es;
            float t[VP];
            float so = 0.f;
#pragma unroll
            for (int j = 0; j < VP; ++j) {
                t[j] = 0.f;
                if (j < V) {
                    const float s = to_f32<T>(srow[j]);
                    const float e = j < V - 1 ? expf(s) : 0.f;
                    t[j] = e;
                    if (j == o) so = s;
                    if (dscore) srow[j] = from_f32<T>(w * (e - (j == o ? ratio : 0.f)));
                }
            }
            const float pos = tree_sum<VP>(t);
            if (out) out[row] = pos - ratio * so + ratio * (logf(ratio) - 1.0f);
        } else {
            const float Vf = (float)V;
            const int64_t o = s_x0[r];
            const float ratio = 1.0f - Vf / (es + Vf);
            const bool same = xi == o;
            float ts[VP], te[VP];
            float sx = 0.f, so = 0.f;
#pragma unroll
            for (int j = 0; j < VP; ++j) {
                ts[j] = 0.f;
                te[j] = 0.f;
                if (j < V) {
                    const float s = to_f32<T>(srow[j]);
                    const float e = expf(s);
                    ts[j] = s;
                    te[j] = e;
                    if (j == xi) sx = s;
                    if (j == o) so = s;
                    if (dscore) {
                        // d pos = (e^s_j - [j==x] e^s_x)/V ; d neg0 = (1 - [j==x])/V
                        const float notx = j == xi ? 0.f : 1.f;
                        const float dpos = notx * e / Vf;
                        const float dneg0 = notx / Vf;
                        const float dneg = same ? ratio * dneg0 : (j == o ? 1.0f / es : 0.f) + dneg0;
                        srow[j] = from_f32<T>(w * (dpos - dneg));
                    }
                }
            }
            const float ssum = tree_sum<VP>(ts);
            const float esum = tree_sum<VP>(te);
            if (out) {
                float neg = ssum / Vf - sx / Vf;
                neg = same ? ratio * neg : so / es + neg;
                const float lr = logf(ratio);
                const float cst = same ? (Vf - 1.0f) / Vf * ratio * (lr - 1.0f)
                                       : ((-lr - 1.0f) / ratio - (Vf - 2.0f)) / Vf;
                const float pos = esum / Vf - expf(sx) / Vf;
                out[row] = pos - neg + cst;
            }
        }
    }
    if (dscore) {
        __syncthreads();
        rows_stage_out((unsigned char*)(dscore + row0 * V), smem, nbytes, ROWS);
    }
}

template <typename T>
static int launch_score_entropy(int graph, const void* score, const float* sigma, const int64_t* x,
                                const int64_t* x0, float* out, void* dscore, const float* gw, const float* gout,
                                int64_t M, int64_t L, int V, hipStream_t st) {
    if (rows_path_ok(V, (int)sizeof(T), 0, score, dscore, nullptr)) {
        SEDD_DISPATCH_VP(V, hipLaunchKernelGGL((score_entropy_rows_kernel<T, VP>),
                                               dim3((unsigned)((M + RowsCfg<VP>::ROWS - 1) / RowsCfg<VP>::ROWS)),
                                               dim3(RowsCfg<VP>::ROWS), rows_lds(RowsCfg<VP>::ROWS, V * (int)sizeof(T)),
                                               st, graph, (const T*)score, sigma, x, x0, out, (T*)dscore, gw, gout, M,
                                               L, V));
        SEDD_LAUNCH_CHECK();
        return SEDD_OK;
    }
    SEDD_DISPATCH_GROUP(V, hipLaunchKernelGGL((score_entropy_kernel<T, G>), dim3(sedd_grid_for(M, TPB / G)),
                                              dim3(TPB), 0, st, graph, (const T*)score, sigma, x, x0, out,
                                              (T*)dscore, gw, gout, M, L, V));
    SEDD_LAUNCH_CHECK();
    return SEDD_OK;
}

static int check_se_args(int graph, const void* score, int dtype, const float* sigma, const int64_t* x,
                         const int64_t* x0, int64_t M, int64_t L, int V) {
    SEDD_CHECK_ARG(graph == SEDD_ABSORB || graph == SEDD_UNIFORM, "score_entropy: unknown graph %d", graph);
    SEDD_CHECK_ARG(dtype == SEDD_F32 || dtype == SEDD_BF16, "score_entropy: unknown dtype %d", dtype);
    SEDD_CHECK_ARG(M == 0 || (score && sigma && x && x0), "score_entropy: null pointer");
    SEDD_CHECK_ARG(M >= 0 && L > 0 && V > 1 && M % L == 0, "score_entropy: bad sizes M=%ld L=%ld V=%d", (long)M,
                   (long)L, V);
    return SEDD_OK;
}

extern "C" int sedd_score_entropy_fwd(int graph, const void* score, int dtype, const float* sigma,
                                      const int64_t* x, const int64_t* x0, float* out, void* dscore,
                                      const float* gw, int64_t M, int64_t L, int V, sedd_stream_t stream) {
    int rc = check_se_args(graph, score, dtype, sigma, x, x0, M, L, V);
    if (rc) return rc;
    SEDD_CHECK_ARG(M == 0 || out, "score_entropy_fwd: out is null");
    SEDD_CHECK_ARG(!dscore || gw, "score_entropy_fwd: dscore requested without gw");
    if (M == 0) return SEDD_OK;
    if (dtype == SEDD_F32)
        return launch_score_entropy<float>(graph, score, sigma, x, x0, out, dscore, gw, nullptr, M, L, V,
                                           (hipStream_t)stream);
    return launch_score_entropy<bf16_t>(graph, score, sigma, x, x0, out, dscore, gw, nullptr, M, L, V,
                                        (hipStream_t)stream);
}

extern "C" int sedd_score_entropy_bwd(int graph, const void* score, int dtype, const float* sigma,
                                      const int64_t* x, const int64_t* x0, const float* gout, void* dscore,
                                      int64_t M, int64_t L, int V, sedd_stream_t stream) {
    int rc = check_se_args(graph, score, dtype, sigma, x, x0, M, L, V);
    if (rc) return rc;
    SEDD_CHECK_ARG(M == 0 || (gout && dscore), "score_entropy_bwd: null pointer");
    if (M == 0) return SEDD_OK;
    if (dtype == SEDD_F32)
        return launch_score_entropy<float>(graph, score, sigma, x, x0, nullptr, dscore, nullptr, gout, M, L, V,
                                           (hipStream_t)stream);
    return launch_score_entropy<bf16_t>(graph, score, sigma, x, x0, nullptr, dscore, nullptr, gout, M, L, V,
                                        (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------ gumbel argmax
template <typename T, int G>
__global__ void __launch_bounds__(TPB) gumbel_argmax_kernel(const T* __restrict__ p, const float* __restrict__ u,
                                                            int64_t* __restrict__ out, int64_t M, int V) {
    GroupIdx<G> g;
    for (int64_t row = g.first; row < M; row += g.stride) {
        const T* prow = p + row * V;
        const float* urow = u + row * V;
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int j = g.lane; j < V; j += G) {
            const float v = gumbel_ratio(to_f32<T>(prow[j]), urow[j]);
            if (argmax_better(v, j, bv, bi)) { bv = v; bi = j; }
        }
        group_argmax<G>(bv, bi);
        if (g.lane == 0) out[row] = bi;
    }
}

template <typename T, int VP>
__global__ void __launch_bounds__(RowsCfg<VP>::ROWS)
    gumbel_argmax_rows_kernel(const T* __restrict__ p, const float* __restrict__ u, int64_t* __restrict__ out,
                              int64_t M, int V) {
    constexpr int ROWS = RowsCfg<VP>::ROWS;
    extern __shared__ __align__(16) unsigned char smem[];
    const int64_t row0 = (int64_t)blockIdx.x * ROWS;
    const int nrows = (int)(M - row0 < ROWS ? M - row0 : ROWS);
    unsigned char* s_u = smem + rows_lds(ROWS, V * (int)sizeof(T));
    rows_stage_in((const unsigned char*)(p + row0 * V), smem, nrows * V * (int)sizeof(T), ROWS);
    rows_stage_in((const unsigned char*)(u + row0 * V), s_u, nrows * V * 4, ROWS);
    rows_stage_wait();
    const int r = threadIdx.x;
    if (r >= nrows) return;
    const T* prow = (const T*)smem + r * V;
    const float* urow = (const float*)s_u + r * V;
    ArgmaxScan am;
#pragma unroll
    for (int j = 0; j < VP; ++j) {
        if (j < V) {
            const float v = gumbel_ratio_fast(to_f32<T>(prow[j]), urow[j]);
            if (j == 0) am.first(v);
            else am.push(v, j);
        }
    }
    out[row0 + r] = am.result();
}

extern "C" int sedd_gumbel_argmax(const void* p, int dtype, const float* u, int64_t* out, int64_t M, int V,
                                  sedd_stream_t stream) {
    SEDD_CHECK_ARG(dtype == SEDD_F32 || dtype == SEDD_BF16, "gumbel_argmax: unknown dtype %d", dtype);
    SEDD_CHECK_ARG(M == 0 || (p && u && out), "gumbel_argmax: null pointer");
    SEDD_CHECK_ARG(M >= 0 && V > 0, "gumbel_argmax: bad sizes M=%ld V=%d", (long)M, V);
    if (M == 0) return SEDD_OK;
    hipStream_t st = (hipStream_t)stream;
    const int esz = dtype == SEDD_F32 ? 4 : 2;
    if (rows_path_ok(V, esz, V, p, u, nullptr)) {
        if (dtype == SEDD_F32) {
            SEDD_DISPATCH_VP(V, hipLaunchKernelGGL((gumbel_argmax_rows_kernel<float, VP>),
                                                   dim3((unsigned)((M + RowsCfg<VP>::ROWS - 1) / RowsCfg<VP>::ROWS)),
                                                   dim3(RowsCfg<VP>::ROWS),
                                                   rows_lds(RowsCfg<VP>::ROWS, V * 4) + rows_lds(RowsCfg<VP>::ROWS, V * 4),
                                                   st, (const float*)p, u, out, M, V));
        } else {
            SEDD_DISPATCH_VP(V, hipLaunchKernelGGL((gumbel_argmax_rows_kernel<bf16_t, VP>),
                                                   dim3((unsigned)((M + RowsCfg<VP>::ROWS - 1) / RowsCfg<VP>::ROWS)),
                                                   dim3(RowsCfg<VP>::ROWS),
                                                   rows_lds(RowsCfg<VP>::ROWS, V * 2) + rows_lds(RowsCfg<VP>::ROWS, V * 4),
                                                   st, (const bf16_t*)p, u, out, M, V));
        }
        SEDD_LAUNCH_CHECK();
        return SEDD_OK;
    }
    if (dtype == SEDD_F32) {
        SEDD_DISPATCH_GROUP(V, hipLaunchKernelGGL((gumbel_argmax_kernel<float, G>),
                                                  dim3(sedd_grid_for(M, TPB / G)), dim3(TPB), 0, st,
                                                  (const float*)p, u, out, M, V));
    } else {
        SEDD_DISPATCH_GROUP(V, hipLaunchKernelGGL((gumbel_argmax_kernel<bf16_t, G>),
                                                  dim3(sedd_grid_for(M, TPB / G)), dim3(TPB), 0, st,
                                                  (const bf16_t*)p, u, out, M, V));
    }
    SEDD_LAUNCH_CHECK();
    return SEDD_OK;
}

// ------------------------------------------------------------------------------------- fused predictor step
// Euler (sampling.py:61-80): p = onehot(x) + dt*dsigma * reverse_rate(x, exp(logscore))   (graph_lib.py:79-96)
// Analytic (sampling.py:90-99) / Denoiser (:107-132): p = staggered_score(exp(logscore), ds) *
// transp_transition(x, ds) (graph_lib.py:264-269, :226-234 | :162-165, :147-154), Denoiser drops the mask column.
// then catsample.py:12-13.  Unmasked rows of the absorbing graph are fixed points (SURVEY section 4
// KAT-unmasked-fixed-point): p is a one-hot at x, so neither the row's scores nor its noise are read.
template <typename T, int G>
__global__ void __launch_bounds__(TPB) predictor_step_kernel(int graph, int step, const T* __restrict__ ls,
                                                             const int64_t* __restrict__ x,
                                                             const float* __restrict__ coef,
                                                             const float* __restrict__ u,
                                                             int64_t* __restrict__ xnew, int64_t M, int64_t L,
                                                             int V) {
    GroupIdx<G> g;
    const int m = V - 1;
    const int Vu = (graph == SEDD_ABSORB && step == SEDD_STEP_DENOISE) ? V - 1 : V;
    for (int64_t row = g.first; row < M; row += g.stride) {
        const int64_t xi64 = x[row];
        const int xi = (int)xi64;
        const float cf = coef[row / L];
        const T* lrow = ls + row * V;
        const float* urow = u + row * Vu;
        if (graph == SEDD_ABSORB && xi != m) {
            int64_t res = xi64;
            if (step != SEDD_STEP_EULER) {
                // p_x = (s_x e^{ds}) * e^{-ds}; every other entry is (+-)0, so x wins unless p_x underflowed to 0
                const float pi = (expf(to_f32<T>(lrow[xi])) * expf(cf)) * expf(-cf);
                res = (pi > 0.f || pi != pi) ? xi64 : 0;
            }
            if (g.lane == 0) xnew[row] = res;
            continue;
        }
        // pass 1: sum of scores
        float ssum = 0.f;
        for (int j = g.lane; j < V; j += G) {
            const float s = expf(to_f32<T>(lrow[j]));
            bool inc;
            if (step == SEDD_STEP_EULER) inc = j != xi;  // reverse_rate zeroes the diagonal before summing
            else inc = true;                              // staggered_score sums every column
            if (graph == SEDD_UNIFORM && step == SEDD_STEP_EULER) ssum += inc ? (1.0f / (float)V) * s : 0.f;
            else ssum += inc ? s : 0.f;
        }
        ssum = group_sum<G>(ssum);
        float e_p = 0.f, e_n = 0.f, one_m_en = 0.f, extra = 0.f, a_u = 0.f, tval = 0.f, tdiag = 0.f;
        if (step != SEDD_STEP_EULER) {
            e_p = expf(cf);
            e_n = expf(-cf);
            one_m_en = 1.0f - e_n;
            if (graph == SEDD_ABSORB) {
                extra = (1.0f - e_p) * ssum;
            } else {
                a_u = (e_n - 1.0f) / ((float)V * e_n);
                tval = (1.0f * one_m_en) / (float)V;
                float ts = 0.f;
                for (int j = g.lane; j < V; j += G) ts += j != xi ? tval : 0.f;
                tdiag = 1.0f - group_sum<G>(ts);
            }
        }
        // pass 2: probabilities, Gumbel ratio, arg-max
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int j = g.lane; j < Vu; j += G) {
            const float s = expf(to_f32<T>(lrow[j]));
            float p;
            if (step == SEDD_STEP_EULER) {
                float rj;
                if (graph == SEDD_ABSORB) rj = j == xi ? -ssum : s;  // row is masked here: transp_rate = 1 off-diag
                else rj = j == xi ? -ssum : (1.0f / (float)V) * s;
                p = (j == xi ? 1.0f : 0.f) + cf * rj;
            } else if (graph == SEDD_ABSORB) {
                float stag = s * e_p;
                if (j == m) stag += extra;
                const float tr = j == m ? e_n + one_m_en : one_m_en;
                p = stag * tr;
            } else {
                const float stag = a_u * ssum + s / e_n;
                p = stag * (j == xi ? tdiag : tval);
            }
            const float v = gumbel_ratio(p, urow[j]);
            if (argmax_better(v, j, bv, bi)) { bv = v; bi = j; }
        }
        group_argmax<G>(bv, bi);
        if (g.lane == 0) xnew[row] = bi;
    }
}

// What bounds this kernel (round-2 measurements at the cfg-5 shard, 512 x 1024 rows, V = 26, bf16; 90 MB algorithmic):
// LDS-DMA alone moves the tiles in 13.5 us, but SQ_INSTS_VALU is 11.4 M wave-instructions per launch — ~2000 per
// surviving row (OCML expf 13, compensated log 9, IEEE division 11, selects / compares / moves ~25 per vocabulary entry)
// — and a SIMD issues one VALU instruction per 4 cycles: a ~19 us VALU floor, above the HBM floor.  Persistent variants
// (two LDS buffers with four lanes per row; one early-released buffer with the row copied to registers) overlapped the
// loads but ran 33-41 us: four lanes per row double the instruction count, one lane per row makes a stage's arithmetic a
// 4-6 us serial chain per wave behind which the block's next stage waits (tools/pred_bench.py has the harness).  Two
// things found on the way apply here: `__syncthreads()` = vmcnt(0) + barrier drains every in-flight LDS-DMA, and LDS
// reads placed under the `j < V` guards compile to one round trip per element (the row is now read unguarded, in one
// batch: 29.7 -> 28.2 us).  An upper bound on what cheaper arithmetic could buy (v_log_f32 * ln 2 and v_rcp_f32 in place of
// the compensated log and the IEEE division, tokens no longer bit-exact; a certified version would fall back to the exact
// scan on near-ties): 28.5 -> 25.8 us — and with only 10 % of the rows doing any arithmetic the kernel still takes 23 us, so
// the VALU count explains a third of the gap to the DMA-only time at most; not pursued.  Bytes are not the limit either:
// staging only the chunks that touch a masked row (the state read first, the DMA exec-masked by a per-row flag; tokens
// still bit-exact) moved 45 % fewer bytes at the bench's 50 % mask fraction and ran 32.2 us instead of 28.5 (26.6 vs 25.4 at
// 10 %).  What remains is the serial chain of a block (scalars -> DMA -> wait -> per-row arithmetic -> store, ~9 us) times
// the 2.7 rounds of blocks that 40 KB of LDS per block lets the chip hold.
// Philox4x32-10 (Salmon et al., SC'11; the counter-based generator behind curand / torch's device RNG): four 32-bit words
// from a 128-bit counter and a 64-bit key.  sedd_predictor_step_philox draws the uniform of (row, vocabulary slot j) as
// word (j & 3) of philox(counter = {row_lo, row_hi, j >> 2, offset}, key = {seed_lo, seed_hi}), u = (word >> 8) * 2^-24
// in [0, 1) — a function of (seed, offset, row, j) only, so results do not depend on the launch geometry.
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0;
        c1 = lo1;
        c2 = hi0 ^ c3 ^ k1;
        c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// PHILOX: the uniforms are generated in registers (above) instead of being staged from `u`: M (V e + 16) bytes instead
// of M (V e + 4 Vu + 16), at ~10 quarter-rate integer multiplies per uniform.
template <typename T, int VP, int graph, int step, bool PHILOX = false>
__global__ void __launch_bounds__(RowsCfg<VP>::ROWS)
    predictor_step_rows_kernel(const T* __restrict__ ls, const int64_t* __restrict__ x,
                               const float* __restrict__ coef, const float* __restrict__ u,
                               int64_t* __restrict__ xnew, int64_t M, int64_t L, int V, uint64_t seed = 0,
                               uint32_t offset = 0) {
    constexpr int ROWS = RowsCfg<VP>::ROWS;
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ int s_cnt;
    __shared__ unsigned short s_list[ROWS];
    __shared__ int s_x[ROWS];
    __shared__ float s_cf[ROWS];
    const int m = V - 1;
    const int Vu = (graph == SEDD_ABSORB && step == SEDD_STEP_DENOISE) ? V - 1 : V;
    const int64_t row0 = (int64_t)blockIdx.x * ROWS;
    const int nrows = (int)(M - row0 < ROWS ? M - row0 : ROWS);
    unsigned char* s_u = smem + rows_lds(ROWS, V * (int)sizeof(T));
    const int tid = threadIdx.x;
    if (tid == 0) s_cnt = 0;
    rows_stage_in((const unsigned char*)(ls + row0 * V), smem, nrows * V * (int)sizeof(T), ROWS);
    if constexpr (!PHILOX) rows_stage_in((const unsigned char*)(u + row0 * Vu), s_u, nrows * Vu * 4, ROWS);
    const int64_t xi64 = tid < nrows ? x[row0 + tid] : 0;
    const float cf_own = tid < nrows ? coef[(row0 + tid) / L] : 0.f;  // fetched under the staging latency
    rows_stage_wait();
    bool active = false;
    if (tid < nrows) {
        if (graph == SEDD_ABSORB && (int)xi64 != m) {
            int64_t res = xi64;
            if (step != SEDD_STEP_EULER) {
                // p_x = (s_x e^{ds}) * e^{-ds}; every other entry is (+-)0, so x wins unless p_x underflowed to 0
                const float cf = cf_own;
                const float pi = (expf(to_f32<T>(((const T*)smem)[tid * V + (int)xi64])) * expf(cf)) * expf(-cf);
                res = (pi > 0.f || pi != pi) ? xi64 : 0;
            }
            xnew[row0 + tid] = res;
        } else {
            active = true;
            s_x[tid] = (int)xi64;
            s_cf[tid] = cf_own;
        }
    }
    rows_compact(active, tid, &s_cnt, s_list);
    __syncthreads();
    // surviving rows go to a block-dependent rotation of the waves, so retired waves spread over all SIMDs
    const int slot = (tid + ROWS - 64 * (blockIdx.x % (ROWS / 64))) % ROWS;
    if (slot >= s_cnt) return;
    const int r = s_list[slot];
    const int64_t row = row0 + r;
    const int xi = s_x[r];
    const float cf = s_cf[r];
    const T* lrow = (const T*)smem + r * V;
    const float* urow = (const float*)s_u + r * Vu;
    float e[VP], t[VP], un[VP];
    // The row is read with all VP slots unconditionally and back to back (slots past the row's end read the next row or
    // the slack behind the tile and are never consumed: every use below sits under a wave-uniform j < V guard).  Reads
    // inside those guards compile to one LDS round trip per element: 2 V serial latencies per row.
    {
        T raw[VP];
#pragma unroll
        for (int j = 0; j < VP; ++j) raw[j] = lrow[j];
        if constexpr (PHILOX) {
#pragma unroll
            for (int q4 = 0; q4 < VP / 4; ++q4) {
                if (4 * q4 < Vu) {  // (wave-uniform)
                    uint32_t w[4];
                    philox4x32_10((uint32_t)row, (uint32_t)((uint64_t)row >> 32), (uint32_t)q4, offset, (uint32_t)seed,
                                  (uint32_t)(seed >> 32), w);
#pragma unroll
                    for (int c = 0; c < 4; ++c) un[4 * q4 + c] = (float)(w[c] >> 8) * 5.9604644775390625e-8f;
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) un[4 * q4 + c] = 0.f;
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < VP; ++j) un[j] = urow[j];
        }
#pragma unroll
        for (int j = 0; j < VP; ++j) e[j] = to_f32<T>(raw[j]);
    }
    const bool euler = step == SEDD_STEP_EULER;
    const float invV = 1.0f / (float)V;
#pragma unroll
    for (int j = 0; j < VP; ++j) {
        t[j] = 0.f;
        if (j < V) {
            const float s = expf(e[j]);
            e[j] = s;
            const bool inc = euler ? j != xi : true;  // reverse_rate zeroes the diagonal before summing
            if (graph == SEDD_UNIFORM && euler) t[j] = inc ? invV * s : 0.f;
            else t[j] = inc ? s : 0.f;
        }
    }
    const float ssum = tree_sum<VP>(t);
    float e_p = 0.f, e_n = 0.f, one_m_en = 0.f, extra = 0.f, a_u = 0.f, tval = 0.f, tdiag = 0.f;
    if (!euler) {
        e_p = expf(cf);
        e_n = expf(-cf);
        one_m_en = 1.0f - e_n;
        if (graph == SEDD_ABSORB) {
            extra = (1.0f - e_p) * ssum;
        } else {
            a_u = (e_n - 1.0f) / ((float)V * e_n);
            tval = (1.0f * one_m_en) / (float)V;
#pragma unroll
            for (int j = 0; j < VP; ++j) t[j] = (j < V && j != xi) ? tval : 0.f;
            tdiag = 1.0f - tree_sum<VP>(t);
        }
    }
    ArgmaxScan am;
#pragma unroll
    for (int j = 0; j < VP; ++j) {
        if (j < Vu) {
            const float s = e[j];
            float p;
            if (euler) {
                float rj;
                if (graph == SEDD_ABSORB) rj = j == xi ? -ssum : s;  // masked row: transp_rate = 1 off the diagonal
                else rj = j == xi ? -ssum : invV * s;
                p = (j == xi ? 1.0f : 0.f) + cf * rj;
            } else if (graph == SEDD_ABSORB) {
                float stag = s * e_p;
                if (j == m) stag += extra;
                const float tr = j == m ? e_n + one_m_en : one_m_en;
                p = stag * tr;
            } else {
                const float stag = a_u * ssum + s / e_n;
                p = stag * (j == xi ? tdiag : tval);
            }
            const float v = gumbel_ratio_fast(p, un[j]);
            if (j == 0) am.first(v);
            else am.push(v, j);
        }
    }
    xnew[row] = am.result();
}

// the uniforms of sedd_predictor_step_philox as a tensor: thread -> (row, group of four slots)
__global__ void __launch_bounds__(TPB) philox_uniform_kernel(uint64_t seed, uint32_t offset, float* __restrict__ u,
                                                             int64_t M, int Vu, int nq) {
    const int64_t total = M * nq;
    for (int64_t t = (int64_t)blockIdx.x * TPB + threadIdx.x; t < total; t += (int64_t)gridDim.x * TPB) {
        const int64_t row = t / nq;
        const int q4 = (int)(t % nq);
        uint32_t w[4];
        philox4x32_10((uint32_t)row, (uint32_t)((uint64_t)row >> 32), (uint32_t)q4, offset, (uint32_t)seed,
                      (uint32_t)(seed >> 32), w);
        float* dst = u + row * Vu + 4 * q4;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (4 * q4 + c < Vu) dst[c] = (float)(w[c] >> 8) * 5.9604644775390625e-8f;
    }
}

extern "C" int sedd_philox_uniform(uint64_t seed, uint64_t offset, float* u, int64_t M, int Vu, sedd_stream_t stream) {
    SEDD_CHECK_ARG(M >= 0 && Vu > 0, "philox_uniform: bad sizes M=%ld Vu=%d", (long)M, Vu);
    SEDD_CHECK_ARG(offset < (1ull << 32), "philox_uniform: offset must fit 32 bits");
    SEDD_CHECK_ARG(M == 0 || u, "philox_uniform: null pointer");
    if (M == 0) return SEDD_OK;
    const int nq = (Vu + 3) / 4;
    const int64_t blocks = (M * nq + TPB - 1) / TPB;
    hipLaunchKernelGGL(philox_uniform_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(TPB), 0,
                       (hipStream_t)stream, seed, (uint32_t)offset, u, M, Vu, nq);
    SEDD_LAUNCH_CHECK();
    return SEDD_OK;
}

extern "C" int sedd_predictor_step_philox(int graph, int step, const void* logscore, int dtype, const int64_t* x,
                                          const float* coef, uint64_t seed, uint64_t offset, int64_t* xnew, int64_t M,
                                          int64_t L, int V, sedd_stream_t stream) {
    SEDD_CHECK_ARG(graph == SEDD_ABSORB || graph == SEDD_UNIFORM, "predictor_step_philox: unknown graph %d", graph);
    SEDD_CHECK_ARG(step >= SEDD_STEP_EULER && step <= SEDD_STEP_DENOISE, "predictor_step_philox: unknown step %d", step);
    SEDD_CHECK_ARG(dtype == SEDD_F32 || dtype == SEDD_BF16, "predictor_step_philox: unknown dtype %d", dtype);
    SEDD_CHECK_ARG(M == 0 || (logscore && x && coef && xnew), "predictor_step_philox: null pointer");
    SEDD_CHECK_ARG(M >= 0 && L > 0 && V > 1 && M % L == 0, "predictor_step_philox: bad sizes M=%ld L=%ld V=%d", (long)M,
                   (long)L, V);
    SEDD_CHECK_ARG(offset < (1ull << 32), "predictor_step_philox: offset must fit 32 bits");
    if (M == 0) return SEDD_OK;
    const int esz = dtype == SEDD_F32 ? 4 : 2;
    if (!rows_path_ok(V, esz, 0, logscore, nullptr, nullptr)) {
        sedd_set_error("predictor_step_philox: only the row-per-lane path (V <= 64, V * elem_size not a multiple of 32, "
                       "16-byte aligned log-score) generates its noise in-kernel; V=%d", V);
        return SEDD_ERR_UNSUPPORTED_SHAPE;
    }
    hipStream_t st = (hipStream_t)stream;
#define PRED_PHILOX(T, GR, ST)                                                                                        \
    SEDD_DISPATCH_VP(V, hipLaunchKernelGGL((predictor_step_rows_kernel<T, VP, GR, ST, true>),                         \
                                           dim3((unsigned)((M + RowsCfg<VP>::ROWS - 1) / RowsCfg<VP>::ROWS)),         \
                                           dim3(RowsCfg<VP>::ROWS), rows_lds(RowsCfg<VP>::ROWS, V * (int)sizeof(T)) + 256, \
                                           st, (const T*)logscore, x, coef, (const float*)nullptr, xnew, M, L, V, seed,   \
                                           (uint32_t)offset))
#define PRED_PHILOX_ST(T, GR)                                                        \
    do {                                                                             \
        if (step == SEDD_STEP_EULER) PRED_PHILOX(T, GR, SEDD_STEP_EULER);            \
        else if (step == SEDD_STEP_ANALYTIC) PRED_PHILOX(T, GR, SEDD_STEP_ANALYTIC); \
        else PRED_PHILOX(T, GR, SEDD_STEP_DENOISE);                                  \
    } while (0)
    if (dtype == SEDD_F32) {
        if (graph == SEDD_ABSORB) PRED_PHILOX_ST(float, SEDD_ABSORB);
        else PRED_PHILOX_ST(float, SEDD_UNIFORM);
    } else {
        if (graph == SEDD_ABSORB) PRED_PHILOX_ST(bf16_t, SEDD_ABSORB);
        else PRED_PHILOX_ST(bf16_t, SEDD_UNIFORM);
    }
#undef PRED_PHILOX_ST
#undef PRED_PHILOX
    SEDD_LAUNCH_CHECK();
    return SEDD_OK;
}

extern "C" int sedd_predictor_step(int graph, int step, const void* logscore, int dtype, const int64_t* x,
                                   const float* coef, const float* u, int64_t* xnew, int64_t M, int64_t L, int V,
                                   sedd_stream_t stream) {
    SEDD_CHECK_ARG(graph == SEDD_ABSORB || graph == SEDD_UNIFORM, "predictor_step: unknown graph %d", graph);
    SEDD_CHECK_ARG(step >= SEDD_STEP_EULER && step <= SEDD_STEP_DENOISE, "predictor_step: unknown step %d", step);
    SEDD_CHECK_ARG(dtype == SEDD_F32 || dtype == SEDD_BF16, "predictor_step: unknown dtype %d", dtype);
    SEDD_CHECK_ARG(M == 0 || (logscore && x && coef && u && xnew), "predictor_step: null pointer");
    SEDD_CHECK_ARG(M >= 0 && L > 0 && V > 1 && M % L == 0, "predictor_step: bad sizes M=%ld L=%ld V=%d", (long)M,
                   (long)L, V);
    if (M == 0) return SEDD_OK;
    hipStream_t st = (hipStream_t)stream;
    const int esz = dtype == SEDD_F32 ? 4 : 2;
    const int Vu = (graph == SEDD_ABSORB && step == SEDD_STEP_DENOISE) ? V - 1 : V;
    if (rows_path_ok(V, esz, Vu, logscore, u, nullptr)) {
#define PRED_ROWS(T, GR, ST)                                                                                          \
    SEDD_DISPATCH_VP(V, hipLaunchKernelGGL((predictor_step_rows_kernel<T, VP, GR, ST>),                               \
                                           dim3((unsigned)((M + RowsCfg<VP>::ROWS - 1) / RowsCfg<VP>::ROWS)),         \
                                           dim3(RowsCfg<VP>::ROWS),                                                   \
                                           rows_lds(RowsCfg<VP>::ROWS, V * (int)sizeof(T)) +                          \
                                               rows_lds(RowsCfg<VP>::ROWS, Vu * 4) + 256,                             \
                                           st, (const T*)logscore, x, coef, u, xnew, M, L, V))
#define PRED_ROWS_ST(T, GR)                                                    \
    do {                                                                       \
        if (step == SEDD_STEP_EULER) PRED_ROWS(T, GR, SEDD_STEP_EULER);        \
        else if (step == SEDD_STEP_ANALYTIC) PRED_ROWS(T, GR, SEDD_STEP_ANALYTIC); \
        else PRED_ROWS(T, GR, SEDD_STEP_DENOISE);                              \
    } while (0)
        if (dtype == SEDD_F32) {
            if (graph == SEDD_ABSORB) PRED_ROWS_ST(float, SEDD_ABSORB);
            else PRED_ROWS_ST(float, SEDD_UNIFORM);
        } else {
            if (graph == SEDD_ABSORB) PRED_ROWS_ST(bf16_t, SEDD_ABSORB);
            else PRED_ROWS_ST(bf16_t, SEDD_UNIFORM);
        }
#undef PRED_ROWS_ST
#undef PRED_ROWS
        SEDD_LAUNCH_CHECK();
        return SEDD_OK;
    }
    if (dtype == SEDD_F32) {
        SEDD_DISPATCH_GROUP(V, hipLaunchKernelGGL((predictor_step_kernel<float, G>),
                                                  dim3(sedd_grid_for(M, TPB / G)), dim3(TPB), 0, st, graph, step,
                                                  (const float*)logscore, x, coef, u, xnew, M, L, V));
    } else {
        SEDD_DISPATCH_GROUP(V, hipLaunchKernelGGL((predictor_step_kernel<bf16_t, G>),
                                                  dim3(sedd_grid_for(M, TPB / G)), dim3(TPB), 0, st, graph, step,
                                                  (const bf16_t*)logscore, x, coef, u, xnew, M, L, V));
    }
    SEDD_LAUNCH_CHECK();
    return SEDD_OK;
}

// ------------------------------------------------------------------------------ stand-alone Graph methods
// mode 0: transp_rate (graph_lib.py:218-221 / :139-145)      mode 1: reverse_rate (:79-93)
// mode 2: staggered_score (:264-269 / :162-165)              mode 3: transp_transition (:226-234 / :147-154)
template <int G>
__global__ void __launch_bounds__(TPB) graph_method_kernel(int graph, int mode, const int64_t* __restrict__ x,
                                                           const float* __restrict__ score,
                                                           const float* __restrict__ sig, float* __restrict__ out,
                                                           int64_t M, int64_t L, int V) {
    GroupIdx<G> g;
    const int m = V - 1;
    const float invV = 1.0f / (float)V;
    for (int64_t row = g.first; row < M; row += g.stride) {
        const int xi = x ? (int)x[row] : -1;
        const float sg = sig ? sig[row / L] : 0.f;
        const float* srow = score ? score + row * V : nullptr;
        float* orow = out + row * V;
        auto trate = [&](int j) -> float {
            if (graph == SEDD_ABSORB) return (j == xi ? -1.f : 0.f) + (xi == m ? 1.f : 0.f);
            return j == xi ? -((float)(V - 1)) / (float)V : invV;
        };
        if (mode == 0) {
            for (int j = g.lane; j < V; j += G) orow[j] = trate(j);
        } else if (mode == 1) {
            float s = 0.f;
            for (int j = g.lane; j < V; j += G) s += j == xi ? 0.f : trate(j) * srow[j];
            s = group_sum<G>(s);
            for (int j = g.lane; j < V; j += G) orow[j] = j == xi ? -s : trate(j) * srow[j];
        } else if (mode == 2) {
            float s = 0.f;
            for (int j = g.lane; j < V; j += G) s += srow[j];
            s = group_sum<G>(s);
            if (graph == SEDD_ABSORB) {
                const float e = expf(sg);
                const float extra = (1.0f - e) * s;
                for (int j = g.lane; j < V; j += G) {
                    float v = srow[j] * e;
                    if (j == m) v += extra;
                    orow[j] = v;
                }
            } else {
                const float w = expf(-sg);
                const float a = (w - 1.0f) / ((float)V * w);
                for (int j = g.lane; j < V; j += G) orow[j] = a * s + srow[j] / w;
            }
        } else {
            const float en = expf(-sg);
            const float om = 1.0f - en;
            if (graph == SEDD_ABSORB) {
                for (int j = g.lane; j < V; j += G)
                    orow[j] = (j == xi ? en : 0.f) + (xi == m ? om : 0.f);
            } else {
                const float tval = (1.0f * om) / (float)V;
                float ts = 0.f;
                for (int j = g.lane; j < V; j += G) ts += j != xi ? tval : 0.f;
                const float tdiag = 1.0f - group_sum<G>(ts);
                for (int j = g.lane; j < V; j += G) orow[j] = j == xi ? tdiag : tval;
            }
        }
    }
}

static int launch_graph_method(int graph, int mode, const int64_t* x, const float* score, const float* sig,
                               float* out, int64_t M, int64_t L, int V, sedd_stream_t stream, const char* name) {
    SEDD_CHECK_ARG(graph == SEDD_ABSORB || graph == SEDD_UNIFORM, "%s: unknown graph %d", name, graph);
    SEDD_CHECK_ARG(M == 0 || out, "%s: out is null", name);
    SEDD_CHECK_ARG(M >= 0 && L > 0 && V > 1 && M % L == 0, "%s: bad sizes M=%ld L=%ld V=%d", name, (long)M, (long)L,
                   V);
    if (M == 0) return SEDD_OK;
    SEDD_DISPATCH_GROUP(V, hipLaunchKernelGGL((graph_method_kernel<G>), dim3(sedd_grid_for(M, TPB / G)), dim3(TPB),
                                              0, (hipStream_t)stream, graph, mode, x, score, sig, out, M, L, V));
    SEDD_LAUNCH_CHECK();
    return SEDD_OK;
}

extern "C" int sedd_transp_rate(int graph, const int64_t* x, float* out, int64_t M, int V, sedd_stream_t stream) {
    SEDD_CHECK_ARG(M == 0 || (x), "transp_rate: x is null");
    return launch_graph_method(graph, 0, x, nullptr, nullptr, out, M, 1, V, stream, "transp_rate");
}
extern "C" int sedd_reverse_rate(int graph, const int64_t* x, const float* score, float* out, int64_t M, int V,
                                 sedd_stream_t stream) {
    SEDD_CHECK_ARG(M == 0 || (x && score), "reverse_rate: null pointer");
    return launch_graph_method(graph, 1, x, score, nullptr, out, M, 1, V, stream, "reverse_rate");
}
extern "C" int sedd_staggered_score(int graph, const float* score, const float* dsigma, float* out, int64_t M,
                                    int64_t L, int V, sedd_stream_t stream) {
    SEDD_CHECK_ARG(M == 0 || (score && dsigma), "staggered_score: null pointer");
    return launch_graph_method(graph, 2, nullptr, score, dsigma, out, M, L, V, stream, "staggered_score");
}
extern "C" int sedd_transp_transition(int graph, const int64_t* x, const float* sigma, float* out, int64_t M,
                                      int64_t L, int V, sedd_stream_t stream) {
    SEDD_CHECK_ARG(M == 0 || (x && sigma), "transp_transition: null pointer");
    return launch_graph_method(graph, 3, x, nullptr, sigma, out, M, L, V, stream, "transp_transition");
}
