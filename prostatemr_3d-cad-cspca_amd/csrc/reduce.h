// reduce.h -- per-(n,c) reductions over the voxel axis of an (N,V,C) NDHWC tensor.
//
// Stage 1: grid (chunks, N). Each 256-thread block owns `chunkV` consecutive voxels of one sample and
//          ALL channels; lanes run along C (coalesced, channel-contiguous HBM reads), partial sums stay
//          in registers, then one LDS pass folds the voxel-sub-lanes. Output: fp32 partials
//          [N][nchunks][C][NS] -- deterministic (no atomics).
// Stage 2: finalize kernel folds the chunks in fp64 (tiny).
#pragma once
#include "common.h"

#include <stdlib.h>
#define M1_RED_THREADS 256

static inline int m1_red_chunkV(long long V, int C, int N) {
    // >= ~16 elements per thread, and about 512 blocks per launch (2 per CU): every block ends with a fold of its NS x VEC
    // register sums through shuffles + LDS that costs as much as ~10 voxel iterations, so more, smaller chunks lose
    // (measured at batch 2: 1024 chunks per sample -> 256 = -34 % on the SE backward reduction, -3.7 % per step)
    long long per_block = (long long)M1_RED_THREADS * 16 / (C < 256 ? (C < 1 ? 1 : C) : 256);
    if (per_block < 16) per_block = 16;
    long long chunk = per_block;
    int tgt = M1_CFG("M1_RED_BLOCKS", 512); { if (tgt < 1) tgt = 1; }
    int maxc = tgt / (N < 1 ? 1 : N); if (maxc < 32) maxc = 32;
    const long long cap = cdiv_ll(V, maxc);
    if (chunk < cap) chunk = cap;
    if (chunk > V) chunk = V;
    return (int)chunk;
}
static inline int m1_red_nchunks(long long V, int C, int N) { return (int)cdiv_ll(V, m1_red_chunkV(V, C, N)); }

// Block barrier behind LDS traffic only.  __syncthreads() also waits vmcnt(0): a barrier that follows the partial-row stores of a fold
// step waits for their round trip to memory (~3 us per step, measured: the 5-sum SE backward reduction of a (4,10,20,20,128) tensor
// took 21 us next to a 9.5 us apply pass over the same data).
__device__ __forceinline__ void m1_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

static inline int m1_pow2_ge(int c) { int p = 1; while (p < c) p <<= 1; return p; }

template <int NS> struct M1ParamOut { float* ptr[NS]; int acc[NS]; };

// Functor contract:  __device__ void operator()(int n, long long v, int c, float* acc) const;   acc[NS] += ...
template <int NS, typename F>
__global__ void __launch_bounds__(M1_RED_THREADS) m1_reduce_nc_kernel(F f, long long V, int C, int chunkV,
                                                                      int nchunks, float* __restrict__ partial) {
    __shared__ float red[M1_RED_THREADS * NS];
    const int n = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    const long long v0 = (long long)chunk * chunkV;
    long long v1 = v0 + chunkV; if (v1 > V) v1 = V;
    int cpad = 1; while (cpad < C && cpad < M1_RED_THREADS) cpad <<= 1;   // lanes along C
    const int vs = tid / cpad, nvs = M1_RED_THREADS / cpad, cl = tid % cpad;
    for (int cbase = 0; cbase < C; cbase += cpad) {
        const int c = cbase + cl;
        float acc[NS];
#pragma unroll
        for (int k = 0; k < NS; ++k) acc[k] = 0.f;
        if (c < C)
            for (long long v = v0 + vs; v < v1; v += nvs) f(n, v, c, acc);
#pragma unroll
        for (int k = 0; k < NS; ++k) red[tid * NS + k] = acc[k];
        __syncthreads();
        if (vs == 0 && c < C) {
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                float s = 0.f;
                for (int j = 0; j < nvs; ++j) s += red[(j * cpad + cl) * NS + k];
                partial[(((size_t)n * nchunks + chunk) * C + c) * NS + k] = s;
            }
        }
        m1_lds_barrier();                                  // (not __syncthreads: the stores above need not have landed)
    }
}

// out[n][c][k] = sum_chunks partial (fp64 accumulate) -> float.  One block per (n,c): threads stride the chunks.
// stats_V > 0 (NS == 2 only): writes {mean, rstd} computed in fp64 from (sum, sum of squares) instead.
template <int NS>
__global__ void __launch_bounds__(256) m1_reduce_finalize_kernel(const float* __restrict__ partial, int N, int C, int nchunks,
                                                                 float* __restrict__ out, long long stats_V, float eps,
                                                                 int accumulate, float* __restrict__ out2 = nullptr, int csplit = 0) {
    // one BLOCK per (n,c): the fold is a chain of dependent cache-line loads, 256 lanes keep it 4x shorter than a wave
    __shared__ double red[4][NS];
    const int i = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = i / C, c = i % C;
    double s[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) s[k] = 0.0;
    for (int j = threadIdx.x; j < nchunks; j += 256)
#pragma unroll
        for (int k = 0; k < NS; ++k) s[k] += (double)partial[(((size_t)n * nchunks + j) * C + c) * NS + k];
#pragma unroll
    for (int k = 0; k < NS; ++k) { s[k] = wave_sum_d(s[k]); if (lane == 0) red[wave][k] = s[k]; }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < NS; ++k) s[k] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
        // channels >= csplit (csplit > 0) belong to a second output tensor of C - csplit channels
        float* o = out + (size_t)i * NS;
        if (csplit > 0) o = c < csplit ? out + ((size_t)n * csplit + c) * NS : out2 + ((size_t)n * (C - csplit) + (c - csplit)) * NS;
        if (stats_V > 0 && NS == 2) {
            const double mean = s[0] / (double)stats_V;
            double var = s[NS - 1] / (double)stats_V - mean * mean;
            if (var < 0.0) var = 0.0;
            o[0] = (float)mean;
            o[NS - 1] = (float)(1.0 / sqrt(var + (double)eps));
        } else {
#pragma unroll
            for (int k = 0; k < NS; ++k) o[k] = (accumulate ? o[k] : 0.f) + (float)s[k];
        }
    }
}
template <int NS>
static inline int m1_reduce_finalize_launch(const float* partial, int N, int C, int nchunks, float* out, long long stats_V,
                                            float eps, hipStream_t st, int accumulate = 0, float* out2 = nullptr, int csplit = 0) {
    hipLaunchKernelGGL((m1_reduce_finalize_kernel<NS>), dim3(N * C), dim3(256), 0, st, partial, N, C, nchunks, out,
                       stats_V, eps, accumulate, out2, csplit);
    return m1_check_launch();
}

// Same fold, plus the parameter gradients that are sums over the batch of the per-sample sums (gamma/beta of an
// InstanceNorm, the SE gate input): ONE WAVE per channel walks the samples.  pout[k] (nullable) receives
// sum_n out[n][c][k], added to its previous value when pacc[k] != 0.
template <int NS>
__global__ void __launch_bounds__(256) m1_reduce_finalize_params_kernel(const float* __restrict__ partial, int N, int C, int nchunks,
                                                                        float* __restrict__ out, M1ParamOut<NS> po) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;
    double tot[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) tot[k] = 0.0;
    // up to 4 samples at a time: their partial rows are loaded together (independent loads in flight), one sample at a time the
    // fold was N dependent round trips to L2 / HBM (5 - 8 us per launch at the stacked batch of 4, 62 launches per C3 step)
    constexpr int NB = 4;
    for (int n0 = 0; n0 < N; n0 += NB) {
        double s[NB][NS];
#pragma unroll
        for (int q = 0; q < NB; ++q)
#pragma unroll
            for (int k = 0; k < NS; ++k) s[q][k] = 0.0;
#pragma unroll 2
        for (int j = lane; j < nchunks; j += 64) {
#pragma unroll
            for (int q = 0; q < NB; ++q)
                if (n0 + q < N) {
#pragma unroll
                    for (int k = 0; k < NS; ++k) s[q][k] += (double)partial[(((size_t)(n0 + q) * nchunks + j) * C + c) * NS + k];
                }
        }
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            if (n0 + q < N) {                                  // (wave-uniform)
#pragma unroll
                for (int k = 0; k < NS; ++k) { s[q][k] = wave_sum_d(s[q][k]); tot[k] += (double)(float)s[q][k]; }
                if (lane == 0) {
#pragma unroll
                    for (int k = 0; k < NS; ++k) out[((size_t)(n0 + q) * C + c) * NS + k] = (float)s[q][k];
                }
            }
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < NS; ++k)
            if (po.ptr[k]) po.ptr[k][c] = (po.acc[k] ? po.ptr[k][c] : 0.f) + (float)tot[k];
    }
}
// The same for MANY partial rows per sample (the per-tile rows of a res0 / res1 conv epilogue: up to ~4,000): one BLOCK per channel,
// 256 lanes stride the rows of up to 4 samples at a time.  One wave per channel walked 4,000 rows in 62 dependent L2 round trips with
// 2 - 8 blocks on the whole chip (25 us at C = 8, 10.6 us at C = 16; round 6).
template <int NS>
__global__ void __launch_bounds__(256) m1_reduce_finalize_params_wide_kernel(const float* __restrict__ partial, int N, int C, int nchunks,
                                                                             float* __restrict__ out, M1ParamOut<NS> po) {
    constexpr int NB = 4;
    __shared__ double red[4][NB][NS];
    const int c = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double tot[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) tot[k] = 0.0;
    for (int n0 = 0; n0 < N; n0 += NB) {
        double s[NB][NS];
#pragma unroll
        for (int q = 0; q < NB; ++q)
#pragma unroll
            for (int k = 0; k < NS; ++k) s[q][k] = 0.0;
        for (int j = threadIdx.x; j < nchunks; j += 256) {
#pragma unroll
            for (int q = 0; q < NB; ++q)
                if (n0 + q < N) {
#pragma unroll
                    for (int k = 0; k < NS; ++k) s[q][k] += (double)partial[(((size_t)(n0 + q) * nchunks + j) * C + c) * NS + k];
                }
        }
#pragma unroll
        for (int q = 0; q < NB; ++q)
#pragma unroll
            for (int k = 0; k < NS; ++k) { s[q][k] = wave_sum_d(s[q][k]); if (lane == 0) red[wave][q][k] = s[q][k]; }
        __syncthreads();
        if (threadIdx.x == 0) {
#pragma unroll
            for (int q = 0; q < NB; ++q)
                if (n0 + q < N) {
#pragma unroll
                    for (int k = 0; k < NS; ++k) {
                        const float v = (float)((red[0][q][k] + red[1][q][k]) + (red[2][q][k] + red[3][q][k]));
                        out[((size_t)(n0 + q) * C + c) * NS + k] = v; tot[k] += (double)v;
                    }
                }
        }
        m1_lds_barrier();
    }
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < NS; ++k)
            if (po.ptr[k]) po.ptr[k][c] = (po.acc[k] ? po.ptr[k][c] : 0.f) + (float)tot[k];
    }
}
template <int NS>
static inline int m1_reduce_finalize_params_launch(const float* partial, int N, int C, int nchunks, float* out,
                                                   const M1ParamOut<NS>& po, hipStream_t st) {
    if (nchunks > M1_CFG("M1_FINP_WIDE", 128)) {
        hipLaunchKernelGGL((m1_reduce_finalize_params_wide_kernel<NS>), dim3(C), dim3(256), 0, st, partial, N, C, nchunks, out, po);
        return m1_check_launch();
    }
    hipLaunchKernelGGL((m1_reduce_finalize_params_kernel<NS>), dim3((C + 3) / 4), dim3(256), 0, st, partial, N, C, nchunks, out, po);
    return m1_check_launch();
}

template <typename F, typename = void> struct M1RedUnroll { static constexpr int value = 2; };
template <typename F> struct M1RedUnroll<F, decltype((void)F::kUnroll)> { static constexpr int value = F::kUnroll; };

// Vector variant: a lane owns VEC consecutive channels (one 16-byte load per tensor per voxel) instead of one.
// Functor contract:  static constexpr int kVec;  __device__ void vec(int n, long long v, int c0, float (*acc)[kVec]) const;
// (body: block (n, chunk) of the reduction -- the batched launches of norm.hip walk several reductions with it)
template <int NS, int VEC, typename F>
__device__ __forceinline__ void m1_reduce_nc_vec_body(const F& f, long long V, int C, int chunkV, int nchunks, float* __restrict__ partial,
                                                      int n, int chunk, float (*red)[M1_RED_THREADS * VEC]) {
    const int tid = threadIdx.x;
    const long long v0 = (long long)chunk * chunkV;
    long long v1 = v0 + chunkV; if (v1 > V) v1 = V;
    const int cg = C / VEC;
    int cpad = 1; while (cpad < cg && cpad < M1_RED_THREADS) cpad <<= 1;   // lanes along the channel groups
    const int vs = tid / cpad, nvs = M1_RED_THREADS / cpad, cl = tid % cpad;
    for (int gbase = 0; gbase < cg; gbase += cpad) {
        const int gi = gbase + cl;
        float acc[NS][VEC];
#pragma unroll
        for (int k = 0; k < NS; ++k)
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[k][e] = 0.f;
        if (gi < cg) {
            // M1RedUnroll<F>: voxels in flight per lane (independent 16-byte loads issued back to back; the streaming
            // functors run at 2-3 waves per SIMD, so the loads of one voxel alone cannot cover the HBM latency)
            constexpr int U = M1RedUnroll<F>::value;
            long long v = v0 + vs;
            if constexpr (U > 1) {
                for (; v + (long long)(U - 1) * nvs < v1; v += (long long)U * nvs) {
#pragma unroll
                    for (int u = 0; u < U; ++u) f.vec(n, v + (long long)u * nvs, gi * VEC, acc);
                }
            }
            for (; v < v1; v += nvs) f.vec(n, v, gi * VEC, acc);
        }
        // fold the voxel sub-lanes: xor-shuffles inside a wave (lanes cpad apart share a channel group), then the
        // 4 waves (or, for >= 64 channel groups, the voxel sub-lane rows) through LDS -- every sum in its own LDS plane, one barrier,
        // and the row-0 lanes write their NS x VEC results as 16-byte stores (round 6: the fold used to run one sum at a time with
        // two __syncthreads each, and every barrier waited for the 4-byte partial stores in front of it)
        const int wcol = cpad < 64 ? cpad : 64;                      // distinct channel groups per wave
        const int rows = M1_RED_THREADS / (cpad < 64 ? 64 : cpad);
        const int row = cpad < 64 ? (tid >> 6) : vs;
        const bool writer = cpad >= 64 || (tid & 63) < cpad;
        // (the shuffle distance is the OUTER loop: the NS x VEC exchanges of one distance are independent and pipeline; with the
        //  distance loop inside, each of the 40 sums of the SE backward ran its own serial chain of ds_bpermute + wait -- 80 - 120
        //  dependent LDS round trips, ~10 us of a 20 us launch on the deep levels' tensors)
        if (wcol < 64) {
            for (int o = 32; o >= wcol; o >>= 1) {
#pragma unroll
                for (int k = 0; k < NS; ++k)
#pragma unroll
                    for (int e = 0; e < VEC; ++e) acc[k][e] += __shfl_xor(acc[k][e], o, 64);
            }
        }
        if (writer) {
#pragma unroll
            for (int k = 0; k < NS; ++k)
#pragma unroll
                for (int e = 0; e < VEC; ++e) red[k][(row * cpad + cl) * VEC + e] = acc[k][e];
        }
        m1_lds_barrier();
        if (writer && row == 0 && gi < cg) {
            float o[VEC * NS];                                        // [e][k]: the partial row's layout for channels gi*VEC..
#pragma unroll
            for (int q = 0; q < VEC * NS; ++q) o[q] = 0.f;
            for (int j = 0; j < rows; ++j) {                          // (rows outside: the NS x VEC reads of one row are independent)
#pragma unroll
                for (int k = 0; k < NS; ++k)
#pragma unroll
                    for (int e = 0; e < VEC; ++e) o[e * NS + k] += red[k][(j * cpad + cl) * VEC + e];
            }
            float* dst = partial + (((size_t)n * nchunks + chunk) * C + (size_t)gi * VEC) * NS;
            if constexpr ((VEC * NS) % 4 == 0) {                      // (VEC is 4 or 8: dst is 16-byte aligned)
#pragma unroll
                for (int q = 0; q < VEC * NS; q += 4) *reinterpret_cast<float4*>(dst + q) = make_float4(o[q], o[q + 1], o[q + 2], o[q + 3]);
            } else {
#pragma unroll
                for (int q = 0; q < VEC * NS; ++q) dst[q] = o[q];
            }
        }
        if (gbase + cpad < cg) m1_lds_barrier();
    }
}
template <int NS, int VEC, typename F>
__global__ void __launch_bounds__(M1_RED_THREADS) m1_reduce_nc_vec_kernel(F f, long long V, int C, int chunkV,
                                                                          int nchunks, float* __restrict__ partial) {
    __shared__ __attribute__((aligned(16))) float red[NS][M1_RED_THREADS * VEC];   // all NS sums at once: ONE barrier per fold
    m1_reduce_nc_vec_body<NS, VEC, F>(f, V, C, chunkV, nchunks, partial, blockIdx.y, blockIdx.x, red);
}

template <typename F, typename = void> struct M1RedVec { static constexpr int value = 0; };
template <typename F> struct M1RedVec<F, decltype((void)F::kVec)> { static constexpr int value = F::kVec; };

template <int NS, typename F>
static inline int m1_reduce_nc_launch(const F& f, int N, long long V, int C, float* partial, hipStream_t st) {
    const int chunkV = m1_red_chunkV(V, C, N), nchunks = m1_red_nchunks(V, C, N);
    dim3 grid(nchunks, N);
    constexpr int VEC = M1RedVec<F>::value;
    if constexpr (VEC > 0) {
        if (C % VEC == 0) {
            hipLaunchKernelGGL((m1_reduce_nc_vec_kernel<NS, VEC, F>), grid, dim3(M1_RED_THREADS), 0, st, f, V, C, chunkV, nchunks,
                               partial);
            return m1_check_launch();
        }
    }
    hipLaunchKernelGGL((m1_reduce_nc_kernel<NS, F>), grid, dim3(M1_RED_THREADS), 0, st, f, V, C, chunkV, nchunks,
                       partial);
    return m1_check_launch();
}
