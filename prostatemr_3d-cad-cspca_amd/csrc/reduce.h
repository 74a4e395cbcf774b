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

static inline int m1_pow2_ge(int c) { int p = 1; while (p < c) p <<= 1; return p; }

// ---- finalize by the LAST block instead of a second launch (round 6) ---------------------------------------------------------
// The ~150 finalize launches of a C3 step (5-11 us each, a few dozen blocks, alone on their stream between two dependent kernels)
// fold the partial rows a reduction or a conv epilogue just wrote.  With a ticket per sample the block that arrives last does the
// fold itself: every block writes its rows WRITE-THROUGH (agent-scope `sc1` stores: the bytes leave the XCD's L2, no cache-wide
// write-back), drains them (`s_waitcnt vmcnt(0)`), takes a ticket (one relaxed agent-scope atomic per block, self-resetting:
// atomicInc wraps at nblocks - 1); only the last arriver pays an L1 invalidate (agent acquire) and reads the rows.  Round 3's
// version of this idea (M1_RED_LASTBLOCK) cost +4 ms per step because EVERY block ran a __threadfence(): a write-back of its
// XCD's whole dirty L2 plus an L1 invalidate (MI355X_MICROARCH.md, "Workgroup dispatch ... inter-workgroup visibility").
// The fold order is fixed (rows in index order per lane group, lane groups in index order): results do not depend on which
// block arrives last.  Tickets come from a caller-owned, zero-initialised pool (m1_tickets_set); without one the finalize
// launches stay.
unsigned* m1_ticket_take(int n);          // config.hip: n consecutive counters of the registered pool (nullptr: none / M1_RED_TAIL=0)

template <int NS> struct M1ParamOut { float* ptr[NS]; int acc[NS]; };

// what the last block does with the folded sums s[NS] of (n, c):
//   mode 1: out[n][c][k] (+ accumulate), or {mean, rstd} when stats_V > 0 (NS == 2); channels >= csplit go to out2 (m1_reduce_finalize_kernel)
//   mode 2: out[n][c][k] = s (per-sample sums, float) and po.ptr[k][c] (+)= sum_n out[n][c][k]            (m1_reduce_finalize_params_kernel)
//   mode 3: as 2, only the batch sums are wanted (out = scratch [N][C][NS])                                (column sums: bias gradients)
template <int NS> struct M1Fin {
    int mode; unsigned* tickets;          // tickets[0..N-1] per sample, tickets[N] for the batch fold
    float* out; float* out2; int csplit; long long stats_V; float eps; int accumulate;
    M1ParamOut<NS> po;
};

__device__ __forceinline__ void m1_st_wt(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Call from ALL threads of a 256-thread block after its partial rows were stored with m1_st_wt.  partial: [N][nrows][C][NS].
template <int NS>
__device__ __forceinline__ void m1_reduce_tail(const M1Fin<NS>& fin, const float* __restrict__ partial, int n, int N, int C, int nrows,
                                               double* __restrict__ dred /* LDS, >= 256 doubles */) {
    __shared__ unsigned s_old;
    const int tid = threadIdx.x;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) s_old = atomicInc(fin.tickets + n, (unsigned)(nrows - 1));
    __syncthreads();
    if (s_old != (unsigned)(nrows - 1)) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    int CW = 1; while (CW < C && CW < 256) CW <<= 1;
    const int G = 256 / CW, cl = tid % CW, g = tid / CW;
    const float* base = partial + (size_t)n * nrows * C * NS;
    for (int c0 = 0; c0 < C; c0 += CW) {
        const int c = c0 + cl;
        double s[NS];
#pragma unroll
        for (int k = 0; k < NS; ++k) s[k] = 0.0;
        if (c < C) {
#pragma unroll 4
            for (int j = g; j < nrows; j += G) {
                const float* r = base + ((size_t)j * C + c) * NS;
#pragma unroll
                for (int k = 0; k < NS; ++k) s[k] += (double)r[k];
            }
        }
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            if (G > 1) {
                dred[tid] = s[k];
                __syncthreads();
                if (g == 0) { double t = 0.0; for (int q = 0; q < G; ++q) t += dred[q * CW + cl]; s[k] = t; }
                __syncthreads();
            }
        }
        if (g == 0 && c < C) {
            if (fin.mode == 1) {
                float* o = fin.out + ((size_t)n * C + c) * NS;
                if (fin.csplit > 0) o = c < fin.csplit ? fin.out + ((size_t)n * fin.csplit + c) * NS : fin.out2 + ((size_t)n * (C - fin.csplit) + (c - fin.csplit)) * NS;
                if (fin.stats_V > 0 && NS == 2) {
                    const double mean = s[0] / (double)fin.stats_V;
                    double var = s[NS - 1] / (double)fin.stats_V - mean * mean;
                    if (var < 0.0) var = 0.0;
                    o[0] = (float)mean; o[NS - 1] = (float)(1.0 / sqrt(var + (double)fin.eps));
                } else {
#pragma unroll
                    for (int k = 0; k < NS; ++k) o[k] = (fin.accumulate ? o[k] : 0.f) + (float)s[k];
                }
            } else {
#pragma unroll
                for (int k = 0; k < NS; ++k) m1_st_wt(fin.out + ((size_t)n * C + c) * NS + k, (float)s[k]);
            }
        }
    }
    if (fin.mode == 1) return;
    // batch fold: the last of the N per-sample finishers sums the per-sample (float) sums in sample order
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) s_old = N > 1 ? atomicInc(fin.tickets + N, (unsigned)(N - 1)) : 0u;
    __syncthreads();
    if (s_old != (unsigned)(N - 1)) return;
    if (N > 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    for (int c = tid; c < C; c += 256) {
        double tot[NS];
#pragma unroll
        for (int k = 0; k < NS; ++k) tot[k] = 0.0;
        for (int q = 0; q < N; ++q)
#pragma unroll
            for (int k = 0; k < NS; ++k) tot[k] += (double)fin.out[((size_t)q * C + c) * NS + k];
#pragma unroll
        for (int k = 0; k < NS; ++k)
            if (fin.po.ptr[k]) fin.po.ptr[k][c] = (fin.po.acc[k] ? fin.po.ptr[k][c] : 0.f) + (float)tot[k];
    }
}

// Functor contract:  __device__ void operator()(int n, long long v, int c, float* acc) const;   acc[NS] += ...
template <int NS, typename F>
__global__ void __launch_bounds__(M1_RED_THREADS) m1_reduce_nc_kernel(F f, long long V, int C, int chunkV,
                                                                      int nchunks, float* __restrict__ partial, M1Fin<NS> fin) {
    __shared__ __attribute__((aligned(8))) float red[M1_RED_THREADS * (NS < 2 ? 2 : NS)];
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
                float* dst = partial + (((size_t)n * nchunks + chunk) * C + c) * NS + k;
                if (fin.mode) m1_st_wt(dst, s); else *dst = s;
            }
        }
        __syncthreads();
    }
    if (fin.mode) m1_reduce_tail<NS>(fin, partial, n, (int)gridDim.y, C, nchunks, reinterpret_cast<double*>(red));
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
template <int NS>
static inline int m1_reduce_finalize_params_launch(const float* partial, int N, int C, int nchunks, float* out,
                                                   const M1ParamOut<NS>& po, hipStream_t st) {
    hipLaunchKernelGGL((m1_reduce_finalize_params_kernel<NS>), dim3((C + 3) / 4), dim3(256), 0, st, partial, N, C, nchunks, out, po);
    return m1_check_launch();
}

template <typename F, typename = void> struct M1RedUnroll { static constexpr int value = 1; };
template <typename F> struct M1RedUnroll<F, decltype((void)F::kUnroll)> { static constexpr int value = F::kUnroll; };

// Vector variant: a lane owns VEC consecutive channels (one 16-byte load per tensor per voxel) instead of one.
// Functor contract:  static constexpr int kVec;  __device__ void vec(int n, long long v, int c0, float (*acc)[kVec]) const;
template <int NS, int VEC, typename F>
__global__ void __launch_bounds__(M1_RED_THREADS) m1_reduce_nc_vec_kernel(F f, long long V, int C, int chunkV,
                                                                          int nchunks, float* __restrict__ partial, M1Fin<NS> fin) {
    __shared__ __attribute__((aligned(8))) float red[M1_RED_THREADS * (VEC < 2 ? 2 : VEC)];   // one sum at a time (NS passes)
    const int n = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
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
        // 4 waves (or, for >= 64 channel groups, the voxel sub-lane rows) through LDS, one sum at a time
        const int wcol = cpad < 64 ? cpad : 64;                      // distinct channel groups per wave
        const int rows = M1_RED_THREADS / (cpad < 64 ? 64 : cpad);
        const int row = cpad < 64 ? (tid >> 6) : vs;
        const bool writer = cpad >= 64 || (tid & 63) < cpad;
#pragma unroll
        for (int k = 0; k < NS; ++k) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                float s = acc[k][e];
                for (int o = 32; o >= wcol && wcol < 64; o >>= 1) s += __shfl_xor(s, o, 64);
                if (writer) red[(row * cpad + cl) * VEC + e] = s;
            }
            __syncthreads();
            if (writer && row == 0 && gi < cg) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    float s = 0.f;
                    for (int j = 0; j < rows; ++j) s += red[(j * cpad + cl) * VEC + e];
                    float* dst = partial + (((size_t)n * nchunks + chunk) * C + gi * VEC + e) * NS + k;
                    if (fin.mode) m1_st_wt(dst, s); else *dst = s;
                }
            }
            __syncthreads();
        }
    }
    if (fin.mode) m1_reduce_tail<NS>(fin, partial, n, (int)gridDim.y, C, nchunks, reinterpret_cast<double*>(red));
}

template <typename F, typename = void> struct M1RedVec { static constexpr int value = 0; };
template <typename F> struct M1RedVec<F, decltype((void)F::kVec)> { static constexpr int value = F::kVec; };

// `fin` (optional): the finalize the caller would launch next.  *fused = true: the reduction's last blocks did it (tickets were
// available), the caller skips its finalize launch.
template <int NS, typename F>
static inline int m1_reduce_nc_launch(const F& f, int N, long long V, int C, float* partial, hipStream_t st,
                                      const M1Fin<NS>* fin = nullptr, bool* fused = nullptr) {
    const int chunkV = m1_red_chunkV(V, C, N), nchunks = m1_red_nchunks(V, C, N);
    dim3 grid(nchunks, N);
    M1Fin<NS> fz{};
    if (fused) *fused = false;
    if (fin && fin->mode && fused) {
        unsigned* t = m1_ticket_take(N + 1);
        if (t) { fz = *fin; fz.tickets = t; *fused = true; }
    }
    constexpr int VEC = M1RedVec<F>::value;
    if constexpr (VEC > 0) {
        if (C % VEC == 0) {
            hipLaunchKernelGGL((m1_reduce_nc_vec_kernel<NS, VEC, F>), grid, dim3(M1_RED_THREADS), 0, st, f, V, C, chunkV, nchunks,
                               partial, fz);
            return m1_check_launch();
        }
    }
    hipLaunchKernelGGL((m1_reduce_nc_kernel<NS, F>), grid, dim3(M1_RED_THREADS), 0, st, f, V, C, chunkV, nchunks,
                       partial, fz);
    return m1_check_launch();
}
// the stand-alone reduction followed by its finalize, fused into one launch when tickets are available
template <int NS> static inline M1Fin<NS> m1_fin_out(float* out, long long stats_V, float eps, int accumulate = 0, float* out2 = nullptr, int csplit = 0) {
    M1Fin<NS> f{}; f.mode = 1; f.out = out; f.out2 = out2; f.csplit = csplit; f.stats_V = stats_V; f.eps = eps; f.accumulate = accumulate; return f;
}
template <int NS> static inline M1Fin<NS> m1_fin_params(float* sums, const M1ParamOut<NS>& po, int mode = 2) {
    M1Fin<NS> f{}; f.mode = mode; f.out = sums; f.po = po; return f;
}
