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
    static int tgt = -1; if (tgt < 0) { const char* e = getenv("M1_RED_BLOCKS"); tgt = e ? atoi(e) : 512; if (tgt < 1) tgt = 1; }
    int maxc = tgt / (N < 1 ? 1 : N); if (maxc < 32) maxc = 32;
    const long long cap = cdiv_ll(V, maxc);
    if (chunk < cap) chunk = cap;
    if (chunk > V) chunk = V;
    return (int)chunk;
}
static inline int m1_red_nchunks(long long V, int C, int N) { return (int)cdiv_ll(V, m1_red_chunkV(V, C, N)); }

static inline int m1_pow2_ge(int c) { int p = 1; while (p < c) p <<= 1; return p; }

// Same fold as m1_reduce_finalize_kernel / m1_reduce_finalize_params_kernel below, but done by the LAST block of the reduction
// to finish (a ticket per sample, then one over the samples for the parameter sums): the stand-alone finalize launches are
// ~150 kernels of 3-4 us per C3 step on the critical chain between a reduction and its consumer.  Fixed summation order (row
// slices, fp64), so the result does not depend on which block comes last.  mode 0: off (the caller launches the finalize kernel).
// Correct (the whole GPU suite passes with M1_RED_LASTBLOCK=1) but SLOWER than the launches it removes: see m1_red_lastblock_on.
template <int NS> struct M1ParamOut { float* ptr[NS]; int acc[NS]; };
template <int NS> struct M1RedFin {
    int mode;                  // 1: out[n][c][k] (or {mean, rstd} when stats_V > 0), 2: + parameter sums over the samples (po)
    int joint;                 // mode 1: all N * nchunks partial rows are ONE sample (column sums)
    float* out; long long stats_V; float eps; int accumulate; float* out2; int csplit;
    M1ParamOut<NS> po;
    unsigned* cnt;             // N + 1 tickets, zero on entry and zero again on exit
};
#define M1_RED_NCNT 4096
static __device__ unsigned m1_red_tickets[M1_RED_NCNT];      // (one array per translation unit; zero-initialised with the module)
static inline unsigned* m1_red_ticket_slot(int n) {          // n consecutive tickets; a slot comes round again after ~400 launches
    static unsigned* base = nullptr; static unsigned next = 0;
    if (!base && hipGetSymbolAddress((void**)&base, HIP_SYMBOL(m1_red_tickets)) != hipSuccess) return nullptr;
    if (next + (unsigned)n > M1_RED_NCNT) next = 0;
    unsigned* p = base + next; next += (unsigned)n;
    return p;
}
static inline bool m1_red_lastblock_on() {
    // OFF by default: measured +4.0 ms per C3 step (27.0 -> 31.0) and +1.6 ms per C2 step.  The ticket needs agent-scope release /
    // acquire fences, and on this 8-XCD part (one non-coherent L2 per XCD) those are a write-back and an invalidate of the
    // XCD's whole L2 -- executed by every block of every reduction.  A kernel boundary does the same once.
    static int en = -1; if (en < 0) { const char* e = getenv("M1_RED_LASTBLOCK"); en = e ? atoi(e) : 0; }
    return en != 0;
}
__device__ __forceinline__ float m1_ld_dev(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int NS>
__device__ __forceinline__ void m1_red_finish(const M1RedFin<NS>& fin, const float* partial, int N, int C, int nchunks) {
    __shared__ int s_last;
    __shared__ double s_red[M1_RED_THREADS];
    __threadfence();                                           // this block's partials are visible device-wide ...
    __syncthreads();                                           // ... all of them, before its ticket
    const int n = fin.joint ? 0 : (int)blockIdx.y;
    const unsigned total = fin.joint ? gridDim.x * gridDim.y : gridDim.x;
    if (threadIdx.x == 0) { const unsigned t = atomicAdd(fin.cnt + n, 1u); s_last = t == total - 1; if (s_last) fin.cnt[n] = 0; }
    __syncthreads();
    if (!s_last) return;
    __threadfence();                                           // (the partial rows were never read by this CU before: no stale L1 lines)
    const int rows = (int)total, P = C * NS;
    const size_t row_stride = (size_t)P;
    const float* const p0 = partial + (size_t)n * rows * row_stride;
    // all 256 threads: pair (c,k) x slice of the rows; a slice is summed in row order (fp64), the slices are combined in slice
    // order -- the same additions whichever block ends up doing them
    int SL = 1; while (SL * 2 * P <= M1_RED_THREADS && SL * 2 <= rows) SL *= 2;
    const int per = (rows + SL - 1) / SL;
    for (int base = 0; base < P; base += M1_RED_THREADS / SL) {
        const int npair = P - base < M1_RED_THREADS / SL ? P - base : M1_RED_THREADS / SL;
        const int t = threadIdx.x, pl = t % npair, sl = t / npair;
        double s = 0.0;
        if (sl < SL) {
            const int j0 = sl * per, j1 = j0 + per < rows ? j0 + per : rows;
            const float* q = p0 + base + pl;
#pragma unroll 4
            for (int j = j0; j < j1; ++j) s += (double)q[(size_t)j * row_stride];
        }
        s_red[t] = s;
        __syncthreads();
        if (sl == 0) {
            for (int q = 1; q < SL; ++q) s += s_red[q * npair + pl];
            s_red[pl] = s;                                     // (slot pl was this thread's own)
        }
        __syncthreads();
        if (t < npair) {
            const int idx = base + t, c = idx / NS, k = idx - c * NS;
            float* o = fin.out + ((size_t)n * C + c) * NS;
            if (fin.csplit > 0) o = c < fin.csplit ? fin.out + ((size_t)n * fin.csplit + c) * NS : fin.out2 + ((size_t)n * (C - fin.csplit) + (c - fin.csplit)) * NS;
            if (fin.mode == 1 && fin.stats_V > 0 && NS == 2) {   // (NS == 2: both sums of channel c sit in this round, P is even)
                if (k == 0) {
                    const double mean = s_red[t] / (double)fin.stats_V;
                    double var = s_red[t + 1] / (double)fin.stats_V - mean * mean;
                    if (var < 0.0) var = 0.0;
                    o[0] = (float)mean; o[1] = (float)(1.0 / sqrt(var + (double)fin.eps));
                }
            } else o[k] = (fin.mode == 1 && fin.accumulate ? o[k] : 0.f) + (float)s_red[t];
        }
        __syncthreads();
    }
    if (fin.mode != 2) return;
    // parameter sums over the samples: the last sample to be finalised
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) { const unsigned t = atomicAdd(fin.cnt + N, 1u); s_last = t == (unsigned)N - 1; if (s_last) fin.cnt[N] = 0; }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    for (int idx = threadIdx.x; idx < P; idx += M1_RED_THREADS) {
        const int c = idx / NS, k = idx - c * NS;
        double tot = 0.0;
        for (int m = 0; m < N; ++m) tot += (double)m1_ld_dev(fin.out + ((size_t)m * C + c) * NS + k);
        float* pp = nullptr; int pa = 0;                       // (static indices: a run-time index would move the argument struct to scratch)
#pragma unroll
        for (int kk = 0; kk < NS; ++kk) if (kk == k) { pp = fin.po.ptr[kk]; pa = fin.po.acc[kk]; }
        if (pp) pp[c] = (pa ? pp[c] : 0.f) + (float)tot;
    }
}

// Functor contract:  __device__ void operator()(int n, long long v, int c, float* acc) const;   acc[NS] += ...
template <int NS, typename F>
__global__ void __launch_bounds__(M1_RED_THREADS) m1_reduce_nc_kernel(F f, long long V, int C, int chunkV,
                                                                      int nchunks, float* __restrict__ partial, M1RedFin<NS> fin) {
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
        __syncthreads();
    }
    if (fin.mode) m1_red_finish<NS>(fin, partial, (int)gridDim.y, C, nchunks);
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
    for (int n = 0; n < N; ++n) {
        double s[NS];
#pragma unroll
        for (int k = 0; k < NS; ++k) s[k] = 0.0;
        for (int j = lane; j < nchunks; j += 64)
#pragma unroll
            for (int k = 0; k < NS; ++k) s[k] += (double)partial[(((size_t)n * nchunks + j) * C + c) * NS + k];
#pragma unroll
        for (int k = 0; k < NS; ++k) { s[k] = wave_sum_d(s[k]); tot[k] += (double)(float)s[k]; }
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < NS; ++k) out[((size_t)n * C + c) * NS + k] = (float)s[k];
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
                                                                          int nchunks, float* __restrict__ partial, M1RedFin<NS> fin) {
    __shared__ float red[M1_RED_THREADS * VEC];                      // one sum at a time (NS passes)
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
                    partial[(((size_t)n * nchunks + chunk) * C + gi * VEC + e) * NS + k] = s;
                }
            }
            __syncthreads();
        }
    }
    if (fin.mode) m1_red_finish<NS>(fin, partial, (int)gridDim.y, C, nchunks);
}

template <typename F, typename = void> struct M1RedVec { static constexpr int value = 0; };
template <typename F> struct M1RedVec<F, decltype((void)F::kVec)> { static constexpr int value = F::kVec; };

// fin (optional): the fold the caller would launch next, done by the last block instead.  Returns with fin->mode = 0 when the
// switch is off or no ticket slot is available: the caller then launches its finalize kernel as before.
template <int NS, typename F>
static inline int m1_reduce_nc_launch(const F& f, int N, long long V, int C, float* partial, hipStream_t st, M1RedFin<NS>* finp = nullptr) {
    const int chunkV = m1_red_chunkV(V, C, N), nchunks = m1_red_nchunks(V, C, N);
    dim3 grid(nchunks, N);
    M1RedFin<NS> fin{};
    if (finp && finp->mode && m1_red_lastblock_on()) {
        finp->cnt = m1_red_ticket_slot(N + 1);
        if (finp->cnt) fin = *finp; else finp->mode = 0;
    } else if (finp) finp->mode = 0;
    constexpr int VEC = M1RedVec<F>::value;
    if constexpr (VEC > 0) {
        if (C % VEC == 0) {
            hipLaunchKernelGGL((m1_reduce_nc_vec_kernel<NS, VEC, F>), grid, dim3(M1_RED_THREADS), 0, st, f, V, C, chunkV, nchunks,
                               partial, fin);
            return m1_check_launch();
        }
    }
    hipLaunchKernelGGL((m1_reduce_nc_kernel<NS, F>), grid, dim3(M1_RED_THREADS), 0, st, f, V, C, chunkV, nchunks,
                       partial, fin);
    return m1_check_launch();
}
