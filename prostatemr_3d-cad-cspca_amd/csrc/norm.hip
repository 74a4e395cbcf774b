// norm.hip -- InstanceNormalization (eps inside rsqrt, biased variance) [+ LeakyReLU], forward and backward.
// Replaces tfa.layers.InstanceNormalization + relu(alpha=0.1) at network_blocks.py:38-44,54-60,104,128 and
// networks.py:473,575-576.  Backward math: SURVEY.md App. F.
#include "common.h"
#include "reduce.h"
#include "gather.h"

extern "C" size_t m1_reduce_ws_floats(int N, long long V, int C, int nsums) {
    return (size_t)N * m1_red_nchunks(V, C, N) * C * nsums + (size_t)N * C * nsums + 64;
}

// ------------------------------------------------------------------------------------------------
// statistics: mean and rstd per (n,c)
// ------------------------------------------------------------------------------------------------
template <typename T>
struct StatsF {
    const T* x; long long V; int C;
    __device__ void operator()(int n, long long v, int c, float* acc) const {
        float xv = Act<T>::ld(x + ((size_t)n * V + v) * C + c);
        acc[0] += xv; acc[1] += xv * xv;
    }
    static constexpr int kVec = sizeof(T) == 2 ? 8 : 4;
    __device__ void vec(int n, long long v, int c0, float (*acc)[kVec]) const {
        float xv[kVec];
        VecIO<T, kVec>::ld(x + ((size_t)n * V + v) * C + c0, xv);
#pragma unroll
        for (int e = 0; e < kVec; ++e) { acc[0][e] += xv[e]; acc[1][e] += xv[e] * xv[e]; }
    }
};

template <typename T>
static int stats_impl(const void* x, int N, long long V, int C, float eps, float* stats, float* ws, hipStream_t st) {
    StatsF<T> f{(const T*)x, V, C};
    int rc = m1_reduce_nc_launch<2>(f, N, V, C, ws, st);
    if (rc) return rc;
    return m1_reduce_finalize_launch<2>(ws, N, C, m1_red_nchunks(V, C, N), stats, V, eps, st);
}

int m1_stats_internal(const void* x, int N, long long V, int C, int dtype, float eps, float* stats, float* ws, hipStream_t st) {
    return dtype == M1_BF16 ? stats_impl<bf16_t>(x, N, V, C, eps, stats, ws, st) : stats_impl<float>(x, N, V, C, eps, stats, ws, st);
}
// Partial rows per sample a statistics workspace holds: one per 64 voxels, and on small volumes up to 256 (one per 16 voxels) so that
// the pointwise kernel -- one row per (sample, wave) -- is not held to 8 / 60 waves per column slice on the (5,10,10) / (10,20,20)
// levels by the capacity of its own statistics rows (round 6: 64 -> 256 at (4,10,20,20) ran 8 tiles per wave in a row, 27-32 us).
long long m1_stats_rows_cap(long long V) {
    const long long a = (V + 63) / 64; long long b = (V + 15) / 16; if (b > 256) b = 256;
    if (!M1_CFG("M1_STATS_ROWS_SMALL", 1)) b = 0;          // (0: one row per 64 voxels everywhere, round 5)
    return a > b ? a : b;
}
// floats needed by either the fused-epilogue partials (one per 64-row tile at worst) or the stand-alone reduction
size_t m1_stats_ws_floats(int N, long long V, int C) {
    const size_t fused = (size_t)N * (size_t)m1_stats_rows_cap(V) * C * 2;
    const size_t alone = m1_reduce_ws_floats(N, V, C, 2);
    return (fused > alone ? fused : alone) + 64;
}

extern "C" int m1_instnorm_stats(const void* x, int N, long long V, int C, int dtype, float eps, float* stats,
                                 float* ws, void* stream) {
    if (!x || !stats || !ws || N <= 0 || V <= 0 || C <= 0) return M1_ERR_BAD_ARG;
    M1ProfScope ps("instnorm_stats", 0.0, (double)N * V * C * (dtype == M1_BF16 ? 2 : 4), (hipStream_t)stream);
    return dtype == M1_BF16 ? stats_impl<bf16_t>(x, N, V, C, eps, stats, ws, (hipStream_t)stream)
                            : stats_impl<float>(x, N, V, C, eps, stats, ws, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------
// apply: y = lrelu( (x-mean)*rstd*gamma + beta )
// ------------------------------------------------------------------------------------------------
template <typename T, int VEC>
__global__ void __launch_bounds__(256) in_apply_kernel(const T* __restrict__ x, const float* __restrict__ stats,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float slope,
                                                       T* __restrict__ y, long long V, int C) {
    const int n = blockIdx.y;
    const long long per = V * (C / VEC);
    const T* xn = x + (size_t)n * V * C;
    T* yn = y + (size_t)n * V * C;
    const int cg = C / VEC;
    // the launch keeps gridDim.x*blockDim.x a multiple of the channel groups: a thread's channels never change, their
    // parameters are loaded once (4*VEC scalar loads per 16-byte vector otherwise -- the kernel was L1/issue bound)
    const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long long)gridDim.x * blockDim.x;
    const int c0 = (int)(i0 % cg) * VEC;
    float mean[VEC], rstd[VEC], gm[VEC], bt[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        const int c = c0 + k;
        mean[k] = stats[((size_t)n * C + c) * 2]; rstd[k] = stats[((size_t)n * C + c) * 2 + 1]; gm[k] = gamma[c]; bt[k] = beta[c];
    }
    for (long long i = i0; i < per; i += stride) {
        float v[VEC];
        VecIO<T, VEC>::ld(xn + i * VEC, v);
#pragma unroll
        for (int k = 0; k < VEC; ++k) v[k] = lrelu_f((v[k] - mean[k]) * rstd[k] * gm[k] + bt[k], slope);
        VecIO<T, VEC>::st(yn + i * VEC, v);
    }
}

template <typename T>
static int apply_impl(const void* x, const float* stats, const float* gamma, const float* beta, float slope, void* y,
                      int N, long long V, int C, hipStream_t st) {
    constexpr int VW = sizeof(T) == 2 ? 8 : 4;
    long long per;
    if (C % VW == 0) {
        per = V * (C / VW);
        int gx = m1_grid_for(per, C / VW);
        hipLaunchKernelGGL((in_apply_kernel<T, VW>), dim3(gx, N), dim3(256), 0, st, (const T*)x, stats, gamma, beta,
                           slope, (T*)y, V, C);
    } else {
        per = V * C;
        int gx = m1_grid_for(per, C);
        hipLaunchKernelGGL((in_apply_kernel<T, 1>), dim3(gx, N), dim3(256), 0, st, (const T*)x, stats, gamma, beta,
                           slope, (T*)y, V, C);
    }
    return m1_check_launch();
}

extern "C" int m1_instnorm_apply(const void* x, const float* stats, const float* gamma, const float* beta,
                                 float slope, void* y, int N, long long V, int C, int dtype, void* stream) {
    if (m1_debug_skip("in_apply")) return M1_OK;
    if (!x || !stats || !gamma || !beta || !y || N <= 0 || V <= 0 || C <= 0) return M1_ERR_BAD_ARG;
    M1ProfScope ps("instnorm_apply", 0.0, 2.0 * N * V * C * (dtype == M1_BF16 ? 2 : 4), (hipStream_t)stream);
    return dtype == M1_BF16 ? apply_impl<bf16_t>(x, stats, gamma, beta, slope, y, N, V, C, (hipStream_t)stream)
                            : apply_impl<float>(x, stats, gamma, beta, slope, y, N, V, C, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------
// backward (App. F):  a = lrelu(y), y = gamma*xh + beta, xh = (x-mean)*rstd
//   dy = da * lrelu'(y);  dbeta = sum dy;  dgamma = sum dy*xh;  dx = gamma*rstd*(dy - dbeta/V - xh*dgamma/V)
// ------------------------------------------------------------------------------------------------
template <typename T>
struct InBwdF {
    const T* x; const T* da; const float* stats; const float* gamma; const float* beta; float slope;
    long long V; int C;
    __device__ void operator()(int n, long long v, int c, float* acc) const {
        const size_t idx = ((size_t)n * V + v) * C + c;
        const float mean = stats[((size_t)n * C + c) * 2], rstd = stats[((size_t)n * C + c) * 2 + 1];
        const float xh = (Act<T>::ld(x + idx) - mean) * rstd;
        const float yv = gamma[c] * xh + beta[c];
        const float dy = Act<T>::ld(da + idx) * lrelu_g(yv, slope);
        acc[0] += dy; acc[1] += dy * xh;
    }
    static constexpr int kVec = sizeof(T) == 2 ? 8 : 4;
    __device__ void vec(int n, long long v, int c0, float (*acc)[kVec]) const {
        const size_t idx = ((size_t)n * V + v) * C + c0;
        float xv[kVec], dv[kVec];
        VecIO<T, kVec>::ld(x + idx, xv); VecIO<T, kVec>::ld(da + idx, dv);
#pragma unroll
        for (int e = 0; e < kVec; ++e) {
            const int c = c0 + e;
            const float mean = stats[((size_t)n * C + c) * 2], rstd = stats[((size_t)n * C + c) * 2 + 1];
            const float xh = (xv[e] - mean) * rstd;
            const float dy = dv[e] * lrelu_g(gamma[c] * xh + beta[c], slope);
            acc[0][e] += dy; acc[1][e] += dy * xh;
        }
    }
};

template <typename T, int VEC>
__global__ void __launch_bounds__(256) in_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ da,
                                                           const float* __restrict__ stats,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float slope,
                                                           const float* __restrict__ sums /*[N][C][2]*/,
                                                           T* __restrict__ dx, long long V, int C) {
    const int n = blockIdx.y;
    const int cg = C / VEC;
    const long long per = V * cg;
    const size_t base = (size_t)n * V * C;
    const float invV = 1.0f / (float)V;
    const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long long)gridDim.x * blockDim.x;
    const int c0 = (int)(i0 % cg) * VEC;                 // invariant per thread (see in_apply_kernel)
    float mean[VEC], rstd[VEC], gm[VEC], bt[VEC], s0[VEC], s1[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        const int c = c0 + k;
        mean[k] = stats[((size_t)n * C + c) * 2]; rstd[k] = stats[((size_t)n * C + c) * 2 + 1]; gm[k] = gamma[c]; bt[k] = beta[c];
        s0[k] = sums[((size_t)n * C + c) * 2] * invV; s1[k] = sums[((size_t)n * C + c) * 2 + 1] * invV;
    }
    for (long long i = i0; i < per; i += stride) {
        float xv[VEC], dv[VEC];
        VecIO<T, VEC>::ld(x + base + i * VEC, xv);
        VecIO<T, VEC>::ld(da + base + i * VEC, dv);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const float xh = (xv[k] - mean[k]) * rstd[k];
            const float dy = dv[k] * lrelu_g(gm[k] * xh + bt[k], slope);
            dv[k] = gm[k] * rstd[k] * (dy - s0[k] - xh * s1[k]);
        }
        VecIO<T, VEC>::st(dx + base + i * VEC, dv);
    }
}

// the apply half of the backward, from partial sums a data-gradient epilogue already emitted (m1_conv3d_dgrad_inbwd):
// partial [N][nparts][C][2], sums scratch [N][C][2]
template <typename T>
static int bwd_from_partials(const void* x, const float* stats, const float* gamma, const float* beta, float slope, const void* dy, void* dx,
                             float* dgamma, float* dbeta, int N, long long V, int C, const float* partial, int nparts, float* sums,
                             hipStream_t st, int accumulate) {
    M1ParamOut<2> po{{dbeta, dgamma}, {accumulate, accumulate}};
    int rc = m1_reduce_finalize_params_launch<2>(partial, N, C, nparts, sums, po, st);
    if (rc) return rc;
    constexpr int VW = sizeof(T) == 2 ? 8 : 4;
    if (C % VW == 0) {
        long long per = V * (C / VW);
        int gx = m1_grid_for(per, C / VW);
        hipLaunchKernelGGL((in_bwd_apply_kernel<T, VW>), dim3(gx, N), dim3(256), 0, st, (const T*)x, (const T*)dy, stats, gamma, beta, slope, sums, (T*)dx, V, C);
    } else {
        long long per = V * C;
        int gx = m1_grid_for(per, C);
        hipLaunchKernelGGL((in_bwd_apply_kernel<T, 1>), dim3(gx, N), dim3(256), 0, st, (const T*)x, (const T*)dy, stats, gamma, beta, slope, sums, (T*)dx, V, C);
    }
    return m1_check_launch();
}
extern "C" int m1_instnorm_bwd_partials(const void* x, const float* stats, const float* gamma, const float* beta, float slope, const void* dy,
                                        void* dx, float* dgamma, float* dbeta, int N, long long V, int C, int dtype, const float* partial,
                                        int nparts, float* sums, int accumulate, void* stream) {
    if (!x || !stats || !gamma || !beta || !dy || !dx || !dgamma || !dbeta || !partial || !sums || nparts <= 0 || N <= 0 || V <= 0 || C <= 0)
        return M1_ERR_BAD_ARG;
    M1ProfScope ps("instnorm_bwd", 0.0, 3.0 * N * V * C * (dtype == M1_BF16 ? 2 : 4), (hipStream_t)stream);
    return dtype == M1_BF16 ? bwd_from_partials<bf16_t>(x, stats, gamma, beta, slope, dy, dx, dgamma, dbeta, N, V, C, partial, nparts, sums, (hipStream_t)stream, accumulate)
                            : bwd_from_partials<float>(x, stats, gamma, beta, slope, dy, dx, dgamma, dbeta, N, V, C, partial, nparts, sums, (hipStream_t)stream, accumulate);
}

template <typename T>
static int bwd_impl(const void* x, const float* stats, const float* gamma, const float* beta, float slope,
                    const void* dy, void* dx, float* dgamma, float* dbeta, int N, long long V, int C, float* ws,
                    hipStream_t st, int accumulate) {
    InBwdF<T> f{(const T*)x, (const T*)dy, stats, gamma, beta, slope, V, C};
    const int nchunks = m1_red_nchunks(V, C, N);
    float* sums = ws + (size_t)N * nchunks * C * 2;
    M1ParamOut<2> po{{dbeta, dgamma}, {accumulate, accumulate}};      // dbeta = sum_n sums[.][0], dgamma = sum_n sums[.][1]
    int rc = m1_reduce_nc_launch<2>(f, N, V, C, ws, st);
    if (rc) return rc;
    rc = m1_reduce_finalize_params_launch<2>(ws, N, C, nchunks, sums, po, st); if (rc) return rc;
    constexpr int VW = sizeof(T) == 2 ? 8 : 4;
    if (C % VW == 0) {
        long long per = V * (C / VW);
        int gx = m1_grid_for(per, C / VW);
        hipLaunchKernelGGL((in_bwd_apply_kernel<T, VW>), dim3(gx, N), dim3(256), 0, st, (const T*)x, (const T*)dy,
                           stats, gamma, beta, slope, sums, (T*)dx, V, C);
    } else {
        long long per = V * C;
        int gx = m1_grid_for(per, C);
        hipLaunchKernelGGL((in_bwd_apply_kernel<T, 1>), dim3(gx, N), dim3(256), 0, st, (const T*)x, (const T*)dy,
                           stats, gamma, beta, slope, sums, (T*)dx, V, C);
    }
    return m1_check_launch();
}

extern "C" int m1_instnorm_bwd(const void* x, const float* stats, const float* gamma, const float* beta, float slope,
                               const void* dy, void* dx, float* dgamma, float* dbeta, int N, long long V, int C,
                               int dtype, float* ws, int accumulate, void* stream) {
    if (m1_debug_skip("in_bwd")) return M1_OK;
    if (!x || !stats || !gamma || !beta || !dy || !dx || !dgamma || !dbeta || !ws) return M1_ERR_BAD_ARG;
    M1ProfScope ps("instnorm_bwd", 0.0, 5.0 * N * V * C * (dtype == M1_BF16 ? 2 : 4), (hipStream_t)stream);
    return dtype == M1_BF16
               ? bwd_impl<bf16_t>(x, stats, gamma, beta, slope, dy, dx, dgamma, dbeta, N, V, C, ws, (hipStream_t)stream, accumulate)
               : bwd_impl<float>(x, stats, gamma, beta, slope, dy, dx, dgamma, dbeta, N, V, C, ws, (hipStream_t)stream, accumulate);
}

// ------------------------------------------------------------------------------------------------
// column sum over all (n, v): out[c] = sum x[n,v,c]  (bias gradients). ws as for nsums=1.
// ------------------------------------------------------------------------------------------------
template <typename T>
struct ColSumF {
    const T* x; long long V; int C;
    __device__ void operator()(int n, long long v, int c, float* acc) const {
        acc[0] += Act<T>::ld(x + ((size_t)n * V + v) * C + c);
    }
    static constexpr int kVec = sizeof(T) == 2 ? 8 : 4;
    __device__ void vec(int n, long long v, int c0, float (*acc)[kVec]) const {
        float xv[kVec];
        VecIO<T, kVec>::ld(x + ((size_t)n * V + v) * C + c0, xv);
#pragma unroll
        for (int e = 0; e < kVec; ++e) acc[0][e] += xv[e];
    }
};
// Internal (used by conv wgrad): ws must hold N*nchunks*C floats.
int m1_colsum_internal(const void* x, int N, long long V, int C, int dtype, float* out, float* ws, hipStream_t st, int accumulate) {
    int rc;
    if (dtype == M1_BF16) { ColSumF<bf16_t> f{(const bf16_t*)x, V, C}; rc = m1_reduce_nc_launch<1>(f, N, V, C, ws, st); }
    else { ColSumF<float> f{(const float*)x, V, C}; rc = m1_reduce_nc_launch<1>(f, N, V, C, ws, st); }
    if (rc) return rc;
    // partial is [N*nchunks][C][1]: fold all rows as one sample
    return m1_reduce_finalize_launch<1>(ws, 1, C, N * m1_red_nchunks(V, C, N), out, 0, 0.f, st, accumulate);
}

// ---- deferred bias gradients of the transposed convolutions (round 6) ------------------------------------------------------------
// db[c] (+)= sum_{n,v} dy[n,v,c] is a reduction + a fold of its own per Conv3DTranspose (m1_convT3d_wgrad cannot fuse it: d(out) is the
// shifted operand there): 19 + 19 launches of 4 - 12 us on the data-gradient chain of a C3 step, read by nobody before the optimiser.
// Under m1_wgrad_defer they are queued like the folds of the weight-gradient copies and run as TWO batched launches from
// m1_wgrad_fold_pending (the caller keeps dy and the workspace alive until then, as it does for the copies).
#include <mutex>
#include <vector>
#define CS_MAX 24
struct ColSumJob { const void* x; long long V; int C, N, chunkV, nchunks; float* partial; float* out; int acc; };
struct ColSumBatch { int n; int pref[CS_MAX + 1]; ColSumJob j[CS_MAX]; };
static_assert(sizeof(ColSumBatch) <= 3800, "batch must fit the kernel argument segment");
template <typename T>
__global__ void __launch_bounds__(M1_RED_THREADS) colsum_batch_kernel(ColSumBatch B) {
    constexpr int VEC = ColSumF<T>::kVec;
    __shared__ __attribute__((aligned(16))) float red[1][M1_RED_THREADS * VEC];
    int k = 0;
    while (k + 1 < B.n && (int)blockIdx.x >= B.pref[k + 1]) ++k;
    const ColSumJob& q = B.j[k];
    const int b = blockIdx.x - B.pref[k], n = b / q.nchunks, chunk = b - n * q.nchunks;
    ColSumF<T> f{(const T*)q.x, q.V, q.C};
    m1_reduce_nc_vec_body<1, VEC, ColSumF<T>>(f, q.V, q.C, q.chunkV, q.nchunks, q.partial, n, chunk, red);
}
// one block per (job, channel): the N * nchunks partial rows in fp64, as m1_reduce_finalize_kernel<1> folds them
__global__ void __launch_bounds__(256) colsum_finalize_batch_kernel(ColSumBatch B) {
    __shared__ double red[4];
    int k = 0;
    while (k + 1 < B.n && (int)blockIdx.x >= B.pref[k + 1]) ++k;
    const ColSumJob& q = B.j[k];
    const int c = blockIdx.x - B.pref[k], rows = q.N * q.nchunks, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double s = 0.0;
    for (int j = threadIdx.x; j < rows; j += 256) s += (double)q.partial[(size_t)j * q.C + c];
    s = wave_sum_d(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) q.out[c] = (q.acc ? q.out[c] : 0.f) + (float)((red[0] + red[1]) + (red[2] + red[3]));
}
static std::mutex g_cs_mu;
static std::vector<ColSumJob> g_cs_pending[2];          // [bf16, fp32]
// true: queued (nothing launched); false: the caller runs the reduction now (shape outside the vector kernel, or a second gradient
// into the same db in this batch -- two read-modify-writes of db must not share a launch)
bool m1_colsum_defer(const void* x, int N, long long V, int C, int dtype, float* out, float* ws, int accumulate) {
    if (!M1_CFG("M1_BIAS_DEFER", 1)) return false;
    const int VEC = dtype == M1_BF16 ? 8 : 4;
    if (C % VEC || (long long)N * m1_red_nchunks(V, C, N) >= (1 << 20)) return false;
    std::lock_guard<std::mutex> lk(g_cs_mu);
    std::vector<ColSumJob>& v = g_cs_pending[dtype == M1_BF16 ? 0 : 1];
    for (const ColSumJob& q : g_cs_pending[0]) if (q.out == out) return false;
    for (const ColSumJob& q : g_cs_pending[1]) if (q.out == out) return false;
    v.push_back(ColSumJob{x, V, C, N, m1_red_chunkV(V, C, N), m1_red_nchunks(V, C, N), ws, out, accumulate});
    return true;
}
int m1_colsum_launch_pending(hipStream_t st) {
    std::lock_guard<std::mutex> lk(g_cs_mu);
    int rc = M1_OK;
    for (int t = 0; t < 2 && rc == M1_OK; ++t) {
        std::vector<ColSumJob>& v = g_cs_pending[t];
        size_t i = 0;
        while (i < v.size() && rc == M1_OK) {
            ColSumBatch B{}, Fb{}; B.pref[0] = 0; Fb.pref[0] = 0;
            while (i < v.size() && B.n < CS_MAX) {
                const ColSumJob& q = v[i++];
                B.j[B.n] = q; B.pref[B.n + 1] = B.pref[B.n] + q.N * q.nchunks; ++B.n;
                Fb.j[Fb.n] = q; Fb.pref[Fb.n + 1] = Fb.pref[Fb.n] + q.C; ++Fb.n;
            }
            if (t == 0) hipLaunchKernelGGL(colsum_batch_kernel<bf16_t>, dim3((unsigned)B.pref[B.n]), dim3(M1_RED_THREADS), 0, st, B);
            else hipLaunchKernelGGL(colsum_batch_kernel<float>, dim3((unsigned)B.pref[B.n]), dim3(M1_RED_THREADS), 0, st, B);
            rc = m1_check_launch(); if (rc) break;
            hipLaunchKernelGGL(colsum_finalize_batch_kernel, dim3((unsigned)Fb.pref[Fb.n]), dim3(256), 0, st, Fb);
            rc = m1_check_launch();
        }
        v.clear();
    }
    g_cs_pending[0].clear(); g_cs_pending[1].clear();
    return rc;
}
void m1_colsum_drop_pending() { std::lock_guard<std::mutex> lk(g_cs_mu); g_cs_pending[0].clear(); g_cs_pending[1].clear(); }
