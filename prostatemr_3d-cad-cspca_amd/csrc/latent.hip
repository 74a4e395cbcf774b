// latent.hip -- per-scale latent head of the hierarchical probabilistic U-Net: reparameterised sample
// (networks.py:640-647,664-671,688-695,712-719) and KL(q||p) of diagonal Gaussians (networks.py:373-385;
// tfp MultivariateNormalDiag / kl_divergence), forward and backward (App. F).
#include "common.h"

#define LOGSIG_CLIP 0.1f
__device__ __forceinline__ float clipls(float v) { return fminf(fmaxf(v, -LOGSIG_CLIP), LOGSIG_CLIP); }
__device__ __forceinline__ float clipmask(float v) { return (v >= -LOGSIG_CLIP && v <= LOGSIG_CLIP) ? 1.f : 0.f; }

// The N(0,1) draw of element i of a latent head when the caller injects none (m1_latent_sample_rng_*): a pure function of
// (seed, step, stream id, element index) -- Philox4x32-10 (the dropout stream's generator, common.h) + Box-Muller -- so the backward
// pass regenerates the draw the forward made, nothing is stored, and no generator launch sits in the captured step
// (MultivariateNormalDiag(...).sample(), networks.py:647: mu + sigma * eps).
struct LatRng { const uint64_t* rng; uint64_t stream_id; };
__device__ __forceinline__ float latent_eps(const LatRng& q, long long i) {
    const uint64_t seed = q.rng[0] + q.stream_id * 0x9E3779B97F4A7C15ull, base = q.rng[1] << 36;
    const uint4 r = philox4x32_10(seed, base + (uint64_t)i);
    const float u1 = ((float)(r.x >> 8) + 0.5f) * (1.0f / 16777216.0f), u2 = (float)(r.y >> 8) * (1.0f / 16777216.0f);
    return sqrtf(-2.f * logf(u1)) * cosf(6.28318530718f * u2);
}

template <typename T>
__global__ void latent_sample_fwd_kernel(const T* __restrict__ ml, const T* __restrict__ eps, T* __restrict__ z,
                                         long long NV, int L, int mode, LatRng q) {
    const long long tot = NV * L;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (long long)gridDim.x * blockDim.x) {
        const long long v = i / L; const int d = (int)(i % L);
        const float mu = Act<T>::ld(ml + v * 2 * L + d);
        float r = mu;
        // mode 2: two passes of the reference stacked along the batch axis -- the first half of the batch is the sampling pass
        // (networks.py:348; eps holds N/2 samples), the second half the prob_mean pass (:349)
        if (mode == 0 || (mode == 2 && i < tot / 2))
            r = fmaf(expf(clipls(Act<T>::ld(ml + v * 2 * L + L + d))), q.rng ? latent_eps(q, i) : Act<T>::ld(eps + i), mu);
        Act<T>::st(z + i, r);
    }
}

template <typename T>
__global__ void latent_sample_bwd_kernel(const T* __restrict__ ml, const T* __restrict__ eps, const T* __restrict__ dz,
                                         T* __restrict__ dml, long long NV, int L, int mode, LatRng q) {
    const long long tot = NV * L;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (long long)gridDim.x * blockDim.x) {
        const long long v = i / L; const int d = (int)(i % L);
        const float g = Act<T>::ld(dz + i);
        Act<T>::st(dml + v * 2 * L + d, g);
        float gl = 0.f;
        if (mode == 0 || (mode == 2 && i < tot / 2)) {
            const float ls = Act<T>::ld(ml + v * 2 * L + L + d);
            gl = g * expf(clipls(ls)) * (q.rng ? latent_eps(q, i) : Act<T>::ld(eps + i)) * clipmask(ls);
        }
        Act<T>::st(dml + v * 2 * L + L + d, gl);
    }
}

// single block: deterministic sum; kl[0] = (1/N) * sum_{n,v} KL_voxel
template <typename T>
__global__ void __launch_bounds__(1024) kl_fwd_kernel(const T* __restrict__ mq, const T* __restrict__ mp, float* __restrict__ kl,
                                                      long long NV, int L, int N) {
    __shared__ double red[16];
    double s = 0.0;
    for (long long v = threadIdx.x; v < NV; v += blockDim.x) {
        float t = 0.f;
        for (int d = 0; d < L; ++d) {
            const float lq = clipls(Act<T>::ld(mq + v * 2 * L + L + d)), lp = clipls(Act<T>::ld(mp + v * 2 * L + L + d));
            const float sp_inv = expf(-lp), rs = expf(lq) * sp_inv;
            const float dm = (Act<T>::ld(mq + v * 2 * L + d) - Act<T>::ld(mp + v * 2 * L + d)) * sp_inv;
            t += rs * rs + dm * dm - 1.f + 2.f * (lp - lq);
        }
        s += 0.5 * (double)t;
    }
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) tot += red[i];
        kl[0] = (float)(tot / (double)N);
    }
}

// NVall >= NV: the gradient tensors hold NVall voxels (a batch of which only the first N samples enter this KL term, M1Net's stacked
// passes): the rest is written as zeros here instead of by a fill + copy of autograd's slice backward
template <typename T>
__global__ void kl_bwd_kernel(const T* __restrict__ mq, const T* __restrict__ mp, const float* __restrict__ dkl,
                              T* __restrict__ dmq, T* __restrict__ dmp, long long NV, int L, int N, long long NVall) {
    const long long tot = NV * L, totall = NVall * L;
    const float sc = dkl[0] / (float)N;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < totall; i += (long long)gridDim.x * blockDim.x) {
        const long long v = i / L; const int d = (int)(i % L);
        if (i >= tot) {
            Act<T>::st(dmq + v * 2 * L + d, 0.f); Act<T>::st(dmp + v * 2 * L + d, 0.f);
            Act<T>::st(dmq + v * 2 * L + L + d, 0.f); Act<T>::st(dmp + v * 2 * L + L + d, 0.f);
            continue;
        }
        const float lqr = Act<T>::ld(mq + v * 2 * L + L + d), lpr = Act<T>::ld(mp + v * 2 * L + L + d);
        const float lq = clipls(lqr), lp = clipls(lpr);
        const float ip2 = expf(-2.f * lp);                 // 1/sigma_p^2
        const float rs2 = expf(2.f * (lq - lp));           // (sigma_q/sigma_p)^2
        const float dm = Act<T>::ld(mq + v * 2 * L + d) - Act<T>::ld(mp + v * 2 * L + d);
        Act<T>::st(dmq + v * 2 * L + d, sc * dm * ip2);
        Act<T>::st(dmp + v * 2 * L + d, -sc * dm * ip2);
        Act<T>::st(dmq + v * 2 * L + L + d, sc * (rs2 - 1.f) * clipmask(lqr));
        Act<T>::st(dmp + v * 2 * L + L + d, sc * (1.f - rs2 - dm * dm * ip2) * clipmask(lpr));
    }
}

static inline int gxl(long long n) { long long b = cdiv_ll(n, 256); return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b)); }

static int latent_fwd(const void* ml, const void* eps, LatRng q, void* z, int N, long long V, int L, int mode, int dtype, hipStream_t st) {
    if (!ml || !z || (mode != 1 && !eps && !q.rng) || N <= 0 || V <= 0 || L <= 0 || mode < 0 || mode > 2 || (mode == 2 && (N & 1))) return M1_ERR_BAD_ARG;
    const long long NV = (long long)N * V;
    if (dtype == M1_BF16) hipLaunchKernelGGL(latent_sample_fwd_kernel<bf16_t>, dim3(gxl(NV * L)), dim3(256), 0, st, (const bf16_t*)ml, (const bf16_t*)eps, (bf16_t*)z, NV, L, mode, q);
    else hipLaunchKernelGGL(latent_sample_fwd_kernel<float>, dim3(gxl(NV * L)), dim3(256), 0, st, (const float*)ml, (const float*)eps, (float*)z, NV, L, mode, q);
    return m1_check_launch();
}
static int latent_bwd(const void* ml, const void* eps, LatRng q, const void* dz, void* dml, int N, long long V, int L, int mode, int dtype, hipStream_t st) {
    if (!ml || !dz || !dml || (mode != 1 && !eps && !q.rng) || N <= 0 || V <= 0 || L <= 0 || mode < 0 || mode > 2 || (mode == 2 && (N & 1))) return M1_ERR_BAD_ARG;
    const long long NV = (long long)N * V;
    if (dtype == M1_BF16) hipLaunchKernelGGL(latent_sample_bwd_kernel<bf16_t>, dim3(gxl(NV * L)), dim3(256), 0, st, (const bf16_t*)ml, (const bf16_t*)eps, (const bf16_t*)dz, (bf16_t*)dml, NV, L, mode, q);
    else hipLaunchKernelGGL(latent_sample_bwd_kernel<float>, dim3(gxl(NV * L)), dim3(256), 0, st, (const float*)ml, (const float*)eps, (const float*)dz, (float*)dml, NV, L, mode, q);
    return m1_check_launch();
}
extern "C" int m1_latent_sample_fwd(const void* ml, const void* eps, void* z, int N, long long V, int L, int mode, int dtype,
                                    void* stream) {
    return latent_fwd(ml, eps, LatRng{nullptr, 0}, z, N, V, L, mode, dtype, (hipStream_t)stream);
}
extern "C" int m1_latent_sample_bwd(const void* ml, const void* eps, const void* dz, void* dml, int N, long long V, int L,
                                    int mode, int dtype, void* stream) {
    return latent_bwd(ml, eps, LatRng{nullptr, 0}, dz, dml, N, V, L, mode, dtype, (hipStream_t)stream);
}
// the same with the draws made in the kernel from the dropout / sampling stream state rng = {seed, step} (device) and a stream id
extern "C" int m1_latent_sample_rng_fwd(const void* ml, const uint64_t* rng, uint64_t stream_id, void* z, int N, long long V, int L, int mode,
                                        int dtype, void* stream) {
    if (!rng) return M1_ERR_BAD_ARG;
    return latent_fwd(ml, nullptr, LatRng{rng, stream_id}, z, N, V, L, mode, dtype, (hipStream_t)stream);
}
extern "C" int m1_latent_sample_rng_bwd(const void* ml, const uint64_t* rng, uint64_t stream_id, const void* dz, void* dml, int N, long long V,
                                        int L, int mode, int dtype, void* stream) {
    if (!rng) return M1_ERR_BAD_ARG;
    return latent_bwd(ml, nullptr, LatRng{rng, stream_id}, dz, dml, N, V, L, mode, dtype, (hipStream_t)stream);
}
extern "C" int m1_kl_fwd(const void* ml_q, const void* ml_p, float* kl, int N, long long V, int L, int dtype, void* stream) {
    if (!ml_q || !ml_p || !kl || N <= 0 || V <= 0 || L <= 0) return M1_ERR_BAD_ARG;
    const long long NV = (long long)N * V; hipStream_t st = (hipStream_t)stream;
    if (dtype == M1_BF16) hipLaunchKernelGGL(kl_fwd_kernel<bf16_t>, dim3(1), dim3(1024), 0, st, (const bf16_t*)ml_q, (const bf16_t*)ml_p, kl, NV, L, N);
    else hipLaunchKernelGGL(kl_fwd_kernel<float>, dim3(1), dim3(1024), 0, st, (const float*)ml_q, (const float*)ml_p, kl, NV, L, N);
    return m1_check_launch();
}
static int kl_bwd_entry(const void* ml_q, const void* ml_p, const float* dkl, void* dml_q, void* dml_p, int N, long long V,
                        int L, int Nall, int dtype, void* stream) {
    if (!ml_q || !ml_p || !dkl || !dml_q || !dml_p || N <= 0 || V <= 0 || L <= 0 || Nall < N) return M1_ERR_BAD_ARG;
    const long long NV = (long long)N * V, NVall = (long long)Nall * V; hipStream_t st = (hipStream_t)stream;
    if (dtype == M1_BF16) hipLaunchKernelGGL(kl_bwd_kernel<bf16_t>, dim3(gxl(NVall * L)), dim3(256), 0, st, (const bf16_t*)ml_q, (const bf16_t*)ml_p, dkl, (bf16_t*)dml_q, (bf16_t*)dml_p, NV, L, N, NVall);
    else hipLaunchKernelGGL(kl_bwd_kernel<float>, dim3(gxl(NVall * L)), dim3(256), 0, st, (const float*)ml_q, (const float*)ml_p, dkl, (float*)dml_q, (float*)dml_p, NV, L, N, NVall);
    return m1_check_launch();
}
extern "C" int m1_kl_bwd(const void* ml_q, const void* ml_p, const float* dkl, void* dml_q, void* dml_p, int N, long long V,
                         int L, int dtype, void* stream) {
    return kl_bwd_entry(ml_q, ml_p, dkl, dml_q, dml_p, N, V, L, N, dtype, stream);
}
// the KL term reads the first N of Nall samples of ml_q / ml_p (the sampling half of a stacked pass): dml_q / dml_p hold Nall samples,
// the gradient of the rest is written as zeros
extern "C" int m1_kl_bwd_first(const void* ml_q, const void* ml_p, const float* dkl, void* dml_q, void* dml_p, int N, long long V,
                               int L, int Nall, int dtype, void* stream) {
    return kl_bwd_entry(ml_q, ml_p, dkl, dml_q, dml_p, N, V, L, Nall, dtype, stream);
}
