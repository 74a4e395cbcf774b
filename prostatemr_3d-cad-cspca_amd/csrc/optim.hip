// optim.hip -- Keras Adam(amsgrad=True) (train_model.py:120) fused with the L2
// regulariser gradient 2*lambda*w of networks.py:456-460 (kernel AND bias; App. C-7), on flat fp32 buffers.
//   lr_t = lr*sqrt(1-b2^t)/(1-b1^t);  m,v EMA;  vhat = max(vhat, v);  w -= lr_t*m/(sqrt(vhat)+eps)
// lr and the step counter live in device memory so that a captured hipGraph can be replayed.
#include "common.h"

__global__ void __launch_bounds__(256) adam_amsgrad_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                           float* __restrict__ v, float* __restrict__ vhat, long long n,
                                                           long long n_kernel, long long n_bias, float l2k, float l2b,
                                                           float gscale, const float* __restrict__ lr_dev, float b1, float b2,
                                                           float eps, const int* __restrict__ step_dev) {
    const int t = step_dev[0];
    const float lr_t = lr_dev[0] * sqrtf(1.f - powf(b2, (float)t)) / (1.f - powf(b1, (float)t));
    const long long n4 = n >> 2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4 + (n & 3); i += (long long)gridDim.x * blockDim.x) {
        if (i < n4) {
            const long long e = i << 2;
            float4 pv = *reinterpret_cast<float4*>(p + e), gv = *reinterpret_cast<const float4*>(g + e);
            float4 mv = *reinterpret_cast<float4*>(m + e), vv = *reinterpret_cast<float4*>(v + e), hv = *reinterpret_cast<float4*>(vhat + e);
            float pp[4] = {pv.x, pv.y, pv.z, pv.w}, gg[4] = {gv.x, gv.y, gv.z, gv.w}, mm[4] = {mv.x, mv.y, mv.z, mv.w};
            float vq[4] = {vv.x, vv.y, vv.z, vv.w}, hh[4] = {hv.x, hv.y, hv.z, hv.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const long long idx = e + k;
                const float lam = idx < n_kernel ? l2k : (idx < n_kernel + n_bias ? l2b : 0.f);
                const float gr = gg[k] * gscale + 2.f * lam * pp[k];
                mm[k] = b1 * mm[k] + (1.f - b1) * gr;
                vq[k] = b2 * vq[k] + (1.f - b2) * gr * gr;
                hh[k] = fmaxf(hh[k], vq[k]);
                pp[k] -= lr_t * mm[k] / (sqrtf(hh[k]) + eps);
            }
            *reinterpret_cast<float4*>(p + e) = make_float4(pp[0], pp[1], pp[2], pp[3]);
            *reinterpret_cast<float4*>(m + e) = make_float4(mm[0], mm[1], mm[2], mm[3]);
            *reinterpret_cast<float4*>(v + e) = make_float4(vq[0], vq[1], vq[2], vq[3]);
            *reinterpret_cast<float4*>(vhat + e) = make_float4(hh[0], hh[1], hh[2], hh[3]);
        } else {
            const long long idx = (n4 << 2) + (i - n4);
            const float lam = idx < n_kernel ? l2k : (idx < n_kernel + n_bias ? l2b : 0.f);
            const float gr = g[idx] * gscale + 2.f * lam * p[idx];
            const float mm = b1 * m[idx] + (1.f - b1) * gr, vq = b2 * v[idx] + (1.f - b2) * gr * gr;
            const float hh = fmaxf(vhat[idx], vq);
            m[idx] = mm; v[idx] = vq; vhat[idx] = hh;
            p[idx] -= lr_t * mm / (sqrtf(hh) + eps);
        }
    }
}

__global__ void step_inc_kernel(int* step, unsigned long long* rng) { if (step) step[0] += 1; if (rng) rng[1] += 1; }

extern "C" int m1_adam_amsgrad(float* p, const float* g, float* m, float* v, float* vhat, long long n, long long n_kernel,
                               long long n_bias, float l2_kernel, float l2_bias, float grad_scale, const float* lr_dev,
                               float beta1, float beta2, float eps, const int* step_dev, void* stream) {
    if (m1_debug_skip("adam")) return M1_OK;
    if (!p || !g || !m || !v || !vhat || !lr_dev || !step_dev || n <= 0) return M1_ERR_BAD_ARG;
    if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v | (uintptr_t)vhat) & 15) return M1_ERR_BAD_ARG;
    long long blocks = cdiv_ll((n >> 2) + 3, 256); if (blocks > 4096) blocks = 4096;
    M1ProfScope ps("adam_amsgrad", 0.0, 36.0 * n, (hipStream_t)stream);
    hipLaunchKernelGGL(adam_amsgrad_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, vhat, n, n_kernel,
                       n_bias, l2_kernel, l2_bias, grad_scale, lr_dev, beta1, beta2, eps, step_dev);
    return m1_check_launch();
}

// advances the optimizer step counter and the dropout/sampling stream counter (rng[1]) by one
extern "C" int m1_step_advance(int* step_dev, uint64_t* rng_dev, void* stream) {
    hipLaunchKernelGGL(step_inc_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step_dev, (unsigned long long*)rng_dev);
    return m1_check_launch();
}
