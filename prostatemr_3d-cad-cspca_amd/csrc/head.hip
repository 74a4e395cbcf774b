// head.hip -- output heads: per-voxel softmax over the class axis (networks.py:754, 388-390) of the main
// logits and the deep-supervision logits (networks.py:739-741,751).  The reference upsamples the feature
// map (UpSampling3D) and then applies the 1x1x1 logits conv; a 1x1x1 conv commutes with nearest repeat, so
// the conv runs at native resolution and the repeat is index arithmetic here (bit-identical per voxel).
// Also: standalone dropout (network_blocks.py:142-143, networks.py:462) and fp32<->bf16 casts.
#include "common.h"

#define MAX_HEADS 4
#define MAX_NC 8

struct HeadsP {
    const void* logits[MAX_HEADS];
    void* dlogits[MAX_HEADS];
    int u0[MAX_HEADS], u1[MAX_HEADS], u2[MAX_HEADS];
    int nheads, N, D, H, W, nc;
};

template <typename T>
__global__ void softmax_heads_fwd_kernel(HeadsP p, float* __restrict__ probs) {
    const long long V = (long long)p.D * p.H * p.W, tot = V * p.N * p.nheads;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (long long)gridDim.x * blockDim.x) {
        const int h = (int)(i % p.nheads); const long long gv = i / p.nheads;
        const int n = (int)(gv / V); long long r = gv % V;
        const int w = (int)(r % p.W); r /= p.W; const int hh = (int)(r % p.H); const int d = (int)(r / p.H);
        const int Dh = p.D / p.u0[h], Hh = p.H / p.u1[h], Wh = p.W / p.u2[h];
        const long long src = (((long long)n * Dh + d / p.u0[h]) * Hh + hh / p.u1[h]) * Wh + w / p.u2[h];
        const T* lp = (const T*)p.logits[h] + src * p.nc;
        float v[MAX_NC], m = -3.4e38f;
        for (int c = 0; c < p.nc; ++c) { v[c] = Act<T>::ld(lp + c); m = fmaxf(m, v[c]); }
        float s = 0.f;
        for (int c = 0; c < p.nc; ++c) { v[c] = expf(v[c] - m); s += v[c]; }
        const float inv = 1.f / s;
        float* o = probs + gv * (p.nheads * p.nc) + h * p.nc;
        for (int c = 0; c < p.nc; ++c) o[c] = v[c] * inv;
    }
}

// dlogits_h[n,q,c] = sum_{v in window(q)} p_c * (dp_c - sum_k p_k dp_k)
template <typename T>
__global__ void softmax_heads_bwd_kernel(HeadsP p, const float* __restrict__ probs, const float* __restrict__ dprobs, int h) {
    const int Dh = p.D / p.u0[h], Hh = p.H / p.u1[h], Wh = p.W / p.u2[h];
    const long long Vh = (long long)Dh * Hh * Wh, tot = Vh * p.N;
    const int stride = p.nheads * p.nc;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(i / Vh); long long r = i % Vh;
        const int qw = (int)(r % Wh); r /= Wh; const int qh = (int)(r % Hh); const int qd = (int)(r / Hh);
        float g[MAX_NC];
        for (int c = 0; c < p.nc; ++c) g[c] = 0.f;
        for (int a = 0; a < p.u0[h]; ++a)
            for (int b = 0; b < p.u1[h]; ++b)
                for (int e = 0; e < p.u2[h]; ++e) {
                    const long long v = (((long long)n * p.D + qd * p.u0[h] + a) * p.H + qh * p.u1[h] + b) * p.W + qw * p.u2[h] + e;
                    const float* pr = probs + v * stride + h * p.nc;
                    const float* dp = dprobs + v * stride + h * p.nc;
                    float dot = 0.f;
                    for (int c = 0; c < p.nc; ++c) dot += pr[c] * dp[c];
                    for (int c = 0; c < p.nc; ++c) g[c] += pr[c] * (dp[c] - dot);
                }
        T* o = (T*)p.dlogits[h] + i * p.nc;
        for (int c = 0; c < p.nc; ++c) Act<T>::st(o + c, g[c]);
    }
}

static int fill_heads(HeadsP& p, const m1_head_t* heads, int nheads, int N, int D, int H, int W, int nc, bool need_d) {
    if (!heads || nheads < 1 || nheads > MAX_HEADS || nc < 1 || nc > MAX_NC || N <= 0 || D <= 0 || H <= 0 || W <= 0) return M1_ERR_BAD_ARG;
    p.nheads = nheads; p.N = N; p.D = D; p.H = H; p.W = W; p.nc = nc;
    for (int i = 0; i < MAX_HEADS; ++i) {
        if (i < nheads) {
            if (!heads[i].logits || (need_d && !heads[i].dlogits)) return M1_ERR_BAD_ARG;
            if (heads[i].u0 <= 0 || heads[i].u1 <= 0 || heads[i].u2 <= 0 || D % heads[i].u0 || H % heads[i].u1 || W % heads[i].u2) return M1_ERR_UNSUPPORTED;
            p.logits[i] = heads[i].logits; p.dlogits[i] = heads[i].dlogits; p.u0[i] = heads[i].u0; p.u1[i] = heads[i].u1; p.u2[i] = heads[i].u2;
        } else { p.logits[i] = nullptr; p.dlogits[i] = nullptr; p.u0[i] = p.u1[i] = p.u2[i] = 1; }
    }
    return M1_OK;
}
static inline int gxh(long long n) { long long b = cdiv_ll(n, 256); return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b)); }

extern "C" int m1_softmax_heads_fwd(const m1_head_t* heads, int nheads, float* probs, int N, int D, int H, int W, int nc,
                                    int dtype, void* stream) {
    HeadsP p; int rc = fill_heads(p, heads, nheads, N, D, H, W, nc, false); if (rc) return rc;
    if (!probs) return M1_ERR_BAD_ARG;
    const long long tot = (long long)N * D * H * W * nheads; hipStream_t st = (hipStream_t)stream;
    if (dtype == M1_BF16) hipLaunchKernelGGL(softmax_heads_fwd_kernel<bf16_t>, dim3(gxh(tot)), dim3(256), 0, st, p, probs);
    else hipLaunchKernelGGL(softmax_heads_fwd_kernel<float>, dim3(gxh(tot)), dim3(256), 0, st, p, probs);
    return m1_check_launch();
}

extern "C" int m1_softmax_heads_bwd(const m1_head_t* heads, int nheads, const float* probs, const float* dprobs, int N, int D,
                                    int H, int W, int nc, int dtype, void* stream) {
    HeadsP p; int rc = fill_heads(p, heads, nheads, N, D, H, W, nc, true); if (rc) return rc;
    if (!probs || !dprobs) return M1_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    for (int h = 0; h < nheads; ++h) {
        const long long tot = (long long)N * (D / p.u0[h]) * (H / p.u1[h]) * (W / p.u2[h]);
        if (dtype == M1_BF16) hipLaunchKernelGGL(softmax_heads_bwd_kernel<bf16_t>, dim3(gxh(tot)), dim3(256), 0, st, p, probs, dprobs, h);
        else hipLaunchKernelGGL(softmax_heads_bwd_kernel<float>, dim3(gxh(tot)), dim3(256), 0, st, p, probs, dprobs, h);
    }
    return m1_check_launch();
}

// ---------------- dropout (forward and backward are the same map) ----------------
template <typename T>
__global__ void dropout_kernel(const T* __restrict__ x, T* __restrict__ y, long long n, float rate, const uint64_t* __restrict__ rng,
                               uint64_t layer_id) {
    const uint64_t seed = rng[0] + layer_id * 0x9E3779B97F4A7C15ull, base = rng[1] << 36;
    const float sc = 1.f / (1.f - rate);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        Act<T>::st(y + i, philox_keep(seed, base, (uint64_t)i, rate) ? Act<T>::ld(x + i) * sc : 0.f);
}
// ---------------------------------------------------------------------------------------------------------
// Focal loss on the softmax heads (losses.py:32-49): per head h of nc classes at each voxel
//   s = sum_c p_c ; q_c = clip(p_c / s, eps, 1-eps) ; fl = sum_c alpha_c * y_c^2 * (1-q_c)^gamma * (-log q_c)
//   loss = mean_h mean_n sum_voxels fl
// forward: per-(block, head) partial sums, folded in a fixed order by one block (run-to-run identical);
// backward: dL/dp_k = go/(N*heads) * ( g_k/s - sum_c g_c p_c / s^2 ), g_c = alpha_c y_c^2 f'(q_c) inside the clip range,
//   f'(q) = gamma (1-q)^(gamma-1) log q - (1-q)^gamma / q.
// ---------------------------------------------------------------------------------------------------------
#define FOCAL_EPS 1e-7f
struct FocalP {
    const float* probs; const void* y; float alpha[MAX_NC];
    long long NV, V; int nheads, nc, N, ydt; float gamma;
};
template <typename TY> __device__ __forceinline__ float focal_ld_y(const void* y, long long i);
template <> __device__ __forceinline__ float focal_ld_y<float>(const void* y, long long i) { return ((const float*)y)[i]; }
template <> __device__ __forceinline__ float focal_ld_y<unsigned short>(const void* y, long long i) {
    return __uint_as_float((unsigned)((const unsigned short*)y)[i] << 16);
}
__device__ __forceinline__ float focal_pow(float b, float e) { return e == 2.f ? b * b : (e == 1.f ? b : (e == 0.f ? 1.f : powf(b, e))); }

template <typename TY>
__global__ void __launch_bounds__(256) focal_fwd_kernel(FocalP p, float* __restrict__ partial) {
    float acc[MAX_HEADS];
    for (int h = 0; h < MAX_HEADS; ++h) acc[h] = 0.f;
    const int n = blockIdx.y;
    for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < p.V; v += (long long)gridDim.x * 256) {
        const long long gv = (long long)n * p.V + v;
        float y[MAX_NC];
        for (int c = 0; c < p.nc; ++c) y[c] = focal_ld_y<TY>(p.y, gv * p.nc + c);
        const float* pp = p.probs + gv * (p.nheads * p.nc);
        for (int h = 0; h < p.nheads; ++h) {
            float s = 0.f;
            for (int c = 0; c < p.nc; ++c) s += pp[h * p.nc + c];
            float fl = 0.f;
            for (int c = 0; c < p.nc; ++c) {
                const float q = fminf(fmaxf(pp[h * p.nc + c] / s, FOCAL_EPS), 1.f - FOCAL_EPS);
                fl += p.alpha[c] * ((y[c] * focal_pow(1.f - q, p.gamma)) * (y[c] * -logf(q)));
            }
            acc[h] += fl;
        }
    }
    __shared__ float red[4][MAX_HEADS];
    for (int h = 0; h < p.nheads; ++h) {
        float a = acc[h];
        for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][h] = a;
    }
    __syncthreads();
    if (threadIdx.x < p.nheads)
        partial[((long long)n * p.nheads + threadIdx.x) * gridDim.x + blockIdx.x] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
// one block: loss = (1/(N*heads)) sum_{n,h} sum_blocks partial  (fixed order)
__global__ void __launch_bounds__(256) focal_finish_kernel(const float* __restrict__ partial, int rows, int per, float scale,
                                                            float* __restrict__ loss) {
    __shared__ float red[256];
    float a = 0.f;
    for (int r = 0; r < rows; ++r)
        for (int i = threadIdx.x; i < per; i += 256) a += partial[(long long)r * per + i];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) loss[0] = red[0] * scale;
}
template <typename TY>
__global__ void __launch_bounds__(256) focal_bwd_kernel(FocalP p, const float* __restrict__ go, float* __restrict__ dprobs) {
    const float sc = go[0] / (float)((long long)p.N * p.nheads);
    for (long long gv = (long long)blockIdx.x * 256 + threadIdx.x; gv < p.NV; gv += (long long)gridDim.x * 256) {
        float y[MAX_NC];
        for (int c = 0; c < p.nc; ++c) y[c] = focal_ld_y<TY>(p.y, gv * p.nc + c);
        const float* pp = p.probs + gv * (p.nheads * p.nc);
        float* dp = dprobs + gv * (p.nheads * p.nc);
        for (int h = 0; h < p.nheads; ++h) {
            float s = 0.f, pc[MAX_NC], g[MAX_NC], gp = 0.f;
            for (int c = 0; c < p.nc; ++c) { pc[c] = pp[h * p.nc + c]; s += pc[c]; }
            const float inv = 1.f / s;
            for (int c = 0; c < p.nc; ++c) {
                const float r = pc[c] / s;
                float gc = 0.f;
                if (r >= FOCAL_EPS && r <= 1.f - FOCAL_EPS && y[c] != 0.f) {
                    const float om = 1.f - r;
                    const float d = p.gamma == 0.f ? 0.f : p.gamma * focal_pow(om, p.gamma - 1.f) * logf(r);
                    gc = p.alpha[c] * y[c] * y[c] * (d - focal_pow(om, p.gamma) / r);
                }
                g[c] = gc; gp += gc * pc[c];
            }
            for (int c = 0; c < p.nc; ++c) dp[h * p.nc + c] = sc * (g[c] - gp * inv) * inv;
        }
    }
}
static int focal_fill(FocalP& p, const float* probs, const void* y, int y_dtype, const float* alpha, float gamma, int N,
                      long long V, int nheads, int nc) {
    if (!probs || !y || !alpha || N <= 0 || V <= 0) return M1_ERR_BAD_ARG;
    if (nheads <= 0 || nheads > MAX_HEADS || nc <= 0 || nc > MAX_NC || (y_dtype != M1_F32 && y_dtype != M1_BF16)) return M1_ERR_UNSUPPORTED;
    p.probs = probs; p.y = y; p.V = V; p.NV = (long long)N * V; p.nheads = nheads; p.nc = nc; p.N = N; p.ydt = y_dtype; p.gamma = gamma;
    for (int c = 0; c < MAX_NC; ++c) p.alpha[c] = c < nc ? alpha[c] : 0.f;
    return M1_OK;
}
extern "C" size_t m1_focal_ws_floats(int N, long long V, int nheads) {
    if (N <= 0 || V <= 0 || nheads <= 0) return 0;
    long long blocks = (V + 1023) / 1024; if (blocks > 256) blocks = 256; if (blocks < 1) blocks = 1;
    return (size_t)N * nheads * blocks;
}
extern "C" int m1_focal_fwd(const float* probs, const void* y_true, int y_dtype, const float* alpha, float gamma, int N,
                            long long V, int nheads, int nc, float* ws, float* loss, void* stream) {
    FocalP p; int rc = focal_fill(p, probs, y_true, y_dtype, alpha, gamma, N, V, nheads, nc);
    if (rc != M1_OK) return rc;
    if (!ws || !loss || N > 65535) return M1_ERR_BAD_ARG;
    const int blocks = (int)(m1_focal_ws_floats(N, V, nheads) / ((size_t)N * nheads));
    if (y_dtype == M1_F32) hipLaunchKernelGGL(focal_fwd_kernel<float>, dim3(blocks, N), dim3(256), 0, (hipStream_t)stream, p, ws);
    else hipLaunchKernelGGL(focal_fwd_kernel<unsigned short>, dim3(blocks, N), dim3(256), 0, (hipStream_t)stream, p, ws);
    hipLaunchKernelGGL(focal_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, ws, N * nheads, blocks,
                       1.f / (float)((long long)N * nheads), loss);
    return m1_check_launch();
}
extern "C" int m1_focal_bwd(const float* probs, const void* y_true, int y_dtype, const float* alpha, float gamma, int N,
                            long long V, int nheads, int nc, const float* dloss, float* dprobs, void* stream) {
    FocalP p; int rc = focal_fill(p, probs, y_true, y_dtype, alpha, gamma, N, V, nheads, nc);
    if (rc != M1_OK) return rc;
    if (!dloss || !dprobs) return M1_ERR_BAD_ARG;
    long long blocks = (p.NV + 255) / 256; if (blocks > 4096) blocks = 4096;
    if (y_dtype == M1_F32) hipLaunchKernelGGL(focal_bwd_kernel<float>, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, p, dloss, dprobs);
    else hipLaunchKernelGGL(focal_bwd_kernel<unsigned short>, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, p, dloss, dprobs);
    return m1_check_launch();
}

extern "C" int m1_dropout(const void* x, void* y, long long n, float rate, const uint64_t* rng, uint64_t layer_id, int dtype,
                          void* stream) {
    if (!x || !y || n <= 0 || rate <= 0.f || rate >= 1.f || !rng) return M1_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == M1_BF16) hipLaunchKernelGGL(dropout_kernel<bf16_t>, dim3(gxh(n)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, n, rate, rng, layer_id);
    else hipLaunchKernelGGL(dropout_kernel<float>, dim3(gxh(n)), dim3(256), 0, st, (const float*)x, (float*)y, n, rate, rng, layer_id);
    return m1_check_launch();
}

// ---------------- casts ----------------
template <typename S, typename D>
__global__ void cast_kernel(const S* __restrict__ x, D* __restrict__ y, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        Act<D>::st(y + i, Act<S>::ld(x + i));
}
extern "C" int m1_cast(const void* x, int sdt, void* y, int ddt, long long n, void* stream) {
    if (!x || !y || n <= 0) return M1_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (sdt == M1_F32 && ddt == M1_BF16) hipLaunchKernelGGL((cast_kernel<float, bf16_t>), dim3(gxh(n)), dim3(256), 0, st, (const float*)x, (bf16_t*)y, n);
    else if (sdt == M1_BF16 && ddt == M1_F32) hipLaunchKernelGGL((cast_kernel<bf16_t, float>), dim3(gxh(n)), dim3(256), 0, st, (const bf16_t*)x, (float*)y, n);
    else if (sdt == M1_F32 && ddt == M1_F32) hipLaunchKernelGGL((cast_kernel<float, float>), dim3(gxh(n)), dim3(256), 0, st, (const float*)x, (float*)y, n);
    else if (sdt == M1_BF16 && ddt == M1_BF16) hipLaunchKernelGGL((cast_kernel<bf16_t, bf16_t>), dim3(gxh(n)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, n);
    else return M1_ERR_UNSUPPORTED;
    return m1_check_launch();
}
