// attn.hip -- the non-GEMM pieces of GridAttentionBlock3D (network_blocks.py:113-124), fwd + bwd:
//   sigma  = sigmoid( psi . lrelu(theta + upsample(phi)) + b_psi )       (B:113-119)
//   y      = upsample(sigma) * x                                           (B:120-124)
// The nearest-neighbour UpSampling3D is index arithmetic; its backward is a window sum.
// The three 1x1x1 convolutions (theta, phi, W) go through the conv entry points.
#include "common.h"
#include "reduce.h"

struct Geo {
    int N, Dt, Ht, Wt, Dp, Hp, Wp, C, ud, uh, uw;
};

__device__ __forceinline__ long long phi_voxel(const Geo& g, long long v /* fine voxel within sample */) {
    const int w = (int)(v % g.Wt); long long r = v / g.Wt;
    const int h = (int)(r % g.Ht); const int d = (int)(r / g.Ht);
    return ((long long)(d / g.ud) * g.Hp + (h / g.uh)) * g.Wp + (w / g.uw);
}

// lanes-per-voxel: power of two <= 64 covering C/VEC
static inline int lanes_per_voxel(int C, int VEC) { int l = 1; while (l < 64 && l * VEC < C) l <<= 1; return l; }

template <typename T, int VEC>
__global__ void __launch_bounds__(256) gate_sigma_fwd_kernel(const T* __restrict__ theta, const T* __restrict__ phi,
                                                             const float* __restrict__ wpsi, const float* __restrict__ bpsi,
                                                             T* __restrict__ sigma, Geo g, int lpv) {
    const long long Vt = (long long)g.Dt * g.Ht * g.Wt, Vp = (long long)g.Dp * g.Hp * g.Wp;
    const long long total = Vt * g.N;
    const int vpb = 256 / lpv, sub = threadIdx.x % lpv, vloc = threadIdx.x / lpv;
    for (long long gv = (long long)blockIdx.x * vpb + vloc; gv < total + vloc; gv += (long long)gridDim.x * vpb) {
        const bool ok = gv < total;   // keep all lanes in the shuffle
        float s = 0.f;
        if (ok) {
            const int n = (int)(gv / Vt); const long long v = gv % Vt;
            const T* tp = theta + (size_t)gv * g.C;
            const T* pp = phi + ((size_t)n * Vp + phi_voxel(g, v)) * g.C;
            for (int c = sub * VEC; c < g.C; c += lpv * VEC) {
                float a[VEC], b[VEC];
                VecIO<T, VEC>::ld(tp + c, a); VecIO<T, VEC>::ld(pp + c, b);
#pragma unroll
                for (int k = 0; k < VEC; ++k) s = fmaf(lrelu_f(a[k] + b[k], 0.1f), wpsi[c + k], s);
            }
        }
        for (int o = lpv >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (ok && sub == 0) Act<T>::st(sigma + gv, sigmoid_f(s + bpsi[0]));
    }
}

template <typename T, int VEC>
__global__ void __launch_bounds__(256) gate_dtheta_kernel(const T* __restrict__ theta, const T* __restrict__ phi,
                                                          const float* __restrict__ wpsi, const T* __restrict__ sigma,
                                                          const T* __restrict__ dsigma, T* __restrict__ dtheta, Geo g) {
    const long long Vt = (long long)g.Dt * g.Ht * g.Wt, Vp = (long long)g.Dp * g.Hp * g.Wp;
    const int cg = g.C / VEC;
    const long long per = Vt * g.N * cg;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (long long)gridDim.x * blockDim.x) {
        const int c0 = (int)(i % cg) * VEC; const long long gv = i / cg;
        const int n = (int)(gv / Vt); const long long v = gv % Vt;
        const float sg = Act<T>::ld(sigma + gv);
        const float dpsi = Act<T>::ld(dsigma + gv) * sg * (1.f - sg);
        float a[VEC], b[VEC];
        VecIO<T, VEC>::ld(theta + (size_t)gv * g.C + c0, a);
        VecIO<T, VEC>::ld(phi + ((size_t)n * Vp + phi_voxel(g, v)) * g.C + c0, b);
#pragma unroll
        for (int k = 0; k < VEC; ++k) a[k] = dpsi * wpsi[c0 + k] * lrelu_g(a[k] + b[k], 0.1f);
        VecIO<T, VEC>::st(dtheta + (size_t)gv * g.C + c0, a);
    }
}

// dwpsi[c] = sum_{n,v} dpsi * f[c];  dbpsi = sum dpsi  (second accumulator; read back from channel 0)
template <typename T>
struct GateWF {
    const T* theta; const T* phi; const T* sigma; const T* dsigma; Geo g;
    __device__ void operator()(int n, long long v, int c, float* acc) const {
        const long long Vt = (long long)g.Dt * g.Ht * g.Wt, Vp = (long long)g.Dp * g.Hp * g.Wp;
        const long long gv = (long long)n * Vt + v;
        const float sg = Act<T>::ld(sigma + gv);
        const float dpsi = Act<T>::ld(dsigma + gv) * sg * (1.f - sg);
        const float f = lrelu_f(Act<T>::ld(theta + (size_t)gv * g.C + c) +
                                Act<T>::ld(phi + ((size_t)n * Vp + phi_voxel(g, v)) * g.C + c), 0.1f);
        acc[0] += dpsi * f; acc[1] += dpsi;
    }
    static constexpr int kVec = sizeof(T) == 2 ? 8 : 4;
    __device__ void vec(int n, long long v, int c0, float (*acc)[kVec]) const {
        const long long Vt = (long long)g.Dt * g.Ht * g.Wt, Vp = (long long)g.Dp * g.Hp * g.Wp;
        const long long gv = (long long)n * Vt + v;
        const float sg = Act<T>::ld(sigma + gv);
        const float dpsi = Act<T>::ld(dsigma + gv) * sg * (1.f - sg);
        float th[kVec], ph[kVec];
        VecIO<T, kVec>::ld(theta + (size_t)gv * g.C + c0, th);
        VecIO<T, kVec>::ld(phi + ((size_t)n * Vp + phi_voxel(g, v)) * g.C + c0, ph);
#pragma unroll
        for (int e = 0; e < kVec; ++e) { acc[0][e] += dpsi * lrelu_f(th[e] + ph[e], 0.1f); acc[1][e] += dpsi; }
    }
};
__global__ void gate_w_finalize_kernel(const float* __restrict__ sums /*[N][C][2]*/, int N, int C,
                                       float* __restrict__ dwpsi, float* __restrict__ dbpsi, int acc) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0, b = 0.0;
    for (int n = 0; n < N; ++n) { s += sums[((size_t)n * C + c) * 2]; b += sums[((size_t)n * C + c) * 2 + 1]; }
    dwpsi[c] = (acc ? dwpsi[c] : 0.f) + (float)s;
    if (c == 0) dbpsi[0] = (acc ? dbpsi[0] : 0.f) + (float)b;
}

// dphi[n,p,c] = sum_{v in window(p)} dtheta[n,v,c].  One block per coarse voxel: lanes = channel groups x window
// lanes (the window is up to 4*16*16 = 1024 fine voxels at res0), folded through LDS.
template <typename T, int VEC>
__global__ void __launch_bounds__(256) window_sum_kernel(const T* __restrict__ dtheta, T* __restrict__ dphi, Geo g) {
    __shared__ float red[256 * VEC];
    const long long Vp = (long long)g.Dp * g.Hp * g.Wp;
    const int cg = g.C / VEC;
    int CGP = 1; while (CGP < cg && CGP < 256) CGP <<= 1;
    const int WL = 256 / CGP, cl = threadIdx.x % CGP, wl = threadIdx.x / CGP;
    const int win = g.ud * g.uh * g.uw;
    for (long long gp = blockIdx.x; gp < Vp * g.N; gp += gridDim.x) {
        const int n = (int)(gp / Vp); long long r = gp % Vp;
        const int pw = (int)(r % g.Wp); r /= g.Wp; const int ph = (int)(r % g.Hp); const int pd = (int)(r / g.Hp);
        for (int cb = 0; cb < cg; cb += CGP) {
            const int c0 = (cb + cl) * VEC;
            float s[VEC];
#pragma unroll
            for (int k = 0; k < VEC; ++k) s[k] = 0.f;
            if (cb + cl < cg)
                for (int wi = wl; wi < win; wi += WL) {
                    const int e = wi % g.uw, b = (wi / g.uw) % g.uh, a = wi / (g.uw * g.uh);
                    const long long v = (((long long)n * g.Dt + pd * g.ud + a) * g.Ht + ph * g.uh + b) * g.Wt + pw * g.uw + e;
                    float t[VEC];
                    VecIO<T, VEC>::ld(dtheta + (size_t)v * g.C + c0, t);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) s[k] += t[k];
                }
#pragma unroll
            for (int k = 0; k < VEC; ++k) red[threadIdx.x * VEC + k] = s[k];
            __syncthreads();
            if (wl == 0 && cb + cl < cg) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) s[k] = 0.f;
                for (int q = 0; q < WL; ++q)
#pragma unroll
                    for (int k = 0; k < VEC; ++k) s[k] += red[(q * CGP + cl) * VEC + k];
                VecIO<T, VEC>::st(dphi + (size_t)gp * g.C + c0, s);
            }
            __syncthreads();
        }
    }
}

// The whole non-GEMM backward of sigma = sigmoid(psi . lrelu(theta + phi_up) + b) in ONE pass over theta (round 6): a block owns coarse
// (phi) voxels; for every fine voxel of the window it forms d(theta) = d(psi) w_psi lrelu'(theta + phi), stores it, adds the STORED
// (rounded) value into the window sum d(phi) -- what window_sum_kernel read back -- and keeps running sums of d(w_psi) = sum d(psi) f
// and d(b_psi) = sum d(psi) for ONE partial row per block, folded in a fixed order by gate_w_fold_kernel.  Before: gate_dtheta +
// window_sum + the GateWF reduction (a second pass over theta, phi, sigma) + two finalize launches on the chain of every gate.
template <typename T, int VEC>
__global__ void __launch_bounds__(256) gate_bwd_fused_kernel(const T* __restrict__ theta, const T* __restrict__ phi, const float* __restrict__ wpsi,
                                                             const T* __restrict__ sigma, const T* __restrict__ dsigma, T* __restrict__ dtheta,
                                                             T* __restrict__ dphi, float* __restrict__ partial /*[gridDim.x][C][2]*/, Geo g) {
    __shared__ float red[256 * VEC];
    const long long Vp = (long long)g.Dp * g.Hp * g.Wp;
    const int cg = g.C / VEC;                                   // (host: cg <= 256)
    int CGP = 1; while (CGP < cg) CGP <<= 1;
    const int WL = 256 / CGP, cl = threadIdx.x % CGP, wl = threadIdx.x / CGP;
    const int win = g.ud * g.uh * g.uw;
    const bool act = cl < cg;
    const int c0 = cl * VEC;
    float wp[VEC], aw[VEC], ab = 0.f;
#pragma unroll
    for (int k = 0; k < VEC; ++k) { wp[k] = act ? wpsi[c0 + k] : 0.f; aw[k] = 0.f; }
    for (long long gp = blockIdx.x; gp < Vp * g.N; gp += gridDim.x) {
        const int n = (int)(gp / Vp); long long r = gp % Vp;
        const int pw = (int)(r % g.Wp); r /= g.Wp; const int ph = (int)(r % g.Hp); const int pd = (int)(r / g.Hp);
        float s[VEC], b[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) { s[k] = 0.f; b[k] = 0.f; }
        if (act) {
            VecIO<T, VEC>::ld(phi + (size_t)gp * g.C + c0, b);
            for (int wi = wl; wi < win; wi += WL) {
                const int e = wi % g.uw, bb = (wi / g.uw) % g.uh, a = wi / (g.uw * g.uh);
                const long long v = (((long long)n * g.Dt + pd * g.ud + a) * g.Ht + ph * g.uh + bb) * g.Wt + pw * g.uw + e;
                const float sg = Act<T>::ld(sigma + v);
                const float dps = Act<T>::ld(dsigma + v) * sg * (1.f - sg);
                float t[VEC];
                VecIO<T, VEC>::ld(theta + (size_t)v * g.C + c0, t);
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    const float f = t[k] + b[k];
                    aw[k] += dps * lrelu_f(f, 0.1f);
                    t[k] = dps * wp[k] * lrelu_g(f, 0.1f);
                }
                ab += dps;
                VecIO<T, VEC>::st(dtheta + (size_t)v * g.C + c0, t);
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    float rv = t[k];
                    if constexpr (sizeof(T) == 2) rv = bf2f(f2bf(rv));
                    s[k] += rv;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < VEC; ++k) red[threadIdx.x * VEC + k] = s[k];
        __syncthreads();
        if (wl == 0 && act) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) s[k] = 0.f;
            for (int q = 0; q < WL; ++q)
#pragma unroll
                for (int k = 0; k < VEC; ++k) s[k] += red[(q * CGP + cl) * VEC + k];
            VecIO<T, VEC>::st(dphi + (size_t)gp * g.C + c0, s);
        }
        __syncthreads();
    }
    // the block's partial row of {d(w_psi)[c], d(b_psi)}: window lanes folded in lane-row order
#pragma unroll
    for (int k = 0; k < VEC; ++k) red[threadIdx.x * VEC + k] = aw[k];
    __syncthreads();
    if (wl == 0 && act) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            float t = 0.f;
            for (int q = 0; q < WL; ++q) t += red[(q * CGP + cl) * VEC + k];
            partial[((size_t)blockIdx.x * g.C + c0 + k) * 2] = t;
        }
    }
    __syncthreads();
    red[threadIdx.x] = ab;
    __syncthreads();
    if (wl == 0 && act) {
        float t = 0.f;
        for (int q = 0; q < WL; ++q) t += red[q * CGP + cl];
#pragma unroll
        for (int k = 0; k < VEC; ++k) partial[((size_t)blockIdx.x * g.C + c0 + k) * 2 + 1] = t;      // (every channel lane of a voxel saw the same d(psi))
    }
}
// d(w_psi)[c] (+)= sum over the partial rows (fp64, fixed order), d(b_psi) from channel 0's second sum: one block per channel
__global__ void __launch_bounds__(256) gate_w_fold_kernel(const float* __restrict__ partial, int rows, int C, float* __restrict__ dwpsi,
                                                          float* __restrict__ dbpsi, int acc) {
    __shared__ double red[4][2];
    const int c = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double s = 0.0, b = 0.0;
    for (int j = threadIdx.x; j < rows; j += 256) { s += (double)partial[((size_t)j * C + c) * 2]; b += (double)partial[((size_t)j * C + c) * 2 + 1]; }
    s = wave_sum_d(s); b = wave_sum_d(b);
    if (lane == 0) { red[wave][0] = s; red[wave][1] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        dwpsi[c] = (acc ? dwpsi[c] : 0.f) + (float)((red[0][0] + red[1][0]) + (red[2][0] + red[3][0]));
        if (c == 0) dbpsi[0] = (acc ? dbpsi[0] : 0.f) + (float)((red[0][1] + red[1][1]) + (red[2][1] + red[3][1]));
    }
}

static inline int gx_for(long long per) { long long b = cdiv_ll(per, 256); return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b)); }

template <typename T>
static int gate_fwd_impl(const void* theta, const void* phi, const float* wpsi, const float* bpsi, void* sigma, const Geo& g,
                         hipStream_t st) {
    constexpr int VW = sizeof(T) == 2 ? 8 : 4;
    const long long total = (long long)g.N * g.Dt * g.Ht * g.Wt;
    if (g.C % VW == 0) {
        const int lpv = lanes_per_voxel(g.C, VW);
        hipLaunchKernelGGL((gate_sigma_fwd_kernel<T, VW>), dim3(gx_for(total * lpv)), dim3(256), 0, st, (const T*)theta,
                           (const T*)phi, wpsi, bpsi, (T*)sigma, g, lpv);
    } else {
        const int lpv = lanes_per_voxel(g.C, 1);
        hipLaunchKernelGGL((gate_sigma_fwd_kernel<T, 1>), dim3(gx_for(total * lpv)), dim3(256), 0, st, (const T*)theta,
                           (const T*)phi, wpsi, bpsi, (T*)sigma, g, lpv);
    }
    return m1_check_launch();
}

template <typename T>
static int gate_bwd_impl(const void* theta, const void* phi, const float* wpsi, const void* sigma, const void* dsigma,
                         void* dtheta, void* dphi, float* dwpsi, float* dbpsi, const Geo& g, float* ws, hipStream_t st, int acc) {
    constexpr int VW = sizeof(T) == 2 ? 8 : 4;
    const long long Vt = (long long)g.Dt * g.Ht * g.Wt, Vp = (long long)g.Dp * g.Hp * g.Wp;
    if (g.C % VW == 0 && g.C / VW <= 256 && M1_CFG("M1_GATE_BWD_FUSED", 1)) {       // one pass + one fold (see gate_bwd_fused_kernel)
        // partial rows = blocks: as many as the caller's reduction workspace holds (m1_reduce_ws_floats(N, Vt, C, 2)), at most 4096
        long long rows = (long long)(m1_reduce_ws_floats(g.N, Vt, g.C, 2) / ((size_t)g.C * 2));
        if (rows > 4096) rows = 4096;
        if (rows > Vp * g.N) rows = Vp * g.N;
        if (rows >= 1) {
            hipLaunchKernelGGL((gate_bwd_fused_kernel<T, VW>), dim3((unsigned)rows), dim3(256), 0, st, (const T*)theta, (const T*)phi, wpsi,
                               (const T*)sigma, (const T*)dsigma, (T*)dtheta, (T*)dphi, ws, g);
            int rc0 = m1_check_launch(); if (rc0) return rc0;
            hipLaunchKernelGGL(gate_w_fold_kernel, dim3(g.C), dim3(256), 0, st, ws, (int)rows, g.C, dwpsi, dbpsi, acc);
            return m1_check_launch();
        }
    }
    if (g.C % VW == 0) {
        hipLaunchKernelGGL((gate_dtheta_kernel<T, VW>), dim3(gx_for(Vt * g.N * (g.C / VW))), dim3(256), 0, st, (const T*)theta,
                           (const T*)phi, wpsi, (const T*)sigma, (const T*)dsigma, (T*)dtheta, g);
        hipLaunchKernelGGL((window_sum_kernel<T, VW>), dim3((unsigned)(Vp * g.N > 4096 ? 4096 : Vp * g.N)), dim3(256), 0, st, (const T*)dtheta,
                           (T*)dphi, g);
    } else {
        hipLaunchKernelGGL((gate_dtheta_kernel<T, 1>), dim3(gx_for(Vt * g.N * g.C)), dim3(256), 0, st, (const T*)theta,
                           (const T*)phi, wpsi, (const T*)sigma, (const T*)dsigma, (T*)dtheta, g);
        hipLaunchKernelGGL((window_sum_kernel<T, 1>), dim3((unsigned)(Vp * g.N > 4096 ? 4096 : Vp * g.N)), dim3(256), 0, st, (const T*)dtheta,
                           (T*)dphi, g);
    }
    int rc = m1_check_launch(); if (rc) return rc;
    GateWF<T> f{(const T*)theta, (const T*)phi, (const T*)sigma, (const T*)dsigma, g};
    rc = m1_reduce_nc_launch<2>(f, g.N, Vt, g.C, ws, st); if (rc) return rc;
    const int nchunks = m1_red_nchunks(Vt, g.C, g.N);
    float* sums = ws + (size_t)g.N * nchunks * g.C * 2;
    rc = m1_reduce_finalize_launch<2>(ws, g.N, g.C, nchunks, sums, 0, 0.f, st); if (rc) return rc;
    hipLaunchKernelGGL(gate_w_finalize_kernel, dim3((g.C + 255) / 256), dim3(256), 0, st, sums, g.N, g.C, dwpsi, dbpsi, acc);
    return m1_check_launch();
}

static int make_geo(Geo& g, int N, int Dt, int Ht, int Wt, int Dp, int Hp, int Wp, int C) {
    if (N <= 0 || Dt <= 0 || Ht <= 0 || Wt <= 0 || Dp <= 0 || Hp <= 0 || Wp <= 0 || C <= 0) return M1_ERR_BAD_ARG;
    if (Dt % Dp || Ht % Hp || Wt % Wp) return M1_ERR_UNSUPPORTED;   // UpSampling3D(size=shape//shape) then add must match
    g = Geo{N, Dt, Ht, Wt, Dp, Hp, Wp, C, Dt / Dp, Ht / Hp, Wt / Wp};
    return M1_OK;
}

extern "C" int m1_gate_sigma_fwd(const void* theta, const void* phi, const float* wpsi, const float* bpsi, void* sigma,
                                 int N, int Dt, int Ht, int Wt, int Dp, int Hp, int Wp, int C, int dtype, void* stream) {
    if (m1_debug_skip("gate")) return M1_OK;
    if (!theta || !phi || !wpsi || !bpsi || !sigma) return M1_ERR_BAD_ARG;
    Geo g; int rc = make_geo(g, N, Dt, Ht, Wt, Dp, Hp, Wp, C); if (rc) return rc;
    M1ProfScope ps("gate_sigma_fwd", 0.0, (double)N * Dt * Ht * Wt * C * (dtype == M1_BF16 ? 2 : 4), (hipStream_t)stream);
    return dtype == M1_BF16 ? gate_fwd_impl<bf16_t>(theta, phi, wpsi, bpsi, sigma, g, (hipStream_t)stream)
                            : gate_fwd_impl<float>(theta, phi, wpsi, bpsi, sigma, g, (hipStream_t)stream);
}

extern "C" int m1_gate_sigma_bwd(const void* theta, const void* phi, const float* wpsi, const void* sigma,
                                 const void* dsigma, void* dtheta, void* dphi, float* dwpsi, float* dbpsi, int N, int Dt,
                                 int Ht, int Wt, int Dp, int Hp, int Wp, int C, int dtype, float* ws, int accumulate,
                                 void* stream) {
    if (m1_debug_skip("gate")) return M1_OK;
    if (!theta || !phi || !wpsi || !sigma || !dsigma || !dtheta || !dphi || !dwpsi || !dbpsi || !ws) return M1_ERR_BAD_ARG;
    Geo g; int rc = make_geo(g, N, Dt, Ht, Wt, Dp, Hp, Wp, C); if (rc) return rc;
    M1ProfScope ps("gate_sigma_bwd", 0.0, 4.0 * N * Dt * Ht * Wt * C * (dtype == M1_BF16 ? 2 : 4), (hipStream_t)stream);
    return dtype == M1_BF16
               ? gate_bwd_impl<bf16_t>(theta, phi, wpsi, sigma, dsigma, dtheta, dphi, dwpsi, dbpsi, g, ws, (hipStream_t)stream, accumulate)
               : gate_bwd_impl<float>(theta, phi, wpsi, sigma, dsigma, dtheta, dphi, dwpsi, dbpsi, g, ws, (hipStream_t)stream, accumulate);
}

// ------------------------------------------------------------------------------------------------
// y = upsample(sigma) * x
// ------------------------------------------------------------------------------------------------
struct MulGeo { int N, D, H, W, C, s0, s1, s2, Ds, Hs, Ws; };

__device__ __forceinline__ long long sig_index(const MulGeo& g, long long gv) {
    const long long V = (long long)g.D * g.H * g.W;
    const int n = (int)(gv / V); long long r = gv % V;
    const int w = (int)(r % g.W); r /= g.W; const int h = (int)(r % g.H); const int d = (int)(r / g.H);
    return (((long long)n * g.Ds + d / g.s0) * g.Hs + h / g.s1) * g.Ws + w / g.s2;
}

template <typename T, int VEC, bool ACC = false>
__global__ void __launch_bounds__(256) mul_sigma_kernel(const T* __restrict__ x, const T* __restrict__ sigma,
                                                        T* __restrict__ y, MulGeo g) {
    const int cg = g.C / VEC;
    const long long per = (long long)g.N * g.D * g.H * g.W * cg;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (long long)gridDim.x * blockDim.x) {
        const long long gv = i / cg;
        const float s = Act<T>::ld(sigma + sig_index(g, gv));
        float a[VEC];
        VecIO<T, VEC>::ld(x + i * VEC, a);
#pragma unroll
        for (int k = 0; k < VEC; ++k) a[k] *= s;
        if (ACC) {                                        // y += (gradient of a tensor with several consumers)
            float b[VEC];
            VecIO<T, VEC>::ld(y + i * VEC, b);
#pragma unroll
            for (int k = 0; k < VEC; ++k) a[k] += b[k];
        }
        VecIO<T, VEC>::st(y + i * VEC, a);
    }
}

// sigma AND y = upsample(sigma) * x in one pass (round 6): lpv lanes per coarse voxel compute its sigma (all lanes end with the sum),
// store it, and walk the voxel's window of x with the ROUNDED value -- exactly what gate_sigma_fwd + mul_sigma produce, one launch and
// one read of sigma fewer on the forward chain of every gate.  Ci = gate (inter) channels, mg.C = channels of x.
template <typename T, int VEC>
__global__ void __launch_bounds__(256) gate_sigma_mul_fwd_kernel(const T* __restrict__ theta, const T* __restrict__ phi,
                                                                 const float* __restrict__ wpsi, const float* __restrict__ bpsi,
                                                                 T* __restrict__ sigma, const T* __restrict__ x, T* __restrict__ y,
                                                                 Geo g, MulGeo mg, int lpv) {
    const long long Vt = (long long)g.Dt * g.Ht * g.Wt, Vp = (long long)g.Dp * g.Hp * g.Wp;
    const long long total = Vt * g.N;
    const int vpb = 256 / lpv, sub = threadIdx.x % lpv, vloc = threadIdx.x / lpv;
    for (long long gv = (long long)blockIdx.x * vpb + vloc; gv < total + vloc; gv += (long long)gridDim.x * vpb) {
        const bool ok = gv < total;   // keep all lanes in the shuffle
        float s = 0.f;
        int n = 0; long long v = 0;
        if (ok) {
            n = (int)(gv / Vt); v = gv % Vt;
            const T* tp = theta + (size_t)gv * g.C;
            const T* pp = phi + ((size_t)n * Vp + phi_voxel(g, v)) * g.C;
            for (int c = sub * VEC; c < g.C; c += lpv * VEC) {
                float a[VEC], b[VEC];
                VecIO<T, VEC>::ld(tp + c, a); VecIO<T, VEC>::ld(pp + c, b);
#pragma unroll
                for (int k = 0; k < VEC; ++k) s = fmaf(lrelu_f(a[k] + b[k], 0.1f), wpsi[c + k], s);
            }
        }
        for (int o = lpv >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (!ok) continue;
        float sg = sigmoid_f(s + bpsi[0]);
        if (sub == 0) Act<T>::st(sigma + gv, sg);
        if constexpr (sizeof(T) == 2) sg = bf2f(f2bf(sg));          // the multiply reads the STORED sigma
        const int pw = (int)(v % g.Wt); const long long r = v / g.Wt; const int ph = (int)(r % g.Ht), pd = (int)(r / g.Ht);
        for (int a = 0; a < mg.s0; ++a)
            for (int b = 0; b < mg.s1; ++b)
                for (int e = 0; e < mg.s2; ++e) {
                    const long long fv = (((long long)n * mg.D + pd * mg.s0 + a) * mg.H + ph * mg.s1 + b) * mg.W + pw * mg.s2 + e;
                    for (int c = sub * VEC; c < mg.C; c += lpv * VEC) {
                        float xv[VEC];
                        VecIO<T, VEC>::ld(x + (size_t)fv * mg.C + c, xv);
#pragma unroll
                        for (int k = 0; k < VEC; ++k) xv[k] *= sg;
                        VecIO<T, VEC>::st(y + (size_t)fv * mg.C + c, xv);
                    }
                }
    }
}

// dsigma[n,p] = sum_{v in window(p)} sum_c dy*x   (lpv lanes per coarse voxel)
// DX != 0 (round 6): the same pass also writes dx = sigma[p] * dy (DX == 2: dx += ...) -- the two kernels of the multiply's backward
// read dy twice and were two launches on the data-gradient chain of every gate
template <typename T, int VEC, int DX = 0>
__global__ void __launch_bounds__(256) mul_sigma_dsig_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                             T* __restrict__ dsigma, MulGeo g, int lpv,
                                                             const T* __restrict__ sigma = nullptr, T* __restrict__ dx = nullptr) {
    const long long Vs = (long long)g.Ds * g.Hs * g.Ws, total = Vs * g.N;
    const int vpb = 256 / lpv, sub = threadIdx.x % lpv, vloc = threadIdx.x / lpv;
    for (long long gp = (long long)blockIdx.x * vpb + vloc; gp < total + vloc; gp += (long long)gridDim.x * vpb) {
        const bool ok = gp < total;
        float s = 0.f;
        if (ok) {
            const int n = (int)(gp / Vs); long long r = gp % Vs;
            const int pw = (int)(r % g.Ws); r /= g.Ws; const int ph = (int)(r % g.Hs); const int pd = (int)(r / g.Hs);
            float sg = 0.f;
            if constexpr (DX != 0) sg = Act<T>::ld(sigma + gp);
            for (int a = 0; a < g.s0; ++a)
                for (int b = 0; b < g.s1; ++b)
                    for (int e = 0; e < g.s2; ++e) {
                        const long long v = (((long long)n * g.D + pd * g.s0 + a) * g.H + ph * g.s1 + b) * g.W + pw * g.s2 + e;
                        for (int c = sub * VEC; c < g.C; c += lpv * VEC) {
                            float xv[VEC], dv[VEC];
                            VecIO<T, VEC>::ld(x + (size_t)v * g.C + c, xv); VecIO<T, VEC>::ld(dy + (size_t)v * g.C + c, dv);
#pragma unroll
                            for (int k = 0; k < VEC; ++k) s = fmaf(xv[k], dv[k], s);
                            if constexpr (DX != 0) {
                                float o[VEC];
#pragma unroll
                                for (int k = 0; k < VEC; ++k) o[k] = sg * dv[k];
                                if constexpr (DX == 2) {
                                    float old[VEC];
                                    VecIO<T, VEC>::ld(dx + (size_t)v * g.C + c, old);
#pragma unroll
                                    for (int k = 0; k < VEC; ++k) o[k] += old[k];
                                }
                                VecIO<T, VEC>::st(dx + (size_t)v * g.C + c, o);
                            }
                        }
                    }
        }
        for (int o = lpv >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (ok && sub == 0) Act<T>::st(dsigma + gp, s);
    }
}

static int make_mulgeo(MulGeo& g, int N, int D, int H, int W, int C, int s0, int s1, int s2) {
    if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0 || s0 <= 0 || s1 <= 0 || s2 <= 0) return M1_ERR_BAD_ARG;
    if (D % s0 || H % s1 || W % s2) return M1_ERR_UNSUPPORTED;
    g = MulGeo{N, D, H, W, C, s0, s1, s2, D / s0, H / s1, W / s2};
    return M1_OK;
}

template <typename T, bool ACC = false>
static int mul_fwd_impl(const void* x, const void* sigma, void* y, const MulGeo& g, hipStream_t st) {
    constexpr int VW = sizeof(T) == 2 ? 8 : 4;
    const long long V = (long long)g.N * g.D * g.H * g.W;
    if (g.C % VW == 0)
        hipLaunchKernelGGL((mul_sigma_kernel<T, VW, ACC>), dim3(gx_for(V * (g.C / VW))), dim3(256), 0, st, (const T*)x, (const T*)sigma, (T*)y, g);
    else
        hipLaunchKernelGGL((mul_sigma_kernel<T, 1, ACC>), dim3(gx_for(V * g.C)), dim3(256), 0, st, (const T*)x, (const T*)sigma, (T*)y, g);
    return m1_check_launch();
}

extern "C" int m1_mul_sigma_fwd(const void* x, const void* sigma, void* y, int N, int D, int H, int W, int C, int s0, int s1,
                                int s2, int dtype, void* stream) {
    if (m1_debug_skip("gate")) return M1_OK;
    if (!x || !sigma || !y) return M1_ERR_BAD_ARG;
    MulGeo g; int rc = make_mulgeo(g, N, D, H, W, C, s0, s1, s2); if (rc) return rc;
    M1ProfScope ps("mul_sigma_fwd", 0.0, 2.0 * N * D * H * W * C * (dtype == M1_BF16 ? 2 : 4), (hipStream_t)stream);
    return dtype == M1_BF16 ? mul_fwd_impl<bf16_t>(x, sigma, y, g, (hipStream_t)stream)
                            : mul_fwd_impl<float>(x, sigma, y, g, (hipStream_t)stream);
}

template <typename T>
static int mul_bwd_impl(const void* x, const void* sigma, const void* dy, void* dx, void* dsigma, const MulGeo& g, int accumulate,
                        hipStream_t st) {
    constexpr int VW = sizeof(T) == 2 ? 8 : 4;
    const long long total = (long long)g.N * g.Ds * g.Hs * g.Ws;
    if (g.C % VW == 0 && M1_CFG("M1_GATE_MUL_BWD_FUSED", 1)) {          // one pass: dsigma and dx (+)= sigma_up * dy
        const int lpv = lanes_per_voxel(g.C, VW);
        if (accumulate)
            hipLaunchKernelGGL((mul_sigma_dsig_kernel<T, VW, 2>), dim3(gx_for(total * lpv)), dim3(256), 0, st, (const T*)x, (const T*)dy, (T*)dsigma, g, lpv, (const T*)sigma, (T*)dx);
        else
            hipLaunchKernelGGL((mul_sigma_dsig_kernel<T, VW, 1>), dim3(gx_for(total * lpv)), dim3(256), 0, st, (const T*)x, (const T*)dy, (T*)dsigma, g, lpv, (const T*)sigma, (T*)dx);
        return m1_check_launch();
    }
    int rc = accumulate ? mul_fwd_impl<T, true>(dy, sigma, dx, g, st) : mul_fwd_impl<T>(dy, sigma, dx, g, st);   // dx (+)= sigma_up * dy
    if (rc) return rc;
    if (g.C % VW == 0) {
        const int lpv = lanes_per_voxel(g.C, VW);
        hipLaunchKernelGGL((mul_sigma_dsig_kernel<T, VW>), dim3(gx_for(total * lpv)), dim3(256), 0, st, (const T*)x, (const T*)dy, (T*)dsigma, g, lpv);
    } else {
        const int lpv = lanes_per_voxel(g.C, 1);
        hipLaunchKernelGGL((mul_sigma_dsig_kernel<T, 1>), dim3(gx_for(total * lpv)), dim3(256), 0, st, (const T*)x, (const T*)dy, (T*)dsigma, g, lpv);
    }
    return m1_check_launch();
}

extern "C" int m1_mul_sigma_bwd(const void* x, const void* sigma, const void* dy, void* dx, void* dsigma, int N, int D, int H,
                                int W, int C, int s0, int s1, int s2, int dtype, int accumulate_dx, void* stream) {
    if (m1_debug_skip("gate")) return M1_OK;
    if (!x || !sigma || !dy || !dx || !dsigma) return M1_ERR_BAD_ARG;
    MulGeo g; int rc = make_mulgeo(g, N, D, H, W, C, s0, s1, s2); if (rc) return rc;
    M1ProfScope ps("mul_sigma_bwd", 0.0, 4.0 * N * D * H * W * C * (dtype == M1_BF16 ? 2 : 4), (hipStream_t)stream);
    return dtype == M1_BF16 ? mul_bwd_impl<bf16_t>(x, sigma, dy, dx, dsigma, g, accumulate_dx, (hipStream_t)stream)
                            : mul_bwd_impl<float>(x, sigma, dy, dx, dsigma, g, accumulate_dx, (hipStream_t)stream);
}

// sigma = gate_sigma_fwd(theta, phi), y = mul_sigma_fwd(x, sigma) as ONE launch (B:113-124).  M1_ERR_UNSUPPORTED (nothing launched):
// channel counts that are no multiple of a 16-byte vector, or a sigma grid that is not x's grid / (s0, s1, s2) -- take the two calls.
template <typename T>
static int gate_mul_fwd_impl(const void* theta, const void* phi, const float* wpsi, const float* bpsi, void* sigma, const void* x, void* y,
                             const Geo& g, const MulGeo& mg, hipStream_t st) {
    constexpr int VW = sizeof(T) == 2 ? 8 : 4;
    if (g.C % VW || mg.C % VW) return M1_ERR_UNSUPPORTED;
    const int lpv = lanes_per_voxel(g.C > mg.C ? g.C : mg.C, VW);
    const long long total = (long long)g.N * g.Dt * g.Ht * g.Wt;
    hipLaunchKernelGGL((gate_sigma_mul_fwd_kernel<T, VW>), dim3(gx_for(total * lpv)), dim3(256), 0, st, (const T*)theta, (const T*)phi, wpsi, bpsi,
                       (T*)sigma, (const T*)x, (T*)y, g, mg, lpv);
    return m1_check_launch();
}
extern "C" int m1_gate_sigma_mul_fwd(const void* theta, const void* phi, const float* wpsi, const float* bpsi, void* sigma, const void* x,
                                     void* y, int N, int Dt, int Ht, int Wt, int Dp, int Hp, int Wp, int Ci, int D, int H, int W, int Cx,
                                     int s0, int s1, int s2, int dtype, void* stream) {
    if (!theta || !phi || !wpsi || !bpsi || !sigma || !x || !y) return M1_ERR_BAD_ARG;
    if (dtype != M1_F32 && dtype != M1_BF16) return M1_ERR_BAD_ARG;
    Geo g; int rc = make_geo(g, N, Dt, Ht, Wt, Dp, Hp, Wp, Ci); if (rc) return rc;
    MulGeo mg; rc = make_mulgeo(mg, N, D, H, W, Cx, s0, s1, s2); if (rc) return rc;
    if (mg.Ds != Dt || mg.Hs != Ht || mg.Ws != Wt) return M1_ERR_UNSUPPORTED;
    if (!M1_CFG("M1_GATE_FWD_FUSED", 1)) return M1_ERR_UNSUPPORTED;
    if (m1_debug_skip("gate")) return M1_OK;
    const int es = dtype == M1_BF16 ? 2 : 4;
    M1ProfScope ps("gate_sigma_mul_fwd", 0.0, ((double)N * Dt * Ht * Wt * Ci + 2.0 * N * D * H * W * Cx) * es, (hipStream_t)stream);
    return dtype == M1_BF16 ? gate_mul_fwd_impl<bf16_t>(theta, phi, wpsi, bpsi, sigma, x, y, g, mg, (hipStream_t)stream)
                            : gate_mul_fwd_impl<float>(theta, phi, wpsi, bpsi, sigma, x, y, g, mg, (hipStream_t)stream);
}
