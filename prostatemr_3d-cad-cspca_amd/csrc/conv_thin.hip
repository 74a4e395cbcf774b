// conv_thin.hip -- bf16 layers with a handful of channels on ONE side, where a matrix-core tile is mostly padding and the
// implicit-GEMM gather moves 16-byte segments for 4..6 useful bytes (networks.py:478 the image stem 2|3 -> 32 at full
// resolution; networks.py:737-751 the 2-channel logit heads' data gradients):
//
//   thin_fwd_kernel    Conv3D forward, <= 4 input channels, 1x3x3 / 3x3x3, stride 1, <= 32 output channels: vector FMAs on a
//                      register tile of 4 voxels x 8 output channels per lane (input tile with halo and the whole kernel in LDS as
//                      fp32), bias, bf16 rounding, InstanceNorm statistics of the rounded outputs as one partial row per (sample,
//                      block), 16-byte stores.  158 -> ~50 us per launch at (4,20,160,160).
//   thin_pw_dgrad_kernel  pointwise data gradient with <= 8 gradient channels (dX[v][:] = dY[v][:] W^T): a streaming kernel, one
//                      16-byte store per lane.
//
// Both read the fp32 weights themselves and round them to bf16 first, exactly as the packed panels of the matrix-core kernels
// do: the two paths agree to accumulation order.
#include "gather.h"
#include "reduce.h"

#define TH_TW 32            // tile: 8 rows x 32 columns of one (sample, depth) slice
#define TH_TH 8
#define TH_MAXK 108         // taps * input channels
#define TH_MAXOC 32

struct ThinP {
    const bf16_t* x; bf16_t* out; const float* w; const float* bias;
    long long wST, wSC, wSO;
    int N, D, H, W, Cin, OC;
    int kd, kh, kw, pd, ph, pw;
    int tiles_w, tiles_h, tiles_ps, ntiles, nsplit;
    float* stat_partial;      // [N][nsplit][OC][2] or nullptr
};

__global__ void __launch_bounds__(256) thin_fwd_kernel(ThinP p) {
    __shared__ __attribute__((aligned(16))) float Ws[TH_MAXK * TH_MAXOC];            // [k][oc], bf16-rounded
    __shared__ float Xs[3 * (TH_TH + 2) * (TH_TW + 2) * 4];                         // [kd][row][col][ci]
    __shared__ float red[4][TH_MAXOC][2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Cin = p.Cin, OC = p.OC, KT = p.kd * p.kh * p.kw * Cin;
    const int XW = TH_TW + p.kw - 1, XH = TH_TH + p.kh - 1;
    for (int e = tid; e < KT * TH_MAXOC; e += 256) {
        const int k = e / TH_MAXOC, oc = e % TH_MAXOC, t = k / Cin, ci = k % Cin;
        Ws[e] = oc < OC ? bf2f(f2bf(p.w[(long long)t * p.wST + (long long)ci * p.wSC + (long long)oc * p.wSO])) : 0.f;
    }
    // lane -> (8 output channels, 4 consecutive voxels of a tile row): 4 channel groups x 64 voxel groups
    const int ocg = tid & 3, vg = tid >> 2, vr = vg >> 3, vc = (vg & 7) * 4;
    const int oc0 = ocg * 8;
    float bias_r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bias_r[j] = (p.bias && oc0 + j < OC) ? p.bias[oc0 + j] : 0.f;
    float ssum[8], ssq[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { ssum[j] = 0.f; ssq[j] = 0.f; }
    const bool want_stats = p.stat_partial != nullptr;
    auto flush = [&](int n) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float s = ssum[j], q = ssq[j];
#pragma unroll
            for (int o = 4; o < 64; o <<= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
            if (lane < 4) { red[wave][oc0 + j][0] = s; red[wave][oc0 + j][1] = q; }
            ssum[j] = 0.f; ssq[j] = 0.f;
        }
        __syncthreads();
        if (tid < OC) {
            const float s = (red[0][tid][0] + red[1][tid][0]) + (red[2][tid][0] + red[3][tid][0]);
            const float q = (red[0][tid][1] + red[1][tid][1]) + (red[2][tid][1] + red[3][tid][1]);
            float* dst = p.stat_partial + (((long long)n * p.nsplit + blockIdx.x) * OC + tid) * 2;
            dst[0] = s; dst[1] = q;
        }
        __syncthreads();
    };
    int cur_n = 0;
    const int xrows = p.kd * XH * XW;
    for (int kt = blockIdx.x; kt < p.ntiles; kt += p.nsplit) {
        int r = kt; const int tw = r % p.tiles_w; r /= p.tiles_w; const int th = r % p.tiles_h; r /= p.tiles_h; const int od = r % p.D; const int n = r / p.D;
        if (want_stats) { for (; cur_n < n; ++cur_n) flush(cur_n); }
        __syncthreads();                                  // (the previous tile's readers are done; first pass: Ws is in place)
        const int id0 = od - p.pd, ih0 = th * TH_TH - p.ph, iw0 = tw * TH_TW - p.pw;
        for (int e = tid; e < xrows; e += 256) {
            const int a = e / (XH * XW), r2 = e - a * (XH * XW), b = r2 / XW, c = r2 - b * XW;
            const int id = id0 + a, ih = ih0 + b, iw = iw0 + c;
            const bool ok = (unsigned)id < (unsigned)p.D && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
            const bf16_t* src = p.x + ((((long long)n * p.D + id) * p.H + ih) * p.W + iw) * Cin;
            for (int ci = 0; ci < Cin; ++ci) Xs[e * Cin + ci] = ok ? bf2f(src[ci]) : 0.f;
        }
        __syncthreads();
        float acc[4][8];
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[v][j] = bias_r[j];
        // per (depth tap, row tap, input channel): the 6 input values under the lane's 4 voxels serve the 3 column taps
        for (int ab = 0; ab < p.kd * p.kh; ++ab) {
            const int a = ab / p.kh, b = ab - a * p.kh;
            const float* xr = Xs + ((a * XH + b + vr) * XW + vc) * Cin;
            for (int ci = 0; ci < Cin; ++ci) {
                float xv[6];
#pragma unroll
                for (int q = 0; q < 6; ++q) xv[q] = xr[q * Cin + ci];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float* wk = Ws + ((ab * 3 + c) * Cin + ci) * TH_MAXOC + oc0;
                    const float4 w0 = *reinterpret_cast<const float4*>(wk), w1 = *reinterpret_cast<const float4*>(wk + 4);
                    const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
                    for (int v = 0; v < 4; ++v)
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[v][j] = fmaf(xv[v + c], wv[j], acc[v][j]);
                }
            }
        }
        const int oh = th * TH_TH + vr;
        if (oh < p.H && oc0 < OC) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int ow = tw * TH_TW + vc + v;
                if (ow >= p.W) continue;
                uint4 o;
                o.x = (unsigned)f2bf(acc[v][0]) | ((unsigned)f2bf(acc[v][1]) << 16); o.y = (unsigned)f2bf(acc[v][2]) | ((unsigned)f2bf(acc[v][3]) << 16);
                o.z = (unsigned)f2bf(acc[v][4]) | ((unsigned)f2bf(acc[v][5]) << 16); o.w = (unsigned)f2bf(acc[v][6]) | ((unsigned)f2bf(acc[v][7]) << 16);
                *reinterpret_cast<uint4*>(p.out + ((((long long)n * p.D + od) * p.H + oh) * p.W + ow) * OC + oc0) = o;
                if (want_stats) {
                    const unsigned u[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float lo = __uint_as_float(u[q] << 16), hi = __uint_as_float(u[q] & 0xffff0000u);
                        ssum[2 * q] += lo; ssq[2 * q] += lo * lo; ssum[2 * q + 1] += hi; ssq[2 * q + 1] += hi * hi;
                    }
                }
            }
        }
    }
    if (want_stats) { for (; cur_n < p.N; ++cur_n) flush(cur_n); }
}

struct ThinD {
    const bf16_t* dy; bf16_t* dx; const float* w;
    long long wSC, wSO; long long nvox; int CC, OC, accumulate;
};
// dx[v][oc] (+)= sum_c dy[v][c] * w[c*wSC + oc*wSO]
__global__ void __launch_bounds__(256) thin_pw_dgrad_kernel(ThinD p) {
    extern __shared__ float Wd[];                         // [c][OC]
    for (int e = threadIdx.x; e < p.CC * p.OC; e += 256) {
        const int c = e / p.OC, oc = e % p.OC;
        Wd[e] = bf2f(f2bf(p.w[(long long)c * p.wSC + (long long)oc * p.wSO]));
    }
    __syncthreads();
    const int og = p.OC >> 3;
    const long long tot = p.nvox * og;
    for (long long u = (long long)blockIdx.x * 256 + threadIdx.x; u < tot; u += (long long)gridDim.x * 256) {
        const long long v = u / og; const int oc0 = (int)(u - v * og) * 8;
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const bf16_t* d = p.dy + v * p.CC;
        for (int c = 0; c < p.CC; ++c) {
            const float dv = bf2f(d[c]);
            const float4 w0 = *reinterpret_cast<const float4*>(Wd + c * p.OC + oc0), w1 = *reinterpret_cast<const float4*>(Wd + c * p.OC + oc0 + 4);
            a[0] = fmaf(dv, w0.x, a[0]); a[1] = fmaf(dv, w0.y, a[1]); a[2] = fmaf(dv, w0.z, a[2]); a[3] = fmaf(dv, w0.w, a[3]);
            a[4] = fmaf(dv, w1.x, a[4]); a[5] = fmaf(dv, w1.y, a[5]); a[6] = fmaf(dv, w1.z, a[6]); a[7] = fmaf(dv, w1.w, a[7]);
        }
        uint4* dst = reinterpret_cast<uint4*>(p.dx + v * p.OC + oc0);
        if (p.accumulate) {                               // round(new) + old, as the matrix-core epilogues do
            const uint4 o = *dst; const unsigned u4[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                a[2 * q] = bf2f(f2bf(a[2 * q])) + __uint_as_float(u4[q] << 16);
                a[2 * q + 1] = bf2f(f2bf(a[2 * q + 1])) + __uint_as_float(u4[q] & 0xffff0000u);
            }
        }
        uint4 o;
        o.x = (unsigned)f2bf(a[0]) | ((unsigned)f2bf(a[1]) << 16); o.y = (unsigned)f2bf(a[2]) | ((unsigned)f2bf(a[3]) << 16);
        o.z = (unsigned)f2bf(a[4]) | ((unsigned)f2bf(a[5]) << 16); o.w = (unsigned)f2bf(a[6]) | ((unsigned)f2bf(a[7]) << 16);
        *dst = o;
    }
}

// 1 = the problem was taken (*rc = result), 0 = not a thin layer: the caller goes on to the matrix-core kernels
int m1_thin_conv_try(const GatherSpec& g, hipStream_t st, int* rc) {
    int en = M1_CFG("M1_THIN", 1);
    if (!en || g.dtype != M1_BF16 || g.nsrc != 1 || g.w2 || g.oc_off || g.cc_off || g.ib_x) return 0;
    const long long Vout = (long long)g.OD * g.OH * g.OW;
    void* out = g.out; int acc = g.accumulate;
    if (g.nout > 1) return 0;
    if (g.nout == 1) { if (!g.outs[0] || g.outC[0] != g.OC) return 0; out = g.outs[0]; acc = g.outAcc[0]; }
    if (!out || (((uintptr_t)out) & 15)) return 0;
    const int CC = g.srcC[0], taps = g.kd * g.kh * g.kw;
    if (g.mode == 0 && CC <= 4 && g.sd == 1 && g.sh == 1 && g.sw == 1 && g.kh == 3 && g.kw == 3 && (g.kd == 1 || g.kd == 3) &&
        g.OC % 8 == 0 && g.OC <= TH_MAXOC && !acc && taps * CC <= TH_MAXK && g.ID == g.OD && g.IH == g.OH && g.IW == g.OW &&
        ((long long)g.N * Vout >= 65536 || en == 2)) {
        ThinP p{};
        p.x = (const bf16_t*)g.src[0]; p.out = (bf16_t*)out; p.w = g.w; p.bias = g.bias; p.wST = g.wST; p.wSC = g.wSC; p.wSO = g.wSO;
        p.N = g.N; p.D = g.OD; p.H = g.OH; p.W = g.OW; p.Cin = CC; p.OC = g.OC; p.kd = g.kd; p.kh = g.kh; p.kw = g.kw; p.pd = g.pd; p.ph = g.ph; p.pw = g.pw;
        p.tiles_w = (g.OW + TH_TW - 1) / TH_TW; p.tiles_h = (g.OH + TH_TH - 1) / TH_TH; p.tiles_ps = g.OD * p.tiles_h * p.tiles_w;
        const long long nt = (long long)g.N * p.tiles_ps;
        if (nt >= (1ll << 30)) return 0;
        p.ntiles = (int)nt;
        long long nsplit = 1024; if (nsplit > nt) nsplit = nt;
        const long long cap = (Vout + 63) / 64;           // partial rows per sample the statistics workspace holds (m1_stats_ws_floats)
        if (nsplit > cap) nsplit = cap;
        p.nsplit = (int)nsplit;
        const bool stats = g.stats_out && g.stats_ws;
        if (g.stats_out && !stats) return 0;
        p.stat_partial = stats ? g.stats_ws : nullptr;
        m1_note_kernel("thin_fwd");
        hipLaunchKernelGGL(thin_fwd_kernel, dim3((unsigned)p.nsplit), dim3(256), 0, st, p);
        *rc = m1_check_launch();
        if (!*rc && stats) *rc = m1_reduce_finalize_launch<2>(g.stats_ws, g.N, g.OC, p.nsplit, g.stats_out, Vout, g.stats_eps, st);
        return 1;
    }
    if (g.mode == 1 && taps == 1 && g.sd == 1 && g.sh == 1 && g.sw == 1 && CC <= 8 && g.OC % 8 == 0 && g.OC <= 1024 && !g.stats_out && !g.bias &&
        g.ID == g.OD && g.IH == g.OH && g.IW == g.OW) {
        ThinD p{};
        p.dy = (const bf16_t*)g.src[0]; p.dx = (bf16_t*)out; p.w = g.w; p.wSC = g.wSC; p.wSO = g.wSO; p.nvox = (long long)g.N * Vout;
        p.CC = CC; p.OC = g.OC; p.accumulate = acc;
        long long nb = cdiv_ll(p.nvox * (g.OC / 8), 256 * 4); if (nb > 4096) nb = 4096; if (nb < 1) nb = 1;
        m1_note_kernel("thin_pw_dgrad");
        hipLaunchKernelGGL(thin_pw_dgrad_kernel, dim3((unsigned)nb), dim3(256), (size_t)CC * g.OC * sizeof(float), st, p);
        *rc = m1_check_launch();
        return 1;
    }
    return 0;
}
