// conv_direct.hip -- generic (any channel count, any concat, fp32/bf16 storage, fp32 math) NDHWC
// Conv3D / Conv3DTranspose with TF 'same' semantics: forward, data-gradient, weight-gradient.
//
// One LDS-tiled gather kernel serves four roles:
//   mode 0 "forward gather"     in = o*s + k - p        : Conv3D forward,           Conv3DTranspose dgrad
//   mode 1 "transposed gather"  in = (o + p - k)/s      : Conv3D dgrad,             Conv3DTranspose forward
// Mode 1 walks one output PARITY CLASS (o mod s) per blockIdx.y so that only the taps that actually hit
// that class are visited (no zero-stuffed work for stride 2).
// The virtual channel concat (tf.concat axis=-1 at networks.py:596-623,653-725) is walked in the loader.
//
// This is the correctness-first / odd-shape path (stem Cin=2..3, latent z, class logits).  The MFMA
// implicit-GEMM path in conv_mfma.hip takes the wide layers.
#include "common.h"
#include "reduce.h"
#include "gather.h"

struct GatherP {
    const void* src[M1_MAX_SRC];
    int srcC[M1_MAX_SRC];
    int srcOff[M1_MAX_SRC + 1];
    int nsrc;
    int CC;           // contraction channels (sum of srcC)
    int ID, IH, IW;   // spatial dims of the gathered tensor(s)
    void* out;
    int OC;           // channels of `out`
    int OD, OH, OW;   // spatial dims of `out`
    int N;
    const float* w;
    long long wST, wSC, wSO;  // w[tap*wST + cc*wSC + (oc+oc_off)*wSO]
    int oc_off, cc_off, accumulate;
    const float* bias;
    int kd, kh, kw;
    int mode;
    int sd, sh, sw, pd, ph, pw;
};

#define GT_V 64   // voxels per block
#define GT_O 64   // output channels per block
#define GT_K 16   // contraction chunk

template <typename T>
__global__ void __launch_bounds__(256) conv_gather_kernel(GatherP p) {
    __shared__ float A_s[GT_V][GT_K + 1];
    __shared__ float W_s[GT_K][GT_O];
    __shared__ int q_d[GT_V], q_h[GT_V], q_w[GT_V];
    __shared__ const T* s_src[M1_MAX_SRC];
    __shared__ int s_srcC[M1_MAX_SRC], s_srcOff[M1_MAX_SRC + 1];

    const int tid = threadIdx.x;
    const int ocTiles = (p.OC + GT_O - 1) / GT_O;
    const int n = blockIdx.z / ocTiles;
    const int oc0 = (blockIdx.z % ocTiles) * GT_O;

    int pdc = 0, phc = 0, pwc = 0, QD = p.OD, QH = p.OH, QW = p.OW;
    if (p.mode == 1) {
        const int pc = blockIdx.y;
        pwc = pc % p.sw; phc = (pc / p.sw) % p.sh; pdc = pc / (p.sw * p.sh);
        QD = p.OD > pdc ? (p.OD - pdc + p.sd - 1) / p.sd : 0;
        QH = p.OH > phc ? (p.OH - phc + p.sh - 1) / p.sh : 0;
        QW = p.OW > pwc ? (p.OW - pwc + p.sw - 1) / p.sw : 0;
    }
    const long long QV = (long long)QD * QH * QW;
    const long long v0 = (long long)blockIdx.x * GT_V;
    if (v0 >= QV) return;

    if (tid < GT_V) {
        long long lin = v0 + tid;
        if (lin < QV) {
            q_w[tid] = (int)(lin % QW); long long r = lin / QW;
            q_h[tid] = (int)(r % QH); q_d[tid] = (int)(r / QH);
        } else { q_d[tid] = -1; q_h[tid] = 0; q_w[tid] = 0; }
    }
    if (tid < M1_MAX_SRC) { s_src[tid] = (const T*)p.src[tid]; s_srcC[tid] = p.srcC[tid]; }
    if (tid <= M1_MAX_SRC) s_srcOff[tid] = p.srcOff[tid];
    __syncthreads();

    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

    const int tv = tid >> 4, to = tid & 15;   // compute mapping: 4 voxels x 4 oc per thread
    const int lc = tid & 15, lv = tid >> 4;   // loader mapping
    const int nsrc = p.nsrc;

    for (int kd = 0; kd < p.kd; ++kd) {
        int offd;
        if (p.mode == 0) offd = kd - p.pd;
        else { int t = pdc + p.pd - kd; if (t % p.sd != 0) continue; offd = t / p.sd; }
        for (int kh = 0; kh < p.kh; ++kh) {
            int offh;
            if (p.mode == 0) offh = kh - p.ph;
            else { int t = phc + p.ph - kh; if (t % p.sh != 0) continue; offh = t / p.sh; }
            for (int kw = 0; kw < p.kw; ++kw) {
                int offw;
                if (p.mode == 0) offw = kw - p.pw;
                else { int t = pwc + p.pw - kw; if (t % p.sw != 0) continue; offw = t / p.sw; }
                const long long tap = ((long long)kd * p.kh + kh) * p.kw + kw;

                long long aoff[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int v = lv + 16 * i;
                    aoff[i] = -1;
                    if (q_d[v] >= 0) {
                        int id, ih, iw;
                        if (p.mode == 0) { id = q_d[v] * p.sd + offd; ih = q_h[v] * p.sh + offh; iw = q_w[v] * p.sw + offw; }
                        else { id = q_d[v] + offd; ih = q_h[v] + offh; iw = q_w[v] + offw; }
                        if (id >= 0 && id < p.ID && ih >= 0 && ih < p.IH && iw >= 0 && iw < p.IW)
                            aoff[i] = (((long long)n * p.ID + id) * p.IH + ih) * p.IW + iw;
                    }
                }
                for (int c0 = 0; c0 < p.CC; c0 += GT_K) {
                    {   // A tile: lanes along the contraction channel (coalesced NDHWC reads)
                        const int c = c0 + lc;
                        int s = 0;
                        while (s + 1 < nsrc && c >= s_srcOff[s + 1]) ++s;
                        const T* sp = s_src[s];
                        const int sC = s_srcC[s], cl = c - s_srcOff[s];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            float val = 0.f;
                            if (c < p.CC && aoff[i] >= 0) val = Act<T>::ld(sp + aoff[i] * sC + cl);
                            A_s[lv + 16 * i][lc] = val;
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {   // W tile
                        const int e = tid + 256 * i;
                        const int cw = e >> 6, ow = e & 63;
                        const int c = c0 + cw, oc = oc0 + ow;
                        float val = 0.f;
                        if (c < p.CC && oc < p.OC) val = p.w[tap * p.wST + (long long)(c + p.cc_off) * p.wSC + (long long)(oc + p.oc_off) * p.wSO];
                        W_s[cw][ow] = val;
                    }
                    __syncthreads();
#pragma unroll
                    for (int cc = 0; cc < GT_K; ++cc) {
                        float a[4], b[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) a[i] = A_s[tv * 4 + i][cc];
                        const float4 bv = *reinterpret_cast<const float4*>(&W_s[cc][to * 4]);
                        b[0] = bv.x; b[1] = bv.y; b[2] = bv.z; b[3] = bv.w;
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
                    }
                    __syncthreads();
                }
            }
        }
    }

    T* out = (T*)p.out;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int v = tv * 4 + i;
        if (q_d[v] < 0) continue;
        int od = q_d[v], oh = q_h[v], ow = q_w[v];
        if (p.mode == 1) { od = od * p.sd + pdc; oh = oh * p.sh + phc; ow = ow * p.sw + pwc; }
        const long long o = (((long long)n * p.OD + od) * p.OH + oh) * p.OW + ow;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int oc = oc0 + to * 4 + j;
            if (oc < p.OC) {
                float r = acc[i][j];
                if (p.bias) r += p.bias[oc + p.oc_off];
                if (p.accumulate) r += Act<T>::ld(out + o * p.OC + oc);
                Act<T>::st(out + o * p.OC + oc, r);
            }
        }
    }
}

static int launch_gather(const GatherP& p, int dtype, hipStream_t st) {
    int classes = 1;
    long long maxQV = (long long)p.OD * p.OH * p.OW;
    if (p.mode == 1) {
        classes = p.sd * p.sh * p.sw;
        maxQV = (long long)((p.OD + p.sd - 1) / p.sd) * ((p.OH + p.sh - 1) / p.sh) * ((p.OW + p.sw - 1) / p.sw);
    }
    const int ocTiles = (p.OC + GT_O - 1) / GT_O;
    dim3 grid((unsigned)cdiv_ll(maxQV, GT_V), classes, p.N * ocTiles);
    m1_note_kernel("conv_gather");
    if (dtype == M1_BF16) hipLaunchKernelGGL(conv_gather_kernel<bf16_t>, grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL(conv_gather_kernel<float>, grid, dim3(256), 0, st, p);
    return m1_check_launch();
}

// ------------------------------------------------------------------------------------------------
// weight gradient:  R[tap][a+a_off][b+b_off] += sum_{n,v} A[n, v*s + k - p][a] * B[n, v][b]
// ------------------------------------------------------------------------------------------------
struct WgradP : WgradSpec {
    long long vox_per_split;
};

#define WG_T 64
#define WG_K 16

template <typename T>
__global__ void __launch_bounds__(256) conv_wgrad_kernel(WgradP p) {
    __shared__ float A_s[WG_K][WG_T];
    __shared__ float B_s[WG_K][WG_T];
    __shared__ long long a_off_s[WG_K], b_off_s[WG_K];

    const int tid = threadIdx.x;
    const int bTiles = (p.CB + WG_T - 1) / WG_T;
    const int a0 = (blockIdx.x / bTiles) * WG_T, b0 = (blockIdx.x % bTiles) * WG_T;
    const int tap = blockIdx.y;
    const int kw = tap % p.kw, kh = (tap / p.kw) % p.kh, kd = tap / (p.kw * p.kh);
    const long long BV = (long long)p.BD * p.BH * p.BW, TV = BV * p.N;
    const long long vbeg = (long long)blockIdx.z * p.vox_per_split;
    long long vend = vbeg + p.vox_per_split; if (vend > TV) vend = TV;

    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    const int ta = tid >> 4, tb = tid & 15;
    const T* A = (const T*)p.A; const T* B = (const T*)p.B;

    for (long long vs = vbeg; vs < vend; vs += WG_K) {
        if (tid < WG_K) {
            const long long v = vs + tid;
            long long ao = -1, bo = -1;
            if (v < vend) {
                const int n = (int)(v / BV); long long r = v % BV;
                const int bw = (int)(r % p.BW); r /= p.BW;
                const int bh = (int)(r % p.BH); const int bd = (int)(r / p.BH);
                bo = v;
                const int ad = bd * p.sd + kd - p.pd, ah = bh * p.sh + kh - p.ph, aw = bw * p.sw + kw - p.pw;
                if (ad >= 0 && ad < p.AD && ah >= 0 && ah < p.AH && aw >= 0 && aw < p.AW)
                    ao = (((long long)n * p.AD + ad) * p.AH + ah) * p.AW + aw;
            }
            a_off_s[tid] = ao; b_off_s[tid] = bo;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i, k = e >> 6, c = e & 63;
            const long long ao = a_off_s[k], bo = b_off_s[k];
            float av = 0.f, bv = 0.f;
            if (ao >= 0 && a0 + c < p.CA) av = Act<T>::ld(A + ao * p.CA + a0 + c);
            if (bo >= 0 && ao >= 0 && b0 + c < p.CB) bv = Act<T>::ld(B + bo * p.CB + b0 + c);
            A_s[k][c] = av; B_s[k][c] = bv;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < WG_K; ++k) {
            const float4 av = *reinterpret_cast<const float4*>(&A_s[k][ta * 4]);
            const float4 bv = *reinterpret_cast<const float4*>(&B_s[k][tb * 4]);
            const float a[4] = {av.x, av.y, av.z, av.w}, b[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int a = a0 + ta * 4 + i;
        if (a >= p.CA) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int b = b0 + tb * 4 + j;
            if (b < p.CB) atomicAdd(p.R + (long long)tap * p.RT + (long long)(a + p.a_off) * p.RSA + (b + p.b_off), acc[i][j]);
        }
    }
}

int m1_direct_wgrad(const WgradSpec& spec, hipStream_t st) {
    WgradP p; static_cast<WgradSpec&>(p) = spec; p.vox_per_split = 0;
    const int dtype = spec.dtype;
    const int aTiles = (p.CA + WG_T - 1) / WG_T, bTiles = (p.CB + WG_T - 1) / WG_T;
    const int taps = p.kd * p.kh * p.kw;
    const long long TV = (long long)p.N * p.BD * p.BH * p.BW;
    long long splits = cdiv_ll(2048, (long long)aTiles * bTiles * taps);
    const long long max_splits = cdiv_ll(TV, 256);
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    long long vps = cdiv_ll(TV, splits);
    vps = cdiv_ll(vps, WG_K) * WG_K;
    splits = cdiv_ll(TV, vps);
    p.vox_per_split = vps;
    dim3 grid(aTiles * bTiles, taps, (unsigned)splits);
    m1_note_kernel("conv_wgrad_direct");
    if (dtype == M1_BF16) hipLaunchKernelGGL(conv_wgrad_kernel<bf16_t>, grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL(conv_wgrad_kernel<float>, grid, dim3(256), 0, st, p);
    return m1_check_launch();
}

// ------------------------------------------------------------------------------------------------
// skinny weight gradient: one side has only a handful of channels (stem: Cin = 2..3; class logits / latent heads:
// Cout = 2..6; latent z members).  Lanes run along the WIDE side's channels (coalesced), every thread keeps the
// NT x SM partial sums of its wide channel in registers, voxel-lanes are folded through LDS, one atomic per output.
// ------------------------------------------------------------------------------------------------
template <typename T, int NT, int SM, bool SMALL_A>
__global__ void __launch_bounds__(256) wgrad_skinny_kernel(WgradP p) {
    __shared__ float red[256];
    const int tid = threadIdx.x;
    const int wide = SMALL_A ? p.CB : p.CA, small = SMALL_A ? p.CA : p.CB;
    int WP = 1; while (WP < wide) WP <<= 1;
    const int VL = 256 / WP, wc = tid % WP, vl = tid / WP;
    const long long BV = (long long)p.BD * p.BH * p.BW, TV = BV * p.N;
    const long long vbeg = (long long)blockIdx.x * p.vox_per_split;
    long long vend = vbeg + p.vox_per_split; if (vend > TV) vend = TV;
    const T* A = (const T*)p.A; const T* B = (const T*)p.B;
    float acc[NT][SM];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int k = 0; k < SM; ++k) acc[t][k] = 0.f;
    if (wc < wide) {
        for (long long v = vbeg + vl; v < vend; v += VL) {
            const int n = (int)(v / BV); long long r = v % BV;
            const int bw = (int)(r % p.BW); r /= p.BW;
            const int bh = (int)(r % p.BH); const int bd = (int)(r / p.BH);
            float bval[SM];
            if (SMALL_A) bval[0] = Act<T>::ld(B + v * p.CB + wc);
            else {
#pragma unroll
                for (int k = 0; k < SM; ++k) bval[k] = k < small ? Act<T>::ld(B + v * p.CB + k) : 0.f;
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int kw = t % p.kw, kh = (t / p.kw) % p.kh, kd = t / (p.kw * p.kh);
                const int ad = bd * p.sd + kd - p.pd, ah = bh * p.sh + kh - p.ph, aw = bw * p.sw + kw - p.pw;
                if (ad < 0 || ad >= p.AD || ah < 0 || ah >= p.AH || aw < 0 || aw >= p.AW) continue;
                const long long ao = (((long long)n * p.AD + ad) * p.AH + ah) * p.AW + aw;
                if (SMALL_A) {
#pragma unroll
                    for (int k = 0; k < SM; ++k) if (k < small) acc[t][k] = fmaf(Act<T>::ld(A + ao * p.CA + k), bval[0], acc[t][k]);
                } else {
                    const float av = Act<T>::ld(A + ao * p.CA + wc);
#pragma unroll
                    for (int k = 0; k < SM; ++k) acc[t][k] = fmaf(av, bval[k], acc[t][k]);
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int k = 0; k < SM; ++k) {
            if (k >= small) continue;                   // block-uniform
            red[tid] = acc[t][k];
            __syncthreads();
            if (vl == 0 && wc < wide) {
                float s = 0.f;
                for (int q = 0; q < VL; ++q) s += red[q * WP + wc];
                const int a = SMALL_A ? k : wc, b = SMALL_A ? wc : k;
                atomicAdd(p.R + (long long)t * p.RT + (long long)(a + p.a_off) * p.RSA + (b + p.b_off), s);
            }
            __syncthreads();
        }
}

bool m1_skinny_wgrad_supported(const WgradSpec& g) {
    const int nt = g.kd * g.kh * g.kw;
    if (nt != 1 && nt != 9 && nt != 27) return false;
    if (g.CA <= 4 && g.CB <= 256) return true;
    if (g.CB <= 4 && g.CA <= 256) return true;
    if (g.CB <= 8 && g.CA <= 256 && nt == 1) return true;
    return false;
}

template <typename T>
static int skinny_launch(WgradP p, hipStream_t st) {
    const int nt = p.kd * p.kh * p.kw;
    const long long TV = (long long)p.N * p.BD * p.BH * p.BW;
    long long blocks = cdiv_ll(TV, 512); if (blocks > 768) blocks = 768; if (blocks < 1) blocks = 1;   // atomics on few addresses: keep the block count moderate
    p.vox_per_split = cdiv_ll(TV, blocks);
    blocks = cdiv_ll(TV, p.vox_per_split);
    const dim3 grid((unsigned)blocks), blk(256);
    const bool smallA = p.CA <= 4 && p.CB <= 256;
    m1_note_kernel("wgrad_skinny");
#define SK(NT_, SM_, SA_) hipLaunchKernelGGL((wgrad_skinny_kernel<T, NT_, SM_, SA_>), grid, blk, 0, st, p)
    if (smallA) { if (nt == 1) SK(1, 4, true); else if (nt == 9) SK(9, 4, true); else SK(27, 4, true); }
    else if (p.CB <= 4) { if (nt == 1) SK(1, 4, false); else if (nt == 9) SK(9, 4, false); else SK(27, 4, false); }
    else SK(1, 8, false);
#undef SK
    return m1_check_launch();
}
int m1_skinny_wgrad(const WgradSpec& spec, hipStream_t st) {
    WgradP p; static_cast<WgradSpec&>(p) = spec; p.vox_per_split = 0;
    return spec.dtype == M1_BF16 ? skinny_launch<bf16_t>(p, st) : skinny_launch<float>(p, st);
}

// ------------------------------------------------------------------------------------------------
// spec-based entry (dispatch.hip builds the specs)
// ------------------------------------------------------------------------------------------------
int m1_direct_gather(const GatherSpec& g, hipStream_t st) {
    GatherP p{};
    int off = 0;
    p.nsrc = g.nsrc;
    for (int i = 0; i < M1_MAX_SRC; ++i) {
        if (i < g.nsrc) { p.src[i] = g.src[i]; p.srcC[i] = g.srcC[i]; p.srcOff[i] = off; off += g.srcC[i]; }
        else { p.src[i] = nullptr; p.srcC[i] = 0; p.srcOff[i] = off; }
    }
    p.srcOff[M1_MAX_SRC] = off; p.CC = off;
    p.ID = g.ID; p.IH = g.IH; p.IW = g.IW; p.out = g.out; p.OC = g.OC; p.OD = g.OD; p.OH = g.OH; p.OW = g.OW; p.N = g.N;
    p.w = g.w; p.wST = g.wST; p.wSC = g.wSC; p.wSO = g.wSO; p.oc_off = g.oc_off; p.cc_off = g.cc_off; p.accumulate = g.accumulate;
    p.bias = g.bias; p.kd = g.kd; p.kh = g.kh; p.kw = g.kw; p.mode = g.mode;
    p.sd = g.sd; p.sh = g.sh; p.sw = g.sw; p.pd = g.pd; p.ph = g.ph; p.pw = g.pw;
    return launch_gather(p, g.dtype, st);
}
