// gather.h -- the logical problems behind the six conv entry points, shared by the direct (conv_direct.hip)
// and matrix-core (conv_mfma.hip, wgrad_mfma.hip) implementations.  dispatch.hip builds these specs.
#pragma once
#include "common.h"

// out[o][oc] = bias[oc+oc_off] + sum_tap sum_c  X[gather(o,tap)][c] * w[tap*wST + (c+cc_off)*wSC + (oc+oc_off)*wSO]
//   mode 0: gather = o*s + tap - p  (Conv3D forward, Conv3DTranspose dgrad)
//   mode 1: gather = (o + p - tap)/s when divisible (Conv3D dgrad, Conv3DTranspose forward)
struct GatherSpec {
    const void* src[M1_MAX_SRC]; int srcC[M1_MAX_SRC]; int nsrc;   // virtual concat of the contraction axis
    int ID, IH, IW;            // extent of the gathered tensors
    void* out; int OC;         // output tensor and its channel count
    // nout > 0: the OC output channels are a virtual concat too -- columns [outOff(m), outOff(m)+outC[m]) go to tensor outs[m]
    // (own row pitch outC[m], own accumulate flag, nullptr = not wanted).  The data gradient of a conv over a concat is ONE
    // launch for all members then (dispatch.hip dgrad_common); `out` is unused.
    int nout; void* outs[M1_MAX_SRC]; int outC[M1_MAX_SRC]; int outAcc[M1_MAX_SRC];
    int OD, OH, OW, N;
    const float* w; long long wST, wSC, wSO; int oc_off, cc_off;
    const float* bias;
    // a second weight tensor (conv1 || conv4 of an SE block as one problem, network_blocks.py:53,64): output columns >= oc_split
    // (forward: [y1 | y4]) or contraction channels >= c_split (data gradient over [dy1 | dy4]) come from w2 with its own strides;
    // bias2 / stats_out2 belong to the columns >= oc_split
    const float* w2; long long w2ST, w2SC, w2SO; int oc_split, c_split;
    const float* bias2; float* stats_out2;
    int kd, kh, kw, mode, sd, sh, sw, pd, ph, pw;
    int dtype, accumulate;
    float* stats_out;          // optional (Conv3D forward only): per-(n,oc) {mean, rstd} of the output for the InstanceNorm
    float* stats_ws;           //   that follows; fused into the epilogue when the tiling allows, else a reduction pass
    float stats_eps;
    // optional (data gradients with ONE output tensor): the output is d(a) of a = lrelu(IN(x)) (network_blocks.py:54-58) -- the
    // kernel that writes it also emits the two sums the InstanceNorm backward needs, per tile and (sample, channel):
    // {sum dy, sum dy*xh}, dy = out * lrelu'(gamma*xh + beta), xh = (x - mean)*rstd, into ib_partial [N][*ib_nparts][OC][2]
    // ("emit them as partials from the kernel that produces da").  *ib_nparts = 0 when the kernel that took
    // the problem cannot (the caller then runs the stand-alone reduction).
    const void* ib_x; const float* ib_stats; const float* ib_gamma; const float* ib_beta; float ib_slope;
    float* ib_partial; int* ib_nparts;
    int ib_cap;                // rows per sample `ib_partial` holds (a kernel that would write more leaves *ib_nparts = 0)
};

int m1_stats_internal(const void* x, int N, long long V, int C, int dtype, float eps, float* stats, float* ws, hipStream_t st);
size_t m1_stats_ws_floats(int N, long long V, int C);

// R[tap*RT + (a+a_off)*RSA + (b+b_off)] += sum_{n,v} A[n, v*s + tap - p][a] * B[n, v][b]
struct WgradSpec {
    const void* A; int CA; int AD, AH, AW;
    const void* B; int CB; int BD, BH, BW;
    int N;
    float* R; long long RT, RSA; int a_off, b_off;
    int kd, kh, kw, sd, sh, sw, pd, ph, pw;
    int dtype;
    float* rx; long long rx_floats;   // caller-owned scratch for per-split partial copies of the gradient (inside the wgrad `ws`)
    float* bsum;      // optional: bsum[b + b_off] += sum_v B[v][b] (bias gradient of a Conv3D, fused into the tap whose
    int bsum_tap;     // shifted partner is always inside the volume: tap index bsum_tap); NULL = off
};

int m1_direct_gather(const GatherSpec& g, hipStream_t st);
int m1_direct_wgrad(const WgradSpec& g, hipStream_t st);
bool m1_skinny_wgrad_supported(const WgradSpec& g);
int m1_skinny_wgrad(const WgradSpec& g, hipStream_t st);
bool m1_mfma_supported(const GatherSpec& g);
size_t m1_mfma_ws_bytes(const GatherSpec& g);
int m1_mfma_gather(const GatherSpec& g, void* ws, int ws_packed, hipStream_t st);
// layers with <= 4 (forward) / <= 8 (pointwise data gradient) channels on the thin side (conv_thin.hip): 1 = taken, *rc = result
int m1_thin_conv_try(const GatherSpec& g, hipStream_t st, int* rc);
int m1_pack_batch_internal(const void* const* jobs_dev, const int* prefix_dev, int njobs, int total_blocks, hipStream_t st);
bool m1_mfma_wgrad_supported(const WgradSpec& g);
int m1_mfma_wgrad(const WgradSpec& g, hipStream_t st);
bool m1_tf_wgrad_supported(const WgradSpec& g);      // tap-fused variant (wgrad_tf.hip)
// nw / nb: floats of the whole weight / bias gradient the spec's R / bsum point into.  M1_ERR_WORKSPACE / UNSUPPORTED:
// nothing was launched, the caller takes the per-tap kernel instead.
int m1_tf_wgrad(const WgradSpec& g, long long nw, int nb, hipStream_t st);
int m1_tf_wgrad_multi(const WgradSpec& g, long long nw, int nb, hipStream_t st, int nmem, const void* const* Am, const int* a_offs, long long rx_mem);
// partial-copy scratch (the spec's rx region when it holds `floats`, else nullptr = take the atomic path) and the fold of
// `ncopies` copies of stride `stride` floats into g.R / g.bsum (bias sums sit at offset nw inside a copy)
static inline float* m1_wg_rx_get(const WgradSpec& g, long long floats) { return (g.rx && g.rx_floats >= floats) ? g.rx : nullptr; }
int m1_wg_rx_finish(float* rx, long long stride, int ncopies, const WgradSpec& g, long long nw, hipStream_t st);
int m1_mfma_wgrad_ex(const WgradSpec& g, long long nw, int nb, hipStream_t st);
bool m1_tap_wgrad_supported(const WgradSpec& g);     // per-tap kernel on the transpose read (wgrad_tap.hip), >= 64 channels
int m1_tap_wgrad(const WgradSpec& g, long long nw, int nb, hipStream_t st);
// tap-fused kernel on 32x32x16 MFMAs for stride-1 layers with multiples of 64 channels on both sides (wgrad_t3.hip); nmem
// equal-width concat members in one launch (nmem = 1: Am / a_offs unused)
bool m1_t3_wgrad_supported(const WgradSpec& g);
int m1_t3_wgrad(const WgradSpec& g, long long nw, int nb, hipStream_t st, int nmem, const void* const* Am, const int* a_offs, long long rx_mem);
// fp32, few channels (<= 32 on a side or up to 32 tile pairs), stride 1: tap-fused kernel on 32x32 tiles (wgrad_t3s.hip)
bool m1_t3s_wgrad_supported(const WgradSpec& g);
int m1_t3s_wgrad(const WgradSpec& g, long long nw, int nb, hipStream_t st);
// fp32 pointwise (1x1x1, stride 1) weight gradient: operand-stream-bound GEMM over the voxel list (wgrad_t3s.hip)
bool m1_pwf_wgrad_supported(const WgradSpec& g);
int m1_pwf_wgrad(const WgradSpec& g, long long nw, int nb, hipStream_t st);
int m1_colsum_internal(const void* x, int N, long long V, int C, int dtype, float* out, float* ws, hipStream_t st, int accumulate);

// deferred-fold switch of m1_wg_rx_finish (wgrad_tf.hip): returns the previous setting
long long m1_stats_rows_cap(long long V);       // partial rows per sample of a statistics workspace (norm.hip)
int m1_fold_defer_set(int on);
int m1_fold_defer_get();
// deferred bias gradients of the transposed convs (norm.hip): queued under m1_wgrad_defer, launched by m1_wgrad_fold_pending
bool m1_colsum_defer(const void* x, int N, long long V, int C, int dtype, float* out, float* ws, int accumulate);
int m1_colsum_launch_pending(hipStream_t st);
void m1_colsum_drop_pending();
