/* Independent plain-C loop restatement of the third-party op semantics the M1 hot path rests on
 * (SURVEY.md App. B-1..B-6).  TEST INFRASTRUCTURE ONLY -- "parity unpinned" (see oracle/m1_oracle.py):
 * it follows the documented behaviour of tensorflow 2.5.0 / tensorflow_addons 0.14.0 /
 * tensorflow_probability 0.13.0 (tf2.5/requirements.txt:1,5,7), not a TensorFlow run.
 *
 * It shares no code with oracle/m1_oracle.py (torch) and exists so that the two restatements pin each
 * other (KAT-7), and as the scalar "port" CPU baseline for single ops.
 *
 * Call sites in the reference (relative to /root/reference/tf2.5/scripts/model/unets/):
 *   conv3d_same            network_blocks.py:37,39,41,43 ; networks.py:472,526
 *   conv3d_transpose_same  networks.py:496-499,505-507,513-514,520,546-553
 *   instance_norm          network_blocks.py:38,40,42,44,104 ; networks.py:473
 *   kl_mvn_diag            networks.py:375  (tfp.distributions.kl_divergence)
 *
 * Build: gcc -O2 -shared -fPIC -o oracle/_build/libm1naive.so oracle/naive_ops.c -lm
 * All tensors are NDHWC, double precision, C-contiguous.
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

static int ceil_div(int a, int b) { return (a + b - 1) / b; }

/* App. B-1: out=ceil(in/s); pad_total=max((out-1)*s+k-in,0); pad_before=pad_total/2 (floor). */
static void same_pad(int in, int k, int s, int *out, int *pb) {
    int o = ceil_div(in, s);
    int tot = (o - 1) * s + k - in;
    if (tot < 0) tot = 0;
    *out = o;
    *pb = tot / 2;
}

/* x: (N,D,H,W,Ci)  w: (kd,kh,kw,Ci,Co)  b: (Co) or NULL  y: (N,OD,OH,OW,Co) */
int naive_conv3d_same(const double *x, const double *w, const double *b, double *y,
                      int N, int D, int H, int W, int Ci, int Co,
                      int kd, int kh, int kw, int sd, int sh, int sw) {
    int OD, OH, OW, pd, ph, pw;
    same_pad(D, kd, sd, &OD, &pd);
    same_pad(H, kh, sh, &OH, &ph);
    same_pad(W, kw, sw, &OW, &pw);
    for (int n = 0; n < N; ++n)
    for (int od = 0; od < OD; ++od)
    for (int oh = 0; oh < OH; ++oh)
    for (int ow = 0; ow < OW; ++ow) {
        double *yo = y + ((((size_t)n * OD + od) * OH + oh) * OW + ow) * Co;
        for (int co = 0; co < Co; ++co) yo[co] = b ? b[co] : 0.0;
        for (int a = 0; a < kd; ++a) {
            int id = od * sd + a - pd;
            if (id < 0 || id >= D) continue;
            for (int c = 0; c < kh; ++c) {
                int ih = oh * sh + c - ph;
                if (ih < 0 || ih >= H) continue;
                for (int e = 0; e < kw; ++e) {
                    int iw = ow * sw + e - pw;
                    if (iw < 0 || iw >= W) continue;
                    const double *xi = x + ((((size_t)n * D + id) * H + ih) * W + iw) * Ci;
                    const double *wt = w + (((size_t)a * kh + c) * kw + e) * Ci * Co;
                    for (int ci = 0; ci < Ci; ++ci) {
                        double xv = xi[ci];
                        const double *wr = wt + (size_t)ci * Co;
                        for (int co = 0; co < Co; ++co) yo[co] += xv * wr[co];
                    }
                }
            }
        }
    }
    return 0;
}

/* App. B-2, written as a SCATTER (the torch restatement is a gather/crop -- different code path):
 * x: (N,D,H,W,Ci)  w: (kd,kh,kw,Co,Ci)  y: (N,D*sd,H*sh,W*sw,Co);  j = i*s + k - pb, pb=max(k-s,0)/2 */
int naive_conv3d_transpose_same(const double *x, const double *w, const double *b, double *y,
                                int N, int D, int H, int W, int Ci, int Co,
                                int kd, int kh, int kw, int sd, int sh, int sw) {
    int OD = D * sd, OH = H * sh, OW = W * sw;
    int pd = (kd - sd > 0 ? kd - sd : 0) / 2;
    int ph = (kh - sh > 0 ? kh - sh : 0) / 2;
    int pw = (kw - sw > 0 ? kw - sw : 0) / 2;
    size_t total = (size_t)N * OD * OH * OW;
    for (size_t v = 0; v < total; ++v)
        for (int co = 0; co < Co; ++co) y[v * Co + co] = b ? b[co] : 0.0;
    for (int n = 0; n < N; ++n)
    for (int id = 0; id < D; ++id)
    for (int ih = 0; ih < H; ++ih)
    for (int iw = 0; iw < W; ++iw) {
        const double *xi = x + ((((size_t)n * D + id) * H + ih) * W + iw) * Ci;
        for (int a = 0; a < kd; ++a) {
            int od = id * sd + a - pd;
            if (od < 0 || od >= OD) continue;
            for (int c = 0; c < kh; ++c) {
                int oh = ih * sh + c - ph;
                if (oh < 0 || oh >= OH) continue;
                for (int e = 0; e < kw; ++e) {
                    int ow = iw * sw + e - pw;
                    if (ow < 0 || ow >= OW) continue;
                    double *yo = y + ((((size_t)n * OD + od) * OH + oh) * OW + ow) * Co;
                    const double *wt = w + (((size_t)a * kh + c) * kw + e) * Co * Ci;
                    for (int co = 0; co < Co; ++co) {
                        const double *wr = wt + (size_t)co * Ci;
                        double acc = 0.0;
                        for (int ci = 0; ci < Ci; ++ci) acc += xi[ci] * wr[ci];
                        yo[co] += acc;
                    }
                }
            }
        }
    }
    return 0;
}

/* App. B-3: per (n,c) mean and BIASED variance over D*H*W, eps inside the rsqrt.
 * slope: leaky-relu negative slope applied afterwards (1.0 = none). */
int naive_instance_norm(const double *x, const double *gamma, const double *beta, double *y,
                        int N, size_t V, int C, double eps, double slope) {
    for (int n = 0; n < N; ++n)
    for (int c = 0; c < C; ++c) {
        const double *xp = x + (size_t)n * V * C + c;
        double *yp = y + (size_t)n * V * C + c;
        double mu = 0.0;
        for (size_t v = 0; v < V; ++v) mu += xp[v * C];
        mu /= (double)V;
        double var = 0.0;
        for (size_t v = 0; v < V; ++v) { double d = xp[v * C] - mu; var += d * d; }
        var /= (double)V;
        double r = 1.0 / sqrt(var + eps);
        for (size_t v = 0; v < V; ++v) {
            double t = (xp[v * C] - mu) * r * gamma[c] + beta[c];
            yp[v * C] = t >= 0.0 ? t : slope * t;
        }
    }
    return 0;
}

/* App. B-6: ml_* = (N,V,2L) head outputs [mu | logsigma]; logsigma clipped to +-clip (networks.py:642).
 * out[n] = sum_voxels KL(q||p);  returns mean_n out[n] through *kl (networks.py:376-377). */
int naive_kl_mvn_diag(const double *ml_q, const double *ml_p, double *kl,
                      int N, size_t V, int L, double clip) {
    double tot = 0.0;
    for (int n = 0; n < N; ++n) {
        double inst = 0.0;
        for (size_t v = 0; v < V; ++v) {
            const double *q = ml_q + ((size_t)n * V + v) * 2 * L;
            const double *p = ml_p + ((size_t)n * V + v) * 2 * L;
            double s = 0.0;
            for (int d = 0; d < L; ++d) {
                double lq = q[L + d], lp = p[L + d];
                lq = lq < -clip ? -clip : (lq > clip ? clip : lq);
                lp = lp < -clip ? -clip : (lp > clip ? clip : lp);
                double sq = exp(lq), sp = exp(lp);
                double dm = (q[d] - p[d]) / sp;
                double rs = sq / sp;
                s += rs * rs + dm * dm - 1.0 + 2.0 * (lp - lq);
            }
            inst += 0.5 * s;
        }
        tot += inst;
    }
    *kl = tot / (double)N;
    return 0;
}
