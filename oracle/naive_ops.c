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

/* ------------------------------------------------------------------------------------------------------------------
 * KAT-7: the WHOLE deterministic M1 forward in plain C loops -- an independent code path for the
 * wiring of networks.py:568-630 (stem, 4 strided SE encoders, 4 attention gates, transposed-conv up-path with channel
 * concats, 4 SE decoders, 1x1x1 logits) and of network_blocks.py:48-80,106-130, written from the reference, sharing
 * nothing with oracle/m1_oracle.py.  dense_skip = deep_supervision = probabilistic = False, att_sub_samp = (1,1,1).
 *
 * Parameters arrive as an array of pointers consumed IN THIS ORDER (each conv: kernel, bias; each norm: gamma, beta):
 *   conve0, norme0,
 *   serse1..serse4            each: conv1,norm1, conv2,norm2, conv3,norm3, conv4,norm4, conv6, conv7
 *   att0..att3                each: theta, phi, psi, W, normW
 *   convtd3, sersd3, convtd2, sersd2, convtd1, sersd1, convtd0, sersd0, logits
 * filters[5], strides[15], kernels[15] as in the M1 constructor (networks.py:38-40).
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct { const double **p; int cur, n; } cursor_t;
static const double *nextp(cursor_t *c) { return c->cur < c->n ? c->p[c->cur++] : NULL; }
typedef struct { double *v; int D, H, W, C; } vol_t;                  /* one NDHWC tensor of batch size N (N is global) */

static size_t vox(const vol_t *t) { return (size_t)t->D * t->H * t->W; }
static vol_t vol_new(int N, int D, int H, int W, int C) {
    vol_t t = { (double *)malloc(sizeof(double) * (size_t)N * D * H * W * C), D, H, W, C };
    return t;
}

static vol_t conv_layer(cursor_t *c, int N, const vol_t *x, int Co, const int *k, const int *s) {
    const double *w = nextp(c), *b = nextp(c);
    vol_t y = vol_new(N, ceil_div(x->D, s[0]), ceil_div(x->H, s[1]), ceil_div(x->W, s[2]), Co);
    naive_conv3d_same(x->v, w, b, y.v, N, x->D, x->H, x->W, x->C, Co, k[0], k[1], k[2], s[0], s[1], s[2]);
    return y;
}
static vol_t convT_layer(cursor_t *c, int N, const vol_t *x, int Co, const int *k, const int *s) {
    const double *w = nextp(c), *b = nextp(c);
    vol_t y = vol_new(N, x->D * s[0], x->H * s[1], x->W * s[2], Co);
    naive_conv3d_transpose_same(x->v, w, b, y.v, N, x->D, x->H, x->W, x->C, Co, k[0], k[1], k[2], s[0], s[1], s[2]);
    return y;
}
static void norm_inplace(cursor_t *c, int N, vol_t *x, double slope) {
    const double *g = nextp(c), *b = nextp(c);
    naive_instance_norm(x->v, g, b, x->v, N, vox(x), x->C, 1e-3, slope);        /* element-wise after the statistics: in place is safe */
}
static vol_t concat2(int N, const vol_t *a, const vol_t *b) {                   /* tf.concat([a, b], axis=-1) */
    vol_t y = vol_new(N, a->D, a->H, a->W, a->C + b->C);
    size_t V = (size_t)N * vox(a);
    for (size_t v = 0; v < V; ++v) {
        memcpy(y.v + v * y.C, a->v + v * a->C, sizeof(double) * a->C);
        memcpy(y.v + v * y.C + a->C, b->v + v * b->C, sizeof(double) * b->C);
    }
    return y;
}
static vol_t concatn(int N, const vol_t *const *ts, int n) {                       /* tf.concat(ts, axis=-1) */
    int C = 0; for (int i = 0; i < n; ++i) C += ts[i]->C;
    vol_t y = vol_new(N, ts[0]->D, ts[0]->H, ts[0]->W, C);
    size_t V = (size_t)N * vox(ts[0]);
    for (size_t v = 0; v < V; ++v) {
        int o = 0;
        for (int i = 0; i < n; ++i) { memcpy(y.v + v * C + o, ts[i]->v + v * ts[i]->C, sizeof(double) * ts[i]->C); o += ts[i]->C; }
    }
    return y;
}
static const int ONE3[3] = { 1, 1, 1 }, K333[3] = { 3, 3, 3 };

/* network_blocks.py:48-80 */
static vol_t se_block(cursor_t *c, int N, const vol_t *x, int F, const int *k, const int *s, int red) {
    vol_t a = conv_layer(c, N, x, F / 4, k, s);            norm_inplace(c, N, &a, 0.1);       /* B:53-55 */
    vol_t b = conv_layer(c, N, &a, F / 4, K333, ONE3);     norm_inplace(c, N, &b, 0.1);       /* B:56-58 */
    vol_t x_ = conv_layer(c, N, &b, F, ONE3, ONE3);        norm_inplace(c, N, &x_, 1.0);      /* B:59-60 */
    vol_t r = conv_layer(c, N, x, F, k, s);                norm_inplace(c, N, &r, 1.0);       /* B:63-65 (C_in != F always) */
    const double *w6 = nextp(c), *b6 = nextp(c), *w7 = nextp(c), *b7 = nextp(c);
    int Fr = F / red;
    size_t V = vox(&x_);
    double *pool = (double *)malloc(sizeof(double) * F), *hid = (double *)malloc(sizeof(double) * Fr);
    double *gate = (double *)malloc(sizeof(double) * F);
    for (int n = 0; n < N; ++n) {
        double *xn = x_.v + (size_t)n * V * F, *rn = r.v + (size_t)n * V * F;
        for (int f = 0; f < F; ++f) {                                                          /* B:68 GlobalAveragePooling3D */
            double sum = 0.0;
            for (size_t v = 0; v < V; ++v) sum += xn[v * F + f];
            pool[f] = sum / (double)V;
        }
        for (int j = 0; j < Fr; ++j) {                                                         /* B:70-71 */
            double t = b6[j];
            for (int f = 0; f < F; ++f) t += pool[f] * w6[(size_t)f * Fr + j];
            hid[j] = t >= 0.0 ? t : 0.1 * t;
        }
        for (int f = 0; f < F; ++f) {                                                          /* B:72-73 */
            double t = b7[f];
            for (int j = 0; j < Fr; ++j) t += hid[j] * w7[(size_t)j * F + f];
            gate[f] = 1.0 / (1.0 + exp(-t));
        }
        for (size_t v = 0; v < V; ++v)
            for (int f = 0; f < F; ++f) {                                                      /* B:74-78: multiply, multiply, leaky */
                double u = xn[v * F + f] * gate[f] * rn[v * F + f];
                xn[v * F + f] = u >= 0.0 ? u : 0.1 * u;
            }
    }
    free(pool); free(hid); free(gate); free(a.v); free(b.v); free(r.v);
    return x_;
}

/* network_blocks.py:106-130 with sub_samp = (1,1,1) */
static vol_t gate_block(cursor_t *c, int N, const vol_t *x, const vol_t *g, int Fi) {
    vol_t th = conv_layer(c, N, x, Fi, ONE3, ONE3);                                            /* B:111 */
    vol_t ph = conv_layer(c, N, g, Fi, ONE3, ONE3);                                            /* B:112 */
    int ud = th.D / ph.D, uh = th.H / ph.H, uw = th.W / ph.W;                                  /* B:113-116 nearest repeat */
    const double *wpsi = nextp(c), *bpsi = nextp(c);
    vol_t y = vol_new(N, x->D, x->H, x->W, x->C);
    for (int n = 0; n < N; ++n)
    for (int d = 0; d < th.D; ++d)
    for (int h = 0; h < th.H; ++h)
    for (int w = 0; w < th.W; ++w) {
        size_t it = (((size_t)n * th.D + d) * th.H + h) * th.W + w;
        size_t ip = (((size_t)n * ph.D + d / ud) * ph.H + h / uh) * ph.W + w / uw;
        double psi = bpsi[0];
        for (int f = 0; f < Fi; ++f) {
            double t = th.v[it * Fi + f] + ph.v[ip * Fi + f];                                   /* B:117 */
            t = t >= 0.0 ? t : 0.1 * t;
            psi += t * wpsi[f];                                                                /* B:118 */
        }
        double sig = 1.0 / (1.0 + exp(-psi));                                                  /* B:119 (sub_samp 1: B:120-123 identity) */
        for (int f = 0; f < x->C; ++f) y.v[it * x->C + f] = sig * x->v[it * x->C + f];         /* B:124 */
    }
    vol_t wy = conv_layer(c, N, &y, Fi, ONE3, ONE3);                                           /* B:127 */
    norm_inplace(c, N, &wy, 1.0);                                                              /* B:128 */
    free(th.v); free(ph.v); free(y.v);
    return wy;
}

int naive_m1_det_forward(const double *x, int N, int D, int H, int W, int Cin, const int *filters, const int *strides,
                         const int *kernels, const int *se_red, int num_classes, const double **params, int nparams,
                         double *logits) {
    cursor_t c = { params, 0, nparams };
    const int *F = filters;
    #define S(i) (strides + 3 * (i))
    #define K(i) (kernels + 3 * (i))
    vol_t in = { (double *)x, D, H, W, Cin };
    vol_t x0 = conv_layer(&c, N, &in, F[0], K(0), S(0)); norm_inplace(&c, N, &x0, 0.1);       /* N:574-576 */
    vol_t e1 = se_block(&c, N, &x0, F[1], K(1), S(1), se_red[1]);                              /* N:579 */
    vol_t e2 = se_block(&c, N, &e1, F[2], K(2), S(2), se_red[2]);
    vol_t e3 = se_block(&c, N, &e2, F[3], K(3), S(3), se_red[3]);
    vol_t em = se_block(&c, N, &e3, F[4], K(4), S(4), se_red[4]);                              /* N:582 */
    vol_t a0 = gate_block(&c, N, &x0, &em, F[0]);                                              /* N:585-588 */
    vol_t a1 = gate_block(&c, N, &e1, &em, F[1]);
    vol_t a2 = gate_block(&c, N, &e2, &em, F[2]);
    vol_t a3 = gate_block(&c, N, &e3, &em, F[3]);
    vol_t d3 = convT_layer(&c, N, &em, F[3], K(4), S(4));                                      /* N:591 */
    vol_t c3 = concat2(N, &d3, &a3);                                                           /* N:596 */
    vol_t u3 = se_block(&c, N, &c3, F[3], K(3), ONE3, se_red[3]);                              /* N:597 */
    vol_t d2 = convT_layer(&c, N, &u3, F[2], K(3), S(3));                                      /* N:600 */
    vol_t c2 = concat2(N, &d2, &a2);                                                           /* N:606 */
    vol_t u2 = se_block(&c, N, &c2, F[2], K(2), ONE3, se_red[2]);
    vol_t d1 = convT_layer(&c, N, &u2, F[1], K(2), S(2));                                      /* N:610 */
    vol_t c1 = concat2(N, &d1, &a1);                                                           /* N:615 */
    vol_t u1 = se_block(&c, N, &c1, F[1], K(1), ONE3, se_red[1]);
    vol_t d0 = convT_layer(&c, N, &u1, F[0], K(1), S(1));                                      /* N:619 */
    vol_t c0 = concat2(N, &d0, &a0);                                                           /* N:623 */
    vol_t u0 = se_block(&c, N, &c0, F[0], K(0), ONE3, se_red[0]);                              /* N:624 */
    vol_t lg = conv_layer(&c, N, &u0, num_classes, ONE3, ONE3);                                /* N:627 */
    #undef S
    #undef K
    int ok = (c.cur == nparams);               /* every parameter consumed exactly once */
    memcpy(logits, lg.v, sizeof(double) * (size_t)N * vox(&lg) * num_classes);
    vol_t all[] = { x0, e1, e2, e3, em, a0, a1, a2, a3, d3, c3, u3, d2, c2, u2, d1, c1, u1, d0, c0, u0, lg };
    for (size_t i = 0; i < sizeof(all) / sizeof(all[0]); ++i) free(all[i].v);
    return ok ? 0 : -1;
}

/* ------------------------------------------------------------------------------------------------------------------
 * The hierarchical PROBABILISTIC M1 train-time forward in plain C loops (networks.py:297-385, 631-728), again written from
 * the reference and sharing nothing with oracle/m1_oracle.py: prior and posterior M1Core with their latent branches
 * (1x1x1 mu/log-sigma heads, reparameterised draw, transposed-conv latent decoder dec_hi*, SE blocks sersp*), the four
 * core passes of a train step that feed its outputs (q_sample, q_mean, p_sample_z_q, p_sample_z_q_mean), the stitching
 * decoder and KL(Q||P).  dense_skip optional (the nested decoder of N:591-621); deep supervision does not touch these
 * outputs in probabilistic mode (its heads are not evaluated here); att_sub_samp = (1,1,1), dropout off.
 *
 * Parameter order of ONE core: the deterministic order documented above (conve0 ... sersd0, logits; with dense_skip
 * convtd3 is followed by convtd3_up1, _up2, _up3, convtd2 by convtd2_up1, _up2, convtd1 by convtd1_up1), then for the levels
 * 3, 2, 1, 0:  mu_logsig{l} (kernel, bias -- only when latent_dims[3-l] != 0), dec_hi{l} (kernel, bias), sersp{l} (SE block).
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct { vol_t ml[4]; vol_t z[4]; int L[4]; vol_t feat; } prob_out_t;     /* index 0 = level 3 (coarsest) ... 3 = level 0 */

static vol_t slice_ch(int N, const vol_t *a, int c0, int c1) {
    vol_t y = vol_new(N, a->D, a->H, a->W, c1 - c0);
    size_t V = (size_t)N * vox(a);
    for (size_t v = 0; v < V; ++v) memcpy(y.v + v * y.C, a->v + v * a->C + c0, sizeof(double) * (size_t)(c1 - c0));
    return y;
}
/* One M1Core.__call__(inputs, prob_mean, prob_z_q) (networks.py:568-728).  z_given[i] != NULL: the latent of level 3-i is
 * that tensor (prob_z_q); else eps[i] != NULL: mu + exp(clip(logsigma, +-0.1)) * eps[i] (distrib.sample()); else mu. */
static int core_prob_forward(const double **params, int nparams, int N, const vol_t *in, const int *F, const int *strides,
                             const int *kernels, const int *se_red, int num_classes, const int *Ldims, int dense,
                             const double *const *z_given, const double *const *eps, prob_out_t *out) {
    cursor_t c = { params, 0, nparams };
    #define S(i) (strides + 3 * (i))
    #define K(i) (kernels + 3 * (i))
    vol_t x0 = conv_layer(&c, N, in, F[0], K(0), S(0)); norm_inplace(&c, N, &x0, 0.1);        /* N:574-576 */
    vol_t e1 = se_block(&c, N, &x0, F[1], K(1), S(1), se_red[1]);                              /* N:579-582 */
    vol_t e2 = se_block(&c, N, &e1, F[2], K(2), S(2), se_red[2]);
    vol_t e3 = se_block(&c, N, &e2, F[3], K(3), S(3), se_red[3]);
    vol_t em = se_block(&c, N, &e3, F[4], K(4), S(4), se_red[4]);
    vol_t a0 = gate_block(&c, N, &x0, &em, F[0]);                                              /* N:585-588 */
    vol_t a1 = gate_block(&c, N, &e1, &em, F[1]);
    vol_t a2 = gate_block(&c, N, &e2, &em, F[2]);
    vol_t a3 = gate_block(&c, N, &e3, &em, F[3]);
    /* nested decoder, N:589-624; dense_skip: every transposed conv output is carried up to all finer stages (N:592-594, 601-603, 612) */
    vol_t nil = { NULL, 0, 0, 0, 0 };
    vol_t d3 = convT_layer(&c, N, &em, F[3], K(4), S(4));
    vol_t d3u1 = nil, d3u2 = nil, d3u3 = nil, d2u1 = nil, d2u2 = nil, d1u1 = nil;
    if (dense) {
        d3u1 = convT_layer(&c, N, &d3, F[2], K(3), S(3));
        d3u2 = convT_layer(&c, N, &d3u1, F[1], K(2), S(2));
        d3u3 = convT_layer(&c, N, &d3u2, F[0], K(1), S(1));
    }
    vol_t c3 = concat2(N, &d3, &a3);                                                            /* N:595 */
    vol_t u3 = se_block(&c, N, &c3, F[3], K(3), ONE3, se_red[3]);
    vol_t d2 = convT_layer(&c, N, &u3, F[2], K(3), S(3));
    vol_t c2;
    if (dense) {
        d2u1 = convT_layer(&c, N, &d2, F[1], K(2), S(2));
        d2u2 = convT_layer(&c, N, &d2u1, F[0], K(1), S(1));
        const vol_t *ts[] = { &d2, &d3u1, &a2 }; c2 = concatn(N, ts, 3);                        /* N:603 */
    } else c2 = concat2(N, &d2, &a2);
    vol_t u2 = se_block(&c, N, &c2, F[2], K(2), ONE3, se_red[2]);
    vol_t d1 = convT_layer(&c, N, &u2, F[1], K(2), S(2));
    vol_t c1;
    if (dense) {
        d1u1 = convT_layer(&c, N, &d1, F[0], K(1), S(1));
        const vol_t *ts[] = { &d1, &d2u1, &d3u2, &a1 }; c1 = concatn(N, ts, 4);                 /* N:613 */
    } else c1 = concat2(N, &d1, &a1);
    vol_t u1 = se_block(&c, N, &c1, F[1], K(1), ONE3, se_red[1]);
    vol_t d0 = convT_layer(&c, N, &u1, F[0], K(1), S(1));
    vol_t c0;
    if (dense) { const vol_t *ts[] = { &d0, &d1u1, &d2u2, &d3u3, &a0 }; c0 = concatn(N, ts, 5); }  /* N:621 */
    else c0 = concat2(N, &d0, &a0);
    vol_t u0 = se_block(&c, N, &c0, F[0], K(0), ONE3, se_red[0]);
    vol_t lg = conv_layer(&c, N, &u0, num_classes, ONE3, ONE3);                                /* N:627 (not an output here) */
    /* ---- latent branch, N:632-723: level 3 reads convm, the others the running decoder features ---- */
    const vol_t *skip[4] = { &c3, &c2, &c1, &c0 };                                             /* uconv3_ ... uconv0_ */
    vol_t feat = em; int feat_owned = 0;
    for (int i = 0; i < 4; ++i) {
        const int lvl = 3 - i, L = Ldims[i];
        out->L[i] = L; out->ml[i].v = NULL; out->z[i].v = NULL;
        vol_t dec_in = feat; int dec_in_owned = 0;
        if (L != 0) {
            vol_t ml = conv_layer(&c, N, &feat, 2 * L, ONE3, ONE3);                            /* mu_logsig{lvl}: N:637, 660, 684, 708 */
            vol_t z = vol_new(N, feat.D, feat.H, feat.W, L);
            size_t V = (size_t)N * vox(&feat);
            for (size_t v = 0; v < V; ++v)
                for (int l = 0; l < L; ++l) {
                    double mu = ml.v[v * 2 * L + l], ls = ml.v[v * 2 * L + L + l];
                    ls = ls < -0.1 ? -0.1 : (ls > 0.1 ? 0.1 : ls);                             /* N:640 clip_by_value */
                    if (z_given && z_given[i]) z.v[v * L + l] = z_given[i][v * L + l];         /* N:643 */
                    else if (eps && eps[i])   z.v[v * L + l] = mu + exp(ls) * eps[i][v * L + l];   /* N:645 sample() */
                    else                      z.v[v * L + l] = mu;                             /* N:644 .loc */
                }
            out->ml[i] = ml; out->z[i] = z;
            dec_in = concat2(N, &z, &feat); dec_in_owned = 1;                                  /* N:651 tf.concat([z, features]) */
        }
        vol_t up = convT_layer(&c, N, &dec_in, F[lvl], K(lvl + 1), S(lvl + 1));                /* dec_hi{lvl}: N:548-555 */
        vol_t cat = concat2(N, &up, skip[i]);                                                  /* N:650-651 */
        vol_t nf = se_block(&c, N, &cat, F[lvl], K(lvl), ONE3, se_red[lvl]);                   /* sersp{lvl}: N:556-563 */
        if (dec_in_owned) free(dec_in.v);
        free(up.v); free(cat.v);
        if (feat_owned) free(feat.v);
        feat = nf; feat_owned = 1;
    }
    out->feat = feat;
    #undef S
    #undef K
    int ok = (c.cur == nparams);
    vol_t all[] = { x0, e1, e2, e3, em, a0, a1, a2, a3, d3, c3, u3, d2, c2, u2, d1, c1, u1, d0, c0, u0, lg, d3u1, d3u2, d3u3, d2u1, d2u2, d1u1 };
    for (size_t i = 0; i < sizeof(all) / sizeof(all[0]); ++i) free(all[i].v);
    return ok ? 0 : -1;
}
static void prob_out_free(prob_out_t *o) {
    for (int i = 0; i < 4; ++i) { free(o->ml[i].v); free(o->z[i].v); }
    free(o->feat.v);
}

/* networks.py:297-385: train_conv = stitch(prior(image, z = posterior(image+label, mean).latents).decoder_features) and
 * prob_kl = sum_levels mean_n sum_voxels KL(q_sample.dist || prior(image, z = q_sample.latents).dist).
 * x: (N,D,H,W,Cin) with the label as channel Cin-(num_classes-1)-1 (the off-by-one slice of N:300-301 as written);
 * eps_q[i]: N(0,1) draws of latent level 3-i for the sampled posterior pass (NULL where latent_dims[i] == 0). */
int naive_m1_prob_train_forward(const double *x, int N, int D, int H, int W, int Cin, const int *filters, const int *strides,
                                const int *kernels, const int *se_red, int num_classes, const int *latent_dims, int dense_skip,
                                const double *const *eps_q, const double **prior_params, int n_prior,
                                const double **post_params, int n_post, const double *stitch_w, const double *stitch_b,
                                double *train_conv, double *kl_out) {
    vol_t in = { (double *)x, D, H, W, Cin };
    const int nl = num_classes - 1;
    vol_t image = slice_ch(N, &in, 0, Cin - nl);                                               /* N:300 inputs[..., :-(nc-1)] */
    vol_t label = slice_ch(N, &in, Cin - nl - 1, Cin - 1);                                     /* N:301 inputs[..., -(nc-1)-1:-1] */
    vol_t post_in = concat2(N, &image, &label);                                                /* N:348 */
    prob_out_t qs, qm, pz, pzm;
    int rc = 0;
    rc |= core_prob_forward(post_params, n_post, N, &post_in, filters, strides, kernels, se_red, num_classes, latent_dims, dense_skip, NULL, eps_q, &qs);   /* N:348 */
    rc |= core_prob_forward(post_params, n_post, N, &post_in, filters, strides, kernels, se_red, num_classes, latent_dims, dense_skip, NULL, NULL, &qm);    /* N:349 */
    const double *zs[4], *zm[4];
    for (int i = 0; i < 4; ++i) { zs[i] = qs.z[i].v; zm[i] = qm.z[i].v; }
    rc |= core_prob_forward(prior_params, n_prior, N, &image, filters, strides, kernels, se_red, num_classes, latent_dims, dense_skip, zs, NULL, &pz);      /* N:351 */
    rc |= core_prob_forward(prior_params, n_prior, N, &image, filters, strides, kernels, se_red, num_classes, latent_dims, dense_skip, zm, NULL, &pzm);     /* N:352 */
    naive_conv3d_same(pzm.feat.v, stitch_w, stitch_b, train_conv, N, pzm.feat.D, pzm.feat.H, pzm.feat.W, pzm.feat.C, num_classes,
                      1, 1, 1, 1, 1, 1);                                                       /* N:356, B:277 */
    double kl = 0.0;
    for (int i = 0; i < 4; ++i)
        if (latent_dims[i] != 0) {                                                             /* N:373-379 */
            double k = 0.0;
            naive_kl_mvn_diag(qs.ml[i].v, pz.ml[i].v, &k, N, vox(&qs.ml[i]), latent_dims[i], 0.1);
            kl += k;
        }
    *kl_out = kl;                                                                              /* N:385 */
    prob_out_free(&qs); prob_out_free(&qm); prob_out_free(&pz); prob_out_free(&pzm);
    free(image.v); free(label.v); free(post_in.v);
    return rc;
}

