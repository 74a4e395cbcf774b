/* m1hip.h -- C ABI of the MI355X-native M1 hot path (libm1hip.so, gfx950 only).
 *
 * The reference (DIAGNijmegen/prostateMR_3D-CAD-csPCa, tf2.5/) owns NO native code: every op below is,
 * in the reference, a call into a TensorFlow / TF-Addons / TF-Probability layer.  Each entry point
 * cites the reference call site(s) whose arithmetic it replaces, relative to
 * tf2.5/scripts/model/unets/  (N: = networks.py, B: = network_blocks.py).
 *
 * Conventions
 *   - plain pointers and sizes only; the caller (PyTorch, or any host) owns every buffer, including
 *     workspaces; no device allocation, no hidden synchronisation; every launch goes to the caller's
 *     hipStream_t (graph-capturable: no memset / memcpy nodes, zero fills are kernels).
 *   - process-global host state the library DOES keep (all of it behind mutexes, none of it device memory):
 *     the tuning-switch table (m1_config_*), the queue of deferred weight-gradient folds between m1_wgrad_defer(1)
 *     and m1_wgrad_fold_pending / _drop (the queued jobs point into caller-owned workspaces, which the caller keeps
 *     alive until then), the test hook m1_set_force_direct, the opt-in profiler (m1_prof_*) and the opt-in
 *     kernel-choice log (m1_debug_kernels).  One thread drives the library at a time per process.
 *   - results are bit-reproducible run to run in the default configuration: every reduction across blocks goes
 *     through per-split partials folded in a fixed order.  Floating-point atomics remain compiled in behind
 *     switches only (M1_WG_DET=0, m1_set_force_direct: the generic fp32 reference kernels of conv_direct.hip).
 *   - activations are NDHWC, C-contiguous; `dtype` selects their storage type
 *     (M1_F32 / M1_BF16); parameters, statistics, reductions and gradients of parameters are fp32.
 *   - every function returns 0 on success or a negative m1_status.
 *   - `void* stream` is a hipStream_t.
 */
#ifndef M1HIP_H
#define M1HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum m1_dtype { M1_F32 = 0, M1_BF16 = 1 };

enum m1_status {
    M1_OK = 0,
    M1_ERR_BAD_ARG = -1,
    M1_ERR_UNSUPPORTED = -2,
    M1_ERR_LAUNCH = -3,
    M1_ERR_WORKSPACE = -4
};

#define M1_MAX_SRC 6

/* One member of a virtual channel-concat (tf.concat(axis=-1), N:596,604,606,613,615,621,623,653,677,
 * 701,725).  The concat is never materialised: kernels walk the list. */
typedef struct {
    const void* ptr; /* (N, D, H, W, C) */
    int C;
    int _pad;
} m1_src_t;

/* Geometry of one Conv3D(padding='same') or Conv3DTranspose(padding='same').
 * For Conv3D:           (D,H,W) = input extent,  output = ceil(in/s)       
 * For Conv3DTranspose:  (D,H,W) = input extent,  output = in*s              */
typedef struct {
    int N, D, H, W;
    int Cin, Cout;
    int kd, kh, kw;
    int sd, sh, sw;
    int dtype;
    int nsrc;                 /* number of concat members forming the Cin axis */
    m1_src_t src[M1_MAX_SRC]; /* sum of C == Cin */
} m1_conv_desc_t;

const char* m1_status_name(int status);
int m1_abi_version(void);

/* ---- Conv3D(padding='same') + bias : B:37,39,41,43,100-103 ; N:472,526,529-531,534-537 ; B:275 ----
 * w: Keras layout (kd,kh,kw,Cin,Cout) fp32; bias (Cout) fp32 or NULL; y: (N,OD,OH,OW,Cout). */
/* ws: caller-owned scratch of m1_conv_ws_bytes(d, transposed, role) bytes, 256-byte aligned (packed bf16/fp32
 * weight panels for the matrix-core kernels; bias-gradient partials for wgrad). role: 0 fwd, 1 dgrad, 2 wgrad. */
size_t m1_conv_ws_bytes(const m1_conv_desc_t* d, int transposed, int role);
/* Packed-panel refresh (no reference counterpart: the panels are this library's own copy of the Keras kernels,
 * train_model.py:231's optimizer step changes them once per step).  The first pack of a panel (ws_packed == 0) leaves
 * a job record in front of it inside ws; ws must therefore be ZERO-FILLED when it is first handed over.
 * m1_conv_pack_jobs writes the device addresses of the records of (d, transposed, role in {0,1}) into jobs_out
 * (room for M1_MAX_SRC entries) and returns their number; m1_pack_batch re-packs, in one launch, every filled record
 * of the device array jobs_dev[njobs] from the current weight values (records never filled are skipped).
 * block_prefix_dev (optional, device, njobs + 1 ints, prefix[0] = 0, prefix[njobs] = total_blocks): blocks of 256 threads per
 * job, sized by the caller in proportion to each job's weight count (>= 1 each); NULL = 48 blocks for every job. */
int m1_conv_pack_jobs(const m1_conv_desc_t* d, int transposed, int role, void* ws, void** jobs_out);
int m1_pack_batch(const void* const* jobs_dev, const int* block_prefix_dev, int njobs, int total_blocks, void* stream);
/* ws_packed != 0: ws still holds the weight panels a previous call with the SAME descriptor geometry, role and
 * (unchanged) weights left there -- the pack pass is skipped (the prior / posterior cores run twice per step). */
/* stats (optional, (N,Cout,2) fp32): {mean, rstd} (eps 1e-3, biased variance) of y per (n, channel) for the
 * InstanceNormalization that follows (B:54-60, N:575) -- accumulated from the stored (rounded) outputs in the
 * conv's own epilogue, so the statistics cost no extra pass over y. */
int m1_conv3d_fwd(const m1_conv_desc_t* d, const float* w, const float* bias, void* y, float* stats, void* ws,
                  int ws_packed, void* stream);
/* dx[i]: gradient buffer of concat member i (same shape/dtype as src[i]) or NULL to skip it.
 * accumulate (NULL = all 0): accumulate[i] != 0 -> dx[i] += instead of dx[i] = : a tensor read by several layers
 * (an SE block's input feeds conv1 and conv4, B:53,64; an encoder output also feeds its attention gate, N:584-590)
 * gets its gradient summed by the kernels' own epilogues, in call order, instead of by separate add passes. */
int m1_conv3d_dgrad(const m1_conv_desc_t* d, const float* w, const void* dy, void* const* dx, const int* accumulate,
                    void* ws, int ws_packed, void* stream);
/* 1 when both pair entry points below accept this shape (they return M1_ERR_UNSUPPORTED otherwise): ask before building a graph. */
int m1_conv3d_pair_supported(const m1_conv_desc_t* d, int C1);
/* Data gradient of a single-input Conv3D whose input is a = lrelu(IN(x)) -- conv2 / conv3 of an SEResNetBottleNeck (B:54-59): `da`
 * = d(a), and the kernel that writes it also emits the two sums the InstanceNorm backward needs (dbeta = sum dy,
 * dgamma = sum dy*xh, dy = da*lrelu'(gamma*xh+beta)) per tile into partial [N][*nparts][Cin][2] -- no separate reduction pass over
 * (x, da).  partial: N * partial_rows * Cin * 2 + N * Cin * 2 + 64 floats with partial_rows = m1_conv3d_dgrad_inbwd_rows(d) (a kernel
 * that would write more rows than the caller states leaves *nparts = 0).  *nparts = 0 on return: the kernel that took this shape has no such
 * epilogue; da is complete, finish with m1_instnorm_bwd; otherwise with m1_instnorm_bwd_partials. */
int m1_conv3d_dgrad_inbwd_rows(const m1_conv_desc_t* d);
int m1_conv3d_dgrad_inbwd(const m1_conv_desc_t* d, const float* w, const void* dy, void* da, const void* x, const float* stats,
                          const float* gamma, const float* beta, float slope, float* partial, int partial_rows, int* nparts,
                          void* ws, int ws_packed, void* stream);
/* conv1 || conv4 of an SEResNetBottleNeck as ONE problem (B:53 and B:64 apply Conv3D(F/4, k, s) and Conv3D(F, k, s) to the same
 * input): d->Cout = C1 + C4; w1 (kd,kh,kw,Cin,C1), w4 (kd,kh,kw,Cin,C4) stay separate Keras tensors.  Forward writes y1 (…,C1) and
 * y4 (…,C4) and, optionally, both (N,C,2) statistics tensors; the data gradient contracts over the virtual concat [dy1 | dy4].
 * ws: m1_conv_ws_bytes(d, 0, role) of the SAME descriptor (roles 0 / 1), same zero-fill and ws_packed contract as m1_conv3d_fwd /
 * m1_conv3d_dgrad.  M1_ERR_UNSUPPORTED (nothing launched): take the two single convs instead (channel counts that are not
 * multiples of one 16-byte segment, the halo-tile member-group regime, m1_set_force_direct). */
int m1_conv3d_pair_fwd(const m1_conv_desc_t* d, const float* w1, const float* b1, const float* w4, const float* b4, int C1,
                       void* y1, void* y4, float* stats1, float* stats4, void* ws, int ws_packed, void* stream);
int m1_conv3d_pair_dgrad(const m1_conv_desc_t* d, const float* w1, const float* w4, int C1, const void* dy1, const void* dy4,
                         void* const* dx, const int* accumulate, void* ws, int ws_packed, void* stream);
/* dw (kd,kh,kw,Cin,Cout) and db (Cout): accumulate == 0 -> overwritten (zeroed inside first);
 * accumulate != 0 -> added to what is there (the caller's flat gradient buffer, zeroed once per step: a weight
 * shared by several passes -- prior / posterior cores run twice per step -- sums without any extra copy). The
 * same flag has the same meaning on every other parameter-gradient output of this ABI. */
int m1_conv3d_wgrad(const m1_conv_desc_t* d, const void* dy, float* dw, float* db, void* ws, int accumulate,
                    void* stream);
/* test hook: 1 = route every conv through the generic direct kernels (no matrix cores). */
int m1_set_force_direct(int on);
/* Tuning switches ("M1_..." names): one table for the whole library.  A switch's value is its built-in default, or the
 * environment variable of the same name as the process started, or the last m1_config_set; m1_config_unset drops the override.
 * A change takes effect at the next launch that consults the switch (there are no per-call-site caches).  m1_config_get returns
 * M1_ERR_UNSUPPORTED for a switch nothing has consulted or set yet. */
int m1_config_set(const char* name, int value);
int m1_config_unset(const char* name);
int m1_config_get(const char* name, int* value);
/* debug probe (tools/dbg/stress_lds.py): `blocks` workgroups fill 31 KB of static LDS with a pattern and re-read it `spins` times; *bad
 * (device, unsigned) counts the words that changed.  Launched next to another kernel it shows whether that kernel writes LDS outside
 * its own allocation. */
int m1_debug_lds_canary(unsigned* bad, int blocks, int spins, void* stream);
/* debug probes (csrc/debug.hip; hip/ops.py M1_DEBUG_TRACE / M1_DEBUG_POISON=2; never launched unless a debug switch asks):
 * m1_debug_checksum: *slot (device) = a deterministic 64-bit checksum of the nbytes at p (4-byte aligned), ONE single-block kernel --
 * capturable into a hipGraph, so the outputs of every op of a replayed step can be compared between two processes.
 * m1_debug_scribble: `blocks` workgroups (0 = 512) that each take a whole CU (160 KB of LDS, four waves with 256 VGPRs + 256 AGPRs)
 * and leave the NaN pattern 0x7FC07FC0 in every LDS word and every vector register; `spins` ~microseconds each block stays resident
 * (so that all CUs are covered).  A kernel that reads LDS or a register it never wrote then yields NaN instead of a value that
 * depends on its predecessor on that CU. */
int m1_debug_checksum(const void* p, long long nbytes, unsigned long long* slot, void* stream);
/* Kernel-choice log: which kernel the dispatch put behind the conv-like launches since the log was last cleared, as a comma-separated
 * list ("conv_t3:bn160:ks2,wgrad_t3:kws16:big1,conv_mfma:128x128:w8:ks1", ...; valid until the next call).  mode 1 / 0: return the
 * log, clear it and switch logging on / off; mode < 0: return it only.  Off by default.  The op tests of the special kernels assert
 * on it: a shape the special kernel declines would otherwise compare the generic kernel with itself and stay green. */
const char* m1_debug_kernels(int mode);
int m1_debug_scribble(int blocks, int spins, void* stream);

/* ---- Conv3DTranspose(padding='same') + bias : N:496-499,505-507,513-514,520,546-553 ----
 * w: Keras layout (kd,kh,kw,Cout,Cin) fp32; y: (N, D*sd, H*sh, W*sw, Cout). */
int m1_convT3d_fwd(const m1_conv_desc_t* d, const float* w, const float* bias, void* y, void* ws, int ws_packed,
                   void* stream);
int m1_convT3d_dgrad(const m1_conv_desc_t* d, const float* w, const void* dy, void* const* dx, const int* accumulate,
                     void* ws, int ws_packed, void* stream);
int m1_convT3d_wgrad(const m1_conv_desc_t* d, const void* dy, float* dw, float* db, void* ws, int accumulate,
                     void* stream);

/* ---- tfa.layers.InstanceNormalization (eps 1e-3) [+ LeakyReLU(slope)] : B:38,40,42,44,54-60,104,128;
 *      N:473,575-576.  x,y: (N,V,C).  stats: (N,C,2) fp32 = {mean, rstd}.
 *      ws: fp32 workspace of m1_reduce_ws_floats(N,V,C,nsums) floats. ---- */
size_t m1_reduce_ws_floats(int N, long long V, int C, int nsums);
int m1_instnorm_stats(const void* x, int N, long long V, int C, int dtype, float eps, float* stats, float* ws,
                      void* stream);
int m1_instnorm_apply(const void* x, const float* stats, const float* gamma, const float* beta, float slope,
                      void* y, int N, long long V, int C, int dtype, void* stream);
/* dy = grad wrt the (activated) output; writes dx (grad wrt raw x); dgamma, dbeta (C): see `accumulate`. */
int m1_instnorm_bwd(const void* x, const float* stats, const float* gamma, const float* beta, float slope,
                    const void* dy, void* dx, float* dgamma, float* dbeta, int N, long long V, int C, int dtype,
                    float* ws, int accumulate, void* stream);

/* the same from the partial sums of m1_conv3d_dgrad_inbwd: partial [N][nparts][C][2]; sums: (N,C,2) floats of scratch */
int m1_instnorm_bwd_partials(const void* x, const float* stats, const float* gamma, const float* beta, float slope,
                             const void* dy, void* dx, float* dgamma, float* dbeta, int N, long long V, int C, int dtype,
                             const float* partial, int nparts, float* sums, int accumulate, void* stream);

/* ---- SE gate + multiplicative residual combine : B:68-78 ----
 * gate: g = sigmoid(W7 . lrelu(W6 . beta3 + b6) + b7)  (GAP(IN3(.)) == beta3 exactly).
 * W6: (F,Fr) W7: (Fr,F) Keras (1,1,1,Cin,Cout) layout.  hidden: (Fr) pre-activation, saved for bwd. */
int m1_se_gate_fwd(const float* beta3, const float* W6, const float* b6, const float* W7, const float* b7,
                   int F, int Fr, float* hidden, float* g, void* stream);
/* dg: the F + Fr float scratch written by m1_se_combine_bwd (contents are consumed and overwritten). */
int m1_se_gate_bwd(const float* beta3, const float* W6, const float* W7, const float* hidden, const float* g,
                   const float* dg, int F, int Fr, float* dbeta3_add, float* dW6, float* db6, float* dW7,
                   float* db7, int accumulate, void* stream);
#define M1_SE_GATE_BATCH 16
/* All gates of a core pass in one launch (they depend on parameters only): same arguments as m1_se_gate_fwd per job. */
typedef struct {
    const float* beta3; const float* W6; const float* b6; const float* W7; const float* b7;
    float* hidden; float* g;
    int F, Fr;
} m1_se_gate_fwd_job_t;
int m1_se_gate_fwd_batch(const m1_se_gate_fwd_job_t* jobs /* host array */, int njobs, void* stream);
/* The gate backward yields parameter gradients only (nothing on the data-gradient chain waits for it): a caller may
 * collect the jobs of a whole backward pass and run them in two launches.  Same arguments as m1_se_gate_bwd.  Jobs
 * that name the same destination buffers (one SE block evaluated by several passes of the cores) must have
 * accumulate = 1; they are applied one after the other in array order, so the sums are run-to-run identical. */
typedef struct {
    const float* beta3; const float* W6; const float* W7; const float* hidden; const float* g;
    float* dg;                 /* the F + Fr scratch of m1_se_combine_bwd (must stay alive until the batch has run) */
    float* dbeta3_add; float* dW6; float* db6; float* dW7; float* db7;
    int F, Fr, accumulate, _pad;
} m1_se_gate_job_t;
int m1_se_gate_bwd_batch(const m1_se_gate_job_t* jobs /* host array */, int njobs, void* stream);
/* out = dropout( lrelu( IN3(y3) * g * IN4(y4) ) ); y3,y4 raw conv outputs (N,V,F); stats3/4 (N,F,2).
 * Dropout state is DEVICE resident so a captured graph can be replayed: rng[0] = seed, rng[1] = step
 * counter (advanced by m1_step_advance); layer_id separates the streams of different layers. The mask is
 * a pure function of (rng, layer_id, element index): backward can regenerate it.  keep_mask (optional, bf16 and
 * F % 8 == 0 only, N*V*F/8 bytes): the forward also stores the keep bits (bit idx & 7 of byte idx >> 3) and a backward
 * given the same buffer reads them instead of re-running Philox (the backward passes are instruction bound:
 * 10 Philox rounds per 4 elements were ~45 % of their instructions).
 * stats4 == gamma4 == beta4 == NULL: the identity residual of B:63 (C_in == filters: no conv4 / norm4) -- y4 is then the block's
 * INPUT tensor (N,V,F) and out = dropout( lrelu( IN3(y3) * g * y4 ) ). */
int m1_se_combine_fwd(const void* y3, const void* y4, const float* stats3, const float* stats4,
                      const float* gamma3, const float* beta3, const float* gamma4, const float* beta4,
                      const float* g, void* out, int N, long long V, int F, int dtype, float drop_rate,
                      const uint64_t* rng, uint64_t layer_id, unsigned char* keep_mask, void* stream);
/* writes dy3, dy4 (grads wrt the RAW conv outputs, i.e. through both InstanceNorms); dgamma3,dbeta3,dgamma4,
 * dbeta4 (F each) per `accumulate`; dg is scratch for m1_se_gate_bwd, F + Fr floats, first F always overwritten.
 * ws: m1_reduce_ws_floats(N,V,F,5).  Identity residual (stats4 == gamma4 == beta4 == NULL): dy4 is the gradient wrt the block input
 * through the residual factor (no normalisation to go back through), dgamma4 / dbeta4 are not touched (may be NULL). */
int m1_se_combine_bwd(const void* y3, const void* y4, const float* stats3, const float* stats4,
                      const float* gamma3, const float* beta3, const float* gamma4, const float* beta4,
                      const float* g, const void* dout, void* dy3, void* dy4, float* dgamma3, float* dbeta3,
                      float* dgamma4, float* dbeta4, float* dg, int N, long long V, int F, int dtype,
                      float drop_rate, const uint64_t* rng, uint64_t layer_id, const unsigned char* keep_mask,
                      float* ws, int accumulate, void* stream);

/* The same for TWO stacked passes of a core that share everything in front of their first dropout draw (round 6).  A training step
 * evaluates each core twice on the same input (N:348-349 posterior, N:351-352 prior; here stacked along the batch axis): with
 * Monte-Carlo dropout the two passes differ from the first SE block's dropout on (N:579-582) -- the stem and that block's convolutions
 * and norms are the same computation twice.  y3 / y4 / stats3 / stats4 hold N samples; out (and keep_mask) hold 2N: output sample n
 * reads input sample n % N and draws its own keep mask (the dropout stream is indexed by the OUTPUT element, exactly as if the inputs
 * had been duplicated).  Backward: dout holds 2N samples; dy3 / dy4 (N samples) and the parameter sums take the SUM of the two halves'
 * gradients (lrelu' and both factors are common to the halves).  Not for the identity residual. */
int m1_se_combine_dup_fwd(const void* y3, const void* y4, const float* stats3, const float* stats4,
                          const float* gamma3, const float* beta3, const float* gamma4, const float* beta4,
                          const float* g, void* out, int N, long long V, int F, int dtype, float drop_rate,
                          const uint64_t* rng, uint64_t layer_id, unsigned char* keep_mask, void* stream);
int m1_se_combine_dup_bwd(const void* y3, const void* y4, const float* stats3, const float* stats4,
                          const float* gamma3, const float* beta3, const float* gamma4, const float* beta4,
                          const float* g, const void* dout, void* dy3, void* dy4, float* dgamma3, float* dbeta3,
                          float* dgamma4, float* dbeta4, float* dg, int N, long long V, int F, int dtype,
                          float drop_rate, const uint64_t* rng, uint64_t layer_id, const unsigned char* keep_mask,
                          float* ws, int accumulate, void* stream);

/* ---- grid attention gate pieces : B:113-124 ----
 * theta: (N, Dt,Ht,Wt, C) ; phi: (N, Dp,Hp,Wp, C) nearest-upsampled by (Dt/Dp,...) ;
 * sigma[n,v] = sigmoid( sum_c lrelu(theta+phi_up)[c]*wpsi[c] + bpsi ) : (N,Dt,Ht,Wt) stored as dtype */
int m1_gate_sigma_fwd(const void* theta, const void* phi, const float* wpsi, const float* bpsi, void* sigma,
                      int N, int Dt, int Ht, int Wt, int Dp, int Hp, int Wp, int C, int dtype, void* stream);
/* dtheta (like theta) is written; dphi (like phi) = window-sum of dtheta; dwpsi (C), dbpsi (1) overwritten.
 * ws: m1_reduce_ws_floats(N, Dt*Ht*Wt, C, 2) floats */
int m1_gate_sigma_bwd(const void* theta, const void* phi, const float* wpsi, const void* sigma,
                      const void* dsigma, void* dtheta, void* dphi, float* dwpsi, float* dbpsi, int N, int Dt,
                      int Ht, int Wt, int Dp, int Hp, int Wp, int C, int dtype, float* ws, int accumulate,
                      void* stream);
/* y = sigma_up * x : x (N,D,H,W,C), sigma (N,D/ss0,H/ss1,W/ss2) */
int m1_mul_sigma_fwd(const void* x, const void* sigma, void* y, int N, int D, int H, int W, int C, int s0,
                     int s1, int s2, int dtype, void* stream);
/* Both of the above as ONE launch (B:113-124): sigma (N,Dt,Ht,Wt) is written AND y = sigma_up * x with the stored (rounded) sigma;
 * x, y (N,D,H,W,Cx), Ci = channels of theta / phi, (Dt,Ht,Wt) = (D/s0, H/s1, W/s2).  M1_ERR_UNSUPPORTED (nothing launched) for channel
 * counts that are no multiple of a 16-byte vector or a mismatching sigma grid: take the two calls. */
int m1_gate_sigma_mul_fwd(const void* theta, const void* phi, const float* wpsi, const float* bpsi, void* sigma, const void* x,
                          void* y, int N, int Dt, int Ht, int Wt, int Dp, int Hp, int Wp, int Ci, int D, int H, int W, int Cx,
                          int s0, int s1, int s2, int dtype, void* stream);
/* accumulate_dx != 0: dx += sigma_up * dy (x also feeds other layers, see m1_conv3d_dgrad); dsigma is always overwritten */
int m1_mul_sigma_bwd(const void* x, const void* sigma, const void* dy, void* dx, void* dsigma, int N, int D,
                     int H, int W, int C, int s0, int s1, int s2, int dtype, int accumulate_dx, void* stream);

/* ---- latent head : N:640-647 (x4 levels) and KL N:373-385 ----
 * ml: (N,V,2L) = [mu | logsigma]; z = mu + exp(clip(logsigma,+-0.1))*eps  (mode 0) or mu (mode 1);
 * mode 2: N even, two passes stacked along the batch axis -- samples [0, N/2) as mode 0 with eps of N/2 samples (N:348),
 * samples [N/2, N) as mode 1 (N:349). */
int m1_latent_sample_fwd(const void* ml, const void* eps, void* z, int N, long long V, int L, int mode,
                         int dtype, void* stream);
int m1_latent_sample_bwd(const void* ml, const void* eps, const void* dz, void* dml, int N, long long V, int L,
                         int mode, int dtype, void* stream);
/* The same with the N(0,1) draws made INSIDE the kernel (no draw tensor, no generator launch in the step): element i of the sampling
 * pass takes Box-Muller of Philox4x32-10(seed = rng[0] + stream_id * golden, counter = (rng[1] << 36) + i), rng = the device-resident
 * {seed, step} pair of the dropout stream (m1_step_advance moves it).  The backward regenerates the same draws from the same state:
 * call it before the step counter advances.  stream_id: distinct per latent head. */
int m1_latent_sample_rng_fwd(const void* ml, const uint64_t* rng, uint64_t stream_id, void* z, int N, long long V, int L, int mode,
                             int dtype, void* stream);
int m1_latent_sample_rng_bwd(const void* ml, const uint64_t* rng, uint64_t stream_id, const void* dz, void* dml, int N, long long V,
                             int L, int mode, int dtype, void* stream);
/* kl[0] = mean_n sum_v KL(q||p) (fp32, overwritten). */
int m1_kl_fwd(const void* ml_q, const void* ml_p, float* kl, int N, long long V, int L, int dtype, void* stream);
/* dml_q, dml_p = dkl[0] * dKL/d(ml_*)  */
int m1_kl_bwd(const void* ml_q, const void* ml_p, const float* dkl, void* dml_q, void* dml_p, int N,
              long long V, int L, int dtype, void* stream);
/* the same when the KL term reads only the first N of Nall samples of ml_q / ml_p (the sampling half of two stacked passes, N:348,351):
 * dml_q / dml_p are (Nall, V, 2L); the gradient of the samples >= N is written as zeros by the kernel */
int m1_kl_bwd_first(const void* ml_q, const void* ml_p, const float* dkl, void* dml_q, void* dml_p, int N, long long V,
                    int L, int Nall, int dtype, void* stream);

/* ---- output heads : softmax(logits) (N:754, N:388-390) with deep-supervision heads upsampled by
 *      nearest repeat (N:739-741,751).  logits_h: (N, D/u0, H/u1, W/u2, nc) per head;
 *      probs: (N,D,H,W,nheads*nc) fp32. ---- */
typedef struct {
    const void* logits;
    void* dlogits;
    int u0, u1, u2;
    int _pad;
} m1_head_t;
int m1_softmax_heads_fwd(const m1_head_t* heads, int nheads, float* probs, int N, int D, int H, int W, int nc,
                         int dtype, void* stream);
int m1_softmax_heads_bwd(const m1_head_t* heads, int nheads, const float* probs, const float* dprobs, int N,
                         int D, int H, int W, int nc, int dtype, void* stream);

/* ---- Focal loss on the softmax heads : losses.py:32-49 (FL: renormalise, clip [1e-7, 1-1e-7], -y log p, * y (1-p)^gamma,
 * * alpha, sum over D,H,W,C, mean over the batch; loss(): mean over the y_pred.shape[-1] / nc heads) ----
 * probs (N,V,nheads*nc) fp32 as written by m1_softmax_heads_fwd; y_true (N,V,nc) fp32 or bf16; alpha: nc HOST floats.
 * fwd: ws of m1_focal_ws_floats floats (per-block partials, folded in a fixed order); loss: 1 device float.
 * bwd: dprobs = dloss[0] * dLoss/dprobs (dloss: 1 device float, e.g. autograd's incoming gradient). */
size_t m1_focal_ws_floats(int N, long long V, int nheads);
int m1_focal_fwd(const float* probs, const void* y_true, int y_dtype, const float* alpha, float gamma, int N, long long V,
                 int nheads, int nc, float* ws, float* loss, void* stream);
int m1_focal_bwd(const float* probs, const void* y_true, int y_dtype, const float* alpha, float gamma, int N, long long V,
                 int nheads, int nc, const float* dloss, float* dprobs, void* stream);

/* ---- MonteCarloDropout / Dropout : B:142-143 ; N:462-463 (Philox4x32-10, mask regenerated in bwd) ---- */
int m1_dropout(const void* x, void* y, long long n, float rate, const uint64_t* rng, uint64_t layer_id, int dtype,
               void* stream);

/* ---- dtype conversion of activations (fp32 <-> bf16) ---- */
int m1_cast(const void* x, int src_dtype, void* y, int dst_dtype, long long n, void* stream);

/* ---- Keras Adam(amsgrad=True) + L2 regulariser gradient : train_model.py:120 ; N:456-460 ----
 * p,g,m,v,vhat: flat fp32 of n elements; g += 2*lambda*p first (lambda per contiguous range:
 * [0,n_kernel) -> l2_kernel, [n_kernel,n_kernel+n_bias) -> l2_bias, rest -> 0); grad_scale multiplies g
 * (1/world_size after a sum all-reduce). lr_dev[0] = learning rate and step_dev[0] = 1-based step, both in
 * device memory (graph replay). All five buffers 16-byte aligned. */
int m1_adam_amsgrad(float* p, const float* g, float* m, float* v, float* vhat, long long n, long long n_kernel,
                    long long n_bias, float l2_kernel, float l2_bias, float grad_scale, const float* lr_dev,
                    float beta1, float beta2, float eps, const int* step_dev, void* stream);
/* step_dev[0] += 1 (may be NULL); rng_dev[1] += 1 (may be NULL). */
int m1_step_advance(int* step_dev, uint64_t* rng_dev, void* stream);

/* Deferred folds of the weight-gradient partial copies (no reference counterpart; the reference's tf.GradientTape sums weight
 * gradients inside each op).  m1_conv3d_wgrad / m1_convT3d_wgrad split the voxels over blocks, every block stores its partial
 * tile into a copy inside `ws`, and a fold kernel adds the copies into dw / db in a fixed order.  After m1_wgrad_defer(1) the
 * weight-gradient entry points QUEUE that fold instead of launching it (process-wide, host side); m1_wgrad_fold_pending runs
 * everything queued in a few batched launches on `stream` -- the caller keeps every `ws` alive and orders `stream` behind the
 * weight-gradient launches until then; dw / db are incomplete before.  m1_wgrad_fold_drop forgets the queue (gradients
 * discarded).  A second fold into the same block of dw while one is queued first runs the queue on that call's stream. */
int m1_wgrad_defer(int on);
int m1_wgrad_fold_pending(void* stream);
int m1_wgrad_fold_drop(void);

/* ---- opt-in per-kernel-family timing with hipEvents on the launch stream (bench.py roofline) ---- */
int m1_prof_enable(int on);
int m1_prof_reset(void);
/* after a stream sync: fills up to max_n records; returns count. */
typedef struct {
    char name[48];
    double total_ms;
    double flops; /* algorithmic flops summed over launches */
    double bytes; /* algorithmic bytes summed over launches */
    long long launches;
} m1_prof_rec_t;
int m1_prof_read(m1_prof_rec_t* out, int max_n);

#ifdef __cplusplus
}
#endif
#endif /* M1HIP_H */
