// dispatch.hip -- the six public conv entry points.  Each builds the logical problem (gather.h) for its role and
// runs it on the matrix-core kernels when the channel counts allow 16-byte segments, else on the generic
// direct kernels.  A concat whose members are partly unaligned (the latent z of 1..3 channels in front of the
// feature map, networks.py:652-653) is split: aligned members -> MFMA, the rest -> direct kernel accumulating
// into the same output.
#include "common.h"
#include "gather.h"
#include "reduce.h"
#include <stdio.h>
#include <stdlib.h>

static inline bool desc_ok(const m1_conv_desc_t* d) {
    if (!(d && d->N > 0 && d->D > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0 && d->kd > 0 && d->kh > 0 &&
          d->kw > 0 && d->sd > 0 && d->sh > 0 && d->sw > 0 && d->nsrc >= 1 && d->nsrc <= M1_MAX_SRC)) return false;
    if (d->dtype != M1_F32 && d->dtype != M1_BF16) return false;
    int c = 0;
    for (int i = 0; i < d->nsrc; ++i) { if (!d->src[i].ptr || d->src[i].C <= 0) return false; c += d->src[i].C; }
    return c == d->Cin;
}
static inline double esz(int dt) { return dt == M1_BF16 ? 2.0 : 4.0; }
static inline void same_pad(int in, int k, int s, int* out, int* pb) {
    int o = (in + s - 1) / s;
    int tot = (o - 1) * s + k - in; if (tot < 0) tot = 0;
    *out = o; *pb = tot / 2;
}
static inline int convT_pb(int k, int s) { return (k - s > 0 ? k - s : 0) / 2; }

struct Geo { int OD, OH, OW, pd, ph, pw; };
static inline Geo conv_geo(const m1_conv_desc_t* d) {
    Geo g; same_pad(d->D, d->kd, d->sd, &g.OD, &g.pd); same_pad(d->H, d->kh, d->sh, &g.OH, &g.ph); same_pad(d->W, d->kw, d->sw, &g.OW, &g.pw);
    return g;
}
static inline Geo convT_geo(const m1_conv_desc_t* d) {
    Geo g; g.OD = d->D * d->sd; g.OH = d->H * d->sh; g.OW = d->W * d->sw;
    g.pd = convT_pb(d->kd, d->sd); g.ph = convT_pb(d->kh, d->sh); g.pw = convT_pb(d->kw, d->sw);
    return g;
}
// algorithmic work: flops = 2*MAC; bytes = every logical input read once + output written once
static inline double conv_macs(const m1_conv_desc_t* d, bool T) {
    Geo g = T ? convT_geo(d) : conv_geo(d);
    const double taps = (double)d->kd * d->kh * d->kw;
    const double vox = T ? (double)d->D * d->H * d->W : (double)g.OD * g.OH * g.OW;
    return (double)d->N * vox * taps * d->Cin * d->Cout;
}
static inline double conv_bytes(const m1_conv_desc_t* d, bool T) {
    Geo g = T ? convT_geo(d) : conv_geo(d);
    return ((double)d->N * d->D * d->H * d->W * d->Cin + (double)d->N * g.OD * g.OH * g.OW * d->Cout) * esz(d->dtype);
}

// M1_PROF_DETAIL=1: the profiler hooks (prof.hip) key conv records by geometry, not only by entry point
struct ProfName { char s[48]; };
static ProfName prof_name(const char* fam, const m1_conv_desc_t* d) {
    int detail = M1_CFG("M1_PROF_DETAIL", 0);
    ProfName n; 
    if (!detail) { snprintf(n.s, sizeof(n.s), "%s", fam); return n; }
    snprintf(n.s, sizeof(n.s), "%s %dx%dx%d c%d>%d k%d%d%d s%d%d%d m%d", fam, d->D, d->H, d->W, d->Cin, d->Cout, d->kd, d->kh, d->kw,
             d->sd, d->sh, d->sw, d->nsrc);
    return n;
}

extern "C" const char* m1_status_name(int s) {
    switch (s) {
        case M1_OK: return "M1_OK";
        case M1_ERR_BAD_ARG: return "M1_ERR_BAD_ARG";
        case M1_ERR_UNSUPPORTED: return "M1_ERR_UNSUPPORTED";
        case M1_ERR_LAUNCH: return "M1_ERR_LAUNCH";
        case M1_ERR_WORKSPACE: return "M1_ERR_WORKSPACE";
        default: return "M1_ERR_UNKNOWN";
    }
}
extern "C" int m1_abi_version(void) { return 1; }

static int g_force_direct = 0;
extern "C" int m1_set_force_direct(int on) { g_force_direct = on; return M1_OK; }

// ---- spec builders -------------------------------------------------------------------------------------------
// forward-type problems: the concat `d->src` is the CONTRACTION axis, `out` has all Cout channels
static GatherSpec fwd_spec(const m1_conv_desc_t* d, bool T, const float* w, const float* bias, void* y) {
    GatherSpec g{}; Geo q = T ? convT_geo(d) : conv_geo(d);
    g.nsrc = d->nsrc;
    for (int i = 0; i < d->nsrc; ++i) { g.src[i] = d->src[i].ptr; g.srcC[i] = d->src[i].C; }
    g.ID = d->D; g.IH = d->H; g.IW = d->W; g.out = y; g.OC = d->Cout; g.OD = q.OD; g.OH = q.OH; g.OW = q.OW; g.N = d->N;
    g.w = w; g.bias = bias; g.kd = d->kd; g.kh = d->kh; g.kw = d->kw; g.sd = d->sd; g.sh = d->sh; g.sw = d->sw;
    g.pd = q.pd; g.ph = q.ph; g.pw = q.pw; g.dtype = d->dtype;
    if (!T) { g.mode = 0; g.wST = (long long)d->Cin * d->Cout; g.wSC = d->Cout; g.wSO = 1; }     // w[tap][ci][co]
    else    { g.mode = 1; g.wST = (long long)d->Cout * d->Cin; g.wSC = 1; g.wSO = d->Cin; }      // w[tap][co][ci]
    return g;
}
// data-gradient problems: dy (Cout channels) is the contraction axis, one spec per concat member
static GatherSpec dgrad_spec(const m1_conv_desc_t* d, bool T, const float* w, const void* dy, void* dx, int member, int ch_off) {
    GatherSpec g{}; Geo q = T ? convT_geo(d) : conv_geo(d);
    g.nsrc = 1; g.src[0] = dy; g.srcC[0] = d->Cout;
    g.ID = q.OD; g.IH = q.OH; g.IW = q.OW; g.out = dx; g.OC = d->src[member].C; g.OD = d->D; g.OH = d->H; g.OW = d->W; g.N = d->N;
    g.w = w; g.bias = nullptr; g.oc_off = ch_off; g.kd = d->kd; g.kh = d->kh; g.kw = d->kw; g.sd = d->sd; g.sh = d->sh; g.sw = d->sw;
    g.pd = q.pd; g.ph = q.ph; g.pw = q.pw; g.dtype = d->dtype;
    if (!T) { g.mode = 1; g.wST = (long long)d->Cin * d->Cout; g.wSC = 1; g.wSO = d->Cout; }     // out = ci, contraction = co
    else    { g.mode = 0; g.wST = (long long)d->Cout * d->Cin; g.wSC = d->Cin; g.wSO = 1; }
    return g;
}

// the data gradient of ALL concat members as one problem: Cin output columns spread over the members' gradient tensors
// (GatherSpec::outs).  One launch instead of one per member: dY is gathered once per tile instead of once per member, 32-channel
// members stop running on 32-column tiles (the loader-bound shape), and a 5-member dense-skip concat costs 1 launch, not 5.
static bool dgrad_fused_ok(const m1_conv_desc_t* d) {
    int en = M1_CFG("M1_DGRAD_FUSED", 1);
    if (!en || d->nsrc < 2) return false;
    const int seg = d->dtype == M1_BF16 ? 8 : 4;
    for (int i = 0; i < d->nsrc; ++i) if (d->src[i].C % seg) return false;
    // the wide, shallow layers (res0/res1: <= 64 dY channels, stride 1, >= 32,768 voxels) run on the halo-tile kernel, whose blocks
    // own one 32-column weight slice each: per-member launches are faster there (192->32 at res0: 0.76 vs 1.13 ms per step fused)
    const long long vox = (long long)d->N * d->D * d->H * d->W;
    int hf = M1_CFG("M1_DGRAD_FUSED_HALO", 0);
    if (!hf && d->dtype == M1_BF16 && d->Cout <= 64 && d->sd == 1 && d->sh == 1 && d->sw == 1 && vox >= 32768 && d->kd * d->kh * d->kw > 1) return false;
    return true;
}
static GatherSpec dgrad_fused_spec(const m1_conv_desc_t* d, bool T, const float* w, const void* dy, void* const* dx, const int* accumulate) {
    GatherSpec g = dgrad_spec(d, T, w, dy, nullptr, 0, 0);
    g.OC = d->Cin; g.oc_off = 0; g.out = nullptr; g.accumulate = 0;
    g.nout = d->nsrc;
    for (int i = 0; i < d->nsrc; ++i) {
        g.outs[i] = dx ? dx[i] : nullptr; g.outC[i] = d->src[i].C; g.outAcc[i] = (accumulate && accumulate[i]) ? 1 : 0;
    }
    return g;
}

static inline size_t align256(size_t v) { return (v + 255) / 256 * 256; }

// Every channel count runs on the matrix-core kernels (unaligned members are zero-padded to whole 16-byte K
// segments inside the kernel); the generic direct kernels remain as the m1_set_force_direct reference path.
static void split_members(const GatherSpec& g, GatherSpec* mf, GatherSpec* dr) {
    *mf = g; *dr = g; mf->nsrc = 0; dr->nsrc = 0;
    if (g_force_direct || !m1_mfma_supported(g)) { *dr = g; return; }
    *mf = g;
}

static size_t gather_ws_bytes(const GatherSpec& g) {
    GatherSpec mf, dr; split_members(g, &mf, &dr);
    return mf.nsrc ? align256(m1_mfma_ws_bytes(mf)) : 0;
}

static int run_gather(const GatherSpec& g, void* ws, int ws_packed, hipStream_t st) {
    GatherSpec mf, dr; split_members(g, &mf, &dr);
    int rc = M1_OK;
    if (mf.nsrc) {
        if (!ws) return M1_ERR_WORKSPACE;
        rc = m1_mfma_gather(mf, ws, ws_packed, st); if (rc) return rc;
        if (dr.nsrc) { dr.accumulate = 1; dr.bias = nullptr; rc = m1_direct_gather(dr, st); }
        return rc;
    }
    return m1_direct_gather(dr, st);
}

// ---- forward of a wide concat as one halo-tile launch per member group ----------------------------------------------
// The halo-tile kernel (conv_halo.hip) stages an input tile once for all taps but holds at most 64 contraction channels; the
// dense-skip concats of the full model (160 channels at res0, 256 at res1: networks.py:604-623) fell back to the per-tap
// gather (160->32 at res0: 470 us, bound by 9x re-reads from L2).  Split the members into groups of 8/16/32/64 channels: the
// first group writes y (+ bias), the others add into it in their epilogue, the last one also emits the InstanceNorm statistics
// of the sum.  Each group is an ordinary conv over its members with a channel offset into the weights (own panel, own job).
struct FwdGroups { int n; int first[M1_MAX_SRC], count[M1_MAX_SRC], coff[M1_MAX_SRC]; bool T; };
// A concat of wide members plus a tiny one that is not a whole 16-byte segment per voxel (the latent z of 1..3 channels in front of
// the feature map: networks.py:652-653, dec_hi = Conv3DTranspose([z, f])): as ONE problem the odd member takes the whole contraction
// off the LDS-DMA loader (515 -> 256 at res4: 144 us against 69 us for 512 -> 256).  Split: the run of 64-byte-aligned members is one
// conv that writes y (+ bias) on the fast path, the odd members follow as tiny convs that add into y.  Forward and transposed forward.
static bool odd_split(const m1_conv_desc_t* d, bool T, FwdGroups* fg) {
    int en = M1_CFG("M1_ODD_SPLIT", 1);
    if (!en || g_force_direct || d->nsrc < 2) return false;
    const int seg = d->dtype == M1_BF16 ? 8 : 4, chunk = 4 * seg;
    int a0 = -1, a1 = -1, nodd = 0, call = 0;
    for (int i = 0; i < d->nsrc; ++i) {
        const int c = d->src[i].C;
        if (c % chunk == 0) { if (a0 < 0) a0 = i; a1 = i; call += c; }
        else if (c < seg) ++nodd;
        else return false;
    }
    if (!nodd || a0 < 0 || call < 64) return false;
    for (int i = a0; i <= a1; ++i) if (d->src[i].C % chunk) return false;        // the aligned members must be one run
    fg->n = 0; fg->T = T;
    int off = 0, offs[M1_MAX_SRC];
    for (int i = 0; i < d->nsrc; ++i) { offs[i] = off; off += d->src[i].C; }
    fg->first[0] = a0; fg->count[0] = a1 - a0 + 1; fg->coff[0] = offs[a0]; fg->n = 1;
    if (a0 > 0) { fg->first[fg->n] = 0; fg->count[fg->n] = a0; fg->coff[fg->n] = 0; fg->n++; }
    if (a1 + 1 < d->nsrc) { fg->first[fg->n] = a1 + 1; fg->count[fg->n] = d->nsrc - a1 - 1; fg->coff[fg->n] = offs[a1 + 1]; fg->n++; }
    return true;
}
static bool fwd_groups(const m1_conv_desc_t* d, bool T, FwdGroups* fg) {
    if (odd_split(d, T, fg)) return true;
    fg->T = T;
    int en = M1_CFG("M1_HALO_GROUPS", 1);
    if (!en || T || g_force_direct || d->dtype != M1_BF16 || d->nsrc < 2 || d->Cin <= 64) return false;
    if (d->kd * d->kh * d->kw < 2) return false;
    Geo q = conv_geo(d);
    if (q.OW % 8 || (long long)d->N * q.OD * q.OH * q.OW < 32768) return false;
    auto ok = [](int c) { return c == 8 || c == 16 || c == 32 || c == 64; };
    fg->n = 0; int cur = 0, off = 0;
    for (int i = 0; i < d->nsrc; ++i) {
        const int c = d->src[i].C;
        if (!ok(c)) return false;
        if (fg->n > 0 && ok(cur + c) && cur + c <= 64) { fg->count[fg->n - 1]++; cur += c; }
        else { fg->first[fg->n] = i; fg->count[fg->n] = 1; fg->coff[fg->n] = off; fg->n++; cur = c; }
        off += c;
    }
    return fg->n >= 2;
}
static GatherSpec fwd_group_spec(const m1_conv_desc_t* d, const FwdGroups& fg, int gi, const float* w, const float* bias, void* y) {
    GatherSpec g = fwd_spec(d, fg.T, w, gi == 0 ? bias : nullptr, y);
    g.nsrc = fg.count[gi];
    for (int i = 0; i < g.nsrc; ++i) { g.src[i] = d->src[fg.first[gi] + i].ptr; g.srcC[i] = d->src[fg.first[gi] + i].C; }
    g.cc_off = fg.coff[gi];
    g.accumulate = gi > 0 ? 1 : 0;
    return g;
}

// ---- stem weight gradient (Cin < 8: the image channels, networks.py:472) -----------------------------------------------
// 3 (or 2) input channels are 6 (4) bytes per voxel: no 16-byte segments for the LDS-DMA loaders, so this layer used to run
// on the generic per-tap kernel, re-reading dY once per tap (157 us at batch 2 = the slowest weight gradient of the step for
// 0.2 % of its flops).  Instead: X -> X8 (zero-padded to 8 channels, one 16-byte segment per voxel) in the workspace, the
// tap-fused kernel on (X8, dY) into a scratch gradient with 8 input rows per tap, and a fold of its first Cin rows.
static bool stem_wanted(const m1_conv_desc_t* d, bool T) {
    int en = M1_CFG("M1_STEM_TF", 1);
    // (fp32: the same with 4 channels = one 16-byte segment per voxel, on the fp32 tap-fused kernel)
    return en && !T && !g_force_direct && d->nsrc == 1 && d->src[0].C < (d->dtype == M1_BF16 ? 8 : 4) && d->Cin == d->src[0].C;
}
static size_t stem_ws_bytes(const m1_conv_desc_t* d, bool T) {
    if (!stem_wanted(d, T)) return 0;
    return align256((size_t)d->N * d->D * d->H * d->W * 8 * 2) + align256((size_t)d->kd * d->kh * d->kw * 8 * d->Cout * sizeof(float));
}
// Zero fill as a KERNEL.  hipMemsetAsync on a capturing stream becomes a memset node of the step's hipGraph, and on this ROCm release
// a replayed graph does not keep a memset node ordered between the kernel nodes around it (round 5: from the third replay on, the
// padded-stem weight gradient -- zero r8, accumulate into it, read it -- differed from run to run, in order on ONE stream; eager
// launches never; tools/probes/graph_memset_probe.py).  M1_MEMSET_KERNEL=0 restores the memset (the reproducer of that finding).
__global__ void __launch_bounds__(256) m1_zero_kernel(unsigned* __restrict__ head, int nhead, uint4* __restrict__ p, long long n16,
                                                      unsigned* __restrict__ tail, int ntail) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256) p[i] = make_uint4(0, 0, 0, 0);
    if (blockIdx.x == 0 && (int)threadIdx.x < nhead) head[threadIdx.x] = 0u;
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0u;
}
// Any 4-byte aligned buffer of a multiple of 4 bytes (dw / db at an odd float offset of a caller's flat gradient buffer): up to three
// dwords in front of the first 16-byte boundary, the 16-byte body, up to three dwords behind it -- always a kernel, never a memset
// node (round-5 advisor: the old fallback to hipMemsetAsync for unaligned pointers brought the replay defect back without an error).
// M1_MEMSET_KERNEL=0 restores the memset = the reproducer of that defect.
static int m1_zero_async(void* p, size_t bytes, hipStream_t st) {
    if (!bytes) return M1_OK;
    if (!M1_CFG("M1_MEMSET_KERNEL", 1)) return hipMemsetAsync(p, 0, bytes, st) == hipSuccess ? M1_OK : M1_ERR_LAUNCH;
    if (((uintptr_t)p & 3) || (bytes & 3)) return M1_ERR_BAD_ARG;
    size_t ndw = bytes >> 2;
    int nhead = (int)(((16 - ((uintptr_t)p & 15)) & 15) >> 2);
    if ((size_t)nhead > ndw) nhead = (int)ndw;
    unsigned* head = (unsigned*)p;
    unsigned char* body = (unsigned char*)p + 4 * (size_t)nhead;
    const long long n16 = (long long)((ndw - nhead) >> 2);
    const int ntail = (int)((ndw - nhead) & 3);
    long long g = (n16 + 255) / 256; if (g > 2048) g = 2048; if (g < 1) g = 1;
    hipLaunchKernelGGL(m1_zero_kernel, dim3((unsigned)g), dim3(256), 0, st, head, nhead, (uint4*)body, n16, (unsigned*)(body + (n16 << 4)), ntail);
    return m1_check_launch();
}
__global__ void __launch_bounds__(256) stem_pad8_kernel(const unsigned short* __restrict__ x, uint4* __restrict__ x8, long long nvox, int C) {
    for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < nvox; v += (long long)gridDim.x * 256) {
        unsigned short e[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int c = 0; c < C; ++c) e[c] = x[v * C + c];
        uint4 o;
        o.x = e[0] | ((unsigned)e[1] << 16); o.y = e[2] | ((unsigned)e[3] << 16);
        o.z = e[4] | ((unsigned)e[5] << 16); o.w = e[6] | ((unsigned)e[7] << 16);
        x8[v] = o;
    }
}
__global__ void __launch_bounds__(256) stem_pad4f_kernel(const float* __restrict__ x, float4* __restrict__ x4, long long nvox, int C) {
    for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < nvox; v += (long long)gridDim.x * 256) {
        float e[4] = {0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < C; ++c) e[c] = x[v * C + c];
        x4[v] = make_float4(e[0], e[1], e[2], e[3]);
    }
}
// dw[t][ci][co] += r8[t][ci][co], ci < Cin   (dw was zeroed above when accumulate == 0); CP = padded channels of r8 (8 / 4)
__global__ void __launch_bounds__(256) stem_fold_kernel(const float* __restrict__ r8, float* __restrict__ dw, int taps, int Cin, int Cout, int CP) {
    const int n = taps * Cin * Cout;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int co = i % Cout, ci = (i / Cout) % Cin, t = i / (Cout * Cin);
        dw[i] += r8[((size_t)t * CP + ci) * Cout + co];
    }
}

// Scratch for the per-split partial copies of a weight gradient (+ bias sums): compact copies of ONE concat member's block
// ([tap][C_member][Cout] + Cout floats), up to 512 of them, at most 64 MB.  It lives in the caller's wgrad workspace -- one region
// per call, so weight gradients that run concurrently on different streams never share it (the library keeps no device memory
// of its own).  All weight-gradient kernels fold copies in a fixed order instead of using floating-point atomics.
static size_t wgrad_rx_bytes(const m1_conv_desc_t* d) {
    const size_t taps = (size_t)d->kd * d->kh * d->kw;
    size_t cmax = 8;                                           // (the padded stem runs as an 8-channel member)
    for (int i = 0; i < d->nsrc; ++i) if ((size_t)d->src[i].C > cmax) cmax = d->src[i].C;
    const size_t other = cmax > (size_t)d->Cout ? cmax : (size_t)d->Cout;      // (transposed conv: the roles of the two sides swap)
    const size_t stride = taps * cmax * d->Cout + other;
    size_t b = 512 * stride * sizeof(float);
    const long long cap = (long long)M1_CFG("M1_WG_RX_MB", 64) << 20;
    if (b > (size_t)cap) b = (size_t)cap;
    if (b < 2 * stride * sizeof(float)) b = 2 * stride * sizeof(float);
    return align256(b);
}
// every concat member has a region of its own (wgrad_rx_bytes each): with deferred folds (m1_wgrad_defer) the copies of
// member i must survive the kernels of member i+1
static size_t wgrad_rx_total(const m1_conv_desc_t* d) { return wgrad_rx_bytes(d) * (size_t)(d->nsrc > 0 ? d->nsrc : 1); }

// ---- workspace query ---------------------------------------------------------------------------------------------
extern "C" size_t m1_conv_ws_bytes(const m1_conv_desc_t* d, int transposed, int role) {
    if (!desc_ok(d)) return 0;
    const bool T = transposed != 0;
    if (role == 0) {
        Geo q0 = T ? convT_geo(d) : conv_geo(d);
        size_t panels = 0;
        FwdGroups fg;
        if (fwd_groups(d, T, &fg)) { for (int gi = 0; gi < fg.n; ++gi) panels += gather_ws_bytes(fwd_group_spec(d, fg, gi, nullptr, nullptr, nullptr)); }
        else panels = gather_ws_bytes(fwd_spec(d, T, nullptr, nullptr, nullptr));
        return panels + align256(m1_stats_ws_floats(d->N, (long long)q0.OD * q0.OH * q0.OW, d->Cout) * sizeof(float)) + 256;
    }
    if (role == 1) {
        if (dgrad_fused_ok(d) && !g_force_direct && m1_mfma_supported(dgrad_fused_spec(d, T, nullptr, nullptr, nullptr, nullptr)))
            return gather_ws_bytes(dgrad_fused_spec(d, T, nullptr, nullptr, nullptr, nullptr)) + 256;
        size_t m = 0; int off = 0;
        for (int i = 0; i < d->nsrc; ++i) {
            m += gather_ws_bytes(dgrad_spec(d, T, nullptr, nullptr, nullptr, i, off));
            off += d->src[i].C;
        }
        return m + 256;
    }
    Geo q = T ? convT_geo(d) : conv_geo(d);
    return align256(m1_reduce_ws_floats(d->N, (long long)q.OD * q.OH * q.OW, d->Cout, 1) * sizeof(float)) + 256 + stem_ws_bytes(d, T) +
           wgrad_rx_total(d);
}

// ---- packed-weight panel records ----------------------------------------------------------------------------------
// Device addresses of the pack-job records inside `ws` (one per matrix-core panel: role 0 -> 1, role 1 -> one per
// concat member).  A record is filled by the first (lazy) pack of its panel; m1_pack_batch re-runs filled records.
extern "C" int m1_conv_pack_jobs(const m1_conv_desc_t* d, int transposed, int role, void* ws, void** jobs_out) {
    if (!desc_ok(d) || !ws || !jobs_out) return 0;
    const bool T = transposed != 0;
    int n = 0;
    if (role == 0) {
        FwdGroups fg;
        if (fwd_groups(d, T, &fg)) {
            size_t woff = 0;
            for (int gi = 0; gi < fg.n; ++gi) {
                const size_t b = gather_ws_bytes(fwd_group_spec(d, fg, gi, nullptr, nullptr, nullptr));
                if (b) jobs_out[n++] = (unsigned char*)ws + woff;
                woff += b;
            }
        } else if (gather_ws_bytes(fwd_spec(d, T, nullptr, nullptr, nullptr))) jobs_out[n++] = ws;
    } else if (role == 1 && dgrad_fused_ok(d) && !g_force_direct && m1_mfma_supported(dgrad_fused_spec(d, T, nullptr, nullptr, nullptr, nullptr))) {
        if (gather_ws_bytes(dgrad_fused_spec(d, T, nullptr, nullptr, nullptr, nullptr))) jobs_out[n++] = ws;
    } else if (role == 1) {
        int off = 0; size_t woff = 0;
        for (int i = 0; i < d->nsrc; ++i) {
            const size_t b = gather_ws_bytes(dgrad_spec(d, T, nullptr, nullptr, nullptr, i, off));
            if (b) jobs_out[n++] = (unsigned char*)ws + woff;
            woff += b; off += d->src[i].C;
        }
    }
    return n;
}
extern "C" int m1_pack_batch(const void* const* jobs_dev, const int* block_prefix_dev, int njobs, int total_blocks, void* stream) {
    if (m1_debug_skip("pack")) return M1_OK;
    if (njobs < 0 || (njobs > 0 && !jobs_dev) || (block_prefix_dev && total_blocks < njobs)) return M1_ERR_BAD_ARG;
    M1ProfScope ps("pack_batch", 0.0, 0.0, (hipStream_t)stream);
    return m1_pack_batch_internal(jobs_dev, block_prefix_dev, njobs, total_blocks, (hipStream_t)stream);
}

// ---- Conv3D ------------------------------------------------------------------------------------------------------
extern "C" int m1_conv3d_fwd(const m1_conv_desc_t* d, const float* w, const float* bias, void* y, float* stats, void* ws,
                             int ws_packed, void* stream) {
    if (m1_debug_skip("conv_fwd")) return M1_OK;
    if (!desc_ok(d) || !w || !y) return M1_ERR_BAD_ARG;
    M1ProfScope ps(prof_name("conv3d_fwd", d).s, 2.0 * conv_macs(d, false), conv_bytes(d, false), (hipStream_t)stream);
    FwdGroups fg;
    if (ws && fwd_groups(d, false, &fg)) {
        size_t woff = 0, total = 0;
        for (int gi = 0; gi < fg.n; ++gi) total += gather_ws_bytes(fwd_group_spec(d, fg, gi, nullptr, nullptr, nullptr));
        for (int gi = 0; gi < fg.n; ++gi) {
            GatherSpec gg = fwd_group_spec(d, fg, gi, w, bias, y);
            if (stats && gi == fg.n - 1) {
                gg.stats_out = stats; gg.stats_eps = 1e-3f;
                gg.stats_ws = reinterpret_cast<float*>((unsigned char*)ws + total);
            }
            int rc = run_gather(gg, (unsigned char*)ws + woff, ws_packed, (hipStream_t)stream); if (rc) return rc;
            woff += gather_ws_bytes(gg);
        }
        return M1_OK;
    }
    GatherSpec g = fwd_spec(d, false, w, bias, y);
    if (stats) {
        if (!ws) return M1_ERR_WORKSPACE;
        g.stats_out = stats; g.stats_eps = 1e-3f;        // tfa InstanceNormalization epsilon
        g.stats_ws = reinterpret_cast<float*>((unsigned char*)ws + gather_ws_bytes(g));
        GatherSpec mf, dr; split_members(g, &mf, &dr);
        if (!mf.nsrc) {                                   // direct path: conv, then the stand-alone reduction
            int rc = m1_direct_gather(dr, (hipStream_t)stream); if (rc) return rc;
            Geo q = conv_geo(d);
            return m1_stats_internal(y, d->N, (long long)q.OD * q.OH * q.OW, d->Cout, d->dtype, 1e-3f, stats, g.stats_ws, (hipStream_t)stream);
        }
    }
    return run_gather(g, ws, ws_packed, (hipStream_t)stream);
}
// ---- conv1 || conv4 of an SE block (network_blocks.py:53,64: same input, same kernel size and strides) as ONE problem --------
// Forward: one conv with C1 + C4 output columns, written to two tensors ([y1 | y4]) with their own bias vectors and InstanceNorm
// statistics: the im2col operand is gathered once, and the C1 = F/4 columns that alone would run on a loader-bound 32-column
// tile ride on the 128-column tiles of their big twin.  Data gradient: one contraction over the virtual concat [dy1 | dy4] with a
// panel packed from both weight tensors (no read-modify-write of dx by a second launch).  d->Cout = C1 + C4.
static bool pair_ok(const m1_conv_desc_t* d, int C1) {
    const int seg = d->dtype == M1_BF16 ? 8 : 4;
    if (!desc_ok(d) || C1 <= 0 || C1 >= d->Cout || C1 % seg || (d->Cout - C1) % seg || g_force_direct) return false;
    for (int i = 0; i < d->nsrc; ++i) if (d->src[i].C % seg) return false;
    return true;
}
extern "C" int m1_conv3d_pair_fwd(const m1_conv_desc_t* d, const float* w1, const float* b1, const float* w4, const float* b4, int C1,
                                  void* y1, void* y4, float* stats1, float* stats4, void* ws, int ws_packed, void* stream) {
    if (m1_debug_skip("conv_fwd")) return M1_OK;
    if (!d || !w1 || !w4 || !y1 || !y4 || !ws || (stats1 == nullptr) != (stats4 == nullptr)) return M1_ERR_BAD_ARG;
    FwdGroups fg;
    if (!pair_ok(d, C1) || fwd_groups(d, false, &fg)) return M1_ERR_UNSUPPORTED;
    M1ProfScope ps(prof_name("conv3d_fwd", d).s, 2.0 * conv_macs(d, false), conv_bytes(d, false), (hipStream_t)stream);
    const int C4 = d->Cout - C1;
    GatherSpec g = fwd_spec(d, false, w1, b1, nullptr);
    g.wST = (long long)d->Cin * C1; g.wSC = C1; g.wSO = 1;
    g.w2 = w4; g.w2ST = (long long)d->Cin * C4; g.w2SC = C4; g.w2SO = 1; g.oc_split = C1; g.bias2 = b4;
    g.nout = 2; g.outs[0] = y1; g.outC[0] = C1; g.outAcc[0] = 0; g.outs[1] = y4; g.outC[1] = C4; g.outAcc[1] = 0;
    if (!m1_mfma_supported(g)) return M1_ERR_UNSUPPORTED;
    if (stats1) {
        g.stats_out = stats1; g.stats_out2 = stats4; g.stats_eps = 1e-3f;
        g.stats_ws = reinterpret_cast<float*>((unsigned char*)ws + gather_ws_bytes(g));
    }
    return m1_mfma_gather(g, ws, ws_packed, (hipStream_t)stream);
}
extern "C" int m1_conv3d_pair_dgrad(const m1_conv_desc_t* d, const float* w1, const float* w4, int C1, const void* dy1, const void* dy4,
                                    void* const* dx, const int* accumulate, void* ws, int ws_packed, void* stream) {
    if (m1_debug_skip("conv_dgrad")) return M1_OK;
    if (!d || !w1 || !w4 || !dy1 || !dy4 || !dx || !ws) return M1_ERR_BAD_ARG;
    if (!pair_ok(d, C1)) return M1_ERR_UNSUPPORTED;
    M1ProfScope ps(prof_name("conv3d_dgrad", d).s, 2.0 * conv_macs(d, false), conv_bytes(d, false), (hipStream_t)stream);
    const int C4 = d->Cout - C1;
    GatherSpec g;
    if (d->nsrc >= 2) {
        if (!dgrad_fused_ok(d)) return M1_ERR_UNSUPPORTED;
        g = dgrad_fused_spec(d, false, w1, dy1, dx, accumulate);
    } else {
        g = dgrad_spec(d, false, w1, dy1, dx[0], 0, 0);
        g.accumulate = accumulate && accumulate[0] ? 1 : 0;
        if (!dx[0]) return M1_OK;
    }
    g.nsrc = 2; g.src[0] = dy1; g.srcC[0] = C1; g.src[1] = dy4; g.srcC[1] = C4;
    g.wST = (long long)d->Cin * C1; g.wSC = 1; g.wSO = C1;
    g.w2 = w4; g.w2ST = (long long)d->Cin * C4; g.w2SC = 1; g.w2SO = C4; g.c_split = C1;
    if (!m1_mfma_supported(g)) return M1_ERR_UNSUPPORTED;
    bool any = false;
    for (int i = 0; i < d->nsrc; ++i) any |= dx[i] != nullptr;
    return any ? m1_mfma_gather(g, ws, ws_packed, (hipStream_t)stream) : M1_OK;
}

// 1 when BOTH m1_conv3d_pair_fwd and m1_conv3d_pair_dgrad take this shape (neither would return M1_ERR_UNSUPPORTED): the caller
// decides between the pair and the two single convs BEFORE it builds its autograd graph (a refusal in the middle of a backward pass
// has no fallback).  Mirrors every gate of the two entry points.
extern "C" int m1_conv3d_pair_supported(const m1_conv_desc_t* d, int C1) {
    if (!d || g_force_direct) return 0;
    FwdGroups fg;
    if (!pair_ok(d, C1) || fwd_groups(d, false, &fg)) return 0;
    const int C4 = d->Cout - C1;
    GatherSpec g = fwd_spec(d, false, nullptr, nullptr, nullptr);
    g.oc_split = C1; g.nout = 2; g.outC[0] = C1; g.outC[1] = C4;
    if (!m1_mfma_supported(g)) return 0;
    GatherSpec q;
    if (d->nsrc >= 2) {
        if (!dgrad_fused_ok(d)) return 0;
        q = dgrad_fused_spec(d, false, nullptr, nullptr, nullptr, nullptr);
    } else q = dgrad_spec(d, false, nullptr, nullptr, nullptr, 0, 0);
    q.nsrc = 2; q.srcC[0] = C1; q.srcC[1] = C4; q.c_split = C1;
    return m1_mfma_supported(q) ? 1 : 0;
}

extern "C" int m1_convT3d_fwd(const m1_conv_desc_t* d, const float* w, const float* bias, void* y, void* ws, int ws_packed,
                              void* stream) {
    if (m1_debug_skip("conv_fwd")) return M1_OK;
    if (!desc_ok(d) || !w || !y) return M1_ERR_BAD_ARG;
    M1ProfScope ps(prof_name("convT3d_fwd", d).s, 2.0 * conv_macs(d, true), conv_bytes(d, true), (hipStream_t)stream);
    FwdGroups fg;
    if (ws && fwd_groups(d, true, &fg)) {                 // (odd_split: the aligned run first, the tiny members add into y)
        size_t woff = 0;
        for (int gi = 0; gi < fg.n; ++gi) {
            GatherSpec gg = fwd_group_spec(d, fg, gi, w, bias, y);
            int rc = run_gather(gg, (unsigned char*)ws + woff, ws_packed, (hipStream_t)stream); if (rc) return rc;
            woff += gather_ws_bytes(gg);
        }
        return M1_OK;
    }
    return run_gather(fwd_spec(d, true, w, bias, y), ws, ws_packed, (hipStream_t)stream);
}
static int dgrad_common(const m1_conv_desc_t* d, bool T, const float* w, const void* dy, void* const* dx, const int* accumulate,
                        void* ws, int ws_packed, hipStream_t st) {
    if (dgrad_fused_ok(d) && !g_force_direct) {
        GatherSpec gf = dgrad_fused_spec(d, T, w, dy, dx, accumulate);
        if (m1_mfma_supported(gf)) {
            bool any = false;
            for (int i = 0; i < d->nsrc; ++i) any |= dx[i] != nullptr;
            return any ? run_gather(gf, ws, ws_packed, st) : M1_OK;
        }
    }
    int off = 0; size_t woff = 0;                         // member i's panel lives at its own offset of ws (cacheable)
    for (int i = 0; i < d->nsrc; ++i) {
        GatherSpec g = dgrad_spec(d, T, w, dy, dx[i], i, off);
        g.accumulate = accumulate && accumulate[i] ? 1 : 0;
        if (dx[i]) {
            int rc = run_gather(g, ws ? (unsigned char*)ws + woff : nullptr, ws_packed, st); if (rc) return rc;
        }
        woff += gather_ws_bytes(g);
        off += d->src[i].C;
    }
    return M1_OK;
}
extern "C" int m1_conv3d_dgrad(const m1_conv_desc_t* d, const float* w, const void* dy, void* const* dx, const int* accumulate,
                               void* ws, int ws_packed, void* stream) {
    if (m1_debug_skip("conv_dgrad")) return M1_OK;
    if (!desc_ok(d) || !w || !dy || !dx) return M1_ERR_BAD_ARG;
    M1ProfScope ps(prof_name("conv3d_dgrad", d).s, 2.0 * conv_macs(d, false), conv_bytes(d, false), (hipStream_t)stream);
    return dgrad_common(d, false, w, dy, dx, accumulate, ws, ws_packed, (hipStream_t)stream);
}
// Data gradient of a single-input Conv3D whose input is a = lrelu(IN(x)) (conv2 / conv3 of an SE block, network_blocks.py:54-59):
// the kernel that writes d(a) also emits the per-tile sums the InstanceNorm backward needs ({sum dy, sum dy*xh} per sample and
// channel) into `partial` [N][*nparts][Cin][2].  *nparts = 0: the kernel that took the shape has no such epilogue -- d(a) is
// complete, the caller runs m1_instnorm_bwd instead of m1_instnorm_bwd_partials.
// rows per sample the partial buffer of m1_conv3d_dgrad_inbwd must hold (`partial` = N * rows * Cin * 2 floats): the most any kernel
// behind it writes -- one row per epilogue tile (>= 64 voxels each) or one per chunk of the split-K finish reduction
extern "C" int m1_conv3d_dgrad_inbwd_rows(const m1_conv_desc_t* d) {
    if (!desc_ok(d)) return 0;
    const long long V = (long long)d->D * d->H * d->W;
    const long long tiles = m1_stats_rows_cap(V), chunks = m1_red_nchunks(V, d->Cin, d->N);     // (>= one row per 64 voxels)
    const long long r = tiles > chunks ? tiles : chunks;
    return r > 0x3fffffff ? 0x3fffffff : (int)r;
}
extern "C" int m1_conv3d_dgrad_inbwd(const m1_conv_desc_t* d, const float* w, const void* dy, void* da, const void* x, const float* stats,
                                     const float* gamma, const float* beta, float slope, float* partial, int partial_rows, int* nparts,
                                     void* ws, int ws_packed, void* stream) {
    if (!desc_ok(d) || !w || !dy || !da || !x || !stats || !gamma || !beta || !partial || !nparts || d->nsrc != 1 || partial_rows < 0) return M1_ERR_BAD_ARG;
    M1ProfScope ps(prof_name("conv3d_dgrad", d).s, 2.0 * conv_macs(d, false), conv_bytes(d, false), (hipStream_t)stream);
    *nparts = 0;
    if (g_force_direct) { void* dxs[1] = {da}; int acc0[1] = {0}; return dgrad_common(d, false, w, dy, dxs, acc0, ws, ws_packed, (hipStream_t)stream); }
    GatherSpec g = dgrad_spec(d, false, w, dy, da, 0, 0);
    g.ib_x = x; g.ib_stats = stats; g.ib_gamma = gamma; g.ib_beta = beta; g.ib_slope = slope; g.ib_partial = partial; g.ib_nparts = nparts; g.ib_cap = partial_rows;
    return run_gather(g, ws, ws_packed, (hipStream_t)stream);
}
extern "C" int m1_convT3d_dgrad(const m1_conv_desc_t* d, const float* w, const void* dy, void* const* dx, const int* accumulate,
                                void* ws, int ws_packed, void* stream) {
    if (m1_debug_skip("conv_dgrad")) return M1_OK;
    if (!desc_ok(d) || !w || !dy || !dx) return M1_ERR_BAD_ARG;
    M1ProfScope ps(prof_name("convT3d_dgrad", d).s, 2.0 * conv_macs(d, true), conv_bytes(d, true), (hipStream_t)stream);
    return dgrad_common(d, true, w, dy, dx, accumulate, ws, ws_packed, (hipStream_t)stream);
}

// ---- weight gradients ------------------------------------------------------------------------------------------
#include <stdlib.h>
static bool tf_wanted(const WgradSpec& g) {
    int maxc = M1_CFG("M1_TF_MAXC", 128);
    // 32x32 channel tiles re-read dY once per 32 input channels and X once per 32 output channels: a win while the OUTPUT side
    // is narrow (conv1 of an SE block: F/4 <= 32 channels -- 512->32 at res2: 1.70 -> 1.08 ms per step), a loss beyond (512->128: 2x slower)
    if (g.CA > 64 && g.CB > 32) return false;
    return g.CA <= maxc && g.CB <= maxc && m1_tf_wgrad_supported(g);
}
static bool m1_tf_wgrad_supported_stem(const m1_conv_desc_t* d, const Geo& q) {
    WgradSpec g{};
    const int CP = d->dtype == M1_BF16 ? 8 : 4;
    g.N = d->N; g.kd = d->kd; g.kh = d->kh; g.kw = d->kw; g.sd = d->sd; g.sh = d->sh; g.sw = d->sw;
    g.pd = q.pd; g.ph = q.ph; g.pw = q.pw; g.dtype = d->dtype;
    g.CA = CP; g.AD = d->D; g.AH = d->H; g.AW = d->W; g.CB = d->Cout; g.BD = q.OD; g.BH = q.OH; g.BW = q.OW;
    g.RT = (long long)CP * d->Cout; g.RSA = d->Cout;
    return d->dtype == M1_BF16 ? m1_tf_wgrad_supported(g) : m1_t3s_wgrad_supported(g);
}
static int wgrad_common(const m1_conv_desc_t* d, bool T, const void* dy, float* dw, float* db, void* ws, hipStream_t st,
                        int accumulate) {
    Geo q = T ? convT_geo(d) : conv_geo(d);
    const size_t nw = (size_t)d->kd * d->kh * d->kw * d->Cin * d->Cout;
    if (!accumulate && m1_zero_async(dw, nw * sizeof(float), st) != M1_OK) return M1_ERR_LAUNCH;
    // Conv3D bias gradient rides on the matrix-core wgrad of the first concat member (all-ones fragment); the
    // transposed conv (its dOut is the SHIFTED operand) and the direct path keep the separate column-sum pass
    const bool fuse_db = db && !T && !g_force_direct;
    if (fuse_db && !accumulate && m1_zero_async(db, (size_t)d->Cout * sizeof(float), st) != M1_OK) return M1_ERR_LAUNCH;
    int off = 0;
    const int nbias = fuse_db ? d->Cout : 0;
    float* rx = nullptr; long long rx_floats = 0;              // partial-copy scratch: the tail of the caller's workspace
    if (ws && wgrad_rx_bytes(d)) {
        rx = reinterpret_cast<float*>((unsigned char*)ws + align256(m1_reduce_ws_floats(d->N, (long long)q.OD * q.OH * q.OW, d->Cout, 1) * sizeof(float)) +
                                      256 + stem_ws_bytes(d, T));
        rx_floats = (long long)(wgrad_rx_bytes(d) / sizeof(float));
    }
    if (stem_wanted(d, T) && ws && m1_tf_wgrad_supported_stem(d, q)) {
        const int taps = d->kd * d->kh * d->kw, Cin = d->Cin;
        const long long nvox = (long long)d->N * d->D * d->H * d->W;
        unsigned char* base = (unsigned char*)ws + align256(m1_reduce_ws_floats(d->N, (long long)q.OD * q.OH * q.OW, d->Cout, 1) * sizeof(float)) + 256;
        const bool f32 = d->dtype == M1_F32;
        const int CP = f32 ? 4 : 8;                       // padded channels: one 16-byte segment per voxel
        uint4* x8 = reinterpret_cast<uint4*>(base);
        float* r8 = reinterpret_cast<float*>(base + align256((size_t)nvox * 16));
        const size_t nw8 = (size_t)taps * CP * d->Cout;
        if (m1_zero_async(r8, nw8 * sizeof(float), st) != M1_OK) return M1_ERR_LAUNCH;
        long long pb = (nvox + 255) / 256; if (pb > 8192) pb = 8192;
        if (f32) hipLaunchKernelGGL(stem_pad4f_kernel, dim3((unsigned)pb), dim3(256), 0, st, (const float*)d->src[0].ptr, reinterpret_cast<float4*>(x8), nvox, Cin);
        else hipLaunchKernelGGL(stem_pad8_kernel, dim3((unsigned)pb), dim3(256), 0, st, (const unsigned short*)d->src[0].ptr, x8, nvox, Cin);
        WgradSpec g{};
        g.N = d->N; g.R = r8; g.kd = d->kd; g.kh = d->kh; g.kw = d->kw; g.sd = d->sd; g.sh = d->sh; g.sw = d->sw;
        g.pd = q.pd; g.ph = q.ph; g.pw = q.pw; g.dtype = d->dtype;
        g.A = x8; g.CA = CP; g.AD = d->D; g.AH = d->H; g.AW = d->W;
        g.B = dy; g.CB = d->Cout; g.BD = q.OD; g.BH = q.OH; g.BW = q.OW;
        g.RT = (long long)CP * d->Cout; g.RSA = d->Cout; g.a_off = 0; g.b_off = 0;
        g.rx = rx; g.rx_floats = rx_floats;
        if (fuse_db) { g.bsum = db; g.bsum_tap = (q.pd * d->kh + q.ph) * d->kw + q.pw; }
        const int was = m1_fold_defer_set(0);          // (stem_fold_kernel below reads the folded 8-channel gradient)
        int rc = f32 ? m1_t3s_wgrad(g, (long long)nw8, nbias, st) : m1_tf_wgrad(g, (long long)nw8, nbias, st);
        m1_fold_defer_set(was);
        if (rc == M1_OK) {
            hipLaunchKernelGGL(stem_fold_kernel, dim3((taps * Cin * d->Cout + 255) / 256), dim3(256), 0, st, r8, dw, taps, Cin, d->Cout, CP);
            return m1_check_launch();
        }
        if (rc != M1_ERR_UNSUPPORTED && rc != M1_ERR_WORKSPACE) return rc;      // declined: nothing launched, take the generic path
    }
    for (int i = 0; i < d->nsrc; ++i) {
        WgradSpec g{};
        g.N = d->N; g.R = dw; g.kd = d->kd; g.kh = d->kh; g.kw = d->kw; g.sd = d->sd; g.sh = d->sh; g.sw = d->sw;
        g.pd = q.pd; g.ph = q.ph; g.pw = q.pw; g.dtype = d->dtype;
        g.rx = rx ? rx + (long long)i * rx_floats : nullptr; g.rx_floats = rx_floats;          // this member's own copy region
        if (!T) {   // dw[tap][ci][co] = sum X[v*s+tap-p][ci] * dY[v][co]
            g.A = d->src[i].ptr; g.CA = d->src[i].C; g.AD = d->D; g.AH = d->H; g.AW = d->W;
            g.B = dy; g.CB = d->Cout; g.BD = q.OD; g.BH = q.OH; g.BW = q.OW;
            g.RT = (long long)d->Cin * d->Cout; g.RSA = d->Cout; g.a_off = off; g.b_off = 0;
            if (fuse_db && i == 0) { g.bsum = db; g.bsum_tap = (q.pd * d->kh + q.ph) * d->kw + q.pw; }
        } else {    // dw[tap][co][ci] = sum dOut[i*s+tap-pb][co] * In[i][ci]
            g.A = dy; g.CA = d->Cout; g.AD = q.OD; g.AH = q.OH; g.AW = q.OW;
            g.B = d->src[i].ptr; g.CB = d->src[i].C; g.BD = d->D; g.BH = d->H; g.BW = d->W;
            g.RT = (long long)d->Cout * d->Cin; g.RSA = d->Cin; g.a_off = 0; g.b_off = off;
        }
        int rc = M1_ERR_UNSUPPORTED;
        int wlog = M1_CFG("M1_WG_LOG", 0);
        // >= 64 channels on both sides, stride 1: the 64x64-tile tap-fused kernel on 32x32x16 MFMAs, one launch for a run of
        // equal-width members (dY staged once per kd slice for all of them, wgrad_t3.hip)
        if (!g_force_direct && rx && m1_t3_wgrad_supported(g)) {
            int n = 1;                                    // (transposed conv: the members are on the dY-less side, one call each)
            while (!T && i + n < d->nsrc && d->src[i + n].C == d->src[i].C) ++n;
            const void* Am[M1_MAX_SRC]; int aoffs[M1_MAX_SRC]; int o = off;
            for (int m = 0; m < n; ++m) { Am[m] = d->src[i + m].ptr; aoffs[m] = o; o += d->src[i + m].C; }
            rc = m1_t3_wgrad(g, (long long)nw, nbias, st, n, Am, aoffs, rx_floats);
            if (wlog) fprintf(stderr, "wgrad %s N%d B %dx%dx%d CA %d x%d CB %d k%d%d%d s%d%d%d -> t3 rc %d\n", T ? "convT" : "conv", g.N, g.BD, g.BH, g.BW, g.CA, n, g.CB, g.kd, g.kh, g.kw, g.sd, g.sh, g.sw, rc);
            if (rc == M1_OK) { off = o; i += n - 1; continue; }
            if (rc != M1_ERR_UNSUPPORTED && rc != M1_ERR_WORKSPACE) return rc;
            rc = M1_ERR_UNSUPPORTED;
        }
        if (!g_force_direct && rx && g.dtype == M1_F32 && !T && m1_pwf_wgrad_supported(g)) {
            rc = m1_pwf_wgrad(g, (long long)nw, nbias, st);
            if (wlog) fprintf(stderr, "wgrad conv N%d B %dx%dx%d CA %d CB %d k111 -> pwf rc %d\n", g.N, g.BD, g.BH, g.BW, g.CA, g.CB, rc);
            if (rc == M1_OK) { off += d->src[i].C; continue; }
            if (rc != M1_ERR_UNSUPPORTED && rc != M1_ERR_WORKSPACE) return rc;
            rc = M1_ERR_UNSUPPORTED;
        }
        // fp32 layers the 64x64-tile kernel does not take (few channels on a side): the 32x32-tile tap-fused fp32 kernel
        if (!g_force_direct && rx && g.dtype == M1_F32 && m1_t3s_wgrad_supported(g)) {
            rc = m1_t3s_wgrad(g, (long long)nw, nbias, st);
            if (wlog) fprintf(stderr, "wgrad %s N%d B %dx%dx%d CA %d CB %d k%d%d%d s%d%d%d -> t3s rc %d\n", T ? "convT" : "conv", g.N, g.BD, g.BH, g.BW, g.CA, g.CB, g.kd, g.kh, g.kw, g.sd, g.sh, g.sw, rc);
            if (rc == M1_OK) { off += d->src[i].C; continue; }
            if (rc != M1_ERR_UNSUPPORTED && rc != M1_ERR_WORKSPACE) return rc;
            rc = M1_ERR_UNSUPPORTED;
        }
        // a run of members with the same channel count on the tap-fused kernel: ONE launch (blockIdx.z = member)
        int multi = M1_CFG("M1_TF_MULTI", 1);
        if (multi && !T && !g_force_direct && rx && tf_wanted(g)) {
            int n = 1;
            while (i + n < d->nsrc && d->src[i + n].C == d->src[i].C) ++n;
            if (n > 1) {
                const void* Am[M1_MAX_SRC]; int aoffs[M1_MAX_SRC]; int o = off;
                for (int m = 0; m < n; ++m) { Am[m] = d->src[i + m].ptr; aoffs[m] = o; o += d->src[i + m].C; }
                rc = m1_tf_wgrad_multi(g, (long long)nw, nbias, st, n, Am, aoffs, rx_floats);
                if (wlog) fprintf(stderr, "wgrad conv N%d B %dx%dx%d CA %d x%d CB %d -> tap-fused multi rc %d\n", g.N, g.BD, g.BH, g.BW, g.CA, n, g.CB, rc);
                if (rc == M1_OK) { off = o; i += n - 1; continue; }
                if (rc != M1_ERR_UNSUPPORTED && rc != M1_ERR_WORKSPACE) return rc;
                rc = M1_ERR_UNSUPPORTED;
            }
        }
        if (!g_force_direct && tf_wanted(g)) rc = m1_tf_wgrad(g, (long long)nw, nbias, st);   // may decline (no launch)
        if (wlog) fprintf(stderr, "wgrad %s N%d B %dx%dx%d CA %d CB %d k%d%d%d s%d%d%d -> %s\n", T ? "convT" : "conv", g.N, g.BD, g.BH, g.BW, g.CA, g.CB,
                          g.kd, g.kh, g.kw, g.sd, g.sh, g.sw, rc == M1_OK ? "tap-fused" : (m1_tap_wgrad_supported(g) ? "tap" : "mfma"));
        if (rc == M1_OK) { off += d->src[i].C; continue; }
        if (rc != M1_ERR_UNSUPPORTED && rc != M1_ERR_WORKSPACE) return rc;
        if (g_force_direct == 2 && m1_skinny_wgrad_supported(g)) rc = m1_skinny_wgrad(g, st);   // test hook for that kernel
        else if (!g_force_direct && m1_tap_wgrad_supported(g)) rc = m1_tap_wgrad(g, (long long)nw, nbias, st);
        else if (!g_force_direct && m1_mfma_wgrad_supported(g)) rc = m1_mfma_wgrad_ex(g, (long long)nw, nbias, st);
        else rc = m1_direct_wgrad(g, st);
        if (rc) return rc;
        off += d->src[i].C;
    }
    if (db && !fuse_db) {
        if (!ws) return M1_ERR_WORKSPACE;
        const long long Vd = (long long)q.OD * q.OH * q.OW;
        if (!g_force_direct && m1_fold_defer_get() && m1_colsum_defer(dy, d->N, Vd, d->Cout, d->dtype, db, (float*)ws, accumulate)) return M1_OK;
        return m1_colsum_internal(dy, d->N, Vd, d->Cout, d->dtype, db, (float*)ws, st, accumulate);
    }
    return M1_OK;
}
extern "C" int m1_conv3d_wgrad(const m1_conv_desc_t* d, const void* dy, float* dw, float* db, void* ws, int accumulate,
                               void* stream) {
    if (m1_debug_skip("conv_wgrad")) return M1_OK;
    if (!desc_ok(d) || !dy || !dw) return M1_ERR_BAD_ARG;
    M1ProfScope ps(prof_name("conv3d_wgrad", d).s, 2.0 * conv_macs(d, false), conv_bytes(d, false), (hipStream_t)stream);
    return wgrad_common(d, false, dy, dw, db, ws, (hipStream_t)stream, accumulate);
}
extern "C" int m1_convT3d_wgrad(const m1_conv_desc_t* d, const void* dy, float* dw, float* db, void* ws, int accumulate,
                                void* stream) {
    if (m1_debug_skip("conv_wgrad")) return M1_OK;
    if (!desc_ok(d) || !dy || !dw) return M1_ERR_BAD_ARG;
    M1ProfScope ps(prof_name("convT3d_wgrad", d).s, 2.0 * conv_macs(d, true), conv_bytes(d, true), (hipStream_t)stream);
    return wgrad_common(d, true, dy, dw, db, ws, (hipStream_t)stream, accumulate);
}
