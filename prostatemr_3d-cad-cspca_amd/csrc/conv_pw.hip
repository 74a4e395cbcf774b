// conv_pw.hip -- pointwise (1x1x1, stride 1) bf16 convolutions as a streaming GEMM: conv3 of every SE block (F/4 -> F,
// network_blocks.py:58-60), the attention-gate projections (network_blocks.py:100-117) and their data gradients.
//
//   out[v][oc] = bias[oc] + sum_c X[v][c] * Wp[oc][c]          v = voxel of the NDHWC tensors, c over the virtual concat
//
// These layers move 4-16x more output than input bytes and have 16..256-deep contractions: they are HBM-bound, and in
// conv_mfma.hip (row tables, LDS operand tiles, a barrier per 64 deep stage, an LDS output tile) the fixed cost of a tile
// is the whole run time.  Here:
// * a block keeps its <= 32-wide slice of the packed weight panel in LDS (loaded once; wider slices cost > 128 VGPRs of
//   per-lane epilogue state -- the input is re-read per slice from L2 / the Infinity Cache instead);
// * a WAVE owns 32 consecutive voxels per step: the MFMA B operand (8 consecutive channels of one voxel per lane) is
//   exactly a 16-byte global load of the NDHWC row -- no LDS staging, no barrier in the loop;
// * the weights are the A operand, so a lane ends up with 4 consecutive output channels of one voxel: bias, bf16
//   rounding (v_cvt_pk_bf16_f32), optional accumulate, InstanceNorm statistics and an 8-byte store in registers;
// * statistics: running per-lane sums, one partial per (sample, wave of the grid), written when the walk leaves the sample.
#include "conv_mfma.h"
#include "reduce.h"
#include <stdlib.h>

typedef unsigned u32x4p_t __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(2))) __bf16 pbf2_t;
typedef __attribute__((ext_vector_type(2))) float pf2_t;
__device__ __forceinline__ unsigned pw_cvt_pk(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector((pf2_t){a, b}, pbf2_t));
}
__device__ __forceinline__ int pw_swz(int row, int seg) { return seg ^ ((-(row >> 2)) & 3); }

struct PwP {
    MfmaP m;
    int kpad, nseg;              // panel row length (elements); valid 16-byte K segments (= CC / 8)
    int nwaves;                  // waves of the grid along y (= statistics partials per sample)
    int V;                       // voxels per sample
    long long Mtot;              // voxels in all
};

template <int TN, int NCH>
__global__ void __launch_bounds__(256) conv_pw_kernel(PwP p) {
    constexpr int BN = TN * 16, NW = 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];     // [NCH][BN][64] weights, swizzled
    const MfmaP& m = p.m;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fs = lane >> 4;
    const int oc0 = blockIdx.x * BN;

    // ---- weight slice -> LDS (plain loads + ds_write: once per block) ----
    {
        const bf16_t* wp = (const bf16_t*)m.wp + m.cls_woff[0];
        for (int q = tid; q < NCH * BN * 4; q += 256) {
            const int ch = q / (BN * 4), r = (q / 4) % BN, s = q & 3;
            const uint4 v = *reinterpret_cast<const uint4*>(wp + (long long)(oc0 + r) * p.kpad + (ch * 4 + s) * 8);
            *reinterpret_cast<uint4*>(smem + ((ch * BN + r) * 4 + pw_swz(r, s)) * 16) = v;
        }
    }
    __syncthreads();

    // ---- this lane's K segment of every chunk: member pointer + channel offset, row pitch; beyond the concat: zeros ----
    const bf16_t* xp[NCH]; int xC[NCH];
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
        const int kseg = q * 4 + fs;
        int c = kseg * 8, s = 0;
        while (s < m.nsrc - 1 && c >= m.srcC[s]) { c -= m.srcC[s]; ++s; }
        const bool ok = kseg < p.nseg;
        xp[q] = ok ? (const bf16_t*)m.src[s] + c : nullptr; xC[q] = ok ? m.srcC[s] : 0;
    }
    // ---- epilogue invariants (lane: voxel i*16 + fr of the wave tile, channels oc0 + j*16 + fs*4 .. +3) ----
    bf16_t* o_base[TN]; int o_C[TN], o_nv[TN], o_acc[TN], o_fast[TN]; float bias_r[TN][4];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int oc = oc0 + j * 16 + fs * 4;
        int nv = m.OCn - oc; nv = nv > 4 ? 4 : (nv < 0 ? 0 : nv);
        o_base[j] = nullptr; o_C[j] = 0; o_acc[j] = 0; o_fast[j] = 0;
        if (nv > 0) {
            const OutRef o = m1_out_ref(m, oc);
            if (o.base) { o_base[j] = (bf16_t*)o.base + o.col; o_C[j] = o.C; o_acc[j] = o.acc; o_fast[j] = nv == 4 && (o.C & 3) == 0 && (o.col & 3) == 0; }
            else nv = 0;
        }
        o_nv[j] = nv;
#pragma unroll
        for (int r = 0; r < 4; ++r) bias_r[j][r] = (oc + r < m.OCn) ? m1_bias_at(m, oc + r) : 0.f;
    }
    const unsigned char* const w_rd = smem + fr * 64 + pw_swz(fr, fs) * 16;     // fragment of oc row j*16 + fr: + (q*BN + j*16)*64

    const bool want_stats = m.stat_partial != nullptr;
    // InstanceNorm-backward sums instead of statistics (MfmaP::ib_x: the output is d(a), a = lrelu(IN(x)); round 4)
    const bool ib = m.ib_x != nullptr;
    float ib_g[TN][4], ib_b[TN][4];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int oc = oc0 + j * 16 + fs * 4 + r;
            ib_g[j][r] = (ib && oc < m.OCn) ? m.ib_gamma[oc] : 0.f; ib_b[j][r] = (ib && oc < m.OCn) ? m.ib_beta[oc] : 0.f;
        }
    float ssum[TN][4], ssq[TN][4];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) { ssum[j][r] = 0.f; ssq[j][r] = 0.f; }
    const int part = blockIdx.y * NW + wave;            // this wave's statistics slot
    int cur_n = 0;
    auto flush = [&](int n) {           // one partial per (sample, wave): 16-lane fold, lanes fr == 0 write their 4 channels
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s = ssum[j][r], q = ssq[j][r];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
                const int oc = oc0 + j * 16 + fs * 4 + r;
                if (fr == 0 && oc < m.OCn) {
                    float* dst = m.stat_partial + (((long long)n * p.nwaves + part) * m.OC + oc) * 2;
                    dst[0] = s; dst[1] = q;
                }
                ssum[j][r] = 0.f; ssq[j][r] = 0.f;
            }
    };

    const long long step = (long long)p.nwaves * 32;
    // voxel fragments: 16 bytes per lane and chunk, straight from the NDHWC rows; the next tile's are fetched before this
    // tile's MFMAs (the loop has no other latency hiding than the waves per SIMD)
    u32x4p_t xf[2][NCH], xn[2][NCH];
    auto fetch = [&](long long v0, u32x4p_t (&x)[2][NCH]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const long long v = v0 + i * 16 + fr;
#pragma unroll
            for (int q = 0; q < NCH; ++q) {
                x[i][q] = (u32x4p_t){0u, 0u, 0u, 0u};
                if (v < p.Mtot && xp[q]) x[i][q] = *reinterpret_cast<const u32x4p_t*>(xp[q] + v * xC[q]);
            }
        }
    };
    fetch((long long)part * 32, xf);
    for (long long v0 = (long long)part * 32; v0 < p.Mtot; v0 += step) {
        if (want_stats) { const int n = (int)(v0 / p.V); for (; cur_n < n; ++cur_n) flush(cur_n); }
        fetch(v0 + step, xn);
        bool vok[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) vok[i] = v0 + i * 16 + fr < p.Mtot;
        f32x4_t acc[2][TN];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < NCH; ++q)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const u32x4p_t wf = *reinterpret_cast<const u32x4p_t*>(w_rd + (q * BN + j * 16) * 64);
#pragma unroll
                for (int i = 0; i < 2; ++i)       // D[oc][voxel]: weights as A, voxels as B
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf), __builtin_bit_cast(bf16x8_t, xf[i][q]), acc[i][j], 0, 0, 0);
            }
        // ---- epilogue in registers ----
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (!vok[i]) continue;
            const long long orow = v0 + i * 16 + fr;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (o_nv[j] == 0) continue;
                bf16_t* dst = o_base[j] + orow * o_C[j];
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] + bias_r[j][r];
                if (o_fast[j]) {
                    if (o_acc[j]) {
                        const uint2 ov = *reinterpret_cast<const uint2*>(dst);
                        const unsigned r01 = pw_cvt_pk(v[0], v[1]), r23 = pw_cvt_pk(v[2], v[3]);
                        v[0] = __uint_as_float(r01 << 16) + __uint_as_float(ov.x << 16); v[1] = __uint_as_float(r01 & 0xffff0000u) + __uint_as_float(ov.x & 0xffff0000u);
                        v[2] = __uint_as_float(r23 << 16) + __uint_as_float(ov.y << 16); v[3] = __uint_as_float(r23 & 0xffff0000u) + __uint_as_float(ov.y & 0xffff0000u);
                    }
                    uint2 o;
                    o.x = pw_cvt_pk(v[0], v[1]); o.y = pw_cvt_pk(v[2], v[3]);
                    *reinterpret_cast<uint2*>(dst) = o;
                    if (want_stats) {
                        const float r0 = __uint_as_float(o.x << 16), r1 = __uint_as_float(o.x & 0xffff0000u);
                        const float r2 = __uint_as_float(o.y << 16), r3 = __uint_as_float(o.y & 0xffff0000u);
                        if (ib) {             // {sum dy, sum dy * xh}, dy = d(a) * lrelu'(gamma xh + beta)
                            const int oc = oc0 + j * 16 + fs * 4, n = (int)(orow / p.V);
                            const float* msp = m.ib_stats + ((long long)n * m.OC + oc) * 2;
                            const float4 ms0 = *reinterpret_cast<const float4*>(msp), ms1 = *reinterpret_cast<const float4*>(msp + 4);
                            const uint2 xw = *reinterpret_cast<const uint2*>((const bf16_t*)m.ib_x + orow * m.OC + oc);
                            const float x0 = (__uint_as_float(xw.x << 16) - ms0.x) * ms0.y, x1 = (__uint_as_float(xw.x & 0xffff0000u) - ms0.z) * ms0.w;
                            const float x2 = (__uint_as_float(xw.y << 16) - ms1.x) * ms1.y, x3 = (__uint_as_float(xw.y & 0xffff0000u) - ms1.z) * ms1.w;
                            const float d0 = r0 * lrelu_g(ib_g[j][0] * x0 + ib_b[j][0], m.ib_slope), d1 = r1 * lrelu_g(ib_g[j][1] * x1 + ib_b[j][1], m.ib_slope);
                            const float d2 = r2 * lrelu_g(ib_g[j][2] * x2 + ib_b[j][2], m.ib_slope), d3 = r3 * lrelu_g(ib_g[j][3] * x3 + ib_b[j][3], m.ib_slope);
                            ssum[j][0] += d0; ssq[j][0] += d0 * x0; ssum[j][1] += d1; ssq[j][1] += d1 * x1;
                            ssum[j][2] += d2; ssq[j][2] += d2 * x2; ssum[j][3] += d3; ssq[j][3] += d3 * x3;
                        } else {
                        ssum[j][0] += r0; ssq[j][0] += r0 * r0; ssum[j][1] += r1; ssq[j][1] += r1 * r1;
                        ssum[j][2] += r2; ssq[j][2] += r2 * r2; ssum[j][3] += r3; ssq[j][3] += r3 * r3;
                        }
                    }
                } else {                      // rows that are not 8-byte tiled (1..3 channels), or a partial group
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (r < o_nv[j]) {
                            float a = v[r];
                            if (o_acc[j]) a = bf2f(f2bf(a)) + bf2f(dst[r]);
                            const bf16_t e = f2bf(a); dst[r] = e;
                            const float vr = bf2f(e); ssum[j][r] += vr; ssq[j][r] += vr * vr;
                        }
                }
            }
        }
    #pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < NCH; ++q) xf[i][q] = xn[i][q];
    }
    if (want_stats) { for (; cur_n < m.N; ++cur_n) flush(cur_n); }
}

// ------------------------------------------------------------------------------------------------
static bool pw_plan(const MfmaP& m, int OCpad, int BN, PwP& p) {
    int en = M1_CFG("M1_PW", 1);
    if (!en) return false;
    if (m.nclasses != 1 || m.ksplit != 1 || m.cls_ntaps[0] != 1) return false;
    if (m.tdd[0] != 0 || m.tdh[0] != 0 || m.tdw[0] != 0) return false;
    if (m.sd != 1 || m.sh != 1 || m.sw != 1) return false;
    if (m.ID != m.OD || m.IH != m.OH || m.IW != m.OW) return false;
    int CC = 0;
    for (int i = 0; i < m.nsrc; ++i) { if (m.srcC[i] % 8) return false; CC += m.srcC[i]; }
    const int kpad = m.cls_kpad[0];
    // contractions of up to 64 channels only: measured against conv_mfma (us, forward / data gradient) 16->64 at (4,20,80,80)
    // 40 / 18 vs 60 / 25, 32->32 at (2,20,160,160) 49 / 30 vs 62 / 47, 32->128 on par, but 128->128 42 / 42 vs 37 / 26 and
    // 256->256 41 / 33 vs 30 / 16 -- with 4 or 8 chunks of voxel fragments in registers the waves per SIMD run out
    if (kpad % 32 || kpad / 32 > 2 || CC > kpad) return false;
    if (!(BN == 16 || BN == 32) || OCpad % BN) return false;      // (wider slices: > 128 VGPRs of per-lane epilogue state)
    p = PwP{}; p.m = m; p.kpad = kpad; p.nseg = CC / 8;
    p.V = m.OD * m.OH * m.OW; p.Mtot = (long long)m.N * p.V;
    if (p.V % 32) return false;                   // a wave tile never straddles two samples
    return true;
}
static int pw_nwaves(const PwP& p, int OCpad, int BN) {
    const int slices = OCpad / BN;
    int tg = M1_CFG("M1_PW_BLOCKS", 2048);
    long long blocks = tg / slices; if (blocks < 1) blocks = 1;
    const long long need = (p.Mtot + 127) / 128;
    if (blocks > need) blocks = need;
    return (int)blocks * 4;
}
bool m1_pw_conv_supported(const MfmaP& mp, int OCpad, int BN) { PwP p; return pw_plan(mp, OCpad, BN, p); }
int m1_pw_conv_stat_parts(const MfmaP& mp, int OCpad, int BN) { PwP p; return pw_plan(mp, OCpad, BN, p) ? pw_nwaves(p, OCpad, BN) : 0; }

int m1_pw_conv(const MfmaP& mp, int OCpad, int BN, hipStream_t st) {
    PwP p;
    if (!pw_plan(mp, OCpad, BN, p)) return M1_ERR_UNSUPPORTED;
    // fused statistics: the caller chose the partial count (<= what its workspace holds, a multiple of 4 waves)
    p.nwaves = mp.stat_partial ? mp.stat_tiles : pw_nwaves(p, OCpad, BN);
    if (p.nwaves < 4 || p.nwaves % 4) return M1_ERR_BAD_ARG;
    const int nch = p.kpad / 32;
    const int NCH = nch <= 1 ? 1 : 2;
    if (NCH * 32 != p.kpad) {
        // the panel rows are kpad long; chunks beyond kpad would read the next row: only exact powers of two take this path
        return M1_ERR_UNSUPPORTED;
    }
    void (*kern)(PwP) = nullptr;
#define PK(TN_, NCH_) if (BN == TN_ * 16 && NCH == NCH_) kern = conv_pw_kernel<TN_, NCH_>;
    PK(1, 1) PK(1, 2) PK(2, 1) PK(2, 2)
#undef PK
    if (!kern) return M1_ERR_UNSUPPORTED;
    const size_t smem = (size_t)NCH * BN * 64;
    if (smem > 48 * 1024) {
        static const void* done[16]; static int ndone = 0;
        bool seen = false;
        for (int q = 0; q < ndone; ++q) seen |= done[q] == (const void*)kern;
        if (!seen) {
            if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) != hipSuccess) return M1_ERR_LAUNCH;
            if (ndone < 16) done[ndone++] = (const void*)kern;
        }
    }
    m1_note_kernel("conv_pw:bn%d", BN);
    hipLaunchKernelGGL(kern, dim3(OCpad / BN, p.nwaves / 4), dim3(256), smem, st, p);
    return m1_check_launch();
}
