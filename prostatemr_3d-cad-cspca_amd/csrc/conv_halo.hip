// conv_halo.hip -- halo-tile implicit-GEMM convolution (bf16) for the wide, shallow layers (res0/res1: hundreds of
// thousands of voxels, <= 64 contraction channels): the case where conv_mfma.hip's per-tap im2col gather re-reads every
// input voxel once per tap from L2 and is bound by the L2 -> LDS path.
//
//   out[o][oc] = bias[oc] + sum_t sum_c X[o*s - p + off_t][c] * Wp[oc][t*CC + c]       (same packed panel as conv_mfma)
//
// * A block owns one <= 32-wide slice of the output channels and keeps that slice of the weight panel RESIDENT in LDS
//   ([chunk][oc][64 B], swizzled) while it walks 256-voxel output tiles (TH rows x TW columns of one (n, d) slice; 128 with
//   256 threads).
// * Per tile the input voxels of the tile INCLUDING the stencil halo are staged once, by LDS-DMA
//   (global_load_lds_dwordx4), as voxel-major rows holding the whole channel concat; a tap is a row offset into
//   that tile, so every input voxel crosses L2 -> LDS once per tile instead of once per tap.
// * K order is [tap][member][channel] as in the panel; the per-lane fragment offsets of all K chunks are tile
//   invariant and live in registers; fragment reads are ds_read_b128 by inline asm, one chunk ahead of the MFMAs,
//   the tiles themselves run in a multi-stage DMA pipeline with counted vmcnt waits (as wgrad_tf.hip).
// * Epilogue in registers: the weights are the MFMA's A operand, so a lane holds 4 consecutive output channels of one
//   voxel -- bias, v_cvt_pk_bf16_f32, the optional out += (fetched ahead of the next tile's DMA), running InstanceNorm
//   statistics (one partial per sample and block) and an 8-byte store; no LDS output tile, one barrier per tile.
// * The K-chunk / DMA-piece counts of the M1 layer shapes are template parameters (straight-line tile loop, immediate vmcnt).
// * Parity-class mode (template B1 > 0; round 4): the data gradient of a (1,2,2)-strided 1x3x3 conv / the forward of the matching
//   transposed conv (res0 <-> res1) is four stride-1 problems in the q-domain (o = 2q + parity), one per (h, w) parity, with 4 / 2 /
//   2 / 1 taps.  One block stages the dY tile (halo +-1) ONCE, keeps the weight slices of all four classes in LDS, runs the four K
//   ranges [0,B1) [B1,B2) [B2,B3) [B3,NCH) into four accumulator sets and writes the 2x2-interleaved output rows -- conv_mfma ran the
//   classes as separate blocks on 32-column tiles (117 TFLOP/s on 64 -> 32 at (2,20,160,160)).
#include "conv_mfma.h"
#include "reduce.h"
#include <stdlib.h>
#include <stdio.h>

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

__device__ __attribute__((aligned(64))) unsigned int m1_zero_page_h[16];

__device__ __forceinline__ void glds16h(const void* g, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ u32x4_t lds_read128h(unsigned lds_addr) {
    u32x4_t v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(lds_addr) : "memory");
    return v;
}
__device__ __forceinline__ void lds_waith(u32x4_t& v) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v)); }
__device__ __forceinline__ void lds_tieh(u32x4_t& v) { asm volatile("" : "+v"(v)); }     // no instruction: ordering only
__device__ __forceinline__ void wait_vmh(int n) {
    switch (n) {
#define HW(N_) case N_: asm volatile("s_waitcnt vmcnt(" #N_ ")" ::: "memory"); break;
        HW(0) HW(1) HW(2) HW(3) HW(4) HW(5) HW(6) HW(7) HW(8) HW(9) HW(10) HW(11) HW(12) HW(13) HW(14) HW(15) HW(16)
        HW(17) HW(18) HW(19) HW(20) HW(21) HW(22) HW(23) HW(24) HW(25) HW(26) HW(27) HW(28) HW(29) HW(30) HW(31) HW(32)
        HW(33) HW(34) HW(35) HW(36) HW(37) HW(38) HW(39) HW(40)
#undef HW
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

#define HL_MAX_XIT 10        // LDS-DMA pieces per thread for one input tile
#define HL_MAX_CH 18         // K chunks (of 32) whose fragment offsets live in registers (18 = 9 taps x 64 channels; the run-time variant on 512 threads sits at the 256-VGPR limit)

struct HaloP {
    MfmaP m;
    int TW, TH, tiles_w, tiles_h, ntiles, nsplit;
    int KDs, IHt, IWt;               // input tile extent (slices, rows, columns) incl. halo
    int dmin, hmin, wmin;            // smallest tap offsets
    int sde, she, swe, pde, phe, pwe;// in = o*s - p + off  (mode 1, stride 1: s = 1, p = 0)
    int PX, spr;                     // input tile row pitch (bytes) = CC*2, 16-byte slots per row
    int x_slots, x_bytes;            // slots (rounded to 256) / bytes of one input tile
    int nchunks, nseg, kpad;
    int b_bytes, c_bytes, stages;
    int BNh;                         // output channels per block (16 or 32)
    int nthr;                        // 256 or 512 threads: the output tile is nthr/2 voxels
    int tiles_per_sample;
    int spr_sh, tw_sh;               // log2 of spr / TW
    int ncls;                        // 1, or 4 = (h, w) parity classes (mode 1, stride (1,2,2)): tiles walk the q-domain QH x QW
    int QH, QW;                      // extent the tiles cover (= OH, OW for one class; OH/2, OW/2 for four)
};

// physical 16-byte slot of logical slot `sl` in tile row `row` (conflict-free ds_read_b128 of 16 consecutive rows)
__device__ __forceinline__ int x_swz(int row, int sl, int spr) {
    return spr == 8 ? (sl ^ (row & 7)) : (spr == 4 ? (sl ^ ((row >> 1) & 3)) : sl);
}
__device__ __forceinline__ int b_swz(int row, int seg) { return seg ^ ((-(row >> 2)) & 3); }

// pack two floats to bf16x2 (round to nearest even): one v_cvt_pk_bf16_f32
typedef __attribute__((ext_vector_type(2))) __bf16 hbf2_t;
typedef __attribute__((ext_vector_type(2))) float hf2_t;
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector((hf2_t){a, b}, hbf2_t));
}
// 8-byte load the compiler does not track (its own bookkeeping would wait vmcnt(0) at the first use, draining the DMA issued behind it)
__device__ __forceinline__ unsigned long long gload8_untracked(const void* p) {
    unsigned long long v;
    asm volatile("global_load_dwordx2 %0, %1, off" : "=&v"(v) : "v"(p) : "memory");
    return v;
}
template <int N_> __device__ __forceinline__ void wait_vm_c() {
    static_assert(N_ >= 0 && N_ <= 63, "vmcnt immediate");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory");
}

// NCH / NXIT > 0: the number of K chunks / of LDS-DMA pieces per thread are compile-time (the common layer shapes: straight-line
// tile loop, immediate wait counts); 0 = taken from the plan at run time (predicated unrolling up to HL_MAX_CH / HL_MAX_XIT)
template <int TN, int NTHR, int NCH, int NXIT, int B1 = 0, int B2 = 0, int B3 = 0>
__global__ void __launch_bounds__(NTHR) conv_halo_kernel(HaloP p) {
    constexpr int TM = 2, BN = TN * 16, SEG = 8, NW = NTHR / 64;      // each wave owns 32 voxels x BN channels
    constexpr bool RT = NCH == 0;
    constexpr bool CLS = B1 > 0;                                      // four parity classes: K chunks [0,B1) [B1,B2) [B2,B3) [B3,NCH)
    constexpr int NC = CLS ? 4 : 1;
    static_assert(!CLS || (NCH > 0 && B1 < B2 && B2 < B3 && B3 < NCH), "class boundaries");
#define cls_of(ch) (CLS ? ((ch) >= B1) + ((ch) >= B2) + ((ch) >= B3) : 0)
#define cls_lo(c) ((c) == 0 ? 0 : ((c) == 1 ? B1 : ((c) == 2 ? B2 : B3)))
    constexpr int CH = RT ? HL_MAX_CH : NCH, XIT = RT ? HL_MAX_XIT : NXIT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const Bs = smem;                                   // [nchunks][BN][64]
    float* const red = reinterpret_cast<float*>(smem + p.b_bytes);    // [NW][BN][2]  statistics of one sample, per wave
    unsigned char* const Xs0 = smem + p.b_bytes + p.c_bytes;          // [stages][x_bytes]
    const MfmaP& m = p.m;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int oc0 = blockIdx.x * BN;
    const unsigned char* zero_pg = reinterpret_cast<const unsigned char*>(m1_zero_page_h);
    const int fr = lane & 15, fs = lane >> 4;
    const int spt_sh = p.spr_sh;                                      // log2 of the 16-byte K segments per tap (= slots per row)

    // ---- weight panel slice -> LDS, once ----
    {
        const bf16_t* wp = (const bf16_t*)m.wp + m.cls_woff[0];
        const int nslot = p.nchunks * BN * 4;
        for (int q0 = 0; q0 < nslot; q0 += NTHR) {
            const int q = q0 + tid;                                   // LDS slot: chunk, oc row, physical 16-byte slot
            const int ch = q / (BN * 4), r = (q / 4) % BN, s = b_swz(r, q & 3);
            const unsigned char* src;
            if constexpr (CLS) {         // class c's matrix [OCpad][kpad_c] sits at cls_woff[c]
                const int c = cls_of(ch), lo = cls_lo(c), kp = m.cls_kpad[c];
                src = (q < nslot) ? reinterpret_cast<const unsigned char*>((const bf16_t*)m.wp + m.cls_woff[c] + (long long)(oc0 + r) * kp + ((ch - lo) * 4 + s) * SEG) : zero_pg;
            } else
            src = (q < nslot) ? reinterpret_cast<const unsigned char*>(wp + (long long)(oc0 + r) * p.kpad + (ch * 4 + s) * SEG)
                                                   : zero_pg;
            if (q0 + wave * 64 < nslot) glds16h(src, Bs + (q0 + wave * 64) * 16);
        }
    }

    // ---- per-lane description of its input-tile DMA pieces: address at tile origin 0 + what the bounds check needs ----
    const bf16_t* x_ptr[XIT]; int x_C[XIT], x_pk[XIT];
    const int nxit = RT ? p.x_slots / NTHR : NXIT;
    const int nchunks = RT ? p.nchunks : NCH;
    const int x_rows = p.KDs * p.IHt * p.IWt;
#pragma unroll
    for (int it = 0; it < XIT; ++it) {
        const int q = it * NTHR + tid;
        const int row = q >> spt_sh, slp = q & (p.spr - 1);
        const int sl = x_swz(row, slp, p.spr);                        // XOR swizzles are involutions
        const int dd = row / (p.IHt * p.IWt); const int r2 = row - dd * (p.IHt * p.IWt);
        const int hh = r2 / p.IWt, ww = r2 - hh * p.IWt;
        int c = sl * SEG, s = 0;                                      // channel on the concat axis -> member
        while (s < m.nsrc - 1 && c >= m.srcC[s]) { c -= m.srcC[s]; ++s; }
        x_C[it] = m.srcC[s];
        x_ptr[it] = (const bf16_t*)m.src[s] + (long long)((dd * m.IH + hh) * m.IW + ww) * m.srcC[s] + c;
        x_pk[it] = row < x_rows ? (dd | (hh << 8) | (ww << 16)) : -1;
    }

    // ---- tile walk (incremental decode: column tile, row tile, depth, sample) ----
    int q_kt = blockIdx.y, q_tw, q_th, q_od, q_n;
    { int r = q_kt; q_tw = r % p.tiles_w; r /= p.tiles_w; q_th = r % p.tiles_h; r /= p.tiles_h; q_od = r % m.OD; q_n = r / m.OD; }
    int s_tw, s_th, s_od, s_n;
    { int r = p.nsplit; s_tw = r % p.tiles_w; r /= p.tiles_w; s_th = r % p.tiles_h; r /= p.tiles_h; s_od = r % m.OD; s_n = r / m.OD; }
    auto advance = [&](int& kt, int& tw, int& th, int& od, int& n) {
        kt += p.nsplit;
        tw += s_tw; int c = tw >= p.tiles_w; tw -= c ? p.tiles_w : 0;
        th += s_th + c; c = th >= p.tiles_h; th -= c ? p.tiles_h : 0;
        od += s_od + c; c = od >= m.OD; od -= c ? m.OD : 0;
        n += s_n + c;
    };
    auto issue = [&](int st) {
        const int live = q_kt < p.ntiles;
        const int id0 = q_od * p.sde - p.pde + p.dmin, ih0 = q_th * p.TH * p.she - p.phe + p.hmin, iw0 = q_tw * p.TW * p.swe - p.pwe + p.wmin;
        const int lin0 = ((q_n * m.ID + id0) * m.IH + ih0) * m.IW + iw0;
        unsigned char* Xs = Xs0 + st * p.x_bytes;
        const long long zp = (long long)zero_pg;
        // the whole input tile (halo included) inside the volume: no per-piece bounds checks (rows beyond the tile: pk < 0)
        const bool inside = live && id0 >= 0 && id0 + p.KDs <= m.ID && ih0 >= 0 && ih0 + p.IHt <= m.IH && iw0 >= 0 && iw0 + p.IWt <= m.IW;
        if (inside) {
#pragma unroll
            for (int it = 0; it < XIT; ++it) {
                if (!RT || it < nxit) {
                    const long long real = (long long)(x_ptr[it] + (long long)lin0 * x_C[it]);
                    const int ok = x_pk[it] >= 0;
                    const unsigned char* src = reinterpret_cast<const unsigned char*>(zp + ((real - zp) & -(long long)ok));
                    glds16h(src, Xs + (it * NTHR + wave * 64) * 16);
                }
            }
        } else {
#pragma unroll
            for (int it = 0; it < XIT; ++it) {
                if (!RT || it < nxit) {
                    const int pk = x_pk[it];
                    const int dd = pk & 0xff, hh = (pk >> 8) & 0xff, ww = (pk >> 16) & 0xff;
                    // (bitwise, not &&: one straight line of compares; the select is arithmetic so that no branch guards the address)
                    const int ok = live & (int)(pk >= 0) & (int)((unsigned)(id0 + dd) < (unsigned)m.ID) & (int)((unsigned)(ih0 + hh) < (unsigned)m.IH) &
                                   (int)((unsigned)(iw0 + ww) < (unsigned)m.IW);
                    const long long real = (long long)(x_ptr[it] + (long long)lin0 * x_C[it]);
                    const unsigned char* src = reinterpret_cast<const unsigned char*>(zp + ((real - zp) & -(long long)ok));
                    glds16h(src, Xs + (it * NTHR + wave * 64) * 16);
                }
            }
        }
        advance(q_kt, q_tw, q_th, q_od, q_n);
    };

    // ---- fragment offsets (tile invariant): voxels per (chunk, 16-voxel tile), weights per 16-channel tile ----
    const unsigned lds0 = (unsigned)(unsigned long long)(lptr_t)smem;
    const int tw_sh = p.tw_sh;
    unsigned a_off[CH][TM];
#pragma unroll
    for (int q = 0; q < CH; ++q) {
        int kseg = q * 4 + fs, t0 = 0;
        if constexpr (CLS) {             // segment inside its class; the class's taps start at cls_first
            const int c = cls_of(q);
            kseg = (q - cls_lo(c)) * 4 + fs; t0 = m.cls_first[c];
            if (kseg >= m.cls_ntaps[c] * p.spr) kseg = 0;
        } else if (kseg >= p.nseg) kseg = 0;                          // K padding: the panel holds zeros there
        const int t = t0 + (kseg >> spt_sh), sl = kseg & (p.spr - 1);
        const int ddr = m.tdd[t] - p.dmin, dhr = m.tdh[t] - p.hmin, dwr = m.tdw[t] - p.wmin;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mv = wave * 32 + i * 16 + fr;
            const int th = mv >> tw_sh, tw = mv & (p.TW - 1);
            const int row = (ddr * p.IHt + th * p.she + dhr) * p.IWt + tw * p.swe + dwr;
            a_off[q][i] = lds0 + p.b_bytes + p.c_bytes + row * p.PX + x_swz(row, sl, p.spr) * 16;
        }
    }
    unsigned b_off[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) b_off[j] = lds0 + (j * 16 + fr) * 64 + b_swz(fr, fs) * 16;

    // ---- epilogue invariants.  The weights are the MFMA's A operand, so a lane ends up with FOUR CONSECUTIVE OUTPUT
    // CHANNELS (oc = j*16 + fs*4 + r) of ONE voxel (wave*32 + i*16 + fr): bias, rounding, the optional add, the statistics
    // and an 8-byte store all happen in registers -- no LDS tile, no block barrier in the epilogue ----
    int e_th[TM], e_row[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int mv = wave * 32 + i * 16 + fr;
        e_th[i] = mv >> tw_sh; e_row[i] = CLS ? 2 * (e_th[i] * m.OW + (mv & (p.TW - 1))) : e_th[i] * m.OW + (mv & (p.TW - 1));
    }
    bf16_t* o_base[TN]; int o_C[TN], o_nv[TN], o_acc[TN], o_fast[TN]; float bias_r[TN][4];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int oc = oc0 + j * 16 + fs * 4;
        int nv = m.OCn - oc; nv = nv > 4 ? 4 : (nv < 0 ? 0 : nv);
        o_base[j] = nullptr; o_C[j] = 0; o_acc[j] = 0; o_fast[j] = 0;
        if (nv > 0) {
            const OutRef o = m1_out_ref(m, oc);       // (4 channels never straddle two destination tensors: widths are multiples of 8)
            if (o.base) { o_base[j] = (bf16_t*)o.base + o.col; o_C[j] = o.C; o_acc[j] = o.acc; o_fast[j] = nv == 4 && (o.C & 3) == 0 && (o.col & 3) == 0; }
            else nv = 0;
        }
        o_nv[j] = nv;
#pragma unroll
        for (int r = 0; r < 4; ++r) bias_r[j][r] = (oc + r < m.OCn) ? m1_bias_at(m, oc + r) : 0.f;
    }
    // The bias values are global loads: consume them HERE.  Left to their first use in the tile loop the compiler puts an
    // s_waitcnt vmcnt(0) in front of the epilogue of every tile, which also drains the input DMA of the next tile (loads
    // retire in order) and serialises the pipeline.
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(bias_r[j][r]));
    // InstanceNorm-backward sums instead of statistics (MfmaP::ib_x: the output is d(a), a = lrelu(IN(x))): per lane gamma / beta of
    // its 4 channels in registers, {mean, rstd} of every sample in LDS behind the statistics fold rows (read per tile: the sample
    // changes along the walk), x fetched with the out += values ahead of the next tile's DMA
    const bool ib = !CLS && m.ib_x != nullptr;                        // (class mode: a data gradient without statistics of any kind)
    float* const ib_ms = red + NW * BN * 2;                           // [N][BN][2]
    float ib_g[TN][4], ib_b[TN][4];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int oc = oc0 + j * 16 + fs * 4 + r;
            ib_g[j][r] = (ib && oc < m.OCn) ? m.ib_gamma[oc] : 0.f; ib_b[j][r] = (ib && oc < m.OCn) ? m.ib_beta[oc] : 0.f;
        }
    if (ib) {
        for (int e = tid; e < m.N * BN; e += NTHR) {
            const int n = e / BN, c = e - n * BN, oc = oc0 + c < m.OCn ? oc0 + c : 0;
            ib_ms[e * 2] = m.ib_stats[((long long)n * m.OC + oc) * 2]; ib_ms[e * 2 + 1] = m.ib_stats[((long long)n * m.OC + oc) * 2 + 1];
        }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(ib_g[j][r]), "+v"(ib_b[j][r]));
    if (ib) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
    // running {sum, sum of squares} of the ROUNDED outputs of the current sample, per lane; folded over the block and written as
    // ONE partial per (sample, block) when the walk leaves the sample: [N][nsplit][OC][2], fixed order -> deterministic
    float ssum[TN][4], ssq[TN][4];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) { ssum[j][r] = 0.f; ssq[j][r] = 0.f; }
    int cur_n = 0;
    bool any_acc = false;
#pragma unroll
    for (int j = 0; j < TN; ++j) any_acc |= o_acc[j] != 0;
    any_acc = __builtin_amdgcn_readfirstlane(__any(any_acc)) != 0;
    const bool want_stats = !CLS && m.stat_partial != nullptr;
    auto flush = [&](int n) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s = ssum[j][r], q = ssq[j][r];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
                if (fr == 0) { red[(wave * BN + j * 16 + fs * 4 + r) * 2] = s; red[(wave * BN + j * 16 + fs * 4 + r) * 2 + 1] = q; }
                ssum[j][r] = 0.f; ssq[j][r] = 0.f;
            }
        __syncthreads();
        if (tid < BN && oc0 + tid < m.OCn) {
            float s = 0.f, q = 0.f;
            for (int w = 0; w < NW; ++w) { s += red[(w * BN + tid) * 2]; q += red[(w * BN + tid) * 2 + 1]; }
            float* dst = m.stat_partial + (((long long)n * p.nsplit + blockIdx.y) * m.OC + oc0 + tid) * 2;
            dst[0] = s; dst[1] = q;
        }
        __syncthreads();
    };

    // ---- pipeline over the tiles ----
    const int S = p.stages, npiece = nxit;
    int c_kt = blockIdx.y, c_tw = q_tw, c_th = q_th, c_od = q_od, c_n = q_n;       // the tile being computed
    for (int s = 0; s < S - 1; ++s) issue(s);
    int st = 0;
    for (; c_kt < p.ntiles; advance(c_kt, c_tw, c_th, c_od, c_n)) {
        if (want_stats) { for (; cur_n < c_n; ++cur_n) flush(cur_n); }
        if constexpr (RT) wait_vmh(npiece * (S - 2));
        else { if (S == 2) wait_vm_c<0>(); else if (S == 3) wait_vm_c<NXIT>(); else wait_vm_c<2 * NXIT>(); }
        __builtin_amdgcn_s_barrier();
        // out += launches: fetch what is there BEFORE the next tile's DMA is issued -- loads retire in order, so a load issued
        // behind the DMA could only be waited for together with it
        const int oh0 = c_th * p.TH;
        const int row0 = CLS ? ((c_n * m.OD + c_od) * m.OH + 2 * oh0) * m.OW + 2 * c_tw * p.TW
                             : ((c_n * m.OD + c_od) * m.OH + oh0) * m.OW + c_tw * p.TW;
        unsigned long long oldv[NC][TM][TN], xv[TM][TN];
        if (ib) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    xv[i][j] = 0ull;
                    if (o_fast[j] && oh0 + e_th[i] < p.QH)
                        xv[i][j] = gload8_untracked((const bf16_t*)m.ib_x + (long long)(row0 + e_row[i]) * m.OC + oc0 + j * 16 + fs * 4);
                }
        }
        if (any_acc) {
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    oldv[c][i][j] = 0ull;
                    if (o_acc[j] && o_fast[j] && oh0 + e_th[i] < p.QH)
                        oldv[c][i][j] = gload8_untracked(o_base[j] + (long long)(row0 + e_row[i] + (c >> 1) * m.OW + (c & 1)) * o_C[j]);
                }
        }
        int stn = st + S - 1; if (stn >= S) stn -= S;
        issue(stn);
        const unsigned sb = (unsigned)(st * p.x_bytes);

        f32x4_t acc[NC][TM][TN];
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[c][i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        u32x4_t af[2][TM], bf[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[0][i] = lds_read128h(a_off[0][i] + sb);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[0][j] = lds_read128h(b_off[j]);
#pragma unroll
        for (int q = 0; q < CH; ++q) {
            if (!RT || q < nchunks) {
                const int cur = q & 1;
                lds_waith(af[cur][0]);
#pragma unroll
                for (int i = 1; i < TM; ++i) lds_tieh(af[cur][i]);
#pragma unroll
                for (int j = 0; j < TN; ++j) lds_tieh(bf[cur][j]);
                if (q + 1 < CH && (!RT || q + 1 < nchunks)) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) af[cur ^ 1][i] = lds_read128h(a_off[q + 1 < CH ? q + 1 : q][i] + sb);
#pragma unroll
                    for (int j = 0; j < TN; ++j) bf[cur ^ 1][j] = lds_read128h(b_off[j] + (q + 1) * BN * 64);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)      // D[oc][voxel]: weights as A, voxels as B
                        acc[cls_of(q)][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, bf[cur][j]),
                                                                            __builtin_bit_cast(bf16x8_t, af[cur][i]), acc[cls_of(q)][i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        // ---- epilogue in registers ----
        if (any_acc || ib) {      // the fetched values: everything but the DMA pieces issued behind them has landed
            if constexpr (RT) wait_vm_c<0>(); else wait_vm_c<NXIT>();
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (any_acc) {
#pragma unroll
                        for (int c = 0; c < NC; ++c) asm volatile("" : "+v"(oldv[c][i][j]));
                    }
                    if (ib) asm volatile("" : "+v"(xv[i][j]));
                }
        }
#pragma unroll
        for (int c = 0; c < NC; ++c)                      // class c = (h parity, w parity): output voxel (2 qh + c/2, 2 qw + c%2)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if (oh0 + e_th[i] >= p.QH) continue;
            const long long orow = row0 + e_row[i] + (CLS ? (c >> 1) * m.OW + (c & 1) : 0);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (o_nv[j] == 0) continue;
                bf16_t* dst = o_base[j] + orow * o_C[j];
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = acc[c][i][j][r] + bias_r[j][r];
                if (o_fast[j]) {
                    if (o_acc[j]) {           // out += : an earlier launch wrote the other concat members' share
                        const uint2 ov = make_uint2((unsigned)oldv[c][i][j], (unsigned)(oldv[c][i][j] >> 32));
                        const float b[4] = {__uint_as_float(ov.x << 16), __uint_as_float(ov.x & 0xffff0000u),
                                            __uint_as_float(ov.y << 16), __uint_as_float(ov.y & 0xffff0000u)};
                        const unsigned r01 = cvt_pk_bf16(v[0], v[1]), r23 = cvt_pk_bf16(v[2], v[3]);
                        v[0] = __uint_as_float(r01 << 16) + b[0]; v[1] = __uint_as_float(r01 & 0xffff0000u) + b[1];
                        v[2] = __uint_as_float(r23 << 16) + b[2]; v[3] = __uint_as_float(r23 & 0xffff0000u) + b[3];
                    }
                    uint2 o;
                    o.x = cvt_pk_bf16(v[0], v[1]); o.y = cvt_pk_bf16(v[2], v[3]);
                    *reinterpret_cast<uint2*>(dst) = o;
                    if (want_stats) {
                        const float r0 = __uint_as_float(o.x << 16), r1 = __uint_as_float(o.x & 0xffff0000u);
                        const float r2 = __uint_as_float(o.y << 16), r3 = __uint_as_float(o.y & 0xffff0000u);
                        if (ib) {             // {sum dy, sum dy * xh}, dy = d(a) * lrelu'(gamma xh + beta)
                            const float4 ms0 = *reinterpret_cast<const float4*>(ib_ms + ((c_n * BN + j * 16 + fs * 4) * 2));
                            const float4 ms1 = *reinterpret_cast<const float4*>(ib_ms + ((c_n * BN + j * 16 + fs * 4) * 2) + 4);
                            const uint2 xw = make_uint2((unsigned)xv[i][j], (unsigned)(xv[i][j] >> 32));
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
        if (++st == S) st = 0;
    }
    if (want_stats) { for (; cur_n < m.N; ++cur_n) flush(cur_n); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
#undef cls_of
#undef cls_lo

// ------------------------------------------------------------------------------------------------
// parity-class mode: the class boundaries are compile-time -- the shapes of the res0 <-> res1 transitions, both tap orders
typedef void (*halo_kern_t)(HaloP);
static halo_kern_t halo_cls_kernel(const MfmaP& mp, const HaloP& p) {
    const int nxit = p.x_slots / p.nthr;
    const int b1 = mp.cls_kpad[0] / 32, b2 = b1 + mp.cls_kpad[1] / 32, b3 = b2 + mp.cls_kpad[2] / 32;
    halo_kern_t kern = nullptr;
#define HC(NCH_, NXIT_, B1_, B2_, B3_) if (!kern && p.nchunks == NCH_ && nxit == NXIT_ && b1 == B1_ && b2 == B2_ && b3 == B3_) \
        kern = p.BNh == 32 ? conv_halo_kernel<2, 512, NCH_, NXIT_, B1_, B2_, B3_> : conv_halo_kernel<1, 512, NCH_, NXIT_, B1_, B2_, B3_>;
    HC(18, 5, 8, 12, 16) HC(18, 5, 2, 6, 10)           // 64 gradient channels: 4 / 2 / 2 / 1 taps (or reversed) x 2 chunks
    HC(5, 2, 2, 3, 4) HC(5, 2, 1, 2, 3)                // 16 gradient channels: a tap is half a chunk
    HC(9, 3, 4, 6, 8) HC(9, 3, 1, 3, 5)                // 32 gradient channels
#undef HC
    return kern;
}
static bool halo_plan_nthr(const MfmaP& m, int OCpad, HaloP& p, int force_nthr);
// 256-voxel tiles on 512 threads; when those do not fit (a stride-2 layer's input tile is 4x its output tile: 32 -> 64 k133 s122 at
// res0 needs 74 KB per stage) 128-voxel tiles on 256 threads
static bool halo_plan(const MfmaP& m, int OCpad, HaloP& p) {
    if (halo_plan_nthr(m, OCpad, p, 0)) return true;
    return halo_plan_nthr(m, OCpad, p, 256);
}
static bool halo_plan_nthr(const MfmaP& m, int OCpad, HaloP& p, int force_nthr) {
    if (m.ksplit != 1) return false;
    // four (h, w) parity classes of a (1,2,2)-strided 1x3x3 layer's data gradient / transposed forward: stride-1 problems in the q-domain
    const bool cls4 = m.mode == 1 && m.nclasses == 4 && m.sd == 1 && m.sh == 2 && m.sw == 2 && m.OH % 2 == 0 && m.OW % 2 == 0 &&
                      m.IH == m.OH / 2 && m.IW == m.OW / 2 && m.ID == m.OD && !m.stat_partial && !m.ib_x;
    { int hc = M1_CFG("M1_HALO_CLASSES", 1); if (cls4 && !hc) return false; }
    if (m.nclasses != 1 && !cls4) return false;
    if (!(m.mode == 0 || (m.mode == 1 && m.sd == 1 && m.sh == 1 && m.sw == 1) || cls4)) return false;
    { int hs = M1_CFG("M1_HALO_STRIDED", 1);       // 0: strided layers stay on conv_mfma (A/B switch)
      if (!hs && m.mode == 0 && (m.sd > 1 || m.sh > 1 || m.sw > 1)) return false; }
    int CC = 0;
    for (int i = 0; i < m.nsrc; ++i) { if (m.srcC[i] % 8) return false; CC += m.srcC[i]; }
    if (!(CC == 8 || CC == 16 || CC == 32 || CC == 64)) return false;
    const int QH = cls4 ? m.OH / 2 : m.OH, QW = cls4 ? m.OW / 2 : m.OW;
    if (QW % 8) return false;
    if ((long long)m.N * m.ID * m.IH * m.IW >= (1ll << 31) - (1 << 20) || (long long)m.N * m.OD * m.OH * m.OW >= (1ll << 31) - (1 << 20)) return false;
    int nt = m.cls_ntaps[0];
    if (cls4) { nt = 0; for (int c = 0; c < 4; ++c) { if (m.cls_first[c] != nt) return false; nt += m.cls_ntaps[c]; } }
    if (nt < 2 || m.cls_first[0] != 0) return false;                // (a 1x1x1 conv has no halo to share)
    p = HaloP{}; p.m = m; p.ncls = cls4 ? 4 : 1; p.QH = QH; p.QW = QW;
    int dmin = 127, dmax = -127, hmin = 127, hmax = -127, wmin = 127, wmax = -127;
    for (int t = 0; t < nt; ++t) {
        dmin = m.tdd[t] < dmin ? m.tdd[t] : dmin; dmax = m.tdd[t] > dmax ? m.tdd[t] : dmax;
        hmin = m.tdh[t] < hmin ? m.tdh[t] : hmin; hmax = m.tdh[t] > hmax ? m.tdh[t] : hmax;
        wmin = m.tdw[t] < wmin ? m.tdw[t] : wmin; wmax = m.tdw[t] > wmax ? m.tdw[t] : wmax;
    }
    p.dmin = dmin; p.hmin = hmin; p.wmin = wmin;
    if (m.mode == 0) { p.sde = m.sd; p.she = m.sh; p.swe = m.sw; p.pde = m.pd; p.phe = m.ph; p.pwe = m.pw; }
    else { p.sde = p.she = p.swe = 1; p.pde = p.phe = p.pwe = 0; }
    { int nt_ = M1_CFG("M1_HALO_THREADS", 512); p.nthr = nt_ == 512 ? 512 : 256; }
    if (force_nthr) p.nthr = force_nthr;
    p.TW = QW % 32 == 0 ? 32 : (QW % 16 == 0 ? 16 : 8);
    if (p.nthr == 512) {      // 256-voxel tiles unless their row padding wastes clearly more than 128-voxel tiles would
        const int th5 = 256 / p.TW, th2 = 128 / p.TW;
        const double e5 = (double)QH / ((QH + th5 - 1) / th5 * th5), e2 = (double)QH / ((QH + th2 - 1) / th2 * th2);
        if (e5 < 0.9 * e2) p.nthr = 256;
    }
    if (cls4 && p.nthr != 512) return false;
    const int BMh = p.nthr / 2;
    p.TH = BMh / p.TW;
    p.tiles_w = QW / p.TW; p.tiles_h = (QH + p.TH - 1) / p.TH;
    p.tiles_per_sample = m.OD * p.tiles_h * p.tiles_w;
    const long long nt_all = (long long)m.N * p.tiles_per_sample;
    if (nt_all >= (1ll << 30)) return false;
    p.ntiles = (int)nt_all;
    p.KDs = dmax - dmin + 1; p.IHt = (p.TH - 1) * p.she + (hmax - hmin + 1); p.IWt = (p.TW - 1) * p.swe + (wmax - wmin + 1);
    if (p.KDs > 255 || p.IHt > 255 || p.IWt > 255) return false;
    p.PX = CC * 2; p.spr = CC / 8;
    p.spr_sh = p.spr == 8 ? 3 : (p.spr == 4 ? 2 : (p.spr == 2 ? 1 : 0)); p.tw_sh = p.TW == 32 ? 5 : (p.TW == 16 ? 4 : 3);
    const int x_rows = p.KDs * p.IHt * p.IWt;
    p.x_slots = (x_rows * p.spr + p.nthr - 1) / p.nthr * p.nthr;
    if (p.x_slots > HL_MAX_XIT * p.nthr) return false;
    p.x_bytes = p.x_slots * 16 + 256;                                // (+ slack: fragment reads of K-padding segments stay inside)
    p.kpad = m.cls_kpad[0]; p.nchunks = p.kpad / 32; p.nseg = nt * p.spr;
    if (cls4) { p.kpad = 0; for (int c = 0; c < 4; ++c) p.kpad += m.cls_kpad[c]; p.nchunks = p.kpad / 32; }
    if (p.nchunks > HL_MAX_CH) return false;
    p.BNh = (m.OCn > 16 && OCpad >= 32) ? 32 : 16;
    if (OCpad % p.BNh) return false;
    p.b_bytes = p.nchunks * p.BNh * 64;
    p.c_bytes = (p.nthr / 64) * p.BNh * 2 * 4;                        // statistics fold, one row per wave
    if (m.ib_x) p.c_bytes += m.N * p.BNh * 2 * 4;                     // + {mean, rstd} of the block's channels, every sample (InstanceNorm-backward sums)
    const int fixed = p.b_bytes + p.c_bytes;
    int kb = M1_CFG("M1_HALO_LDS_KB", 160);
    int S = (kb * 1024 - fixed) / p.x_bytes;
    if (S > 4) S = 4;
    { int fs = M1_CFG("M1_HALO_STAGES", 0); if (fs >= 2 && fs < S) S = fs; }
    while (S > 2 && (p.x_slots / p.nthr) * (S - 2) > 40) --S;
    if (S < 2) return false;
    p.stages = S;
    if (cls4 && !halo_cls_kernel(m, p)) return false;
    return true;
}

// blocks along y: every block walks tiles blockIdx.y, + nsplit, ... (neighbouring tiles run at the same time and share halos in L2)
static int halo_nsplit(const HaloP& p, int OCpad) {
    const int slices = OCpad / p.BNh;
    int nsplit = 256 / slices; if (nsplit < 1) nsplit = 1;
    { int tg = M1_CFG("M1_HALO_BLOCKS", 0); if (tg > 0) nsplit = tg / slices > 0 ? tg / slices : 1; }
    if (nsplit > p.ntiles) nsplit = p.ntiles;
    return nsplit;
}

bool m1_halo_conv_supported(const MfmaP& mp, int OCpad) { HaloP p; return halo_plan(mp, OCpad, p); }
// statistics partials per sample that the kernel writes (MfmaP::stat_tiles): one per block row
int m1_halo_conv_stat_parts(const MfmaP& mp, int OCpad) { HaloP p; return halo_plan(mp, OCpad, p) ? halo_nsplit(p, OCpad) : 0; }

int m1_halo_conv(const MfmaP& mp, int OCpad, hipStream_t st) {
    HaloP p;
    if (!halo_plan(mp, OCpad, p)) return M1_ERR_UNSUPPORTED;
    const int slices = OCpad / p.BNh;
    const int nsplit = halo_nsplit(p, OCpad);
    p.nsplit = nsplit;
    if (mp.stat_partial && mp.stat_tiles != nsplit) return M1_ERR_BAD_ARG;
    const size_t smem = (size_t)p.b_bytes + p.c_bytes + (size_t)p.stages * p.x_bytes;
    // compile-time (K chunks, DMA pieces per thread) for the layer shapes of the M1 configurations; anything else: run-time counts
    const int nxit = p.x_slots / p.nthr;
    void (*kern)(HaloP) = nullptr;
    if (p.ncls == 4) {
        kern = halo_cls_kernel(mp, p);
        if (!kern) return M1_ERR_UNSUPPORTED;
    } else
    if (p.nthr == 512) {
#define HK(NCH_, NXIT_) if (!kern && p.nchunks == NCH_ && nxit == NXIT_) kern = p.BNh == 32 ? conv_halo_kernel<2, 512, NCH_, NXIT_> : conv_halo_kernel<1, 512, NCH_, NXIT_>;
        HK(18, 6) HK(9, 3) HK(5, 2) HK(3, 1) HK(14, 4) HK(7, 2)
#undef HK
    } else {
#define HK(NCH_, NXIT_) if (!kern && p.nchunks == NCH_ && nxit == NXIT_) kern = p.BNh == 32 ? conv_halo_kernel<2, 256, NCH_, NXIT_> : conv_halo_kernel<1, 256, NCH_, NXIT_>;
        HK(9, 9)                       // 32 -> 64 / 16, k133, s122 at res0 -> res1 (the strided SE blocks' conv1 / conv4)
#undef HK
    }
    { int lg = M1_CFG("M1_HALO_LOG", 0);
      if (lg) fprintf(stderr, "halo: nthr %d BN %d nchunks %d nxit %d stages %d TW %d TH %d %s\n", p.nthr, p.BNh, p.nchunks, nxit, p.stages, p.TW, p.TH, kern ? "static" : "runtime"); }
    if (!kern) kern = p.nthr == 512 ? (p.BNh == 32 ? conv_halo_kernel<2, 512, 0, 0> : conv_halo_kernel<1, 512, 0, 0>)
                                    : (p.BNh == 32 ? conv_halo_kernel<2, 256, 0, 0> : conv_halo_kernel<1, 256, 0, 0>);
    {
        static const void* done[24]; static int ndone = 0;
        bool seen = false;
        for (int q = 0; q < ndone; ++q) seen |= done[q] == (const void*)kern;
        if (!seen) {
            if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return M1_ERR_LAUNCH;
            if (ndone < 24) done[ndone++] = (const void*)kern;
        }
    }
    m1_note_kernel(p.ncls == 4 ? "conv_halo_cls:bn%d" : "conv_halo:bn%d", p.BNh);
    hipLaunchKernelGGL(kern, dim3(slices, nsplit), dim3(p.nthr), smem, st, p);
    return m1_check_launch();
}
