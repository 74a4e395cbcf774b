// conv_halo.hip -- halo-tile implicit-GEMM convolution (bf16) for the wide, shallow layers (res0/res1: hundreds of
// thousands of voxels, <= 64 contraction channels): the case where conv_mfma.hip's per-tap im2col gather re-reads every
// input voxel once per tap from L2 and is bound by the L2 -> LDS path.
//
//   out[o][oc] = bias[oc] + sum_t sum_c X[o*s - p + off_t][c] * Wp[oc][t*CC + c]       (same packed panel as conv_mfma)
//
// * A block owns one <= 32-wide slice of the output channels and keeps that slice of the weight panel RESIDENT in LDS
//   ([chunk][oc][64 B], swizzled) while it walks 128-voxel output tiles (TH rows x TW columns of one (n, d) slice).
// * Per tile the input voxels of the tile INCLUDING the stencil halo are staged once, by LDS-DMA
//   (global_load_lds_dwordx4), as voxel-major rows holding the whole channel concat; a tap is a row offset into
//   that tile, so every input voxel crosses L2 -> LDS once per tile instead of once per tap.
// * K order is [tap][member][channel] as in the panel; the per-lane fragment offsets of all K chunks are tile
//   invariant and live in registers; fragment reads are ds_read_b128 by inline asm, one chunk ahead of the MFMAs,
//   the tiles themselves run in a multi-stage DMA pipeline with counted vmcnt waits (as wgrad_tf.hip).
// * Epilogue as conv_mfma: bias, bf16 rounding, tile staged in LDS, 16-byte coalesced stores, fused InstanceNorm
//   statistics partials.
#include "conv_mfma.h"
#include "reduce.h"
#include <stdlib.h>

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

#define HL_MAX_XIT 8         // LDS-DMA pieces per thread for one input tile
#define HL_MAX_CH 20         // K chunks (of 32) whose fragment offsets live in registers

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
};

// physical 16-byte slot of logical slot `sl` in tile row `row` (conflict-free ds_read_b128 of 16 consecutive rows)
__device__ __forceinline__ int x_swz(int row, int sl, int spr) {
    return spr == 8 ? (sl ^ (row & 7)) : (spr == 4 ? (sl ^ ((row >> 1) & 3)) : sl);
}
__device__ __forceinline__ int b_swz(int row, int seg) { return seg ^ ((-(row >> 2)) & 3); }

template <int TN, int NTHR>
__global__ void __launch_bounds__(NTHR) conv_halo_kernel(HaloP p) {
    constexpr int TM = 2, BN = TN * 16, SEG = 8, CP = BN + SEG, HL_BM = NTHR / 2;     // each wave owns 32 voxels x BN channels
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const Bs = smem;                                   // [nchunks][BN][64]
    bf16_t* const C_s = reinterpret_cast<bf16_t*>(smem + p.b_bytes);  // [128][CP]
    float* const red = reinterpret_cast<float*>(smem + p.b_bytes + HL_BM * CP * 2);   // [NTHR][2]
    unsigned char* const Xs0 = smem + p.b_bytes + p.c_bytes;          // [stages][x_bytes]
    const MfmaP& m = p.m;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int oc0 = blockIdx.x * BN;
    const unsigned char* zero_pg = reinterpret_cast<const unsigned char*>(m1_zero_page_h);
    const int fr = lane & 15, fs = lane >> 4;
    const int spt = p.spr;                                            // 16-byte K segments per tap (= slots per row)

    // ---- weight panel slice -> LDS, once ----
    {
        const bf16_t* wp = (const bf16_t*)m.wp + m.cls_woff[0];
        const int nslot = p.nchunks * BN * 4;
        for (int q0 = 0; q0 < nslot; q0 += NTHR) {
            const int q = q0 + tid;                                   // LDS slot: chunk, oc row, physical 16-byte slot
            const int ch = q / (BN * 4), r = (q / 4) % BN, s = b_swz(r, q & 3);
            const unsigned char* src = (q < nslot) ? reinterpret_cast<const unsigned char*>(wp + (long long)(oc0 + r) * p.kpad + (ch * 4 + s) * SEG)
                                                   : zero_pg;
            if (q0 + wave * 64 < nslot) glds16h(src, Bs + (q0 + wave * 64) * 16);
        }
    }

    // ---- per-lane description of its input-tile DMA pieces ----
    const bf16_t* x_base[HL_MAX_XIT]; int x_C[HL_MAX_XIT], x_rel[HL_MAX_XIT], x_pk[HL_MAX_XIT], x_co[HL_MAX_XIT];
    const int nxit = p.x_slots / NTHR;
    const int x_rows = p.KDs * p.IHt * p.IWt;
#pragma unroll
    for (int it = 0; it < HL_MAX_XIT; ++it) {
        const int q = it * NTHR + tid;
        const int row = q / p.spr, slp = q - row * p.spr;
        const int sl = x_swz(row, slp, p.spr);                        // XOR swizzles are involutions
        const int dd = row / (p.IHt * p.IWt); const int r2 = row - dd * (p.IHt * p.IWt);
        const int hh = r2 / p.IWt, ww = r2 - hh * p.IWt;
        int c = sl * SEG, s = 0;                                      // channel on the concat axis -> member
        while (s < m.nsrc - 1 && c >= m.srcC[s]) { c -= m.srcC[s]; ++s; }
        x_base[it] = (const bf16_t*)m.src[s]; x_C[it] = m.srcC[s]; x_co[it] = c;
        x_rel[it] = (dd * m.IH + hh) * m.IW + ww;
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
        const bool live = q_kt < p.ntiles;
        const int id0 = q_od * p.sde - p.pde + p.dmin, ih0 = q_th * p.TH * p.she - p.phe + p.hmin, iw0 = q_tw * p.TW * p.swe - p.pwe + p.wmin;
        const int lin0 = ((q_n * m.ID + id0) * m.IH + ih0) * m.IW + iw0;
        unsigned char* Xs = Xs0 + st * p.x_bytes;
#pragma unroll
        for (int it = 0; it < HL_MAX_XIT; ++it) {
            if (it < nxit) {
                const int pk = x_pk[it];
                const int dd = pk & 0xff, hh = (pk >> 8) & 0xff, ww = (pk >> 16) & 0xff;
                const bool ok = live && pk >= 0 && (unsigned)(id0 + dd) < (unsigned)m.ID && (unsigned)(ih0 + hh) < (unsigned)m.IH &&
                                (unsigned)(iw0 + ww) < (unsigned)m.IW;
                // branch-free select (the compiler turns the ?: into a branch around the 64-bit address arithmetic)
                const long long real = (long long)(x_base[it] + (long long)(lin0 + x_rel[it]) * x_C[it] + x_co[it]);
                const long long zp = (long long)zero_pg;
                const unsigned char* src = reinterpret_cast<const unsigned char*>(zp + ((real - zp) & -(long long)ok));
                glds16h(src, Xs + (it * NTHR + wave * 64) * 16);
            }
        }
        advance(q_kt, q_tw, q_th, q_od, q_n);
    };

    // ---- fragment offsets (tile invariant): A per (chunk, 16-row tile), B per 16-column tile ----
    const unsigned lds0 = (unsigned)(unsigned long long)(lptr_t)smem;
    unsigned a_off[HL_MAX_CH][TM];
#pragma unroll
    for (int q = 0; q < HL_MAX_CH; ++q) {
        int kseg = q * 4 + fs; if (kseg >= p.nseg) kseg = 0;          // K padding: the panel holds zeros there
        const int t = kseg / spt, sl = kseg - t * spt;
        const int ddr = m.tdd[t] - p.dmin, dhr = m.tdh[t] - p.hmin, dwr = m.tdw[t] - p.wmin;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mv = wave * 32 + i * 16 + fr;
            const int th = mv / p.TW, tw = mv - th * p.TW;
            const int row = (ddr * p.IHt + th * p.she + dhr) * p.IWt + tw * p.swe + dwr;
            a_off[q][i] = lds0 + p.b_bytes + p.c_bytes + row * p.PX + x_swz(row, sl, p.spr) * 16;
        }
    }
    unsigned b_off[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) b_off[j] = lds0 + (j * 16 + fr) * 64 + b_swz(fr, fs) * 16;

    // ---- pipeline over the tiles ----
    const int S = p.stages, npiece = nxit;
    int c_kt = blockIdx.y, c_tw = q_tw, c_th = q_th, c_od = q_od, c_n = q_n;       // the tile being computed
    for (int s = 0; s < S - 1; ++s) issue(s);
    int st = 0;
    for (; c_kt < p.ntiles; advance(c_kt, c_tw, c_th, c_od, c_n)) {
        wait_vmh(npiece * (S - 2));
        __builtin_amdgcn_s_barrier();
        int stn = st + S - 1; if (stn >= S) stn -= S;
        issue(stn);
        const unsigned sb = (unsigned)(st * p.x_bytes);

        f32x4_t acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        u32x4_t af[2][TM], bf[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[0][i] = lds_read128h(a_off[0][i] + sb);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[0][j] = lds_read128h(b_off[j]);
#pragma unroll
        for (int q = 0; q < HL_MAX_CH; ++q) {
            if (q < p.nchunks) {
                const int cur = q & 1;
                lds_waith(af[cur][0]);
#pragma unroll
                for (int i = 1; i < TM; ++i) lds_tieh(af[cur][i]);
#pragma unroll
                for (int j = 0; j < TN; ++j) lds_tieh(bf[cur][j]);
                if (q + 1 < HL_MAX_CH && q + 1 < p.nchunks) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) af[cur ^ 1][i] = lds_read128h(a_off[q + 1 < HL_MAX_CH ? q + 1 : q][i] + sb);
#pragma unroll
                    for (int j = 0; j < TN; ++j) bf[cur ^ 1][j] = lds_read128h(b_off[j] + (q + 1) * BN * 64);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af[cur][i]),
                                                                            __builtin_bit_cast(bf16x8_t, bf[cur][j]), acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        // ---- epilogue: acc (+bias) -> bf16 -> LDS tile -> 16-byte stores (+ statistics partials) ----
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = j * 16 + fr;
                const float bv = (oc0 + col < m.OCn) ? m1_bias_at(m, oc0 + col) : 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = wave * 32 + i * 16 + fs * 4 + r;
                    Act<bf16_t>::st(C_s + row * CP + col, acc[i][j][r] + bv);
                }
            }
        __syncthreads();
        const int oh0 = c_th * p.TH, ow0 = c_tw * p.TW;
        const long long slice0 = ((long long)c_n * m.OD + c_od) * m.OH;
        bool any_acc = m.accumulate != 0;
        for (int q = 0; q < m.nout; ++q) any_acc |= m.outAcc[q] != 0;
        if (any_acc) {            // out += : fold what is there into the tile first, so that the statistics below (a conv run as
            // one launch per member group: the last group owns them) and the stores see the sum
            constexpr int SPRa = BN / SEG;
            for (int e = tid; e < HL_BM * SPRa; e += NTHR) {
                const int row = e / SPRa, cs = e % SPRa;
                const int th = row / p.TW, tw = row - th * p.TW;
                const int oc = oc0 + cs * SEG;
                if (oh0 + th >= m.OH || oc >= m.OCn) continue;
                const OutRef o = m1_out_ref(m, oc);
                if (!o.base || !o.acc) continue;
                const long long orow = (slice0 + oh0 + th) * m.OW + ow0 + tw;
                const bf16_t* src = (const bf16_t*)o.base + orow * o.C + o.col;
                bf16_t* ct = C_s + row * CP + cs * SEG;
                if (o.C % SEG != 0 || (m.nout == 0 && oc + SEG > m.OCn)) {
                    for (int k = 0; k < SEG && oc + k < m.OCn; ++k) Act<bf16_t>::st(ct + k, Act<bf16_t>::ld(ct + k) + Act<bf16_t>::ld(src + k));
                } else {
                    float a[SEG], b[SEG];
                    VecIO<bf16_t, SEG>::ld(ct, a); VecIO<bf16_t, SEG>::ld(src, b);
#pragma unroll
                    for (int k = 0; k < SEG; ++k) a[k] += b[k];
                    VecIO<bf16_t, SEG>::st(ct, a);
                }
            }
            __syncthreads();
        }
        if (m.stat_partial) {
            constexpr int G = NTHR / BN;
            const int col = tid % BN, rg = tid / BN;
            float s = 0.f, ss = 0.f;
            for (int row = rg; row < HL_BM; row += G) {
                if (oh0 + row / p.TW < m.OH) { const float v = Act<bf16_t>::ld(C_s + row * CP + col); s += v; ss += v * v; }
            }
            red[tid * 2] = s; red[tid * 2 + 1] = ss;
            __syncthreads();
            if (rg == 0 && oc0 + col < m.OCn) {
                for (int q = 1; q < G; ++q) { s += red[(q * BN + col) * 2]; ss += red[(q * BN + col) * 2 + 1]; }
                const long long tile = (long long)c_n * p.tiles_per_sample + ((long long)c_od * p.tiles_h + c_th) * p.tiles_w + c_tw;
                float* dst = m.stat_partial + ((tile * m.OC) + oc0 + col) * 2;
                dst[0] = s; dst[1] = ss;
            }
        }
        constexpr int SPR = BN / SEG;
        for (int e = tid; e < HL_BM * SPR; e += NTHR) {
            const int row = e / SPR, cs = e % SPR;
            const int th = row / p.TW, tw = row - th * p.TW;
            const int oc = oc0 + cs * SEG;
            if (oh0 + th >= m.OH || oc >= m.OCn) continue;
            const OutRef o = m1_out_ref(m, oc);
            if (!o.base) continue;
            const long long orow = (slice0 + oh0 + th) * m.OW + ow0 + tw;
            const uint4 v = *reinterpret_cast<const uint4*>(C_s + row * CP + cs * SEG);
            bf16_t* dst = (bf16_t*)o.base + orow * o.C + o.col;
            if (o.C % SEG != 0 || (m.nout == 0 && oc + SEG > m.OCn)) {
                const bf16_t* ve = reinterpret_cast<const bf16_t*>(&v);
                for (int k = 0; k < SEG && oc + k < m.OCn; ++k) dst[k] = ve[k];
            } else {
                *reinterpret_cast<uint4*>(dst) = v;
            }
        }
        if (++st == S) st = 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ------------------------------------------------------------------------------------------------
static bool halo_plan(const MfmaP& m, int OCpad, HaloP& p) {
    if (m.nclasses != 1 || m.ksplit != 1) return false;
    if (!(m.mode == 0 || (m.mode == 1 && m.sd == 1 && m.sh == 1 && m.sw == 1))) return false;
    int CC = 0;
    for (int i = 0; i < m.nsrc; ++i) { if (m.srcC[i] % 8) return false; CC += m.srcC[i]; }
    if (!(CC == 8 || CC == 16 || CC == 32 || CC == 64)) return false;
    if (m.OW % 8) return false;
    if ((long long)m.N * m.ID * m.IH * m.IW >= (1ll << 31) - (1 << 20) || (long long)m.N * m.OD * m.OH * m.OW >= (1ll << 31) - (1 << 20)) return false;
    const int nt = m.cls_ntaps[0];
    if (nt < 2 || m.cls_first[0] != 0) return false;                // (a 1x1x1 conv has no halo to share)
    p = HaloP{}; p.m = m;
    int dmin = 127, dmax = -127, hmin = 127, hmax = -127, wmin = 127, wmax = -127;
    for (int t = 0; t < nt; ++t) {
        dmin = m.tdd[t] < dmin ? m.tdd[t] : dmin; dmax = m.tdd[t] > dmax ? m.tdd[t] : dmax;
        hmin = m.tdh[t] < hmin ? m.tdh[t] : hmin; hmax = m.tdh[t] > hmax ? m.tdh[t] : hmax;
        wmin = m.tdw[t] < wmin ? m.tdw[t] : wmin; wmax = m.tdw[t] > wmax ? m.tdw[t] : wmax;
    }
    p.dmin = dmin; p.hmin = hmin; p.wmin = wmin;
    if (m.mode == 0) { p.sde = m.sd; p.she = m.sh; p.swe = m.sw; p.pde = m.pd; p.phe = m.ph; p.pwe = m.pw; }
    else { p.sde = p.she = p.swe = 1; p.pde = p.phe = p.pwe = 0; }
    { static int nt_ = -1; if (nt_ < 0) { const char* e = getenv("M1_HALO_THREADS"); nt_ = e ? atoi(e) : 512; } p.nthr = nt_ == 512 ? 512 : 256; }
    p.TW = m.OW % 32 == 0 ? 32 : (m.OW % 16 == 0 ? 16 : 8);
    if (p.nthr == 512) {      // 256-voxel tiles unless their row padding wastes clearly more than 128-voxel tiles would
        const int th5 = 256 / p.TW, th2 = 128 / p.TW;
        const double e5 = (double)m.OH / ((m.OH + th5 - 1) / th5 * th5), e2 = (double)m.OH / ((m.OH + th2 - 1) / th2 * th2);
        if (e5 < 0.9 * e2) p.nthr = 256;
    }
    const int BMh = p.nthr / 2;
    p.TH = BMh / p.TW;
    p.tiles_w = m.OW / p.TW; p.tiles_h = (m.OH + p.TH - 1) / p.TH;
    p.tiles_per_sample = m.OD * p.tiles_h * p.tiles_w;
    const long long nt_all = (long long)m.N * p.tiles_per_sample;
    if (nt_all >= (1ll << 30)) return false;
    p.ntiles = (int)nt_all;
    p.KDs = dmax - dmin + 1; p.IHt = (p.TH - 1) * p.she + (hmax - hmin + 1); p.IWt = (p.TW - 1) * p.swe + (wmax - wmin + 1);
    if (p.KDs > 255 || p.IHt > 255 || p.IWt > 255) return false;
    p.PX = CC * 2; p.spr = CC / 8;
    const int x_rows = p.KDs * p.IHt * p.IWt;
    p.x_slots = (x_rows * p.spr + p.nthr - 1) / p.nthr * p.nthr;
    if (p.x_slots > HL_MAX_XIT * p.nthr) return false;
    p.x_bytes = p.x_slots * 16 + 256;                                // (+ slack: fragment reads of K-padding segments stay inside)
    p.kpad = m.cls_kpad[0]; p.nchunks = p.kpad / 32; p.nseg = nt * p.spr;
    if (p.nchunks > HL_MAX_CH) return false;
    p.BNh = (m.OCn > 16 && OCpad >= 32) ? 32 : 16;
    if (OCpad % p.BNh) return false;
    p.b_bytes = p.nchunks * p.BNh * 64;
    p.c_bytes = BMh * (p.BNh + 8) * 2 + p.nthr * 2 * 4;
    const int fixed = p.b_bytes + p.c_bytes;
    static int kb = -1; if (kb < 0) { const char* e = getenv("M1_HALO_LDS_KB"); kb = e ? atoi(e) : 160; }
    int S = (kb * 1024 - fixed) / p.x_bytes;
    if (S > 4) S = 4;
    { static int fs = -1; if (fs < 0) { const char* e = getenv("M1_HALO_STAGES"); fs = e ? atoi(e) : 0; } if (fs >= 2 && fs < S) S = fs; }
    while (S > 2 && (p.x_slots / p.nthr) * (S - 2) > 40) --S;
    if (S < 2) return false;
    p.stages = S;
    return true;
}

bool m1_halo_conv_supported(const MfmaP& mp, int OCpad) { HaloP p; return halo_plan(mp, OCpad, p); }
int m1_halo_conv_tiles_per_sample(const MfmaP& mp) { HaloP p; return halo_plan(mp, 32, p) ? p.tiles_per_sample : 0; }

int m1_halo_conv(const MfmaP& mp, int OCpad, hipStream_t st) {
    HaloP p;
    if (!halo_plan(mp, OCpad, p)) return M1_ERR_UNSUPPORTED;
    const int slices = OCpad / p.BNh;
    int nsplit = 256 / slices; if (nsplit < 1) nsplit = 1;
    { static int tg = -1; if (tg < 0) { const char* e = getenv("M1_HALO_BLOCKS"); tg = e ? atoi(e) : 0; } if (tg > 0) nsplit = tg / slices > 0 ? tg / slices : 1; }
    if (nsplit > p.ntiles) nsplit = p.ntiles;
    p.nsplit = nsplit;
    const size_t smem = (size_t)p.b_bytes + p.c_bytes + (size_t)p.stages * p.x_bytes;
    void (*kern)(HaloP) = p.nthr == 512 ? (p.BNh == 32 ? conv_halo_kernel<2, 512> : conv_halo_kernel<1, 512>)
                                        : (p.BNh == 32 ? conv_halo_kernel<2, 256> : conv_halo_kernel<1, 256>);
    {
        static const void* done[4]; static int ndone = 0;
        bool seen = false;
        for (int q = 0; q < ndone; ++q) seen |= done[q] == (const void*)kern;
        if (!seen) {
            if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return M1_ERR_LAUNCH;
            if (ndone < 4) done[ndone++] = (const void*)kern;
        }
    }
    hipLaunchKernelGGL(kern, dim3(slices, nsplit), dim3(p.nthr), smem, st, p);
    return m1_check_launch();
}
