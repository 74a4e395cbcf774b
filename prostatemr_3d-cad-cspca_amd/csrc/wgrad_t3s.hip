// wgrad_t3s.hip -- tap-fused fp32 weight gradient for the layers with FEW channels (the SE bottleneck convs: 8 / 16 / 32 channels
// on one or both sides, stride 1, (3,3,3) or (1,3,3)) on v_mfma_f32_32x32x2_f32:
//
//   R[tap][a][b] += sum_{n,v} A[n, v + tap - p][a] * B[n, v][b]            the 9 (kh,kw) taps of one kd slice per block
//
// The per-tap kernel (wgrad_mfma.hip) reads X and dY once per tap: at (32,256,256) a 32 -> 32 (1,3,3) layer moves 9 x 0.54 GB for
// 39 GFLOP and takes 1.7 ms -- 17 ms of the 56 ms C5 step (BASELINE.json configs[3], fp32) went to these layers.  Here
// * a block owns the 9 taps of one kd slice of ONE 32 x 32 channel tile (channels beyond CA / CB are zero rows: the LDS-DMA
//   fetches nothing for them); per K-tile of 128 output voxels (TH rows x KWs columns of one (n,d) slice) it stages the dY rows
//   and the X rows INCLUDING the tap halo once, by LDS-DMA (rows of 128 bytes, no swizzle: a half wave reads 32 consecutive dwords);
// * 12 waves = one block per CU: wave = (voxel quarter vq, tap row kh) accumulates the 32 x 32 tiles of its 3 kw taps over its 32
//   voxels of the K-tile (3 MFMAs per voxel pair from 3 + 1 dword reads); the four quarters are summed in a fixed order through
//   LDS when the block is done;
// * tile table, fixed DMA piece slots, per-split partial copies + fixed-order fold as in wgrad_t3.hip (no float atomics).
#include "common.h"
#include "gather.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4s_t;
template <int TL> struct TsAcc;
template <> struct TsAcc<32> { typedef f32x16_t type; static constexpr int NE = 16, VG = 2;
    static __device__ __forceinline__ type mfma(float a, float b, type c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); } };
template <> struct TsAcc<16> { typedef f32x4s_t type; static constexpr int NE = 4, VG = 4;
    static __device__ __forceinline__ type mfma(float a, float b, type c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); } };
typedef __attribute__((address_space(3))) void* lptr_t;
typedef int i32x4_t __attribute__((ext_vector_type(4)));

#define TS_WAVES 12
#define TS_THREADS (TS_WAVES * 64)
#define TS_KT 128           // voxel slots per K-tile (4 quarters of 32); 64 for the stride-2 variant (its X tile is 4x the dY tile)

struct TSP {
    const float* A; const float* B; float* Rx; long long rx_stride, rx_bias;
    int CA, CB, AD, AH, AW, BD, BH, BW, N;
    int pd, ph, pw, KD, sd;
    int tiles_w, tiles_h, ntiles, nsplit, stages;
    int nau;                 // 32-channel units of A (blockIdx.z = unit * KD + kd), b units on blockIdx.x
    int want_bsum;
};

__device__ __forceinline__ void ts_dma(i32x4_t rs, unsigned lds, unsigned voff) {
    // (M0 is written here without a clobber: "m0" is a reserved register to hipcc -- it warns on the clobber -- and these kernels contain no
    // compiler-generated M0 use that a stale value could reach; tools/isa_async_check.py / tests/test_build_props.py verify that on the ISA)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" :: "s"(lds), "v"(voff), "s"(rs) : "memory");
}

// STR = stride of the gather in H and W (1, or 2: strided convs and, with the roles of the two sides swapped, transposed convs;
// TF-SAME pad_before 0).  No de-interleaving as in the bf16 kernel: a dword fragment read is conflict-free whatever the rows.
// TL = channel tile: 32 (v_mfma_f32_32x32x2_f32, a voxel pair per MFMA) or 16 (v_mfma_f32_16x16x4_f32, four voxels per MFMA) for the
// layers with <= 16 channels on a side -- padded to 32 they did 4-16x the work (8 -> 8 at (32,256,256): 0.92 ms).
template <int KWS, int STR, int TL>
__global__ void __launch_bounds__(TS_THREADS, 3) wgrad_t3s_kernel(TSP p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int vq = wave / 3, kh = wave - 3 * vq;
    constexpr int KT = STR == 1 ? TS_KT : TS_KT / 2, VPQ = KT / 4;
    constexpr int KWs = KWS, TH = KT / KWS, AWt = STR == 1 ? KWs + 2 : 2 * KWs + 2, AHt = STR == 1 ? TH + 2 : 2 * TH + 1, arows = AHt * AWt;
    constexpr int nA = (arows + 7) / 8, nB = KT / 8;          // 1 KB pieces (8 rows of 128 bytes)
    constexpr int A_ITS = (nA + TS_WAVES - 1) / TS_WAVES, B_ITS = (nB + TS_WAVES - 1) / TS_WAVES, NP = A_ITS + B_ITS;
    constexpr int stage_bytes = (nA + nB) * 1024;
    const int kd = (int)blockIdx.z % p.KD, au = (int)blockIdx.z / p.KD;
    const int a_base = au * TL, b_base = (int)blockIdx.x * TL;
    typedef typename TsAcc<TL>::type acc_t;
    constexpr int NE = TsAcc<TL>::NE, VG = TsAcc<TL>::VG, CL = 64 / VG;      // accumulator registers, voxels per MFMA, lanes per voxel
    constexpr unsigned OOB = 0x80000000u;
    const unsigned lds0 = (unsigned)(unsigned long long)(lptr_t)smem;

    // ---- this lane's LDS-DMA pieces: tile row r = hh * AWt + ww holds 32 channels (a 16-byte slot = 4 channels) ----
    unsigned vo[NP]; int pk[NP]; int dst[NP];
    const unsigned trash = lds0 + (unsigned)(p.stages * stage_bytes);
#pragma unroll
    for (int it = 0; it < NP; ++it) {
        vo[it] = OOB; pk[it] = 0;
        if (it < A_ITS) {
            const int q = wave + TS_WAVES * it;
            const bool real = q < nA;
            dst[it] = real ? q * 1024 : -1;
            const int s = q * 64 + lane, row = s >> 3, sl = s & 7;
            const int hh = row / AWt, ww = row - hh * AWt;
            pk[it] = hh | (ww << 8);
            if (real && row < arows && sl * 4 < TL && a_base + sl * 4 < p.CA) vo[it] = (unsigned)(((hh * p.AW + ww) * p.CA + a_base + sl * 4) * 4);
        } else {
            const int q = wave + TS_WAVES * (it - A_ITS);
            const bool real = q < nB;
            dst[it] = real ? (nA + q) * 1024 : -1;
            const int s = q * 64 + lane, kk = s >> 3, sl = s & 7;
            const int th = kk / KWs, tw = kk - th * KWs;
            pk[it] = th;
            if (real && sl * 4 < TL && b_base + sl * 4 < p.CB) vo[it] = (unsigned)(((th * p.BW + tw) * p.CB + b_base + sl * 4) * 4);
        }
    }

    // ---- tile table (see wgrad_t3.hip) ----
    const int my_tiles = (p.ntiles - (int)blockIdx.y + p.nsplit - 1) / p.nsplit;
    constexpr int RED_BYTES = 9 * 3 * NE * 64 * 4;             // the three other quarters' accumulators at the end
    const int pipe_bytes = p.stages * stage_bytes + 1024;
    int* const tab = reinterpret_cast<int*>(smem + (pipe_bytes > RED_BYTES ? pipe_bytes : RED_BYTES));
    for (int t = tid; t < my_tiles + p.stages; t += TS_THREADS) {
        int e[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (t < my_tiles) {
            int r = (int)blockIdx.y + t * p.nsplit;
            const int twi = r % p.tiles_w; r /= p.tiles_w;
            const int thi = r % p.tiles_h; r /= p.tiles_h;
            const int bd = r % p.BD, n = r / p.BD;
            const int ad = bd * p.sd + kd - p.pd, ah0 = thi * TH * STR - p.ph, aw0 = twi * KWs * STR - p.pw, bh0 = thi * TH;
            const long long alin0 = (((long long)n * p.AD + ad) * p.AH + ah0) * p.AW + aw0;
            const long long blin0 = (((long long)n * p.BD + bd) * p.BH + bh0) * p.BW + twi * KWs;
            const unsigned long long pa = (unsigned long long)(p.A + alin0 * p.CA), pb = (unsigned long long)(p.B + blin0 * p.CB);
            e[0] = (int)(unsigned)pa; e[1] = (int)((unsigned)(pa >> 32) & 0xffffu); e[2] = (unsigned)ad < (unsigned)p.AD ? 0x7fffffff : 0;
            e[3] = (ah0 & 0xffff) | (aw0 << 16);
            e[4] = (int)(unsigned)pb; e[5] = (int)((unsigned)(pb >> 32) & 0xffffu); e[6] = 0x7fffffff; e[7] = bh0;
        }
        reinterpret_cast<int4*>(tab)[2 * t] = make_int4(e[0], e[1], e[2], e[3]);
        reinterpret_cast<int4*>(tab)[2 * t + 1] = make_int4(e[4], e[5], e[6], e[7]);
    }
    __syncthreads();
    int q_e = 0;
    int4 ea, eb;
    auto fetch = [&]() {
        ea = reinterpret_cast<const int4*>(tab)[2 * q_e]; eb = reinterpret_cast<const int4*>(tab)[2 * q_e + 1];
        ++q_e;
    };
    auto issue = [&](int st) {
        i32x4_t ra, rb;
        ra.x = __builtin_amdgcn_readfirstlane(ea.x); ra.y = __builtin_amdgcn_readfirstlane(ea.y);
        ra.z = __builtin_amdgcn_readfirstlane(ea.z); ra.w = 0x00020000;
        rb.x = __builtin_amdgcn_readfirstlane(eb.x); rb.y = __builtin_amdgcn_readfirstlane(eb.y);
        rb.z = __builtin_amdgcn_readfirstlane(eb.z); rb.w = 0x00020000;
        const int ah0 = (ea.w << 16) >> 16, aw0 = ea.w >> 16, bh0 = eb.w;
        const unsigned S0 = lds0 + (unsigned)(st * stage_bytes);
#pragma unroll
        for (int it = 0; it < NP; ++it) {
            const unsigned d = dst[it] >= 0 ? S0 + (unsigned)dst[it] : trash;
            if (it < A_ITS) {
                const unsigned hh = (unsigned)(pk[it] & 0xff), ww = (unsigned)(pk[it] >> 8);
                unsigned o = (hh + (unsigned)ah0) < (unsigned)p.AH ? vo[it] : OOB;
                o = (ww + (unsigned)aw0) < (unsigned)p.AW ? o : OOB;
                ts_dma(ra, d, o);
            } else {
                ts_dma(rb, d, (unsigned)(pk[it] + bh0) < (unsigned)p.BH ? vo[it] : OOB);
            }
        }
    };

    // fragments: lane l = (voxel of the pair g = l >> 5, channel c = l & 31): one dword per operand and MFMA.  Voxel slot of step
    // k2: 32 vq + 2 k2 + g = tile row (32 vq + 2 k2) / KWs, column (2 k2) % KWs + g  (32 and KWs are multiples of each other)
    const int g = lane / CL, c = lane % CL;
    const int thq = (VPQ * vq) / KWs, twq = (VPQ * vq) % KWs;
    const unsigned char* const aL = smem + (((STR * thq + kh) * AWt + STR * (twq + g)) * 32 + c) * 4;
    const unsigned char* const bL = smem + nA * 1024 + ((VPQ * vq + g) * 32 + c) * 4;

    acc_t acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int e = 0; e < NE; ++e) acc[t][e] = 0.f;
    const bool do_bsum = p.want_bsum && au == 0 && kd == 0 && kh == 0;
    float accb = 0.f;

    const int S = p.stages;                                    // 2 or 3
    for (int s = 0; s < S - 1; ++s) { fetch(); issue(s); }
    int st = 0;
    for (int kt = blockIdx.y; kt < p.ntiles; kt += p.nsplit) {
        fetch();
        if (S == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        int stn = st + S - 1; if (stn >= S) stn -= S;
        issue(stn);
        const unsigned char* const ap = aL + st * stage_bytes; const unsigned char* const bp = bL + st * stage_bytes;
#pragma unroll 4
        for (int k2 = 0; k2 < VPQ / VG; ++k2) {
            const int th = (VG * k2) / KWs, tw0 = (VG * k2) % KWs;
            const unsigned char* a = ap + STR * (th * AWt + tw0) * 128;
            const float bf = *reinterpret_cast<const float*>(bp + (VG * k2) * 128);
            const float af0 = *reinterpret_cast<const float*>(a), af1 = *reinterpret_cast<const float*>(a + 128),
                        af2 = *reinterpret_cast<const float*>(a + 256);
            if (do_bsum) accb += bf;
            acc[0] = TsAcc<TL>::mfma(af0, bf, acc[0]);
            acc[1] = TsAcc<TL>::mfma(af1, bf, acc[1]);
            acc[2] = TsAcc<TL>::mfma(af2, bf, acc[2]);
        }
        if (++st == S) st = 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                           // every wave is done reading the stages: they become the reduction buffer

    // ---- quarters 1..3 hand their accumulators to quarter 0 (fixed order of additions) ----
    float* const red = reinterpret_cast<float*>(smem);         // [quarter - 1][kh][tap][e][lane]
    float* const redb = red + 9 * 3 * NE * 64;                 // bias sums [quarter][lane] (kh == 0 waves)  -- inside the table area: done with it
    if (vq > 0) {
        float* r = red + ((vq - 1) * 3 + kh) * 3 * NE * 64;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int e = 0; e < NE; ++e) r[(t * NE + e) * 64 + lane] = acc[t][e];
    }
    if (do_bsum) redb[vq * 64 + lane] = accb;
    __syncthreads();
    if (vq > 0) return;
#pragma unroll 1
    for (int q = 0; q < 3; ++q) {                              // (rolled: 48 loads in flight at a time, not 144)
        const float* r = red + (q * 3 + kh) * 3 * NE * 64;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int e = 0; e < NE; ++e) acc[t][e] += r[(t * NE + e) * 64 + lane];
    }
    float* Rx = p.Rx + (long long)blockIdx.y * p.rx_stride;
    if (do_bsum) {
        float sb = redb[lane] + redb[64 + lane];               // (quarter order 0, 1, 2, 3; then the two voxels of a pair)
        sb += redb[128 + lane]; sb += redb[192 + lane];
        sb += __shfl_xor(sb, 32);                              // ... then the voxels of a group
        if (TL == 16) sb += __shfl_xor(sb, 16);
        if (lane < CL && b_base + lane < p.CB) Rx[p.rx_bias + b_base + lane] = sb;
    }
    // D[a][b]: 32x32 tile: lane holds b = lane & 31, a = (e & 3) + 8 (e >> 2) + 4 (lane >> 5); 16x16 tile: b = lane & 15, a = 4 (lane >> 4) + e
    const int b = b_base + c;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const long long tap = (long long)(kd * 3 + kh) * 3 + t;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int a = a_base + (TL == 32 ? (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5) : 4 * (lane >> 4) + e);
            if (a < p.CA && b < p.CB) Rx[(tap * p.CA + a) * p.CB + b] = acc[t][e];
        }
    }
#endif
}

static bool ts_plan(const WgradSpec& g, TSP& p, int* kws_out) {
    int en = M1_CFG("M1_WG_T3S", 1);
    if (!en || g.dtype != M1_F32) return false;
    if (g.CA % 4 || g.CB % 4 || g.CA < 4 || g.CB < 4) return false;
    if (!(g.kh == 3 && g.kw == 3 && (g.kd == 1 || g.kd == 3))) return false;
    const bool s1 = g.sh == 1 && g.sw == 1 && g.sd == 1, s2 = g.sh == 2 && g.sw == 2 && (g.sd == 1 || g.sd == 2) && g.ph == 0 && g.pw == 0;
    if (!s1 && !s2) return false;
    if (g.BW % 8) return false;
    const int TLh = (g.CA <= 16 || g.CB <= 16) ? 16 : 32;
    const int nau = (g.CA + TLh - 1) / TLh, nbu = (g.CB + TLh - 1) / TLh;
    // every tile pair re-reads its operands: the wide stride-1 layers have wgrad_t3f; the strided / transposed ones (small volumes
    // at the deep levels) only this kernel
    if (nau * nbu > (s2 ? 128 : 32)) return false;
    if ((long long)(g.AH + 4) * g.AW * g.CA * 4 >= (1ll << 31) - 4096 || (long long)(g.BH + 20) * g.BW * g.CB * 4 >= (1ll << 31) - 4096) return false;
    int kws = g.BW % 32 == 0 ? 32 : (g.BW % 16 == 0 ? 16 : 8);
    const int KT = s2 ? TS_KT / 2 : TS_KT;
    p = TSP{};
    p.A = (const float*)g.A; p.B = (const float*)g.B;
    p.CA = g.CA; p.CB = g.CB; p.AD = g.AD; p.AH = g.AH; p.AW = g.AW; p.BD = g.BD; p.BH = g.BH; p.BW = g.BW; p.N = g.N;
    p.pd = g.pd; p.ph = g.ph; p.pw = g.pw; p.KD = g.kd; p.sd = g.sd;
    const int TH = KT / kws;
    p.tiles_w = g.BW / kws; p.tiles_h = (g.BH + TH - 1) / TH;
    const long long nt = (long long)g.N * g.BD * p.tiles_h * p.tiles_w;
    if (nt >= (1ll << 30) || nt < 4) return false;
    p.ntiles = (int)nt; p.nau = nau;
    *kws_out = kws;
    return true;
}
bool m1_t3s_wgrad_supported(const WgradSpec& g) { TSP p; int k; return ts_plan(g, p, &k); }

int m1_t3s_wgrad(const WgradSpec& g, long long nw, int nb, hipStream_t st) {
    TSP p; int kws;
    if (!ts_plan(g, p, &kws)) return M1_ERR_UNSUPPORTED;
    const int TLh = (g.CA <= 16 || g.CB <= 16) ? 16 : 32;
    const int nbu = (g.CB + TLh - 1) / TLh;
    const long long per_split = (long long)nbu * p.nau * g.kd;
    int tgt = M1_CFG("M1_T3S_BLOCKS", 256);
    long long nsplit = tgt / per_split; if (nsplit < 1) nsplit = 1;
    const long long nloc = (long long)g.kd * 9 * g.CA * g.CB;
    const long long stride = nloc + g.CB;
    if (!g.rx || g.rx_floats < stride) return M1_ERR_WORKSPACE;
    if (nsplit * stride > g.rx_floats) nsplit = g.rx_floats / stride;
    if (nsplit > p.ntiles / 4) nsplit = p.ntiles / 4;          // >= 4 K-tiles per block
    if (nsplit > 512) nsplit = 512;
    if (nsplit < 1) nsplit = 1;
    p.nsplit = (int)nsplit;
    p.Rx = g.rx; p.rx_stride = stride; p.rx_bias = nloc;
    p.want_bsum = g.bsum != nullptr;
    const bool s2 = g.sh == 2;
    const int KT = s2 ? TS_KT / 2 : TS_KT;
    const int TH = KT / kws, arows = s2 ? (2 * TH + 1) * (2 * kws + 2) : (TH + 2) * (kws + 2), nA = (arows + 7) / 8;
    const int stage_bytes = (nA + KT / 8) * 1024;
    const long long tiles_per_block = (p.ntiles + nsplit - 1) / nsplit;
    const size_t red_bytes = (size_t)9 * 3 * (TLh == 32 ? 16 : 4) * 64 * 4;
    int S = 3;
    auto need = [&](int s) { const size_t pipe = (size_t)s * stage_bytes + 1024; return (pipe > red_bytes ? pipe : red_bytes) + (size_t)(tiles_per_block + s) * 32 + 1024; };
    while (S >= 2 && need(S) > 160 * 1024) --S;
    if (S < 2) return M1_ERR_UNSUPPORTED;
    p.stages = S;
    size_t smem = need(S);
    if (smem < red_bytes + 4 * 64 * 4 + 1024) smem = red_bytes + 4 * 64 * 4 + 1024;     // the bias sums sit behind the reduction buffer
    void (*kern)(TSP) = nullptr;
#define TSK(K_, S_) (TLh == 32 ? wgrad_t3s_kernel<K_, S_, 32> : wgrad_t3s_kernel<K_, S_, 16>)
    kern = s2 ? (kws == 32 ? TSK(32, 2) : (kws == 16 ? TSK(16, 2) : TSK(8, 2))) : (kws == 32 ? TSK(32, 1) : (kws == 16 ? TSK(16, 1) : TSK(8, 1)));
#undef TSK
    {
        static const void* done[16]; static int ndone = 0;
        bool seen = false;
        for (int q = 0; q < ndone; ++q) seen |= done[q] == (const void*)kern;
        if (!seen) {
            if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return M1_ERR_LAUNCH;
            if (ndone < 16) done[ndone++] = (const void*)kern;
        }
    }
    m1_note_kernel("wgrad_t3s");
    hipLaunchKernelGGL(kern, dim3((unsigned)nbu, (unsigned)nsplit, (unsigned)(p.nau * g.kd)), dim3(TS_THREADS), smem, st, p);
    int rc = m1_check_launch(); if (rc) return rc;
    return m1_wg_rx_finish(p.Rx, stride, (int)nsplit, g, nloc, st);
}

// ------------------------------------------------------------------------------------------------------------------------------
// Pointwise (1x1x1, stride 1) fp32 weight gradient: R[a][b] += sum_v X[v][a] * dY[v][b], a plain GEMM whose contraction axis is the
// voxel list.  The per-tap kernel ran it at 2.6 TFLOP/s (128 -> 128 at (32,64,64): 1.6 ms for 134 MB of operands); this one is
// bound by the operand stream: a block owns one 64 x 64 channel tile and walks K-tiles of 128 consecutive voxels (both operands by
// LDS-DMA, 256-byte rows, two stages); 8 waves = (voxel quarter, a half) x 2 MFMAs (32 x 32 x 2) per voxel pair; the quarters are
// summed in a fixed order through LDS, per-split partial copies + fold as everywhere.
#define PW_WAVES 8
#define PW_THREADS (PW_WAVES * 64)
#define PW_KT 128
struct PWP { const float* A; const float* B; float* Rx; long long rx_stride, rx_bias; int CA, CB; long long V; int ntiles, nsplit, want_bsum; };

__global__ void __launch_bounds__(PW_THREADS, 2) wgrad_pwf_kernel(PWP p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int vq = wave >> 1, ah = wave & 1;
    constexpr int nT = PW_KT / 4;                              // 1 KB pieces of one operand tile (4 rows of 256 bytes)
    constexpr int ITS = nT / PW_WAVES, NP = 2 * ITS;           // 4 + 4 pieces per wave and stage
    constexpr int stage_bytes = 2 * nT * 1024;                 // 64 KB
    const int a_base = (int)blockIdx.z * 64, b_base = (int)blockIdx.x * 64;
    constexpr unsigned OOB = 0x80000000u;
    const unsigned lds0 = (unsigned)(unsigned long long)(lptr_t)smem;
    unsigned voA[ITS], voB[ITS];
#pragma unroll
    for (int it = 0; it < ITS; ++it) {
        const int s = (wave + PW_WAVES * it) * 64 + lane, row = s >> 4, sl = s & 15;
        voA[it] = a_base + sl * 4 < p.CA ? (unsigned)((row * p.CA + a_base + sl * 4) * 4) : OOB;
        voB[it] = b_base + sl * 4 < p.CB ? (unsigned)((row * p.CB + b_base + sl * 4) * 4) : OOB;
    }
    auto issue = [&](long long tile, int st) {
        // the tile's first voxel travels in the resource base, its number of valid rows in the byte range (rows past the end: zeros)
        const long long v0 = tile * PW_KT;
        long long left = p.V - v0; if (left < 0) left = 0; if (left > PW_KT) left = PW_KT;
        const unsigned long long pa = (unsigned long long)(p.A + v0 * p.CA), pb = (unsigned long long)(p.B + v0 * p.CB);
        i32x4_t ra, rb;
        ra.x = __builtin_amdgcn_readfirstlane((int)(unsigned)pa); ra.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(pa >> 32) & 0xffffu));
        ra.z = __builtin_amdgcn_readfirstlane((int)(left * p.CA * 4)); ra.w = 0x00020000;
        rb.x = __builtin_amdgcn_readfirstlane((int)(unsigned)pb); rb.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(pb >> 32) & 0xffffu));
        rb.z = __builtin_amdgcn_readfirstlane((int)(left * p.CB * 4)); rb.w = 0x00020000;
        const unsigned S0 = lds0 + (unsigned)(st * stage_bytes);
#pragma unroll
        for (int it = 0; it < ITS; ++it) ts_dma(ra, S0 + (unsigned)((wave + PW_WAVES * it) * 1024), voA[it]);
#pragma unroll
        for (int it = 0; it < ITS; ++it) ts_dma(rb, S0 + (unsigned)((nT + wave + PW_WAVES * it) * 1024), voB[it]);
    };
    const int g = lane >> 5, c = lane & 31;
    const unsigned char* const aL = smem + ((32 * vq + g) * 64 + ah * 32 + c) * 4;
    const unsigned char* const bL = smem + nT * 1024 + ((32 * vq + g) * 64 + c) * 4;
    f32x16_t acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    const bool do_bsum = p.want_bsum && blockIdx.z == 0 && ah == 0;
    float accb0 = 0.f, accb1 = 0.f;
    long long kt = blockIdx.y;
    issue(kt, 0);
    int st = 0;
    for (; kt < p.ntiles; kt += p.nsplit) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        issue(kt + p.nsplit, st ^ 1);                          // (a tile past the end: zero rows, never read)
        const unsigned char* const ap = aL + st * stage_bytes; const unsigned char* const bp = bL + st * stage_bytes;
#pragma unroll 4
        for (int k2 = 0; k2 < 16; ++k2) {
            const float af = *reinterpret_cast<const float*>(ap + k2 * 512);
            const float bf0 = *reinterpret_cast<const float*>(bp + k2 * 512), bf1 = *reinterpret_cast<const float*>(bp + k2 * 512 + 128);
            if (do_bsum) { accb0 += bf0; accb1 += bf1; }
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf1, acc[1], 0, 0, 0);
        }
        st ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float* const red = reinterpret_cast<float*>(smem);         // [quarter - 1][a half][tile][e][lane], then the bias sums [quarter][2][lane]
    float* const redb = red + 6 * 32 * 64;
    if (vq > 0) {
        float* r = red + ((vq - 1) * 2 + ah) * 32 * 64;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) r[(t * 16 + e) * 64 + lane] = acc[t][e];
    }
    if (do_bsum) { redb[(vq * 2) * 64 + lane] = accb0; redb[(vq * 2 + 1) * 64 + lane] = accb1; }
    __syncthreads();
    if (vq > 0) return;
#pragma unroll 1
    for (int q = 0; q < 3; ++q) {
        const float* r = red + (q * 2 + ah) * 32 * 64;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][e] += r[(t * 16 + e) * 64 + lane];
    }
    float* Rx = p.Rx + (long long)blockIdx.y * p.rx_stride;
    if (do_bsum) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float sb = redb[h * 64 + lane];
            sb += redb[(2 + h) * 64 + lane]; sb += redb[(4 + h) * 64 + lane]; sb += redb[(6 + h) * 64 + lane];
            sb += __shfl_xor(sb, 32);
            const int b = b_base + h * 32 + lane;
            if (lane < 32 && b < p.CB) Rx[p.rx_bias + b] = sb;
        }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int b = b_base + t * 32 + (lane & 31);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int a = a_base + ah * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
            if (a < p.CA && b < p.CB) Rx[(long long)a * p.CB + b] = acc[t][e];
        }
    }
#endif
}

bool m1_pwf_wgrad_supported(const WgradSpec& g) {
    int en = M1_CFG("M1_WG_PWF", 1);
    if (!en || g.dtype != M1_F32) return false;
    if (g.kd != 1 || g.kh != 1 || g.kw != 1 || g.sd != 1 || g.sh != 1 || g.sw != 1) return false;
    if (g.AD != g.BD || g.AH != g.BH || g.AW != g.BW) return false;
    if (g.CA % 4 || g.CB % 4 || g.CA < 4 || g.CB < 4) return false;
    const long long V = (long long)g.N * g.BD * g.BH * g.BW;
    if (V < 4 * PW_KT) return false;
    if ((long long)PW_KT * (g.CA > g.CB ? g.CA : g.CB) * 4 >= (1ll << 31) - 4096) return false;
    return ((g.CA + 63) / 64) * ((g.CB + 63) / 64) <= 64;
}
int m1_pwf_wgrad(const WgradSpec& g, long long nw, int nb, hipStream_t st) {
    if (!m1_pwf_wgrad_supported(g)) return M1_ERR_UNSUPPORTED;
    PWP p{};
    p.A = (const float*)g.A; p.B = (const float*)g.B; p.CA = g.CA; p.CB = g.CB;
    p.V = (long long)g.N * g.BD * g.BH * g.BW;
    p.ntiles = (int)((p.V + PW_KT - 1) / PW_KT);
    const int nau = (g.CA + 63) / 64, nbu = (g.CB + 63) / 64;
    int tgt = M1_CFG("M1_PWF_BLOCKS", 256);     // one block per CU (128 KB of LDS)
    long long nsplit = tgt / (nau * nbu); if (nsplit < 1) nsplit = 1;
    const long long nloc = (long long)g.CA * g.CB, stride = nloc + g.CB;
    if (!g.rx || g.rx_floats < stride) return M1_ERR_WORKSPACE;
    if (nsplit * stride > g.rx_floats) nsplit = g.rx_floats / stride;
    if (nsplit > p.ntiles / 2) nsplit = p.ntiles / 2;
    if (nsplit > 512) nsplit = 512;
    if (nsplit < 1) nsplit = 1;
    p.nsplit = (int)nsplit; p.Rx = g.rx; p.rx_stride = stride; p.rx_bias = nloc; p.want_bsum = g.bsum != nullptr;
    const size_t smem = 2 * (size_t)(2 * (PW_KT / 4) * 1024);
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)wgrad_pwf_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return M1_ERR_LAUNCH;
        attr = true;
    }
    m1_note_kernel("wgrad_pwf");
    hipLaunchKernelGGL(wgrad_pwf_kernel, dim3((unsigned)nbu, (unsigned)nsplit, (unsigned)nau), dim3(PW_THREADS), smem, st, p);
    int rc = m1_check_launch(); if (rc) return rc;
    return m1_wg_rx_finish(p.Rx, stride, (int)nsplit, g, nloc, st);
}
