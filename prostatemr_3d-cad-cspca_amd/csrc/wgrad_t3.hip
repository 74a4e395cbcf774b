// wgrad_t3.hip -- tap-fused bf16 weight gradient for the layers with >= 64 channels on both sides (stride 1; stride (1|2,2,2) and
// transposed convs through the parity-plane staging, STR = 2), on
// v_mfma_f32_32x32x16_bf16:
//
//   R[tap][a][b] += sum_{n,v} A[n, v + tap - p][a] * B[n, v][b]            the 9 (kh,kw) taps of one kd slice per block
//
// The per-tap kernel (wgrad_tap.hip) stages both operands once per tap and reads two LDS fragments per 16x16x32 MFMA; the
// 32x32-tile tap-fused kernel (wgrad_tf.hip) re-reads dY once per 32 input channels.  Here
// * a block owns the 9 taps of ONE kd slice of TWO 64 x 64 channel tiles that share one operand: one A tile (64 input
//   channels) with two B tiles (128 output channels of dY) when C_out is a multiple of 128 ("BIGB"), else two A tiles -- two
//   64-channel units of the concat, of one member or of two -- with one B tile ("BIGA").  Per K-tile of <= 64 output voxels (TH
//   rows x KWs columns of one (n,d) slice) it stages, by LDS-DMA, the B rows and the A rows INCLUDING the tap halo: X and dY
//   cross L2 -> LDS once for 9 taps and two channel tiles;
// * 12 waves = ONE block per CU with 3 waves on every SIMD (6-wave blocks at "two per CU" measured ~1.07 resident: the second
//   block rarely finds 3 free wave slots on the right SIMDs).  Wave w = (tile pair half w / 6, a half wa, tap row kh) accumulates
//   the 32(a) x 64(b) tile of its 3 kw taps: 6 MFMAs (32x32x16) per 16-voxel k-step from 6 A + 4 B transpose reads
//   (ds_read_b64_tr_b16, compiler-visible: constant offsets fold into the instruction) -- 1.7 LDS reads per MFMA of twice the
//   work of a 16x16x32 one; the dY fragments of a k-step are shared by the 3 taps; 96 accumulator registers;
// * 128-byte LDS rows (64 channels); the two 64-byte halves of a row swap when bit 1 of its tile column is set, so that the 4
//   rows x 64 bytes a 32-lane group of a transpose read touches fall on 4 different bank quarters (0 conflicts measured);
// * the geometry of every K-tile of the block (buffer-resource words of the tile origins) is computed once into an LDS table:
//   12 waves repeating ~120 scalar instructions of 64-bit index arithmetic per tile executed 7.2 scalar instructions per
//   MFMA on the CU's shared scalar unit (2.1 now; worth 3 % by itself, the occupancy point above was the larger one);
//   piece slots have fixed kinds and no branches for the same reason;
// * K-tiles are dealt round-robin to the blocks of a channel tile (blockIdx.y); every block stores its partial tiles into its
//   own compact copy of the member's gradient block, folded in a fixed order by m1_wg_rx_finish (no float atomics);
//   equal-width members of a concat share the launch.
// wgrad_t3f_kernel below is the fp32 variant of the same block on v_mfma_f32_32x32x2_f32 (plain dword fragment reads, two stages);
// wgrad_t3s.hip holds the fp32 kernels for layers with few channels and for pointwise layers.
#include "common.h"
#include "gather.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef int i32x4_t __attribute__((ext_vector_type(4)));

#define T3_WAVES 12
#define T3_THREADS (T3_WAVES * 64)
#define T3_KT 64            // voxel slots per K-tile (4 k-steps of 16)

struct T3P {
    const bf16_t* A0; const bf16_t* A1; const bf16_t* A2; const bf16_t* A3; const bf16_t* A4; const bf16_t* A5; int nmem;
    const bf16_t* B; float* Rx; long long rx_stride, rx_mem, rx_bias;
    int CA, CB, AD, AH, AW, BD, BH, BW, N;
    int pd, ph, pw, KD, sd;
    int KWs, TH;
    int aTiles, nunits;      // 64-channel units of the concat: unit u = (member u / aTiles, channels 64 (u % aTiles) ..)
    int tiles_w, tiles_h, ntiles, nsplit, stages;
    int want_bsum;
};

// transpose read, compiler-visible (it packs the two halves of a fragment into one register quadruple, folds constant
// offsets into the instruction and schedules the lgkmcnt waits).  The LDS-DMA below is issued from inline asm, so the compiler
// never sees a pending LDS write that it would drain with vmcnt(0) in front of every read.
__device__ __forceinline__ s16x4_t t3_tr(const unsigned char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p);
}
__device__ __forceinline__ bf16x8_t t3_frag(s16x4_t lo, s16x4_t hi) {
    return __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
// 64 lanes x 16 bytes, global (buffer resource `rs`, per-lane byte offset `voff`; out of range -> zeros) -> LDS at the
// wave-uniform byte address `lds` + 16 * lane.  M0 carries the LDS base of an LDS-DMA.
__device__ __forceinline__ void t3_dma(i32x4_t rs, unsigned lds, unsigned voff) {
    // (M0 is written here without a clobber: "m0" is a reserved register to hipcc -- it warns on the clobber -- and these kernels contain no
    // compiler-generated M0 use that a stale value could reach; tools/isa_async_check.py / tests/test_build_props.py verify that on the ISA)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" :: "s"(lds), "v"(voff), "s"(rs) : "memory");
}
// at most P * (S - 2) LDS-DMA pieces of this wave still in flight (every wave issues P pieces per stage)
template <int P> __device__ __forceinline__ void t3_wait_stages(int S) {
    if (S == 3) { if (P == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); }
    else if (S == 4) { if (P == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); }
    else if (S == 5) { if (P == 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(21)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ float t3_sum8(bf16x8_t f) {       // sum of the 8 bf16 of a fragment
    const uint4 v = __builtin_bit_cast(uint4, f);
    float s = __uint_as_float(v.x << 16) + __uint_as_float(v.x & 0xffff0000u);
    s += __uint_as_float(v.y << 16) + __uint_as_float(v.y & 0xffff0000u);
    s += __uint_as_float(v.z << 16) + __uint_as_float(v.z & 0xffff0000u);
    s += __uint_as_float(v.w << 16) + __uint_as_float(v.w & 0xffff0000u);
    return s;
}
__device__ __forceinline__ const bf16_t* t3_member(const T3P& p, int m) {      // (uniform selects: a dynamic index would move the struct to scratch)
    const bf16_t* a = p.A0;
    if (m == 1) a = p.A1;
    if (m == 2) a = p.A2;
    if (m == 3) a = p.A3;
    if (m == 4) a = p.A4;
    if (m == 5) a = p.A5;
    return a;
}

// KWS = columns of a K-tile: 8 / 16 / 32 (TH = 64 / KWS rows; fragment addresses = per-lane base + compile-time offsets), or
// 0 = any multiple of 4 up to 32 (whole-row tiles of the (10,20,20) level: per-lane address table).
// BIGB: one A tile + two B tiles (else two A tiles + one B tile).
// STR = stride of the gather in H and W: 1, or 2 (strided convs; the transposed convs, whose dOut is the gathered side) with
// TF-SAME pad_before 0.  The A tile is then staged DE-INTERLEAVED: 4 parity planes [row parity][column parity] of
// (TH + 1) x (KWs + 2) rows, so that the voxels tw, tw+1, .. of a tap -- input columns 2 tw + kw -- are CONSECUTIVE rows of
// plane (kh & 1, kw & 1) at (th + (kh >> 1), tw + (kw >> 1)): the same conflict-free transpose reads and the same compile-time
// offsets as stride 1 (interleaved, 4 voxels would sit 256 bytes apart: one bank quarter, 2-way conflicts at best).
template <int KWS, bool BIGB, int STR>
__global__ void __launch_bounds__(T3_THREADS, 3) wgrad_t3_kernel(T3P p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wsel = wave >= 6 ? 1 : 0, w6 = wave - 6 * wsel;
    const int wa = w6 & 1, kh = w6 >> 1;
    // A tile rows: stride 1 [TH + 2][KWs + 2]; stride 2 [4 planes][TH + 1][KWs + 2] (plane width KWs + 1, padded to even)
    const int KWs = KWS ? KWS : p.KWs, TH = KWS ? T3_KT / KWS : p.TH, AWt = KWs + 2, AHt = STR == 1 ? TH + 2 : TH + 1;
    const int PLANE = AHt * AWt, arows = (STR == 1 ? 1 : 4) * PLANE;
    const int nA = (arows + 7) / 8;                            // 1 KB pieces of ONE A tile (rows of 128 bytes)
    constexpr int nB = T3_KT / 8;                              // ... of one B tile
    constexpr int NTA = BIGB ? 1 : 2, NTB = BIGB ? 2 : 1;      // tiles of each kind per stage
    constexpr int A_ITS = STR == 2 ? 5 : (BIGB ? 2 : 3);       // piece slots per wave: A_ITS for A, NP - A_ITS for B
    constexpr int NP = STR == 2 ? 7 : 4;
    const int nAtot = NTA * nA;
    const int stage_bytes = (nAtot + NTB * nB) * 1024;
    const int kd = (int)blockIdx.z % p.KD, zu = (int)blockIdx.z / p.KD;
    // the two 64 x 64 tiles of the block: (unit u0, b tile bt0) and (u1, bt1); a unit past the end is a ghost (zeros, not stored)
    const int u0 = BIGB ? zu : 2 * zu, u1 = BIGB ? zu : 2 * zu + 1;
    const int bt0 = BIGB ? 2 * (int)blockIdx.x : (int)blockIdx.x, bt1 = BIGB ? bt0 + 1 : bt0;
    const int m0 = u0 / p.aTiles, m1u = u1 < p.nunits ? u1 / p.aTiles : m0;
    const bf16_t* const A0base = t3_member(p, m0) + (u0 - m0 * p.aTiles) * 64;
    const bf16_t* const A1base = t3_member(p, m1u) + (u1 - m1u * p.aTiles) * 64;
    const long long dA1 = (const char*)A1base - (const char*)A0base;          // second A tile = first + this many bytes
    const bool ghost1 = !BIGB && u1 >= p.nunits;
    constexpr unsigned OOB = 0x80000000u;
    const unsigned lds0 = (unsigned)(unsigned long long)(lptr_t)smem;

    // ---- this lane's LDS-DMA pieces (tile invariant).  A tile row r = hh * AWt + ww holds 64 channels (128 bytes); its two
    //      64-byte halves swap when bit 1 of the tile column ww is set (B: of the voxel slot).  Stage = [A tiles][B tiles].
    //      Every wave issues 4 pieces (1 KB) per stage, kinds fixed per slot; a slot past the tiles fetches nothing (offset out
    //      of range) into a scratch KB behind the stages: one compile-time vmcnt for all waves, no branches. ----
    unsigned vo[NP]; int pk[NP]; int dst[NP]; int second[NP];
    const unsigned trash = lds0 + (unsigned)(p.stages * stage_bytes);
#pragma unroll
    for (int it = 0; it < NP; ++it) {
        vo[it] = OOB; pk[it] = 0;
        if (it < A_ITS) {
            const int q = wave + T3_WAVES * it;                // piece of the A tiles
            const int t1 = q >= nA ? 1 : 0, ql = q - t1 * nA;
            const bool real = q < nAtot;
            dst[it] = real ? q * 1024 : -1; second[it] = t1;
            const int s = ql * 64 + lane, row = s >> 3, slp = s & 7;
            const int pl = row / PLANE, rp = row - pl * PLANE;  // (stride 1: one plane)
            const int hp = rp / AWt, wp = rp - hp * AWt;
            const int sl = slp ^ (((wp >> 1) & 1) << 2);
            const int hh = STR == 1 ? hp : 2 * hp + (pl >> 1), ww = STR == 1 ? wp : 2 * wp + (pl & 1);      // input row / column of the tile
            pk[it] = hh | (ww << 8);
            if (real && row < arows) vo[it] = (unsigned)(((hh * p.AW + ww) * p.CA + sl * 8) * 2);
        } else {
            const int q = wave + T3_WAVES * (it - A_ITS);      // piece of the B tiles
            const int t1 = q >= nB ? 1 : 0, ql = q - t1 * nB;
            const bool real = q < NTB * nB;
            dst[it] = real ? (nAtot + q) * 1024 : -1; second[it] = t1;
            const int s = ql * 64 + lane, kk = s >> 3, slp = s & 7;
            const int sl = slp ^ (((kk >> 1) & 1) << 2);
            const int th = kk / KWs, tw = kk - th * KWs;
            pk[it] = th;
            if (real && kk < TH * KWs) vo[it] = (unsigned)(((th * p.BW + tw) * p.CB + (t1 ? bt1 : bt0) * 64 + sl * 8) * 2);
        }
    }

    // ---- tile table: the geometry of every K-tile of this block (buffer-resource words of its A and B tile origins, the tile's
    //      position for the halo tests), computed ONCE, in parallel over the threads, into LDS (32 bytes per tile). ----
    const int my_tiles = (p.ntiles - (int)blockIdx.y + p.nsplit - 1) / p.nsplit;
    int* const tab = reinterpret_cast<int*>(smem + p.stages * stage_bytes + 1024);
    for (int t = tid; t < my_tiles + p.stages; t += T3_THREADS) {
        int e[8] = {0, 0, 0, 0, 0, 0, 0, 0};                   // (entries past the end: num_records 0 = every lane fetches zeros)
        if (t < my_tiles) {
            int r = (int)blockIdx.y + t * p.nsplit;
            const int twi = r % p.tiles_w; r /= p.tiles_w;
            const int thi = r % p.tiles_h; r /= p.tiles_h;
            const int bd = r % p.BD, n = r / p.BD;
            const int ad = bd * p.sd + kd - p.pd, ah0 = thi * TH * STR - p.ph, aw0 = twi * KWs * STR - p.pw, bh0 = thi * TH;
            const long long alin0 = (((long long)n * p.AD + ad) * p.AH + ah0) * p.AW + aw0;
            const long long blin0 = (((long long)n * p.BD + bd) * p.BH + bh0) * p.BW + twi * KWs;
            const unsigned long long pa = (unsigned long long)(A0base + alin0 * p.CA), pb = (unsigned long long)(p.B + blin0 * p.CB);
            e[0] = (int)(unsigned)pa; e[1] = (int)((unsigned)(pa >> 32) & 0xffffu); e[2] = (unsigned)ad < (unsigned)p.AD ? 0x7fffffff : 0;
            e[3] = (ah0 & 0xffff) | (aw0 << 16);               // (the 4th resource word is a constant: the slot carries the tile position)
            e[4] = (int)(unsigned)pb; e[5] = (int)((unsigned)(pb >> 32) & 0xffffu); e[6] = 0x7fffffff; e[7] = bh0;
        }
        reinterpret_cast<int4*>(tab)[2 * t] = make_int4(e[0], e[1], e[2], e[3]);
        reinterpret_cast<int4*>(tab)[2 * t + 1] = make_int4(e[4], e[5], e[6], e[7]);
    }
    __syncthreads();
    int q_e = 0;                                               // table entry of the NEXT tile to issue
    int4 ea, eb;
    auto fetch = [&]() {                                       // (issued ahead of the stage wait + barrier: the latency hides there)
        ea = reinterpret_cast<const int4*>(tab)[2 * q_e]; eb = reinterpret_cast<const int4*>(tab)[2 * q_e + 1];
        ++q_e;
    };
    const unsigned dA1lo = (unsigned)(unsigned long long)dA1; const int dA1hi = (int)(dA1 >> 32);
    auto issue = [&](int st) {
        i32x4_t ra0, ra1, rb;
        ra0.x = __builtin_amdgcn_readfirstlane(ea.x); ra0.y = __builtin_amdgcn_readfirstlane(ea.y);
        ra0.z = __builtin_amdgcn_readfirstlane(ea.z); ra0.w = 0x00020000;
        rb.x = __builtin_amdgcn_readfirstlane(eb.x); rb.y = __builtin_amdgcn_readfirstlane(eb.y);
        rb.z = __builtin_amdgcn_readfirstlane(eb.z); rb.w = 0x00020000;
        ra1 = ra0;
        if (!BIGB) {
            const unsigned long long b1 = (((unsigned long long)(unsigned)ra0.y << 32) | (unsigned)ra0.x) + (((unsigned long long)(unsigned)dA1hi << 32) | dA1lo);
            ra1.x = (int)(unsigned)b1; ra1.y = (int)((unsigned)(b1 >> 32) & 0xffffu);
            ra1.z = ghost1 ? 0 : ra0.z;
        }
        const int ah0 = (ea.w << 16) >> 16, aw0 = ea.w >> 16, bh0 = eb.w;      // (vector registers: every lane read the same entry)
        const unsigned S0 = lds0 + (unsigned)(st * stage_bytes);
#pragma unroll
        for (int it = 0; it < NP; ++it) {
            const unsigned d = dst[it] >= 0 ? S0 + (unsigned)dst[it] : trash;
            if (it < A_ITS) {   // halo rows / columns outside the volume fetch zeros (per-lane compares: vector work, not scalar)
                const unsigned hh = (unsigned)(pk[it] & 0xff), ww = (unsigned)(pk[it] >> 8);
                unsigned o = (hh + (unsigned)ah0) < (unsigned)p.AH ? vo[it] : OOB;
                o = (ww + (unsigned)aw0) < (unsigned)p.AW ? o : OOB;
                t3_dma((!BIGB && second[it]) ? ra1 : ra0, d, o);
            } else {
                t3_dma(rb, d, (unsigned)(pk[it] + bh0) < (unsigned)p.BH ? vo[it] : OOB);
            }
        }
    };

    // ---- fragment read addresses.  32x32x16 operand: lane l supplies column l & 31, k = 8 (l >> 5) .. +7; a transpose read
    //      serves [4 voxels][16 channels] per 16-lane group: lane (g2 = l >> 4, i = l & 15) points at voxel slot
    //      kk = 16 ks + 8 (g2 >> 1) + 4 h + (i >> 2), 16-channel sub-block g2 & 1, 8-byte piece i & 3 ----
    const int g2 = lane >> 4, i = lane & 15, kg = g2 >> 1;
    const int cpart = (g2 & 1) * 32 + (i & 3) * 8;
    const unsigned char* const At = smem + (BIGB ? 0 : wsel * nA * 1024);                  // this wave's A tile / B tile of a stage
    const unsigned char* const Bt = smem + nAtot * 1024 + (BIGB ? wsel * nB * 1024 : 0);
    // KWS != 0: voxel slot -> (row th, column tw) splits into a per-lane and a compile-time part (both multiples of 4 columns
    // apart, so the half swap -- bit 1 of the column -- is the lane's): address = lane base [kw] + constant (ks, h)
    const unsigned char* aL[3]; const unsigned char* bL[2];
    unsigned a_tab[KWS ? 1 : 4][2];      // KWS == 0: per-lane byte offsets of the A rows of (ks, h), tap kw = 0
    if (KWS) {
        const int th_l = KWS == 8 ? kg : 0, tw_l = (KWS == 8 ? 0 : 8 * kg) + (i >> 2);
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            // tap (kh, kw) of voxel (th, tw): stride 1 row (th + kh, tw + kw); stride 2 plane (kh & 1, kw & 1), row (th + kh / 2, tw + kw / 2)
            const int pl = STR == 1 ? 0 : (kh & 1) * 2 + (kw & 1);
            const int hp = th_l + (STR == 1 ? kh : kh >> 1), wp = tw_l + (STR == 1 ? kw : kw >> 1);
            aL[kw] = At + (pl * PLANE + hp * AWt + wp) * 128 + (((wa ^ (wp >> 1)) & 1) << 6) + cpart;
        }
    } else {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int kk = ks * 16 + 8 * kg + 4 * h + (i >> 2);
                const bool real = kk < TH * KWs;                 // (empty slots of a short tile: B is zero there, A reads any valid row)
                const int th = real ? kk / KWs : 0, tw = real ? kk - th * KWs : (i >> 2);
                const int r0 = STR == 1 ? (th + kh) * AWt + tw : ((kh & 1) * 2 * PLANE + (th + (kh >> 1)) * AWt + tw);
                a_tab[KWS ? 0 : ks][h] = (unsigned)(r0 * 128 + (((wa ^ (tw >> 1)) & 1) << 6) + cpart);
            }
        aL[0] = aL[1] = aL[2] = At;
    }
    {   // B rows are the voxel slots themselves: slot kk = per-lane part 8 kg + (i >> 2) + constant 16 ks + 4 h, whatever KWs
        const int kk_l = 8 * kg + (i >> 2);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) bL[nb] = Bt + kk_l * 128 + (((nb ^ (kk_l >> 1)) & 1) << 6) + cpart;
    }
    // generic tables: tap kw = row + kw: +128 bytes; the half swap flips with bit 1 of the column: always for kw = 2, for
    // kw = 1 iff the column is odd -- a lane's columns all have the parity of (i >> 2) (KWs is a multiple of 4)
    const unsigned m1 = ((i >> 2) & 1) ? 64u : 0u;

    f32x16_t acc[3][2];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][nb][e] = 0.f;
    // this wave's 64 x 64 tile: unit uw (member mw, first channel aw of the member), b tile btw
    const int uw = wsel ? u1 : u0, btw = wsel ? bt1 : bt0;
    const int mw = wsel ? m1u : m0, aw = (uw - mw * p.aTiles) * 64, b0 = btw * 64;
    const bool do_bsum = p.want_bsum && uw == 0 && kd == 0 && w6 == 0;
    float accb0 = 0.f, accb1 = 0.f;

    // ---- S-deep pipeline: while tile j is on the MFMAs, tiles j+1 .. j+S-1 are in flight ----
    const int S = p.stages;
    for (int s = 0; s < S - 1; ++s) { fetch(); issue(s); }
    int st = 0;
    for (int kt = blockIdx.y; kt < p.ntiles; kt += p.nsplit) {
        fetch();
        t3_wait_stages<NP>(S);                                 // this wave's pieces of tile kt have landed ...
        __builtin_amdgcn_s_barrier();                          // ... and everybody's; everybody is also done with tile kt - nsplit
        int stn = st + S - 1; if (stn >= S) stn -= S;
        issue(stn);                                            // refill the buffer tile kt - nsplit was read from
        const int sb = st * stage_bytes;
        const unsigned char* a0p = aL[0] + sb; const unsigned char* a1p = aL[1] + sb; const unsigned char* a2p = aL[2] + sb;
        const unsigned char* b0p = bL[0] + sb; const unsigned char* b1p = bL[1] + sb;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8_t bf0, bf1, af0, af1, af2;
            const int bo0 = ks * 16 * 128, bo1 = bo0 + 4 * 128;
            bf0 = t3_frag(t3_tr(b0p + bo0), t3_tr(b0p + bo1));
            bf1 = t3_frag(t3_tr(b1p + bo0), t3_tr(b1p + bo1));
            if (KWS) {
                constexpr int AW_ = KWS + 2;
                const int th_c = KWS == 8 ? 2 * ks : (KWS == 16 ? ks : ks >> 1), tw_c = KWS == 32 ? 16 * (ks & 1) : 0;
                const int ao0 = (th_c * AW_ + tw_c) * 128, ao1 = ao0 + 4 * 128;
                af0 = t3_frag(t3_tr(a0p + ao0), t3_tr(a0p + ao1));
                af1 = t3_frag(t3_tr(a1p + ao0), t3_tr(a1p + ao1));
                af2 = t3_frag(t3_tr(a2p + ao0), t3_tr(a2p + ao1));
            } else {
                const unsigned char* s0 = At + sb;
                unsigned aA = a_tab[KWS ? 0 : ks][0], aB = a_tab[KWS ? 0 : ks][1];
                asm volatile("" : "+v"(aA), "+v"(aB));       // (keeps the 16 derived tap addresses out of loop-invariant registers)
                af0 = t3_frag(t3_tr(s0 + aA), t3_tr(s0 + aB));
                if (STR == 1) af1 = t3_frag(t3_tr(s0 + ((aA + 128u) ^ m1)), t3_tr(s0 + ((aB + 128u) ^ m1)));
                else af1 = t3_frag(t3_tr(s0 + aA + (unsigned)(PLANE * 128)), t3_tr(s0 + aB + (unsigned)(PLANE * 128)));   // the odd-column plane, same row
                if (STR == 1) af2 = t3_frag(t3_tr(s0 + ((aA + 256u) ^ 64u)), t3_tr(s0 + ((aB + 256u) ^ 64u)));
                else af2 = t3_frag(t3_tr(s0 + ((aA + 128u) ^ m1)), t3_tr(s0 + ((aB + 128u) ^ m1)));                     // the even plane, next column
            }
            if (do_bsum) { accb0 += t3_sum8(bf0); accb1 += t3_sum8(bf1); }
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af0, bf0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af0, bf1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af1, bf0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af1, bf1, acc[1][1], 0, 0, 0);
            acc[2][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af2, bf0, acc[2][0], 0, 0, 0);
            acc[2][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af2, bf1, acc[2][1], 0, 0, 0);
        }
        if (++st == S) st = 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the padding stages of the tail

    // ---- D[a][b] of a 32x32 tile: lane holds b = lane & 31, a = (e & 3) + 8 (e >> 2) + 4 (lane >> 5) ----
    if (uw >= p.nunits) return;                                // ghost tile (odd number of units)
    float* Rx = p.Rx + (long long)mw * p.rx_mem + (long long)blockIdx.y * p.rx_stride;
    if (do_bsum) {
        accb0 += __shfl_xor(accb0, 32); accb1 += __shfl_xor(accb1, 32);
        if (lane < 32) { Rx[p.rx_bias + b0 + lane] = accb0; Rx[p.rx_bias + b0 + 32 + lane] = accb1; }
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const long long tap = (long long)(kd * 3 + kh) * 3 + t;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const int b = b0 + nb * 32 + (lane & 31);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int a = aw + wa * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                Rx[(tap * p.CA + a) * p.CB + b] = acc[t][nb][e];
            }
        }
    }
#endif
}

// ------------------------------------------------------------------------------------------------------------------------------
// fp32 variant (the C5 configuration, BASELINE.json configs[3]): the same block (12 waves, 9 taps of one kd slice of two 64 x 64
// channel tiles, haloed X tile + dY tiles staged once per K-tile by LDS-DMA, tile table, fixed piece slots, per-split partial
// copies) on v_mfma_f32_32x32x2_f32.  An fp32 MFMA does 1/16 of the work of a bf16 one per cycle, so the kernel is bound by MFMA
// issue with room to spare everywhere else: rows are 256 bytes (64 channels), fragments are plain ds_read_b32 (lane = channel,
// lanes 0-31 / 32-63 = two consecutive voxels: 32 consecutive dwords per half wave, conflict-free without a swizzle), two stages.
// Stride 1, K-tile columns 8 / 16 / 32.  It replaces the register-transposing wgrad_mfma_kernel<float> (16-42 % MFMA-busy, half of
// the C5 step) on the layers with >= 64 channels on both sides.
template <int KWS, bool BIGB>
__global__ void __launch_bounds__(T3_THREADS, 3) wgrad_t3f_kernel(T3P p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wsel = wave >= 6 ? 1 : 0, w6 = wave - 6 * wsel;
    const int wa = w6 & 1, kh = w6 >> 1;
    constexpr int KWs = KWS, TH = T3_KT / KWS, AWt = KWs + 2, AHt = TH + 2, arows = AHt * AWt;
    constexpr int nA = (arows + 3) / 4;                        // 1 KB pieces of ONE A tile (rows of 256 bytes)
    constexpr int nB = T3_KT / 4;                              // ... of one B tile
    constexpr int NTA = BIGB ? 1 : 2, NTB = BIGB ? 2 : 1;
    constexpr int nAtot = NTA * nA, nBtot = NTB * nB;
    constexpr int A_ITS = (nAtot + T3_WAVES - 1) / T3_WAVES, B_ITS = (nBtot + T3_WAVES - 1) / T3_WAVES, NP = A_ITS + B_ITS;
    constexpr int stage_bytes = (nAtot + nBtot) * 1024;
    const float* const pA0 = reinterpret_cast<const float*>(p.A0);
    const int kd = (int)blockIdx.z % p.KD, zu = (int)blockIdx.z / p.KD;
    const int u0 = BIGB ? zu : 2 * zu, u1 = BIGB ? zu : 2 * zu + 1;
    const int bt0 = BIGB ? 2 * (int)blockIdx.x : (int)blockIdx.x, bt1 = BIGB ? bt0 + 1 : bt0;
    const int m0 = u0 / p.aTiles, m1u = u1 < p.nunits ? u1 / p.aTiles : m0;
    const float* const A0base = reinterpret_cast<const float*>(t3_member(p, m0)) + (u0 - m0 * p.aTiles) * 64;
    const float* const A1base = reinterpret_cast<const float*>(t3_member(p, m1u)) + (u1 - m1u * p.aTiles) * 64;
    const long long dA1 = (const char*)A1base - (const char*)A0base;
    const bool ghost1 = !BIGB && u1 >= p.nunits;
    constexpr unsigned OOB = 0x80000000u;
    const unsigned lds0 = (unsigned)(unsigned long long)(lptr_t)smem;

    unsigned vo[NP]; int pk[NP]; int dst[NP]; int second[NP];
    const unsigned trash = lds0 + (unsigned)(p.stages * stage_bytes);
#pragma unroll
    for (int it = 0; it < NP; ++it) {
        vo[it] = OOB; pk[it] = 0;
        if (it < A_ITS) {
            const int q = wave + T3_WAVES * it;
            const int t1 = q >= nA ? 1 : 0, ql = q - t1 * nA;
            const bool real = q < nAtot;
            dst[it] = real ? q * 1024 : -1; second[it] = t1;
            const int s = ql * 64 + lane, row = s >> 4, sl = s & 15;           // 16 slots of 16 bytes per row
            const int hh = row / AWt, ww = row - hh * AWt;
            pk[it] = hh | (ww << 8);
            if (real && row < arows) vo[it] = (unsigned)(((hh * p.AW + ww) * p.CA + sl * 4) * 4);
        } else {
            const int q = wave + T3_WAVES * (it - A_ITS);
            const int t1 = q >= nB ? 1 : 0, ql = q - t1 * nB;
            const bool real = q < nBtot;
            dst[it] = real ? (nAtot + q) * 1024 : -1; second[it] = t1;
            const int s = ql * 64 + lane, kk = s >> 4, sl = s & 15;
            const int th = kk / KWs, tw = kk - th * KWs;
            pk[it] = th;
            if (real) vo[it] = (unsigned)(((th * p.BW + tw) * p.CB + (t1 ? bt1 : bt0) * 64 + sl * 4) * 4);
        }
    }

    const int my_tiles = (p.ntiles - (int)blockIdx.y + p.nsplit - 1) / p.nsplit;
    int* const tab = reinterpret_cast<int*>(smem + p.stages * stage_bytes + 1024);
    for (int t = tid; t < my_tiles + p.stages; t += T3_THREADS) {
        int e[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (t < my_tiles) {
            int r = (int)blockIdx.y + t * p.nsplit;
            const int twi = r % p.tiles_w; r /= p.tiles_w;
            const int thi = r % p.tiles_h; r /= p.tiles_h;
            const int bd = r % p.BD, n = r / p.BD;
            const int ad = bd + kd - p.pd, ah0 = thi * TH - p.ph, aw0 = twi * KWs - p.pw, bh0 = thi * TH;
            const long long alin0 = (((long long)n * p.AD + ad) * p.AH + ah0) * p.AW + aw0;
            const long long blin0 = (((long long)n * p.BD + bd) * p.BH + bh0) * p.BW + twi * KWs;
            const unsigned long long pa = (unsigned long long)(A0base + alin0 * p.CA);
            const unsigned long long pb = (unsigned long long)(reinterpret_cast<const float*>(p.B) + blin0 * p.CB);
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
    const unsigned dA1lo = (unsigned)(unsigned long long)dA1; const int dA1hi = (int)(dA1 >> 32);
    auto issue = [&](int st) {
        i32x4_t ra0, ra1, rb;
        ra0.x = __builtin_amdgcn_readfirstlane(ea.x); ra0.y = __builtin_amdgcn_readfirstlane(ea.y);
        ra0.z = __builtin_amdgcn_readfirstlane(ea.z); ra0.w = 0x00020000;
        rb.x = __builtin_amdgcn_readfirstlane(eb.x); rb.y = __builtin_amdgcn_readfirstlane(eb.y);
        rb.z = __builtin_amdgcn_readfirstlane(eb.z); rb.w = 0x00020000;
        ra1 = ra0;
        if (!BIGB) {
            const unsigned long long b1 = (((unsigned long long)(unsigned)ra0.y << 32) | (unsigned)ra0.x) + (((unsigned long long)(unsigned)dA1hi << 32) | dA1lo);
            ra1.x = (int)(unsigned)b1; ra1.y = (int)((unsigned)(b1 >> 32) & 0xffffu);
            ra1.z = ghost1 ? 0 : ra0.z;
        }
        const int ah0 = (ea.w << 16) >> 16, aw0 = ea.w >> 16, bh0 = eb.w;
        const unsigned S0 = lds0 + (unsigned)(st * stage_bytes);
#pragma unroll
        for (int it = 0; it < NP; ++it) {
            const unsigned d = dst[it] >= 0 ? S0 + (unsigned)dst[it] : trash;
            if (it < A_ITS) {
                const unsigned hh = (unsigned)(pk[it] & 0xff), ww = (unsigned)(pk[it] >> 8);
                unsigned o = (hh + (unsigned)ah0) < (unsigned)p.AH ? vo[it] : OOB;
                o = (ww + (unsigned)aw0) < (unsigned)p.AW ? o : OOB;
                t3_dma((!BIGB && second[it]) ? ra1 : ra0, d, o);
            } else {
                t3_dma(rb, d, (unsigned)(pk[it] + bh0) < (unsigned)p.BH ? vo[it] : OOB);
            }
        }
    };

    // fragments: lane l = (voxel of the pair g = l >> 5, channel c = l & 31): one dword per operand and MFMA
    const int g = lane >> 5, c = lane & 31;
    const unsigned char* const At = smem + (BIGB ? 0 : wsel * nA * 1024);
    const unsigned char* const Bt = smem + nAtot * 1024 + (BIGB ? wsel * nB * 1024 : 0);
    const unsigned char* const aL = At + ((kh * AWt + g) * 64 + wa * 32 + c) * 4;      // tap (kh, 0) of voxel slot g
    const unsigned char* const bL = Bt + (g * 64 + c) * 4;

    f32x16_t acc[3][2];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][nb][e] = 0.f;
    const int uw = wsel ? u1 : u0, btw = wsel ? bt1 : bt0;
    const int mw = wsel ? m1u : m0, aw = (uw - mw * p.aTiles) * 64, b0 = btw * 64;
    const bool do_bsum = p.want_bsum && uw == 0 && kd == 0 && w6 == 0;
    float accb0 = 0.f, accb1 = 0.f;

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
#pragma unroll 8
        for (int k2 = 0; k2 < T3_KT / 2; ++k2) {               // voxel slots 2 k2 + g: row th = 2 k2 / KWs, column 2 k2 % KWs + g
            const int th = (2 * k2) / KWs, tw0 = (2 * k2) % KWs;
            const unsigned char* a = ap + (th * AWt + tw0) * 256;
            const unsigned char* b = bp + (2 * k2) * 256;
            const float bf0 = *reinterpret_cast<const float*>(b), bf1 = *reinterpret_cast<const float*>(b + 128);
            const float af0 = *reinterpret_cast<const float*>(a), af1 = *reinterpret_cast<const float*>(a + 256),
                        af2 = *reinterpret_cast<const float*>(a + 512);
            if (do_bsum) { accb0 += bf0; accb1 += bf1; }
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af0, bf0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af0, bf1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af1, bf0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af1, bf1, acc[1][1], 0, 0, 0);
            acc[2][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af2, bf0, acc[2][0], 0, 0, 0);
            acc[2][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af2, bf1, acc[2][1], 0, 0, 0);
        }
        if (++st == S) st = 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    if (uw >= p.nunits) return;
    float* Rx = p.Rx + (long long)mw * p.rx_mem + (long long)blockIdx.y * p.rx_stride;
    if (do_bsum) {
        accb0 += __shfl_xor(accb0, 32); accb1 += __shfl_xor(accb1, 32);
        if (lane < 32) { Rx[p.rx_bias + b0 + lane] = accb0; Rx[p.rx_bias + b0 + 32 + lane] = accb1; }
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const long long tap = (long long)(kd * 3 + kh) * 3 + t;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const int b = b0 + nb * 32 + (lane & 31);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int a = aw + wa * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                Rx[(tap * p.CA + a) * p.CB + b] = acc[t][nb][e];
            }
        }
    }
#endif
}

// fills the tile geometry; false = shape outside this kernel
static bool t3_plan(const WgradSpec& g, T3P& p) {
    int en = M1_CFG("M1_WG_T3", 1);
    if (!en || (g.dtype != M1_BF16 && g.dtype != M1_F32)) return false;
    const bool f32 = g.dtype == M1_F32;
    { int ef = M1_CFG("M1_WG_T3F", 1); if (f32 && !ef) return false; }
    if (g.CA < 64 || g.CB < 64 || g.CA % 64 || g.CB % 64) return false;
    if (!(g.kh == 3 && g.kw == 3 && (g.kd == 1 || g.kd == 3))) return false;
    const bool s1 = g.sh == 1 && g.sw == 1 && g.sd == 1, s2 = g.sh == 2 && g.sw == 2 && (g.sd == 1 || g.sd == 2) && g.ph == 0 && g.pw == 0;
    if (!s1 && !s2) return false;
    if (f32 && (!s1 || g.BW % 8)) return false;                // fp32 kernel: stride 1, K-tile columns 8 / 16 / 32
    if (s2 && g.CB % 128) return false;                        // (two de-interleaved A tiles do not fit a stage)
    if (g.BW % 4 || g.BW < 8) return false;
    // (DMA offsets are 32-bit and relative to the tile origin, which travels in the 64-bit resource base)
    const long long esz = f32 ? 4 : 2;
    if ((long long)(g.AH + 4) * g.AW * g.CA * esz >= (1ll << 31) - 4096 || (long long)(g.BH + 4) * g.BW * g.CB * esz >= (1ll << 31) - 4096) return false;
    p = T3P{};
    p.B = (const bf16_t*)g.B;
    p.CA = g.CA; p.CB = g.CB; p.AD = g.AD; p.AH = g.AH; p.AW = g.AW; p.BD = g.BD; p.BH = g.BH; p.BW = g.BW; p.N = g.N;
    p.pd = g.pd; p.ph = g.ph; p.pw = g.pw; p.KD = g.kd; p.sd = g.sd;
    // K-tile: TH rows x KWs columns <= 64 voxels, KWs a multiple of 4 (4-voxel transpose groups never straddle a row)
    int kws = 0;
    for (int c : {32, 16, 8}) {
        if (f32 && c == 32 && g.CB % 128) continue;            // (fp32, two X tiles per block: two stages of 32-column tiles exceed the LDS)
        if (g.BW % c == 0) { kws = c; break; }
    }
    if (!kws) { if (g.BW <= 32) kws = g.BW; else { for (int c = 28; c >= 8; c -= 4) if (g.BW % c == 0) { kws = c; break; } } }
    if (!kws) return false;
    p.KWs = kws; p.TH = T3_KT / kws;
    if (p.TH + 2 > 255 || p.KWs + 2 > 255) return false;
    const int nA = s1 ? ((p.TH + 2) * (p.KWs + 2) + 7) / 8 : (4 * (p.TH + 1) * (p.KWs + 2) + 7) / 8;
    if (!f32 && nA > (s1 ? 18 : 60)) return false;             // (the kernel's fixed piece slots: 24 / 36 / 60 pieces for the A tiles)
    if (f32 && !(kws == 8 || kws == 16 || kws == 32)) return false;
    if (2 * p.TH + 1 > 255 || 2 * p.KWs + 3 > 255) return false;
    p.tiles_w = g.BW / p.KWs; p.tiles_h = (g.BH + p.TH - 1) / p.TH;
    const long long nt = (long long)g.N * g.BD * p.tiles_h * p.tiles_w;
    if (nt >= (1ll << 30) || nt < 8) return false;
    p.ntiles = (int)nt;
    return true;
}
bool m1_t3_wgrad_supported(const WgradSpec& g) { T3P p; return t3_plan(g, p); }

// `nmem` members of one Conv3D concat at once (nmem = 1: Am / a_offs may be null): member m = (Am[m], a_off a_offs[m]), all with
// g.CA channels; copies of member m live at g.rx + m * rx_mem
int m1_t3_wgrad(const WgradSpec& g, long long nw, int nb, hipStream_t st, int nmem, const void* const* Am, const int* a_offs, long long rx_mem) {
    T3P p;
    if (!t3_plan(g, p)) return M1_ERR_UNSUPPORTED;
    if (nmem < 1 || nmem > M1_MAX_SRC) return M1_ERR_UNSUPPORTED;
    const bf16_t* am[M1_MAX_SRC] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    for (int m = 0; m < nmem; ++m) am[m] = (const bf16_t*)(nmem > 1 ? Am[m] : g.A);
    p.A0 = am[0]; p.A1 = am[1]; p.A2 = am[2]; p.A3 = am[3]; p.A4 = am[4]; p.A5 = am[5]; p.nmem = nmem;
    p.aTiles = g.CA / 64; p.nunits = p.aTiles * nmem;
    const bool bigb = g.CB % 128 == 0;                         // two dY tiles per block, else two units of the concat
    if (!bigb && p.nunits < 2) return M1_ERR_UNSUPPORTED;      // (a single 64 -> 64 tile: the per-tap kernel)
    const int gx = bigb ? g.CB / 128 : g.CB / 64, gzu = bigb ? p.nunits : (p.nunits + 1) / 2;
    const long long per_split = (long long)gx * gzu * g.kd;    // blocks per voxel split
    int tgt = M1_CFG("M1_T3_BLOCKS", 256);      // one block per CU
    long long nsplit = tgt / per_split; if (nsplit < 1) nsplit = 1;
    const long long nloc = (long long)g.kd * 9 * g.CA * g.CB;
    const long long stride = nloc + g.CB;                      // compact copy of one member's block (+ bias sums)
    if (!g.rx || g.rx_floats < stride) return M1_ERR_WORKSPACE;
    if (nsplit * stride > g.rx_floats) nsplit = g.rx_floats / stride;
    if (nsplit > p.ntiles / 8) nsplit = p.ntiles / 8;          // >= 8 K-tiles per block: its prologue and its partial tiles must amortise
    if (nsplit < 1) nsplit = 1;
    // too small to fill the chip at that depth (single layers of the (10,20,20) level): the per-tap kernel, with 27 x more
    // blocks per voxel split, is the better fit
    int minb = M1_CFG("M1_T3_MIN_BLOCKS", 128);
    if (nsplit * per_split < minb) return M1_ERR_UNSUPPORTED;
    p.nsplit = (int)nsplit;
    p.Rx = g.rx; p.rx_stride = stride; p.rx_bias = nloc; p.rx_mem = nmem > 1 ? rx_mem : 0;
    p.want_bsum = g.bsum != nullptr;
    // LDS: the stages, the scratch KB of the empty piece slots, the tile table (32 bytes per K-tile of a block + the padding stages)
    const bool s2 = g.sh == 2, f32 = g.dtype == M1_F32;
    const int nA = f32 ? ((p.TH + 2) * (p.KWs + 2) + 3) / 4 : (!s2 ? ((p.TH + 2) * (p.KWs + 2) + 7) / 8 : (4 * (p.TH + 1) * (p.KWs + 2) + 7) / 8);
    const int stage_bytes = f32 ? (bigb ? nA + 32 : 2 * nA + 16) * 1024 : (bigb ? nA + 16 : 2 * nA + 8) * 1024;
    const long long tiles_per_block = (p.ntiles + nsplit - 1) / nsplit;
    int S = f32 ? 3 : 4;
    { int fs = M1_CFG("M1_T3_STAGES", 0); if (fs >= 2 && fs <= 5) S = fs; }
    if (f32 && S > 3) S = 3;
    while (S >= 2 && (size_t)S * stage_bytes + 1024 + (size_t)(tiles_per_block + S) * 32 > 160 * 1024) --S;
    if (S < 2) return M1_ERR_UNSUPPORTED;
    p.stages = S;
    const size_t smem = (size_t)S * stage_bytes + 1024 + (size_t)(tiles_per_block + S) * 32;
    void (*kern)(T3P) = nullptr;
    if (f32) kern = bigb ? (p.KWs == 8 ? wgrad_t3f_kernel<8, true> : (p.KWs == 16 ? wgrad_t3f_kernel<16, true> : wgrad_t3f_kernel<32, true>))
                         : (p.KWs == 8 ? wgrad_t3f_kernel<8, false> : (p.KWs == 16 ? wgrad_t3f_kernel<16, false> : wgrad_t3f_kernel<32, false>));
    else if (s2) kern = p.KWs == 8 ? wgrad_t3_kernel<8, true, 2> : (p.KWs == 16 ? wgrad_t3_kernel<16, true, 2> : (p.KWs == 32 ? wgrad_t3_kernel<32, true, 2> : wgrad_t3_kernel<0, true, 2>));
    else if (bigb) kern = p.KWs == 8 ? wgrad_t3_kernel<8, true, 1> : (p.KWs == 16 ? wgrad_t3_kernel<16, true, 1> : (p.KWs == 32 ? wgrad_t3_kernel<32, true, 1> : wgrad_t3_kernel<0, true, 1>));
    else kern = p.KWs == 8 ? wgrad_t3_kernel<8, false, 1> : (p.KWs == 16 ? wgrad_t3_kernel<16, false, 1> : (p.KWs == 32 ? wgrad_t3_kernel<32, false, 1> : wgrad_t3_kernel<0, false, 1>));
    {
        static const void* done[20]; static int ndone = 0;
        bool seen = false;
        for (int q = 0; q < ndone; ++q) seen |= done[q] == (const void*)kern;
        if (!seen) {
            if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return M1_ERR_LAUNCH;
            if (ndone < 20) done[ndone++] = (const void*)kern;
        }
    }
    m1_note_kernel(f32 ? "wgrad_t3f:kws%d:big%d" : (s2 ? "wgrad_t3:s2:kws%d:big%d" : "wgrad_t3:kws%d:big%d"), p.KWs, (int)bigb);
    hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)nsplit, (unsigned)(gzu * g.kd)), dim3(T3_THREADS), smem, st, p);
    int rc = m1_check_launch(); if (rc) return rc;
    for (int m = 0; m < nmem; ++m) {                      // one fold per member (its own block of R; the bias sums ride on member 0)
        WgradSpec gm = g;
        if (nmem > 1) gm.a_off = a_offs[m];
        if (m) gm.bsum = nullptr;
        rc = m1_wg_rx_finish(p.Rx + (long long)m * p.rx_mem, stride, (int)nsplit, gm, nloc, st); if (rc) return rc;
    }
    return M1_OK;
}
