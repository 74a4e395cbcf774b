// wgrad_tf.hip -- tap-fused weight gradients (bf16) for the wide, shallow layers (res0..res2: many voxels, <= 32
// channels per tile), where the per-tap kernel of wgrad_mfma.hip re-reads both operands once per tap.
//
//   R[tap][a][b] += sum_{n,v} A[n, v*s + tap - p][a] * B[n, v][b]            all taps of a (1,3,3)/(3,3,3) kernel at once
//
// * One block owns a 32(a) x 32(b) channel tile and walks K-tiles of 64 or 128 output voxels (2 or 4 k-steps of 32: rows x
//   KWs columns, KWs = 32/16/8 dividing the row length; 128 for the (1,3,3) kernels, see tf_plan).  Per K-tile it stages, by LDS-DMA (global_load_lds_dwordx4,
//   no staging registers), the 64 B rows and the A rows of the tile INCLUDING the tap halo -- every A voxel is loaded
//   once for all taps -- in their natural voxel-major layout.
// * The MFMA wants K(=voxel)-contiguous fragments: ds_read_b64_tr_b16 (gfx950 transpose read) takes a
//   [4 voxels][16 channels] block per 16-lane group and hands lane i channel i of the 4 voxels, so a tap is just a
//   different ROW offset into the same A tile (no register transposes, no unaligned LDS reads).
//   Probe of the instruction's lane mapping: tools/probes/tr_b16_probe.hip.
// * wave w accumulates the 16x16 tile (w>>1, w&1) of every tap: NT x 4 accumulator registers; the B fragment of a
//   k-step is read once and reused by all taps.
// * K-tiles are dealt round-robin to the blocks of a channel tile; every block stores its partial tile into its own copy of the
//   member's gradient block and a fold (tf_finish_*, queued and run in batches: m1_wgrad_defer) adds the copies in a fixed
//   order -- no floating-point atomics.  The equal-width members of a concat share one launch (blockIdx.z = member).
#include "common.h"
#include "gather.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __attribute__((aligned(64))) unsigned int m1_zero_page_w[16];

__device__ __forceinline__ void glds16w(const void* g, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)lds_wave_base, 16, 0, 0);
}
// 4 voxels x 16 channels block -> this lane's channel, 4 consecutive voxels (see header)
__device__ __forceinline__ s16x4_t tr_read(const unsigned char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p);
}

#define TF_MAX_NKS 4        // k-steps (of 32 voxels) per K-tile: 2, or 4 for the <= 16-channel layers (see tf_plan)
#define TF_MAX_AIT 10       // LDS-DMA pieces per thread for the A tile

struct TfP {
    const bf16_t* A; const bf16_t* B; float* R; float* bsum;
    int CA, CB, AD, AH, AW, BD, BH, BW, N;
    long long RT, RSA; int a_off, b_off;
    int sd, sh, sw, pd, ph, pw;
    int KWs, TH;             // k-step = (32/KWs) rows x KWs columns; K-tile = TH = nks*(32/KWs) rows x KWs columns
    int AHt, AWt;            // A tile extent per kd slice (rows, columns) incl. halo
    int spra, sprb;          // 16-byte slots per LDS row = min(C, 32)/8
    int a_slots, b_slots;    // slots of the A / B tile (a_slots rounded up to whole waves)
    int a_bytes, b_bytes;    // LDS bytes per buffer (with slack for the 32-byte fragment reads of short rows)
    int tiles_w, tiles_h; long long ntiles;
    int bTiles, nsplit;
    int stages;              // LDS pipeline depth (tiles staged ahead of the MFMAs + 1)
    float* Rx; long long rx_stride;      // per-XCD private copies of R (+ bias sums behind it): see the epilogue
    long long rx_bias;       // offset of the bias sums inside one copy
    int nks;                 // k-steps per K-tile (template TF_NKS of the kernel)
    // several concat members of one conv in ONE launch (same geometry and channel count, blockIdx.z = member): a per-member launch
    // of ~256 blocks leaves one block per CU; member m reads Am[m] and stores into the copies at Rx + m * rx_mem
    int nmem; const bf16_t* Am[M1_MAX_SRC]; long long rx_mem;
};

typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
// transpose-read by inline asm: hipcc cannot tell an LDS-DMA still in flight from the buffer being read and would wait
// vmcnt(0) in front of every compiler-visible LDS read (no prefetch depth at all)
__device__ __forceinline__ u32x2_t tr_read_asm(unsigned lds_addr) {
    u32x2_t v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(lds_addr) : "memory");
    return v;
}
__device__ __forceinline__ void lds_wait2(u32x2_t& a, u32x2_t& b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ bf16x8_t frag8(u32x2_t lo, u32x2_t hi) {
    return __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(lo, hi, 0, 1, 2, 3));
}
// at most n LDS-DMA pieces of this wave still in flight (n wave-uniform)
__device__ __forceinline__ void wait_vm(int n) {
    switch (n) {
#define TF_W(N_) case N_: asm volatile("s_waitcnt vmcnt(" #N_ ")" ::: "memory"); break;
        TF_W(0) TF_W(1) TF_W(2) TF_W(3) TF_W(4) TF_W(5) TF_W(6) TF_W(7) TF_W(8) TF_W(9) TF_W(10) TF_W(11) TF_W(12)
        TF_W(13) TF_W(14) TF_W(15) TF_W(16) TF_W(17) TF_W(18) TF_W(19) TF_W(20) TF_W(21) TF_W(22) TF_W(23) TF_W(24)
        TF_W(25) TF_W(26) TF_W(27) TF_W(28) TF_W(29) TF_W(30) TF_W(31) TF_W(32) TF_W(33) TF_W(34) TF_W(35) TF_W(36)
        TF_W(37) TF_W(38) TF_W(39) TF_W(40) TF_W(41) TF_W(42) TF_W(43) TF_W(44) TF_W(45) TF_W(46) TF_W(47) TF_W(48)
#undef TF_W
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

#define TF_MAX_STAGES 8

template <int KD, int KH, int KW, int KPARTS, int TF_NKS>
__global__ void __launch_bounds__(256) wgrad_tf_kernel(TfP p) {
#if defined(__HIP_DEVICE_COMPILE__)      // (buffer-resource builtins do not exist in the host pass)
    constexpr int NT = KD * KH * KW, NG = KD * KH;           // taps; tap groups (the KW taps of one (kd,kh) share a row offset)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // waves -> (16x16 output tile, tap part): with <= 16 channels on a side there are only 2 or 1 tiles and the waves that
    // share a tile split the taps between them (tap t belongs to part t % kparts)
    constexpr int kparts = KPARTS, ntile = 4 / KPARTS;       // host: ntile = (CA > 16 ? 2 : 1) * (CB > 16 ? 2 : 1)
    const int ntb = p.CB > 16 ? 2 : 1;
    const int tile_id = wave % ntile, part = wave / ntile;
    const int ta = tile_id / ntb, tb = tile_id % ntb;
    const int a0 = (blockIdx.x / p.bTiles) * 32, b0 = (blockIdx.x % p.bTiles) * 32;
    const int mem = p.nmem > 1 ? (int)blockIdx.z : 0;
    const bf16_t* Abase = p.A;
    if (p.nmem > 1) {         // (a chain of uniform selects: a dynamic index would move the whole argument struct to scratch memory)
        Abase = p.Am[0];
#pragma unroll
        for (int m = 1; m < M1_MAX_SRC; ++m) if (mem == m) Abase = p.Am[m];
    }
    const int PA = p.spra * 16, PB = p.sprb * 16;            // LDS row pitch (bytes)
    const int stage_bytes = p.a_bytes + p.b_bytes;           // [A tile][B tile] per stage
    const unsigned char* zero_pg = reinterpret_cast<const unsigned char*>(m1_zero_page_w);

    // ---- per-lane description of its LDS-DMA pieces (the same for every K-tile).  Every wave issues nait + 1 pieces
    //      per stage (tiles padded to whole 256-slot rounds, padding fetches the zero page): static vmcnt arithmetic ----
    // (buffer loads: tile origin in the resource base, constant 32-bit lane offsets, 2^31 = out of range -> zeros)
    constexpr unsigned OOB = 0x80000000u;
    unsigned a_vo[TF_MAX_AIT]; int a_pk[TF_MAX_AIT];
    const int nait = p.a_slots >> 8;
#pragma unroll
    for (int it = 0; it < TF_MAX_AIT; ++it) {
        const int q = it * 256 + tid;
        const int row = q / p.spra, slp = q - row * p.spra;
        const int dd = row / (p.AHt * p.AWt); const int r2 = row - dd * (p.AHt * p.AWt);
        const int hh = r2 / p.AWt, ww = r2 - hh * p.AWt;
        // 64-byte rows: the two 32-byte halves of a row swap when bit 3 of the tile column is set -- lane groups 0/1 of a
        // transpose read sit 8 columns apart and would otherwise hit the same banks
        const int sl = p.spra == 4 ? (slp ^ (((ww >> 3) & 1) << 1)) : slp;
        a_pk[it] = dd | (hh << 8) | (ww << 16);
        a_vo[it] = (row < KD * p.AHt * p.AWt) ? (unsigned)((((dd * p.AH + hh) * p.AW + ww) * p.CA + a0 + sl * 8) * 2) : OOB;
    }
    constexpr int NBIT = (TF_NKS * 32 * 4 + 255) / 256;      // LDS-DMA pieces per thread for the B tile (max)
    const int nbit = (p.b_slots + 255) >> 8;
    unsigned b_vo[NBIT]; int b_th[NBIT];
#pragma unroll
    for (int it = 0; it < NBIT; ++it) {
        const int q = it * 256 + tid;
        const int row = q / p.sprb, slp = q - row * p.sprb;
        const int th = row / p.KWs, tw = row - th * p.KWs;
        const int sl = p.sprb == 4 ? (slp ^ (((tw >> 3) & 1) << 1)) : slp;
        b_th[it] = th;
        b_vo[it] = q < p.b_slots ? (unsigned)(((th * p.BW + tw) * p.CB + b0 + sl * 8) * 2) : OOB;
    }

    // tile counter of the NEXT tile to issue, decoded incrementally (no divisions in the loop)
    int q_kt = blockIdx.y, q_tw, q_th, q_bd, q_n;
    { int r = q_kt; q_tw = r % p.tiles_w; r /= p.tiles_w; q_th = r % p.tiles_h; r /= p.tiles_h; q_bd = r % p.BD; q_n = r / p.BD; }
    int s_tw, s_th, s_bd, s_n;
    { int r = p.nsplit; s_tw = r % p.tiles_w; r /= p.tiles_w; s_th = r % p.tiles_h; r /= p.tiles_h; s_bd = r % p.BD; s_n = r / p.BD; }
    auto issue = [&](int st) {
        const bool live = q_kt < (int)p.ntiles;
        const int twi = q_tw, thi = q_th, bd = q_bd, n = q_n;
        q_kt += p.nsplit;
        q_tw += s_tw; int c = q_tw >= p.tiles_w; q_tw -= c ? p.tiles_w : 0;
        q_th += s_th + c; c = q_th >= p.tiles_h; q_th -= c ? p.tiles_h : 0;
        q_bd += s_bd + c; c = q_bd >= p.BD; q_bd -= c ? p.BD : 0;
        q_n += s_n + c;
        const int ad0 = bd * p.sd - p.pd, ah0 = thi * p.TH * p.sh - p.ph, aw0 = twi * p.KWs * p.sw - p.pw;
        const long long lin0 = (((long long)n * p.AD + ad0) * p.AH + ah0) * p.AW + aw0;
        const int bh0 = thi * p.TH;
        const long long blin0 = (((long long)n * p.BD + bd) * p.BH + bh0) * p.BW + twi * p.KWs;
        unsigned char* As = smem + st * stage_bytes;
        unsigned char* Bs = As + p.a_bytes;
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)(Abase + lin0 * p.CA), 0, live ? 0x7fffffff : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(p.B + blin0 * p.CB), 0, live ? 0x7fffffff : 0, 0x00020000);
        const bool a_inner = ad0 >= 0 && ad0 + KD - 1 < p.AD && ah0 >= 0 && ah0 + p.AHt - 1 < p.AH && aw0 >= 0 && aw0 + p.AWt - 1 < p.AW;
        if (a_inner) {                                     // uniform: no per-lane work
#pragma unroll
            for (int it = 0; it < TF_MAX_AIT; ++it)
                if (it < nait) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lptr_t)(As + (it * 256 + wave * 64) * 16), 16, a_vo[it], 0, 0, 0);
        } else {
#pragma unroll
            for (int it = 0; it < TF_MAX_AIT; ++it) {
                if (it < nait) {
                    const int pk = a_pk[it];
                    const int dd = pk & 0xff, hh = (pk >> 8) & 0xff, ww = (pk >> 16) & 0xff;
                    const bool ok = (unsigned)(ad0 + dd) < (unsigned)p.AD && (unsigned)(ah0 + hh) < (unsigned)p.AH && (unsigned)(aw0 + ww) < (unsigned)p.AW;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lptr_t)(As + (it * 256 + wave * 64) * 16), 16, ok ? a_vo[it] : OOB, 0, 0, 0);
                }
            }
        }
        const bool b_inner = bh0 + p.TH <= p.BH;
#pragma unroll
        for (int it = 0; it < NBIT; ++it)
            if (it < nbit)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lptr_t)(Bs + (it * 256 + wave * 64) * 16), 16,
                                                         (b_inner || bh0 + b_th[it] < p.BH) ? b_vo[it] : OOB, 0, 0, 0);
    };

    // ---- fragment read addresses: lane (g = lane>>4, i = lane&15) supplies voxel 8g + 4h + (i>>2), 8-byte piece i&3 ----
    const int g = lane >> 4, i = lane & 15;
    const unsigned lds0 = (unsigned)(unsigned long long)(lptr_t)smem;
    unsigned a_ad[TF_NKS][2][KW], b_ad[TF_NKS][2];            // byte addresses inside stage 0 for tap row (kd,kh) = (0,0)
#pragma unroll
    for (int ks = 0; ks < TF_NKS; ++ks)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int kk = ks * 32 + 8 * g + 4 * h + (i >> 2);
            const int th = kk / p.KWs, tw = kk - th * p.KWs;
#pragma unroll
            for (int kw = 0; kw < KW; ++kw) {
                const int ww = tw * p.sw + kw;
                const int half = p.spra == 4 ? (ta ^ ((ww >> 3) & 1)) : ta;
                a_ad[ks][h][kw] = lds0 + ((th * p.sh) * p.AWt + ww) * PA + half * 32 + (i & 3) * 8;
            }
            const int halfb = p.sprb == 4 ? (tb ^ ((tw >> 3) & 1)) : tb;
            b_ad[ks][h] = lds0 + p.a_bytes + kk * PB + halfb * 32 + (i & 3) * 8;
        }
    const int grp_pitch = p.AWt * PA;                          // bytes between tap rows kh and kh+1 (kd: x AHt)

    f32x4_t acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const bool do_bsum = p.bsum != nullptr && a0 == 0 && ta == 0 && part == 0 && mem == 0;
    f32x4_t accb = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, make_uint4(0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u));

    // ---- S-deep pipeline: while tile j is on the MFMAs, tiles j+1 .. j+S-1 are in flight ----
    const int S = p.stages;
    const int npiece = nait + nbit;                            // LDS-DMA pieces per wave per stage
    const long long kt0 = blockIdx.y, step = p.nsplit;
    for (int s = 0; s < S - 1; ++s) issue(s);
    int st = 0;
    for (long long kt = kt0; kt < p.ntiles; kt += step) {
        wait_vm(npiece * (S - 2));                     // this wave's pieces of tile kt have landed ...
        __builtin_amdgcn_s_barrier();                          // ... and everybody's; everybody is also done with tile kt - step
        int stn = st + S - 1; if (stn >= S) stn -= S;
        issue(stn);                                            // refill the buffer tile kt - step was read from
        const unsigned sb = (unsigned)(st * stage_bytes);
        // fragment reads run one unit (= the KH*KW taps of one kd slice for one k-step) ahead of the MFMAs
        constexpr int NU = TF_NKS * KD, CH = KH * KW;
        u32x2_t bl[2], bh[2], al[2][CH], ah[2][CH];
        auto rd_unit = [&](int u, int set) {
            const int ks = u / KD, kd = u % KD;
            if (kd == 0) { bl[ks & 1] = tr_read_asm(b_ad[ks][0] + sb); bh[ks & 1] = tr_read_asm(b_ad[ks][1] + sb); }
#pragma unroll
            for (int kh = 0; kh < KH; ++kh) {
                const unsigned ro = sb + (unsigned)((kd * p.AHt + kh) * grp_pitch);
#pragma unroll
                for (int kw = 0; kw < KW; ++kw) {
                    if (((kd * CH + kh * KW + kw) % kparts) == part) {
                        al[set][kh * KW + kw] = tr_read_asm(a_ad[ks][0][kw] + ro);
                        ah[set][kh * KW + kw] = tr_read_asm(a_ad[ks][1][kw] + ro);
                    }
                }
            }
        };
        rd_unit(0, 0);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int ks = u / KD, kd = u % KD, set = u & 1;
            if (kd == 0) lds_wait2(bl[ks & 1], bh[ks & 1]);
#pragma unroll
            for (int c = 0; c < CH; ++c) lds_wait2(al[set][c], ah[set][c]);
            if (u + 1 < NU) rd_unit(u + 1, set ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            const bf16x8_t bfr = frag8(bl[ks & 1], bh[ks & 1]);
            if (do_bsum && kd == 0) accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, bfr, accb, 0, 0, 0);
#pragma unroll
            for (int c = 0; c < CH; ++c)
                if (((kd * CH + c) % kparts) == part)
                    acc[kd * CH + c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag8(al[set][c], ah[set][c]), bfr, acc[kd * CH + c], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (++st == S) st = 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the padding stages of the tail

    // ---- D[a][b]: lane holds a = 4*(lane>>4) + r, b = lane&15 ----
    // Float atomics are executed at the memory side on this multi-XCD part (~7 G requests/s in total, measured), which
    // would cost more than the whole K loop.  Every block instead STORES its partial tile into its own copy of R
    // (copy = K-split index); tf_finish_kernel adds the copies.
    float* Rx = p.Rx + (long long)mem * p.rx_mem + (long long)blockIdx.y * p.rx_stride;
    const int b = b0 + tb * 16 + i;
    if (do_bsum && g == 0 && b < p.CB) Rx[p.rx_bias + b] = accb[0];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int a = a0 + ta * 16 + g * 4 + r;
            if ((t % kparts) == part && a < p.CA && b < p.CB) Rx[((long long)t * p.CA + a) * p.CB + b] = acc[t][r];
        }
#endif
}


// R[idx(i)] += sum over the copies of Rx[copy][idx(i)] for the NT x CA x CB block of one concat member (+ its bias sums).
// A block owns EL consecutive elements; its 256/EL lane rows stride the copies, fold through LDS, and ONE lane adds the
// total into R -- fixed order, no atomics: the weight gradient of these layers is run-to-run deterministic.
struct TfFin { float* Rx; long long stride; int ncopies; float* R; int NT, CA, CB; long long RT, RSA; int a_off, b_off;
               float* bsum; long long rx_bias; int nb; int EL; int bias_serial; };
// (bodies take the block index / block count of THEIR fold, so that one launch can serve many folds: tf_finish_batch_kernel)
__device__ __forceinline__ void tf_fold_generic(const TfFin& f, unsigned vb, float* red) {
    const long long n = (long long)f.NT * f.CA * f.CB;
    const int e = threadIdx.x % f.EL, y = threadIdx.x / f.EL, YL = 256 / f.EL;
    const long long i = (long long)vb * f.EL + e;
    const long long idx = i;                              // the copies are compact: [tap][a][b] of this member, then CB bias sums
    float* dst = nullptr;
    if (i < n) {
        const int b = (int)(i % f.CB); const long long q = i / f.CB; const int a = (int)(q % f.CA), t = (int)(q / f.CA);
        dst = f.R + (long long)t * f.RT + (long long)(a + f.a_off) * f.RSA + b + f.b_off;
    } else if (i < n + f.nb) { dst = f.bsum + (i - n) + f.b_off; }
    float s = 0.f;
    if (dst) {
#pragma unroll 8
        for (int c = y; c < f.ncopies; c += YL) s += f.Rx[(long long)c * f.stride + idx];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (y == 0 && dst) {
        for (int q = 1; q < YL; ++q) s += red[q * f.EL + e];
        *dst += s;
    }
}
__global__ void __launch_bounds__(256) tf_finish_kernel(TfFin f) {
    __shared__ float red[256];
    tf_fold_generic(f, blockIdx.x, red);
}

// Few copies of a large block (deep layers: 2..16 voxel splits of a multi-MB weight block): one thread folds 4 consecutive
// elements over all copies with 16-byte loads -- bandwidth-bound, where the kernel above (built for hundreds of copies of a
// small block) spends its time in 55k nearly empty blocks.  Same fixed order of additions.
__device__ __forceinline__ void tf_fold_vec(const TfFin& f, unsigned vb, unsigned nblk) {
    const long long n = (long long)f.NT * f.CA * f.CB, n4 = n >> 2;
    for (long long q = (long long)vb * 256 + threadIdx.x; q < n4; q += (long long)nblk * 256) {
        const long long i = q << 2;
        float4 s = *reinterpret_cast<const float4*>(f.Rx + i);
        for (int c = 1; c < f.ncopies; ++c) {
            const float4 v = *reinterpret_cast<const float4*>(f.Rx + (long long)c * f.stride + i);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        const int b = (int)(i % f.CB); const long long r = i / f.CB; const int a = (int)(r % f.CA), t = (int)(r / f.CA);
        float* dst = f.R + (long long)t * f.RT + (long long)(a + f.a_off) * f.RSA + b + f.b_off;
        dst[0] += s.x; dst[1] += s.y; dst[2] += s.z; dst[3] += s.w;
    }
    if (vb == 0)
        for (int j = threadIdx.x; j < f.nb; j += 256) {
            float s = 0.f;
            for (int c = 0; c < f.ncopies; ++c) s += f.Rx[(long long)c * f.stride + n + j];
            f.bsum[j + f.b_off] += s;
        }
}
__global__ void __launch_bounds__(256) tf_finish_vec_kernel(TfFin f) { tf_fold_vec(f, blockIdx.x, gridDim.x); }

// Many copies (tens to hundreds of voxel splits): a block owns 256 consecutive elements, its 4 waves take every 4th copy with
// 16-byte loads (1 KB contiguous per wave and copy), fold through LDS in wave order, one store.  Fixed order, coalesced.
__device__ __forceinline__ void tf_fold_wide(const TfFin& f, unsigned vb, float4 (*red)[64]) {
    const long long n = (long long)f.NT * f.CA * f.CB;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long i = ((long long)vb * 64 + lane) << 2;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n) {
#pragma unroll 4
        for (int c = w; c < f.ncopies; c += 4) {
            const float4 v = *reinterpret_cast<const float4*>(f.Rx + (long long)c * f.stride + i);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    red[w][lane] = s;
    __syncthreads();
    if (w == 0 && i < n) {
#pragma unroll
        for (int q = 1; q < 4; ++q) { const float4 v = red[q][lane]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
        const int b = (int)(i % f.CB); const long long r = i / f.CB; const int a = (int)(r % f.CA), t = (int)(r / f.CA);
        float* dst = f.R + (long long)t * f.RT + (long long)(a + f.a_off) * f.RSA + b + f.b_off;
        dst[0] += s.x; dst[1] += s.y; dst[2] += s.z; dst[3] += s.w;
    }
    // bias sums (block 0 of the fold): the copies are spread over the lane rows -- 256 / nbp rows stride the copies with four loads in
    // flight, then fold through LDS in row order.  (Round 6: one thread per bias element used to walk ALL copies, 512 dependent loads
    // of a single block -- the tail of a batched launch of thousands of blocks, 100-160 us.)
    if (vb == 0 && f.bias_serial) {                        // (M1_TF_BIAS_SERIAL=1: the round-5 form, for A/B)
        for (int j = threadIdx.x; j < f.nb; j += 256) {
            float sb = 0.f;
            for (int c = 0; c < f.ncopies; ++c) sb += f.Rx[(long long)c * f.stride + n + j];
            f.bsum[j + f.b_off] += sb;
        }
    } else if (vb == 0 && f.nb > 0) {
        __syncthreads();                                   // (red is reused)
        float* const rb = reinterpret_cast<float*>(red);   // 256 floats
        int nbp = 1; while (nbp < f.nb && nbp < 256) nbp <<= 1;
        const int rows = 256 / nbp, j = threadIdx.x % nbp, r0 = threadIdx.x / nbp;
        for (int j0 = 0; j0 < f.nb; j0 += nbp) {
            float sb = 0.f;
            if (j0 + j < f.nb) {
#pragma unroll 4
                for (int c = r0; c < f.ncopies; c += rows) sb += f.Rx[(long long)c * f.stride + n + j0 + j];
            }
            rb[threadIdx.x] = sb;
            __syncthreads();
            if (r0 == 0 && j0 + j < f.nb) {
                for (int q = 1; q < rows; ++q) sb += rb[q * nbp + j];
                f.bsum[j0 + j + f.b_off] += sb;
            }
            __syncthreads();
        }
    }
}
__global__ void __launch_bounds__(256) tf_finish_wide_kernel(TfFin f) {
    __shared__ float4 red[4][64];
    tf_fold_wide(f, blockIdx.x, red);
}

// Many folds in one launch (m1_wgrad_defer / m1_wgrad_fold_pending): a fold of one conv member is 5-10 us of a few dozen blocks;
// ~130 of them per C3 step, each alone on its stream, were 4 % of the step.  The jobs travel by value in the kernel arguments.
#define TF_FOLD_MAX 24
struct TfFoldBatch { int n; int pref[TF_FOLD_MAX + 1]; int mode[TF_FOLD_MAX]; TfFin f[TF_FOLD_MAX]; };
static_assert(sizeof(TfFoldBatch) <= 3800, "fold batch must fit the kernel argument segment");
__global__ void __launch_bounds__(256) tf_finish_batch_kernel(TfFoldBatch B) {
    __shared__ float4 red[4][64];
    int j = 0;
    while (j + 1 < B.n && (int)blockIdx.x >= B.pref[j + 1]) ++j;
    const unsigned vb = blockIdx.x - B.pref[j], nblk = B.pref[j + 1] - B.pref[j];
    const TfFin& f = B.f[j];
    if (B.mode[j] == 1) tf_fold_vec(f, vb, nblk);
    else if (B.mode[j] == 2) tf_fold_wide(f, vb, red);
    else tf_fold_generic(f, vb, reinterpret_cast<float*>(red));
}

static inline bool tf_chan_ok(int c) { return c == 8 || c == 16 || (c >= 32 && c % 32 == 0); }

// fills the launch geometry; false = shape outside this kernel (the per-tap kernel of wgrad_mfma.hip takes it)
static bool tf_plan(const WgradSpec& g, TfP& p) {
    if (g.dtype != M1_BF16) return false;
    const bool k133 = g.kd == 1 && g.kh == 3 && g.kw == 3, k333 = g.kd == 3 && g.kh == 3 && g.kw == 3;
    if (!k133 && !k333) return false;
    if (!tf_chan_ok(g.CA) || !tf_chan_ok(g.CB)) return false;
    if (g.BW % 8) return false;
    if ((long long)g.N * g.AD * g.AH * g.AW >= (1ll << 31) - (1 << 20) || (long long)g.N * g.BD * g.BH * g.BW >= (1ll << 31) - (1 << 20)) return false;
    p = TfP{};
    p.A = (const bf16_t*)g.A; p.B = (const bf16_t*)g.B; p.R = g.R; p.bsum = g.bsum;
    p.CA = g.CA; p.CB = g.CB; p.AD = g.AD; p.AH = g.AH; p.AW = g.AW; p.BD = g.BD; p.BH = g.BH; p.BW = g.BW; p.N = g.N;
    p.RT = g.RT; p.RSA = g.RSA; p.a_off = g.a_off; p.b_off = g.b_off;
    p.sd = g.sd; p.sh = g.sh; p.sw = g.sw; p.pd = g.pd; p.ph = g.ph; p.pw = g.pw;
    p.KWs = g.BW % 32 == 0 ? 32 : (g.BW % 16 == 0 ? 16 : 8);
    // K-tile: 128 voxels (4 k-steps per barrier and DMA round trip) for the (1,3,3) kernels and wherever both sides have <= 16
    // channels; 64 for the other (3,3,3) layers, whose 3-slice input tiles would leave one block per CU.  With the members of a
    // concat in one launch (enough blocks per CU) the larger tile pays: 64 -> 32 at (2,20,160,160) 146 -> 112 us, 320 -> 64 at
    // (2,20,80,80) 358 -> 294 us, -1.2 % per C3 step.  M1_TF_NKS4: 2 (default), 1 = only the <= 16-channel layers, 0 = always 64.
    int n4 = M1_CFG("M1_TF_NKS4", 2);
    const int nks = ((n4 && g.CA <= 16 && g.CB <= 16) || (n4 == 2 && g.kd == 1)) ? 4 : 2;
    p.nks = nks;
    p.TH = nks * (32 / p.KWs);
    p.AHt = (p.TH - 1) * g.sh + g.kh; p.AWt = (p.KWs - 1) * g.sw + g.kw;
    p.spra = (g.CA < 32 ? g.CA : 32) / 8; p.sprb = (g.CB < 32 ? g.CB : 32) / 8;
    const int a_rows = g.kd * p.AHt * p.AWt;
    p.a_slots = (a_rows * p.spra + 255) / 256 * 256; p.b_slots = nks * 32 * p.sprb;
    if (p.a_slots > TF_MAX_AIT * 256 || p.AHt > 255 || p.AWt > 255) return false;
    p.a_bytes = p.a_slots * 16; p.b_bytes = (p.b_slots + 255) / 256 * 256 * 16 + 64;      // (+64: slack for the 32-byte fragment reads of short rows)
    p.tiles_w = g.BW / p.KWs; p.tiles_h = (g.BH + p.TH - 1) / p.TH;
    p.ntiles = (long long)g.N * g.BD * p.tiles_h * p.tiles_w;
    if (p.ntiles >= (1ll << 30)) return false;
    // as many stages as fit ~76 KB (two blocks per CU), at least 3, bounded by the vmcnt range
    const int sb = p.a_bytes + p.b_bytes, npiece = p.a_slots / 256 + (p.b_slots + 255) / 256;
    int S = (76 * 1024) / sb;
    if (S < 3) S = 3;
    if (S > TF_MAX_STAGES) S = TF_MAX_STAGES;
    while (S > 3 && npiece * (S - 2) > 48) --S;
    { int fs = M1_CFG("M1_TF_STAGES", 0); if (fs >= 3) S = fs; }
    p.stages = S;
    return npiece * (S - 2) <= 48 && (size_t)S * sb <= 160 * 1024;
}
// 64x64 variant: both sides multiples of 64 channels
bool m1_tf_wgrad_supported(const WgradSpec& g) { TfP p; return tf_plan(g, p); }

// ---- deferred folds: queued (process-wide, the autograd engine calls the weight gradients from its own thread) and run
//      in a few batched launches by m1_wgrad_fold_pending ----
#include <mutex>
#include <vector>
struct TfPending { TfFin f; int mode; int blocks; };
static std::mutex g_fold_mu;
static std::vector<TfPending> g_fold_pending;
static int g_fold_defer = 0;
static int fold_launch_pending_locked(hipStream_t st) {
    size_t i = 0;
    while (i < g_fold_pending.size()) {
        TfFoldBatch B{}; B.pref[0] = 0;
        while (i < g_fold_pending.size() && B.n < TF_FOLD_MAX) {
            const TfPending& q = g_fold_pending[i++];
            B.f[B.n] = q.f; B.mode[B.n] = q.mode; B.pref[B.n + 1] = B.pref[B.n] + q.blocks; ++B.n;
        }
        hipLaunchKernelGGL(tf_finish_batch_kernel, dim3((unsigned)B.pref[B.n]), dim3(256), 0, st, B);
        int rc = m1_check_launch(); if (rc) { g_fold_pending.clear(); return rc; }
    }
    g_fold_pending.clear();
    return M1_OK;
}
int m1_fold_defer_set(int on) { std::lock_guard<std::mutex> lk(g_fold_mu); const int was = g_fold_defer; g_fold_defer = on ? 1 : 0; return was; }
extern "C" int m1_wgrad_defer(int on) { std::lock_guard<std::mutex> lk(g_fold_mu); g_fold_defer = on ? 1 : 0; return M1_OK; }
int m1_fold_defer_get() { std::lock_guard<std::mutex> lk(g_fold_mu); return g_fold_defer; }
extern "C" int m1_wgrad_fold_drop(void) { { std::lock_guard<std::mutex> lk(g_fold_mu); g_fold_pending.clear(); } m1_colsum_drop_pending(); return M1_OK; }
extern "C" int m1_wgrad_fold_pending(void* stream) {
    if (m1_debug_skip("fold")) return m1_wgrad_fold_drop();
    {
        std::lock_guard<std::mutex> lk(g_fold_mu);
        const int rc = fold_launch_pending_locked((hipStream_t)stream); if (rc) return rc;
    }
    return m1_colsum_launch_pending((hipStream_t)stream);        // (the transposed convs' bias gradients queued with the folds)
}

// shared with the per-tap kernel (wgrad_mfma.hip), which uses the same partial-copy scheme for small weight tensors
int m1_wg_rx_finish(float* rx, long long stride, int ncopies, const WgradSpec& g, long long nw, hipStream_t st) {
    TfFin f{rx, stride, ncopies, g.R, g.kd * g.kh * g.kw, g.CA, g.CB, g.RT, g.RSA, g.a_off, g.b_off, g.bsum, nw, g.bsum ? g.CB : 0, 32,
            M1_CFG("M1_TF_BIAS_SERIAL", 0)};
    const long long n = (long long)f.NT * f.CA * f.CB + f.nb;
    int mode = 0; long long blocks;
    if (ncopies <= 16 && f.CB % 4 == 0 && n >= (1 << 16)) {
        mode = 1; blocks = (n / 4 + 255) / 256; if (blocks > 4096) blocks = 4096;
    } else if (f.CB % 4 == 0 && n - f.nb >= (1 << 15)) {            // (n - nb) / 256 >= 128 blocks
        mode = 2; blocks = (n - f.nb + 255) / 256;
    } else {
        while (f.EL > 4 && n / f.EL < 128) f.EL >>= 1;           // small blocks of R: more lane rows per element, more blocks
        blocks = (n + f.EL - 1) / f.EL;
    }
    {
        std::lock_guard<std::mutex> lk(g_fold_mu);
        if (g_fold_defer) {
            // Two folds into the same block of R (a parameter used by two passes) must not share a launch (both read-modify-write
            // R).  The queue may hold folds whose partial copies were written on OTHER streams (SE shortcut / gate branches,
            // weight-gradient streams) that `st` is not ordered behind, so it is NOT flushed here: this fold runs now, on its own
            // stream, right behind the kernel that wrote its copies; the queued one follows at m1_wgrad_fold_pending, which the
            // caller issues on a stream that has joined every stream used since (ops.join_side_streams).
            bool dup = false;
            for (const TfPending& q : g_fold_pending)
                if (q.f.R == f.R && q.f.a_off == f.a_off && q.f.b_off == f.b_off) { dup = true; break; }
            if (!dup) {
                g_fold_pending.push_back(TfPending{f, mode, (int)blocks});
                return M1_OK;
            }
        }
    }
    if (mode == 1) hipLaunchKernelGGL(tf_finish_vec_kernel, dim3((unsigned)blocks), dim3(256), 0, st, f);
    else if (mode == 2) hipLaunchKernelGGL(tf_finish_wide_kernel, dim3((unsigned)blocks), dim3(256), 0, st, f);
    else hipLaunchKernelGGL(tf_finish_kernel, dim3((unsigned)blocks), dim3(256), 0, st, f);
    return m1_check_launch();
}

#define TF_MAX_COPY_BYTES (96ll << 20)
static int tf_wgrad_launch(const WgradSpec& g, long long nw, int nb, hipStream_t st, int nmem, const void* const* Am, const int* a_offs, long long rx_mem);
// `nmem` members of one Conv3D concat at once: member m = (Am[m], a_off a_offs[m]), all with g.CA channels; copies of member m
// live at g.rx + m * rx_mem (rx_mem >= g.rx_floats of one member)
int m1_tf_wgrad_multi(const WgradSpec& g, long long nw, int nb, hipStream_t st, int nmem, const void* const* Am, const int* a_offs, long long rx_mem) {
    return tf_wgrad_launch(g, nw, nb, st, nmem, Am, a_offs, rx_mem);
}
// nw / nb: floats of the whole weight / bias gradient that g.R / g.bsum point into
int m1_tf_wgrad(const WgradSpec& g, long long nw, int nb, hipStream_t st) { return tf_wgrad_launch(g, nw, nb, st, 1, nullptr, nullptr, 0); }
static int tf_wgrad_launch(const WgradSpec& g, long long nw, int nb, hipStream_t st, int nmem, const void* const* Am, const int* a_offs, long long rx_mem) {
    TfP p;
    if (!tf_plan(g, p)) return M1_ERR_UNSUPPORTED;
    const int TS = 32;
    const int aTiles = (g.CA + TS - 1) / TS; p.bTiles = (g.CB + TS - 1) / TS;
    const int ctiles = aTiles * p.bTiles;
    // blocks per launch (M1_TF_SPLIT, 0 = by size): 256 (one per CU) up to ~24k K-tiles, 512 beyond -- isolated, the 64 -> 32 layer
    // at (20,160,160) runs 102 -> 67 us with 512, but every block adds a partial copy to fold and in the captured step other
    // kernels fill the idle SIMDs: C3 (4 volumes per launch) -2 % with 512, C2 (2 volumes) +1 %
    int tgt_env = M1_CFG("M1_TF_SPLIT", 0);
    const int tgt = tgt_env > 0 ? tgt_env : (p.ntiles >= 24576 ? 512 : 256);
    long long nsplit = (tgt + ctiles - 1) / ctiles;
    const long long nloc = (long long)g.kd * g.kh * g.kw * g.CA * g.CB;
    const long long stride = nloc + g.CB;                      // compact copy of this member's block (+ bias sums)
    if (nsplit * stride * 4 > TF_MAX_COPY_BYTES) nsplit = TF_MAX_COPY_BYTES / (stride * 4);
    if (!g.rx || g.rx_floats < stride) return M1_ERR_WORKSPACE;
    if (nsplit * stride > g.rx_floats) nsplit = g.rx_floats / stride;      // as many copies as the caller's scratch holds
    if (nsplit > p.ntiles) nsplit = p.ntiles;
    if (nsplit < 1) return M1_ERR_UNSUPPORTED;
    p.nsplit = (int)nsplit;
    p.Rx = g.rx; p.rx_stride = stride; p.rx_bias = nloc;
    const size_t smem = (size_t)p.stages * (p.a_bytes + p.b_bytes);
    dim3 grid(ctiles, (unsigned)nsplit, 1);
    p.nmem = 1; p.rx_mem = 0;
    if (nmem > 1) {
        if (nmem > M1_MAX_SRC) return M1_ERR_UNSUPPORTED;
        p.nmem = nmem; p.rx_mem = rx_mem; grid.z = (unsigned)nmem;
        for (int m = 0; m < nmem; ++m) p.Am[m] = (const bf16_t*)Am[m];
    }
    const int kparts = 4 / ((g.CA > 16 ? 2 : 1) * (g.CB > 16 ? 2 : 1));
    void (*kern)(TfP) = nullptr;
#define TF_PICK(KD_, KP_) if ((g.kd == KD_) && kparts == KP_ && p.nks == 2) kern = wgrad_tf_kernel<KD_, 3, 3, KP_, 2>;
    TF_PICK(1, 1) TF_PICK(1, 2) TF_PICK(1, 4) TF_PICK(3, 1) TF_PICK(3, 2) TF_PICK(3, 4)
#undef TF_PICK
    if (p.nks == 4 && kparts == 4) kern = g.kd == 1 ? wgrad_tf_kernel<1, 3, 3, 4, 4> : (g.kd == 3 ? wgrad_tf_kernel<3, 3, 3, 4, 4> : nullptr);
    if (p.nks == 4 && g.kd == 1 && kparts == 1) kern = wgrad_tf_kernel<1, 3, 3, 1, 4>;
    if (p.nks == 4 && g.kd == 1 && kparts == 2) kern = wgrad_tf_kernel<1, 3, 3, 2, 4>;
    if (!kern) return M1_ERR_UNSUPPORTED;
    {   // raise the dynamic-LDS limit once per instantiation
        static const void* done[8]; static int ndone = 0;
        bool seen = false;
        for (int q = 0; q < ndone; ++q) seen |= done[q] == (const void*)kern;
        if (!seen) {
            if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return M1_ERR_LAUNCH;
            if (ndone < 8) done[ndone++] = (const void*)kern;
        }
    }
    m1_note_kernel("wgrad_tf");
    hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, p);
    int rc = m1_check_launch(); if (rc) return rc;
    if (nmem <= 1) return m1_wg_rx_finish(p.Rx, stride, (int)nsplit, g, nloc, st);
    for (int m = 0; m < nmem; ++m) {                      // one fold per member (its own block of R; the bias sums ride on member 0)
        WgradSpec gm = g; gm.a_off = a_offs[m]; if (m) gm.bsum = nullptr;
        rc = m1_wg_rx_finish(p.Rx + (long long)m * rx_mem, stride, (int)nsplit, gm, nloc, st); if (rc) return rc;
    }
    return M1_OK;
}
