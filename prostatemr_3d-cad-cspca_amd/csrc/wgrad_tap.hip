// wgrad_tap.hip -- per-tap bf16 weight gradient for the deep layers (>= 64 channels on both sides) built on the
// gfx950 transpose read instead of register transposes:
//
//   R[tap][a][b] += sum_{n,v} A[n, v*s + tap - p][a] * B[n, v][b]            one tap per blockIdx.z
//
// * wgrad_mfma.hip stages both operands through VGPRs, transposes 8x8 blocks in registers and writes them with
//   ds_write_b128 (<= 79 B/clk/CU): for a 128x128 tile that store path alone costs as many cycles as the MFMAs.
//   Here the 64-voxel operand tiles go global -> LDS by LDS-DMA in their natural voxel-major layout and are read as
//   K-contiguous fragments with ds_read_b64_tr_b16 (2 LDS cycles per wave-instruction): no staging registers, no
//   transposes, no ds_write, 16 fragment reads per 16 MFMAs for the 64x64 wave tile.
// * Voxel tiles are 2-D (rows x columns of one (n, d) slice, as wgrad_tf.hip) so that the per-lane DMA source
//   offsets are tile invariant; the tap only moves the tile origin.  Out-of-volume voxels fetch a zero page.
// * 32-byte pieces of a row are XOR-swizzled with row bits {0,1,3} (8 pieces) / {1,3} (4) / {3} (2) so that the two
//   16-lane groups of a transpose read (rows r..r+3 and r+8..r+11) hit disjoint banks.
// * Few voxel splits per (tap, tile) (~2 blocks per CU in total): float atomics into R.
#include "common.h"
#include "gather.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __attribute__((aligned(64))) unsigned int m1_zero_page_t[16];

__device__ __forceinline__ void glds16t(const void* g, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ u32x2_t tr_read_t(unsigned lds_addr) {
    u32x2_t v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(lds_addr) : "memory");
    return v;
}
__device__ __forceinline__ void lds_wait_t(u32x2_t& a, u32x2_t& b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ bf16x8_t frag8t(u32x2_t lo, u32x2_t hi) {
    return __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(lo, hi, 0, 1, 2, 3));
}
__device__ __forceinline__ void wait_vmt(int n) {
    switch (n) {
#define TW_(N_) case N_: asm volatile("s_waitcnt vmcnt(" #N_ ")" ::: "memory"); break;
        TW_(0) TW_(1) TW_(2) TW_(3) TW_(4) TW_(5) TW_(6) TW_(7) TW_(8) TW_(9) TW_(10) TW_(11) TW_(12) TW_(13) TW_(14) TW_(15) TW_(16)
        TW_(17) TW_(18) TW_(19) TW_(20) TW_(21) TW_(22) TW_(23) TW_(24)
#undef TW_
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}
// XOR applied to the 32-byte piece index of row kk (np = pieces per row, a power of two <= 8)
__device__ __forceinline__ int piece_swz(int kk, int np) {
    return np == 8 ? ((kk & 3) | (((kk >> 3) & 1) << 2)) : (np == 4 ? (((kk >> 1) & 1) | (((kk >> 3) & 1) << 1)) : (np == 2 ? ((kk >> 3) & 1) : 0));
}

struct TapP {
    const bf16_t* A; const bf16_t* B; float* R; float* bsum; int bsum_tap;
    int CA, CB, AD, AH, AW, BD, BH, BW, N;
    long long RT, RSA; int a_off, b_off;
    int kd, kh, kw, sd, sh, sw, pd, ph, pw;
    int KWs, TH;                 // K-tile = TH rows x KWs columns = 64 voxels (2 k-steps of 32)
    int tiles_w, tiles_h, ntiles, nsplit, bTiles, stages;
    float* Rx; long long rx_stride, rx_bias;     // per-split partial copies of R (+ bias sums), folded by m1_wg_rx_finish
    int ctiles, taps, xcd_total;                 // xcd_total != 0: 1-D grid of that many blocks, XCD-aware order (see the kernel)
};

template <int SUBA, int SUBB>     // wave tile = SUBA*16 x SUBB*16 channels, block = 2x2 waves
__global__ void __launch_bounds__(256) wgrad_tap_kernel(TapP p) {
#if defined(__HIP_DEVICE_COMPILE__)     // (the buffer-resource builtins do not exist in the host pass)
    constexpr int TA = SUBA * 32, TB = SUBB * 32, PA = TA * 2, PB = TB * 2, NPA = TA / 16, NPB = TB / 16;
    constexpr int SPA = TA / 8, SPB = TB / 8;                   // 16-byte slots per row
    constexpr int AIT = 64 * SPA / 256, BIT = 64 * SPB / 256;   // DMA pieces per thread per stage (1, 2 or 4)
    constexpr int A_BYTES = 64 * PA, B_BYTES = 64 * PB, STAGE = A_BYTES + B_BYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wa = wave >> 1, wb = wave & 1;
    // XCD-aware block order (p.xcd_total != 0: 1-D grid padded to a multiple of 8): consecutive block ids go round-robin
    // over the 8 XCDs, each with its own L2.  All taps of one voxel split read the same dY tiles and X tiles one row / plane
    // apart; numbered tap-fastest inside a contiguous per-XCD range they run side by side on ONE XCD and share its L2 instead
    // of every XCD fetching every tile (measured HBM fetch of this kernel: 6.7x its operands without the remap).
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (p.xcd_total) {
        const unsigned L = (blockIdx.x & 7u) * (p.xcd_total >> 3) + (blockIdx.x >> 3);
        if (L >= (unsigned)(p.ctiles * p.nsplit * p.taps)) return;
        bz = L % p.taps; by = (L / p.taps) % p.nsplit; bx = L / (p.taps * p.nsplit);
    }
    const int a0 = (bx / p.bTiles) * TA, b0 = (bx % p.bTiles) * TB;
    const int tap = bz;
    const int tkw = tap % p.kw, tkh = (tap / p.kw) % p.kh, tkd = tap / (p.kw * p.kh);
    const unsigned char* zero_pg = reinterpret_cast<const unsigned char*>(m1_zero_page_t);

    // ---- per-lane DMA pieces: LDS slot q -> (row kk = voxel of the tile, physical slot); tile invariant ----
    // (buffer loads: the tile origin goes into the resource base, the lane keeps a constant 32-bit byte offset; an offset
    //  of 2^31 is past num_records = 2^31 - 1, so the hardware range check returns the zeros of the padding)
    constexpr unsigned OOB = 0x80000000u;
    unsigned a_vo[AIT], b_vo[BIT]; int a_pk[AIT], b_th[BIT];
#pragma unroll
    for (int it = 0; it < AIT; ++it) {
        const int q = it * 256 + tid, kk = q / SPA, slp = q % SPA;
        const int th = kk / p.KWs, tw = kk - th * p.KWs;
        const int sl = (((slp >> 1) ^ piece_swz(kk, NPA)) << 1) | (slp & 1);
        a_pk[it] = (th * p.sh) | ((tw * p.sw) << 8);
        // (rows beyond TH: a row length that does not divide 64 leaves the tail of the 64-voxel tile empty)
        a_vo[it] = (a0 + sl * 8 < p.CA && th < p.TH) ? (unsigned)((((th * p.sh) * p.AW + tw * p.sw) * p.CA + a0 + sl * 8) * 2) : OOB;
    }
#pragma unroll
    for (int it = 0; it < BIT; ++it) {
        const int q = it * 256 + tid, kk = q / SPB, slp = q % SPB;
        const int th = kk / p.KWs, tw = kk - th * p.KWs;
        const int sl = (((slp >> 1) ^ piece_swz(kk, NPB)) << 1) | (slp & 1);
        b_th[it] = th;
        b_vo[it] = (b0 + sl * 8 < p.CB && th < p.TH) ? (unsigned)(((th * p.BW + tw) * p.CB + b0 + sl * 8) * 2) : OOB;
    }

    int q_kt = by, q_tw, q_th, q_bd, q_n;
    { int r = q_kt; q_tw = r % p.tiles_w; r /= p.tiles_w; q_th = r % p.tiles_h; r /= p.tiles_h; q_bd = r % p.BD; q_n = r / p.BD; }
    int s_tw, s_th, s_bd, s_n;
    { int r = p.nsplit; s_tw = r % p.tiles_w; r /= p.tiles_w; s_th = r % p.tiles_h; r /= p.tiles_h; s_bd = r % p.BD; s_n = r / p.BD; }
    auto issue = [&](int st) {
        const bool live = q_kt < p.ntiles;
        const int twi = q_tw, thi = q_th, bd = q_bd, n = q_n;
        q_kt += p.nsplit;
        q_tw += s_tw; int c = q_tw >= p.tiles_w; q_tw -= c ? p.tiles_w : 0;
        q_th += s_th + c; c = q_th >= p.tiles_h; q_th -= c ? p.tiles_h : 0;
        q_bd += s_bd + c; c = q_bd >= p.BD; q_bd -= c ? p.BD : 0;
        q_n += s_n + c;
        const int ad = bd * p.sd + tkd - p.pd, ah0 = thi * p.TH * p.sh + tkh - p.ph, aw0 = twi * p.KWs * p.sw + tkw - p.pw;
        const bool dok = live && (unsigned)ad < (unsigned)p.AD;
        const long long alin0 = (((long long)n * p.AD + ad) * p.AH + ah0) * p.AW + aw0;
        const int bh0 = thi * p.TH;
        const long long blin0 = (((long long)n * p.BD + bd) * p.BH + bh0) * p.BW + twi * p.KWs;
        unsigned char* As = smem + st * STAGE;
        unsigned char* Bs = As + A_BYTES;
        // num_records 0 = every lane out of range (tile past the end / d slice outside the volume): all zeros
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + alin0 * p.CA), 0, dok ? 0x7fffffff : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(p.B + blin0 * p.CB), 0, live ? 0x7fffffff : 0, 0x00020000);
        const bool a_inner = ah0 >= 0 && ah0 + (p.TH - 1) * p.sh < p.AH && aw0 >= 0 && aw0 + (p.KWs - 1) * p.sw < p.AW;
        const bool b_inner = bh0 + p.TH <= p.BH;
        if (a_inner) {                                     // uniform: no per-lane work at all
#pragma unroll
            for (int it = 0; it < AIT; ++it)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lptr_t)(As + (it * 256 + wave * 64) * 16), 16, a_vo[it], 0, 0, 0);
        } else {
#pragma unroll
            for (int it = 0; it < AIT; ++it) {
                const int hh = a_pk[it] & 0xff, ww = a_pk[it] >> 8;
                const bool ok = (unsigned)(ah0 + hh) < (unsigned)p.AH && (unsigned)(aw0 + ww) < (unsigned)p.AW;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lptr_t)(As + (it * 256 + wave * 64) * 16), 16, ok ? a_vo[it] : OOB, 0, 0, 0);
            }
        }
        if (b_inner) {
#pragma unroll
            for (int it = 0; it < BIT; ++it)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lptr_t)(Bs + (it * 256 + wave * 64) * 16), 16, b_vo[it], 0, 0, 0);
        } else {
#pragma unroll
            for (int it = 0; it < BIT; ++it)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lptr_t)(Bs + (it * 256 + wave * 64) * 16), 16, bh0 + b_th[it] < p.BH ? b_vo[it] : OOB, 0, 0, 0);
        }
    };

    // ---- fragment addresses of sub-tile 0 (sub-tile s: address ^ (s*32)) ----
    const int g = lane >> 4, i = lane & 15;
    const unsigned lds0 = (unsigned)(unsigned long long)(lptr_t)smem;
    unsigned a_ad[2][2], b_ad[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int kk = ks * 32 + 8 * g + 4 * h + (i >> 2);
            a_ad[ks][h] = lds0 + kk * PA + (((wa * SUBA) ^ piece_swz(kk, NPA)) * 32) + (i & 3) * 8;
            b_ad[ks][h] = lds0 + A_BYTES + kk * PB + (((wb * SUBB) ^ piece_swz(kk, NPB)) * 32) + (i & 3) * 8;
        }

    f32x4_t acc[SUBA][SUBB];
#pragma unroll
    for (int x = 0; x < SUBA; ++x)
#pragma unroll
        for (int y = 0; y < SUBB; ++y) acc[x][y] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const bool do_bsum = p.bsum != nullptr && tap == p.bsum_tap && a0 == 0 && wa == 0;
    f32x4_t accb[SUBB];
#pragma unroll
    for (int y = 0; y < SUBB; ++y) accb[y] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, make_uint4(0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u));

    const int S = p.stages;
    constexpr int npiece = AIT + BIT;
    for (int s = 0; s < S - 1; ++s) issue(s);
    int st = 0;
    for (int kt = by; kt < p.ntiles; kt += p.nsplit) {
        wait_vmt(npiece * (S - 2));
        __builtin_amdgcn_s_barrier();
        int stn = st + S - 1; if (stn >= S) stn -= S;
        issue(stn);
        const unsigned sb = (unsigned)(st * STAGE);
        u32x2_t al[2][SUBA], ah[2][SUBA], bl[2][SUBB], bh[2][SUBB];
        auto rd = [&](int ks) {
#pragma unroll
            for (int x = 0; x < SUBA; ++x) { al[ks][x] = tr_read_t((a_ad[ks][0] + sb) ^ (x * 32)); ah[ks][x] = tr_read_t((a_ad[ks][1] + sb) ^ (x * 32)); }
#pragma unroll
            for (int y = 0; y < SUBB; ++y) { bl[ks][y] = tr_read_t((b_ad[ks][0] + sb) ^ (y * 32)); bh[ks][y] = tr_read_t((b_ad[ks][1] + sb) ^ (y * 32)); }
        };
        rd(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int x = 0; x < SUBA; ++x) lds_wait_t(al[ks][x], ah[ks][x]);
#pragma unroll
            for (int y = 0; y < SUBB; ++y) lds_wait_t(bl[ks][y], bh[ks][y]);
            if (ks == 0) rd(1);
            __builtin_amdgcn_sched_barrier(0);
            bf16x8_t bf[SUBB];
#pragma unroll
            for (int y = 0; y < SUBB; ++y) bf[y] = frag8t(bl[ks][y], bh[ks][y]);
            if (do_bsum) {
#pragma unroll
                for (int y = 0; y < SUBB; ++y) accb[y] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, bf[y], accb[y], 0, 0, 0);
            }
#pragma unroll
            for (int x = 0; x < SUBA; ++x) {
                const bf16x8_t af = frag8t(al[ks][x], ah[ks][x]);
#pragma unroll
                for (int y = 0; y < SUBB; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf[y], acc[x][y], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (++st == S) st = 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- D[a][b]: lane holds a = 4*(lane>>4) + r, b = lane&15.  Plain stores into this split's copy when there is one
    //      (the memory-side float atomics of ~500 blocks drain for tens of microseconds after the last wave has finished) ----
    float* const Rx = p.Rx ? p.Rx + (long long)by * p.rx_stride : nullptr;
#pragma unroll
    for (int y = 0; y < SUBB; ++y) {
        const int b = b0 + (wb * SUBB + y) * 16 + i;
        if (do_bsum && g == 0 && b < p.CB) { if (Rx) Rx[p.rx_bias + b] = accb[y][0]; else atomicAdd(p.bsum + b + p.b_off, accb[y][0]); }
#pragma unroll
        for (int x = 0; x < SUBA; ++x)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int a = a0 + (wa * SUBA + x) * 16 + g * 4 + r;
                if (a < p.CA && b < p.CB) {
                    const long long idx = (long long)tap * p.RT + (long long)(a + p.a_off) * p.RSA + (b + p.b_off);
                    if (Rx) Rx[((long long)tap * p.CA + a) * p.CB + b] = acc[x][y][r]; else atomicAdd(p.R + idx, acc[x][y][r]);
                }
            }
    }
#endif
}

static inline int tap_side(int c) { return c > 64 ? 128 : 64; }

static bool tap_plan(const WgradSpec& g, TapP& p) {
    int en = M1_CFG("M1_WG_TAP", 1);
    if (!en || g.dtype != M1_BF16) return false;
    if (g.CA < 64 || g.CB < 64 || g.CA % 8 || g.CB % 8) return false;
    if (g.BW % 8 && g.BW > 32) return false;
    if ((long long)g.N * g.AD * g.AH * g.AW >= (1ll << 31) - (1 << 20) || (long long)g.N * g.BD * g.BH * g.BW >= (1ll << 31) - (1 << 20)) return false;
    p = TapP{};
    p.A = (const bf16_t*)g.A; p.B = (const bf16_t*)g.B; p.R = g.R; p.bsum = g.bsum; p.bsum_tap = g.bsum_tap;
    p.CA = g.CA; p.CB = g.CB; p.AD = g.AD; p.AH = g.AH; p.AW = g.AW; p.BD = g.BD; p.BH = g.BH; p.BW = g.BW; p.N = g.N;
    p.RT = g.RT; p.RSA = g.RSA; p.a_off = g.a_off; p.b_off = g.b_off;
    p.kd = g.kd; p.kh = g.kh; p.kw = g.kw; p.sd = g.sd; p.sh = g.sh; p.sw = g.sw; p.pd = g.pd; p.ph = g.ph; p.pw = g.pw;
    // tile = TH rows x KWs columns <= 64 voxels; rows of up to 32 voxels that are no multiple of 8 (W = 20 at the (10,20,20) level)
    // take the whole row: 3 x 20 = 60 voxels + 4 empty slots (89 % of the MFMA work is real, against 62 % for 16 x 4 tiles)
    p.KWs = g.BW % 32 == 0 ? 32 : (g.BW % 16 == 0 ? 16 : (g.BW % 8 == 0 ? 8 : g.BW));
    p.TH = 64 / p.KWs;
    if ((p.TH - 1) * g.sh > 255 || (p.KWs - 1) * g.sw > 255) return false;
    p.tiles_w = g.BW / p.KWs; p.tiles_h = (g.BH + p.TH - 1) / p.TH;
    const long long nt = (long long)g.N * g.BD * p.tiles_h * p.tiles_w;
    if (nt >= (1ll << 30) || nt < 8) return false;
    p.ntiles = (int)nt;
    return true;
}
bool m1_tap_wgrad_supported(const WgradSpec& g) { TapP p; return tap_plan(g, p); }

int m1_tap_wgrad(const WgradSpec& g, long long nw, int nb, hipStream_t st) {
    TapP p;
    if (!tap_plan(g, p)) return M1_ERR_UNSUPPORTED;
    const int TA = tap_side(g.CA), TB = tap_side(g.CB);
    const int taps = g.kd * g.kh * g.kw;
    int tgt = M1_CFG("M1_WG_TAP_BLOCKS", 512);
    const int aTiles = (g.CA + TA - 1) / TA; p.bTiles = (g.CB + TB - 1) / TB;
    const int ctiles = aTiles * p.bTiles;
    // round DOWN: 2 blocks per CU x 256 CUs = 512 slots; one block more than that is a second round for its whole XCD
    int rdn = M1_CFG("M1_WG_FLOOR", 1);
    long long nsplit = rdn ? tgt / ((long long)ctiles * taps) : (tgt + (long long)ctiles * taps - 1) / ((long long)ctiles * taps);
    if (nsplit > p.ntiles / 4) nsplit = p.ntiles / 4;
    if (nsplit < 1) nsplit = 1;
    p.nsplit = (int)nsplit;
    const int stage = 64 * (TA + TB) * 2;
    int S = (76 * 1024) / stage; if (S > 4) S = 4; if (S < 2) S = 2;
    { int fs = M1_CFG("M1_WG_TAP_STAGES", 0); if (fs >= 2) S = fs; }
    p.stages = S;
    const size_t smem = (size_t)S * stage;
    void (*kern)(TapP) = nullptr;
    if (TA == 128 && TB == 128) kern = wgrad_tap_kernel<4, 4>;
    else if (TA == 128 && TB == 64) kern = wgrad_tap_kernel<4, 2>;
    else if (TA == 64 && TB == 128) kern = wgrad_tap_kernel<2, 4>;
    else kern = wgrad_tap_kernel<2, 2>;
    {
        static const void* done[4]; static int ndone = 0;
        bool seen = false;
        for (int q = 0; q < ndone; ++q) seen |= done[q] == (const void*)kern;
        if (!seen) {
            if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return M1_ERR_LAUNCH;
            if (ndone < 4) done[ndone++] = (const void*)kern;
        }
    }
    // partial copies + fixed-order fold instead of atomics (compact per-member copies, see wgrad_mfma.hip): bit-reproducible
    // weight gradients; M1_WG_DET=0 restores the atomic path
    int det = M1_CFG("M1_WG_DET", 1);
    const long long stride = (long long)taps * g.CA * g.CB + g.CB;
    p.Rx = nullptr; p.rx_stride = stride; p.rx_bias = (long long)taps * g.CA * g.CB;
    if (det && nsplit >= 2) {
        long long fit = g.rx ? g.rx_floats / stride : 0;
        if (fit >= 2) { if (nsplit > fit) nsplit = fit; p.Rx = g.rx; }
        else nsplit = 1;
        p.nsplit = (int)nsplit;
    }
    int xr = M1_CFG("M1_WG_XCD", 1);
    p.ctiles = ctiles; p.taps = taps; p.xcd_total = 0;
    if (xr && taps > 1) {
        p.xcd_total = (int)(((long long)ctiles * nsplit * taps + 7) / 8 * 8);
        m1_note_kernel("wgrad_tap");
        hipLaunchKernelGGL(kern, dim3((unsigned)p.xcd_total), dim3(256), smem, st, p);
    } else {
        m1_note_kernel("wgrad_tap");
        hipLaunchKernelGGL(kern, dim3(ctiles, (unsigned)nsplit, taps), dim3(256), smem, st, p);
    }
    int rc = m1_check_launch(); if (rc) return rc;
    if (p.Rx) return m1_wg_rx_finish(p.Rx, stride, (int)nsplit, g, p.rx_bias, st);
    return M1_OK;
}
