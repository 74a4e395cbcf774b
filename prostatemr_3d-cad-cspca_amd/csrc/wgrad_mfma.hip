// wgrad_mfma.hip -- weight gradients of Conv3D / Conv3DTranspose on the matrix cores.
//
//   R[tap][a][b] += sum_{n,v} A[n, v*s + tap - p][a] * B[n, v][b]        M = a, N = b, K = voxels (split-K)
//
// Both operands are voxel-major in memory (NDHWC) while the MFMA wants K(=voxel)-contiguous fragments, so the
// staging path transposes SEG x SEG blocks IN REGISTERS (8x8 bf16 via 16-bit interleaves, 4x4 fp32 for free by
// register renaming) and writes K-contiguous 16-byte segments into the same swizzled 64-byte-row LDS image the
// forward kernel uses; the fragment reads / MFMA issue are then identical to conv_mfma.hip.
// One block = one (tap, a-tile, b-tile, voxel-split); partial results are combined with fp32 atomics in L2.
#include "common.h"
#include "gather.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

template <typename T> struct WT;
template <> struct WT<bf16_t> { static constexpr int SEG = 8; };
template <> struct WT<float> { static constexpr int SEG = 4; };

// LDS image [row][64 B]: byte offset of 16-byte segment `seg` of logical row `row`.  Rows 8..15 of every 16 are
// pair-swapped and the slot is XOR-ed with row bits 2..5 so that BOTH the ds_read_b128 fragment reads (16 rows x
// one slot per lane group) and the staging ds_write_b128 (8 lanes = 8 channel groups 8 rows apart, same slot) are
// bank-conflict free (checked exhaustively against the gfx950 lane groups, tools/lds_swizzle_check.py).
__device__ __forceinline__ int woff(int row, int seg) {
    return ((row ^ ((row >> 3) & 1)) << 6) + (((seg ^ (-(row >> 2)) ^ (row >> 4)) & 3) << 4);
}

struct WgP : WgradSpec { long long vox_per_split; float* Rx; long long rx_stride, rx_bias;       // Rx: per-split partial copies
             int xcd_total, xcd_gx, xcd_gy, xcd_gz; };                                             // XCD-aware 1-D launch (see the kernel)

// r[j] = 16 bytes of voxel j (SEG channels); returns o[c] = 16 bytes of channel c (SEG voxels)
__device__ __forceinline__ void transpose_unit(const uint4 (&r)[8], uint4 (&o)[8], bf16_t) {
    // 8x8 of 16-bit: pairwise interleave rows (2i,2i+1) per dword, then a register-only 4x4 dword transpose
    unsigned lo[4][4], hi[4][4];     // [pair i][dword j]: lo = channel 2j of rows (2i,2i+1), hi = channel 2j+1
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned x[4] = {r[2 * i].x, r[2 * i].y, r[2 * i].z, r[2 * i].w};
        const unsigned y[4] = {r[2 * i + 1].x, r[2 * i + 1].y, r[2 * i + 1].z, r[2 * i + 1].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            lo[i][j] = (x[j] & 0xffffu) | (y[j] << 16);
            hi[i][j] = (x[j] >> 16) | (y[j] & 0xffff0000u);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        o[2 * j] = make_uint4(lo[0][j], lo[1][j], lo[2][j], lo[3][j]);
        o[2 * j + 1] = make_uint4(hi[0][j], hi[1][j], hi[2][j], hi[3][j]);
    }
}
__device__ __forceinline__ void transpose_unit(const uint4 (&r)[8], uint4 (&o)[8], float) {
    o[0] = make_uint4(r[0].x, r[1].x, r[2].x, r[3].x);
    o[1] = make_uint4(r[0].y, r[1].y, r[2].y, r[3].y);
    o[2] = make_uint4(r[0].z, r[1].z, r[2].z, r[3].z);
    o[3] = make_uint4(r[0].w, r[1].w, r[2].w, r[3].w);
}

template <typename T, int TA, int TB, int KB>
__global__ void __launch_bounds__(256) wgrad_mfma_kernel(WgP p) {
    constexpr int SEG = WT<T>::SEG;
    constexpr int KS1 = 4 * SEG;                           // voxels per 64-byte LDS row
    constexpr int KS = KS1 * KB;                           // voxels per K-step (KB row-blocks: keeps all 256 threads loading)
    constexpr int TM = TA / 2 / 16, TN = TB / 2 / 16;      // 2x2 waves
    constexpr int UA = (TA / SEG) * 4 * KB, UB = (TB / SEG) * 4 * KB;   // SEGxSEG units per K-step
    constexpr int NU = (UA + UB + 255) / 256;
    __shared__ __attribute__((aligned(16))) unsigned char A_s[KB * TA * 64];
    __shared__ __attribute__((aligned(16))) unsigned char B_s[KB * TB * 64];
    __shared__ long long a_off_s[KS], b_off_s[KS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int bTiles = (p.CB + TB - 1) / TB;
    // XCD-aware block order (p.xcd_total != 0, see wgrad_tap.hip): the ab-tiles and taps of one voxel split share one XCD's L2
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (p.xcd_total) {
        const unsigned L = (blockIdx.x & 7u) * (p.xcd_total >> 3) + (blockIdx.x >> 3);
        if (L >= (unsigned)(p.xcd_gx * p.xcd_gy * p.xcd_gz)) return;
        bx = L % p.xcd_gx; by = (L / p.xcd_gx) % p.xcd_gy; bz = L / (p.xcd_gx * p.xcd_gy);
    }
    const int a0 = (bx / bTiles) * TA, b0 = (bx % bTiles) * TB;
    const int tap = by;
    const int kw = tap % p.kw, kh = (tap / p.kw) % p.kh, kd = tap / (p.kw * p.kh);
    // voxel indices fit 32 bits (checked on the host): 32-bit divisions only
    const int BV = p.BD * p.BH * p.BW, TV = BV * p.N;
    const int vbeg = (int)(bz * p.vox_per_split);
    int vend = vbeg + (int)p.vox_per_split; if (vend > TV) vend = TV;
    const T* A = (const T*)p.A; const T* B = (const T*)p.B;

    f32x4_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fs = lane >> 4;
    // fused bias gradient: column sums of the B tile by one extra MFMA with an all-ones A fragment (only the
    // blocks of one always-in-bounds tap and the first a-tile do it, only the wm == 0 waves)
    const bool do_bsum = p.bsum != nullptr && tap == p.bsum_tap && a0 == 0 && wm == 0;
    f32x4_t accb[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) accb[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    uint4 ones;
    if constexpr (sizeof(T) == 2) ones = make_uint4(0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u);
    else ones = make_uint4(0x3F800000u, 0x3F800000u, 0x3F800000u, 0x3F800000u);

    for (int vs = vbeg; vs < vend; vs += KS) {
        for (int t = tid; t < KS; t += 256) {
            const int v = vs + t;
            long long ao = -1, bo = -1;
            if (v < vend) {
                const int n = v / BV; int r = v - n * BV;
                const int q1 = r / p.BW; const int bw = r - q1 * p.BW;
                const int bd = q1 / p.BH; const int bh = q1 - bd * p.BH;
                const int ad = bd * p.sd + kd - p.pd, ah = bh * p.sh + kh - p.ph, aw = bw * p.sw + kw - p.pw;
                if (ad >= 0 && ad < p.AD && ah >= 0 && ah < p.AH && aw >= 0 && aw < p.AW) {
                    ao = (((long long)n * p.AD + ad) * p.AH + ah) * p.AW + aw;
                    bo = v;        // a voxel whose shifted partner is outside the volume contributes nothing
                }
            }
            a_off_s[t] = ao; b_off_s[t] = bo;
        }
        __syncthreads();           // also fences the previous step's fragment reads before A_s/B_s are rewritten
#pragma unroll
        for (int q = 0; q < NU; ++q) {
            const int u = tid + 256 * q;
            if (u < UA + UB) {
                const bool isA = u < UA;
                const int uu = isA ? u : u - UA;
                const int ncg = (isA ? TA : TB) / SEG;
                const int cg = uu % ncg, ks = uu / ncg;                 // channel group, k segment (0 .. 4*KB-1)
                const int C = isA ? p.CA : p.CB, c0 = (isA ? a0 : b0) + cg * SEG;
                const T* base = isA ? A : B;
                uint4 r[8], o[8];
#pragma unroll
                for (int j = 0; j < SEG; ++j) {
                    const long long off = isA ? a_off_s[ks * SEG + j] : b_off_s[ks * SEG + j];
                    uint4 v = make_uint4(0, 0, 0, 0);
                    if (off >= 0 && c0 < C) {
                        if (C % SEG == 0) v = *reinterpret_cast<const uint4*>(base + off * C + c0);
                        else {          // member not 16-byte tiled (stem input, latent z, class logits): element loads, zero pad
                            union { uint4 q; T e[SEG]; } u; u.q = make_uint4(0, 0, 0, 0);
#pragma unroll
                            for (int k = 0; k < SEG; ++k) if (c0 + k < C) u.e[k] = base[off * C + c0 + k];
                            v = u.q;
                        }
                    }
                    r[j] = v;
                }
                transpose_unit(r, o, T());
                unsigned char* dst = (isA ? A_s : B_s) + (ks >> 2) * ((isA ? TA : TB) * 64);
#pragma unroll
                for (int c = 0; c < SEG; ++c) {
                    const int row = cg * SEG + c;
                    *reinterpret_cast<uint4*>(dst + woff(row, ks & 3)) = o[c];
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
        uint4 af[TM], bfr[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = wm * (TA / 2) + i * 16 + fr;
            af[i] = *reinterpret_cast<const uint4*>(A_s + kb * TA * 64 + woff(row, fs));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int row = wn * (TB / 2) + j * 16 + fr;
            bfr[j] = *reinterpret_cast<const uint4*>(B_s + kb * TB * 64 + woff(row, fs));
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if constexpr (sizeof(T) == 2) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af[i]),
                                                                        __builtin_bit_cast(bf16x8_t, bfr[j]), acc[i][j], 0, 0, 0);
                } else {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(af[i].x), __uint_as_float(bfr[j].x), acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(af[i].y), __uint_as_float(bfr[j].y), acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(af[i].z), __uint_as_float(bfr[j].z), acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(af[i].w), __uint_as_float(bfr[j].w), acc[i][j], 0, 0, 0);
                }
            }
        if (do_bsum) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if constexpr (sizeof(T) == 2) {
                    accb[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ones), __builtin_bit_cast(bf16x8_t, bfr[j]), accb[j], 0, 0, 0);
                } else {
                    accb[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, __uint_as_float(bfr[j].x), accb[j], 0, 0, 0);
                    accb[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, __uint_as_float(bfr[j].y), accb[j], 0, 0, 0);
                    accb[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, __uint_as_float(bfr[j].z), accb[j], 0, 0, 0);
                    accb[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, __uint_as_float(bfr[j].w), accb[j], 0, 0, 0);
                }
            }
        }
        }
    }
    // small weight tensors: every split stores into its own copy (m1_wg_rx_finish folds them) -- hundreds of blocks
    // adding into the same few cache lines serialise at the memory-side atomic unit (~85 ns per request and line)
    float* const Rx = p.Rx ? p.Rx + (long long)bz * p.rx_stride : nullptr;
    if (do_bsum && fs == 0) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int b = b0 + wn * (TB / 2) + j * 16 + fr;
            if (b < p.CB) { if (Rx) Rx[p.rx_bias + b] = accb[j][0]; else atomicAdd(p.bsum + b + p.b_off, accb[j][0]); }
        }
    }
    // D[i = a][j = b]: lane holds a = 4*fs + r, b = fr
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int b = b0 + wn * (TB / 2) + j * 16 + fr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int a = a0 + wm * (TA / 2) + i * 16 + fs * 4 + r;
                if (a < p.CA && b < p.CB) {
                    const long long idx = (long long)tap * p.RT + (long long)(a + p.a_off) * p.RSA + (b + p.b_off);
                    if (Rx) Rx[((long long)tap * p.CA + a) * p.CB + b] = acc[i][j][r]; else atomicAdd(p.R + idx, acc[i][j][r]);
                }
            }
        }
}

bool m1_mfma_wgrad_supported(const WgradSpec& g) {
    return (long long)g.N * g.BD * g.BH * g.BW < (1ll << 31) - 4096;       // 32-bit voxel arithmetic in the kernel
}

template <typename T, int TA, int TB>
static int launch_wg(WgP p, hipStream_t st) {
    constexpr int KB = (TA + TB) <= 64 ? 8 : ((TA + TB) <= 128 ? 4 : 2);
    constexpr int KS = 4 * WT<T>::SEG * KB;
    const int aTiles = (p.CA + TA - 1) / TA, bTiles = (p.CB + TB - 1) / TB;
    const int taps = p.kd * p.kh * p.kw;
    const long long TV = (long long)p.N * p.BD * p.BH * p.BW;
    // ~2 blocks per CU: every extra voxel split adds TA*TB float atomics per tap, and those are executed at the memory
    // side on this part (measured: 1536 -> 512 target blocks = -28 % on the 128x128x27-tap layers).  Rounded DOWN: 512 = 2 per
    // CU; one block more is a second round for its whole XCD (-8 % on the wgrad family).  M1_WG_BLOCKS overrides.
    int tgt = M1_CFG("M1_WG_BLOCKS", 512);
    int rdn = M1_CFG("M1_WG_FLOOR", 1);
    long long splits = rdn ? tgt / ((long long)aTiles * bTiles * taps) : cdiv_ll(tgt, (long long)aTiles * bTiles * taps);
    const long long max_splits = cdiv_ll(TV, 4 * KS);
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    long long vps = cdiv_ll(cdiv_ll(TV, splits), KS) * KS;
    splits = cdiv_ll(TV, vps);
    p.vox_per_split = vps;
    dim3 grid(aTiles * bTiles, taps, (unsigned)splits);
    // Partial copies instead of atomics: every voxel split stores its tile into its own compact copy ([tap][a][b] of THIS member +
    // CB bias sums) and m1_wg_rx_finish folds the copies in a fixed order -- no floating-point atomics, so the weight gradient is
    // bit-reproducible run to run (M1_WG_DET=0 restores the atomic path for large blocks).  Fewer splits when the caller's
    // scratch holds fewer copies; one split needs none (a single add per element and launch is ordered by the stream).
    bool partial = false;
    const long long stride = (long long)taps * p.CA * p.CB + p.CB;
    p.rx_stride = stride; p.rx_bias = (long long)taps * p.CA * p.CB;
    int det = M1_CFG("M1_WG_DET", 1);
    const bool small = (long long)taps * p.CA * p.CB <= 32768 && splits >= 24;
    if (splits >= 2 && (det || small)) {
        long long fit = p.rx ? p.rx_floats / stride : 0;
        if (fit >= 2) {
            if (splits > fit) { splits = fit; vps = cdiv_ll(cdiv_ll(TV, splits), KS) * KS; splits = cdiv_ll(TV, vps); p.vox_per_split = vps; grid.z = (unsigned)splits; }
            if (splits >= 2) { p.Rx = p.rx; partial = true; }
        } else if (det) { splits = 1; vps = cdiv_ll(TV, KS) * KS; p.vox_per_split = vps; grid.z = 1; }
    }
    if (!partial) { p.Rx = nullptr; }
    int xr = M1_CFG("M1_WG_XCD", 1);
    p.xcd_total = 0; p.xcd_gx = (int)grid.x; p.xcd_gy = (int)grid.y; p.xcd_gz = (int)grid.z;
    if (xr && (long long)grid.x * grid.y > 1 && splits > 1) {
        p.xcd_total = (int)(((long long)grid.x * grid.y * grid.z + 7) / 8 * 8);
        m1_note_kernel("wgrad_mfma");
        hipLaunchKernelGGL((wgrad_mfma_kernel<T, TA, TB, KB>), dim3((unsigned)p.xcd_total), dim3(256), 0, st, p);
    } else {
        m1_note_kernel("wgrad_mfma");
        hipLaunchKernelGGL((wgrad_mfma_kernel<T, TA, TB, KB>), grid, dim3(256), 0, st, p);
    }
    int rc = m1_check_launch(); if (rc) return rc;
    if (partial) return m1_wg_rx_finish(p.Rx, stride, (int)splits, p, p.rx_bias, st);
    return M1_OK;
}

static inline int pick_t(int c) { return c > 64 ? 128 : (c > 32 ? 64 : 32); }

template <typename T>
static int run_wg(const WgradSpec& g, long long nw, int nb, hipStream_t st) {
    WgP p; static_cast<WgradSpec&>(p) = g; p.vox_per_split = 0; p.Rx = nullptr; p.rx_stride = nw > 0 ? nw + nb : 0; p.rx_bias = nw;
    const int ta = pick_t(g.CA), tb = pick_t(g.CB);
#define WG_CASE(A_, B_) if (ta == A_ && tb == B_) return launch_wg<T, A_, B_>(p, st);
    WG_CASE(128, 128) WG_CASE(128, 64) WG_CASE(128, 32)
    WG_CASE(64, 128) WG_CASE(64, 64) WG_CASE(64, 32)
    WG_CASE(32, 128) WG_CASE(32, 64) WG_CASE(32, 32)
#undef WG_CASE
    return M1_ERR_UNSUPPORTED;
}

int m1_mfma_wgrad(const WgradSpec& g, hipStream_t st) {
    return g.dtype == M1_BF16 ? run_wg<bf16_t>(g, 0, 0, st) : run_wg<float>(g, 0, 0, st);
}
// nw / nb: floats of the whole weight / bias gradient g.R / g.bsum point into (enables the partial-copy epilogue)
int m1_mfma_wgrad_ex(const WgradSpec& g, long long nw, int nb, hipStream_t st) {
    return g.dtype == M1_BF16 ? run_wg<bf16_t>(g, nw, nb, st) : run_wg<float>(g, nw, nb, st);
}
