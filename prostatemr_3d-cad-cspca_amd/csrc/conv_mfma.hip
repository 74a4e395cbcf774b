// conv_mfma.hip -- implicit-GEMM Conv3D / Conv3DTranspose (TF 'same') on the CDNA4 matrix cores.
//
//   D[voxel][oc] = sum_{tap} sum_{c} X[voxel (+) tap][c] * Wp[oc][tap][c]        M = voxels, N = oc, K = taps*C
//
// * One kernel serves conv forward / convT dgrad (mode 0, in = o*s + k - p) and conv dgrad / convT forward
//   (mode 1, in = (o + p - k)/s, one output PARITY CLASS per blockIdx.y with only its own taps).
// * A operand: im2col gathered on the fly, 16-byte segments (8 bf16 / 4 fp32 channels) straight from the NDHWC
//   tensors of the virtual concat (never materialised); zero fill outside the volume (TF 'same' padding).
// * B operand: weights pre-packed K-contiguous per output channel ([oc][tap][c], same dtype as activations).
// * LDS tiles are 64-byte rows (one K-chunk = 32 bf16 / 16 fp32) with the segment XOR-swizzle
//   seg' = seg ^ ((-(row>>2))&3), which makes every ds_read_b128 fragment read conflict-free on gfx950's
//   16-lane b128 groups; double buffered, global loads for chunk i+1 are in flight while chunk i is on the MFMAs.
// * wave64 tiles of 16x16: bf16 -> v_mfma_f32_16x16x32_bf16 (one per tile per chunk), fp32 ->
//   4 x v_mfma_f32_16x16x4_f32 (exact fp32).  fp32 accumulation in both.
// * epilogue: + bias, convert, stage the block tile in LDS, 16-byte coalesced NDHWC stores.
#include "common.h"
#include "gather.h"
#include "reduce.h"

#include "conv_mfma.h"
#include "conv_t3.h"
#include <stdlib.h>

__device__ __forceinline__ int swz(int row, int seg) { return seg ^ ((-(row >> 2)) & 3); }

// n (< SEG) elements starting at p, zero padded to one 16-byte segment (element-wise loads: p need not be aligned)
template <typename T>
__device__ __forceinline__ uint4 load_partial_seg(const T* p, int n) {
    constexpr int SEG = MT<T>::SEG;
    union { uint4 v; T e[SEG]; } u;
    u.v = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < SEG; ++k) if (k < n) u.e[k] = p[k];
    return u.v;
}

// 64 zero bytes: the source of every LDS-DMA piece that falls outside the volume / beyond the K range
__device__ __attribute__((aligned(64))) unsigned int m1_zero_page[16];

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
// one wave-instruction: 64 lanes x 16 bytes from per-lane global addresses to lds_wave_base + lane*16
__device__ __forceinline__ void glds16(const void* g, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)lds_wave_base, 16, 0, 0);
}

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4_t lds_read128(unsigned lds_addr) {
    u32x4_t v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(lds_addr) : "memory");
    return v;
}
// all outstanding LDS reads have landed; tying the fragment makes its consumers wait behind this statement
__device__ __forceinline__ void lds_wait(u32x4_t& v) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v)); }
// no instruction: only orders the consumers of v behind the preceding (volatile) wait
__device__ __forceinline__ void lds_tie(u32x4_t& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ unsigned lds_addr_of(const void* p) { return (unsigned)(unsigned long long)(lptr_t)p; }

// GLDS = true (every concat member a multiple of one 64-byte K-chunk): both operands go global -> LDS by LDS-DMA
// (global_load_lds_dwordx4), no staging registers and no ds_write pass -- the VGPR -> LDS store path (<= 85 B/clk/CU)
// was the bound of the register-staged loop.  The DMA writes lane-linear, so the XOR swizzle is applied on the SOURCE
// side: the lane that owns LDS slot (row, s) fetches K-segment s ^ swz(row).
// KG > 1 (round 6, LDS-DMA path): K GROUPS inside the block instead of split-K across blocks for the deep levels' small launches
// (M = 4,000 / 500 voxels per sample: a few hundred tiles).  The block carries KG wave groups of WM x WN waves; group g walks the g-th
// K range of the tile with its own pipeline buffers, the groups' accumulators are added through LDS in group order, and group 0 runs
// the ordinary epilogue -- InstanceNorm statistics (or InstanceNorm-backward sums) included.  Against split-K over blockIdx.y: the same
// waves per CU, no fp32 slabs, no finish / statistics pass over them (FinishStatsF: 20 launches of 15-26 us per C3 step).
template <typename T, int BM, int BN, int WM, int WN, int KC, bool GLDS, int KG = 1>
__global__ void __launch_bounds__(WM * WN * 64 * KG) conv_mfma_kernel(MfmaP p) {
#if defined(__HIP_DEVICE_COMPILE__)      // (buffer-resource builtins do not exist in the host pass)
    constexpr int NTHR = WM * WN * 64, NW = WM * WN;        // threads / waves of ONE K group: 4 waves, or 8 for the 128x128 tile of the deep layers
    static_assert(KG == 1 || GLDS, "K groups: LDS-DMA path only");
    constexpr int SEG = MT<T>::SEG;
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64;
    constexpr int A_LD = BM * 4 / NTHR;                 // 16-B loads per thread per chunk for A
    constexpr int B_LD = (BN * 4 + NTHR - 1) / NTHR;    // for B
    static_assert((NW == 4 || NW == 8 || NW == 16) && (BM * 4) % NTHR == 0 && BN % 16 == 0, "tile config");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // tables first (they must survive the epilogue tile, which reuses the pipeline buffers)
    int4* rowinfo = reinterpret_cast<int4*>(smem);                                 // [BM] {n, bd, bh, bw}
    int* outrow = reinterpret_cast<int*>(rowinfo + BM);                            // [BM] output voxel index or -1
    const T** s_src = reinterpret_cast<const T**>(outrow + BM);                    // [6]
    int* s_srcC = reinterpret_cast<int*>(s_src + M1_MAX_SRC);                      // [6]
    int* s_srcSeg = s_srcC + M1_MAX_SRC;                                           // [6]
    int* s_tap = s_srcSeg + M1_MAX_SRC;                                            // [27] packed dd|dh|dw
    int* s_tdl = s_tap + MF_MAX_TAPS;                                              // [27] the same as a linear voxel delta
    constexpr int TBL_BYTES = (BM * 20 + M1_MAX_SRC * 16 + 2 * MF_MAX_TAPS * 4 + 15) / 16 * 16;
    // (tid, wave: inside the K group -- everything below is group-local; the tables are written by every group with the same values)
    const int kg = KG > 1 ? __builtin_amdgcn_readfirstlane((int)threadIdx.x / NTHR) : 0;
    const int tid = KG > 1 ? (int)threadIdx.x - kg * NTHR : (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int GROUP_BYTES = 2 * KC * (A_BYTES + B_BYTES);
    unsigned char* A_s = smem + TBL_BYTES + kg * GROUP_BYTES;   // [2][KC][BM][64]
    unsigned char* B_s = A_s + 2 * KC * A_BYTES;         // [2][KC][BN][64]

    const int wm = wave / WN, wn = wave % WN;
    const int cls = blockIdx.y / p.ksplit, ksp = (blockIdx.y % p.ksplit) * KG + kg;     // K range of this group: ksp of ksplit * KG
    const int oc0 = blockIdx.z * BN;

    int pdc = 0, phc = 0, pwc = 0, QD = p.OD, QH = p.OH, QW = p.OW;
    if (p.mode == 1) {
        pwc = cls % p.sw; phc = (cls / p.sw) % p.sh; pdc = cls / (p.sw * p.sh);
        QD = p.OD > pdc ? (p.OD - pdc + p.sd - 1) / p.sd : 0;
        QH = p.OH > phc ? (p.OH - phc + p.sh - 1) / p.sh : 0;
        QW = p.OW > pwc ? (p.OW - pwc + p.sw - 1) / p.sw : 0;
    }
    const long long QV = (long long)QD * QH * QW, Mtot = QV * p.N;
    // Workgroups go to the 8 XCDs round-robin (block b -> XCD b & 7): hand each XCD one contiguous range of M tiles so
    // that the halo voxels neighbouring tiles gather are shared through that XCD's L2.  gridDim.x = 8 * ceil(tiles / 8).
    const long long tile = (long long)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    long long m0 = tile * BM, m_end = Mtot;
    if (p.tps > 0) {               // tiles per sample: the last tile of a sample is cut at the sample's end
        if (tile >= (long long)p.tps * p.N) return;
        const long long tn = tile / p.tps;
        m0 = tn * QV + (tile - tn * p.tps) * BM; m_end = (tn + 1) * QV;
    }
    if (m0 >= Mtot) return;

    for (int r = tid; r < BM; r += NTHR) {
        const long long m = m0 + r;
        int4 ri = make_int4(-1, 0, 0, 0); int orow = -1;
        if (m < m_end) {
            int n, qw, qh, qd;
            if (Mtot < (1ll << 31)) {         // (every tensor the 31-bit LDS-DMA offsets can address: 32-bit divisions, ~4x fewer instructions)
                const unsigned mm = (unsigned)m, qv = (unsigned)QV;
                n = (int)(mm / qv); const unsigned lin = mm - (unsigned)n * qv;
                const unsigned t1 = lin / (unsigned)QW; qw = (int)(lin - t1 * (unsigned)QW);
                qd = (int)(t1 / (unsigned)QH); qh = (int)(t1 - (unsigned)qd * (unsigned)QH);
            } else {
                n = (int)(m / QV); long long lin = m % QV;
                qw = (int)(lin % QW); lin /= QW; qh = (int)(lin % QH); qd = (int)(lin / QH);
            }
            if (p.mode == 0) {
                ri = make_int4(n, qd * p.sd - p.pd, qh * p.sh - p.ph, qw * p.sw - p.pw);
                orow = (int)m;
            } else {
                ri = make_int4(n, qd, qh, qw);
                orow = (int)((((long long)n * p.OD + qd * p.sd + pdc) * p.OH + qh * p.sh + phc) * p.OW + qw * p.sw + pwc);
            }
        }
        rowinfo[r] = ri; outrow[r] = orow;
    }
    if (tid < M1_MAX_SRC) { s_src[tid] = (const T*)p.src[tid]; s_srcC[tid] = p.srcC[tid]; s_srcSeg[tid] = p.srcSeg[tid]; }
    const int ntaps = p.cls_ntaps[cls], tfirst = p.cls_first[cls];
    if (tid < ntaps) {
        const int t = tfirst + tid;
        s_tap[tid] = ((int)p.tdd[t] & 0xff) | (((int)p.tdh[t] & 0xff) << 8) | (((int)p.tdw[t] & 0xff) << 16);
        s_tdl[tid] = ((int)p.tdd[t] * p.IH + (int)p.tdh[t]) * p.IW + (int)p.tdw[t];
    }
    __syncthreads();

    const int spt = p.spt;                               // segments per tap
    const int nseg = ntaps * spt;
    const int nchunks_all = (nseg + 3) >> 2;
    const int cps = (nchunks_all + p.ksplit * KG - 1) / (p.ksplit * KG);        // chunks per split
    const int c_beg = ksp * cps;
    const int nchunks = (c_beg + cps <= nchunks_all ? cps : nchunks_all - c_beg);   // may be <= 0 for a trailing split
    const int kpad = p.cls_kpad[cls];
    const T* wp = (const T*)p.wp + p.cls_woff[cls];

    f32x4_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    uint4 ra[GLDS ? 1 : KC][GLDS ? 1 : A_LD], rb[GLDS ? 1 : KC][GLDS ? 1 : B_LD];
    const int lrow = tid >> 2;                           // loader: 4 lanes cover one 64-byte row
    // register staging: logical segment tid&3, swizzled when written; LDS-DMA: slot tid&3, swizzled when fetched
    // (rows tid>>2 + (NTHR/4)*i and (tid + NTHR*i)>>2 all share (row>>2)&3 = (tid>>4)&3)
    const int lseg = GLDS ? ((tid & 3) ^ ((-(tid >> 4)) & 3)) : (tid & 3);
    const unsigned char* zero_pg = reinterpret_cast<const unsigned char*>(m1_zero_page);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);

    // incremental loader state (aligned case): current tap / concat member / channel offset and per-row bases
    int st_tap = 0, st_s = 0, st_c = 0;
    const T* st_ptr[A_LD];
    auto st_set_tap = [&]() {          // voxel bases of this thread's rows for tap st_tap, member st_s
        const int tp = s_tap[st_tap];
        const int dd = (signed char)(tp & 0xff), dh = (signed char)((tp >> 8) & 0xff), dw = (signed char)((tp >> 16) & 0xff);
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            const int4 ri = rowinfo[lrow + (NTHR / 4) * i];
            const int id = ri.y + dd, ih = ri.z + dh, iw = ri.w + dw;
            st_ptr[i] = nullptr;
            if (ri.x >= 0 && id >= 0 && id < p.ID && ih >= 0 && ih < p.IH && iw >= 0 && iw < p.IW) {
                const long long vox = (((long long)ri.x * p.ID + id) * p.IH + ih) * p.IW + iw;
                st_ptr[i] = s_src[st_s] + vox * s_srcC[st_s] + lseg * SEG;
            }
        }
    };
    // LDS-DMA variant: buffer loads -- the member tensor / weight panel is the resource, a lane keeps a 32-bit byte offset
    // (2^31 = out of range: the hardware range check returns the zeros of the padding), the chunk offset is the scalar
    // offset: no per-lane 64-bit address arithmetic in the loop.  Per row: the byte offset of its voxel under tap delta 0 in
    // the current member and a bit mask of the taps that stay inside the volume; a tap is then one scalar byte delta.
    constexpr unsigned OOB = 0x80000000u;
    unsigned b_vo[B_LD];
    unsigned rmask[A_LD]; int rvox[A_LD], rbyte[A_LD];
    int tap_db = 0;                                      // byte delta of tap st_tap in member st_s
    __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.src[0], 0, 0, 0x00020000);
    auto glds_set_member = [&]() {
        const int sC = __builtin_amdgcn_readfirstlane(s_srcC[st_s]);
#pragma unroll
        for (int i = 0; i < A_LD; ++i) rbyte[i] = (rvox[i] * sC + lseg * SEG) * (int)sizeof(T);
        rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.src[__builtin_amdgcn_readfirstlane(st_s)], 0, 0x7fffffff, 0x00020000);
    };
    auto glds_set_tap = [&]() {
        tap_db = __builtin_amdgcn_readfirstlane(s_tdl[st_tap]) * __builtin_amdgcn_readfirstlane(s_srcC[st_s]) * (int)sizeof(T);
    };
    if constexpr (GLDS) {
        // tap validity per row: the taps are the OUTER loop (one LDS read per tap, not per (row, tap)); tap offsets are -1 .. 2
        int4 ris[A_LD];
#pragma unroll
        for (int i = 0; i < A_LD; ++i) { ris[i] = rowinfo[lrow + (NTHR / 4) * i]; rmask[i] = 0; }
#pragma unroll 9
        for (int t = 0; t < ntaps; ++t) {
            const int tp = s_tap[t];
            const int dd = (signed char)(tp & 0xff), dh = (signed char)((tp >> 8) & 0xff), dw = (signed char)((tp >> 16) & 0xff);
#pragma unroll
            for (int i = 0; i < A_LD; ++i) {
                const bool ok = ris[i].x >= 0 && (unsigned)(ris[i].y + dd) < (unsigned)p.ID && (unsigned)(ris[i].z + dh) < (unsigned)p.IH &&
                                (unsigned)(ris[i].w + dw) < (unsigned)p.IW;
                rmask[i] |= (ok ? 1u : 0u) << t;
            }
        }
#pragma unroll
        for (int i = 0; i < A_LD; ++i) rvox[i] = ((ris[i].x * p.ID + ris[i].y) * p.IH + ris[i].z) * p.IW + ris[i].w;
    }
    if constexpr (GLDS) {
#pragma unroll
        for (int i = 0; i < B_LD; ++i) {
            const int e = tid + NTHR * i;
            b_vo[i] = (unsigned)(((oc0 + (e >> 2)) * kpad + lseg * SEG) * (int)sizeof(T));
        }
    }
    if (p.aligned && nchunks > 0) {
        const int cpt = spt >> 2;                          // chunks per tap
        int c;
        if (p.korder == 2) { const int cg = c_beg / (2 * ntaps), rem = c_beg - cg * 2 * ntaps; st_tap = rem >> 1; c = (cg * 2 + (rem & 1)) * 4 * SEG; }
        else if (p.korder) { const int cg = c_beg / ntaps; st_tap = c_beg - cg * ntaps; c = cg * 4 * SEG; }    // [chunk][tap] K order
        else { st_tap = c_beg / cpt; c = (c_beg - st_tap * cpt) * 4 * SEG; }                               // [tap][chunk]
        while (c >= s_srcC[st_s]) { c -= s_srcC[st_s]; ++st_s; }
        st_c = c;
        if constexpr (GLDS) { glds_set_member(); glds_set_tap(); } else st_set_tap();
    }

    // loads the KC chunks of pipeline stage `sg` into registers (zero beyond this block's K range)
    auto prefetch_stage = [&](int sg) {
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            const int rel = sg * KC + kc;
            const int chunk = c_beg + rel;
            const bool live = rel < nchunks;
            // ---- B: packed weights, rows = output channels ----
#pragma unroll
            for (int i = 0; i < B_LD; ++i) {
                const int e = tid + NTHR * i;                 // (row, seg) = (e>>2, e&3)
                uint4 v = make_uint4(0, 0, 0, 0);
                if (live && e < BN * 4)
                    v = *reinterpret_cast<const uint4*>(wp + (long long)(oc0 + (e >> 2)) * kpad + (long long)(chunk * 4 + (e & 3)) * SEG);
                rb[kc][i] = v;
            }
            // ---- A: gathered activations ----
            if (!live) {
#pragma unroll
                for (int i = 0; i < A_LD; ++i) ra[kc][i] = make_uint4(0, 0, 0, 0);
            } else if (p.aligned) {
#pragma unroll
                for (int i = 0; i < A_LD; ++i)
                    ra[kc][i] = st_ptr[i] ? *reinterpret_cast<const uint4*>(st_ptr[i] + st_c) : make_uint4(0, 0, 0, 0);
                st_c += 4 * SEG;
                if (st_c >= s_srcC[st_s]) {                // next concat member, or next tap
                    st_c = 0;
                    if (++st_s == p.nsrc) { st_s = 0; ++st_tap; }
                    if (st_tap < ntaps) st_set_tap();
                }
            } else {
                const int kseg = chunk * 4 + lseg;
                const T* sp = nullptr; int sC = 0, coff = 0, dd = 0, dh = 0, dw = 0;
                const bool kvalid = kseg < nseg;
                if (kvalid) {
                    const int tap_i = kseg / spt;
                    int cs = kseg - tap_i * spt, s = 0;
                    while (cs >= s_srcSeg[s]) { cs -= s_srcSeg[s]; ++s; }
                    sp = s_src[s]; sC = s_srcC[s]; coff = cs * SEG;
                    const int tp = s_tap[tap_i];
                    dd = (signed char)(tp & 0xff); dh = (signed char)((tp >> 8) & 0xff); dw = (signed char)((tp >> 16) & 0xff);
                }
#pragma unroll
                for (int i = 0; i < A_LD; ++i) {
                    const int4 ri = rowinfo[lrow + (NTHR / 4) * i];
                    uint4 v = make_uint4(0, 0, 0, 0);
                    const int id = ri.y + dd, ih = ri.z + dh, iw = ri.w + dw;
                    if (kvalid && coff < sC && ri.x >= 0 && id >= 0 && id < p.ID && ih >= 0 && ih < p.IH && iw >= 0 && iw < p.IW) {
                        const long long vox = (((long long)ri.x * p.ID + id) * p.IH + ih) * p.IW + iw;
                        if (sC % SEG == 0) v = *reinterpret_cast<const uint4*>(sp + vox * sC + coff);
                        else v = load_partial_seg<T>(sp + vox * sC + coff, sC - coff < SEG ? sC - coff : SEG);
                    }
                    ra[kc][i] = v;
                }
            }
        }
    };
    // LDS-DMA: issue the KC chunks of stage `sg` straight into pipeline buffer `buf`
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)wp, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_0 = __builtin_amdgcn_make_buffer_rsrc((void*)wp, 0, 0, 0x00020000);     // all lanes out of range
    auto issue_chunk = [&](int kc, int buf, int chunk, const __amdgpu_buffer_rsrc_t ra, const __amdgpu_buffer_rsrc_t rb) {
#pragma unroll
        for (int i = 0; i < B_LD; ++i)
            if ((NW * (i + 1)) * 64 <= BN * 4 || (wave_u + NW * i) * 64 < BN * 4)      // compile-time true for full rounds, else wave-uniform
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lptr_t)(B_s + (buf * KC + kc) * B_BYTES + (wave_u + NW * i) * 1024), 16,
                                                         b_vo[i], chunk * 64, 0, 0);
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            const unsigned vo = ((rmask[i] >> st_tap) & 1u) ? (unsigned)(rbyte[i] + tap_db) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lptr_t)(A_s + (buf * KC + kc) * A_BYTES + (wave_u + NW * i) * 1024), 16,
                                                     vo, st_c * (int)sizeof(T), 0, 0);
        }
    };
    // K order [32-channel chunk][tap] (p.korder): the 27 taps of one chunk follow each other, so the rows a block gathers for tap
    // t+1 are the rows it (and its neighbours) just fetched for tap t, one voxel over -- they are still in L2.  In [tap][chunk]
    // order a 128-row tile walks all Cin channels (128 KB at Cin = 512) before it returns to a voxel: with ~64 tiles in flight
    // per XCD that is twice its 4 MB L2, and every tap re-fetches from the Infinity Cache / HBM.
    auto advance_chunk = [&]() {
        if (p.korder == 2) {
            if (((st_c / (4 * SEG)) & 1) == 0) st_c += 4 * SEG;           // second chunk of the pair, same tap
            else {
                st_c -= 4 * SEG;
                if (++st_tap == ntaps) {
                    st_tap = 0; st_c += 8 * SEG;
                    if (st_c >= s_srcC[st_s]) { st_c = 0; if (++st_s < p.nsrc) glds_set_member(); }
                }
                if (st_s < p.nsrc) glds_set_tap();
            }
        } else if (p.korder) {
            if (++st_tap == ntaps) {
                st_tap = 0; st_c += 4 * SEG;
                if (st_c >= s_srcC[st_s]) { st_c = 0; if (++st_s < p.nsrc) glds_set_member(); }
            }
            if (st_s < p.nsrc) glds_set_tap();
        } else {
            st_c += 4 * SEG;
            if (st_c >= s_srcC[st_s]) {
                st_c = 0;
                if (++st_s == p.nsrc) { st_s = 0; ++st_tap; }
                if (st_tap < ntaps) { glds_set_member(); glds_set_tap(); }
            }
        }
    };
    auto issue = [&](int sg, int buf) {
        const int rel0 = sg * KC;
        if (rel0 + KC <= nchunks) {                        // every chunk of the stage is inside the K range (all but the last stage)
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) { issue_chunk(kc, buf, c_beg + rel0 + kc, rs_a, rs_b); advance_chunk(); }
        } else {
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) {
                const bool live = rel0 + kc < nchunks;
                issue_chunk(kc, buf, c_beg + rel0 + kc, live ? rs_a : rs_0, live ? rs_b : rs_0);
                if (live) advance_chunk();
            }
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
#pragma unroll
            for (int i = 0; i < A_LD; ++i) {
                const int row = lrow + (NTHR / 4) * i;
                *reinterpret_cast<uint4*>(A_s + (buf * KC + kc) * A_BYTES + row * 64 + swz(row, lseg) * 16) = ra[kc][i];
            }
#pragma unroll
            for (int i = 0; i < B_LD; ++i) {
                const int e = tid + NTHR * i;
                if (e < BN * 4) {
                    const int row = e >> 2;
                    *reinterpret_cast<uint4*>(B_s + (buf * KC + kc) * B_BYTES + row * 64 + swz(row, e & 3) * 16) = rb[kc][i];
                }
            }
        }
    };

    const int fr = lane & 15, fs = lane >> 4;
    // fragment read addresses of tile (i or j) = base + 1024*tile (16 rows x 64 B; the swizzle term is tile-invariant)
    const unsigned a_rd = lds_addr_of(A_s) + (wm * (BM / WM) + fr) * 64 + swz(fr, fs) * 16;
    const unsigned b_rd = lds_addr_of(B_s) + (wn * (BN / WN) + fr) * 64 + swz(fr, fs) * 16;
    // (K groups share the block's barriers: every group runs the stages of a FULL range; beyond its own range a group stages zeros)
    const int nstages = KG > 1 ? (cps + KC - 1) / KC : (nchunks + KC - 1) / KC;
    if (nstages > 0) {
        if constexpr (GLDS) issue(0, 0);
        else { prefetch_stage(0); stage(0); }
    }
    __syncthreads();
    for (int it = 0; it < nstages; ++it) {
        const int buf = it & 1;
        if (it + 1 < nstages) { if constexpr (GLDS) issue(it + 1, buf ^ 1); else prefetch_stage(it + 1); }
        if constexpr (GLDS) {
            // Fragment reads by inline asm: hipcc cannot tell an LDS-DMA in flight from the buffer being read and would
            // wait vmcnt(0) before the first ds_read of the stage (no load/compute overlap inside a block).  The reads
            // of chunk kc+1 are issued before the MFMAs of chunk kc.
            u32x4_t af[KC][TM], bfr[KC][TN];
            auto rd = [&](int kc) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[kc][i] = lds_read128(a_rd + (buf * KC + kc) * A_BYTES + i * 1024);
#pragma unroll
                for (int j = 0; j < TN; ++j) bfr[kc][j] = lds_read128(b_rd + (buf * KC + kc) * B_BYTES + j * 1024);
            };
            rd(0);
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) {
                lds_wait(af[kc][0]);                       // one s_waitcnt for the whole chunk, the other fragments are tied to it
#pragma unroll
                for (int i = 1; i < TM; ++i) lds_tie(af[kc][i]);
#pragma unroll
                for (int j = 0; j < TN; ++j) lds_tie(bfr[kc][j]);
                if (kc + 1 < KC) rd(kc + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        if constexpr (sizeof(T) == 2) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af[kc][i]),
                                                                                __builtin_bit_cast(bf16x8_t, bfr[kc][j]), acc[i][j], 0, 0, 0);
                        } else {
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(af[kc][i][q]), __uint_as_float(bfr[kc][j][q]), acc[i][j], 0, 0, 0);
                        }
                    }
                if (kc + 1 < KC) __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            uint4 af[TM], bfr[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = wm * (BM / WM) + i * 16 + fr;
                af[i] = *reinterpret_cast<const uint4*>(A_s + (buf * KC + kc) * A_BYTES + row * 64 + swz(row, fs) * 16);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = wn * (BN / WN) + j * 16 + fr;
                bfr[j] = *reinterpret_cast<const uint4*>(B_s + (buf * KC + kc) * B_BYTES + row * 64 + swz(row, fs) * 16);
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
        }
        }
        if constexpr (!GLDS) { if (it + 1 < nstages) stage(buf ^ 1); }
        __syncthreads();          // (GLDS: the barrier's fence also waits for the stage in flight, vmcnt(0))
    }

    if constexpr (KG > 1) {        // add the groups' accumulators in group order (fixed: deterministic); groups > 0 are done then
        float* const xr = reinterpret_cast<float*>(smem + TBL_BYTES);          // [(KG - 1)][NTHR][TM * TN * 4]: the pipeline buffers are free
        if (kg > 0) {
            float* dst = xr + ((size_t)(kg - 1) * NTHR + tid) * (TM * TN * 4);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) *reinterpret_cast<f32x4_t*>(dst + (i * TN + j) * 4) = acc[i][j];
        }
        __syncthreads();
        if (kg > 0) return;
#pragma unroll
        for (int g = 1; g < KG; ++g) {
            const float* src = xr + ((size_t)(g - 1) * NTHR + tid) * (TM * TN * 4);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] += *reinterpret_cast<const f32x4_t*>(src + (i * TN + j) * 4);
        }
        __syncthreads();           // (group 0 alone from here: the epilogue tile overwrites what was just read)
    }
    if (p.ksplit > 1) {            // partial K range: plain stores into this split's own fp32 slab; the finish kernel adds the
                                   // slabs in a fixed order (run-to-run deterministic, unlike fp32 atomics).  A trailing split
                                   // with no chunks stores its zeros, so the slabs need no memset.
        float* slab = p.acc32 + (long long)ksp * p.slab_elems;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int oc = oc0 + wn * (BN / WN) + j * 16 + fr;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int orow = outrow[wm * (BM / WM) + i * 16 + fs * 4 + r];
                    if (orow >= 0 && oc < p.OCn) slab[(long long)orow * p.OC + oc] = acc[i][j][r];
                }
            }
        return;
    }
    // ---- epilogue: acc (+bias) -> T -> LDS tile [BM][BN] (row pitch BN+SEG to spread banks) -> 16-B stores ----
    constexpr int CP = BN + SEG;                          // pitch in elements
    T* C_s = reinterpret_cast<T*>(A_s);                   // reuses the A/B buffers (all waves are past the last sync)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = wn * (BN / WN) + j * 16 + fr;
            const float bv = (oc0 + col < p.OCn) ? m1_bias_at(p, oc0 + col) : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wm * (BM / WM) + i * 16 + fs * 4 + r;
                Act<T>::st(C_s + row * CP + col, acc[i][j][r] + bv);
            }
        }
    __syncthreads();
    if (p.stat_partial) {          // per-tile column sums of the stored (rounded) values -> deterministic partials
        constexpr int EPI_BYTES = (int)((BM * CP * sizeof(T) + 15) / 16 * 16), PIPE_BYTES = 2 * KC * (A_BYTES + B_BYTES) * KG;
        float* red = reinterpret_cast<float*>(A_s + (PIPE_BYTES > EPI_BYTES ? PIPE_BYTES : EPI_BYTES));   // 2 KB scratch behind both
        constexpr int G = NTHR / BN;                      // row groups
        const int col = tid % BN, rg = tid / BN;
        float s = 0.f, ss = 0.f;
        if (p.ib_x) {              // InstanceNorm backward of the tensor this tile is the gradient of: {sum dy, sum dy * xh}
            const int oc = oc0 + col < p.OCn ? oc0 + col : 0, n = (int)(m0 / QV);
            const float mean = p.ib_stats[((long long)n * p.OC + oc) * 2], rstd = p.ib_stats[((long long)n * p.OC + oc) * 2 + 1];
            const float gm = p.ib_gamma[oc], bt = p.ib_beta[oc];
            const T* xb = (const T*)p.ib_x + oc;
            // all x values of this thread's rows first (independent loads in flight together): one row at a time the loop was
            // BM/G dependent global round trips -- 16 on the 32-column tiles of the SE bottleneck convs, 85 us for a 21 us layer
            constexpr int RPT = (BM + G - 1) / G;
            if constexpr (RPT > 32) {                 // (wide tiles with few row groups: keep the rolled loop)
                for (int row = rg; row < BM; row += G) {
                    const int orow = outrow[row];
                    if (orow >= 0) {
                        const float xh = (Act<T>::ld(xb + (long long)orow * p.OC) - mean) * rstd;
                        const float dy = Act<T>::ld(C_s + row * CP + col) * lrelu_g(gm * xh + bt, p.ib_slope);
                        s += dy; ss += dy * xh;
                    }
                }
            } else {
            float xv[RPT];
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                const int row = rg + i * G;
                const int orow = row < BM ? outrow[row] : -1;
                xv[i] = orow >= 0 ? Act<T>::ld(xb + (long long)orow * p.OC) : 0.f;
            }
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                const int row = rg + i * G;
                if (row < BM && outrow[row] >= 0) {
                    const float xh = (xv[i] - mean) * rstd;
                    const float dy = Act<T>::ld(C_s + row * CP + col) * lrelu_g(gm * xh + bt, p.ib_slope);
                    s += dy; ss += dy * xh;
                }
            }
            }
        } else
        for (int row = rg; row < BM; row += G) {
            if (outrow[row] >= 0) { const float v = Act<T>::ld(C_s + row * CP + col); s += v; ss += v * v; }
        }
        red[tid * 2] = s; red[tid * 2 + 1] = ss;
        __syncthreads();
        if (rg == 0 && oc0 + col < p.OCn) {
            for (int q = 1; q < G; ++q) { s += red[(q * BN + col) * 2]; ss += red[(q * BN + col) * 2 + 1]; }
            // mode 0: tiles are sample-major, stat_tiles per sample
            float* dst = p.stat_partial + ((tile * p.OC) + oc0 + col) * 2;
            dst[0] = s; dst[1] = ss;
        }
    }
    constexpr int SPR = BN / SEG;                         // 16-B segments per tile row
    for (int e = tid; e < BM * SPR; e += NTHR) {
        const int row = e / SPR, cs = e % SPR;
        const int orow = outrow[row];
        const int oc = oc0 + cs * SEG;
        if (orow < 0 || oc >= p.OCn) continue;
        const OutRef o = m1_out_ref(p, oc);                // (a segment never straddles two destination tensors: their widths are multiples of SEG)
        if (!o.base) continue;
        uint4 v = *reinterpret_cast<const uint4*>(C_s + row * CP + cs * SEG);
        T* dst = (T*)o.base + (long long)orow * o.C + o.col;
        if (o.C % SEG != 0 || (p.nout == 0 && oc + SEG > p.OCn)) {       // output row not 16-byte tiled (dz: 1..3 channels; class logits)
            union { uint4 q; T e[SEG]; } ve; ve.q = v;
#pragma unroll
            for (int k = 0; k < SEG; ++k) {               // (constant indices: a run-time bound would put the segment in scratch memory)
                if (oc + k < p.OCn) {
                    float a = Act<T>::ld(&ve.e[k]);
                    if (o.acc) a += Act<T>::ld(dst + k);
                    Act<T>::st(dst + k, a);
                }
            }
            continue;
        }
        if (o.acc) {
            float a[SEG], b[SEG];
            VecIO<T, SEG>::ld(reinterpret_cast<const T*>(&v), a);
            VecIO<T, SEG>::ld(dst, b);
#pragma unroll
            for (int k = 0; k < SEG; ++k) a[k] += b[k];
            VecIO<T, SEG>::st(dst, a);
        } else {
            *reinterpret_cast<uint4*>(dst) = v;
        }
    }
#endif
}

template <typename T, int BM, int BN, int KC, int NTHR, int KG = 1>
static constexpr size_t mfma_smem_bytes() {
    size_t tbl = (BM * 20 + M1_MAX_SRC * 16 + 2 * MF_MAX_TAPS * 4 + 15) / 16 * 16;      // = TBL_BYTES of the kernel
    size_t pipe = (size_t)KC * (2 * BM * 64 + 2 * BN * 64) * KG;
    size_t epi = ((size_t)BM * (BN + MT<T>::SEG) * sizeof(T) + 15) / 16 * 16;
    return tbl + (pipe > epi ? pipe : epi) + NTHR * 2 * sizeof(float);
}

// ------------------------------------------------------------------------------------------------
// weight packing:  wp[class][oc][tap_i*CC + c] = w[wtap*wST + c*wSC + (oc+oc_off)*wSO]   (zero padded)
// ------------------------------------------------------------------------------------------------
struct PackP {
    const float* w; long long wST, wSC, wSO; int oc_off, cc_off, OCn, OCpad, CC;
    const float* w2; long long w2ST, w2SC, w2SO; int oc_split, c_split;     // second weight tensor (GatherSpec::w2)
    int nsrc, spt, SEG; int srcC[M1_MAX_SRC], srcSeg[M1_MAX_SRC];
    int nclasses; int cls_ntaps[MF_MAX_CLASSES], cls_first[MF_MAX_CLASSES], cls_kpad[MF_MAX_CLASSES];
    long long cls_woff[MF_MAX_CLASSES];
    unsigned char wtap[MF_MAX_TAPS];
    int korder;              // 0: K = [tap][concat channel]; 1: K = [64-byte chunk of the concat][tap] (MfmaP::korder)
};
// source offset (without the output-channel term) of the first column of packed K-segment kseg (SEG columns = SEG
// consecutive channels of one concat member under one tap) of class cls; *nvalid = its real (non-padding) columns
__device__ __forceinline__ bool pack_seg_pos(const PackP& p, int cls, int kseg, int* nvalid, int* wtap, int* chan) {
    int tap_i, seg;
    if (p.korder == 2) {     // [pair of chunks][tap][chunk of the pair]: a tap reads 128 contiguous bytes of a voxel row
        const int chunk = kseg >> 2, nt = p.cls_ntaps[cls], cg = chunk / (2 * nt), rem = chunk - cg * 2 * nt;
        tap_i = rem >> 1; seg = (cg * 2 + (rem & 1)) * 4 + (kseg & 3);
    } else if (p.korder) {   // [chunk][tap]: 4 segments per chunk (every member is a multiple of one chunk)
        const int chunk = kseg >> 2, nt = p.cls_ntaps[cls], cg = chunk / nt;
        tap_i = chunk - cg * nt; seg = cg * 4 + (kseg & 3);
    } else { tap_i = kseg / p.spt; seg = kseg - tap_i * p.spt; }
    int c = 0, s = 0;
    while (s < p.nsrc && seg >= p.srcSeg[s]) { seg -= p.srcSeg[s]; c += p.srcC[s]; ++s; }
    if (s >= p.nsrc || tap_i >= p.cls_ntaps[cls]) { *nvalid = 0; return false; }
    const int rem = p.srcC[s] - seg * p.SEG;
    *nvalid = rem < p.SEG ? rem : p.SEG;
    *chan = c + seg * p.SEG;                                // channel on the (unpadded) concat axis
    *wtap = p.wtap[p.cls_first[cls] + tap_i];
    return true;
}
// A pack job leaves its own parameters in front of its panel (first M1_PACK_JOB_BYTES of the workspace region), so that
// m1_pack_batch can refresh every panel of a model in ONE launch after the optimiser step (per-layer pack launches are
// ~5 us each, ~150 of them per step otherwise).
#define M1_PACK_JOB_BYTES 512
#define M1_PACK_MAGIC 0x4d315031u
struct PackJob { PackP p; void* out; int dtype; unsigned magic; };
static_assert(sizeof(PackJob) <= M1_PACK_JOB_BYTES, "pack job record");

// one thread per 16-byte panel segment (SEG consecutive K columns of one output channel)
template <typename T>
__device__ __forceinline__ void pack_class_elems(const PackP& p, int cls, T* __restrict__ out, long long first, long long stride) {
    constexpr int SEG = MT<T>::SEG;
    const int kpad = p.cls_kpad[cls], nks = kpad / SEG;
    // the output-channel axis is the contiguous one of the SOURCE (forward panels) or the K axis is (data-gradient panels: a thread
    // reads 32 contiguous bytes, consecutive threads consecutive segments).  Forward panels: a wave covers 16 output channels x 4
    // segments -- 64-byte runs of the source per load and 64-byte runs of the panel per row (one thread per channel along the whole
    // wave wrote 64 panel rows, 16 bytes each: 332 us for the ~120 M panel elements of C3)
    const bool oc_fast = p.wSO == 1;
    const int ocg = p.OCpad >> 4, ksg = (nks + 3) >> 2;
    const long long tot = oc_fast ? (long long)ocg * ksg * 64 : (long long)p.OCpad * nks;
    for (long long u = first; u < tot; u += stride) {
        int oc, kseg;
        if (oc_fast) {
            const long long grp = u >> 6; const int l = (int)(u & 63);
            oc = (int)(grp % ocg) * 16 + (l & 15); kseg = (int)(grp / ocg) * 4 + (l >> 4);
            if (kseg >= nks) continue;
        } else { oc = (int)(u / nks); kseg = (int)(u % nks); }
        int nvalid, wtap = 0, c = 0;
        pack_seg_pos(p, cls, kseg, &nvalid, &wtap, &c);
        if (oc >= p.OCn) nvalid = 0;
        // which weight tensor owns (output column oc, contraction channel c)
        const float* w = p.w; long long wST = p.wST, wSC = p.wSC, wSO = p.wSO; int ocl = oc + p.oc_off, cl = c + p.cc_off;
        if (p.w2 && ((p.oc_split > 0 && oc >= p.oc_split) || (p.c_split > 0 && c >= p.c_split))) {
            w = p.w2; wST = p.w2ST; wSC = p.w2SC; wSO = p.w2SO;
            if (p.oc_split > 0) ocl = oc - p.oc_split + p.oc_off; else cl = c - p.c_split;
        }
        const long long so = (long long)wtap * wST + (long long)cl * wSC + (long long)ocl * wSO;
        float v[SEG];
#pragma unroll
        for (int j = 0; j < SEG; ++j) v[j] = j < nvalid ? w[so + (long long)j * wSC] : 0.f;
        VecIO<T, SEG>::st(out + p.cls_woff[cls] + (long long)oc * kpad + (long long)kseg * SEG, v);
    }
}
template <typename T>
__global__ void pack_weights_kernel(PackP p, T* __restrict__ out, PackJob* __restrict__ job) {
    if (job && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        job->p = p; job->out = out; job->dtype = sizeof(T) == 2 ? M1_BF16 : M1_F32; job->magic = M1_PACK_MAGIC;
    }
    pack_class_elems<T>(p, blockIdx.y, out, (long long)blockIdx.x * blockDim.x + threadIdx.x, (long long)gridDim.x * blockDim.x);
}
// grid (blocks per job, jobs): re-runs recorded jobs from their device-resident records
__global__ void __launch_bounds__(256) pack_batch_kernel(const PackJob* const* __restrict__ jobs) {
    __shared__ PackJob j;
    const unsigned* src = reinterpret_cast<const unsigned*>(jobs[blockIdx.y]);
    for (int i = threadIdx.x; i < (int)(sizeof(PackJob) / 4); i += blockDim.x) reinterpret_cast<unsigned*>(&j)[i] = src[i];
    __syncthreads();
    if (j.magic != M1_PACK_MAGIC) return;
    const long long first = (long long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long long)gridDim.x * blockDim.x;
    for (int cls = 0; cls < j.p.nclasses; ++cls) {
        if (j.dtype == M1_BF16) pack_class_elems<bf16_t>(j.p, cls, (bf16_t*)j.out, first, stride);
        else pack_class_elems<float>(j.p, cls, (float*)j.out, first, stride);
    }
}
// balanced variant: 1-D grid, block b serves job j with prefix[j] <= b < prefix[j+1] (the caller sizes the block count of a
// job by its weight count: with a fixed 48 blocks per job the 3.5 M-weight res3/res4 kernels set the run time of the launch)
__global__ void __launch_bounds__(256) pack_batch_balanced_kernel(const PackJob* const* __restrict__ jobs, const int* __restrict__ prefix,
                                                                  int njobs) {
    __shared__ PackJob j;
    int lo = 0, hi = njobs - 1;                      // last j with prefix[j] <= blockIdx.x
    const int b = blockIdx.x;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (prefix[mid] <= b) lo = mid; else hi = mid - 1; }
    const unsigned* src = reinterpret_cast<const unsigned*>(jobs[lo]);
    for (int i = threadIdx.x; i < (int)(sizeof(PackJob) / 4); i += blockDim.x) reinterpret_cast<unsigned*>(&j)[i] = src[i];
    __syncthreads();
    if (j.magic != M1_PACK_MAGIC) return;
    const int nb = prefix[lo + 1] - prefix[lo];
    const long long first = (long long)(b - prefix[lo]) * blockDim.x + threadIdx.x, stride = (long long)nb * blockDim.x;
    for (int cls = 0; cls < j.p.nclasses; ++cls) {
        if (j.dtype == M1_BF16) pack_class_elems<bf16_t>(j.p, cls, (bf16_t*)j.out, first, stride);
        else pack_class_elems<float>(j.p, cls, (float*)j.out, first, stride);
    }
}
int m1_pack_batch_internal(const void* const* jobs_dev, const int* prefix_dev, int njobs, int total_blocks, hipStream_t st) {
    if (njobs <= 0) return M1_OK;
    if (prefix_dev && total_blocks > 0)
        hipLaunchKernelGGL(pack_batch_balanced_kernel, dim3((unsigned)total_blocks), dim3(256), 0, st,
                           reinterpret_cast<const PackJob* const*>(jobs_dev), prefix_dev, njobs);
    else
        hipLaunchKernelGGL(pack_batch_kernel, dim3(48, (unsigned)njobs), dim3(256), 0, st, reinterpret_cast<const PackJob* const*>(jobs_dev));
    return m1_check_launch();
}

// out = T(sum_ks slab[ks] + bias) [+ out]   (split-K finish, fixed summation order); the slabs are [voxel][OC], the output may be
// spread over several tensors (MfmaP::outs)
struct FinOut { int nout; void* outs[M1_MAX_SRC]; int outC[M1_MAX_SRC]; int outOff[M1_MAX_SRC + 1]; int outAcc[M1_MAX_SRC]; };
template <typename T>
__global__ void splitk_finish_kernel(const float* __restrict__ acc32, int ksplit, const float* __restrict__ bias, T* __restrict__ out,
                                     long long n, int OC, int accumulate, FinOut fo, const float* __restrict__ bias2, int bias_split) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int oc = (int)(i % OC);
        float v = (bias_split > 0 && oc >= bias_split) ? (bias2 ? bias2[oc - bias_split] : 0.f) : (bias ? bias[oc] : 0.f);
        for (int k = 0; k < ksplit; ++k) v += acc32[(long long)k * n + i];
        T* dst = out + i; int acc = accumulate;
        if (fo.nout > 0) {
            int m = 0;
            while (m + 1 < fo.nout && oc >= fo.outOff[m + 1]) ++m;
            if (!fo.outs[m]) continue;
            dst = (T*)fo.outs[m] + (i / OC) * fo.outC[m] + (oc - fo.outOff[m]); acc = fo.outAcc[m];
        }
        if (acc) v += Act<T>::ld(dst);
        Act<T>::st(dst, v);
    }
}

// The same finish as ONE pass with the InstanceNorm statistics of the (rounded) result: the functor of the generic per-(n,c)
// reduction (reduce.h: a block owns a run of voxels of one sample, lanes along the channels) folds the slabs, adds the bias,
// stores the output and returns {v, v^2} -- the stand-alone statistics pass over the finished tensor (one more read of it,
// one more launch: 28 per C3 step on the split-K layers of the deep levels) disappears.
template <typename T> struct FinishStatsF {
    static constexpr int kVec = 16 / (int)sizeof(T);
    static constexpr int kUnroll = 2;
    const float* acc32; int ksplit; long long slab; const float* bias; const float* bias2; int bias_split;
    T* out; long long V; int OC; FinOut fo;
    // ib_x != nullptr: the sums of the InstanceNorm BACKWARD of the tensor `out` is the gradient of (GatherSpec::ib_*) instead
    const T* ib_x; const float* ib_stats; const float* ib_gamma; const float* ib_beta; float ib_slope;
    __device__ void operator()(int n, long long v, int c, float* acc) const {      // (scalar form of the contract; the launcher gates on OC % kVec == 0)
        const long long row = (long long)n * V + v, i = row * OC + c;
        float a = (bias_split > 0 && c >= bias_split) ? (bias2 ? bias2[c - bias_split] : 0.f) : (bias ? bias[c] : 0.f);
        for (int k = 0; k < ksplit; ++k) a += acc32[(long long)k * slab + i];
        T* dst = out + i;
        if (fo.nout > 0) {
            int m = 0;
            while (m + 1 < fo.nout && c >= fo.outOff[m + 1]) ++m;
            dst = (T*)fo.outs[m] + row * fo.outC[m] + (c - fo.outOff[m]);
        }
        Act<T>::st(dst, a);
        float r = a;
        if constexpr (sizeof(T) == 2) r = bf2f(f2bf(r));
        if (ib_x) {
            const float xh = (Act<T>::ld(ib_x + i) - ib_stats[((long long)n * OC + c) * 2]) * ib_stats[((long long)n * OC + c) * 2 + 1];
            const float dy = r * lrelu_g(ib_gamma[c] * xh + ib_beta[c], ib_slope);
            acc[0] += dy; acc[1] += dy * xh;
        } else { acc[0] += r; acc[1] += r * r; }
    }
    __device__ void vec(int n, long long v, int c0, float (*acc)[kVec]) const {
        const long long row = (long long)n * V + v, i = row * OC + c0;
        float a[kVec];
#pragma unroll
        for (int e = 0; e < kVec; ++e) {
            const int oc = c0 + e;
            a[e] = (bias_split > 0 && oc >= bias_split) ? (bias2 ? bias2[oc - bias_split] : 0.f) : (bias ? bias[oc] : 0.f);
        }
        for (int k = 0; k < ksplit; ++k) {
            const float* sp = acc32 + (long long)k * slab + i;
#pragma unroll
            for (int q = 0; q < kVec; q += 4) {
                const float4 t = *reinterpret_cast<const float4*>(sp + q);
                a[q] += t.x; a[q + 1] += t.y; a[q + 2] += t.z; a[q + 3] += t.w;
            }
        }
        T* dst = out + i;
        if (fo.nout > 0) {                                   // (a 16-byte group never straddles two destination tensors)
            int m = 0;
            while (m + 1 < fo.nout && c0 >= fo.outOff[m + 1]) ++m;
            dst = (T*)fo.outs[m] + row * fo.outC[m] + (c0 - fo.outOff[m]);
        }
        VecIO<T, kVec>::st(dst, a);
        float xv[kVec];
        if (ib_x) VecIO<T, kVec>::ld(ib_x + i, xv);
#pragma unroll
        for (int e = 0; e < kVec; ++e) {                     // the statistics of what was STORED (rounded), as the epilogues compute them
            float r = a[e];
            if constexpr (sizeof(T) == 2) r = bf2f(f2bf(r));
            if (ib_x) {
                const int c = c0 + e;
                const float xh = (xv[e] - ib_stats[((long long)n * OC + c) * 2]) * ib_stats[((long long)n * OC + c) * 2 + 1];
                const float dy = r * lrelu_g(ib_gamma[c] * xh + ib_beta[c], ib_slope);
                acc[0][e] += dy; acc[1][e] += dy * xh;
            } else { acc[0][e] += r; acc[1][e] += r * r; }
        }
    }
};

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static inline int seg_of(int dtype) { return dtype == M1_BF16 ? 8 : 4; }
static inline int pick_bn(int ocn) { return ocn > 64 ? 128 : (ocn > 32 ? 64 : (ocn > 16 ? 32 : 16)); }
struct Plan { int BN, BM, ksplit; };
// 128-row tiles unless that leaves the 256 CUs short of work (res2: 250 tiles, res3/res4: M = 4,000 / 500 voxels)
// -> 64 rows, and if still short, split K (taps x channels, up to 13,824 deep there) over blockIdx.y.
// (Measured: keeping 128-row tiles and splitting K harder instead is slower -- the fp32 atomics cost more than the
// saved L2 -> LDS traffic.)
// bn160: the conv1 || conv4 pair forward (32 + 128 or 64 + 256 output columns): 160-column tiles (5 MFMA column tiles per wave)
// instead of 128-column tiles of which the last is 75 / 50 % empty
static inline bool want_bn160(const GatherSpec& g) {
    int en = M1_CFG("M1_BN160", 1);
    return en && g.dtype == M1_BF16 && g.w2 && g.oc_split > 0 && g.mode == 0 && g.OC % 160 == 0 && g.OC <= 320;
}
static inline Plan make_plan(int ocn, long long maxM, int ncls, int min_nchunks, bool bn160 = false) {
    Plan pl; pl.BN = bn160 ? 160 : pick_bn(ocn);
    const int ntile = (ocn + pl.BN - 1) / pl.BN;
    pl.BM = (bn160 || cdiv_ll(maxM, 128) * ncls * ntile >= 256) ? 128 : 64;      // (the 160-column tile exists with 128 rows only)
    const long long blocks = cdiv_ll(maxM, pl.BM) * ncls * ntile;
    pl.ksplit = 1;
    {   // very deep contractions (>= 200 chunks: the dense-skip concats of the full model) with a 128-wide oc tile and only
        // ~250 row tiles: 128x128 tiles (64x64 per wave: fewer fragment reads and weight re-reads per MFMA) and slab
        // split-K to get the blocks back (-19 % on the 512->128-channel res2 layer).  M1_PLAN128=<target blocks>, 0 = off
        int t128 = M1_CFG("M1_PLAN128", 768);
        const long long b128 = cdiv_ll(maxM, 128) * ncls * ntile;
        if (t128 > 0 && pl.BN == 128 && b128 >= 100 && b128 < t128 && min_nchunks >= 200) {
            pl.BM = 128;
            long long want = cdiv_ll(t128, b128), cap = min_nchunks / 64;
            pl.ksplit = (int)(want < cap ? want : cap);
            if (pl.ksplit < 1) pl.ksplit = 1;
            return pl;
        }
    }
    if (blocks < 256 && min_nchunks >= 16) {
        long long want = cdiv_ll(512, blocks), cap = min_nchunks / 8;
        pl.ksplit = (int)(want < cap ? want : cap);
        if (pl.ksplit < 1) pl.ksplit = 1;
        if (pl.ksplit > 32) pl.ksplit = 32;
    }
    return pl;
}
static inline long long spec_maxM(const GatherSpec& g) {
    if (g.mode == 0) return (long long)g.N * g.OD * g.OH * g.OW;
    return (long long)g.N * ((g.OD + g.sd - 1) / g.sd) * ((g.OH + g.sh - 1) / g.sh) * ((g.OW + g.sw - 1) / g.sw);
}
static inline int spec_ncls(const GatherSpec& g) { return g.mode == 1 ? g.sd * g.sh * g.sw : 1; }

bool m1_mfma_supported(const GatherSpec& g) {
    if (g.kd * g.kh * g.kw > MF_MAX_TAPS) return false;
    if (g.mode == 1 && g.sd * g.sh * g.sw > MF_MAX_CLASSES) return false;
    return true;
}

static inline int spec_spt(const GatherSpec& g, int SEG) {
    int s = 0; for (int i = 0; i < g.nsrc; ++i) s += (g.srcC[i] + SEG - 1) / SEG; return s;
}
static void build_classes(const GatherSpec& g, int CC, int SEG, int OCpad, MfmaP* mp, PackP* pp, long long* total_elems) {
    const int ncls = g.mode == 1 ? g.sd * g.sh * g.sw : 1;
    int t = 0; long long woff = 0;
    for (int c = 0; c < ncls; ++c) {
        const int pwc = g.mode == 1 ? c % g.sw : 0, phc = g.mode == 1 ? (c / g.sw) % g.sh : 0, pdc = g.mode == 1 ? c / (g.sw * g.sh) : 0;
        const int first = t;
        for (int a = 0; a < g.kd; ++a) {
            int od;
            if (g.mode == 0) od = a - 0; else { int v = pdc + g.pd - a; if (v % g.sd) continue; od = v / g.sd; }
            for (int b = 0; b < g.kh; ++b) {
                int oh;
                if (g.mode == 0) oh = b; else { int v = phc + g.ph - b; if (v % g.sh) continue; oh = v / g.sh; }
                for (int e = 0; e < g.kw; ++e) {
                    int ow;
                    if (g.mode == 0) ow = e; else { int v = pwc + g.pw - e; if (v % g.sw) continue; ow = v / g.sw; }
                    if (mp) { mp->tdd[t] = (signed char)od; mp->tdh[t] = (signed char)oh; mp->tdw[t] = (signed char)ow; }
                    if (pp) pp->wtap[t] = (unsigned char)((a * g.kh + b) * g.kw + e);
                    ++t;
                }
            }
        }
        const int nt = t - first;
        const int nseg = nt * spec_spt(g, SEG);
        const int kpad = ((nseg + 3) / 4) * 4 * SEG;
        if (mp) { mp->cls_ntaps[c] = nt; mp->cls_first[c] = first; mp->cls_kpad[c] = kpad; mp->cls_woff[c] = woff; }
        if (pp) { pp->cls_ntaps[c] = nt; pp->cls_first[c] = first; pp->cls_kpad[c] = kpad; pp->cls_woff[c] = woff; }
        woff += (long long)OCpad * kpad;
    }
    if (mp) mp->nclasses = ncls;
    if (pp) pp->nclasses = ncls;
    *total_elems = woff;
}

static inline int min_class_chunks(const GatherSpec& g, int CC, int SEG) {
    // smallest K-chunk count over the parity classes (mode 1) -- the split factor must suit every class
    MfmaP tmp{}; long long tot = 0;
    build_classes(g, CC, SEG, 16, &tmp, nullptr, &tot);
    int mn = 1 << 30;
    for (int c = 0; c < tmp.nclasses; ++c) { int n = tmp.cls_kpad[c] / (4 * SEG); if (n < mn) mn = n; }
    return mn;
}
static inline size_t out_elems(const GatherSpec& g) { return (size_t)g.N * g.OD * g.OH * g.OW * g.OC; }

size_t m1_mfma_ws_bytes(const GatherSpec& g) {
    const int SEG = seg_of(g.dtype);
    int CC = 0; for (int i = 0; i < g.nsrc; ++i) CC += g.srcC[i];
    // (the workspace query does not know whether the call will be the conv1 || conv4 pair, which takes 160-column tiles and may
    //  then split K differently: size for both plans)
    size_t best = 0;
    const bool maybe160 = g.dtype == M1_BF16 && g.mode == 0 && g.OC % 160 == 0 && g.OC <= 320;
    int t3bn = 0, t3ks = 1;
    const bool t3 = m1_ct3_plan(g, &t3bn, &t3ks);
    for (int v = 0; v < 3; ++v) {
        if ((v == 1 && !maybe160) || (v == 2 && !t3)) continue;
        Plan pl = make_plan(g.OC, spec_maxM(g), spec_ncls(g), min_class_chunks(g, CC, SEG), v == 1);
        if (v == 2) { pl.BN = t3bn; pl.BM = 256; pl.ksplit = t3ks; }
        const int BN = pl.BN, OCpad = (g.OC + BN - 1) / BN * BN;
        long long tot = 0;
        build_classes(g, CC, SEG, OCpad, nullptr, nullptr, &tot);
        size_t bytes = ((size_t)tot * (g.dtype == M1_BF16 ? 2 : 4) + 255) / 256 * 256;
        if (pl.ksplit > 1) bytes += out_elems(g) * sizeof(float) * pl.ksplit;
        if (bytes > best) best = bytes;
    }
    return best + 256 + M1_PACK_JOB_BYTES;
}

template <typename T, int BM, int BN, int WM, int WN, int KC>
static int launch_cfg_kc(const MfmaP& mp, long long maxM, int OCpad, hipStream_t st) {
    dim3 grid((unsigned)(cdiv_ll(cdiv_ll(maxM, BM), 8) * 8), mp.nclasses * mp.ksplit, OCpad / BN);
    const size_t smem = mfma_smem_bytes<T, BM, BN, KC, WM * WN * 64>();
    auto kern = mp.aligned ? conv_mfma_kernel<T, BM, BN, WM, WN, KC, true> : conv_mfma_kernel<T, BM, BN, WM, WN, KC, false>;
    static bool attr_set = false;     // per instantiation
    if (smem > 48 * 1024 && !attr_set) {
        if (hipFuncSetAttribute((const void*)conv_mfma_kernel<T, BM, BN, WM, WN, KC, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return M1_ERR_LAUNCH;
        if (hipFuncSetAttribute((const void*)conv_mfma_kernel<T, BM, BN, WM, WN, KC, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return M1_ERR_LAUNCH;
        attr_set = true;
    }
    m1_note_kernel("conv_mfma:%dx%d:w%d:ks%d", BM, BN, WM * WN, mp.ksplit);
    hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), smem, st, mp);
    return m1_check_launch();
}

// K groups inside the block (conv_mfma_kernel<.., KG>): the small 64-row tiles of the deep levels, bf16, LDS-DMA path, two chunks per stage
template <typename T, int BM, int BN, int WM, int WN, int KG>
static int launch_cfg_kg(const MfmaP& mp, long long maxM, int OCpad, hipStream_t st) {
    constexpr size_t smem = mfma_smem_bytes<T, BM, BN, 2, WM * WN * 64, KG>();
    if constexpr (sizeof(T) != 2 || smem > 160 * 1024 - 256 || WM * WN * 64 * KG > 1024) return M1_ERR_UNSUPPORTED;
    else {
        dim3 grid((unsigned)(cdiv_ll(cdiv_ll(maxM, BM), 8) * 8), mp.nclasses * mp.ksplit, OCpad / BN);
        static bool attr_set = false;     // per instantiation
        if (!attr_set) {
            if (hipFuncSetAttribute((const void*)conv_mfma_kernel<T, BM, BN, WM, WN, 2, true, KG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return M1_ERR_LAUNCH;
            attr_set = true;
        }
        m1_note_kernel("conv_mfma:%dx%d:w%d:ks%d:kg%d", BM, BN, WM * WN, mp.ksplit, KG);
        hipLaunchKernelGGL((conv_mfma_kernel<T, BM, BN, WM, WN, 2, true, KG>), grid, dim3(WM * WN * 64 * KG), smem, st, mp);
        return m1_check_launch();
    }
}
// the K groups a tile shape can carry in 160 KB of LDS / 1,024 threads (0: none)
static inline int mfma_kg_max(int BM, int BN, int nthr) {
    const int group = 2 * 2 * (BM + BN) * 64;             // two stages of two chunks
    int k = (150 * 1024) / group; if (k > 1024 / nthr) k = 1024 / nthr; if (k > 4) k = 4;
    return k >= 2 ? k : 0;
}

// two 64-byte K-chunks per pipeline stage (one barrier per 64 bf16 / 32 fp32 of K) once the K loop is long enough
template <typename T, int BM, int BN, int WM, int WN>
static int launch_cfg(const MfmaP& mp, long long maxM, int OCpad, hipStream_t st, int kg = 1) {
    if (kg == 2) return launch_cfg_kg<T, BM, BN, WM, WN, 2>(mp, maxM, OCpad, st);
    if (kg == 3) return launch_cfg_kg<T, BM, BN, WM, WN, 3>(mp, maxM, OCpad, st);
    if (kg == 4) return launch_cfg_kg<T, BM, BN, WM, WN, 4>(mp, maxM, OCpad, st);
    int minchunks = 1 << 30;
    for (int c = 0; c < mp.nclasses; ++c) { const int n = mp.cls_kpad[c] / (4 * MT<T>::SEG) / mp.ksplit; if (n < minchunks) minchunks = n; }
    return minchunks >= 6 ? launch_cfg_kc<T, BM, BN, WM, WN, 2>(mp, maxM, OCpad, st)
                          : launch_cfg_kc<T, BM, BN, WM, WN, 1>(mp, maxM, OCpad, st);
}

template <typename T>
static int run_mfma(const GatherSpec& g, void* ws, int ws_packed, hipStream_t st) {
    constexpr int SEG = MT<T>::SEG;
    MfmaP mp{}; PackP pp{};
    int CC = 0;
    for (int i = 0; i < M1_MAX_SRC; ++i) {
        mp.src[i] = i < g.nsrc ? g.src[i] : nullptr; mp.srcC[i] = i < g.nsrc ? g.srcC[i] : (1 << 30);
        mp.srcSeg[i] = i < g.nsrc ? (g.srcC[i] + SEG - 1) / SEG : (1 << 30);
        pp.srcC[i] = i < g.nsrc ? g.srcC[i] : 0; pp.srcSeg[i] = i < g.nsrc ? (g.srcC[i] + SEG - 1) / SEG : 0;
        if (i < g.nsrc) CC += g.srcC[i];
    }
    mp.nsrc = g.nsrc; mp.CC = CC; mp.spt = spec_spt(g, SEG);
    pp.nsrc = g.nsrc; pp.spt = mp.spt; pp.SEG = SEG; mp.ID = g.ID; mp.IH = g.IH; mp.IW = g.IW; mp.out = g.out; mp.OC = g.OC; mp.OCn = g.OC;
    void* panel = reinterpret_cast<unsigned char*>(ws) + M1_PACK_JOB_BYTES;    // [job record][panels][split-K accumulator]
    mp.OD = g.OD; mp.OH = g.OH; mp.OW = g.OW; mp.N = g.N; mp.wp = panel; mp.bias = g.bias; mp.mode = g.mode;
    mp.sd = g.sd; mp.sh = g.sh; mp.sw = g.sw; mp.pd = g.pd; mp.ph = g.ph; mp.pw = g.pw; mp.accumulate = g.accumulate;
    mp.bias2 = g.bias2; mp.bias_split = g.oc_split;
    mp.nout = g.nout; mp.outOff[0] = 0;
    for (int i = 0; i < g.nout; ++i) { mp.outs[i] = g.outs[i]; mp.outC[i] = g.outC[i]; mp.outAcc[i] = g.outAcc[i]; mp.outOff[i + 1] = mp.outOff[i] + g.outC[i]; }
    Plan pl = make_plan(g.OC, spec_maxM(g), spec_ncls(g), min_class_chunks(g, CC, SEG), want_bn160(g));
    // stride-1 3x3x3 / 1x3x3 matrix-core layers: the staged-run kernel on 32x32x16 MFMAs (conv_t3.hip) with its own tiling
    int t3bn = 0, t3ks = 1;
    const bool t3 = sizeof(T) == 2 && m1_ct3_plan(g, &t3bn, &t3ks);
    if (t3) { pl.BN = t3bn; pl.BM = 256; pl.ksplit = t3ks; }
    // K groups inside the block instead of split-K across blocks (conv_mfma_kernel<.., KG>): the 64-row tiles of the deep levels whose
    // split the plan put at 2..4 (one kind of tile per launch: every class then has >= 8 chunks per group).  M1_MFMA_KG=0: split-K slabs
    int kgroups = 1;
    if (sizeof(T) == 2 && !t3 && pl.BM == 64 && (pl.BN == 64 || pl.BN == 128) && pl.ksplit >= 2 && M1_CFG("M1_MFMA_KG", 1)) {
        bool al = true;
        for (int i = 0; i < g.nsrc; ++i) al = al && g.srcC[i] % (4 * SEG) == 0;
        const int kmax = mfma_kg_max(64, pl.BN, 256);
        // (only where the unsplit tiles already cover the chip: the groups of a block share its CU, split-K blocks spread over CUs --
        //  on the 32..128 tiles of the (5,10,10) level the in-block form was 40-60 % slower)
        const long long tiles64 = cdiv_ll(spec_maxM(g), 64) * spec_ncls(g) * ((g.OC + pl.BN - 1) / pl.BN);
        if (al && kmax >= 2 && pl.ksplit <= 4 && tiles64 >= 200) { kgroups = pl.ksplit < kmax ? pl.ksplit : kmax; pl.ksplit = 1; }
    }
    const int BN = pl.BN, OCpad = (g.OC + BN - 1) / BN * BN;
    long long tot = 0;
    build_classes(g, CC, SEG, OCpad, &mp, &pp, &tot);
    mp.ksplit = pl.ksplit; mp.acc32 = nullptr; mp.aligned = 1; mp.stat_partial = nullptr; mp.stat_tiles = 0;
    const long long Vout = (long long)g.OD * g.OH * g.OW;
    // M1_CONV8 (default 1): 128x128 tiles on 8 waves (64x32 per wave) for the deep layers -- the same waves per CU as two 4-wave
    // 64x128 / 128x128 blocks with half the weight-tile re-reads.  Round 1 (2-volume launches): -8 % on the 512->128 forward, +4 % on
    // its data gradient, neutral end to end; with the stacked passes (4 volumes per launch): 647 -> 624 us forward, 613 -> 557 us
    // data gradient, -1.7 % per C3 step, C2 neutral.  256x128 tiles on 8 waves (one block per CU) were slower (702 / 678 us).
    int c8 = M1_CFG("M1_CONV8", 1);
    const bool use8 = !t3 && kgroups == 1 && c8 && BN == 128 && (pl.ksplit == 1 || (c8 >= 2 && pl.BM == 128)) &&
                      cdiv_ll(spec_maxM(g), 128) * spec_ncls(g) * (OCpad / 128) * pl.ksplit >= 160;
    const int bm_eff = t3 ? 256 : (use8 ? 128 : pl.BM);
    // (the staged-run kernel tiles every sample on its own: its tiles never straddle samples)
    // (the implicit-GEMM kernel tiles the samples one by one when V is no multiple of the tile -- MfmaP::tps, round 6: the deep levels'
    //  4,000 / 500 voxels per sample; the halo / pointwise kernels have their own rows and ignore it)
    const int tiles_ps = t3 ? m1_ct3_tiles_per_sample(g.OD, g.OH, g.OW) : (int)cdiv_ll(Vout, bm_eff);      // epilogue partial rows per sample
    const bool per_sample = !t3 && g.N > 1 && Vout % bm_eff != 0 && spec_ncls(g) == 1 && pl.ksplit == 1 && M1_CFG("M1_MFMA_TPS", 1) &&
                            (long long)tiles_ps * bm_eff * g.N < (1ll << 31);
    bool fuse_stats = g.stats_out && g.stats_ws && g.mode == 0 && pl.ksplit == 1 && (g.N == 1 || Vout % bm_eff == 0 || t3 || per_sample) &&
                      !g.accumulate;           // (this kernel's statistics come from its own tile, before the add)
    if (fuse_stats) { mp.stat_partial = g.stats_ws; mp.stat_tiles = tiles_ps; }
    // InstanceNorm-backward sums from the epilogue of a data gradient (GatherSpec::ib_*): one output tensor, one parity class
    // (rows run sample-major like a forward conv's), tiles that do not straddle samples
    const bool ib_want = g.ib_x && g.ib_partial && g.ib_nparts && !g.stats_out && !g.accumulate && g.nout <= 1 && spec_ncls(g) == 1 &&
                         (g.nout == 0 || (g.outs[0] && !g.outAcc[0] && g.outC[0] == g.OC));
    if (g.ib_nparts) *g.ib_nparts = 0;
    bool ib_epi = ib_want && pl.ksplit == 1 && (g.N == 1 || Vout % bm_eff == 0 || t3 || per_sample) && tiles_ps <= g.ib_cap;
    mp.tps = (per_sample && (fuse_stats || ib_epi)) ? tiles_ps : 0;
    if (ib_epi) {
        mp.stat_partial = g.ib_partial; mp.stat_tiles = tiles_ps;
        mp.ib_x = g.ib_x; mp.ib_stats = g.ib_stats; mp.ib_gamma = g.ib_gamma; mp.ib_beta = g.ib_beta; mp.ib_slope = g.ib_slope;
    }
    for (int i = 0; i < g.nsrc; ++i) if (g.srcC[i] % (4 * SEG)) mp.aligned = 0;
    for (int i = 0; i < g.nsrc; ++i)          // the LDS-DMA loader addresses a member with 31-bit byte offsets
        if ((long long)g.N * g.ID * g.IH * g.IW * g.srcC[i] * (long long)sizeof(T) >= (1ll << 31) - 4096) mp.aligned = 0;
    if (tot * (long long)sizeof(T) >= (1ll << 31) - 4096) mp.aligned = 0;
    if (pl.ksplit > 1) {
        const size_t wbytes = ((size_t)tot * sizeof(T) + 255) / 256 * 256;
        mp.acc32 = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(panel) + wbytes);
        mp.slab_elems = (long long)out_elems(g);
    }
    // wide, shallow bf16 layers: the halo-tile kernel (conv_halo.hip) stages every input voxel once per tile, not per tap
    const long long maxM = mp.tps > 0 ? (long long)mp.tps * g.N * bm_eff : spec_maxM(g);      // (conv_mfma launches size their grid by it)
    bool halo = false;
    if constexpr (sizeof(T) == 2) {
        int hen = M1_CFG("M1_HALO", 1);
        halo = !t3 && hen && (maxM >= 32768 || hen == 2) && m1_halo_conv_supported(mp, OCpad);      // (M1_HALO=2: no size floor, tests)
    }
    // K order of the panel and of the LDS-DMA gather: chunk-major ([64-byte chunk][tap]) for multi-tap problems on the
    // implicit-GEMM kernel (see advance_chunk); the halo kernel and single-tap problems keep [tap][channel]
    { int ko = M1_CFG("M1_KORDER", 2);     // 2: pairs of chunks when every member allows
      int maxtaps = 0; for (int c = 0; c < mp.nclasses; ++c) maxtaps = mp.cls_ntaps[c] > maxtaps ? mp.cls_ntaps[c] : maxtaps;
      mp.korder = (ko && mp.aligned && !halo && maxtaps > 1) ? 1 : 0;
      if (mp.korder && ko == 2) { bool pair = true; for (int i = 0; i < g.nsrc; ++i) pair &= g.srcC[i] % (8 * SEG) == 0; if (pair) mp.korder = 2; }
      if (t3) mp.korder = 1;                              // (the staged-run kernel reads [chunk][tap] panels)
      pp.korder = mp.korder; }
    pp.w = g.w; pp.wST = g.wST; pp.wSC = g.wSC; pp.wSO = g.wSO; pp.oc_off = g.oc_off; pp.cc_off = g.cc_off; pp.OCn = g.OC; pp.OCpad = OCpad; pp.CC = CC;
    pp.w2 = g.w2; pp.w2ST = g.w2ST; pp.w2SC = g.w2SC; pp.w2SO = g.w2SO; pp.oc_split = g.oc_split; pp.c_split = g.c_split;
    int maxk = 0; for (int c = 0; c < pp.nclasses; ++c) maxk = pp.cls_kpad[c] > maxk ? pp.cls_kpad[c] : maxk;
    long long pblocks = cdiv_ll((long long)OCpad * maxk, 256); if (pblocks > 2048) pblocks = 2048; if (pblocks < 1) pblocks = 1;
    int rc = M1_OK;
    if (!ws_packed) {          // the caller may keep the panel of an unchanged weight across calls (2+2 core passes per step)
        hipLaunchKernelGGL(pack_weights_kernel<T>, dim3((unsigned)pblocks, pp.nclasses), dim3(256), 0, st, pp, (T*)panel, reinterpret_cast<PackJob*>(ws));
        rc = m1_check_launch(); if (rc) return rc;
    }

    const bool small = pl.BM == 64;
    int rc2;
    if (halo) {
        // InstanceNorm-backward sums from the halo kernel's register epilogue (round 4): one partial row per (sample, block row), the
        // same rows its statistics use; whole 4-channel groups only
        if (ib_want && !ib_epi && pl.ksplit == 1 && g.OC % 4 == 0) {
            mp.ib_x = g.ib_x; mp.ib_stats = g.ib_stats; mp.ib_gamma = g.ib_gamma; mp.ib_beta = g.ib_beta; mp.ib_slope = g.ib_slope;
            ib_epi = true;
        }
        const int tps = m1_halo_conv_stat_parts(mp, OCpad);
        if (ib_epi && g.OC % 4 == 0 && tps > 0 && tps <= g.ib_cap) { mp.stat_partial = g.ib_partial; mp.stat_tiles = tps; fuse_stats = false; }
        else {
            if (ib_epi) { ib_epi = false; mp.ib_x = nullptr; }
            if (g.stats_out && g.stats_ws && g.mode == 0 && tps > 0 && tps <= (Vout + 63) / 64) {
                mp.stat_partial = g.stats_ws; mp.stat_tiles = tps; fuse_stats = true;
            } else { mp.stat_partial = nullptr; mp.stat_tiles = 0; fuse_stats = false; }
        }
    }
    // pointwise layers: the streaming kernel (conv_pw.hip) -- no operand tiles, no barriers, epilogue in registers
    bool pw = false; int pwBN = 0;
    if constexpr (sizeof(T) == 2) {
        pwBN = BN > 32 ? 32 : BN;
        if (!halo && !t3 && m1_pw_conv_supported(mp, OCpad, pwBN)) {
            int parts = m1_pw_conv_stat_parts(mp, OCpad, pwBN);
            const int cap = (int)m1_stats_rows_cap(Vout) / 4 * 4;     // what the statistics workspace holds per sample
            if (parts > cap) parts = cap;
            // InstanceNorm-backward sums from the streaming kernel's register epilogue (round 4): one partial row per (sample, wave)
            // (its rows go to the caller's partial buffer: as many waves as THAT holds)
            const int parts_ib = parts < g.ib_cap / 4 * 4 ? parts : g.ib_cap / 4 * 4;
            const bool ib_pw = ib_want && pl.ksplit == 1 && g.OC % 4 == 0 && Vout % 32 == 0 && parts_ib >= 4;
            if (ib_pw) {
                mp.ib_x = g.ib_x; mp.ib_stats = g.ib_stats; mp.ib_gamma = g.ib_gamma; mp.ib_beta = g.ib_beta; mp.ib_slope = g.ib_slope;
                mp.stat_partial = g.ib_partial; mp.stat_tiles = parts_ib; ib_epi = true; fuse_stats = false;
            } else {
                if (ib_epi) { ib_epi = false; mp.ib_x = nullptr; }
                if (g.stats_out && g.stats_ws && g.mode == 0 && !g.accumulate && parts >= 4) {
                    mp.stat_partial = g.stats_ws; mp.stat_tiles = parts; fuse_stats = true;
                } else { mp.stat_partial = nullptr; mp.stat_tiles = 0; fuse_stats = false; }
            }
            pw = true;
        }
    }
    { int lg = M1_CFG("M1_MFMA_LOG", 0);
      if (lg) fprintf(stderr, "mfma: mode %d N%d out %dx%dx%d CC %d OC %d k%d taps s%d%d%d nsrc %d -> %s BM %d BN %d ksplit %d korder %d stats %d\n", g.mode, g.N, g.OD, g.OH, g.OW, CC, g.OC,
                      g.kd * g.kh * g.kw, g.sd, g.sh, g.sw, g.nsrc, t3 ? "t3" : (halo ? "halo" : (pw ? "pw" : "mfma")), bm_eff, BN, pl.ksplit, mp.korder, fuse_stats ? 1 : 0); }
    if (t3) rc2 = m1_ct3_conv(mp, BN, OCpad, st);
    else if (halo) rc2 = m1_halo_conv(mp, OCpad, st);
    else if (pw) rc2 = m1_pw_conv(mp, OCpad, pwBN, st);
    else
    switch (BN) {
        case 128: rc2 = use8 ? launch_cfg<T, 128, 128, 2, 4>(mp, maxM, OCpad, st)
                             : (small ? launch_cfg<T, 64, 128, 1, 4>(mp, maxM, OCpad, st, mp.aligned ? kgroups : 1) : launch_cfg<T, 128, 128, 2, 2>(mp, maxM, OCpad, st)); break;
        case 160: if constexpr (sizeof(T) == 2) rc2 = launch_cfg<T, 128, 160, 2, 2>(mp, maxM, OCpad, st); else rc2 = M1_ERR_UNSUPPORTED; break;
        case 64: {
            // 64 / 32 columns on 8 waves of 32x32 / 32x16 instead of 4 waves of 32x64 / 32x32 (M1_F32_W8: bit 0 fp32, bit 1 bf16).  fp32: an
            // MFMA is 1/16 of a bf16 one per cycle, the LDS has room for the extra fragment reads and the extra waves hide the operand
            // latency (54 % MFMA-busy with 4 waves; C5 40.18 -> 39.95 ms).  bf16: these tiles are latency-bound (10-15 % MFMA-busy): C2
            // 7.86 -> 7.59 ms, C3 neutral
            int w8 = M1_CFG("M1_F32_W8", 3);
            if (((sizeof(T) == 4 && (w8 & 1)) || (sizeof(T) == 2 && (w8 & 2))) && !small) rc2 = launch_cfg<T, 128, 64, 4, 2>(mp, maxM, OCpad, st);
            else rc2 = small ? launch_cfg<T, 64, 64, 2, 2>(mp, maxM, OCpad, st, mp.aligned ? kgroups : 1) : launch_cfg<T, 128, 64, 4, 1>(mp, maxM, OCpad, st);
            break; }
        case 32: {
            int w8 = M1_CFG("M1_F32_W8", 3);
            if (((sizeof(T) == 4 && (w8 & 1)) || (sizeof(T) == 2 && (w8 & 2))) && !small) rc2 = launch_cfg<T, 128, 32, 4, 2>(mp, maxM, OCpad, st);
            else rc2 = small ? launch_cfg<T, 64, 32, 4, 1>(mp, maxM, OCpad, st) : launch_cfg<T, 128, 32, 4, 1>(mp, maxM, OCpad, st);
            break; }
        default: rc2 = small ? launch_cfg<T, 64, 16, 4, 1>(mp, maxM, OCpad, st) : launch_cfg<T, 128, 16, 4, 1>(mp, maxM, OCpad, st); break;
    }
    if (rc2) return rc2;
    if (ib_epi) *g.ib_nparts = mp.stat_tiles;
    // (two output tensors -- the conv1 || conv4 pair: statistics per tensor, stats_out for [0, oc_split), stats_out2 for the rest)
    auto stats_fallback = [&]() -> int {
        if (g.nout == 2 && g.stats_out2) {
            int r = m1_stats_internal(g.outs[0], g.N, Vout, g.outC[0], g.dtype, g.stats_eps, g.stats_out, g.stats_ws, st); if (r) return r;
            return m1_stats_internal(g.outs[1], g.N, Vout, g.outC[1], g.dtype, g.stats_eps, g.stats_out2, g.stats_ws, st);
        }
        return m1_stats_internal(g.out, g.N, Vout, g.OC, g.dtype, g.stats_eps, g.stats_out, g.stats_ws, st);
    };
    if (g.stats_out) {
        if (fuse_stats) {           // partial layout [N][tiles][OC][2] is exactly what the generic finalize folds (fp64, per wave)
            rc2 = m1_reduce_finalize_launch<2>(g.stats_ws, g.N, g.OC, mp.stat_tiles, g.stats_out, Vout, g.stats_eps, st, 0,
                                               g.stats_out2, g.stats_out2 ? g.oc_split : 0);
            if (rc2) return rc2;
        } else if (pl.ksplit <= 1) {
            return stats_fallback();
        }
    }
    if (pl.ksplit <= 1) return rc2;
    const long long ne = (long long)out_elems(g);
    long long fb = cdiv_ll(ne, 256); if (fb > 2048) fb = 2048;
    FinOut fo{}; fo.nout = mp.nout;
    for (int i = 0; i < mp.nout; ++i) { fo.outs[i] = mp.outs[i]; fo.outC[i] = mp.outC[i]; fo.outOff[i] = mp.outOff[i]; fo.outAcc[i] = mp.outAcc[i]; }
    fo.outOff[mp.nout] = mp.outOff[mp.nout];
    {   // finish + InstanceNorm statistics in one pass (see FinishStatsF)
        int fs = M1_CFG("M1_FINISH_STATS", 1);
        constexpr int VEC = 16 / (int)sizeof(T);
        bool ok = fs && g.stats_out && g.stats_ws && g.mode == 0 && !g.accumulate && g.OC % VEC == 0 && mp.nout <= 2;
        for (int i = 0; i < mp.nout; ++i) ok = ok && mp.outs[i] && !mp.outAcc[i] && mp.outC[i] % VEC == 0;
        if (mp.nout == 2 && !g.stats_out2) ok = false;          // (two output tensors: the conv1 || conv4 pair with its two statistics)
        constexpr int VECI = 16 / (int)sizeof(T);
        if (ib_want && fs && g.OC % VECI == 0 && m1_red_nchunks(Vout, g.OC, g.N) <= g.ib_cap) {      // the same pass with the InstanceNorm-backward sums (one launch: no finalize here)
            FinishStatsF<T> f{mp.acc32, pl.ksplit, ne, g.bias, g.bias2, g.oc_split, (T*)g.out, Vout, g.OC, fo,
                              (const T*)g.ib_x, g.ib_stats, g.ib_gamma, g.ib_beta, g.ib_slope};
            rc2 = m1_reduce_nc_launch<2>(f, g.N, Vout, g.OC, g.ib_partial, st); if (rc2) return rc2;
            *g.ib_nparts = m1_red_nchunks(Vout, g.OC, g.N);
            return M1_OK;
        }
        if (ok) {
            FinishStatsF<T> f{mp.acc32, pl.ksplit, ne, g.bias, g.bias2, g.oc_split, (T*)g.out, Vout, g.OC, fo, nullptr, nullptr, nullptr, nullptr, 0.f};
            rc2 = m1_reduce_nc_launch<2>(f, g.N, Vout, g.OC, g.stats_ws, st); if (rc2) return rc2;
            return m1_reduce_finalize_launch<2>(g.stats_ws, g.N, g.OC, m1_red_nchunks(Vout, g.OC, g.N), g.stats_out, Vout, g.stats_eps, st, 0,
                                                g.stats_out2, g.stats_out2 ? g.oc_split : 0);
        }
    }
    hipLaunchKernelGGL(splitk_finish_kernel<T>, dim3((unsigned)fb), dim3(256), 0, st, mp.acc32, pl.ksplit, g.bias, (T*)g.out, ne, g.OC, g.accumulate, fo, g.bias2, g.oc_split);
    rc2 = m1_check_launch(); if (rc2) return rc2;
    if (g.stats_out) return stats_fallback();
    return M1_OK;
}

int m1_mfma_gather(const GatherSpec& g, void* ws, int ws_packed, hipStream_t st) {
    if (!ws) return M1_ERR_WORKSPACE;
    { int rc = M1_OK; if (m1_thin_conv_try(g, st, &rc)) return rc; }
    return g.dtype == M1_BF16 ? run_mfma<bf16_t>(g, ws, ws_packed, st) : run_mfma<float>(g, ws, ws_packed, st);
}
