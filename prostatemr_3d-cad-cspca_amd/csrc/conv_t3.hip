// conv_t3.hip -- stride-1 Conv3D forward / data gradient (3x3x3 and 1x3x3, TF 'same') for the matrix-core layers, bf16, on
// v_mfma_f32_32x32x16_bf16 with the input staged ONCE per output tile:
//
//   D[v][oc] = sum_{dd,dh,dw in -1..1} sum_c X[v + (dd H + dh) W + dw][c] * Wp[oc][tap(dd,dh,dw)][c]       v = flattened (d,h,w)
//
// conv_mfma.hip gathers the A operand from L2 once per tap (27 x per 128-column tile) on 16x16x32 MFMAs with 0.75 KB of LDS fragment
// reads per MFMA -- the LDS is as busy as the matrix cores and the kernel sits at ~24 % MFMA-busy.  Here
// * an output tile is a RUN of 256 consecutive voxels of one sample in flattened (d,h,w) order; for one 32-channel chunk of the
//   concat and one kd slice the run EXTENDED BY W + 1 voxels on both sides is staged once by LDS-DMA (64-byte rows): the nine (dh,dw)
//   taps of the slice are row offsets (dh W + dw) into it -- the input crosses L2 -> LDS 3 x (1 + 2 (W+1) / 256) times instead of
//   27 x, whatever the row length (no 2-D tile geometry: W = 40, 20, 10 run the same code).  A neighbour in flattened order that is
//   NOT a neighbour in the volume (h or w wrapped) must read zeros: every lane keeps the LDS address of its voxel under each of the
//   nine (dh,dw) shifts (18 registers, computed once), and an invalid (voxel, tap) pair points at a ZERO ROW of the staged image (the
//   first row behind the run, which the DMA fills with zeros like every row outside the sample) -- TF-SAME padding costs no
//   instruction in the loop.  (A first version masked the fragments in registers: 48 v_and per kh row and wave, 7 VALU instructions
//   per MFMA, the VALU pipe as busy as the matrix pipe; a padded flattening with pad columns / rows pushed the 500-tile res2 launches
//   over two rounds of 256 blocks.)  d wraps fall outside the sample and are zero-filled by the buffer range check;
// * weights stream per kh row: the three (dw) taps of one (chunk, dd, dh) are a stage of a 3-deep ring, read from the same packed
//   panel conv_mfma uses ([oc][chunk][tap][32 channels], korder 1);
// * 8 waves = 4 (voxel quarters of 64) x 2 (column halves): a wave accumulates a 64 x (32 NJ) tile in 2 x NJ 32x32 tiles -- one LDS
//   fragment read per MFMA of twice the work of a 16x16x32 one (conv_mfma: 0.75 KB per half-size MFMA); the 160-column tile of the
//   conv1 || conv4 pair forward gives the two column waves 3 + 2 tiles, paired on the SIMDs so that every SIMD carries 5;
// * one barrier per kh row (24 / 30 / 36 MFMAs per wave): counted vmcnt, DMA issued from inline asm two rows ahead (weights) and
//   one kd slice ahead (input), fragment reads compiler-visible;
// * XOR swizzle on the DMA source side (LDS slot s of row r holds 16-byte piece s ^ ((r >> 2) & 3)): 16 consecutive rows of a
//   ds_read_b128 lane group cover all 64 banks;
// * epilogue as conv_mfma's: bias, rounding, LDS tile, InstanceNorm statistics (or InstanceNorm-backward sums) of the rounded
//   values per tile, 16-byte stores spread over the destination tensors; split-K over the channel chunks into fp32 slabs.
#include "common.h"
#include "gather.h"
#include "conv_mfma.h"
#include "conv_t3.h"
#include <type_traits>

typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef int i32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

#define CT3_BM 256
#define CT3_THREADS 512
#define CT3_NAS 2            // A-piece slots per wave in each of the two issue phases of a kd slice: 2 * 2 * 8 KB >= (256 + 2 W + 2) * 64 B

__device__ __forceinline__ void ct3_dma(i32x4_t rs, unsigned lds, unsigned voff, unsigned soff) {
    // (M0 is written here without a clobber: "m0" is a reserved register to hipcc -- it warns on the clobber -- and these kernels contain no
    // compiler-generated M0 use that a stale value could reach; tools/isa_async_check.py / tests/test_build_props.py verify that on the ISA)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(lds), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
template <int N> __device__ __forceinline__ void ct3_vmwait() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
__device__ __forceinline__ u32x4_t ct3_lds(unsigned addr) {                 // ds_read_b128 at a 32-bit LDS address (compiler-visible)
    return *reinterpret_cast<const __attribute__((address_space(3))) u32x4_t*>((unsigned long long)addr);
}

struct Ct3P {
    int V, HW, tps;          // voxels per sample, per (d) plane; tiles per sample
    int halo, arows, nA;     // W + 1; rows of a staged run (256 + 2 halo); 1 KB pieces (16 rows) of run + zero row
    int KD;                  // kd slices (3 or 1)
    int nchunks;             // 32-channel chunks of the concat
    int cps;                 // chunks per K split
    int OCpad;
    int smax;                // stages of a block at most (cps * KD): the descriptor tables are sized by it
};

// compile-time loop: f(std::integral_constant<int, I>{}) for I = 0 .. N - 1 (immediates of the inline-asm reads must be constants)
template <int I, int N, typename F> __device__ __forceinline__ void ct3_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); ct3_for<I + 1, N>(f); }
}
template <int N> __device__ __forceinline__ void ct3_lgkmwait() { asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(N) : "memory"); }
// fragment read, issued where it stands (volatile): the compiler would otherwise sink every read to its MFMA and wait for it there
template <int IMM> __device__ __forceinline__ void ct3_rd(u32x4_t& v, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(IMM));
}

#define CT3_ASTRIDE (22 * 1024)      // bytes of one A buffer (compile time: the buffer parity is an immediate of the reads): W <= 44

template <int NJ0, int NJ1>
__global__ void __launch_bounds__(CT3_THREADS, 2) conv_t3_kernel(MfmaP p, Ct3P q) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BN = 32 * (NJ0 + NJ1), NJ = NJ0 > NJ1 ? NJ0 : NJ1;
    constexpr int BT = BN * 64;                                  // bytes of one tap's weight tile (BN rows of 64 bytes)
    constexpr int BSTAGE = 3 * BT;                               // one kh row: 3 taps
    constexpr int NB = 3 * BN / 16;                              // its 1 KB pieces
    constexpr int NBS = (NB + 7) / 8;                            // piece slots per wave
    constexpr int NR = 2 + NJ;                                   // fragment reads per k-step
    constexpr int RING = BN >= 256 ? 2 : 3;                      // weight stages in LDS (256 columns: 48 KB each, two fit)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* const s_tap = reinterpret_cast<int*>(smem);             // [27] panel tap index of (dd,dh,dw), -1 = no such tap
    int* const outrow = reinterpret_cast<int*>(smem + 128);      // [256] output voxel (global row) or -1
    unsigned char* const A_s = smem + 128 + CT3_BM * 4;
    const unsigned lds0 = (unsigned)(unsigned long long)(lptr_t)smem;
    const unsigned ldsA = lds0 + 128 + CT3_BM * 4, ldsB = ldsA + 2 * CT3_ASTRIDE, trash = ldsB + RING * BSTAGE;
    // descriptor tables behind the scratch KB: per stage (chunk, kd slice) the A resource + run shift, per interval the 3 panel offsets
    int4* const tabA = reinterpret_cast<int4*>(smem + 128 + CT3_BM * 4 + 2 * CT3_ASTRIDE + RING * BSTAGE + 1024);     // [S + 2][2]
    int4* const tabB = tabA + 2 * (q.smax + 2);                                                                     // [Q + 3]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2;                     // (waves w and w + 4 share a SIMD: one of each column half)
    const int oc0 = blockIdx.z * BN, ksp = blockIdx.y;
    const long long tile = (long long)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);      // XCD b & 7 owns a contiguous range of tiles
    if (tile >= (long long)q.tps * p.N) return;
    const int n = (int)(tile / q.tps), m0 = (int)(tile % q.tps) * CT3_BM;
    const int W = p.IW, H = p.IH;
    const int nt = p.cls_ntaps[0];

    if (tid < 27) {
        const int dd = tid / 9 - 1, dh = (tid / 3) % 3 - 1, dw = tid % 3 - 1;
        int found = -1;
        for (int t = 0; t < nt; ++t) {
            const int ed = p.mode == 0 ? (int)p.tdd[t] - p.pd : (int)p.tdd[t], eh = p.mode == 0 ? (int)p.tdh[t] - p.ph : (int)p.tdh[t],
                      ew = p.mode == 0 ? (int)p.tdw[t] - p.pw : (int)p.tdw[t];
            if (ed == dd && eh == dh && ew == dw) found = t;
        }
        s_tap[tid] = found;
    }
    if (tid < CT3_BM) outrow[tid] = m0 + tid < q.V ? n * q.V + m0 + tid : -1;
    // K range of this block: chunks [c_beg, c_end) x kd slices -> stages; a stage = 3 intervals (kh rows)
    const int c_beg = ksp * q.cps, c_end = c_beg + q.cps < q.nchunks ? c_beg + q.cps : q.nchunks;
    const int S = (c_end > c_beg ? c_end - c_beg : 0) * q.KD, Q = 3 * S;
    __syncthreads();
    for (int st = tid; st < S + 2; st += CT3_THREADS) {           // A descriptors (past the end: zero-size resource)
        int4 e0 = make_int4(0, 0, 0, 0), e1 = make_int4(0, 2, 0, 0);
        if (st < S) {
            const int c = c_beg + st / q.KD, dd = q.KD == 3 ? st % 3 - 1 : 0;
            int ch = c * 32, m = 0;
            while (m + 1 < p.nsrc && ch >= p.srcC[m]) { ch -= p.srcC[m]; ++m; }
            const int Cs = p.srcC[m];
            const unsigned long long base = (unsigned long long)((const bf16_t*)p.src[m] + (long long)n * q.V * Cs);
            e0 = make_int4((int)(unsigned)base, (int)((unsigned)(base >> 32) & 0xffffu), q.V * Cs * 2, 0x00020000);
            e1 = make_int4(m0 - q.halo + dd * q.HW, Cs * 2, ch * 2, 0);          // run origin (voxel), row pitch, channel byte offset
        }
        tabA[2 * st] = e0; tabA[2 * st + 1] = e1;
    }
    for (int qi = tid; qi < Q + 3; qi += CT3_THREADS) {           // B panel offsets of the 3 taps of interval qi (-1: nothing)
        int4 e = make_int4(-1, -1, -1, 0);
        if (qi < Q) {
            const int st = qi / 3, dhi = qi - 3 * st, c = c_beg + st / q.KD, ddi = q.KD == 3 ? st % 3 : 1;
            const int* tp = s_tap + (ddi * 3 + dhi) * 3;
            e.x = tp[0] >= 0 ? (c * nt + tp[0]) * 64 : -1; e.y = tp[1] >= 0 ? (c * nt + tp[1]) * 64 : -1; e.z = tp[2] >= 0 ? (c * nt + tp[2]) * 64 : -1;
        }
        tabB[qi] = e;
    }

    // ---- fragment read addresses, all precomputed.  A: rows of this lane's two voxels under the nine (dh,dw) shifts, k-step 0 (the
    //      second half of the chunk = the same ^ 32, the buffer parity is an immediate); slot = (2 ks + kh8) ^ ((row >> 2) & 3).  A
    //      shift that leaves the plane (TF-SAME padding) points at the zero row behind the run.
    //      B: column rows of the wave's tiles in the three ring slots. ----
    const int kh8 = lane >> 5;                                   // which 8 of a 16-deep k-step this lane supplies
    unsigned aad[9][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int loc = wm * 64 + i * 32 + (lane & 31), m = m0 + loc;
        const int hw = m % q.HW, h = hw / W, w = hw - h * W;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int hh = h + t / 3 - 1, ww = w + t % 3 - 1;
            const bool ok = (unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W;
            const int r = ok ? q.halo + loc + (t / 3 - 1) * W + (t % 3 - 1) : q.arows;
            aad[t][i] = ldsA + (unsigned)(r * 64 + ((((r >> 2) & 3) ^ kh8) << 4));
        }
    }
    const int bcol0 = wn == 0 ? 0 : NJ0 * 32;
    unsigned bad[RING][NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int jj = (wn != 0 && j >= NJ1) ? NJ1 - 1 : j;          // (a tile this column wave does not have: re-read its last one)
        const int row = bcol0 + jj * 32 + (lane & 31);
#pragma unroll
        for (int sl = 0; sl < RING; ++sl) bad[sl][j] = ldsB + (unsigned)(sl * BSTAGE + row * 64 + ((kh8 ^ ((row >> 2) & 3)) << 4));
    }

    // ---- DMA slots of this lane.  A piece = 16 rows x 64 B of the staged run; B piece = 16 panel rows x 64 B of one tap.
    //      A slot past the pieces fetches nothing into the scratch KB (the counts stay uniform, no branches). ----
    constexpr unsigned OOB = 0x80000000u;
    int a_vox[2][CT3_NAS]; unsigned a_dst[2][CT3_NAS]; int a_ssl[2][CT3_NAS];
#pragma unroll
    for (int ph = 0; ph < 2; ++ph)
#pragma unroll
        for (int sl = 0; sl < CT3_NAS; ++sl) {
            const int j = (ph * CT3_NAS + sl) * 8 + wave;
            const int r = j * 16 + (lane >> 2);
            a_vox[ph][sl] = j < q.nA && r < q.arows ? r : -(1 << 22);      // row of the run (behind it, the zero row included: far below zero -> zeros)
            a_dst[ph][sl] = j < q.nA ? ldsA + (unsigned)(j * 1024) : trash;
            a_ssl[ph][sl] = (((lane & 3) ^ ((r >> 2) & 3)) << 4);
        }
    unsigned b_vo[NBS]; unsigned b_dst[NBS]; int b_tap[NBS];
    const int kpad = p.cls_kpad[0];
#pragma unroll
    for (int sl = 0; sl < NBS; ++sl) {
        const int j = sl * 8 + wave;
        const int tp = j / (BN / 16), og = j - tp * (BN / 16);
        const int row = og * 16 + (lane >> 2);
        b_tap[sl] = j < NB ? tp : 3;
        b_dst[sl] = j < NB ? ldsB + (unsigned)(tp * BT + og * 1024) : trash;
        b_vo[sl] = j < NB ? (unsigned)(((oc0 + row) * kpad) * 2 + (((lane & 3) ^ ((row >> 2) & 3)) << 4)) : OOB;
    }
    __syncthreads();

    const unsigned long long wpa = (unsigned long long)((const bf16_t*)p.wp + p.cls_woff[0]);
    i32x4_t rs_b; rs_b.x = (int)(unsigned)wpa; rs_b.y = (int)((unsigned)(wpa >> 32) & 0xffffu); rs_b.z = 0x7fffffff; rs_b.w = 0x00020000;

    // phase ph (0 / 1) of the A stage `st` into buffer `par`: one table entry, no branches
    auto issue_a = [&](int st, int par, int ph) {
        const int4 e0 = tabA[2 * st], e1 = tabA[2 * st + 1];
        i32x4_t rs;
        rs.x = __builtin_amdgcn_readfirstlane(e0.x); rs.y = __builtin_amdgcn_readfirstlane(e0.y);
        rs.z = __builtin_amdgcn_readfirstlane(e0.z); rs.w = 0x00020000;
        const int vsh = e1.x, Cs2 = e1.y, cb = e1.z;                   // (vector registers: every lane read the same entry)
#pragma unroll
        for (int sl = 0; sl < CT3_NAS; ++sl) {
            const int g = a_vox[ph][sl] + vsh;                   // voxel of the sample (negative / beyond it: out of range -> zeros)
            const unsigned vo = (unsigned)(__mul24(g, Cs2) + cb + a_ssl[ph][sl]);
            ct3_dma(rs, a_dst[ph][sl] == trash ? trash : a_dst[ph][sl] + (unsigned)(par * CT3_ASTRIDE), g < 0 ? OOB : vo, 0);
        }
    };
    // weights of interval qi into ring slot `slot`: taps (dd, dh, -1..1) of its chunk
    auto issue_b = [&](int qi, int slot) {
        const int4 e = tabB[qi];
        const int o0 = __builtin_amdgcn_readfirstlane(e.x), o1 = __builtin_amdgcn_readfirstlane(e.y), o2 = __builtin_amdgcn_readfirstlane(e.z);
#pragma unroll
        for (int sl = 0; sl < NBS; ++sl) {
            const int o = b_tap[sl] == 0 ? o0 : (b_tap[sl] == 1 ? o1 : (b_tap[sl] == 2 ? o2 : -1));
            i32x4_t rs = rs_b; rs.z = o >= 0 ? 0x7fffffff : 0;
            ct3_dma(rs, b_dst[sl] == trash ? trash : b_dst[sl] + (unsigned)(slot * BSTAGE), b_vo[sl], o >= 0 ? (unsigned)o : 0u);
        }
    };

    f32x16_t acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // one kh row = 6 k-steps (3 taps x 2 halves of the 32-channel chunk), each 2 A + NJ B fragments -> 2 x NJ MFMAs.  The reads of
    // k-step k + 1 are issued BEFORE the MFMAs of k-step k (two fragment sets), with a counted lgkmcnt: LDS returns in order.
    auto compute = [&](auto parc, auto dhc, auto&& pre) {
        constexpr int PAR = decltype(parc)::value, DHI = decltype(dhc)::value;
        constexpr int SLOT = RING == 3 ? DHI : ((PAR + DHI) & 1);       // interval 3 st + dh: st = (even) + PAR
        u32x4_t fa[2][2], fb[2][NJ];
        auto reads = [&](auto kc) {                               // k-step K = 2 tap + half
            constexpr int K = decltype(kc)::value, DWI = K >> 1, KS = K & 1, BUF = K & 1;
#pragma unroll
            for (int i = 0; i < 2; ++i) ct3_rd<PAR * CT3_ASTRIDE>(fa[BUF][i], KS ? aad[DHI * 3 + DWI][i] ^ 32u : aad[DHI * 3 + DWI][i]);
#pragma unroll
            for (int j = 0; j < NJ; ++j) ct3_rd<DWI * BT>(fb[BUF][j], KS ? bad[SLOT][j] ^ 32u : bad[SLOT][j]);
        };
        reads(std::integral_constant<int, 0>{});
        pre();                                                    // (the DMA issue of the interval: behind the first reads, whose latency it covers)
        ct3_for<0, 6>([&](auto kc) {
            constexpr int K = decltype(kc)::value, BUF = K & 1;
            // the wait is TIED to the fragments of this k-step (in/out operands): the compiler knows nothing of the asynchronous
            // return of an inline-asm read and would otherwise schedule the MFMAs in front of the wait
            if constexpr (K + 1 < 6) reads(std::integral_constant<int, K + 1>{});
            constexpr int LEFT = K + 1 < 6 ? NR : 0;
            if constexpr (NJ == 2) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(fa[BUF][0]), "+v"(fa[BUF][1]), "+v"(fb[BUF][0]), "+v"(fb[BUF][1]) : "n"(LEFT) : "memory");
            else if constexpr (NJ == 4) asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(fa[BUF][0]), "+v"(fa[BUF][1]), "+v"(fb[BUF][0]), "+v"(fb[BUF][1]), "+v"(fb[BUF][2]), "+v"(fb[BUF][3]) : "n"(LEFT) : "memory");
            else asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(fa[BUF][0]), "+v"(fa[BUF][1]), "+v"(fb[BUF][0]), "+v"(fb[BUF][1]), "+v"(fb[BUF][2]) : "n"(LEFT) : "memory");
            // (NJ0 != NJ1, the 3 + 2 split of the 160-column tile: ONE code path -- the narrow column wave reads the fragment of its
            //  last tile a second time, so that the lgkmcnt counts stay the same, and skips the MFMAs of the tile it does not have)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if (NJ0 != NJ1 && j >= NJ1 && wn != 0) continue;
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa[BUF][i]), __builtin_bit_cast(bf16x8_t, fb[BUF][j]), acc[i][j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    // ---- pipeline.  Issue groups in order: A(0) [2 phases], B(0), B(1); then at interval qi: B(qi + 2) and, in its first two kh
    //      rows, the two phases of A(stage + 1).  Before interval qi only the group issued at qi - 1 may still be in flight.
    //      Stages go in pairs: the A buffer of a stage is a compile-time constant. ----
    if (Q > 0) {
        issue_a(0, 0, 0); issue_a(0, 0, 1);
        issue_b(0, 0);
        if constexpr (RING == 3) issue_b(1, 1);
        for (int st0 = 0; st0 < S; st0 += 2) {
            ct3_for<0, 2>([&](auto parc) {
                constexpr int PAR = decltype(parc)::value;
                const int st = st0 + PAR;
                if (st < S) {
                    ct3_for<0, 3>([&](auto dhc) {
                        constexpr int DHI = decltype(dhc)::value;
                        const int qi = 3 * st + DHI;
                        // ring of 3: only the group issued one interval ago may be in flight; ring of 2: nothing (the weights of this
                        // interval were issued one interval ago)
                        if constexpr (RING == 2) ct3_vmwait<0>();
                        else if constexpr (DHI == 0) ct3_vmwait<NBS>(); else ct3_vmwait<NBS + CT3_NAS>();
                        __builtin_amdgcn_s_barrier();
                        compute(parc, dhc, [&]() {
                            if constexpr (RING == 3) issue_b(qi + 2, (DHI + 2) % 3); else issue_b(qi + 1, (PAR + DHI + 1) & 1);
                            if constexpr (DHI < 2) issue_a(st + 1, PAR ^ 1, DHI);
                        });
                    });
                }
            });
        }
        ct3_vmwait<0>();
    }
    __syncthreads();

    // ---- D of a 32x32 tile: lane holds column (lane & 31), rows (e & 3) + 8 (e >> 2) + 4 (lane >> 5) ----
    const int NJw = wn == 0 ? NJ0 : NJ1;
    const int a_bytes = CT3_ASTRIDE;
    if (p.ksplit > 1) {
        float* slab = p.acc32 + (long long)ksp * p.slab_elems;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if (j >= NJw) continue;
                const int oc = oc0 + bcol0 + j * 32 + (lane & 31);
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int orow = outrow[wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)];
                    if (orow >= 0 && oc < p.OCn) slab[(long long)orow * p.OC + oc] = acc[i][j][e];
                }
            }
        return;
    }
    constexpr int CP = BN + 8;
    bf16_t* const C_s = reinterpret_cast<bf16_t*>(A_s);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if (j >= NJw) continue;
            const int col = bcol0 + j * 32 + (lane & 31);
            const float bv = (oc0 + col < p.OCn) ? m1_bias_at(p, oc0 + col) : 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                C_s[row * CP + col] = f2bf(acc[i][j][e] + bv);
            }
        }
    __syncthreads();
    if (p.stat_partial) {
        float* red = reinterpret_cast<float*>(smem + 128 + CT3_BM * 4 + 2 * a_bytes + RING * BSTAGE);       // the ring's scratch KB + the tables (done with both)
        constexpr int G = CT3_THREADS / BN;
        const int col = tid % BN, rg = tid / BN;
        float s = 0.f, ss = 0.f;
        if (rg < G) {
            if (p.ib_x) {
                const int oc = oc0 + col < p.OCn ? oc0 + col : 0;
                const float mean = p.ib_stats[((long long)n * p.OC + oc) * 2], rstd = p.ib_stats[((long long)n * p.OC + oc) * 2 + 1];
                const float gm = p.ib_gamma[oc], bt = p.ib_beta[oc];
                const bf16_t* xb = (const bf16_t*)p.ib_x + oc;
                for (int row = rg; row < CT3_BM; row += G) {
                    const int orow = outrow[row];
                    if (orow >= 0) {
                        const float xh = (bf2f(xb[(long long)orow * p.OC]) - mean) * rstd;
                        const float dy = bf2f(C_s[row * CP + col]) * lrelu_g(gm * xh + bt, p.ib_slope);
                        s += dy; ss += dy * xh;
                    }
                }
            } else {
                for (int row = rg; row < CT3_BM; row += G)
                    if (outrow[row] >= 0) { const float v = bf2f(C_s[row * CP + col]); s += v; ss += v * v; }
            }
        }
        red[tid * 2] = s; red[tid * 2 + 1] = ss;
        __syncthreads();
        if (rg == 0 && oc0 + col < p.OCn) {
            for (int g2 = 1; g2 < G; ++g2) { s += red[(g2 * BN + col) * 2]; ss += red[(g2 * BN + col) * 2 + 1]; }
            float* dst = p.stat_partial + ((tile * p.OC) + oc0 + col) * 2;
            dst[0] = s; dst[1] = ss;
        }
    }
    constexpr int SPR = BN / 8;
    for (int e = tid; e < CT3_BM * SPR; e += CT3_THREADS) {
        const int row = e / SPR, cs = e % SPR;
        const int orow = outrow[row];
        const int oc = oc0 + cs * 8;
        if (orow < 0 || oc >= p.OCn) continue;
        const OutRef o = m1_out_ref(p, oc);
        if (!o.base) continue;
        const uint4 v = *reinterpret_cast<const uint4*>(C_s + row * CP + cs * 8);
        bf16_t* dst = (bf16_t*)o.base + (long long)orow * o.C + o.col;
        if (o.acc) {
            float a[8], b[8];
            VecIO<bf16_t, 8>::ld(reinterpret_cast<const bf16_t*>(&v), a);
            VecIO<bf16_t, 8>::ld(dst, b);
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] += b[k];
            VecIO<bf16_t, 8>::st(dst, a);
        } else *reinterpret_cast<uint4*>(dst) = v;
    }
#endif
}

// ---- host ----
static inline size_t ct3_smem(int BN, int smax) {
    size_t tab = (size_t)(smax + 2) * 32 + (size_t)(3 * smax + 3) * 16;           // descriptor tables
    if (tab < CT3_THREADS * 2 * sizeof(float)) tab = CT3_THREADS * 2 * sizeof(float);      // (the statistics scratch reuses the region)
    return 128 + CT3_BM * 4 + 2 * (size_t)CT3_ASTRIDE + (BN >= 256 ? 2 : 3) * 3 * (size_t)BN * 64 + 1024 + tab;
}

// the shapes this kernel takes, and how: BN (columns per block), K splits.  false: conv_mfma / conv_halo keep the problem.
bool m1_ct3_plan(const GatherSpec& g, int* BN_out, int* ksplit_out) {
    const int en = M1_CFG("M1_CONV_T3", 1);
    if (!en || g.dtype != M1_BF16) return false;
    if (g.sd != 1 || g.sh != 1 || g.sw != 1) return false;
    if (!(g.kh == 3 && g.kw == 3 && (g.kd == 3 || g.kd == 1))) return false;
    if (g.ID != g.OD || g.IH != g.OH || g.IW != g.OW) return false;
    if (g.mode == 0 && !(g.ph == 1 && g.pw == 1 && g.pd == (g.kd == 3 ? 1 : 0))) return false;
    int CC = 0;
    for (int i = 0; i < g.nsrc; ++i) { if (g.srcC[i] % 32) return false; CC += g.srcC[i]; }
    if (g.OC % 8) return false;
    for (int i = 0; i < g.nout; ++i) if (g.outC[i] % 8) return false;
    const long long V = (long long)g.OD * g.OH * g.OW;
    const int W = g.IW;
    const int arows = CT3_BM + 2 * (W + 1), nA = (arows + 16) / 16;          // (+ the zero row)
    if (nA * 1024 > CT3_ASTRIDE || nA > 2 * CT3_NAS * 8) return false;      // (W <= 46: res2 and deeper)
    if (V * 768 * 2 >= (1ll << 31) || V < 64) return false;
    const int minc = M1_CFG("M1_CT3_MINC", 96), minoc = M1_CFG("M1_CT3_MINOC", 96);
    const long long minm = M1_CFG("M1_CT3_MINM", 4096);
    if (CC < minc || g.OC < minoc || (long long)g.N * V < minm) return false;
    for (int i = 0; i < g.nsrc; ++i) if (V * g.srcC[i] * 2 >= (1ll << 31) - 4096) return false;
    const long long tiles = (long long)g.N * cdiv_ll(V, CT3_BM);
    const int nchunks = CC / 32, S = nchunks * g.kd;
    const int cus = M1_CFG("M1_CT3_CUS", 256);
    // Cost model (us, fitted to same-box measurements of the C3 layers, profiles/r04_conv_t3_layers.txt): one block per CU, so a launch
    // takes rounds = ceil(blocks / 256) block times; a block = a fixed part (prologue: tables, addresses, first DMA round trip;
    // epilogue: LDS tile, statistics, stores) + its kh-row intervals, whose time grows with the tile width more slowly than the MFMA
    // count (fewer fragment reads and less staged input per MFMA); split-K adds the slab round trip of the finish pass.
    double best = 1e30; int bBN = 0, bks = 1;
    for (int BN : {128, 160, 192, 256}) {
        const int ntile = (g.OC + BN - 1) / BN;
        const double t_int = BN == 128 ? 1.5 : (BN == 160 ? 1.85 : (BN == 192 ? 2.15 : 2.85)), t_fix = 10.0 + 0.03 * BN;
        for (int ks = 1; ks <= 8; ++ks) {
            if (ks > 1 && nchunks / ks < 4) break;
            if (ct3_smem(BN, (nchunks + ks - 1) / ks * g.kd) > 160 * 1024) continue;
            const long long blocks = tiles * ntile * ks;
            const long long rounds = cdiv_ll(blocks, cus);
            const int cps = (nchunks + ks - 1) / ks;
            double t = (double)rounds * (t_fix + 3.0 * cps * g.kd * t_int);
            if (ks > 1) t += 6.0 + (double)g.N * V * g.OC * (4.0 * ks + 2.0) / 3.0e6;      // slabs written + read back, output written (3 TB/s)
            if (t < best) { best = t; bBN = BN; bks = ks; }
        }
    }
    if (!bBN) return false;
    const int fbn = M1_CFG("M1_CT3_BN", 0), fks = M1_CFG("M1_CT3_KSPLIT", 0);
    if (fbn == 128 || fbn == 160 || fbn == 192 || fbn == 256) bBN = fbn;
    if (fks >= 1) bks = fks;
    if (ct3_smem(bBN, (nchunks + bks - 1) / bks * g.kd) > 160 * 1024) return false;
    *BN_out = bBN; *ksplit_out = bks;
    return true;
}

// tiles (= statistics partial rows) per sample: runs of 256 voxels
int m1_ct3_tiles_per_sample(int D, int H, int W) { return (int)cdiv_ll((long long)D * H * W, CT3_BM); }

int m1_ct3_conv(const MfmaP& mp, int BN, int OCpad, hipStream_t st) {
    Ct3P q{};
    const long long V = (long long)mp.OD * mp.OH * mp.OW;
    q.V = (int)V; q.HW = mp.IH * mp.IW; q.tps = (int)cdiv_ll(V, CT3_BM);
    q.halo = mp.IW + 1; q.arows = CT3_BM + 2 * q.halo; q.nA = (q.arows + 16) / 16;
    q.KD = mp.cls_ntaps[0] == 27 ? 3 : 1;
    q.nchunks = mp.CC / 32; q.cps = (q.nchunks + mp.ksplit - 1) / mp.ksplit; q.OCpad = OCpad; q.smax = q.cps * q.KD;
    if (mp.cls_ntaps[0] != 27 && mp.cls_ntaps[0] != 9) return M1_ERR_UNSUPPORTED;
    const size_t smem = ct3_smem(BN, q.smax);
    void (*kern)(MfmaP, Ct3P) = BN == 128 ? conv_t3_kernel<2, 2> : (BN == 160 ? conv_t3_kernel<3, 2> : (BN == 192 ? conv_t3_kernel<3, 3> : conv_t3_kernel<4, 4>));
    {
        static const void* done[4]; static int ndone = 0;
        bool seen = false;
        for (int i = 0; i < ndone; ++i) seen |= done[i] == (const void*)kern;
        if (!seen) {
            if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return M1_ERR_LAUNCH;
            if (ndone < 4) done[ndone++] = (const void*)kern;
        }
    }
    const long long tiles = (long long)q.tps * mp.N;
    dim3 grid((unsigned)(cdiv_ll(tiles, 8) * 8), (unsigned)mp.ksplit, (unsigned)(OCpad / BN));
    m1_note_kernel("conv_t3:bn%d:ks%d", BN, mp.ksplit);
    hipLaunchKernelGGL(kern, grid, dim3(CT3_THREADS), smem, st, mp, q);
    return m1_check_launch();
}
