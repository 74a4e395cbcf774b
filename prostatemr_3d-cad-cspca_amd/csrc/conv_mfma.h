// conv_mfma.h -- launch parameters shared by the implicit-GEMM conv kernels (conv_mfma.hip, conv_halo.hip)
#pragma once
#include "common.h"
#include "gather.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#define MF_MAX_TAPS 27
#define MF_MAX_CLASSES 8

struct MfmaP {
    const void* src[M1_MAX_SRC];
    int srcC[M1_MAX_SRC];
    int srcSeg[M1_MAX_SRC];     // 16-byte K segments per member = ceil(C / SEG): a member that is not a multiple of SEG
                                // (latent z: 1..3 channels, stem input: 2..3) is zero-padded to whole segments in K space
    int nsrc, CC, spt;          // CC = contraction channels per tap (sum srcC); spt = segments per tap (sum srcSeg)
    int ID, IH, IW;             // gathered tensor extent
    void* out;
    int OC, OCn;                // out row stride (channels) / channels computed by this launch
    int nout;                   // > 0: columns are spread over several tensors (GatherSpec::outs): member m owns [outOff[m], outOff[m+1])
    void* outs[M1_MAX_SRC]; int outC[M1_MAX_SRC]; int outOff[M1_MAX_SRC + 1]; int outAcc[M1_MAX_SRC];
    int OD, OH, OW, N;
    const void* wp;             // packed weights
    const float* bias;
    const float* bias2; int bias_split;   // columns >= bias_split take bias2[col - bias_split] (0 = one bias vector)
    int mode, sd, sh, sw, pd, ph, pw;
    int nclasses;
    int cls_ntaps[MF_MAX_CLASSES], cls_first[MF_MAX_CLASSES], cls_kpad[MF_MAX_CLASSES];
    long long cls_woff[MF_MAX_CLASSES];     // element offset of the class matrix in wp
    signed char tdd[MF_MAX_TAPS], tdh[MF_MAX_TAPS], tdw[MF_MAX_TAPS];   // gather offsets per (class-ordered) tap
    int tap_pk[MF_MAX_TAPS];    // the same, packed dd | dh<<8 | dw<<16 (scalar loads in the LDS-DMA loader)
    int accumulate;             // out += result (used when another kernel already wrote the other concat members)
    int ksplit;                 // > 1: blockIdx.y = cls*ksplit + ks; partial sums go to acc32 with fp32 atomics
    float* acc32;               // [ksplit][out voxels][OC] fp32 slabs (ksplit > 1 only)
    long long slab_elems;
    int aligned;                // every concat member is a multiple of one 64-byte K-chunk: incremental addressing
    int korder;                 // 1: K runs [64-byte chunk of the concat][tap] (needs aligned), 0: [tap][concat channel]
    const void* ib_x;           // != nullptr: stat_partial receives the InstanceNorm-BACKWARD sums {sum dy, sum dy*xh} of the (rounded)
    const float* ib_stats; const float* ib_gamma; const float* ib_beta; float ib_slope;   //   outputs instead (GatherSpec::ib_*)
    int tps;                    // > 0 (conv_mfma_kernel only, one class): the M tiles run PER SAMPLE, tps of them each, so that no tile straddles two
                                //   samples and the per-tile statistics stay per-sample when V % BM != 0 (round 6)
    float* stat_partial;        // fused InstanceNorm statistics: [N][stat_tiles][OC][2] = {sum, sum of squares} of the ROUNDED
    int stat_tiles;             //   outputs, one partial per 64/128-row tile (mode 0, tiles never straddle samples) or, in the
                                //   halo kernel, per (sample, block row)
};

// where output column `oc` of a launch lives: tensor base (nullptr = nobody wants it), its row pitch, the column inside it,
// and whether the kernel adds to what is there
struct OutRef { void* base; int C, col, acc; };
__device__ __forceinline__ OutRef m1_out_ref(const MfmaP& p, int oc) {
    if (p.nout == 0) return OutRef{p.out, p.OC, oc, p.accumulate};
    int m = 0;
    while (m + 1 < p.nout && oc >= p.outOff[m + 1]) ++m;
    return OutRef{p.outs[m], p.outC[m], oc - p.outOff[m], p.outAcc[m]};
}

__device__ __forceinline__ float m1_bias_at(const MfmaP& p, int oc) {
    if (p.bias_split > 0 && oc >= p.bias_split) return p.bias2 ? p.bias2[oc - p.bias_split] : 0.f;
    return p.bias ? p.bias[oc] : 0.f;
}

template <typename T> struct MT;
template <> struct MT<bf16_t> { static constexpr int SEG = 8; };
template <> struct MT<float> { static constexpr int SEG = 4; };


// halo-tile variant (conv_halo.hip): takes the same parameters and the same packed panel
bool m1_halo_conv_supported(const MfmaP& mp, int OCpad);
int m1_halo_conv_stat_parts(const MfmaP& mp, int OCpad);
int m1_halo_conv(const MfmaP& mp, int OCpad, hipStream_t st);

// pointwise (1x1x1, stride 1) streaming variant (conv_pw.hip): same parameters and panel; BN = output channels per block
bool m1_pw_conv_supported(const MfmaP& mp, int OCpad, int BN);
int m1_pw_conv_stat_parts(const MfmaP& mp, int OCpad, int BN);
int m1_pw_conv(const MfmaP& mp, int OCpad, int BN, hipStream_t st);
