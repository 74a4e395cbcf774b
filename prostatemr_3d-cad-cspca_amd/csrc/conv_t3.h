// conv_t3.h -- staged-run implicit-GEMM conv for the stride-1 matrix-core layers (conv_t3.hip); takes conv_mfma's parameters and panel
#pragma once
#include "conv_mfma.h"

// the shapes the kernel takes and how it tiles them (columns per block, K splits); false: conv_mfma / conv_halo keep the problem
bool m1_ct3_plan(const GatherSpec& g, int* BN, int* ksplit);
int m1_ct3_tiles_per_sample(int D, int H, int W);      // = rows per sample of its statistics partials
int m1_ct3_conv(const MfmaP& mp, int BN, int OCpad, hipStream_t st);
