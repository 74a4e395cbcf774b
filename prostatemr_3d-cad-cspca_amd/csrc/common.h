// common.h -- shared device/host helpers for libm1hip (gfx950 / CDNA4 only, wave = 64).
#pragma once
#include <stdlib.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/m1hip.h"

#define M1_WAVE 64
#define M1_LRELU_EPS_UNUSED 0

typedef unsigned short bf16_t;  // raw bf16 storage

// ---- bf16 <-> f32 (round-to-nearest-even, NaN preserved) ----
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);  // quiet NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

template <typename T> struct Act;
template <> struct Act<float> {
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Act<bf16_t> {
    static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
    static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};

// ---- vector load/store of VEC consecutive channels as floats (16 B per lane when VEC = 4 (f32) / 8 (bf16)) ----
template <typename T, int VEC> struct VecIO;
template <> struct VecIO<float, 1> {
    static __device__ __forceinline__ void ld(const float* p, float* o) { o[0] = p[0]; }
    static __device__ __forceinline__ void st(float* p, const float* o) { p[0] = o[0]; }
};
template <> struct VecIO<float, 4> {
    static __device__ __forceinline__ void ld(const float* p, float* o) {
        float4 v = *reinterpret_cast<const float4*>(p);
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    }
    static __device__ __forceinline__ void st(float* p, const float* o) {
        *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
    }
};
template <> struct VecIO<float, 8> {
    static __device__ __forceinline__ void ld(const float* p, float* o) {
        VecIO<float, 4>::ld(p, o); VecIO<float, 4>::ld(p + 4, o + 4);
    }
    static __device__ __forceinline__ void st(float* p, const float* o) {
        VecIO<float, 4>::st(p, o); VecIO<float, 4>::st(p + 4, o + 4);
    }
};
template <> struct VecIO<bf16_t, 1> {
    static __device__ __forceinline__ void ld(const bf16_t* p, float* o) { o[0] = bf2f(p[0]); }
    static __device__ __forceinline__ void st(bf16_t* p, const float* o) { p[0] = f2bf(o[0]); }
};
template <> struct VecIO<bf16_t, 4> {
    static __device__ __forceinline__ void ld(const bf16_t* p, float* o) {
        uint2 v = *reinterpret_cast<const uint2*>(p);
        o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
        o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
    }
    static __device__ __forceinline__ void st(bf16_t* p, const float* o) {
        uint2 v;
        v.x = (unsigned)f2bf(o[0]) | ((unsigned)f2bf(o[1]) << 16);
        v.y = (unsigned)f2bf(o[2]) | ((unsigned)f2bf(o[3]) << 16);
        *reinterpret_cast<uint2*>(p) = v;
    }
};
template <> struct VecIO<bf16_t, 8> {
    static __device__ __forceinline__ void ld(const bf16_t* p, float* o) {
        uint4 v = *reinterpret_cast<const uint4*>(p);
        o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
        o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
        o[4] = __uint_as_float(v.z << 16); o[5] = __uint_as_float(v.z & 0xffff0000u);
        o[6] = __uint_as_float(v.w << 16); o[7] = __uint_as_float(v.w & 0xffff0000u);
    }
    static __device__ __forceinline__ void st(bf16_t* p, const float* o) {
        uint4 v;
        v.x = (unsigned)f2bf(o[0]) | ((unsigned)f2bf(o[1]) << 16);
        v.y = (unsigned)f2bf(o[2]) | ((unsigned)f2bf(o[3]) << 16);
        v.z = (unsigned)f2bf(o[4]) | ((unsigned)f2bf(o[5]) << 16);
        v.w = (unsigned)f2bf(o[6]) | ((unsigned)f2bf(o[7]) << 16);
        *reinterpret_cast<uint4*>(p) = v;
    }
};

// ---- tuning switches: ONE table for the whole library (config.hip) ----
// M1_CFG("M1_X", default) = the switch's current value: the default, or the environment variable of the same name when the process
// started with it, or what m1_config_set (include/m1hip.h) stored last -- a caller can change a switch between two launches and put
// it back (tests run the same op under several settings in one process).  The call site keeps a pointer to its table entry: one load.
struct M1CfgEntry { char name[40]; int def; int def_known; int env; int env_set; int ovr; int has_ovr; volatile int v; };
M1CfgEntry* m1_cfg_entry(const char* name, int def);
bool m1_debug_skip(const char* name);
#define M1_CFG(NAME, DEF) ([]() -> int { static M1CfgEntry* const e_ = m1_cfg_entry(NAME, DEF); return e_->v; }())

// blocks of 256 threads for `per` vector elements whose channel group is (index % cg): at most ~2048 blocks, and
// gridDim.x*256 a multiple of cg so that a thread keeps its channel group across its grid-stride loop
static inline int m1_grid_for(long long per, int cg) {
    int cap = M1_CFG("M1_EW_BLOCKS", 2048); { if (cap < 1) cap = 1; }
    long long g = (per + 255) / 256; if (g > cap) g = cap; if (g < 1) g = 1;
    // (g*256) % cg == 0  <=>  g is a multiple of cg / gcd(cg, 256)
    long long a = cg, b = 256; while (b) { long long t = a % b; a = b; b = t; }
    const long long need = cg / a;
    g = (g + need - 1) / need * need;
    return (int)g;
}
__device__ __forceinline__ float lrelu_f(float x, float slope) { return x >= 0.f ? x : slope * x; }
__device__ __forceinline__ float lrelu_g(float x, float slope) { return x >= 0.f ? 1.f : slope; }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + __expf(-x)); }

// ---- wave / block reductions (wave64 shuffles) ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- Philox4x32-10 (counter-based RNG; the keep-mask is a pure function of (seed, element index)) ----
__device__ __forceinline__ void philox_round(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t k0,
                                             uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    // (one 32x32 -> 64 multiply per product: v_mad_u64_u32 instead of a v_mul_hi_u32 + v_mul_lo_u32 pair, both quarter rate)
    const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
    uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}
__device__ __forceinline__ uint4 philox4x32_10(uint64_t seed, uint64_t ctr) {
    uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = 0x243F6A88u, c3 = 0x85A308D3u;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        philox_round(c0, c1, c2, c3, k0, k1);
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return make_uint4(c0, c1, c2, c3);
}
// keep-decision of element `idx` of the dropout stream (seed, base): U[0,1) >= rate  
__device__ __forceinline__ bool philox_keep(uint64_t seed, uint64_t base, uint64_t idx, float rate) {
    uint64_t e = base + idx;
    uint4 r = philox4x32_10(seed, e >> 2);
    uint32_t w = (e & 3) == 0 ? r.x : (e & 3) == 1 ? r.y : (e & 3) == 2 ? r.z : r.w;
    float u = (float)(w >> 8) * (1.0f / 16777216.0f);
    return u >= rate;
}

// VEC consecutive elements starting at idx0 (one Philox call per 4 consecutive stream positions when aligned)
// ALIGNED: the caller guarantees (base + idx0) % 4 == 0 (no code for the element-wise path)
template <int VEC, bool ALIGNED = false>
__device__ __forceinline__ void philox_keep_vec(uint64_t seed, uint64_t base, uint64_t idx0, float rate, bool* keep) {
    const uint64_t e0 = base + idx0;
    if constexpr (VEC % 4 == 0) {
        if (ALIGNED || (e0 & 3) == 0) {
#pragma unroll
            for (int q = 0; q < VEC / 4; ++q) {
                const uint4 r = philox4x32_10(seed, (e0 >> 2) + q);
                keep[4 * q + 0] = (float)(r.x >> 8) * (1.0f / 16777216.0f) >= rate;
                keep[4 * q + 1] = (float)(r.y >> 8) * (1.0f / 16777216.0f) >= rate;
                keep[4 * q + 2] = (float)(r.z >> 8) * (1.0f / 16777216.0f) >= rate;
                keep[4 * q + 3] = (float)(r.w >> 8) * (1.0f / 16777216.0f) >= rate;
            }
            return;
        }
    }
#pragma unroll
    for (int k = 0; k < VEC; ++k) keep[k] = philox_keep(seed, base, idx0 + k, rate);
}

// ---- host helpers ----
static inline int m1_check_launch() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? M1_OK : M1_ERR_LAUNCH;
}
static inline long long cdiv_ll(long long a, long long b) { return (a + b - 1) / b; }

// kernel-choice log (prof.hip; m1_debug_kernels in include/m1hip.h): the dispatch names the kernel behind every conv-like launch so
// that a test of a special kernel can assert it ran (a declined shape would otherwise compare the generic kernel with itself)
void m1_note_kernel(const char* fmt, ...);

// profiler hooks (prof.hip)
struct M1ProfScope {
    int slot;
    hipStream_t s;
    M1ProfScope(const char* name, double flops, double bytes, hipStream_t stream);
    ~M1ProfScope();
};
