// debug.hip -- opt-in probes for state-dependent results (never launched by the product path unless a debug switch asks for them):
//
//   m1_debug_checksum   a deterministic 64-bit checksum of a device buffer into a slot of a device log, as ONE kernel launch: it can be
//                       captured into the step's hipGraph, so the outputs of every op of a REPLAYED step can be compared between two
//                       processes (hip/ops.py M1_DEBUG_TRACE; tools/dbg/first_diff.py names the first op whose output differs).
//   m1_debug_scribble   leaves a NaN pattern in every byte of LDS, every VGPR and every AGPR of all CUs: a kernel that reads LDS or a
//                       register it never wrote (a partial tile, an accumulator that was not zeroed, a cross-lane read of lanes that
//                       were masked off) then produces NaN in an ordinary in-order run instead of a value that depends on which kernel
//                       ran on that CU before (hip/ops.py M1_DEBUG_POISON=2 launches it in front of every entry point).
#include "common.h"

__global__ void __launch_bounds__(1024) m1_checksum_kernel(const unsigned char* __restrict__ p, long long nbytes,
                                                           unsigned long long* __restrict__ slot) {
    __shared__ unsigned long long red[16];
    const long long nw = nbytes >> 2;
    const unsigned* w = reinterpret_cast<const unsigned*>(p);
    unsigned long long s = 0;
    for (long long i = threadIdx.x; i < nw; i += 1024) s += (unsigned long long)w[i] * (unsigned long long)((i & 0xffff) + 1);
    if (threadIdx.x == 0) for (long long b = nw << 2; b < nbytes; ++b) s += (unsigned long long)p[b] * 0x9E3779B97F4A7C15ull;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned lo = __shfl_xor((unsigned)s, o, 64), hi = __shfl_xor((unsigned)(s >> 32), o, 64);
        s += ((unsigned long long)hi << 32) | lo;
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int i = 0; i < 16; ++i) t += red[i];
        *slot = t;
    }
}

extern "C" int m1_debug_checksum(const void* p, long long nbytes, unsigned long long* slot, void* stream) {
    if (!p || !slot || nbytes < 0 || ((uintptr_t)p & 3)) return M1_ERR_BAD_ARG;
    hipLaunchKernelGGL(m1_checksum_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (const unsigned char*)p, nbytes, slot);
    return m1_check_launch();
}

// One workgroup per CU at a time (it asks for the whole 160 KB of LDS), four waves = one per SIMD, each with the whole register file of
// its SIMD (256 VGPRs + 256 AGPRs).  0x7FC07FC0 is a quiet NaN as fp32 and as a pair of bf16.
#define SCR_C10(k, p) #k #p "0", #k #p "1", #k #p "2", #k #p "3", #k #p "4", #k #p "5", #k #p "6", #k #p "7", #k #p "8", #k #p "9"
#define SCR_C100(k, p) SCR_C10(k, p##0), SCR_C10(k, p##1), SCR_C10(k, p##2), SCR_C10(k, p##3), SCR_C10(k, p##4), SCR_C10(k, p##5), \
                       SCR_C10(k, p##6), SCR_C10(k, p##7), SCR_C10(k, p##8), SCR_C10(k, p##9)
#define SCR_CALL(k) SCR_C10(k, ), SCR_C10(k, 1), SCR_C10(k, 2), SCR_C10(k, 3), SCR_C10(k, 4), SCR_C10(k, 5), SCR_C10(k, 6), SCR_C10(k, 7), \
                    SCR_C10(k, 8), SCR_C10(k, 9), SCR_C100(k, 1), SCR_C10(k, 20), SCR_C10(k, 21), SCR_C10(k, 22), SCR_C10(k, 23), \
                    SCR_C10(k, 24), #k "250", #k "251", #k "252", #k "253", #k "254", #k "255"

__global__ void __launch_bounds__(256) m1_scribble_kernel(int spins) {
    extern __shared__ unsigned scr_lds[];
    const unsigned pat = 0x7FC07FC0u;
    for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 256) scr_lds[i] = pat;
    __syncthreads();
    for (int s = 0; s < spins; ++s) __builtin_amdgcn_s_sleep(32);          // stay resident until every CU has taken a workgroup
    const unsigned sp = __builtin_amdgcn_readfirstlane(pat);
    asm volatile(".set scr_i, 0\n .rept 256\n v_mov_b32 v[scr_i], %0\n .set scr_i, scr_i+1\n .endr\n"
                 ".set scr_i, 0\n .rept 256\n v_accvgpr_write_b32 a[scr_i], v0\n .set scr_i, scr_i+1\n .endr\n"
                 :: "s"(sp) : SCR_CALL(v), SCR_CALL(a));
}

extern "C" int m1_debug_scribble(int blocks, int spins, void* stream) {
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)m1_scribble_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return M1_ERR_LAUNCH;
        attr = true;
    }
    if (blocks <= 0) blocks = 512;
    hipLaunchKernelGGL(m1_scribble_kernel, dim3(blocks), dim3(256), 160 * 1024, (hipStream_t)stream, spins < 0 ? 0 : spins);
    return m1_check_launch();
}
