// grid_barrier_norm.hip -- probe (round 6, review item 6): is ONE launch with a grid barrier cheaper than THREE launches for the
// statistics -> finalize -> apply chain of an InstanceNorm + LeakyReLU on the deep levels' tensors?
//
//   A: k_stats (partial sums per voxel chunk) -> k_final (fold, mean / rstd) -> k_apply            three launches on one stream
//   B: k_fused: the same three phases in one launch; between them a device-wide counter barrier (lane-0 release fence -> relaxed
//      atomic arrive -> relaxed polling loads + s_sleep -> acquire fence -> __syncthreads; the hand-off form MI355X_MICROARCH.md
//      prescribes), all blocks co-resident (grid <= 256), spins bounded.
// bf16 (N, V, C) tensors, 256-thread blocks, a lane owns 8 channels.  Both variants run 300 times back to back on one stream between
// two events; the outputs are compared.  Build / run: tools/probes/grid_barrier_norm.sh
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned short bf16;
__device__ __forceinline__ float bf2f(bf16 v) { return __uint_as_float((unsigned)v << 16); }
__device__ __forceinline__ bf16 f2bf(float f) { unsigned u = __float_as_uint(f); u += 0x7fffu + ((u >> 16) & 1u); return (bf16)(u >> 16); }

// phase 1: block (chunk, n): sums of x and x^2 over its voxels, per channel -> partial[n][chunk][C][2]
__device__ void phase_stats(const bf16* x, long long V, int C, int chunkV, int nchunks, float* partial, int n, int chunk, float* red) {
    const int cg = C / 8, tid = threadIdx.x, cl = tid % cg, vs = tid / cg, nvs = 256 / cg;
    const long long v0 = (long long)chunk * chunkV; long long v1 = v0 + chunkV; if (v1 > V) v1 = V;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (long long v = v0 + vs; v < v1; v += nvs) {
        const uint4 u = *reinterpret_cast<const uint4*>(x + ((size_t)n * V + v) * C + cl * 8);
        const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float a = __uint_as_float(w[k] << 16), b = __uint_as_float(w[k] & 0xffff0000u);
            s[2 * k] += a; q[2 * k] += a * a; s[2 * k + 1] += b; q[2 * k + 1] += b * b;
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) { red[(tid * 8 + k) * 2] = s[k]; red[(tid * 8 + k) * 2 + 1] = q[k]; }
    __syncthreads();
    if (vs == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float a = 0.f, b = 0.f;
            for (int j = 0; j < nvs; ++j) { a += red[((j * cg + cl) * 8 + k) * 2]; b += red[((j * cg + cl) * 8 + k) * 2 + 1]; }
            float* dst = partial + (((size_t)n * nchunks + chunk) * C + cl * 8 + k) * 2;
            dst[0] = a; dst[1] = b;
        }
    }
    __syncthreads();
}
// phase 3: the block's voxel chunk through (x - mean) rstd gamma + beta, LeakyReLU(0.1)
__device__ void phase_apply(const bf16* x, bf16* y, long long V, int C, int chunkV, int n, int chunk, const float* ms /*[C][2]*/,
                            const float* gamma, const float* beta) {
    const int cg = C / 8, tid = threadIdx.x, cl = tid % cg, vs = tid / cg, nvs = 256 / cg;
    const long long v0 = (long long)chunk * chunkV; long long v1 = v0 + chunkV; if (v1 > V) v1 = V;
    float sc[8], sh[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int c = cl * 8 + k; const float m = ms[c * 2], r = ms[c * 2 + 1];
        sc[k] = r * gamma[c]; sh[k] = beta[c] - m * sc[k];
    }
    for (long long v = v0 + vs; v < v1; v += nvs) {
        const size_t o = ((size_t)n * V + v) * C + cl * 8;
        const uint4 u = *reinterpret_cast<const uint4*>(x + o);
        const unsigned w[4] = {u.x, u.y, u.z, u.w}; unsigned r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float a = __uint_as_float(w[k] << 16) * sc[2 * k] + sh[2 * k], b = __uint_as_float(w[k] & 0xffff0000u) * sc[2 * k + 1] + sh[2 * k + 1];
            a = a > 0.f ? a : 0.1f * a; b = b > 0.f ? b : 0.1f * b;
            r[k] = (unsigned)f2bf(a) | ((unsigned)f2bf(b) << 16);
        }
        *reinterpret_cast<uint4*>(y + o) = make_uint4(r[0], r[1], r[2], r[3]);
    }
}

__global__ void __launch_bounds__(256) k_stats(const bf16* x, long long V, int C, int chunkV, int nchunks, float* partial) {
    __shared__ float red[256 * 16];
    phase_stats(x, V, C, chunkV, nchunks, partial, blockIdx.y, blockIdx.x, red);
}
__global__ void __launch_bounds__(256) k_final(const float* partial, int C, int nchunks, long long V, float* stats) {
    __shared__ double red[4][2];
    const int n = blockIdx.x / C, c = blockIdx.x % C, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double a = 0.0, b = 0.0;
    for (int j = threadIdx.x; j < nchunks; j += 256) { a += partial[(((size_t)n * nchunks + j) * C + c) * 2]; b += partial[(((size_t)n * nchunks + j) * C + c) * 2 + 1]; }
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
    if (lane == 0) { red[wave][0] = a; red[wave][1] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        a = red[0][0] + red[1][0] + red[2][0] + red[3][0]; b = red[0][1] + red[1][1] + red[2][1] + red[3][1];
        const double m = a / (double)V; double var = b / (double)V - m * m; if (var < 0) var = 0;
        stats[(size_t)blockIdx.x * 2] = (float)m; stats[(size_t)blockIdx.x * 2 + 1] = (float)(1.0 / sqrt(var + 1e-3));
    }
}
__global__ void __launch_bounds__(256) k_apply(const bf16* x, bf16* y, long long V, int C, int chunkV, const float* stats, const float* gamma,
                                               const float* beta) {
    phase_apply(x, y, V, C, chunkV, blockIdx.y, blockIdx.x, stats + (size_t)blockIdx.y * C * 2, gamma, beta);
}

// one launch: phase 1 -> barrier -> every block folds the rows of ITS sample (fp64, through LDS) -> phase 3
__global__ void __launch_bounds__(256) k_fused(const bf16* x, bf16* y, long long V, int C, int chunkV, int nchunks, float* partial, float* stats,
                                               const float* gamma, const float* beta, unsigned* bar /*[2]: arrivals, departures*/, int* err) {
    __shared__ float red[256 * 16];
    const int n = blockIdx.y, chunk = blockIdx.x, nblk = gridDim.x * gridDim.y;
    phase_stats(x, V, C, chunkV, nchunks, partial, n, chunk, red);
    // ---- device-wide barrier ----
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)nblk) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1 << 22)) { *err = 1; break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    // ---- phase 2: mean / rstd of this sample's channels: thread t folds value t (and t + 256 ...) of the C x 2 sums over the rows ----
    // (rows spread over the lane rows, eight loads in flight: a thread walking all rows alone is 64 dependent L2 round trips)
    float* ms = red;                                               // [C][2], then scratch [rows][C*2] behind it
    const int nv = C * 2, rws = 256 / nv > 0 ? 256 / nv : 1;       // lane rows per value (C <= 128: 1..4)
    float* scr = red + 512;
    for (int e0 = 0; e0 < nv; e0 += 256) {
        const int e = e0 + (int)threadIdx.x % (nv < 256 ? nv : 256), r0 = (int)threadIdx.x / (nv < 256 ? nv : 256);
        double a = 0.0;
        if (e < nv && r0 < rws) {
#pragma unroll 8
            for (int j = r0; j < nchunks; j += rws) a += (double)partial[((size_t)n * nchunks + j) * nv + e];
        }
        scr[threadIdx.x] = (float)a;
        __syncthreads();
        if (r0 == 0 && e < nv) { float t = 0.f; for (int q = 0; q < rws; ++q) t += scr[q * (nv < 256 ? nv : 256) + (int)threadIdx.x]; ms[e] = t; }
        __syncthreads();
    }
    float mf = 0.f, rf = 0.f;
    if (threadIdx.x < C) {
        const double m = (double)ms[threadIdx.x * 2] / (double)V; double var = (double)ms[threadIdx.x * 2 + 1] / (double)V - m * m; if (var < 0) var = 0;
        mf = (float)m; rf = (float)(1.0 / sqrt(var + 1e-3));
    }
    __syncthreads();
    if (threadIdx.x < C) {
        ms[threadIdx.x * 2] = mf; ms[threadIdx.x * 2 + 1] = rf;
        if (chunk == 0) { stats[((size_t)n * C + threadIdx.x) * 2] = mf; stats[((size_t)n * C + threadIdx.x) * 2 + 1] = rf; }
    }
    __syncthreads();
    phase_apply(x, y, V, C, chunkV, n, chunk, ms, gamma, beta);
    // ---- the last block to leave re-arms the barrier (no memset between launches / graph replays) ----
    if (threadIdx.x == 0) {
        const unsigned d = __hip_atomic_fetch_add(bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (d == (unsigned)nblk - 1) { __hip_atomic_store(bar + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(bar, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    }
}

int main() {
    const int shapes[][3] = {{4, 4000, 64}, {4, 4000, 256}, {4, 500, 128}, {4, 500, 512}, {4, 32000, 32}, {4, 32000, 128}};   // N, V, C (C <= 128 for k_fused's LDS: see below)
    printf("%-22s %10s %10s %10s %8s\n", "N x V x C (bf16)", "3 launches", "1 launch", "ratio", "max|d|");
    for (auto& sh : shapes) {
        const int N = sh[0], C = sh[2]; const long long V = sh[1];
        if (C > 256 || 256 % (C / 8)) continue;
        int nchunks = 256 / N; while ((V + nchunks - 1) / nchunks < 256 / (C / 8) && nchunks > 1) nchunks >>= 1;     // >= one voxel per lane row
        const int chunkV = (int)((V + nchunks - 1) / nchunks); nchunks = (int)((V + chunkV - 1) / chunkV);
        const size_t ne = (size_t)N * V * C;
        std::vector<bf16> hx(ne); srand(1);
        for (size_t i = 0; i < ne; ++i) { float f = (float)rand() / RAND_MAX * 2.f - 1.f; unsigned u; memcpy(&u, &f, 4); hx[i] = (bf16)(u >> 16); }
        std::vector<float> hg(C, 1.f), hb(C, 0.1f);
        bf16 *x, *y0, *y1; float *partial, *stats, *gamma, *beta; unsigned* bar; int* err;
        CK(hipMalloc(&x, ne * 2)); CK(hipMalloc(&y0, ne * 2)); CK(hipMalloc(&y1, ne * 2));
        CK(hipMalloc(&partial, (size_t)N * nchunks * C * 2 * 4)); CK(hipMalloc(&stats, (size_t)N * C * 2 * 4));
        CK(hipMalloc(&gamma, C * 4)); CK(hipMalloc(&beta, C * 4)); CK(hipMalloc(&bar, 8)); CK(hipMalloc(&err, 4));
        CK(hipMemcpy(x, hx.data(), ne * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(gamma, hg.data(), C * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(beta, hb.data(), C * 4, hipMemcpyHostToDevice)); CK(hipMemset(bar, 0, 8)); CK(hipMemset(err, 0, 4));
        hipStream_t st; CK(hipStreamCreate(&st)); hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const dim3 grid(nchunks, N);
        auto runA = [&]() {
            hipLaunchKernelGGL(k_stats, grid, dim3(256), 0, st, x, V, C, chunkV, nchunks, partial);
            hipLaunchKernelGGL(k_final, dim3(N * C), dim3(256), 0, st, partial, C, nchunks, V, stats);
            hipLaunchKernelGGL(k_apply, grid, dim3(256), 0, st, x, y0, V, C, chunkV, stats, gamma, beta);
        };
        auto runB = [&]() { hipLaunchKernelGGL(k_fused, grid, dim3(256), 0, st, x, y1, V, C, chunkV, nchunks, partial, stats, gamma, beta, bar, err); };
        float tA = 0, tB = 0; const int IT = 300;
        for (int rep = 0; rep < 2; ++rep) {
            for (int i = 0; i < 20; ++i) runA();
            CK(hipEventRecord(e0, st)); for (int i = 0; i < IT; ++i) runA(); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&tA, e0, e1));
            for (int i = 0; i < 20; ++i) runB();
            CK(hipEventRecord(e0, st)); for (int i = 0; i < IT; ++i) runB(); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&tB, e0, e1));
        }
        std::vector<bf16> h0(ne), h1(ne); int herr = 0;
        CK(hipMemcpy(h0.data(), y0, ne * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), y1, ne * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        float md = 0.f;
        for (size_t i = 0; i < ne; ++i) { unsigned a = (unsigned)h0[i] << 16, b = (unsigned)h1[i] << 16; float fa, fb; memcpy(&fa, &a, 4); memcpy(&fb, &b, 4); md = fmaxf(md, fabsf(fa - fb)); }
        char name[64]; snprintf(name, sizeof name, "%d x %lld x %d (%d blk)", N, V, C, N * nchunks);
        printf("%-22s %8.2f us %8.2f us %10.2f %8.1e%s\n", name, tA * 1e3f / IT, tB * 1e3f / IT, tB / tA, md, herr ? "  BARRIER TIMEOUT" : "");
        hipFree(x); hipFree(y0); hipFree(y1); hipFree(partial); hipFree(stats); hipFree(gamma); hipFree(beta); hipFree(bar); hipFree(err);
    }
    return 0;
}
