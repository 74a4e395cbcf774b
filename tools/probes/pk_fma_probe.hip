// Minimal probe for the round-4 finding -- it does NOT reproduce the effect (0 mismatches next to either aggressor below):
// the library's own kernels are needed as neighbours (build with `make -C prostatemr_3d-cad-cspca_amd/csrc clean && make ... NOPK=`, then
// `VICTIM=stemnostats python tools/dbg/stress_posterior.py part:conv3`: 48 of 60 replays wrong).  Kept as the starting point of a reproducer: do packed fp32 FMAs (v_pk_fma_f32) return the same bits as scalar FMAs
// (v_fmac_f32) when the wave shares the GPU with an MFMA kernel that keeps its accumulators in AGPRs?
//   hipcc --offload-arch=gfx950 -O3 -o pk_fma_probe tools/probes/pk_fma_probe.hip && ./pk_fma_probe
// victim: every lane accumulates 8 sums over LDS-resident data twice -- once with float2 math (the compiler emits v_pk_fma_f32), once
// with inline-asm v_fmac_f32 -- and counts lanes whose two results differ in any bit.  aggressor: an MFMA loop with 16 accumulator tiles.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f2 __attribute__((ext_vector_type(2)));

__global__ void __launch_bounds__(256) victim(const float* __restrict__ xin, const float* __restrict__ win, unsigned* __restrict__ bad, int iters) {
    __shared__ __attribute__((aligned(16))) float Ws[27 * 32];
    __shared__ float Xs[10 * 34 * 3];
    for (int e = threadIdx.x; e < 27 * 32; e += 256) Ws[e] = win[e];
    unsigned nb = 0;
    for (int it = 0; it < iters; ++it) {
        __syncthreads();
        for (int e = threadIdx.x; e < 10 * 34 * 3; e += 256) Xs[e] = xin[(e + it * 131 + blockIdx.x * 17) & 65535];
        __syncthreads();
        const int vg = threadIdx.x >> 2, vr = vg >> 3, vc = (vg & 7) * 4, oc0 = (threadIdx.x & 3) * 8;
        f2 accp[4][4]; float accs[4][8];
        for (int v = 0; v < 4; ++v) for (int j = 0; j < 4; ++j) { accp[v][j] = (f2){0.f, 0.f}; accs[v][2 * j] = 0.f; accs[v][2 * j + 1] = 0.f; }
        for (int ab = 0; ab < 3; ++ab) {
            const float* xr = Xs + ((ab + vr) * 34 + vc) * 3;
            for (int ci = 0; ci < 3; ++ci) {
                float xv[6];
#pragma unroll
                for (int q = 0; q < 6; ++q) xv[q] = xr[q * 3 + ci];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float* wk = Ws + ((ab * 3 + c) * 3 + ci) * 32 + oc0;
                    const float4 w0 = *reinterpret_cast<const float4*>(wk), w1 = *reinterpret_cast<const float4*>(wk + 4);
                    const f2 wp[4] = {(f2){w0.x, w0.y}, (f2){w0.z, w0.w}, (f2){w1.x, w1.y}, (f2){w1.z, w1.w}};
                    const float ws_[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const f2 xx = (f2){xv[v + c], xv[v + c]};
#pragma unroll
                        for (int j = 0; j < 4; ++j) accp[v][j] = __builtin_elementwise_fma(xx, wp[j], accp[v][j]);
#pragma unroll
                        for (int j = 0; j < 8; ++j) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(accs[v][j]) : "v"(xv[v + c]), "v"(ws_[j]));
                    }
                }
            }
        }
        for (int v = 0; v < 4; ++v) for (int j = 0; j < 4; ++j)
            nb += (__float_as_uint(accp[v][j].x) != __float_as_uint(accs[v][2 * j])) | (__float_as_uint(accp[v][j].y) != __float_as_uint(accs[v][2 * j + 1]));
    }
    if (nb) atomicAdd(bad, nb);
}

__global__ void __launch_bounds__(256) aggressor(const bf16x8* __restrict__ a, float* __restrict__ out, int iters) {
    f32x16 acc[16];
    for (int t = 0; t < 16; ++t) for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    bf16x8 x = a[threadIdx.x], y = a[threadIdx.x + 256];
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[t], 0, 0, 0);
    float s = 0.f;
    for (int t = 0; t < 16; ++t) for (int e = 0; e < 16; ++e) s += acc[t][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// second aggressor: AGPRs as a spill space (v_accvgpr_write / v_accvgpr_read traffic between MFMAs), as the compiler does in conv_pw
__global__ void __launch_bounds__(256) aggressor2(const bf16x8* __restrict__ a, float* __restrict__ out, int iters) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 acc[8];
    for (int t = 0; t < 8; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 x = a[threadIdx.x], y = a[threadIdx.x + 256];
    float keep = (float)threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, acc[t], 0, 0, 0);
        asm volatile("v_accvgpr_write_b32 a0, %1\n v_accvgpr_write_b32 a1, %1\n v_accvgpr_write_b32 a2, %1\n v_accvgpr_write_b32 a3, %1\n s_nop 2\n"
                     "v_accvgpr_read_b32 %0, a0\n v_accvgpr_read_b32 %0, a1\n v_accvgpr_read_b32 %0, a2\n v_accvgpr_read_b32 %0, a3\n"
                     : "=v"(keep) : "v"(keep) : "a0", "a1", "a2", "a3");
    }
    float s = keep;
    for (int t = 0; t < 8; ++t) for (int e = 0; e < 4; ++e) s += acc[t][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    float *x, *w, *o; unsigned* bad; bf16x8* a;
    hipMalloc(&x, 65536 * 4); hipMalloc(&w, 27 * 32 * 4); hipMalloc(&o, 2048 * 256 * 4); hipMalloc(&bad, 4); hipMalloc(&a, 512 * 16);
    std::vector<float> hx(65536), hw(27 * 32);
    for (size_t i = 0; i < hx.size(); ++i) hx[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = (float)((i * 40503u) % 1999) / 4000.f - 0.25f;
    hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice); hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    hipMemset(a, 0x3c, 512 * 16);
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    for (int mode = 0; mode < 3; ++mode) {
        unsigned total = 0; int hits = 0;
        for (int rep = 0; rep < 50; ++rep) {
            hipMemsetAsync(bad, 0, 4, s1); hipStreamSynchronize(s1);
            if (mode == 1) hipLaunchKernelGGL(aggressor, dim3(1024), dim3(256), 0, s2, a, o, 400);
            if (mode == 2) hipLaunchKernelGGL(aggressor2, dim3(2048), dim3(256), 0, s2, a, o, 1500);
            hipLaunchKernelGGL(victim, dim3(1024), dim3(256), 0, s1, x, w, bad, 40);
            hipDeviceSynchronize();
            unsigned h; hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost); total += h; hits += h != 0;
        }
        printf("%s: %u lanes with packed != scalar FMA results, in %d of 50 launches\n", mode == 0 ? "alone" : (mode == 1 ? "next to the MFMA kernel" : "next to the MFMA + accvgpr kernel"), total, hits);
    }
    return 0;
}
