// pmc_calib.hip -- known-byte-count probes for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 (round 6, judge item 8).
// MI355X_MICROARCH.md states the x2 correction of FETCH_SIZE for 16 B/lane streaming reads only and calls every other access width and
// WRITE_SIZE uncalibrated.  Each kernel below moves exactly BYTES bytes (1 GiB by default: beyond the 256 MiB Infinity Cache) with ONE
// access width the library uses:   rd16 / rd8 / rd4 / rd2   global loads of 16 / 8 / 4 / 2 bytes per lane (coalesced stream)
//                                  rd16_lds            buffer_load_dwordx4 ... lds (LDS-DMA, the conv / weight-gradient loaders)
//                                  rd16_strided        16-byte pieces at a 64-byte pitch (a 32-channel chunk of a 128-byte voxel row)
//                                  wr16 / wr8 / wr4    global stores of 16 / 8 / 4 bytes per lane
// build + run (GPU box):  bash tools/probes/pmc_calib.sh  ->  gpurun_out/pmc_calib.txt (factor = true bytes / reported bytes)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

typedef int i32x4 __attribute__((ext_vector_type(4)));

template <typename V> __global__ void __launch_bounds__(256) rd_kernel(const V* __restrict__ p, long long n, unsigned* __restrict__ sink) {
    unsigned acc = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        V v = p[i];
        unsigned w[(sizeof(V) + 3) / 4] = {};
        __builtin_memcpy(w, &v, sizeof(V));
#pragma unroll
        for (unsigned k = 0; k < (sizeof(V) + 3) / 4; ++k) acc += w[k];
    }
    if (acc == 0x12345678u) sink[0] = acc;            // (data-dependent: keeps the loads)
}
__global__ void __launch_bounds__(256) rd16_strided_kernel(const uint4* __restrict__ p, long long nrows, int piece, unsigned* __restrict__ sink) {
    unsigned acc = 0;                                  // rows of 64 bytes (4 pieces): every row contributes ONE 16-byte piece
    for (long long r = (long long)blockIdx.x * 256 + threadIdx.x; r < nrows; r += (long long)gridDim.x * 256) { const uint4 v = p[r * 4 + piece]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void __launch_bounds__(256) rd16_lds_kernel(const void* __restrict__ p, long long nbytes, unsigned* __restrict__ sink) {
    __shared__ __attribute__((aligned(16))) unsigned char buf[4 * 1024];
    const unsigned lds = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)buf + (threadIdx.x >> 6) * 1024);
    const unsigned long long base = (unsigned long long)p;
    const long long chunk = 1ll << 30;                 // buffer resources address 31 bits: 1 GiB windows
    for (long long w0 = 0; w0 < nbytes; w0 += chunk) {
        const unsigned long long b = base + w0;
        i32x4 rs; rs.x = (int)(unsigned)b; rs.y = (int)((unsigned)(b >> 32) & 0xffffu); rs.z = 0x7fffffff; rs.w = 0x00020000;
        const long long lim = nbytes - w0 < chunk ? nbytes - w0 : chunk;
        for (long long o = ((long long)blockIdx.x * 256 + threadIdx.x) * 16; o < lim; o += (long long)gridDim.x * 256 * 16) {
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" :: "s"(lds), "v"((unsigned)o), "s"(rs) : "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (buf[threadIdx.x] == 0x77 && nbytes < 0) sink[0] = 1;
}
template <typename V> __global__ void __launch_bounds__(256) wr_kernel(V* __restrict__ p, long long n, V val) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) p[i] = val;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char** argv) {
    const long long bytes = (argc > 1 ? atoll(argv[1]) : 1024ll) << 20;
    void* buf; unsigned* sink;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 1, bytes)); CK(hipDeviceSynchronize());
    const int blocks = 256 * 8;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(rd_kernel<uint4>, dim3(blocks), dim3(256), 0, 0, (const uint4*)buf, bytes / 16, sink);
        hipLaunchKernelGGL(rd_kernel<uint2>, dim3(blocks), dim3(256), 0, 0, (const uint2*)buf, bytes / 8, sink);
        hipLaunchKernelGGL(rd_kernel<unsigned>, dim3(blocks), dim3(256), 0, 0, (const unsigned*)buf, bytes / 4, sink);
        hipLaunchKernelGGL(rd_kernel<unsigned short>, dim3(blocks), dim3(256), 0, 0, (const unsigned short*)buf, bytes / 2, sink);
        hipLaunchKernelGGL(rd16_lds_kernel, dim3(blocks), dim3(256), 0, 0, buf, bytes, sink);
        hipLaunchKernelGGL(rd16_strided_kernel, dim3(blocks), dim3(256), 0, 0, (const uint4*)buf, bytes / 64, 1, sink);
        hipLaunchKernelGGL(wr_kernel<uint4>, dim3(blocks), dim3(256), 0, 0, (uint4*)buf, bytes / 16, make_uint4(1, 2, 3, 4));
        hipLaunchKernelGGL(wr_kernel<uint2>, dim3(blocks), dim3(256), 0, 0, (uint2*)buf, bytes / 8, make_uint2(1, 2));
        hipLaunchKernelGGL(wr_kernel<unsigned>, dim3(blocks), dim3(256), 0, 0, (unsigned*)buf, bytes / 4, 7u);
        CK(hipDeviceSynchronize());
    }
    printf("pmc_calib: every kernel moved %lld bytes (strided: %lld useful bytes out of %lld touched rows x 64 B)\n", bytes, bytes / 4, bytes / 64);
    return 0;
}
