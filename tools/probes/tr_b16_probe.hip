// Probe of ds_read_b64_tr_b16 on gfx950: LDS holds u16 ids (id = byte_offset / 2); each lane passes an 8-byte-aligned
// address; print which ids every lane receives.   hipcc --offload-arch=gfx950 -O2 tr_b16_probe.hip -o tr_probe && ./tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void probe(const int* __restrict__ lane_addr, uint16_t* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const unsigned base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)lds;
    const unsigned a = base + lane_addr[threadIdx.x];
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    out[threadIdx.x * 4 + 0] = v[0] & 0xffff; out[threadIdx.x * 4 + 1] = v[0] >> 16;
    out[threadIdx.x * 4 + 2] = v[1] & 0xffff; out[threadIdx.x * 4 + 3] = v[1] >> 16;
}
int main() {
    int h_addr[64]; uint16_t h_out[256];
    int *d_addr; uint16_t* d_out;
    hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
    for (int pat = 0; pat < 3; ++pat) {
        for (int l = 0; l < 64; ++l) {
            const int g = l >> 4, i = l & 15;
            if (pat == 0) h_addr[l] = g * 1024 + (i >> 2) * 64 + (i & 3) * 8;      // rows of 64 B, lane i -> row i>>2, piece i&3
            else if (pat == 1) h_addr[l] = g * 1024 + i * 64;                       // lane i -> row i, piece 0
            else h_addr[l] = g * 1024 + (i & 3) * 64 + (i >> 2) * 8;               // lane i -> row i&3, piece i>>2
        }
        hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_addr, d_out);
        hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
        printf("pattern %d (ids are u16 indices; row = id/32 within group of 512, col = id%%32)\n", pat);
        for (int l = 0; l < 64; ++l) {
            printf("  lane %2d addr %4d(id %4d):", l, h_addr[l], h_addr[l] / 2);
            for (int j = 0; j < 4; ++j) printf(" %4d[r%d c%2d]", h_out[l * 4 + j], (h_out[l * 4 + j] % 512) / 32, h_out[l * 4 + j] % 32);
            printf("\n");
            if (l == 19) { printf("  ...\n"); l = 47; }
        }
    }
    return 0;
}
