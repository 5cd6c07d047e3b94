// Read-only HBM streaming ceiling: every lane xors 16-byte loads of a 12.7 GB buffer (same size as the state matrix).
// build: hipcc --offload-arch=gfx950 -O3 read_bw.hip -o read_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32;

template <int UNROLL>
__global__ __launch_bounds__(256) void k_read(const uint4* __restrict__ p, long n16, u32* out) {
    u32 acc = 0;
    const long stride = (long)gridDim.x * 256 * UNROLL;
    for (long i = (long)blockIdx.x * 256 * UNROLL + threadIdx.x; i < n16; i += stride) {
        uint4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = (i + 256L * u < n16) ? p[i + 256L * u] : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

__global__ __launch_bounds__(256) void k_copy(const uint4* __restrict__ p, uint4* __restrict__ q, long n16) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) q[i] = p[i];
}

int main() {
    const long bytes = 15000000L * 848;
    const long n16 = bytes / 16;
    uint4 *p, *q; u32* out;
    hipMalloc(&p, bytes); hipMalloc(&q, bytes); hipMalloc(&out, 4);
    hipMemset(p, 1, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int bpc = 2; bpc <= 16; bpc *= 2) {
        const int grid = 256 * bpc;
        float best4 = 1e9, best8 = 1e9, bestc = 1e9;
        for (int it = 0; it < 6; ++it) {
            float ms;
            hipEventRecord(e0); hipLaunchKernelGGL(k_read<4>, dim3(grid), dim3(256), 0, 0, p, n16, out); hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1); if (it && ms < best4) best4 = ms;
            hipEventRecord(e0); hipLaunchKernelGGL(k_read<8>, dim3(grid), dim3(256), 0, 0, p, n16, out); hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1); if (it && ms < best8) best8 = ms;
            hipEventRecord(e0); hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, p, q, n16); hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1); if (it && ms < bestc) bestc = ms;
        }
        printf("blocks/CU %2d: read x4 %.3f ms = %.0f GB/s | read x8 %.3f ms = %.0f GB/s | copy %.3f ms = %.0f GB/s (read+write)\n", bpc, best4,
               bytes / best4 / 1e6, best8, bytes / best8 / 1e6, bestc, 2.0 * bytes / bestc / 1e6);
    }
    return 0;
}
