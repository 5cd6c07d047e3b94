// Micro-benchmark: do VALU instructions issue under a running MFMA?  Per iteration a wave issues M fp4 MX MFMAs
// (independent accumulators) and V full-rate VALU ops; time per iteration for (M,0), (0,V), (M,V), interleaved or
// clustered, at 1 and 2 waves per SIMD.  build: hipcc --offload-arch=gfx950 -O3 mfma_valu.hip -o mfma_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32;
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

#define X8(r) "v_xor_b32 %" #r ", %" #r ", %8\n\t"
#define VALU16 X8(0) X8(1) X8(2) X8(3) X8(4) X8(5) X8(6) X8(7) X8(0) X8(1) X8(2) X8(3) X8(4) X8(5) X8(6) X8(7)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int MODE>   // int8 32x32x32 MFMA: 0 alone, 2 followed by 144 VALU, 3 interleaved
__global__ __launch_bounds__(256) void ki(u32* out, int iters, u32 seed) {
    u32 a0 = seed * (threadIdx.x + 1), a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19, c = seed;
    v16i acc[9];
    for (int i = 0; i < 9; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0;
    v4i fa = {(int)a0, (int)a1, (int)a2, (int)a3}, fb = {(int)a4, (int)a5, (int)a6, (int)a7};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb, acc[i], 0, 0, 0);
            if (MODE == 3)
                asm volatile(VALU16 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        }
        if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 9; ++i)
                asm volatile(VALU16 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        }
    }
    int s = 0;
    for (int i = 0; i < 9; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (u32)s;
}

template <int MODE>   // 0: MFMA only, 1: VALU only, 2: 9 MFMA then 144 VALU, 3: interleaved 1 MFMA : 16 VALU
__global__ __launch_bounds__(256) void k(u32* out, int iters, u32 seed) {
    u32 a0 = seed * (threadIdx.x + 1), a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19, c = seed;
    v16f acc[9];
    for (int i = 0; i < 9; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    v8i fa = {(int)a0, (int)a1, (int)a2, (int)a3, 0, 0, 0, 0}, fb = {(int)a4, (int)a5, (int)a6, (int)a7, 0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            if (MODE == 0 || MODE == 2 || MODE == 3)
                acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa, fb, acc[i], 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            if (MODE == 3)
                asm volatile(VALU16 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        }
        if (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int i = 0; i < 9; ++i)
                asm volatile(VALU16 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        }
    }
    float s = 0.f;
    for (int i = 0; i < 9; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (u32)s;
}

template <int MODE, bool I8 = false>
void run(const char* name) {
    u32* out;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(u32));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    for (int wps = 1; wps <= 4; wps *= 2) {          // 256-thread blocks = one wave per SIMD each
        const int blocks = 256 * wps;
        auto fn = I8 ? ki<MODE> : k<MODE>;
        hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), 0, 0, out, 10, 3u);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), 0, 0, out, iters, 3u);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // SIMD-cycles per wave-iteration: time * clock / (iterations * waves per SIMD)
        printf("%-34s waves/SIMD %d: %8.3f ms -> %7.1f cycles per wave-iteration on its SIMD @2.4GHz (%.1f per SIMD-iteration-slot)\n", name, wps, ms,
               ms * 1e-3 * 2.4e9 / iters, ms * 1e-3 * 2.4e9 / iters / wps);
    }
    hipFree(out);
}

int main() {
    run<0>("9 MFMA fp4 32x32x64");
    run<1>("144 VALU (v_xor)");
    run<2>("9 MFMA then 144 VALU");
    run<3>("9 x (1 MFMA + 16 VALU)");
    run<0, true>("9 MFMA i8 32x32x32");
    run<2, true>("9 MFMA i8 then 144 VALU");
    run<3, true>("9 x (1 MFMA i8 + 16 VALU)");
    return 0;
}
