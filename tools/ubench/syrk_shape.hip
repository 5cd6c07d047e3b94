// Micro-benchmark: what a workgroup SHAPE of the S3 contraction can reach with no real data -- per k-step (64 bins) a wave issues
// TA x TB fp4 MFMAs (32x32x64, its own accumulators), reads its TA + TB operand tiles with ds_read_b128 (double-buffered in
// registers, requested one k-step ahead) and its share of the workgroup's LDS-DMA pieces (1 KiB global_load_lds_dwordx4 from an
// L2-resident buffer), everything interleaved between the MFMAs; one raw barrier per two k-steps.  Shapes:
//     8 waves x (3 x 3 tiles)  = k_s3_syrk_fp4 (2 waves per SIMD, 192 x 384 cells, 18 tiles per k-step)
//     4 waves x (4 x 6 tiles)  = one wave per SIMD, 384 accumulator registers (256 x 384 cells, 20 tiles per k-step)
//     4 waves x (4 x 4 tiles)  = one wave per SIMD, 256 x 256 cells, 16 tiles per k-step
// Prints the matrix rate reached (PFLOP/s-equivalent) and the cycles per MFMA at the measured clock.
// build: hipcc --offload-arch=gfx950 -O3 syrk_shape.hip -o syrk_shape
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned int u32;
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

#include <utility>
template <int TA, int TB, int NW, int NL, bool MFMA, bool LOADS>
struct Wave {
    v16f acc[TA][TB];
    v4i oa[2][TA], ob[2][TB];
    u32 lds0;
    const char* gsrc;
    char* ldst;
    static constexpr int NM = TA * TB, NR = TA + TB;
    static constexpr int STRIDE = (NM - NR) / NL > 0 ? (NM - NR) / NL : 1;

    // slot I of k-step parity KS: MFMA I, then one non-matrix instruction: the reads of the next k-step first, then the DMA pieces
    template <int KS, int I>
    __device__ __forceinline__ void slot(int piece0) {
        constexpr int a = I / TB, b = I % TB, cur = KS, nxt = KS ^ 1;
        if (MFMA)
            acc[a][b] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(__builtin_shufflevector(oa[cur][a], oa[cur][a], 0, 1, 2, 3, -1, -1, -1, -1),
                                                                        __builtin_shufflevector(ob[cur][b], ob[cur][b], 0, 1, 2, 3, -1, -1, -1, -1),
                                                                        acc[a][b], 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        __builtin_amdgcn_sched_barrier(0);
        if (LOADS) {
            if constexpr (I < TA)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(oa[nxt][I]) : "v"(lds0), "n"(I * 1024) : "memory");
            else if constexpr (I < NR)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ob[nxt][I - TA]) : "v"(lds0), "n"(I * 1024) : "memory");
            else if constexpr ((I - NR) % STRIDE == 0 && (I - NR) / STRIDE < NL) {
                constexpr int j = (I - NR) / STRIDE;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + (long)((piece0 + j) & 31) * 8192),
                                                 (__attribute__((address_space(3))) void*)(ldst + j * 1024), 16, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    template <int KS, int... I>
    __device__ __forceinline__ void kstep(int piece0, std::integer_sequence<int, I...>) {
        (slot<KS, I>(piece0), ...);
        if (LOADS) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int t = 0; t < TA; ++t) asm volatile("" : "+v"(oa[KS ^ 1][t]));
#pragma unroll
            for (int t = 0; t < TB; ++t) asm volatile("" : "+v"(ob[KS ^ 1][t]));
        }
    }
};

template <int TA, int TB, int NW, int NL, bool MFMA, bool LOADS>
__global__ __launch_bounds__(64 * NW, 1) void k(const char* __restrict__ src, u32* __restrict__ out, int iters, long src_bytes) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    Wave<TA, TB, NW, NL, MFMA, LOADS> W;
#pragma unroll
    for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int b = 0; b < TB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) W.acc[a][b][r] = 0.f;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int t = 0; t < TA; ++t) W.oa[p][t] = v4i{0x22222222, 0, 0x22222222, 0};
#pragma unroll
        for (int t = 0; t < TB; ++t) W.ob[p][t] = v4i{0x22222222, 0x22222222, 0, 0};
    }
    W.lds0 = (u32)(uintptr_t)smem + (u32)lane * 16 + (u32)w * (TA + TB) * 1024;
    W.gsrc = src + ((long)blockIdx.x * 65536) % (src_bytes - 262144) + lane * 16;
    W.ldst = smem + 65536 + w * NL * 1024;
    for (int it = 0; it < iters; ++it) {
        W.template kstep<0>(it * 2 * NL, std::make_integer_sequence<int, TA * TB>{});
        W.template kstep<1>(it * 2 * NL + NL, std::make_integer_sequence<int, TA * TB>{});
        if (LOADS) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");
            __builtin_amdgcn_s_barrier();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int b = 0; b < TB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += W.acc[a][b][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)s;
}

template <int TA, int TB, int NW, int NL, bool MFMA, bool LOADS>
void run(const char* name, const char* src, u32* out, long src_bytes) {
    auto fn = k<TA, TB, NW, NL, MFMA, LOADS>;
    const size_t lds = 144 * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, blocks = 256;
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(64 * NW), lds, 0, src, out, 50, src_bytes);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(fn, dim3(blocks), dim3(64 * NW), lds, 0, src, out, iters, src_bytes);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    hipError_t err = hipGetLastError();
    const double mf = (double)blocks * NW * iters * 2 * TA * TB;        // MFMAs
    const double per_simd = mf / 1024.0;
    printf("%-44s %8.3f ms  %6.2f PFLOP/s-eq  %6.1f ns per k-step  %5.1f cycles per MFMA and SIMD at 2.3 GHz  (%s)\n", name, best,
           MFMA ? mf * 131072.0 / (best * 1e-3) / 1e15 : 0.0, best * 1e6 / (iters * 2), MFMA ? best * 1e-3 * 2.3e9 / per_simd : 0.0,
           hipGetErrorString(err));
}

int main() {
    const long src_bytes = 8L << 20;
    char* src;
    u32* out;
    hipMalloc(&src, src_bytes);
    hipMemset(src, 0x22, src_bytes);
    hipMalloc(&out, 256 * 512 * sizeof(u32));
    run<3, 3, 8, 3, true, false>("8 waves x 3x3, MFMAs only", src, out, src_bytes);
    run<3, 3, 8, 3, false, true>("8 waves x 3x3, reads + DMA only", src, out, src_bytes);
    run<3, 3, 8, 3, true, true>("8 waves x 3x3, all (2.25 pieces per k-step: 3)", src, out, src_bytes);
    run<4, 6, 4, 5, true, false>("4 waves x 4x6, MFMAs only", src, out, src_bytes);
    run<4, 6, 4, 5, false, true>("4 waves x 4x6, reads + DMA only", src, out, src_bytes);
    run<4, 6, 4, 5, true, true>("4 waves x 4x6, all (5 pieces per k-step)", src, out, src_bytes);
    run<4, 4, 4, 4, true, false>("4 waves x 4x4, MFMAs only", src, out, src_bytes);
    run<4, 4, 4, 4, true, true>("4 waves x 4x4, all (4 pieces per k-step)", src, out, src_bytes);
    run<3, 6, 4, 5, true, true>("4 waves x 3x6, all (4.5 pieces per k-step: 5)", src, out, src_bytes);
    return 0;
}
