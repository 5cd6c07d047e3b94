// Micro-benchmark: issue rate and dependent-chain latency of v_fma_f64 (VGPR operands, and with one SGPR-pair operand
// as in k_score_s2_bin) vs waves per SIMD.  build: hipcc --offload-arch=gfx950 -O3 f64_rate.hip -o f64_rate
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP8(X) X X X X X X X X
// 8 independent accumulators
#define IND_V "v_fma_f64 %0, %8, %9, %0\n\tv_fma_f64 %1, %8, %9, %1\n\tv_fma_f64 %2, %8, %9, %2\n\tv_fma_f64 %3, %8, %9, %3\n\t" \
              "v_fma_f64 %4, %8, %9, %4\n\tv_fma_f64 %5, %8, %9, %5\n\tv_fma_f64 %6, %8, %9, %6\n\tv_fma_f64 %7, %8, %9, %7\n\t"
#define IND_S "v_fma_f64 %0, %8, %10, %0\n\tv_fma_f64 %1, %8, %10, %1\n\tv_fma_f64 %2, %8, %10, %2\n\tv_fma_f64 %3, %8, %10, %3\n\t" \
              "v_fma_f64 %4, %8, %10, %4\n\tv_fma_f64 %5, %8, %10, %5\n\tv_fma_f64 %6, %8, %10, %6\n\tv_fma_f64 %7, %8, %10, %7\n\t"
// one dependent chain
#define DEP_V "v_fma_f64 %0, %8, %9, %0\n\tv_fma_f64 %0, %8, %9, %0\n\tv_fma_f64 %0, %8, %9, %0\n\tv_fma_f64 %0, %8, %9, %0\n\t" \
              "v_fma_f64 %0, %8, %9, %0\n\tv_fma_f64 %0, %8, %9, %0\n\tv_fma_f64 %0, %8, %9, %0\n\tv_fma_f64 %0, %8, %9, %0\n\t"
#define DEP_S "v_fma_f64 %0, %8, %10, %0\n\tv_fma_f64 %0, %8, %10, %0\n\tv_fma_f64 %0, %8, %10, %0\n\tv_fma_f64 %0, %8, %10, %0\n\t" \
              "v_fma_f64 %0, %8, %10, %0\n\tv_fma_f64 %0, %8, %10, %0\n\tv_fma_f64 %0, %8, %10, %0\n\tv_fma_f64 %0, %8, %10, %0\n\t"
// two chains
#define DEP2_S "v_fma_f64 %0, %8, %10, %0\n\tv_fma_f64 %1, %8, %10, %1\n\tv_fma_f64 %0, %8, %10, %0\n\tv_fma_f64 %1, %8, %10, %1\n\t" \
               "v_fma_f64 %0, %8, %10, %0\n\tv_fma_f64 %1, %8, %10, %1\n\tv_fma_f64 %0, %8, %10, %0\n\tv_fma_f64 %1, %8, %10, %1\n\t"
#define DEP4_S "v_fma_f64 %0, %8, %10, %0\n\tv_fma_f64 %1, %8, %10, %1\n\tv_fma_f64 %2, %8, %10, %2\n\tv_fma_f64 %3, %8, %10, %3\n\t" \
               "v_fma_f64 %0, %8, %10, %0\n\tv_fma_f64 %1, %8, %10, %1\n\tv_fma_f64 %2, %8, %10, %2\n\tv_fma_f64 %3, %8, %10, %3\n\t"
#define ADD_V "v_add_f64 %0, %8, %0\n\tv_add_f64 %1, %8, %1\n\tv_add_f64 %2, %8, %2\n\tv_add_f64 %3, %8, %3\n\t" \
              "v_add_f64 %4, %8, %4\n\tv_add_f64 %5, %8, %5\n\tv_add_f64 %6, %8, %6\n\tv_add_f64 %7, %8, %7\n\t"
#define BODY(S) asm volatile(REP8(S) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y), "s"(sc))

template <int OP>
__global__ __launch_bounds__(256) void k(double* out, int iters, double seed, double sc) {
    double a0 = seed * threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const double x = 1.0000001, y = 0.999999;
    for (int it = 0; it < iters; ++it) {
        if (OP == 0) BODY(IND_V);
        if (OP == 1) BODY(IND_S);
        if (OP == 2) BODY(DEP_V);
        if (OP == 3) BODY(DEP_S);
        if (OP == 4) BODY(DEP2_S);
        if (OP == 5) BODY(DEP4_S);
        if (OP == 6) BODY(ADD_V);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int OP>
void run(const char* name) {
    double* out;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(double));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (int wps = 1; wps <= 8; wps *= 2) {
        const int blocks = 256 * wps;
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 10, 3.0, 0.5);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 3.0, 0.5);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double winst = (double)blocks * 4 * iters * 64;
        printf("%-22s waves/SIMD %d: %8.3f ms -> %6.2f SIMD-cycles per wave-inst @2.4GHz (%.1f TFLOP/s)\n", name, wps, ms,
               2.4e9 * 1024.0 / (winst / (ms * 1e-3)), winst * 64 * 2 / (ms * 1e-3) / 1e12);
    }
    hipFree(out);
}

int main() {
    run<0>("fma_f64 8 chains vgpr");
    run<1>("fma_f64 8 chains sgpr");
    run<2>("fma_f64 1 chain vgpr");
    run<3>("fma_f64 1 chain sgpr");
    run<4>("fma_f64 2 chains sgpr");
    run<5>("fma_f64 4 chains sgpr");
    run<6>("add_f64 8 chains");
    return 0;
}
