// Micro-benchmark: issue rate of the integer VALU ops the counting core is made of, vs waves per SIMD.
// Inline asm so the optimiser cannot fold the chains.  build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32;

#define REP8(X) X X X X X X X X
#define BODY(INSTR)                                                                                   \
    asm volatile(REP8(INSTR(%0, %1, %2, %3) INSTR(%1, %2, %3, %4) INSTR(%2, %3, %4, %5) INSTR(%3, %4, %5, %6) \
                      INSTR(%4, %5, %6, %7) INSTR(%5, %6, %7, %0) INSTR(%6, %7, %0, %1) INSTR(%7, %0, %1, %2)) \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7))

#define I_BFI(d, a, b, c) "v_bfi_b32 " #d ", " #a ", " #b ", " #c "\n\t"
#define I_BCNT(d, a, b, c) "v_bcnt_u32_b32 " #d ", " #a ", " #d "\n\t"
#define I_LSHL(d, a, b, c) "v_lshlrev_b32 " #d ", 3, " #a "\n\t"
#define I_BITOP3(d, a, b, c) "v_bitop3_b32 " #d ", " #a ", " #b ", " #c " bitop3:0x96\n\t"
#define I_FMA(d, a, b, c) "v_fma_f32 " #d ", " #a ", " #b ", " #c "\n\t"
#define I_AND(d, a, b, c) "v_and_b32 " #d ", " #a ", " #b "\n\t"
#define I_LSHLOR(d, a, b, c) "v_lshl_or_b32 " #d ", " #a ", 3, " #b "\n\t"
#define I_ADDDPP(d, a, b, c) "v_add_u32_dpp " #d ", " #a ", " #b " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
#define I_CNDMASK(d, a, b, c) "v_cndmask_b32 " #d ", " #a ", " #b ", vcc\n\t"

template <int OP>
__global__ __launch_bounds__(256) void k(u32* out, int iters, u32 seed) {
    u32 a0 = seed * (threadIdx.x + 1), a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    for (int it = 0; it < iters; ++it) {
        if (OP == 0) BODY(I_BFI);
        if (OP == 1) BODY(I_BCNT);
        if (OP == 2) BODY(I_LSHL);
        if (OP == 3) BODY(I_BITOP3);
        if (OP == 4) BODY(I_FMA);
        if (OP == 5) BODY(I_AND);
        if (OP == 6) BODY(I_LSHLOR);
        if (OP == 7) BODY(I_ADDDPP);
        if (OP == 8) BODY(I_CNDMASK);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

template <int OP>
void run(const char* name) {
    u32* out;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(u32));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    for (int wps = 1; wps <= 8; wps *= 2) {       // waves per SIMD = blocks per CU (256-thread blocks = 4 waves)
        const int blocks = 256 * wps;
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 10, 3u);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 3u);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double winst = (double)blocks * 4 * iters * 64;   // wave-instructions
        printf("%-10s waves/SIMD %d: %8.3f ms  %7.1f G wave-inst/s chip -> %.2f SIMD-cycles per wave-inst @2.4GHz\n", name, wps, ms,
               winst / (ms * 1e-3) / 1e9, 2.4e9 * 1024.0 / (winst / (ms * 1e-3)));
    }
    hipFree(out);
}

int main() {
    run<4>("fma_f32");
    run<0>("bfi");
    run<1>("bcnt");
    run<2>("lshlrev");
    run<3>("bitop3");
    run<5>("and");
    run<6>("lshl_or");
    run<7>("add_dpp");
    run<8>("cndmask");
    return 0;
}
