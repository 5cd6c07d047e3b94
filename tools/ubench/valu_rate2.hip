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


#define I_LSHR(d, a, b, c) "v_lshrrev_b32 " #d ", 3, " #a "\n\t"
#define I_ALIGNBIT(d, a, b, c) "v_alignbit_b32 " #d ", " #a ", " #b ", 5\n\t"
#define I_PERM(d, a, b, c) "v_perm_b32 " #d ", " #a ", " #b ", " #c "\n\t"
#define I_BFE(d, a, b, c) "v_bfe_u32 " #d ", " #a ", 3, 5\n\t"
#define I_ANDOR(d, a, b, c) "v_and_or_b32 " #d ", " #a ", " #b ", " #c "\n\t"
#define I_OR3(d, a, b, c) "v_or3_b32 " #d ", " #a ", " #b ", " #c "\n\t"
#define I_ADD(d, a, b, c) "v_add_u32 " #d ", " #a ", " #b "\n\t"
#define I_ADD3(d, a, b, c) "v_add3_u32 " #d ", " #a ", " #b ", " #c "\n\t"
#define I_MAD24(d, a, b, c) "v_mad_u32_u24 " #d ", " #a ", " #b ", " #c "\n\t"
#define I_MUL24(d, a, b, c) "v_mul_u32_u24 " #d ", " #a ", " #b "\n\t"
#define I_DOT4(d, a, b, c) "v_dot4_u32_u8 " #d ", " #a ", " #b ", " #c "\n\t"
#define I_XOR(d, a, b, c) "v_xor_b32 " #d ", " #a ", " #b "\n\t"
#define I_MOV(d, a, b, c) "v_mov_b32 " #d ", " #a "\n\t"
#define I_SAD(d, a, b, c) "v_sad_u8 " #d ", " #a ", " #b ", " #c "\n\t"
#define I_LSHLADD(d, a, b, c) "v_lshl_add_u32 " #d ", " #a ", 3, " #b "\n\t"
#define I_PKADD16(d, a, b, c) "v_pk_add_u16 " #d ", " #a ", " #b "\n\t"
#define I_PKLSHL16(d, a, b, c) "v_pk_lshlrev_b16 " #d ", 3, " #a "\n\t"
#define I_CND64(d, a, b, c) "v_cndmask_b32_e64 " #d ", " #a ", " #b ", s[4:5]\n\t"
#define I_MOVDPP(d, a, b, c) "v_mov_b32_dpp " #d ", " #a " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
#define I_SUBREV(d, a, b, c) "v_sub_u32 " #d ", " #a ", " #b "\n\t"
#define I_MULLO(d, a, b, c) "v_mul_lo_u32 " #d ", " #a ", " #b "\n\t"
#define I_CMPEQ(d, a, b, c) "v_cmp_eq_u32 vcc, " #a ", " #b "\n\t"
#define I_ADDC(d, a, b, c) "v_addc_co_u32 " #d ", vcc, " #a ", " #b ", vcc\n\t"
#define I_BFM(d, a, b, c) "v_bfm_b32 " #d ", " #a ", " #b "\n\t"
#define I_LSHL64(d, a, b, c) "v_lshlrev_b64 v[20:21], 3, v[22:23]\n\t"
#define I_NOT(d, a, b, c) "v_not_b32 " #d ", " #a "\n\t"
#define I_MAX(d, a, b, c) "v_max_u32 " #d ", " #a ", " #b "\n\t"
#define I_MIN3(d, a, b, c) "v_min3_u32 " #d ", " #a ", " #b ", " #c "\n\t"
#define I_CVTU8(d, a, b, c) "v_cvt_f32_ubyte1 " #d ", " #a "\n\t"
#define I_SDWA(d, a, b, c) "v_lshlrev_b32_sdwa " #d ", " #a ", " #b " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n\t"
#define I_ADDSDWA(d, a, b, c) "v_add_u32_sdwa " #d ", " #a ", " #b " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n\t"

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
        if (OP == 9) BODY(I_LSHR);
        if (OP == 10) BODY(I_ALIGNBIT);
        if (OP == 11) BODY(I_PERM);
        if (OP == 12) BODY(I_BFE);
        if (OP == 13) BODY(I_ANDOR);
        if (OP == 14) BODY(I_OR3);
        if (OP == 15) BODY(I_ADD);
        if (OP == 16) BODY(I_ADD3);
        if (OP == 17) BODY(I_MAD24);
        if (OP == 18) BODY(I_MUL24);
        if (OP == 19) BODY(I_DOT4);
        if (OP == 20) BODY(I_XOR);
        if (OP == 21) BODY(I_MOV);
        if (OP == 22) BODY(I_SAD);
        if (OP == 23) BODY(I_LSHLADD);
        if (OP == 24) BODY(I_PKADD16);
        if (OP == 25) BODY(I_PKLSHL16);
        if (OP == 26) BODY(I_CND64);
        if (OP == 27) BODY(I_MOVDPP);
        if (OP == 28) BODY(I_SUBREV);
        if (OP == 29) BODY(I_MULLO);
        if (OP == 30) BODY(I_CMPEQ);
        if (OP == 31) BODY(I_ADDC);
        if (OP == 32) BODY(I_BFM);
        if (OP == 33) BODY(I_NOT);
        if (OP == 34) BODY(I_MAX);
        if (OP == 35) BODY(I_MIN3);
        if (OP == 36) BODY(I_CVTU8);
        if (OP == 37) BODY(I_SDWA);
        if (OP == 38) BODY(I_ADDSDWA);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

template <int OP>
void run(const char* name) {
    u32* out;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(u32));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (int wps = 1; wps <= 4; wps *= 4) {       // waves per SIMD = blocks per CU (256-thread blocks = 4 waves)
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
    run<9>("lshr");
    run<10>("alignbit");
    run<11>("perm");
    run<12>("bfe");
    run<13>("andor");
    run<14>("or3");
    run<15>("add");
    run<16>("add3");
    run<17>("mad24");
    run<18>("mul24");
    run<19>("dot4");
    run<20>("xor");
    run<21>("mov");
    run<22>("sad");
    run<23>("lshladd");
    run<24>("pkadd16");
    run<25>("pklshl16");
    run<26>("cnd64");
    run<27>("movdpp");
    run<28>("subrev");
    run<29>("mullo");
    run<30>("cmpeq");
    run<31>("addc");
    run<32>("bfm");
    run<33>("not");
    run<34>("max");
    run<35>("min3");
    run<36>("cvtu8");
    run<37>("sdwa");
    run<38>("addsdwa");
    return 0;
}
