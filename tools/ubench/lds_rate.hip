// Micro-benchmark: LDS read throughput of a CU by access width and address pattern, and the same gathers with the two VALU
// instructions of k_s3_score_bl around them.  1024 threads per workgroup (4 waves per SIMD), one workgroup per CU.
// build: hipcc --offload-arch=gfx950 -O3 lds_rate.hip -o lds_rate
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef unsigned int u32;
typedef u32 u32x2 __attribute__((ext_vector_type(2)));
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

// PAT: 0 = lane * width (linear, conflict free), 1 = b32 at (19 * (lane & 31) + s) * 4 (the score kernel's pattern, s uniform),
//      2 = b32 all lanes the same address (broadcast), 3 = b32 two halves with different s
template <int W, int PAT, int VALU>
__global__ __launch_bounds__(1024) void k(u32* out, int iters) {
    extern __shared__ char lds[];
    for (int i = threadIdx.x; i < 16384; i += 1024) ((u32*)lds)[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    u32 addr;
    if (PAT == 0) addr = lane * W;
    else if (PAT == 1) addr = (19 * (lane & 31) + 3) * 4;
    else if (PAT == 2) addr = 64;
    else addr = (19 * (lane & 31) + (lane >> 5) * 7) * 4;
    u32 acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0, acc4 = 0, acc5 = 0, acc6 = 0, acc7 = 0;
    u32 xb = addr | (addr << 16), xw = 0x0c080400u;
    for (int it = 0; it < iters; ++it) {
        if (VALU == 0) {
            if (W == 4) {
                u32 t0, t1, t2, t3, t4, t5, t6, t7;
                asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:4096\n ds_read_b32 %2, %8 offset:8192\n ds_read_b32 %3, %8 offset:12288\n"
                             "ds_read_b32 %4, %8 offset:16384\n ds_read_b32 %5, %8 offset:20480\n ds_read_b32 %6, %8 offset:24576\n ds_read_b32 %7, %8 offset:28672\n"
                             "s_waitcnt lgkmcnt(0)\n"
                             : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7) : "v"(addr));
                acc0 ^= t0 ^ t1 ^ t2 ^ t3 ^ t4 ^ t5 ^ t6 ^ t7;
            } else if (W == 2) {
                u32 t0, t1, t2, t3, t4, t5, t6, t7;
                asm volatile("ds_read_u16 %0, %8\n ds_read_u16 %1, %8 offset:4096\n ds_read_u16 %2, %8 offset:8192\n ds_read_u16 %3, %8 offset:12288\n"
                             "ds_read_u16 %4, %8 offset:16384\n ds_read_u16 %5, %8 offset:20480\n ds_read_u16 %6, %8 offset:24576\n ds_read_u16 %7, %8 offset:28672\n"
                             "s_waitcnt lgkmcnt(0)\n"
                             : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7) : "v"(addr));
                acc0 ^= t0 ^ t1 ^ t2 ^ t3 ^ t4 ^ t5 ^ t6 ^ t7;
            } else if (W == 8) {
                u32x2 t0, t1, t2, t3, t4, t5, t6, t7;
                asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:4096\n ds_read_b64 %2, %8 offset:8192\n ds_read_b64 %3, %8 offset:12288\n"
                             "ds_read_b64 %4, %8 offset:16384\n ds_read_b64 %5, %8 offset:20480\n ds_read_b64 %6, %8 offset:24576\n ds_read_b64 %7, %8 offset:28672\n"
                             "s_waitcnt lgkmcnt(0)\n"
                             : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7) : "v"(addr));
                acc0 ^= t0.x ^ t1.y ^ t2.x ^ t3.y ^ t4.x ^ t5.y ^ t6.x ^ t7.y;
            } else {
                u32x4 t0, t1, t2, t3, t4, t5, t6, t7;
                asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:4096\n ds_read_b128 %2, %8 offset:8192\n ds_read_b128 %3, %8 offset:12288\n"
                             "ds_read_b128 %4, %8 offset:16384\n ds_read_b128 %5, %8 offset:20480\n ds_read_b128 %6, %8 offset:24576\n ds_read_b128 %7, %8 offset:28672\n"
                             "s_waitcnt lgkmcnt(0)\n"
                             : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7) : "v"(addr));
                acc0 ^= t0.x ^ t1.y ^ t2.z ^ t3.w ^ t4.x ^ t5.y ^ t6.z ^ t7.w;
            }
        } else {
            // the score kernel's block of eight: sdwa add, gather, add
            u32 t0, t1, t2, t3, t4, t5, t6, t7;
            asm volatile("v_add_u32_sdwa %0, %16, %17 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:BYTE_0\n"
                         "v_add_u32_sdwa %1, %16, %17 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:BYTE_1\n"
                         "v_add_u32_sdwa %2, %16, %17 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:BYTE_2\n"
                         "v_add_u32_sdwa %3, %16, %17 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:BYTE_3\n"
                         "v_add_u32_sdwa %4, %16, %17 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:BYTE_0\n"
                         "v_add_u32_sdwa %5, %16, %17 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:BYTE_1\n"
                         "v_add_u32_sdwa %6, %16, %17 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:BYTE_2\n"
                         "v_add_u32_sdwa %7, %16, %17 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:BYTE_3\n"
                         "ds_read_b32 %0, %0\n ds_read_b32 %1, %1 offset:4096\n ds_read_b32 %2, %2 offset:8192\n ds_read_b32 %3, %3 offset:12288\n"
                         "ds_read_b32 %4, %4 offset:16384\n ds_read_b32 %5, %5 offset:20480\n ds_read_b32 %6, %6 offset:24576\n ds_read_b32 %7, %7 offset:28672\n"
                         "s_waitcnt lgkmcnt(4)\n v_add_u32 %8, %8, %0\n v_add_u32 %9, %9, %1\n v_add_u32 %10, %10, %2\n v_add_u32 %11, %11, %3\n"
                         "s_waitcnt lgkmcnt(0)\n v_add_u32 %12, %12, %4\n v_add_u32 %13, %13, %5\n v_add_u32 %14, %14, %6\n v_add_u32 %15, %15, %7\n"
                         : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7), "+v"(acc0), "+v"(acc1), "+v"(acc2),
                           "+v"(acc3), "+v"(acc4), "+v"(acc5), "+v"(acc6), "+v"(acc7)
                         : "v"(xb), "v"(xw));
        }
    }
    out[blockIdx.x * 1024 + threadIdx.x] = acc0 + acc1 + acc2 + acc3 + acc4 + acc5 + acc6 + acc7;
}

template <int W, int PAT, int VALU>
void run(const char* name) {
    u32* out;
    hipMalloc(&out, 256 * 1024 * 4);
    hipFuncSetAttribute((const void*)k<W, PAT, VALU>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    hipLaunchKernelGGL((k<W, PAT, VALU>), dim3(256), dim3(1024), 65536, 0, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<W, PAT, VALU>), dim3(256), dim3(1024), 65536, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double winst = 16.0 * 8 * iters;                       // wave-level LDS instructions per CU
    printf("%-44s %8.3f ms -> %5.2f ns = %5.2f CU cycles @2.4GHz per wave instruction, %6.1f B/clk/CU\n", name, ms, ms * 1e6 / winst,
           ms * 1e-3 * 2.4e9 / winst, 64.0 * W / (ms * 1e-3 * 2.4e9 / winst));
    hipFree(out);
}

int main() {
    run<4, 0, 0>("ds_read_b32 linear");
    run<4, 1, 0>("ds_read_b32 stride 19 words, 32 lanes x 2");
    run<4, 2, 0>("ds_read_b32 broadcast");
    run<4, 3, 0>("ds_read_b32 stride 19, halves differ");
    run<2, 0, 0>("ds_read_u16 linear");
    run<8, 0, 0>("ds_read_b64 linear");
    run<16, 0, 0>("ds_read_b128 linear");
    run<4, 1, 1>("sdwa add + ds_read_b32 (stride 19) + add");
    run<4, 3, 1>("sdwa add + ds_read_b32 (halves differ) + add");
    return 0;
}
