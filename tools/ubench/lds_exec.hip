// Micro-benchmark: does a ds_read_b32 with half of the wave masked off in EXEC cost the LDS pipe half the time?
// (The S3 score kernel could mask the half wave whose bin is in the modal state.)  1024 threads per workgroup, one per CU.
// Also the price of the scalar work that would set EXEC per gather (two s_bfe_i32 + branch on execz).
// build: hipcc --offload-arch=gfx950 -O3 lds_exec.hip -o lds_exec
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef unsigned int u32;

// MODE 0: all 64 lanes; 1: low 32 lanes; 2: halves alternate from read to read; 3: low 16 lanes; 4: EXEC from two s_bfe_i32 of a
// mask word per read (all bits set: nothing is skipped, the price of the decision); 5: the same with a mask that drops ~half of the halves
// and skips the read when EXEC is empty
template <int MODE>
__global__ __launch_bounds__(1024) void k(u32* out, int iters, u32 m_lo, u32 m_hi) {
    extern __shared__ char lds[];
    for (int i = threadIdx.x; i < 16384; i += 1024) ((u32*)lds)[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const u32 addr = (19 * (lane & 31) + 3) * 4;
    u32 acc = 0;
    for (int it = 0; it < iters; ++it) {
        u32 t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0, t6 = 0, t7 = 0;
        if (MODE <= 3) {
            asm volatile(
                "s_mov_b64 exec, %9\n ds_read_b32 %0, %8\n s_mov_b64 exec, %10\n ds_read_b32 %1, %8 offset:4096\n"
                "s_mov_b64 exec, %9\n ds_read_b32 %2, %8 offset:8192\n s_mov_b64 exec, %10\n ds_read_b32 %3, %8 offset:12288\n"
                "s_mov_b64 exec, %9\n ds_read_b32 %4, %8 offset:16384\n s_mov_b64 exec, %10\n ds_read_b32 %5, %8 offset:20480\n"
                "s_mov_b64 exec, %9\n ds_read_b32 %6, %8 offset:24576\n s_mov_b64 exec, %10\n ds_read_b32 %7, %8 offset:28672\n"
                "s_mov_b64 exec, -1\n s_waitcnt lgkmcnt(0)\n"
                : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3), "+v"(t4), "+v"(t5), "+v"(t6), "+v"(t7)
                : "v"(addr),
                  "s"(MODE == 0 ? ~0ull : MODE == 3 ? 0xffffull : 0xffffffffull),
                  "s"(MODE == 0 ? ~0ull : MODE == 1 ? 0xffffffffull : MODE == 3 ? 0xffffull : 0xffffffff00000000ull));
        } else {
#define RD(T, OFF, BIT)                                                                                          \
    "s_bfe_i32 exec_lo, %9, " #BIT " | 0x10000\n s_bfe_i32 exec_hi, %10, " #BIT " | 0x10000\n s_cbranch_execz 1f\n" \
    "ds_read_b32 %" #T ", %8 offset:" #OFF "\n1:\n"
            asm volatile(RD(0, 0, 0) RD(1, 4096, 1) RD(2, 8192, 2) RD(3, 12288, 3) RD(4, 16384, 4) RD(5, 20480, 5) RD(6, 24576, 6) RD(7, 28672, 7)
                         "s_mov_b64 exec, -1\n s_waitcnt lgkmcnt(0)\n"
                         : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3), "+v"(t4), "+v"(t5), "+v"(t6), "+v"(t7)
                         : "v"(addr), "s"(m_lo), "s"(m_hi));
#undef RD
        }
        acc ^= t0 ^ t1 ^ t2 ^ t3 ^ t4 ^ t5 ^ t6 ^ t7;
    }
    out[blockIdx.x * 1024 + threadIdx.x] = acc;
}

template <int MODE>
void run(const char* name, u32 m_lo = ~0u, u32 m_hi = ~0u) {
    u32* out;
    hipMalloc(&out, 256 * 1024 * 4);
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(1024), 65536, 0, out, 10, m_lo, m_hi);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(1024), 65536, 0, out, iters, m_lo, m_hi);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double winst = 16.0 * 8 * iters;
    printf("%-66s %8.3f ms -> %5.2f ns per read slot of a wave and CU\n", name, ms, ms * 1e6 / winst);
    hipFree(out);
}

int main() {
    run<0>("ds_read_b32, all 64 lanes");
    run<1>("ds_read_b32, low 32 lanes (EXEC)");
    run<2>("ds_read_b32, halves alternating");
    run<3>("ds_read_b32, low 16 lanes");
    run<4>("EXEC from s_bfe_i32 x2 + execz branch, masks all ones");
    run<5>("same, lo mask 0x55, hi mask 0x33 (2 full, 4 half, 2 skipped of 8)", 0x55u, 0x33u);
    run<5>("same, lo mask 0x11, hi mask 0x22 (4 half, 4 skipped of 8)", 0x11u, 0x22u);
    run<5>("same, masks zero (everything skipped)", 0u, 0u);
    return 0;
}
