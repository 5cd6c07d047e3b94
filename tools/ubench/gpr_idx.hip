// Micro-benchmark: what do the pieces of the modal-state S3 score loop (epg_s3_sparse.hip) cost a SIMD?  VGPR index mode
// (s_set_gpr_idx_on / _idx / a plain M0 write in front of an indexed v_add_u32), EXEC switches, v_readlane + scalar decode.
// 1024 threads per workgroup (4 waves per SIMD), one workgroup per CU; time per inner step of ONE wave's instruction stream
// and per CU.  build: hipcc --offload-arch=gfx950 -O3 gpr_idx.hip -o gpr_idx
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef unsigned int u32;
typedef int v16i __attribute__((ext_vector_type(16)));

// MODE 0: 8 plain v_add_u32                         1: 8 x (s_set_gpr_idx_on s, 0xa ; v_add_u32 indexed)
//      2: mode on once, 8 x (s_set_gpr_idx_idx s ; v_add)     3: mode on once, 8 x (s_mov_b32 m0, s ; v_add)
//      4: 8 x (s_mov_b64 exec, half ; v_add_u32)              5: 8 x (v_readlane ; s_bfe ; s_lshr) then 8 v_add
//      6: like 1 but with s_nop 0 between the two             7: 8 x (s_set_gpr_idx_on ; v_lshl_add_u32 indexed src2)
template <int MODE>
__global__ __launch_bounds__(1024) void k(u32* out, int iters, int i0, int i1) {
    v16i acc;
    for (int e = 0; e < 16; ++e) acc[e] = threadIdx.x + e;
    u32 t = threadIdx.x * 3 + 1, u = threadIdx.x;
    const unsigned long long lo = 0xffffffffull, hi = 0xffffffff00000000ull;
    const int a = __builtin_amdgcn_readfirstlane(i0), b = __builtin_amdgcn_readfirstlane(i1);
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            asm volatile("v_add_u32 v40, %1, v40\n v_add_u32 v41, %1, v41\n v_add_u32 v42, %1, v42\n v_add_u32 v43, %1, v43\n"
                         "v_add_u32 v44, %1, v44\n v_add_u32 v45, %1, v45\n v_add_u32 v46, %1, v46\n v_add_u32 v47, %1, v47\n"
                         : "+{v[40:55]}"(acc) : "v"(t));
        } else if (MODE == 1 || MODE == 6) {
#define ST(S) "s_set_gpr_idx_on " S ", 0xa\n" NOP "v_add_u32 v40, %1, v40\n"
#define NOP ""
            if (MODE == 1)
                asm volatile(ST("%2") ST("%3") ST("%2") ST("%3") ST("%2") ST("%3") ST("%2") ST("%3") "s_set_gpr_idx_off\n"
                             : "+{v[40:55]}"(acc) : "v"(t), "s"(a), "s"(b) : "m0");
#undef NOP
#define NOP "s_nop 0\n"
            if (MODE == 6)
                asm volatile(ST("%2") ST("%3") ST("%2") ST("%3") ST("%2") ST("%3") ST("%2") ST("%3") "s_set_gpr_idx_off\n"
                             : "+{v[40:55]}"(acc) : "v"(t), "s"(a), "s"(b) : "m0");
#undef NOP
#undef ST
        } else if (MODE == 2) {
#define ST(S) "s_set_gpr_idx_idx " S "\n v_add_u32 v40, %1, v40\n"
            asm volatile("s_set_gpr_idx_on %2, 0xa\n" ST("%2") ST("%3") ST("%2") ST("%3") ST("%2") ST("%3") ST("%2") ST("%3") "s_set_gpr_idx_off\n"
                         : "+{v[40:55]}"(acc) : "v"(t), "s"(a), "s"(b) : "m0");
#undef ST
        } else if (MODE == 3) {
#define ST(S) "s_mov_b32 m0, " S "\n v_add_u32 v40, %1, v40\n"
            asm volatile("s_set_gpr_idx_on %2, 0xa\n" ST("%4") ST("%5") ST("%4") ST("%5") ST("%4") ST("%5") ST("%4") ST("%5") "s_set_gpr_idx_off\n"
                         : "+{v[40:55]}"(acc) : "v"(t), "s"(a), "s"(b), "s"(a | 0xa000), "s"(b | 0xa000) : "m0");
#undef ST
        } else if (MODE == 4) {
#define ST(M, R) "s_mov_b64 exec, " M "\n v_add_u32 " R ", %1, " R "\n"
            asm volatile(ST("%2", "v40") ST("%3", "v41") ST("%2", "v42") ST("%3", "v43") ST("%2", "v44") ST("%3", "v45") ST("%2", "v46") ST("%3", "v47")
                         "s_mov_b64 exec, -1\n"
                         : "+{v[40:55]}"(acc) : "v"(t), "s"(lo), "s"(hi));
#undef ST
        } else if (MODE == 5) {
#define ST(L, R) "v_readlane_b32 s40, %1, " L "\n s_bfe_u32 s41, s40, 0x80008\n s_lshr_b32 s42, s40, 16\n s_and_b32 s43, s41, s42\n v_add_u32 " R ", s43, " R "\n"
            asm volatile(ST("1", "v40") ST("2", "v41") ST("3", "v42") ST("4", "v43") ST("5", "v44") ST("6", "v45") ST("7", "v46") ST("8", "v47")
                         : "+{v[40:55]}"(acc) : "v"(u) : "s40", "s41", "s42", "s43", "scc");
#undef ST
        } else if (MODE == 7) {
#define ST(S, D) "s_set_gpr_idx_on " S ", 0x4\n v_lshl_add_u32 " D ", %2, 7, v40\n"
            u32 d0, d1, d2, d3;
            asm volatile(ST("%6", "%0") ST("%7", "%1") ST("%6", "%2") ST("%7", "%3") ST("%6", "%0") ST("%7", "%1") ST("%6", "%2") ST("%7", "%3")
                         "s_set_gpr_idx_off\n"
                         : "=&v"(d0), "=&v"(d1), "+v"(t), "=&v"(d3), "+{v[40:55]}"(acc) : "v"(u), "s"(a), "s"(b) : "m0");
            u ^= d0 ^ d1 ^ d3;
#undef ST
        }
    }
    u32 s = t ^ u;
    for (int e = 0; e < 16; ++e) s ^= (u32)acc[e];
    out[blockIdx.x * 1024 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name) {
    u32* out;
    hipMalloc(&out, 256 * 1024 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(1024), 0, 0, out, 10, 3, 5);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(1024), 0, 0, out, iters, 3, 5);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // per CU: 16 waves each doing iters * 8 steps
    const double ns_step_cu = ms * 1e6 / ((double)iters * 8 * 16);
    printf("%-58s %8.3f ms   %.2f ns per step and CU (%.1f cycles of a SIMD per step of one of its 4 waves)\n", name, ms, ns_step_cu,
           ns_step_cu * 4 * 2.4);
    hipFree(out);
}

int main() {
    run<0>("0: v_add_u32");
    run<1>("1: s_set_gpr_idx_on + indexed v_add_u32");
    run<6>("6: s_set_gpr_idx_on + s_nop 0 + indexed v_add_u32");
    run<2>("2: s_set_gpr_idx_idx + indexed v_add_u32");
    run<3>("3: s_mov_b32 m0 + indexed v_add_u32");
    run<4>("4: s_mov_b64 exec + v_add_u32");
    run<5>("5: v_readlane + 3 SALU + v_add_u32 (sgpr)");
    run<7>("7: s_set_gpr_idx_on + v_lshl_add_u32 (src2 indexed)");
    return 0;
}
