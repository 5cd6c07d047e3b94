// What makes the histogram store of the count pass cost 7-17 % of the launch when it is 4 % of the bytes?  
// A read stream over one 4 GiB block with a write stream of 1152 bytes per 27136 read (36 of 848: the count pass's ratio) into
// ANOTHER 4 GiB block, for every combination of
//   read pattern   0 = every lane 16 contiguous bytes (1 KiB per wave instruction)
//                  1 = the count pass's: 16 rows of 848 bytes per instruction, four lanes x 16 bytes per row, 14 instructions per half tile
//   store pattern  2 = each wave stores its super-tile's 1152 bytes when it is done with it (the count pass)
//                  3 = the four waves of a workgroup meet at a barrier and store their 4608 contiguous bytes together
//                  4 G = a wave takes G CONSECUTIVE super-tiles and stores their G x 1152 bytes as one run when it is done with them (4 = like 2)
//   occupancy      workgroups per CU (dynamic LDS keeps more from fitting)
// printed as the ratio to the same kernel without the store, for the first `targets` other blocks (memory classes show as two levels).
// build: hipcc --offload-arch=gfx950 -O3 store_cost.hip -o store_cost       usage: store_cost [targets=10] [matrix GiB=12]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <functional>
#include <vector>
typedef unsigned int u32;

constexpr long ROW = 848, TILE = 32 * ROW, HT = 32 * 36;

template <int READ, int STORE>
__global__ __launch_bounds__(256) void k_var(const char* __restrict__ X, char* __restrict__ H, long nsuper, u32* out) {
    extern __shared__ char pad[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32 acc = 0;
    constexpr int G = STORE >= 4 ? STORE / 4 : 1;                                        // super-tiles per run (STORE = 4 G)
    const long nruns = nsuper / G;
    for (long run = (long)blockIdx.x * 4 + wave; run < nruns; run += (long)gridDim.x * 4) {
        for (int slot = 0; slot < G; ++slot) {
            const long st = run * G + slot;                                               // a wave's G super-tiles are consecutive
            if (READ == 0) {
                const uint4* p = reinterpret_cast<const uint4*>(X + st * TILE);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    uint4 v[14];
#pragma unroll
                    for (int i = 0; i < 14; ++i) {
                        int idx = (h * 14 + i) * 64 + lane;
                        idx = idx < 1696 ? idx : 1695;
                        v[i] = p[idx];
                    }
#pragma unroll
                    for (int i = 0; i < 14; ++i) acc ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
                }
            } else {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const char* rowp = X + (st * 32 + h * 16 + (lane >> 2)) * ROW;
                    uint4 v[14];
#pragma unroll
                    for (int i = 0; i < 14; ++i) {
                        int c = 4 * i + (lane & 3);
                        c = c < 52 ? c : 52;
                        v[i] = *reinterpret_cast<const uint4*>(rowp + 16 * c);
                    }
#pragma unroll
                    for (int i = 0; i < 14; ++i) acc ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
                }
            }
        }
        if (STORE == 3) __syncthreads();
        if (STORE >= 2) {                                                                 // every byte of H is written exactly once
            char* dst = H + run * (G * HT);
            for (int c = lane; c < 72 * G; c += 64) *reinterpret_cast<uint4*>(dst + 16 * c) = make_uint4(acc, 1, 2, 3);
        }
    }
    if (acc == 0x12345678u) out[0] = acc + pad[0];
}

static float timed(hipEvent_t e0, hipEvent_t e1, int reps, const std::function<void()>& f) {
    float best = 1e9;
    for (int it = 0; it < reps + 1; ++it) {
        float ms;
        (void)hipEventRecord(e0);
        f();
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (it && ms < best) best = ms;
    }
    return best;
}

template <int READ, int STORE>
static float run(const char* X, char* H, long nsuper, int wg_per_cu, u32* out, hipEvent_t e0, hipEvent_t e1) {
    const int shmem = wg_per_cu >= 8 ? 0 : (160 * 1024 / wg_per_cu) - 1024;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_var<READ, STORE>), hipFuncAttributeMaxDynamicSharedMemorySize, shmem);
    return timed(e0, e1, 3, [&] { hipLaunchKernelGGL((k_var<READ, STORE>), dim3(256 * wg_per_cu), dim3(256), shmem, 0, X, H, nsuper, out); });
}

int main(int argc, char** argv) {
    const int targets = argc > 1 ? atoi(argv[1]) : 10;
    const long BLK = 4L << 30, XB = argc > 2 ? atol(argv[2]) << 30 : 12L << 30;
    std::vector<char*> blk;
    {   // the read stream: ONE allocation of 12 GiB like the state matrix, so that its 0.5 GB of stores are twice the 256 MB
        // memory-side cache (with a 4 GiB matrix the 180 MB of stores never reach the memory inside the launch: every pattern
        // with long runs then looks free, 1.01 / 1.05, and the count pass does not behave like that)
        char* p = nullptr;
        if (hipMalloc(&p, XB) != hipSuccess) return 1;
        (void)hipMemset(p, 7, XB);
        blk.push_back(p);
    }
    for (int i = 1; i < targets + 1; ++i) {
        char* p = nullptr;
        if (hipMalloc(&p, BLK) != hipSuccess) break;
        (void)hipMemset(p, 1 + i, BLK);
        blk.push_back(p);
    }
    u32* out;
    (void)hipMalloc(&out, 4);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const long nsuper = XB / TILE;
    const int nb = (int)blk.size();
#define ROWOF(READ, STORE, WG)                                                                                             \
    {                                                                                                                      \
        const float base = run<READ, 0>(blk[0], blk[1], nsuper, WG, out, e0, e1);                                          \
        printf("read %d store %d  %2d wg/CU: alone %.3f ms = %4.0f GB/s; with the store into block j:", READ, STORE, WG, base, \
               nsuper * TILE / base / 1e6);                                                                                \
        for (int j = 1; j < nb; ++j) printf(" %.3f", run<READ, STORE>(blk[0], blk[j], nsuper, WG, out, e0, e1) / base);     \
        printf("\n");                                                                                                      \
        fflush(stdout);                                                                                                    \
    }
    for (int wg : {3}) {
        ROWOF(0, 2, wg)
        ROWOF(0, 3, wg)
        ROWOF(0, 4, wg)
        ROWOF(0, 8, wg)
        ROWOF(0, 16, wg)
        ROWOF(0, 32, wg)
        ROWOF(0, 64, wg)
        ROWOF(1, 2, wg)
        ROWOF(1, 8, wg)
        ROWOF(1, 16, wg)
        ROWOF(1, 32, wg)
    }
    return 0;
}
