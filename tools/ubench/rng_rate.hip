// Micro-benchmark: 128-bit blocks of counter-based random numbers per second -- Philox4x32-10 / -7 (four 32x32->64 multiplies
// per round: v_mul_lo_u32 + v_mul_hi_u32) against Threefry4x32-20 / -12 (adds, rotates, xors only) -- and the bare rates of
// v_mul_lo_u32 / v_mul_hi_u32 / v_mul_u32_u24 / v_alignbit_b32.  build: hipcc --offload-arch=gfx950 -O3 rng_rate.hip -o rng_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32;
typedef unsigned long long u64;

template <int ROUNDS>
__device__ __forceinline__ void philox(u32 (&c)[4], u32 k0, u32 k1) {
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const u64 p0 = (u64)0xD2511F53u * c[0];
        const u64 p1 = (u64)0xCD9E8D57u * c[2];
        const u32 n0 = (u32)(p1 >> 32) ^ c[1] ^ k0;
        const u32 n2 = (u32)(p0 >> 32) ^ c[3] ^ k1;
        c[0] = n0; c[1] = (u32)p1; c[2] = n2; c[3] = (u32)p0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

__device__ __forceinline__ u32 rotl(u32 x, int r) { return __builtin_amdgcn_alignbit(x, x, 32 - r); }

template <int ROUNDS>
__device__ __forceinline__ void threefry(u32 (&x)[4], u32 k0, u32 k1, u32 k2, u32 k3) {
    const u32 ks[5] = {k0, k1, k2, k3, 0x1BD11BDAu ^ k0 ^ k1 ^ k2 ^ k3};
    const int R[8][2] = {{10, 26}, {11, 21}, {13, 27}, {23, 5}, {6, 20}, {17, 11}, {25, 10}, {18, 20}};
    x[0] += ks[0]; x[1] += ks[1]; x[2] += ks[2]; x[3] += ks[3];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        if (r & 1) {
            x[0] += x[3]; x[3] = rotl(x[3], R[r & 7][0]) ^ x[0];
            x[2] += x[1]; x[1] = rotl(x[1], R[r & 7][1]) ^ x[2];
        } else {
            x[0] += x[1]; x[1] = rotl(x[1], R[r & 7][0]) ^ x[0];
            x[2] += x[3]; x[3] = rotl(x[3], R[r & 7][1]) ^ x[2];
        }
        if ((r & 3) == 3) {
            const int s = r / 4 + 1;
            x[0] += ks[s % 5]; x[1] += ks[(s + 1) % 5]; x[2] += ks[(s + 2) % 5]; x[3] += ks[(s + 3) % 5] + s;
        }
    }
}

template <int WHICH>
__global__ __launch_bounds__(256) void k(u32* out, int iters, u32 seed) {
    const u32 id = blockIdx.x * 256 + threadIdx.x;
    u32 acc = 0;
    for (int it = 0; it < iters; ++it) {
        u32 c[4] = {id, (u32)it, acc & 1u, 0x6e756c6cu};
        if (WHICH == 0) philox<10>(c, seed, 77u);
        if (WHICH == 1) philox<7>(c, seed, 77u);
        if (WHICH == 2) threefry<20>(c, seed, 77u, 1u, 2u);
        if (WHICH == 3) threefry<12>(c, seed, 77u, 1u, 2u);
        if (WHICH == 4) { c[0] = c[0] * (seed | 1u); c[1] = c[1] * c[0]; c[2] = c[2] * c[1]; c[3] = c[3] * c[2]; }
        if (WHICH == 5) { c[0] = __umulhi(c[0], seed | 1u); c[1] = __umulhi(c[1], c[0]); c[2] = __umulhi(c[2], c[1]); c[3] = __umulhi(c[3], c[2]); }
        if (WHICH == 6) { c[0] = __umul24(c[0], seed | 1u); c[1] = __umul24(c[1], c[0]); c[2] = __umul24(c[2], c[1]); c[3] = __umul24(c[3], c[2]); }
        if (WHICH == 7) { c[0] = rotl(c[0], 5) ^ seed; c[1] = rotl(c[1], 7) ^ c[0]; c[2] = rotl(c[2], 9) ^ c[1]; c[3] = rotl(c[3], 11) ^ c[2]; }
        acc ^= c[0] ^ c[1] ^ c[2] ^ c[3];
    }
    out[id] = acc;
}

template <int WHICH>
void run(const char* name, double ops_per_iter) {
    u32* out;
    const int blocks = 256 * 8;
    hipMalloc(&out, blocks * 256 * sizeof(u32));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    hipLaunchKernelGGL(k<WHICH>, dim3(blocks), dim3(256), 0, 0, out, 10, 3u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<WHICH>, dim3(blocks), dim3(256), 0, 0, out, iters, 3u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double wi = (double)blocks * 4 * iters;            // wave-iterations
    const double cyc = 2.4e9 * 1024.0 * (ms * 1e-3) / wi;    // SIMD-cycles per wave-iteration (8 waves per SIMD resident)
    printf("%-18s %8.3f ms  %8.1f SIMD-cycles per wave-iteration (%.1f per op of %g)  -> %.2f T lane-blocks/s\n", name, ms, cyc,
           cyc / ops_per_iter, ops_per_iter, wi * 64 / (ms * 1e-3) / 1e12);
    hipFree(out);
}

int main() {
    run<0>("philox4x32-10", 1);
    run<1>("philox4x32-7", 1);
    run<2>("threefry4x32-20", 1);
    run<3>("threefry4x32-12", 1);
    run<4>("4 x mul_lo_u32", 4);
    run<5>("4 x mul_hi_u32", 4);
    run<6>("4 x mul_u32_u24", 4);
    run<7>("4 x (alignbit+xor)", 4);
    return 0;
}
