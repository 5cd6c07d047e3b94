// Paired-mode null groups (reference helpers.py:183-194): every row of the concatenation [A|B] is shuffled by an
// independent uniform permutation and cut into two groups (the first ga and the next gb shuffled columns).  Only
// the per-state counts of the two groups enter the scores, so the kernel draws the group membership directly:
// sequential selection sampling over the row's columns (column c joins group A with probability
// need_A / remaining, group B with need_B / remaining) is exactly the law of (first ga, next gb) of a uniform
// permutation.  Randomness: Philox4x32-10 keyed by the seed, counter = (global row, column block) -- the result
// depends only on (seed, row0 + row), never on the launch geometry or on which GPU owns the row.
// One lane per row; the lane's two histograms live in LDS as private uint16 columns (no atomics).  gfx950 only.
#include "epg_common.h"

namespace epg {

__device__ __forceinline__ void philox4x32_10(u32 (&c)[4], u32 k0, u32 k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const u64 p0 = (u64)0xD2511F53u * c[0];
        const u64 p1 = (u64)0xCD9E8D57u * c[2];
        const u32 n0 = (u32)(p1 >> 32) ^ c[1] ^ k0;
        const u32 n2 = (u32)(p0 >> 32) ^ c[3] ^ k1;
        c[0] = n0; c[1] = (u32)p1; c[2] = n2; c[3] = (u32)p0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

__global__ __launch_bounds__(256) void k_null_hist(const char* __restrict__ XA, int NA, long ldxa, const char* __restrict__ XB,
                                                    int NB, long ldxb, long R, int S, int ga, int gb, u64 seed, long row0,
                                                    u16* __restrict__ HA, u16* __restrict__ HB) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u16* h = reinterpret_cast<u16*>(smem);   // [2][S][256]
    const int tid = threadIdx.x;
    const long row = (long)blockIdx.x * 256 + tid;
    for (int e = tid; e < 2 * S * 256; e += 256) h[e] = 0;
    __syncthreads();
    if (row < R) {
        const u64 grow = (u64)(row0 + row);
        const int M = NA + NB;
        u32 needA = (u32)ga, needB = (u32)gb, rem = (u32)M;
        const char* pa = XA + row * ldxa;
        const char* pb = XB + row * ldxb;
        for (int c0 = 0; c0 < M && (needA | needB); c0 += 4) {
            u32 ctr[4] = {(u32)grow, (u32)(grow >> 32), (u32)(c0 >> 2), 0u};
            philox4x32_10(ctr, (u32)seed, (u32)(seed >> 32));
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = c0 + k;
                if (c >= M) break;
                const u32 pick = (u32)(((u64)ctr[k] * rem) >> 32);      // uniform in [0, rem)
                const int x = (int)(unsigned char)(c < NA ? pa[c] : pb[c - NA]);
                if (pick < needA) {
                    if (x < S) h[(0 * S + x) * 256 + tid] += 1;
                    --needA;
                } else if (pick < needA + needB) {
                    if (x < S) h[(1 * S + x) * 256 + tid] += 1;
                    --needB;
                }
                --rem;
            }
        }
        for (int s = 0; s < S; ++s) {
            HA[row * S + s] = h[(0 * S + s) * 256 + tid];
            HB[row * S + s] = h[(1 * S + s) * 256 + tid];
        }
    }
}

// quiescent from cached histograms of the two real groups (scores.py:294-303)
int null_hist_impl(const int8_t* XA, int32_t NA, int64_t ldxa, const int8_t* XB, int32_t NB, int64_t ldxb, int64_t R, int32_t S,
                   int32_t ga, int32_t gb, uint64_t seed, int64_t row0, uint16_t* HA, uint16_t* HB, hipStream_t st) {
    if (R < 0 || NA < 1 || NB < 1 || ldxa < NA || ldxb < NB || S < 1 || S > 31)
        return fail(EPG_ERR_INVALID_ARG, "null_hist: bad shape");
    if (ga < 0 || gb < 0 || (long)ga + gb > (long)NA + NB)
        return fail(EPG_ERR_INVALID_ARG, "null_hist: group sizes %d + %d exceed the %d columns", ga, gb, NA + NB);
    if (NA + NB > 65535) return fail(EPG_ERR_UNSUPPORTED, "null_hist: more than 65535 columns");
    if (R == 0) return EPG_OK;
    if (!XA || !XB || !HA || !HB) return fail(EPG_ERR_INVALID_ARG, "null_hist: NULL argument");
    const size_t shmem = (size_t)2 * S * 256 * 2;
    hipLaunchKernelGGL(k_null_hist, dim3((unsigned)((R + 255) / 256)), dim3(256), shmem, st, reinterpret_cast<const char*>(XA), NA,
                       (long)ldxa, reinterpret_cast<const char*>(XB), NB, (long)ldxb, (long)R, S, ga, gb, (u64)seed, (long)row0, HA, HB);
    EPG_LAUNCH_CHECK("k_null_hist");
    return EPG_OK;
}

}  // namespace epg
