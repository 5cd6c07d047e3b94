// S3 path (biosample-pair saliency).  gfx950 only.
//
// Expected pass (reference expected.py:165-204 s3Calc): C[a,b,i,j] = #{bins : x[a] == i and x[b] == j}, a != b.
//   k_s3_hist: a block owns TA "a" biosamples x 64 "b" biosamples x a slice of <= 65535 bins and keeps their
//   TA*64*S*S co-occurrence counters as packed uint16 pairs in LDS (<= 150 KB); each lane is one b, reads its
//   state byte (a coalesced 64-byte row segment per wave and bin) and issues one ds_add_u32 per a.  The slice
//   length bounds every counter by 65535; the flush adds the non-zero counters to the int32 global array.
//   Integer arithmetic only => exact and independent of launch geometry.  Bound: LDS atomic rate
//   (R*N*(N-1) increments), see DESIGN.md.
//
// Score pass (reference scores.py:455-506 s3Score):
//   T[a,b,i,j] = kl(float32(1)/P, q[a,b,i,j]) in float32 (scores.py:479-480), stored transposed as
//   T2[b][j][a][i] so that the N*S values needed for one (b, j) are contiguous;
//   score[bin, s] = sum_{b: x_b == s} sum_{a != b} T[a, b, x_a, s]   (closed form of scores.py:496-498).
//   k_s3_score: a block owns one b and a slice of bins; for every state s present in the slice's column b it
//   stages the tile T2[b][s][:][:] (N*S floats, <= 150 KB) in LDS and, for each bin with x_b == s, a wave
//   gathers tile[a][x_a] over all a (coalesced row read, LDS gather), reduces in float64 and adds the result to
//   the float64 score.  The diagonal a == b contributes 0 because q[a,a,:,:] == 0 (kl masks q == 0).
#include "epg_common.h"

#include <stdlib.h>

namespace epg {

constexpr int S3_TB = 64;             // b biosamples per block (= lanes of a wave)
constexpr int S3_SLICE = 65535;       // bins per slice: packed uint16 counters cannot overflow
constexpr int S3_LDS_BUDGET = 150 * 1024;

// uint16 counters per lane: S*S rounded up to 2 (mod 4), i.e. an ODD number of dwords, so that the 64 lanes of a wave
// hitting the same (state, state) cell -- the common case on real data, 71 % of cells are one state -- spread over all
// 32 LDS banks (2-way, the minimum for 64 lanes) instead of 4-way
__host__ __device__ inline int s3_lane_stride(int S) {
    int v = S * S;
    while ((v & 3) != 2) ++v;
    return v;
}

constexpr int S3H_THREADS = 1024;     // 16 waves per block: the block owns the CU's LDS, so occupancy must come from its size
constexpr int S3H_UNROLL = 4;         // bins per wave iteration (independent loads in flight)

__global__ __launch_bounds__(S3H_THREADS) void k_s3_hist(const char* __restrict__ X, long R, int N, long ldx, int S, int TA,
                                                          int n_btiles, int* __restrict__ counts) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u32* hist = reinterpret_cast<u32*>(smem);
    const int SS = S * S;
    const int SSP = s3_lane_stride(S);            // per-lane counter block, padded to an odd number of dwords
    const int words = TA * S3_TB * SSP / 2;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int NW = S3H_THREADS / 64;
    const long tile = blockIdx.x;                 // (a-tile, b-tile)
    const int a0 = (int)(tile / n_btiles) * TA;
    const int b = (int)(tile % n_btiles) * S3_TB + lane;
    const long r0 = (long)blockIdx.y * S3_SLICE;
    const long r1 = r0 + S3_SLICE < R ? r0 + S3_SLICE : R;
    // symmetry: C[b,a,j,i] == C[a,b,i,j].  Only pairs a < b are counted; the flush writes both entries.  A tile whose
    // a range lies entirely at or above its b range has no such pair.
    if (a0 >= (int)(tile % n_btiles) * S3_TB + S3_TB - 1) return;

    for (int w = threadIdx.x; w < words; w += S3H_THREADS) hist[w] = 0;
    __syncthreads();

    const bool b_ok = b < N;
    const int bcl = b_ok ? b : 0;
    for (long rb = r0 + (long)wave * S3H_UNROLL; rb < r1; rb += (long)NW * S3H_UNROLL) {
        int xb[S3H_UNROLL], xa[S3H_UNROLL][4];
#pragma unroll
        for (int u = 0; u < S3H_UNROLL; ++u) {
            const long row = rb + u < r1 ? rb + u : r1 - 1;        // clamp: loads stay in bounds, masked below
            const char* rp = X + row * ldx;
            xb[u] = (int)(unsigned char)rp[bcl];
#pragma unroll
            for (int ta = 0; ta < 4; ++ta) xa[u][ta] = (int)(unsigned char)rp[a0 + ta < N ? a0 + ta : 0];
        }
#pragma unroll
        for (int u = 0; u < S3H_UNROLL; ++u) {
            if (rb + u >= r1 || !b_ok || xb[u] >= S) continue;
#pragma unroll
            for (int ta = 0; ta < 4; ++ta) {
                const int a = a0 + ta;
                if (ta >= TA || a >= b || xa[u][ta] >= S) continue;          // a < b only (b < N is b_ok)
                const int idx = (ta * S3_TB + lane) * SSP + xa[u][ta] * S + xb[u];
                atomicAdd(&hist[idx >> 1], 1u << (16 * (idx & 1)));
            }
        }
    }
    __syncthreads();
    for (int w = threadIdx.x; w < words; w += S3H_THREADS) {
        const u32 v = hist[w];
        if (!v) continue;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const u32 c = h ? v >> 16 : v & 0xffffu;
            if (!c) continue;
            const int idx = 2 * w + h;
            const int ta = idx / (S3_TB * SSP);
            const int rem = idx - ta * S3_TB * SSP;
            const int bl = rem / SSP, ij = rem - bl * SSP;      // ij >= SS is padding and stays zero
            const long a = a0 + ta, bb = (long)(tile % n_btiles) * S3_TB + bl;
            const int i = ij / S, j = ij - i * S;
            atomicAdd(&counts[((a * N + bb) * SS) + ij], (int)c);
            atomicAdd(&counts[((bb * N + a) * SS) + j * S + i], (int)c);      // the mirrored pair (b, a)
        }
    }
}

// T2[b][j][a][i] = float32 kl(float32(1)/P, q[a,b,i,j]); float32 arithmetic like scores.py:479-480.
__global__ void k_s3_table(const float* __restrict__ q, int N, int S, float* __restrict__ T2) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)N * N * S * S;
    if (e >= total) return;
    // e indexes T2: ((b*S + j)*N + a)*S + i
    const int i = (int)(e % S);
    long t = e / S;
    const int a = (int)(t % N);
    t /= N;
    const int j = (int)(t % S);
    const int b = (int)(t / S);
    const float qv = q[(((long)a * N + b) * S + i) * S + j];
    const float obs = 1.0f / (float)((long)N * (N - 1));
    float v = 0.0f;
    if (qv != 0.0f) {
        const float r = obs / qv;
        if (r > 0.0f) v = obs * log2f(r);
    }
    T2[e] = v;
}

constexpr int S3_SCORE_SLICE = 4096;
constexpr int S3S_THREADS = 512;
constexpr int S3S_MAX_LANES_PER_BIN = 64;

// One lane = one bin.  For the block's biosample b and every state s present in the slice's column b: stage the
// tile T2[b][s][:][:] in LDS, compact the bins with x_b == s into a list, and let each lane walk its bin's row
// (16-byte loads, next chunk prefetched) gathering tile[a][x_a] for all a -- no cross-lane reduction, 64 bins per
// wave in flight.  Four float partial sums per 16-state chunk are folded into a float64 accumulator (each float sum
// has 4 terms, so the float64 total carries ~1e-7 relative error, far inside the 2e-6 test bound).
__global__ __launch_bounds__(S3S_THREADS) void k_s3_score(const char* __restrict__ X, long R, int N, long ldx, int S,
                                                           const float* __restrict__ T2, double* __restrict__ out64) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* tile = reinterpret_cast<float*>(smem);                                              // [N*S]
    unsigned short* list = reinterpret_cast<unsigned short*>(smem + (size_t)N * S * 4);         // [S3_SCORE_SLICE]
    unsigned char* col = reinterpret_cast<unsigned char*>(list + S3_SCORE_SLICE);               // [S3_SCORE_SLICE]
    __shared__ u32 present;
    __shared__ u32 nlist;
    const int b = blockIdx.x % N;                 // b fastest: blocks of one slice run together (rows stay in L2)
    const long slice = blockIdx.x / N;
    const long r0 = slice * S3_SCORE_SLICE;
    const int nb = (int)((R - r0) < S3_SCORE_SLICE ? (R - r0) : S3_SCORE_SLICE);
    const int nchunks = (N + 15) >> 4;

    if (threadIdx.x == 0) present = 0;
    __syncthreads();
    u32 mine = 0;
    for (int k = threadIdx.x; k < nb; k += S3S_THREADS) {
        const unsigned char v = (unsigned char)X[(r0 + k) * ldx + b];
        col[k] = v;
        if (v < S) mine |= 1u << v;
    }
    if (mine) atomicOr(&present, mine);
    __syncthreads();
    const u32 have = present;

    for (int s = 0; s < S; ++s) {
        if (!((have >> s) & 1u)) continue;       // block-uniform
        __syncthreads();                           // previous tile and list fully consumed
        if (threadIdx.x == 0) nlist = 0;
        const float* src = T2 + ((long)b * S + s) * N * S;
        // tile: 16-byte loads, all of a thread's loads in flight at once (a scalar-load loop costs one L2 round trip
        // per 512 floats and dominated the pass); T2 tiles are 16-byte aligned when N*S is a multiple of 4
        const int n4 = ((N * S) & 3) == 0 ? (N * S) >> 2 : 0;
        {
            const float4* src4 = reinterpret_cast<const float4*>(src);
            float4* tile4 = reinterpret_cast<float4*>(tile);
            constexpr int U = 8;
            for (int e0 = threadIdx.x; e0 < n4; e0 += S3S_THREADS * U) {
                float4 v[U];
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (e0 + u * S3S_THREADS < n4) v[u] = src4[e0 + u * S3S_THREADS];
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (e0 + u * S3S_THREADS < n4) tile4[e0 + u * S3S_THREADS] = v[u];
            }
            for (int e = 4 * n4 + threadIdx.x; e < N * S; e += S3S_THREADS) tile[e] = src[e];
        }
        __syncthreads();
        // compact the bins with x_b == s: one LDS atomic per wave (ballot + lane prefix), not one per bin
        for (int k0 = 0; k0 < nb; k0 += S3S_THREADS) {
            const int k = k0 + threadIdx.x;
            const bool hit = k < nb && col[k] == (unsigned char)s;
            const unsigned long long m = __ballot(hit);
            const int lane = threadIdx.x & 63;
            u32 base = 0;
            if (lane == 0 && m) base = atomicAdd(&nlist, (u32)__popcll(m));
            base = __shfl(base, 0);
            if (hit) list[base + __popcll(m & ((1ull << lane) - 1))] = (unsigned short)k;
        }
        __syncthreads();
        const int n = (int)nlist;
        // a quad per bin: quad lane j walks chunks j, j+4, ... of the bin's row, four chunks in flight; the walk is a
        // chain of dependent L2 round trips, so four lanes per bin cut its length four-fold and keep the rare states'
        // passes (a handful of bins) from idling the block
        // lanes per bin: as many as keep the block full in ONE round (a rare state's handful of bins gets 16-64 lanes
        // each and one short dependent-load chain; the dominant state gets 1-4 lanes per bin and several rounds)
        int G = 1;
        while (G < S3S_MAX_LANES_PER_BIN && 2 * G * n <= S3S_THREADS) G <<= 1;
        if (G < 4 && n > 0) G = 4;
        const int BPR = S3S_THREADS / G;                               // bins per round
        const int j = threadIdx.x & (G - 1);
        for (int t = threadIdx.x / G; t < ((n + BPR - 1) / BPR) * BPR; t += BPR) {
            const bool live = t < n;
            const long row = r0 + (live ? list[t] : list[0]);
            const char* rp = X + row * ldx;
            double acc = 0.0;
            // a 16-byte chunk may run past N: into row padding or the next row (masked by the a < N test below); only
            // the matrix's very last row of a tightly packed matrix must not be over-read
            const bool tail_unsafe = row == R - 1 && ldx < 16L * nchunks;
            auto load_chunk = [&](int c) -> uint4 {
                if (c == nchunks - 1 && tail_unsafe) {
                    u32 w4[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
                    for (int a = 16 * c; a < N; ++a) {
                        const int o = a - 16 * c;
                        w4[o >> 2] = (w4[o >> 2] & ~(0xffu << (8 * (o & 3)))) | ((u32)(unsigned char)rp[a] << (8 * (o & 3)));
                    }
                    return make_uint4(w4[0], w4[1], w4[2], w4[3]);
                }
                return ld16(rp + 16 * c);
            };
            constexpr int PF = 4;
            uint4 ring[PF];
#pragma unroll
            for (int p = 0; p < PF; ++p) ring[p] = (live && j + G * p < nchunks) ? load_chunk(j + G * p) : make_uint4(~0u, ~0u, ~0u, ~0u);
            for (int c0 = j; c0 < nchunks && live; c0 += G * PF) {
#pragma unroll
                for (int p = 0; p < PF; ++p) {
                    const int c = c0 + G * p;
                    if (c >= nchunks) break;
                    const uint4 cur = ring[p];
                    if (c + G * PF < nchunks) ring[p] = load_chunk(c + G * PF);
                    const u32 w[4] = {cur.x, cur.y, cur.z, cur.w};
                    const int abase = 16 * c;
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        float part = 0.f;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int a = abase + 4 * d + k;
                            const u32 x = (w[d] >> (8 * k)) & 0xffu;
                            if (a < N && x < (u32)S) part += tile[a * S + (int)x];
                        }
                        acc += (double)part;
                    }
                }
            }
            for (int off = 1; off < G; off <<= 1) acc += __shfl_xor(acc, off);
            if (live && j == 0) atomicAdd(&out64[row * S + s], acc);
        }
    }
}

__global__ __launch_bounds__(256) void k_f64_to_f32(const double* __restrict__ in, long n, float* __restrict__ out) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = (float)in[i];
}

// ---------------------------------------------------------------------------------------------------------------
static int s3_ta(int S) {
    int ta = S3_LDS_BUDGET / (S3_TB * s3_lane_stride(S) * 2);
    if (ta > 4) ta = 4;
    return ta;
}

int64_t s3_mfma_ws_bytes(int64_t R, int N);
int hist_s3_mfma(const char* X, int64_t R, int32_t N, int64_t ldx, int32_t S, int32_t* counts, void* ws, hipStream_t st);

int64_t s3_table_bytes(int N, int S) { return align_up((int64_t)N * N * S * S * 4, 256); }
int64_t s3_ws_bytes(int64_t R, int N, int S) {
    const int64_t score = s3_table_bytes(N, S) + align_up(R * S * 8, 256);     // table + float64 accumulator
    const int64_t hist = s3_mfma_ws_bytes(R, N);                               // transposed state matrix
    return score > hist ? score : hist;
}

int hist_s3_impl(const int8_t* X8, int64_t R, int32_t N, int64_t ldx, int32_t S, int32_t* counts, void* ws, int64_t ws_bytes,
                 hipStream_t st) {
    if (R < 0 || N < 2 || ldx < N || S < 1) return fail(EPG_ERR_INVALID_ARG, "hist_s3: bad shape R=%lld N=%d ldx=%lld S=%d", (long long)R, N, (long long)ldx, S);
    if (S > 31) return fail(EPG_ERR_UNSUPPORTED, "hist_s3: S=%d > 31 not supported by this build", S);
    if (R == 0) return EPG_OK;
    if (!X8 || !counts) return fail(EPG_ERR_INVALID_ARG, "hist_s3: NULL argument");
    // matrix-core path when the caller gave room for the transposed matrix (S <= 30: padding rows use pattern 30);
    // EPG_S3_HIST=lds forces the LDS-atomic kernel (A/B measurements, and the fallback for S = 31 / no workspace)
    static const bool force_lds = [] { const char* e = getenv("EPG_S3_HIST"); return e && e[0] == 'l'; }();
    if (!force_lds && S <= 30 && ws && ws_bytes >= s3_mfma_ws_bytes(R, N))
        return hist_s3_mfma(reinterpret_cast<const char*>(X8), R, N, ldx, S, counts, ws, st);
    const int TA = s3_ta(S);
    if (TA < 1) return fail(EPG_ERR_UNSUPPORTED, "hist_s3: S=%d needs more LDS than a CU has", S);
    const int n_atiles = (N + TA - 1) / TA, n_btiles = (N + S3_TB - 1) / S3_TB;
    const long nslices = (R + S3_SLICE - 1) / S3_SLICE;
    if (nslices > 65535) return fail(EPG_ERR_UNSUPPORTED, "hist_s3: R=%lld too large for one call (max %lld bins)", (long long)R, 65535LL * S3_SLICE);
    const size_t shmem = (size_t)(TA * S3_TB * s3_lane_stride(S) / 2) * 4;
    EPG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_s3_hist), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    hipLaunchKernelGGL(k_s3_hist, dim3((unsigned)(n_atiles * n_btiles), (unsigned)nslices), dim3(S3H_THREADS), shmem, st,
                       reinterpret_cast<const char*>(X8), (long)R, N, (long)ldx, S, TA, n_btiles, counts);
    EPG_LAUNCH_CHECK("k_s3_hist");
    return EPG_OK;
}

int score_s3_impl(const int8_t* X8, int64_t R, int32_t N, int64_t ldx, int32_t S, const float* q, double* out64, float* out32,
                  void* ws, int64_t ws_bytes, hipStream_t st) {
    if (R < 0 || N < 2 || ldx < N || S < 1) return fail(EPG_ERR_INVALID_ARG, "score_s3: bad shape");
    if (S > 31) return fail(EPG_ERR_UNSUPPORTED, "score_s3: S=%d > 31 not supported by this build", S);
    if (R == 0) return EPG_OK;
    if (!X8 || !q || !ws) return fail(EPG_ERR_INVALID_ARG, "score_s3: NULL argument");
    const size_t shmem = (size_t)N * S * 4 + (size_t)S3_SCORE_SLICE * 3;
    if (shmem > 160 * 1024 - 64) return fail(EPG_ERR_UNSUPPORTED, "score_s3: N*S = %d exceeds the LDS tile (N*S*4 + 8 KB <= 160 KB)", N * S);
    const int64_t tb = s3_table_bytes(N, S);
    const int64_t need = tb + (out64 ? 0 : align_up(R * S * 8, 256));
    if (ws_bytes < need) return fail(EPG_ERR_WORKSPACE, "score_s3: workspace %lld < %lld bytes", (long long)ws_bytes, (long long)need);
    float* T2 = reinterpret_cast<float*>(ws);
    double* acc = out64 ? out64 : reinterpret_cast<double*>(reinterpret_cast<char*>(ws) + tb);
    const long total = (long)N * N * S * S;
    hipLaunchKernelGGL(k_s3_table, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, q, N, S, T2);
    EPG_LAUNCH_CHECK("k_s3_table");
    EPG_HIP(hipMemsetAsync(acc, 0, (size_t)R * S * 8, st));
    const long nslices = (R + S3_SCORE_SLICE - 1) / S3_SCORE_SLICE;
    if (nslices * N > 0x7fffffffL) return fail(EPG_ERR_UNSUPPORTED, "score_s3: R*N too large for one call");
    EPG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_s3_score), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    hipLaunchKernelGGL(k_s3_score, dim3((unsigned)(nslices * N)), dim3(S3S_THREADS), shmem, st, reinterpret_cast<const char*>(X8), (long)R, N,
                       (long)ldx, S, T2, acc);
    EPG_LAUNCH_CHECK("k_s3_score");
    if (out32) {
        long blocks = ((long)R * S + 255) / 256;
        if (blocks > num_cus() * 8L) blocks = num_cus() * 8L;
        hipLaunchKernelGGL(k_f64_to_f32, dim3((unsigned)blocks), dim3(256), 0, st, acc, (long)R * S, out32);
        EPG_LAUNCH_CHECK("k_f64_to_f32");
    }
    return EPG_OK;
}

}  // namespace epg
