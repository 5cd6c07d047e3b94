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
//   T[a,b,i,j] = kl(float32(1)/P, q[a,b,i,j]) in float32 (scores.py:479-480), kept in q's layout, with a zero
//   diagonal a == b and zero rows up to a multiple of S3S_ACH biosamples;
//   score[bin, s] = sum_{b: x_b == s} sum_{a != b} T[a, b, x_a, s]   (closed form of scores.py:496-498).
//   k_s3_score: a block owns one b and a slice of bins of the transposed matrix; it streams over a in phases of
//   S3S_ACH, staging the tables T[a][b] into LDS as padded tiles, and every thread gathers tab[a][x_b][x_a] for its
//   16 bins; float32 partial sums of one phase are folded into float64 accumulators.  See DESIGN.md.
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

// T[a][b][i][j] = float32 kl(float32(1)/P, q[a,b,i,j]); float32 operands and results like scores.py:479-480, every operation
// correctly rounded (s3_table_entry, epg_common.h).  Same layout as q.
// rows a in [N, Nceil) and the diagonal a == b are zero, so that the score kernel needs no predicates
__global__ void k_s3_table(const float* __restrict__ q, int N, int Nceil, int S, float* __restrict__ T) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long SS = (long)S * S, total = (long)N * N * SS;
    if (e >= (long)Nceil * N * SS) return;
    const long ab = e / SS;
    if (e >= total || ab / N == ab % N) { T[e] = 0.0f; return; }
    const float qv = q[e];
    const float obs = 1.0f / (float)((long)N * (N - 1));
    const float v = s3_table_entry(qv, obs);
    T[e] = v;
}

// ---------------------------------------------------------------------------------------------------------------
// S3 score, a-streaming form.  score[bin, x_b] += sum_{a != b} T[a, b, x_a, x_b]  for every biosample b.
// A block owns one b and a slice of S3S_THREADS*16 bins; every thread owns 16 consecutive bins: their x_b (16 bytes of
// XT[b], loop invariant) and 16 accumulators.  The block then streams over a: per a one 16-byte load of XT[a] per
// thread (1 KiB per wave, fully coalesced) and 16 gathers from the 18x18 table T[a][b] staged in LDS, padded and
// transposed to [x_b][32] so that "not a state" (31 in the sanitised XT) reads a zero and needs no branch.
// No per-state passes, no bin lists, no cross-lane reduction, no dependent-load chains (the loads of the next a do not
// depend on anything).  Tables are staged S3S_ACH biosamples at a time (one barrier per S3S_ACH values of a).
// Partial sums of S3S_ACH float terms are folded into float64 accumulators (error ~1e-7 relative).
// ---------------------------------------------------------------------------------------------------------------
constexpr int S3S_THREADS = 256;
constexpr int S3S_BPT = 16;                       // bins per thread
constexpr int S3S_SLICE = S3S_THREADS * S3S_BPT;  // bins per block
constexpr int S3S_ACH = 4;                        // biosamples a per staging phase
// Floats per table row.  Rows are stored in the order 31 - x_b, so the LDS bank of a gather is (x_a - x_b) mod 32.
// With rows in natural order the bank is x_a + x_b and the pairs (i, j) and (j, i) -- equally likely for every
// state distribution -- always collide; measured: half of the LDS cycles were bank conflicts.
constexpr int S3S_LD = 33;
constexpr int S3S_TAB = 32 * S3S_LD * 4;          // bytes of one padded table
constexpr int S3S_DUMMY = 31 * S3S_LD + 32;       // a slot no gather reads (column 32)
constexpr double S3S_FIX = 1125899906842624.0;    // 2^50: fixed-point unit of the score cells (see the end of k_s3_score)

// Everything the loop touches is unconditional: XT4 holds 4*state (31 -> 124 for "not a state" and for padded bins),
// the table has zero rows for a >= N and a zero diagonal a == b, invalid staging elements go to a dummy LDS slot and
// out-of-range addresses are clamped.  (With per-element predicates hipcc turned the loads into flat loads with
// selected addresses and spilled the default value to scratch.)  Per-thread offsets are unsigned 32-bit byte offsets
// from wave-uniform bases, which is the scalar-base + vector-offset form of global_load.
template <int STG>
struct S3Stage {
    u32 soff[STG], doff[STG];                     // byte offsets: source inside T[a0][b], slot inside a phase buffer
    float v[STG];
};

__device__ __forceinline__ float ldf(const char* base, u32 off) { return *reinterpret_cast<const float*>(base + off); }

template <int PH, int STG>
__device__ __forceinline__ void s3_phase(const char* __restrict__ XTs, long Rp, u32 toff, int N, int Nceil, int a0,
                                         const char* __restrict__ Tb, long a_stride, char* tabc, S3Stage<STG>& sg,
                                         const u32 (&rowoff)[S3S_BPT], uint4 (&raw)[S3S_ACH], double (&acc)[S3S_BPT]) {
    // tables of the next phase (requested one phase ago) -> the other buffer; then request the phase after
    char* dst = tabc + (PH ^ 1) * S3S_ACH * S3S_TAB;
#pragma unroll
    for (int k = 0; k < STG; ++k) *reinterpret_cast<float*>(dst + sg.doff[k]) = sg.v[k];
    if (a0 + 2 * S3S_ACH < Nceil) {                // wave-uniform
        const char* src = Tb + (long)(a0 + 2 * S3S_ACH) * a_stride;
#pragma unroll
        for (int k = 0; k < STG; ++k) sg.v[k] = ldf(src, sg.soff[k]);
    }
    uint4 nraw[S3S_ACH];
#pragma unroll
    for (int ai = 0; ai < S3S_ACH; ++ai) {
        int a = a0 + S3S_ACH + ai;
        a = a < N ? a : N - 1;                     // rows past N read a valid row; their tables are zero
        nraw[ai] = *reinterpret_cast<const uint4*>(XTs + (long)a * Rp + toff);
    }
    float part[S3S_BPT];
#pragma unroll
    for (int u = 0; u < S3S_BPT; ++u) part[u] = 0.f;
#pragma unroll
    for (int ai = 0; ai < S3S_ACH; ++ai) {
        const u32 w[4] = {raw[ai].x, raw[ai].y, raw[ai].z, raw[ai].w};
#pragma unroll
        for (int u = 0; u < S3S_BPT; ++u) {
            const u32 xa4 = (w[u >> 2] >> (8 * (u & 3))) & 0xffu;              // 4 * x_a
            part[u] += *reinterpret_cast<const float*>(tabc + (rowoff[u] + xa4) + (PH * S3S_ACH + ai) * S3S_TAB);
        }
    }
#pragma unroll
    for (int u = 0; u < S3S_BPT; ++u) acc[u] += (double)part[u];
#pragma unroll
    for (int ai = 0; ai < S3S_ACH; ++ai) raw[ai] = nraw[ai];
    __syncthreads();
}

template <int STG>
__global__ __launch_bounds__(S3S_THREADS, STG <= 6 ? 3 : 2) void k_s3_score(const char* __restrict__ XT4, long Rp, long R, int N, int Nceil, int S,
                                                              const float* __restrict__ T, double* __restrict__ out64) {
    __shared__ float tab[2][S3S_ACH][32][S3S_LD];  // [phase][a][31 - x_b][x_a], zero outside S x S
    char* tabc = reinterpret_cast<char*>(&tab[0][0][0][0]);
    // b fastest: the N blocks of a slice share its XT rows in L2.  (Round 2 tried patches of PB biosamples x PS slices so that
    // the resident blocks also share table columns -- PMC shows 360 GB of fabric reads per 1 M bins --: 96 x 8 and 64 x 12
    // run the same 100 ms, 16 x 48 and 8 x 96 are slower; the kernel is bound by the LDS gathers and their VALU, not by memory.)
    const int b = blockIdx.x % N;
    const long slice0 = (long)(blockIdx.x / N) * S3S_SLICE;
    const long r0 = slice0 + (long)threadIdx.x * S3S_BPT;
    const char* XTs = XT4 + slice0;               // wave-uniform base of this slice
    const u32 toff = (u32)((r0 < Rp ? r0 : Rp - 16) - slice0);   // threads past the end gather from valid memory, write nothing
    const int SS = S * S;
    const long a_stride = (long)N * SS * 4;       // bytes between T[a][b] and T[a+1][b]

    for (int e = threadIdx.x; e < 2 * S3S_ACH * 32 * S3S_LD; e += S3S_THREADS) (&tab[0][0][0][0])[e] = 0.f;
    const uint4 xbv = *reinterpret_cast<const uint4*>(XTs + (long)b * Rp + toff);
    const u32 xbw[4] = {xbv.x, xbv.y, xbv.z, xbv.w};
    u32 rowoff[S3S_BPT];                            // byte offset of row x_b inside a padded table
#pragma unroll
    for (int u = 0; u < S3S_BPT; ++u) rowoff[u] = (124u - ((xbw[u >> 2] >> (8 * (u & 3))) & 0xffu)) * (u32)S3S_LD;   // 4 * (31 - x_b) * LD
    double acc[S3S_BPT];
#pragma unroll
    for (int u = 0; u < S3S_BPT; ++u) acc[u] = 0.0;

    // staging map of this thread, computed once: element e = tid + k*THREADS of a phase's ACH*S*S table values ->
    // source offset in T[.][b] and padded, transposed LDS slot (tab[..][a][x_b = j][x_a = i])
    S3Stage<STG> sg;
#pragma unroll
    for (int k = 0; k < STG; ++k) {
        const int e = threadIdx.x + k * S3S_THREADS;
        const int ai = e / SS, ij = e - ai * SS;
        const int i = ij / S, j = ij - i * S;
        const bool ok = e < S3S_ACH * SS;
        sg.soff[k] = ok ? (u32)(ai * N * SS + ij) * 4u : 0u;
        sg.doff[k] = (ok ? (u32)((ai * 32 + 31 - j) * S3S_LD + i) : (u32)S3S_DUMMY) * 4u;
    }
    const char* Tb = reinterpret_cast<const char*>(T + (long)b * SS);
    __syncthreads();
    {   // phase 0 directly, phase 1 requested
#pragma unroll
        for (int k = 0; k < STG; ++k) sg.v[k] = ldf(Tb, sg.soff[k]);
#pragma unroll
        for (int k = 0; k < STG; ++k) *reinterpret_cast<float*>(tabc + sg.doff[k]) = sg.v[k];
        if (S3S_ACH < Nceil) {
            const char* src = Tb + (long)S3S_ACH * a_stride;
#pragma unroll
            for (int k = 0; k < STG; ++k) sg.v[k] = ldf(src, sg.soff[k]);
        }
    }
    uint4 raw[S3S_ACH];
#pragma unroll
    for (int ai = 0; ai < S3S_ACH; ++ai) raw[ai] = *reinterpret_cast<const uint4*>(XTs + (long)(ai < N ? ai : N - 1) * Rp + toff);
    __syncthreads();
    for (int a0 = 0; a0 < Nceil; a0 += 2 * S3S_ACH) {
        s3_phase<0, STG>(XTs, Rp, toff, N, Nceil, a0, Tb, a_stride, tabc, sg, rowoff, raw, acc);
        if (a0 + S3S_ACH < Nceil) s3_phase<1, STG>(XTs, Rp, toff, N, Nceil, a0 + S3S_ACH, Tb, a_stride, tabc, sg, rowoff, raw, acc);
    }
#pragma unroll
    for (int u = 0; u < S3S_BPT; ++u) {
        const long row = r0 + u;
        const u32 xb = ((xbw[u >> 2] >> (8 * (u & 3))) & 0xffu) >> 2;
        // The 833 blocks (one per b) of a slice add into the same [bin, state] cells in whatever order they finish.  The
        // cells are therefore 64-bit FIXED-POINT integers (2^-50 units): integer addition is associative, so the score is
        // bit-identical from run to run and for any partition of the bins, like S1 and S2.  |score| < 2^13 by a wide
        // margin (real values stay below ~20); rounding a contribution costs <= 2^-51, 833 of them < 4e-13 absolute.
        if (row < R && xb < (u32)S)
            atomicAdd(reinterpret_cast<u64*>(&out64[row * S + xb]), (u64)__double2ll_rn(acc[u] * S3S_FIX));
    }
}

// fixed-point cells -> float64 in place and / or float32
__global__ __launch_bounds__(256) void k_s3_fix_finish(double* __restrict__ cells, long n, int want64, float* __restrict__ out32) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const double v = (double)reinterpret_cast<const long long*>(cells)[i] * (1.0 / S3S_FIX);
        if (want64) cells[i] = v;
        if (out32) out32[i] = (float)v;
    }
}

// ---------------------------------------------------------------------------------------------------------------
static int s3_ta(int S) {
    int ta = S3_LDS_BUDGET / (S3_TB * s3_lane_stride(S) * 2);
    if (ta > 4) ta = 4;
    return ta;
}

int64_t s3_xt_bytes(int64_t R, int N);
int transpose_states(const char* X, int64_t R, int32_t N, int64_t ldx, int32_t S, char* XT, int64_t Rp, int shift, hipStream_t st);
int64_t s3_gemm_ws_bytes(int64_t R, int N, int S);
int64_t s3_gemm_ws_min_bytes(int64_t R, int N, int S);
int hist_s3_gemm(const char* X, int64_t R, int32_t N, int64_t ldx, int32_t S, int32_t* counts, void* ws, int64_t ws_bytes, hipStream_t st);
bool s3_lanes_ok(int N, int S);
int64_t s3_lanes_ws_bytes(int64_t R, int N, int S);
int score_s3_lanes(const int8_t* X8, int64_t R, int32_t N, int64_t ldx, int32_t S, const float* q, double* out64, float* out32, void* ws,
                   int64_t ws_bytes, hipStream_t st);


int wide_hist_s3(const int8_t* X, int64_t R, int32_t N, int64_t ldx, int32_t S, int32_t* counts, hipStream_t st);
int wide_score_s3(const int8_t* X, int64_t R, int32_t N, int64_t ldx, int32_t S, const float* q, double* out64, float* out32, void* ws,
                  int64_t ws_bytes, hipStream_t st);

static int s3_nceil(int N) { return (N + S3S_ACH - 1) / S3S_ACH * S3S_ACH; }
int64_t s3_table_bytes(int N, int S) { return align_up((int64_t)s3_nceil(N) * N * S * S * 4, 256); }
int64_t s3_ws_bytes(int64_t R, int N, int S) {
    if (S > 31) return align_up(R * S * 8, 256) + 256;       // the wide models (epg_wide.hip): the fixed-point cells, nothing else
    // score: table + transposed state matrix + float64 accumulator; expected: transposed state matrix (+ a chunk of the
    // precomputed fp4 one-hot operand for the default kernel)
    int64_t score = s3_table_bytes(N, S) + s3_xt_bytes(R, N) + align_up(R * S * 8, 256);
    if (s3_lanes_ok(N, S) && s3_lanes_ws_bytes(R, N, S) > score) score = s3_lanes_ws_bytes(R, N, S);
    const int64_t expected = s3_gemm_ws_bytes(R, N, S);
    return score > expected ? score : expected;
}

int hist_s3_impl(const int8_t* X8, int64_t R, int32_t N, int64_t ldx, int32_t S, int32_t* counts, void* ws, int64_t ws_bytes,
                 hipStream_t st) {
    if (R < 0 || N < 2 || ldx < N || S < 1) return fail(EPG_ERR_INVALID_ARG, "hist_s3: bad shape R=%lld N=%d ldx=%lld S=%d", (long long)R, N, (long long)ldx, S);
    if (S > 127) return fail(EPG_ERR_UNSUPPORTED, "hist_s3: S=%d > 127 (states are int8)", S);
    if (R == 0) return EPG_OK;
    if (!X8 || !counts) return fail(EPG_ERR_INVALID_ARG, "hist_s3: NULL argument");
    if (ws && (reinterpret_cast<uintptr_t>(ws) & 15)) return fail(EPG_ERR_INVALID_ARG, "hist_s3: the workspace must be 16-byte aligned");
    if (S > 31) return wide_hist_s3(X8, R, N, ldx, S, counts, st);                      // the wide models: epg_wide.hip
    // matrix-core path (epg_s3_gemm.hip: the precomputed one-hot fp4 contraction) when the caller's workspace holds the transposed
    // matrix and at least a 16 K-bin chunk of the operand (S <= 30: padding rows use pattern 30); the LDS-counter kernel below is
    // the path for S = 31, for no / too small a workspace, and -- epg_test_force(3, 1) -- for the tests on any shape
    const bool force_lds = g_force[FORCE_S3_HIST_LDS] != 0;
    if (!force_lds && S <= 30 && ws && ws_bytes >= s3_gemm_ws_min_bytes(R, N, S))
        return hist_s3_gemm(reinterpret_cast<const char*>(X8), R, N, ldx, S, counts, ws, ws_bytes, st);
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
    if (S > 127) return fail(EPG_ERR_UNSUPPORTED, "score_s3: S=%d > 127 (states are int8)", S);
    if (R == 0) return EPG_OK;
    if (!X8 || !q || !ws) return fail(EPG_ERR_INVALID_ARG, "score_s3: NULL argument");
    if (reinterpret_cast<uintptr_t>(ws) & 15) return fail(EPG_ERR_INVALID_ARG, "score_s3: the workspace must be 16-byte aligned");
    if (S > 31) return wide_score_s3(X8, R, N, ldx, S, q, out64, out32, ws, ws_bytes, st);   // the wide models: epg_wide.hip
    // default: the biosample-lane kernel (epg_s3_lanes.hip, S <= 21) when the workspace holds its table; k_s3_score below is the
    // path for S > 21 and smaller workspaces (epg_test_force(1, 1): on any shape).  Round 3's modal-state kernel (per-biosample base
    // table + gathers only for biosamples off the modal state: the same integers from ~1/3 of the gathers, 1.43 x SLOWER -- a SIMD
    // pays ~3.7 cycles per wave instruction of any kind, tools/ubench/gpr_idx.hip) was deleted in round 5.
    const bool force_bins = g_force[FORCE_S3_SCORE_BINS] != 0;
    if (!force_bins && s3_lanes_ok(N, S) && ws_bytes >= s3_lanes_ws_bytes(R, N, S) - (out64 ? align_up(R * S * 8, 256) : 0))
        return score_s3_lanes(X8, R, N, ldx, S, q, out64, out32, ws, ws_bytes, st);
    const int64_t tb = s3_table_bytes(N, S), xtb = s3_xt_bytes(R, N);
    const int64_t need = tb + xtb + (out64 ? 0 : align_up(R * S * 8, 256));
    if (ws_bytes < need) return fail(EPG_ERR_WORKSPACE, "score_s3: workspace %lld < %lld bytes", (long long)ws_bytes, (long long)need);
    float* T = reinterpret_cast<float*>(ws);
    char* XT = reinterpret_cast<char*>(ws) + tb;
    double* acc = out64 ? out64 : reinterpret_cast<double*>(reinterpret_cast<char*>(ws) + tb + xtb);
    const int Nceil = s3_nceil(N);
    const long total = (long)Nceil * N * S * S;
    hipLaunchKernelGGL(k_s3_table, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, q, N, Nceil, S, T);
    EPG_LAUNCH_CHECK("k_s3_table");
    const long Rp = align_up(R, 32);
    int rc = transpose_states(reinterpret_cast<const char*>(X8), R, N, ldx, S, XT, Rp, 2, st);   // bytes = 4 * state
    if (rc) return rc;
    EPG_HIP(hipMemsetAsync(acc, 0, (size_t)R * S * 8, st));
    const long nslices = (R + S3S_SLICE - 1) / S3S_SLICE;
    if (nslices * N > 0x7fffffffL) return fail(EPG_ERR_UNSUPPORTED, "score_s3: R*N too large for one call");
    // staging elements per thread: ceil(ACH * S * S / THREADS)
    const int stg = (S3S_ACH * S * S + S3S_THREADS - 1) / S3S_THREADS;
    const dim3 grid((unsigned)(nslices * N)), block(S3S_THREADS);
    if (stg <= 6) hipLaunchKernelGGL(k_s3_score<6>, grid, block, 0, st, XT, Rp, (long)R, N, Nceil, S, T, acc);
    else if (stg <= 10) hipLaunchKernelGGL(k_s3_score<10>, grid, block, 0, st, XT, Rp, (long)R, N, Nceil, S, T, acc);
    else hipLaunchKernelGGL(k_s3_score<16>, grid, block, 0, st, XT, Rp, (long)R, N, Nceil, S, T, acc);
    EPG_LAUNCH_CHECK("k_s3_score");
    {
        long blocks = ((long)R * S + 255) / 256;
        if (blocks > num_cus() * 8L) blocks = num_cus() * 8L;
        hipLaunchKernelGGL(k_s3_fix_finish, dim3((unsigned)blocks), dim3(256), 0, st, acc, (long)R * S, out64 ? 1 : 0, out32);
        EPG_LAUNCH_CHECK("k_s3_fix_finish");
    }
    return EPG_OK;
}

}  // namespace epg
