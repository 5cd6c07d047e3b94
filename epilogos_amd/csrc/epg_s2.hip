// S2 path (state-pair saliency) from cached per-bin histograms, plus the paired-mode extras.  gfx950 only.
#include "epg_common.h"

namespace epg {

// ---------------------------------------------------------------------------------------------------------------
// S2 expected: C[i,j] += sum_b h_i*h_j (i != j), h_i*(h_i - 1) (i == j)      (expected.py:146-158 s2Calc)
// One thread per ordered pair (i, j) (up to 4 pairs per thread for S <= 31), bins staged through LDS in
// batches; exact integer arithmetic (u64 accumulators).  Reads 2*S bytes per bin.
// ---------------------------------------------------------------------------------------------------------------
constexpr int S2H_BATCH = 128;

__global__ __launch_bounds__(256) void k_s2_hist_from_binhist(const u16* __restrict__ H, long R, int S,
                                                               u64* __restrict__ counts) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u32* s_h = reinterpret_cast<u32*>(smem);  // [S2H_BATCH][S] as u32
    const int SS = S * S;
    u64 acc[4] = {0, 0, 0, 0};
    int pi[4], pj[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int p = threadIdx.x + 256 * k;
        pi[k] = p < SS ? p / S : -1;
        pj[k] = p < SS ? p % S : 0;
    }
    const long nb = (R + S2H_BATCH - 1) / S2H_BATCH;
    for (long batch = blockIdx.x; batch < nb; batch += gridDim.x) {
        const long r0 = batch * S2H_BATCH;
        const int rows = (int)((R - r0) < S2H_BATCH ? (R - r0) : S2H_BATCH);
        __syncthreads();
        for (int e = threadIdx.x; e < rows * S; e += 256) s_h[e] = H[r0 * S + e];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (pi[k] < 0) continue;
            const int i = pi[k], j = pj[k];
            const u32 dj = (i == j) ? 1u : 0u;
            u64 a = 0;
            for (int r = 0; r < rows; ++r) {
                const u32 hi = s_h[r * S + i];
                const u32 hj = s_h[r * S + j];
                a += (u64)hi * (u64)(hj - (hi ? dj : 0u));  // h_i == 0 contributes 0 (no u32 wrap)
            }
            acc[k] += a;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (pi[k] >= 0 && acc[k]) atomicAdd(&counts[pi[k] * S + pj[k]], acc[k]);
}

// ---------------------------------------------------------------------------------------------------------------
// S2 score tables: LH[c] = log2(c) for c = 0..maxc (LH[0] = 0, never used);
// LPQ[i,j] = log2(perms * q[i,j]) or +inf-marker where q == 0.
// log2(p/q) with p = num/perms is evaluated as LH[a] + LH[b] - LPQ[i,j], num = a*b (DESIGN.md, S2 numerics).
// ---------------------------------------------------------------------------------------------------------------
constexpr double LPQ_MASKED = 1e300;

__global__ void k_s2_tables(const float* __restrict__ q, int S, long perms, int maxc, double* __restrict__ LH,
                            double* __restrict__ LPQ) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t <= maxc) LH[t] = t > 0 ? log2((double)t) : 0.0;
    if (t < S * S) {
        const double qd = (double)q[t];
        LPQ[t] = qd == 0.0 ? LPQ_MASKED : log2((double)perms * qd);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// S2 score: score[b, j] = sum_{i ascending} kl(p_ij, q_ij), p_ij = (h_i*h_j - [i==j]*h_i) / perms
// (scores.py:404-412,443-451).  A wave scores BPW = 64 / S bins at a time, lane = (bin, j); the bin's
// histogram and log2 of its counts sit in LDS and are broadcast along i.
// ---------------------------------------------------------------------------------------------------------------
template <typename OT>
__global__ __launch_bounds__(256) void k_score_s2_from_hist(const u16* __restrict__ H, long R, int S, double inv_perms,
                                                             int maxc, const double* __restrict__ gLH,
                                                             const double* __restrict__ gLPQ, OT* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // layout: LPQ [S*S] doubles | per wave: lh [BPW*S] doubles | per wave: hh [BPW*S] u32
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int BPW = 64 / S;
    double* s_lpq = reinterpret_cast<double*>(smem);
    double* s_lh = s_lpq + S * S + wave * BPW * S;
    u32* s_hh = reinterpret_cast<u32*>(s_lpq + S * S + 4 * BPW * S) + wave * BPW * S;
    for (int e = threadIdx.x; e < S * S; e += 256) s_lpq[e] = gLPQ[e];
    __syncthreads();

    const int bsub = lane / S, j = lane - bsub * S;
    const bool active = bsub < BPW;
    const long ngroups = (R + BPW - 1) / BPW;
    for (long grp = (long)blockIdx.x * 4 + wave; grp < ngroups; grp += (long)gridDim.x * 4) {
        const long row = grp * BPW + bsub;
        const bool valid = active && row < R;
        u32 hj = 0;
        if (valid) hj = H[row * S + j];
        if (hj > (u32)maxc) hj = (u32)maxc;  // cannot happen for consistent inputs; keeps the gather in bounds
        const double lj = gLH[hj];
        const double ljm1 = gLH[hj ? hj - 1 : 0];
        __builtin_amdgcn_wave_barrier();
        if (active) {
            s_hh[bsub * S + j] = hj;
            s_lh[bsub * S + j] = lj;
        }
        __builtin_amdgcn_wave_barrier();
        double acc = 0.0;
        if (valid && hj) {
            const u32* hh = s_hh + bsub * S;
            const double* lh = s_lh + bsub * S;
            for (int i = 0; i < S; ++i) {
                const u32 hi = hh[i];
                const u32 other = (i == j) ? hj - 1 : hj;
                const u32 num = hi * other;  // <= 65535^2 < 2^32
                const double lpq = s_lpq[i * S + j];
                if (num != 0 && lpq != LPQ_MASKED) {
                    const double p = (double)num * inv_perms;
                    const double lg = (lh[i] + ((i == j) ? ljm1 : lj)) - lpq;
                    acc += p * lg;
                }
            }
        }
        if (valid) out[row * S + j] = (OT)acc;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// paired extras
// ---------------------------------------------------------------------------------------------------------------
// delta = a - b; dist[b] = sum_s delta^2 * sign(sum_s delta) in float32, with numpy's pairwise_sum order for a
// contiguous float32 row (8 partial sums, tree-combined, remainder appended) so the float32 result matches
// np.sum(axis=1) of scores.py:231-232 bit for bit.
__device__ __forceinline__ float np_rowsum_f32(const float* v, int n) {
    if (n < 8) {
        float r = 0.f;
        for (int i = 0; i < n; ++i) r += v[i];
        return r;
    }
    float r[8];
    for (int k = 0; k < 8; ++k) r[k] = v[k];
    int i = 8;
    for (; i < n - (n % 8); i += 8)
        for (int k = 0; k < 8; ++k) r[k] += v[i + k];
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += v[i];
    return res;
}

__global__ __launch_bounds__(256) void k_pair_finish(const float* __restrict__ a, const float* __restrict__ b, long R,
                                                      int S, float* __restrict__ delta, float* __restrict__ dist) {
    const long row = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= R) return;
    float d[32], d2[32];
    for (int s = 0; s < S; ++s) {
        d[s] = a[row * S + s] - b[row * S + s];
        d2[s] = d[s] * d[s];
        delta[row * S + s] = d[s];
    }
    if (dist) {
        const float sd = np_rowsum_f32(d, S);
        const float sq = np_rowsum_f32(d2, S);
        const float sg = sd > 0.f ? 1.f : (sd < 0.f ? -1.f : sd);  // np.sign: 0 -> 0, nan -> nan
        dist[row] = sq * sg;
    }
}

// quiescent from the two groups' histograms: every column of A and of B equals qstate  (scores.py:294-303)
__global__ __launch_bounds__(256) void k_quiescent_from_hist(const u16* __restrict__ HA, int NA, const u16* __restrict__ HB,
                                                              int NB, long R, int S, int qstate, uint8_t* __restrict__ mask) {
    const long row = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= R) return;
    mask[row] = (HA[row * S + qstate] == NA && HB[row * S + qstate] == NB) ? 1 : 0;
}

// quiescent straight from the state matrices: 16 lanes per row
__global__ __launch_bounds__(256) void k_quiescent(const char* __restrict__ XA, int NA, long ldxa, const char* __restrict__ XB,
                                                    int NB, long ldxb, long R, int qstate, uint8_t* __restrict__ mask) {
    const int sub = threadIdx.x & 15;
    const long row = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    bool ok = true;
    if (row < R) {
        const char* pa = XA + row * ldxa;
        for (int n = sub; n < NA; n += 16) ok = ok && ((int)(signed char)pa[n] == qstate);
        const char* pb = XB + row * ldxb;
        for (int n = sub; n < NB; n += 16) ok = ok && ((int)(signed char)pb[n] == qstate);
    }
    int v = ok ? 1 : 0;
    for (int off = 8; off > 0; off >>= 1) v &= __shfl_xor(v, off, 16);
    if (row < R && sub == 0) mask[row] = (uint8_t)v;
}

// ---------------------------------------------------------------------------------------------------------------
// host-side dispatch
// ---------------------------------------------------------------------------------------------------------------
int hist_s2_from_binhist_impl(const uint16_t* H, int64_t R, int32_t S, int64_t* counts, hipStream_t st) {
    if (R < 0 || S < 1 || S > 31) return fail(EPG_ERR_INVALID_ARG, "hist_s2: bad shape R=%lld S=%d", (long long)R, S);
    if (R == 0) return EPG_OK;
    if (!H || !counts) return fail(EPG_ERR_INVALID_ARG, "hist_s2: NULL argument");
    const long nb = (R + S2H_BATCH - 1) / S2H_BATCH;
    long blocks = nb < num_cus() * 4L ? nb : num_cus() * 4L;
    const size_t shmem = (size_t)S2H_BATCH * S * 4;
    hipLaunchKernelGGL(k_s2_hist_from_binhist, dim3((int)blocks), dim3(256), shmem, st, H, (long)R, S,
                       reinterpret_cast<u64*>(counts));
    EPG_LAUNCH_CHECK("k_s2_hist_from_binhist");
    return EPG_OK;
}

int64_t s2_table_bytes(int maxc, int S) { return align_up((int64_t)(maxc + 1) * 8, 256) + align_up((int64_t)S * S * 8, 256); }

int score_s2_from_hist_impl(const uint16_t* H, int64_t R, int32_t N, int32_t S, int64_t perms, const float* q,
                            double* out64, float* out32, void* ws, int64_t ws_bytes, hipStream_t st) {
    if (R < 0 || N < 1 || S < 1 || S > 31 || perms < 1) return fail(EPG_ERR_INVALID_ARG, "score_s2: bad shape R=%lld N=%d S=%d perms=%lld", (long long)R, N, S, (long long)perms);
    if (R == 0) return EPG_OK;
    if (!H || !q || !ws) return fail(EPG_ERR_INVALID_ARG, "score_s2: NULL argument");
    if (ws_bytes < s2_table_bytes(N, S)) return fail(EPG_ERR_WORKSPACE, "score_s2: workspace %lld < %lld bytes", (long long)ws_bytes, (long long)s2_table_bytes(N, S));
    double* LH = reinterpret_cast<double*>(ws);
    double* LPQ = reinterpret_cast<double*>(reinterpret_cast<char*>(ws) + align_up((int64_t)(N + 1) * 8, 256));
    const int nt = (N + 1) > S * S ? (N + 1) : S * S;
    hipLaunchKernelGGL(k_s2_tables, dim3((nt + 255) / 256), dim3(256), 0, st, q, S, (long)perms, N, LH, LPQ);
    EPG_LAUNCH_CHECK("k_s2_tables");
    const int BPW = 64 / S;
    const long ngroups = (R + BPW - 1) / BPW;
    long blocks = (ngroups + 3) / 4;
    if (blocks > num_cus() * 8L) blocks = num_cus() * 8L;
    const size_t shmem = (size_t)S * S * 8 + (size_t)4 * BPW * S * 8 + (size_t)4 * BPW * S * 4;
    const double inv = 1.0 / (double)perms;
    if (out32) hipLaunchKernelGGL((k_score_s2_from_hist<float>), dim3((int)blocks), dim3(256), shmem, st, H, (long)R, S, inv, N, LH, LPQ, out32);
    if (out64) hipLaunchKernelGGL((k_score_s2_from_hist<double>), dim3((int)blocks), dim3(256), shmem, st, H, (long)R, S, inv, N, LH, LPQ, out64);
    EPG_LAUNCH_CHECK("k_score_s2_from_hist");
    return EPG_OK;
}

// "%.5f" then strtod then float32: v * 1e5 is exact in double (24-bit x 17-bit significands), rint is half-even on that
// exact value like the correctly rounded decimal conversion, and k / 1e5 is the correctly rounded quotient, i.e. the
// double nearest to the decimal k * 10^-5 that the parser produces.
__device__ __forceinline__ float text_roundtrip_f5(float v) { return (float)(rint((double)v * 1e5) / 1e5); }

__global__ __launch_bounds__(256) void k_pair_metrics(const float* __restrict__ delta, long R, int S, int roundtrip,
                                                       float* __restrict__ dist, int* __restrict__ maxdiff) {
    const long row = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= R) return;
    float sq = 0.f, sd = 0.f, best = -1.f;
    int arg = S;
    for (int s = 0; s < S; ++s) {
        float d = delta[row * S + s];
        if (roundtrip) d = text_roundtrip_f5(d);
        sq = __fadd_rn(sq, __fmul_rn(d, d));          // np.square, then a separate add: no fused multiply-add
        sd = __fadd_rn(sd, d);
        if (fabsf(d) >= best) { best = fabsf(d); arg = s + 1; }   // >= : ties go to the higher state
    }
    const float sg = sd > 0.f ? 1.f : (sd < 0.f ? -1.f : sd);     // np.sign
    dist[row] = __fmul_rn(sq, sg);
    maxdiff[row] = arg;
}

int pair_metrics_impl(const float* delta, int64_t R, int32_t S, int32_t roundtrip, float* dist, int32_t* maxdiff, hipStream_t st) {
    if (R < 0 || S < 1) return fail(EPG_ERR_INVALID_ARG, "pair_metrics: bad shape");
    if (R == 0) return EPG_OK;
    if (!delta || !dist || !maxdiff) return fail(EPG_ERR_INVALID_ARG, "pair_metrics: NULL argument");
    hipLaunchKernelGGL(k_pair_metrics, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, st, delta, (long)R, S, roundtrip, dist, maxdiff);
    EPG_LAUNCH_CHECK("k_pair_metrics");
    return EPG_OK;
}

int pair_finish_impl(const float* a, const float* b, int64_t R, int32_t S, float* delta, float* dist, hipStream_t st) {
    if (R < 0 || S < 1 || S > 32) return fail(EPG_ERR_INVALID_ARG, "pair_finish: bad shape");
    if (R == 0) return EPG_OK;
    if (!a || !b || !delta) return fail(EPG_ERR_INVALID_ARG, "pair_finish: NULL argument");
    hipLaunchKernelGGL(k_pair_finish, dim3((int)((R + 255) / 256)), dim3(256), 0, st, a, b, (long)R, S, delta, dist);
    EPG_LAUNCH_CHECK("k_pair_finish");
    return EPG_OK;
}

int quiescent_impl(const int8_t* XA, int32_t NA, int64_t ldxa, const int8_t* XB, int32_t NB, int64_t ldxb, int64_t R,
                   int32_t qstate, uint8_t* mask, hipStream_t st) {
    if (R < 0 || NA < 1 || NB < 1 || ldxa < NA || ldxb < NB) return fail(EPG_ERR_INVALID_ARG, "quiescent: bad shape");
    if (R == 0) return EPG_OK;
    if (!XA || !XB || !mask) return fail(EPG_ERR_INVALID_ARG, "quiescent: NULL argument");
    if (qstate < 0) {  // filtering off (run.py:113: -q 0 -> -1): nothing is quiescent
        EPG_HIP(hipMemsetAsync(mask, 0, (size_t)R, st));
        return EPG_OK;
    }
    const long threads = (long)R * 16;
    hipLaunchKernelGGL(k_quiescent, dim3((int)((threads + 255) / 256)), dim3(256), 0, st, reinterpret_cast<const char*>(XA), NA,
                       (long)ldxa, reinterpret_cast<const char*>(XB), NB, (long)ldxb, (long)R, qstate, mask);
    EPG_LAUNCH_CHECK("k_quiescent");
    return EPG_OK;
}

}  // namespace epg
