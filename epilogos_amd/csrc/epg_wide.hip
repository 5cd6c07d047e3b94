// State models of 32 .. 127 states (round 3).  gfx950 only.
//
// The reference takes any numStates (helpers.py:9-17); every model it bundles has 15, 18 or 25 states, and the kernels of
// this engine are built around that: a five-bit state decode, 31 packed counters, (S, S) tiles in registers.  So that the
// ABI never refuses a model the reference accepts, the entry points hand a model above 31 states to the kernels below --
// the same arithmetic, restated plainly, bytes decoded whole (0 .. S - 1 a state, anything else "not a state"), no claim to
// speed: LDS or global atomics for counts, one float64 log2 per S2 term, the S3 table evaluated per term.  The integer results
// are exact, the float64 scores follow the reference's order of summation where it has one (S2: ascending i), so the parity
// tests use the same tolerances as the fast kernels' (tests/test_hip_wide_models.py).  Per-bin counts stay uint16 (N <= 65 535).
#include "epg_common.h"

namespace epg {

constexpr int W_SMAX = 127;

// ---------------------------------------------------------------------------------------------------------------
// S2 counts from the per-bin histograms: C[i, j] += sum_b h_i h_j - [i == j] h_i  (expected.py:137,146-158), optionally of
// h = h_A + h_B (paired mode).  A block stages 128 bins; a thread owns cells t, t + 256, ... of the upper triangle incl. the
// diagonal and mirrors them at the end.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_w_hist_s2(const u16* __restrict__ H, const u16* __restrict__ H2, long R, int S, u64* __restrict__ counts) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u16* sh = reinterpret_cast<u16*>(smem);                        // [128][S]
    const int ncell = S * (S + 1) / 2;
    for (long r0 = (long)blockIdx.x * 128; r0 < R; r0 += (long)gridDim.x * 128) {
        const int rows = (int)(R - r0 < 128 ? R - r0 : 128);
        __syncthreads();
        for (int e = threadIdx.x; e < rows * S; e += 256) {
            u32 v = H[r0 * S + e];
            if (H2) v += H2[r0 * S + e];
            sh[e] = (u16)v;
        }
        __syncthreads();
        for (int cell = threadIdx.x; cell < ncell; cell += 256) {
            // cell -> (i <= j): row i of the triangle starts at i S - i (i - 1) / 2
            int i = 0;
            while ((i + 1) * S - (i + 1) * i / 2 <= cell) ++i;
            const int j = i + (cell - (i * S - i * (i - 1) / 2));
            u64 acc = 0;
            for (int r = 0; r < rows; ++r) {
                const u32 hi = sh[r * S + i], hj = sh[r * S + j];
                acc += (u64)hi * (i == j ? (hi ? hi - 1 : 0) : hj);
            }
            if (acc) {
                atomicAdd(&counts[i * S + j], acc);
                if (i != j) atomicAdd(&counts[j * S + i], acc);
            }
        }
    }
}

int wide_hist_s2_from_binhist(const uint16_t* H, const uint16_t* H2, int64_t R, int32_t S, int64_t* counts, hipStream_t st) {
    long blocks = (R + 127) / 128;
    if (blocks > num_cus() * 4L) blocks = num_cus() * 4L;
    hipLaunchKernelGGL(k_w_hist_s2, dim3((unsigned)blocks), dim3(256), (size_t)128 * S * 2, st, H, H2, (long)R, S, reinterpret_cast<u64*>(counts));
    EPG_LAUNCH_CHECK("k_w_hist_s2");
    return EPG_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// S2 scores from the per-bin histograms: p[i, j] = (h_i h_j - [i == j] h_i) / perms, score[j] = sum_i kl(p[i, j], q[i, j]) in
// ascending i (scores.py:371,412,426-452).  One wave per bin, a lane takes the columns j = lane, lane + 64.
// ---------------------------------------------------------------------------------------------------------------
template <typename OT>
__global__ __launch_bounds__(256) void k_w_score_s2(const u16* __restrict__ H, long R, int S, double perms, const float* __restrict__ q,
                                                   OT* __restrict__ out) {
    __shared__ u16 sh[4][128];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long r = (long)blockIdx.x * 4 + wave; r < R; r += (long)gridDim.x * 4) {
        for (int s = lane; s < S; s += 64) sh[wave][s] = H[r * S + s];
        __builtin_amdgcn_wave_barrier();
        for (int j = lane; j < S; j += 64) {
            const u32 hj = sh[wave][j];
            double acc = 0.0;
            for (int i = 0; i < S; ++i) {
                const u32 hi = sh[wave][i];
                const unsigned long long num = (unsigned long long)hi * (i == j ? (hi ? hi - 1 : 0) : hj);
                acc += kl_term((double)num / perms, (double)q[i * S + j]);   // a true division, like scores.py:449-451
            }
            out[r * S + j] = (OT)acc;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

int wide_score_s2_from_hist(const uint16_t* H, int64_t R, int32_t S, int64_t perms, const float* q, double* out64, float* out32, hipStream_t st) {
    long blocks = (R + 3) / 4;
    if (blocks > num_cus() * 16L) blocks = num_cus() * 16L;
    if (out32) hipLaunchKernelGGL(k_w_score_s2<float>, dim3((unsigned)blocks), dim3(256), 0, st, H, (long)R, S, (double)perms, q, out32);
    if (out64) hipLaunchKernelGGL(k_w_score_s2<double>, dim3((unsigned)blocks), dim3(256), 0, st, H, (long)R, S, (double)perms, q, out64);
    EPG_LAUNCH_CHECK("k_w_score_s2");
    return EPG_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// S3 counts: C[a, b, x_a, x_b] += 1 for a != b (expected.py:183-200).  A thread takes a (bin, a) and walks b.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_w_hist_s3(const unsigned char* __restrict__ X, long R, int N, long ldx, int S, int* __restrict__ counts) {
    const long total = R * N;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long r = e / N;
        const int a = (int)(e - r * N);
        const unsigned char* row = X + r * ldx;
        const int i = row[a];
        if (i >= S) continue;
        for (int b = 0; b < N; ++b) {
            const int j = row[b];
            if (b != a && j < S) atomicAdd(&counts[(((long)a * N + b) * S + i) * S + j], 1);
        }
    }
}

int wide_hist_s3(const int8_t* X, int64_t R, int32_t N, int64_t ldx, int32_t S, int32_t* counts, hipStream_t st) {
    long blocks = (R * N + 255) / 256;
    if (blocks > num_cus() * 16L) blocks = num_cus() * 16L;
    hipLaunchKernelGGL(k_w_hist_s3, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<const unsigned char*>(X), (long)R, N, (long)ldx, S, counts);
    EPG_LAUNCH_CHECK("k_w_hist_s3");
    return EPG_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// S3 scores: T = kl(float32(1) / P, q) in float32 (scores.py:479-480); score[bin, x_b] += sum_{a != b} T[a, b, x_a, x_b].  A thread
// takes a (bin, b), sums its terms in float64 and adds them to the (bin, state) cell as a 2^-50 fixed-point integer -- integer
// adds commute, so the scores do not depend on the order of the atomics (like every other S3 score kernel of this library).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float w_s3_t(float qv, float obs) { return s3_table_entry(qv, obs); }   // (epg_common.h)

__global__ __launch_bounds__(256) void k_w_score_s3(const unsigned char* __restrict__ X, long R, int N, long ldx, int S, const float* __restrict__ q,
                                                   long long* __restrict__ cells) {
    const long total = R * N;
    const float obs = 1.0f / (float)((long)N * (N - 1));
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long r = e / N;
        const int b = (int)(e - r * N);
        const unsigned char* row = X + r * ldx;
        const int j = row[b];
        if (j >= S) continue;
        double acc = 0.0;
        for (int a = 0; a < N; ++a) {
            const int i = row[a];
            if (a != b && i < S) acc += (double)w_s3_t(q[(((long)a * N + b) * S + i) * S + j], obs);
        }
        atomicAdd(reinterpret_cast<unsigned long long*>(&cells[r * S + j]), (unsigned long long)__double2ll_rn(acc * 1125899906842624.0));
    }
}

__global__ __launch_bounds__(256) void k_w_fix_finish(double* __restrict__ cells, long n, int want64, float* __restrict__ out32) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const double v = (double)reinterpret_cast<const long long*>(cells)[i] * (1.0 / 1125899906842624.0);
        if (want64) cells[i] = v;
        if (out32) out32[i] = (float)v;
    }
}

int wide_score_s3(const int8_t* X, int64_t R, int32_t N, int64_t ldx, int32_t S, const float* q, double* out64, float* out32, void* ws,
                  int64_t ws_bytes, hipStream_t st) {
    const int64_t need = out64 ? 0 : align_up(R * S * 8, 256);
    if (ws_bytes < need) return fail(EPG_ERR_WORKSPACE, "score_s3: workspace %lld < %lld bytes", (long long)ws_bytes, (long long)need);
    double* acc = out64 ? out64 : reinterpret_cast<double*>(ws);
    EPG_HIP(hipMemsetAsync(acc, 0, (size_t)R * S * 8, st));
    long blocks = (R * N + 255) / 256;
    if (blocks > num_cus() * 16L) blocks = num_cus() * 16L;
    hipLaunchKernelGGL(k_w_score_s3, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<const unsigned char*>(X), (long)R, N, (long)ldx, S, q,
                       reinterpret_cast<long long*>(acc));
    EPG_LAUNCH_CHECK("k_w_score_s3");
    blocks = (R * S + 255) / 256;
    if (blocks > num_cus() * 8L) blocks = num_cus() * 8L;
    hipLaunchKernelGGL(k_w_fix_finish, dim3((unsigned)blocks), dim3(256), 0, st, acc, (long)R * S, out64 ? 1 : 0, out32);
    EPG_LAUNCH_CHECK("k_w_fix_finish");
    return EPG_OK;
}

}  // namespace epg
