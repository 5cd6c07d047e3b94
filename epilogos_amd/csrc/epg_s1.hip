// S1 path: per-bin histogram (K1), expected S1 counts, S1 score (direct and from cached histograms).
// gfx950 only.  See epg_count.h for the counting core and DESIGN.md for the roofline of each kernel.
#include "epg_count.h"

#include <stdlib.h>
#include <string.h>

namespace epg {

typedef unsigned short k1_v2u16 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------------------------------------
// K1: X[R, ldx] int8 -> H[R, S] uint16 (+ counts[S] += column sums).  HBM-bound: N bytes read, 2*S written per bin.
// Restates np.unique(row, return_counts=True) of scores.py:341/444 and expected.py:111-113,152.
// ---------------------------------------------------------------------------------------------------------------
// SC = number of states the counting core decodes (a template parameter: its loops live in registers); Sout <= SC = the
// state model's size = columns of H.  A model between two instantiations runs on the next larger one and only its own
// columns are stored (an occurrence of a state >= Sout is "not a state", like in the reference's inputs it cannot occur).
// FULL: Sout == SC at compile time (the reference's 15-, 18- and 25-state models).
// The body is shared by the one-matrix kernel and the several-parts kernel: `loop(enter, epilogue, finish)` runs the tile loop,
// enter(H) names the histogram array the following tiles belong to; nmax = the widest part (how often the packed uint16
// running counts must be flushed).
// PAIRS (S2 jobs, round 5): the same launch also adds the S2 state-pair counts of its bins into counts2[S * S] --
// C[i,j] += sum_b h_i h_j (i != j), h_i (h_i - 1) (i == j), expected.py:146-158 -- from the 32 rows of a super-tile that are
// staged in LDS for the H store anyway: re-packed as bin PAIRS per state (word (i, k) = counts of state i in bins 2k and 2k + 1)
// and contracted with v_dot2_u32_u16 in register tiles like k_s2_hist_wave (epg_s2.hip: a lane owns a 3 x 3 block of state pairs
// and every third pair word) -- ~5 wave instructions per bin on top of the counting core's ~45, instead of a pass of its own over
// H (0.23 ms per 15 M bins, at 0.29 of its bytes: VERDICT r4).  Needs FULL (a compile-time row width) and counts < 4096.
constexpr int K1P_LD = 17;                             // pair-matrix row stride in words (16 pair words + 1: rows on different banks)

template <int SC, bool FULL, bool PAIRS, typename Loop>
__device__ __forceinline__ void bin_hist_body(int Sout_, u64* __restrict__ counts, int nmax, u64* __restrict__ counts2, Loop&& loop) {
    constexpr int S = SC;
    constexpr int ND = (S + 1) / 2;
    const int Sout = FULL ? S : Sout_;
    const int ROWB = 2 * Sout;                         // bytes of one row of H
    __shared__ u64 s_cnt[S + 1];
    __shared__ __attribute__((aligned(16))) char s_stage[4][32 * 2 * S];
    constexpr int PG = (S + 2) / 3, PSP = 3 * PG, PROLES = PG * (PG + 1) / 2, PSLICE = PROLES <= 64 ? 64 / PROLES : 1;
    static_assert(!PAIRS || (FULL && PROLES <= 64), "the pair counts need a compile-time row width and at most 64 block roles");
    __shared__ u32 s_pair[PAIRS ? 4 * PSP * K1P_LD : 1];
    __shared__ u64 s_c2[PAIRS ? S * S : 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 3, b = lane >> 2;
    if (threadIdx.x <= S) s_cnt[threadIdx.x] = 0;
    if (PAIRS) {
        for (int e = threadIdx.x; e < S * S; e += 256) s_c2[e] = 0;
        for (int e = threadIdx.x; e < 4 * PSP * K1P_LD; e += 256) s_pair[e] = 0;     // (padded state rows stay zero)
    }
    __syncthreads();
    u16* H = nullptr;
    // the lane's block of state pairs (gi <= gj, groups of three states) and its slice of the pair words
    int pgi = -1, pgj = 0, pkq = 0;
    u64 pacc[PAIRS ? 9 : 1], prs[PAIRS ? 3 : 1];
    if (PAIRS) {
        const int role = lane < PROLES * PSLICE ? lane / PSLICE : -1;
        pkq = lane % PSLICE;
        if (role >= 0) {
            int t = role, i = 0;
            while (t >= PG - i) { t -= PG - i; ++i; }
            pgi = i; pgj = i + t;
        }
#pragma unroll
        for (int c = 0; c < 9; ++c) pacc[c] = 0;
#pragma unroll
        for (int c = 0; c < 3; ++c) prs[c] = 0;
    }
    auto pair_counts = [&](const char* stage, int rows) {        // the staged super-tile: [32][S] uint16, `rows` of them real
        u32* sp = s_pair + wave * PSP * K1P_LD;
        const u16* raw = reinterpret_cast<const u16*>(stage);
        __builtin_amdgcn_wave_barrier();
        for (int e = lane; e < S * 16; e += 64) {
            const int i = e >> 4, k = e & 15;
            const u32 lo = 2 * k < rows ? raw[(2 * k) * S + i] : 0u, hi = 2 * k + 1 < rows ? raw[(2 * k + 1) * S + i] : 0u;
            sp[i * K1P_LD + k] = lo | (hi << 16);
        }
        __builtin_amdgcn_wave_barrier();
        if (pgi >= 0) {
            const u32* pi = sp + 3 * pgi * K1P_LD;
            const u32* pj = sp + 3 * pgj * K1P_LD;
            u32 part[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, ps[3] = {0, 0, 0};
            const bool diag = pgi == pgj;
            for (int k = pkq; k < 16; k += PSLICE) {
                const u32 av[3] = {pi[k], pi[K1P_LD + k], pi[2 * K1P_LD + k]};
                const u32 bv[3] = {pj[k], pj[K1P_LD + k], pj[2 * K1P_LD + k]};
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int u = 0; u < 3; ++u)
                        part[3 * t + u] = __builtin_amdgcn_udot2(__builtin_bit_cast(k1_v2u16, av[t]), __builtin_bit_cast(k1_v2u16, bv[u]), part[3 * t + u], false);
                if (diag) {
#pragma unroll
                    for (int t = 0; t < 3; ++t)
                        ps[t] = __builtin_amdgcn_udot2(__builtin_bit_cast(k1_v2u16, av[t]), __builtin_bit_cast(k1_v2u16, 0x00010001u), ps[t], false);
                }
            }
#pragma unroll
            for (int c = 0; c < 9; ++c) pacc[c] += part[c];          // (6 words x 2 bins x 4095^2 < 2^32 per super-tile)
#pragma unroll
            for (int c = 0; c < 3; ++c) prs[c] += ps[c];
        }
        __builtin_amdgcn_wave_barrier();
    };

    // running state counts of this lane's bins, as packed uint16 pairs; flushed to LDS before a half can overflow
    u32 accp[ND];
#pragma unroll
    for (int m = 0; m < ND; ++m) accp[m] = 0;
    const int flush_every = 65535 / nmax > 1 ? 65535 / nmax - 1 : 1;
    int since = 0;
    auto flush = [&]() {
        if (j == 0) {
#pragma unroll
            for (int m = 0; m < ND; ++m) {
                const u32 lo = accp[m] & 0xffffu, hi = accp[m] >> 16;
                if (lo) atomicAdd(&s_cnt[2 * m], (u64)lo);
                if (hi) atomicAdd(&s_cnt[2 * m + 1], (u64)hi);
            }
        }
#pragma unroll
        for (int m = 0; m < ND; ++m) accp[m] = 0;
        since = 0;
    };

    auto enter = [&](u16* Hpart) { H = Hpart; };
    auto epilogue = [&](int half, long row, bool valid, u32 (&cnt)[S]) {
        u32 d[ND];
        pack_reduce<S>(cnt, d);
        if (counts) {
#pragma unroll
            for (int m = 0; m < ND; ++m) accp[m] += valid ? d[m] : 0u;
            if (++since >= flush_every) flush();
        }
        if (H) {
            // stage the bin's uint16 row: d[] already is that layout, one dword per pair of states
            char* srow = &s_stage[wave][(half * 16 + b) * ROWB];
            if constexpr (FULL && (S & 1) == 0) {        // even, full width: whole dwords
#pragma unroll
                for (int k = 0; k < (ND + 3) / 4; ++k) {
                    const u32 v = sel4(d[4 * k], 4 * k + 1 < ND ? d[4 * k + 1] : 0u, 4 * k + 2 < ND ? d[4 * k + 2] : 0u,
                                       4 * k + 3 < ND ? d[4 * k + 3] : 0u, j);
                    if (4 * k + j < ND) *reinterpret_cast<u32*>(srow + 4 * (4 * k + j)) = v;
                }
            } else {
#pragma unroll
                for (int k = 0; k < (S + 3) / 4; ++k) {
                    const u32 v = (j & 2) ? (2 * k + 1 < ND ? d[2 * k + 1] : 0u) : d[2 * k];
                    const u32 c = (j & 1) ? v >> 16 : v & 0xffffu;
                    if (4 * k + j < Sout) *reinterpret_cast<u16*>(srow + 2 * (4 * k + j)) = (u16)c;
                }
            }
        }
    };
    auto finish = [&](long st, long row0, int rows) {
        if (H) store_staged(s_stage[wave], reinterpret_cast<char*>(H) + row0 * ROWB, rows * ROWB, lane);
        if constexpr (PAIRS) pair_counts(s_stage[wave], rows);
    };
    loop(enter, epilogue, finish);

    if (counts) {
        flush();
        __syncthreads();
        if ((int)threadIdx.x < Sout && s_cnt[threadIdx.x]) atomicAdd(&counts[threadIdx.x], s_cnt[threadIdx.x]);
    }
    if constexpr (PAIRS) {
        // C[i,j] = sum h_i h_j (both orders);  C[i,i] = sum h_i^2 - sum h_i.  The block's lanes meet in an LDS copy of C, then one
        // global atomic per cell and block (k_s2_hist_wave's epilogue)
        if (pgi >= 0) {
            const bool diag = pgi == pgj;
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const int i = 3 * pgi + t, jj = 3 * pgj + u;
                    u64 v = pacc[3 * t + u];
                    if (diag && t == u) v -= prs[t];             // two's complement
                    if (i >= S || jj >= S || !v) continue;
                    atomicAdd(&s_c2[i * S + jj], v);
                    if (!diag) atomicAdd(&s_c2[jj * S + i], v);
                }
        }
        __syncthreads();
        for (int e = threadIdx.x; e < S * S; e += 256)
            if (s_c2[e]) atomicAdd(&counts2[e], s_c2[e]);
    }
}

template <int SC, int NG, bool FULL>
__global__ __launch_bounds__(256) void k_bin_hist(const char* __restrict__ X, long R, int N, long ldx, int Sout_,
                                                   u16* __restrict__ H, u64* __restrict__ counts) {
    bin_hist_body<SC, FULL, false>(Sout_, counts, N, nullptr, [&](auto&& enter, auto&& epilogue, auto&& finish) {
        enter(H);
        tile_loop<SC, NG>(X, R, N, ldx, epilogue, finish);
    });
}

// K1 + the S2 pair counts of the same bins (see bin_hist_body): what an S2 job's count pass launches
template <int SC, int NG>
__global__ __launch_bounds__(256) void k_bin_hist_s2(const char* __restrict__ X, long R, int N, long ldx, u16* __restrict__ H,
                                                      u64* __restrict__ counts, u64* __restrict__ counts2) {
    bin_hist_body<SC, true, true>(SC, counts, N, counts2, [&](auto&& enter, auto&& epilogue, auto&& finish) {
        enter(H);
        tile_loop<SC, NG>(X, R, N, ldx, epilogue, finish);
    });
}

// Several matrices in ONE launch (epg_bin_hist_parts): the 2 x 24 chromosome parts of a paired genome are 48 launches of ~80 us
// otherwise, each with its ramp and tail (round 4: the count phase of BASELINE config 5 ran at 0.36 of its bytes).
template <int SC, int NG, bool FULL>
__global__ __launch_bounds__(256) void k_bin_hist_parts(const KhParts pt, int Sout_, u64* __restrict__ counts, int nmax) {
    bin_hist_body<SC, FULL, false>(Sout_, counts, nmax, nullptr, [&](auto&& enter, auto&& epilogue, auto&& finish) {
        tile_loop_parts<SC, NG>(pt, [&](int part) { enter(pt.h[part]); }, epilogue, finish);
    });
}

// Any S <= 127, any N, any alignment, never reads past a row's N bytes: one wave per bin, LDS atomics.
// Used for the matrix's last row(s) when ldx < 16*ceil(N/16) (the fast kernel's last chunk would over-read).  Decodes
// the low five bits of a byte like the fast kernel (see epg_count.h) so that both treat every byte value alike.
__global__ __launch_bounds__(256) void k_bin_hist_safe(const char* __restrict__ X, long row_begin, long row_end, int N,
                                                        long ldx, int S, u16* __restrict__ H, u64* __restrict__ counts, int mask = 31) {
    __shared__ u32 s_h[4][128];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long row = row_begin + (long)blockIdx.x * 4 + wave; row < row_end; row += (long)gridDim.x * 4) {
        for (int s = lane; s < S; s += 64) s_h[wave][s] = 0;
        __builtin_amdgcn_wave_barrier();
        const char* rp = X + row * ldx;
        for (int n = lane; n < N; n += 64) {
            const int v = (unsigned char)rp[n] & mask;       // 31: the fast kernels' five-bit decode; 255: the wide models
            if (v < S) atomicAdd(&s_h[wave][v], 1u);
        }
        __builtin_amdgcn_wave_barrier();
        for (int s = lane; s < S; s += 64) {
            const u32 c = s_h[wave][s];
            if (H) H[row * S + s] = (u16)c;
            if (counts && c) atomicAdd(&counts[s], (u64)c);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// S1 lookup table: T[c, s] = kl(c / N, q[s]) for c = 0..N  (scores.py:343 rowObsS1 + scores.py:550 klScoreND);
// T64 is the float64 value, T32 its float32 rounding (the reference's store, scores.py:317).
// ---------------------------------------------------------------------------------------------------------------
__global__ void k_s1_table(const float* __restrict__ q, int N, int S, double* __restrict__ T64, float* __restrict__ T32) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)(N + 1) * S) return;
    const int c = (int)(e / S), s = (int)(e % S);
    const double p = (double)c / (double)N;
    const double v = kl_term(p, (double)q[s]);
    T64[e] = v;
    T32[e] = (float)v;
}

// ---------------------------------------------------------------------------------------------------------------
// S1 score straight from the state matrix: N bytes read + S*sizeof(OT) written per bin (scores.py:309-317).
// ---------------------------------------------------------------------------------------------------------------
template <int S, int NG, typename OT>
__global__ __launch_bounds__(256) void k_score_s1(const char* __restrict__ X, long R, int N, long ldx,
                                                   const OT* __restrict__ T, OT* __restrict__ out) {
    constexpr int ND = (S + 1) / 2;
    constexpr int ROWB = S * (int)sizeof(OT);
    __shared__ __attribute__((aligned(16))) char s_stage[4][32 * ROWB];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 3, b = lane >> 2;
    auto epilogue = [&](int half, long row, bool valid, u32 (&cnt)[S]) {
        u32 d[ND];
        pack_reduce<S>(cnt, d);
        OT* srow = reinterpret_cast<OT*>(&s_stage[wave][(half * 16 + b) * ROWB]);
#pragma unroll
        for (int k = 0; k < (S + 3) / 4; ++k) {
            const u32 v = (j & 2) ? (2 * k + 1 < ND ? d[2 * k + 1] : 0u) : d[2 * k];
            const u32 c = (j & 1) ? v >> 16 : v & 0xffffu;
            const int s = 4 * k + j;
            if (s < S) srow[s] = (c && valid) ? T[(long)c * S + s] : (OT)0;
        }
    };
    auto finish = [&](long st, long row0, int rows) {
        store_staged_nt(s_stage[wave], reinterpret_cast<char*>(out) + row0 * ROWB, rows * ROWB, lane);
    };
    tile_loop<S, NG>(X, R, N, ldx, epilogue, finish);
}

// S1 score from cached histograms: 2*S bytes read + S*sizeof(OT) written per bin.  A lane handles 4 consecutive
// elements (8 bytes of H in, 16/32 bytes out); consecutive lanes take consecutive quads so every load and store
// instruction covers whole contiguous lines; 4 quads in flight per lane.
// LDS_T: the table is first copied into LDS (nent entries) and gathered from there.  From memory a 64-lane gather of four-byte
// entries walks up to 64 lines through the L1 (one line per clock), four gathers per 16 bytes of output: the kernel ran at
// the texture-address rate, 3.9 TB/s of its 108 B/bin; an LDS gather costs a few cycles.
template <typename OT, bool LDS_T>
__global__ __launch_bounds__(LDS_T ? 1024 : 256) void k_score_s1_from_hist(const u16* __restrict__ H, long total, int S,
                                                                             const OT* __restrict__ Tg, int nent, OT* __restrict__ out,
                                                                             u64* __restrict__ zero_counts, int rev) {
    extern __shared__ __attribute__((aligned(16))) char smem_t[];
    // the job's state counts are spent once the table exists (this launch is ordered after k_s1_combine): zero them for the
    // next job's accumulation here instead of in a launch of its own
    if (zero_counts && blockIdx.x == 0 && (int)threadIdx.x < S) zero_counts[threadIdx.x] = 0;
    const OT* T = Tg;
    if (LDS_T) {
        OT* Ts = reinterpret_cast<OT*>(smem_t);
        for (int e = threadIdx.x; e < nent; e += blockDim.x) Ts[e] = Tg[e];
        __syncthreads();
        T = Ts;
    }
    // a count is looked up when it is in 1 .. N (N = nent / S - 1); 0 scores 0 (scores.py:550), and so does anything larger: the
    // rows between the parts of a batch's flat histogram buffer (engine.hist_rows_flat) are never written by the count pass
    const u32 cmax = (u32)(nent / S - 1);
    const long nquads = total >> 2;
    const long stride = (long)gridDim.x * blockDim.x;
    long qd = (long)blockIdx.x * blockDim.x + threadIdx.x;
    // rev: the grid walks the quads from the LAST to the first.  The histograms were written front to back by the count pass
    // just before, so the ones written last are the ones still in the memory-side cache (DESIGN.md, K1 / score pass).
    const long qtop = nquads - 1, qsgn = rev ? -1 : 1, qoff = rev ? qtop : 0;     // quad index = qoff + qsgn * qd
    for (; qd + 3 * stride < nquads; qd += 4 * stride) {
        uint2 h[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) h[u] = *reinterpret_cast<const uint2*>(H + 4 * (qoff + qsgn * (qd + u * stride)));
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long e0 = 4 * (qoff + qsgn * (qd + u * stride));
            int s = (int)(e0 % S);
            const u32 c[4] = {h[u].x & 0xffffu, h[u].x >> 16, h[u].y & 0xffffu, h[u].y >> 16};
            OT v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                v[k] = c[k] - 1u < cmax ? T[(long)c[k] * S + s] : (OT)0;
                s = s + 1 == S ? 0 : s + 1;
            }
            // non-temporal: nobody on the device reads the scores again (store_staged_nt, epg_common.h)
            if constexpr (sizeof(OT) == 4) {
                typedef float v4f_nt __attribute__((ext_vector_type(4)));
                const v4f_nt vv = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
                __builtin_nontemporal_store(vv, reinterpret_cast<v4f_nt*>(out + e0));
            } else {
                typedef double v2d_nt __attribute__((ext_vector_type(2)));
                const v2d_nt lo = {(double)v[0], (double)v[1]}, hi = {(double)v[2], (double)v[3]};
                __builtin_nontemporal_store(lo, reinterpret_cast<v2d_nt*>(out + e0));
                __builtin_nontemporal_store(hi, reinterpret_cast<v2d_nt*>(out + e0 + 2));
            }
        }
    }
    for (; qd < nquads; qd += stride) {
        const long e0 = 4 * (qoff + qsgn * qd);
        int s = (int)(e0 % S);
        for (int k = 0; k < 4; ++k) {
            const u32 c = H[e0 + k];
            out[e0 + k] = c - 1u < cmax ? T[(long)c * S + s] : (OT)0;
            s = s + 1 == S ? 0 : s + 1;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (total & 3)) {
        const long e = (nquads << 2) + threadIdx.x;
        const u32 c = H[e];
        out[e] = c - 1u < cmax ? T[(long)c * S + (int)(e % S)] : (OT)0;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// normalise: q = float32( double(C) / double(sum C) )   (expectedCombination.py:42).  Two tiny kernels: integer sum
// (exact, order-independent), then the divide.
// ---------------------------------------------------------------------------------------------------------------
template <typename IT>
__global__ __launch_bounds__(256) void k_sum_int(const IT* __restrict__ C, long n, long long* __restrict__ total) {
    long long acc = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) acc += (long long)C[i];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    __shared__ long long s_part[4];
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const long long t = s_part[0] + s_part[1] + s_part[2] + s_part[3];
        if (t) atomicAdd(reinterpret_cast<u64*>(total), (u64)t);
    }
}

template <typename IT>
__global__ __launch_bounds__(256) void k_normalise(const IT* __restrict__ C, long n, const long long* __restrict__ total,
                                                    float* __restrict__ q) {
    const double tot = (double)(*total);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        q[i] = (float)((double)C[i] / tot);
}

// Short arrays -- the S1 and S2 count arrays (18 and 324 entries) -- in ONE block and ONE launch: the sum stays in the block, so
// the memset and the two kernels above (three launches, ~15 us of a 0.45 ms S2 job on an eighth of the genome) become one.
template <typename IT>
__global__ __launch_bounds__(256) void k_normalise_small(const IT* __restrict__ C, int n, float* __restrict__ q) {
    __shared__ long long s_part[4];
    long long acc = 0;
    for (int i = threadIdx.x; i < n; i += 256) acc += (long long)C[i];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    const double tot = (double)(s_part[0] + s_part[1] + s_part[2] + s_part[3]);
    for (int i = threadIdx.x; i < n; i += 256) q[i] = (float)((double)C[i] / tot);
}

// ---------------------------------------------------------------------------------------------------------------
// host-side dispatch
// ---------------------------------------------------------------------------------------------------------------
// persistent grid size in blocks per CU.  Round 3: two blocks (2 waves per SIMD) measure 3-4 % faster than four on 1.9-7.5 M-bin
// shards and 1 % on 15 M bins (profiles/r03i_*); epg_test_force(4, n) sets another for A/B runs (tools/kbench.py)
static int blocks_per_cu() { return g_force[FORCE_K1_BLOCKS_PER_CU] > 0 ? g_force[FORCE_K1_BLOCKS_PER_CU] : 2; }
static int grid_for_tiles(long R) {
    const long nsuper = (R + 31) >> 5;
    long blocks = (nsuper + 3) / 4;
    const long cap = (long)num_cus() * blocks_per_cu();
    if (blocks > cap) {
        // persistent grid: a wave walks super-tiles st, st + waves, ...; with the full grid the waves of a 1.9 M-bin shard (an
        // eighth of the genome) get 14 or 15 of them -- the launch ends with 5 % of its time spent by the waves that got 15.
        // Take the fewest iterations the full grid allows, then the fewest blocks that still cover the tiles in that many.
        const long iters = (nsuper + cap * 4 - 1) / (cap * 4);
        blocks = (nsuper + iters * 4 - 1) / (iters * 4);
    }
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

template <int S, int NG>
static void launch_bin_hist(const char* X, long R, int N, long ldx, int Sout, u16* H, u64* counts, hipStream_t st) {
    // odd S stores uint16 by uint16 either way; with the row width as a run-time value that path measured 2.30 ms
    // against 2.64 ms for the compile-time one (15 M x 833, S = 15), so only even S takes the compile-time width
    constexpr bool FULL = (S & 1) == 0;
    hipLaunchKernelGGL((k_bin_hist<S, NG, FULL>), dim3(grid_for_tiles(R)), dim3(256), 0, st, X, R, N, ldx, Sout, H, counts);
}

// a state model of another size: the counting core of the next instantiated size, any-N load loop, only Sout columns stored
template <int SC>
static void launch_bin_hist_any(const char* X, long R, int N, long ldx, int Sout, u16* H, u64* counts, hipStream_t st) {
    hipLaunchKernelGGL((k_bin_hist<SC, 0, false>), dim3(grid_for_tiles(R)), dim3(256), 0, st, X, R, N, ldx, Sout, H, counts);
}

template <int S>
static void dispatch_bin_hist_ng(const char* X, long R, int N, long ldx, int Sout, u16* H, u64* counts, hipStream_t st) {
    const int ng = (N + 127) / 128;
    switch (ng) {
        case 1: launch_bin_hist<S, 1>(X, R, N, ldx, Sout, H, counts, st); break;
        case 2: launch_bin_hist<S, 2>(X, R, N, ldx, Sout, H, counts, st); break;
        case 3: launch_bin_hist<S, 3>(X, R, N, ldx, Sout, H, counts, st); break;
        case 4: launch_bin_hist<S, 4>(X, R, N, ldx, Sout, H, counts, st); break;
        case 5: launch_bin_hist<S, 5>(X, R, N, ldx, Sout, H, counts, st); break;
        case 6: launch_bin_hist<S, 6>(X, R, N, ldx, Sout, H, counts, st); break;
        case 7: launch_bin_hist<S, 7>(X, R, N, ldx, Sout, H, counts, st); break;
        case 8: launch_bin_hist<S, 8>(X, R, N, ldx, Sout, H, counts, st); break;
        default: launch_bin_hist<S, 0>(X, R, N, ldx, Sout, H, counts, st); break;
    }
}

// rows the fast kernel may touch: all of them unless its 16-byte last chunk could run past the allocation
static long fast_rows(long R, int N, long ldx) {
    const long chunks = (N + 15) / 16;
    const long over = 16 * chunks - ldx;           // bytes a row's last chunk reaches past the row pitch
    if (over <= 0) return R;
    const long unsafe = (over + ldx - 1) / ldx;    // trailing rows whose last chunk would end past X + R*ldx (N < 16: several)
    return R > unsafe ? R - unsafe : 0;
}

int bin_hist_impl(const int8_t* X8, int64_t R, int32_t N, int64_t ldx, int32_t S, uint16_t* H, int64_t* counts,
                  hipStream_t st) {
    if (R < 0 || N < 1 || ldx < N || S < 1) return fail(EPG_ERR_INVALID_ARG, "bin_hist: bad shape R=%lld N=%d ldx=%lld S=%d", (long long)R, N, (long long)ldx, S);
    if (S > 127) return fail(EPG_ERR_UNSUPPORTED, "bin_hist: S=%d > 127 (states are int8)", S);
    if (N > 65535) return fail(EPG_ERR_UNSUPPORTED, "bin_hist: N=%d > 65535 (uint16 per-bin counts)", N);
    if (R == 0) return EPG_OK;
    if (!X8) return fail(EPG_ERR_INVALID_ARG, "bin_hist: X is NULL");
    if (H && (reinterpret_cast<uintptr_t>(H) & 15)) return fail(EPG_ERR_INVALID_ARG, "bin_hist: H must be 16-byte aligned");
    const char* X = reinterpret_cast<const char*>(X8);
    u64* cnt = reinterpret_cast<u64*>(counts);
    if (S > 31) {                                  // the wide models (epg_wide.hip): one wave per bin, the whole byte decoded
        hipLaunchKernelGGL(k_bin_hist_safe, dim3((unsigned)((R + 3) / 4 < num_cus() * 32L ? (R + 3) / 4 : num_cus() * 32L)), dim3(256), 0, st, X, 0L,
                           (long)R, N, ldx, S, H, cnt, 255);
        EPG_LAUNCH_CHECK("k_bin_hist_safe");
        return EPG_OK;
    }
    const long Rf = fast_rows(R, N, ldx);
    if (Rf > 0) {
        // the reference's models have their own instantiation; any other size runs on the next larger counting core
        if (S == 15) dispatch_bin_hist_ng<15>(X, Rf, N, ldx, S, H, cnt, st);
        else if (S == 18) dispatch_bin_hist_ng<18>(X, Rf, N, ldx, S, H, cnt, st);
        else if (S == 25) dispatch_bin_hist_ng<25>(X, Rf, N, ldx, S, H, cnt, st);
        else if (S < 15) launch_bin_hist_any<15>(X, Rf, N, ldx, S, H, cnt, st);
        else if (S < 18) launch_bin_hist_any<18>(X, Rf, N, ldx, S, H, cnt, st);
        else if (S < 25) launch_bin_hist_any<25>(X, Rf, N, ldx, S, H, cnt, st);
        else launch_bin_hist_any<31>(X, Rf, N, ldx, S, H, cnt, st);
        EPG_LAUNCH_CHECK("k_bin_hist");
    }
    if (Rf < R) {
        hipLaunchKernelGGL(k_bin_hist_safe, dim3((unsigned)((R - Rf + 3) / 4 < 1024 ? (R - Rf + 3) / 4 : 1024)), dim3(256), 0, st, X, Rf, (long)R, N, ldx, S, H, cnt);
        EPG_LAUNCH_CHECK("k_bin_hist_safe");
    }
    return EPG_OK;
}

// ---- K1 with the S2 pair counts folded in (epg_bin_hist_s2)
int hist_s2_from_binhist_impl(const uint16_t*, const uint16_t*, int64_t, int32_t, int64_t*, hipStream_t);

template <int S>
static void dispatch_bin_hist_s2(const char* X, long R, int N, long ldx, u16* H, u64* counts, u64* counts2, hipStream_t st) {
#define EPG_K1S2(NGV) hipLaunchKernelGGL((k_bin_hist_s2<S, NGV>), dim3(grid_for_tiles(R)), dim3(256), 0, st, X, R, N, ldx, H, counts, counts2)
    switch ((N + 127) / 128) {
        case 1: EPG_K1S2(1); break;
        case 2: EPG_K1S2(2); break;
        case 3: EPG_K1S2(3); break;
        case 4: EPG_K1S2(4); break;
        case 5: EPG_K1S2(5); break;
        case 6: EPG_K1S2(6); break;
        case 7: EPG_K1S2(7); break;
        default: EPG_K1S2(8); break;
    }
#undef EPG_K1S2
}

// H = per-bin histograms of X, counts2[S * S] += the S2 pair counts of its bins (and counts[S] += the state counts when given): ONE
// launch for the reference's 15-, 18- and 25-state models on rows of up to 1024 columns; anything else: the count pass, then
// the pair-count pass over H (the same integers).
int bin_hist_s2_impl(const int8_t* X8, int64_t R, int32_t N, int64_t ldx, int32_t S, uint16_t* H, int64_t* counts, int64_t* counts2,
                     hipStream_t st) {
    if (!H || !counts2) return fail(EPG_ERR_INVALID_ARG, "bin_hist_s2: H and counts2 are required");
    const bool fused = (S == 15 || S == 18 || S == 25) && R > 0 && N >= 1 && N <= 1024 && ldx >= N && X8 &&
                       !(reinterpret_cast<uintptr_t>(H) & 15) && fast_rows(R, N, ldx) == R;
    if (!fused) {
        int rc = bin_hist_impl(X8, R, N, ldx, S, H, counts, st);
        if (rc) return rc;
        return hist_s2_from_binhist_impl(H, nullptr, R, S, counts2, st);
    }
    const char* X = reinterpret_cast<const char*>(X8);
    u64* c1 = reinterpret_cast<u64*>(counts);
    u64* c2 = reinterpret_cast<u64*>(counts2);
    if (S == 18) dispatch_bin_hist_s2<18>(X, R, N, ldx, H, c1, c2, st);
    else if (S == 15) dispatch_bin_hist_s2<15>(X, R, N, ldx, H, c1, c2, st);
    else dispatch_bin_hist_s2<25>(X, R, N, ldx, H, c1, c2, st);
    EPG_LAUNCH_CHECK("k_bin_hist_s2");
    return EPG_OK;
}

// ---- several matrices, one launch
template <int S, int NG>
static void launch_bin_hist_parts(const KhParts& pt, int Sout, u64* counts, int nmax, hipStream_t st) {
    constexpr bool FULL = NG > 0 && (S & 1) == 0;          // (the any-width instantiation stores column by column, like launch_bin_hist_any)
    const long nsuper = pt.st0[pt.n];
    hipLaunchKernelGGL((k_bin_hist_parts<S, NG, FULL>), dim3(grid_for_tiles(nsuper * 32)), dim3(256), 0, st, pt, Sout, counts, nmax);
}

template <int S>
static void dispatch_bin_hist_parts(int ng, const KhParts& pt, int Sout, u64* counts, int nmax, hipStream_t st) {
    switch (ng) {
        case 1: launch_bin_hist_parts<S, 1>(pt, Sout, counts, nmax, st); break;
        case 2: launch_bin_hist_parts<S, 2>(pt, Sout, counts, nmax, st); break;
        case 3: launch_bin_hist_parts<S, 3>(pt, Sout, counts, nmax, st); break;
        case 4: launch_bin_hist_parts<S, 4>(pt, Sout, counts, nmax, st); break;
        case 5: launch_bin_hist_parts<S, 5>(pt, Sout, counts, nmax, st); break;
        case 6: launch_bin_hist_parts<S, 6>(pt, Sout, counts, nmax, st); break;
        case 7: launch_bin_hist_parts<S, 7>(pt, Sout, counts, nmax, st); break;
        case 8: launch_bin_hist_parts<S, 8>(pt, Sout, counts, nmax, st); break;
        default: launch_bin_hist_parts<S, 0>(pt, Sout, counts, nmax, st); break;
    }
}

// epg_bin_hist over `nparts` matrices in as few launches as their widths allow: parts whose widths share a load schedule
// (the same number of 128-byte groups per row; more than eight: the any-width loop) go into one launch, KH_MAXP at a time.  The
// same integers as nparts calls of bin_hist_impl (tests/test_hip_parity.py); restates the per-file loop of run.py:236-257 +
// expected.py:111-113 for resident files.
int bin_hist_parts_impl(int32_t nparts, const int8_t* const* X, const int64_t* R, const int32_t* N, const int64_t* ldx, int32_t S,
                        uint16_t* const* H, int64_t* counts, hipStream_t st) {
    if (nparts < 0 || (nparts && (!X || !R || !N || !ldx))) return fail(EPG_ERR_INVALID_ARG, "bin_hist_parts: NULL argument array");
    if (S < 1 || S > 127) return fail(EPG_ERR_UNSUPPORTED, "bin_hist_parts: S=%d outside 1..127", S);
    for (int p = 0; p < nparts; ++p) {
        if (R[p] < 0 || (R[p] && (N[p] < 1 || ldx[p] < N[p]))) return fail(EPG_ERR_INVALID_ARG, "bin_hist_parts: bad shape of part %d", p);
        if (R[p] && N[p] > 65535) return fail(EPG_ERR_UNSUPPORTED, "bin_hist_parts: N=%d > 65535 (uint16 per-bin counts)", N[p]);
        if (R[p] && !X[p]) return fail(EPG_ERR_INVALID_ARG, "bin_hist_parts: X of part %d is NULL", p);
        if (H && H[p] && (reinterpret_cast<uintptr_t>(H[p]) & 15)) return fail(EPG_ERR_INVALID_ARG, "bin_hist_parts: H of part %d must be 16-byte aligned", p);
    }
    if (S > 31) {                                  // the wide models: part by part
        for (int p = 0; p < nparts; ++p) {
            const int rc = bin_hist_impl(X[p], R[p], N[p], ldx[p], S, H ? H[p] : nullptr, counts, st);
            if (rc) return rc;
        }
        return EPG_OK;
    }
    u64* cnt = reinterpret_cast<u64*>(counts);
    // schedule class of a part: groups per row (1..8), 0 = any width; classes are launched one after the other
    auto cls = [](int n) { const int ng = (n + 127) / 128; return ng <= 8 ? ng : 0; };
    bool seen[9] = {false, false, false, false, false, false, false, false, false};
    for (int p = 0; p < nparts; ++p)
        if (R[p]) seen[cls(N[p])] = true;
    const bool generic = !(S == 15 || S == 18 || S == 25);      // another model size: the any-width loop of the next larger core
    for (int c = 0; c <= 8; ++c) {
        if (!seen[c]) continue;
        for (int p0 = 0; p0 < nparts;) {
            KhParts pt;
            memset(&pt, 0, sizeof(pt));
            long supers = 0;
            int nmax = 1, p = p0;
            for (; p < nparts && pt.n < KH_MAXP; ++p) {
                if (!R[p] || cls(N[p]) != c) continue;
                const long Rf = fast_rows(R[p], N[p], ldx[p]);
                if (Rf < R[p]) {                   // trailing rows the 16-byte loads could run past: byte-granular kernel
                    hipLaunchKernelGGL(k_bin_hist_safe, dim3((unsigned)((R[p] - Rf + 3) / 4 < 1024 ? (R[p] - Rf + 3) / 4 : 1024)), dim3(256), 0, st,
                                       reinterpret_cast<const char*>(X[p]), Rf, (long)R[p], N[p], (long)ldx[p], S, H ? H[p] : nullptr, cnt);
                    EPG_LAUNCH_CHECK("k_bin_hist_safe");
                }
                if (Rf == 0) continue;
                const int k = pt.n++;
                pt.x[k] = reinterpret_cast<const char*>(X[p]);
                pt.h[k] = H ? H[p] : nullptr;
                pt.rows[k] = Rf;
                pt.ldx[k] = ldx[p];
                pt.n_cols[k] = N[p];
                pt.st0[k] = supers;
                supers += (Rf + 31) / 32;
                if (N[p] > nmax) nmax = N[p];
            }
            pt.st0[pt.n] = supers;
            p0 = p;
            if (pt.n == 0) break;
            const int ng = generic ? 0 : c;
            if (S == 15) dispatch_bin_hist_parts<15>(ng, pt, S, cnt, nmax, st);
            else if (S == 18) dispatch_bin_hist_parts<18>(ng, pt, S, cnt, nmax, st);
            else if (S == 25) dispatch_bin_hist_parts<25>(ng, pt, S, cnt, nmax, st);
            else if (S < 15) launch_bin_hist_parts<15, 0>(pt, S, cnt, nmax, st);
            else if (S < 18) launch_bin_hist_parts<18, 0>(pt, S, cnt, nmax, st);
            else if (S < 25) launch_bin_hist_parts<25, 0>(pt, S, cnt, nmax, st);
            else launch_bin_hist_parts<31, 0>(pt, S, cnt, nmax, st);
            EPG_LAUNCH_CHECK("k_bin_hist_parts");
        }
    }
    return EPG_OK;
}

template <int S, int NG, typename OT>
static void launch_score_s1(const char* X, long R, int N, long ldx, const OT* T, OT* out, hipStream_t st) {
    hipLaunchKernelGGL((k_score_s1<S, NG, OT>), dim3(grid_for_tiles(R)), dim3(256), 0, st, X, R, N, ldx, T, out);
}

template <typename OT>
static bool dispatch_score_s1(const char* X, long R, int N, long ldx, int S, const OT* T, OT* out, hipStream_t st) {
    if (S != 18) return false;
    const int ng = (N + 127) / 128;
    switch (ng) {
        case 1: launch_score_s1<18, 1, OT>(X, R, N, ldx, T, out, st); break;
        case 2: launch_score_s1<18, 2, OT>(X, R, N, ldx, T, out, st); break;
        case 3: launch_score_s1<18, 3, OT>(X, R, N, ldx, T, out, st); break;
        case 4: launch_score_s1<18, 4, OT>(X, R, N, ldx, T, out, st); break;
        case 5: launch_score_s1<18, 5, OT>(X, R, N, ldx, T, out, st); break;
        case 6: launch_score_s1<18, 6, OT>(X, R, N, ldx, T, out, st); break;
        case 7: launch_score_s1<18, 7, OT>(X, R, N, ldx, T, out, st); break;
        case 8: launch_score_s1<18, 8, OT>(X, R, N, ldx, T, out, st); break;
        default: launch_score_s1<18, 0, OT>(X, R, N, ldx, T, out, st); break;
    }
    return true;
}

static int64_t s1_table_bytes(int N, int S) { return align_up((int64_t)(N + 1) * S * 8, 256) + align_up((int64_t)(N + 1) * S * 4, 256); }

int64_t s1_ws_bytes(int64_t R, int N, int S) {
    // table + room for a cached histogram (used when the fused kernel does not cover this S)
    return s1_table_bytes(N, S) + 256 + align_up(R * S * 2, 256);
}

static int build_s1_table(const float* q, int N, int S, void* ws, double** T64, float** T32, hipStream_t st) {
    *T64 = reinterpret_cast<double*>(ws);
    *T32 = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + align_up((int64_t)(N + 1) * S * 8, 256));
    const long n = (long)(N + 1) * S;
    hipLaunchKernelGGL(k_s1_table, dim3((int)((n + 255) / 256)), dim3(256), 0, st, q, N, S, *T64, *T32);
    EPG_LAUNCH_CHECK("k_s1_table");
    return EPG_OK;
}

static int check_score_from_hist_args(const uint16_t* H, const double* out64, const float* out32) {
    if ((reinterpret_cast<uintptr_t>(H) & 7) != 0) return fail(EPG_ERR_INVALID_ARG, "score_s1_from_binhist: H must be 8-byte aligned");
    if ((out32 && (reinterpret_cast<uintptr_t>(out32) & 15)) || (out64 && (reinterpret_cast<uintptr_t>(out64) & 15)))
        return fail(EPG_ERR_INVALID_ARG, "score_s1_from_binhist: outputs must be 16-byte aligned");
    return EPG_OK;
}

template <typename OT>
static int launch_score_s1_from_hist_t(const uint16_t* H, long total, int32_t S, const OT* T, int nent, OT* out, u64* zero_counts, hipStream_t st) {
    const size_t tbytes = (size_t)nent * sizeof(OT);
    const int rev = 1;      // the grid walks H from its last row to the first: what the count pass wrote last is still cached
    if (tbytes <= 150 * 1024) {             // the table in LDS: blocks of 16 waves, as many per CU as tables fit (<= 2)
        const int per_cu = tbytes <= 75 * 1024 ? 2 : 1;
        long nb = (total / 4 + 4095) / 4096;
        if (nb > (long)num_cus() * per_cu) nb = (long)num_cus() * per_cu;
        if (nb < 1) nb = 1;
        auto kern = k_score_s1_from_hist<OT, true>;
        static DynLds lds_attr;                        // one per OT instantiation
        EPG_HIP(ensure_dyn_lds(lds_attr, reinterpret_cast<const void*>(kern), 150 * 1024));
        hipLaunchKernelGGL(kern, dim3((int)nb), dim3(1024), tbytes, st, H, total, S, T, nent, out, zero_counts, rev);
    } else {
        long nb = (total / 4 + 1023) / 1024;
        if (nb > num_cus() * 8L) nb = num_cus() * 8L;  // 8 blocks/CU measured best (16: +5 %); nt stores: no gain
        if (nb < 1) nb = 1;
        hipLaunchKernelGGL((k_score_s1_from_hist<OT, false>), dim3((int)nb), dim3(256), 0, st, H, total, S, T, nent, out, zero_counts, rev);
    }
    EPG_LAUNCH_CHECK("k_score_s1_from_hist");
    return EPG_OK;
}

static int launch_score_s1_from_hist(const uint16_t* H, int64_t R, int32_t N, int32_t S, const double* T64, const float* T32,
                                     double* out64, float* out32, hipStream_t st, u64* zero_counts = nullptr) {
    const long total = (long)R * S;
    const int nent = (N + 1) * S;
    int rc = EPG_OK;
    if (out32) rc = launch_score_s1_from_hist_t<float>(H, total, S, T32, nent, out32, zero_counts, st);
    if (!rc && out64) rc = launch_score_s1_from_hist_t<double>(H, total, S, T64, nent, out64, out32 ? nullptr : zero_counts, st);
    return rc;
}

int score_s1_from_hist_impl(const uint16_t* H, int64_t R, int32_t N, int32_t S, const float* q, double* out64,
                            float* out32, void* ws, int64_t ws_bytes, hipStream_t st) {
    if (R < 0 || N < 1 || S < 1) return fail(EPG_ERR_INVALID_ARG, "score_s1: bad shape");
    if (R == 0) return EPG_OK;
    if (!H || !q || !ws) return fail(EPG_ERR_INVALID_ARG, "score_s1_from_binhist: NULL argument");
    if (ws_bytes < s1_table_bytes(N, S)) return fail(EPG_ERR_WORKSPACE, "score_s1: workspace %lld < %lld bytes", (long long)ws_bytes, (long long)s1_table_bytes(N, S));
    int rc = check_score_from_hist_args(H, out64, out32);
    if (rc) return rc;
    double* T64; float* T32;
    rc = build_s1_table(q, N, S, ws, &T64, &T32, st);
    if (rc) return rc;
    return launch_score_s1_from_hist(H, R, N, S, T64, T32, out64, out32, st);
}

// The same score pass with the lookup table supplied by the caller (device pointers, [N + 1, S] row-major; either may be NULL
// together with its output): the command line builds T[c, s] = kl(c / N, q[s]) on the host with numpy's log2 -- the
// reference's own arithmetic (scores.py:343,550) -- so that scores_*.txt.gz equals the reference's byte for byte; the device
// table above differs from it in the last bit of a few float64 logarithms, which moves ~0.1 % of the float32 stores.
int score_s1_from_hist_table_impl(const uint16_t* H, int64_t R, int32_t N, int32_t S, const double* T64, const float* T32,
                                  double* out64, float* out32, hipStream_t st) {
    if (R < 0 || N < 1 || S < 1) return fail(EPG_ERR_INVALID_ARG, "score_s1: bad shape");
    if (R == 0) return EPG_OK;
    if (!H || (out64 && !T64) || (out32 && !T32)) return fail(EPG_ERR_INVALID_ARG, "score_s1_from_binhist_table: NULL argument");
    if ((T64 && (reinterpret_cast<uintptr_t>(T64) & 7)) || (T32 && (reinterpret_cast<uintptr_t>(T32) & 3)))
        return fail(EPG_ERR_INVALID_ARG, "score_s1_from_binhist_table: misaligned table");
    int rc = check_score_from_hist_args(H, out64, out32);
    if (rc) return rc;
    return launch_score_s1_from_hist(H, R, N, S, T64, T32, out64, out32, st);
}

// ---------------------------------------------------------------------------------------------------------------
// STEP 2 and the S1 table of STEP 3 in ONE launch (a whole S1 job on a 1.9 M-bin shard is ~0.35 ms of kernel time: a memset,
// two normalise kernels and the table kernel, each with its ~1.7 us boundary and its host call, were a measurable part of
// it): q = float32(double(C) / double(sum C)) (expectedCombination.py:42), T[c, s] = kl(c / N, q[s]) (scores.py:343,550).
// Every block works q out for itself from the S counts and then fills 256 entries of the table: as ONE block of 1024 threads
// (each with 15 float64 log2 in a row) this launch took 16.5 us, a tenth of which is left.  The counts are zeroed for the next
// job by the score kernel that follows (no block of this one knows when the others have read them).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_s1_combine(const u64* __restrict__ counts, int N, int S, float* __restrict__ q,
                                                     double* __restrict__ T64, float* __restrict__ T32) {
    __shared__ long long s_c[128];
    __shared__ double s_q[128];
    const int t = threadIdx.x;
    if (t < S) s_c[t] = (long long)counts[t];
    __syncthreads();
    if (t < S) {
        long long tot = 0;
        for (int s = 0; s < S; ++s) tot += s_c[s];
        const float qf = (float)((double)s_c[t] / (double)tot);
        if (blockIdx.x == 0) q[t] = qf;
        s_q[t] = (double)qf;
    }
    __syncthreads();
    const int e = blockIdx.x * 256 + t;
    if (e < (N + 1) * S) {
        const int c = e / S, s = e - c * S;
        const double v = kl_term((double)c / (double)N, s_q[s]);
        T64[e] = v;
        T32[e] = (float)v;
    }
}

template <typename IT>
static int normalise_impl(const IT* C, int64_t n, float* q, void* ws, int64_t ws_bytes, hipStream_t st);

int combine_score_s1_impl(int64_t* counts, int32_t rezero, const uint16_t* H, int64_t R, int32_t N, int32_t S, float* q,
                          double* out64, float* out32, void* ws, int64_t ws_bytes, hipStream_t st) {
    if (R < 0 || N < 1 || S < 1) return fail(EPG_ERR_INVALID_ARG, "combine_score_s1: bad shape R=%lld N=%d S=%d", (long long)R, N, S);
    if (S > 127) return fail(EPG_ERR_UNSUPPORTED, "combine_score_s1: S=%d > 127 (states are int8)", S);
    if (!counts || !q || !ws || (R > 0 && !H)) return fail(EPG_ERR_INVALID_ARG, "combine_score_s1: NULL argument");
    if (ws_bytes < s1_table_bytes(N, S)) return fail(EPG_ERR_WORKSPACE, "combine_score_s1: workspace %lld < %lld bytes", (long long)ws_bytes, (long long)s1_table_bytes(N, S));
    int rc = check_score_from_hist_args(H, out64, out32);
    if (rc) return rc;
    double* T64 = reinterpret_cast<double*>(ws);
    float* T32 = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + align_up((int64_t)(N + 1) * S * 8, 256));
    const long nent = (long)(N + 1) * S;
    hipLaunchKernelGGL(k_s1_combine, dim3((unsigned)((nent + 255) / 256)), dim3(256), 0, st, reinterpret_cast<const u64*>(counts), N, S, q, T64, T32);
    EPG_LAUNCH_CHECK("k_s1_combine");
    if (R == 0 || (!out32 && !out64)) {
        if (rezero) EPG_HIP(hipMemsetAsync(counts, 0, (size_t)S * 8, st));
        return EPG_OK;
    }
    return launch_score_s1_from_hist(H, R, N, S, T64, T32, out64, out32, st, rezero ? reinterpret_cast<u64*>(counts) : nullptr);
}

int score_s1_impl(const int8_t* X8, int64_t R, int32_t N, int64_t ldx, int32_t S, const float* q, double* out64,
                  float* out32, void* ws, int64_t ws_bytes, hipStream_t st) {
    if (R < 0 || N < 1 || ldx < N || S < 1) return fail(EPG_ERR_INVALID_ARG, "score_s1: bad shape");
    if (S > 127) return fail(EPG_ERR_UNSUPPORTED, "score_s1: S=%d > 127 (states are int8)", S);
    if (R == 0) return EPG_OK;
    if (!X8 || !q || !ws) return fail(EPG_ERR_INVALID_ARG, "score_s1: NULL argument");
    if ((out32 && (reinterpret_cast<uintptr_t>(out32) & 15)) || (out64 && (reinterpret_cast<uintptr_t>(out64) & 15)))
        return fail(EPG_ERR_INVALID_ARG, "score_s1: outputs must be 16-byte aligned");
    const long Rf = fast_rows(R, N, ldx);
    const bool fused = S == 18 && Rf == R;
    if (!fused) {
        // histogram into the workspace, then score from it
        if (ws_bytes < s1_ws_bytes(R, N, S)) return fail(EPG_ERR_WORKSPACE, "score_s1: workspace %lld < %lld bytes", (long long)ws_bytes, (long long)s1_ws_bytes(R, N, S));
        u16* H = reinterpret_cast<u16*>(reinterpret_cast<char*>(ws) + s1_table_bytes(N, S));
        int rc = bin_hist_impl(X8, R, N, ldx, S, H, nullptr, st);
        if (rc) return rc;
        return score_s1_from_hist_impl(H, R, N, S, q, out64, out32, ws, s1_table_bytes(N, S), st);
    }
    if (ws_bytes < s1_table_bytes(N, S)) return fail(EPG_ERR_WORKSPACE, "score_s1: workspace %lld < %lld bytes", (long long)ws_bytes, (long long)s1_table_bytes(N, S));
    double* T64; float* T32;
    int rc = build_s1_table(q, N, S, ws, &T64, &T32, st);
    if (rc) return rc;
    const char* X = reinterpret_cast<const char*>(X8);
    if (out32) dispatch_score_s1<float>(X, R, N, ldx, S, T32, out32, st);
    if (out64) dispatch_score_s1<double>(X, R, N, ldx, S, T64, out64, st);
    EPG_LAUNCH_CHECK("k_score_s1");
    return EPG_OK;
}

template <typename IT>
static int normalise_impl(const IT* C, int64_t n, float* q, void* ws, int64_t ws_bytes, hipStream_t st) {
    if (n < 1 || !C || !q || !ws) return fail(EPG_ERR_INVALID_ARG, "normalise: bad argument");
    if (ws_bytes < 8) return fail(EPG_ERR_WORKSPACE, "normalise: workspace needs 8 bytes");
    if (n <= 4096) {
        hipLaunchKernelGGL((k_normalise_small<IT>), dim3(1), dim3(256), 0, st, C, (int)n, q);
        EPG_LAUNCH_CHECK("k_normalise_small");
        return EPG_OK;
    }
    long long* total = reinterpret_cast<long long*>(ws);
    EPG_HIP(hipMemsetAsync(total, 0, 8, st));
    long blocks = (n + 255) / 256;
    if (blocks > num_cus() * 8L) blocks = num_cus() * 8L;
    hipLaunchKernelGGL((k_sum_int<IT>), dim3((int)blocks), dim3(256), 0, st, C, (long)n, total);
    hipLaunchKernelGGL((k_normalise<IT>), dim3((int)blocks), dim3(256), 0, st, C, (long)n, total, q);
    EPG_LAUNCH_CHECK("k_normalise");
    return EPG_OK;
}

int normalise_i64_impl(const int64_t* C, int64_t n, float* q, void* ws, int64_t ws_bytes, hipStream_t st) {
    return normalise_impl<long long>(reinterpret_cast<const long long*>(C), n, q, ws, ws_bytes, st);
}
int normalise_i32_impl(const int32_t* C, int64_t n, float* q, void* ws, int64_t ws_bytes, hipStream_t st) {
    return normalise_impl<int>(C, n, q, ws, ws_bytes, st);
}

}  // namespace epg
