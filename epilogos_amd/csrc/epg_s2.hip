// S2 path (state-pair saliency) from cached per-bin histograms, plus the paired-mode extras.  gfx950 only.
#include "epg_common.h"

#include <stdlib.h>
#include <string.h>

namespace epg {

// ---------------------------------------------------------------------------------------------------------------
// S2 expected: C[i,j] += sum_b h_i*h_j (i != j), h_i*(h_i - 1) (i == j)      (expected.py:146-158 s2Calc)
// The matrix is symmetric: unordered state pairs i <= j plus sum_b h_i per state (the diagonal's correction).  Exact integers.
// (Round 1's block-wide kernel -- one thread per pair, three block barriers per 128 bins: 70 % of the wave cycles waiting -- was
// replaced by the wave-level kernel below in round 2 and deleted in round 5.)
// ---------------------------------------------------------------------------------------------------------------
typedef unsigned short v2u16 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------------------------------------
// Every WAVE on its own.  A wave owns tiles of 64 bins:
// it fetches a tile's 128*S bytes one tile ahead with whole-line loads, parks them in its own LDS slot, re-packs them as
// bin PAIRS per state (word (i, k) = counts of state i in bins 2k and 2k+1, row stride 33 words) and contracts them with
// v_dot2_u32_u16 in REGISTER TILES: a lane owns a 3 x 3 block of state pairs (groups gi <= gj of three states) and every
// third pair-word k, so six LDS words feed nine dot products (the one-pair-per-thread form read two words per product).
// Diagonal blocks also sum their three states' counts (the -sum h_i of the diagonal).  Only wave-level barriers.
// Exact: a lane's partial sums are < 2^32 while counts are < 4096 (11 words x 2 products); a tile holding a larger count
// takes a plain u64 path.  H2 (may be NULL): a second histogram array of the same shape, added bin by bin before the products --
// the counts of the column concatenation [A|B] of paired mode (helpers.py:173) from the two groups' histograms (halves cannot
// carry: a bin's count over both groups is at most N_A + N_B <= 65535).
// ---------------------------------------------------------------------------------------------------------------
constexpr int S2W_LD = 33;                   // pair-matrix row stride in words
// S2W_MAXPASS = ceil(#block roles / 64): 1 for S <= 30 (55 roles), 2 for S = 31 (66)
template <int S2W_MAXPASS>
__global__ __launch_bounds__(256, S2W_MAXPASS == 1 ? 4 : 2) void k_s2_hist_wave(const u16* __restrict__ H, const u16* __restrict__ H2, long R, int S,
                                                       u64* __restrict__ counts) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ng = (S + 2) / 3, SP = 3 * ng;                 // state groups; rows of the pair matrix (padded rows stay zero)
    const int nroles = ng * (ng + 1) / 2;
    const int nslice = nroles <= 64 ? 64 / nroles : 1;       // lanes per role, each taking every nslice-th pair word
    const int raw_bytes = 64 * S * 2;
    const int wave_bytes = ((raw_bytes + 15) & ~15) + SP * S2W_LD * 4;
    char* s_raw = smem + (size_t)wave * wave_bytes;
    u32* s_p = reinterpret_cast<u32*>(s_raw + ((raw_bytes + 15) & ~15));
    for (int e = lane; e < SP * S2W_LD; e += 64) s_p[e] = 0;
    // block roles of this lane (one per pass)
    int gi[S2W_MAXPASS], gj[S2W_MAXPASS], kq[S2W_MAXPASS];
    u64 acc[S2W_MAXPASS][9], rs[S2W_MAXPASS][3];
#pragma unroll
    for (int p = 0; p < S2W_MAXPASS; ++p) {
        const int role = nroles <= 64 ? (p == 0 && lane < nroles * nslice ? lane / nslice : -1) : lane + 64 * p;
        gi[p] = -1; gj[p] = 0; kq[p] = nroles <= 64 ? lane % nslice : 0;
        if (role >= 0 && role < nroles) {
            int t = role, i = 0;
            while (t >= ng - i) { t -= ng - i; ++i; }
            gi[p] = i; gj[p] = i + t;
        }
#pragma unroll
        for (int c = 0; c < 9; ++c) acc[p][c] = 0;
#pragma unroll
        for (int c = 0; c < 3; ++c) rs[p][c] = 0;
    }
    const int nchunks = (raw_bytes + 15) >> 4;               // 16-byte pieces of a tile: at most three per lane (S <= 31: 248)
    const long total_bytes = R * S * 2;
    const long ntiles = (R + 63) >> 6;
    const long stride = (long)gridDim.x * 4;
    auto fetch = [&](long tile, int c) -> uint4 {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (tile >= ntiles || c >= nchunks) return v;
        // the grid walks the tiles from the LAST to the first: the count pass wrote H front to back just before, its tail is
        // what the memory-side cache still holds (a sum: the order does not matter to the result)
        const long off = (ntiles - 1 - tile) * raw_bytes + 16L * c;
        if (off + 16 <= total_bytes && 16 * c + 16 <= raw_bytes) {
            v = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(H) + off);
            if (H2) {
                const uint4 w = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(H2) + off);
                v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
            }
            return v;
        }
        u16 t[8] = {0, 0, 0, 0, 0, 0, 0, 0};                   // the piece at the end of the tile / of H
        for (int k = 0; k < 8; ++k)
            if (off + 2 * k < total_bytes && 16 * c + 2 * k < raw_bytes) {
                t[k] = *reinterpret_cast<const u16*>(reinterpret_cast<const char*>(H) + off + 2 * k);
                if (H2) t[k] = (u16)(t[k] + *reinterpret_cast<const u16*>(reinterpret_cast<const char*>(H2) + off + 2 * k));
            }
        return make_uint4(t[0] | (u32)t[1] << 16, t[2] | (u32)t[3] << 16, t[4] | (u32)t[5] << 16, t[6] | (u32)t[7] << 16);
    };
    long tile = (long)blockIdx.x * 4 + wave;
    uint4 pre[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) pre[c] = fetch(tile, lane + 64 * c);
    for (; tile < ntiles; tile += stride) {
        __builtin_amdgcn_wave_barrier();
        u32 m = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (lane + 64 * c < nchunks) *reinterpret_cast<uint4*>(s_raw + 16 * (lane + 64 * c)) = pre[c];
            m |= pre[c].x | pre[c].y | pre[c].z | pre[c].w;
        }
        const bool big = __any((m & 0xF000F000u) != 0);       // a count >= 4096 somewhere in the tile
#pragma unroll
        for (int c = 0; c < 4; ++c) pre[c] = fetch(tile + stride, lane + 64 * c);   // lands while this tile is contracted
        __builtin_amdgcn_wave_barrier();
        const u16* raw = reinterpret_cast<const u16*>(s_raw);
        if (!big) {
            for (int e = lane; e < S * 32; e += 64) {
                const int i = e >> 5, k = e & 31;
                s_p[i * S2W_LD + k] = (u32)raw[(2 * k) * S + i] | ((u32)raw[(2 * k + 1) * S + i] << 16);
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int p = 0; p < S2W_MAXPASS; ++p) {
                if (gi[p] < 0) continue;
                const u32* pi = s_p + 3 * gi[p] * S2W_LD;
                const u32* pj = s_p + 3 * gj[p] * S2W_LD;
                u32 part[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, ps[3] = {0, 0, 0};
                const bool diag = gi[p] == gj[p];
                for (int k = kq[p]; k < 32; k += nslice) {
                    const u32 a0 = pi[k], a1 = pi[S2W_LD + k], a2 = pi[2 * S2W_LD + k];
                    const u32 b0 = pj[k], b1 = pj[S2W_LD + k], b2 = pj[2 * S2W_LD + k];
                    const u32 av[3] = {a0, a1, a2}, bv[3] = {b0, b1, b2};
#pragma unroll
                    for (int t = 0; t < 3; ++t)
#pragma unroll
                        for (int u = 0; u < 3; ++u)
                            part[3 * t + u] = __builtin_amdgcn_udot2(__builtin_bit_cast(v2u16, av[t]), __builtin_bit_cast(v2u16, bv[u]), part[3 * t + u], false);
                    if (diag) {
#pragma unroll
                        for (int t = 0; t < 3; ++t)
                            ps[t] = __builtin_amdgcn_udot2(__builtin_bit_cast(v2u16, av[t]), __builtin_bit_cast(v2u16, 0x00010001u), ps[t], false);
                    }
                }
#pragma unroll
                for (int c = 0; c < 9; ++c) acc[p][c] += part[c];
#pragma unroll
                for (int c = 0; c < 3; ++c) rs[p][c] += ps[c];
            }
        } else {                                               // counts >= 4096: 64-bit products straight from the raw tile
            const long r0 = (ntiles - 1 - tile) * 64;
            const int rows = (int)(R - r0 < 64 ? R - r0 : 64);
#pragma unroll
            for (int p = 0; p < S2W_MAXPASS; ++p) {
                if (gi[p] < 0) continue;
                for (int r = kq[p]; r < rows; r += nslice) {
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        const int i = 3 * gi[p] + t;
                        const u64 hi = i < S ? raw[r * S + i] : 0;
#pragma unroll
                        for (int u = 0; u < 3; ++u) {
                            const int j = 3 * gj[p] + u;
                            acc[p][3 * t + u] += hi * (u64)(j < S ? raw[r * S + j] : 0);
                        }
                        if (gi[p] == gj[p]) rs[p][t] += hi;
                    }
                }
            }
        }
    }
    // C[i,j] = sum h_i h_j (both orders);  C[i,i] = sum h_i^2 - sum h_i.  The block's lanes first meet in an LDS copy of C
    // (the staging memory is free now), then one global atomic per cell and block: with every lane adding its nine sums
    // straight to global memory 7.7 M atomics queued up on 324 addresses and the kernel took 3.9 ms.
    __syncthreads();
    u64* s_c = reinterpret_cast<u64*>(smem);
    for (int e = threadIdx.x; e < S * S; e += 256) s_c[e] = 0;
    __syncthreads();
#pragma unroll
    for (int p = 0; p < S2W_MAXPASS; ++p) {
        if (gi[p] < 0) continue;
        const bool diag = gi[p] == gj[p];
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int i = 3 * gi[p] + t, j = 3 * gj[p] + u;
                u64 v = acc[p][3 * t + u];
                if (diag && t == u) v -= rs[p][t];             // two's complement
                if (i >= S || j >= S || !v) continue;
                atomicAdd(&s_c[i * S + j], v);
                if (!diag) atomicAdd(&s_c[j * S + i], v);
            }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < S * S; e += 256)
        if (s_c[e]) atomicAdd(&counts[e], s_c[e]);
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
        const double v = qd == 0.0 ? LPQ_MASKED : log2((double)perms * qd);
        LPQ[t] = v;
        // LPQT[j][i] = LPQ[i][j] / P: a column of LPQ as 8*S contiguous bytes, already divided by the number of ordered pairs
        // (k_score_s2_bin works on tables that carry the 1/P of p = num / P: one multiply per score less)
        LPQ[S * S + 1 + (t % S) * S + t / S] = v * (1.0 / (double)perms);
    }
    if (t == 0) {                                   // LPQ[S*S] = number of masked (q == 0) entries: selects the score kernel
        int nz = 0;
        for (int e = 0; e < S * S; ++e) nz += q[e] == 0.0f;
        LPQ[S * S] = (double)nz;
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
                                                             const double* __restrict__ gLPQ, OT* __restrict__ out,
                                                             int fast_exists) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // layout: LPQ [S*S] doubles | per wave: lh [BPW*S] doubles | per wave: hh [BPW*S] u32
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int BPW = 64 / S;
    double* s_lpq = reinterpret_cast<double*>(smem);
    double* s_lh = s_lpq + S * S + wave * BPW * S;
    u32* s_hh = reinterpret_cast<u32*>(s_lpq + S * S + 4 * BPW * S) + wave * BPW * S;
    if (fast_exists && gLPQ[S * S] == 0.0) return;   // no masked entry and a fast instantiation for this S: that one runs
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
// S2 score, one LANE per bin (no q == 0; S = 15 / 18 / 25).  With p = num / P and log2(p/q) = log2(num) - LPQ[i,j]
// (LPQ = log2(P q)), num = h_i h_j (i != j) or h_j (h_j - 1) (i == j), the sum over i collapses to
//     score[j] = h_j / P * ( A - G_j + LPQ[j,j] + LH[h_j] (n - h_j - 1) + (h_j - 1) LH[h_j - 1] )
//     A = sum_i h_i LH[h_i],  n = sum_i h_i,  G_j = sum_i h_i LPQ[i,j]
// Round 5 (late): the 324 FMAs of G were 42 % of the kernel's 774 VALU instructions per bin, the rest bookkeeping around them.  The
// tables now carry the 1 / P (LH' = LH / P in the block's LDS copy, LPQ' = LPQ / P in the transposed copy k_s2_tables writes), the
// last term comes from a second LDS table U'[h] = (h - 1) LH'[h - 1] fetched with LH'[h] in the first loop (no gather, no
// subtract-multiply in the second), n is summed from the packed words before it: score[j] = h_j ((A' - G'_j + LPQ'[j,j]) + c_j),
// c_j = fma(LH'[h_j], (n - 1) - h_j, U'[h_j]) formed in the first loop (two arrays of S doubles live, as before: 128 VGPRs).
// (h_i = 0 and h_j = 0 give exact zeros like the reference's masked terms).  Everything that depends on (i, j) is the
// S x S matrix-vector product G = h . LPQ.  LPQ does not depend on the bin, so with a lane per bin its entries are
// WAVE-UNIFORM: they are read with scalar loads (column j of LPQ = 8*S contiguous bytes of the transposed copy) and enter
// v_fma_f64 as SGPR operands -- one float64 FMA per (i, j) and nothing else in the inner loop: no LDS broadcast, no idle
// lanes (round 1's (bin, j)-per-lane kernel ran 54 of 64 lanes and spent three float64 operations plus one LDS read per
// term: 1.40 ms for 15 M bins at S = 18; deleted in round 5).  A lane reads its bin's 2*S bytes of H directly (the wave's rows are one
// contiguous span) one iteration ahead, gathers LH[h_i] from the log table in LDS, and stages its S outputs in LDS so
// that the wave writes whole lines.  Float64 throughout; differs from the reference's i-ascending sum by ~1e-13 relative.
// ---------------------------------------------------------------------------------------------------------------
template <int S>
__device__ __forceinline__ void load_hrow(const u16* __restrict__ H, long row, long R, u32 (&w)[(S + 1) / 2]) {
    constexpr int ND = (S + 1) / 2;
    if (row >= R) {
#pragma unroll
        for (int m = 0; m < ND; ++m) w[m] = 0;
        return;
    }
    const char* p = reinterpret_cast<const char*>(H + row * S);          // 2*S bytes, 2-byte aligned (4 for even S)
    if constexpr ((S & 1) == 0) {
#pragma unroll
        for (int m = 0; m < ND; ++m) __builtin_memcpy(&w[m], p + 4 * m, 4);
    } else {
#pragma unroll
        for (int m = 0; m < ND - 1; ++m) __builtin_memcpy(&w[m], p + 4 * m, 4);
        w[ND - 1] = *reinterpret_cast<const u16*>(p + 4 * (ND - 1));       // the last state stands alone: never read past the row
    }
}

template <int S, typename OT, bool LDS_LH>
__global__ __launch_bounds__(256) void k_score_s2_bin(const u16* __restrict__ H, long R, double inv_perms, int maxc,
                                                       const double* __restrict__ gLH, const double* __restrict__ gLPQ,
                                                       OT* __restrict__ out) {
    if (gLPQ[S * S] != 0.0) return;                  // some q == 0: the general kernel runs instead
    constexpr int ND = (S + 1) / 2;
    constexpr int ROWB = S * (int)sizeof(OT);
    __shared__ __attribute__((aligned(16))) char s_stage[4][64 * ROWB];
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* s_LH = reinterpret_cast<double*>(smem);               // [maxc + 1] pairs (LH'[h], U'[h])
    if (LDS_LH) {
        for (int e = threadIdx.x; e <= maxc; e += 256) {
            s_LH[2 * e] = gLH[e] * inv_perms;
            s_LH[2 * e + 1] = e > 0 ? (double)(e - 1) * (gLH[e - 1] * inv_perms) : 0.0;
        }
        __syncthreads();
    }
    const double* __restrict__ LPQT = gLPQ + S * S + 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long ntiles = (R + 63) >> 6;
    const long stride = (long)gridDim.x * 4;
    long tile = (long)blockIdx.x * 4 + wave;
    u32 wnext[ND];
    load_hrow<S>(H, tile * 64 + lane, tile < ntiles ? R : 0, wnext);
    for (; tile < ntiles; tile += stride) {
        u32 w[ND];
#pragma unroll
        for (int m = 0; m < ND; ++m) w[m] = wnext[m];
        load_hrow<S>(H, (tile + stride) * 64 + lane, tile + stride < ntiles ? R : 0, wnext);   // travels while this tile is scored
        // n = the row's sum, from the packed words (two 16-bit partial sums: no carry between the halves below 65536 columns)
        u32 sw = 0;
#pragma unroll
        for (int m = 0; m < ND; ++m) sw += w[m];
        const double n1 = (double)((sw & 0xffffu) + (sw >> 16)) - 1.0;
        double hd[S], cj[S];                           // cj = LH'[h_j] (n - 1 - h_j) + U'[h_j]: all of the bracket that is per state
        double A = 0.0;
#pragma unroll
        for (int i = 0; i < S; ++i) {
            u32 h = (i & 1) ? w[i >> 1] >> 16 : w[i >> 1] & 0xffffu;
            h = h > (u32)maxc ? (u32)maxc : h;       // cannot happen for consistent inputs; keeps the gather in bounds
            hd[i] = (double)h;
            double lh, um;
            if (LDS_LH) {
                const double2 t = *reinterpret_cast<const double2*>(&s_LH[2 * h]);
                lh = t.x;
                um = t.y;
            } else {
                lh = gLH[h] * inv_perms;
                um = h ? (hd[i] - 1.0) * (gLH[h - 1] * inv_perms) : 0.0;
            }
            A = fma(hd[i], lh, A);
            cj[i] = fma(lh, n1 - hd[i], um);
        }
        OT* srow = reinterpret_cast<OT*>(&s_stage[wave][lane * ROWB]);
        // LPQ is re-read through the scalar cache for every tile: as a loop invariant its 2*S*S dwords would be hoisted,
        // would not fit the 102 SGPRs and would be spilled to VGPR lanes (618 v_readlane per tile in the first build)
        // (constant address space: the loads stay scalar even though the pointer is opaque to the optimiser)
        typedef const double __attribute__((address_space(4)))* cdp;
        const double* lpq_ = LPQT;
        asm volatile("" : "+s"(lpq_));
        const cdp lpq = (cdp)lpq_;
#pragma unroll
        for (int j = 0; j < S; ++j) {
            double G = 0.0;
#pragma unroll
            for (int i = 0; i < S; ++i) G = fma(hd[i], lpq[j * S + i], G);        // scalar operand: LPQ is wave-uniform
            const double br = ((A - G) + lpq[j * S + j]) + cj[j];
            // h_j = 0: hd = 0 and br is finite (every table entry is); the + 0.0 of the fma makes it the reference's +0.0, not -0.0
            srow[j] = (OT)__builtin_fma(hd[j], br, 0.0);
        }
        __builtin_amdgcn_wave_barrier();
        const long row0 = tile * 64;
        const int rows = (int)(R - row0 < 64 ? R - row0 : 64);
        store_staged_nt(s_stage[wave], reinterpret_cast<char*>(out) + row0 * ROWB, rows * ROWB, lane);
        __builtin_amdgcn_wave_barrier();
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

// Coalesced copy of `nbytes` (a multiple of 4; src 16-byte aligned) from global memory into a wave's LDS buffer: consecutive
// lanes take consecutive 16-byte chunks, so every load instruction covers whole lines (the mirror image of store_staged).
__device__ __forceinline__ void load_staged(char* lds, const char* src, int nbytes, int lane) {
    const int nchunks = nbytes >> 4;
    for (int c = lane; c < nchunks; c += 64)
        *reinterpret_cast<uint4*>(lds + 16 * c) = *reinterpret_cast<const uint4*>(src + 16 * c);
    for (int o = (nchunks << 4) + 4 * lane; o + 4 <= nbytes; o += 256)
        *reinterpret_cast<u32*>(lds + o) = *reinterpret_cast<const u32*>(src + o);
    // an odd number of uint16 (odd state count x odd number of rows): the last two bytes.  (Round 4: they were left out -- the count
    // of the last state in the LAST bin of a part read stale LDS, one wrong delta per file of a 15-state model whose bin count mod 64
    // is odd; found by tests/test_hip_parity.py::test_paired_s1_several_parts_in_one_launch.)
    if ((nbytes & 2) && lane == 0) *reinterpret_cast<u16*>(lds + nbytes - 2) = *reinterpret_cast<const u16*>(src + nbytes - 2);
}

// d * d as numpy computes it -- rounded to float32 BEFORE it is added.  `__fmul_rn` + `__fadd_rn` are ordinary multiplies and adds to
// the optimiser (hipcc's default -ffp-contract=fast comes with the header they are inlined from, not with this function's pragma):
// in the loops unrolled for a compile-time S it fused them into v_fma_f32 and STEP 4's distance lost its last bit.  The empty asm is
// opaque: the product exists as a register value before anything can be added to it.
__device__ __forceinline__ float sq_nofma(float d) {
    float p = d * d;
    asm volatile("" : "+v"(p));
    return p;
}

// One lane per row, rows handed over through LDS: a wave's 64 rows of a and b are 64*S contiguous floats each, fetched with
// whole-line loads (a lane reading its own 72-byte row straight from memory made every load instruction touch 64 lines:
// 1.37 ms for 15 M bins, 2.4 TB/s); the lane then walks its row in LDS with numpy's blocked order -- eight partial sums in
// registers, no per-lane arrays (a float d[32] indexed by a runtime loop lives in scratch: 4.9 ms) -- writes delta over
// its row of a, and the wave stores the 64 delta rows as whole lines.
__global__ __launch_bounds__(256) void k_pair_finish(const float* __restrict__ a, const float* __restrict__ b, long R,
                                                      int S, float* __restrict__ delta, float* __restrict__ dist, int TR) {
#pragma clang fp contract(off)   // numpy squares, rounds, then adds: no fused multiply-add anywhere in here
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rowb = S * 4;
    char* sa = smem + (size_t)wave * 2 * TR * rowb;        // TR rows per wave and turn: 64, fewer for the wide state models
    char* sb = sa + TR * rowb;
    const long ntiles = (R + TR - 1) / TR;
    for (long tile = (long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long)gridDim.x * 4) {
        const long row0 = tile * TR;
        const int rows = (int)(R - row0 < TR ? R - row0 : TR);
        load_staged(sa, reinterpret_cast<const char*>(a + row0 * S), rows * rowb, lane);
        load_staged(sb, reinterpret_cast<const char*>(b + row0 * S), rows * rowb, lane);
        __builtin_amdgcn_wave_barrier();
        if (lane < rows) {
            const float* pa = reinterpret_cast<const float*>(sa + lane * rowb);
            const float* pb = reinterpret_cast<const float*>(sb + lane * rowb);
            float* pd = reinterpret_cast<float*>(sa + lane * rowb);
            float sd, sq;
            if (S < 8) {                                                   // numpy: plain loop below eight elements
                sd = 0.f;
                sq = 0.f;
                for (int s = 0; s < S; ++s) {
                    const float d = pa[s] - pb[s];
                    pd[s] = d;
                    sd += d;
                    sq += sq_nofma(d);
                }
            } else {
                float rd[8], rq[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float d = pa[k] - pb[k];
                    pd[k] = d;
                    rd[k] = d;
                    rq[k] = sq_nofma(d);
                }
                int i = 8;
                for (; i < S - (S % 8); i += 8) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float d = pa[i + k] - pb[i + k];
                        pd[i + k] = d;
                        rd[k] += d;
                        rq[k] += sq_nofma(d);
                    }
                }
                sd = ((rd[0] + rd[1]) + (rd[2] + rd[3])) + ((rd[4] + rd[5]) + (rd[6] + rd[7]));
                sq = ((rq[0] + rq[1]) + (rq[2] + rq[3])) + ((rq[4] + rq[5]) + (rq[6] + rq[7]));
                for (; i < S; ++i) {
                    const float d = pa[i] - pb[i];
                    pd[i] = d;
                    sd += d;
                    sq += sq_nofma(d);
                }
            }
            if (dist) {
                const float sg = sd > 0.f ? 1.f : (sd < 0.f ? -1.f : sd);  // np.sign: 0 -> 0, nan -> nan
                dist[row0 + lane] = sq * sg;
            }
        }
        __builtin_amdgcn_wave_barrier();
        store_staged_nt(sa, reinterpret_cast<char*>(delta + row0 * S), rows * rowb, lane);
        __builtin_amdgcn_wave_barrier();
    }
}

// quiescent straight from the state matrices: 16 lanes per row, 16-byte chunks compared as four words against the
// quiescent state in every byte.  Most bins fail within their first chunks: a wave (four rows) stops reading a group as
// soon as all of its rows have failed.
__device__ __forceinline__ bool chunk_is(const char* p, int n, u32 pat) {          // n valid bytes (1..16) at p
    if (n >= 16) {
        const uint4 v = ld16(p);
        return ((v.x ^ pat) | (v.y ^ pat) | (v.z ^ pat) | (v.w ^ pat)) == 0;
    }
    bool ok = true;
    for (int k = 0; k < n; ++k) ok = ok && ((u32)(unsigned char)p[k] == (pat & 0xffu));
    return ok;
}

__global__ __launch_bounds__(256) void k_quiescent(const char* __restrict__ XA, int NA, long ldxa, const char* __restrict__ XB,
                                                    int NB, long ldxb, long R, int qstate, uint8_t* __restrict__ mask) {
    const int sub = threadIdx.x & 15;
    const long row = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const u32 pat = (u32)(qstate & 0xff) * 0x01010101u;
    int ok = row < R ? 1 : 0;                                                      // the row's verdict so far, same in its 16 lanes
    for (int g = 0; g < 2; ++g) {
        const char* p = (g ? XB + row * ldxb : XA + row * ldxa);
        const int N = g ? NB : NA;
        for (int c0 = 0; c0 < N; c0 += 256) {
            if (!__any(ok)) break;                                                 // every row of this wave already failed
            const int c = c0 + 16 * sub;
            int v = (ok && c < N) ? (chunk_is(p + c, N - c, pat) ? 1 : 0) : ok;
            for (int off = 8; off > 0; off >>= 1) v &= __shfl_xor(v, off, 16);
            ok = v;
        }
    }
    const int v = ok;
    if (row < R && sub == 0) mask[row] = (uint8_t)v;
}

// ---------------------------------------------------------------------------------------------------------------
// Paired S1, everything after the null groups in ONE pass over the four histograms of a bin (round 3): the scores of A, B
// and the two null groups are gathers from S1 tables (the caller's, [N + 1, S] float32 per group width; scores.py:223-232 store
// float32 scores), delta = scoreA - scoreB, the null distance = sign(sum nd) sum nd^2 of nd = nullA - nullB in numpy's pairwise
// order, and STEP 4's per-bin reduction of delta (roiAndVisualPairwise.py:347-354) -- the arithmetic of k_score_s1_from_hist,
// k_pair_finish and k_pair_metrics, bit for bit, without writing and re-reading four score arrays: 144 B read and 85 B written per
// bin instead of ~950.  One lane per bin; a wave's 64 rows of the four histograms come in and its 64 delta rows go out as whole
// lines through LDS; the tables (52 KB for 379 + 342 biosamples) are copied into LDS once per block.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float text_roundtrip_f5(float v);
constexpr int PF_WAVES_MAX = 12;                     // waves per workgroup: as many 64-row staging areas as fit next to the tables

// SC: the state count at compile time (the loops over the states unroll and their 4 S LDS reads overlap -- with a run-time S and
// six waves per CU the kernel waited out every read: 2.2 ms for 15 M bins against 0.9); 0 = any S.
// With the state count at compile time a lane first takes its 2 S counts of A and B into registers; the wave's 64 delta rows then
// go where those histogram rows were (64 x 4 S bytes either way), so a wave stages 8 S instead of 12 S bytes per bin and twelve
// waves instead of six fit next to 52 KB of tables -- the kernel is latency-bound, occupancy is what it lacks.
// Several PARTS in one launch (round 4): the command line holds a chromosome file's bins as a part of their own -- 24 for hg19 --
// and a launch per part paid the copy of the tables into LDS, the ramp and the tail 24 times (1.5 ms of kernel time for 15 M bins
// against 1.0 in one launch).  The parts' pointers and row counts travel in the kernel argument (PF_MAXP at a time); a wave's
// tile index walks through the parts' tiles in order.  The quiescence mask of scores.py:294-303 (every column of A and of B holds
// the quiescent state <=> its count equals the group's width) comes out of the same pass when a part asks for it.
constexpr int PF_MAXP = 24;
struct PfParts {
    const u16* ha[PF_MAXP];
    const u16* hb[PF_MAXP];
    const u16* hna[PF_MAXP];
    const u16* hnb[PF_MAXP];
    float* delta[PF_MAXP];
    float* ndist[PF_MAXP];
    float* rdist[PF_MAXP];
    int* maxdiff[PF_MAXP];
    unsigned char* mask[PF_MAXP];          // NULL: no mask for this part
    long rows[PF_MAXP];
    long tile0[PF_MAXP + 1];               // first tile (64 rows) of every part, and their total
    int n;
};

template <int SC>
__global__ __launch_bounds__(64 * PF_WAVES_MAX) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_pair_fused_s1(const PfParts pt, int S_, const float* __restrict__ TA,
                                                                  int entA, const float* __restrict__ TB, int entB, const float* __restrict__ TnA,
                                                                  int entnA, const float* __restrict__ TnB, int entnB, int NA, int NB, int qstate) {
#pragma clang fp contract(off)   // numpy squares, rounds, then adds: no fused multiply-add anywhere in here
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int S = SC ? SC : S_;
    // tables first (a null table that is the real group's table is not copied twice), then the waves' staging areas
    float* tA = reinterpret_cast<float*>(smem);
    float* tB = tA + entA;
    float* tnA = TnA == TA ? tA : tB + entB;
    float* tnB = TnB == TB ? tB : (TnA == TA ? tB + entB : tnA + entnA);
    float* tend = (TnB == TB ? (TnA == TA ? tB + entB : tnA + entnA) : tnB + entnB);
    for (int e = threadIdx.x; e < entA; e += blockDim.x) tA[e] = TA[e];
    for (int e = threadIdx.x; e < entB; e += blockDim.x) tB[e] = TB[e];
    if (TnA != TA) for (int e = threadIdx.x; e < entnA; e += blockDim.x) tnA[e] = TnA[e];
    if (TnB != TB) for (int e = threadIdx.x; e < entnB; e += blockDim.x) tnB[e] = TnB[e];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hb = 2 * S, rowb = 4 * S;
    const size_t tab_bytes = ((size_t)(reinterpret_cast<char*>(tend) - smem) + 15) & ~(size_t)15;
    const int nwaves = blockDim.x >> 6;
    char* stage = smem + tab_bytes + (size_t)wave * 64 * (4 * hb + (SC ? 0 : rowb));
    char* sA = stage, *sB = sA + 64 * hb, *snA = sB + 64 * hb, *snB = snA + 64 * hb, *sD = SC ? sA : snB + 64 * hb;
    const long ntiles = pt.tile0[pt.n];
    int part = 0;
    // Round 5: the NEXT tile's four histogram blocks are fetched into registers before this tile is worked on (compile-time S,
    // whole tiles of 64 rows only: 8 S chunks of 16 bytes per array, ceil(S / 8) per lane).  Before, a wave loaded, waited,
    // computed, stored -- twelve waves per CU with one tile each in flight a third of the time: 2.9 TB/s of the pass's own bytes.
    constexpr int PCH = SC ? (SC + 7) / 8 : 1, PNCH = 8 * SC;
    uint4 pre0[PCH], pre1[PCH], pre2[PCH], pre3[PCH];
#pragma unroll
    for (int k = 0; k < PCH; ++k) pre0[k] = pre1[k] = pre2[k] = pre3[k] = make_uint4(0, 0, 0, 0);
    bool have_pre = false;
    const long tstride = (long)gridDim.x * nwaves;
    for (long tile = (long)blockIdx.x * nwaves + wave; tile < ntiles; tile += tstride) {
        while (tile >= pt.tile0[part + 1]) ++part;                   // (a wave's tiles ascend: the part only moves forward)
        const u16* __restrict__ HA = pt.ha[part];
        const u16* __restrict__ HB = pt.hb[part];
        const u16* __restrict__ HnA = pt.hna[part];
        const u16* __restrict__ HnB = pt.hnb[part];
        float* __restrict__ delta = pt.delta[part];
        float* __restrict__ ndist = pt.ndist[part];
        float* __restrict__ rdist = pt.rdist[part];
        int* __restrict__ maxdiff = pt.maxdiff[part];
        unsigned char* __restrict__ mask = pt.mask[part];
        const long R = pt.rows[part];
        const long row0 = (tile - pt.tile0[part]) * 64;
        const int rows = (int)(R - row0 < 64 ? R - row0 : 64);
        if (SC && have_pre) {
#pragma unroll
            for (int k = 0; k < PCH; ++k) {
                const int c = lane + 64 * k;
                if (c < PNCH) {
                    *reinterpret_cast<uint4*>(sA + 16 * c) = pre0[k];
                    *reinterpret_cast<uint4*>(sB + 16 * c) = pre1[k];
                    *reinterpret_cast<uint4*>(snA + 16 * c) = pre2[k];
                    *reinterpret_cast<uint4*>(snB + 16 * c) = pre3[k];
                }
            }
        } else {
            load_staged(sA, reinterpret_cast<const char*>(HA + row0 * S), rows * hb, lane);
            load_staged(sB, reinterpret_cast<const char*>(HB + row0 * S), rows * hb, lane);
            load_staged(snA, reinterpret_cast<const char*>(HnA + row0 * S), rows * hb, lane);
            load_staged(snB, reinterpret_cast<const char*>(HnB + row0 * S), rows * hb, lane);
        }
        have_pre = false;
        if (SC) {
            const long nt = tile + tstride;
            if (nt < ntiles) {
                int np = part;
                while (nt >= pt.tile0[np + 1]) ++np;
                const long nrow0 = (nt - pt.tile0[np]) * 64;
                if (pt.rows[np] - nrow0 >= 64) {                     // a whole tile: its loads run under this tile's arithmetic
                    const char* g0 = reinterpret_cast<const char*>(pt.ha[np] + nrow0 * S);
                    const char* g1 = reinterpret_cast<const char*>(pt.hb[np] + nrow0 * S);
                    const char* g2 = reinterpret_cast<const char*>(pt.hna[np] + nrow0 * S);
                    const char* g3 = reinterpret_cast<const char*>(pt.hnb[np] + nrow0 * S);
#pragma unroll
                    for (int k = 0; k < PCH; ++k) {
                        const int c = lane + 64 * k;
                        if (c < PNCH) {
                            pre0[k] = *reinterpret_cast<const uint4*>(g0 + 16 * c);
                            pre1[k] = *reinterpret_cast<const uint4*>(g1 + 16 * c);
                            pre2[k] = *reinterpret_cast<const uint4*>(g2 + 16 * c);
                            pre3[k] = *reinterpret_cast<const uint4*>(g3 + 16 * c);
                        }
                    }
                    have_pre = true;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < rows) {
            const u16* ha = reinterpret_cast<const u16*>(sA + lane * hb);
            const u16* hbq = reinterpret_cast<const u16*>(sB + lane * hb);
            const u16* hna = reinterpret_cast<const u16*>(snA + lane * hb);
            const u16* hnb = reinterpret_cast<const u16*>(snB + lane * hb);
            float* pd = reinterpret_cast<float*>(sD + lane * rowb);
            // a score = the table entry of (count, state); count 0 scores 0 (k_score_s1_from_hist)
            auto sc = [S](const float* t, u32 c, int s) { return c ? t[(long)c * S + s] : 0.0f; };
            // SC: the counts of A and B into registers before any lane's delta row overwrites them (every lane's reads are
            // issued before the first write: a wave's LDS operations execute in order)
            u32 ca[SC ? SC : 1], cb[SC ? SC : 1];
            if (mask) mask[row0 + lane] = qstate >= 0 && ha[qstate] == (u16)NA && hbq[qstate] == (u16)NB;   // (before any delta row lands)
            if (SC) {
#pragma unroll
                for (int s = 0; s < SC; ++s) {
                    ca[s] = ha[s];
                    cb[s] = hbq[s];
                }
                __builtin_amdgcn_wave_barrier();
                asm volatile("" ::: "memory");
            }
            // delta and STEP 4's reduction of it: ascending states, "%.5f" round trip, ties to the higher state (k_pair_metrics)
            float sq = 0.f, sd = 0.f, best = -1.f;
            int arg = S;
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const float d0 = sc(tA, SC ? ca[SC ? s : 0] : (u32)ha[s], s) - sc(tB, SC ? cb[SC ? s : 0] : (u32)hbq[s], s);
                pd[s] = d0;
                const float d = text_roundtrip_f5(d0);
                sq = __fadd_rn(sq, sq_nofma(d));
                sd = __fadd_rn(sd, d);
                if (fabsf(d) >= best) { best = fabsf(d); arg = s + 1; }
            }
            const float sg = sd > 0.f ? 1.f : (sd < 0.f ? -1.f : sd);
            rdist[row0 + lane] = __fmul_rn(sq, sg);
            maxdiff[row0 + lane] = arg;
            // the null distance: numpy's pairwise order over nd = nullA - nullB (k_pair_finish)
            auto nd = [&](int s) { return sc(tnA, hna[s], s) - sc(tnB, hnb[s], s); };
            float nsd, nsq;
            if (S < 8) {
                nsd = 0.f;
                nsq = 0.f;
                for (int s = 0; s < S; ++s) {
                    const float d = nd(s);
                    nsd += d;
                    nsq += sq_nofma(d);
                }
            } else {
                float rd[8], rq[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float d = nd(k);
                    rd[k] = d;
                    rq[k] = sq_nofma(d);
                }
                int i = 8;
                for (; i < S - (S % 8); i += 8) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float d = nd(i + k);
                        rd[k] += d;
                        rq[k] += sq_nofma(d);
                    }
                }
                nsd = ((rd[0] + rd[1]) + (rd[2] + rd[3])) + ((rd[4] + rd[5]) + (rd[6] + rd[7]));
                nsq = ((rq[0] + rq[1]) + (rq[2] + rq[3])) + ((rq[4] + rq[5]) + (rq[6] + rq[7]));
                for (; i < S; ++i) {
                    const float d = nd(i);
                    nsd += d;
                    nsq += sq_nofma(d);
                }
            }
            const float nsg = nsd > 0.f ? 1.f : (nsd < 0.f ? -1.f : nsd);
            ndist[row0 + lane] = nsq * nsg;
        }
        __builtin_amdgcn_wave_barrier();
        store_staged_nt(sD, reinterpret_cast<char*>(delta + row0 * S), rows * rowb, lane);
        __builtin_amdgcn_wave_barrier();
    }
}

int pair_scores_s1_parts_impl(int32_t nparts, const uint16_t* const* HA, const uint16_t* const* HB, const uint16_t* const* HnA,
                              const uint16_t* const* HnB, const int64_t* R, int32_t S, int32_t NA, int32_t NB, int32_t ga, int32_t gb,
                              const float* TA, const float* TB, const float* TnA, const float* TnB, float* const* delta, float* const* ndist,
                              float* const* rdist, int32_t* const* maxdiff, uint8_t* const* mask, int32_t qstate, hipStream_t st) {
    if (nparts < 0 || S < 1 || S > 127 || NA < 1 || NB < 1 || ga < 1 || gb < 1 || qstate >= S) return fail(EPG_ERR_INVALID_ARG, "pair_scores_s1: bad shape");
    if (nparts == 0) return EPG_OK;
    if (!HA || !HB || !HnA || !HnB || !R || !TA || !TB || !TnA || !TnB || !delta || !ndist || !rdist || !maxdiff)
        return fail(EPG_ERR_INVALID_ARG, "pair_scores_s1: NULL argument");
    if (NA > 65535 || NB > 65535) return fail(EPG_ERR_INVALID_ARG, "pair_scores_s1: more than 65535 columns");
    for (int p = 0; p < nparts; ++p) {
        if (R[p] < 0) return fail(EPG_ERR_INVALID_ARG, "pair_scores_s1: part %d has %lld rows", p, (long long)R[p]);
        if (R[p] == 0) continue;
        if (!HA[p] || !HB[p] || !HnA[p] || !HnB[p] || !delta[p] || !ndist[p] || !rdist[p] || !maxdiff[p])
            return fail(EPG_ERR_INVALID_ARG, "pair_scores_s1: NULL argument in part %d", p);
        if ((reinterpret_cast<uintptr_t>(HA[p]) | reinterpret_cast<uintptr_t>(HB[p]) | reinterpret_cast<uintptr_t>(HnA[p]) |
             reinterpret_cast<uintptr_t>(HnB[p]) | reinterpret_cast<uintptr_t>(delta[p])) & 15)
            return fail(EPG_ERR_INVALID_ARG, "pair_scores_s1: histograms and delta must be 16-byte aligned (part %d)", p);
    }
    const int entA = (NA + 1) * S, entB = (NB + 1) * S, entnA = (ga + 1) * S, entnB = (gb + 1) * S;
    size_t tab = (size_t)(entA + entB + (TnA == TA ? 0 : entnA) + (TnB == TB ? 0 : entnB)) * 4;
    tab = (tab + 15) & ~(size_t)15;
    const bool sc = S == 18 || S == 15 || S == 25;                                   // the instantiations with S at compile time
    const size_t per_wave = (size_t)64 * (4 * 2 * S + (sc ? 0 : 4 * S));
    // the tables of wide groups do not fit next to the staging areas: the caller takes the four score passes, pair_finish and
    // pair_metrics instead (same results)
    if (tab + 4 * per_wave > 160 * 1024)
        return fail(EPG_ERR_UNSUPPORTED, "pair_scores_s1: %zu bytes of LDS needed for groups of %d / %d / %d / %d columns", tab + 4 * per_wave, NA, NB, ga, gb);
    int waves = (int)((160 * 1024 - tab) / per_wave);
    if (waves > PF_WAVES_MAX) waves = PF_WAVES_MAX;
    {
        const char* e = exp_env("EPG_PAIR_WAVES");                                    // (experiments build only)
        if (e && atoi(e) >= 1 && atoi(e) < waves) waves = atoi(e);
    }
    const size_t shmem = tab + (size_t)waves * per_wave;
#define PF_LAUNCH(SC)                                                                                                            \
    do {                                                                                                                         \
        static DynLds lds_attr;                                                                                                  \
        EPG_HIP(ensure_dyn_lds(lds_attr, reinterpret_cast<const void*>(k_pair_fused_s1<SC>), 160 * 1024));                       \
        hipLaunchKernelGGL(k_pair_fused_s1<SC>, dim3((unsigned)blocks), dim3(64 * waves), shmem, st, pt, S, TA, entA, TB, entB, TnA, entnA, \
                           TnB, entnB, NA, NB, qstate);                                                                          \
    } while (0)
    for (int p0 = 0; p0 < nparts;) {                                                  // PF_MAXP parts with rows per launch
        PfParts pt;
        memset(&pt, 0, sizeof(pt));
        long tiles = 0;
        int p = p0;
        for (; p < nparts && pt.n < PF_MAXP; ++p) {
            if (R[p] == 0) continue;
            const int k = pt.n++;
            pt.ha[k] = HA[p]; pt.hb[k] = HB[p]; pt.hna[k] = HnA[p]; pt.hnb[k] = HnB[p];
            pt.delta[k] = delta[p]; pt.ndist[k] = ndist[p]; pt.rdist[k] = rdist[p]; pt.maxdiff[k] = maxdiff[p];
            pt.mask[k] = mask ? mask[p] : nullptr;
            pt.rows[k] = R[p];
            pt.tile0[k] = tiles;
            tiles += (R[p] + 63) / 64;
        }
        pt.tile0[pt.n] = tiles;
        p0 = p;
        if (pt.n == 0) break;
        long blocks = (tiles + waves - 1) / waves;
        if (blocks > num_cus()) blocks = num_cus();
        if (S == 18) PF_LAUNCH(18);
        else if (S == 15) PF_LAUNCH(15);
        else if (S == 25) PF_LAUNCH(25);
        else PF_LAUNCH(0);
        EPG_LAUNCH_CHECK("k_pair_fused_s1");
    }
#undef PF_LAUNCH
    return EPG_OK;
}

int pair_scores_s1_impl(const uint16_t* HA, const uint16_t* HB, const uint16_t* HnA, const uint16_t* HnB, int64_t R, int32_t S, int32_t NA,
                        int32_t NB, int32_t ga, int32_t gb, const float* TA, const float* TB, const float* TnA, const float* TnB, float* delta,
                        float* ndist, float* rdist, int32_t* maxdiff, hipStream_t st) {
    if (R < 0) return fail(EPG_ERR_INVALID_ARG, "pair_scores_s1: bad shape");
    if (R > 0 && (!HA || !HB || !HnA || !HnB || !delta || !ndist || !rdist || !maxdiff)) return fail(EPG_ERR_INVALID_ARG, "pair_scores_s1: NULL argument");
    return pair_scores_s1_parts_impl(1, &HA, &HB, &HnA, &HnB, &R, S, NA, NB, ga, gb, TA, TB, TnA, TnB, &delta, &ndist, &rdist, &maxdiff, nullptr, -1, st);
}

// ---------------------------------------------------------------------------------------------------------------
// host-side dispatch
// ---------------------------------------------------------------------------------------------------------------
int wide_hist_s2_from_binhist(const uint16_t* H, const uint16_t* H2, int64_t R, int32_t S, int64_t* counts, hipStream_t st);
int wide_score_s2_from_hist(const uint16_t* H, int64_t R, int32_t S, int64_t perms, const float* q, double* out64, float* out32, hipStream_t st);

int hist_s2_from_binhist_impl(const uint16_t* H, const uint16_t* H2, int64_t R, int32_t S, int64_t* counts, hipStream_t st) {
    if (R < 0 || S < 1 || S > 127) return fail(EPG_ERR_INVALID_ARG, "hist_s2: bad shape R=%lld S=%d", (long long)R, S);
    if (R == 0) return EPG_OK;
    if (!H || !counts) return fail(EPG_ERR_INVALID_ARG, "hist_s2: NULL argument");
    if (S > 31) return wide_hist_s2_from_binhist(H, H2, R, S, counts, st);          // the wide models: epg_wide.hip
    if ((reinterpret_cast<uintptr_t>(H) & 15) || (reinterpret_cast<uintptr_t>(H2) & 15)) return fail(EPG_ERR_INVALID_ARG, "hist_s2: H must be 16-byte aligned");
    const long ntiles = (R + 63) / 64;
    long blocks = (ntiles + 3) / 4;
    if (blocks > num_cus() * 8L) blocks = num_cus() * 8L;
    const int SP = 3 * ((S + 2) / 3);
    size_t shmem = 4 * ((((size_t)64 * S * 2 + 15) & ~(size_t)15) + (size_t)SP * S2W_LD * 4);
    if (shmem < (size_t)S * S * 8) shmem = (size_t)S * S * 8;
    if (S <= 30) hipLaunchKernelGGL(k_s2_hist_wave<1>, dim3((int)blocks), dim3(256), shmem, st, H, H2, (long)R, S, reinterpret_cast<u64*>(counts));
    else hipLaunchKernelGGL(k_s2_hist_wave<2>, dim3((int)blocks), dim3(256), shmem, st, H, H2, (long)R, S, reinterpret_cast<u64*>(counts));
    EPG_LAUNCH_CHECK("k_s2_hist_wave");
    return EPG_OK;
}

int64_t s2_table_bytes(int maxc, int S) { return align_up((int64_t)(maxc + 1) * 8, 256) + align_up((int64_t)(2 * S * S + 1) * 8, 256); }

int score_s2_from_hist_impl(const uint16_t* H, int64_t R, int32_t N, int32_t S, int64_t perms, const float* q,
                            double* out64, float* out32, void* ws, int64_t ws_bytes, hipStream_t st) {
    if (R < 0 || N < 1 || S < 1 || S > 127 || perms < 1) return fail(EPG_ERR_INVALID_ARG, "score_s2: bad shape R=%lld N=%d S=%d perms=%lld", (long long)R, N, S, (long long)perms);
    if (R == 0) return EPG_OK;
    if (!H || !q || !ws) return fail(EPG_ERR_INVALID_ARG, "score_s2: NULL argument");
    if (S > 31) return wide_score_s2_from_hist(H, R, S, perms, q, out64, out32, st);   // the wide models: epg_wide.hip
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
    // both kernels are launched; LPQ[S*S] (the number of q == 0 entries, known on the device only) decides which works
    const int fast = S == 15 || S == 18 || S == 25;
    if (out32) hipLaunchKernelGGL((k_score_s2_from_hist<float>), dim3((int)blocks), dim3(256), shmem, st, H, (long)R, S, inv, N, LH, LPQ, out32, fast);
    if (out64) hipLaunchKernelGGL((k_score_s2_from_hist<double>), dim3((int)blocks), dim3(256), shmem, st, H, (long)R, S, inv, N, LH, LPQ, out64, fast);
    EPG_LAUNCH_CHECK("k_score_s2_from_hist");
    const bool lds_lh = N < 4096;
    const size_t lh_bytes = lds_lh ? (size_t)(N + 1) * 16 : 0;                    // pairs (LH'[h], U'[h])
    const long ntiles = (R + 63) / 64;
    long bblocks = (ntiles + 3) / 4;
    if (bblocks > num_cus() * 4L) bblocks = num_cus() * 4L;
    // k_score_s2_bin holds a static staging area of 4 waves x 64 rows x S outputs next to the log table: the table goes to LDS
    // only while both fit the 64 KB a workgroup gets without asking (S = 25 with float64 outputs: N <= 1791)
    const bool lds32 = lds_lh && (size_t)256 * S * 4 + lh_bytes <= 65536, lds64 = lds_lh && (size_t)256 * S * 8 + lh_bytes <= 65536;
#define EPG_S2_FAST(SV)                                                                                                         \
    if (S == SV) {                                                                                                              \
        if (out32 && lds32) hipLaunchKernelGGL((k_score_s2_bin<SV, float, true>), dim3((int)bblocks), dim3(256), lh_bytes, st, H, (long)R, inv, N, LH, LPQ, out32);   \
        if (out32 && !lds32) hipLaunchKernelGGL((k_score_s2_bin<SV, float, false>), dim3((int)bblocks), dim3(256), 0, st, H, (long)R, inv, N, LH, LPQ, out32);         \
        if (out64 && lds64) hipLaunchKernelGGL((k_score_s2_bin<SV, double, true>), dim3((int)bblocks), dim3(256), lh_bytes, st, H, (long)R, inv, N, LH, LPQ, out64);  \
        if (out64 && !lds64) hipLaunchKernelGGL((k_score_s2_bin<SV, double, false>), dim3((int)bblocks), dim3(256), 0, st, H, (long)R, inv, N, LH, LPQ, out64);        \
    }
    EPG_S2_FAST(15) EPG_S2_FAST(18) EPG_S2_FAST(25)
#undef EPG_S2_FAST
    EPG_LAUNCH_CHECK("k_score_s2_bin");
    return EPG_OK;
}

// "%.5f" then strtod then float32: v * 1e5 is exact in double (24-bit x 17-bit significands), rint is half-even on that
// exact value like the correctly rounded decimal conversion, and k / 1e5 is the correctly rounded quotient, i.e. the
// double nearest to the decimal k * 10^-5 that the parser produces.
// Round 4: below 2^17 in magnitude the quotient is taken as k * 1e-5 -- one multiply where the float64 division is ~35 instructions,
// eighteen times per bin in the one-pass paired kernel.  It is the same float32: with q = k / 10^5 in [2^E, 2^(E+1)), E <= 17, a
// float32 rounding boundary is m = j 2^(E-24) with j odd, and |q - m| = |k 2^s - j 10^5| / (10^5 2^s), s = 24 - E >= 7; the
// numerator is an integer and cannot be 0 (2^s would have to divide 10^5 j = 2^5 5^5 j), so q lies >= 2^28 / 10^5 = 2684 float64
// ulps from every boundary, while k * double(1e-5) is within 1.5 ulps of q -- and so is strtod's correctly rounded value.
__device__ __forceinline__ float text_roundtrip_f5(float v) {
    const double k = rint((double)v * 1e5);
    return fabsf(v) < 131072.0f ? (float)(k * 1e-5) : (float)(k / 1e5);
}

__global__ __launch_bounds__(256) void k_pair_metrics(const float* __restrict__ delta, long R, int S, int roundtrip,
                                                       float* __restrict__ dist, int* __restrict__ maxdiff, int TR) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rowb = S * 4;
    char* sd_ = smem + (size_t)wave * TR * rowb;                     // the wave's 64 rows, fetched as whole lines
    const long ntiles = (R + TR - 1) / TR;
    for (long tile = (long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long)gridDim.x * 4) {
        const long row0 = tile * TR;
        const int rows = (int)(R - row0 < TR ? R - row0 : TR);
        load_staged(sd_, reinterpret_cast<const char*>(delta + row0 * S), rows * rowb, lane);
        __builtin_amdgcn_wave_barrier();
        if (lane < rows) {
            const float* pr = reinterpret_cast<const float*>(sd_ + lane * rowb);
            float sq = 0.f, sd = 0.f, best = -1.f;
            int arg = S;
            for (int s = 0; s < S; ++s) {
                float d = pr[s];
                if (roundtrip) d = text_roundtrip_f5(d);
                sq = __fadd_rn(sq, sq_nofma(d));          // np.square, then a separate add: no fused multiply-add
                sd = __fadd_rn(sd, d);
                if (fabsf(d) >= best) { best = fabsf(d); arg = s + 1; }   // >= : ties go to the higher state
            }
            const float sg = sd > 0.f ? 1.f : (sd < 0.f ? -1.f : sd);     // np.sign
            dist[row0 + lane] = __fmul_rn(sq, sg);
            maxdiff[row0 + lane] = arg;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

int pair_metrics_impl(const float* delta, int64_t R, int32_t S, int32_t roundtrip, float* dist, int32_t* maxdiff, hipStream_t st) {
    if (R < 0 || S < 1) return fail(EPG_ERR_INVALID_ARG, "pair_metrics: bad shape");
    if (R == 0) return EPG_OK;
    if (!delta || !dist || !maxdiff) return fail(EPG_ERR_INVALID_ARG, "pair_metrics: NULL argument");
    if (reinterpret_cast<uintptr_t>(delta) & 15) return fail(EPG_ERR_INVALID_ARG, "pair_metrics: delta must be 16-byte aligned");
    const int TR = tile_rows(S * 4);
    long blocks = ((R + TR - 1) / TR + 3) / 4;
    if (blocks > num_cus() * 8L) blocks = num_cus() * 8L;
    hipLaunchKernelGGL(k_pair_metrics, dim3((unsigned)blocks), dim3(256), (size_t)4 * TR * S * 4, st, delta, (long)R, S, roundtrip, dist, maxdiff, TR);
    EPG_LAUNCH_CHECK("k_pair_metrics");
    return EPG_OK;
}

int pair_finish_impl(const float* a, const float* b, int64_t R, int32_t S, float* delta, float* dist, hipStream_t st) {
    if (R < 0 || S < 1 || S > 127) return fail(EPG_ERR_INVALID_ARG, "pair_finish: bad shape");   // numpy's pairwise sum recurses above 128
    if (R == 0) return EPG_OK;
    if (!a || !b || !delta) return fail(EPG_ERR_INVALID_ARG, "pair_finish: NULL argument");
    if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(delta)) & 15)
        return fail(EPG_ERR_INVALID_ARG, "pair_finish: a, b and delta must be 16-byte aligned");
    const int TR = tile_rows(2 * S * 4);
    long blocks = ((R + TR - 1) / TR + 3) / 4;
    if (blocks > num_cus() * 8L) blocks = num_cus() * 8L;
    hipLaunchKernelGGL(k_pair_finish, dim3((int)blocks), dim3(256), (size_t)4 * 2 * TR * S * 4, st, a, b, (long)R, S, delta, dist, TR);
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
