// Per-bin state counting core (gfx950).
//
// Work split: a quad (4 lanes) owns one bin, a wave owns 16 consecutive bins, a 256-thread block 64.  A row of N
// state bytes is cut into 16-byte chunks; chunk c of the row goes to quad lane c & 3, load slot c >> 2, so one
// global_load_dwordx4 per slot covers 64 contiguous bytes of each of the wave's 16 rows.  Two slots (32 bytes per
// lane) make a "group": its 8 dwords are bit-transposed in registers into five 32-bit planes P0..P4 (bit b of
// every state byte), from which the indicator word of each state is three ANDs away and v_bcnt_u32_b32
// accumulates it.  Cost: 38 VALU for the transpose + 16 + 2*S for the decode per 32 bytes (~2.8 VALU/byte at
// S = 18), no LDS, no atomics, no data-dependent control flow (so skewed real data -- 71 % of cells in one
// state -- runs at the same speed as uniform data).
//
// Bytes that are not states (row padding, chunks past the row end, rows past R) are forced to 0xFF, which has
// bit 4 and bits 2..3 set and therefore decodes to state 31: never counted for S <= 31.  For the same reason a
// negative / out-of-range input byte is simply not counted (the expected pass notices sum(counts) != R*N).
#pragma once
#include "epg_common.h"

namespace epg {

// one group: 8 dwords = 32 state bytes -> cnt[s] += #bytes equal to s
template <int S>
__device__ __forceinline__ void count_group(const u32 (&w)[8], u32 (&cnt)[S]) {
    // stage 1: low nibbles of dword pairs (2i, 2i+1) share a byte
    u32 n0 = bfi(0x0f0f0f0fu, w[0], w[1] << 4);
    u32 n1 = bfi(0x0f0f0f0fu, w[2], w[3] << 4);
    u32 n2 = bfi(0x0f0f0f0fu, w[4], w[5] << 4);
    u32 n3 = bfi(0x0f0f0f0fu, w[6], w[7] << 4);
    // stage 2: bit pairs (b0,b1) and (b2,b3)
    u32 m0 = bfi(0x33333333u, n0, n1 << 2), m1 = bfi(0x33333333u, n2, n3 << 2);
    u32 r0 = bfi(0xccccccccu, n1, n0 >> 2), r1 = bfi(0xccccccccu, n3, n2 >> 2);
    // stage 3: single bit planes; sample (dword k = 4a+2g+h, byte j) sits at bit 8j + 4h + 2g + a in every plane
    const u32 P0 = bfi(0x55555555u, m0, m1 << 1), P1 = bfi(0xaaaaaaaau, m1, m0 >> 1);
    const u32 P2 = bfi(0x55555555u, r0, r1 << 1), P3 = bfi(0xaaaaaaaau, r1, r0 >> 1);
    // bit 4 plane, same sample order; the bfi masks leave no garbage behind
    u32 c00 = bfi(0x10101010u, w[0], w[4] << 1), c01 = bfi(0x10101010u, w[1], w[5] << 1);
    u32 c10 = bfi(0x10101010u, w[2], w[6] << 1), c11 = bfi(0x10101010u, w[3], w[7] << 1);
    u32 d0 = bfi(0x30303030u, c00, c10 << 2), d1 = bfi(0x30303030u, c01, c11 << 2);
    const u32 P4 = bfi(0xf0f0f0f0u, d1, d0 >> 4);

    u32 A[4], B[4];
    A[3] = P0 & P1; A[1] = P0 ^ A[3]; A[2] = P1 ^ A[3]; A[0] = ~(P0 | P1);
    B[3] = P2 & P3; B[1] = P2 ^ B[3]; B[2] = P3 ^ B[3]; B[0] = ~(P2 | P3);
    const u32 nP4 = ~P4;
    u32 M[2][4];
#pragma unroll
    for (int mid = 0; mid < 4; ++mid) {
        M[0][mid] = B[mid] & nP4;
        M[1][mid] = B[mid] & P4;
    }
#pragma unroll
    for (int s = 0; s < S; ++s) cnt[s] += (u32)__builtin_popcount(A[s & 3] & M[s >> 4][(s >> 2) & 3]);
}

// Per-launch constants of the row geometry (wave-uniform, live in SGPRs)
struct RowGeom {
    int chunks;     // ceil(N / 16)
    int last;       // chunks - 1
    u32 tail[4];    // OR-mask for the last chunk: 0xFF on bytes >= N - 16*last
};

__device__ __forceinline__ RowGeom make_geom(int N) {
    RowGeom g;
    g.chunks = (N + 15) >> 4;
    g.last = g.chunks - 1;
    const int t = N - 16 * g.last;  // 1..16 valid bytes in the last chunk
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        int v = t - 4 * d;
        v = v < 0 ? 0 : (v > 4 ? 4 : v);
        g.tail[d] = v == 4 ? 0u : (0xffffffffu << (8 * v));
    }
    return g;
}

// load slot i (chunk c = 4i + j) of the lane's row; slots that can touch the row end are clamped and masked
template <bool MAYBE_TAIL>
__device__ __forceinline__ void load_slot(const char* rowp, int i, int j, const RowGeom& g, u32* w) {
    const int c = 4 * i + j;
    if (!MAYBE_TAIL) {
        const uint4 v = ld16(rowp + 16 * c);
        w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
    } else {
        const int cc = c < g.last ? c : g.last;
        const uint4 v = ld16(rowp + 16 * cc);
        const u32 inv = c > g.last ? 0xffffffffu : 0u;
        const bool is_last = c == g.last;
        w[0] = v.x | inv | (is_last ? g.tail[0] : 0u);
        w[1] = v.y | inv | (is_last ? g.tail[1] : 0u);
        w[2] = v.z | inv | (is_last ? g.tail[2] : 0u);
        w[3] = v.w | inv | (is_last ? g.tail[3] : 0u);
    }
}

// Count one row's states for this lane's share of chunks.  NG > 0: compile-time number of groups
// (128*(NG-1) < N <= 128*NG), all 2*NG loads are issued up front.  NG == 0: any N, one group in flight.
template <int S, int NG>
__device__ __forceinline__ void count_row(const char* rowp, int j, const RowGeom& g, u32 (&cnt)[S]) {
    if constexpr (NG > 0) {
        u32 w[NG][8];
#pragma unroll
        for (int t = 0; t < NG; ++t) {
            if (t < NG - 1) {
                load_slot<false>(rowp, 2 * t, j, g, &w[t][0]);
                load_slot<false>(rowp, 2 * t + 1, j, g, &w[t][4]);
            } else {
                load_slot<true>(rowp, 2 * t, j, g, &w[t][0]);
                load_slot<true>(rowp, 2 * t + 1, j, g, &w[t][4]);
            }
        }
#pragma unroll
        for (int t = 0; t < NG; ++t) count_group<S>(w[t], cnt);
    } else {
        const int ngroups = (g.chunks + 7) >> 3;
        for (int t = 0; t < ngroups; ++t) {
            u32 w[8];
            if (t < ngroups - 1) {
                load_slot<false>(rowp, 2 * t, j, g, &w[0]);
                load_slot<false>(rowp, 2 * t + 1, j, g, &w[4]);
            } else {
                load_slot<true>(rowp, 2 * t, j, g, &w[0]);
                load_slot<true>(rowp, 2 * t + 1, j, g, &w[4]);
            }
            count_group<S>(w, cnt);
        }
    }
}

// pick x[j] for a quad lane j in 0..3 (3 v_cndmask)
__device__ __forceinline__ u32 sel4(u32 x0, u32 x1, u32 x2, u32 x3, int j) {
    const u32 lo = (j & 1) ? x1 : x0;
    const u32 hi = (j & 1) ? x3 : x2;
    return (j & 2) ? hi : lo;
}

}  // namespace epg
