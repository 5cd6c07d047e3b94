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
// bits 0..4 set and therefore decodes to state 31: never counted for S <= 31.  Only bits 0..4 of a byte are decoded:
// an input byte b is counted as state b & 31 when that is < S and not at all otherwise.  So -1 (0xFF) and S..31 are
// "not a state" (the count check sum(counts) != R*N reports them), while a byte in 32..254 whose low five bits are < S
// aliases that state -- the contract of the ABI is bytes in 0..31 or 0xFF; the host parser (epg_io.cpp) stores anything
// outside 0..30 as -1 so that files cannot produce an aliasing byte.
#pragma once
#include "epg_common.h"

namespace epg {

// gfx950 VALU issue rates measured with tools/ubench/valu_rate*.hip (SIMD cycles per wave64 instruction):
//   2: v_bitop3_b32, v_and/or/xor/not, v_add/sub_u32, v_lshrrev_b32, v_mov, v_fma_f32
//   4: v_lshlrev_b32, v_bfi_b32, v_bcnt_u32_b32, v_lshl_or/and_or/or3/add3, v_perm, v_bfe, v_cndmask, every DPP/SDWA op
// so the transpose below selects with v_bitop3 (not v_bfi), shifts right where it can, and doubles with v_add.
#define EPG_B3(a, b, c, tt) ((u32)__builtin_amdgcn_bitop3_b32((int)(a), (int)(b), (int)(c), (tt)))

// (m & x) | (~m & y) as one full-rate v_bitop3_b32
__device__ __forceinline__ u32 sel(u32 m, u32 x, u32 y) { return EPG_B3(m, x, y, 0xCA); }

// x << 1 as a full-rate v_add_u32 (LLVM would canonicalise x + x back to the half-rate v_lshlrev_b32)
__device__ __forceinline__ u32 dbl(u32 x) {
    u32 r;
    asm("v_add_u32 %0, %1, %1" : "=v"(r) : "v"(x));
    return r;
}

// one group: 8 dwords = 32 state bytes -> cnt[s] += #bytes equal to s
template <int S>
__device__ __forceinline__ void count_group(const u32 (&w)[8], u32 (&cnt)[S]) {
    // stage 1: low nibbles of dword pairs (2i, 2i+1) share a byte
    const u32 n0 = sel(0x0f0f0f0fu, w[0], w[1] << 4);
    const u32 n1 = sel(0x0f0f0f0fu, w[2], w[3] << 4);
    const u32 n2 = sel(0x0f0f0f0fu, w[4], w[5] << 4);
    const u32 n3 = sel(0x0f0f0f0fu, w[6], w[7] << 4);
    // stage 2: bit pairs (b0,b1) and (b2,b3)
    const u32 m0 = sel(0x33333333u, n0, n1 << 2), m1 = sel(0x33333333u, n2, n3 << 2);
    const u32 r0 = sel(0xccccccccu, n1, n0 >> 2), r1 = sel(0xccccccccu, n3, n2 >> 2);
    // stage 3: single bit planes; sample (dword k = 4a+2g+h, byte j) sits at bit 8j + 4h + 2g + a in every plane
    const u32 P0 = sel(0x55555555u, m0, dbl(m1)), P1 = sel(0xaaaaaaaau, m1, m0 >> 1);
    const u32 P2 = sel(0x55555555u, r0, dbl(r1)), P3 = sel(0xaaaaaaaau, r1, r0 >> 1);
    // bit 4 plane, same sample order; the selects leave no garbage behind
    const u32 c00 = sel(0x10101010u, w[0], dbl(w[4])), c01 = sel(0x10101010u, w[1], dbl(w[5]));
    const u32 c10 = sel(0x10101010u, w[2], dbl(w[6])), c11 = sel(0x10101010u, w[3], dbl(w[7]));
    const u32 d0 = sel(0x30303030u, c00, c10 << 2), d1 = sel(0x30303030u, c01, c11 << 2);
    const u32 P4 = sel(0xf0f0f0f0u, d1, d0 >> 4);

    // decode: L[k] = samples whose low three bits equal k (one bitop3 each), then one bitop3 per state for bits 3,4
    u32 L[8];
    L[0] = EPG_B3(P0, P1, P2, 0x01); L[1] = EPG_B3(P0, P1, P2, 0x10); L[2] = EPG_B3(P0, P1, P2, 0x04); L[3] = EPG_B3(P0, P1, P2, 0x40);
    L[4] = EPG_B3(P0, P1, P2, 0x02); L[5] = EPG_B3(P0, P1, P2, 0x20); L[6] = EPG_B3(P0, P1, P2, 0x08); L[7] = EPG_B3(P0, P1, P2, 0x80);
#pragma unroll
    for (int s = 0; s < S; ++s) {
        u32 ind;
        switch (s >> 3) {                       // (bit3, bit4) of s
            case 0: ind = EPG_B3(L[s & 7], P3, P4, 0x10); break;   // L & ~P3 & ~P4
            case 1: ind = EPG_B3(L[s & 7], P3, P4, 0x40); break;   // L &  P3 & ~P4
            case 2: ind = EPG_B3(L[s & 7], P3, P4, 0x20); break;   // L & ~P3 &  P4
            default: ind = EPG_B3(L[s & 7], P3, P4, 0x80); break;  // L &  P3 &  P4
        }
        cnt[s] += (u32)__builtin_popcount(ind);
    }
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
        // any N: groups in batches of four, the batch's eight loads issued before its first count (one group at a time
        // measured 2.81-2.86 ms against 2.40 ms for 7.5 M bins x 1698 on one box)
        const int ngroups = (g.chunks + 7) >> 3;
        for (int t0 = 0; t0 < ngroups; t0 += 4) {
            u32 w[4][8];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int t = t0 + k;
                if (t < ngroups - 1) {                               // wave-uniform
                    load_slot<false>(rowp, 2 * t, j, g, &w[k][0]);
                    load_slot<false>(rowp, 2 * t + 1, j, g, &w[k][4]);
                } else if (t == ngroups - 1) {
                    load_slot<true>(rowp, 2 * t, j, g, &w[k][0]);
                    load_slot<true>(rowp, 2 * t + 1, j, g, &w[k][4]);
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (t0 + k < ngroups) count_group<S>(w[k], cnt);
        }
    }
}

// Tile loop.  A wave owns "super-tiles" of 32 consecutive bins = two 16-bin tiles counted back to back; NG > 0
// issues all 2*NG loads of a tile up front (13.5 KB in flight per wave at N = 833), NG == 0 handles any N four groups
// at a time.  epilogue(half, row, valid, cnt) gets the lane's partial counts of one tile; finish(st, row0, rows) runs
// once per super-tile so the outputs of 32 bins can be written as whole 128-byte lines (32 rows of H are 1152 bytes
// = 9 lines; 36-byte row pieces written straight from the quads cost 0.35 ms of a 2.4 ms launch: partial-line
// writes interleaved with the read stream).
// Round 2 (profiles/r02d, r02e): with the H store the kernel keeps ~20 % fewer read requests in flight than without it
// while no L2->fabric stall counter moves; the guess that a counting wave's vmcnt (loads and stores in one in-order counter)
// puts the store latency on its load path was tested with a dedicated store wave per block (three counting waves hand
// their super-tiles' H rows to the fourth through an LDS ring): same placement-dependent levels, same mean (2.51 against
// 2.41 ms over four alternating bench runs each, 2.48 against 2.44 ms over eight placements) -- removed.
// Measured alternatives that were NOT faster and were removed: (a) refilling group t of the next tile right after
// counting it, with inline-asm loads and hand-counted vmcnt (2.46 vs 2.33 ms); (b) flat line-granular loads staged
// through LDS so that no 128-byte line is requested twice (2.40 ms); (c) nt / sc1 / sc0 sc1 loads (3.1-3.4 ms).
template <int S, int NG, int NW = 4, typename Epilogue, typename Finish>
__device__ __forceinline__ void tile_loop(const char* __restrict__ X, long R, int N, long ldx, Epilogue&& epilogue, Finish&& finish) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 3, b = lane >> 2;
    const RowGeom g = make_geom(N);
    const long nsuper = (R + 31) >> 5;
    for (long st = (long)blockIdx.x * NW + wave; st < nsuper; st += (long)gridDim.x * NW) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const long row = st * 32 + half * 16 + b;
            const bool valid = row < R;
            u32 cnt[S];
#pragma unroll
            for (int s = 0; s < S; ++s) cnt[s] = 0;
            count_row<S, NG>(X + (valid ? row : R - 1) * ldx, j, g, cnt);
            epilogue(half, row, valid, cnt);
        }
        const long row0 = st * 32;
        finish(st, row0, (int)(R - row0 < 32 ? R - row0 : 32));
    }
}

// The same loop over SEVERAL matrices ("parts": the chromosome files of a genome, both groups of a paired run) in one launch:
// the parts' pointers and shapes travel in the kernel argument, super-tiles are numbered through the parts in order (a
// super-tile never straddles two parts), a wave's super-tile index ascends, so its part only moves forward.  enter(part)
// tells the caller which part the following epilogue / finish calls belong to.  Every part must have a width in the
// instantiation's range (128 (NG - 1) < N <= 128 NG, or NG == 0).
constexpr int KH_MAXP = 48;
struct KhParts {
    const char* x[KH_MAXP];
    u16* h[KH_MAXP];                       // (not used by the loop: the caller's enter() picks it up)
    long rows[KH_MAXP];
    long ldx[KH_MAXP];
    long st0[KH_MAXP + 1];                 // first super-tile (32 rows) of every part, and their total
    int n_cols[KH_MAXP];
    int n;
};

template <int S, int NG, int NW = 4, typename Enter, typename Epilogue, typename Finish>
__device__ __forceinline__ void tile_loop_parts(const KhParts& pt, Enter&& enter, Epilogue&& epilogue, Finish&& finish) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 3, b = lane >> 2;
    const long nsuper = pt.st0[pt.n];
    int part = -1;
    const char* X = nullptr;
    long R = 0, ldx = 0, base = 0, next = 0;
    RowGeom g = make_geom(16);
    for (long st = (long)blockIdx.x * NW + wave; st < nsuper; st += (long)gridDim.x * NW) {
        if (st >= next) {
            do { ++part; next = pt.st0[part + 1]; } while (st >= next);
            X = pt.x[part];
            R = pt.rows[part];
            ldx = pt.ldx[part];
            base = pt.st0[part];
            g = make_geom(pt.n_cols[part]);
            enter(part);
        }
        const long lst = st - base;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const long row = lst * 32 + half * 16 + b;
            const bool valid = row < R;
            u32 cnt[S];
#pragma unroll
            for (int s = 0; s < S; ++s) cnt[s] = 0;
            count_row<S, NG>(X + (valid ? row : R - 1) * ldx, j, g, cnt);
            epilogue(half, row, valid, cnt);
        }
        const long row0 = lst * 32;
        finish(lst, row0, (int)(R - row0 < 32 ? R - row0 : 32));
    }
}

// Pack pairs of per-lane counts into uint16 halves and sum over the quad: d[m] = cnt[2m] | cnt[2m+1] << 16,
// every lane of the quad gets the bin's totals (a total never exceeds N <= 65535, so halves cannot carry).
// Half the DPP adds of an unpacked reduction, and d[] is already the uint16 row layout of H.
template <int S>
__device__ __forceinline__ void pack_reduce(const u32 (&cnt)[S], u32 (&d)[(S + 1) / 2]) {
#pragma unroll
    for (int m = 0; m < (S + 1) / 2; ++m) {
        const u32 hi = 2 * m + 1 < S ? cnt[2 * m + 1] : 0u;
        d[m] = quad_sum(cnt[2 * m] | (hi << 16));
    }
}

// pick x[j] for a quad lane j in 0..3 (3 v_cndmask)
__device__ __forceinline__ u32 sel4(u32 x0, u32 x1, u32 x2, u32 x3, int j) {
    const u32 lo = (j & 1) ? x1 : x0;
    const u32 hi = (j & 1) ? x3 : x2;
    return (j & 2) ? hi : lo;
}

}  // namespace epg
