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

// load group t (slots 2t, 2t+1) of a row; the last group of a row is the only one that needs tail handling
template <int NG>
__device__ __forceinline__ void load_group(const char* rowp, int t, int j, const RowGeom& g, u32 (&w)[8]) {
    if (t < NG - 1) {
        load_slot<false>(rowp, 2 * t, j, g, &w[0]);
        load_slot<false>(rowp, 2 * t + 1, j, g, &w[4]);
    } else {
        load_slot<true>(rowp, 2 * t, j, g, &w[0]);
        load_slot<true>(rowp, 2 * t + 1, j, g, &w[4]);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Software-pipelined tile loop (NG > 0): every wave keeps one whole tile (2*NG loads of 1 KiB) in flight.
// Group t of the NEXT tile is requested into the registers group t of the CURRENT tile has just been counted
// out of, so a tile's VALU work overlaps the HBM latency of the next tile inside the wave, with one register
// set.  hipcc cannot be made to keep that order (it hoists the refills into a second register set and waits for
// all of them before the first count), so the loads are inline asm and the vmcnt bookkeeping is ours:
//   steady state, before counting group t: the VMEM ops issued after group t's two loads are
//   2*(NG-1-t) loads of later groups + (<= 3 stores of the previous epilogue) + 2*t refills already issued
//   => s_waitcnt vmcnt(2*(NG-1)) is sufficient whatever the stores do (loads retire in order; a store that
//   retires early only lowers the counter, one that is still pending only makes us wait longer);
//   last tile of the wave (no refills): vmcnt(2*(NG-1-t)).
// Compiler-generated VMEM in the epilogue must be stores only (its own s_waitcnt accounting does not see the
// asm loads; a compiler-inserted vmcnt(0) would be correct but would drain the prefetch).
// ---------------------------------------------------------------------------------------------------------------
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

// POL: cache policy of the streaming loads: 0 default, 1 nt, 2 sc1, 3 sc0 sc1 (the last three bypass the CU's L1)
template <int OFF, int POL>
__device__ __forceinline__ void gload16_asm(u32x4& dst, const char* p) {
    if constexpr (POL == 0) asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(dst) : "v"(p), "i"(OFF) : "memory");
    if constexpr (POL == 1) asm volatile("global_load_dwordx4 %0, %1, off offset:%2 nt" : "=v"(dst) : "v"(p), "i"(OFF) : "memory");
    if constexpr (POL == 2) asm volatile("global_load_dwordx4 %0, %1, off offset:%2 sc1" : "=v"(dst) : "v"(p), "i"(OFF) : "memory");
    if constexpr (POL == 3) asm volatile("global_load_dwordx4 %0, %1, off offset:%2 sc0 sc1" : "=v"(dst) : "v"(p), "i"(OFF) : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vm(u32x4& a, u32x4& b) {
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "i"(N) : "memory");
}

template <int I>
struct IC { static constexpr int value = I; };

template <int N, int I = 0, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(IC<I>{});
        static_for<N, I + 1>(f);
    }
}

template <int S, int NG, int POL, typename Epilogue>
__device__ __forceinline__ void tile_loop_pipelined(const char* __restrict__ X, long R, int N, long ldx, Epilogue&& epilogue) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // keep the tile index in SGPRs
    const int j = lane & 3, b = lane >> 2;
    const RowGeom g = make_geom(N);
    const long ntiles = (R + 15) >> 4;
    const long stride = (long)gridDim.x * 4;
    long tile = (long)blockIdx.x * 4 + wave;
    if (tile >= ntiles) return;

    // lane constants of the row's last group: clamped chunk offsets (relative to the lane base 16*j) and OR-masks
    constexpr int TA = 2 * (NG - 1), TB = 2 * (NG - 1) + 1;      // slots of the last group
    const int cA = 4 * TA + j, cB = 4 * TB + j;
    const long offA = 16L * (cA < g.last ? cA : g.last) - 16L * j;
    const long offB = 16L * (cB < g.last ? cB : g.last) - 16L * j;
    u32 fix[8];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        fix[d] = (cA > g.last ? 0xffffffffu : 0u) | (cA == g.last ? g.tail[d] : 0u);
        fix[4 + d] = (cB > g.last ? 0xffffffffu : 0u) | (cB == g.last ? g.tail[d] : 0u);
    }

    u32x4 wa[NG], wb[NG];   // slot 2t and 2t+1 of group t
    auto issue_group = [&](auto tc, const char* lanep) {
        constexpr int t = decltype(tc)::value;
        if constexpr (t < NG - 1) {
            gload16_asm<128 * t, POL>(wa[t], lanep);
            gload16_asm<128 * t + 64, POL>(wb[t], lanep);
        } else {
            gload16_asm<0, POL>(wa[t], lanep + offA);
            gload16_asm<0, POL>(wb[t], lanep + offB);
        }
    };
    {
        const long row = tile * 16 + b;
        const char* lanep = X + (row < R ? row : R - 1) * ldx + 16 * j;
        static_for<NG>([&](auto tc) { issue_group(tc, lanep); });
    }
    // one tile; HAS_NEXT is compile-time so the steady-state loop has no branches around the asm statements
    // (a branch makes hipcc merge the "+v" operands of the waits through v_mov copies placed BEFORE the wait)
    auto do_tile = [&](auto hn) {
        constexpr bool HAS_NEXT = decltype(hn)::value != 0;
        const long row = tile * 16 + b;
        const long nrow = row + stride * 16;
        const char* nlanep = X + (nrow < R ? nrow : R - 1) * ldx + 16 * j;
        u32 cnt[S];
#pragma unroll
        for (int s = 0; s < S; ++s) cnt[s] = 0;
        static_for<NG>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            wait_vm<HAS_NEXT ? 2 * (NG - 1) : 2 * (NG - 1 - t)>(wa[t], wb[t]);
            u32 w[8] = {wa[t].x, wa[t].y, wa[t].z, wa[t].w, wb[t].x, wb[t].y, wb[t].z, wb[t].w};
            if constexpr (t == NG - 1) {
#pragma unroll
                for (int d = 0; d < 8; ++d) w[d] |= fix[d];
            }
            count_group<S>(w, cnt);
            // pin: every count of this group is complete before the refill below overwrites its registers
            // (the asm statements keep their order; without this hipcc sinks the counting under all refills)
#pragma unroll
            for (int s = 0; s < S; ++s) asm volatile("" : "+v"(cnt[s]));
            if constexpr (HAS_NEXT) issue_group(tc, nlanep);
        });
        epilogue(row, row < R, cnt);
    };
    for (; tile + stride < ntiles; tile += stride) do_tile(IC<1>{});
    do_tile(IC<0>{});
}

// Plain tile loop: NG > 0 issues all loads of a tile up front, NG == 0 handles any N one group at a time.
template <int S, int NG, typename Epilogue>
__device__ __forceinline__ void tile_loop_simple(const char* __restrict__ X, long R, int N, long ldx, Epilogue&& epilogue) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 3, b = lane >> 2;
    const RowGeom g = make_geom(N);
    const long ntiles = (R + 15) >> 4;
    for (long tile = (long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long)gridDim.x * 4) {
        const long row = tile * 16 + b;
        const bool valid = row < R;
        u32 cnt[S];
#pragma unroll
        for (int s = 0; s < S; ++s) cnt[s] = 0;
        count_row<S, NG>(X + (valid ? row : R - 1) * ldx, j, g, cnt);
        epilogue(row, valid, cnt);
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
