// S3 expected pass as a matrix-core contraction of one-hot operands (gfx950: fp4 MX v_mfma_scale_f32_32x32x64_f8f6f4 by
// default, int8 v_mfma_i32_32x32x32_i8 as the alternative).
//
// C[a,b,i,j] = #{bins : x[a] == i and x[b] == j} is G = E^T E for the one-hot expansion E[bin, (sample, state)]
// (reference expected.py:183-200 increments exactly these cells, one bin at a time).  This is a genuine dense
// contraction -- K = all bins, M = N = biosamples*states = 14 994 at N = 833 -- and the one place on this path where
// the matrix cores are the right tool (SURVEY 7 "S3"); everything else is histogram/elementwise work.
//
//  * k_transpose_states writes XT[sample][bin] (bins padded with 31, states outside [0,S) turned into 31) so that the
//    bins of ONE (sample, state) row are contiguous bytes.
//  * one wave per pair of 96-row blocks (3x3 tiles of 32x32, 144 accumulator registers).  The one-hot operand is never
//    materialised in memory: lane l of tile t owns row m = 32t + (l & 31) = (sample, state) and turns its state bytes
//    into one-hot bytes / nibbles with a SWAR equality test against its own state (one VALU per bin).  A and B
//    operands come from the same routine, so both see bins in the same order and the K sum pairs them.
//  * symmetry: only block pairs bm <= bn are computed; off-diagonal blocks also write the mirrored cells
//    C[b,a,j,i].  The diagonal a == b is skipped (it stays 0 like the reference's).  Integer (or exact float32)
//    accumulation, int32 atomics combine the K splits.
//  * kernels, ms per 1 M bins at N = 833, S = 18:  int8, per-lane loads, operands of step k+1 built under the MFMAs of
//    step k (sched_group_barrier) 138;  B: int8, per-lane loads, two-deep load ring 122;  int8, operand bytes through
//    LDS 93 (with the sched_group_barrier pipelining 103-112);  E: fp4 MX, bytes through LDS 83;  F: E with the A
//    operands built once per workgroup 71.  tools/ubench/mfma_valu.hip: on a SIMD the time of nine MFMAs and of the VALU
//    work between them ADD UP at 1, 2 and 4 waves (9 fp4 MFMA 400 cycles, 144 v_xor 360, together 830-890), so the
//    one-hot VALU (about one instruction per row and bin) costs as much as the matrix work: F spends ~400 cycles of
//    MFMA and ~350 of VALU per 64-bin step and measures 908.
#include "epg_common.h"

#include <stdlib.h>

namespace epg {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int MF_T = 3;               // tiles per block side
constexpr int MF_ROWS = 32 * MF_T;    // (sample, state) rows per block

__global__ __launch_bounds__(256) void k_transpose_states(const char* __restrict__ X, long R, int N, long ldx, int S,
                                                           char* __restrict__ XT, long Rp, int shift, int bad, int* __restrict__ dirty) {
    __shared__ unsigned char tile[64][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const long b0 = (long)blockIdx.x * 64;
    const int s0 = blockIdx.y * 64;
    bool seen_bad = false;                            // a byte of the matrix proper that is not a state
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const long bin = b0 + ty + 4 * i;
        const int smp = s0 + tx;
        unsigned char v = (unsigned char)bad;         // "not a state": bins past R, states outside [0, S)
        if (bin < R && smp < N) {
            v = (unsigned char)X[bin * ldx + smp];
            if (v >= S) { v = (unsigned char)bad; seen_bad = true; }
        }
        tile[ty + 4 * i][tx] = (unsigned char)(v << shift);
    }
    if (dirty && __any(seen_bad) && (threadIdx.x & 63) == 0) atomicOr(dirty, 1);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int smp = s0 + ty + 4 * i;
        const long bin = b0 + tx;
        if (smp < N && bin < Rp) XT[(long)smp * Rp + bin] = (char)tile[tx][ty + 4 * i];
    }
}

// The same transpose with 16-byte global accesses (round 5): a thread loads 16 state bytes of one bin, the tile goes through LDS, a
// thread stores 16 bins of one biosample.  The byte-per-thread form above moved 25 GB at 1.25 TB/s (20 ms per transpose of the
// 15 M-bin genome, two per S3 job).  Needs rows of at least N bytes readable in 16-byte pieces: a piece that would reach past the
// row pitch is read byte by byte.
__global__ __launch_bounds__(256) void k_transpose_states16(const char* __restrict__ X, long R, int N, long ldx, int S,
                                                             char* __restrict__ XT, long Rp, int shift, int bad, int* __restrict__ dirty) {
    constexpr int LD = 68;                            // tile row pitch in bytes: 17 dwords, the four 16-bin groups fall on different banks
    __shared__ __attribute__((aligned(16))) unsigned char tile[64 * LD];
    const int t = threadIdx.x;
    // one-dimensional grid, the biosample tile fastest: the ceil(N / 64) blocks that share 64 bins run side by side, so the second
    // half of every 128-byte line of the state matrix they read is still in L2 (bin tile fastest: each line was fetched twice)
    const int nst = (N + 63) / 64;
    const long b0 = (long)(blockIdx.x / nst) * 64;
    const int s0 = (int)(blockIdx.x % nst) * 64;
    bool seen_bad = false;
    {
        const int row = t >> 2, c = t & 3;
        const long bin = b0 + row;
        const int smp0 = s0 + 16 * c;
        unsigned char v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = (unsigned char)bad;
        if (bin < R && smp0 < N) {
            const char* p = X + bin * ldx + smp0;
            if (smp0 + 16 <= ldx) {
                const uint4 w = ld16(p);
                __builtin_memcpy(v, &w, 16);
            } else {
                for (int i = 0; i < 16 && smp0 + i < N; ++i) v[i] = (unsigned char)p[i];
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (smp0 + i >= N) v[i] = (unsigned char)bad;
                else if (v[i] >= S) { v[i] = (unsigned char)bad; seen_bad = true; }
            }
        }
        u32 w4[4];
#pragma unroll
        for (int d = 0; d < 4; ++d)
            w4[d] = ((u32)(unsigned char)(v[4 * d] << shift)) | ((u32)(unsigned char)(v[4 * d + 1] << shift) << 8) |
                    ((u32)(unsigned char)(v[4 * d + 2] << shift) << 16) | ((u32)(unsigned char)(v[4 * d + 3] << shift) << 24);
#pragma unroll
        for (int d = 0; d < 4; ++d) *reinterpret_cast<u32*>(&tile[row * LD + 16 * c + 4 * d]) = w4[d];
    }
    if (dirty && __any(seen_bad) && (t & 63) == 0) atomicOr(dirty, 1);
    __syncthreads();
    {
        const int sl = t >> 2, k = t & 3;             // local biosample, group of 16 bins
        const int smp = s0 + sl;
        const long bin = b0 + 16 * k;
        if (smp < N && bin < Rp) {
            u32 w4[4];
#pragma unroll
            for (int d = 0; d < 4; ++d)
                w4[d] = (u32)tile[(16 * k + 4 * d) * LD + sl] | ((u32)tile[(16 * k + 4 * d + 1) * LD + sl] << 8) |
                        ((u32)tile[(16 * k + 4 * d + 2) * LD + sl] << 16) | ((u32)tile[(16 * k + 4 * d + 3) * LD + sl] << 24);
            if (bin + 16 <= Rp) {
                *reinterpret_cast<uint4*>(XT + (long)smp * Rp + bin) = make_uint4(w4[0], w4[1], w4[2], w4[3]);
            } else {
                for (int i = 0; i < 16 && bin + i < Rp; ++i) XT[(long)smp * Rp + bin + i] = (char)(w4[i >> 2] >> (8 * (i & 3)));
            }
        }
    }
}

// bytes of w are in [0, 31] (XT is sanitised), pat = the lane's state in every byte (<= 30): four full-rate VALU
__device__ __forceinline__ u32 eq_bytes(u32 w, u32 pat) {
    const u32 t = w ^ pat;                                        // 0 where equal, < 32 elsewhere
    const u32 u = t + 0x7f7f7f7fu;                                // bit 7 of a byte set <=> byte != pattern (no carries)
    return ~(u >> 7) & 0x01010101u;                               // 1 where equal
}

__device__ __forceinline__ v4i onehot16(const uint4 raw, u32 pat) {
    v4i r;
    r.x = (int)eq_bytes(raw.x, pat);
    r.y = (int)eq_bytes(raw.y, pat);
    r.z = (int)eq_bytes(raw.z, pat);
    r.w = (int)eq_bytes(raw.w, pat);
    return r;
}

// Variant B (S < 14, where a block's 96 rows can span more than 8 biosamples): every lane loads its own 16 bytes.  No
// operand double-buffering inside a wave (the other wave of the SIMD fills the matrix pipe while this one builds its
// operands); the registers hold a two-deep ring of raw loads, so a load has two whole steps to land.  122 ms per 1 M
// bins at N = 833, S = 18: bounded by the 64 B/clk L1 path, every byte travels ~14 times.
__global__ __launch_bounds__(64, 2) void k_s3_hist_mfma_b(const char* __restrict__ XT, long Rp, long ksplit_len, int N, int S,
                                                          int nblk, int* __restrict__ counts) {
    const int lane = threadIdx.x;
    const int NS = N * S;
    int p = blockIdx.x, bm = 0;
    while (p >= nblk - bm) { p -= nblk - bm; ++bm; }
    const int bn = bm + p;
    const long kbeg = (long)blockIdx.y * ksplit_len;
    const long kend = kbeg + ksplit_len < Rp ? kbeg + ksplit_len : Rp;

    const char* pA[MF_T];
    const char* pB[MF_T];
    u32 patA[MF_T], patB[MF_T];
#pragma unroll
    for (int t = 0; t < MF_T; ++t) {
        const int m = (bm * MF_T + t) * 32 + (lane & 31);
        const int n = (bn * MF_T + t) * 32 + (lane & 31);
        const int am = m < NS ? m / S : 0, im = m < NS ? m % S : 30;
        const int an = n < NS ? n / S : 0, in_ = n < NS ? n % S : 30;
        pA[t] = XT + (long)am * Rp + 16 * (lane >> 5) + kbeg;
        pB[t] = XT + (long)an * Rp + 16 * (lane >> 5) + kbeg;
        patA[t] = (u32)im * 0x01010101u;
        patB[t] = (u32)in_ * 0x01010101u;
    }
    v16i acc[MF_T][MF_T];
#pragma unroll
    for (int a = 0; a < MF_T; ++a)
#pragma unroll
        for (int b = 0; b < MF_T; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0;

    const long nsteps = (kend - kbeg) / 32;       // kbeg, kend are multiples of 32
    uint4 ra[2][MF_T], rb[2][MF_T];
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        const long off = d < nsteps ? 32L * d : 0;
#pragma unroll
        for (int t = 0; t < MF_T; ++t) {
            ra[d][t] = *reinterpret_cast<const uint4*>(pA[t] + off);
            rb[d][t] = *reinterpret_cast<const uint4*>(pB[t] + off);
        }
    }
    auto step = [&](const int d, long k) {         // consume ring slot d (step k), refill it with step k + 2
        v4i fa[MF_T], fb[MF_T];
#pragma unroll
        for (int t = 0; t < MF_T; ++t) {
            fa[t] = onehot16(ra[d][t], patA[t]);
            fb[t] = onehot16(rb[d][t], patB[t]);
        }
        const long off = k + 2 < nsteps ? 32 * (k + 2) : 0;
#pragma unroll
        for (int t = 0; t < MF_T; ++t) {
            ra[d][t] = *reinterpret_cast<const uint4*>(pA[t] + off);
            rb[d][t] = *reinterpret_cast<const uint4*>(pB[t] + off);
        }
        __builtin_amdgcn_s_setprio(1);                      // the wave that has its operands gets the matrix pipe
#pragma unroll
        for (int a = 0; a < MF_T; ++a)
#pragma unroll
            for (int b = 0; b < MF_T; ++b) acc[a][b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[a], fb[b], acc[a][b], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    long k = 0;
    for (; k + 1 < nsteps; k += 2) {
        step(0, k);
        step(1, k + 1);
    }
    if (k < nsteps) step(0, k);

    const long SS = (long)S * S;
#pragma unroll
    for (int ta = 0; ta < MF_T; ++ta)
#pragma unroll
        for (int tb = 0; tb < MF_T; ++tb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int v = acc[ta][tb][r];
                if (!v) continue;
                const int m = (bm * MF_T + ta) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int n = (bn * MF_T + tb) * 32 + (lane & 31);
                if (m >= NS || n >= NS) continue;
                const int a = m / S, i = m - a * S, b = n / S, j = n - b * S;
                if (a == b) continue;
                atomicAdd(&counts[((long)a * N + b) * SS + i * S + j], v);
                if (bm != bn) atomicAdd(&counts[((long)b * N + a) * SS + j * S + i], v);
            }
}

// Operand bytes through LDS (variants E and F).  The 96 rows of a block belong to at most 7 biosamples and the 18
// state rows of a biosample all need the same 16 bytes, so loading per lane moves every byte ~14 times through the
// 64 B/clk L1 path, which is what bounded variants A/B (PMC: neither the matrix pipe nor the VALU above 40 %).  Here one
// global_load_dwordx4 per side fetches the distinct bytes of four k-steps (8 biosamples x 128 bins), one ds_write_b128
// parks them in LDS and the lanes pick their 16 bytes with broadcast ds_read_b128 (256 B/clk).  One wave per
// workgroup: no barriers, LDS operations of a wave execute in order.
constexpr int MC_SMP = 8;                 // biosample slots per side (96 rows span at most 7 biosamples of >= 15 states)
constexpr int MC_CH = 8;                  // 16-byte chunks per macro-step (128 bins = 4 k-steps)
constexpr int MC_LD = MC_CH + 1;          // chunk stride in uint4: 144 B between biosamples keeps the b128 groups on distinct banks

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

// Variant E: the same contraction on the MX path, v_mfma_scale_f32_32x32x64_f8f6f4 with both operands in fp4 (E2M1) and
// unit block scales (E8M0 127): one instruction covers 64 bins in the time the int8 one covers 32.  A one-hot entry is
// the fp4 encoding of 1.0 (0b0010); counts accumulate in float32, exact below 2^24, which the K split guarantees.  A
// lane's 32 bins are 32 state bytes -> eight dwords -> four dwords of nibbles; which bin lands in which nibble does
// not matter as long as A and B agree, and both come from this routine.
__device__ __forceinline__ u32 eq_pair_fp4(u32 w0, u32 w1, u32 pat) {
    const u32 d0 = 0x80808080u - (w0 ^ pat);                     // bit 7 of a byte set <=> byte == pattern (bytes < 32)
    const u32 d1 = 0x80808080u - (w1 ^ pat);
    return ((d0 >> 6) & 0x02020202u) | ((d1 >> 2) & 0x20202020u);
}

__device__ __forceinline__ v8i onehot32_fp4(const uint4 r0, const uint4 r1, u32 pat) {
    v8i r;
    r[0] = (int)eq_pair_fp4(r0.x, r0.y, pat);
    r[1] = (int)eq_pair_fp4(r0.z, r0.w, pat);
    r[2] = (int)eq_pair_fp4(r1.x, r1.y, pat);
    r[3] = (int)eq_pair_fp4(r1.z, r1.w, pat);
    r[4] = r[5] = r[6] = r[7] = 0;                               // fp4 operands use four registers; the rest is not encoded
    return r;
}

__global__ __launch_bounds__(64, 2) void k_s3_hist_mfma_e(const char* __restrict__ XT, long Rp, long ksplit_len, int N, int S,
                                                          int nblk, int* __restrict__ counts) {
    __shared__ uint4 lds[2][2][MC_SMP][MC_LD];
    const int lane = threadIdx.x;
    const int NS = N * S;
    int p = blockIdx.x, bm = 0;
    while (p >= nblk - bm) { p -= nblk - bm; ++bm; }
    const int bn = bm + p;
    const long kbeg = (long)blockIdx.y * ksplit_len;
    const long kend = kbeg + ksplit_len < Rp ? kbeg + ksplit_len : Rp;
    const long nsteps = (kend - kbeg) / 64;                     // K = 64 bins per MFMA; kbeg, kend are multiples of 64

    // first biosample of each side's block; a lane's rows address slots relative to it
    const int sA0 = (bm * MF_ROWS) / S, sB0 = (bn * MF_ROWS) / S;
    u32 rdA[MF_T], rdB[MF_T], patA[MF_T], patB[MF_T];           // LDS byte offsets of this lane's rows inside a buffer
#pragma unroll
    for (int t = 0; t < MF_T; ++t) {
        const int m = (bm * MF_T + t) * 32 + (lane & 31);
        const int n = (bn * MF_T + t) * 32 + (lane & 31);
        const int am = m < NS ? m / S : N - 1, im = m < NS ? m % S : 30;      // rows past N*S match nothing (S <= 30)
        const int an = n < NS ? n / S : N - 1, in_ = n < NS ? n % S : 30;
        int la = am - sA0, lb = an - sB0;
        la = la < MC_SMP ? la : MC_SMP - 1;                                    // only rows past N*S can exceed the slots
        lb = lb < MC_SMP ? lb : MC_SMP - 1;
        rdA[t] = (u32)(((0 * MC_SMP + la) * MC_LD + 2 * (lane >> 5)) * 16);
        rdB[t] = (u32)(((1 * MC_SMP + lb) * MC_LD + 2 * (lane >> 5)) * 16);
        patA[t] = (u32)im * 0x01010101u;
        patB[t] = (u32)in_ * 0x01010101u;
    }
    // staging role of this lane: biosample slot lane >> 3, chunk lane & 7
    const int slot = lane >> 3, chunk = lane & 7;
    const int gsa = sA0 + slot < N ? sA0 + slot : N - 1, gsb = sB0 + slot < N ? sB0 + slot : N - 1;
    const char* gA = XT + (long)gsa * Rp;
    const char* gB = XT + (long)gsb * Rp;
    const u32 wrA = (u32)(((0 * MC_SMP + slot) * MC_LD + chunk) * 16), wrB = (u32)(((1 * MC_SMP + slot) * MC_LD + chunk) * 16);
    char* ldsc = reinterpret_cast<char*>(&lds[0][0][0][0]);
    constexpr u32 BUF = 2 * MC_SMP * MC_LD * 16;
    auto gload = [&](long M, uint4& va, uint4& vb) {             // macro-step M: bins kbeg + 128 M + 16 chunk ..
        long off = kbeg + 128 * M + 16 * chunk;
        off = off < Rp - 16 ? off : Rp - 16;                     // tail chunks past the slice are never consumed
        va = *reinterpret_cast<const uint4*>(gA + off);
        vb = *reinterpret_cast<const uint4*>(gB + off);
    };

    v16f acc[MF_T][MF_T];                                        // exact: a wave's K range is < 2^24 bins
#pragma unroll
    for (int a = 0; a < MF_T; ++a)
#pragma unroll
        for (int b = 0; b < MF_T; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    uint4 va, vb;
    gload(0, va, vb);
    *reinterpret_cast<uint4*>(ldsc + wrA) = va;
    *reinterpret_cast<uint4*>(ldsc + wrB) = vb;
    gload(1, va, vb);
    const long nmacro = (nsteps + 1) / 2;                        // 128 bins = two k-steps of 64
    for (long M = 0; M < nmacro; ++M) {
        const u32 cur = (u32)(M & 1) * BUF, nxt = BUF - cur;
        *reinterpret_cast<uint4*>(ldsc + nxt + wrA) = va;
        *reinterpret_cast<uint4*>(ldsc + nxt + wrB) = vb;
        gload(M + 2, va, vb);
        const int ns = nsteps - 2 * M < 2 ? (int)(nsteps - 2 * M) : 2;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            if (s2 >= ns) break;                                 // wave-uniform
            v8i fa[MF_T], fb[MF_T];
#pragma unroll
            for (int t = 0; t < MF_T; ++t) {
                const char* pa = ldsc + cur + rdA[t] + 64 * s2;
                const char* pb = ldsc + cur + rdB[t] + 64 * s2;
                fa[t] = onehot32_fp4(*reinterpret_cast<const uint4*>(pa), *reinterpret_cast<const uint4*>(pa + 16), patA[t]);
                fb[t] = onehot32_fp4(*reinterpret_cast<const uint4*>(pb), *reinterpret_cast<const uint4*>(pb + 16), patB[t]);
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int a = 0; a < MF_T; ++a)
#pragma unroll
                for (int b = 0; b < MF_T; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa[a], fb[b], acc[a][b], 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            __builtin_amdgcn_s_setprio(0);
        }
    }

    const long SS = (long)S * S;
#pragma unroll
    for (int ta = 0; ta < MF_T; ++ta)
#pragma unroll
        for (int tb = 0; tb < MF_T; ++tb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int v = (int)acc[ta][tb][r];
                if (!v) continue;
                const int m = (bm * MF_T + ta) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int n = (bn * MF_T + tb) * 32 + (lane & 31);
                if (m >= NS || n >= NS) continue;
                const int a = m / S, i = m - a * S, b = n / S, j = n - b * S;
                if (a == b) continue;
                atomicAdd(&counts[((long)a * N + b) * SS + i * S + j], v);
                if (bm != bn) atomicAdd(&counts[((long)b * N + a) * SS + j * S + i], v);
            }
}

// Variant F (default): variant E with the A operands shared by a workgroup.  In E a wave builds six one-hot tile operands per step
// (~34 VALU each) for nine MFMAs and is VALU-bound at a quarter of the fp4 rate.  Here a workgroup of eight waves takes
// one A block and eight consecutive B blocks: the three A tile operands of a step are built ONCE (wave w builds step w
// of the next 512-bin macro-step, per-lane loads, one ds_write_b128 per tile) and every wave reads them back with
// ds_read_b128, so a wave builds 3 + 3/8 operands per step instead of 6.  B bytes go through the wave's private LDS
// staging exactly as in E.  One barrier per 8 steps.
constexpr int MFF_WAVES = 8;
constexpr int MFF_KS = 8;                 // k-steps (of 64 bins) per A macro-step

__global__ __launch_bounds__(64 * MFF_WAVES, 2) void k_s3_hist_mfma_f(const char* __restrict__ XT, long Rp, long ksplit_len, int N,
                                                                       int S, int nblk, int* __restrict__ counts) {
    __shared__ uint4 rawB[MFF_WAVES][2][MC_SMP][MC_LD];
    __shared__ uint4 opA[2][MFF_KS][MF_T][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int NS = N * S;
    // task -> (A block bm, group of eight B blocks starting at bm + 8 g)
    int p = blockIdx.x, bm = 0;
    for (;;) {
        const int ng = (nblk - bm + MFF_WAVES - 1) / MFF_WAVES;
        if (p < ng) break;
        p -= ng;
        ++bm;
    }
    const int bn = bm + MFF_WAVES * p + w;
    const bool active = bn < nblk;                               // wave-uniform
    const long kbeg = (long)blockIdx.y * ksplit_len;
    const long kend = kbeg + ksplit_len < Rp ? kbeg + ksplit_len : Rp;
    const long nAM = (kend - kbeg) / (64 * MFF_KS);              // kbeg, kend are multiples of 512

    // ---- A side: this wave builds k-step w of every macro-step
    const char* aRow[MF_T];
    u32 patA[MF_T];
#pragma unroll
    for (int t = 0; t < MF_T; ++t) {
        const int m = (bm * MF_T + t) * 32 + (lane & 31);
        const int am = m < NS ? m / S : N - 1, im = m < NS ? m % S : 30;
        aRow[t] = XT + (long)am * Rp + kbeg + 64 * w + 32 * (lane >> 5);
        patA[t] = (u32)im * 0x01010101u;
    }
    uint4 ar0, ar1;                                              // raw bytes of the A tile being built (loaded a step ahead)
    auto loadA = [&](long AM, int t) {
        const char* src = aRow[t] + 64L * MFF_KS * AM;
        ar0 = *reinterpret_cast<const uint4*>(src);
        ar1 = *reinterpret_cast<const uint4*>(src + 16);
    };
    auto buildA = [&](int buf, int t) {
        const v8i f = onehot32_fp4(ar0, ar1, patA[t]);
        opA[buf][w][t][lane] = make_uint4((u32)f[0], (u32)f[1], (u32)f[2], (u32)f[3]);
    };

    // ---- B side: private staging as in variant E
    const int bnc = active ? bn : bm;
    const int sB0 = (bnc * MF_ROWS) / S;
    u32 rdB[MF_T], patB[MF_T];
#pragma unroll
    for (int t = 0; t < MF_T; ++t) {
        const int n = (bnc * MF_T + t) * 32 + (lane & 31);
        const int an = n < NS ? n / S : N - 1, in_ = n < NS ? n % S : 30;
        int lb = an - sB0;
        lb = lb < MC_SMP ? lb : MC_SMP - 1;
        rdB[t] = (u32)((lb * MC_LD + 2 * (lane >> 5)) * 16);
        patB[t] = (u32)in_ * 0x01010101u;
    }
    const int slot = lane >> 3, chunk = lane & 7;
    const int gsb = sB0 + slot < N ? sB0 + slot : N - 1;
    const char* gB = XT + (long)gsb * Rp;
    char* rawc = reinterpret_cast<char*>(&rawB[w][0][0][0]);
    const u32 wrB = (u32)((slot * MC_LD + chunk) * 16);
    constexpr u32 BUF = MC_SMP * MC_LD * 16;
    auto gload = [&](long M) {                                   // B macro-step M: bins kbeg + 128 M + 16 chunk ..
        long off = kbeg + 128 * M + 16 * chunk;
        off = off < Rp - 16 ? off : Rp - 16;
        return *reinterpret_cast<const uint4*>(gB + off);
    };

    v16f acc[MF_T][MF_T];
#pragma unroll
    for (int a = 0; a < MF_T; ++a)
#pragma unroll
        for (int b = 0; b < MF_T; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

#pragma unroll
    for (int t = 0; t < MF_T; ++t) {
        loadA(0, t);
        buildA(0, t);
    }
    uint4 vb = gload(0);
    *reinterpret_cast<uint4*>(rawc + wrB) = vb;
    vb = gload(1);
    __syncthreads();
    for (long AM = 0; AM < nAM; ++AM) {
        const int cur = (int)(AM & 1);
        const bool more = AM + 1 < nAM;                          // block-uniform
#pragma unroll
        for (int s = 0; s < MFF_KS; ++s) {
            if ((s & 1) == 0) {                                  // a new private B macro-step every two k-steps
                const long MB = AM * (MFF_KS / 2) + s / 2;
                const u32 nxt = (u32)((MB + 1) & 1) * BUF;
                *reinterpret_cast<uint4*>(rawc + nxt + wrB) = vb;
                vb = gload(MB + 2);
            }
            // one A tile operand of the next macro-step per step: built from bytes requested during the previous step
            if (s >= 1 && s <= MF_T && more) buildA(cur ^ 1, s - 1);
            if (s < MF_T && more) loadA(AM + 1, s);
            if (active) {
                const long MB = AM * (MFF_KS / 2) + s / 2;
                const char* bcur = rawc + (u32)(MB & 1) * BUF + 64 * (s & 1);
                v8i fa[MF_T], fb[MF_T];
#pragma unroll
                for (int t = 0; t < MF_T; ++t) {
                    const uint4 o = opA[cur][s][t][lane];
                    fa[t] = v8i{(int)o.x, (int)o.y, (int)o.z, (int)o.w, 0, 0, 0, 0};
                    const char* pb = bcur + rdB[t];
                    fb[t] = onehot32_fp4(*reinterpret_cast<const uint4*>(pb), *reinterpret_cast<const uint4*>(pb + 16), patB[t]);
                }
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int a = 0; a < MF_T; ++a)
#pragma unroll
                    for (int b = 0; b < MF_T; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa[a], fb[b], acc[a][b], 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                __builtin_amdgcn_s_setprio(0);
            }
        }
        __syncthreads();
    }
    if (!active) return;

    const long SS = (long)S * S;
#pragma unroll
    for (int ta = 0; ta < MF_T; ++ta)
#pragma unroll
        for (int tb = 0; tb < MF_T; ++tb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int v = (int)acc[ta][tb][r];
                if (!v) continue;
                const int m = (bm * MF_T + ta) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int n = (bn * MF_T + tb) * 32 + (lane & 31);
                if (m >= NS || n >= NS) continue;
                const int a = m / S, i = m - a * S, b = n / S, j = n - b * S;
                if (a == b) continue;
                atomicAdd(&counts[((long)a * N + b) * SS + i * S + j], v);
                if (bm != bn) atomicAdd(&counts[((long)b * N + a) * SS + j * S + i], v);
            }
}

int64_t s3_mfma_ws_bytes(int64_t R, int N) { return align_up((int64_t)N * align_up(R, 512) + 64, 256); }

// XT[sample][bin], bins padded to Rp (a multiple of 32), everything that is not a state in [0, S) stored as 31;
// bytes are stored shifted left by `shift` (the S3 score kernel wants 4 * state, a ready-made LDS byte offset)
// `bad` is the code stored for "not a state" (31 for the kernels of this file and k_s3_score; S for k_s3_score_bl, whose
// table rows have exactly one zero column after the S states)
// `dirty` (optional, device int, caller-zeroed): set to 1 when a byte of the first N columns of a row < R is not a state
int transpose_states_flag(const char* X, int64_t R, int32_t N, int64_t ldx, int32_t S, char* XT, int64_t Rp, int shift, int bad, int* dirty,
                          hipStream_t st) {
    // 16-byte stores need XT rows that start 16-byte aligned (Rp a multiple of 16, an aligned base): always so for the workspaces
    // this library lays out; anything else takes the byte-per-thread kernel
    if (Rp % 16 == 0 && (reinterpret_cast<uintptr_t>(XT) & 15) == 0)
        hipLaunchKernelGGL(k_transpose_states16, dim3((unsigned)(((Rp + 63) / 64) * ((N + 63) / 64))), dim3(256), 0, st, X, (long)R, N,
                           (long)ldx, S, XT, (long)Rp, shift, bad, dirty);
    else
        hipLaunchKernelGGL(k_transpose_states, dim3((unsigned)((Rp + 63) / 64), (unsigned)((N + 63) / 64)), dim3(256), 0, st, X, (long)R, N,
                           (long)ldx, S, XT, (long)Rp, shift, bad, dirty);
    EPG_LAUNCH_CHECK("k_transpose_states");
    return EPG_OK;
}
int transpose_states_bad(const char* X, int64_t R, int32_t N, int64_t ldx, int32_t S, char* XT, int64_t Rp, int shift, int bad, hipStream_t st) {
    return transpose_states_flag(X, R, N, ldx, S, XT, Rp, shift, bad, nullptr, st);
}
int transpose_states(const char* X, int64_t R, int32_t N, int64_t ldx, int32_t S, char* XT, int64_t Rp, int shift, hipStream_t st) {
    return transpose_states_bad(X, R, N, ldx, S, XT, Rp, shift, 31, st);
}

int hist_s3_mfma(const char* X, int64_t R, int32_t N, int64_t ldx, int32_t S, int32_t* counts, void* ws, hipStream_t st) {
    // EPG_S3_MFMA selects a kernel for A/B measurements: f (default) = fp4 MX with workgroup-shared A operands, e = fp4 MX
    // per wave, b = int8 with per-lane loads.  E / F need the 96 rows of a block to span at most 8 biosamples (S >= 14).
    static const char choice = [] { const char* e = exp_env("EPG_S3_MFMA"); return e && e[0] != 'g' ? e[0] : 'f'; }();
    const char variant = S >= 14 ? choice : 'b';
    const long Rp = align_up(R, 512);             // whole macro-steps; padded bins hold 31, which matches no row
    char* XT = reinterpret_cast<char*>(ws);
    int rc = transpose_states(X, R, N, ldx, S, XT, Rp, 0, st);
    if (rc) return rc;
    if (variant == 'f') {
        const int nblk = (N * S + MF_ROWS - 1) / MF_ROWS;
        long tasks = 0;
        for (int bm = 0; bm < nblk; ++bm) tasks += (nblk - bm + MFF_WAVES - 1) / MFF_WAVES;
        long splits = (8L * num_cus() + tasks - 1) / tasks;      // one workgroup per CU: eight rounds of tasks
        const long kmacros = Rp / 512;
        const long min_splits = (Rp + (1L << 24) - 513) / ((1L << 24) - 512);
        if (splits < min_splits) splits = min_splits;
        if (splits > kmacros) splits = kmacros;
        if (splits < 1) splits = 1;
        if (splits > 65535) splits = 65535;
        const long ksplit_len = ((kmacros + splits - 1) / splits) * 512;
        const long nsplit = (Rp + ksplit_len - 1) / ksplit_len;
        hipLaunchKernelGGL(k_s3_hist_mfma_f, dim3((unsigned)tasks, (unsigned)nsplit), dim3(64 * MFF_WAVES), 0, st, XT, Rp, ksplit_len, N, S,
                           nblk, counts);
        EPG_LAUNCH_CHECK("k_s3_hist_mfma_f");
        return EPG_OK;
    }
    const int nblk = (N * S + MF_ROWS - 1) / MF_ROWS;
    const long npairs = (long)nblk * (nblk + 1) / 2;
    // split K so that there are a few tasks per wave slot (2 waves per SIMD); int32 atomics combine the splits.  A
    // split stays below 2^24 bins so that the float32 accumulators of the fp4 kernel hold exact integers.
    const long slots = (long)num_cus() * 8;
    long splits = (4 * slots + npairs - 1) / npairs;
    const long ksteps = Rp / 64;
    const long min_splits = (Rp + (1L << 24) - 65) / ((1L << 24) - 64);
    if (splits < min_splits) splits = min_splits;
    if (splits > ksteps) splits = ksteps;
    if (splits < 1) splits = 1;
    if (splits > 65535) splits = 65535;
    const long ksplit_len = ((ksteps + splits - 1) / splits) * 64;
    const long nsplit = (Rp + ksplit_len - 1) / ksplit_len;
    const dim3 grid((unsigned)npairs, (unsigned)nsplit);
    if (variant == 'e') hipLaunchKernelGGL(k_s3_hist_mfma_e, grid, dim3(64), 0, st, XT, Rp, ksplit_len, N, S, nblk, counts);
    else hipLaunchKernelGGL(k_s3_hist_mfma_b, grid, dim3(64), 0, st, XT, Rp, ksplit_len, N, S, nblk, counts);
    EPG_LAUNCH_CHECK("k_s3_hist_mfma");
    return EPG_OK;
}

}  // namespace epg
