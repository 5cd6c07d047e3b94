// Paired-mode null groups (reference helpers.py:183-194): every row of the concatenation [A|B] is shuffled by an
// independent uniform permutation and cut into two groups (the first ga and the next gb shuffled columns).  Only
// the per-state counts of the two groups enter the scores, so the kernel draws the group membership directly:
// sequential selection sampling over the row's columns (column c joins group A with probability
// need_A / remaining, group B with need_B / remaining) is exactly the law of (first ga, next gb) of a uniform
// permutation.  Randomness: Philox4x32-10 keyed by the seed, counter = (global row, group, column block) -- the result
// depends only on (seed, row0 + row), never on the launch geometry or on which GPU owns the row.  gfx950 only.
#include "epg_count.h"

#include <string.h>

#include <stdlib.h>

namespace epg {

__device__ __forceinline__ void philox4x32_10(u32 (&c)[4], u32 k0, u32 k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const u64 p0 = (u64)0xD2511F53u * c[0];
        const u64 p1 = (u64)0xCD9E8D57u * c[2];
        const u32 n0 = (u32)(p1 >> 32) ^ c[1] ^ k0;
        const u32 n2 = (u32)(p0 >> 32) ^ c[3] ^ k1;
        c[0] = n0; c[1] = (u32)p1; c[2] = n2; c[3] = (u32)p0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

constexpr int NH_CH = 64;                 // columns per staged chunk
constexpr int NH_LD = 80;                 // bytes per staged row: 16-lane groups of a ds_read_b128 then cover all 64 banks

// One lane per row, a wave owns 64 rows.  The rows' state bytes are read coalesced (four lanes fetch 64 contiguous bytes
// of a row, sixteen rows per instruction, the next chunk requested while the current one is processed) and handed over
// through the wave's LDS slot, so that every byte is fetched once; a lane reading its own row directly touches 64
// different cache lines per instruction and was 10x slower.  The two group histograms of a row are the low and high
// halves of one uint32 counter per state, [state][lane] in LDS: one ds_add_u32 per column, no read-modify-write.
// Philox counter = (global row, source group, block of four columns): a pure function of (seed, global row).
__global__ __launch_bounds__(256) void k_null_hist(const char* __restrict__ XA, int NA, long ldxa, const char* __restrict__ XB,
                                                    int NB, long ldxb, long R, int S, int ga, int gb, u64 seed, long row0,
                                                    u16* __restrict__ HA, u16* __restrict__ HB) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u32* hist = reinterpret_cast<u32*>(smem);                                    // [S + 1][256], row S takes non-states
    char* stage = smem + (size_t)(S + 1) * 256 * 4;                              // [256][NH_LD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long wrow0 = (long)blockIdx.x * 256 + 64 * wave;                       // first row of this wave
    const long row = wrow0 + lane;
    for (int e = tid; e < (S + 1) * 256; e += 256) hist[e] = 0;
    __syncthreads();
    char* wstage = stage + (size_t)64 * wave * NH_LD;
    const u64 grow = (u64)(row0 + row);
    u32 needA = (u32)ga, needB = (u32)gb, rem = (u32)(NA + NB);
    // staging role of this lane: rows (lane >> 2) + 16 k of the wave, 16-byte piece lane & 3 of the 64-byte chunk
    const int piece = lane & 3, srow = lane >> 2;
    for (int g = 0; g < 2; ++g) {
        const char* X = g ? XB : XA;
        const int Ng = g ? NB : NA;
        const long ldx = g ? ldxb : ldxa;
        auto load_chunk = [&](int c0, uint4 (&v)[4]) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const long r = wrow0 + srow + 16 * k;
                const long c = c0 + 16 * piece;
                v[k] = make_uint4(~0u, ~0u, ~0u, ~0u);                          // 0xFF: not a state
                if (r < R && c < Ng) {
                    const char* src = X + r * ldx + c;
                    if (c + 16 <= ldx) v[k] = *reinterpret_cast<const uint4*>(src);   // whole piece inside the row's pitch
                    else {
                        unsigned char t[16];
                        for (int q = 0; q < 16; ++q) t[q] = c + q < Ng ? (unsigned char)src[q] : 0xFF;
                        v[k] = make_uint4(t[0] | t[1] << 8 | t[2] << 16 | (u32)t[3] << 24, t[4] | t[5] << 8 | t[6] << 16 | (u32)t[7] << 24,
                                          t[8] | t[9] << 8 | t[10] << 16 | (u32)t[11] << 24, t[12] | t[13] << 8 | t[14] << 16 | (u32)t[15] << 24);
                    }
                }
            }
        };
        uint4 pre[4];
        load_chunk(0, pre);
        for (int c0 = 0; c0 < Ng; c0 += NH_CH) {
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < 4; ++k) *reinterpret_cast<uint4*>(wstage + (srow + 16 * k) * NH_LD + 16 * piece) = pre[k];
            __builtin_amdgcn_wave_barrier();
            if (c0 + NH_CH < Ng) load_chunk(c0 + NH_CH, pre);                   // next chunk: in flight during this one
            uint4 mine[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) mine[k] = *reinterpret_cast<const uint4*>(wstage + lane * NH_LD + 16 * k);
            const u32 w[16] = {mine[0].x, mine[0].y, mine[0].z, mine[0].w, mine[1].x, mine[1].y, mine[1].z, mine[1].w,
                               mine[2].x, mine[2].y, mine[2].z, mine[2].w, mine[3].x, mine[3].y, mine[3].z, mine[3].w};
            if (row < R && (needA | needB)) {
#pragma unroll
                for (int b4 = 0; b4 < 16; ++b4) {
                    const int cb = c0 + 4 * b4;
                    if (cb >= Ng) break;                                         // wave-uniform
                    u32 ctr[4] = {(u32)grow, (u32)(grow >> 32), (u32)(cb >> 2), (u32)g};
                    philox4x32_10(ctr, (u32)seed, (u32)(seed >> 32));
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if (cb + k >= Ng) break;                                 // wave-uniform
                        const u32 pick = (u32)(((u64)ctr[k] * rem) >> 32);      // uniform in [0, rem)
                        u32 x = (w[b4] >> (8 * k)) & 0xffu;
                        x = x < (u32)S ? x : (u32)S;
                        const bool inA = pick < needA, inB = !inA && pick < needA + needB;
                        const u32 val = inA ? 1u : (inB ? 0x10000u : 0u);
                        atomicAdd(&hist[x * 256 + tid], val);                    // own column of the counter matrix: no contention
                        needA -= inA;
                        needB -= inB;
                        --rem;
                    }
                }
            } else {
                rem -= (u32)((Ng - c0) < NH_CH ? (Ng - c0) : NH_CH);
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    // rows of this wave back to [row][state], coalesced
    for (int e = lane; e < 64 * S; e += 64) {
        const int r = e / S, st = e - r * S;
        if (wrow0 + r < R) {
            const u32 v = hist[st * 256 + 64 * wave + r];
            HA[(wrow0 + r) * S + st] = (u16)(v & 0xffffu);
            HB[(wrow0 + r) * S + st] = (u16)(v >> 16);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The same law from the two REAL groups' per-bin histograms (round 2).  Only the per-state counts of the shuffled groups are
// needed, and they depend on the row only through its combined histogram h = hA + hB: the counts of the first ga and the next
// gb columns of a uniform permutation of a row with h[s] columns of state s are MULTIVARIATE HYPERGEOMETRIC.  Sampled exactly,
// category by category, by the same selection sampling as above (a column joins A with probability need_A / remaining, B with
// need_B / remaining), except that the LAST category needs no draws at all: whatever is still needed comes from it.  The
// row's most frequent state goes last, so a row costs n - max_s h[s] uniform numbers instead of n (71 % of real cells are
// one state; i.i.d. synthetic rows at the chr1 frequencies: 209 of 721), and the kernel reads 72 bytes per bin instead of 721:
// no state matrix, no LDS hand-over of rows.  Columns that hold no state (n - sum h) are one more category that is drawn
// but not reported.  Philox4x32-10, counter = (global row, block of four draws, tag): a pure function of (seed, global row,
// the row's histograms) -- not of the launch geometry or of the GPU count.  A lane owns a row; rows are handed over through
// LDS both ways so that loads and stores cover whole lines.
// ---------------------------------------------------------------------------------------------------------------
// Several parts (the chromosome files of a genome) in ONE launch: pointers, row counts and shuffle keys travel in the kernel
// argument, tiles are numbered through the parts in order and never straddle two parts; a wave's tile index ascends, so its
// part only moves forward (the pattern of k_pair_fused_s1 and k_bin_hist_parts).  A row's draws depend on (seed, key + row in
// the part, the row's histograms) only -- the same numbers as a launch per part.
constexpr int NH_MAXP = 48;
struct NhParts {
    const u16* ha[NH_MAXP];
    const u16* hb[NH_MAXP];
    u16* oa[NH_MAXP];
    u16* ob[NH_MAXP];
    long rows[NH_MAXP];
    long key[NH_MAXP];                     // row0 of the part: the shuffle key of its first row
    long t0[NH_MAXP + 1];                  // first tile (TR rows) of every part, and their total
    int n;
};

#define NH_PART_STATE                                                                                                   \
    const long ntiles = pt.t0[pt.n];                                                                                    \
    int part = -1;                                                                                                      \
    long next = 0, base = 0, R = 0, row0 = 0;                                                                           \
    const u16* __restrict__ HA = nullptr;                                                                               \
    const u16* __restrict__ HB = nullptr;                                                                               \
    u16* __restrict__ OA = nullptr;                                                                                     \
    u16* __restrict__ OB = nullptr;
#define NH_PART_ENTER                                                                                                   \
    if (tile >= next) {                                                                                                 \
        do { ++part; next = pt.t0[part + 1]; } while (tile >= next);                                                    \
        HA = pt.ha[part]; HB = pt.hb[part]; OA = pt.oa[part]; OB = pt.ob[part];                                         \
        R = pt.rows[part]; row0 = pt.key[part]; base = pt.t0[part];                                                     \
    }                                                                                                                   \
    const long r0 = (tile - base) * TR;

__device__ __forceinline__ void nh_stage_in(char* lds, const char* src, int nbytes, int lane) {
    const int nchunks = nbytes >> 4;
    for (int c = lane; c < nchunks; c += 64) *reinterpret_cast<uint4*>(lds + 16 * c) = *reinterpret_cast<const uint4*>(src + 16 * c);
    for (int o = (nchunks << 4) + 2 * lane; o + 2 <= nbytes; o += 128) *reinterpret_cast<u16*>(lds + o) = *reinterpret_cast<const u16*>(src + o);
}

__global__ __launch_bounds__(256) void k_null_hist_h_seq(const NhParts pt, int S, int n_cols, int ga, int gb, u64 seed, int TR) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rowb = 2 * S;
    char* sa = smem + (size_t)wave * 2 * TR * rowb;               // the wave's TR (64) rows of hA, later of the A group's counts
    char* sb = sa + TR * rowb;
    __builtin_amdgcn_s_setprio(3);                                // (see k_null_hist_h)
    NH_PART_STATE
    for (long tile = (long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long)gridDim.x * 4) {
        NH_PART_ENTER
        const int rows = (int)(R - r0 < TR ? R - r0 : TR);
        nh_stage_in(sa, reinterpret_cast<const char*>(HA + r0 * S), rows * rowb, lane);
        nh_stage_in(sb, reinterpret_cast<const char*>(HB + r0 * S), rows * rowb, lane);
        __builtin_amdgcn_wave_barrier();
        if (lane < rows) {
            u16* pa = reinterpret_cast<u16*>(sa + lane * rowb);
            u16* pb = reinterpret_cast<u16*>(sb + lane * rowb);
            // combined counts into pa (pb becomes the zeroed output row), the most frequent state (first of equals) and the
            // number of columns that hold a state
            u32 tot = 0, best = 0;
            int modal = 0;
            for (int s = 0; s < S; ++s) {
                const u32 h = (u32)pa[s] + pb[s];
                pa[s] = (u16)h;
                pb[s] = 0;
                tot += h;
                if (h > best) { best = h; modal = s; }
            }
            // ONE flat loop over the columns that are drawn -- every state but the modal one, then the columns without a state
            // -- so that the 64 rows of a wave run n - max h steps each (209 +- 12 on chr1-like rows) instead of the sum over
            // the categories of the wave's largest count.  A draw takes SIXTEEN BITS of Philox output (round 4; a byte before):
            // with u in [v, v + 1) / 65536 the column joins A if (v + 1) rem <= 65536 needA, B if v rem >= 65536 needA and
            // (v + 1) rem <= 65536 (needA + needB), neither if v rem >= 65536 (needA + needB); in the remaining ~2 of 65536 cases
            // 32 more bits from a second stream decide -- the outcome is that of a 48-bit uniform number, 8 draws per Philox call.
            u32 A256 = (u32)ga << 16, AB256 = (u32)(ga + gb) << 16, rem = (u32)n_cols;
            const u32 ndraw = (u32)n_cols - best;
            const u64 grow = (u64)(row0 + r0 + lane);
            u32 r0w = 0, r1w = 0, r2w = 0, r3w = 0, cur = 0, nb = 0, calls = 0;          // main stream: 16 bytes per call
            u32 a0w = 0, a1w = 0, a2w = 0, a3w = 0, ahave = 0, acalls = 0;               // second stream for the rare ties
            int s = -1;
            u32 left = 0, inA = 0, inB = 0;
            for (u32 d = 0; d < ndraw; ++d) {
                if (AB256 == 0) break;                                                    // both groups are full: the rest joins neither
                while (left == 0) {                                                       // next non-empty category
                    if (s >= 0 && s < S) { pa[s] = (u16)inA; pb[s] = (u16)inB; }
                    ++s;
                    if (s == modal) ++s;
                    left = s < S ? (u32)pa[s] : (u32)n_cols - tot;
                    inA = 0; inB = 0;
                }
                if ((nb & 1u) == 0) {
                    if (nb == 0) {
                        u32 c[4] = {(u32)grow, (u32)(grow >> 32), calls++, 0x6e756c6cu};
                        philox4x32_10(c, (u32)seed, (u32)(seed >> 32));
                        r0w = c[0]; r1w = c[1]; r2w = c[2]; r3w = c[3];
                        nb = 8;
                    }
                    cur = r0w; r0w = r1w; r1w = r2w; r2w = r3w;
                }
                const u32 byte = cur & 0xffffu;                                           // (sixteen bits)
                cur >>= 16;
                --nb;
                const u32 t = byte * rem, hi = t + rem;
                bool a = hi <= A256;
                bool b = t >= A256 && hi <= AB256;
                if (!(a || b || t >= AB256)) {                                            // the byte's interval straddles a boundary
                    if (ahave == 0) {
                        u32 c[4] = {(u32)grow, (u32)(grow >> 32), acalls++, 0x74696573u};
                        philox4x32_10(c, (u32)seed, (u32)(seed >> 32));
                        a0w = c[0]; a1w = c[1]; a2w = c[2]; a3w = c[3];
                        ahave = 4;
                    }
                    const u64 u48 = ((u64)byte << 32) | a0w;
                    a0w = a1w; a1w = a2w; a2w = a3w;
                    --ahave;
                    const u32 pick = (u32)((u48 * rem) >> 48);                            // uniform in [0, rem)
                    a = pick < (A256 >> 16);
                    b = !a && pick < (AB256 >> 16);
                }
                inA += a; inB += b;
                A256 -= a ? 65536u : 0u;
                AB256 -= (a || b) ? 65536u : 0u;
                --rem;
                --left;
            }
            // the category the loop stopped in, the ones it never reached (nothing joins A or B any more), and the modal
            // state, which takes what is still missing
            if (s >= 0 && s < S) { pa[s] = (u16)inA; pb[s] = (u16)inB; }
            for (int z = s + 1; z < S; ++z)
                if (z != modal) pa[z] = 0;
            pa[modal] = (u16)(A256 >> 16);
            pb[modal] = (u16)((AB256 - A256) >> 16);
        }
        __builtin_amdgcn_wave_barrier();
        store_staged(sa, reinterpret_cast<char*>(OA + r0 * S), rows * rowb, lane);
        store_staged(sb, reinterpret_cast<char*>(OB + r0 * S), rows * rowb, lane);
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Round 3: the same sampler with the categories taken OUT of the draw loop.  Selection sampling does not look at what a
// column holds -- the column at position d joins A with probability need_A / remaining whatever its state -- so with the
// columns of a row laid out [non-modal states in order | columns without a state | modal state] the loop over the first
// m = n - max h positions is the same for every row: one byte of Philox output, a multiply-add, two compares, and ONE BIT of
// outcome per group appended to a per-lane bit string in LDS ([word][lane], flushed every 32 draws by all lanes at once).
// The per-state counts are then range popcounts of that string between the prefix sums of the row's histogram.  The category
// bookkeeping of k_null_hist_h_seq (an LDS store + load and a divergent inner loop whenever ANY lane of the wave crosses into
// its next state, i.e. at nearly every draw) is gone from the loop; the law, the Philox counters (global row, call number,
// tag) and the tie rule are the same, which draw lands in which state is not: same seed, different -- equally distributed --
// null groups.  FULL (ga + gb == n, the command line without -g): a column that does not join A joins B, one bit string.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 nh_range_pop(const u32* __restrict__ bits, u32 o, u32 h) {    // ones in positions [o, o + h)
    u32 cnt = 0, p = o;
    const u32 e = o + h;
    while (p < e) {
        const u32 w = bits[(p >> 5) * 64], lo = p & 31u;
        const u32 n = (32u - lo) < (e - p) ? (32u - lo) : (e - p);                           // 1 .. 32
        cnt += __popc((w >> lo) << (32u - n));                                                 // the n bits from lo on
        p += n;
    }
    return cnt;
}

// One row of the bit-string sampler (ga + gb == n_cols): pa / pb = the row's histograms of the two real groups in LDS on entry,
// the two null groups' on return; bitsA = the lane's outcome-bit words [word * 64]; grow = the row's shuffle key.  Shared by
// k_null_hist_h and the fused count + sample kernel of paired mode (k_pair_count_null).
__device__ __forceinline__ void nh_sample_row_full(u16* __restrict__ pa, u16* __restrict__ pb, u32* __restrict__ bitsA, int S, int n_cols,
                                                   int ga, u64 seed, u64 grow) {
    u32 best = 0;
    int modal = 0;
    for (int s = 0; s < S; ++s) {
        const u32 h = (u32)pa[s] + pb[s];
        pa[s] = (u16)h;
        if (h > best) { best = h; modal = s; }
    }
    int Am1 = (int)(((u32)ga << 16) - 1u);                                        // (need_A << 16) - 1
    const u32 m = (u32)n_cols - best;
    
    u32 calls = 0;
    u32 a0w = 0, a1w = 0, a2w = 0, a3w = 0, ahave = 0, acalls = 0;
    u32 wA = 0;
    // the careful draw at position d: certain unless the 16-bit interval straddles need_A / rem, then 32 more bits decide
    auto draw = [&](u32 v, u32 d) {
        const int rem = n_cols - (int)d;
        const int x = Am1 - (int)__umul24(v, (u32)rem);                           // (need_A << 16) - 1 - v rem
        bool a = x >= rem - 1;                                                    // (v + 1) rem <= need_A << 16
        if (!a && x >= 0) {                                                       // v rem < need_A << 16 < (v + 1) rem
            if (ahave == 0) {
                u32 c[4] = {(u32)grow, (u32)(grow >> 32), acalls++, 0x74696573u};
                philox4x32_10(c, (u32)seed, (u32)(seed >> 32));
                a0w = c[0]; a1w = c[1]; a2w = c[2]; a3w = c[3];
                ahave = 4;
            }
            const u64 u48 = ((u64)v << 32) | a0w;
            a0w = a1w; a1w = a2w; a2w = a3w;
            --ahave;
            const u32 pick = (u32)((u48 * (u32)rem) >> 48);                       // uniform in [0, rem)
            a = pick < (((u32)Am1 + 1u) >> 16);
        }
        wA |= a ? 1u << (d & 31u) : 0u;
        Am1 -= a ? 65536 : 0;
    };
    u32 d = 0;
    for (; d + 8 <= m; d += 8) {
        u32 c[4] = {(u32)grow, (u32)(grow >> 32), calls++, 0x6e756c6cu};
        philox4x32_10(c, (u32)seed, (u32)(seed >> 32));
        const int Am1_0 = Am1;
        const u32 wA_0 = wA;
        const int rem0 = __builtin_amdgcn_readfirstlane(n_cols - (int)d);         // every lane of the wave is at the same d
        const int bit0 = __builtin_amdgcn_readfirstlane((int)(d & 31u));
        u32 low = 0xffffffffu;
        // two draws per word of Philox output, 13 instructions: v_and / v_lshrrev (the two halves), then per draw
        // v_mad_i32_i24 (x), v_cmp_le_i32 (a), v_cndmask (a as 0 / 1), v_lshl_or (the outcome bit), v_mad_i32_i24 (need_A),
        // and one v_min3_u32 for the tie test
#define NH_PAIR(W, K)                                                                                                             \
        {                                                                                                                 \
            u32 v0_, v1_, x0_, x1_, a_;                                                                                   \
            asm volatile(                                                                                                 \
                "v_and_b32_e32 %[v0], 0xffff, %[w]\n\t"                                                                   \
                "v_lshrrev_b32_e32 %[v1], 16, %[w]\n\t"                                                                   \
                "v_mad_i32_i24 %[x0], %[v0], %[nr0], %[am]\n\t"                                                           \
                "v_cmp_le_i32_e32 vcc, %[rm0], %[x0]\n\t"                                                                 \
                "v_cndmask_b32_e64 %[a], 0, 1, vcc\n\t"                                                                   \
                "v_lshl_or_b32 %[wa], %[a], %[b0], %[wa]\n\t"                                                             \
                "v_mad_i32_i24 %[am], %[a], %[m64k], %[am]\n\t"                                                           \
                "v_mad_i32_i24 %[x1], %[v1], %[nr1], %[am]\n\t"                                                           \
                "v_cmp_le_i32_e32 vcc, %[rm1], %[x1]\n\t"                                                                 \
                "v_cndmask_b32_e64 %[a], 0, 1, vcc\n\t"                                                                   \
                "v_lshl_or_b32 %[wa], %[a], %[b1], %[wa]\n\t"                                                             \
                "v_mad_i32_i24 %[am], %[a], %[m64k], %[am]\n\t"                                                           \
                "v_min3_u32 %[lo], %[x0], %[x1], %[lo]"                                                                   \
                : [v0] "=&v"(v0_), [v1] "=&v"(v1_), [x0] "=&v"(x0_), [x1] "=&v"(x1_), [a] "=&v"(a_), [wa] "+v"(wA),       \
                  [am] "+v"(Am1), [lo] "+v"(low)                                                                          \
                : [w] "v"(W), [nr0] "s"((K) - rem0), [nr1] "s"((K) + 1 - rem0), [rm0] "s"(rem0 - (K) - 1),                \
                  [rm1] "s"(rem0 - (K) - 2), [b0] "s"(bit0 + (K)), [b1] "s"(bit0 + (K) + 1), [m64k] "s"(-65536)           \
                : "vcc");                                                                                                 \
        }
        NH_PAIR(c[0], 0)
        NH_PAIR(c[1], 2)
        NH_PAIR(c[2], 4)
        NH_PAIR(c[3], 6)
#undef NH_PAIR
        if (low < (u32)rem0) {                                                    // a tie is possible in this call: repeat it carefully
            Am1 = Am1_0;
            wA = wA_0;
#pragma unroll
            for (int k = 0; k < 8; ++k) draw((k & 1) ? c[k >> 1] >> 16 : c[k >> 1] & 0xffffu, d + k);
        }
        if ((d & 31u) == 24u) {
            bitsA[(d >> 5) * 64] = wA;
            wA = 0;
        }
    }
    if (d < m) {                                                                  // the last one to seven draws
        u32 c[4] = {(u32)grow, (u32)(grow >> 32), calls++, 0x6e756c6cu};
        philox4x32_10(c, (u32)seed, (u32)(seed >> 32));
#pragma unroll
        for (int k = 0; k < 7; ++k)
            if (d + k < m) draw((k & 1) ? c[k >> 1] >> 16 : c[k >> 1] & 0xffffu, d + k);
    }
    if (m & 31u) bitsA[(m >> 5) * 64] = wA;
    // counts per state: range popcounts between the prefix sums of the histogram; the modal state takes what is missing
    u32 o = 0;
    for (int s = 0; s < S; ++s) {
        const u32 h = s == modal ? 0u : (u32)pa[s];
        const u32 ca = nh_range_pop(bitsA, o, h);
        pa[s] = (u16)ca;
        pb[s] = (u16)(h - ca);
        o += h;
    }
    // positions o .. m - 1 are the columns without a state: drawn (they take places in the groups), not reported
    const u32 needA = ((u32)Am1 + 1u) >> 16;                                      // = ga - (A members among the m drawn)
    pa[modal] = (u16)needA;
    pb[modal] = (u16)(best - needA);
}

// Round 4: the draw loop again.  Where round 3's 3.15 ms went: ~14 instructions per draw on the main path, Philox 4 (64 per
// call -- the 32 x 32 multiplies are full rate on gfx950, tools/ubench/rng_rate.hip --, 16 byte-sized draws per call) and ~7 for
// the tie path, which a wave walked whenever ONE of its 64 lanes' bytes straddled need_A / remaining: 22 % of the draws.  Now
//  * a draw takes 16 bits (8 per Philox call: +4 instructions per draw), so a lane ties once in ~65536 draws instead of 256;
//  * the eight draws of a call run SPECULATIVELY with six instructions each -- x = (need_A << 16) - 1 - v * rem by one
//    v_mad_i32_i24 (rem = n - d is scalar), a = x >= rem - 1 (signed), the outcome bit into the string by v_lshl_or, need_A by
//    another v_mad_i32_i24 -- and ONE test per call, umin over the eight (u32)x < rem, catches every possible tie (a tie is
//    0 <= x < rem - 1; the test is conservative); if any lane of the wave fails it (0.8 % of the calls) the wave repeats the
//    call's eight draws from the saved state on the careful path, which resolves ties with 32 more bits like k_null_hist_h_seq.
// Same law, same Philox counters and tie rule as k_null_hist_h_seq (identical outputs, tests/test_hip_s3_null.py); the
// arithmetic is signed 32-bit, hence rows of at most 32767 columns here (the bit strings limit the kernel to 3072 anyway).
// Only the case the command line has without -g: ga + gb == n (a column that does not join A joins B, one bit string).
__global__ __launch_bounds__(256) void k_null_hist_h(const NhParts pt, int S, int n_cols, int ga, int gb, u64 seed, int TR, int NW) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // The sampler is drawn on a second stream UNDER the count pass of the next batch of parts (backend._HipPairedSession): the two
    // share every SIMD, the sampler needs ~80 % of the VALU slots and the count pass ~25 %.  At equal priority the count pass's
    // older waves win the issue arbitration and the sampler, the longer of the two, takes 3.4 ms instead of 2.1 for 15 M bins;
    // with its waves at priority 3 it runs at its own speed and the count pass takes what is left (3.2 instead of 2.5 ms):
    // tools/overlap_probe.py, profiles/r05g_paired_overlap.txt.  Alone on the chip the priority changes nothing.
    __builtin_amdgcn_s_setprio(3);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rowb = 2 * S;
    const size_t per_wave = (size_t)2 * TR * rowb + (size_t)NW * 256;
    char* sa = smem + (size_t)wave * per_wave;                    // the wave's TR rows of hA, later of the A group's counts
    char* sb = sa + TR * rowb;
    u32* bitsA = reinterpret_cast<u32*>(sb + TR * rowb) + lane;   // [word][lane]
    NH_PART_STATE
    for (long tile = (long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long)gridDim.x * 4) {
        NH_PART_ENTER
        const int rows = (int)(R - r0 < TR ? R - r0 : TR);
        nh_stage_in(sa, reinterpret_cast<const char*>(HA + r0 * S), rows * rowb, lane);
        nh_stage_in(sb, reinterpret_cast<const char*>(HB + r0 * S), rows * rowb, lane);
        __builtin_amdgcn_wave_barrier();
        if (lane < rows) {
            u16* pa = reinterpret_cast<u16*>(sa + lane * rowb);
            u16* pb = reinterpret_cast<u16*>(sb + lane * rowb);
            nh_sample_row_full(pa, pb, bitsA, S, n_cols, ga, seed, (u64)(row0 + r0 + lane));
        }
        __builtin_amdgcn_wave_barrier();
        store_staged(sa, reinterpret_cast<char*>(OA + r0 * S), rows * rowb, lane);
        store_staged(sb, reinterpret_cast<char*>(OB + r0 * S), rows * rowb, lane);
        __builtin_amdgcn_wave_barrier();
    }
}

// The same kernel for group sizes that do NOT fill the row (-g: ga + gb < n): a column joins A, B or neither, so there are two
// thresholds ((need_A << 16) - 1 and ((need_A + need_B) << 16) - 1), two outcome bits per position and two bit strings; a draw is
// 13 instructions on the speculative path (twice `v_mad_i32_i24` x, `v_cmp_le_i32`, `v_cndmask`, `v_mad_i32_i24` need; one `v_sub`
// for "B = joined but not A"; two `v_lshl_or`; a share of `v_min3_u32`), a tie at either threshold sends the lane's call through
// the careful path.  Same draws and tie rule as k_null_hist_h_seq: identical outputs.  Round 3 had tried this shape with byte-sized
// draws and found it 5 % slower than the column-by-column kernel; with 16-bit draws: 4.7 -> 3.2 ms per 15 M bins (-g 100).
__global__ __launch_bounds__(256) void k_null_hist_h2(const NhParts pt, int S, int n_cols, int ga, int gb, u64 seed, int TR, int NW) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rowb = 2 * S;
    const size_t per_wave = (size_t)2 * TR * rowb + (size_t)2 * NW * 256;
    char* sa = smem + (size_t)wave * per_wave;
    char* sb = sa + TR * rowb;
    u32* bitsA = reinterpret_cast<u32*>(sb + TR * rowb) + lane;   // [word][lane]
    u32* bitsB = bitsA + NW * 64;
    __builtin_amdgcn_s_setprio(3);                                // (see k_null_hist_h)
    NH_PART_STATE
    for (long tile = (long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long)gridDim.x * 4) {
        NH_PART_ENTER
        const int rows = (int)(R - r0 < TR ? R - r0 : TR);
        nh_stage_in(sa, reinterpret_cast<const char*>(HA + r0 * S), rows * rowb, lane);
        nh_stage_in(sb, reinterpret_cast<const char*>(HB + r0 * S), rows * rowb, lane);
        __builtin_amdgcn_wave_barrier();
        if (lane < rows) {
            u16* pa = reinterpret_cast<u16*>(sa + lane * rowb);
            u16* pb = reinterpret_cast<u16*>(sb + lane * rowb);
            u32 best = 0;
            int modal = 0;
            for (int s = 0; s < S; ++s) {
                const u32 h = (u32)pa[s] + pb[s];
                pa[s] = (u16)h;
                if (h > best) { best = h; modal = s; }
            }
            int Am1 = (int)(((u32)ga << 16) - 1u), ABm1 = (int)(((u32)(ga + gb) << 16) - 1u);
            const u32 m = (u32)n_cols - best;
            const u64 grow = (u64)(row0 + r0 + lane);
            u32 calls = 0;
            u32 a0w = 0, a1w = 0, a2w = 0, a3w = 0, ahave = 0, acalls = 0;
            u32 wA = 0, wB = 0;
            auto draw = [&](u32 v, u32 d) {                                               // the careful draw at position d
                const int rem = n_cols - (int)d;
                const int t = (int)__umul24(v, (u32)rem);
                const int xA = Am1 - t, xAB = ABm1 - t;
                bool a = xA >= rem - 1;                                                   // certainly A
                bool b = xA < 0 && xAB >= rem - 1;                                        // certainly not A, certainly A or B
                if ((!a && xA >= 0) || (xAB < rem - 1 && xAB >= 0)) {                     // the interval straddles a threshold
                    if (ahave == 0) {
                        u32 c[4] = {(u32)grow, (u32)(grow >> 32), acalls++, 0x74696573u};
                        philox4x32_10(c, (u32)seed, (u32)(seed >> 32));
                        a0w = c[0]; a1w = c[1]; a2w = c[2]; a3w = c[3];
                        ahave = 4;
                    }
                    const u64 u48 = ((u64)v << 32) | a0w;
                    a0w = a1w; a1w = a2w; a2w = a3w;
                    --ahave;
                    const u32 pick = (u32)((u48 * (u32)rem) >> 48);                       // uniform in [0, rem)
                    a = pick < (((u32)Am1 + 1u) >> 16);
                    b = !a && pick < (((u32)ABm1 + 1u) >> 16);
                }
                const u32 bit = 1u << (d & 31u);
                wA |= a ? bit : 0u;
                wB |= b ? bit : 0u;
                Am1 -= a ? 65536 : 0;
                ABm1 -= (a || b) ? 65536 : 0;
            };
            u32 d = 0;
            for (; d + 8 <= m; d += 8) {
                u32 c[4] = {(u32)grow, (u32)(grow >> 32), calls++, 0x6e756c6cu};
                philox4x32_10(c, (u32)seed, (u32)(seed >> 32));
                const int Am1_0 = Am1, ABm1_0 = ABm1;
                const u32 wA_0 = wA, wB_0 = wB;
                const int rem0 = __builtin_amdgcn_readfirstlane(n_cols - (int)d);
                const int bit0 = __builtin_amdgcn_readfirstlane((int)(d & 31u));
                u32 low = 0xffffffffu;
#define NH_DRAW2(V, K)                                                                                                            \
                {                                                                                                                 \
                    u32 x0_, x1_, a_, ab_;                                                                                        \
                    asm volatile(                                                                                                 \
                        "v_mad_i32_i24 %[x0], %[v], %[nr], %[am]\n\t"                                                             \
                        "v_mad_i32_i24 %[x1], %[v], %[nr], %[abm]\n\t"                                                            \
                        "v_cmp_le_i32_e32 vcc, %[rm], %[x0]\n\t"                                                                  \
                        "v_cndmask_b32_e64 %[a], 0, 1, vcc\n\t"                                                                   \
                        "v_cmp_le_i32_e32 vcc, %[rm], %[x1]\n\t"                                                                  \
                        "v_cndmask_b32_e64 %[ab], 0, 1, vcc\n\t"                                                                  \
                        "v_lshl_or_b32 %[wa], %[a], %[b], %[wa]\n\t"                                                              \
                        "v_mad_i32_i24 %[am], %[a], %[m64k], %[am]\n\t"                                                           \
                        "v_mad_i32_i24 %[abm], %[ab], %[m64k], %[abm]\n\t"                                                        \
                        "v_sub_u32_e32 %[ab], %[ab], %[a]\n\t"                                                                    \
                        "v_lshl_or_b32 %[wb], %[ab], %[b], %[wb]\n\t"                                                             \
                        "v_min3_u32 %[lo], %[x0], %[x1], %[lo]"                                                                   \
                        : [x0] "=&v"(x0_), [x1] "=&v"(x1_), [a] "=&v"(a_), [ab] "=&v"(ab_), [wa] "+v"(wA), [wb] "+v"(wB),         \
                          [am] "+v"(Am1), [abm] "+v"(ABm1), [lo] "+v"(low)                                                        \
                        : [v] "v"(V), [nr] "s"((K) - rem0), [rm] "s"(rem0 - (K) - 1), [b] "s"(bit0 + (K)), [m64k] "s"(-65536)     \
                        : "vcc");                                                                                                 \
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const u32 v = (k & 1) ? c[k >> 1] >> 16 : c[k >> 1] & 0xffffu;
                    NH_DRAW2(v, k)
                }
#undef NH_DRAW2
                if (low < (u32)rem0) {                                                    // a tie is possible in this call: repeat it carefully
                    Am1 = Am1_0; ABm1 = ABm1_0;
                    wA = wA_0; wB = wB_0;
#pragma unroll
                    for (int k = 0; k < 8; ++k) draw((k & 1) ? c[k >> 1] >> 16 : c[k >> 1] & 0xffffu, d + k);
                }
                if ((d & 31u) == 24u) {
                    bitsA[(d >> 5) * 64] = wA;
                    bitsB[(d >> 5) * 64] = wB;
                    wA = 0;
                    wB = 0;
                }
            }
            if (d < m) {                                                                  // the last one to seven draws
                u32 c[4] = {(u32)grow, (u32)(grow >> 32), calls++, 0x6e756c6cu};
                philox4x32_10(c, (u32)seed, (u32)(seed >> 32));
#pragma unroll
                for (int k = 0; k < 7; ++k)
                    if (d + k < m) draw((k & 1) ? c[k >> 1] >> 16 : c[k >> 1] & 0xffffu, d + k);
            }
            if (m & 31u) {
                bitsA[(m >> 5) * 64] = wA;
                bitsB[(m >> 5) * 64] = wB;
            }
            u32 o = 0;
            for (int s = 0; s < S; ++s) {
                const u32 h = s == modal ? 0u : (u32)pa[s];
                pa[s] = (u16)nh_range_pop(bitsA, o, h);
                pb[s] = (u16)nh_range_pop(bitsB, o, h);
                o += h;
            }
            // positions o .. m - 1 are the columns without a state: drawn (they take places in the groups), not reported
            const u32 needA = ((u32)Am1 + 1u) >> 16, needAB = ((u32)ABm1 + 1u) >> 16;
            pa[modal] = (u16)needA;
            pb[modal] = (u16)(needAB - needA);
        }
        __builtin_amdgcn_wave_barrier();
        store_staged(sa, reinterpret_cast<char*>(OA + r0 * S), rows * rowb, lane);
        store_staged(sb, reinterpret_cast<char*>(OB + r0 * S), rows * rowb, lane);
        __builtin_amdgcn_wave_barrier();
    }
}

int null_hist_parts_impl(int32_t nparts, const uint16_t* const* HA, const uint16_t* const* HB, const int64_t* R, int32_t S, int32_t n_cols,
                         int32_t ga, int32_t gb, uint64_t seed, const int64_t* row0, uint16_t* const* OA, uint16_t* const* OB, hipStream_t st) {
    if (nparts < 0 || S < 1 || S > 127 || n_cols < 1 || n_cols > 65535) return fail(EPG_ERR_INVALID_ARG, "null_hist_from_binhist: bad shape");
    if (ga < 0 || gb < 0 || (long)ga + gb > n_cols)
        return fail(EPG_ERR_INVALID_ARG, "null_hist_from_binhist: group sizes %d + %d exceed the %d columns", ga, gb, n_cols);
    if (nparts && (!HA || !HB || !R || !row0 || !OA || !OB)) return fail(EPG_ERR_INVALID_ARG, "null_hist_from_binhist: NULL argument array");
    for (int p = 0; p < nparts; ++p) {
        if (R[p] < 0) return fail(EPG_ERR_INVALID_ARG, "null_hist_from_binhist: bad shape");
        if (R[p] == 0) continue;
        if (!HA[p] || !HB[p] || !OA[p] || !OB[p]) return fail(EPG_ERR_INVALID_ARG, "null_hist_from_binhist: NULL argument");
        if ((reinterpret_cast<uintptr_t>(HA[p]) | reinterpret_cast<uintptr_t>(HB[p]) | reinterpret_cast<uintptr_t>(OA[p]) | reinterpret_cast<uintptr_t>(OB[p])) & 15)
            return fail(EPG_ERR_INVALID_ARG, "null_hist_from_binhist: histogram arrays must be 16-byte aligned");
    }
    const int TR = tile_rows(2 * 2 * S);
    // the bit-string kernel while a lane's string fits the wave's share of LDS and the two groups fill the row (the command line
    // without -g); two bit strings with -g; otherwise the column-by-column kernel -- same draws, same outputs
    const bool full = ga + gb == n_cols;
    const int NW = (n_cols + 31) / 32;
    const size_t bits_bytes = (size_t)NW * 256;
    for (int p0 = 0; p0 < nparts;) {
        NhParts pt;
        memset(&pt, 0, sizeof(pt));
        long tiles = 0;
        int p = p0;
        for (; p < nparts && pt.n < NH_MAXP; ++p) {
            if (R[p] == 0) continue;
            const int k = pt.n++;
            pt.ha[k] = HA[p]; pt.hb[k] = HB[p]; pt.oa[k] = OA[p]; pt.ob[k] = OB[p];
            pt.rows[k] = R[p];
            pt.key[k] = row0[p];
            pt.t0[k] = tiles;
            tiles += (R[p] + TR - 1) / TR;
        }
        pt.t0[pt.n] = tiles;
        p0 = p;
        if (pt.n == 0) break;
        long blocks = (tiles + 3) / 4;
        if (blocks > num_cus() * 8L) blocks = num_cus() * 8L;
        const bool seq = g_force[FORCE_NULL_SEQ] != 0;          // (tests: the column-by-column kernel on shapes that fit the bit strings)
        if (!seq && !full && 2 * bits_bytes <= 24 * 1024) {     // -g: two thresholds, two bit strings
            const size_t shmem = 4 * ((size_t)2 * TR * 2 * S + 2 * bits_bytes);
            static DynLds lds_attr2;
            EPG_HIP(ensure_dyn_lds(lds_attr2, reinterpret_cast<const void*>(k_null_hist_h2), 160 * 1024));
            hipLaunchKernelGGL(k_null_hist_h2, dim3((unsigned)blocks), dim3(256), shmem, st, pt, S, n_cols, ga, gb, (u64)seed, TR, NW);
            EPG_LAUNCH_CHECK("k_null_hist_h2");
        } else if (!seq && full && bits_bytes <= 24 * 1024) {
            const size_t shmem = 4 * ((size_t)2 * TR * 2 * S + bits_bytes);
            static DynLds lds_attr;
            EPG_HIP(ensure_dyn_lds(lds_attr, reinterpret_cast<const void*>(k_null_hist_h), 160 * 1024));
            hipLaunchKernelGGL(k_null_hist_h, dim3((unsigned)blocks), dim3(256), shmem, st, pt, S, n_cols, ga, gb, (u64)seed, TR, NW);
            EPG_LAUNCH_CHECK("k_null_hist_h");
        } else {
            const size_t shmem = (size_t)4 * 2 * TR * 2 * S;
            hipLaunchKernelGGL(k_null_hist_h_seq, dim3((unsigned)blocks), dim3(256), shmem, st, pt, S, n_cols, ga, gb, (u64)seed, TR);
            EPG_LAUNCH_CHECK("k_null_hist_h_seq");
        }
    }
    return EPG_OK;
}

int null_hist_from_binhist_impl(const uint16_t* HA, const uint16_t* HB, int64_t R, int32_t S, int32_t n_cols, int32_t ga, int32_t gb,
                                uint64_t seed, int64_t row0, uint16_t* OA, uint16_t* OB, hipStream_t st) {
    if (R > 0 && (!HA || !HB || !OA || !OB)) return fail(EPG_ERR_INVALID_ARG, "null_hist_from_binhist: NULL argument");
    return null_hist_parts_impl(1, &HA, &HB, &R, S, n_cols, ga, gb, seed, &row0, &OA, &OB, st);
}

// ---------------------------------------------------------------------------------------------------------------
// Paired mode, count pass AND null draw in one kernel (round 5).  Two kernels on two streams share every SIMD and lose ~30 % of
// the issue slots to each other (count pass 2.44 ms + sampler 2.06 ms alone, 3.3-3.4 ms together: profiles/r05g_*).  Here a
// wave owns tiles of 64 bins of a part: it counts the tile's rows of group A and of group B with k_bin_hist's core (four
// 16-row sub-tiles, quad per row), leaves the two real groups'
// histograms in its LDS slot, writes them out, then every lane draws its row's null groups from them (nh_sample_row_full) and the
// wave writes those out too.  The waves of a CU are in different phases, so its memory pipe and its VALU are busy at the same
// time without a second kernel.  Same integers and the same draws as epg_bin_hist_parts + epg_null_hist_from_binhist_parts
// (tests/test_hip_abi_calls.py).  Compile-time S (15 / 18 / 25) and groups of 128 bytes per row (both widths the same number, <= 4);
// the default group sizes only (ga = NA, gb = NB: one bit string); everything else takes the two kernels.
// ---------------------------------------------------------------------------------------------------------------
constexpr int PC_MAXP = 32;
struct PcParts {
    const char* xa[PC_MAXP];
    const char* xb[PC_MAXP];
    u16* ha[PC_MAXP];
    u16* hb[PC_MAXP];
    u16* oa[PC_MAXP];
    u16* ob[PC_MAXP];
    long rows[PC_MAXP];
    long ldxa[PC_MAXP];
    long ldxb[PC_MAXP];
    long key[PC_MAXP];
    long t0[PC_MAXP + 1];                  // first tile (64 rows) of every part, and their total
    int n;
};

template <int S>
__device__ __forceinline__ void pc_stage_row(char* srow, const u32 (&d)[(S + 1) / 2], int j) {
    constexpr int ND = (S + 1) / 2;
    if constexpr ((S & 1) == 0) {                        // even S: whole dwords
#pragma unroll
        for (int k = 0; k < (ND + 3) / 4; ++k) {
            const u32 v = sel4(d[4 * k], 4 * k + 1 < ND ? d[4 * k + 1] : 0u, 4 * k + 2 < ND ? d[4 * k + 2] : 0u, 4 * k + 3 < ND ? d[4 * k + 3] : 0u, j);
            if (4 * k + j < ND) *reinterpret_cast<u32*>(srow + 4 * (4 * k + j)) = v;
        }
    } else {
#pragma unroll
        for (int k = 0; k < (S + 3) / 4; ++k) {
            const u32 v = (j & 2) ? (2 * k + 1 < ND ? d[2 * k + 1] : 0u) : d[2 * k];
            const u32 c = (j & 1) ? v >> 16 : v & 0xffffu;
            if (4 * k + j < S) *reinterpret_cast<u16*>(srow + 2 * (4 * k + j)) = (u16)c;
        }
    }
}

template <int S, int NG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_pair_count_null(const PcParts pt, int NA, int NB, u64* __restrict__ counts, u64 seed, int NW) {
    constexpr int ND = (S + 1) / 2;
    constexpr int ROWB = 2 * S;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ u64 s_cnt[S + 1];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 3, b = lane >> 2;
    const size_t per_wave = (size_t)2 * 64 * ROWB + (size_t)NW * 256;
    char* sa = smem + (size_t)wave * per_wave;
    char* sb = sa + 64 * ROWB;
    u32* bitsA = reinterpret_cast<u32*>(sb + 64 * ROWB) + lane;
    if (threadIdx.x <= S) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    u32 accp[ND];
#pragma unroll
    for (int m = 0; m < ND; ++m) accp[m] = 0;
    const int nmax = NA > NB ? NA : NB;
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
    const RowGeom gA = make_geom(NA), gB = make_geom(NB);
    const int n_cols = NA + NB;
    const long ntiles = pt.t0[pt.n];
    const long tstride = (long)gridDim.x * 4;
    // a tile's descriptor; `nx` is the tile after this one (its first sub-tile is loaded before this tile's draws start)
    struct Tile { const char* xa; const char* xb; u16* ha; u16* hb; u16* oa; u16* ob; long R, ldxa, ldxb, key, r0; };
    int part = 0;
    auto describe = [&](long tile, Tile& t) {
        while (tile >= pt.t0[part + 1]) ++part;                   // (a wave's tiles ascend: the part only moves forward)
        t.xa = pt.xa[part]; t.xb = pt.xb[part]; t.ha = pt.ha[part]; t.hb = pt.hb[part]; t.oa = pt.oa[part]; t.ob = pt.ob[part];
        t.R = pt.rows[part]; t.ldxa = pt.ldxa[part]; t.ldxb = pt.ldxb[part]; t.key = pt.key[part];
        t.r0 = (tile - pt.t0[part]) * 64;
    };
    // the loads of ONE 16-row sub-tile of one group (2 NG dwordx4 per lane), kept in flight while the previous sub-tile is counted
    u32 wa[NG][8], wb[NG][8];
    auto issue = [&](const char* X, long ldx, long R_, long r0, int sub, const RowGeom& g, u32 (&w)[NG][8]) {
        const long row = r0 + 16 * sub + b;
        const char* rowp = X + (row < R_ ? row : R_ - 1) * ldx;
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
    };
    auto count_into = [&](u32 (&w)[NG][8], char* srow, bool valid) {
        u32 cnt[S];
#pragma unroll
        for (int s = 0; s < S; ++s) cnt[s] = 0;
#pragma unroll
        for (int t = 0; t < NG; ++t) count_group<S>(w[t], cnt);
        u32 d[ND];
        pack_reduce<S>(cnt, d);
        pc_stage_row<S>(srow, d, j);
        if (counts) {
#pragma unroll
            for (int m = 0; m < ND; ++m) accp[m] += valid ? d[m] : 0u;
            if (++since >= flush_every) flush();
        }
    };
    long tile = (long)blockIdx.x * 4 + wave;
    Tile cur, nx;
    if (tile < ntiles) {
        describe(tile, cur);
        issue(cur.xa, cur.ldxa, cur.R, cur.r0, 0, gA, wa);
        issue(cur.xb, cur.ldxb, cur.R, cur.r0, 0, gB, wb);
    }
    for (; tile < ntiles; tile += tstride) {
        const bool more = tile + tstride < ntiles;
        if (more) describe(tile + tstride, nx);
        const int rows = (int)(cur.R - cur.r0 < 64 ? cur.R - cur.r0 : 64);
        // Priorities: the DRAW phase runs at wave priority 3, the count phase at 0 -- like the two-kernel form, where the sampler
        // (the longer chain of VALU work) had to go first.  Measured, job ms on one box: draws first 4.27 / 4.27, count phase first
        // 4.45 / 4.56, no priorities 4.58 / 4.65; (count 1, draws 3) and (count 0, draws 1) are within noise of draws first.
        __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
            const bool valid = cur.r0 + 16 * sub + b < cur.R;
            count_into(wa, sa + (16 * sub + b) * ROWB, valid);
            if (sub < 3) issue(cur.xa, cur.ldxa, cur.R, cur.r0, sub + 1, gA, wa);
            else if (more) issue(nx.xa, nx.ldxa, nx.R, nx.r0, 0, gA, wa);      // in flight during this tile's draws
            count_into(wb, sb + (16 * sub + b) * ROWB, valid);
            if (sub < 3) issue(cur.xb, cur.ldxb, cur.R, cur.r0, sub + 1, gB, wb);
            else if (more) issue(nx.xb, nx.ldxb, nx.R, nx.r0, 0, gB, wb);
        }
        __builtin_amdgcn_wave_barrier();
        store_staged(sa, reinterpret_cast<char*>(cur.ha + cur.r0 * S), rows * ROWB, lane);
        store_staged(sb, reinterpret_cast<char*>(cur.hb + cur.r0 * S), rows * ROWB, lane);
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_setprio(3);
        if (lane < rows)
            nh_sample_row_full(reinterpret_cast<u16*>(sa + lane * ROWB), reinterpret_cast<u16*>(sb + lane * ROWB), bitsA, S, n_cols, NA, seed,
                               (u64)(cur.key + cur.r0 + lane));
        __builtin_amdgcn_wave_barrier();
        store_staged(sa, reinterpret_cast<char*>(cur.oa + cur.r0 * S), rows * ROWB, lane);
        store_staged(sb, reinterpret_cast<char*>(cur.ob + cur.r0 * S), rows * ROWB, lane);
        __builtin_amdgcn_wave_barrier();
        cur = nx;
    }
    if (counts) {
        flush();
        __syncthreads();
        if ((int)threadIdx.x < S && s_cnt[threadIdx.x]) atomicAdd(&counts[threadIdx.x], s_cnt[threadIdx.x]);
    }
}

template <int S, int NG>
static int launch_pair_count_null(const PcParts& pt, int NA, int NB, u64* counts, u64 seed, hipStream_t st) {
    const int NW = (NA + NB + 31) / 32;
    const size_t shmem = 4 * ((size_t)2 * 64 * 2 * S + (size_t)NW * 256);
    static DynLds lds_attr;                                  // (one per <S, NG> instantiation)
    EPG_HIP(ensure_dyn_lds(lds_attr, reinterpret_cast<const void*>(k_pair_count_null<S, NG>), 160 * 1024 - 1024));   // (the kernel also holds a small static array)
    long blocks = (pt.t0[pt.n] + 3) / 4;
    const long per_cu = (long)((160 * 1024 - 1024) / shmem);
    const long cap = num_cus() * (per_cu < 1 ? 1 : per_cu);
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL((k_pair_count_null<S, NG>), dim3((unsigned)blocks), dim3(256), shmem, st, pt, NA, NB, counts, seed, NW);
    EPG_LAUNCH_CHECK("k_pair_count_null");
    return EPG_OK;
}

template <int S>
static int dispatch_pair_count_null(int ng, const PcParts& pt, int NA, int NB, u64* counts, u64 seed, hipStream_t st) {
    switch (ng) {
        case 1: return launch_pair_count_null<S, 1>(pt, NA, NB, counts, seed, st);
        case 2: return launch_pair_count_null<S, 2>(pt, NA, NB, counts, seed, st);
        case 3: return launch_pair_count_null<S, 3>(pt, NA, NB, counts, seed, st);
        default: return launch_pair_count_null<S, 4>(pt, NA, NB, counts, seed, st);
    }
}

// -> EPG_ERR_UNSUPPORTED when the shape is not the fused kernel's (the caller then takes epg_bin_hist_parts +
// epg_null_hist_from_binhist_parts: the same results)
int pair_count_null_parts_impl(int32_t nparts, const int8_t* const* XA, const int8_t* const* XB, const int64_t* R, int32_t NA, int32_t NB,
                               const int64_t* ldxa, const int64_t* ldxb, int32_t S, uint16_t* const* HA, uint16_t* const* HB, int64_t* counts,
                               uint64_t seed, const int64_t* row0, uint16_t* const* OA, uint16_t* const* OB, hipStream_t st) {
    if (nparts < 0 || NA < 1 || NB < 1 || S < 1) return fail(EPG_ERR_INVALID_ARG, "pair_count_null: bad shape");
    if (nparts && (!XA || !XB || !R || !ldxa || !ldxb || !HA || !HB || !row0 || !OA || !OB)) return fail(EPG_ERR_INVALID_ARG, "pair_count_null: NULL argument array");
    const int ng = (NA + 127) / 128;
    if (!(S == 15 || S == 18 || S == 25) || ng != (NB + 127) / 128 || ng > 4 || NA + NB > 3072)
        return fail(EPG_ERR_UNSUPPORTED, "pair_count_null: S=%d, widths %d + %d are not the fused kernel's", S, NA, NB);
    for (int p = 0; p < nparts; ++p) {
        if (R[p] < 0) return fail(EPG_ERR_INVALID_ARG, "pair_count_null: bad shape of part %d", p);
        if (R[p] == 0) continue;
        if (!XA[p] || !XB[p] || !HA[p] || !HB[p] || !OA[p] || !OB[p]) return fail(EPG_ERR_INVALID_ARG, "pair_count_null: NULL argument of part %d", p);
        if ((reinterpret_cast<uintptr_t>(HA[p]) | reinterpret_cast<uintptr_t>(HB[p]) | reinterpret_cast<uintptr_t>(OA[p]) | reinterpret_cast<uintptr_t>(OB[p])) & 15)
            return fail(EPG_ERR_INVALID_ARG, "pair_count_null: histogram arrays must be 16-byte aligned");
        // the 16-byte loads of a row's last chunk must stay inside the row pitch (engine.alloc_states pads to 16)
        if (ldxa[p] < 16L * ((NA + 15) / 16) || ldxb[p] < 16L * ((NB + 15) / 16))
            return fail(EPG_ERR_UNSUPPORTED, "pair_count_null: row pitch of part %d is not padded to 16 bytes", p);
    }
    u64* cnt = reinterpret_cast<u64*>(counts);
    for (int p0 = 0; p0 < nparts;) {
        PcParts pt;
        memset(&pt, 0, sizeof(pt));
        long tiles = 0;
        int p = p0;
        for (; p < nparts && pt.n < PC_MAXP; ++p) {
            if (R[p] == 0) continue;
            const int k = pt.n++;
            pt.xa[k] = reinterpret_cast<const char*>(XA[p]); pt.xb[k] = reinterpret_cast<const char*>(XB[p]);
            pt.ha[k] = HA[p]; pt.hb[k] = HB[p]; pt.oa[k] = OA[p]; pt.ob[k] = OB[p];
            pt.rows[k] = R[p]; pt.ldxa[k] = ldxa[p]; pt.ldxb[k] = ldxb[p]; pt.key[k] = row0[p];
            pt.t0[k] = tiles;
            tiles += (R[p] + 63) / 64;
        }
        pt.t0[pt.n] = tiles;
        p0 = p;
        if (pt.n == 0) break;
        int rc;
        if (S == 18) rc = dispatch_pair_count_null<18>(ng, pt, NA, NB, cnt, (u64)seed, st);
        else if (S == 15) rc = dispatch_pair_count_null<15>(ng, pt, NA, NB, cnt, (u64)seed, st);
        else rc = dispatch_pair_count_null<25>(ng, pt, NA, NB, cnt, (u64)seed, st);
        if (rc) return rc;
    }
    return EPG_OK;
}

// quiescent from cached histograms of the two real groups (scores.py:294-303): every column of A and of B holds the
// quiescent state <=> its count equals the group's width
__global__ __launch_bounds__(256) void k_quiescent_h(const u16* __restrict__ HA, const u16* __restrict__ HB, long R, int S, int NA, int NB,
                                                      int qstate, uint8_t* __restrict__ mask) {
    const long row = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= R) return;
    mask[row] = (uint8_t)(HA[row * S + qstate] == (u16)NA && HB[row * S + qstate] == (u16)NB);
}

int quiescent_from_binhist_impl(const uint16_t* HA, const uint16_t* HB, int64_t R, int32_t S, int32_t NA, int32_t NB, int32_t qstate,
                                uint8_t* mask, hipStream_t st) {
    if (R < 0 || S < 1 || NA < 1 || NB < 1 || NA > 65535 || NB > 65535 || qstate >= S)
        return fail(EPG_ERR_INVALID_ARG, "quiescent_from_binhist: bad shape");
    if (R == 0) return EPG_OK;
    if (!HA || !HB || !mask) return fail(EPG_ERR_INVALID_ARG, "quiescent_from_binhist: NULL argument");
    if (qstate < 0) {  // filtering off (run.py:113: -q 0 -> -1): nothing is quiescent
        EPG_HIP(hipMemsetAsync(mask, 0, (size_t)R, st));
        return EPG_OK;
    }
    hipLaunchKernelGGL(k_quiescent_h, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, st, HA, HB, (long)R, S, NA, NB, qstate, mask);
    EPG_LAUNCH_CHECK("k_quiescent_h");
    return EPG_OK;
}


int null_hist_impl(const int8_t* XA, int32_t NA, int64_t ldxa, const int8_t* XB, int32_t NB, int64_t ldxb, int64_t R, int32_t S,
                   int32_t ga, int32_t gb, uint64_t seed, int64_t row0, uint16_t* HA, uint16_t* HB, hipStream_t st) {
    if (R < 0 || NA < 1 || NB < 1 || ldxa < NA || ldxb < NB || S < 1)
        return fail(EPG_ERR_INVALID_ARG, "null_hist: bad shape");
    if (S > 31) return fail(EPG_ERR_UNSUPPORTED, "null_hist: the matrix-scanning kernel decodes five bits; for S=%d count the groups with "
                                                 "epg_bin_hist and draw with epg_null_hist_from_binhist", S);
    if (ga < 0 || gb < 0 || (long)ga + gb > (long)NA + NB)
        return fail(EPG_ERR_INVALID_ARG, "null_hist: group sizes %d + %d exceed the %d columns", ga, gb, NA + NB);
    if (NA + NB > 65535) return fail(EPG_ERR_UNSUPPORTED, "null_hist: more than 65535 columns");
    if (R == 0) return EPG_OK;
    if (!XA || !XB || !HA || !HB) return fail(EPG_ERR_INVALID_ARG, "null_hist: NULL argument");
    const size_t shmem = (size_t)(S + 1) * 256 * 4 + (size_t)256 * NH_LD;
    hipLaunchKernelGGL(k_null_hist, dim3((unsigned)((R + 255) / 256)), dim3(256), shmem, st, reinterpret_cast<const char*>(XA), NA,
                       (long)ldxa, reinterpret_cast<const char*>(XB), NB, (long)ldxb, (long)R, S, ga, gb, (u64)seed, (long)row0, HA, HB);
    EPG_LAUNCH_CHECK("k_null_hist");
    return EPG_OK;
}

}  // namespace epg
