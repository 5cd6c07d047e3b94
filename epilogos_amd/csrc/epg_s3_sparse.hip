// S3 score, modal-state form (round 3).  gfx950 only.  SELECTABLE (EPG_S3_SCORE=sparse), NOT the default: it returns the
// dense kernel's integers from a third of its gathers and is slower -- 56.0 against 39.1 ms per 500 K bins at N = 833, S = 18,
// chr1-like states (profiles/r03d_*).  Why, measured (tools/ubench/gpr_idx.hip): a SIMD of this chip retires a wave
// instruction of ANY kind -- SALU, VALU, LDS -- every ~3.7 cycles (4 waves per SIMD; `v_add_u32` alone 2.9, `s_set_gpr_idx_on` +
// indexed `v_add_u32` 7.5, `s_mov_b64 exec` + `v_add_u32` 6.6, `v_readlane` + 3 SALU + `v_add_u32` 18.4), so a gather kernel
// is bound by its instruction COUNT.  The dense kernel spends 3 instructions per gather instruction (address, ds_read,
// accumulate) and 48 of them per half wave and phase; this kernel needs ~17 per live pair of bins (which accumulator a
// gather belongs to is data: two index switches, two address and two accumulate instructions per pair under EXEC halves,
// item fetch and decode) for 0.35 x 48 of them -- 1.4x the dense kernel's instruction count, and 1.43x its time.  A form with one
// bin per wave (no halves, ~6 instructions per item) would need 64-biosample slabs, halve the bins per workgroup and be
// bound by the slab stream at twice today's volume.  The algebra below is exact and cuts the terms to 8 % of N (N - 1) when
// applied to both biosamples of a pair; on this hardware model neither level pays.
//
// Reference: scores.py:455-506 s3Score.  score[bin, x_b] += sum_{a != b} T[a, b, x_a, x_b]  with T = kl(float32(1)/P, q)
// in float32 (scores.py:479-480) -- P = N (N - 1) table terms per bin, 693 056 at N = 833.
//
// k_s3_score_bl (epg_s3_lanes.hip) does every one of those terms as an LDS gather and sits at the rate a CU issues
// ds_read_b32 (76 ms per 1 M bins).  Real epigenomes are mostly ONE state (71 % of the cells of the EpiMap matrix are
// "quiescent"), and with m that state
//     T[a, b, i, j] = T[a, b, m, j] + D[a, b, i, j],      D[a, b, m, j] = 0,
// so for a (bin, b)
//     sum_{a != b} T[a, b, x_a, x_b] = B1[b][x_b] + sum_{a != b, x_a != m} D[a, b, x_a, x_b],     B1[b][j] = sum_{a != b} T[a, b, m, j]:
// a per-biosample table looked up once, and table gathers only for the biosamples a that are NOT in the modal state in
// this bin -- 29 % of them.  In the 32-bit fixed point of the dense kernel (entries rounded once to units of
// max|T| N / 2^30, then integer arithmetic) the identity is exact: both kernels produce the same integers.
//
// The kernel keeps the dense kernel's frame -- a lane is a biosample b (32 consecutive ones per half wave), a half wave
// is one bin at a time, the (a, 32 b) slab of the table streams through two LDS buffers by LDS-DMA from a loader wave,
// one barrier per a, epilogue through LDS into 64-bit fixed-point cells -- and changes what a phase does:
//  * for every (wave, a) a prep kernel has listed the bins of the wave's two halves in which a is off the modal state
//    (k_sp_count / k_sp_write); the k-th live bin of half 0 and the k-th of half 1 share a gather instruction (the
//    shorter list is padded with a dummy bin).  A phase issues ~0.32 K gathers instead of K.
//  * WHICH accumulator a gather belongs to is now data: the accumulators (one VGPR per bin of a half wave) are
//    addressed through the VGPR index mode (s_set_gpr_idx_on: M0 is added to the register number of selected
//    operands), which the compiler cannot be asked for -- the whole a loop of a gather wave is one asm statement with
//    the accumulators pinned to v[40:104] and the packed x_b bytes to v[106:122].  The two halves of an instruction
//    work on different bins, hence different registers: the address part runs under EXEC = lower / upper half, the
//    accumulation uses DPP row masks (rows 0-1 / rows 2-3), each with its own index.
//  * a list item is one dword per gather instruction (k0, x0, k1, x1: per half the bin = accumulator index and the slab row).
//    The lists reach the wave through VECTOR loads, 64 items per global_load_dword, two registers deep, and are handed to the
//    scalar side by v_readlane: the first build fetched them with s_load_dwordx16 and ran 2x SLOWER than the dense kernel --
//    scalar loads share lgkmcnt with the LDS reads and return out of order, so every batch of four gathers waited for a
//    full trip to memory of the next batch's items (3.6 us per phase).  vmcnt counts in order and nothing else uses it here.
//  * the address of a gather is (x_b * row stride + 4 * lane), kept in a register per bin next to its accumulator, plus
//    128 * x_a' from the item: one v_lshl_add_u32 per half, four items per EXEC switch.
//  * the slab is laid out [x_b][x_a'][b] (x_a' = the 17 states other than m): 128 consecutive bytes per (x_b, x_a'), so a
//    lane's bank is its lane number whatever the states are -- conflict free like the dense kernel's layout, and 39 KB
//    instead of 42.75 KB per (a, chunk).  THREE slab buffers (117 KB of LDS): the loader wave runs two slabs ahead, so a
//    slab has a whole phase more than its own to arrive (the dense kernel's two buffers give it one; its "stream alone"
//    time is a latency, not a bandwidth).  The third buffer lies beyond the 16-bit ds_read offset: the prep kernel adds the
//    difference, in rows of 128 bytes, to the x fields of the items of every third phase.
// Accumulator and address cost two registers per (bin, b) whether or not b is in the modal state, so a workgroup covers
// 15 waves x 2 x 48 = 1440 bins per slab, like the dense kernel.
#include "epg_common.h"

#include <stdlib.h>

#include <vector>

namespace epg {

int transpose_states_bad(const char* X, int64_t R, int32_t N, int64_t ldx, int32_t S, char* XT, int64_t Rp, int shift, int bad, hipStream_t st);
int bin_hist_impl(const int8_t* X8, int64_t R, int32_t N, int64_t ldx, int32_t S, uint16_t* H, int64_t* counts, hipStream_t st);
int score_s3_lanes(const int8_t* X8, int64_t R, int32_t N, int64_t ldx, int32_t S, const float* q, double* out64, float* out32, void* ws,
                   int64_t ws_bytes, hipStream_t st);
int64_t s3_lanes_ws_bytes(int64_t R, int N, int S);

constexpr int SP_GW = 15;                          // gather waves; wave SP_GW of the workgroup is the loader
constexpr int SP_THREADS = 64 * (SP_GW + 1);
constexpr int SP_K = 48;                           // bins of a half wave = accumulator (and address) registers of a gather wave
constexpr int SP_WB = 2 * SP_K;                    // bins of a wave
constexpr int SP_SLICE = SP_GW * SP_WB;            // bins of a workgroup (1920)
constexpr int SP_SMAX = 19;                        // 19 * 18 * 128 = 43 KiB per slab; the third buffer's bias must fit an item's 8-bit x field
constexpr int SP_EB = 16;                          // bins of a half wave per round of the epilogue
constexpr int SP_BATCH = 4;                        // gather instructions per batch of items (two s_load_dwordx16)
constexpr int SP_BLOCK = 64;                       // items (dwords) per vector load of a wave: 16 batches
constexpr int SP_BUF2_IMM = 65408;                 // the largest multiple of 128 a ds_read offset can hold

static inline int sp_chb(int S) { return (int)align_up((int64_t)S * (S - 1) * 128, 1024); }   // bytes of one (chunk, a) slab: whole 1 KiB DMA pieces
static inline int sp_nchunk(int N) { return (N + 31) / 32; }

__device__ __forceinline__ float sp_t(float qv, float obs) {   // k_s3_table's arithmetic (float32 like scores.py:479-480)
    float v = 0.0f;
    if (qv != 0.0f) {
        const float r = obs / qv;
        if (r > 0.0f) v = obs * log2f(r);
    }
    return v;
}

// max |T| over the off-diagonal table (bits of a non-negative float order like unsigned integers)
__global__ __launch_bounds__(256) void k_sp_tmax(const float* __restrict__ q, int N, int S, u32* __restrict__ maxbits) {
    const long SS = (long)S * S, total = (long)N * N * SS;
    const float obs = 1.0f / (float)((long)N * (N - 1));
    u32 m = 0;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long ab = e / SS;
        if (ab / N == ab % N) continue;
        const u32 bits = __float_as_uint(fabsf(sp_t(q[e], obs)));
        m = bits > m ? bits : m;
    }
    for (int o = 32; o; o >>= 1) {
        const u32 t = (u32)__shfl_xor((int)m, o);
        m = t > m ? t : m;
    }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(maxbits, m);
}

// scal[0] = unit (score per fixed-point step), scal[1] = 1 / unit.  Units of max|T| N / 2^30: an entry is < 2^30 / N, a
// difference of two < 2^31 / N, and the N - 1 differences of a (bin, b) add up in an int32 without overflow.
__global__ void k_sp_unit(const u32* __restrict__ maxbits, int N, double* __restrict__ scal) {
    const double mx = (double)__uint_as_float(*maxbits);
    const double unit = mx * (double)N / 1073741824.0;
    scal[0] = unit;
    scal[1] = unit > 0.0 ? 1.0 / unit : 0.0;
}

__device__ __forceinline__ int sp_fix(const float* __restrict__ q, int N, int S, int a, int b, int i, int j, float obs, double inv) {
    return (int)__double2ll_rn((double)sp_t(q[(((long)a * N + b) * S + i) * S + j], obs) * inv);
}

// TD[c][a][x_b = j][x'][l] = fix(T[a, 32 c + l, i, j]) - fix(T[a, 32 c + l, m, j]) with i = x' + (x' >= m); zero for b >= N,
// a == b and in the padding of a slab
__global__ __launch_bounds__(256) void k_sp_td_build(const float* __restrict__ q, int N, int S, int m, int chw /*dwords per (c, a)*/, int nchunk,
                                                    const double* __restrict__ scal, int* __restrict__ TD) {
    const long total = (long)nchunk * N * chw;
    const float obs = 1.0f / (float)((long)N * (N - 1));
    const double inv = scal[1];
    const int rowj = (S - 1) * 32;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long ca = e / chw;
        const int wd = (int)(e - ca * chw);
        const int c = (int)(ca / N), a = (int)(ca - (long)c * N);
        const int j = wd / rowj, rem = wd - j * rowj, xp = rem >> 5, l = rem & 31;
        const int b = 32 * c + l;
        int v = 0;
        if (j < S && b < N && b != a) {
            const int i = xp + (xp >= m ? 1 : 0);
            v = sp_fix(q, N, S, a, b, i, j, obs, inv) - sp_fix(q, N, S, a, b, m, j, obs, inv);
        }
        TD[e] = v;
    }
}

// B1[b][j] = sum_{a != b} fix(T[a, b, m, j])
__global__ __launch_bounds__(256) void k_sp_base(const float* __restrict__ q, int N, int S, int m, const double* __restrict__ scal,
                                                long long* __restrict__ B1) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N * S) return;
    const int b = e / S, j = e - b * S;
    const float obs = 1.0f / (float)((long)N * (N - 1));
    const double inv = scal[1];
    long long s = 0;
    for (int a = 0; a < N; ++a)
        if (a != b) s += sp_fix(q, N, S, a, b, m, j, obs, inv);
    B1[e] = s;
}

// ---- the item lists.  Wave slot ws = 15 * slice + wave owns bins [128 ws, 128 ws + 128) of the (padded) matrix: half h
// the 64 from 128 ws + 64 h on.  XT4 holds 4 * state per byte ("not a state" = 4 S).
__device__ __forceinline__ int sp_live(const unsigned char* __restrict__ p, int m4, int s4, unsigned char* k_out, unsigned char* x_out) {
    int n = 0;
#pragma unroll
    for (int g = 0; g < SP_K / 16; ++g) {
        const uint4 v = *reinterpret_cast<const uint4*>(p + 16 * g);     // wave slots start at multiples of 128 bins
        const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int sv = (int)((w[u >> 2] >> (8 * (u & 3))) & 0xffu);
            if (sv != m4 && sv < s4) {
                if (k_out) { k_out[n] = (unsigned char)(16 * g + u); x_out[n] = (unsigned char)(sv >> 2); }
                ++n;
            }
        }
    }
    return n;
}

// NB[ws][a] = batches of four gather instructions of (ws, a) (at least one: an all-dummy batch keeps the kernel's batch
// stream free of empty phases); TOT[ws] = their sum over a
__global__ __launch_bounds__(256) void k_sp_count(const char* __restrict__ XT4, long Rp, int N, int S, int m, u32* __restrict__ NB,
                                                 u32* __restrict__ TOT) {
    __shared__ u32 s_sum;
    if (threadIdx.x == 0) s_sum = 0;
    __syncthreads();
    const long ws = blockIdx.x;
    const unsigned char* base = reinterpret_cast<const unsigned char*>(XT4) + ws * SP_WB;
    u32 mine = 0;
    for (int a = threadIdx.x; a < N; a += 256) {
        const unsigned char* p = base + (long)a * Rp;
        const int n0 = sp_live(p, 4 * m, 4 * S, nullptr, nullptr), n1 = sp_live(p + SP_K, 4 * m, 4 * S, nullptr, nullptr);
        const int n = n0 > n1 ? n0 : n1;
        const u32 nb = n ? (u32)((n + SP_BATCH - 1) / SP_BATCH) : 1u;
        NB[ws * N + a] = nb;
        mine += nb;
    }
    atomicAdd(&s_sum, mine);
    __syncthreads();
    if (threadIdx.x == 0) TOT[ws] = s_sum;
}

// Items of wave slot ws0 + blockIdx.x: one dword per gather instruction, k0 | x0 << 8 | k1 << 16 | x1 << 24, the phases of a
// slot one after the other from LIST + 64 BASE[slot] dwords on (NB gives the lengths, in batches of four).  x = the slab
// row x_a' plus, in every third phase, `bias2` rows: what the third slab buffer lies beyond the reach of a ds_read offset.
__global__ __launch_bounds__(256) void k_sp_write(const char* __restrict__ XT4, long Rp, int N, int S, int m, long ws0, const u32* __restrict__ NB,
                                                 const unsigned long long* __restrict__ BASE, u32* __restrict__ LIST, int bias2) {
    __shared__ u32 s_part[256];
    const long ws = ws0 + blockIdx.x;
    const unsigned char* base = reinterpret_cast<const unsigned char*>(XT4) + ws * SP_WB;
    const int per = (N + 255) / 256;
    const int a_lo = threadIdx.x * per < N ? threadIdx.x * per : N, a_hi = a_lo + per < N ? a_lo + per : N;
    u32 sum = 0;
    for (int a = a_lo; a < a_hi; ++a) sum += NB[ws * N + a];
    s_part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 run = 0;
        for (int t = 0; t < 256; ++t) { const u32 v = s_part[t]; s_part[t] = run; run += v; }
    }
    __syncthreads();
    u32* out = LIST + BASE[blockIdx.x] * SP_BLOCK + (unsigned long long)s_part[threadIdx.x] * SP_BATCH;
    unsigned char k0[SP_K], x0[SP_K], k1[SP_K], x1[SP_K];
    for (int a = a_lo; a < a_hi; ++a) {
        const unsigned char* p = base + (long)a * Rp;
        const int n0 = sp_live(p, 4 * m, 4 * S, k0, x0), n1 = sp_live(p + SP_K, 4 * m, 4 * S, k1, x1);
        const u32 n = NB[ws * N + a] * SP_BATCH;
        const u32 bias = (a % 3 == 2) ? (u32)bias2 : 0u;
        for (u32 it = 0; it < n; ++it) {
            // a dummy half reads the dummy address register and adds into the dummy accumulator (index K): row 0 of the slab
            u32 ka = SP_K, xa = 0, kb = SP_K, xb = 0;
            if ((int)it < n0) { ka = k0[it]; const int i = x0[it]; xa = (u32)(i - (i > m ? 1 : 0)); }
            if ((int)it < n1) { kb = k1[it]; const int i = x1[it]; xb = (u32)(i - (i > m ? 1 : 0)); }
            out[it] = ka | ((xa + bias) << 8) | (kb << 16) | ((xb + bias) << 24);
        }
        out += n;
    }
}

// ---- the loader's side of a phase (as in epg_s3_lanes.hip)
template <int NP>
__device__ __forceinline__ void sp_request(const char* __restrict__ td_a, u32 loff, int npieces, char* dst) {
#pragma unroll
    for (int r = 0; r < NP; ++r) {
        const int p = r < npieces ? r : npieces - 1;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(td_a + p * 1024 + loff),
                                         (__attribute__((address_space(3))) void*)(dst + p * 1024), 16, 0, 0);
    }
}
template <int NPF>
__device__ __forceinline__ void sp_touch(u32 (&sink)[NPF], const char* __restrict__ td_a, u32 toff, u32 tmax) {
#pragma unroll
    for (int r = 0; r < NPF; ++r) {
        u32 o = toff + (u32)r * 8192u;
        o = o < tmax ? o : tmax;
        asm volatile("global_load_dword %0, %1, %2" : "+v"(sink[r]) : "v"(o), "s"(td_a));
    }
}

// One batch of four gather instructions (items at lanes s28 .. s28 + 3 of v123).  OFS = text of the ds_read offset of the
// phase's slab buffer.  Item: k0 | x0 << 8 | k1 << 16 | x1 << 24.  ACC0 = v24, ABASE0 = v74 (index K = the dummies).
#define SP_BATCH_BODY(OFS)                                                                                       \
    "s_add_u32 s29, s28, 1\n s_add_u32 s30, s28, 2\n s_add_u32 s31, s28, 3\n"                                   \
    "v_readlane_b32 s36, v123, s28\n v_readlane_b32 s37, v123, s29\n"                                            \
    "v_readlane_b32 s38, v123, s30\n v_readlane_b32 s39, v123, s31\n"                                            \
    "s_bfe_u32 s40, s36, 0x80008\n s_bfe_u32 s41, s37, 0x80008\n s_bfe_u32 s42, s38, 0x80008\n s_bfe_u32 s43, s39, 0x80008\n" \
    "s_lshr_b32 s44, s36, 16\n s_lshr_b32 s45, s37, 16\n s_lshr_b32 s46, s38, 16\n s_lshr_b32 s47, s39, 16\n"   \
    "s_lshr_b32 s48, s36, 24\n s_lshr_b32 s49, s37, 24\n s_lshr_b32 s50, s38, 24\n s_lshr_b32 s51, s39, 24\n"   \
    "s_mov_b64 exec, %[lo]\n"                                                                                    \
    "s_set_gpr_idx_on s36, 0x4\n v_lshl_add_u32 %[ad0], s40, 7, v74\n"                                           \
    "s_set_gpr_idx_on s37, 0x4\n v_lshl_add_u32 %[ad1], s41, 7, v74\n"                                           \
    "s_set_gpr_idx_on s38, 0x4\n v_lshl_add_u32 %[ad2], s42, 7, v74\n"                                           \
    "s_set_gpr_idx_on s39, 0x4\n v_lshl_add_u32 %[ad3], s43, 7, v74\n"                                           \
    "s_mov_b64 exec, %[hi]\n"                                                                                    \
    "s_set_gpr_idx_on s44, 0x4\n v_lshl_add_u32 %[ad0], s48, 7, v74\n"                                           \
    "s_set_gpr_idx_on s45, 0x4\n v_lshl_add_u32 %[ad1], s49, 7, v74\n"                                           \
    "s_set_gpr_idx_on s46, 0x4\n v_lshl_add_u32 %[ad2], s50, 7, v74\n"                                           \
    "s_set_gpr_idx_on s47, 0x4\n v_lshl_add_u32 %[ad3], s51, 7, v74\n"                                           \
    "s_mov_b64 exec, -1\n"                                                                                       \
    "ds_read_b32 %[r0], %[ad0] offset:" OFS "\n ds_read_b32 %[r1], %[ad1] offset:" OFS "\n"                      \
    "ds_read_b32 %[r2], %[ad2] offset:" OFS "\n ds_read_b32 %[r3], %[ad3] offset:" OFS "\n"                      \
    "s_waitcnt lgkmcnt(0)\n"                                                                                     \
    "s_mov_b64 exec, %[lo]\n"                                                                                    \
    "s_set_gpr_idx_on s36, 0xa\n v_add_u32 v24, %[r0], v24\n"                                                    \
    "s_set_gpr_idx_on s37, 0xa\n v_add_u32 v24, %[r1], v24\n"                                                    \
    "s_set_gpr_idx_on s38, 0xa\n v_add_u32 v24, %[r2], v24\n"                                                    \
    "s_set_gpr_idx_on s39, 0xa\n v_add_u32 v24, %[r3], v24\n"                                                    \
    "s_mov_b64 exec, %[hi]\n"                                                                                    \
    "s_set_gpr_idx_on s44, 0xa\n v_add_u32 v24, %[r0], v24\n"                                                    \
    "s_set_gpr_idx_on s45, 0xa\n v_add_u32 v24, %[r1], v24\n"                                                    \
    "s_set_gpr_idx_on s46, 0xa\n v_add_u32 v24, %[r2], v24\n"                                                    \
    "s_set_gpr_idx_on s47, 0xa\n v_add_u32 v24, %[r3], v24\n"                                                    \
    "s_mov_b64 exec, -1\n"                                                                                       \
    "s_set_gpr_idx_off\n"
// after a batch: next four lanes of the item register; after 16 batches the prefetched register takes over and the one
// after it is requested.  Then the phase's batch count; falls through at the end of the phase.
#define SP_BATCH_TAIL(LOOP, TAG)                                                                                 \
    "s_add_u32 s28, s28, 4\n"                                                                                    \
    "s_cmp_lt_u32 s28, 64\n"                                                                                     \
    "s_cbranch_scc1 LSP_NOSW" TAG "_%=\n"                                                                        \
    "s_waitcnt vmcnt(0)\n"                                                                                       \
    "v_mov_b32 v123, v124\n"                                                                                     \
    "global_load_dword v124, %[off64], s[20:21]\n"                                                               \
    "s_add_u32 s20, s20, 256\n s_addc_u32 s21, s21, 0\n"                                                         \
    "s_mov_b32 s28, 0\n"                                                                                         \
    "LSP_NOSW" TAG "_%=:\n"                                                                                      \
    "s_sub_u32 s24, s24, 1\n"                                                                                    \
    "s_cmp_eq_u32 s24, 0\n"                                                                                      \
    "s_cbranch_scc0 " LOOP "_%=\n"
// end of a phase: barrier, next phase's batch count (two phases of lookahead on the scalar path)
#define SP_PHASE_END(NEXT)                                                                                       \
    "s_barrier\n"                                                                                                \
    "s_sub_u32 s25, s25, 1\n"                                                                                    \
    "s_cmp_eq_u32 s25, 0\n"                                                                                      \
    "s_cbranch_scc1 LSP_DONE_%=\n"                                                                               \
    "s_mov_b32 s24, s26\n"                                                                                       \
    "s_mov_b32 s26, s27\n"                                                                                       \
    "s_load_dword s27, s[22:23], 0x0\n"                                                                          \
    "s_add_u32 s22, s22, 4\n s_addc_u32 s23, s23, 0\n"                                                           \
    "s_branch " NEXT "_%=\n"

typedef int v32i __attribute__((ext_vector_type(32)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int NP, int BUF1, int BUF2>
__global__ __launch_bounds__(SP_THREADS) void k_s3_score_sp(const char* __restrict__ XT4, long Rp, long R, int N, int S, int chb,
                                                            const char* __restrict__ TD, int nslices, const u32* __restrict__ NB,
                                                            const unsigned long long* __restrict__ BASE, const u32* __restrict__ LIST,
                                                            long ws0, const long long* __restrict__ B1, u64* __restrict__ cells, int ahead) {
    extern __shared__ __attribute__((aligned(1024))) char tab[];
    constexpr int BUF2_IMM = BUF2 < SP_BUF2_IMM ? BUF2 : SP_BUF2_IMM;   // what is missing comes with the items' x fields
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = blockIdx.x / nslices;
    const int slice = blockIdx.x - c * nslices;
    const long slice0 = (long)slice * SP_SLICE;
    // the gathers address LDS by absolute offsets: the dynamic segment must start at 0
    if ((u32)(size_t)(__attribute__((address_space(3))) char*)tab != 0u) __builtin_trap();

    if (wv == SP_GW) {
        // ---- loader wave: slab a + 2 is requested while the gather waves work on slab a (three buffers); before the barrier
        // that opens phase a + 1 only slab a + 1 must have landed -- vmcnt counts in order, so "at most the NP loads of slab
        // a + 2 (and the touches) outstanding" says exactly that
        constexpr int NPF = (NP + 7) / 8;
        static_assert(NP + NPF <= 63, "vmcnt has six bits");
        const char* td = TD + (long)c * N * chb;
        const u32 loff = (u32)lane * 16u, toff = (u32)lane * 128u, tmax = (u32)chb - 128u;
        const int npieces = chb >> 10;
        const int rank = (int)(blockIdx.x >> 3) & 31;                    // workgroups are dealt to the 8 XCDs round robin
        u32 sink[NPF];
#pragma unroll
        for (int r = 0; r < NPF; ++r) sink[r] = 0;
        sp_request<NP>(td, loff, npieces, tab);
        sp_request<NP>(td + (long)(1 < N ? 1 : 0) * chb, loff, npieces, tab + BUF1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");      // slab 0 has landed
        __builtin_amdgcn_s_barrier();
        for (int a = 0; a < N; ++a) {
            const int a2 = a + 2 < N ? a + 2 : N - 1;                    // past the end: a valid slab nobody reads
            const int bsel = (a + 2) % 3;
            sp_request<NP>(td + (long)a2 * chb, loff, npieces, tab + (bsel == 0 ? 0 : bsel == 1 ? BUF1 : BUF2));
            if (ahead > 0 && ((a + ahead) & 31) == rank && a + ahead < N) {
                sp_touch<NPF>(sink, td + (long)(a + ahead) * chb, toff, tmax);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP + NPF) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");   // slab a + 1 has landed
            }
            __builtin_amdgcn_s_barrier();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // nothing may still be writing LDS when the epilogue reuses it
#pragma unroll
        for (int r = 0; r < NPF; ++r) asm volatile("" : "+v"(sink[r])::"memory");
        __builtin_amdgcn_s_barrier();                                   // (the gather waves wait here before their epilogue)
    } else {
        // ---- gather waves
        const int l = lane & 31, h = lane >> 5;
        const long bin0 = slice0 + wv * SP_WB + h * SP_K;               // first bin of this half wave; Rp is a multiple of SP_SLICE
        const int b = 32 * c + l, bl = b < N ? b : N - 1;
        const char* pb = XT4 + (long)bl * Rp + bin0;
        const u32 lane4 = (u32)l * 4u, jst = (u32)(S - 1) * 128u;
        // per bin: the byte offset of (x_b, lane) in a slab; the item adds 128 x_a'
        v32i ab0;
        v16i ab1;
#pragma unroll
        for (int g = 0; g < SP_K / 16; ++g) {
            const uint4 v = *reinterpret_cast<const uint4*>(pb + 16 * g);
            const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                u32 j = ((w[u >> 2] >> (8 * (u & 3))) & 0xffu) >> 2;
                j = j < (u32)S ? j : 0u;                                 // "not a state": any valid row, dropped at the end
                const int kk = 16 * g + u;
                if (kk < 32) ab0[kk] = (int)(j * jst + lane4);
                else ab1[kk - 32] = (int)(j * jst + lane4);
            }
        }
        int abd = (int)lane4;                                           // the dummy bin's address: row 0
        v32i acc0;
        v16i acc1;
#pragma unroll
        for (int k = 0; k < 32; ++k) acc0[k] = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) acc1[k] = 0;
        int accd = 0;
        const long wl = (long)slice * SP_GW + wv;                       // wave slot within this launch
        const u32* nbp = NB + (ws0 + wl) * N;
        const u32* lst = LIST + BASE[wl] * SP_BLOCK;
        const u32 off64 = (u32)lane * 4u;
        const unsigned long long lo = 0xffffffffull, hi = 0xffffffff00000000ull;
        u32 ad0, ad1, ad2, ad3, r0, r1, r2, r3;
        // s20:21 next block of items to request, s22:23 NB pointer, s24 batches left in this phase, s25 phases left, s26 / s27
        // NB of the next two phases, s28 lane of the next item in v123 (v124: the block after it), s29-31, s36-51 scratch
        asm volatile(
            "s_mov_b64 s[20:21], %[lst]\n"
            "s_mov_b64 s[22:23], %[nbp]\n"
            "s_mov_b32 s25, %[n]\n"
            "global_load_dword v123, %[off64], s[20:21]\n"
            "global_load_dword v124, %[off64], s[20:21] offset:256\n"
            "s_add_u32 s20, s20, 512\n s_addc_u32 s21, s21, 0\n"
            "s_load_dword s24, s[22:23], 0x0\n"
            "s_load_dword s26, s[22:23], 0x4\n"
            "s_load_dword s27, s[22:23], 0x8\n"
            "s_add_u32 s22, s22, 12\n s_addc_u32 s23, s23, 0\n"
            "s_mov_b32 s28, 0\n"
            "s_waitcnt vmcnt(1) lgkmcnt(0)\n"
            "s_barrier\n"
            "LSP_P0_%=:\n" SP_BATCH_BODY("0") SP_BATCH_TAIL("LSP_P0", "0") SP_PHASE_END("LSP_P1")
            "LSP_P1_%=:\n" SP_BATCH_BODY("%[buf1]") SP_BATCH_TAIL("LSP_P1", "1") SP_PHASE_END("LSP_P2")
            "LSP_P2_%=:\n" SP_BATCH_BODY("%[buf2]") SP_BATCH_TAIL("LSP_P2", "2") SP_PHASE_END("LSP_P0")
            "LSP_DONE_%=:\n"
            "s_waitcnt vmcnt(0) lgkmcnt(0)\n"
            : "+{v[24:55]}"(acc0), "+{v[56:71]}"(acc1), "+{v72}"(accd), [ad0] "=&v"(ad0), [ad1] "=&v"(ad1), [ad2] "=&v"(ad2), [ad3] "=&v"(ad3),
              [r0] "=&v"(r0), [r1] "=&v"(r1), [r2] "=&v"(r2), [r3] "=&v"(r3)
            : "{v[74:105]}"(ab0), "{v[106:121]}"(ab1), "{v122}"(abd), [lst] "s"(lst), [nbp] "s"(nbp), [n] "s"(N), [off64] "v"(off64), [lo] "s"(lo),
              [hi] "s"(hi), [buf1] "n"(BUF1), [buf2] "n"(BUF2_IMM)
            : "memory", "scc", "v123", "v124", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s36", "s37",
              "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51");
        __builtin_amdgcn_s_barrier();                                   // the loader has drained its last (unused) requests

        // Epilogue (as in k_s3_score_bl): the 32 lanes of a half wave hold one bin's sums for 32 biosamples; with the
        // per-biosample base B1[b][x_b] added they are summed per (bin, state) in LDS -- the slab buffers are free, the last
        // barrier is behind every wave, a wave only touches its own 2 x 16 x S cells -- and the non-zero cells go out, one
        // address per lane.  Integer adds: exact, and independent of the order.
        long long* red = reinterpret_cast<long long*>(tab) + (long)wv * (2 * SP_EB * S);
        const bool live = b < N;
        const long long* b1 = B1 + (long)bl * S;
#pragma unroll
        for (int g = 0; g < SP_K / SP_EB; ++g) {
            for (int e = lane; e < 2 * SP_EB * S; e += 64) red[e] = 0;
            const uint4 v = *reinterpret_cast<const uint4*>(pb + 16 * g);
            const u32 w[4] = {v.x, v.y, v.z, v.w};
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
            for (int u = 0; u < SP_EB; ++u) {
                const u32 j = ((w[u >> 2] >> (8 * (u & 3))) & 0xffu) >> 2;
                const int kk = SP_EB * g + u;
                const int av = kk < 32 ? acc0[kk & 31] : acc1[kk & 15];
                if (live && j < (u32)S)
                    __hip_atomic_fetch_add(&red[(h * SP_EB + u) * S + j], (long long)av + b1[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            for (int e = lane; e < 2 * SP_EB * S; e += 64) {
                const long long val = red[e];
                const int hb = e / S;                                    // half * SP_EB + bin of the round
                const long row = slice0 + wv * SP_WB + (hb / SP_EB) * SP_K + SP_EB * g + (hb % SP_EB);
                if (val != 0 && row < R) atomicAdd(&cells[row * S + (e - hb * S)], (u64)val);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
    }
}

// fixed-point cells -> float64 in place and / or float32
__global__ __launch_bounds__(256) void k_sp_unit_finish(double* __restrict__ cells, long n, int want64, float* __restrict__ out32,
                                                        const double* __restrict__ scal) {
    const double unit = scal[0];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const double v = (double)reinterpret_cast<const long long*>(cells)[i] * unit;
        if (want64) cells[i] = v;
        if (out32) out32[i] = (float)v;
    }
}

bool s3_sparse_ok(int N, int S) { return S >= 2 && S <= SP_SMAX && N >= 2; }

struct SpLayout {
    int64_t td, xt, cells, b1, nb, tot, base, misc, list_off, total_fixed;
    long Rp, nws;
};
static SpLayout sp_layout(int64_t R, int N, int S, bool own_cells) {
    SpLayout L;
    L.Rp = align_up(R, SP_SLICE);
    L.nws = L.Rp / SP_WB;
    int64_t o = 0;
    L.misc = o; o += 1024;                                               // unit, 1/unit, maxbits, state counts
    L.td = o; o += align_up((int64_t)sp_nchunk(N) * N * sp_chb(S), 256);
    L.b1 = o; o += align_up((int64_t)N * S * 8, 256);
    L.xt = o; o += align_up((int64_t)N * L.Rp + 64, 256);
    L.cells = o; o += own_cells ? align_up(R * S * 8, 256) : 0;
    L.nb = o; o += align_up(L.nws * (int64_t)(N + 4) * 4, 256);
    L.tot = o; o += align_up(L.nws * 4, 256);
    L.base = o; o += align_up(L.nws * 8, 256);
    L.list_off = o;
    L.total_fixed = o;
    return L;
}
static inline int64_t sp_min_list_bytes(int N) { return ((int64_t)N * (SP_K / SP_BATCH) * SP_BATCH / SP_BLOCK + 3) * SP_BLOCK * 4 * SP_GW; }   // one slice, every bin live
// what the kernel wants on top of its fixed buffers: room for the lists of up to 512 K bins with a third of the cells off
// the modal state (fewer bins per launch otherwise)
int64_t s3_sparse_ws_bytes(int64_t R, int N, int S) {
    const SpLayout L = sp_layout(R, N, S, true);
    const int64_t nws = L.nws < 5462 ? L.nws : 5462;
    int64_t lists = align_up(nws * ((int64_t)N * (SP_K / 3 / SP_BATCH + 1) * SP_BATCH + 3 * SP_BLOCK) * 4, 256);
    if (lists < sp_min_list_bytes(N)) lists = sp_min_list_bytes(N);
    return L.total_fixed + lists;
}

// EPG_S3_SCORE: 'l' = the dense biosample-lane kernel, 's' = this kernel whatever the state frequencies are, default: this
// kernel when at least 45 % of the cells are in one state.  Returns 1 when the caller should run the dense kernel.
int score_s3_sparse(const int8_t* X8, int64_t R, int32_t N, int64_t ldx, int32_t S, const float* q, double* out64, float* out32, void* ws,
                    int64_t ws_bytes, hipStream_t st, bool forced) {
    const SpLayout L = sp_layout(R, N, S, out64 == nullptr);
    if (ws_bytes < L.total_fixed + sp_min_list_bytes(N))
        return fail(EPG_ERR_WORKSPACE, "score_s3: workspace %lld < %lld bytes", (long long)ws_bytes, (long long)s3_sparse_ws_bytes(R, N, S));
    char* base = reinterpret_cast<char*>(ws);
    double* scal = reinterpret_cast<double*>(base + L.misc);
    u32* maxbits = reinterpret_cast<u32*>(base + L.misc + 64);
    int64_t* scnt = reinterpret_cast<int64_t*>(base + L.misc + 128);       // S <= 31 state counts
    int* TD = reinterpret_cast<int*>(base + L.td);
    long long* B1 = reinterpret_cast<long long*>(base + L.b1);
    char* XT = base + L.xt;
    double* acc = out64 ? out64 : reinterpret_cast<double*>(base + L.cells);
    u32* NB = reinterpret_cast<u32*>(base + L.nb);
    u32* TOT = reinterpret_cast<u32*>(base + L.tot);
    unsigned long long* BASE = reinterpret_cast<unsigned long long*>(base + L.base);
    u32* LIST = reinterpret_cast<u32*>(base + L.list_off);
    const int64_t list_cap = (ws_bytes - L.list_off) / (SP_BLOCK * 4);    // blocks of 64 items

    // the modal state of THIS matrix (any state gives the same integers; the most frequent one the fewest gathers)
    EPG_HIP(hipMemsetAsync(scnt, 0, 32 * 8, st));
    int rc = bin_hist_impl(X8, R, N, ldx, S, nullptr, scnt, st);
    if (rc) return rc;
    int64_t hc[32];
    EPG_HIP(hipMemcpyAsync(hc, scnt, (size_t)S * 8, hipMemcpyDeviceToHost, st));
    EPG_HIP(hipStreamSynchronize(st));
    int m = 0;
    int64_t tot = 0;
    for (int s = 0; s < S; ++s) { tot += hc[s]; if (hc[s] > hc[m]) m = s; }
    // a byte that is not a state is counted nowhere: the identity above takes every biosample to be in SOME state (one outside
    // the lists counts as modal), so a matrix with such bytes goes to the dense kernel, where they add nothing
    if (tot != (int64_t)R * N) return 1;
    if (!forced && (double)hc[m] < 0.45 * (double)tot) return 1;                    // the caller runs the dense kernel

    const int chb = sp_chb(S), nchunk = sp_nchunk(N);
    const int bias2 = 2 * chb > SP_BUF2_IMM ? (2 * chb - SP_BUF2_IMM) / 128 : 0;
    const int grid_cap = num_cus() * 16;
    EPG_HIP(hipMemsetAsync(maxbits, 0, 4, st));
    {
        const long total = (long)N * N * S * S;
        long blocks = (total + 255) / 256;
        if (blocks > grid_cap) blocks = grid_cap;
        hipLaunchKernelGGL(k_sp_tmax, dim3((unsigned)blocks), dim3(256), 0, st, q, N, S, maxbits);
        EPG_LAUNCH_CHECK("k_sp_tmax");
        hipLaunchKernelGGL(k_sp_unit, dim3(1), dim3(1), 0, st, maxbits, N, scal);
        EPG_LAUNCH_CHECK("k_sp_unit");
        const long words = (long)nchunk * N * (chb / 4);
        blocks = (words + 255) / 256;
        if (blocks > grid_cap * 4L) blocks = grid_cap * 4L;
        hipLaunchKernelGGL(k_sp_td_build, dim3((unsigned)blocks), dim3(256), 0, st, q, N, S, m, chb / 4, nchunk, scal, TD);
        EPG_LAUNCH_CHECK("k_sp_td_build");
        hipLaunchKernelGGL(k_sp_base, dim3((unsigned)((N * S + 255) / 256)), dim3(256), 0, st, q, N, S, m, scal, B1);
        EPG_LAUNCH_CHECK("k_sp_base");
    }
    rc = transpose_states_bad(reinterpret_cast<const char*>(X8), R, N, ldx, S, XT, L.Rp, 2, S, st);   // bytes = 4 * state, 4 * S = "not a state"
    if (rc) return rc;
    EPG_HIP(hipMemsetAsync(acc, 0, (size_t)R * S * 8, st));
    hipLaunchKernelGGL(k_sp_count, dim3((unsigned)L.nws), dim3(256), 0, st, XT, L.Rp, N, S, m, NB, TOT);
    EPG_LAUNCH_CHECK("k_sp_count");
    std::vector<u32> tot_h((size_t)L.nws);
    EPG_HIP(hipMemcpyAsync(tot_h.data(), TOT, (size_t)L.nws * 4, hipMemcpyDeviceToHost, st));
    EPG_HIP(hipStreamSynchronize(st));

    // launches of whole slices whose lists fit the workspace; BASE[] = first block of each wave slot within its launch's list
    // (two blocks of slack per slot: the wave requests two blocks past the last one it uses)
    const long nslices_all = L.Rp / SP_SLICE;
    std::vector<unsigned long long> base_h((size_t)L.nws);
    std::vector<long> launch_first;
    {
        long s0 = 0;
        while (s0 < nslices_all) {
            unsigned long long used = 0;
            long s1 = s0;
            while (s1 < nslices_all) {
                unsigned long long need = 0;
                for (int w = 0; w < SP_GW; ++w)
                    need += ((unsigned long long)tot_h[(size_t)(s1 * SP_GW + w)] * SP_BATCH + SP_BLOCK - 1) / SP_BLOCK + 2;
                if (used + need > (unsigned long long)list_cap) {
                    if (s1 > s0) break;
                    return fail(EPG_ERR_WORKSPACE, "score_s3: workspace too small for the item lists of one slice");
                }
                for (int w = 0; w < SP_GW; ++w) {
                    base_h[(size_t)(s1 * SP_GW + w)] = used;
                    used += ((unsigned long long)tot_h[(size_t)(s1 * SP_GW + w)] * SP_BATCH + SP_BLOCK - 1) / SP_BLOCK + 2;
                }
                ++s1;
            }
            launch_first.push_back(s0);
            s0 = s1;
        }
        launch_first.push_back(nslices_all);
    }
    EPG_HIP(hipMemcpyAsync(BASE, base_h.data(), (size_t)L.nws * 8, hipMemcpyHostToDevice, st));
    EPG_HIP(hipStreamSynchronize(st));                                  // base_h goes out of scope: the copy must have read it

    const size_t red = (size_t)SP_GW * 2 * SP_EB * S * 8;
    static const int ahead = [] { const char* e = getenv("EPG_S3_AHEAD"); return e ? atoi(e) : 6; }();   // phases between touch and use
    const int npieces = chb >> 10;
    for (size_t ci = 0; ci + 1 < launch_first.size(); ++ci) {
        const long sl0 = launch_first[ci], nsl = launch_first[ci + 1] - sl0;
        const long ws0 = sl0 * SP_GW;
        hipLaunchKernelGGL(k_sp_write, dim3((unsigned)(nsl * SP_GW)), dim3(256), 0, st, XT, L.Rp, N, S, m, ws0, NB, BASE + ws0, LIST, bias2);
        EPG_LAUNCH_CHECK("k_sp_write");
        if (nsl * nchunk > 0x7fffffffL) return fail(EPG_ERR_UNSUPPORTED, "score_s3: R*N too large for one call");
        const char* XTc = XT + sl0 * SP_SLICE;                            // the launch's first bin; rows stay Rp apart
        u64* cellc = reinterpret_cast<u64*>(acc) + sl0 * SP_SLICE * S;
        const long Rc = R - sl0 * SP_SLICE;                               // rows from the launch's first bin to the end of the matrix
#define SP_LAUNCH(NP, B1OFF)                                                                                                     \
    do {                                                                                                                         \
        size_t shmem = (size_t)2 * (B1OFF) + chb;                                                                                \
        if (shmem < red) shmem = red;                                                                                            \
        EPG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_s3_score_sp<NP, B1OFF, 2 * (B1OFF)>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem)); \
        hipLaunchKernelGGL((k_s3_score_sp<NP, B1OFF, 2 * (B1OFF)>), dim3((unsigned)(nsl * nchunk)), dim3(SP_THREADS), shmem, st, XTc, L.Rp, Rc, N, S, chb, \
                           reinterpret_cast<const char*>(TD), (int)nsl, NB, BASE + ws0, LIST, ws0, B1, cellc, ahead);           \
    } while (0)
        // the second and third buffer start at chb and 2 chb (one instantiation per slab size of the supported state models)
        switch (npieces) {
            case 1: SP_LAUNCH(1, 1024); break;     case 2: SP_LAUNCH(2, 2048); break;     case 3: SP_LAUNCH(3, 3072); break;
            case 4: SP_LAUNCH(4, 4096); break;     case 5: SP_LAUNCH(5, 5120); break;     case 6: SP_LAUNCH(6, 6144); break;
            case 7: SP_LAUNCH(7, 7168); break;     case 9: SP_LAUNCH(9, 9216); break;     case 12: SP_LAUNCH(12, 12288); break;
            case 14: SP_LAUNCH(14, 14336); break;  case 17: SP_LAUNCH(17, 17408); break;  case 20: SP_LAUNCH(20, 20480); break;
            case 23: SP_LAUNCH(23, 23552); break;  case 27: SP_LAUNCH(27, 27648); break;  case 30: SP_LAUNCH(30, 30720); break;
            case 34: SP_LAUNCH(34, 34816); break;  case 39: SP_LAUNCH(39, 39936); break;  case 43: SP_LAUNCH(43, 44032); break;
            default: return fail(EPG_ERR_UNSUPPORTED, "score_s3: no modal-state kernel for S=%d", S);
        }
#undef SP_LAUNCH
        EPG_LAUNCH_CHECK("k_s3_score_sp");
    }
    {
        long blocks = ((long)R * S + 255) / 256;
        if (blocks > num_cus() * 8L) blocks = num_cus() * 8L;
        hipLaunchKernelGGL(k_sp_unit_finish, dim3((unsigned)blocks), dim3(256), 0, st, acc, (long)R * S, out64 ? 1 : 0, out32, scal);
        EPG_LAUNCH_CHECK("k_sp_unit_finish");
    }
    return EPG_OK;
}

}  // namespace epg
