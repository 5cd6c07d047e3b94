// S3 score, biosample-lane form (round 2).  gfx950 only.
//
// Reference: scores.py:455-506 s3Score.  score[bin, x_b] += sum_{a != b} T[a, b, x_a, x_b]  with
// T = kl(float32(1)/P, q) in float32 (scores.py:479-480) -- P = N (N - 1) table gathers per bin.
//
// k_s3_score (epg_s3.hip) gives a lane 16 bins of ONE biosample b: the 64 lanes of a gather then want up to 14 different
// (x_a, x_b) cells of one 18 x 18 table and collide in the 32 LDS banks (PMC: ~45 % of the LDS cycles); it measures 2.3 ns
// of a CU per 64 terms.  tools/ubench/lds_rate.hip: a CU retires one conflict-free ds_read_b32 of a wave per 1.12 ns, with an
// sdwa add in front and an add behind it in the same time.  That is the floor of any gather kernel, and this one is built
// to sit on it:
//
//  * a lane is a BIOSAMPLE b (32 consecutive b per half wave), a half wave is one bin at a time.  x_a is then the same for
//    the 32 lanes of a gather and only x_b differs, and with the table chunk of (a, 32 b) laid out [x_b][b][x_a] with an ODD
//    number SI >= S + 1 of dwords per (x_b, b) the bank of lane l is (SI * l + x_a) mod 32: a bijection of l, whatever the
//    states are.  Every gather is conflict free by construction (PMC: SQ_LDS_BANK_CONFLICT is 2 % of
//    SQ_LDS_IDX_ACTIVE, the epilogue's same-address adds; 45 % in k_s3_score).
//  * the table is stored as 32-bit FIXED POINT in units of max|T| * N / 2^30 (2^31 in round 2; see k_s3_tq_unit): the N terms of one (bin, b) add up in a plain
//    int32 without overflow, integer adds are exact and commute (bit-identical scores for any launch geometry, like the
//    64-bit cells they are added to), and the only error is the rounding of a table entry, <= unit / 2 ~ 7e-12 absolute at
//    N = 833 against terms of 1e-6 .. 3e-5, where the float32 partial sums of k_s3_score gave 1e-7 of a score.  (16-bit
//    codes were built too -- half the bytes to stream, room for a ring of three buffers, 78 ms --: real tables are made of
//    few distinct values, log2(R / small count), so their rounding errors do not average out; 4e-6 of a score at N = 833.)
//  * per gather: v_add_u32_sdwa (a 16-bit half of the packed, loop-invariant x_b offsets + a byte of 4 x_a), ds_read_b32 with
//    the buffer in the immediate offset, v_add_u32 into the bin's accumulator.  48 bins per half wave: 24 VGPRs of x_b
//    offsets, 48 accumulators, 12 of x_a bytes (each dwordx4 is re-requested for the next a as soon as its 16 gathers are
//    issued), 16 for two blocks of eight gathers in flight.
//  * the chunk of a + 1 -- 18 x 32 x 19 dwords = 42.75 KiB, contiguous in the pre-arranged table -- goes from L2 straight
//    into the other LDS buffer by global_load_lds_dwordx4 while the gathers of a run, issued by a LOADER wave (the
//    sixteenth wave of the workgroup; the other fifteen gather): vmcnt counts in order, so when the gather waves issued
//    the table loads themselves every wait for state bytes was also a wait for the table (106 ms per 1 M bins).  One raw
//    barrier per a.  Workgroups are ordered with the bin slice fastest, so the ~256 resident ones walk the same chunks and
//    share them in L2 (PMC: 91 % hits, 27 GB from memory per 1 M bins).
//  * epilogue: the 32 lanes of a half wave hold one bin's sums for 32 biosamples; they are added up per (bin, state) in LDS
//    first (see there).
//  Measured at N = 833, S = 18, 1 M bins: 76.3 ms (k_s3_score 98.4).  Per a and workgroup the LDS pipe is busy 0.81 us with
//  gathers and 0.16 us with the 43 KiB the loader streams in, a barrier costs ~0.1 us, and a phase takes 1.19 us: the
//  kernel is LDS bound (EPG_S3_SCORE_DBG: gathers alone 58 ms, the table stream alone 38 ms, barriers and epilogue alone 9 ms).
#include "epg_common.h"

#include <stdlib.h>

namespace epg {

int transpose_states_bad(const char* X, int64_t R, int32_t N, int64_t ldx, int32_t S, char* XT, int64_t Rp, int shift, int bad, hipStream_t st);

constexpr int BL_GW = 15;                          // gather waves; wave BL_GW of the workgroup is the loader
constexpr int BL_THREADS = 64 * (BL_GW + 1);
constexpr int BL_BW = 48;                          // bins of a half wave
constexpr int BL_SLICE = BL_GW * 2 * BL_BW;        // bins of a workgroup (1440)
constexpr int BL_BUF1 = 65024;                     // byte offset of the second LDS buffer: the largest 512-multiple in a ds_read immediate
constexpr int BL_SMAX = 20;                        // 20 * 32 * 21 * 4 = 53760 bytes per chunk: 53 table loads in flight (vmcnt has six bits)
constexpr int BL_EB = 16;                          // bins of a half wave per round of the epilogue

static inline int bl_si(int S) { return (S + 1) | 1; }                                       // dwords per (x_b, b): odd, >= S + 1
static inline int bl_chb(int S) { return (int)align_up((int64_t)S * 32 * bl_si(S) * 4, 1024); }   // bytes of one (chunk, a): whole wave loads
static inline int bl_nchunk(int N) { return (N + 31) / 32; }

__device__ __forceinline__ float s3_t(float qv, float obs) { return s3_table_entry(qv, obs); }   // (epg_common.h)

// max |T| over the off-diagonal table (bits of a non-negative float order like unsigned integers)
__global__ __launch_bounds__(256) void k_s3_tq_max(const float* __restrict__ q, int N, int S, u32* __restrict__ maxbits) {
    const long SS = (long)S * S, total = (long)N * N * SS;
    const float obs = 1.0f / (float)((long)N * (N - 1));
    u32 m = 0;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long ab = e / SS;
        if (ab / N == ab % N) continue;
        const u32 bits = __float_as_uint(fabsf(s3_t(q[e], obs)));
        m = bits > m ? bits : m;
    }
    for (int o = 32; o; o >>= 1) {
        const u32 t = (u32)__shfl_xor((int)m, o);
        m = t > m ? t : m;
    }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(maxbits, m);
}

// scal[0] = unit (score per fixed-point step), scal[1] = 1 / unit.  max|T| N / 2^30 (round 3 halved the range for the modal-state
// kernel, whose table held DIFFERENCES of two entries; that kernel was deleted in round 5, the unit stays: scores are bit-identical
// to every earlier run)
__global__ void k_s3_tq_unit(const u32* __restrict__ maxbits, int N, double* __restrict__ scal) {
    const double mx = (double)__uint_as_float(*maxbits);
    const double unit = mx * (double)N / 1073741824.0;
    scal[0] = unit;
    scal[1] = unit > 0.0 ? 1.0 / unit : 0.0;
}

// TQ[c][a][x_b = j][l][x_a = i] = rint(T[a, 32 c + l, i, j] / unit) as int32; zero for i >= S (the column "not a state"
// reads), b >= N, a == b and in the padding of a chunk
__global__ __launch_bounds__(256) void k_s3_tq_build(const float* __restrict__ q, int N, int S, int SI, int chw /*dwords per (c, a)*/,
                                                    int nchunk, const double* __restrict__ scal, int* __restrict__ TQ) {
    const long total = (long)nchunk * N * chw;
    const float obs = 1.0f / (float)((long)N * (N - 1));
    const double inv = scal[1];
    const int rowj = 32 * SI;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long ca = e / chw;
        const int wd = (int)(e - ca * chw);
        const int c = (int)(ca / N), a = (int)(ca - (long)c * N);
        const int j = wd / rowj, rem = wd - j * rowj, l = rem / SI, i = rem - l * SI;
        const int b = 32 * c + l;
        int v = 0;
        if (j < S && i < S && b < N && b != a) v = (int)__double2ll_rn((double)s3_t(q[(((long)a * N + b) * S + i) * S + j], obs) * inv);
        TQ[e] = v;
    }
}

__device__ __forceinline__ u32 sel_byte(const uint4& v, int u) {      // byte u of 16, u a compile-time constant after unrolling
    const u32 w = (u >> 2) == 0 ? v.x : (u >> 2) == 1 ? v.y : (u >> 2) == 2 ? v.z : v.w;
    return (w >> (8 * (u & 3))) & 0xffu;
}

// Eight gathers, issue half: address = (16-bit half of the packed x_b offsets) + (byte of a word of 4 x_a) in one
// v_add_u32_sdwa, then ds_read_b32 with the buffer in the immediate offset (the dynamic LDS segment starts at address 0:
// the kernel has no static __shared__).  Written as asm because the compiler (a) hoists the loop-invariant half-word
// extraction out of the a loop -- twice the registers, and the kernel spills --, (b) adds the (zero) LDS base to every
// address with a VALU instruction and (c) under this register pressure waits for every gather before issuing the next.
#define BL_A1(T, XB, HW, XW, BY) "v_add_u32_sdwa %" #T ", %" #XB ", %" #XW " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_" #HW " src1_sel:BYTE_" #BY "\n"
#define BL_R1(T) "ds_read_b32 %" #T ", %" #T " offset:%14\n"
template <int BUF>
__device__ __forceinline__ void bl_issue8(u32 (&t)[8], u32 xb0, u32 xb1, u32 xb2, u32 xb3, u32 xw0, u32 xw1) {
    asm volatile(BL_A1(0, 8, 0, 12, 0) BL_A1(1, 8, 1, 12, 1) BL_A1(2, 9, 0, 12, 2) BL_A1(3, 9, 1, 12, 3)
                 BL_A1(4, 10, 0, 13, 0) BL_A1(5, 10, 1, 13, 1) BL_A1(6, 11, 0, 13, 2) BL_A1(7, 11, 1, 13, 3)
                 BL_R1(0) BL_R1(1) BL_R1(2) BL_R1(3) BL_R1(4) BL_R1(5) BL_R1(6) BL_R1(7)
                 : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]), "=&v"(t[6]), "=&v"(t[7])
                 : "v"(xb0), "v"(xb1), "v"(xb2), "v"(xb3), "v"(xw0), "v"(xw1), "n"(BUF)
                 : "memory");
}
// consume half of a block: wait until at most PENDING younger LDS reads are outstanding (they return in order), then
// accumulate four.  The lgkm counter has four bits, so a wave never has more than 12 gathers in flight here.
template <int PENDING, int H>
__device__ __forceinline__ void bl_consume4(int* acc, const u32 (&t)[8]) {
    asm volatile("s_waitcnt lgkmcnt(%8)\n"
                 "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %5\n v_add_u32 %2, %2, %6\n v_add_u32 %3, %3, %7\n"
                 : "+v"(acc[4 * H]), "+v"(acc[4 * H + 1]), "+v"(acc[4 * H + 2]), "+v"(acc[4 * H + 3])
                 : "v"(t[4 * H]), "v"(t[4 * H + 1]), "v"(t[4 * H + 2]), "v"(t[4 * H + 3]), "n"(PENDING)
                 : "memory");
}
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
// 16 state bytes of the next a, scalar base + 32-bit lane offset, and the waits for them, by hand: the three loads of a
// phase complete in order, so "xa[g] has arrived" is always "at most two younger loads outstanding"
template <int OFF>
__device__ __forceinline__ void bl_load_x(u32x4& dst, const char* base, u32 voff) {
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(base), "n"(OFF));
}
template <int N>
__device__ __forceinline__ void bl_wait_vm(u32x4& x) {        // "+v": nothing that reads x moves above the wait
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(x) : "n"(N));
}

// block B of eight bins (of BL_BW / 8): its four registers of x_b offsets and its two words of x_a bytes
template <int BUF, int B>
__device__ __forceinline__ void bl_issue(u32 (&t)[8], const u32 (&xbo)[BL_BW / 2], const u32x4 (&xa)[BL_BW / 16]) {
    bl_issue8<BUF>(t, xbo[4 * B], xbo[4 * B + 1], xbo[4 * B + 2], xbo[4 * B + 3], xa[B / 2][(2 * B) & 3], xa[B / 2][(2 * B + 1) & 3]);
}

// one a of a gather wave: 48 gathers per half wave from LDS buffer PH, software-pipelined by hand over two sets of eight
// registers, and the state bytes of the next a into the registers of this one as soon as their gathers are issued.
// The pipeline runs THROUGH the barrier: the last four gathers of a phase (t1[4..7]) are still in flight when the wave
// arrives, and are accumulated after the first block of the next phase has been issued -- with the LDS queue drained and
// refilled at every barrier a phase cost ~150 cycles more.  (Those four reads were queued before the barrier; what the
// loader sends into their buffer afterwards is a round trip to L2 behind them.)
template <int PH>
__device__ __forceinline__ void bl_phase(const char* __restrict__ xs_next, u32 xoff, const u32 (&xbo)[BL_BW / 2], u32x4 (&xa)[BL_BW / 16],
                                         int (&acc)[BL_BW], u32 (&t1)[8], int dbg) {
    constexpr int BUF = PH * BL_BUF1;
    static_assert(BL_BW == 48, "six blocks of eight bins, three dwordx4 of states");
    u32 t0[8];
    if (dbg & 4) {                                                  // measurement: no gathers, no state loads
        __builtin_amdgcn_s_barrier();
        return;
    }
    bl_wait_vm<2>(xa[0]);
    bl_issue<BUF, 0>(t0, xbo, xa);
    bl_consume4<8, 1>(acc + 40, t1);                                // the previous phase's last four (zeros before the first)
    bl_consume4<4, 0>(acc, t0);
    bl_issue<BUF, 1>(t1, xbo, xa);
    bl_load_x<0>(xa[0], xs_next, xoff);                             // the same 16 bins of the next a
    bl_consume4<8, 1>(acc, t0);
    bl_consume4<4, 0>(acc + 8, t1);
    bl_wait_vm<2>(xa[1]);
    bl_issue<BUF, 2>(t0, xbo, xa);
    bl_consume4<8, 1>(acc + 8, t1);
    bl_consume4<4, 0>(acc + 16, t0);
    bl_issue<BUF, 3>(t1, xbo, xa);
    bl_load_x<16>(xa[1], xs_next, xoff);
    bl_consume4<8, 1>(acc + 16, t0);
    bl_consume4<4, 0>(acc + 24, t1);
    bl_wait_vm<2>(xa[2]);
    bl_issue<BUF, 4>(t0, xbo, xa);
    bl_consume4<8, 1>(acc + 24, t1);
    bl_consume4<4, 0>(acc + 32, t0);
    bl_issue<BUF, 5>(t1, xbo, xa);
    bl_load_x<32>(xa[2], xs_next, xoff);
    bl_consume4<8, 1>(acc + 32, t0);
    bl_consume4<4, 0>(acc + 40, t1);
    __builtin_amdgcn_s_barrier();
}

// The loader's side of a phase: one chunk -> an LDS buffer, NP loads of 1 KiB whatever the chunk size (pieces past the end
// repeat the last one).  A chunk has ONE phase to land: there are two buffers, because the span of three is out of reach of
// the 16-bit ds_read immediate.  That is enough when its lines are in L2 and not when they come from memory (9 % of the
// requests did, and the kernel ran at the pace of the slowest of a chunk's 43 loads: 1.31 us per a with 0.81 us of gathers).
// So ONE workgroup per XCD and chunk -- the one whose rank among the workgroups of its XCD equals the chunk index mod 32 --
// touches one dword of every 128-byte line of the chunk `ahead` phases before it is due: the 32 workgroups that share an L2
// walk the chunks together, so all of them then find it there.  (Every workgroup touching every chunk two phases ahead
// doubled the L2 traffic and was slower than no touch at all.)
template <int NP>
__device__ __forceinline__ void bl_request(const char* __restrict__ tq_a, u32 loff, int npieces, char* dst) {
#pragma unroll
    for (int r = 0; r < NP; ++r) {
        const int p = r < npieces ? r : npieces - 1;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tq_a + p * 1024 + loff),
                                         (__attribute__((address_space(3))) void*)(dst + p * 1024), 16, 0, 0);
    }
}
template <int NPF>
__device__ __forceinline__ void bl_touch(u32 (&sink)[NPF], const char* __restrict__ tq_a, u32 toff, u32 tmax) {
#pragma unroll
    for (int r = 0; r < NPF; ++r) {
        u32 o = toff + (u32)r * 8192u;
        o = o < tmax ? o : tmax;
        // "+v": the register stays this variable's from the first touch to the final wait -- as a plain output the compiler
        // may hand the register of the PREVIOUS touch, whose load is still in flight, to something else
        asm volatile("global_load_dword %0, %1, %2" : "+v"(sink[r]) : "v"(o), "s"(tq_a));
    }
}

template <int NP>
__global__ __launch_bounds__(BL_THREADS) void k_s3_score_bl(const char* __restrict__ XT4, long Rp, long R, int N, int S, int SI, int chb,
                                                            const char* __restrict__ TQ, int nslices, u64* __restrict__ cells, int ahead,
                                                            int dbg) {
    extern __shared__ __attribute__((aligned(1024))) char tab[];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = blockIdx.x / nslices;
    const long slice0 = (long)(blockIdx.x - c * nslices) * BL_SLICE;
    // the gathers address LDS by absolute offsets (immediates 0 and BL_BUF1): the dynamic segment must start at 0
    if ((u32)(size_t)(__attribute__((address_space(3))) char*)tab != 0u) __builtin_trap();

    if (wv == BL_GW) {
        // ---- loader wave, at wave priority 3 (round 5): it shares its SIMD with four gather waves whose instruction streams never
        // pause, and every one of its few instructions -- the LDS-DMA requests of chunk a + 1, the wait, the barrier all sixteen
        // waves meet at -- queued behind theirs: 77.4 -> 62.8 ms per 1 M bins at N = 833 (the "gathers alone" figure above is 58),
        // one line.  The same bits (tests/test_hip_s3_n833.py, test_hip_s3_stress.py).
        __builtin_amdgcn_s_setprio(3);
        constexpr int NPF = (NP + 7) / 8;
        const char* tq = TQ + (long)c * N * chb;
        const long tstride = (dbg & 1) ? 0 : chb;                        // measurement: every phase loads the chunk of a = 0
        const u32 loff = (u32)lane * 16u, toff = (u32)lane * 128u, tmax = (u32)chb - 128u;
        const int npieces = chb >> 10;
        const int rank = (int)(blockIdx.x >> 3) & 31;                    // workgroups are dealt to the 8 XCDs round robin
        u32 sink[NPF];
#pragma unroll
        for (int r = 0; r < NPF; ++r) sink[r] = 0;
        bl_request<NP>(tq, loff, npieces, tab);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int a = 0; a < N; ++a) {
            const int a1 = a + 1 < N ? a + 1 : N - 1;                    // past the end: a valid chunk nobody reads
            if (!(dbg & 2)) bl_request<NP>(tq + a1 * tstride, loff, npieces, tab + ((a + 1) & 1) * BL_BUF1);
            if (ahead > 0 && ((a + ahead) & 31) == rank && a + ahead < N) {
                bl_touch<NPF>(sink, tq + (a + ahead) * tstride, toff, tmax);   // younger than the request: it does not wait for them
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPF) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // chunk a + 1 has landed
            }
            __builtin_amdgcn_s_barrier();
        }
#pragma unroll
        for (int r = 0; r < NPF; ++r) asm volatile("s_waitcnt vmcnt(0)" : "+v"(sink[r])::"memory");
    } else {
        // ---- gather waves
        const int l = lane & 31, h = lane >> 5;
        const long bin0 = slice0 + wv * (2 * BL_BW) + h * BL_BW;        // first bin of this half wave; Rp is a multiple of BL_SLICE
        const int b = 32 * c + l, bl = b < N ? b : N - 1;
        // loop invariants: byte offset of (x_b, l) inside a chunk for each of the 48 bins, two per register
        u32 xbo[BL_BW / 2];
        {
            const char* pb = XT4 + (long)bl * Rp + bin0;
            const u32 lpart = (u32)l * SI * 4u, jstride = 32u * SI * 4u;
#pragma unroll
            for (int g = 0; g < BL_BW / 16; ++g) {
                const uint4 v = *reinterpret_cast<const uint4*>(pb + 16 * g);
#pragma unroll
                for (int u = 0; u < 16; u += 2) {
                    u32 j0 = sel_byte(v, u) >> 2, j1 = sel_byte(v, u + 1) >> 2;
                    j0 = j0 < (u32)S ? j0 : 0u;                          // "not a state": any valid row, dropped at the end
                    j1 = j1 < (u32)S ? j1 : 0u;
                    xbo[(16 * g + u) >> 1] = (j0 * jstride + lpart) | ((j1 * jstride + lpart) << 16);
                }
            }
        }
        int acc[BL_BW];
#pragma unroll
        for (int k = 0; k < BL_BW; ++k) acc[k] = 0;

        u32x4 xa[BL_BW / 16];
        const char* xs = XT4 + slice0 + wv * (2 * BL_BW);                // wave-uniform: the wave's 2 x 48 bins of biosample 0
        const u32 xoff = (u32)h * BL_BW;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // the x_b loads above: from here on vmcnt is counted by hand
        bl_load_x<0>(xa[0], xs, xoff);
        bl_load_x<16>(xa[1], xs, xoff);
        bl_load_x<32>(xa[2], xs, xoff);
        __builtin_amdgcn_s_barrier();

        u32 t1[8];                                                      // the gathers in flight across a barrier
#pragma unroll
        for (int k = 0; k < 8; ++k) t1[k] = 0;
        for (int a = 0; a < N; a += 2) {
            {
                const int an = a + 1 < N ? a + 1 : a;                   // past the end: valid addresses, results unused
                bl_phase<0>(xs + (long)an * Rp, xoff, xbo, xa, acc, t1, dbg);
            }
            if (a + 1 < N) {
                const int an = a + 2 < N ? a + 2 : a + 1;
                bl_phase<1>(xs + (long)an * Rp, xoff, xbo, xa, acc, t1, dbg);
            }
        }
        if (!(dbg & 4)) bl_consume4<0, 1>(acc + 40, t1);
        // the state loads of the clamped "next a" are still landing in xa: keep those registers until they have
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(xa[0]), "+v"(xa[1]), "+v"(xa[2])::"memory");

        // Epilogue.  The 32 lanes of a half wave hold one bin's sums for 32 biosamples, most of them in the same state: added
        // straight to the 64-bit cells they would be 32 atomics on one address, which the L2 serialises (measured: a third
        // of the kernel).  They are first added up per (bin, state) in LDS -- the table buffers are free, the last barrier is
        // behind every wave, and a wave only touches its own 2 x 16 x S cells -- and the non-zero cells go out, one address
        // per lane.  Integer adds: exact, and independent of the order.
        long long* red = reinterpret_cast<long long*>(tab) + (long)wv * (2 * BL_EB * S);
        const bool live = b < N;
        const char* pb = XT4 + (long)bl * Rp + bin0;
#pragma unroll
        for (int g = 0; g < BL_BW / BL_EB; ++g) {
            for (int e = lane; e < 2 * BL_EB * S; e += 64) red[e] = 0;
            const uint4 v = *reinterpret_cast<const uint4*>(pb + 16 * g);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
            for (int u = 0; u < BL_EB; ++u) {
                const u32 j = sel_byte(v, u) >> 2;
                if (live && j < (u32)S)
                    __hip_atomic_fetch_add(&red[(h * BL_EB + u) * S + j], (long long)acc[BL_EB * g + u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            for (int e = lane; e < 2 * BL_EB * S; e += 64) {
                const long long val = red[e];
                const int hb = e / S;                                    // half * BL_EB + bin of the round
                const long row = slice0 + wv * (2 * BL_BW) + (hb / BL_EB) * BL_BW + BL_EB * g + (hb % BL_EB);
                if (val != 0 && row < R) atomicAdd(&cells[row * S + (e - hb * S)], (u64)val);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
    }
}

// fixed-point cells -> float64 in place and / or float32
__global__ __launch_bounds__(256) void k_s3_unit_finish(double* __restrict__ cells, long n, int want64, float* __restrict__ out32,
                                                        const double* __restrict__ scal) {
    const double unit = scal[0];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const double v = (double)reinterpret_cast<const long long*>(cells)[i] * unit;
        if (want64) cells[i] = v;
        if (out32) out32[i] = (float)v;
    }
}

bool s3_lanes_ok(int N, int S) { return S >= 1 && S <= BL_SMAX && N >= 2; }
static int64_t bl_table_bytes(int N, int S) { return align_up((int64_t)bl_nchunk(N) * N * bl_chb(S), 256) + 256; }   // + maxbits, unit, 1/unit
static int64_t bl_xt_bytes(int64_t R, int N) { return align_up((int64_t)N * align_up(R, BL_SLICE) + 64, 256); }
int64_t s3_lanes_ws_bytes(int64_t R, int N, int S) { return bl_table_bytes(N, S) + bl_xt_bytes(R, N) + align_up(R * S * 8, 256); }

int score_s3_lanes(const int8_t* X8, int64_t R, int32_t N, int64_t ldx, int32_t S, const float* q, double* out64, float* out32, void* ws,
                   int64_t ws_bytes, hipStream_t st) {
    const int64_t tb = bl_table_bytes(N, S), xtb = bl_xt_bytes(R, N);
    const int64_t need = tb + xtb + (out64 ? 0 : align_up(R * S * 8, 256));
    if (ws_bytes < need) return fail(EPG_ERR_WORKSPACE, "score_s3: workspace %lld < %lld bytes", (long long)ws_bytes, (long long)need);
    char* base = reinterpret_cast<char*>(ws);
    int* TQ = reinterpret_cast<int*>(base);
    double* scal = reinterpret_cast<double*>(base + tb - 256);
    u32* maxbits = reinterpret_cast<u32*>(base + tb - 256 + 64);
    char* XT = base + tb;
    double* acc = out64 ? out64 : reinterpret_cast<double*>(base + tb + xtb);
    const int SI = bl_si(S), chb = bl_chb(S), nchunk = bl_nchunk(N);
    const int grid_cap = num_cus() * 16;

    EPG_HIP(hipMemsetAsync(maxbits, 0, 4, st));
    {
        const long total = (long)N * N * S * S;
        long blocks = (total + 255) / 256;
        if (blocks > grid_cap) blocks = grid_cap;
        hipLaunchKernelGGL(k_s3_tq_max, dim3((unsigned)blocks), dim3(256), 0, st, q, N, S, maxbits);
        EPG_LAUNCH_CHECK("k_s3_tq_max");
        hipLaunchKernelGGL(k_s3_tq_unit, dim3(1), dim3(1), 0, st, maxbits, N, scal);
        EPG_LAUNCH_CHECK("k_s3_tq_unit");
        const long words = (long)nchunk * N * (chb / 4);
        blocks = (words + 255) / 256;
        if (blocks > grid_cap * 4L) blocks = grid_cap * 4L;
        hipLaunchKernelGGL(k_s3_tq_build, dim3((unsigned)blocks), dim3(256), 0, st, q, N, S, SI, chb / 4, nchunk, scal, TQ);
        EPG_LAUNCH_CHECK("k_s3_tq_build");
    }
    const long Rp = align_up(R, BL_SLICE);
    int rc = transpose_states_bad(reinterpret_cast<const char*>(X8), R, N, ldx, S, XT, Rp, 2, S, st);   // bytes = 4 * state, 4 * S = "not a state"
    if (rc) return rc;
    EPG_HIP(hipMemsetAsync(acc, 0, (size_t)R * S * 8, st));
    const long nslices = Rp / BL_SLICE;
    if (nslices * nchunk > 0x7fffffffL) return fail(EPG_ERR_UNSUPPORTED, "score_s3: R*N too large for one call");
    size_t shmem = (size_t)BL_BUF1 + chb;
    const size_t red = (size_t)BL_GW * 2 * BL_EB * S * 8;
    if (shmem < red) shmem = red;
    // EPG_S3_SCORE_DBG (measurements only, results are wrong): 1 = every phase loads the same chunk, 2 = no table loads, 4 = no gathers
    static const int dbg = [] { const char* e = exp_env("EPG_S3_SCORE_DBG"); return e ? atoi(e) : 0; }();
    static const int ahead = [] { const char* e = exp_env("EPG_S3_AHEAD"); return e ? atoi(e) : 6; }();   // phases between touch and use
    const int npieces = chb >> 10;                                        // 1 .. 53 for S <= 20; 32 at S = 15, 43 at S = 18
#define BL_LAUNCH(NP)                                                                                                            \
    do {                                                                                                                         \
        EPG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_s3_score_bl<NP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem)); \
        hipLaunchKernelGGL(k_s3_score_bl<NP>, dim3((unsigned)(nslices * nchunk)), dim3(BL_THREADS), shmem, st, XT, Rp, (long)R, N, S, SI, chb, \
                           reinterpret_cast<const char*>(TQ), (int)nslices, reinterpret_cast<u64*>(acc), ahead, dbg);             \
    } while (0)
    // (The kernel's MEMORY side, round 6: FETCH_SIZE says 195-209 KB per bin on 8-15 M bins -- the 990 MB table is streamed once per
    // ~5000 bins, 3.2 TB/s over the launch: the workgroups that share an L2 do not stay within the ~90 phases of each other that
    // its 4 MB can bridge.  One launch per 728 slices (1.05 M bins), which puts them back in step every 73 rounds, cuts that to
    // 137 KB per bin and costs 1.3 % of time (924 against 912 ms per 15 M bins; groups of 364: 964 ms, of 2912: 910 ms) -- the gathers,
    // not the table stream, are what the kernel waits for, so it stays one launch.  profiles/r06f_s3_score_launch_groups.txt.)
    if (npieces <= 16) BL_LAUNCH(16);
    else if (npieces <= 32) BL_LAUNCH(32);
    else if (npieces <= 43) BL_LAUNCH(43);
    else BL_LAUNCH(53);
#undef BL_LAUNCH
    EPG_LAUNCH_CHECK("k_s3_score_bl");
    {
        long blocks = ((long)R * S + 255) / 256;
        if (blocks > num_cus() * 8L) blocks = num_cus() * 8L;
        hipLaunchKernelGGL(k_s3_unit_finish, dim3((unsigned)blocks), dim3(256), 0, st, acc, (long)R * S, out64 ? 1 : 0, out32, scal);
        EPG_LAUNCH_CHECK("k_s3_unit_finish");
    }
    return EPG_OK;
}

}  // namespace epg
