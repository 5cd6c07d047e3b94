// S3 expected pass as a pure fp4 matrix-core contraction of a PRECOMPUTED one-hot operand (gfx950).
//
// C[a,b,i,j] = #{bins : x[a] == i and x[b] == j} = E^T E for the one-hot expansion E[bin, (sample, state)]
// (reference expected.py:183-200).  Rounds 1-2 built the one-hot nibbles inside the contraction kernel, from state
// bytes (deleted in round 6), and measured that on a gfx950 SIMD the VALU time of that build and the MFMA time ADD UP (tools/ubench/
// mfma_valu.hip): 9 MFMAs of 64 bins cost ~400 cycles, the operand build ~350 more, and every operand is rebuilt by each of
// the ~157 workgroups that need it.  Here every operand is built ONCE per chunk of bins into HBM, already in the register
// layout of v_mfma_scale_f32_32x32x64_f8f6f4 (fp4 E2M1, 1.0 = 0b0010, unit block scales), and the contraction kernel is
// left with loads and matrix instructions:
//   k_s3_onehot_fp4   XT[sample][bin] -> E4[kstep][tile32][1 KiB]: tile = 32 consecutive (sample, state) rows x 64 bins,
//                     stored [half][row][16 B] = lane-linear for both the LDS-DMA that stages it and the ds_read_b128 that
//                     feeds it to the MFMA (7.5 KB per bin; a chunk of 256 K bins = 1.9 GB of workspace at N = 833).
//   k_s3_syrk_fp4     workgroup = 8 waves = 2 x 4 blocks of 96 rows (192 x 384 of C), wave = one 96 x 96 block pair (3 x 3
//                     MFMA tiles, 144 accumulator registers); per 128 bins the 18 + 18 operand tiles (36 KiB) are staged
//                     ONCE per workgroup by global_load_lds_dwordx4 (no registers, no VALU) into a 3-deep LDS ring, two
//                     stages ahead (counted vmcnt, raw s_barrier: one barrier per 18 MFMAs of a wave), and every wave reads
//                     its six tiles per 64 bins with ds_read_b128.  Only block pairs bm <= bn are computed; the epilogue
//                     writes the mirrored cells too.  Workgroups are ordered in 8 x 4 patches of the (bm-pair, bn-quad)
//                     grid and dealt to the XCDs in contiguous runs, so the 32 workgroups that share an L2 share operand
//                     panels.  Accumulation is float32 of exact integers (a chunk is < 2^24 bins); int32 atomics add the
//                     chunk to the caller's counts.
// Rows past N*S are zero tiles, bins past R hold "not a state": no edge cases inside the loop.
#include "epg_common.h"

#include <stdlib.h>

namespace epg {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

// Workgroup shape: 2 x 4 waves (192 x 384 cells of C, 18 operand tiles per k-step, one workgroup per CU).  Building with
// -DEPG_S3_WN=2 gives 2 x 2 waves (192 x 192, 12 tiles per k-step, 76 KB of LDS, TWO workgroups per CU -- confirmed with the
// occupancy API -- so that one workgroup's waves could own the matrix pipe while the other waits at its barrier): measured
// 70 ms per 1 M bins against 42, a third more operand traffic per MFMA outweighs the overlap.
#ifndef EPG_S3_WN
#define EPG_S3_WN 4
#endif
constexpr int G_WM = 2, G_WN = EPG_S3_WN;               // waves of a workgroup along the A / B side
constexpr int G_NW = G_WM * G_WN;
constexpr int G_BLK = 96;                               // rows of a wave's block (3 tiles of 32)
constexpr int G_BM = G_WM * G_BLK, G_BN = G_WN * G_BLK; // rows of C per workgroup
constexpr int G_TA = G_BM / 32, G_TB = G_BN / 32;       // operand tiles per k-step
constexpr int G_KS = 2;                                 // k-steps (of 64 bins) per stage
constexpr int G_LOADS = (G_KS * (G_TA + G_TB) + G_NW - 1) / G_NW;   // 1 KiB LDS-DMA loads per wave and stage (6; 5 of 40 slots for 2 x 4)
constexpr int G_LA = G_LOADS - 3;                       // of which between the MFMAs of a stage's first / second k-step: G_LA / 3
constexpr int G_STAGE_TILES = G_KS * (G_TA + G_TB);     // 36 for 2 x 4 waves: the surplus load slots (36..39) land in the scrap
constexpr int G_STAGE_BYTES = G_STAGE_TILES * 1024;
// LDS ring of STAGES stages (template parameter of the kernel: 3 or 4) + one scrap KiB per wave for dummy / surplus loads
__host__ __device__ constexpr int g_lds_bytes(int stages) { return stages * G_STAGE_BYTES + G_NW * 1024; }
constexpr int G_RING_DEFAULT = 3;
static_assert(g_lds_bytes(4) <= 160 * 1024, "the ring of four must fit a CU's LDS");
constexpr int G_PATCH_P = 8, G_PATCH_Q = G_WN == 2 ? 8 : 4;           // workgroup ordering: patches of the (P, Q) task grid
static_assert(G_LA == 2 || G_LA == 3, "five or six loads per wave and stage");
// bins per chunk the workspace size is quoted for (multiple of 512; < 2^24): 2 M bins = 15 GB of operand at N = 833, of 288.  Every
// chunk ends in the epilogue's ~2 atomics per cell of counts: 8 M bins in chunks of 1 / 2 / 4 M measure 283.0 / 276.2 / 281.2 ms
// for the expected phase (profiles/r06b_s3_chunk_overlap_ab.txt; 1 M until round 5)
constexpr long G_KC_MAX = 2097152;
constexpr long G_KC_MIN = 16384;
constexpr long G_REDUCED_MIN_BINS = 262144;             // calls shorter than this take the full contraction (see hist_s3_gemm)

__host__ __device__ inline int g_rows_padded(int NS) { return (NS + G_BN - 1) / G_BN * G_BN; }

// SWAR one-hot of 32 state bytes (two uint4) against the lane's state: fp4 nibbles, 0b0010 where equal
// (both operands come from this routine, so which bin lands in which nibble does not matter)
__device__ __forceinline__ u32 g_eq_pair_fp4(u32 w0, u32 w1, u32 pat) {
    const u32 d0 = 0x80808080u - (w0 ^ pat);
    const u32 d1 = 0x80808080u - (w1 ^ pat);
    return ((d0 >> 6) & 0x02020202u) | ((d1 >> 2) & 0x20202020u);
}

// gate / want (also k_s3_syrk_fp4): the launch does its work only when *gate == want (gate == nullptr: always) -- the reduced
// and the full contraction are both enqueued and the "a byte of this call is not a state" flag on the device picks one
__global__ __launch_bounds__(256) void k_s3_onehot_fp4(const char* __restrict__ XT, long Rp, int N, int S, int NS, int NT, long k0,
                                                        long nksteps, uint4* __restrict__ E4, const int* __restrict__ gate, int want) {
    if (gate && *gate != want) return;
    const int lane = threadIdx.x & 63;
    // (kstep, tile), tile fastest: 1 KiB per wave, in order; a bounded grid strides over them (a gated-off launch of one
    // block per four tiles -- 1.9 M blocks per 1 M bins -- took 0.39 ms to do nothing)
    for (long id = (long)blockIdx.x * 4 + (threadIdx.x >> 6); id < nksteps * NT; id += (long)gridDim.x * 4) {
        const int tile = (int)(id % NT);
        const long kstep = id / NT;
        const int r = tile * 32 + (lane & 31);
        const int a = r < NS ? r / S : N - 1;
        const u32 pat = (u32)(r < NS ? r - a * S : 30) * 0x01010101u;    // rows past N*S match nothing (states <= 29 or 31)
        const char* src = XT + (long)a * Rp + k0 + 64 * kstep + 32 * (lane >> 5);
        const uint4 r0 = *reinterpret_cast<const uint4*>(src), r1 = *reinterpret_cast<const uint4*>(src + 16);
        E4[id * 64 + lane] = make_uint4(g_eq_pair_fp4(r0.x, r0.y, pat), g_eq_pair_fp4(r0.z, r0.w, pat), g_eq_pair_fp4(r1.x, r1.y, pat),
                                        g_eq_pair_fp4(r1.z, r1.w, pat));
    }
}

// Workgroup tasks (P = G_WM-tuple of A blocks, Q = G_WN-tuple of B blocks; needed when some bm <= bn, i.e. P <= g_pmax(Q)),
// patch by patch.
__host__ __device__ inline int g_pmax(int Q) { return (G_WN * Q + G_WN - 1) / G_WM; }

static int g_ntasks(int NQ) {
    int n = 0;
    for (int Q = 0; Q < NQ; ++Q) n += g_pmax(Q) + 1;
    return n;
}

__global__ void k_s3_tasks(int NQ, int* __restrict__ tasks) {
    if (threadIdx.x || blockIdx.x) return;
    int n = 0;
    for (int qq = 0; qq * G_PATCH_Q < NQ; ++qq)
        for (int pp = 0; pp * G_PATCH_P <= g_pmax(qq * G_PATCH_Q + G_PATCH_Q - 1); ++pp)
            for (int Q = qq * G_PATCH_Q; Q < qq * G_PATCH_Q + G_PATCH_Q && Q < NQ; ++Q)
                for (int P = pp * G_PATCH_P; P < pp * G_PATCH_P + G_PATCH_P && P <= g_pmax(Q); ++P) tasks[n++] = P | (Q << 16);
}

// the six operand tiles of one k-step: issued as a batch, consumed one batch of nine MFMAs later
struct GOps {
    v4i a[3], b[3];
};

template <int KS_IMM>
__device__ __forceinline__ void g_read(GOps& o, u32 aA, u32 aB) {
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(o.a[t]) : "v"(aA), "n"((KS_IMM * (G_TA + G_TB) + t) * 1024) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(o.b[t]) : "v"(aB), "n"((KS_IMM * (G_TA + G_TB) + t) * 1024) : "memory");
    }
}

__device__ __forceinline__ void g_wait_lds(GOps& o) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(o.a[0]), "+v"(o.a[1]), "+v"(o.a[2]), "+v"(o.b[0]), "+v"(o.b[1]), "+v"(o.b[2])::"memory");
}

#define G_MFMA(I, J)                                                                                                             \
    acc[I][J] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(__builtin_shufflevector(o.a[I], o.a[I], 0, 1, 2, 3, -1, -1, -1, -1), \
                                                                __builtin_shufflevector(o.b[J], o.b[J], 0, 1, 2, 3, -1, -1, -1, -1), \
                                                                acc[I][J], 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f)

// nine MFMAs of one k-step (fp4 operands: four registers are read)
__device__ __forceinline__ void g_mfma(const GOps& o, bool active, v16f (&acc)[3][3]) {
    if (!active) return;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) G_MFMA(a, b);
    __builtin_amdgcn_s_setprio(0);
}

// The same with the wave's five LDS-DMA loads of a later stage spread between the MFMAs.  Issuing an LDS-DMA load costs the
// wave ~60-180 cycles of issue time (MI355X_MICROARCH.md, price list); five in a row right after the barrier, in both waves
// of a SIMD at once, left the matrix pipe idle for that long (1M bins: 55.6 ms; interleaved: see DESIGN.md).  sched_barrier
// pins the order the optimiser would otherwise restore.
// The same with NJ of the wave's LDS-DMA loads (j = J0 ..) spread between the MFMAs.
template <int J0, int NJ, typename Load>
__device__ __forceinline__ void g_mfma_loads(const GOps& o, bool active, v16f (&acc)[3][3], Load&& load) {
    static_assert(NJ == 2 || NJ == 3, "two or three loads per batch");
    if (!active) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) load(J0 + j);
        return;
    }
    __builtin_amdgcn_s_setprio(1);
    G_MFMA(0, 0); G_MFMA(0, 1);
    __builtin_amdgcn_sched_barrier(0);
    load(J0);
    __builtin_amdgcn_sched_barrier(0);
    G_MFMA(0, 2); G_MFMA(1, 0); G_MFMA(1, 1);
    __builtin_amdgcn_sched_barrier(0);
    load(J0 + 1);
    __builtin_amdgcn_sched_barrier(0);
    G_MFMA(1, 2); G_MFMA(2, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (NJ == 3) load(J0 + 2);
    __builtin_amdgcn_sched_barrier(0);
    G_MFMA(2, 1); G_MFMA(2, 2);
    __builtin_amdgcn_s_setprio(0);
}

// Epilogue of both contraction kernels (the LDS ring is free: every wave has passed a barrier after its last operand read).
// Direct cells C[a,b,i,j]: a lane holds column n = (b, j), consecutive lanes consecutive j -- the atomics of
// one instruction fall into a few 72-byte runs.  Mirrored cells C[b,a,j,i] want consecutive lanes on consecutive i, i.e.
// lanes along m: the tile goes through a padded 32 x 33 LDS scratch (the ring is free once every wave has passed the
// barrier below) and comes back transposed.  With lanes along n the mirrored atomics touched 32 lines per instruction
// and an epilogue cost ~1.8 ms per launch (3.5-3.9 ms per chunk with the two K splits).
__device__ __forceinline__ void g_epilogue(v16f (&acc)[3][3], char* smem, int w, int lane, int bm, int bn, int N, int S,
                                           int* __restrict__ counts, int* __restrict__ marg) {
    const int NS = N * S;
    const long SS = (long)S * S;
    float* scr = reinterpret_cast<float*>(smem) + w * (32 * 33);
    const int lm = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int ta = 0; ta < 3; ++ta)
#pragma unroll
        for (int tb = 0; tb < 3; ++tb) {
            const int m0 = (bm * 3 + ta) * 32, n0 = (bn * 3 + tb) * 32;
            {
                const int n = n0 + lm;
                const int b = n / S, j = n - b * S;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int v = (int)acc[ta][tb][r];
                    const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const int a = m / S, i = m - a * S;
                    if (v && m < NS && n < NS && a != b) atomicAdd(&counts[((long)a * N + b) * SS + i * S + j], v);
                    // the diagonal cell of (a, i) is the number of bins in which biosample a is in state i: the marginals
                    if (marg && v && m == n && m < NS) atomicAdd(&marg[a * 32 + i], v);
                }
            }
            if (bm != bn) {
#pragma unroll
                for (int r = 0; r < 16; ++r) scr[((r & 3) + 8 * (r >> 2) + 4 * lh) * 33 + lm] = acc[ta][tb][r];
                __builtin_amdgcn_wave_barrier();
                const int m = m0 + lm;
                const int a = m / S, i = m - a * S;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int nl = (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const int v = (int)scr[lm * 33 + nl];
                    const int n = n0 + nl;
                    const int b = n / S, j = n - b * S;
                    if (v && m < NS && n < NS && a != b) atomicAdd(&counts[((long)b * N + a) * SS + j * S + i], v);
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
}

template <int G_STAGES>
__global__ __launch_bounds__(64 * G_NW, 2) void k_s3_syrk_fp4(const char* __restrict__ E4, int NT, long nstages, long stages_per_split,
                                                         const int* __restrict__ tasks, int ntasks, int N, int S,
                                                         int* __restrict__ counts, int dbg, const int* __restrict__ gate, int want,
                                                         int* __restrict__ marg) {
    if (gate && *gate != want) return;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = w / G_WN, wn = w % G_WN;
    // XCD b % 8 takes a contiguous run of the patch-ordered task list (bijective for any ntasks)
    const int xcd = blockIdx.x & 7, q8 = ntasks >> 3, r8 = ntasks & 7;
    const int tix = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
    const int task = tasks[tix];
    const int P = task & 0xffff, Q = task >> 16;
    const int bm = G_WM * P + wm, bn = G_WN * Q + wn;
    const bool active = bm <= bn && !(dbg & 8);                             // wave-uniform (dbg 8: data path only, no MFMAs)
    const long g0 = (long)blockIdx.y * stages_per_split;
    const int G = (int)(nstages - g0 < stages_per_split ? nstages - g0 : stages_per_split);   // < 2^24 / 128
    if (G <= 0) return;

    // this wave's five load slots of a stage: slot = w + 8 j -> (k-step, tile) -> source offset inside the stage's two k-steps
    long soff[G_LOADS];
    u32 doff[G_LOADS];
    bool surplus[G_LOADS];
    char* const scrap = smem + G_STAGES * G_STAGE_BYTES + w * 1024;
#pragma unroll
    for (int j = 0; j < G_LOADS; ++j) {
        const int slot = w + G_NW * j;
        surplus[j] = slot >= G_STAGE_TILES;                                // 2 x 4 waves: slots 36..39, a duplicate load into the scrap
        const int s = surplus[j] ? 0 : slot;
        const int ks = s / (G_TA + G_TB), t = s - ks * (G_TA + G_TB);
        const int gt = (dbg & 1) ? t : (t < G_TA ? P * G_TA + t : Q * G_TB + (t - G_TA));   // dbg 1: every workgroup the same panels
        soff[j] = ((long)ks * NT + gt) * 1024 + lane * 16;
        doff[j] = (u32)slot * 1024;
    }
    const char* src0 = E4 + g0 * G_KS * (long)NT * 1024;
    const long stage_stride = (long)G_KS * NT * 1024;
    auto issue1 = [&](int g, int slot, int j) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src0 + g * stage_stride + soff[j]),
                                         (__attribute__((address_space(3))) void*)(surplus[j] ? scrap : smem + slot * G_STAGE_BYTES + doff[j]), 16, 0, 0);
    };
    auto issue = [&](int g, int slot) {
#pragma unroll
        for (int j = 0; j < G_LOADS; ++j) issue1(g, slot, j);
    };

    v16f acc[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const u32 lds0 = (u32)(uintptr_t)smem + (u32)lane * 16;
    const u32 baseA = lds0 + (u32)(wm * 3) * 1024, baseB = lds0 + (u32)(G_TA + wn * 3) * 1024;
    // Software pipeline.  LDS-DMA runs two to three stages ahead (ring of three: a stage's slot is refilled only after the
    // barrier that every wave passes once its last operands of that stage sit in registers).  The operand tiles of k-step
    // k + 1 are requested BEFORE the nine MFMAs of k-step k are issued and waited for after them, so the 48 KiB a
    // workgroup reads from LDS per k-step (~190 LDS cycles) and the read latency hide under matrix work instead of
    // standing between the two waves of a SIMD after every barrier (first version: 2400 cycles per stage for 1584 of MFMA).
    issue(0, 0);
#pragma unroll
    for (int s = 1; s < G_STAGES - 1; ++s) issue(G > s ? s : 0, s);   // a chunk shorter than the pipeline refetches stage 0: never read
#pragma unroll
    for (int j = G_LA; j < G_LOADS; ++j)                   // the last ring slot's first G_LA loads go out in the loop's first turn
        issue1(G > G_STAGES - 1 ? G_STAGES - 1 : 0, G_STAGES - 1, j);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((G_STAGES - 2) * G_LOADS + 3) : "memory");
    __builtin_amdgcn_s_barrier();
    GOps o0, o1;
    g_read<0>(o0, baseA, baseB);
    g_wait_lds(o0);
    // A wave's five DMA loads of a stage go out between MFMAs, two in one batch and three in the next, never in a row:
    // issuing an LDS-DMA load holds a wave up for 60-180 cycles, and the two waves of a SIMD reach the same point together.
    // Stage g + 2 is fetched into the slot of stage g - 1: loads j = 2, 3, 4 between the MFMAs of k-step 1 of stage g - 1
    // (after that turn's barrier: the slot is free) and j = 0, 1 between those of k-step 0 of stage g.  Issue order per wave:
    // ... s(g+1){2,3,4} s(g+1){0,1} s(g+2){2,3,4} s(g+2){0,1} | wait for stage g + 1 = vmcnt(5).  Past the last stage the
    // same instructions fetch stage 0 into the ring's scrap slots, so the loop has no load-dependent control flow.
    auto loads_for = [&](int gs, int into_slot) {
        const bool real = gs < G && !(dbg & 16);           // dbg 16: no operand traffic in the loop (dummy loads of stage 0)
        const char* lsrc = src0 + (real ? (long)gs * stage_stride : 0L);
        char* ldst = smem + into_slot * G_STAGE_BYTES;
        return [=, &soff, &doff, &surplus](int j) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(lsrc + soff[j]),
                                             (__attribute__((address_space(3))) void*)(real && !surplus[j] ? ldst + doff[j] : scrap), 16, 0, 0);
        };
    };
    int slot = 0, slot_prev = G_STAGES - 1;                // ring slots of stage g and g - 1 (no division in the loop)
    for (int g = 0; g < G; ++g) {
        const u32 so = (u32)slot * G_STAGE_BYTES;
        const int slot1 = slot == G_STAGES - 1 ? 0 : slot + 1;
        g_read<1>(o1, baseA + so, baseB + so);
        g_mfma_loads<0, G_LA>(o0, active, acc, loads_for(g + G_STAGES - 1, slot_prev));
        g_wait_lds(o1);
        if (g + 1 < G) {
            // stage g + 1: this wave's loads have landed (the five of stage g + 2 -- real or, at the tail, dummies -- stay in
            // flight); after the barrier everybody's.  Every wave that reaches the barrier holds its last operands of stage g
            // in registers: the slot is free.
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((G_STAGES - 2) * G_LOADS) : "memory");
            __builtin_amdgcn_s_barrier();
            const u32 sn = (u32)slot1 * G_STAGE_BYTES;
            g_read<0>(o0, baseA + sn, baseB + sn);
        }
        g_mfma_loads<G_LA, 3>(o1, active, acc, loads_for(g + G_STAGES, slot));
        if (g + 1 < G) g_wait_lds(o0);
        slot_prev = slot;
        slot = slot1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the tail's dummy loads
    __builtin_amdgcn_s_barrier();
    if (!active || (dbg & 2)) return;                     // dbg 2: no epilogue (measurements only)
    g_epilogue(acc, smem, w, lane, bm, bn, N, S, counts, marg);
}

// (Round 3 built this contraction three more ways -- a ring of four stages, a ping-pong schedule of the SIMD partners with a
// cycle trace of its segments, a one-wave-per-SIMD shape with 4 x 4 tiles -- each bit-identical and within +-3 % of this kernel
// (profiles/r03p_*, r03s_*).  They were deleted in round 5: the library holds one schedule.)

int transpose_states(const char* X, int64_t R, int32_t N, int64_t ldx, int32_t S, char* XT, int64_t Rp, int shift, hipStream_t st);

static long g_chunk_bins(long Rp) { return Rp < G_KC_MAX ? Rp : G_KC_MAX; }
static int64_t g_reduced_bytes(int N, int S);

static int64_t g_fixed_bytes(long Rp, int N, int NQ) {
    return align_up((int64_t)N * Rp + 64, 1024) + align_up((int64_t)g_ntasks(NQ) * 4, 1024);
}

// smallest workspace the kernel can run with (a 16 K-bin chunk of the operand) -- less than that and the caller's
// buffer goes to the build-in-kernel variant instead
int64_t s3_gemm_ws_min_bytes(int64_t R, int N, int S) {
    const long Rp = align_up(R, 512);
    const int NSP = g_rows_padded(N * S), NT = NSP / 32, NQ = NSP / G_BN;
    const long kc = Rp < G_KC_MIN ? Rp : G_KC_MIN;
    return g_fixed_bytes(Rp, N, NQ) + (int64_t)NT * 1024 * (kc / 64);
}

// workspace: XT | task list | E4 chunk
int64_t s3_gemm_ws_bytes(int64_t R, int N, int S) {
    const long Rp = align_up(R, 512);
    const int NSP = g_rows_padded(N * S), NT = NSP / 32, NQ = NSP / G_BN;
    return g_fixed_bytes(Rp, N, NQ) + (int64_t)NT * 1024 * (g_chunk_bins(Rp) / 64) + (S >= 3 ? g_reduced_bytes(N, S) : 0);
}

int transpose_states_flag(const char* X, int64_t R, int32_t N, int64_t ldx, int32_t S, char* XT, int64_t Rp, int shift, int bad, int* dirty,
                          hipStream_t st);

// ---------------------------------------------------------------------------------------------------------------
// The REDUCED contraction.  In a bin every biosample is in exactly one state, so the one-hot column of ONE state per biosample
// is 1 minus the others: with m = S - 1 left out of the operand the contraction runs on N (S - 1) rows -- (17/18)^2 of the
// matrix work at S = 18, 1406 instead of 1640 workgroup tasks at N = 833 -- and the cells of state m follow from the
// marginals M_a[i] = #{bins : x_a = i} (the contraction's own diagonal cells) and the number of bins R:
//     C[a,b,m,j] = M_b[j] - sum_{i != m} C[a,b,i,j]        C[a,b,i,m] = M_a[i] - sum_{j != m} C[a,b,i,j]
//     C[a,b,m,m] = R - sum_{i != m} M_a[i] - sum_{j != m} M_b[j] + sum_{i,j != m} C[a,b,i,j]
// Integers: exact.  The identity needs every state byte of the call to be a state; k_transpose_states reports whether one is
// not, on the device, and BOTH contractions are enqueued, gated on that flag: a clean call (every call of the command line,
// whose parser checks the range) runs the reduced one plus k_s3_reconstruct, a call with a "not a state" byte the full one.
// No host synchronisation, the same counts either way (tests/test_hip_s3_n833.py runs both on one matrix).
// ---------------------------------------------------------------------------------------------------------------
// A block takes RC_P consecutive (a, b) pairs: their reduced cells (RC_P * (S-1)^2 contiguous ints) go through LDS, row and
// column sums are formed there, and every thread then adds whole contiguous runs of the full cells to counts (coalesced
// read-modify-write; nothing else writes counts while this runs).  ~0.8 GB read + 1.8 GB read-modify-write at N = 833.
constexpr int RC_P = 16;
__global__ __launch_bounds__(256) void k_s3_reconstruct(const int* __restrict__ Cr, const int* __restrict__ marg, long R, int N, int S,
                                                         int* __restrict__ counts, const int* __restrict__ gate, int want) {
    if (gate && *gate != want) return;
    extern __shared__ int rc_s[];
    const int S1 = S - 1, c1 = S1 * S1, c2 = S * S;
    int* s_cr = rc_s;                         // [RC_P][S1][S1]
    int* s_row = s_cr + RC_P * c1;            // [RC_P][S1]  sum over j of the reduced cell's row i
    int* s_col = s_row + RC_P * S1;           // [RC_P][S1]  sum over i of column j
    long long* s_mm = reinterpret_cast<long long*>(s_col + RC_P * S1 + ((RC_P * (c1 + 2 * S1)) & 1));   // [RC_P] the (m, m) cell
    const long npairs = (long)N * N;
    for (long p0 = (long)blockIdx.x * RC_P; p0 < npairs; p0 += (long)gridDim.x * RC_P) {
        const int np = (int)(npairs - p0 < RC_P ? npairs - p0 : RC_P);
        __syncthreads();
        for (int e = threadIdx.x; e < np * c1; e += 256) s_cr[e] = Cr[p0 * c1 + e];
        __syncthreads();
        for (int t = threadIdx.x; t < np * 2 * S1; t += 256) {
            const int p = t / (2 * S1), k = t - p * 2 * S1;
            const int* cr = s_cr + p * c1;
            int acc = 0;
            if (k < S1) {
                for (int j = 0; j < S1; ++j) acc += cr[k * S1 + j];
                s_row[p * S1 + k] = acc;
            } else {
                for (int i = 0; i < S1; ++i) acc += cr[i * S1 + (k - S1)];
                s_col[p * S1 + (k - S1)] = acc;
            }
        }
        __syncthreads();
        if ((int)threadIdx.x < np) {
            const int p = threadIdx.x;
            const long ab = p0 + p;
            const int a = (int)(ab / N), b = (int)(ab - (long)a * N);
            long long all = 0, ma = 0, mb = 0;
            for (int i = 0; i < S1; ++i) {
                all += s_row[p * S1 + i];
                ma += marg[a * 32 + i];
                mb += marg[b * 32 + i];
            }
            s_mm[p] = (long long)R - ma - mb + all;
        }
        __syncthreads();
        int* out = counts + p0 * c2;
        const int nout = np * c2;
        for (int e0 = threadIdx.x; e0 < nout; e0 += 256 * 4) {                // four cells per thread and turn: their loads in flight together
            int old[4], add[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + 256 * u;
                old[u] = e < nout ? out[e] : 0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + 256 * u;
                add[u] = 0;
                if (e >= nout) continue;
                const int p = e / c2, rem = e - p * c2, i = rem / S, j = rem - i * S;
                const long ab = p0 + p;
                const int a = (int)(ab / N), b = (int)(ab - (long)a * N);
                if (a == b) continue;                                       // the diagonal stays zero (expected.py:183-200)
                if (i < S1 && j < S1) add[u] = s_cr[p * c1 + i * S1 + j];
                else if (i == S1 && j < S1) add[u] = marg[b * 32 + j] - s_col[p * S1 + j];     // x_a = m, x_b = j
                else if (j == S1 && i < S1) add[u] = marg[a * 32 + i] - s_row[p * S1 + i];     // x_a = i, x_b = m
                else add[u] = (int)s_mm[p];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + 256 * u;
                if (e < nout && add[u]) out[e] = old[u] + add[u];
            }
        }
    }
}

static int64_t g_reduced_bytes(int N, int S) {      // reduced counts | marginals [N][32] | flag
    return align_up((int64_t)N * N * (S - 1) * (S - 1) * 4, 1024) + align_up((int64_t)N * 32 * 4, 1024) + 1024;
}

static int hist_s3_gemm_run(const char* XT, long Rp, int N, int S, int32_t* counts, int* tasks, char* E4, long KC, int dbg_env,
                            const int* gate, int want, int* marg, hipStream_t st);

int hist_s3_gemm(const char* X, int64_t R, int32_t N, int64_t ldx, int32_t S, int32_t* counts, void* ws, int64_t ws_bytes,
                 hipStream_t st) {
    static const int dbg_env = [] { const char* e = exp_env("EPG_S3_DBG"); return e ? atoi(e) : 0; }();   // (experiments build only)
    // the reduced contraction from G_REDUCED_MIN_BINS bins on -- its fixed cost per call (zeroing and re-expanding a 0.8 GB count
    // array: ~1.4 ms at N = 833) is what it saves on ~260 K bins; epg_test_force(2, 1 / 2): the full / the reduced one whatever
    // the call's size (the tests compare the two on their small shapes)
    const int forced = g_force[FORCE_S3_CONTRACTION];
    const bool no_reduced = forced == 1 ? true : forced == 2 ? false : R < G_REDUCED_MIN_BINS;
    const long Rp = align_up(R, 512);                  // whole stages; padded bins hold 31, which matches no row
    const int NS = N * S, NSP = g_rows_padded(NS), NT = NSP / 32, NQ = NSP / G_BN;
    const int ntasks = g_ntasks(NQ);
    char* XT = reinterpret_cast<char*>(ws);
    int* tasks = reinterpret_cast<int*>(XT + align_up((int64_t)N * Rp + 64, 1024));
    char* E4 = reinterpret_cast<char*>(tasks) + align_up((int64_t)ntasks * 4, 1024);
    if (dbg_env & 4) {
        int nblk = -1;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, reinterpret_cast<const void*>(k_s3_syrk_fp4<G_RING_DEFAULT>), 64 * G_NW,
                                                     g_lds_bytes(G_RING_DEFAULT));
        fprintf(stderr, "k_s3_syrk_fp4: %d workgroups of %d threads per CU with %d bytes of LDS\n", nblk, 64 * G_NW, g_lds_bytes(G_RING_DEFAULT));
    }
    static DynLds lds_attr;
    EPG_HIP(ensure_dyn_lds(lds_attr, reinterpret_cast<const void*>(k_s3_syrk_fp4<G_RING_DEFAULT>), g_lds_bytes(G_RING_DEFAULT)));
    // chunk = as many bins of the operand as the caller's workspace holds (every chunk ends in one epilogue of ~2 atomics
    // per cell of counts, so fewer, longer chunks are better); EPG_S3_KC overrides for measurements
    static const long kc_env = [] { const char* e = exp_env("EPG_S3_KC"); return e ? atol(e) / 512 * 512 : 0L; }();   // (experiments build only)
    auto chunk_bins = [&](int64_t bytes_for_e4) {
        long KC = bytes_for_e4 / ((int64_t)NT * 1024) * 64 / 512 * 512;
        if (kc_env > 0 && kc_env < KC) KC = kc_env;
        if (KC > Rp) KC = Rp;
        if (KC > (1L << 24) - 512) KC = (1L << 24) - 512;   // float32 accumulators hold exact integers
        return KC;
    };
    const int64_t fixed = g_fixed_bytes(Rp, N, NQ);
    // the reduced path when the workspace also holds its count array and still a 16 K-bin chunk of the operand
    const int64_t red = g_reduced_bytes(N, S);
    const bool reduced = !no_reduced && S >= 3 && ws_bytes - fixed - red >= (int64_t)NT * 1024 * (G_KC_MIN / 64 < Rp / 64 ? G_KC_MIN / 64 : Rp / 64);
    if (!reduced) {
        const long KC = chunk_bins(ws_bytes - fixed);
        if (KC < 512) return fail(EPG_ERR_WORKSPACE, "hist_s3: workspace too small for the precomputed-operand kernel");
        int rc = transpose_states(X, R, N, ldx, S, XT, Rp, 0, st);
        if (rc) return rc;
        hipLaunchKernelGGL(k_s3_tasks, dim3(1), dim3(1), 0, st, NQ, tasks);
        EPG_LAUNCH_CHECK("k_s3_tasks");
        return hist_s3_gemm_run(XT, Rp, N, S, counts, tasks, E4, KC, dbg_env, nullptr, 0, nullptr, st);
    }
    char* tail = reinterpret_cast<char*>(ws) + (ws_bytes - red) / 1024 * 1024;
    int* Cr = reinterpret_cast<int*>(tail);
    int* marg = reinterpret_cast<int*>(tail + align_up((int64_t)N * N * (S - 1) * (S - 1) * 4, 1024));
    int* dirty = marg + (align_up((int64_t)N * 32 * 4, 1024) / 4);
    const long KC = chunk_bins(tail - E4);
    if (KC < 512) return fail(EPG_ERR_WORKSPACE, "hist_s3: workspace too small for the precomputed-operand kernel");
    EPG_HIP(hipMemsetAsync(tail, 0, (size_t)red - 1024 + 64, st));        // reduced counts, marginals, flag
    int rc = transpose_states_flag(X, R, N, ldx, S, XT, Rp, 0, 31, dirty, st);
    if (rc) return rc;
    // clean call: the contraction without state S - 1 (same XT: a byte S - 1 matches none of the N (S - 1) rows), then the rest
    const int NQ1 = g_rows_padded(N * (S - 1)) / G_BN;
    int* tasks1 = tasks;                                                   // the reduced task list is a prefix-compatible rebuild
    hipLaunchKernelGGL(k_s3_tasks, dim3(1), dim3(1), 0, st, NQ1, tasks1);
    EPG_LAUNCH_CHECK("k_s3_tasks");
    rc = hist_s3_gemm_run(XT, Rp, N, S - 1, Cr, tasks1, E4, KC, dbg_env, dirty, 0, marg, st);
    if (rc) return rc;
    {
        long blocks = ((long)N * N + RC_P - 1) / RC_P;
        if (blocks > num_cus() * 16L) blocks = num_cus() * 16L;
        const int S1 = S - 1;
        const size_t shmem = (size_t)(RC_P * (S1 * S1 + 2 * S1) + 2) * 4 + RC_P * 8;
        hipLaunchKernelGGL(k_s3_reconstruct, dim3((unsigned)blocks), dim3(256), shmem, st, Cr, marg, (long)R, N, S, counts, dirty, 0);
        EPG_LAUNCH_CHECK("k_s3_reconstruct");
    }
    // a call with a "not a state" byte: the full contraction (its launches return at once on a clean call)
    hipLaunchKernelGGL(k_s3_tasks, dim3(1), dim3(1), 0, st, NQ, tasks);
    EPG_LAUNCH_CHECK("k_s3_tasks");
    return hist_s3_gemm_run(XT, Rp, N, S, counts, tasks, E4, KC, dbg_env, dirty, 1, nullptr, st);
}

// the chunk loop of one contraction over the transposed matrix: one-hot operand of a chunk, then the SYRK kernel.
// The two ALTERNATE on the caller's stream (2.3 ms of HBM / VALU work, then 32 ms of matrix work, per 1 M bins).  Round 6 built the
// obvious overlap -- operand of chunk c + 1 into a second buffer on a library-owned helper stream, events between the two -- and
// measured it (profiles/r06b_s3_chunk_overlap_ab.txt, 8 M bins): helper stream at the lowest priority 283.8 ms against 283.0 serial
// (the operand kernel gets no CU before the contraction's last workgroups are out), at normal / high priority +3-4 % (its blocks
// take CUs the contraction then cannot use), confined to 8 / 16 / 32 CUs by a CU mask 533 / 428 / 351 ms (it needs most of the chip
// to reach its bandwidth).  The reason is in the resource report: k_s3_syrk_fp4 takes 256 VGPRs x 2 waves per SIMD, the whole
// register file of every CU it runs on, so no other wave can be resident beside it -- two kernels never share a CU here, and a
// second stream can only hand whole CUs from one to the other.  The code was taken out again; the serial order stays.
static int hist_s3_gemm_run(const char* XT, long Rp, int N, int S, int32_t* counts, int* tasks, char* E4, long KC, int dbg_env,
                            const int* gate, int want, int* marg, hipStream_t st) {
    const int NS = N * S, NSP = g_rows_padded(NS), NT = NSP / 32, NQ = NSP / G_BN;
    const int ntasks = g_ntasks(NQ);
    for (long k0 = 0; k0 < Rp; k0 += KC) {
        const long kc = Rp - k0 < KC ? Rp - k0 : KC;   // multiple of 512
        const long nksteps = kc / 64, nstages = nksteps / G_KS;
        const long waves = nksteps * NT;
        long oh_blocks = (waves + 3) / 4;
        if (oh_blocks > num_cus() * 64L) oh_blocks = num_cus() * 64L;
        hipLaunchKernelGGL(k_s3_onehot_fp4, dim3((unsigned)oh_blocks), dim3(256), 0, st, XT, Rp, N, S, NS, NT, k0, nksteps,
                           reinterpret_cast<uint4*>(E4), gate, want);
        EPG_LAUNCH_CHECK("k_s3_onehot_fp4");
        // Split the chunk's stages over blockIdx.y: few tasks (small N) need it to give every CU work, and with one
        // workgroup per CU the last round of a launch is only partly full -- 1640 tasks on 256 CUs are 6.4 rounds, paid as
        // 7; in two halves 12.8, paid as 13.  int32 atomics combine the splits.
        const long cus = (long)num_cus() * (G_NW == 4 ? 2 : 1);   // workgroups the chip holds at a time
        long splits = (2L * cus + ntasks - 1) / ntasks, best = 0;
        double best_waste = 1e9;
        for (long sp = splits; sp <= splits + 3; ++sp) {
            const double rounds = (double)ntasks * sp / cus;
            const double waste = (double)((ntasks * sp + cus - 1) / cus) / rounds;
            if (waste < best_waste - 0.01) { best_waste = waste; best = sp; }
        }
        splits = best;
        if (splits > nstages) splits = nstages;
        if (splits < 1) splits = 1;
        if (splits > 65535) splits = 65535;
        const long per = (nstages + splits - 1) / splits;
        const long nsplit = (nstages + per - 1) / per;
        hipLaunchKernelGGL(k_s3_syrk_fp4<G_RING_DEFAULT>, dim3((unsigned)ntasks, (unsigned)nsplit), dim3(64 * G_NW), g_lds_bytes(G_RING_DEFAULT), st, E4, NT,
                           nstages, per, tasks, ntasks, N, S, counts, dbg_env, gate, want, marg);
        EPG_LAUNCH_CHECK("k_s3_syrk_fp4");
    }
    return EPG_OK;
}

}  // namespace epg
