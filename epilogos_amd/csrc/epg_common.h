// Shared helpers for the epilogos_amd HIP sources (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "epilogos_amd.h"

typedef unsigned int u32;
typedef unsigned long long u64;
typedef unsigned short u16;

namespace epg {

int fail(int code, const char* fmt, ...);
int num_cus();

// Several entry points have a FALLBACK kernel that other shapes take (wider models, rows too long for the fast kernel's LDS
// budget, calls too small for the reduced contraction ...).  The tests must be able to run a fallback on THEIR small shapes and
// compare it with the default: epg_test_force(which, value) (epg_abi.hip; declared in the header as a test hook) sets one of
// these, 0 = the library decides.  There is no environment variable in any dispatch path of the default build.
// Measurement switches (EPG_*_DBG, chunk sizes ...) exist only in a library built with EPILOGOS_BUILD_EXPERIMENTS=1
// (python -m epilogos_amd.build reads the variable): the default library never looks at the environment.
#ifdef EPILOGOS_BUILD_EXPERIMENTS
#include <stdlib.h>
static inline const char* exp_env(const char* name) { return getenv(name); }
#else
static inline const char* exp_env(const char*) { return nullptr; }
#endif

enum Force { FORCE_NULL_SEQ = 0, FORCE_S3_SCORE_BINS, FORCE_S3_CONTRACTION, FORCE_S3_HIST_LDS, FORCE_K1_BLOCKS_PER_CU, FORCE_COUNT };
extern int g_force[FORCE_COUNT];

#define EPG_HIP(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess)                                                                 \
            return epg::fail(EPG_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e));     \
    } while (0)

#define EPG_LAUNCH_CHECK(name)                                                                \
    do {                                                                                      \
        hipError_t _e = hipGetLastError();                                                    \
        if (_e != hipSuccess)                                                                 \
            return epg::fail(EPG_ERR_HIP, "launch of %s failed: %s", name, hipGetErrorString(_e)); \
    } while (0)

static inline int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) of a kernel that wants more than 64 KB of dynamic LDS: once per kernel AND
// device, safe when two host threads meet at a kernel's first launch (round 5 kept a plain `static bool` per call site: a second
// device, or a second thread that saw the flag before the first had made the call, launched without the attribute).  One
// DynLds object (static, at the call site) per kernel; a bit per device; setting the attribute twice is harmless.
struct DynLds {
    unsigned long long done = 0;
};
static inline hipError_t ensure_dyn_lds(DynLds& a, const void* kernel, int bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (__atomic_load_n(&a.done, __ATOMIC_ACQUIRE) & bit) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) __atomic_fetch_or(&a.done, bit, __ATOMIC_RELEASE);
    return e;
}

// Rows a wave stages per turn in the kernels that keep whole rows of a wave's tile in LDS (four waves per block, 64 KB of LDS
// without asking): 64 for the state models the kernels were built for, 32 / 16 / 8 when `row_bytes` (all staged arrays of a
// row together) grows with a wide model.  A multiple of 8, so that a tile of uint16 or float32 rows starts 16-byte aligned.
static inline int tile_rows(int64_t row_bytes) {
    int tr = 64;
    while (tr > 8 && 4 * tr * row_bytes > 65536) tr >>= 1;
    return tr;
}

// (a & m) | (b & ~m): one v_bfi_b32
__device__ __forceinline__ u32 bfi(u32 m, u32 a, u32 b) { return (a & m) | (b & ~m); }

// 16-byte load from an arbitrarily aligned address (amdhsa enables unaligned access mode; the compiler
// emits a single global_load_dwordx4)
__device__ __forceinline__ uint4 ld16(const char* p) {
    uint4 v;
    __builtin_memcpy(&v, p, 16);
    return v;
}

// sum over the 4 lanes of a quad, result in every lane (two v_add_u32_dpp quad_perm)
__device__ __forceinline__ u32 quad_sum(u32 v) {
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false);  // quad_perm [1,0,3,2]
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false);  // quad_perm [2,3,0,1]
    return v;
}

// Store `nbytes` (a multiple of 2) staged in the wave's LDS buffer to `dst` (16-byte aligned): whole 16-byte chunks
// with dwordx4 stores -- consecutive lanes, consecutive chunks, so every store instruction covers whole 128-byte
// lines -- then a dword tail and, for an odd number of uint16 (odd S, odd number of rows), the last two bytes.
__device__ __forceinline__ void store_staged(const char* lds, char* dst, int nbytes, int lane) {
    const int nchunks = nbytes >> 4;
    for (int c = lane; c < nchunks; c += 64)
        *reinterpret_cast<uint4*>(dst + 16 * c) = *reinterpret_cast<const uint4*>(lds + 16 * c);
    const int tail0 = nchunks << 4;
    for (int o = tail0 + 4 * lane; o + 4 <= nbytes; o += 256)
        *reinterpret_cast<u32*>(dst + o) = *reinterpret_cast<const u32*>(lds + o);
    if ((nbytes & 2) && lane == 0) *reinterpret_cast<u16*>(dst + nbytes - 2) = *reinterpret_cast<const u16*>(lds + nbytes - 2);
}

// The same with NON-TEMPORAL stores, for score outputs (nobody on the device reads them again).  A float32 score array written
// with ordinary stores lingers dirty in L2 and the 256 MB memory-side cache and is written back under whatever streams next --
// the next job's count pass: measured in bench.py's loop of S1 jobs, k_bin_hist inside the step 2.30 -> 2.25 ms on the genome
// and 0.323 -> 0.296 ms on the 1.9 M-bin shard of an 8-GPU split (whose 135 MB of scores fit that cache entirely), the score pass
// itself no slower (0.343 -> 0.337 ms).  Histograms keep ordinary stores: the score pass reads them back at once.
typedef u32 epg_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_staged_nt(const char* lds, char* dst, int nbytes, int lane) {
    const int nchunks = nbytes >> 4;
    for (int c = lane; c < nchunks; c += 64)
        __builtin_nontemporal_store(*reinterpret_cast<const epg_u32x4*>(lds + 16 * c), reinterpret_cast<epg_u32x4*>(dst + 16 * c));
    const int tail0 = nchunks << 4;
    for (int o = tail0 + 4 * lane; o + 4 <= nbytes; o += 256)
        __builtin_nontemporal_store(*reinterpret_cast<const u32*>(lds + o), reinterpret_cast<u32*>(dst + o));
    if ((nbytes & 2) && lane == 0) *reinterpret_cast<u16*>(dst + nbytes - 2) = *reinterpret_cast<const u16*>(lds + nbytes - 2);
}

// kl(p, q) = p * log2(p / q) with the reference's masked-zero semantics (scores.py:550):
// 0 where q == 0, 0 where p/q <= 0.
__device__ __forceinline__ double kl_term(double p, double q) {
    if (q == 0.0) return 0.0;
    const double r = p / q;
    if (!(r > 0.0)) return 0.0;
    return p * log2(r);
}

// One entry of the S3 score table, T = klScoreND(float32(1) / P, q) with both operands float32 (scores.py:479-480; the masked-zero
// rules of scores.py:550).  Every operation of the reference's float32 expression is CORRECTLY ROUNDED here: the quotient (IEEE
// division), the logarithm (evaluated in float64 and rounded once to float32) and the product.  Rounds 1-5 called the device's
// float32 log2f (<= 1 ulp, not the same ulp as numpy's): real tables are made of few distinct values, so a logarithm that is one
// ulp off does not average out over the 832 terms of a (bin, biosample), and with terms of both signs the scores were only good
// to 2e-6 of the float64 restatement.  numpy's own float32 log2 is 1-2 ulp off the correctly rounded value in 0.5-23 % of the
// arguments (SVML on the build container's Xeon, tests/test_oracle_golden.py records it); against ITS table the scores from this
// one agree to < 1e-7 on every test shape, against a correctly rounded host table to the fixed-point rounding (~1e-9).
__device__ __forceinline__ float s3_table_entry(float qv, float obs) {
    if (qv == 0.0f) return 0.0f;
    const float r = __fdiv_rn(obs, qv);
    if (!(r > 0.0f)) return 0.0f;
#ifdef EPG_S3_TABLE_LOG2F                                // (tools/build_ab_lib.py -DEPG_S3_TABLE_LOG2F: rounds 1-5's table, for the record)
    return obs * log2f(r);
#else
    return __fmul_rn(obs, (float)log2((double)r));
#endif
}

}  // namespace epg
