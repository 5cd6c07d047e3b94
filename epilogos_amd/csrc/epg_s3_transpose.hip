// Transposed state matrix XT[biosample][bin] for the S3 kernels (gfx950).
//
// The S3 expected pass (epg_s3_gemm.hip: one-hot fp4 contraction, reference expected.py:165-204) and both S3 score kernels
// (epg_s3_lanes.hip, epg_s3.hip; reference scores.py:455-506) want the bins of ONE biosample as contiguous bytes.  The kernels
// here write that layout once per call into the caller's workspace: bins padded to Rp, every byte that is not a state in
// [0, S) replaced by a code the consumer's tables answer with zero, optionally shifted (4 * state is a ready-made LDS byte
// offset), and -- for the reduced contraction, which needs every byte of the call to be a state -- a device flag that says
// whether one was not.
// (Rounds 1-2 contracted straight from XT with the one-hot operand built inside the MFMA kernel -- k_s3_hist_mfma_b / e / f,
// 71-138 ms per 1 M bins, profiles/r01*, r02* --; since round 2 the operand is precomputed (epg_s3_gemm.hip, 35 ms) and nothing
// dispatched to those kernels any more: deleted in round 6.)
#include "epg_common.h"

namespace epg {

// The transpose with 16-byte global accesses (round 5): a thread loads 16 state bytes of one bin, the tile goes through LDS, a
// thread stores 16 bins of one biosample.  (The byte-per-thread form of rounds 1-4 moved 25 GB at 1.25 TB/s, 20 ms per transpose of
// the 15 M-bin genome, two per S3 job; it stayed as the path for a misaligned XT until round 6 -- which no workspace this library
// lays out produces: the S3 entry points now refuse a workspace that is not 16-byte aligned, and the kernel is gone.)  Needs rows
// of at least N bytes readable in 16-byte pieces: a piece that would reach past the row pitch is read byte by byte.
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

// bytes of XT for a call of R bins (bins padded to whole 512-bin stages of the contraction)
int64_t s3_xt_bytes(int64_t R, int N) { return align_up((int64_t)N * align_up(R, 512) + 64, 256); }

// XT[sample][bin], bins padded to Rp (a multiple of 32), everything that is not a state in [0, S) stored as 31;
// bytes are stored shifted left by `shift` (the S3 score kernel wants 4 * state, a ready-made LDS byte offset)
// `bad` is the code stored for "not a state" (31 for the kernels of this file and k_s3_score; S for k_s3_score_bl, whose
// table rows have exactly one zero column after the S states)
// `dirty` (optional, device int, caller-zeroed): set to 1 when a byte of the first N columns of a row < R is not a state
int transpose_states_flag(const char* X, int64_t R, int32_t N, int64_t ldx, int32_t S, char* XT, int64_t Rp, int shift, int bad, int* dirty,
                          hipStream_t st) {
    // 16-byte stores need XT rows that start 16-byte aligned (Rp a multiple of 16, an aligned base): so for every workspace this
    // library lays out (hist_s3_impl / score_s3_impl refuse a misaligned one)
    if (Rp % 16 != 0 || (reinterpret_cast<uintptr_t>(XT) & 15) != 0)
        return fail(EPG_ERR_INVALID_ARG, "transpose_states: the transposed matrix must be 16-byte aligned with a pitch that is a multiple of 16");
    hipLaunchKernelGGL(k_transpose_states16, dim3((unsigned)(((Rp + 63) / 64) * ((N + 63) / 64))), dim3(256), 0, st, X, (long)R, N,
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

}  // namespace epg
