/*
 * epilogos_amd.h -- C ABI of libepilogos_hip.so, the MI355X (gfx950) scoring engine for epilogos.
 *
 * This is the drop-in boundary for the reference's scoring hot path.  The reference (meuleman/epilogos) is pure
 * Python and has no FFI; the path sits behind three Python entry points
 *     expected.main            (epilogos/expected.py:11)
 *     expectedCombination.main (epilogos/expectedCombination.py:9)
 *     scores.main              (epilogos/scores.py:14)
 * whose per-chunk arithmetic kernels are the functions cited on each prototype below.  A maintainer binds these
 * symbols with ctypes (see INTEGRATION.md for the stub); epilogos_amd/_abi.py is that binding in this repo.
 *
 * Conventions
 *   - Every function returns EPG_OK (0) or a negative EPG_ERR_* code; epg_last_error() returns a thread-local message.
 *   - Caller owns every buffer.  The library allocates nothing, retains no pointer after the call returns, and
 *     enqueues its kernels on `stream` (a hipStream_t; NULL = the default stream).  Calls are asynchronous with
 *     respect to the host: synchronise the stream before reading results.
 *   - All data pointers are DEVICE pointers (e.g. torch.Tensor.data_ptr() of a tensor on the current HIP device).
 *     There is no host/CPU implementation behind this ABI: without a HIP device the calls fail with EPG_ERR_HIP.
 *   - X is the state matrix: row-major int8 [R, ldx], one row per genomic bin, one byte per biosample, 0-based
 *     states (file value - 1, helpers.py:154-155); only the first N bytes of a row are read as states
 *     (ldx >= N; the fast path wants X and ldx 16-byte aligned, any ldx >= N is accepted).  State bytes must be in
 *     0..31 or 0xFF (-1): a byte in [S, 31] or 0xFF is "not a state" and is counted nowhere (the expected pass detects
 *     it: sum(counts) != R*N); the S1/S2 kernels decode only the low five bits, so other byte values are outside the
 *     contract (libepilogos_io's parser stores every file value outside 1..31 as -1).  A model of 32..127 states is handed
 *     to plain kernels that decode the whole byte (csrc/epg_wide.hip): any byte outside [0, S) is then "not a state".
 *   - `counts` outputs ACCUMULATE (+=) so that per-chromosome calls sum into one vector exactly like
 *     expectedCombination.py:30-35; zero them first.  They are what the single RCCL all-reduce runs on.
 *   - One host thread per device; calls on different devices/streams are independent.
 *
 * Limits of this build: 1 <= S <= 127 (states are int8; the tuned kernels serve 1..31 -- the reference's ChromHMM models have
 * 15, 18 and 25 -- and larger models take a generic, slow path; epg_null_hist stops at 31),
 * N <= 65535 (per-bin counts are stored as uint16).
 */
#ifndef EPILOGOS_AMD_H
#define EPILOGOS_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EPG_OK 0
#define EPG_ERR_INVALID_ARG (-1)
#define EPG_ERR_UNSUPPORTED (-2)
#define EPG_ERR_HIP (-3)
#define EPG_ERR_WORKSPACE (-4)

/* Bumped with EVERY change of this file's entry points or of what a workspace / table argument must hold; epilogos_amd/_abi.py
 * reads the number from this header and refuses a library that reports another (a stale build behind EPILOGOS_HIP_LIB).
 *   1  rounds 1-5 (entry points were added without a bump: fixed in 2)
 *   2  round 6: epg_test_force switch 4 replaces the undeclared epg_debug_set_variant; epg_ws_bytes(3, ...) quotes operand chunks
 *      of 2 M bins (1 M before); S3 score tables are correctly rounded float32 (the device's log2f before) */
#define EPG_ABI_VERSION 2

int epg_version(void);
const char* epg_last_error(void);
/* Number of compute units of the current device (used by callers to size nothing; informational) or <0. */
int epg_device_cus(void);

/* ---- per-bin histogram: the HBM-streaming kernel everything in S1/S2 is built on ---------------------------
 * H[b, s] = #{n < N : X[b, n] == s} as uint16 [R, S]  (what np.unique(dataArr[row], return_counts=True) yields:
 * scores.py:341 rowObsS1, scores.py:444 rowObsS2, expected.py:152 s2Calc).
 * counts[s] += sum_b H[b, s]  (expected.py:106-113 s1Calc).  H or counts may be NULL. */
int epg_bin_hist(const int8_t* X, int64_t R, int32_t N, int64_t ldx, int32_t S,
                 uint16_t* H, int64_t* counts, void* stream);

/* epg_bin_hist with the S2 pair counts of the same bins folded into the launch: H as above (required), counts2[S*S] +=
 * sum_b h_i*h_j (i != j), h_i*(h_i-1) (i == j) -- what epg_hist_s2_from_binhist(H) adds, expected.py:146-158 -- and, when
 * counts is not NULL, counts[S] += the state counts.  One launch for 15-, 18- and 25-state models on rows of up to 1024 columns
 * (the pair products come from the rows the kernel stages in LDS for the H store); the two passes otherwise.  Same integers. */
int epg_bin_hist_s2(const int8_t* X, int64_t R, int32_t N, int64_t ldx, int32_t S, uint16_t* H, int64_t* counts,
                    int64_t* counts2, void* stream);

/* epg_bin_hist over `nparts` resident matrices (the chromosome files of a genome; both groups of a paired run) in as few
 * launches as their widths allow (one when every N[p] has the same number of 128-byte groups per row), all parts' state
 * counts into the same counts[S].  X, R, N, ldx, H are HOST arrays of nparts entries (H, or single entries of it, may be NULL).
 * The same integers as nparts calls of epg_bin_hist.  Replaces the per-file loop of run.py:236-257 / expected.py:100-113. */
int epg_bin_hist_parts(int32_t nparts, const int8_t* const* X, const int64_t* R, const int32_t* N, const int64_t* ldx,
                       int32_t S, uint16_t* const* H, int64_t* counts, void* stream);

/* ---- expected-frequency pass (STEP 1) ---------------------------------------------------------------------
 * S1: counts[S]   += state counts                                   -- expected.py:90-116  s1Calc
 * S2: counts[S*S] += sum_b h_i*h_j (i != j), h_i*(h_i-1) (i == j)   -- expected.py:119-162 s2Calc
 * S3: counts[N*N*S*S] (int32) += #{b : X[b,a]==i and X[b,c]==j}, a != c  -- expected.py:165-204 s3Calc */
int epg_hist_s1(const int8_t* X, int64_t R, int32_t N, int64_t ldx, int32_t S, int64_t* counts, void* stream);
int epg_hist_s2(const int8_t* X, int64_t R, int32_t N, int64_t ldx, int32_t S, int64_t* counts,
                void* ws, int64_t ws_bytes, void* stream);
int epg_hist_s2_from_binhist(const uint16_t* H, int64_t R, int32_t S, int64_t* counts, void* stream);
/* Paired mode: the S2 counts of the column concatenation [A|B] (helpers.py:173 readStates, expBool) from the two
 * groups' per-bin histograms, h = HA + HB bin by bin -- neither the concatenated matrix nor a summed histogram is
 * materialised.  (S1 counts of [A|B] are simply epg_bin_hist of A and of B into the same counts.) */
int epg_hist_s2_from_binhist_pair(const uint16_t* HA, const uint16_t* HB, int64_t R, int32_t S, int64_t* counts,
                                  void* stream);
/* epg_hist_s3: with ws == NULL the LDS-counter kernel runs; with epg_ws_bytes(3, R, N, S) bytes of workspace the matrix-core
 * contraction (the transposed matrix, a chunk of the one-hot operand and -- for calls of 262 144 bins or more -- a count array
 * over S - 1 states per biosample from which the cells of the last state are re-derived; a byte that is not a state anywhere in
 * the call switches, on the device, to the contraction over all S states).  The counts are the same integers on every path. */
int epg_hist_s3(const int8_t* X, int64_t R, int32_t N, int64_t ldx, int32_t S, int32_t* counts,
                void* ws, int64_t ws_bytes, void* stream);

/* ---- combination / normalisation (STEP 2) -------------------------------------------------------------------
 * q[k] = (float) ( (double) counts[k] / (double) sum(counts) )      -- expectedCombination.py:42
 * (the sum over files / ranks happens before this call: += accumulation and the RCCL all-reduce). */
int epg_normalise_i64(const int64_t* counts, int64_t n, float* q, void* ws, int64_t ws_bytes, void* stream);
int epg_normalise_i32(const int32_t* counts, int64_t n, float* q, void* ws, int64_t ws_bytes, void* stream);

/* ---- score pass (STEP 3) ------------------------------------------------------------------------------------
 * Workspace size in bytes for the score calls of a given saliency (device memory, caller-allocated; the S3 entry points want it
 * 16-byte aligned -- any device allocation is -- and return EPG_ERR_INVALID_ARG otherwise). */
int64_t epg_ws_bytes(int32_t saliency, int64_t R, int32_t N, int32_t S);

/* S1: score[b, s] = kl(h[b,s]/N, q[s]), kl(p, q) = p*log2(p/q), 0 where q == 0 or p == 0
 *     -- scores.py:259-344 s1Score/rowObsS1 and scores.py:539-550 klScoreND.
 * out64 (double [R,S], the value before the reference's float32 store) and/or out32 (float [R,S], the value the
 * reference stores, scores.py:144,317); either may be NULL. */
int epg_score_s1(const int8_t* X, int64_t R, int32_t N, int64_t ldx, int32_t S, const float* q,
                 double* out64, float* out32, void* ws, int64_t ws_bytes, void* stream);
/* Same from cached per-bin histograms; N is the divisor (group width, scores.py:343). */
int epg_score_s1_from_binhist(const uint16_t* H, int64_t R, int32_t N, int32_t S, const float* q,
                              double* out64, float* out32, void* ws, int64_t ws_bytes, void* stream);

/* STEP 2 + STEP 3 of an S1 job in one call (two launches): q = normalise(counts) as epg_normalise_i64 -- counts is the
 * global, already all-reduced int64[S] vector -- written to q[S]; then the scores of this shard's cached histograms
 * as epg_score_s1_from_binhist.  rezero != 0 additionally leaves counts zeroed for the next job's accumulation
 * (expectedCombination.py:30-42 + scores.py:259-344).  R may be 0 (only q is produced). */
int epg_combine_score_s1(int64_t* counts, int32_t rezero, const uint16_t* H, int64_t R, int32_t N, int32_t S, float* q,
                         double* out64, float* out32, void* ws, int64_t ws_bytes, void* stream);

/* The S1 score pass from cached histograms with the lookup table T[c, s] = kl(c / N, q[s]), c = 0..N, built by the CALLER
 * (device pointers to [N + 1, S] row-major arrays; T64 feeds out64, T32 feeds out32, either pair may be NULL).  The command
 * line builds the table on the host with the reference's own numpy expression (scores.py:343 rowObsS1, :550 klScoreND), so
 * the float32 scores -- and scores_*.txt.gz -- equal the reference's bit for bit. */
int epg_score_s1_from_binhist_table(const uint16_t* H, int64_t R, int32_t N, int32_t S, const double* T64, const float* T32,
                                    double* out64, float* out32, void* stream);

/* Paired S1, everything between the null groups and the output files in one pass over the four per-bin histograms of a bin
 * (real groups A, B of NA, NB columns; null groups of ga, gb columns): with the groups' S1 tables T*[c, s] (device, [width + 1, S]
 * float32, as for epg_score_s1_from_binhist_table; a null table may be the same pointer as its real group's)
 *   delta[bin, s]  = T_A[hA[s], s] - T_B[hB[s], s]                    (scores.py:223-226: float32 scores, float32 difference)
 *   null_dist[bin] = sign(sum nd) * sum nd^2, nd = T_nA[hnA] - T_nB[hnB]  (scores.py:229-232, numpy's float32 pairwise order)
 *   dist, maxdiff  = STEP 4's reduction of delta as it reads it back from the "%.5f" text (roiAndVisualPairwise.py:347-354)
 * -- the results of four epg_score_s1_from_binhist_table, two epg_pair_finish and one epg_pair_metrics, bit for bit.
 * EPG_ERR_UNSUPPORTED when the tables do not fit a CU's LDS next to the staging areas (groups of several thousand columns):
 * the separate calls serve then. */
int epg_pair_scores_s1_from_binhist(const uint16_t* HA, const uint16_t* HB, const uint16_t* HnA, const uint16_t* HnB, int64_t R, int32_t S,
                                    int32_t NA, int32_t NB, int32_t ga, int32_t gb, const float* TA, const float* TB, const float* TnA,
                                    const float* TnB, float* delta, float* null_dist, float* dist, int32_t* maxdiff, void* stream);

/* The same pass over SEVERAL parts in one launch (the command line holds every chromosome file's bins as a part of their own):
 * HA .. maxdiff are HOST arrays of nparts device pointers, R[p] the rows of part p (parts without rows are skipped); the tables
 * and widths are shared.  mask (may be NULL, and so may its entries): uint8 [R[p]] per part, the quiescence mask of
 * scores.py:294-303 -- 1 where every column of A and of B holds state `qstate` (0-based; < 0: nothing is quiescent) -- out of
 * the same pass, which replaces epg_quiescent_from_binhist.  Same results as one call per part, bit for bit; same error codes. */
int epg_pair_scores_s1_parts(int32_t nparts, const uint16_t* const* HA, const uint16_t* const* HB, const uint16_t* const* HnA,
                             const uint16_t* const* HnB, const int64_t* R, int32_t S, int32_t NA, int32_t NB, int32_t ga, int32_t gb,
                             const float* TA, const float* TB, const float* TnA, const float* TnB, float* const* delta, float* const* null_dist,
                             float* const* dist, int32_t* const* maxdiff, uint8_t* const* mask, int32_t qstate, void* stream);

/* S2: p[i,j] = (h_i*h_j - [i==j]*h_i) / perms, score[b, j] = sum_i kl(p[i,j], q[i,j]) in ascending i
 *     -- scores.py:347-452 s2Score/rowObsS2.  perms = N*(N-1) of the ORIGINAL group (scores.py:371,397-398). */
int epg_score_s2(const int8_t* X, int64_t R, int32_t N, int64_t ldx, int32_t S, int64_t perms, const float* q,
                 double* out64, float* out32, void* ws, int64_t ws_bytes, void* stream);
/* From cached histograms; N = number of columns H was counted over (upper bound of any count). */
int epg_score_s2_from_binhist(const uint16_t* H, int64_t R, int32_t N, int32_t S, int64_t perms, const float* q,
                              double* out64, float* out32, void* ws, int64_t ws_bytes, void* stream);

/* S3: T = kl(float32(1)/(N*(N-1)), q) in float32 -- the quotient, the logarithm and the product each correctly rounded --;
 *     score[b, s] = sum_{c: X[b,c]==s} sum_{a != c} T[a,c,X[b,a],s]  -- scores.py:455-506 s3Score.  The sum is exact (fixed-point
 *     integers) where the reference adds 693 056 float32 terms in sequence: within 1e-6 of the float64 sum of numpy's table
 *     (observed <= 2.1e-7), within 1e-4 / 5e-6 abs of the reference's own float32 rows. */
int epg_score_s3(const int8_t* X, int64_t R, int32_t N, int64_t ldx, int32_t S, const float* q,
                 double* out64, float* out32, void* ws, int64_t ws_bytes, void* stream);

/* ---- paired mode extras ---------------------------------------------------------------------------------------
 * delta = a - b (float32); signed_sqdist[b] = sum_s delta^2 * sign(sum_s delta)   -- scores.py:223-232
 * (signed_sqdist may be NULL). */
int epg_pair_finish(const float* a, const float* b, int64_t R, int32_t S, float* delta, float* signed_sqdist,
                    void* stream);
/* Per-bin inputs of the paired STEP 4 -- roiAndVisualPairwise.py:347-354 (readInData):
 *   dist[b]    = sum_s d^2 * sign(sum_s d), float32, states added in ascending order (the reference reduces a
 *                column-major frame, so numpy adds column after column);
 *   maxdiff[b] = 1-based state with the largest |d|, ties to the higher state.
 * d is delta[b, :] as the reference sees it: re-read from the "%.5f" text of pairwiseDelta_*.txt.gz.  roundtrip != 0
 * applies that text round trip (float32 -> 5 decimals, half-even on the exact value -> nearest float32) to the
 * float32 delta of epg_pair_finish; roundtrip == 0 takes delta as already parsed from the file. */
int epg_pair_metrics(const float* delta, int64_t R, int32_t S, int32_t roundtrip, float* dist, int32_t* maxdiff,
                     void* stream);
/* mask[b] = all(XA[b,:]==qstate) && all(XB[b,:]==qstate)                          -- scores.py:294-303 */
int epg_quiescent(const int8_t* XA, int32_t NA, int64_t ldxa, const int8_t* XB, int32_t NB, int64_t ldxb,
                  int64_t R, int32_t qstate, uint8_t* mask, void* stream);
/* The same mask from the two groups' per-bin histograms: all NA columns of A and all NB columns of B hold qstate
 * <=> HA[b, qstate] == NA and HB[b, qstate] == NB. */
int epg_quiescent_from_binhist(const uint16_t* HA, const uint16_t* HB, int64_t R, int32_t S, int32_t NA, int32_t NB,
                               int32_t qstate, uint8_t* mask, void* stream);
/* Per-row uniform shuffle of the concatenation [A|B] (helpers.py:183-184: argsort of i.i.d. uniforms), written
 * as per-bin histograms of the two null halves: HA from the first `ga` shuffled columns, HB from the next `gb`
 * (helpers.py:190-194).  Philox4x32 keyed by (seed, row): reproducible, independent of the launch geometry. */
int epg_null_hist(const int8_t* XA, int32_t NA, int64_t ldxa, const int8_t* XB, int32_t NB, int64_t ldxb,
                  int64_t R, int32_t S, int32_t ga, int32_t gb, uint64_t seed, int64_t row0,
                  uint16_t* HA, uint16_t* HB, void* stream);

/* The same law from the two real groups' per-bin histograms HA, HB (uint16 [R, S], as epg_bin_hist leaves them): the
 * per-state counts of the first `ga` and the next `gb` columns of a uniform permutation of a row depend on the row only
 * through h = HA + HB and are multivariate hypergeometric; drawn exactly by selection sampling per state, the row's most
 * frequent state taking the remainder without a draw.  n_cols = NA + NB (columns without a state, n_cols - sum h, take
 * part in the shuffle and are not reported).  Same seeding contract as epg_null_hist, a DIFFERENT random stream: the two
 * entry points agree in distribution, not draw by draw.  OA / OB may not alias HA / HB. */
/* (epg_null_hist_from_binhist_parts below: the same for several parts -- host arrays of nparts pointers, row counts and
 * shuffle keys row0 -- in one launch; bit-identical to a call per part.) */
int epg_null_hist_from_binhist(const uint16_t* HA, const uint16_t* HB, int64_t R, int32_t S, int32_t n_cols, int32_t ga,
                               int32_t gb, uint64_t seed, int64_t row0, uint16_t* OA, uint16_t* OB, void* stream);
int epg_null_hist_from_binhist_parts(int32_t nparts, const uint16_t* const* HA, const uint16_t* const* HB, const int64_t* R,
                                     int32_t S, int32_t n_cols, int32_t ga, int32_t gb, uint64_t seed, const int64_t* row0,
                                     uint16_t* const* OA, uint16_t* const* OB, void* stream);

/* Paired mode, the default group sizes (ga = NA, gb = NB): count pass of both groups AND the null draw of several resident parts in
 * ONE kernel -- HA / HB as epg_bin_hist_parts leaves them (their state counts into counts[S], which may be NULL), OA / OB as
 * epg_null_hist_from_binhist_parts draws them from HA / HB with the keys row0; the same integers, the same draws.  Every part has NA
 * columns in XA and NB in XB.  Returns EPG_ERR_UNSUPPORTED for shapes outside the fused kernel's (S other than 15 / 18 / 25, widths
 * that differ in their number of 128-byte groups per row or exceed four, a row pitch not padded to 16 bytes): the caller then makes
 * the two calls.  Replaces helpers.py:173,183-194 + expected.py:100-113 for resident files of a paired run. */
int epg_pair_count_null_parts(int32_t nparts, const int8_t* const* XA, const int8_t* const* XB, const int64_t* R, int32_t NA,
                              int32_t NB, const int64_t* ldxa, const int64_t* ldxb, int32_t S, uint16_t* const* HA,
                              uint16_t* const* HB, int64_t* counts, uint64_t seed, const int64_t* row0, uint16_t* const* OA,
                              uint16_t* const* OB, void* stream);

/* ---- test hook (tests/ only; nothing in the package calls it) ----------------------------------------------
 * Several entry points have a fallback kernel that other shapes take; epg_test_force(which, value) makes the next calls take it
 * on any shape so that the tests can compare it with the default on theirs.  which: 0 = the column-by-column null sampler
 * (value 1), 1 = the bin-per-lane S3 score kernel (1), 2 = the S3 contraction (1 = over all S states, 2 = the reduced one
 * whatever the call's size), 3 = the LDS-counter S3 count kernel (1), 4 = blocks per CU of the count pass's persistent grid (value =
 * the number; an A/B measurement switch, same results).  value 0 = the library decides (the default). */
int epg_test_force(int32_t which, int32_t value);

#ifdef __cplusplus
}
#endif
#endif /* EPILOGOS_AMD_H */
