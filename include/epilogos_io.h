/*
 * epilogos_io.h -- C ABI of libepilogos_io.so: native input codec and output writer of the scoring path
 * (SURVEY.md section 8, rows f1 and f2).  Host-side C++ (no GPU): what the reference does with pandas / str.format /
 * single-threaded gzip in
 *     epilogos/helpers.py:80-99    countRows            -> epgio_count_rows
 *     epilogos/helpers.py:152-155  readStates (parse)   -> epgio_open_table / epgio_table_copy_states
 *     epilogos/scores.py:161       locationArr          -> epgio_table_locations
 *     epilogos/scores.py:509-536   writeScores          -> epgio_write_scores
 *
 * Input format (reference README.md:127-134): tab-separated, no header, optionally gzip; columns 1-3 = chromosome,
 * start, end (echoed verbatim to the outputs), columns 4.. = one integer state per biosample, 1-based.
 * Output format (scores.py:530-531): "chr\tstart\tend\t" + S values formatted "%.5f" of the float32, tab separated,
 * one line per bin, gzip.  The writer emits one gzip member per chunk of rows (a multi-member gzip file is a valid
 * gzip file); the decompressed bytes are identical to the reference's.
 *
 * All functions are thread-safe with respect to different handles/paths.  Errors: negative return (or NULL) and
 * epgio_last_error() (thread-local).
 */
#ifndef EPILOGOS_IO_H
#define EPILOGOS_IO_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct epgio_table epgio_table;

const char* epgio_last_error(void);

/* Number of '\n' characters in the (gz) file -- like the reference, a last line without newline is not counted. */
int64_t epgio_count_rows(const char* path);

/* Parse rows [row_lo, row_hi) (row_hi < 0: to the end of the file) with `threads` parser threads (0 = all cores).
 * Returns a handle owning the parsed table, or NULL. */
epgio_table* epgio_open_table(const char* path, int64_t row_lo, int64_t row_hi, int32_t threads);
/* The same for a state model of `max_state` states: up to 31 (what epgio_open_table assumes) values outside 1..31 are stored
 * as "not a state", above that values outside 1..127 -- the GPU kernels of the wide models decode the whole byte. */
epgio_table* epgio_open_table_ex(const char* path, int64_t row_lo, int64_t row_hi, int32_t threads, int32_t max_state);

/* The same, parsing the states STRAIGHT INTO the caller's destination: once the number of rows and state columns is known
 * (after the inflate and the line count), alloc(rows, cols, &ldx, user) is called once, on the calling thread, and returns an
 * int8 buffer of rows * ldx bytes with ldx >= cols (or NULL to give up); every row is parsed into it, bytes cols .. ldx - 1 of
 * a row are set to -1.  What the driver uses with page-locked, row-padded staging buffers: no intermediate matrix, no copy.
 * epgio_table_copy_states is a no-op check on such a table.  threads == 0 here and above means "share": each parallel phase
 * takes epgio_default_threads() divided by the number of files this process is reading at that moment. */
typedef int8_t* (*epgio_alloc_fn)(int64_t rows, int32_t cols, int64_t* ldx, void* user);
epgio_table* epgio_open_table_into(const char* path, int64_t row_lo, int64_t row_hi, int32_t threads, int32_t max_state,
                                   epgio_alloc_fn alloc, void* user);
int64_t epgio_table_rows(const epgio_table* t);
int32_t epgio_table_cols(const epgio_table* t);          /* number of state columns N */
/* Smallest and largest state value of the parsed rows AS WRITTEN IN THE FILE (1-based); 0, 0 for an empty table.  The
 * reference indexes a numStates-long array with value - 1 (expected.py:113) and dies on anything outside 1..numStates;
 * callers compare this range with their state model. */
int epgio_table_state_range(const epgio_table* t, int32_t* lo, int32_t* hi);
/* Copy the 0-based int8 states (file value - 1) into out[r * ldx + c]; bytes c >= N of a row are set to -1.  A value
 * outside 0..30 (0..126 for a table opened for a model of more than 31 states) is stored as -1 = "not a state". */
int epgio_table_copy_states(const epgio_table* t, int8_t* out, int64_t ldx);
/* Verbatim text of the first three columns of every row, each terminated by '\n' ("chr\tstart\tend\n"),
 * concatenated; offsets[r]..offsets[r+1] delimit row r (newline included).  Valid until epgio_close_table. */
const char* epgio_table_locations(const epgio_table* t, const int64_t** offsets);
void epgio_close_table(epgio_table* t);
/* Integer start / end of every row of such a location text ("name\tstart\tend\n" per row) and whether all rows carry the
 * first row's name -- what STEP 4 needs of locationArr (roiSingle.py:95-142) without a text parse of 15 M rows in Python.
 * Fails (-1) when a coordinate is not a plain integer; callers then fall back to a general parser. */
int epgio_parse_locations(const char* loc, const int64_t* loc_off, int64_t R, int64_t* start, int64_t* end,
                          int32_t* same_chrom, int32_t threads);

/* Write R lines: location text of row r (loc + loc_off[r] .. loc_off[r+1], its trailing '\n' dropped) + '\t' + S
 * "%.5f" values + '\n'.
 * gzip_level 1..9 = zlib (the reference's gzip.open default is 9), 0 = the library's fast compressor (about five times
 * the speed of level 6 for files 4-12 % larger); threads 0 = all. */
int epgio_write_scores(const char* path, const char* loc, const int64_t* loc_off, const float* scores, int64_t R,
                       int32_t S, int32_t threads, int32_t gzip_level);

/* The INPUT format, written: R lines "chrom\tstart\tend\t" + N tab-separated 1-based states (value = states[r*ldx + c] + 1),
 * start = start0 + r*step, end = start + step; gzip (multi-member, compressed in parallel) when the path ends in "gz",
 * plain text otherwise.  What bin/preprocess_data_ChromHMM.sh produces for the reference (README.md:127-134); here it
 * feeds the end-to-end runs and tests with matrices at the reference's real width. */
int epgio_write_states(const char* path, const char* chrom, int64_t start0, int64_t step, const int8_t* states, int64_t R,
                       int32_t N, int64_t ldx, int32_t threads, int32_t gzip_level);

/* pairwiseMetrics_*.txt.gz of the paired STEP 4 (roiAndVisualPairwise.py:520-573 writeMetrics): one line per bin,
 *   chromosome \t start \t end \t state name \t "%.5f" of |distance| (float32) \t "+" if distance >= 0 else "-"
 *   [ \t "%.5e" p-value \t "%.5e" adjusted p-value ]          (pvals and mh both NULL or both given)
 * chromosome names: table of nchrom strings (chrom + chrom_off[i] .. chrom_off[i+1]) indexed by chrom_idx[r]; state
 * names likewise, indexed by maxdiff[r] - 1.  Multi-member gzip like epgio_write_scores. */
int epgio_write_metrics(const char* path, const char* chrom, const int64_t* chrom_off, const int32_t* chrom_idx,
                        const int64_t* start, const int64_t* end, const char* names, const int64_t* names_off,
                        const int32_t* maxdiff, const float* dist, const double* pvals, const double* mh, int64_t R,
                        int32_t threads, int32_t gzip_level);

/* Format n float32 values exactly like Python's "%.5f" % float(v) (correctly rounded, "-0.00000" kept), each followed
 * by `sep`; returns the number of bytes written to buf (cap must be >= 48*n).  Exposed for tests. */
int64_t epgio_format_f5(const float* v, int64_t n, char sep, char* buf, int64_t cap);

/* STEP 4 helpers (roiSingle.py:100 `scoreArr.sum(axis=1)`, helpers.py:262-266 rolling max of filter_regions):
 * row sums of a float32 [R, S] matrix (row stride lda) in EXACTLY numpy's order for a contiguous row -- eight running
 * accumulators over groups of eight, their tree sum, then the remainder, S <= 128 -- so that every total has numpy's bits;
 * and the centred rolling maximum pandas computes for Series.rolling(W, center=True).max(): out[i] = max x[i - W/2 .. i +
 * (W - 1)/2], NaN where the window is incomplete.  Both threaded over rows. */
int epgio_row_sums_f32(const float* a, int64_t R, int32_t S, int64_t lda, float* out, int32_t threads);
int epgio_rolling_max_f64(const double* x, int64_t n, int32_t W, double* out, int32_t threads);

/* One gzip member (RFC 1952) holding in[0, n), written by the library's own fast DEFLATE compressor (csrc/epg_deflate.h) --
 * what the writers above use at gzip_level 0.  cap must be >= n + n/8 + 1100; returns the member's size, < 0 on error.
 * Replaces Python's gzip.open(..., "wt") of scores.py:523 (zlib level 9).  Exposed for tests. */
int64_t epgio_gzip_fast(const void* in, int64_t n, void* out, int64_t cap);

/* Blocked gzip.  A BGZF file (htslib's bgzip / tabix: members of at most 64 KiB, each with its compressed size in a 'B','C' extra
 * subfield) is recognised by every reader above and its blocks are inflated IN PARALLEL -- a plain gzip file is one serial stream
 * per file; with threads == 0 the thread count is taken anew every 1024 blocks, so cores that other files' readers give back join
 * in.  Every block's CRC-32 and ISIZE are checked; a file that is not BGZF all the way, or a block that is not what its header
 * and trailer say, goes to the general reader and then to zlib, whose verdict stands.  With the environment variable
 * EPILOGOS_BGZF=1 every writer below emits BGZF (same decompressed bytes; gzip, zlib and this library read it like any
 * multi-member file, tabix-style tools with random access). */

/* The gzip reader on a buffer in memory: own == 2 the parallel BGZF reader (declines anything else), other own != 0 the library's inflate (csrc/epg_inflate.h: every member's CRC-32 and ISIZE
 * checked), own == 0 zlib's; returns the inflated size (copied to out, cap bytes) or < 0 -- the stream was declined / is
 * corrupt / does not fit.  Exposed for the differential fuzz (tools/asan_io.sh, tests/test_native_io.py). */
int64_t epgio_inflate_mem(const void* in, int64_t n, void* out, int64_t cap, int32_t own);

/* Host budget.  Wherever `threads` is 0 above the library uses epgio_default_threads(): EPILOGOS_HOST_THREADS when that is
 * set (the driver gives every rank of a node its share: cores / LOCAL_WORLD_SIZE, capped by the reference's -c,
 * run.py:36,148), else the hardware threads capped by the cgroup CPU quota and by 64.  epgio_thread_census reports the
 * library's runnable threads right now and their peak since the last reset (callers inside the library + workers they
 * started; a caller waiting for its workers does not count). */
int32_t epgio_default_threads(void);
/* Sets that budget (n <= 0: none given).  The environment variable is read once, at the library's first use; the binding
 * passes later values through here so that no native thread ever reads the environment while Python changes it. */
void epgio_set_host_threads(int32_t n);
/* How many files the caller is reading, or is about to read, side by side (worker threads that still have a file to do): the
 * "share" of threads == 0 divides by the larger of this and the readers inside the library at that moment -- a worker between
 * two files must not make the others believe its cores are free.  0 = no plan. */
void epgio_set_reader_plan(int32_t n);
/* The readers keep the text buffers of big files (>= 64 MiB inflated) for the next file instead of returning them to the kernel
 * in the middle of a run; this returns them -- on a detached thread when background != 0.  Call it when nothing is left to read. */
void epgio_release_buffers(int32_t background);
void epgio_thread_census(int32_t* live, int32_t* peak, int32_t reset);

#ifdef __cplusplus
}
#endif
#endif /* EPILOGOS_IO_H */
