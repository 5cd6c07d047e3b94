// extern "C" surface of libepilogos_hip.so -- see include/epilogos_amd.h for the contract of every symbol.
#include "epg_common.h"

namespace epg {

static thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int num_cus() {
    static thread_local int cached_dev = -1;
    static thread_local int cached_cus = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (dev != cached_dev) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        cached_dev = dev;
        cached_cus = cus;
    }
    return cached_cus;
}

// implemented in epg_s1.hip / epg_s2.hip / epg_s3.hip / epg_null.hip
int bin_hist_impl(const int8_t*, int64_t, int32_t, int64_t, int32_t, uint16_t*, int64_t*, hipStream_t);
int64_t s1_ws_bytes(int64_t, int, int);
int score_s1_impl(const int8_t*, int64_t, int32_t, int64_t, int32_t, const float*, double*, float*, void*, int64_t, hipStream_t);
int score_s1_from_hist_impl(const uint16_t*, int64_t, int32_t, int32_t, const float*, double*, float*, void*, int64_t, hipStream_t);
int normalise_i64_impl(const int64_t*, int64_t, float*, void*, int64_t, hipStream_t);
int normalise_i32_impl(const int32_t*, int64_t, float*, void*, int64_t, hipStream_t);
int hist_s2_from_binhist_impl(const uint16_t*, const uint16_t*, int64_t, int32_t, int64_t*, hipStream_t);
int score_s1_from_hist_table_impl(const uint16_t*, int64_t, int32_t, int32_t, const double*, const float*, double*, float*, hipStream_t);
int pair_scores_s1_parts_impl(int32_t, const uint16_t* const*, const uint16_t* const*, const uint16_t* const*, const uint16_t* const*, const int64_t*,
                              int32_t, int32_t, int32_t, int32_t, int32_t, const float*, const float*, const float*, const float*, float* const*,
                              float* const*, float* const*, int32_t* const*, uint8_t* const*, int32_t, hipStream_t);
int pair_scores_s1_impl(const uint16_t*, const uint16_t*, const uint16_t*, const uint16_t*, int64_t, int32_t, int32_t, int32_t, int32_t, int32_t,
                        const float*, const float*, const float*, const float*, float*, float*, float*, int32_t*, hipStream_t);
int combine_score_s1_impl(int64_t*, int32_t, const uint16_t*, int64_t, int32_t, int32_t, float*, double*, float*, void*, int64_t, hipStream_t);
int64_t s2_table_bytes(int, int);
int score_s2_from_hist_impl(const uint16_t*, int64_t, int32_t, int32_t, int64_t, const float*, double*, float*, void*, int64_t, hipStream_t);
int pair_finish_impl(const float*, const float*, int64_t, int32_t, float*, float*, hipStream_t);
int pair_metrics_impl(const float*, int64_t, int32_t, int32_t, float*, int32_t*, hipStream_t);
int quiescent_impl(const int8_t*, int32_t, int64_t, const int8_t*, int32_t, int64_t, int64_t, int32_t, uint8_t*, hipStream_t);
int hist_s3_impl(const int8_t*, int64_t, int32_t, int64_t, int32_t, int32_t*, void*, int64_t, hipStream_t);
int64_t s3_ws_bytes(int64_t, int, int);
int score_s3_impl(const int8_t*, int64_t, int32_t, int64_t, int32_t, const float*, double*, float*, void*, int64_t, hipStream_t);
int null_hist_impl(const int8_t*, int32_t, int64_t, const int8_t*, int32_t, int64_t, int64_t, int32_t, int32_t, int32_t,
                   uint64_t, int64_t, uint16_t*, uint16_t*, hipStream_t);

int null_hist_from_binhist_impl(const uint16_t*, const uint16_t*, int64_t, int32_t, int32_t, int32_t, int32_t, uint64_t, int64_t, uint16_t*,
                                uint16_t*, hipStream_t);

int quiescent_from_binhist_impl(const uint16_t*, const uint16_t*, int64_t, int32_t, int32_t, int32_t, int32_t, uint8_t*, hipStream_t);
int bin_hist_parts_impl(int32_t, const int8_t* const*, const int64_t*, const int32_t*, const int64_t*, int32_t, uint16_t* const*, int64_t*, hipStream_t);
int null_hist_parts_impl(int32_t, const uint16_t* const*, const uint16_t* const*, const int64_t*, int32_t, int32_t, int32_t, int32_t, uint64_t,
                         const int64_t*, uint16_t* const*, uint16_t* const*, hipStream_t);

int pair_count_null_parts_impl(int32_t, const int8_t* const*, const int8_t* const*, const int64_t*, int32_t, int32_t, const int64_t*, const int64_t*,
                               int32_t, uint16_t* const*, uint16_t* const*, int64_t*, uint64_t, const int64_t*, uint16_t* const*, uint16_t* const*,
                               hipStream_t);

int bin_hist_s2_impl(const int8_t*, int64_t, int32_t, int64_t, int32_t, uint16_t*, int64_t*, int64_t*, hipStream_t);

int g_force[FORCE_COUNT] = {0};

}  // namespace epg

using namespace epg;

extern "C" {

int epg_test_force(int32_t which, int32_t value) {
    if (which < 0 || which >= FORCE_COUNT) return fail(EPG_ERR_INVALID_ARG, "test_force: unknown switch %d", which);
    g_force[which] = value;
    return EPG_OK;
}

int epg_version(void) { return EPG_ABI_VERSION; }
const char* epg_last_error(void) { return g_err; }

int epg_device_cus(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(EPG_ERR_HIP, "no HIP device available");
    return num_cus();
}

int epg_bin_hist(const int8_t* X, int64_t R, int32_t N, int64_t ldx, int32_t S, uint16_t* H, int64_t* counts, void* stream) {
    return bin_hist_impl(X, R, N, ldx, S, H, counts, (hipStream_t)stream);
}

int epg_bin_hist_s2(const int8_t* X, int64_t R, int32_t N, int64_t ldx, int32_t S, uint16_t* H, int64_t* counts, int64_t* counts2,
                    void* stream) {
    return bin_hist_s2_impl(X, R, N, ldx, S, H, counts, counts2, (hipStream_t)stream);
}

int epg_bin_hist_parts(int32_t nparts, const int8_t* const* X, const int64_t* R, const int32_t* N, const int64_t* ldx, int32_t S,
                       uint16_t* const* H, int64_t* counts, void* stream) {
    return bin_hist_parts_impl(nparts, X, R, N, ldx, S, H, counts, (hipStream_t)stream);
}

int epg_null_hist_from_binhist_parts(int32_t nparts, const uint16_t* const* HA, const uint16_t* const* HB, const int64_t* R, int32_t S,
                                     int32_t n_cols, int32_t ga, int32_t gb, uint64_t seed, const int64_t* row0, uint16_t* const* OA,
                                     uint16_t* const* OB, void* stream) {
    return null_hist_parts_impl(nparts, HA, HB, R, S, n_cols, ga, gb, seed, row0, OA, OB, (hipStream_t)stream);
}

int epg_pair_count_null_parts(int32_t nparts, const int8_t* const* XA, const int8_t* const* XB, const int64_t* R, int32_t NA, int32_t NB,
                              const int64_t* ldxa, const int64_t* ldxb, int32_t S, uint16_t* const* HA, uint16_t* const* HB, int64_t* counts,
                              uint64_t seed, const int64_t* row0, uint16_t* const* OA, uint16_t* const* OB, void* stream) {
    return pair_count_null_parts_impl(nparts, XA, XB, R, NA, NB, ldxa, ldxb, S, HA, HB, counts, seed, row0, OA, OB, (hipStream_t)stream);
}

int epg_hist_s1(const int8_t* X, int64_t R, int32_t N, int64_t ldx, int32_t S, int64_t* counts, void* stream) {
    if (!counts) return fail(EPG_ERR_INVALID_ARG, "hist_s1: counts is NULL");
    return bin_hist_impl(X, R, N, ldx, S, nullptr, counts, (hipStream_t)stream);
}

int epg_hist_s2_from_binhist(const uint16_t* H, int64_t R, int32_t S, int64_t* counts, void* stream) {
    return hist_s2_from_binhist_impl(H, nullptr, R, S, counts, (hipStream_t)stream);
}

int epg_hist_s2_from_binhist_pair(const uint16_t* HA, const uint16_t* HB, int64_t R, int32_t S, int64_t* counts, void* stream) {
    if (R > 0 && !HB) return fail(EPG_ERR_INVALID_ARG, "hist_s2_pair: HB is NULL");
    return hist_s2_from_binhist_impl(HA, HB, R, S, counts, (hipStream_t)stream);
}

int epg_combine_score_s1(int64_t* counts, int32_t rezero, const uint16_t* H, int64_t R, int32_t N, int32_t S, float* q,
                         double* out64, float* out32, void* ws, int64_t ws_bytes, void* stream) {
    return combine_score_s1_impl(counts, rezero, H, R, N, S, q, out64, out32, ws, ws_bytes, (hipStream_t)stream);
}

int epg_hist_s2(const int8_t* X, int64_t R, int32_t N, int64_t ldx, int32_t S, int64_t* counts, void* ws, int64_t ws_bytes,
                void* stream) {
    if (R > 0 && (!ws || ws_bytes < R * S * 2)) return fail(EPG_ERR_WORKSPACE, "hist_s2: workspace %lld < %lld bytes", (long long)ws_bytes, (long long)(R * S * 2));
    uint16_t* H = reinterpret_cast<uint16_t*>(ws);
    int rc = bin_hist_impl(X, R, N, ldx, S, H, nullptr, (hipStream_t)stream);
    if (rc) return rc;
    return hist_s2_from_binhist_impl(H, nullptr, R, S, counts, (hipStream_t)stream);
}

int epg_hist_s3(const int8_t* X, int64_t R, int32_t N, int64_t ldx, int32_t S, int32_t* counts, void* ws, int64_t ws_bytes,
                void* stream) {
    return hist_s3_impl(X, R, N, ldx, S, counts, ws, ws_bytes, (hipStream_t)stream);
}

int epg_normalise_i64(const int64_t* counts, int64_t n, float* q, void* ws, int64_t ws_bytes, void* stream) {
    return normalise_i64_impl(counts, n, q, ws, ws_bytes, (hipStream_t)stream);
}
int epg_normalise_i32(const int32_t* counts, int64_t n, float* q, void* ws, int64_t ws_bytes, void* stream) {
    return normalise_i32_impl(counts, n, q, ws, ws_bytes, (hipStream_t)stream);
}

int64_t epg_ws_bytes(int32_t saliency, int64_t R, int32_t N, int32_t S) {
    if (R < 0 || N < 1 || S < 1) return fail(EPG_ERR_INVALID_ARG, "ws_bytes: bad shape");
    switch (saliency) {
        case 1: return s1_ws_bytes(R, N, S);
        case 2: return s2_table_bytes(N, S) + align_up(R * S * 2, 256);
        case 3: return s3_ws_bytes(R, N, S);
        default: return fail(EPG_ERR_INVALID_ARG, "ws_bytes: saliency must be 1, 2 or 3");
    }
}

int epg_score_s1(const int8_t* X, int64_t R, int32_t N, int64_t ldx, int32_t S, const float* q, double* out64, float* out32,
                 void* ws, int64_t ws_bytes, void* stream) {
    return score_s1_impl(X, R, N, ldx, S, q, out64, out32, ws, ws_bytes, (hipStream_t)stream);
}

int epg_score_s1_from_binhist(const uint16_t* H, int64_t R, int32_t N, int32_t S, const float* q, double* out64, float* out32,
                              void* ws, int64_t ws_bytes, void* stream) {
    return score_s1_from_hist_impl(H, R, N, S, q, out64, out32, ws, ws_bytes, (hipStream_t)stream);
}

int epg_score_s1_from_binhist_table(const uint16_t* H, int64_t R, int32_t N, int32_t S, const double* T64, const float* T32,
                                    double* out64, float* out32, void* stream) {
    return score_s1_from_hist_table_impl(H, R, N, S, T64, T32, out64, out32, (hipStream_t)stream);
}

int epg_pair_scores_s1_parts(int32_t nparts, const uint16_t* const* HA, const uint16_t* const* HB, const uint16_t* const* HnA,
                             const uint16_t* const* HnB, const int64_t* R, int32_t S, int32_t NA, int32_t NB, int32_t ga, int32_t gb,
                             const float* TA, const float* TB, const float* TnA, const float* TnB, float* const* delta, float* const* null_dist,
                             float* const* dist, int32_t* const* maxdiff, uint8_t* const* mask, int32_t qstate, void* stream) {
    return pair_scores_s1_parts_impl(nparts, HA, HB, HnA, HnB, R, S, NA, NB, ga, gb, TA, TB, TnA, TnB, delta, null_dist, dist, maxdiff, mask, qstate,
                                     (hipStream_t)stream);
}

int epg_pair_scores_s1_from_binhist(const uint16_t* HA, const uint16_t* HB, const uint16_t* HnA, const uint16_t* HnB, int64_t R, int32_t S,
                                    int32_t NA, int32_t NB, int32_t ga, int32_t gb, const float* TA, const float* TB, const float* TnA,
                                    const float* TnB, float* delta, float* null_dist, float* dist, int32_t* maxdiff, void* stream) {
    return pair_scores_s1_impl(HA, HB, HnA, HnB, R, S, NA, NB, ga, gb, TA, TB, TnA, TnB, delta, null_dist, dist, maxdiff, (hipStream_t)stream);
}

int epg_score_s2_from_binhist(const uint16_t* H, int64_t R, int32_t N, int32_t S, int64_t perms, const float* q, double* out64,
                              float* out32, void* ws, int64_t ws_bytes, void* stream) {
    return score_s2_from_hist_impl(H, R, N, S, perms, q, out64, out32, ws, ws_bytes, (hipStream_t)stream);
}

int epg_score_s2(const int8_t* X, int64_t R, int32_t N, int64_t ldx, int32_t S, int64_t perms, const float* q, double* out64,
                 float* out32, void* ws, int64_t ws_bytes, void* stream) {
    const int64_t tb = s2_table_bytes(N, S);
    if (R > 0 && (!ws || ws_bytes < tb + R * S * 2)) return fail(EPG_ERR_WORKSPACE, "score_s2: workspace %lld < %lld bytes", (long long)ws_bytes, (long long)(tb + R * S * 2));
    uint16_t* H = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(ws) + tb);
    int rc = bin_hist_impl(X, R, N, ldx, S, H, nullptr, (hipStream_t)stream);
    if (rc) return rc;
    return score_s2_from_hist_impl(H, R, N, S, perms, q, out64, out32, ws, tb, (hipStream_t)stream);
}

int epg_score_s3(const int8_t* X, int64_t R, int32_t N, int64_t ldx, int32_t S, const float* q, double* out64, float* out32,
                 void* ws, int64_t ws_bytes, void* stream) {
    return score_s3_impl(X, R, N, ldx, S, q, out64, out32, ws, ws_bytes, (hipStream_t)stream);
}

int epg_pair_metrics(const float* delta, int64_t R, int32_t S, int32_t roundtrip, float* dist, int32_t* maxdiff, void* stream) {
    return pair_metrics_impl(delta, R, S, roundtrip, dist, maxdiff, (hipStream_t)stream);
}
int epg_pair_finish(const float* a, const float* b, int64_t R, int32_t S, float* delta, float* signed_sqdist, void* stream) {
    return pair_finish_impl(a, b, R, S, delta, signed_sqdist, (hipStream_t)stream);
}

int epg_quiescent(const int8_t* XA, int32_t NA, int64_t ldxa, const int8_t* XB, int32_t NB, int64_t ldxb, int64_t R,
                  int32_t qstate, uint8_t* mask, void* stream) {
    return quiescent_impl(XA, NA, ldxa, XB, NB, ldxb, R, qstate, mask, (hipStream_t)stream);
}

int epg_null_hist(const int8_t* XA, int32_t NA, int64_t ldxa, const int8_t* XB, int32_t NB, int64_t ldxb, int64_t R, int32_t S,
                  int32_t ga, int32_t gb, uint64_t seed, int64_t row0, uint16_t* HA, uint16_t* HB, void* stream) {
    return null_hist_impl(XA, NA, ldxa, XB, NB, ldxb, R, S, ga, gb, seed, row0, HA, HB, (hipStream_t)stream);
}

int epg_null_hist_from_binhist(const uint16_t* HA, const uint16_t* HB, int64_t R, int32_t S, int32_t n_cols, int32_t ga, int32_t gb,
                               uint64_t seed, int64_t row0, uint16_t* OA, uint16_t* OB, void* stream) {
    return null_hist_from_binhist_impl(HA, HB, R, S, n_cols, ga, gb, seed, row0, OA, OB, (hipStream_t)stream);
}

int epg_quiescent_from_binhist(const uint16_t* HA, const uint16_t* HB, int64_t R, int32_t S, int32_t NA, int32_t NB, int32_t qstate,
                               uint8_t* mask, void* stream) {
    return quiescent_from_binhist_impl(HA, HB, R, S, NA, NB, qstate, mask, (hipStream_t)stream);
}

}  // extern "C"
