// S3 path (biosample-pair saliency) -- placeholder until the tiled kernels land; fails loudly.
#include "epg_common.h"

namespace epg {

int64_t s3_ws_bytes(int64_t, int, int) { return 256; }

int hist_s3_impl(const int8_t*, int64_t, int32_t, int64_t, int32_t, int32_t*, void*, int64_t, hipStream_t) {
    return fail(EPG_ERR_UNSUPPORTED, "hist_s3: not implemented in this build");
}
int score_s3_impl(const int8_t*, int64_t, int32_t, int64_t, int32_t, const float*, double*, float*, void*, int64_t, hipStream_t) {
    return fail(EPG_ERR_UNSUPPORTED, "score_s3: not implemented in this build");
}
int null_hist_impl(const int8_t*, int32_t, int64_t, const int8_t*, int32_t, int64_t, int64_t, int32_t, int32_t, int32_t,
                   uint64_t, int64_t, uint16_t*, uint16_t*, hipStream_t) {
    return fail(EPG_ERR_UNSUPPORTED, "null_hist: not implemented in this build");
}

}  // namespace epg
