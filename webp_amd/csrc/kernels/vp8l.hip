// placeholder until the VP8L kernels land (next commit)
#include "wp_device.h"
extern "C" int wpk_launch_vp8l(int, const wp_vp8l_frame*, const wp_vp8l_frame*, uint8_t*, const uint8_t*, uint8_t*, uint8_t*, int, hipStream_t) { return int(hipErrorNotSupported); }
extern "C" int wpk_launch_alpha(int, const wp_alpha_frame*, const wp_alpha_frame*, const uint8_t*, uint8_t*, hipStream_t) { return int(hipErrorNotSupported); }
