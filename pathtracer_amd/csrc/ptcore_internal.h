// What the translation units of libptcore share beyond include/ptcore.h.  Not exported.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ptcore.h"

#define PTC_HIDDEN __attribute__((visibility("hidden")))

// The context's film: device accumulator (W*H*3 f32; null before pt_set_camera), pixel count, passes
// accumulated so far, and the stream its read-side work runs on.
PTC_HIDDEN float *ptc_accum(pt_context *ctx);
PTC_HIDDEN uint64_t ptc_npix(const pt_context *ctx);
PTC_HIDDEN uint32_t ptc_n_samples(const pt_context *ctx);
PTC_HIDDEN hipStream_t ptc_stream(pt_context *ctx);
PTC_HIDDEN int ptc_device(const pt_context *ctx);
// ColorOps::to_rgb of accum_device / n_samples (an accumulator of the context's resolution on the context's
// device, not necessarily its own) into host memory.
PTC_HIDDEN int ptc_tonemap(pt_context *ctx, const float *accum_device, uint32_t n_samples, uint8_t *rgb_host);
