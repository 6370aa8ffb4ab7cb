// kernels.h -- internal launch functions (one per .hip translation unit).
// Not part of the public ABI; include/heaac_dsp.h is.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include "heaac_dsp.h"

extern "C" {
int heaac_launch_lc(const float *d_tab, const uint16_t *d_rev, int channels,
                    const float *d_coeffs, const HeaacIcs *d_ics,
                    const float *d_state_in, float *d_state_out,
                    void *d_pcm, int pcm_format, size_t n, hipStream_t s);

int heaac_launch_imdct_half(const float *d_tab, const uint16_t *d_rev, int which,
                            float *d_out, const float *d_in, size_t n, hipStream_t s);
}
