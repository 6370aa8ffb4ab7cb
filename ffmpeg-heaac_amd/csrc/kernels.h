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

int heaac_launch_couple(int channels, float *d_pcm, const float *d_cce, const HeaacCoupling *d_cpl,
                        int16_t *d_s16, size_t n, hipStream_t s);

int heaac_launch_interleave(int channels, const HeaacPlaneRef *planes, int len, int pcm_format, int16_t *d_out,
                            size_t n, hipStream_t s);
}

extern "C" {
int heaac_launch_he(const float *d_tab, const uint16_t *d_rev, int cfg,
                    const float *d_coeffs, const HeaacIcs *d_ics,
                    const HeaacSbrFrame *d_sbr, const HeaacSbrHeader *d_hdr, unsigned n_hdr,
                    const HeaacPsFrame *d_ps,
                    const float *d_state_in, float *d_state_out,
                    void *d_pcm, int pcm_format,
                    float *d_ws_W, float *d_ws_X, unsigned *d_queue,
                    unsigned char *d_xtop, const float *d_zero,
                    size_t n, size_t pcm_frame0, int flags, hipStream_t s);

int heaac_launch_ps(const float *d_tab, const HeaacPsFrame *d_ps, const HeaacSbrFrame *d_sbr,
                    const HeaacSbrHeader *d_hdr, unsigned n_hdr, const float *d_state_in, float *d_state_out,
                    int state_words, int off_ps, float *d_ws_X, size_t n, int variants, hipStream_t s);

int heaac_launch_hfps(const float *d_tab, const HeaacSbrFrame *d_sbr, const HeaacSbrHeader *d_hdr,
                      unsigned n_hdr, const HeaacPsFrame *d_ps, const float *d_ws_W,
                      const float *d_state_in, float *d_state_out, int state_words,
                      int off_sbr, int off_ps, float *d_ws_X, size_t n, unsigned *d_queue, unsigned char *d_xtop,
                      hipStream_t s);

int heaac_launch_qmf_analysis(const float *d_tab, const float *d_in, const float *d_xh_in,
                              float *d_xh_out, float *d_W, float scale, size_t n, hipStream_t s);

int heaac_launch_qmf_synthesis(const float *d_tab, const float *d_X, const float *d_v_in,
                               float *d_v_out, float *d_out, float scale, float bias,
                               size_t n, hipStream_t s);

int heaac_launch_qmf_synthesis_ds(const float *d_tab, const float *d_X, const float *d_v_in,
                                  float *d_v_out, float *d_out, float scale, float bias,
                                  size_t n, hipStream_t s);
}

extern "C" {
int heaac_launch_spectral_tools(int channels, float *d_coeffs, const HeaacToolsFrame *d_tools,
                                const int *d_rng_in, int *d_rng_out,
                                const HeaacPredictorState *d_pred_in, HeaacPredictorState *d_pred_out,
                                int stages, const HeaacCceFrame *d_cce, const float *d_cce_coeffs, int n_cce,
                                size_t n, hipStream_t s);
int heaac_launch_fft_calc(const float *d_tab, int nbits, float *d_z, size_t n, hipStream_t s);
int heaac_launch_imdct_mirror(float *d_out, int n, size_t count, hipStream_t s);
}
