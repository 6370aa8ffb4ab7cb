// k_ps.hip -- Parametric Stereo kernel (placeholder until the kernel lands).
#include "kernels.h"
extern "C" int heaac_launch_ps(const float *d_tab, const HeaacPsFrame *d_ps, const HeaacSbrFrame *d_sbr,
                               const HeaacSbrHeader *d_hdr, const float *d_state_in, float *d_state_out,
                               int state_words, int off_ps, float *d_ws_X, size_t n, hipStream_t s)
{
    return HEAAC_ERR_ARG;
}
