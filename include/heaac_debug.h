/* heaac_debug.h -- test-support entry points of libheaac_amd.so.  Not needed to use the library; declared so that
 * nothing the library exports is undeclared (the parity tests that poison the X hand-over workspace use this one:
 * tests/test_he_gpu.py::test_unstored_x_bands_are_never_read). */
#ifndef HEAAC_DEBUG_H
#define HEAAC_DEBUG_H

#include <stddef.h>
#include "heaac_dsp.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Device pointers of workspace set 0 of a device context -- W[chunk][2][32][32][2] floats (the analysis bank's output)
 * and X[chunk][2 channels][38][64][re, im] floats (the hand-over between the HF / PS kernels and the synthesis kernel)
 * -- and the chunk size in frames.  Any of the out pointers may be NULL.  The memory belongs to the context; what it
 * holds between calls is unspecified (the decode calls overwrite it), which is what a test that fills it with NaN
 * relies on.  Returns HEAAC_OK, or HEAAC_ERR_ARG for a NULL context. */
int heaac_debug_workspace(HeaacDevice *dev, float **d_W, float **d_X, size_t *chunk);

/* How many QMF bands of each X row the HF / PS stage stored for the frames of workspace set 0 in the LAST decode call
 * (one byte per frame and output channel: 32, 48 or 64; the bands above are +0 and the synthesis kernel reads them
 * from a page of zeros -- DESIGN.md s4).  Copies 2 * n_frames bytes to host memory after a device synchronisation;
 * n_frames must not exceed the context's chunk.  bench.py reports the shares, since the headline leans on them. */
int heaac_debug_xbands(HeaacDevice *dev, unsigned char *host_out, size_t n_frames);

#ifdef __cplusplus
}
#endif
#endif /* HEAAC_DEBUG_H */
