/* heaac_fft.h -- the reference's transform plugin surface, gfx950-backed.
 *
 * Field-compatible restatement of libavcodec/fft.h:32-53 (struct FFTContext),
 * libavcodec/avfft.h:22-26 (FFTSample, FFTComplex) and the entry points an AAC
 * decoder links against:
 *
 *   ff_fft_init / ff_fft_end          libavcodec/fft.c:81-176, :204-210
 *   ff_fft_permute / ff_fft_calc      libavcodec/fft.h:122-133 (trampolines)
 *   ff_mdct_init / ff_mdct_end        libavcodec/mdct.c:61-105, :228-232
 *   ff_imdct_half / ff_imdct_calc     libavcodec/fft.h:138-145 -> mdct.c:124-179
 *   ff_kbd_window_init                libavcodec/mdct.c:35-54
 *   ff_sine_window_init,
 *   ff_init_ff_sine_windows           libavcodec/mdct_tablegen.h:49-59
 *   av_fft_init / av_fft_permute /
 *   av_fft_calc / av_fft_end          libavcodec/avfft.h:35-49, avfft.c:25-51
 *   av_mdct_init / av_imdct_half /
 *   av_imdct_calc / av_mdct_calc /
 *   av_mdct_end                       libavcodec/avfft.h:51-55, avfft.c:55-90
 *
 * The reference lets per-arch code overwrite the function pointers in
 * ff_fft_init (fft.c:113-115: ff_fft_init_arm / _altivec / _mmx).  This build
 * is one more such backend: the pointers are set to HIP-backed functions
 * (hand-written gfx950 kernels), permutation stays FF_MDCT_PERM_NONE, tables
 * (revtab, tcos, tsin) are filled exactly as the C reference fills them.
 *
 * Scope: the transforms of the HE-AAC decode path only --
 *   ff_fft_init:  inverse = 1, nbits 5, 6, 9
 *   ff_mdct_init: inverse = 1, (nbits, scale) in {(11,1.0), (8,1.0), (7,1/64), (7,-2.0)}
 * anything else returns -1 (no CPU fallback exists in this library).
 *
 * The per-call entry points take HOST pointers, like the reference, and run a
 * batch of one on the GPU (copy in, one kernel, copy out, synchronise): correct
 * but latency-bound.  The throughput path is heaac_imdct_half_batch() /
 * heaac_lc_decode_batch() / heaac_he_decode_batch() in heaac_dsp.h.
 */
#ifndef HEAAC_FFT_H
#define HEAAC_FFT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef float FFTSample;

typedef struct FFTComplex {
    FFTSample re, im;
} FFTComplex;

typedef struct FFTContext FFTContext;

struct FFTContext {
    int nbits;
    int inverse;
    uint16_t *revtab;
    FFTComplex *exptab;
    FFTComplex *exptab1;
    FFTComplex *tmp_buf;
    int mdct_size;
    int mdct_bits;
    FFTSample *tcos;
    FFTSample *tsin;
    void (*fft_permute)(struct FFTContext *s, FFTComplex *z);
    void (*fft_calc)(struct FFTContext *s, FFTComplex *z);
    void (*imdct_calc)(struct FFTContext *s, FFTSample *output, const FFTSample *input);
    void (*imdct_half)(struct FFTContext *s, FFTSample *output, const FFTSample *input);
    void (*mdct_calc)(struct FFTContext *s, FFTSample *output, const FFTSample *input);
    int split_radix;
    int permutation;
#define FF_MDCT_PERM_NONE       0
#define FF_MDCT_PERM_INTERLEAVE 1
};

int  ff_fft_init(FFTContext *s, int nbits, int inverse);
void ff_fft_end(FFTContext *s);
void ff_fft_permute(FFTContext *s, FFTComplex *z);
void ff_fft_calc(FFTContext *s, FFTComplex *z);

int  ff_mdct_init(FFTContext *s, int nbits, int inverse, double scale);
void ff_mdct_end(FFTContext *s);
void ff_imdct_half(FFTContext *s, FFTSample *output, const FFTSample *input);
void ff_imdct_calc(FFTContext *s, FFTSample *output, const FFTSample *input);

void ff_kbd_window_init(float *window, float alpha, int n);
void ff_sine_window_init(float *window, int n);
void ff_init_ff_sine_windows(int index);
extern float *const ff_sine_windows[13];   /* entries 7 (128) and 10 (1024) are backed */

FFTContext *av_fft_init(int nbits, int inverse);
void av_fft_permute(FFTContext *s, FFTComplex *z);
void av_fft_calc(FFTContext *s, FFTComplex *z);
void av_fft_end(FFTContext *s);

FFTContext *av_mdct_init(int nbits, int inverse, double scale);
void av_imdct_calc(FFTContext *s, FFTSample *output, const FFTSample *input);
void av_imdct_half(FFTContext *s, FFTSample *output, const FFTSample *input);
void av_mdct_calc(FFTContext *s, FFTSample *output, const FFTSample *input);   /* forward: not on the decode path, aborts */
void av_mdct_end(FFTContext *s);

#ifdef __cplusplus
}
#endif
#endif /* HEAAC_FFT_H */
