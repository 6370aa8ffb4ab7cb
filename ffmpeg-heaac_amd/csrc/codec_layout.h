// codec_layout.h -- internal: the AVCodec-shaped decoder for streams whose access units carry several output
// elements (channel configurations 3..7, program config elements).  Used by shim.hip; not part of include/*.h.
#pragma once
#include <stdint.h>
#include "heaac_dsp.h"
#include "heaac_parse.h"

struct HeaacLayoutDec;

// What one decoded access unit tells the caller (aacdec.c:2080-2094)
struct HeaacLayoutOut {
    int channels, frame_size, sample_rate;
    int64_t channel_layout;
};

// SBR output mode of a configuration (shim.hip): 0 = 2048 samples at twice the core rate, 1 = "downsampled SBR",
// 1024 samples at the core rate, -1 = the two rates contradict each other
int heaac_sbr_output_mode(const HeaacAacConfig *m);

HeaacLayoutDec *heaac_layout_dec_create(HeaacDevice *dev, const HeaacAacConfig *m4ac, const HeaacAacLayout *layout);
void heaac_layout_dec_destroy(HeaacLayoutDec *d);
// avctx->channels: the layout's channels, plus one for every SCE that carries Parametric Stereo (aacdec.c:203-206)
int heaac_layout_dec_channels(const HeaacLayoutDec *d);
// One access unit -> interleaved int16 in `data` (host).  Returns the bytes consumed (aacdec.c:2102-2107) or -1.
int heaac_layout_dec_frame(HeaacLayoutDec *d, const uint8_t *buf, int size, void *data, int *data_size,
                           HeaacLayoutOut *out);
