/* adts_split.c -- a raw ADTS (.aac) byte buffer -> access units.
 *
 * What it stands for in the reference:
 *   adts_aac_probe / adts_aac_read_header     libavformat/raw.c:666-717   (ID3v2 tag stepped over, score from
 *                                                                          runs of consecutive headers)
 *   ff_aac_ac3_parse + aac_sync               libavcodec/aac_ac3_parser.c:26-100, aac_parser.c:72-95
 *                                             (frame by frame behind each other; bytes no header claims are
 *                                              handed on as packets of their own)
 *   ff_aac_parse_header                       libavcodec/aac_parser.c:29-70 (= heaac_adts_parse_header)
 *
 * The reference's parser is a byte-at-a-time state machine over a stream it cannot look ahead in: after damage it
 * takes the first seven bytes that parse as a header.  This splitter sees the whole buffer, so a header found
 * anywhere but directly behind a good frame is only believed if the probe's rule holds for it -- another header
 * (or the end of the buffer) sits exactly one frame length further on.  Directly behind a good frame a header is
 * accepted as the reference accepts it.  The spans between frames come out as packets of kind JUNK (the reference
 * passes them to the decoder, which refuses them), a last frame cut short by the end of the buffer as TRUNCATED.
 */
#include <string.h>
#include "heaac_parse.h"

/* ff_id3v2_match / ff_id3v2_tag_len (libavformat/id3v2.c:27-50) */
static size_t id3v2_len(const uint8_t *b, size_t size)
{
    if (size < 10 || b[0] != 'I' || b[1] != 'D' || b[2] != '3' || b[3] == 0xff || b[4] == 0xff ||
        ((b[6] | b[7] | b[8] | b[9]) & 0x80))
        return 0;
    size_t len = ((size_t)(b[6] & 0x7f) << 21) + ((size_t)(b[7] & 0x7f) << 14) + ((size_t)(b[8] & 0x7f) << 7) +
                 (b[9] & 0x7f) + 10;
    if (b[5] & 0x10) len += 10;                        /* footer */
    return len;
}

/* frame length if a header starts at `at`, else 0 */
static int header_at(const uint8_t *buf, size_t size, size_t at, HeaacAdtsHeader *h)
{
    HeaacAdtsHeader tmp;
    if (at + 7 > size) return 0;
    if (heaac_adts_parse_header(h ? h : &tmp, buf + at, 7) < 0) return 0;
    return (h ? h : &tmp)->frame_length;
}

int heaac_adts_probe(const uint8_t *buf, size_t size)
{
    /* raw.c:666-702: runs of headers, each exactly one frame length behind the other.  The probe's own header
     * test: sync word, layer 0, frame length >= 7 */
    if (!buf || size < 8) return 0;
    size_t first = id3v2_len(buf, size);
    if (first >= size) return 0;
    const size_t end = size - 7;
    int max_frames = 0, first_frames = 0;
    for (size_t start = first; start < end; ) {
        size_t at = start;
        int frames = 0;
        while (at < end) {
            const unsigned sync = ((unsigned)buf[at] << 8) | buf[at + 1];
            if ((sync & 0xFFF6) != 0xFFF0) break;
            const unsigned flen = ((((unsigned)buf[at + 3] << 24) | ((unsigned)buf[at + 4] << 16) |
                                    ((unsigned)buf[at + 5] << 8) | buf[at + 6]) >> 13) & 0x8FFF;
            if (flen < 7) break;
            at += flen;
            frames++;
        }
        if (frames > max_frames) max_frames = frames;
        if (start == first) first_frames = frames;
        start = at + 1;
    }
    /* AVPROBE_SCORE_MAX = 100 */
    if (first_frames >= 3) return 51;
    if (max_frames > 500) return 50;
    if (max_frames >= 3) return 25;
    return max_frames >= 1 ? 1 : 0;
}

static void put(HeaacAdtsPacket *out, size_t max_out, size_t *n, size_t offset, size_t size, int kind, int header)
{
    if (out && *n < max_out) {
        out[*n].offset = offset;
        out[*n].size = size;
        out[*n].kind = kind;
        out[*n].header_size = header;
    }
    (*n)++;
}

long heaac_adts_split(const uint8_t *buf, size_t size, HeaacAdtsPacket *out, size_t max_out, HeaacAdtsHeader *first_header)
{
    if (!buf && size) return HEAAC_PARSE_ERR_ARG;
    size_t n = 0, at = id3v2_len(buf, size);
    if (at > size) at = size;
    if (at) put(out, max_out, &n, 0, at, HEAAC_ADTS_TAG, 0);
    int in_step = 1, have_first = 0;                   /* in_step: `at` is where a header is expected */
    size_t junk_from = at;
    while (at < size) {
        HeaacAdtsHeader h;
        int flen = header_at(buf, size, at, &h);
        if (flen && !in_step) {
            /* a candidate found while searching: the header behind it must be one too, or the candidate ends
             * exactly where the buffer ends (the last frame).  One that runs past the end cannot be told from
             * payload bytes and stays junk; only a frame directly behind a good one is reported as cut short. */
            const size_t next = at + (size_t)flen;
            if (next != size && !header_at(buf, size, next, NULL)) flen = 0;
        }
        if (!flen) {
            in_step = 0;
            at++;
            continue;
        }
        if (at > junk_from) put(out, max_out, &n, junk_from, at - junk_from, HEAAC_ADTS_JUNK, 0);
        if (!have_first && first_header) *first_header = h;
        have_first = 1;
        const int hs = h.crc_absent ? 7 : 9;
        if (at + (size_t)flen > size) {
            put(out, max_out, &n, at, size - at, HEAAC_ADTS_TRUNCATED, hs);
            at = size;
        } else {
            put(out, max_out, &n, at, (size_t)flen, HEAAC_ADTS_FRAME, hs);
            at += (size_t)flen;
        }
        junk_from = at;
        in_step = 1;
    }
    if (size > junk_from) put(out, max_out, &n, junk_from, size - junk_from, HEAAC_ADTS_JUNK, 0);
    return (long)n;
}
