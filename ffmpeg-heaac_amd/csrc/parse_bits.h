/* parse_bits.h -- bit reader and code trees shared by the host parsers (aac_parse.c, sbr_parse.c).
 * Stands where the reference uses get_bits.h (GetBitContext, get_bits, get_vlc2) and bitstream.c's
 * init_vlc: a binary tree built from the ISO (code, length) pairs, entered through a 10-bit prefix table
 * derived from the tree; no table of the reference's VLC layout exists here. */
#ifndef HEAAC_PARSE_BITS_H
#define HEAAC_PARSE_BITS_H
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* bit reader (MSB first); reading past the end yields zeros and sets `over`                     */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    const uint8_t *buf;
    int size_bits, pos, over;
} Bits;

static void bits_init(Bits *b, const uint8_t *buf, int bytes)
{
    b->buf = buf; b->size_bits = bytes * 8; b->pos = 0; b->over = 0;
}
/* the next 32 bits (zeros past the end), no side effects */
static inline uint32_t peek32(const Bits *b)
{
    const int byte = b->pos >> 3, sh = b->pos & 7, size_bytes = b->size_bits >> 3;
    uint64_t v = 0;
    if (byte + 8 <= size_bytes) {
        memcpy(&v, b->buf + byte, 8);
        v = __builtin_bswap64(v);
    } else {
        for (int i = 0; i < 8; i++) {
            v <<= 8;
            if (byte + i < size_bytes) v |= b->buf[byte + i];
        }
    }
    return (uint32_t)((v << sh) >> 32);
}
static inline void skip(Bits *b, int n)
{
    b->pos += n;
    if (b->pos > b->size_bits) b->over = 1;
}
static inline unsigned bits(Bits *b, int n)          /* n <= 25 */
{
    if (n <= 0) return 0;
    const unsigned v = peek32(b) >> (32 - n);
    skip(b, n);
    return v;
}
static inline unsigned bit1(Bits *b) { return bits(b, 1); }
static inline unsigned peek(Bits *b, int n) { return n > 0 ? peek32(b) >> (32 - n) : 0; }
static inline int bits_left(const Bits *b) { return b->size_bits - b->pos; }

/* ------------------------------------------------------------------------------------------ */
/* code trees                                                                                    */
/* ------------------------------------------------------------------------------------------ */
typedef struct { int16_t child[2]; } Node;           /* >= 0: node index, < 0: -(symbol + 1), 0 at root only */
#define TREE_LUT_BITS 10
#define TREE_MAX_SYMBOLS 289                          /* the largest code book of the path (AAC spectral books 1-4: 81, 5-6: 81,
                                                         7-8: 64, 9-10: 169, 11: 289; SBR <= 121; PS <= 61) */
#define TREE_MAX_NODES (2 * TREE_MAX_SYMBOLS + 2)
/* lut[prefix]: length << 16 | symbol for a code of <= TREE_LUT_BITS bits; 0xff << 16 for a prefix that is no code;
 * otherwise (length 0) the node reached after TREE_LUT_BITS bits, from where the walk goes on bit by bit.
 * Fixed storage: building a tree allocates nothing, so it cannot fail half-way inside pthread_once. */
typedef struct { Node n[TREE_MAX_NODES]; int count; uint32_t lut[1u << TREE_LUT_BITS]; } Tree;

/* Returns 0, or -1 if the book does not fit the fixed storage (a table-generation error, caught by the tests). */
static inline int tree_build(Tree *t, const uint32_t *code32, const uint16_t *code16, const uint8_t *len, int n)
{
    memset(t, 0, sizeof(*t));
    if (n > TREE_MAX_SYMBOLS) return -1;
    t->count = 1;
    for (int s = 0; s < n; s++) {
        const uint32_t c = code32 ? code32[s] : code16[s];
        int at = 0;
        for (int i = len[s] - 1; i >= 0; i--) {
            const int bit = (c >> i) & 1;
            if (i == 0) {
                t->n[at].child[bit] = (int16_t)-(s + 1);
            } else {
                if (t->n[at].child[bit] <= 0) {
                    if (t->count >= TREE_MAX_NODES) return -1;
                    t->n[at].child[bit] = (int16_t)t->count;
                    t->count++;
                }
                at = t->n[at].child[bit];
            }
        }
    }
    for (uint32_t p = 0; p < (1u << TREE_LUT_BITS); p++) {
        int at = 0;
        uint32_t e = 0;
        for (int d = 0; d < TREE_LUT_BITS; d++) {
            const int c = t->n[at].child[(p >> (TREE_LUT_BITS - 1 - d)) & 1];
            if (c < 0) { e = ((uint32_t)(d + 1) << 16) | (uint32_t)(-c - 1); break; }
            if (c == 0) { e = 0xffu << 16; break; }
            at = c;
            e = (uint32_t)at;
        }
        t->lut[p] = e;
    }
    return 0;
}

static inline int tree_read(const Tree *t, Bits *b)
{
    const uint32_t w = peek32(b);
    const uint32_t e = t->lut[w >> (32 - TREE_LUT_BITS)];
    const unsigned len = e >> 16;
    if (len == 0xff) return -1;                       /* not a code of this book */
    if (len) { skip(b, (int)len); return (int)(e & 0xffff); }
    int at = (int)e;
    for (int d = TREE_LUT_BITS; d < 32; d++) {
        const int c = t->n[at].child[(w >> (31 - d)) & 1];
        if (c < 0) { skip(b, d + 1); return -c - 1; }
        if (c == 0) return -1;
        at = c;
    }
    return -1;
}

#endif
