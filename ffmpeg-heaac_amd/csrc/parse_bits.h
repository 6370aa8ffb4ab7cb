/* parse_bits.h -- bit reader and code trees shared by the host parsers (aac_parse.c, sbr_parse.c).
 * Stands where the reference uses get_bits.h (GetBitContext, get_bits, get_vlc2) and bitstream.c's
 * init_vlc: the codes are walked bit by bit through a binary tree built from the ISO (code, length)
 * pairs, so no table of the reference's VLC layout exists here. */
#ifndef HEAAC_PARSE_BITS_H
#define HEAAC_PARSE_BITS_H
#include <stdint.h>
#include <stdlib.h>

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
static inline unsigned bit1(Bits *b)
{
    if (b->pos >= b->size_bits) { b->over = 1; b->pos++; return 0; }
    const unsigned v = (b->buf[b->pos >> 3] >> (7 - (b->pos & 7))) & 1;
    b->pos++;
    return v;
}
static inline unsigned bits(Bits *b, int n)          /* n <= 25 */
{
    unsigned v = 0;
    while (n-- > 0) v = (v << 1) | bit1(b);
    return v;
}
static inline unsigned peek(Bits *b, int n)
{
    Bits t = *b;
    return bits(&t, n);
}
static inline int bits_left(const Bits *b) { return b->size_bits - b->pos; }

/* ------------------------------------------------------------------------------------------ */
/* code trees                                                                                    */
/* ------------------------------------------------------------------------------------------ */
typedef struct { int16_t child[2]; } Node;           /* >= 0: node index, < 0: -(symbol + 1), 0 at root only */
typedef struct { Node *n; int count; } Tree;

static inline void tree_build(Tree *t, const uint32_t *code32, const uint16_t *code16, const uint8_t *len, int n)
{
    int cap = 2 * n + 2;
    t->n = (Node *)calloc(cap, sizeof(Node));
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
                    t->n[at].child[bit] = (int16_t)t->count;
                    t->count++;
                }
                at = t->n[at].child[bit];
            }
        }
    }
}

static inline int tree_read(const Tree *t, Bits *b)
{
    int at = 0;
    for (int depth = 0; depth < 24; depth++) {
        const int c = t->n[at].child[bit1(b)];
        if (c < 0) return -c - 1;
        if (c == 0) return -1;                        /* not a code of this book */
        at = c;
    }
    return -1;
}

#endif
