/* Mints tests/golden/cce_gains.json: every coupling gain a coupling channel element can transmit, formed from the
 * reference's expressions WITH ITS DECLARED TYPES (libavcodec/aacdec.c:1508 `float scale;`, :1528
 * `scale = pow(2., pow(2., (int)get_bits(gb, 2) - 3));`, :1534 `float gain_cache`, :1539 `gain_cache = pow(scale, -gain);`,
 * :1556 `gain_cache = pow(scale, -t) * s;`) -- written from the reference text, sharing nothing with csrc/aac_parse.c or
 * the Python bit writer.   gcc -O2 -std=c99 -ffp-contract=off make_cce_gains.c -lm && ./a.out > cce_gains.json */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

static uint32_t bits_of(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

int main(void)
{
    printf("{\n \"note\": \"float bit patterns; rows = gain_element_scale 0..3, columns = step -128..128\",\n");
    for (int variant = 0; variant < 2; variant++) {
        printf(" \"%s\": [\n", variant ? "negative" : "positive");
        for (int idx = 0; idx < 4; idx++) {
            volatile float scale;
            scale = pow(2., pow(2., idx - 3));
            printf("  [");
            for (int step = -128; step <= 128; step++) {
                float gain_cache;
                if (!variant) gain_cache = pow(scale, -step);
                else { int s = -1; gain_cache = pow(scale, -step) * s; }
                printf("%u%s", bits_of(gain_cache), step < 128 ? ", " : "");
            }
            printf("]%s\n", idx < 3 ? "," : "");
        }
        printf(" ]%s\n", variant ? "" : ",");
    }
    printf("}\n");
    return 0;
}
