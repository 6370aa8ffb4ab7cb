/* fuzz_parse.c -- the host parsers under AddressSanitizer / UBSan on mutated access units (TEST harness).
 *
 * Built by tests/test_parse_fuzz.py from the parser SOURCES (aac_parse.c, sbr_parse.c, sbr_header.c) with
 * -fsanitize=address,undefined, so every out-of-bounds access or undefined shift aborts the run.  Input: a file
 * of seed access units (u32 kind, u32 length, bytes; kind 0 = AAC-LC CPE 48 kHz, 1 = HE-AACv1 CPE 24 kHz,
 * 2 = HE-AACv2 SCE 24 kHz, 3 = AAC-LC CPE 48 kHz with coupling / program config elements, 4 = the same around an SCE:
 * kinds 3 and 4 go through heaac_aac_parse_frame_ex; 5 = a 5.1 access unit (SCE CPE CPE LFE, some with SBR payloads) through
 * heaac_aac_parse_frame_layout, its bytes also read as a program config element at a random bit offset; 6 = an access
 * unit of a program-config 5.1 stream with coupling channel elements through heaac_aac_parse_frame_layout_ex, its
 * layout from the AudioSpecificConfig in the one seed of kind 7).  Every 16th iteration instead walks a buffer of seeds behind ADTS
 * headers, mutated the same ways, through heaac_adts_split and checks that the packets tile it.  Each iteration mutates a seed (bit flips, byte noise, truncation, splice of two
 * seeds, pure noise), parses it on a stream that keeps its state across iterations, and checks what the parser
 * promises: whatever the status, the records it wrote pass validate.h.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "heaac_parse.h"
#include "validate.h"

static uint64_t rs = 0x9e3779b97f4a7c15ull;
static uint32_t rnd(void) { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (uint32_t)(rs >> 16); }

typedef struct { uint32_t kind, len; uint8_t *data; } Seed;

/* What the spectral-tools kernel takes on trust from a channel record (k_tools.hip indexes by these): groups and
 * windows that tile the frame, bands inside it, no noise band wider than the 96 lines its generator table spans. */
static int tools_channel_bad(const HeaacToolsChannel *c, int allow_empty)
{
    const HeaacToolsIcs *ic = &c->ics;
    if (allow_empty && ic->num_window_groups == 0 && ic->max_sfb == 0 && ic->num_windows == 0) return 0;
    if ((ic->num_windows != 1 && ic->num_windows != 8) || ic->num_window_groups < 1 || ic->num_window_groups > 8) return 1;
    if (ic->max_sfb > ic->num_swb || ic->num_swb > 63 || ic->num_window_groups * ic->max_sfb > 120) return 2;
    int wins = 0;
    for (int g = 0; g < ic->num_window_groups; g++) wins += ic->group_len[g];
    if (wins != ic->num_windows) return 3;
    for (int i = 0; i < ic->max_sfb; i++) if (ic->swb_offset[i] >= ic->swb_offset[i + 1]) return 4;
    if (ic->swb_offset[ic->max_sfb] > (ic->num_windows == 8 ? 128 : 1024)) return 5;
    int idx = 0;
    for (int g = 0; g < ic->num_window_groups; g++)
        for (int i = 0; i < ic->max_sfb; i++, idx++)
            if (c->band_type[idx] == HEAAC_NOISE_BT && ic->swb_offset[i + 1] - ic->swb_offset[i] > 96) return 6;
    if (c->pred.pred_sfb_max > ic->num_swb && ic->num_windows == 1 && c->pred.pred_sfb_max > 41) return 7;
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 3) { printf("usage: fuzz_parse seeds.bin iterations\n"); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    Seed seed[256];
    int ns = 0;
    HeaacAacConfig ccfg;
    HeaacAacLayout clay0, clay;
    int have_clay = 0;
    while (ns < 256) {
        uint32_t h[2];
        if (fread(h, 4, 2, f) != 2) break;
        seed[ns].kind = h[0]; seed[ns].len = h[1];
        seed[ns].data = malloc(h[1] ? h[1] : 1);
        if (fread(seed[ns].data, 1, h[1], f) != h[1]) return 2;
        if (h[0] == 7) {                                   /* not an access unit: the coupled stream's configuration */
            if (heaac_asc_layout(&ccfg, &clay0, seed[ns].data, (int)h[1]) != 0) return 2;
            have_clay = 1;
            free(seed[ns].data);
            continue;
        }
        ns++;
    }
    fclose(f);
    clay = clay0;
    if (!ns) return 2;
    const long iters = atol(argv[2]);
    HeaacAacConfig cfg[3];
    memset(cfg, 0, sizeof(cfg));
    for (int k = 0; k < 3; k++) {
        cfg[k].object_type = 2; cfg[k].sbr = k ? 1 : 0; cfg[k].ps = k == 2 ? 1 : 0;
        cfg[k].sampling_index = k ? 6 : 3; cfg[k].sample_rate = k ? 24000 : 48000; cfg[k].chan_config = k == 2 ? 1 : 2;
    }
    HeaacSbrHeaderTable *tab = heaac_sbr_table_create(64);       /* small on purpose: the table fills up */
    HeaacAacStream ast[3]; HeaacSbrStream *sst = malloc(3 * sizeof(*sst));
    memset(ast, 0, sizeof(ast)); heaac_sbr_stream_init(sst, 3);
    float *coeffs = malloc(2 * 1024 * sizeof(float));
    HeaacIcs ics[2]; HeaacToolsFrame *tools = malloc(sizeof(*tools));
    HeaacSbrFrame sbr; HeaacPsFrame ps; HeaacAacFrameInfo info;
    long ok = 0, err = 0, bad_records = 0, started = 0, coupled = 0, adts_frames = 0, adts_bad = 0;
    HeaacAacConfig wide[2];
    memset(wide, 0, sizeof(wide));
    for (int k = 0; k < 2; k++) { wide[k].object_type = 2; wide[k].sampling_index = 3; wide[k].sample_rate = 48000; wide[k].chan_config = k ? 1 : 2; }
    HeaacAacStream wst[2];
    memset(wst, 0, sizeof(wst));
    HeaacCceFrame *cce = malloc(HEAAC_MAX_CCE * sizeof(*cce));
    float *cce_coeffs = malloc(HEAAC_MAX_CCE * 1024 * sizeof(float));
    HeaacIcs cce_ics[HEAAC_MAX_CCE];
    HeaacToolsFrame *cce_tools = malloc(HEAAC_MAX_CCE * sizeof(*cce_tools));
    const HeaacCceOut co = { cce, cce_coeffs, cce_ics, cce_tools };
    /* a 5.1 stream (kind 5) */
    HeaacAacConfig lcfg;
    memset(&lcfg, 0, sizeof(lcfg));
    lcfg.object_type = 2; lcfg.sampling_index = 3; lcfg.sample_rate = 48000; lcfg.chan_config = 6;
    HeaacAacLayout lay;
    heaac_aac_layout_default(&lay, 6);
    HeaacAacStream lst[HEAAC_MAX_ELEMENTS];
    memset(lst, 0, sizeof(lst));
    float *lcoeffs = malloc(HEAAC_MAX_ELEMENTS * 2048 * sizeof(float));
    HeaacIcs lics[HEAAC_MAX_ELEMENTS][2];
    HeaacToolsFrame *ltools = malloc(HEAAC_MAX_ELEMENTS * sizeof(*ltools));
    HeaacAacElementInfo lelem[HEAAC_MAX_ELEMENTS];
    long tools_runs = 0, layouts_ok = 0, layout_units = 0, layout_bad = 0, coupled_units = 0, coupled_links = 0;
    HeaacCceFrame *lcce = malloc(HEAAC_MAX_ELEMENTS * HEAAC_MAX_CCE * sizeof(*lcce));
    HeaacAacElementInfo lcce_elem[HEAAC_MAX_CCE];
    const HeaacCceOut lco = { lcce, cce_coeffs, cce_ics, cce_tools, lcce_elem };
    HeaacAacStream cst[HEAAC_MAX_ELEMENTS];
    memset(cst, 0, sizeof(cst));
    for (long it = 0; it < iters; it++) {
        const Seed *s = &seed[rnd() % ns];
        const int k = (int)s->kind;
        /* exact-size heap copy: ASan sees any read past the access unit */
        uint32_t len = s->len;
        const int mode = (int)(rnd() % 6);
        if (mode == 2 && len > 1) len = 1 + rnd() % len;                      /* truncation */
        if (mode == 5) len = 1 + rnd() % 400;                                  /* pure noise */
        uint8_t *au = malloc(len);
        for (uint32_t i = 0; i < len; i++) au[i] = i < s->len ? s->data[i] : 0;
        if (mode == 0) { const int nf = 1 + (int)(rnd() % 8); for (int j = 0; j < nf; j++) au[rnd() % len] ^= (uint8_t)(1u << (rnd() % 8)); }
        if (mode == 1) { const int nb = 1 + (int)(rnd() % 6); for (int j = 0; j < nb; j++) au[rnd() % len] = (uint8_t)rnd(); }
        if (mode == 3) { const Seed *t = &seed[rnd() % ns]; const uint32_t at = rnd() % len;
                         for (uint32_t i = at; i < len; i++) au[i] = t->data[i % t->len]; }
        if (mode == 5) for (uint32_t i = 0; i < len; i++) au[i] = (uint8_t)rnd();
        /* mode 4: the seed unchanged (keeps streams alive between the damaged frames) */
        if (it % 16 == 15) {
            /* the ADTS walk: this unit behind a 7-byte header, junk in front, twice; the packets must tile the buffer */
            const uint32_t flen = len + 7, total = 3 + 2 * flen;
            uint8_t *buf = malloc(total);
            buf[0] = (uint8_t)rnd(); buf[1] = (uint8_t)rnd(); buf[2] = (uint8_t)rnd();
            for (int rep = 0; rep < 2; rep++) {
                uint8_t *h = buf + 3 + rep * flen;
                h[0] = 0xff; h[1] = 0xf1; h[2] = (uint8_t)(0x40 | (3 << 2)); h[3] = (uint8_t)(0x80 | ((flen >> 11) & 3));
                h[4] = (uint8_t)(flen >> 3); h[5] = (uint8_t)(((flen & 7) << 5) | 0x1f); h[6] = 0xfc;
                memcpy(h + 7, au, len);
            }
            if (rnd() & 1) for (int j = 0; j < 4; j++) buf[rnd() % total] = (uint8_t)rnd();
            HeaacAdtsPacket pk[64];
            HeaacAdtsHeader first;
            const long np = heaac_adts_split(buf, total, pk, 64, &first);
            size_t at = 0;
            for (long j = 0; j < np && j < 64; j++) {
                if (pk[j].offset != at || pk[j].size == 0 || pk[j].kind < 0 || pk[j].kind > 3) adts_bad++;
                at += pk[j].size;
                adts_frames += pk[j].kind == HEAAC_ADTS_FRAME;
            }
            if (np < 0 || (np <= 64 && at != total)) adts_bad++;
            (void)heaac_adts_probe(buf, total);
            free(buf); free(au);
            continue;
        }
        if (k == 6) {
            if (!have_clay) return 2;
            memset(&info, 0, sizeof(info));
            const int r = heaac_aac_parse_frame_layout_ex(&ccfg, &clay, cst, au, (int)len, lcoeffs, &lics[0][0], ltools, lelem,
                                                          (rnd() & 7) ? &lco : NULL, &info);
            free(au);
            if (r >= 0) ok++; else err++;
            if (r == HEAAC_PARSE_OK && info.n_cce) {
                coupled_units++;
                /* every output slot's row holds the same coupling elements, each with its own links */
                for (int c2 = 0; c2 < HEAAC_MAX_CCE; c2++) {
                    int v = 0;
                    for (int e = 0; e < clay.n_elements; e++) {
                        const HeaacCceFrame *c = &lcce[e * HEAAC_MAX_CCE + c2];
                        v |= c->present != lcce[c2].present;
                        if (!c->present) continue;
                        v |= c->n_links > HEAAC_MAX_CCE_LINKS || (c->coupling_point != 0 && c->coupling_point != 1 && c->coupling_point != 3) ||
                             c->ics.num_window_groups < 1 || c->ics.num_window_groups > 8 || c->ics.max_sfb > c->ics.num_swb ||
                             c->ics.num_window_groups * c->ics.max_sfb > 120 || c->seq >= HEAAC_MAX_CCE ||
                             c->outputs_before > clay.n_elements || c->elem_id > 15 || c->seq != lcce[c2].seq ||
                             clay.tag_map[HEAAC_ELEM_CCE][c->elem_id] != c2 + 1;
                        v |= clay.elem[e].type == HEAAC_ELEM_LFE && c->n_links;
                        for (int l = 0; l < c->n_links && l < HEAAC_MAX_CCE_LINKS; l++) v |= c->link[l].target_ch >= clay.elem[e].channels;
                        coupled_links += c->n_links;
                    }
                    if (lcce[c2].present && lcce_elem[c2].sbr_payload_bit >= 0)
                        v |= lcce_elem[c2].sbr_payload_bytes < 1 ||
                             (long)lcce_elem[c2].sbr_payload_bit + 8L * lcce_elem[c2].sbr_payload_bytes - 4 > 8L * (long)len;
                    if (v) { layout_bad++; if (layout_bad < 5) printf("iteration %ld: coupling record of a layout out of range\n", it); }
                }
            }
            if (it % 1024 == 1023) { clay = clay0; memset(cst, 0, sizeof(cst)); }
            continue;
        }
        if (k == 5) {
            memset(&info, 0, sizeof(info));
            const int r = heaac_aac_parse_frame_layout(&lcfg, &lay, lst, au, (int)len, lcoeffs, &lics[0][0], ltools, lelem, &info);
            if (r >= 0) ok++; else err++;
            if (r < 0 && (info.refused & HEAAC_REFUSED_RUN_TOOLS)) {
                /* what a refused unit leaves for the spectral tools: the elements marked present, in seq order */
                tools_runs++;
                for (int e = 0; e < lay.n_elements; e++) {
                    if (!lelem[e].present) continue;
                    int v = lelem[e].seq >= lay.n_elements;
                    for (int c = 0; c < lay.elem[e].channels; c++) v |= tools_channel_bad(&ltools[e].ch[c], 1);
                    if (v) { layout_bad++; if (layout_bad < 5) printf("iteration %ld: refused layout unit leaves a tools record out of range\n", it); }
                }
            }
            if (r == HEAAC_PARSE_OK) {
                layout_units++;
                int seen = 0;
                for (int e = 0; e < lay.n_elements; e++) {
                    const HeaacAacElementInfo *ei = &lelem[e];
                    if (!ei->present) continue;
                    seen++;
                    int v = ei->seq >= lay.n_elements || ei->tag > 15 || ei->type > 3;
                    if (ei->sbr_payload_bit >= 0)
                        v |= ei->sbr_payload_bytes < 1 || (long)ei->sbr_payload_bit + 8L * ei->sbr_payload_bytes - 4 > 8L * (long)len;
                    const HeaacToolsIcs *ic = &ltools[e].ch[0].ics;
                    v |= ic->num_window_groups < 1 || ic->num_window_groups > 8 || ic->max_sfb > ic->num_swb ||
                         ic->num_window_groups * ic->max_sfb > 120;
                    if (v) { layout_bad++; if (layout_bad < 5) printf("iteration %ld: layout element out of range\n", it); }
                }
                if (!seen || lay.tags_mapped > 4) layout_bad++;
            }
            if (it % 1024 == 1023) heaac_aac_layout_default(&lay, 6);        /* the stream starts over: tags are learned again */
            /* the same bytes as a program config element somewhere inside them */
            HeaacAacLayout pl;
            int used = 0;
            const int pr = heaac_aac_layout_from_pce(&pl, au, (int)len, (int)(rnd() % (8 * len)), &used);
            if (pr == 0) {
                layouts_ok++;
                int chs = 0, v = pl.n_elements < 0 || pl.n_elements > HEAAC_MAX_ELEMENTS || used <= 0 || used > 8 * (int)len;
                for (int e = 0; e < pl.n_elements && e < HEAAC_MAX_ELEMENTS; e++) {
                    v |= pl.elem[e].first_channel != chs || pl.elem[e].channels != (pl.elem[e].type == HEAAC_ELEM_CPE ? 2 : 1);
                    v |= pl.slot_of[pl.elem[e].type][pl.elem[e].id] != e + 1;
                    chs += pl.elem[e].channels;
                }
                v |= chs != pl.channels || chs > HEAAC_MAX_LAYOUT_CHANNELS;
                if (v) { layout_bad++; if (layout_bad < 5) printf("iteration %ld: layout out of range\n", it); }
            }
            /* and as the first access unit of a stream that configures itself */
            if (heaac_aac_layout_from_au(&pl, au, (int)len) == 0 && (pl.n_elements < 0 || pl.n_elements > HEAAC_MAX_ELEMENTS)) layout_bad++;
            free(au);
            continue;
        }
        if (k >= 3) {
            /* coupling / program config elements: the wide entry; whatever the status, a record that says `present`
             * stays inside its arrays */
            memset(&info, 0, sizeof(info));
            const int r = heaac_aac_parse_frame_ex(&wide[k - 3], &wst[k - 3], au, (int)len, 2, coeffs, ics, tools, &co, &info);
            free(au);
            if (r >= 0) ok++; else err++;
            if (r == HEAAC_PARSE_OK) {
                for (int s2 = 0; s2 < HEAAC_MAX_CCE; s2++) {
                    const HeaacCceFrame *c = &cce[s2];
                    if (!c->present) continue;
                    coupled++;
                    int v = c->n_links > HEAAC_MAX_CCE_LINKS || (c->coupling_point != 0 && c->coupling_point != 1 && c->coupling_point != 3) ||
                            c->ics.num_window_groups < 1 || c->ics.num_window_groups > 8 || c->ics.max_sfb > c->ics.num_swb ||
                            c->ics.num_window_groups * c->ics.max_sfb > 120 || c->seq >= HEAAC_MAX_CCE;
                    int wins = 0;
                    for (int g = 0; g < c->ics.num_window_groups && g < 8; g++) wins += c->ics.group_len[g];
                    v |= wins != c->ics.num_windows;
                    for (int i = 0; i < c->ics.max_sfb && i < 63; i++) v |= c->ics.swb_offset[i] >= c->ics.swb_offset[i + 1];
                    v |= c->ics.max_sfb && c->ics.swb_offset[c->ics.max_sfb] > (c->ics.num_windows == 8 ? 128 : 1024);
                    for (int l = 0; l < c->n_links && l < HEAAC_MAX_CCE_LINKS; l++) v |= c->link[l].target_ch > 1;
                    if (v) { bad_records++; if (bad_records < 5) printf("iteration %ld: coupling record out of range\n", it); }
                }
            }
            continue;
        }
        memset(&sbr, 0, sizeof(sbr)); memset(&ps, 0, sizeof(ps));
        const int r = heaac_heaac_parse_frame(&cfg[k], &ast[k], &sst[k], tab, au, (int)len, coeffs, ics, tools,
                                              &sbr, k == 2 ? &ps : NULL, &info);
        free(au);
        if (r >= 0) ok++; else err++;
        /* records the spectral tools will run on: a parsed unit's, and what a refused unit leaves for them */
        if (info.channels || (info.refused & HEAAC_REFUSED_RUN_TOOLS)) {
            const int nch = info.channels ? info.channels : cfg[k].chan_config;
            tools_runs += !info.channels;
            for (int c = 0; c < nch; c++) {
                const int v = tools_channel_bad(&tools->ch[c], !info.channels);
                if (v) { bad_records++; if (bad_records < 5) printf("iteration %ld: status %d, tools record breaks rule %d\n", it, r, v); }
            }
        }
        if (!info.channels && (info.refused & ~3)) bad_records++;
        if (info.channels) {                                                   /* the core element parsed: records were written */
            int v = heaac_check_sbr_frame(&sbr, heaac_sbr_table_data(tab), heaac_sbr_table_count(tab), info.channels);
            if (!v && k == 2) v = heaac_check_ps_frame(&ps);
            if (v) { bad_records++; if (bad_records < 5) printf("iteration %ld: status %d, record breaks rule %d\n", it, r, v); }
            started += sbr.start;
        }
        if (it % 4096 == 4095) heaac_sbr_stream_init(&sst[rnd() % 3], 1);      /* a stream now and then starts over */
    }
    printf("iterations %ld: parsed %ld, refused %ld, frames with start = 1: %ld, headers %zu, invalid records %ld\n",
           iters, ok, err, started, heaac_sbr_table_count(tab), bad_records);
    printf("refused units that leave work for the spectral tools %ld\n", tools_runs);
    printf("coupling elements parsed %ld, ADTS frames delivered %ld, ADTS walks that did not tile %ld\n", coupled, adts_frames, adts_bad);
    printf("5.1 units parsed %ld, program config layouts accepted %ld, out of range %ld\n", layout_units, layouts_ok, layout_bad);
    printf("coupled layout units parsed %ld, gain lists landed %ld\n", coupled_units, coupled_links);
    if (adts_bad || layout_bad) return 1;
    free(lcoeffs); free(ltools); free(lcce);
    free(cce); free(cce_coeffs); free(cce_tools);
    heaac_sbr_table_destroy(tab);
    free(sst); free(coeffs); free(tools);
    for (int i = 0; i < ns; i++) free(seed[i].data);
    if (bad_records) return 1;
    printf("ok\n");
    return 0;
}
