/* tables.h -- immutable tables of the HE-AAC DSP path, as one flat float blob.
 *
 * Built once on the host in double precision with libm and rounded to float
 * exactly where the reference rounds (see tables.c), uploaded once per device,
 * staged into LDS by the kernels.  Offsets are in floats from the blob start
 * and are all multiples of 4 (16-byte aligned) so kernels can use float4.
 */
#ifndef HEAAC_TABLES_H
#define HEAAC_TABLES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* FFT twiddles: ff_cos_N[0..N/4] (fft.c:67-79). */
#define TB_COS16      0        /*   5 -> 8   */
#define TB_COS32      8        /*   9 -> 12  */
#define TB_COS64      20       /*  17 -> 20  */
#define TB_COS128     40       /*  33 -> 36  */
#define TB_COS256     76       /*  65 -> 68  */
#define TB_COS512     144      /* 129 -> 132 */
/* MDCT rotation tables tcos[n/4] then tsin[n/4] (mdct.c:94-100). */
#define TB_ROT2048    276      /* 512 + 512  */
#define TB_ROT256     1300     /*  64 +  64  */
#define TB_ROT128S    1428     /*  32 +  32   scale 1/64 (SBR synthesis) */
#define TB_ROT128A    1492     /*  32 +  32   scale -2   (SBR analysis)  */
/* AAC windows (aacdec.c:593-596). */
#define TB_KBD_LONG   1556     /* 1024 */
#define TB_SINE_LONG  2580     /* 1024 */
#define TB_KBD_SHORT  3604     /*  128 */
#define TB_SINE_SHORT 3732     /*  128 */
/* SBR QMF prototype (aacsbr.c:117-123) and noise table (aacsbrdata.h:355). */
#define TB_QMF_US     3860     /*  640 */
#define TB_QMF_DS     4500     /*  320 */
#define TB_NOISE      4820     /* 1024 (re,im pairs) */
/* Parametric Stereo (aacps_tablegen.h:80-209, aacpsdata.c:160-163). */
#define TB_PD_RE      5844     /*  512 */
#define TB_PD_IM      6356     /*  512 */
#define TB_HA         6868     /* 46*8*4 = 1472 */
#define TB_HB         8340     /* 1472 */
#define TB_F20_0_8    9812     /* 8*7*2  = 112 */
#define TB_F34_0_12   9924     /* 12*7*2 = 168 */
#define TB_F34_1_8    10092    /* 112 */
#define TB_F34_2_4    10204    /* 4*7*2 = 56 */
#define TB_QFRACT     10260    /* 2*50*3*2 = 600 */
#define TB_PHIFRACT   10860    /* 2*50*2 = 200 */
#define TB_G1_Q2      11060    /* 7 -> 8 */
/* The MDCT pre-rotation twiddles in the order the register FFT's first layout takes them (k_core2.h, layout A):
 * entry (revtab[k] & 15) * S + (revtab[k] >> 4) = (tcos[k], tsin[k]), S = 32 for N = 2048, 4 for N = 256. */
#define TB_ROTA512    11068    /* 512 (re,im) pairs */
#define TB_ROTA64     12092    /*  64 pairs */
#define TB_TOTAL      12220

/* Split-radix input permutations revtab (fft.c:121-122), as uint16. */
#define RV_512        0
#define RV_64         512
#define RV_32         576
#define RV_TOTAL      608

typedef struct HeaacHostTables {
    float    f[TB_TOTAL];
    uint16_t rev[RV_TOTAL];
} HeaacHostTables;

/* Fill *t.  Pure host code, deterministic. */
void heaac_build_tables(HeaacHostTables *t);

/* Named access for tests (returns count written or -1). */
int heaac_get_table(const char *name, float *dst, int max);

#ifdef __cplusplus
}
#endif
#endif
