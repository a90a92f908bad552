/*
 * qmm_oracle.c — CPU oracle (test infrastructure only, see qmm_oracle.h).
 *
 * Restates, in plain C and in the reference's generic summation order, the arithmetic of the
 * quantized MUL_MAT / MUL_MAT_ID path of the ggml CPU backend, which is the semantic oracle of
 * zhouwg/ggml-hexagon's cDSP mulmat (ggml-dsp.c:1091-1351 is a strip of ggml-cpu.c:6655-6937).
 * Blocks are addressed by byte offset (layouts: ggml/src/ggml-common.h:167-172, 209-214, 285-334).
 *
 * Compiled with -ffp-contract=off so that every f32 operation rounds exactly once, as the
 * reference's scalar C does when built without FMA contraction.
 */
#include "qmm_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- type table */

int qmo_blck_size(int type) {
    switch (type) {
        case QMO_Q4_0: case QMO_Q8_0: case QMO_Q4_1: case QMO_Q5_0: case QMO_Q5_1: case QMO_Q8_1: case QMO_IQ4_NL: return 32;
        case QMO_Q2_K: case QMO_Q3_K: case QMO_Q4_K: case QMO_Q5_K: case QMO_Q6_K: case QMO_Q8_K: case QMO_IQ4_XS: return 256;
        default: return 0;
    }
}

size_t qmo_type_size(int type) {
    switch (type) {
        case QMO_Q4_0: return 18;   /* f16 d, 16 nibble bytes                       ggml-common.h:167-172 */
        case QMO_Q8_0: return 34;   /* f16 d, 32 int8                               ggml-common.h:209-214 */
        case QMO_Q4_1: return 20;   /* f16 d, f16 m, 16 nibble bytes                ggml-common.h:174-185 */
        case QMO_Q5_0: return 22;   /* f16 d, 4 high-bit bytes, 16 nibble bytes     ggml-common.h:187-193 */
        case QMO_Q5_1: return 24;   /* f16 d, f16 m, 4 high-bit bytes, 16 nibbles   ggml-common.h:195-207 */
        case QMO_Q8_1: return 36;   /* f16 d, f16 s = d * sum(qs), 32 int8          ggml-common.h:216-227 */
        case QMO_IQ4_NL: return 18; /* f16 d, 16 bytes of indices into kvalues_iq4nl ggml-common.h:405-410 */
        case QMO_IQ4_XS: return 136; /* f16 d, u16 scales_h, 4 scales_l bytes, 128 bytes of table indices  ggml-common.h:411-418 */
        case QMO_Q2_K: return 84;   /* 16 scale|min nibbles, 64 2-bit bytes, f16 d, f16 dmin  ggml-common.h:253-265 */
        case QMO_Q3_K: return 110;  /* 32 hmask, 64 2-bit bytes, 12 scale bytes, f16 d        ggml-common.h:271-277 */
        case QMO_Q4_K: return 144;  /* f16 d, f16 dmin, 12 scale bytes, 128 nibbles ggml-common.h:285-296 */
        case QMO_Q5_K: return 176;  /* + 32 high-bit bytes before the nibbles       ggml-common.h:298-314 */
        case QMO_Q6_K: return 210;  /* 128 ql, 64 qh, 16 int8 scales, f16 d         ggml-common.h:316-326 */
        case QMO_Q8_K: return 292;  /* f32 d, 256 int8, 16 int16 bsums              ggml-common.h:328-334 */
        default: return 0;
    }
}

size_t qmo_row_size(int type, int64_t k) {
    const int b = qmo_blck_size(type);
    return b ? (size_t)(k / b) * qmo_type_size(type) : 0;
}

int qmo_vec_dot_type(int type) {   /* type_traits_cpu[].vec_dot_type, ggml-cpu.c:256-… */
    switch (type) {
        case QMO_Q4_0: case QMO_Q8_0: case QMO_Q5_0: case QMO_IQ4_NL: return QMO_Q8_0;
        case QMO_Q4_1: case QMO_Q5_1: return QMO_Q8_1;
        case QMO_Q2_K: case QMO_Q3_K: case QMO_Q4_K: case QMO_Q5_K: case QMO_Q6_K: case QMO_IQ4_XS: return QMO_Q8_K;
        default: return -1;
    }
}

/* ---------------------------------------------------------------- fp16 */

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float    u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

float qmo_fp16_to_fp32(uint16_t h) {
    /* IEEE binary16 -> binary32, exact (what GGML_FP16_TO_FP32 yields on every host, ggml-impl.h:323-400) */
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    const uint32_t exp  = (h >> 10) & 0x1fu;
    uint32_t man = h & 0x3ffu;
    if (exp == 0x1f) return u2f(sign | 0x7f800000u | (man << 13));
    if (exp != 0)    return u2f(sign | ((exp + 112u) << 23) | (man << 13));
    if (man == 0)    return u2f(sign);
    int e = -1;                      /* subnormal half: normalise */
    do { man <<= 1; ++e; } while (!(man & 0x400u));
    return u2f(sign | ((uint32_t)(112 - e) << 23) | ((man & 0x3ffu) << 13));
}

uint16_t qmo_fp32_to_fp16(float f) {
    /* binary32 -> binary16, round to nearest even (F16C / the reference's bit trick agree on this) */
    const uint32_t u = f2u(f);
    const uint16_t sign = (uint16_t)((u >> 16) & 0x8000u);
    const uint32_t a = u & 0x7fffffffu;
    if (a >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (a > 0x7f800000u ? 0x200u : 0));
    if (a >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);           /* rounds to >= 65520 -> inf */
    if (a < 0x33000001u)  return sign;                                   /* < 2^-25 (or == ) -> 0     */
    const int e = (int)(a >> 23) - 127;
    uint32_t man = (a & 0x7fffffu) | 0x800000u;
    int shift;
    uint32_t base;
    if (e < -14) { shift = 13 + (-14 - e); base = 0; }                   /* subnormal half */
    else         { shift = 13;             base = (uint32_t)(e + 15) << 10; man &= 0x7fffffu; }
    const uint32_t q = man >> shift;
    const uint32_t r = man & ((1u << shift) - 1u);
    const uint32_t half = 1u << (shift - 1);
    uint32_t out = base + q;
    if (r > half || (r == half && (q & 1u))) out += 1;                   /* carries roll into the exponent */
    return (uint16_t)(sign | out);
}

static inline float rd_f16(const uint8_t *p) { uint16_t h; memcpy(&h, p, 2); return qmo_fp16_to_fp32(h); }

/* ---------------------------------------------------------------- K-quant 6-bit scale/min unpack */

/* get_scale_min_k4, ggml-quants.c:631-638: entry j of the 8 (scale, min) pairs in 12 bytes */
static inline void k4_scale_min(int j, const uint8_t *s, int *sc, int *mn) {
    if (j < 4) {
        *sc = s[j] & 63;
        *mn = s[j + 4] & 63;
    } else {
        *sc = (s[j + 4] & 15) | ((s[j - 4] >> 6) << 4);
        *mn = (s[j + 4] >> 4) | ((s[j]     >> 6) << 4);
    }
}

/* Q3_K: sixteen 6-bit scales out of 12 bytes, as the reference's aux[] shuffle leaves them (ggml-quants.c:1074-1079):
 * scale j = low nibble (j < 8: of byte j, else high nibble of byte j - 8) | two bits of byte 8 + j % 4 (pair j / 4) << 4 */
static inline int q3k_scale(const uint8_t *s, int j) {
    const int lo = j < 8 ? (s[j] & 15) : (s[j - 8] >> 4);
    const int hi = (s[8 + (j & 3)] >> (2 * (j >> 2))) & 3;
    return (lo | (hi << 4)) - 32;
}

static const int8_t iq4nl_values[16] = { -127, -104, -83, -65, -49, -35, -22, -10, 1, 13, 25, 38, 53, 69, 89, 113 };   /* ggml-common.h kvalues_iq4nl */

/* ---------------------------------------------------------------- block unpack */

int qmo_dequantize_row(int type, const void *src, float *dst, int64_t k) {
    const int bs = qmo_blck_size(type);
    if (!bs || type == QMO_Q8_K || type == QMO_Q8_1 || k % bs) return -1;
    const size_t ts = qmo_type_size(type);
    const int64_t nb = k / bs;
    const uint8_t *blk = (const uint8_t *)src;

    for (int64_t i = 0; i < nb; ++i, blk += ts, dst += bs) {
        switch (type) {
        case QMO_Q4_0: {                                  /* ggml-quants.c:255-273 */
            const float d = rd_f16(blk);
            const uint8_t *qs = blk + 2;
            for (int j = 0; j < 16; ++j) {
                dst[j]      = (float)((qs[j] & 15) - 8) * d;
                dst[j + 16] = (float)((qs[j] >> 4) - 8) * d;
            }
        } break;
        case QMO_Q8_0: {                                  /* ggml-quants.c:349-363 */
            const float d = rd_f16(blk);
            const int8_t *qs = (const int8_t *)(blk + 2);
            for (int j = 0; j < 32; ++j) dst[j] = (float)qs[j] * d;
        } break;
        case QMO_Q4_1: {                                  /* ggml-quants.c:275-294: x*d + m */
            const float d = rd_f16(blk), m = rd_f16(blk + 2);
            const uint8_t *qs = blk + 4;
            for (int j = 0; j < 16; ++j) {
                dst[j]      = (float)(qs[j] & 15) * d + m;
                dst[j + 16] = (float)(qs[j] >> 4) * d + m;
            }
        } break;
        case QMO_Q5_0:                                    /* ggml-quants.c:296-320 */
        case QMO_Q5_1: {                                  /* ggml-quants.c:322-347 */
            const int one = type == QMO_Q5_1;
            const float d = rd_f16(blk), m = one ? rd_f16(blk + 2) : 0.0f;
            uint32_t qh; memcpy(&qh, blk + (one ? 4 : 2), 4);
            const uint8_t *qs = blk + (one ? 8 : 6);
            for (int j = 0; j < 16; ++j) {
                const int x0 = (qs[j] & 15) | (int)(((qh >> j) & 1u) << 4);
                const int x1 = (qs[j] >> 4) | (int)(((qh >> (j + 16)) & 1u) << 4);
                if (one) { dst[j] = (float)x0 * d + m;        dst[j + 16] = (float)x1 * d + m; }
                else     { dst[j] = (float)(x0 - 16) * d;     dst[j + 16] = (float)(x1 - 16) * d; }
            }
        } break;
        case QMO_IQ4_NL: {                                /* ggml-quants.c:2436-2453 */
            const float d = rd_f16(blk);
            const uint8_t *qs = blk + 2;
            for (int j = 0; j < 16; ++j) {
                dst[j]      = d * (float)iq4nl_values[qs[j] & 15];
                dst[j + 16] = d * (float)iq4nl_values[qs[j] >> 4];
            }
        } break;
        case QMO_IQ4_XS: {                                /* ggml-quants.c:2454-2475: eight sub-blocks of 32, y = d*(ls-32)*kvalues_iq4nl[q] */
            const float d = rd_f16(blk);
            const unsigned sh = (unsigned)blk[2] | ((unsigned)blk[3] << 8);
            const uint8_t *sl = blk + 4, *qs = blk + 8;
            for (int ib = 0; ib < 8; ++ib, qs += 16) {
                const int ls = (int)((sl[ib / 2] >> (4 * (ib % 2))) & 0xf) | (int)(((sh >> (2 * ib)) & 3) << 4);
                const float dl = d * (float)(ls - 32);
                for (int j = 0; j < 16; ++j) {
                    dst[32 * ib + j]      = dl * (float)iq4nl_values[qs[j] & 15];
                    dst[32 * ib + j + 16] = dl * (float)iq4nl_values[qs[j] >> 4];
                }
            }
        } break;
        case QMO_Q2_K: {                                  /* ggml-quants.c:712-745: 16 sub-blocks of 16, y = d*(sc&15)*q - dmin*(sc>>4) */
            const uint8_t *sc = blk, *qs = blk + 16;
            const float d = rd_f16(blk + 80), dmin = rd_f16(blk + 82);
            for (int g = 0; g < 16; ++g) {                /* sub-block g: half n = g/8, shift 2*((g%8)/2), bytes 32n + 16*(g%2) .. +16 */
                const float dl = d * (float)(sc[g] & 15), ml = dmin * (float)(sc[g] >> 4);
                const uint8_t *q = qs + 32 * (g >> 3) + 16 * (g & 1);
                const int shift = 2 * ((g & 7) >> 1);
                for (int l = 0; l < 16; ++l) dst[16 * g + l] = dl * (float)((q[l] >> shift) & 3) - ml;
            }
        } break;
        case QMO_Q3_K: {                                  /* ggml-quants.c:1056-1106: y = d*(sc-32)*(q2 - (hbit ? 0 : 4)) */
            const uint8_t *hm = blk, *qs = blk + 32, *s12 = blk + 96;
            const float d = rd_f16(blk + 108);
            for (int g = 0; g < 16; ++g) {                /* same walk as Q2_K; the high bit of sub-block g is bit g/2 of hmask[16*(g%2) + l] */
                const float dl = d * (float)q3k_scale(s12, g);
                const uint8_t *q = qs + 32 * (g >> 3) + 16 * (g & 1), *h = hm + 16 * (g & 1);
                const int shift = 2 * ((g & 7) >> 1), bit = g >> 1;
                for (int l = 0; l < 16; ++l)
                    dst[16 * g + l] = dl * (float)((int)((q[l] >> shift) & 3) - (((h[l] >> bit) & 1) ? 0 : 4));
            }
        } break;
        case QMO_Q4_K:                                    /* ggml-quants.c:1280-1302 */
        case QMO_Q5_K: {                                  /* ggml-quants.c:1482-1508 */
            const float d = rd_f16(blk), dmin = rd_f16(blk + 2);
            const uint8_t *sc12 = blk + 4;
            const uint8_t *qh = (type == QMO_Q5_K) ? blk + 16 : NULL;
            const uint8_t *qs = blk + (type == QMO_Q5_K ? 48 : 16);
            for (int pair = 0; pair < 4; ++pair) {        /* 64 weights per pair of sub-blocks */
                int s0, m0, s1, m1;
                k4_scale_min(2 * pair,     sc12, &s0, &m0);
                k4_scale_min(2 * pair + 1, sc12, &s1, &m1);
                const float d0 = d * (float)s0, o0 = dmin * (float)m0;
                const float d1 = d * (float)s1, o1 = dmin * (float)m1;
                const uint8_t *q = qs + 32 * pair;
                for (int l = 0; l < 32; ++l) {
                    int lo = q[l] & 15, hi = q[l] >> 4;
                    if (qh) {
                        lo += ((qh[l] >> (2 * pair))     & 1) << 4;
                        hi += ((qh[l] >> (2 * pair + 1)) & 1) << 4;
                    }
                    dst[64 * pair + l]      = d0 * (float)lo - o0;
                    dst[64 * pair + 32 + l] = d1 * (float)hi - o1;
                }
            }
        } break;
        case QMO_Q6_K: {                                  /* ggml-quants.c:1690-1722 */
            const uint8_t *ql = blk, *qh = blk + 128;
            const int8_t *sc = (const int8_t *)(blk + 192);
            const float d = rd_f16(blk + 208);
            for (int half = 0; half < 2; ++half) {
                float *y = dst + 128 * half;
                const uint8_t *l4 = ql + 64 * half, *h2 = qh + 32 * half;
                const int8_t *s = sc + 8 * half;
                for (int l = 0; l < 32; ++l) {
                    const int is = l >> 4;
                    const int q1 = (int8_t)((l4[l]      & 15) | (((h2[l] >> 0) & 3) << 4)) - 32;
                    const int q2 = (int8_t)((l4[l + 32] & 15) | (((h2[l] >> 2) & 3) << 4)) - 32;
                    const int q3 = (int8_t)((l4[l]      >> 4) | (((h2[l] >> 4) & 3) << 4)) - 32;
                    const int q4 = (int8_t)((l4[l + 32] >> 4) | (((h2[l] >> 6) & 3) << 4)) - 32;
                    /* reference evaluates d * sc * q left to right */
                    y[l]      = d * (float)s[is]     * (float)q1;
                    y[l + 32] = d * (float)s[is + 2] * (float)q2;
                    y[l + 64] = d * (float)s[is + 4] * (float)q3;
                    y[l + 96] = d * (float)s[is + 6] * (float)q4;
                }
            }
        } break;
        default: return -1;
        }
    }
    return 0;
}

/* ---------------------------------------------------------------- activation quantizers */

void qmo_quantize_row_q8_0(const float *x, void *y, int64_t k, int act_mode) {
    uint8_t *out = (uint8_t *)y;
    for (int64_t i = 0; i < k / 32; ++i, x += 32, out += 34) {
        float amax = 0.0f;
        for (int j = 0; j < 32; ++j) { const float a = fabsf(x[j]); if (a > amax) amax = a; }
        const float d = amax / 127.0f;
        const uint16_t dh = qmo_fp32_to_fp16(d);
        memcpy(out, &dh, 2);
        int8_t *qs = (int8_t *)(out + 2);
        if (act_mode == QMO_ACT_X86) {                    /* ggml-cpu-quants.c:806-870 (AVX2 branch) */
            const float id = amax != 0.0f ? 127.0f / amax : 0.0f;
            for (int j = 0; j < 32; ++j) qs[j] = (int8_t)(int)nearbyintf(x[j] * id);  /* ties to even */
        } else {                                          /* ggml-quants.c:194-217 */
            const float id = d != 0.0f ? 1.0f / d : 0.0f;
            for (int j = 0; j < 32; ++j) qs[j] = (int8_t)roundf(x[j] * id);          /* ties away    */
        }
    }
}

/* Q8_1: the bytes of Q8_0 plus s = f16(d * sum of the int8s), with the f32 d (ggml-quants.c:220-252; the AVX2 build multiplies the
 * same f32 d by the integer sum, ggml-cpu-quants.c:1053-…: same value) */
void qmo_quantize_row_q8_1(const float *x, void *y, int64_t k, int act_mode) {
    uint8_t *out = (uint8_t *)y;
    for (int64_t i = 0; i < k / 32; ++i, x += 32, out += 36) {
        uint8_t tmp[34];
        qmo_quantize_row_q8_0(x, tmp, 32, act_mode);
        float amax = 0.0f;
        for (int j = 0; j < 32; ++j) { const float a = fabsf(x[j]); if (a > amax) amax = a; }
        const float d = amax / 127.0f;
        int sum = 0;
        for (int j = 0; j < 32; ++j) sum += (int8_t)tmp[2 + j];
        const uint16_t sh = qmo_fp32_to_fp16((float)sum * d);
        memcpy(out, tmp, 2);
        memcpy(out + 2, &sh, 2);
        memcpy(out + 4, tmp + 2, 32);
    }
}

/* nearest_int, ggml-quants.c:372-377: round-to-nearest-even through the 1.5*2^23 magic constant */
static inline int magic_round(float v) {
    const float t = v + 12582912.0f;
    return (int)(f2u(t) & 0x007fffffu) - 0x00400000;
}

void qmo_quantize_row_q8_K(const float *x, void *y, int64_t k) {   /* ggml-quants.c:2479-2516 */
    uint8_t *out = (uint8_t *)y;
    for (int64_t i = 0; i < k / 256; ++i, x += 256, out += 292) {
        float peak = 0.0f, amax = 0.0f;                   /* signed value at the first max |x| */
        for (int j = 0; j < 256; ++j) {
            const float a = fabsf(x[j]);
            if (a > amax) { amax = a; peak = x[j]; }
        }
        int8_t *qs = (int8_t *)(out + 4);
        int16_t bs[16];
        if (amax == 0.0f) {
            /* reference leaves bsums untouched here (whatever wdata held); we define them as 0.
             * They are only ever multiplied by d == 0. */
            memset(out, 0, 292);
            continue;
        }
        const float iscale = -127.0f / peak;
        for (int j = 0; j < 256; ++j) {
            const int v = magic_round(iscale * x[j]);
            qs[j] = (int8_t)(v < 127 ? v : 127);
        }
        for (int g = 0; g < 16; ++g) {
            int s = 0;
            for (int j = 0; j < 16; ++j) s += qs[16 * g + j];
            bs[g] = (int16_t)s;
        }
        const float d = 1.0f / iscale;
        memcpy(out, &d, 4);
        memcpy(out + 260, bs, 32);
    }
}

/* ---------------------------------------------------------------- row dots (scalar order) */

static float dot_q4_0(int64_t k, const uint8_t *w, const uint8_t *a) {   /* ggml-cpu-quants.c:2591-2607 */
    float acc = 0.0f;
    for (int64_t b = 0; b < k / 32; ++b, w += 18, a += 34) {
        const uint8_t *qs = w + 2;
        const int8_t *y = (const int8_t *)(a + 2);
        int s0 = 0, s1 = 0;
        for (int j = 0; j < 16; ++j) {
            s0 += ((qs[j] & 15) - 8) * y[j];
            s1 += ((qs[j] >> 4) - 8) * y[j + 16];
        }
        acc += (float)(s0 + s1) * rd_f16(w) * rd_f16(a);   /* (sumi*dx)*dy */
    }
    return acc;
}

static float dot_q8_0(int64_t k, const uint8_t *w, const uint8_t *a) {   /* ggml-cpu-quants.c:4004-4015 */
    float acc = 0.0f;
    for (int64_t b = 0; b < k / 32; ++b, w += 34, a += 34) {
        const int8_t *x = (const int8_t *)(w + 2), *y = (const int8_t *)(a + 2);
        int s = 0;
        for (int j = 0; j < 32; ++j) s += x[j] * y[j];
        acc += (float)s * (rd_f16(w) * rd_f16(a));         /* sumi*(dx*dy) */
    }
    return acc;
}

/* Q4_K / Q5_K: ggml-cpu-quants.c:7535-7591 / 8351-8412.  Eight f32 partial sums, lane l collects
 * elements with (index mod 8) == l; the mins term goes to a separate running sum. */
static float dot_q45_K(int type, int64_t k, const uint8_t *w, const uint8_t *a) {
    const size_t ts = qmo_type_size(type);
    float lanes[8] = {0};
    float acc = 0.0f;
    for (int64_t b = 0; b < k / 256; ++b, w += ts, a += 292) {
        const uint8_t *sc12 = w + 4;
        const uint8_t *qh = (type == QMO_Q5_K) ? w + 16 : NULL;
        const uint8_t *qs = w + (type == QMO_Q5_K ? 48 : 16);
        float yd; memcpy(&yd, a, 4);
        const int8_t *q8 = (const int8_t *)(a + 4);
        int16_t bsums[16]; memcpy(bsums, a + 260, 32);

        int8_t wq[256];
        for (int pair = 0; pair < 4; ++pair)
            for (int l = 0; l < 32; ++l) {
                int lo = qs[32 * pair + l] & 15, hi = qs[32 * pair + l] >> 4;
                if (qh) {
                    lo += ((qh[l] >> (2 * pair))     & 1) << 4;
                    hi += ((qh[l] >> (2 * pair + 1)) & 1) << 4;
                }
                wq[64 * pair + l] = (int8_t)lo;
                wq[64 * pair + 32 + l] = (int8_t)hi;
            }
        int sc[8], mn[8];
        for (int j = 0; j < 8; ++j) k4_scale_min(j, sc12, &sc[j], &mn[j]);

        int sum_mins = 0;
        for (int g = 0; g < 16; ++g) sum_mins += bsums[g] * mn[g / 2];

        int32_t part[8] = {0};
        for (int j = 0; j < 8; ++j)
            for (int e = 0; e < 32; ++e)
                part[e & 7] += sc[j] * (int16_t)(q8[32 * j + e] * wq[32 * j + e]);

        const float d = rd_f16(w) * yd;
        for (int l = 0; l < 8; ++l) lanes[l] += d * (float)part[l];
        const float dm = rd_f16(w + 2) * yd;
        acc -= dm * (float)sum_mins;
    }
    for (int l = 0; l < 8; ++l) acc += lanes[l];
    return acc;
}

static float dot_q6_K(int64_t k, const uint8_t *w, const uint8_t *a) {   /* ggml-cpu-quants.c:9423-9465 */
    float lanes[8] = {0};
    for (int64_t b = 0; b < k / 256; ++b, w += 210, a += 292) {
        const uint8_t *ql = w, *qh = w + 128;
        const int8_t *sc = (const int8_t *)(w + 192);
        float yd; memcpy(&yd, a, 4);
        const int8_t *q8 = (const int8_t *)(a + 4);
        int8_t wq[256];
        for (int half = 0; half < 2; ++half) {
            const uint8_t *l4 = ql + 64 * half, *h2 = qh + 32 * half;
            int8_t *o = wq + 128 * half;
            for (int l = 0; l < 32; ++l) {
                o[l]      = (int8_t)(((l4[l]      & 15) | (((h2[l] >> 0) & 3) << 4)) - 32);
                o[l + 32] = (int8_t)(((l4[l + 32] & 15) | (((h2[l] >> 2) & 3) << 4)) - 32);
                o[l + 64] = (int8_t)(((l4[l]      >> 4) | (((h2[l] >> 4) & 3) << 4)) - 32);
                o[l + 96] = (int8_t)(((l4[l + 32] >> 4) | (((h2[l] >> 6) & 3) << 4)) - 32);
            }
        }
        int32_t part[8] = {0};
        for (int g = 0; g < 16; ++g)
            for (int e = 0; e < 16; ++e)
                part[e & 7] += (int)sc[g] * (int16_t)(q8[16 * g + e] * wq[16 * g + e]);
        const float d = rd_f16(w + 208) * yd;
        for (int l = 0; l < 8; ++l) lanes[l] += d * (float)part[l];
    }
    float acc = 0.0f;
    for (int l = 0; l < 8; ++l) acc += lanes[l];
    return acc;
}

/* Q4_1 / Q5_1 against Q8_1 (ggml-cpu-quants.c:2910-2925, 3572-3592): (dx*dy)*sumi + mx*sy per block;
 * Q5_0 / IQ4_NL against Q8_0 (:3227-3248, 12652-12660) */
static float dot_legacy(int type, int64_t k, const uint8_t *w, const uint8_t *a) {
    const size_t ts = qmo_type_size(type);
    const int q81 = type == QMO_Q4_1 || type == QMO_Q5_1;
    float acc = 0.0f;
    for (int64_t b = 0; b < k / 32; ++b, w += ts, a += q81 ? 36 : 34) {
        const int8_t *y = (const int8_t *)(a + (q81 ? 4 : 2));
        int s0 = 0, s1 = 0;
        if (type == QMO_Q4_1) {
            const uint8_t *qs = w + 4;
            for (int j = 0; j < 16; ++j) { s0 += (qs[j] & 15) * y[j]; s1 += (qs[j] >> 4) * y[j + 16]; }
        } else if (type == QMO_IQ4_NL) {
            const uint8_t *qs = w + 2;
            for (int j = 0; j < 16; ++j) { s0 += y[j] * iq4nl_values[qs[j] & 15]; s1 += y[j + 16] * iq4nl_values[qs[j] >> 4]; }
        } else {
            const int one = type == QMO_Q5_1;
            uint32_t qh; memcpy(&qh, w + (one ? 4 : 2), 4);
            const uint8_t *qs = w + (one ? 8 : 6);
            for (int j = 0; j < 16; ++j) {
                const int x0 = ((qs[j] & 15) | (int)(((qh >> j) & 1u) << 4)) - (one ? 0 : 16);
                const int x1 = ((qs[j] >> 4) | (int)(((qh >> (j + 16)) & 1u) << 4)) - (one ? 0 : 16);
                s0 += x0 * y[j];
                s1 += x1 * y[j + 16];
            }
        }
        const int sumi = s0 + s1;
        if (q81)                     acc += (rd_f16(w) * rd_f16(a)) * (float)sumi + rd_f16(w + 2) * rd_f16(a + 2);
        else if (type == QMO_IQ4_NL) acc += (rd_f16(a) * rd_f16(w)) * (float)sumi;            /* d = dy*dx; d * (sumi1 + sumi2) */
        else                         acc += (rd_f16(w) * rd_f16(a)) * (float)sumi;
    }
    return acc;
}

static float dot_q2_K(int64_t k, const uint8_t *w, const uint8_t *a) {   /* ggml-cpu-quants.c:5484-5523 */
    float acc = 0.0f;
    for (int64_t b = 0; b < k / 256; ++b, w += 84, a += 292) {
        const uint8_t *sc = w, *qs = w + 16;
        float yd; memcpy(&yd, a, 4);
        const int8_t *q8 = (const int8_t *)(a + 4);
        int16_t bsums[16]; memcpy(bsums, a + 260, 32);
        int summs = 0;
        for (int j = 0; j < 16; ++j) summs += bsums[j] * (sc[j] >> 4);
        const float dall = yd * rd_f16(w + 80), dmin = yd * rd_f16(w + 82);
        int isum = 0;
        for (int g = 0; g < 16; ++g) {
            const uint8_t *q = qs + 32 * (g >> 3) + 16 * (g & 1);
            const int shift = 2 * ((g & 7) >> 1);
            int l16 = 0;
            for (int l = 0; l < 16; ++l) l16 += q8[16 * g + l] * ((q[l] >> shift) & 3);
            isum += (sc[g] & 15) * l16;
        }
        acc += dall * (float)isum - dmin * (float)summs;
    }
    return acc;
}

static float dot_q3_K(int64_t k, const uint8_t *w, const uint8_t *a) {   /* ggml-cpu-quants.c:6600-6661: eight f32 lanes */
    float lanes[8] = {0};
    for (int64_t b = 0; b < k / 256; ++b, w += 110, a += 292) {
        const uint8_t *hm = w, *qs = w + 32, *s12 = w + 96;
        float yd; memcpy(&yd, a, 4);
        const int8_t *q8 = (const int8_t *)(a + 4);
        int32_t part[8] = {0};
        for (int g = 0; g < 16; ++g) {
            const uint8_t *q = qs + 32 * (g >> 3) + 16 * (g & 1), *h = hm + 16 * (g & 1);
            const int shift = 2 * ((g & 7) >> 1), bit = g >> 1, scale = q3k_scale(s12, g);
            for (int l = 0; l < 16; ++l) {
                const int v = (int)((q[l] >> shift) & 3) - (((h[l] >> bit) & 1) ? 0 : 4);
                part[l & 7] += scale * (int16_t)(q8[16 * g + l] * v);
            }
        }
        const float d = rd_f16(w + 108) * yd;
        for (int l = 0; l < 8; ++l) lanes[l] += d * (float)part[l];
    }
    float acc = 0.0f;
    for (int l = 0; l < 8; ++l) acc += lanes[l];
    return acc;
}

static float dot_iq4_xs(int64_t k, const uint8_t *w, const uint8_t *a) {   /* ggml-cpu-quants.c:12665-13065, the scalar branch (:13037-13063) */
    float sumf = 0.0f;
    for (int64_t b = 0; b < k / 256; ++b, w += 136, a += 292) {
        float yd; memcpy(&yd, a, 4);
        const float d4d8 = rd_f16(w) * yd;
        unsigned h = (unsigned)w[2] | ((unsigned)w[3] << 8);
        const uint8_t *sl = w + 4, *qs = w + 8;
        const int8_t *q8 = (const int8_t *)(a + 4);
        for (int ib = 0; ib < 8; ib += 2) {
            const int ls1 = (int)(sl[ib / 2] & 0xf) | (int)((h << 4) & 0x30);
            const int ls2 = (int)(sl[ib / 2] >> 4) | (int)((h << 2) & 0x30);
            h >>= 4;
            const float d1 = d4d8 * (float)(ls1 - 32), d2 = d4d8 * (float)(ls2 - 32);
            int sumi1 = 0, sumi2 = 0;
            for (int j = 0; j < 16; ++j) { sumi1 += q8[j] * iq4nl_values[qs[j] & 0xf]; sumi2 += q8[j + 16] * iq4nl_values[qs[j] >> 4]; }
            sumf += d1 * (float)(sumi1 + sumi2);
            qs += 16; q8 += 32;
            sumi1 = sumi2 = 0;
            for (int j = 0; j < 16; ++j) { sumi1 += q8[j] * iq4nl_values[qs[j] & 0xf]; sumi2 += q8[j + 16] * iq4nl_values[qs[j] >> 4]; }
            sumf += d2 * (float)(sumi1 + sumi2);
            qs += 16; q8 += 32;
        }
    }
    return sumf;
}

float qmo_vec_dot(int type, int64_t k, const void *w_row, const void *act_row) {
    const uint8_t *w = (const uint8_t *)w_row, *a = (const uint8_t *)act_row;
    switch (type) {
        case QMO_Q4_0: return dot_q4_0(k, w, a);
        case QMO_Q8_0: return dot_q8_0(k, w, a);
        case QMO_Q4_K: case QMO_Q5_K: return dot_q45_K(type, k, w, a);
        case QMO_Q6_K: return dot_q6_K(k, w, a);
        case QMO_Q4_1: case QMO_Q5_0: case QMO_Q5_1: case QMO_IQ4_NL: return dot_legacy(type, k, w, a);
        case QMO_Q2_K: return dot_q2_K(k, w, a);
        case QMO_Q3_K: return dot_q3_K(k, w, a);
        case QMO_IQ4_XS: return dot_iq4_xs(k, w, a);
        default: return NAN;
    }
}

/* ---------------------------------------------------------------- MUL_MAT / MUL_MAT_ID */

static void *quantize_acts(int type, const float *x, int64_t K, int64_t rows, int64_t ldx, int act_mode,
                           size_t *row_bytes) {
    const int vt = qmo_vec_dot_type(type);
    *row_bytes = qmo_row_size(vt, K);
    uint8_t *buf = (uint8_t *)malloc(*row_bytes * (size_t)(rows > 0 ? rows : 1));
    if (!buf) return NULL;
    #pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < rows; ++n) {              /* phase 1, ggml-cpu.c:6807-6842 */
        if (vt == QMO_Q8_0)      qmo_quantize_row_q8_0(x + n * ldx, buf + n * *row_bytes, K, act_mode);
        else if (vt == QMO_Q8_1) qmo_quantize_row_q8_1(x + n * ldx, buf + n * *row_bytes, K, act_mode);
        else                     qmo_quantize_row_q8_K(x + n * ldx, buf + n * *row_bytes, K);
    }
    return buf;
}

int qmo_mul_mat(int type, const void *W, int64_t K, int64_t M,
                const float *x, int64_t N, int64_t ldx, float *dst, int64_t ldd, int act_mode) {
    const int bs = qmo_blck_size(type);
    if (qmo_vec_dot_type(type) < 0 || K % bs) return -1;
    size_t arow;
    uint8_t *acts = (uint8_t *)quantize_acts(type, x, K, N, ldx, act_mode, &arow);
    if (!acts) return -2;
    const size_t wrow = qmo_row_size(type, K);
    #pragma omp parallel for schedule(static)
    for (int64_t m = 0; m < M; ++m)                   /* phase 2, ggml-cpu.c:6655-6743 */
        for (int64_t n = 0; n < N; ++n)
            dst[n * ldd + m] = qmo_vec_dot(type, K, (const uint8_t *)W + m * wrow, acts + n * arow);
    free(acts);
    return 0;
}

int qmo_mul_mat_id(int type, const void *as, int64_t K, int64_t M, int64_t n_expert,
                   const float *b, int64_t ne11, int64_t n_tokens,
                   const int32_t *ids, int64_t n_used, int64_t ids_stride,
                   float *dst, int act_mode) {
    const int bs = qmo_blck_size(type);
    if (qmo_vec_dot_type(type) < 0 || K % bs) return -1;
    size_t arow;
    /* every src1 row is quantized once: rows are (i11 + i12*ne11), ggml-cpu.c:7070-7105 */
    uint8_t *acts = (uint8_t *)quantize_acts(type, b, K, ne11 * n_tokens, K, act_mode, &arow);
    if (!acts) return -2;
    const size_t wrow = qmo_row_size(type, K);
    const size_t wmat = wrow * (size_t)M;
    int rc = 0;
    for (int64_t t = 0; t < n_tokens; ++t)            /* row grouping collapses to this double loop */
        for (int64_t s = 0; s < n_used; ++s) {        /* ggml-cpu.c:7107-7122, 6981-7005 */
            const int32_t e = ids[t * ids_stride + s];
            if (e < 0 || e >= n_expert) { rc = -3; continue; }
            const uint8_t *We = (const uint8_t *)as + (size_t)e * wmat;
            const uint8_t *a = acts + (size_t)((s % ne11) + t * ne11) * arow;
            float *out = dst + (size_t)(s + t * n_used) * (size_t)M;
            #pragma omp parallel for schedule(static)
            for (int64_t m = 0; m < M; ++m) out[m] = qmo_vec_dot(type, K, We + m * wrow, a);
        }
    free(acts);
    return rc;
}
