/*
 * qmm_oracle.h — CPU oracle for the quantized MUL_MAT / MUL_MAT_ID path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product path (ggml-hexagon_amd/csrc) never does and fails loudly without HIP.
 *
 * This is a plain-C restatement (written from the algorithm, not copied) of what the reference's
 * oracle — the ggml CPU backend, which ggml-hexagon's DSP kernel strips line for line
 * (ggml/src/ggml-hexagon/kernels/ggml-dsp.c:1091-1351 == ggml/src/ggml-cpu/ggml-cpu.c:6655-6937) —
 * computes on this path, in its generic ("#else", scalar) summation order.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function here against golden
 * vectors produced by the real reference (oracle/_ref, built from /root/reference by oracle/Makefile;
 * generator tests/golden/make_golden.py) and, when oracle/_ref is present, live against it.
 *
 * Type ids are ggml's enum ggml_type values (ggml/include/ggml.h:351-…).
 */
#ifndef QMM_ORACLE_H
#define QMM_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { QMO_Q4_0 = 2, QMO_Q4_1 = 3, QMO_Q5_0 = 6, QMO_Q5_1 = 7, QMO_Q8_0 = 8, QMO_Q8_1 = 9, QMO_Q2_K = 10, QMO_Q3_K = 11,
       QMO_Q4_K = 12, QMO_Q5_K = 13, QMO_Q6_K = 14, QMO_Q8_K = 15, QMO_IQ4_NL = 20, QMO_IQ4_XS = 23 };

/* activation rounding variant (all three exist in the reference; bytes differ only on rare ties) */
enum { QMO_ACT_REF = 0,   /* quantize_row_q8_0_ref: id = 1/(amax/127), roundf          ggml-quants.c:194-217      */
       QMO_ACT_X86 = 1 }; /* AVX2 quantize_row_q8_0: id = 127/amax, round-to-nearest-even ggml-cpu-quants.c:806-870 */

int    qmo_blck_size(int type);                 /* 32 or 256; 0 if unsupported */
size_t qmo_type_size(int type);                 /* bytes per block            */
size_t qmo_row_size(int type, int64_t k);       /* k/blck*type_size           */
int    qmo_vec_dot_type(int type);              /* Q8_0 for Q4_0/Q5_0/Q8_0/IQ4_NL, Q8_1 for Q4_1/Q5_1, Q8_K for K-quants (ggml-cpu.c:256-…) */

float    qmo_fp16_to_fp32(uint16_t h);
uint16_t qmo_fp32_to_fp16(float f);

/* block unpack, bit-exact spec (ggml-quants.c:255-273, 275-347, 349-363, 712-745, 1056-1106, 1280-1302, 1482-1508, 1690-1722, 2436-2453) */
int qmo_dequantize_row(int type, const void *src, float *dst, int64_t k);

/* activation quantizers (ggml-quants.c:194-217, 2479-2516) */
void qmo_quantize_row_q8_0(const float *x, void *y, int64_t k, int act_mode);
void qmo_quantize_row_q8_1(const float *x, void *y, int64_t k, int act_mode);   /* ggml-quants.c:220-252; AVX2: ggml-cpu-quants.c:1053-… */
void qmo_quantize_row_q8_K(const float *x, void *y, int64_t k);

/* one row dot, scalar summation order (ggml-cpu-quants.c:2591-2607, 2910-2925, 3227-3248, 3572-3592, 4004-4015, 5484-5523,
 * 6600-6661, 7535-7591, 8351-8412, 9423-9465, 12652-12660) */
float qmo_vec_dot(int type, int64_t k, const void *w_row, const void *act_row);

/* dst[n*ldd + m] = W[m,:] . x[n,:]   (ggml-cpu.c:6745-6937 with nth=1, no llamafile path)
 * W: M rows of qmo_row_size(type,K) bytes; x: N rows, ldx floats apart; dst: N rows, ldd floats apart. */
int qmo_mul_mat(int type, const void *W, int64_t K, int64_t M,
                const float *x, int64_t N, int64_t ldx, float *dst, int64_t ldd, int act_mode);

/* MUL_MAT_ID (ggml-cpu.c:6941-7197): as [K,M,n_expert]; b [K, ne11, n_tokens] f32 (ne11 = n_used or 1);
 * ids [n_used, n_tokens] int32 with row stride ids_stride (elements); dst [M, n_used, n_tokens] f32. */
int qmo_mul_mat_id(int type, const void *as, int64_t K, int64_t M, int64_t n_expert,
                   const float *b, int64_t ne11, int64_t n_tokens,
                   const int32_t *ids, int64_t n_used, int64_t ids_stride,
                   float *dst, int act_mode);

#ifdef __cplusplus
}
#endif
#endif
