/*
 * ggml_mi355x_ops.h — C-ABI of the per-layer glue ops that sit either side of the quantized MUL_MAT path
 * (RMS_NORM, ADD/MUL/SUB/DIV, SCALE, UNARY, ROPE, SOFT_MAX, CPY/CONT/DUP, GET_ROWS and the
 * F16/F32 x F32 MUL_MATs of attention), so that a whole transformer layer is one scheduler split and the KV cache
 * lives in HBM.  Same library (libggml_mi355x_qmm.so), same conventions as ggml_mi355x_qmm.h.
 *
 * Reference interface this mirrors: the cDSP entry points take three tensor descriptors,
 *     int ggmlop_dsp_add    (remote_handle64, const dsptensor * src0, const dsptensor * src1, dsptensor * dst)
 *     int ggmlop_dsp_softmax(...), ggmlop_dsp_rmsnorm(...)        (kernels/ggmlop_ap_skel.h:267-273; bodies
 *     kernels/ggml-dsp.c:991-1089 (add), :1353-1365 (softmax / rmsnorm are empty stubs there))
 * with `dsptensor` = {type, ne[4], nb[4], op, op_params[16], flags, data, data_len} (ggmlop_ap_skel.h:234-244).
 * `qmm_tensor` is that descriptor for this boundary (64-bit extents, device pointer).  Semantics of every op are those
 * of the ggml CPU backend (file:line at each kernel in csrc/qmm_ops.hip); parity is checked by the reference's own
 * tests/test-backend-ops.cpp against the CPU backend.
 */
#ifndef GGML_MI355X_OPS_H
#define GGML_MI355X_OPS_H

#include "ggml_mi355x_qmm.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct qmm_tensor {
    void *  data;             /* DEVICE pointer */
    int32_t type;             /* ggml_type: F32 = 0, F16 = 1, I32 = 26, quantized types as in ggml_mi355x_qmm.h */
    int32_t flags;            /* reserved, 0 */
    int64_t ne[4];            /* elements */
    int64_t nb[4];            /* bytes */
    int32_t op_params[16];    /* ggml_tensor::op_params of the node (dst only) */
} qmm_tensor;

enum qmm_op {
    QMM_OP_ADD = 1, QMM_OP_SUB, QMM_OP_MUL, QMM_OP_DIV,
    QMM_OP_SCALE,
    QMM_OP_SILU, QMM_OP_GELU, QMM_OP_GELU_QUICK, QMM_OP_RELU, QMM_OP_TANH, QMM_OP_SIGMOID, QMM_OP_NEG, QMM_OP_EXP,
    QMM_OP_RMS_NORM,          /* op_params[0] = eps (f32) */
    QMM_OP_ROPE,              /* src1 = positions (i32), src2 = frequency factors (f32) or NULL; op_params as ggml_rope_ext */
    QMM_OP_SOFT_MAX,          /* src1 = mask (f32 / f16) or NULL; op_params = {scale, max_bias} */
    QMM_OP_CPY,               /* also CONT and DUP: src0 -> dst, any strides, F32 / F16 either side */
    QMM_OP_GET_ROWS,          /* src0 rows (F32 / F16 / the five quantized types) picked by src1 (i32) -> f32 */
    QMM_OP_MUL_MAT_F,         /* src0 F16 or F32 (any row strides, broadcast over dims 2/3), src1 F32 -> f32 */
    /* fused pairs the plugin forms from consecutive nodes */
    QMM_OP_RMS_NORM_MUL,      /* dst = rms_norm(src0) * src1, src1 one f32 row broadcast over all rows */
    QMM_OP_SILU_MUL,          /* dst = silu(src0) * src1, same shapes, contiguous (SwiGLU of build_ffn) */
    QMM_OP_COUNT
};

/* 1 when the combination of op, types, shapes and strides is implemented, 0 otherwise.  The plugin's supports_op asks this,
 * so the library is the single place that knows the surface.  src1 / src2 may be NULL where the op has none. */
QMM_API int qmm_op_supported(int op, const qmm_tensor * src0, const qmm_tensor * src1, const qmm_tensor * src2,
                             const qmm_tensor * dst);

/* Runs one op on `stream`.  QMM_EUNSUPPORTED for anything qmm_op_supported() rejects; nothing falls back to the CPU. */
QMM_API int qmm_op_compute(qmm_ctx * ctx, int op, const qmm_tensor * src0, const qmm_tensor * src1, const qmm_tensor * src2,
                           const qmm_tensor * dst, void * stream);

#ifdef __cplusplus
}
#endif
#endif
