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
    /* the MoE router (build_moe_ffn) */
    QMM_OP_ARGSORT,           /* i32 indices that order each row; op_params[0] = 0 ascending, 1 descending (ggml_top_k = this + a view) */
    QMM_OP_SUM_ROWS,          /* dst [1, ne1, ne2, ne3] = row sums */
    QMM_OP_NORM,              /* LayerNorm without affine part: (x - mean) / sqrt(var + eps), op_params[0] = eps */
    QMM_OP_COUNT
};

/* 1 when the combination of op, types, shapes and strides is implemented, 0 otherwise.  The plugin's supports_op asks this,
 * so the library is the single place that knows the surface.  src1 / src2 may be NULL where the op has none. */
QMM_API int qmm_op_supported(int op, const qmm_tensor * src0, const qmm_tensor * src1, const qmm_tensor * src2,
                             const qmm_tensor * dst);

/* Runs one op on `stream`.  QMM_EUNSUPPORTED for anything qmm_op_supported() rejects; nothing falls back to the CPU. */
QMM_API int qmm_op_compute(qmm_ctx * ctx, int op, const qmm_tensor * src0, const qmm_tensor * src1, const qmm_tensor * src2,
                           const qmm_tensor * dst, void * stream);

/* The residual add that feeds a norm, in one pass: sum = a + b (same shapes), dst = rms_norm(sum) [* w] (w = one f32 row or NULL).
 * Two results, hence its own entry point.  In llama.cpp's layer: ffn_inp = cur + inpSA -> ffn_norm, and the layer's output
 * add -> the next layer's attn_norm (src/llama-model.cpp:4280-4300, 4340-4346). */
QMM_API int qmm_op_add_rms_norm_supported(const qmm_tensor * a, const qmm_tensor * b, const qmm_tensor * w, const qmm_tensor * sum,
                                          const qmm_tensor * dst);
QMM_API int qmm_op_add_rms_norm(qmm_ctx * ctx, const qmm_tensor * a, const qmm_tensor * b, const qmm_tensor * w, const qmm_tensor * sum,
                                const qmm_tensor * dst, float eps, void * stream);

/* Attention for a batch of up to 8 tokens in one launch: the chain build_attn_mha emits without flash attention
 * (src/llama-graph.cpp:1166-1203): dst = cont(permute(mul_mat(v, soft_max_ext(mul_mat(k, q), mask, scale)), 0, 2, 1, 3)).
 *   q    f32 [D, N, H]       (any row / head strides)        k  f16 [D, n_kv, H_kv]   rows dense, 16-byte aligned
 *   v    f16 [n_kv, Dv, H_kv] the transposed V cache          mask f32 [n_kv, >= N]
 *   dst  f32 [Dv * H, N]
 * D in {64, 128, 256}, n_kv a multiple of 8 (llama.cpp pads it to 32), H a multiple of H_kv (grouped-query attention). */
QMM_API int qmm_attn_decode_supported(const qmm_tensor * q, const qmm_tensor * k, const qmm_tensor * v, const qmm_tensor * mask,
                                      const qmm_tensor * dst);
QMM_API int qmm_attn_decode(qmm_ctx * ctx, const qmm_tensor * q, const qmm_tensor * k, const qmm_tensor * v, const qmm_tensor * mask,
                            const qmm_tensor * dst, float scale, void * stream);

/* The same chain for a prompt batch of any size whose scores fit LDS (n_kv <= 512, a multiple of 32; D = Dv in {64, 128}): one
 * workgroup per (64 tokens, head) keeps the 64 x n_kv score tile in LDS between the two MFMA products, so the f32 score tensor
 * (33 MB per layer at 512 x 512 x 32 heads) never exists.  Same operands and result layout as qmm_attn_decode. */
QMM_API int qmm_attn_prefill_supported(const qmm_tensor * q, const qmm_tensor * k, const qmm_tensor * v, const qmm_tensor * mask,
                                       const qmm_tensor * dst);
QMM_API int qmm_attn_prefill(qmm_ctx * ctx, const qmm_tensor * q, const qmm_tensor * k, const qmm_tensor * v, const qmm_tensor * mask,
                             const qmm_tensor * dst, float scale, void * stream);

/* What follows the q / k / v projections of a few-token batch in build_attn (src/llama-graph.cpp:1306-1365), in one launch:
 *   q_dst = rope(q) (f32; q_dst carries the ROPE node's op_params), k_dst = (f16) rope(k) straight into the K cache view,
 *   v_dst = (f16) v into the (transposed) V cache view, element i of v (in v's index order) to element i of v_dst, as CPY does.
 * k / k_dst and v / v_dst are optional pairs (NULL). */
QMM_API int qmm_rope_kv_store_supported(const qmm_tensor * q, const qmm_tensor * pos, const qmm_tensor * ff, const qmm_tensor * q_dst,
                                        const qmm_tensor * k, const qmm_tensor * k_dst, const qmm_tensor * v, const qmm_tensor * v_dst);
QMM_API int qmm_rope_kv_store(qmm_ctx * ctx, const qmm_tensor * q, const qmm_tensor * pos, const qmm_tensor * ff, const qmm_tensor * q_dst,
                              const qmm_tensor * k, const qmm_tensor * k_dst, const qmm_tensor * v, const qmm_tensor * v_dst, void * stream);

/* The router of a mixture-of-experts block behind its logits (build_moe_ffn, src/llama-graph.cpp:818-858) as one launch:
 *   ids     i32 [n_expert, n_tokens] = argsort(soft_max(logits), descending); its first n_used entries per row are ggml_top_k
 *   weights f32 [n_used * n_tokens] contiguous = the selected probabilities, divided by their sum when `normalise` (norm_w)
 * logits f32 [n_expert <= 64, n_tokens]. */
QMM_API int qmm_moe_router_supported(const qmm_tensor * logits, const qmm_tensor * ids, const qmm_tensor * weights, int64_t n_used);
QMM_API int qmm_moe_router(qmm_ctx * ctx, const qmm_tensor * logits, const qmm_tensor * ids, const qmm_tensor * weights, int64_t n_used,
                           int normalise, void * stream);

/* The same with the logits themselves, for up to 8 tokens (token generation): logits = gate_inp [K, n_expert] (F32) x [K, n_tokens],
 * written to `logits` with the bits qmm_op(MUL_MAT) gives, then the router above; one launch for build_moe_ffn's first six nodes. */
QMM_API int qmm_moe_router_logits_supported(const qmm_tensor * gate_inp, const qmm_tensor * x, const qmm_tensor * logits, const qmm_tensor * ids,
                                            const qmm_tensor * weights, int64_t n_used);
QMM_API int qmm_moe_router_logits(qmm_ctx * ctx, const qmm_tensor * gate_inp, const qmm_tensor * x, const qmm_tensor * logits, const qmm_tensor * ids,
                                  const qmm_tensor * weights, int64_t n_used, int normalise, void * stream);
/* ... and with the RMS norm in front of it (build_moe_ffn's input is ffn_norm = rms_norm(ffn_inp) * w, src/llama-model.cpp:4301-4305): `x` is the
 * un-normed row [K <= 16384, n_tokens <= 8], `normed` receives rms_norm(x, eps) * norm_w with the bits qmm_op(RMS_NORM_MUL) gives (the expert
 * MUL_MAT_IDs read it), the logits are taken against it.  `normed` may be `x` itself (same rows) and nothing else that overlaps it. */
QMM_API int qmm_moe_router_logits_norm_supported(const qmm_tensor * gate_inp, const qmm_tensor * x, const qmm_tensor * norm_w, const qmm_tensor * normed,
                                                 const qmm_tensor * logits, const qmm_tensor * ids, const qmm_tensor * weights, int64_t n_used);
QMM_API int qmm_moe_router_logits_norm(qmm_ctx * ctx, const qmm_tensor * gate_inp, const qmm_tensor * x, const qmm_tensor * norm_w, float eps,
                                       const qmm_tensor * normed, const qmm_tensor * logits, const qmm_tensor * ids, const qmm_tensor * weights,
                                       int64_t n_used, int normalise, void * stream);

/* The other end of the block: out [E, n_tokens] = sum over the used experts of x [E, n_used, n_tokens] * w [1, n_used, n_tokens]
 * (ggml_mul by the router weights, then the ggml_add chain over 2-D views, src/llama-graph.cpp:896-911), in the graph's order. */
QMM_API int qmm_moe_combine_supported(const qmm_tensor * x, const qmm_tensor * w, const qmm_tensor * out);
QMM_API int qmm_moe_combine(qmm_ctx * ctx, const qmm_tensor * x, const qmm_tensor * w, const qmm_tensor * out, void * stream);
/* ... with the residual add and the RMS norm that follow the block in the layer (src/llama-model.cpp:4318-4330, 4216: ffn_moe_out + ffn_inp
 * -> l_out, then the next attn_norm or result_norm): sum [E, n_tokens] = combine(x, w) + b, dst = rms_norm(sum) * nw [E]; one launch,
 * both results the bits of qmm_moe_combine followed by qmm_op_add_rms_norm.  With one token (one workgroup) sum and dst may overlap any input
 * (all inputs are read before the first store); with more tokens they must be clear of x and w, and of b except as the very same rows. */
QMM_API int qmm_moe_combine_add_rms_norm_supported(const qmm_tensor * x, const qmm_tensor * w, const qmm_tensor * b, const qmm_tensor * nw,
                                                   const qmm_tensor * sum, const qmm_tensor * dst);
QMM_API int qmm_moe_combine_add_rms_norm(qmm_ctx * ctx, const qmm_tensor * x, const qmm_tensor * w, const qmm_tensor * b, const qmm_tensor * nw,
                                         const qmm_tensor * sum, const qmm_tensor * dst, float eps, void * stream);

/* qmm_rope_kv_store and qmm_attn_decode as ONE launch for a batch of up to 8 tokens (normal-mode RoPE over the whole head,
 * D <= 128): q [D, H, N] arrives un-roped (q_rope: the ROPE node's descriptor, for its op_params), k_new [D, H_kv, N] and
 * v_new (v_cur^T, [N, Dv * H_kv]) are the batch's projections, k_store / v_store where qmm_rope_kv_store would put them, k / v the
 * cache views the attention reads, j0 the cache row of the batch's first token.  Every workgroup forms the new rows it needs
 * itself and takes them from LDS, so nothing reads rows j0 .. j0 + N of the cache while one workgroup per kv head stores them. */
QMM_API int qmm_attn_decode_rope_supported(const qmm_tensor * q, const qmm_tensor * pos, const qmm_tensor * ff, const qmm_tensor * q_rope,
                                           const qmm_tensor * k_new, const qmm_tensor * k_store, const qmm_tensor * v_new,
                                           const qmm_tensor * v_store, const qmm_tensor * k, const qmm_tensor * v, const qmm_tensor * mask,
                                           const qmm_tensor * dst, int64_t j0);
QMM_API int qmm_attn_decode_rope(qmm_ctx * ctx, const qmm_tensor * q, const qmm_tensor * pos, const qmm_tensor * ff, const qmm_tensor * q_rope,
                                 const qmm_tensor * k_new, const qmm_tensor * k_store, const qmm_tensor * v_new, const qmm_tensor * v_store,
                                 const qmm_tensor * k, const qmm_tensor * v, const qmm_tensor * mask, const qmm_tensor * dst, float scale,
                                 int64_t j0, void * stream);

#ifdef __cplusplus
}
#endif
#endif
