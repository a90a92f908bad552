/*
 * ggml_mi355x_qmm.h — C-ABI of the MI355X (gfx950) quantized MUL_MAT / MUL_MAT_ID kernels.
 *
 * This is the device-side boundary of the hot path: the place where zhouwg/ggml-hexagon crosses from
 * its host backend into the cDSP kernel library.  Each entry point names the reference interface it
 * replaces; paths are relative to the reference tree (ggml/src/ggml-hexagon/...).
 *
 *   reference                                                     here
 *   ------------------------------------------------------------  -----------------------------
 *   ggmlop_dsp_open / _close   (kernels/ggmlop_ap_skel.h:246-262,  qmm_create / qmm_destroy
 *     ggml-hexagon.cpp:4821-4973 ggmlhexagon_init_dsp)
 *   ggmlop_dsp_setclocks       (kernels/ggml-dsp.c:900-946)        — none needed on MI355X —
 *   rpcmem pool                (ggml-hexagon.cpp:4698-4747)        qmm_malloc / qmm_free / qmm_memcpy_*
 *   ggmlop_dsp_mulmat(h, src0, src1, dst)                          qmm_mul_mat / qmm_mul_mat_group
 *     (kernels/ggmlop_ap_skel.h:271, kernels/ggml-dsp.c:1192-1351;
 *      `dsptensor` wire struct kernels/ggmlop_ap_skel.h:234-244)
 *   (MUL_MAT_ID: absent from the reference, ggml-hexagon.cpp:514;  qmm_mul_mat_id
 *    semantics = ggml-cpu.c:6941-7197)
 *   from_float(src1 rows) inside ggmlop_dsp_mulmat                 qmm_quantize_act (exposed for parity tests;
 *     (kernels/ggml-dsp.c:1262-1285)                                 the hot kernels do it in LDS themselves)
 *   dequantize_row_q6_K etc.   (kernels/ggml-dsp.c:696-725)        qmm_dequantize
 *
 * Conventions
 *   - every `const void *` / `float *` data argument is a DEVICE pointer on the context's GPU;
 *   - `stream` is a hipStream_t passed as void* and used verbatim (NULL = HIP's default stream);
 *     qmm_stream() returns a non-blocking stream owned by the context for callers that want one.
 *     Calls are asynchronous on that stream; nothing synchronizes except qmm_synchronize / qmm_memcpy_h2d/_d2h;
 *   - `type` is ggml's enum ggml_type value: Q4_0=2, Q8_0=8, Q4_K=12, Q5_K=13, Q6_K=14 (the north-star's five) and, SURVEY 8f-4,
 *     Q4_1=3, Q5_0=6, Q5_1=7, Q2_K=10, Q3_K=11, IQ4_NL=20 (ggml/src/ggml-common.h:174-207, 253-277, 405-410);
 *   - weights are in GGUF wire layout: rows of blocks, `w_row_bytes` apart (>= K/blck*type_size);
 *   - returns 0 on success, a negative QMM_E* code otherwise (qmm_last_error() has the text, per thread).
 *     Nothing falls back to the CPU;
 *   - a context owns one scratch workspace (prefill operands, split-K partial tiles, MoE lists): issue its compute
 *     calls from one thread and on one stream at a time.  The workspace grows on demand, which synchronizes the
 *     device: run a shape once before capturing it into a hipGraph.  Several contexts per device are fine (the
 *     plugin's logical devices each have their own).
 */
#ifndef GGML_MI355X_QMM_H
#define GGML_MI355X_QMM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(_WIN32)
#  define QMM_API __declspec(dllexport)
#else
#  define QMM_API __attribute__((visibility("default")))
#endif

typedef struct qmm_ctx qmm_ctx;

enum {
    QMM_OK = 0,
    QMM_EINVAL = -1,      /* bad shape / alignment / type */
    QMM_ENODEV = -2,      /* no gfx950 device / HIP runtime failure at init */
    QMM_EHIP = -3,        /* a HIP call failed */
    QMM_ENOMEM = -4,
    QMM_EUNSUPPORTED = -5
};

/* activation rounding variant of the Q8_0 quantizer (see oracle/qmm_oracle.h) */
enum { QMM_ACT_REF = 0, QMM_ACT_X86 = 1 };

/* numerics of the batched (N > QMM_MATVEC_MAX_N) kernel */
enum {
    QMM_PREC_BF16 = 0,     /* dequantized weights and activations rounded to bf16, f32 MFMA accumulate   */
    QMM_PREC_F16_Q8 = 1    /* activations Q8-quantized as the CPU backend does, then f16 MFMA (default)  */
};
#define QMM_MATVEC_MAX_N 8

QMM_API int          qmm_abi_version(void);
QMM_API const char * qmm_last_error(void);
QMM_API int          qmm_device_count(void);

QMM_API qmm_ctx *    qmm_create(int device);
QMM_API void         qmm_destroy(qmm_ctx * ctx);
QMM_API int          qmm_device(const qmm_ctx * ctx);
QMM_API void *       qmm_stream(const qmm_ctx * ctx);
QMM_API int          qmm_device_info(const qmm_ctx * ctx, char * name, size_t name_len,
                                     size_t * mem_free, size_t * mem_total, int * compute_units);
QMM_API int          qmm_set_act_mode(qmm_ctx * ctx, int act_mode);
QMM_API int          qmm_set_precision(qmm_ctx * ctx, int prec);

QMM_API void *       qmm_malloc(qmm_ctx * ctx, size_t bytes);
QMM_API void         qmm_free(qmm_ctx * ctx, void * dptr);
/* page-locked host memory for staging (the rpcmem / ION pool of the reference, ggml-hexagon.cpp:4698-4747): copies from / to it
 * are DMA transfers and the *_async forms below really are asynchronous */
QMM_API void *       qmm_host_malloc(qmm_ctx * ctx, size_t bytes);
QMM_API void         qmm_host_free(qmm_ctx * ctx, void * hptr);
QMM_API int          qmm_memcpy_h2d(qmm_ctx * ctx, void * dst, const void * src, size_t bytes, void * stream);
QMM_API int          qmm_memcpy_d2h(qmm_ctx * ctx, void * dst, const void * src, size_t bytes, void * stream);
QMM_API int          qmm_memcpy_d2d(qmm_ctx * ctx, void * dst, const void * src, size_t bytes, void * stream);
/* the same copies without the host-side wait: ordered on `stream` only (a pageable host buffer may still make the runtime
 * stage or block; results are defined after qmm_synchronize or an event) */
QMM_API int          qmm_memcpy_h2d_async(qmm_ctx * ctx, void * dst, const void * src, size_t bytes, void * stream);
QMM_API int          qmm_memcpy_d2h_async(qmm_ctx * ctx, void * dst, const void * src, size_t bytes, void * stream);
QMM_API int          qmm_memset(qmm_ctx * ctx, void * dst, int value, size_t bytes, void * stream);
QMM_API int          qmm_synchronize(qmm_ctx * ctx, void * stream);

/* Cross-device plumbing for the row split (one process, several devices: ggml-cuda.cu:1365-1673 does the same with
 * cudaMemcpyPeerAsync / cudaMemcpy2DAsync and cudaEvents).  A copy may name memory of any device of this process
 * (qmm_create enables peer access between all gfx950 devices it can); it runs on `stream`, which belongs to `ctx`.
 * qmm_memcpy2d_d2d copies `height` runs of `width` bytes, the runs `dpitch` / `spitch` bytes apart.
 * Events order streams of different contexts: record on one, wait on another. */
typedef struct qmm_event qmm_event;
QMM_API int          qmm_memcpy2d_d2d(qmm_ctx * ctx, void * dst, size_t dpitch, const void * src, size_t spitch,
                                      size_t width, size_t height, void * stream);
QMM_API qmm_event *  qmm_event_create(qmm_ctx * ctx);
QMM_API qmm_event *  qmm_event_create_timing(qmm_ctx * ctx);                             /* an event pair of these can be timed */
QMM_API int          qmm_event_elapsed_ms(qmm_ctx * ctx, qmm_event * start, qmm_event * end, float * ms);   /* both recorded and complete */
QMM_API void         qmm_event_destroy(qmm_ctx * ctx, qmm_event * ev);
QMM_API int          qmm_event_record(qmm_ctx * ctx, qmm_event * ev, void * stream);
QMM_API int          qmm_stream_wait_event(qmm_ctx * ctx, void * stream, qmm_event * ev);
QMM_API int          qmm_event_synchronize(qmm_ctx * ctx, qmm_event * ev);              /* host waits for the event */

QMM_API size_t       qmm_row_size(int type, int64_t k);

/* Planar rows (weight repack at set_tensor; precedent ggml_backend_amx_buffer_set_tensor, ggml/src/ggml-cpu/amx/amx.cpp,
 * which converts to its own layout on upload).  Q4_0, Q8_0 and Q6_K blocks are 18, 34 and 210 bytes, so a lane's 16-byte loads sit at
 * 2-byte alignment in GGUF wire layout.  The planar form keeps every row's bytes in the row and re-lays them as aligned planes:
 *   Q4_0P [d: nb x 2][qs: nb x 16]   Q8_0P [d: nb x 2][qs: nb x 32]   Q6_KP [ql: nb x 128][qh: nb x 64][scales: nb x 16][d: nb x 2]
 * Row size and stride do not change.  qmm_planar_type returns the type code to pass to every entry point of this header for rows
 * in that form (type + 100), or 0 when (type, K, row stride) has none (K % 256, for Q6_K K % 2048, stride % 16).
 * qmm_repack_rows converts `rows` rows IN PLACE, to_planar != 0: wire -> planar, 0: back (byte-exact inverse).  `type` is the
 * wire type; w must be 16-byte aligned. */
QMM_API int qmm_planar_type(int type, int64_t K, int64_t w_row_bytes);
QMM_API int qmm_repack_rows(qmm_ctx * ctx, int type, void * w, int64_t w_row_bytes, int64_t rows, int64_t K, int to_planar, void * stream);

/* dst f32 [rows, K] (contiguous) = bit-exact unpack of `rows` weight rows */
QMM_API int qmm_dequantize(qmm_ctx * ctx, int type, const void * w, int64_t w_row_bytes,
                           int64_t rows, int64_t K, float * dst, void * stream);

/* Activation quantizer in the device layout the kernels use (structure of arrays):
 *   q  int8  [rows, K]
 *   d  f32   [rows, K/32]   for Q8_0 (the fp16-rounded scale, widened)   | [rows, K/256] for Q8_K
 *   bs int16 [rows, K/16]   Q8_K only (may be NULL); for Q8_1 this argument is the f32 array s [rows, K/32]:
 *                           block_q8_1's s = f16(d * sum of the block's int8), widened (ggml-quants.c:220-252)
 * `vec_dot_type` is 8 (Q8_0), 9 (Q8_1) or 15 (Q8_K).  x rows are ldx floats apart. */
QMM_API int qmm_quantize_act(qmm_ctx * ctx, int vec_dot_type, const float * x, int64_t rows, int64_t K,
                             int64_t ldx, int8_t * q, float * d, int16_t * bs, void * stream);

/* dst[n*ldd + m] = sum_k W[m,k] * x[n*ldx + k]    m < M, n < N   (GGML_OP_MUL_MAT, src1/dst f32) */
QMM_API int qmm_mul_mat(qmm_ctx * ctx, int type, const void * w, int64_t w_row_bytes, int64_t K, int64_t M,
                        const float * x, int64_t N, int64_t ldx, float * dst, int64_t ldd, void * stream);

/* several MUL_MATs that share src1 (wq/wk/wv, ffn gate/up) in one launch when N <= QMM_MATVEC_MAX_N */
typedef struct qmm_weight {
    const void * w;
    int64_t      w_row_bytes;
    int64_t      M;
    float *      dst;
    int64_t      ldd;
    int          type;
} qmm_weight;

QMM_API int qmm_mul_mat_group(qmm_ctx * ctx, const qmm_weight * ws, int n_weights, int64_t K,
                              const float * x, int64_t N, int64_t ldx, void * stream);

/* The same for a batch of <= QMM_MATVEC_MAX_N tokens with the two neighbours of the MUL_MAT in a transformer layer folded into the
 * launch (both optional):
 *   norm_w      x is rms_norm(x, norm_eps) * norm_w (one f32 row of K, 16-byte aligned), formed while the kernel stages its
 *               activations: attn_norm in front of wq/wk/wv, ffn_norm in front of ffn_gate/ffn_up (llama.cpp build_norm);
 *   residual[i] dst_i = W_i x + residual[i] (rows ldd_i apart, may be dst_i itself): the residual add behind wo / ffn_down;
 *   swiglu      1 or 2: exactly two matrices of one type and shape, ffn_gate and ffn_up; only ws[0].dst is written:
 *               dst = silu(W_a x) .* (W_b x) with a = swiglu - 1 (build_ffn's ggml_silu + ggml_mul), each wave computing one
 *               row of both. */
typedef struct qmm_mv_extra {
    const float * norm_w;
    float         norm_eps;
    const float * residual[4];
    int           swiglu;
    /* prompt batches (N > 8, round 3; qmm_mul_mat_group_norm_supported): with norm_w the activation prep forms rms_norm(x [+ norm_add]) * norm_w
     * itself; norm_add [N rows, norm_add_ld floats apart] is the residual the graph adds in front of the norm, norm_sum (may be NULL only
     * without norm_add) receives x + norm_add.  norm_sum may be x or norm_add itself (same rows), nothing else that overlaps them. */
    const float * norm_add;
    int64_t       norm_add_ld;
    float *       norm_sum;
    int64_t       norm_sum_ld;
} qmm_mv_extra;

QMM_API int qmm_mul_mat_group_ex(qmm_ctx * ctx, const qmm_weight * ws, int n_weights, int64_t K,
                                 const float * x, int64_t N, int64_t ldx, const qmm_mv_extra * extra, void * stream);
/* 1 when qmm_mul_mat_group_ex takes this group with extra->norm_w at N > 8 tokens: every matrix in a K-quant format whose prefill kernel
 * takes the register-resident Q8_K prep (Q2_K ... Q6_K), K a multiple of 1024 up to 16384, the default precision mode. */
QMM_API int qmm_mul_mat_group_norm_supported(qmm_ctx * ctx, const qmm_weight * ws, int n_weights, int64_t K, int64_t N);

/* Chains: a run of DEPENDENT one-token MUL_MAT groups as one persistent launch (token generation: wo -> ffn_gate/up -> ffn_down ->
 * the next layer's wq/wk/wv have nothing between them once the norm, the residual add and the SwiGLU are folded in, and every
 * launch of its own idles HBM for ~3.5 us of boundary, ramp and first-slice latency).  Between qmm_chain_begin and qmm_chain_end,
 * qmm_mul_mat_group / _ex calls with N == 1 are RECORDED, not launched; every other entry point of this library (and
 * qmm_chain_flush / _end / qmm_synchronize) first launches what was recorded, so the stream sees the calls in their order.  Inside a
 * persistent launch step s+1 starts after every workgroup has published step s (agent-scope arrival counters, write-through
 * stores, sc1 loads), and each wave requests its first weights of step s+1 before that wait.  Results are bit-identical to the
 * same calls made outside a recording.  A launch needs one resident workgroup per CU; a wait that cannot complete times out
 * (20 ms) and qmm_synchronize then returns QMM_EHIP.  GGML_MI355X_CHAIN=0 turns recording off (qmm_chain_begin becomes a no-op).
 * The reference has no counterpart: its AP->cDSP boundary is one FastRPC round trip per MUL_MAT (kernels/ggmlop_ap_skel.c:380-450). */
QMM_API int qmm_chain_begin(qmm_ctx * ctx);
QMM_API int qmm_chain_flush(qmm_ctx * ctx);
QMM_API int qmm_chain_end(qmm_ctx * ctx);
QMM_API int qmm_chain_stats(const qmm_ctx * ctx, int * persistent_launches, int * steps_in_them);
/* diagnostics: the following persistent launches write, per step and workgroup, 8 uint64 stamps of the 100 MHz clock (step start,
 * wait over, activations staged, ..., results published) to `stamps` ([steps][compute units][8], device memory, consecutive
 * launches behind each other); NULL switches it off.  profiles/tools/chain_stamps.py reads them. */
QMM_API int qmm_chain_debug(qmm_ctx * ctx, void * stamps);

/* Which kernels did a call issue?  Between qmm_trace_begin and qmm_trace_end every kernel launch of the MUL_MAT / MUL_MAT_ID entry
 * points appends its name (template arguments as rocprofv3 prints them, ';' behind each) to a host-side list; qmm_trace_end copies the
 * list into buf and returns the number of launches (< 0: error).  Host bookkeeping only, nothing on the device changes.  bench.py
 * buckets its roofline by these labels, so the figures describe the launches the timed pass issues (ggml has no counterpart: its CPU
 * backend has one code path per type, ggml-cpu.c:6745-6937). */
QMM_API int qmm_trace_begin(qmm_ctx * ctx);
QMM_API int qmm_trace_end(qmm_ctx * ctx, char * buf, size_t len);

/* dst = W * (silu(gate) .* up) for a prompt batch (N > QMM_MATVEC_MAX_N): ffn_down with build_ffn's SwiGLU product formed by the
 * activation prep of the MFMA path; gate / up rows are ld_gate / ld_up floats apart.  Default prefill precision only. */
QMM_API int qmm_mul_mat_swiglu_in(qmm_ctx * ctx, int type, const void * w, int64_t w_row_bytes, int64_t K, int64_t M,
                                  const float * gate, int64_t ld_gate, const float * up, int64_t ld_up, int64_t N,
                                  float * dst, int64_t ldd, void * stream);

/* GGML_OP_MUL_MAT_ID.
 *   as   [K, M, n_expert]  experts `expert_bytes` apart
 *   b    f32 [K, ne11, n_tokens]   element (k, i11, t) at b[t*b_nb2/4 + i11*b_nb1/4 + k]   (ne11 = n_used or 1)
 *   ids  int32 [n_used, n_tokens]  DEVICE pointer, rows `ids_nb1` bytes apart (strided views allowed)
 *   dst  f32 [M, n_used, n_tokens] element (m, s, t) at dst[t*d_nb2/4 + s*d_nb1/4 + m]
 * An id outside [0, n_expert) leaves its dst row untouched and makes the call return QMM_EINVAL
 * at the next qmm_synchronize (the CPU backend asserts, ggml-cpu.c:7115). */
QMM_API int qmm_mul_mat_id(qmm_ctx * ctx, int type, const void * as, int64_t w_row_bytes, int64_t expert_bytes,
                           int64_t K, int64_t M, int64_t n_expert,
                           const float * b, int64_t ne11, int64_t b_nb1, int64_t b_nb2,
                           const int32_t * ids, int64_t n_used, int64_t n_tokens, int64_t ids_nb1,
                           float * dst, int64_t d_nb1, int64_t d_nb2, void * stream);

/* Two expert tensors of the same type and shape on the SAME b and ids — ffn_gate_exps and ffn_up_exps, consecutive nodes of
 * llama.cpp's MoE block (src/llama-graph.cpp build_moe_ffn): one mat-vec launch for both at small batches, one expert sort
 * and one activation prep for both at large ones.  dst0 / dst1 have the same strides. */
QMM_API int qmm_mul_mat_id_pair(qmm_ctx * ctx, int type, const void * as0, const void * as1, int64_t w_row_bytes, int64_t expert_bytes,
                                int64_t K, int64_t M, int64_t n_expert,
                                const float * b, int64_t ne11, int64_t b_nb1, int64_t b_nb2,
                                const int32_t * ids, int64_t n_used, int64_t n_tokens, int64_t ids_nb1,
                                float * dst0, float * dst1, int64_t d_nb1, int64_t d_nb2, void * stream);

/* The same two tensors with the SwiGLU that follows them in build_moe_ffn (src/llama-graph.cpp:870-894: ffn_moe_gate, ffn_moe_up,
 * ggml_silu, ggml_mul -> ffn_moe_gate_par): dst[:, s, t] = silu(as_gate[ids[s, t]] . b) * (as_up[ids[s, t]] . b), one launch, for up to 16
 * (token, slot) pairs (token generation; larger batches: qmm_mul_mat_id_pair and the SILU_MUL op).  Same float operations as those. */
QMM_API int qmm_mul_mat_id_swiglu_supported(int64_t n_used, int64_t n_tokens);
QMM_API int qmm_mul_mat_id_swiglu(qmm_ctx * ctx, int type, const void * as_gate, const void * as_up, int64_t w_row_bytes, int64_t expert_bytes,
                                  int64_t K, int64_t M, int64_t n_expert,
                                  const float * b, int64_t ne11, int64_t b_nb1, int64_t b_nb2,
                                  const int32_t * ids, int64_t n_used, int64_t n_tokens, int64_t ids_nb1,
                                  float * dst, int64_t d_nb1, int64_t d_nb2, void * stream);

/* ---- RCCL exchange for a row split driven from ONE process ----------------------------------------------------------
 * Replaces: the reference's only row-split data path, ggml_cuda_op_mul_mat's cudaMemcpyPeerAsync of src1 to every device and of the
 * dst slices back to the main device (ggml/src/ggml-cuda/ggml-cuda.cu:1365-1673; placement of the slices :1603-1625).  Rank r of a
 * communicator is ctxs[r]; one rank per device (RCCL's rule).  Every call is enqueued on the ranks' streams (streams[r], or the
 * context's own stream when `streams` is NULL) and returns at once.  librccl.so is opened on the first qmm_comm_create;
 * QMM_EUNSUPPORTED when it cannot be found (GGML_MI355X_RCCL_LIB names another path).
 * The plugin uses it for split MUL_MATs with GGML_MI355X_RCCL=1 (default: peer copies + events, which need no library);
 * one-process-per-GPU callers exchange through torch.distributed's RCCL instead (ggml-hexagon_amd/rowsplit.py). */
typedef struct qmm_comm qmm_comm;
QMM_API int  qmm_comm_create(qmm_ctx * const * ctxs, int n, qmm_comm ** out);
QMM_API void qmm_comm_destroy(qmm_comm * comm);
QMM_API int  qmm_comm_size(const qmm_comm * comm);
/* bufs[root] (on rank root) -> bufs[r] on every rank, `bytes` each: src1 of a split MUL_MAT */
QMM_API int  qmm_comm_broadcast(qmm_comm * comm, int root, void * const * bufs, size_t bytes, void * const * streams);
/* send[r] (bytes[r] on rank r) -> recv[r] on the root, for every r != root with bytes[r] > 0: the dst slices, one grouped launch */
QMM_API int  qmm_comm_gather(qmm_comm * comm, int root, const void * const * send, void * const * recv, const size_t * bytes,
                             void * const * streams);
/* send[r] (`bytes` on rank r) -> recv[r] (size * bytes on every rank, in rank order) */
QMM_API int  qmm_comm_all_gather(qmm_comm * comm, const void * const * send, void * const * recv, size_t bytes, void * const * streams);

#ifdef __cplusplus
}
#endif
#endif
