"""ctypes binding of include/ggml_mi355x_qmm.h.  Device pointers in, device pointers out; torch is used by
callers only to own HBM buffers and streams.  There is no fallback: if the HIP library cannot be loaded or
the device is not gfx950, construction raises."""
from __future__ import annotations

import ctypes as C

from . import build as _build

Q4_0, Q8_0, Q4_K, Q5_K, Q6_K, Q8_K = 2, 8, 12, 13, 14, 15
Q4_1, Q5_0, Q5_1, Q2_K, Q3_K, IQ4_NL, IQ4_XS = 3, 6, 7, 10, 11, 20, 23
ACT_REF, ACT_X86 = 0, 1
PREC_BF16, PREC_F16_Q8 = 0, 1
MATVEC_MAX_N = 8

EXPORTS = [
    "qmm_abi_version", "qmm_last_error", "qmm_device_count", "qmm_create", "qmm_destroy", "qmm_device", "qmm_stream",
    "qmm_device_info", "qmm_set_act_mode", "qmm_set_precision", "qmm_malloc", "qmm_free", "qmm_host_malloc", "qmm_host_free", "qmm_memcpy_h2d",
    "qmm_memcpy_d2h", "qmm_memcpy_d2d", "qmm_memset", "qmm_synchronize", "qmm_memcpy2d_d2d", "qmm_event_create",
    "qmm_event_destroy", "qmm_event_record", "qmm_stream_wait_event", "qmm_event_synchronize", "qmm_event_create_timing", "qmm_event_elapsed_ms", "qmm_memcpy_h2d_async",
    "qmm_memcpy_d2h_async", "qmm_row_size", "qmm_planar_type", "qmm_repack_rows", "qmm_dequantize",
    "qmm_quantize_act", "qmm_mul_mat", "qmm_mul_mat_group", "qmm_mul_mat_group_ex", "qmm_mul_mat_group_norm_supported", "qmm_mul_mat_swiglu_in", "qmm_mul_mat_id", "qmm_mul_mat_id_pair", "qmm_mul_mat_id_swiglu_supported", "qmm_mul_mat_id_swiglu",
    "qmm_chain_begin", "qmm_chain_flush", "qmm_chain_end", "qmm_chain_stats", "qmm_chain_debug", "qmm_trace_begin", "qmm_trace_end",
    "qmm_comm_create", "qmm_comm_destroy", "qmm_comm_size", "qmm_comm_broadcast", "qmm_comm_gather", "qmm_comm_all_gather",
]
# include/ggml_mi355x_ops.h: the glue ops of a transformer layer
OPS_EXPORTS = [
    "qmm_op_supported", "qmm_op_compute", "qmm_op_add_rms_norm_supported", "qmm_op_add_rms_norm",
    "qmm_attn_decode_supported", "qmm_attn_decode", "qmm_attn_prefill_supported", "qmm_attn_prefill", "qmm_rope_kv_store_supported", "qmm_rope_kv_store", "qmm_attn_decode_rope_supported", "qmm_attn_decode_rope", "qmm_moe_router_supported", "qmm_moe_router", "qmm_moe_router_logits_supported", "qmm_moe_router_logits", "qmm_moe_router_logits_norm_supported", "qmm_moe_router_logits_norm", "qmm_moe_combine_supported", "qmm_moe_combine", "qmm_moe_combine_add_rms_norm_supported", "qmm_moe_combine_add_rms_norm",
]


class QmmMvExtra(C.Structure):
    _fields_ = [("norm_w", C.c_void_p), ("norm_eps", C.c_float), ("residual", C.c_void_p * 4), ("swiglu", C.c_int),
                ("norm_add", C.c_void_p), ("norm_add_ld", C.c_int64), ("norm_sum", C.c_void_p), ("norm_sum_ld", C.c_int64)]


class QmmTensor(C.Structure):
    """qmm_tensor of include/ggml_mi355x_ops.h (the dsptensor of this boundary)"""
    _fields_ = [("data", C.c_void_p), ("type", C.c_int32), ("flags", C.c_int32), ("ne", C.c_int64 * 4), ("nb", C.c_int64 * 4),
                ("op_params", C.c_int32 * 16)]

    @classmethod
    def make(cls, type_, ne, nb=None, data=0, op_params=()):
        es = {0: 4, 1: 2, 26: 4}.get(type_, 1)
        ne = list(ne) + [1] * (4 - len(ne))
        if nb is None:
            nb, acc = [], es
            for n in ne:
                nb.append(acc)
                acc *= n
        t = cls()
        t.data, t.type, t.flags = data, type_, 0
        t.ne[:] = ne
        t.nb[:] = list(nb)
        for i, v in enumerate(op_params):
            t.op_params[i] = v
        return t


# enum qmm_op
(OP_ADD, OP_SUB, OP_MUL, OP_DIV, OP_SCALE, OP_SILU, OP_GELU, OP_GELU_QUICK, OP_RELU, OP_TANH, OP_SIGMOID, OP_NEG, OP_EXP, OP_RMS_NORM,
 OP_ROPE, OP_SOFT_MAX, OP_CPY, OP_GET_ROWS, OP_MUL_MAT_F, OP_RMS_NORM_MUL, OP_SILU_MUL, OP_ARGSORT, OP_SUM_ROWS, OP_NORM) = range(1, 25)


class QmmWeight(C.Structure):
    _fields_ = [("w", C.c_void_p), ("w_row_bytes", C.c_int64), ("M", C.c_int64), ("dst", C.c_void_p),
                ("ldd", C.c_int64), ("type", C.c_int)]


class QmmError(RuntimeError):
    pass


def load_library() -> C.CDLL:
    so = _build.build_qmm()
    lib = C.CDLL(str(so))
    v, i64, i32, sz = C.c_void_p, C.c_int64, C.c_int, C.c_size_t
    lib.qmm_last_error.restype = C.c_char_p
    lib.qmm_create.restype = v
    lib.qmm_create.argtypes = [i32]
    lib.qmm_destroy.argtypes = [v]
    lib.qmm_device.argtypes = [v]
    lib.qmm_stream.restype = v
    lib.qmm_stream.argtypes = [v]
    lib.qmm_device_info.argtypes = [v, C.c_char_p, sz, C.POINTER(sz), C.POINTER(sz), C.POINTER(i32)]
    lib.qmm_set_act_mode.argtypes = [v, i32]
    lib.qmm_set_precision.argtypes = [v, i32]
    lib.qmm_malloc.restype = v
    lib.qmm_malloc.argtypes = [v, sz]
    lib.qmm_free.argtypes = [v, v]
    for f in (lib.qmm_memcpy_h2d, lib.qmm_memcpy_d2h, lib.qmm_memcpy_d2d):
        f.argtypes = [v, v, v, sz, v]
    lib.qmm_memset.argtypes = [v, v, i32, sz, v]
    lib.qmm_synchronize.argtypes = [v, v]
    lib.qmm_memcpy2d_d2d.argtypes = [v, v, sz, v, sz, sz, sz, v]
    lib.qmm_event_create.restype = v
    lib.qmm_event_create.argtypes = [v]
    lib.qmm_event_destroy.argtypes = [v, v]
    lib.qmm_event_record.argtypes = [v, v, v]
    lib.qmm_stream_wait_event.argtypes = [v, v, v]
    lib.qmm_event_synchronize.argtypes = [v, v]
    lib.qmm_memcpy_h2d_async.argtypes = [v, v, v, sz, v]
    lib.qmm_memcpy_d2h_async.argtypes = [v, v, v, sz, v]
    lib.qmm_row_size.restype = sz
    lib.qmm_row_size.argtypes = [i32, i64]
    lib.qmm_dequantize.argtypes = [v, i32, v, i64, i64, i64, v, v]
    lib.qmm_planar_type.argtypes = [i32, i64, i64]
    lib.qmm_repack_rows.argtypes = [v, i32, v, i64, i64, i64, i32, v]
    lib.qmm_quantize_act.argtypes = [v, i32, v, i64, i64, i64, v, v, v, v]
    lib.qmm_mul_mat.argtypes = [v, i32, v, i64, i64, i64, v, i64, i64, v, i64, v]
    lib.qmm_mul_mat_group.argtypes = [v, C.POINTER(QmmWeight), i32, i64, v, i64, i64, v]
    lib.qmm_mul_mat_swiglu_in.argtypes = [v, i32, v, i64, i64, i64, v, i64, v, i64, i64, v, i64, v]
    lib.qmm_mul_mat_group_ex.argtypes = [v, C.POINTER(QmmWeight), i32, i64, v, i64, i64, C.POINTER(QmmMvExtra), v]
    P = C.POINTER(QmmTensor)
    lib.qmm_op_supported.argtypes = [i32, P, P, P, P]
    lib.qmm_op_compute.argtypes = [v, i32, P, P, P, P, v]
    lib.qmm_op_add_rms_norm.argtypes = [v, P, P, P, P, P, C.c_float, v]
    lib.qmm_attn_decode.argtypes = [v, P, P, P, P, P, C.c_float, v]
    lib.qmm_attn_prefill.argtypes = [v, P, P, P, P, P, C.c_float, v]
    lib.qmm_moe_combine.argtypes = [v, P, P, P, v]
    lib.qmm_moe_combine_add_rms_norm.argtypes = [v, P, P, P, P, P, P, C.c_float, v]
    lib.qmm_moe_router.argtypes = [v, P, P, P, i64, i32, v]
    lib.qmm_moe_router_logits.argtypes = [v, P, P, P, P, P, i64, i32, v]
    lib.qmm_moe_router_logits_norm_supported.argtypes = [P, P, P, P, P, P, P, i64]
    lib.qmm_moe_router_logits_norm.argtypes = [v, P, P, P, C.c_float, P, P, P, P, i64, i32, v]
    lib.qmm_rope_kv_store.argtypes = [v, P, P, P, P, P, P, P, P, v]
    lib.qmm_attn_decode_rope.argtypes = [v, P, P, P, P, P, P, P, P, P, P, P, P, C.c_float, i64, v]
    lib.qmm_mul_mat_id.argtypes = [v, i32, v, i64, i64, i64, i64, i64, v, i64, i64, i64, v, i64, i64, i64, v, i64, i64, v]
    lib.qmm_mul_mat_id_pair.argtypes = [v, i32, v, v, i64, i64, i64, i64, i64, v, i64, i64, i64, v, i64, i64, i64, v, v, i64, i64, v]
    lib.qmm_mul_mat_id_swiglu_supported.argtypes = [i64, i64]
    lib.qmm_mul_mat_id_swiglu.argtypes = [v, i32, v, v, i64, i64, i64, i64, i64, v, i64, i64, i64, v, i64, i64, i64, v, i64, i64, v]
    for f in (lib.qmm_chain_begin, lib.qmm_chain_flush, lib.qmm_chain_end):
        f.argtypes = [v]
    lib.qmm_chain_stats.argtypes = [v, C.POINTER(i32), C.POINTER(i32)]
    lib.qmm_chain_debug.argtypes = [v, v]
    lib.qmm_trace_begin.argtypes = [v]
    lib.qmm_trace_end.argtypes = [v, C.c_char_p, sz]
    return lib


class Qmm:
    """One context per GPU (the analogue of the reference's cDSP session, ggml-hexagon.cpp:4821-4973)."""

    def __init__(self, device: int = 0):
        self.lib = load_library()
        self.ctx = self.lib.qmm_create(device)
        if not self.ctx:
            raise QmmError(self.lib.qmm_last_error().decode())
        self.device = device

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.qmm_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise QmmError(f"qmm error {rc}: {self.lib.qmm_last_error().decode()}")

    @staticmethod
    def _stream():
        import torch
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def set_act_mode(self, m):
        self._chk(self.lib.qmm_set_act_mode(self.ctx, m))

    def set_precision(self, p):
        self._chk(self.lib.qmm_set_precision(self.ctx, p))

    def synchronize(self):
        self._chk(self.lib.qmm_synchronize(self.ctx, self._stream()))

    def row_size(self, t, k):
        return self.lib.qmm_row_size(t, k)

    # --- chains: one-token mul_mat_group[_ex] calls between begin and end are recorded and go out as persistent launches
    def chain_begin(self):
        self._chk(self.lib.qmm_chain_begin(self.ctx))

    def chain_flush(self):
        self._chk(self.lib.qmm_chain_flush(self.ctx))

    def chain_end(self):
        self._chk(self.lib.qmm_chain_end(self.ctx))

    def chain_debug(self, stamps=None):
        """stamps: int64 CUDA tensor [steps, cus, 8] (or None to switch off)"""
        self._chk(self.lib.qmm_chain_debug(self.ctx, stamps.data_ptr() if stamps is not None else None))

    def chain_stats(self):
        a, b = C.c_int(0), C.c_int(0)
        self._chk(self.lib.qmm_chain_stats(self.ctx, C.byref(a), C.byref(b)))
        return a.value, b.value

    def trace(self, fn):
        """the kernels `fn()` launches through this context, as a tuple of labels (qmm_trace_begin / qmm_trace_end)"""
        self._chk(self.lib.qmm_trace_begin(self.ctx))
        try:
            fn()
        finally:
            buf = C.create_string_buffer(8192)
            n = self.lib.qmm_trace_end(self.ctx, buf, len(buf))
        if n < 0:
            self._chk(n)
        return tuple(x for x in buf.value.decode().split(";") if x)

    def planar_type(self, t, k, row_bytes):
        return self.lib.qmm_planar_type(t, k, row_bytes)

    def repack_rows(self, t, w, k, to_planar=True):
        """in place: w uint8 [..., rows, row_bytes] (contiguous rows) between GGUF wire layout and the planar layout; returns the
        type code to use for w afterwards"""
        rows = w.numel() // w.shape[-1]
        self._chk(self.lib.qmm_repack_rows(self.ctx, t, w.data_ptr(), w.stride(-2), rows, k, 1 if to_planar else 0, self._stream()))
        return t + 100 if to_planar else t

    # --- torch-tensor conveniences (uint8 weight tensors [M, row_bytes] on the GPU) ---------------
    def dequantize(self, t, w, k):
        import torch
        rows = w.shape[0]
        out = torch.empty((rows, k), dtype=torch.float32, device=w.device)
        self._chk(self.lib.qmm_dequantize(self.ctx, t, w.data_ptr(), w.stride(0), rows, k, out.data_ptr(), self._stream()))
        return out

    def quantize_act(self, vec_dot_type, x):
        import torch
        rows, k = x.shape
        qb = 256 if vec_dot_type == Q8_K else 32
        q = torch.empty((rows, k), dtype=torch.int8, device=x.device)
        d = torch.empty((rows, k // qb), dtype=torch.float32, device=x.device)
        bs = torch.zeros((rows, k // 16), dtype=torch.int16, device=x.device) if vec_dot_type == Q8_K else None
        if vec_dot_type == 9:               # Q8_1: the third array is s = f16(d * sum(q)) per block, as f32
            bs = torch.zeros((rows, k // 32), dtype=torch.float32, device=x.device)
        self._chk(self.lib.qmm_quantize_act(self.ctx, vec_dot_type, x.data_ptr(), rows, k, x.stride(0), q.data_ptr(),
                                            d.data_ptr(), bs.data_ptr() if bs is not None else None, self._stream()))
        return q, d, bs

    def mul_mat(self, t, w, k, x, out=None):
        """w uint8 [M, row_bytes]; x f32 [N, K] -> f32 [N, M]"""
        import torch
        m, n = w.shape[0], x.shape[0]
        if out is None:
            out = torch.empty((n, m), dtype=torch.float32, device=x.device)
        self._chk(self.lib.qmm_mul_mat(self.ctx, t, w.data_ptr(), w.stride(0), k, m, x.data_ptr(), n, x.stride(0),
                                       out.data_ptr(), out.stride(0), self._stream()))
        return out

    def mul_mat_group(self, weights, k, x, outs):
        """weights: list of (type, w uint8 [M,row_bytes]); outs: list of f32 [N, M]"""
        arr = (QmmWeight * len(weights))()
        for i, ((t, w), o) in enumerate(zip(weights, outs)):
            arr[i] = QmmWeight(w.data_ptr(), w.stride(0), w.shape[0], o.data_ptr(), o.stride(0), t)
        self._chk(self.lib.qmm_mul_mat_group(self.ctx, arr, len(weights), k, x.data_ptr(), x.shape[0], x.stride(0), self._stream()))
        return outs

    def mul_mat_group_call(self, weights, k, x, outs):
        """mul_mat_group with the argument marshalling done once: returns a closure that issues the call on the current stream
        (bench.py's passes repeat the same 129 calls; rebuilding the ctypes array per call made the eager prompt pass host-bound on a
        slow host)"""
        arr = (QmmWeight * len(weights))()
        for i, ((t, w), o) in enumerate(zip(weights, outs)):
            arr[i] = QmmWeight(w.data_ptr(), w.stride(0), w.shape[0], o.data_ptr(), o.stride(0), t)
        n, xp, nt, ldx, fn, ctx = len(weights), x.data_ptr(), x.shape[0], x.stride(0), self.lib.qmm_mul_mat_group, self.ctx
        keep = (weights, x, outs)                                   # the closure owns the buffers it points into

        def call():
            rc = fn(ctx, arr, n, k, xp, nt, ldx, self._stream())
            if rc:
                self._chk(rc)
            return keep[2]
        return call

    def mul_mat_group_ex(self, weights, k, x, outs, norm_w=None, eps=0.0, residuals=None, swiglu=0, norm_add=None, norm_sum=None):
        """qmm_mul_mat_group_ex: x -> rms_norm(x, eps) * norm_w while staging (optional), outs[i] = W_i x + residuals[i] (optional);
        prompt batches: norm_add is added to x in front of the norm and norm_sum receives the sum"""
        arr = (QmmWeight * len(weights))()
        for i, ((t, w), o) in enumerate(zip(weights, outs)):
            arr[i] = QmmWeight(w.data_ptr(), w.stride(0), w.shape[0], o.data_ptr(), o.stride(0), t)
        ex = QmmMvExtra()
        ex.norm_w = norm_w.data_ptr() if norm_w is not None else None
        ex.norm_eps = eps
        ex.swiglu = swiglu
        if norm_add is not None:
            ex.norm_add, ex.norm_add_ld = norm_add.data_ptr(), norm_add.stride(0)
        if norm_sum is not None:
            ex.norm_sum, ex.norm_sum_ld = norm_sum.data_ptr(), norm_sum.stride(0)
        for i in range(4):
            r = residuals[i] if residuals is not None and i < len(residuals) else None
            ex.residual[i] = r.data_ptr() if r is not None else None
        self._chk(self.lib.qmm_mul_mat_group_ex(self.ctx, arr, len(weights), k, x.data_ptr(), x.shape[0], x.stride(0), C.byref(ex),
                                                self._stream()))
        return outs

    def op(self, op, dst, src0=None, src1=None, src2=None):
        """qmm_op_compute on QmmTensor descriptors (include/ggml_mi355x_ops.h)"""
        r = lambda t: C.byref(t) if t is not None else None
        self._chk(self.lib.qmm_op_compute(self.ctx, op, r(src0), r(src1), r(src2), r(dst), self._stream()))

    def mul_mat_id(self, t, w, k, b, ids, out=None):
        """w uint8 [n_expert, M, row_bytes]; b f32 [n_tokens, ne11, K]; ids int32 [n_tokens, n_used] (row-strided view ok)
        -> f32 [n_tokens, n_used, M]"""
        import torch
        n_expert, m = w.shape[0], w.shape[1]
        n_tokens, ne11 = b.shape[0], b.shape[1]
        n_used = ids.shape[1]
        assert ids.stride(1) == 1
        if out is None:
            out = torch.empty((n_tokens, n_used, m), dtype=torch.float32, device=b.device)
        self._chk(self.lib.qmm_mul_mat_id(self.ctx, t, w.data_ptr(), w.stride(1), w.stride(0), k, m, n_expert,
                                          b.data_ptr(), ne11, b.stride(1) * 4, b.stride(0) * 4,
                                          ids.data_ptr(), n_used, n_tokens, ids.stride(0) * 4,
                                          out.data_ptr(), out.stride(1) * 4, out.stride(0) * 4, self._stream()))
        return out

    def mul_mat_id_swiglu(self, t, w_gate, w_up, k, b, ids, out):
        """silu(ffn_gate_exps . b) * (ffn_up_exps . b) for a few (token, slot) pairs in one launch"""
        n_expert, m = w_gate.shape[0], w_gate.shape[1]
        n_tokens, ne11 = b.shape[0], b.shape[1]
        n_used = ids.shape[1]
        assert ids.stride(1) == 1 and w_up.shape == w_gate.shape and w_up.stride() == w_gate.stride()
        self._chk(self.lib.qmm_mul_mat_id_swiglu(self.ctx, t, w_gate.data_ptr(), w_up.data_ptr(), w_gate.stride(1), w_gate.stride(0), k, m, n_expert,
                                                 b.data_ptr(), ne11, b.stride(1) * 4, b.stride(0) * 4,
                                                 ids.data_ptr(), n_used, n_tokens, ids.stride(0) * 4,
                                                 out.data_ptr(), out.stride(1) * 4, out.stride(0) * 4, self._stream()))
        return out

    def mul_mat_id_pair(self, t, w0, w1, k, b, ids, out0, out1):
        """two expert tensors of one type and shape on the same b and ids (ffn_gate_exps + ffn_up_exps)"""
        n_expert, m = w0.shape[0], w0.shape[1]
        n_tokens, ne11 = b.shape[0], b.shape[1]
        n_used = ids.shape[1]
        assert ids.stride(1) == 1 and w1.shape == w0.shape and w1.stride() == w0.stride() and out0.stride() == out1.stride()
        self._chk(self.lib.qmm_mul_mat_id_pair(self.ctx, t, w0.data_ptr(), w1.data_ptr(), w0.stride(1), w0.stride(0), k, m, n_expert,
                                               b.data_ptr(), ne11, b.stride(1) * 4, b.stride(0) * 4,
                                               ids.data_ptr(), n_used, n_tokens, ids.stride(0) * 4,
                                               out0.data_ptr(), out1.data_ptr(), out0.stride(1) * 4, out0.stride(0) * 4, self._stream()))
        return out0, out1
