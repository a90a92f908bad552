"""ctypes bindings for the two CPU checkers.  TEST INFRASTRUCTURE ONLY.

* ``Oracle``  — oracle/libqmm_oracle.so, our plain-C restatement (oracle/qmm_oracle.c).
* ``RefGgml`` — oracle/_ref/libggml-{base,cpu}.so, the REAL reference compiled from /root/reference by
  oracle/Makefile.  Used to pin the restatement, to generate tests/golden/, and as the CPU baseline
  (``cpu_baseline.kind == "reference"``) in bench.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Nothing here reads /root/reference at run time.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent

Q4_0, Q8_0, Q4_K, Q5_K, Q6_K, Q8_K = 2, 8, 12, 13, 14, 15
Q4_1, Q5_0, Q5_1, Q8_1, Q2_K, Q3_K, IQ4_NL, IQ4_XS = 3, 6, 7, 9, 10, 11, 20, 23
TYPE_NAMES = {Q4_0: "q4_0", Q8_0: "q8_0", Q4_K: "q4_K", Q5_K: "q5_K", Q6_K: "q6_K", Q8_K: "q8_K",
              Q4_1: "q4_1", Q5_0: "q5_0", Q5_1: "q5_1", Q8_1: "q8_1", Q2_K: "q2_K", Q3_K: "q3_K", IQ4_NL: "iq4_nl", IQ4_XS: "iq4_xs"}
WEIGHT_TYPES = (Q4_0, Q8_0, Q4_K, Q5_K, Q6_K, Q4_1, Q5_0, Q5_1, Q2_K, Q3_K, IQ4_NL, IQ4_XS)


def vec_dot_type(t: int) -> int:
    """type_traits_cpu[t].vec_dot_type (ggml/src/ggml-cpu/ggml-cpu.c:256-…)"""
    return Q8_1 if t in (Q4_1, Q5_1) else Q8_0 if t in (Q4_0, Q8_0, Q5_0, IQ4_NL) else Q8_K
ACT_REF, ACT_X86 = 0, 1


def _ptr(a: np.ndarray):
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def build_oracle() -> Path:
    so = HERE / "libqmm_oracle.so"
    src = HERE / "qmm_oracle.c"
    if not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["make", "-C", str(HERE), "oracle"], check=True, capture_output=True)
    return so


class Oracle:
    def __init__(self):
        self.lib = C.CDLL(str(build_oracle()))
        L = self.lib
        L.qmo_blck_size.restype = C.c_int
        L.qmo_type_size.restype = C.c_size_t
        L.qmo_row_size.restype = C.c_size_t
        L.qmo_row_size.argtypes = [C.c_int, C.c_int64]
        L.qmo_fp16_to_fp32.restype = C.c_float
        L.qmo_fp16_to_fp32.argtypes = [C.c_uint16]
        L.qmo_fp32_to_fp16.restype = C.c_uint16
        L.qmo_fp32_to_fp16.argtypes = [C.c_float]
        L.qmo_dequantize_row.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int64]
        L.qmo_quantize_row_q8_0.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int]
        L.qmo_quantize_row_q8_1.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int]
        L.qmo_quantize_row_q8_K.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.qmo_vec_dot.restype = C.c_float
        L.qmo_vec_dot.argtypes = [C.c_int, C.c_int64, C.c_void_p, C.c_void_p]
        L.qmo_mul_mat.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64,
                                  C.c_void_p, C.c_int64, C.c_int]
        L.qmo_mul_mat_id.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_int64,
                                     C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int]

    def row_size(self, t, k):
        return self.lib.qmo_row_size(t, k)

    def dequantize(self, t, w: np.ndarray, k: int) -> np.ndarray:
        """w: uint8 [rows, row_size] -> f32 [rows, k]"""
        w = np.ascontiguousarray(w, dtype=np.uint8).reshape(-1, self.row_size(t, k))
        out = np.empty((w.shape[0], k), np.float32)
        for r in range(w.shape[0]):
            rc = self.lib.qmo_dequantize_row(t, _ptr(w[r]), _ptr(out[r]), k)
            assert rc == 0
        return out

    def quantize_act(self, t, x: np.ndarray, act_mode=ACT_REF) -> np.ndarray:
        """x f32 [rows, k] -> uint8 [rows, row_size(vec_dot_type)] ; t is the WEIGHT type"""
        x = np.ascontiguousarray(x, np.float32)
        rows, k = x.shape
        vt = self.lib.qmo_vec_dot_type(t)
        out = np.zeros((rows, self.row_size(vt, k)), np.uint8)
        for r in range(rows):
            if vt == Q8_0:
                self.lib.qmo_quantize_row_q8_0(_ptr(x[r]), _ptr(out[r]), k, act_mode)
            elif vt == Q8_1:
                self.lib.qmo_quantize_row_q8_1(_ptr(x[r]), _ptr(out[r]), k, act_mode)
            else:
                self.lib.qmo_quantize_row_q8_K(_ptr(x[r]), _ptr(out[r]), k)
        return out

    def vec_dot(self, t, k, w_row: np.ndarray, a_row: np.ndarray) -> float:
        return float(self.lib.qmo_vec_dot(t, k, _ptr(np.ascontiguousarray(w_row)), _ptr(np.ascontiguousarray(a_row))))

    def mul_mat(self, t, w: np.ndarray, k: int, x: np.ndarray, act_mode=ACT_REF) -> np.ndarray:
        """w uint8 [M, row_size]; x f32 [N, K] -> dst f32 [N, M]"""
        w = np.ascontiguousarray(w, np.uint8).reshape(-1, self.row_size(t, k))
        x = np.ascontiguousarray(x, np.float32)
        m, n = w.shape[0], x.shape[0]
        dst = np.empty((n, m), np.float32)
        rc = self.lib.qmo_mul_mat(t, _ptr(w), k, m, _ptr(x), n, x.shape[1], _ptr(dst), m, act_mode)
        assert rc == 0, rc
        return dst

    def mul_mat_id(self, t, w: np.ndarray, k: int, m: int, b: np.ndarray, ids: np.ndarray, act_mode=ACT_REF):
        """w uint8 [n_expert, M, row_size]; b f32 [n_tokens, ne11, K]; ids int32 [n_tokens, n_used] (may be a
        strided view: the row stride is honoured) -> dst f32 [n_tokens, n_used, M]"""
        n_expert = w.shape[0]
        w = np.ascontiguousarray(w, np.uint8)
        b = np.ascontiguousarray(b, np.float32)
        n_tokens, ne11, _ = b.shape
        assert ids.dtype == np.int32 and ids.strides[1] == 4
        n_used = ids.shape[1]
        stride = ids.strides[0] // 4
        dst = np.empty((n_tokens, n_used, m), np.float32)
        rc = self.lib.qmo_mul_mat_id(t, _ptr(w), k, m, n_expert, _ptr(b), ne11, n_tokens,
                                     C.c_void_p(ids.ctypes.data), n_used, stride, _ptr(dst), act_mode)
        assert rc == 0, rc
        return dst


def _cpu_flags() -> set:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("flags"):
                return set(line.split(":")[1].split())
    except OSError:
        pass
    return set()


def ref_available() -> bool:
    return (HERE / "_ref" / "libggml-base.so").exists() and (HERE / "_ref" / "libggml-cpu.so").exists()


class RefGgml:
    """The real reference's CPU code through ctypes (survey §8c recipe).  ggml_cpu_init() must run first:
    ggml-base converts fp16 through a table that ggml_init fills (ggml-impl.h:498-504)."""

    def __init__(self, variant: str | None = None):
        ref = HERE / "_ref"
        if variant is None:
            need = {"avx512f", "avx512bw", "avx512vl", "avx512dq", "avx512cd", "avx512_vnni", "avx512_vbmi"}
            variant = "avx512" if need <= _cpu_flags() and (ref / "libggml-cpu-avx512.so").exists() else "avx2"
        self.variant = variant
        self.base = C.CDLL(str(ref / "libggml-base.so"), mode=C.RTLD_GLOBAL)
        self.cpu = C.CDLL(str(ref / ("libggml-cpu-avx512.so" if variant == "avx512" else "libggml-cpu.so")),
                          mode=C.RTLD_GLOBAL)
        self.cpu.ggml_cpu_init()
        self.base.ggml_row_size.restype = C.c_size_t
        self.base.ggml_row_size.argtypes = [C.c_int, C.c_int64]

    def row_size(self, t, k):
        return self.base.ggml_row_size(t, k)

    def quantize_weights(self, t, x: np.ndarray) -> np.ndarray:
        """quantize_row_<t>_ref over each row (ggml-quants.c); x f32 [rows,k] -> uint8 [rows,row_size]"""
        x = np.ascontiguousarray(x, np.float32)
        rows, k = x.shape
        out = np.zeros((rows, self.row_size(t, k)), np.uint8)
        fn = getattr(self.base, f"quantize_row_{TYPE_NAMES[t]}_ref")
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        for r in range(rows):
            fn(_ptr(x[r]), _ptr(out[r]), k)
        return out

    def dequantize(self, t, w: np.ndarray, k: int) -> np.ndarray:
        w = np.ascontiguousarray(w, np.uint8).reshape(-1, self.row_size(t, k))
        out = np.empty((w.shape[0], k), np.float32)
        fn = getattr(self.base, f"dequantize_row_{TYPE_NAMES[t]}")
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        for r in range(w.shape[0]):
            fn(_ptr(w[r]), _ptr(out[r]), k)
        return out

    def quantize_act(self, t, x: np.ndarray, impl: str = "cpu") -> np.ndarray:
        """impl='cpu': ggml-cpu's SIMD quantize_row_q8_0/q8_K; impl='ref': ggml-base's *_ref"""
        x = np.ascontiguousarray(x, np.float32)
        rows, k = x.shape
        vt = vec_dot_type(t)
        out = np.zeros((rows, self.row_size(vt, k)), np.uint8)
        if impl == "cpu":
            fn = getattr(self.cpu, f"quantize_row_{TYPE_NAMES[vt]}")
        else:
            fn = getattr(self.base, f"quantize_row_{TYPE_NAMES[vt]}_ref")
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        for r in range(rows):
            fn(_ptr(x[r]), _ptr(out[r]), k)
        return out

    def vec_dot(self, t, k, w_row: np.ndarray, a_row: np.ndarray) -> float:
        vt = TYPE_NAMES[vec_dot_type(t)]
        fn = getattr(self.cpu, f"ggml_vec_dot_{TYPE_NAMES[t]}_{vt}")
        fn.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
        out = C.c_float(0)
        fn(k, C.byref(out), 0, _ptr(np.ascontiguousarray(w_row)), 0, _ptr(np.ascontiguousarray(a_row)), 0, 1)
        return out.value

    def mul_mat(self, t, w: np.ndarray, k: int, x: np.ndarray) -> np.ndarray:
        """Row-by-row use of the reference's own quantizer + vec_dot (the non-llamafile path of
        ggml_compute_forward_mul_mat): dst f32 [N, M]"""
        w = np.ascontiguousarray(w, np.uint8).reshape(-1, self.row_size(t, k))
        acts = self.quantize_act(t, x)
        vt = TYPE_NAMES[vec_dot_type(t)]
        fn = getattr(self.cpu, f"ggml_vec_dot_{TYPE_NAMES[t]}_{vt}")
        fn.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
        n, m = acts.shape[0], w.shape[0]
        dst = np.empty((n, m), np.float32)
        out = C.c_float(0)
        for i in range(n):
            ap = _ptr(acts[i])
            for j in range(m):
                fn(k, C.byref(out), 0, _ptr(w[j]), 0, ap, 0, 1)
                dst[i, j] = out.value
        return dst

    # ---- the reference's real graph path (ggml_compute_forward_mul_mat / _mul_mat_id, multi-threaded,
    #      llamafile sgemm included), driven through the public C API
    class _InitParams(C.Structure):
        _fields_ = [("mem_size", C.c_size_t), ("mem_buffer", C.c_void_p), ("no_alloc", C.c_bool)]

    def _api(self):
        if getattr(self, "_api_ready", False):
            return
        b, c = self.base, self.cpu
        b.ggml_init.restype = C.c_void_p
        b.ggml_init.argtypes = [self._InitParams]
        b.ggml_free.argtypes = [C.c_void_p]
        for name, n in (("ggml_new_tensor_2d", 2), ("ggml_new_tensor_3d", 3)):
            f = getattr(b, name)
            f.restype = C.c_void_p
            f.argtypes = [C.c_void_p, C.c_int] + [C.c_int64] * n
        b.ggml_view_2d.restype = C.c_void_p
        b.ggml_view_2d.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_size_t, C.c_size_t]
        b.ggml_mul_mat.restype = C.c_void_p
        b.ggml_mul_mat.argtypes = [C.c_void_p] * 3
        b.ggml_mul_mat_id.restype = C.c_void_p
        b.ggml_mul_mat_id.argtypes = [C.c_void_p] * 4
        b.ggml_new_graph.restype = C.c_void_p
        b.ggml_new_graph.argtypes = [C.c_void_p]
        b.ggml_build_forward_expand.argtypes = [C.c_void_p, C.c_void_p]
        b.ggml_get_data.restype = C.c_void_p
        b.ggml_get_data.argtypes = [C.c_void_p]
        b.ggml_nbytes.restype = C.c_size_t
        b.ggml_nbytes.argtypes = [C.c_void_p]
        c.ggml_graph_compute_with_ctx.restype = C.c_int
        c.ggml_graph_compute_with_ctx.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        self._api_ready = True

    def _fill(self, tensor, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        assert self.base.ggml_nbytes(tensor) == arr.nbytes, (self.base.ggml_nbytes(tensor), arr.nbytes)
        C.memmove(self.base.ggml_get_data(tensor), arr.ctypes.data, arr.nbytes)

    def graph_mul_mat(self, t, w: np.ndarray, k: int, x: np.ndarray, n_threads: int = 1, repeat: int = 1):
        """dst f32 [N, M] through ggml_mul_mat + ggml_graph_compute_with_ctx.  Returns (dst, seconds/run)."""
        import time
        self._api()
        b = self.base
        w = np.ascontiguousarray(w, np.uint8).reshape(-1, self.row_size(t, k))
        x = np.ascontiguousarray(x, np.float32)
        m, n = w.shape[0], x.shape[0]
        mem = w.nbytes + x.nbytes + n * m * 4 + self.row_size(Q8_K, k) * n * 2 + (64 << 20)
        ctx = b.ggml_init(self._InitParams(mem, None, False))
        try:
            a = b.ggml_new_tensor_2d(ctx, t, k, m)
            bb = b.ggml_new_tensor_2d(ctx, 0, k, n)
            self._fill(a, w)
            self._fill(bb, x)
            d = b.ggml_mul_mat(ctx, a, bb)
            g = b.ggml_new_graph(ctx)
            b.ggml_build_forward_expand(g, d)
            st = self.cpu.ggml_graph_compute_with_ctx(ctx, g, n_threads)
            assert st == 0
            t0 = time.perf_counter()
            for _ in range(repeat):
                self.cpu.ggml_graph_compute_with_ctx(ctx, g, n_threads)
            dt = (time.perf_counter() - t0) / max(repeat, 1)
            out = np.empty((n, m), np.float32)
            C.memmove(out.ctypes.data, b.ggml_get_data(d), out.nbytes)
            return out, dt
        finally:
            b.ggml_free(ctx)

    def graph_mul_mat_id(self, t, w: np.ndarray, k: int, m: int, bsrc: np.ndarray, ids_full: np.ndarray,
                         n_used: int, n_threads: int = 1):
        """as [K,M,n_expert]; b [K, ne11, n_tokens]; ids_full int32 [n_tokens, ids_row] of which the first n_used
        columns are used (n_used < ids_row gives the strided-view form of test-backend-ops.cpp:2097-2102).
        Returns dst f32 [n_tokens, n_used, M]."""
        self._api()
        b = self.base
        w = np.ascontiguousarray(w, np.uint8)
        n_expert = w.shape[0]
        bsrc = np.ascontiguousarray(bsrc, np.float32)
        n_tokens, ne11, _ = bsrc.shape
        ids_full = np.ascontiguousarray(ids_full, np.int32)
        mem = w.nbytes + bsrc.nbytes * 3 + n_tokens * n_used * m * 4 + (64 << 20)
        ctx = b.ggml_init(self._InitParams(mem, None, False))
        try:
            a = b.ggml_new_tensor_3d(ctx, t, k, m, n_expert)
            bb = b.ggml_new_tensor_3d(ctx, 0, k, ne11, n_tokens)
            idt = b.ggml_new_tensor_2d(ctx, 26, ids_full.shape[1], n_tokens)
            self._fill(a, w)
            self._fill(bb, bsrc)
            self._fill(idt, ids_full)
            if n_used != ids_full.shape[1]:
                idt = b.ggml_view_2d(ctx, idt, n_used, n_tokens, ids_full.shape[1] * 4, 0)
            d = b.ggml_mul_mat_id(ctx, a, bb, idt)
            g = b.ggml_new_graph(ctx)
            b.ggml_build_forward_expand(g, d)
            st = self.cpu.ggml_graph_compute_with_ctx(ctx, g, n_threads)
            assert st == 0
            out = np.empty((n_tokens, n_used, m), np.float32)
            C.memmove(out.ctypes.data, b.ggml_get_data(d), out.nbytes)
            return out
        finally:
            b.ggml_free(ctx)
