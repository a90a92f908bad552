#!/usr/bin/env python3
"""Generate tests/golden/golden_<type>.npz from the REAL reference.

Run in the build container only (needs oracle/_ref, i.e. /root/reference compiled by oracle/Makefile):

    make -C oracle ref && python tests/golden/make_golden.py

The reference holds no stored golden files for quant math (all its checks are
computed at test time), so these vectors are *outputs of the reference itself run here*:
  * ggml-base  quantize_row_<t>_ref / dequantize_row_<t>                       (ggml/src/ggml-quants.c)
  * ggml-cpu   quantize_row_q8_0 / quantize_row_q8_K (AVX2 build) and the *_ref forms
  * ggml-cpu   ggml_mul_mat / ggml_mul_mat_id graphs via ggml_graph_compute_with_ctx
               (ggml_compute_forward_mul_mat / _mul_mat_id, ggml/src/ggml-cpu/ggml-cpu.c:6745-7197),
               AVX2 variant, 2 threads
Fixtures are data only (inputs + expected outputs).
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle.pyoracle import Q2_K, Q3_K, Q4_0, Q4_1, Q4_K, Q5_0, Q5_1, Q5_K, Q6_K, Q8_0, IQ4_NL, IQ4_XS, TYPE_NAMES, WEIGHT_TYPES, RefGgml  # noqa: E402
import ggml_hexagon_amd.synth as synth  # noqa: E402

K, M = 512, 40
OUT = Path(__file__).resolve().parent


def edge_blocks(t, rng):
    """one row (K weights) of hand-made edge blocks: zero bytes, all-ones bytes, d<0, d subnormal, d=0"""
    ts, bs = synth.TYPE_SIZE[t], synth.BLCK[t]
    nb = K // bs
    blk = rng.integers(0, 256, (nb, ts), dtype=np.uint8)
    doff = {Q6_K: 208, Q2_K: 80, Q3_K: 108}.get(t, 0)
    two = {Q4_K: 2, Q5_K: 2, Q4_1: 2, Q5_1: 2, Q2_K: 82}          # offset of the second fp16 (dmin / m), where there is one

    def set_d(i, h, off=doff):
        blk[i, off:off + 2] = np.array([h], np.uint16).view(np.uint8)

    for i in range(nb):
        set_d(i, 0x2A00 + 37 * i)                  # ~0.047: sane
        if t in two:
            set_d(i, 0x2C00 + 11 * i, two[t])
    blk[0, :] = 0                                   # all-zero block
    blk[1 % nb, :] = 0xFF                           # every field at max (d = NaN pattern replaced below)
    set_d(1 % nb, 0xABCD)                           # negative d
    if t in two:
        set_d(1 % nb, 0x3555, two[t])
    if nb > 2:
        set_d(2, 0x0001)                            # smallest fp16 subnormal
        if t in two:
            set_d(2, 0x03FF, two[t])                # largest subnormal dmin / m
    if nb > 3:
        set_d(3, 0x0000)                            # d == 0
    if nb > 4:
        set_d(4, 0x7BFF)                            # fp16 max (65504)
    return blk.reshape(1, -1)


def act_rows(rng):
    x = rng.uniform(-1, 1, (8, K)).astype(np.float32)
    x[1, :256] = 0.0                                # an all-zero Q8_K block / eight zero Q8_0 blocks
    x[2] *= 1000.0
    x[3] *= 1e-20
    x[4] = np.round(x[4] * 4) / 4                   # many exact ties after scaling
    x[4, 0] = 127.0 / 8
    x[5, ::2] = 0.0
    x[6] = rng.standard_normal(K).astype(np.float32) * 3
    x[7] = -np.abs(x[7])                            # negative peak
    return x


def main():
    r = RefGgml("avx2")
    only = [a for a in sys.argv[1:]]
    for t in WEIGHT_TYPES:
        if only and TYPE_NAMES[t] not in only:
            continue
        rng = np.random.default_rng(1000 + t)
        w = np.concatenate([
            r.quantize_weights(t, rng.uniform(-1, 1, (M - 9, K)).astype(np.float32)),
            r.quantize_weights(t, (rng.standard_normal((4, K)) * 0.02).astype(np.float32)),
            synth.synth_weights(t, 4, K, seed=t),
            edge_blocks(t, rng),
        ])
        assert w.shape[0] == M
        deq = r.dequantize(t, w, K)
        x = act_rows(rng)
        out = dict(type=np.int32(t), K=np.int32(K), w=w, deq_bits=deq.view(np.uint32), act=x,
                   act_q_cpu=r.quantize_act(t, x, "cpu"), act_q_ref=r.quantize_act(t, x, "ref"))
        # the edge row holds inf/NaN-free but extreme scales; keep it out of the matmul fixtures
        wm = w[:M - 1]
        for n in (1, 5):
            out[f"dst_n{n}"], _ = r.graph_mul_mat(t, wm, K, x[:n], n_threads=2)
        n_expert, n_used, n_tok = 4, 2, 6
        we = np.stack([r.quantize_weights(t, rng.uniform(-1, 1, (16, K)).astype(np.float32)) for _ in range(n_expert)])
        ids_full = np.stack([rng.permutation(n_expert) for _ in range(n_tok)]).astype(np.int32)
        out["id_w"], out["id_ids_full"], out["id_n_used"] = we, ids_full, np.int32(n_used)
        for ne11 in (1, n_used):
            b = rng.uniform(-1, 1, (n_tok, ne11, K)).astype(np.float32)
            out[f"id_b_ne11_{ne11}"] = b
            out[f"id_dst_ne11_{ne11}"] = r.graph_mul_mat_id(t, we, K, 16, b, ids_full, n_used, n_threads=2)
        np.savez_compressed(OUT / f"golden_{TYPE_NAMES[t]}.npz", **out)
        print("wrote", TYPE_NAMES[t], {k: getattr(v, "shape", None) for k, v in out.items()})


if __name__ == "__main__":
    main()
