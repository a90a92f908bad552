"""Synthetic quantized weights in GGUF wire layout (rows of blocks), without running a quantizer.

The hot path never quantizes weights (llama-quantize does that offline), so the bench and the
full-size tests fill weight tensors with random *valid* blocks: random quant bytes and packed
scales, fp16 super-scales drawn so that the dequantized values look like N(0, ~0.02) model
weights (both signs of ``d`` occur, as the real quantizers produce).  Layouts: SURVEY.md Appendix A
(ggml/src/ggml-common.h:167-172, 209-214, 285-334).
"""
from __future__ import annotations

import numpy as np

Q4_0, Q8_0, Q4_K, Q5_K, Q6_K = 2, 8, 12, 13, 14
Q4_1, Q5_0, Q5_1, Q2_K, Q3_K, IQ4_NL, IQ4_XS = 3, 6, 7, 10, 11, 20, 23      # ggml-common.h:174-207, 253-277, 405-418
BLCK = {Q4_0: 32, Q8_0: 32, Q4_K: 256, Q5_K: 256, Q6_K: 256, Q4_1: 32, Q5_0: 32, Q5_1: 32, Q2_K: 256, Q3_K: 256, IQ4_NL: 32, IQ4_XS: 256}
TYPE_SIZE = {Q4_0: 18, Q8_0: 34, Q4_K: 144, Q5_K: 176, Q6_K: 210, Q4_1: 20, Q5_0: 22, Q5_1: 24, Q2_K: 84, Q3_K: 110, IQ4_NL: 18, IQ4_XS: 136}
NAMES = {Q4_0: "q4_0", Q8_0: "q8_0", Q4_K: "q4_K", Q5_K: "q5_K", Q6_K: "q6_K",
         Q4_1: "q4_1", Q5_0: "q5_0", Q5_1: "q5_1", Q2_K: "q2_K", Q3_K: "q3_K", IQ4_NL: "iq4_nl", IQ4_XS: "iq4_xs"}
BY_NAME = {v: k for k, v in NAMES.items()}


def row_size(t: int, k: int) -> int:
    assert k % BLCK[t] == 0
    return k // BLCK[t] * TYPE_SIZE[t]


def _f16_bytes(rng, n, scale, signed=True):
    v = rng.uniform(0.5, 1.5, n).astype(np.float32) * scale
    if signed:
        v *= rng.choice(np.array([-1.0, 1.0], np.float32), n)
    return v.astype(np.float16).view(np.uint8).reshape(n, 2)


def synth_weights(t: int, rows: int, k: int, seed: int = 0, sigma: float = 0.02) -> np.ndarray:
    """uint8 [rows, row_size(t,k)] of random valid blocks."""
    rng = np.random.default_rng(seed)
    nb = rows * (k // BLCK[t])
    blk = rng.integers(0, 256, (nb, TYPE_SIZE[t]), dtype=np.uint8)
    if t == Q4_0:
        blk[:, 0:2] = _f16_bytes(rng, nb, 3 * sigma / 8)
    elif t == Q8_0:
        blk[:, 0:2] = _f16_bytes(rng, nb, 3 * sigma / 127)
    elif t in (Q4_K, Q5_K):
        qmax = 15 if t == Q4_K else 31
        blk[:, 0:2] = _f16_bytes(rng, nb, 6 * sigma / qmax / 40, signed=False)   # d   (sub-scale ~ d*sc, sc<=63)
        blk[:, 2:4] = _f16_bytes(rng, nb, 3 * sigma / 40, signed=False)          # dmin (offset ~ dmin*m)
    elif t == Q6_K:
        blk[:, 208:210] = _f16_bytes(rng, nb, 3 * sigma / 32 / 80)
    elif t in (Q4_1, Q5_1):                                   # w = q*d + m, q in [0, 15 | 31]
        blk[:, 0:2] = _f16_bytes(rng, nb, 6 * sigma / (15 if t == Q4_1 else 31), signed=False)
        blk[:, 2:4] = (-_f16_val(rng, nb, 3 * sigma)).astype(np.float16).view(np.uint8).reshape(nb, 2)
    elif t == Q5_0:
        blk[:, 0:2] = _f16_bytes(rng, nb, 3 * sigma / 16)
    elif t == IQ4_NL:
        blk[:, 0:2] = _f16_bytes(rng, nb, 3 * sigma / 127)
    elif t == IQ4_XS:                                         # w = d*(ls-32)*kvalues[q], |ls-32| <= 32, |kvalues| <= 127
        blk[:, 0:2] = _f16_bytes(rng, nb, 3 * sigma / 127 / 16)
    elif t == Q2_K:                                           # w = d*(sc&15)*q - dmin*(sc>>4), q <= 3
        blk[:, 80:82] = _f16_bytes(rng, nb, 6 * sigma / 3 / 10, signed=False)
        blk[:, 82:84] = _f16_bytes(rng, nb, 3 * sigma / 10, signed=False)
    elif t == Q3_K:                                           # w = d*(sc-32)*(q-4 .. q)
        blk[:, 108:110] = _f16_bytes(rng, nb, 3 * sigma / 4 / 20)
    else:
        raise ValueError(t)
    return blk.reshape(rows, -1)


def _f16_val(rng, n, scale):
    return rng.uniform(0.5, 1.5, n).astype(np.float32) * scale


def synth_weights_torch(t: int, rows: int, k: int, device, seed: int = 0, sigma: float = 0.02):
    """same construction as synth_weights, generated directly in HBM (bench-sized tensors)"""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    nb = rows * (k // BLCK[t])
    blk = torch.randint(0, 256, (nb, TYPE_SIZE[t]), dtype=torch.uint8, device=device, generator=g)

    def f16(scale, signed=True):
        v = (torch.rand(nb, device=device, generator=g) + 0.5) * scale
        if signed:
            v = v * (torch.randint(0, 2, (nb,), device=device, generator=g).float() * 2 - 1)
        return v.to(torch.float16).view(torch.uint8).reshape(nb, 2)

    if t == Q4_0:
        blk[:, 0:2] = f16(3 * sigma / 8)
    elif t == Q8_0:
        blk[:, 0:2] = f16(3 * sigma / 127)
    elif t in (Q4_K, Q5_K):
        qmax = 15 if t == Q4_K else 31
        blk[:, 0:2] = f16(6 * sigma / qmax / 40, signed=False)
        blk[:, 2:4] = f16(3 * sigma / 40, signed=False)
    elif t == Q6_K:
        blk[:, 208:210] = f16(3 * sigma / 32 / 80)
    elif t in (Q4_1, Q5_1):
        blk[:, 0:2] = f16(6 * sigma / (15 if t == Q4_1 else 31), signed=False)
        blk[:, 2:4] = (-(torch.rand(nb, device=device, generator=g) + 0.5) * 3 * sigma).to(torch.float16).view(torch.uint8).reshape(nb, 2)
    elif t == Q5_0:
        blk[:, 0:2] = f16(3 * sigma / 16)
    elif t == IQ4_NL:
        blk[:, 0:2] = f16(3 * sigma / 127)
    elif t == IQ4_XS:
        blk[:, 0:2] = f16(3 * sigma / 127 / 16)
    elif t == Q2_K:
        blk[:, 80:82] = f16(6 * sigma / 3 / 10, signed=False)
        blk[:, 82:84] = f16(3 * sigma / 10, signed=False)
    elif t == Q3_K:
        blk[:, 108:110] = f16(3 * sigma / 4 / 20)
    else:
        raise ValueError(t)
    return blk.reshape(rows, -1)
