"""shared test-data builders"""
import numpy as np


def q8_K_tie_rows(n_rows: int, k: int, seed: int = 0) -> np.ndarray:
    """Activation rows built to sit on the double-rounding edge of quantize_row_q8_K_ref
    (ggml/src/ggml-quants.c:2479-2516): elements x for which  f32(iscale*x)  is EXACTLY a half-integer
    while the unrounded product is not.  The reference rounds the product to f32 first and then adds the
    12582912.f magic (nearest_int, ggml-quants.c:372-377), so these land on ties-to-even; a fused
    multiply-add implementation gets a different int8 for about half of them."""
    f32 = np.float32
    rng = np.random.default_rng(seed)
    out = rng.uniform(-0.2, 0.2, (n_rows, k)).astype(f32)
    for r in range(n_rows):
        for b in range(k // 256):
            blk = out[r, b * 256:(b + 1) * 256]
            m = f32(rng.uniform(0.9, 1.0) * rng.choice([-1.0, 1.0]))
            blk[0] = m
            isc = f32(-127.0) / m
            pos = 1
            for h in rng.permutation(np.arange(-60, 60)):
                target = f32(h + 0.5)
                x0 = f32(target / isc)
                cand = x0
                for _ in range(200):
                    if f32(isc * cand) == target and float(isc) * float(cand) != float(target):
                        blk[pos] = cand
                        pos += 1
                        break
                    cand = np.nextafter(cand, f32(np.inf) if rng.random() < 0.5 else f32(-np.inf))
                if pos >= 200:
                    break
    return out
