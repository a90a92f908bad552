"""Drop-in boundary test: the REFERENCE's own parity harness (tests/test-backend-ops.cpp, built unmodified into
oracle/_ref by oracle/Makefile) loads our ggml backend module through GGML_BACKEND_PATH and compares every MUL_MAT /
MUL_MAT_ID case it supports with the ggml CPU backend (NMSE <= 5e-4, tests/test-backend-ops.cpp:1982-1984, 2075-2077)."""
import os
import re
import subprocess
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parents[1]
TBO = ROOT / "oracle" / "_ref" / "test-backend-ops"
PLUGIN = ROOT / "ggml-hexagon_amd" / "libggml-mi355x.so"


def run_tbo(*args, timeout=900):
    if not TBO.exists() or not PLUGIN.exists():
        pytest.skip("oracle/_ref/test-backend-ops or the plugin module is not built (needs the reference tree at build time)")
    env = dict(os.environ, GGML_BACKEND_PATH=str(PLUGIN))
    p = subprocess.run([str(TBO), *args], env=env, capture_output=True, text=True, timeout=timeout, cwd=str(TBO.parent))
    out = re.sub(r"\x1b\[[0-9;]*m", "", p.stdout + p.stderr)
    return p.returncode, out


def summarize(out):
    ok = len(re.findall(r"\): OK", out))
    fail = [l for l in out.splitlines() if "FAIL" in l or "ERR =" in l]
    unsup = len(re.findall(r"not supported \[", out))
    return ok, fail, unsup


@pytest.mark.parametrize("op,min_ok", [("MUL_MAT", 200), ("MUL_MAT_ID", 100)])
def test_reference_harness_passes(op, min_ok):
    rc, out = run_tbo("test", "-o", op)
    assert "MI355X0" in out, out[-2000:]
    ok, fail, unsup = summarize(out)
    print(f"{op}: {ok} OK, {len(fail)} failed, {unsup} not supported (by design: BF16 / other quant types, F16 src1, permuted views)")
    assert not fail, "\n".join(fail[:20])
    assert rc == 0, out[-3000:]
    assert ok >= min_ok, (ok, out[-2000:])


# the glue ops of a layer; default tolerance of the harness: NMSE <= 1e-7 (tests/test-backend-ops.cpp:325-327)
@pytest.mark.parametrize("op,min_ok", [("ADD", 20), ("MUL", 20), ("DIV", 20), ("SCALE", 1), ("SILU", 1), ("RMS_NORM", 6), ("NORM", 6), ("ARGSORT", 4), ("SUM_ROWS", 1), ("ROPE", 60),
                                       ("SOFT_MAX", 60), ("CPY", 20), ("CONT", 4), ("GET_ROWS", 20)])
def test_reference_harness_passes_glue_ops(op, min_ok):
    rc, out = run_tbo("test", "-o", op)
    ok, fail, unsup = summarize(out)
    print(f"{op}: {ok} OK, {len(fail)} failed, {unsup} not supported")
    assert not fail, "\n".join(fail[:20])
    assert rc == 0, out[-3000:]
    assert ok >= min_ok, (ok, out[-2000:])


def test_supported_surface_is_the_five_quant_types():
    rc, out = run_tbo("test", "-o", "MUL_MAT", "-p",
                      r"type_a=(q4_0|q8_0|q4_K|q5_K|q6_K),type_b=f32,m=16,n=[1-9],k=256,bs=\[1,1\],nr=\[1,1\],per=\[0,1,2,3\]")
    ok, fail, unsup = summarize(out)
    assert not fail and rc == 0
    assert ok >= 45 and unsup == 0, (ok, unsup, out[-1500:])


def test_supported_surface_includes_the_next_seven_formats():
    """Q4_1, Q5_0, Q5_1, Q2_K, Q3_K, IQ4_NL and (round 3) IQ4_XS through the reference's own harness (n = 1..9 covers the mat-vec kernels
    and the first MFMA batch size), MUL_MAT and MUL_MAT_ID"""
    rc, out = run_tbo("test", "-o", "MUL_MAT", "-p",
                      r"type_a=(q4_1|q5_0|q5_1|q2_K|q3_K|iq4_nl|iq4_xs),type_b=f32,m=16,n=[1-9],k=256,bs=\[1,1\],nr=\[1,1\],per=\[0,1,2,3\]")
    ok, fail, unsup = summarize(out)
    assert not fail and rc == 0, "\n".join(fail[:20])
    assert ok >= 63 and unsup == 0, (ok, unsup, out[-1500:])
    rc, out = run_tbo("test", "-o", "MUL_MAT_ID", "-p", r"type_a=(q4_1|q5_0|q5_1|q2_K|q3_K|iq4_nl|iq4_xs)")
    ok, fail, unsup = summarize(out)
    assert not fail and rc == 0, "\n".join(fail[:20])
    assert ok >= 14 and unsup == 0, (ok, unsup, out[-1500:])
