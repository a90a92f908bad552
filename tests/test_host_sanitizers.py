"""The plugin's host logic (csrc/ggml-mi355x.cpp: graph analysis, fusion matching with its aliasing checks, hoists and scratch
redirects, planar-weight bookkeeping, the row-split buffer type) under AddressSanitizer + UBSan, in the CPU-only container
(the reference's own sanitizer builds are CPU-only too, CMakeLists.txt:81-83).  The module is compiled
with -fsanitize=address,undefined against tests/cpp/qmm_stub.cpp, a no-device stand-in of the C-ABI (host memory, no arithmetic),
and driven by the same programs the GPU tests use: the graph fuzzer (both schedules of 40 random transformer graphs), the
planar-weights program (set / get / partial set / copy on repacked tensors) and the reference's libllama on a tiny GGUF (one device,
two logical devices with -sm layer and -sm row).  Results are garbage by construction; what is asserted is that no sanitizer fires."""
import os
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
REF = ROOT / "oracle" / "_ref"
MOD = REF / "asan" / "libggml-mi355x.so"


def _libs():
    out = []
    for n in ("libasan.so", "libubsan.so"):
        p = subprocess.run(["gcc", f"-print-file-name={n}"], capture_output=True, text=True).stdout.strip()
        if not p or not os.path.isabs(p):
            pytest.skip(f"{n} not found")
        out.append(p)
    return " ".join(out)


@pytest.fixture(scope="module")
def env():
    if not (Path("/root/reference/ggml/include/ggml-backend.h").exists() or MOD.exists()):
        pytest.skip("needs the reference tree (ggml headers) to build the sanitizer module")
    if Path("/root/reference/ggml/include/ggml-backend.h").exists():
        subprocess.run(["make", "-C", str(ROOT / "oracle"), "-j8", "ref"], check=True, capture_output=True)
        subprocess.run(["make", "-C", str(ROOT / "oracle"), "asan"], check=True, capture_output=True)
    return dict(os.environ, LD_PRELOAD=_libs(), ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1",
                GGML_BACKEND_PATH=str(MOD), QMM_FUZZ_PLAN_ONLY="1")


def clean(out):
    bad = [l for l in out.splitlines() if "AddressSanitizer" in l or "runtime error:" in l or "LeakSanitizer" in l]
    assert not bad, "\n".join(bad[:10]) + "\n" + out[-3000:]


def test_graph_fuzzer_planning_is_sanitizer_clean(env):
    p = subprocess.run([str(REF / "test-graph-fuzz"), "40"], env=env, capture_output=True, text=True, timeout=600, cwd=str(REF))
    clean(p.stdout + p.stderr)
    assert p.returncode == 0 and "40 OK, 0 FAILED" in p.stdout, (p.stdout + p.stderr)[-2000:]


def test_cross_split_planning_is_sanitizer_clean(env):
    p = subprocess.run([str(REF / "test-cross-split")], env=env, capture_output=True, text=True, timeout=600, cwd=str(REF))
    clean(p.stdout + p.stderr)                      # (numbers cannot match on the stub; the scheduler must have made its splits)
    assert "splits" in p.stdout and "FAILED" in p.stdout


def test_planar_weight_bookkeeping_is_sanitizer_clean(env):
    p = subprocess.run([str(REF / "test-planar-weights")], env=env, capture_output=True, text=True, timeout=600, cwd=str(REF))
    clean(p.stdout + p.stderr)                      # its numeric checks cannot pass on the stub; only the sanitizers are asserted
    assert "checksum" in p.stdout


@pytest.mark.parametrize("mode", ["one-device", "layer-split", "row-split"])
def test_libllama_graphs_are_sanitizer_clean(env, mode, tmp_path):
    exe = REF / "llama-e2e"
    gguf = tmp_path / "tiny.gguf"
    plain = {k: v for k, v in env.items() if k not in ("LD_PRELOAD", "GGML_BACKEND_PATH")}
    subprocess.run([str(exe), "write", "--config", "tiny-mix" if mode == "one-device" else "tiny-q4_k_m", "--gguf", str(gguf)], check=True,
                   capture_output=True, env=plain, cwd=str(REF), timeout=300)
    e = dict(env)
    args = [str(exe), "bench", "--gguf", str(gguf), "--ngl", "99", "-p", "24", "-n", "3", "-r", "1", "-t", "2"]
    if mode != "one-device":
        e["GGML_MI355X_VIRTUAL_DEVICES"] = "2"
        args += ["-sm", "layer" if mode == "layer-split" else "row"]
    p = subprocess.run(args, env=e, capture_output=True, text=True, timeout=600, cwd=str(REF))
    clean(p.stdout + p.stderr)
    assert p.returncode == 0, (p.stdout + p.stderr)[-2000:]
