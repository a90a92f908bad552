"""Row split inside the plugin: tests/cpp/test_split_buffer.cpp, linked against the reference's own ggml
libraries (oracle/_ref), asks the module for "ggml_backend_split_buffer_type" the way llama.cpp does with -sm row, puts
weights in it and runs MUL_MAT on the root device; results are compared with the ggml CPU backend (NMSE <= 5e-4) and
with the unsplit product on one device (bit for bit at N <= 8).  A one-GPU box has one device, so the module is told to
register several logical devices over it (GGML_MI355X_VIRTUAL_DEVICES): every slice allocation, cross-"device" copy,
event wait and 2-D gather of the real multi-GPU path runs, only the fabric under the copies differs."""
import os
import re
import subprocess
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parents[1]
EXE = ROOT / "oracle" / "_ref" / "test-split-buffer"
PLUGIN = ROOT / "ggml-hexagon_amd" / "libggml-mi355x.so"


@pytest.mark.parametrize("n_virtual", [2, 3])
def test_row_split_matches_cpu_and_unsplit(n_virtual):
    if not EXE.exists() or not PLUGIN.exists():
        pytest.skip("oracle/_ref/test-split-buffer or the plugin module is not built (needs the reference tree at build time)")
    env = dict(os.environ, GGML_BACKEND_PATH=str(PLUGIN), GGML_MI355X_VIRTUAL_DEVICES=str(n_virtual))
    p = subprocess.run([str(EXE)], env=env, capture_output=True, text=True, timeout=900, cwd=str(EXE.parent))
    out = p.stdout + p.stderr
    assert f"MI355X devices: {n_virtual}" in out, out[-2000:]
    fails = [l for l in out.splitlines() if l.rstrip().endswith("FAIL")]
    assert not fails, "\n".join(fails[:20])
    m = re.search(r"(\d+) OK, (\d+) FAILED", out)
    assert p.returncode == 0 and m and int(m.group(2)) == 0 and int(m.group(1)) >= 200, out[-3000:]


def test_async_transfers_and_events():
    """set/get/cpy_tensor_async, event record / wait / synchronize and caps.async/events, driven through
    ggml's public API across two logical devices (tests/cpp/test_async_events.cpp)"""
    exe = ROOT / "oracle" / "_ref" / "test-async-events"
    if not exe.exists() or not PLUGIN.exists():
        pytest.skip("oracle/_ref/test-async-events or the plugin module is not built (needs the reference tree at build time)")
    env = dict(os.environ, GGML_BACKEND_PATH=str(PLUGIN), GGML_MI355X_VIRTUAL_DEVICES="2")
    p = subprocess.run([str(exe)], env=env, capture_output=True, text=True, timeout=600, cwd=str(exe.parent))
    out = p.stdout + p.stderr
    m = re.search(r"(\d+) OK, (\d+) FAILED", out)
    assert p.returncode == 0 and m and int(m.group(2)) == 0 and int(m.group(1)) >= 9, out[-3000:]


def test_moe_twin_nodes_go_out_as_one_call():
    """two MUL_MAT_ID nodes on the same src1 and ids in one graph (ffn_up_exps / ffn_gate_exps): the plugin pairs them
    (qmm_mul_mat_id_pair); both results against the CPU backend (tests/cpp/test_moe_pair.cpp)"""
    exe = ROOT / "oracle" / "_ref" / "test-moe-pair"
    if not exe.exists() or not PLUGIN.exists():
        pytest.skip("oracle/_ref/test-moe-pair or the plugin module is not built (needs the reference tree at build time)")
    env = dict(os.environ, GGML_BACKEND_PATH=str(PLUGIN))
    p = subprocess.run([str(exe)], env=env, capture_output=True, text=True, timeout=600, cwd=str(exe.parent))
    out = p.stdout + p.stderr
    m = re.search(r"(\d+) OK, (\d+) FAILED", out)
    assert p.returncode == 0 and m and int(m.group(2)) == 0 and int(m.group(1)) >= 18, out[-3000:]


def test_planar_weights_through_the_plugin():
    """SURVEY 8f-2 (tests/cpp/test_planar_weights.cpp): lazily repacked Q4_0 / Q8_0 / Q6_K weights behave like wire tensors for
    get_tensor / partial set_tensor / tensor copies, match the CPU backend, and give the bits of the un-repacked build"""
    exe = ROOT / "oracle" / "_ref" / "test-planar-weights"
    if not exe.exists() or not PLUGIN.exists():
        pytest.skip("oracle/_ref/test-planar-weights or the plugin module is not built (needs the reference tree at build time)")
    sums = {}
    for repack in ("1", "0"):
        env = dict(os.environ, GGML_BACKEND_PATH=str(PLUGIN), GGML_MI355X_REPACK=repack)
        p = subprocess.run([str(exe)], env=env, capture_output=True, text=True, timeout=600, cwd=str(exe.parent))
        out = p.stdout + p.stderr
        fails = [l for l in out.splitlines() if l.rstrip().endswith("FAIL")]
        assert not fails, "\n".join(fails[:20])
        m = re.search(r"(\d+) OK, (\d+) FAILED", out)
        assert p.returncode == 0 and m and int(m.group(2)) == 0 and int(m.group(1)) == 40, out[-3000:]
        sums[repack] = re.search(r"checksum ([0-9a-f]+)", out).group(1)
    assert sums["1"] == sums["0"], sums


def test_graph_scheduler_on_random_transformer_graphs():
    """tests/cpp/test_graph_fuzz.cpp: 60 random llama / qwen / gemma / gpt-neox style block graphs, allocated by ggml's own graph
    allocator, fused schedule vs one launch per node vs the CPU backend"""
    exe = ROOT / "oracle" / "_ref" / "test-graph-fuzz"
    if not exe.exists() or not PLUGIN.exists():
        pytest.skip("oracle/_ref/test-graph-fuzz or the plugin module is not built (needs the reference tree at build time)")
    env = dict(os.environ, GGML_BACKEND_PATH=str(PLUGIN))
    p = subprocess.run([str(exe), "60"], env=env, capture_output=True, text=True, timeout=900, cwd=str(exe.parent))
    out = p.stdout + p.stderr
    print(out[-1500:])
    fails = [l for l in out.splitlines() if l.rstrip().endswith("FAIL")]
    assert not fails, "\n".join(fails[:20])
    m = re.search(r"(\d+) OK, (\d+) FAILED", out)
    assert p.returncode == 0 and m and int(m.group(2)) == 0 and int(m.group(1)) == 60, out[-3000:]


def test_fusions_survive_forced_buffer_aliasing():
    """tests/cpp/test_fusion_alias.cpp: a later node's buffer placed on an operand the fused launch still reads"""
    exe = ROOT / "oracle" / "_ref" / "test-fusion-alias"
    if not exe.exists() or not PLUGIN.exists():
        pytest.skip("oracle/_ref/test-fusion-alias or the plugin module is not built (needs the reference tree at build time)")
    env = dict(os.environ, GGML_BACKEND_PATH=str(PLUGIN))
    p = subprocess.run([str(exe)], env=env, capture_output=True, text=True, timeout=600, cwd=str(exe.parent))
    out = p.stdout + p.stderr
    print(out[-2500:])
    fails = [l for l in out.splitlines() if l.rstrip().endswith("FAIL")]
    assert not fails, "\n".join(fails[:20])
    m = re.search(r"(\d+) OK, (\d+) FAILED", out)
    assert p.returncode == 0 and m and int(m.group(2)) == 0 and int(m.group(1)) == 61, out[-3000:]
    # the last case's weights exceed the f16 range: the module says so once and re-issues the graph in bf16
    assert "exceeds the f16 range" in out, out[-1500:]


def test_readers_in_another_scheduler_split_get_real_data():
    """tests/cpp/test_cross_split.cpp: an op the device refuses sits inside a layer and reads a tensor a multi-node launch would have
    skipped or sent to the scratch; through ggml_backend_sched over { MI355X, CPU } against the CPU alone"""
    exe = ROOT / "oracle" / "_ref" / "test-cross-split"
    if not exe.exists() or not PLUGIN.exists():
        pytest.skip("oracle/_ref/test-cross-split or the plugin module is not built (needs the reference tree at build time)")
    p = subprocess.run([str(exe)], env=dict(os.environ, GGML_BACKEND_PATH=str(PLUGIN)), capture_output=True, text=True, timeout=600, cwd=str(exe.parent))
    out = p.stdout + p.stderr
    import re
    m = re.search(r"(\d+) OK, (\d+) FAILED", out)
    assert p.returncode == 0 and m and int(m.group(2)) == 0 and int(m.group(1)) == 12, out[-3000:]
