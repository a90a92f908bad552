"""hipcc build recipes (gfx950 only, in-tree outputs so the .so files travel with gpurun snapshots)."""
from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
CSRC = PKG / "csrc"
QMM_SO = PKG / "libggml_mi355x_qmm.so"
PLUGIN_SO = PKG / "libggml-mi355x.so"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
GGML_SRC = Path(os.environ.get("GGML_SRC_DIR", "/root/reference/ggml"))


def _newer(target: Path, sources) -> bool:
    if not target.exists():
        return False
    t = target.stat().st_mtime
    return all(Path(s).stat().st_mtime <= t for s in sources)


def build_qmm(force: bool = False) -> Path:
    """the kernel library behind include/ggml_mi355x_qmm.h and include/ggml_mi355x_ops.h: one object per translation unit
    (qmm_api.hip = the quantized MUL_MAT path, qmm_ops.hip = the glue ops, qmm_comm.hip = the RCCL exchange of a one-process row split), rebuilt only when its sources changed"""
    headers = sorted(CSRC.glob("qmm_*.hiph")) + sorted(CSRC.glob("qmm_*.h")) + sorted((ROOT / "include").glob("ggml_mi355x_*.h"))
    units = [CSRC / "qmm_api.hip", CSRC / "qmm_ops.hip", CSRC / "qmm_comm.hip"]
    if not force and _newer(QMM_SO, units + headers):
        return QMM_SO
    if not shutil.which(HIPCC):
        # a library older than its sources is never used silently: a stale kernel must not be tested or benchmarked
        raise RuntimeError("hipcc not found and libggml_mi355x_qmm.so is " + ("older than its sources" if QMM_SO.exists() else "missing"))
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-fno-slp-vectorize"]
    objs, jobs = [], []
    for u in units:
        obj = CSRC / (u.stem + ".o")
        objs.append(obj)
        if force or not _newer(obj, [u] + headers):
            jobs.append(subprocess.Popen([HIPCC, *flags, "-c", str(u), "-o", str(obj)]))
    for j in jobs:
        if j.wait() != 0:
            raise subprocess.CalledProcessError(j.returncode, j.args)
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(QMM_SO), *map(str, objs), "-ldl"], check=True)
    return QMM_SO


def have_ggml_headers() -> bool:
    return (GGML_SRC / "include" / "ggml-backend.h").exists() and (GGML_SRC / "src" / "ggml-backend-impl.h").exists()


def build_plugin(force: bool = False) -> Path | None:
    """the ggml backend plugin (GGML_BACKEND_DL module).  It is compiled against the ggml headers of the
    llama.cpp tree it will be loaded into (here: the reference tree, in place); when that tree is not
    present the prebuilt module, if any, is kept."""
    src = CSRC / "ggml-mi355x.cpp"
    if not src.exists():
        return None
    srcs = [src, ROOT / "include" / "ggml-mi355x.h", ROOT / "include" / "ggml_mi355x_qmm.h", ROOT / "include" / "ggml_mi355x_ops.h"]
    if not force and _newer(PLUGIN_SO, srcs + [QMM_SO]):
        return PLUGIN_SO
    if not have_ggml_headers() or not (shutil.which("g++") or shutil.which("c++")):
        return PLUGIN_SO if PLUGIN_SO.exists() else None
    cxx = shutil.which("g++") or shutil.which("c++")
    cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-DGGML_BACKEND_DL", "-DGGML_BACKEND_BUILD",
           "-DGGML_BACKEND_SHARED", "-DGGML_SHARED",
           f"-I{GGML_SRC / 'include'}", f"-I{GGML_SRC / 'src'}", f"-I{ROOT / 'include'}",
           str(src), "-o", str(PLUGIN_SO), f"-L{PKG}", "-lggml_mi355x_qmm", "-Wl,-rpath,$ORIGIN"]
    subprocess.run(cmd, check=True)
    return PLUGIN_SO


def build_all(force: bool = False):
    return build_qmm(force), build_plugin(force)


if __name__ == "__main__":
    print(build_all(force=True))
