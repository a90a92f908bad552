"""ggml-mi355x: an MI355X-native implementation of the quantized MUL_MAT / MUL_MAT_ID hot path of ggml.

Layout
  csrc/          hand-written HIP kernels for gfx950 + the C-ABI (include/ggml_mi355x_qmm.h) + the ggml
                 backend plugin (include/ggml-mi355x.h) that unmodified llama.cpp loads via GGML_BACKEND_PATH
  capi.py        ctypes binding of the C-ABI (device pointers in, device pointers out)
  synth.py       synthetic quantized weights in GGUF wire layout
  rowsplit.py    ggml row-split partitioning + the RCCL concat step for one-process-per-GPU runs
  build.py       hipcc build recipes

There is no CPU fallback anywhere in this package: without the HIP library every entry point raises.
"""
from . import synth  # noqa: F401
