"""Import shim: the package directory is ``ggml-hexagon_amd/`` (a hyphen cannot be imported), so
``import ggml_hexagon_amd`` loads that directory under this name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ggml-hexagon_amd")
_spec = importlib.util.spec_from_file_location(
    "ggml_hexagon_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["ggml_hexagon_amd"] = _mod
_spec.loader.exec_module(_mod)
