"""Loader of gaa_host.so, the COMPILED host side of the package's autograd nodes (csrc/gaa_host.cpp): one native call per node forward, the
backward on autograd's device thread without the interpreter.  Built in-tree by csrc/build_host.py (g++ against this interpreter's torch).

`get()` returns the module, or None when the Python twins were asked for (`GAA_NATIVE_HOST=0`, `set_enabled(False)`: rasterizer._RasterizeBound,
binding._MeshFramesTimestep, loss._L1 / _L1Ssim -- the same launches through ctypes, kept as the reference of the host logic and for everything
outside the product default).  A MISSING gaa_host.so is an error, not a silent switch to the slower host: build it (`__graft_entry__.build()`)
or opt out explicitly."""
from __future__ import annotations

import importlib.util
import os

from . import _lib

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST_LIB_PATH = os.path.join(_HERE, "gaa_host.so")
_enabled = os.environ.get("GAA_NATIVE_HOST", "1") != "0"
_mod = None


def set_enabled(flag: bool) -> bool:
    """Process-wide switch between the compiled host (True) and the Python twins (False); returns the previous value."""
    global _enabled
    prev, _enabled = _enabled, bool(flag)
    return prev


def enabled() -> bool:
    return _enabled


def load():
    global _mod
    if _mod is None:
        if not os.path.exists(HOST_LIB_PATH):
            raise RuntimeError(f"{HOST_LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` (g++ against torch's headers), "
                               "or set GAA_NATIVE_HOST=0 to run the Python host side")
        import torch  # noqa: F401  (the module links against torch's libraries)

        _lib.gsr(), _lib.gab(), _lib.gls()      # mapped (after torch) and ABI-checked before the module resolves their symbols
        def _import():
            spec = importlib.util.spec_from_file_location("gaussianavatars_amd.gaa_host", HOST_LIB_PATH)
            m = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(m)
            return m

        try:
            mod = _import()
        except ImportError as e:
            # a module built against another torch / interpreter (an upgrade, another venv): undefined symbols at import.  One rebuild against
            # THIS interpreter, then an error that says what to do -- never a silent switch to the Python host side.
            import subprocess
            import sys

            r = subprocess.run([sys.executable, os.path.join(_HERE, "csrc", "build_host.py"), "--force"], capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"{HOST_LIB_PATH} does not import under this interpreter ({e}) and rebuilding it failed:\n{(r.stdout + r.stderr)[-1500:]}\n"
                                   "build it with `python gaussianavatars_amd/csrc/build_host.py --force`, or set GAA_NATIVE_HOST=0 to run the Python host side") from e
            mod = _import()
        if (mod.GSR_ABI, mod.GAB_ABI, mod.GLS_ABI) != (_lib.GSR_ABI_VERSION, _lib.GAB_ABI_VERSION, _lib.GLS_ABI_VERSION):
            raise RuntimeError(f"gaa_host.so was built for ABI {(mod.GSR_ABI, mod.GAB_ABI, mod.GLS_ABI)}: rebuild it (csrc/build_host.py --force)")
        mod.init(_lib.GSR_LIB_PATH, _lib.GAB_LIB_PATH, _lib.GLS_LIB_PATH)
        _mod = mod
    return _mod


def get():
    """The compiled host module, or None when the Python twins are selected."""
    if not _enabled:
        return None
    return _mod if _mod is not None else load()
