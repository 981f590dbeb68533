"""Import shims for the third-party packages the reference imports unconditionally but which are absent on an
MI355X box (CUDA-only builds or un-vendored submodules -- SURVEY.md Appendix B / D):

    roma                 4 quaternion functions        scene/gaussian_model.py:21, scene/flame_gaussian_model.py:18
    plyfile              PlyData / PlyElement subset   scene/gaussian_model.py:19, scene/dataset_readers.py:23
    simple_knn._C        distCUDA2                     scene/gaussian_model.py:23 (init of un-bound models only)
    nvdiffrast.torch     contexts construct, use raises  mesh_renderer/__init__.py:10 (debug overlay only)
    dearpygui.dearpygui  import-time stub              utils/viewer_utils.py:17 (fps_benchmark_demo.py needs OrbitCamera only)
    tyro                 import-time stub              viewers
    iopath.common.file_io  PathManager.open            utils/pytorch3d_load_obj.py:47 (FlameHead's template OBJ)
    PIL.Image.fromarray  int8 + explicit mode          scene/__init__.py:51 on Pillow >= 11.3 (pillow_compat: a wrapper, not a module)

`install()` registers a shim under the third-party name ONLY when the real package cannot be imported, so an
environment that has the real thing keeps it.  Together with the top-level `diff_gaussian_rasterization` package of this
repository that closes the import closure of train.py / render.py / fps_benchmark_*.py (Appendix D) except torchvision /
lpips, which `train.py` needs for evaluation metrics only (`install(stub_torchvision=True)` stubs it at import time).
"""
from __future__ import annotations

import importlib
import importlib.util
import sys
import types

_REAL = {}   # third-party name -> shim module path inside this package
_SHIMS = {
    "roma": "roma",
    "plyfile": "plyfile",
    "simple_knn": "simple_knn",
    "simple_knn._C": "simple_knn._C",
    "nvdiffrast": "nvdiffrast",
    "nvdiffrast.torch": "nvdiffrast.torch",
    "iopath": "iopath",
    "iopath.common": "iopath.common",
    "iopath.common.file_io": "iopath.common.file_io",
}
_STUBS = ("dearpygui", "dearpygui.dearpygui", "tyro")


def _importable(name: str) -> bool:
    if name in sys.modules:
        return not getattr(sys.modules[name], "__gaussianavatars_amd_shim__", False)
    try:
        return importlib.util.find_spec(name.split(".")[0]) is not None
    except (ImportError, ValueError):
        return False


class _Stub(types.ModuleType):
    """Import-time stand-in: attribute access hands out further stubs, calling one raises."""

    __gaussianavatars_amd_shim__ = True

    def __getattr__(self, item):
        if item.startswith("__"):
            raise AttributeError(item)
        child = _Stub(f"{self.__name__}.{item}")
        setattr(self, item, child)
        return child

    def __call__(self, *a, **k):
        raise RuntimeError(f"{self.__name__} is an import-time stub of gaussianavatars_amd.shims: the real package is not installed")


def pillow_compat() -> bool:
    """The reference composes its training images as `Image.fromarray(np.array(arr * 255.0, dtype=np.byte), "RGB")`
    (scene/__init__.py:51, scene/dataset_readers.py:229): an INT8 array with an explicit mode.  The Pillow it was written against took the
    raw bytes under the given mode; Pillow >= 11.3 checks the dtype against its type map and raises TypeError.  With the real Pillow
    installed, `Image.fromarray` is wrapped so that exactly this call -- explicit mode, int8 data -- is served as before (the same bytes
    viewed as uint8); every other call goes through untouched.  Returns True when the wrapper was installed by this call."""
    try:
        import numpy as np
        from PIL import Image
    except ImportError:
        return False
    if getattr(Image.fromarray, "__gaussianavatars_amd_shim__", False):
        return False
    try:
        Image.fromarray(np.zeros((1, 1, 3), np.int8), "RGB")
        return False            # this Pillow still accepts it: nothing to do
    except TypeError:
        pass
    except Exception:           # noqa: BLE001 -- anything else: leave Pillow alone
        return False
    orig = Image.fromarray

    def fromarray(obj, mode=None):
        if mode is not None and isinstance(obj, np.ndarray) and obj.dtype == np.int8:
            obj = obj.view(np.uint8)
        return orig(obj, mode) if mode is not None else orig(obj)

    fromarray.__gaussianavatars_amd_shim__ = True
    fromarray.__wrapped__ = orig
    fromarray.__doc__ = orig.__doc__
    Image.fromarray = fromarray
    return True


def install(stub_torchvision: bool = False) -> list:
    """Registers the shims that are needed; returns the list of third-party names now served by a shim."""
    served = []
    if pillow_compat():
        served.append("PIL.Image.fromarray(int8, mode)")
    for name, rel in _SHIMS.items():
        top = name.split(".")[0]
        if top in _REAL.get("_kept", ()):   # the real top-level package exists: never mix
            continue
        if _importable(top) and not getattr(sys.modules.get(top), "__gaussianavatars_amd_shim__", False):
            _REAL.setdefault("_kept", set()).add(top)
            continue
        mod = importlib.import_module(f"{__name__}.{rel}")
        mod.__gaussianavatars_amd_shim__ = True
        sys.modules[name] = mod
        served.append(name)
    stubs = list(_STUBS) + (["torchvision", "torchvision.models", "torchvision.transforms", "torchvision.transforms.functional"]
                            if stub_torchvision else [])
    for name in stubs:
        top = name.split(".")[0]
        if _importable(top):
            continue
        if name not in sys.modules:
            sys.modules[name] = _Stub(name)
            if "." in name:
                parent, leaf = name.rsplit(".", 1)
                if parent in sys.modules:
                    setattr(sys.modules[parent], leaf, sys.modules[name])
        served.append(name)
    return served


def uninstall() -> None:
    """Removes every shim / stub this package registered (tests)."""
    for name in list(sys.modules):
        if getattr(sys.modules[name], "__gaussianavatars_amd_shim__", False) and not name.startswith(__name__):
            del sys.modules[name]
    _REAL.clear()
    try:
        from PIL import Image

        if getattr(Image.fromarray, "__gaussianavatars_amd_shim__", False):
            Image.fromarray = Image.fromarray.__wrapped__
    except ImportError:
        pass
