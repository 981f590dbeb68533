#!/usr/bin/env python3
"""Builds gaussianavatars_amd/gaa_host.so (csrc/gaa_host.cpp: the compiled host side of the autograd nodes) with g++ against the torch
headers of THIS interpreter.  No device code in it: the kernels stay in the three hipcc-built libraries, reached through the C ABI.
Called by the Makefile (`make host`) and by __graft_entry__.build().  ~40 s (torch/extension.h)."""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "gaa_host.so")
SRC = os.path.join(HERE, "gaa_host.cpp")
STAMP = OUT + ".stamp"     # what the module was built against: a torch upgrade or another interpreter makes it stale whatever the mtimes say
DEPS = [SRC, __file__] + [os.path.join(HERE, "..", "..", "include", h) for h in ("gsr.h", "gab.h", "gls.h")]


def build_stamp() -> str:
    """torch version, its C++ ABI flag, the interpreter's ABI tag: a module built against anything else fails at import with an undefined symbol."""
    import torch

    return "torch=%s cxx11abi=%d python=%s soabi=%s" % (torch.__version__, int(torch._C._GLIBCXX_USE_CXX11_ABI), sys.version.split()[0],
                                                       sysconfig.get_config_var("SOABI"))


def stamp_matches() -> bool:
    try:
        return open(STAMP).read().strip() == build_stamp()
    except OSError:
        return False


def up_to_date() -> bool:
    return os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in DEPS) and stamp_matches()


def main() -> int:
    if up_to_date() and "--force" not in sys.argv:
        return 0
    import torch
    from torch.utils import cpp_extension as ce

    tl = ce.library_paths()[0]
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function", "-Wno-sign-compare",
           "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DTORCH_EXTENSION_NAME=gaa_host", "-DTORCH_API_INCLUDE_EXTENSION_H",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}"]
    cmd += [f"-I{p}" for p in ce.include_paths()] + [f"-I{rocm}/include", f"-I{sysconfig.get_paths()['include']}"]
    cmd += [SRC, "-o", OUT, f"-L{tl}", "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip", "-ltorch", "-ltorch_python", "-ldl", f"-Wl,-rpath,{tl}"]
    print("building gaa_host.so (g++, torch %s) ..." % torch.__version__, flush=True)
    r = subprocess.run(cmd)
    if r.returncode == 0:
        with open(STAMP, "w") as fh:
            fh.write(build_stamp() + "\n")
    return r.returncode


if __name__ == "__main__":
    sys.exit(main())
