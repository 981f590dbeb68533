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
DEPS = [SRC, __file__] + [os.path.join(HERE, "..", "..", "include", h) for h in ("gsr.h", "gab.h", "gls.h")]


def up_to_date() -> bool:
    return os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in DEPS)


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
    return r.returncode


if __name__ == "__main__":
    sys.exit(main())
