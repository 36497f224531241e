#!/usr/bin/env python3
"""Builds tests/stub/librccl_stub.so (test infrastructure: see rccl_stub.cpp) for gfx950; hipcc cross-compiles without a GPU."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "rccl_stub.cpp")
LIB = os.path.join(HERE, "librccl_stub.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def build(force=False):
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", "-x", "hip", SRC, "-o", LIB, "-lrt"],
                       check=True, cwd=HERE)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
