#!/usr/bin/env python3
"""A/B and diagnostic builds of librtd.so beside the shipped one (same ABI, selected at run time with RTD_LIB):
    python tools/build_variant.py stamps -DRTD_EIG_STAMPS      ->  pythonic-disort_amd/pydisort_amd/librtd_stamps.so
Objects go to pythonic-disort_amd/build_<name>/ (git-ignored like the shipped build's); the .so travels to the GPU box."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("rtd_build", os.path.join(ROOT, "pythonic-disort_amd", "build.py"))
b = importlib.util.module_from_spec(spec)
spec.loader.exec_module(b)
name, flags = sys.argv[1], sys.argv[2:]
b.OBJ_DIR = os.path.join(b.HERE, "build_" + name)
b.LIB = os.path.join(b.OUT_DIR, f"librtd_{name}.so")
b.FLAGS = b.FLAGS + flags
print(b.build())
