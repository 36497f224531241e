#!/usr/bin/env python3
"""Builds librtd.so (the HIP kernels + C ABI of include/rtd.h) for gfx950, in-tree.

hipcc cross-compiles without a GPU.  Objects are rebuilt only when their sources changed.
Usage: python pythonic-disort_amd/build.py [--force]
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "pydisort_amd")
OBJ_DIR = os.path.join(HERE, "build")
LIB = os.path.join(OUT_DIR, "librtd.so")
SOURCES = ["rtd_api.hip", "rtd_eig.hip", "rtd_eig_small.hip", "rtd_bc.hip", "rtd_bc_small.hip", "rtd_bc_tile2.hip", "rtd_bc_wide.hip", "rtd_eval.hip", "rtd_nt.hip", "rtd_bdrf.hip", "rtd_prep.hip"]
HEADERS = [os.path.join(CSRC, "rtd_device.h"), os.path.join(CSRC, "rtd_dd.h"), os.path.join(CSRC, "rtd_bc_common.h"), os.path.join(CSRC, "rtd_bc_tile_common.h"), os.path.join(HERE, "..", "include", "rtd.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-save-temps=obj",
         "-Wno-unused-command-line-argument"] + os.environ.get("RTD_EXTRA_FLAGS", "").split()
HAZARD_CHECK = os.path.join(HERE, "..", "tools", "check_dpp_hazards.py")
# flags of single translation units: what was measured to pay for that unit's kernels (and only there), plus RTD_EXTRA_FLAGS_<stem>
# from the environment for A/B builds (tools/build_variant.py)
PER_SOURCE_FLAGS = {}


def _flags_for(src):
    stem = src.replace(".hip", "")
    return FLAGS + PER_SOURCE_FLAGS.get(src, []) + os.environ.get("RTD_EXTRA_FLAGS_" + stem, "").split()


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src):
    obj = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
    path = os.path.join(CSRC, src)
    if _stale(obj, [path] + HEADERS):
        # compile to a temporary name: the object only takes its final name once the hazard check has passed, so a
        # failed check can never leave an object behind that the next run would take for up to date
        tmp = obj + ".unchecked"
        stem = src.replace(".hip", "")

        def cleanup():
            for f in os.listdir(OBJ_DIR):  # -save-temps leaves ~10 MB of intermediates per source
                if f.startswith(stem + "-h") or f.startswith(stem + ".hip-") or f.startswith(stem + ".o.unchecked-"):
                    os.remove(os.path.join(OBJ_DIR, f))

        try:
            subprocess.run([HIPCC] + _flags_for(src) + ["-c", path, "-o", tmp], check=True, cwd=OBJ_DIR)
            # the kernels update registers with v_fmac_f64_dpp from inline asm: the compiler's hazard recogniser cannot
            # see those writes, so the generated ISA is scanned for a DPP read that follows one too closely
            dev_asm = [os.path.join(OBJ_DIR, f) for f in os.listdir(OBJ_DIR)
                       if "amdgcn" in f and f.endswith(".s") and (f.startswith(stem + "-hip-") or f.startswith(stem + ".o.unchecked-hip-"))]
            if not dev_asm:
                raise RuntimeError(f"{src}: no device assembly (*-hip-amdgcn*.s) found among the -save-temps outputs in "
                                   f"{OBJ_DIR}: the DPP hazard check cannot run (did the toolchain rename them?)")
            chk = subprocess.run([sys.executable, HAZARD_CHECK] + dev_asm, capture_output=True, text=True)
            if chk.returncode != 0:
                sys.stderr.write(chk.stdout + chk.stderr)
                raise RuntimeError(f"{src}: DPP read-after-write hazard in the generated ISA (see above)")
            os.replace(tmp, obj)
        except BaseException:
            if os.path.exists(tmp):
                os.remove(tmp)
            raise
        finally:
            cleanup()
    return obj


def build(force=False):
    os.makedirs(OBJ_DIR, exist_ok=True)
    if force:
        for f in os.listdir(OBJ_DIR):
            os.remove(os.path.join(OBJ_DIR, f))
    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(_compile, SOURCES))
    if _stale(LIB, objs):
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
