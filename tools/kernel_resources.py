#!/usr/bin/env python3
"""Register / LDS / spill figures of every kernel in librtd.so, read from the gfx950 code-object metadata (the AMDGPU
note records of the offload bundle), written to profiles/<round>_kernel_resources.json.  Comments and DESIGN.md quote these
numbers: regenerate the file after every build that changes a kernel instead of editing the numbers by hand.

Usage: python tools/kernel_resources.py [--out profiles/rNN_kernel_resources.json]   (without --out: prints only)
"""
import argparse
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
FIELDS = ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "group_segment_fixed_size",
          "private_segment_fixed_size", "max_flat_workgroup_size", "wavefront_size")


def demangle(names):
    exe = shutil.which("c++filt") or shutil.which("llvm-cxxfilt")
    if not exe:
        return names
    r = subprocess.run([exe], input="\n".join(names), capture_output=True, text=True)
    return r.stdout.splitlines() if r.returncode == 0 else names


def kernels_of(lib):
    tmp = tempfile.mkdtemp(prefix="rtd_co_")
    try:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(lib, local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], capture_output=True, text=True, cwd=tmp, check=True)
        out = {}
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, f)], capture_output=True, text=True).stdout
            # the metadata is YAML: one "- .agpr_count: ..." item per kernel under amdhsa.kernels
            body = notes[notes.index("amdhsa.kernels:"):] if "amdhsa.kernels:" in notes else ""
            for item in re.split(r"\n\s*- \.", body)[1:]:
                item = "." + item
                name = re.search(r"^\s*\.name:\s*(\S+)", item, flags=re.M)
                if not name:
                    continue
                rec = {}
                for k in FIELDS:
                    m = re.search(r"^\s*\." + k + r":\s*(\d+)", item, flags=re.M)
                    if m:
                        rec[k] = int(m.group(1))
                out[name.group(1)] = rec
        names = list(out)
        return {d: out[n] for n, d in zip(names, demangle(names))}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def waves_per_simd(rec):
    """Occupancy bound by registers alone: 512 VGPR+AGPR per SIMD lane, allocated in blocks of 8."""
    regs = rec.get("vgpr_count", 0) + 0  # (vgpr_count of the metadata already includes the AGPRs on gfx90a+ unified files)
    regs = max(8, (regs + 7) // 8 * 8)
    return min(8, 512 // regs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=os.path.join(ROOT, "pythonic-disort_amd", "pydisort_amd", "librtd.so"))
    ap.add_argument("--out", default=None, help="also write the figures as JSON there (default: print only)")
    a = ap.parse_args()
    ks = kernels_of(a.lib)
    for rec in ks.values():
        rec["waves_per_simd_by_registers"] = waves_per_simd(rec)
    doc = {"source": "AMDGPU code-object metadata (llvm-readelf --notes) of " + os.path.relpath(a.lib, ROOT),
           "tool": "tools/kernel_resources.py", "kernels": dict(sorted(ks.items()))}
    if a.out:
        with open(a.out, "w") as f:
            json.dump(doc, f, indent=1)
    for n, r in sorted(ks.items()):
        print(f"{r.get('vgpr_count', 0):4d} vgpr {r.get('agpr_count', 0):4d} agpr {r.get('vgpr_spill_count', 0):4d} vspill "
              f"{r.get('sgpr_spill_count', 0):4d} sspill {r.get('group_segment_fixed_size', 0):6d} lds  {n[:110]}")


if __name__ == "__main__":
    sys.exit(main())
