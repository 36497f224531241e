#!/usr/bin/env python3
"""Scans gfx950 assembly (hipcc -S / -save-temps output) for the one hazard the compiler cannot see: a VALU write made
inside inline asm (our v_fmac_f64_dpp updates) followed within two wait states by a DPP instruction that reads the
written VGPR as its permuted source.  (MI300 ISA section 4.5: "VALU writes VGPR -> VALU DPP reads that VGPR: 2 wait
states"; the hazard recogniser handles compiler-emitted VALU writes only.)  Exit status 1 when a hazard is found.

Usage: python tools/check_dpp_hazards.py file.s [file.s ...]"""
import re
import sys


def instructions(path):
    with open(path) as f:
        for line in f:
            if line.startswith("\t") and not line.startswith("\t;") and not line.startswith("\t."):
                yield line.rstrip("\n")


def regs_of(text):
    used = set()
    for a, b, c in re.findall(r"v\[(\d+):(\d+)\]|\bv(\d+)\b", text):
        if a:
            used |= set(range(int(a), int(b) + 1))
        if c:
            used.add(int(c))
    return used


def scan(path):
    lines = list(instructions(path))
    found = []
    for i, line in enumerate(lines):
        m = re.match(r"\tv_fmac_f64_dpp v\[(\d+):(\d+)\]", line)
        if not m:
            continue
        written = set(range(int(m.group(1)), int(m.group(2)) + 1))
        wait = 0
        for nxt in lines[i + 1:i + 3]:
            if "s_nop" in nxt:
                wait += int(nxt.split()[-1]) + 1
                continue
            if wait >= 2:
                break
            if "_dpp" in nxt and "," in nxt:
                src0 = nxt.split(",")[1]  # the DPP-permuted operand is src0
                if regs_of(src0) & written:
                    found.append((line.strip(), nxt.strip()))
            wait += 1
    return found


if __name__ == "__main__":
    bad = 0
    for path in sys.argv[1:]:
        for a, b in scan(path):
            print(f"{path}: DPP read-after-write hazard:\n    {a}\n    {b}")
            bad += 1
    print(f"checked {len(sys.argv) - 1} file(s): {bad} hazard(s)")
    sys.exit(1 if bad else 0)
