#!/usr/bin/env python3
"""Print value and per-stage ms of bench.py for the library selected by RTD_LIB (A/B tool)."""
import json, subprocess, sys
out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--steps", "5", "--warmup", "2"], capture_output=True, text=True).stdout
d = json.loads(out.strip().splitlines()[-1])
print(f"{d['value']:.0f} col/s", {k: round(v, 2) for k, v in d["roofline"]["kernel_ms_per_step"].items()})
