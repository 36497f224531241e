#!/usr/bin/env python3
"""What the REFERENCE raises for the invalid inputs of tests/frontend_mutations.py (this container only: imports
/root/reference/src): exception type and text per (case, seed) -> tests/golden/frontend_errors.json.
Usage: PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_frontend_error_goldens.py"""
import json
import os
import sys
import warnings

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = ["/root/reference/src", os.path.dirname(HERE)]
import PythonicDISORT  # noqa: E402
import frontend_mutations as F  # noqa: E402

out = {}
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for name in F.MUTATIONS:
        for seed in F.SEEDS:
            kw = F.case(name, seed)
            try:
                PythonicDISORT.pydisort(**{k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in kw.items()})
                out[f"{name}|{seed}"] = ["accepted", ""]
            except Exception as e:  # noqa: BLE001
                out[f"{name}|{seed}"] = [type(e).__name__, str(e)]
json.dump(out, open(os.path.join(HERE, "frontend_errors.json"), "w"), indent=1, sort_keys=True)
print(len(out), "cases;", sum(v[0] == "accepted" for v in out.values()), "accepted by the reference")
