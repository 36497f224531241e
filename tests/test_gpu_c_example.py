"""The C ABI on its own: examples/solve_columns.c -- plain C99, no Python, no C++ -- compiled against include/rtd.h, linked to
librtd.so, run on the GPU, and its printed fluxes and intensities compared with the Python front end on the same inputs (the
quadrature nodes come from a Newton iteration in C there and from numpy.polynomial.legendre.leggauss here: agreement to 1e-12).
What the reference's counterpart would be: one pydisort() call per column and calls of the returned closures
(src/PythonicDISORT/pydisort.py:13-29, _assemble_intensity_and_fluxes.py:170-613)."""
import os
import re
import shutil
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pythonic-disort_amd")]


def test_plain_c_program_against_the_python_front_end(tmp_path):
    import pydisort_amd as amd
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    libdir = os.path.join(ROOT, "pythonic-disort_amd", "pydisort_amd")
    exe = str(tmp_path / "solve_columns")
    subprocess.run([gcc, "-std=c99", "-pedantic", "-Wall", "-Werror", "-O2", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", "solve_columns.c"), os.path.join(libdir, "librtd.so"), "-lm", "-Wl,-rpath," + libdir, "-o", exe],
                   check=True)
    C, L, NQ = 5, 3, 16
    r = subprocess.run([exe, str(C)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = [[float(x) for x in re.findall(r"[-+]?\d\.\d+(?:e[-+]?\d+)?|[-+]?\d+\.\d+(?:e[-+]?\d+)?", ln.split(":", 1)[1])] for ln in r.stdout.splitlines() if ln.startswith("column")]
    assert len(rows) == C and all(len(v) == 5 for v in rows), r.stdout
    c = np.arange(C)[:, None]
    l = np.arange(L)[None, :]
    g = 0.55 + 0.1 * l + 0.01 * c
    cfg = dict(tau_arr=0.4 * (l + 1) * (1.0 + 0.05 * c), omega_arr=np.broadcast_to(0.95 - 0.1 * l, (C, L)).copy(), NQuad=NQ,
               Leg_coeffs_all=g[:, :, None] ** np.arange(NQ + 1)[None, None, :], mu0=0.3 + 0.6 * (np.arange(C) + 1.0) / (C + 1.0),
               I0=np.full(C, 3.0), phi0=np.full(C, 0.5), f_arr=g ** NQ)
    _, sol = amd.pydisort_batch(**cfg)
    tau = np.stack([np.zeros(C), 0.37 * cfg["tau_arr"][:, -1]], axis=1)
    phi = np.array([0.0, 2.0])
    u = sol.u(tau, phi)
    fu, (fd, fdir) = sol.flux_up(tau), sol.flux_down(tau)
    want = np.stack([fu[:, 0], fd[:, 0], fdir[:, 0], u[:, 0, 1, 1], u[:, NQ // 2, 1, 0]], axis=1)
    got = np.array(rows)
    assert np.allclose(got, want, rtol=1e-12, atol=1e-14), np.max(np.abs(got - want) / np.abs(want))
    sol.plan.close()
