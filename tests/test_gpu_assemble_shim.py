"""The inner boundary, executed: ``pydisort_amd._assemble._assemble_intensity_and_fluxes`` is called with the 34 positional
arguments the reference's own ``pydisort`` handed to ITS ``_assemble_intensity_and_fluxes`` (captured by
tests/golden/make_assemble_goldens.py from the reference: delta-M scaled inputs, rescaled sources, BDRF modes as tables) and
the returned callables are held to what the reference's callables returned at the same points."""
import inspect
import os

import numpy as np
import pytest

import goldens

pytestmark = pytest.mark.gpu
DIR = os.path.join(goldens.HERE, "golden", "assemble")
IDS = sorted(f[:-4] for f in os.listdir(DIR) if f.endswith(".npz"))
ILL = {"3a", "5a"}  # omega = 1 - 1e-6


def _args(z, p, names):
    keys = set(z.files)
    out = []
    for k in names:
        if k == "BDRF_Fourier_modes":
            modes = []
            for mi in range(int(z[f"{p}.nbdrf"])):
                if f"{p}.bdrf{mi}.scalar" in keys:
                    modes.append(float(z[f"{p}.bdrf{mi}.scalar"]))
                else:
                    modes.append(goldens.TabulatedBDRF(z[f"{p}.bdrf{mi}.tab"], z[f"{p}.bdrf{mi}.tab0"]))
            out.append(modes)
        elif f"{p}.none.{k}" in keys:
            out.append(None)
        else:
            v = z[f"{p}.arg.{k}"]
            out.append(v[()] if v.ndim == 0 else np.array(v))
    return out


@pytest.mark.parametrize("test_id", IDS)
def test_shim_replays_the_references_positional_arguments(test_id):
    from conftest import record_parity
    from pydisort_amd._assemble import _assemble_intensity_and_fluxes as shim
    assert len(IDS) >= 8
    z = np.load(os.path.join(DIR, test_id + ".npz"))
    names = [str(s) for s in z["names"]]
    phi = z["phi"]
    worst = worst_pw = 0.0
    for ci in range(int(z["ncalls"])):
        p = f"c{ci}"
        res = shim(*_args(z, p, names))   # positional, all 34
        tau = z[f"{p}.tau_pts"]
        want_u0 = z[f"{p}.out.u0"]
        scale = max(np.max(np.abs(want_u0)), 1e-300)
        got_fd = res[1](tau)
        for got, key in ((res[0](tau), "flux_up"), (got_fd[0], "flux_down_diffuse"), (got_fd[1], "flux_down_direct"), (res[2](tau), "u0")):
            want = z[f"{p}.out.{key}"]
            assert np.shape(got) == np.shape(want), key
            fs = max(np.max(np.abs(want)), scale)
            assert np.max(np.abs(got - want)) <= (1e-7 if test_id in ILL else 1e-9) * fs, (key, np.max(np.abs(got - want)) / fs)
        if f"{p}.out.u" in z.files:
            assert len(res) == 4
            a, b = goldens.max_rel_err(res[3](tau, phi), z[f"{p}.out.u"])
            worst, worst_pw = max(worst, a), max(worst_pw, b)
        else:
            assert len(res) == 3
    if worst > 0:
        record_parity("assemble/" + test_id, worst, worst_pw, 1e-7 if test_id in ILL else 1e-9, 1e-6, against="reference")
