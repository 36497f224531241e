"""Pins the CPU oracle (oracle/disort_oracle.py) to the reference: every pydisort call and every
closure evaluation recorded from the reference's 42 pytest cases (tests/golden/ref) is replayed
through the oracle and compared in float64.  Tolerance: 1e-7 of the call's radiation-field scale
(the near-conservative cases, omega = 1-1e-6, are ill-conditioned: two LAPACK orderings of the
same algorithm already differ by ~1e-8 there; everything else agrees to <1e-11)."""
import numpy as np
import pytest

import goldens
from oracle import disort_oracle as O

ILL_CONDITIONED = {"1b", "1e", "2b", "2d", "3a", "3b", "4a", "5a"}  # omega = 1 - 1e-6


def replay(call, solver):
    kw = call["kwargs"]
    res = solver(**kw)
    assert np.allclose(res[0], call["mu_arr"], rtol=0, atol=1e-14)
    fns = dict(zip(["flux_up", "flux_down", "u0", "u"], res[1:]))
    scale = max(max(np.max(np.abs(o), initial=0.0) for o in
                    (ev["out"] if isinstance(ev["out"], tuple) else (ev["out"],))) for ev in call["evals"])
    worst = 0.0
    for ev in call["evals"]:
        got = fns[ev["name"]](*ev["args"], **ev["kwargs"])
        gots = got if isinstance(got, tuple) else (got,)
        wants = ev["out"] if isinstance(ev["out"], tuple) else (ev["out"],)
        assert len(gots) == len(wants)
        for g, w in zip(gots, wants):
            assert np.shape(g) == np.shape(w), (ev["name"], np.shape(g), np.shape(w))
            worst = max(worst, float(np.max(np.abs(np.asarray(g) - w), initial=0.0)) / scale)
    return worst


@pytest.mark.parametrize("test_id", goldens.list_ids())
def test_oracle_matches_reference(test_id):
    tol = 1e-7 if test_id in ILL_CONDITIONED else 1e-10
    for call in goldens.load(test_id):
        assert replay(call, O.pydisort) < tol


@pytest.mark.parametrize("name", ["cfg1_q4", "cfg2_q32"])
def test_oracle_on_baseline_literal_configs(name):
    """BASELINE.json configs[0] (TP1 with 4 streams: the CPU plumbing case) and configs[1] (32-stream TP5-like)."""
    import os
    from pydisort_amd import synthetic
    kw, tau_pts = synthetic.literal_cases()[name]
    z = np.load(os.path.join(goldens.HERE, "golden", "synth", name + ".npz"))
    mu_arr, Fp, Fm, u0, u = O.pydisort(**kw)
    scale = np.max(np.abs(z["u"]))
    assert np.max(np.abs(u(tau_pts, z["phi"]) - z["u"])) / scale < 1e-10
    assert np.allclose(Fp(tau_pts), z["flux_up"], rtol=1e-10, atol=1e-13 * scale)


@pytest.mark.parametrize("name", ["q72", "q96", "q128"])
def test_oracle_beyond_64_streams(name):
    """72 / 96 / 128 streams (pydisort_amd.synthetic.many_stream_cases) against the reference run here.  The two float64
    implementations of the same algorithm agree to ~3e-10 of the field scale at these sizes (eigenvector conditioning),
    hence 1e-8."""
    import os
    from pydisort_amd import synthetic
    kw, tau_pts = synthetic.many_stream_cases()[name]
    z = np.load(os.path.join(goldens.HERE, "golden", "synth", name + ".npz"))
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mu_arr, Fp, Fm, u0, u = O.pydisort(**kw)
    scale = np.max(np.abs(z["u"]))
    assert np.max(np.abs(u(tau_pts, z["phi"]) - z["u"])) / scale < 1e-8
    assert np.allclose(Fp(tau_pts), z["flux_up"], rtol=1e-9, atol=1e-11 * scale)


@pytest.mark.parametrize("name", ["q96_L20", "q72_L50", "q128_L50"])
def test_oracle_on_the_timed_many_stream_workloads_at_full_depth(name):
    """The 66 ... 128-stream workloads that are timed (96 x 20 x 48, 72 x 50 x 36, 128 x 50 x 64: pydisort_amd.synthetic.
    many_stream_deep_cases) at their full depth, first column, against the reference run here (round-4 verdict: the only
    128-stream golden was 2 layers deep)."""
    import os
    import warnings
    from pydisort_amd import synthetic
    maker_kw, nf, ncol = synthetic.many_stream_deep_cases()[name]
    cfg = synthetic.cfg4_columns(1, **maker_kw)
    z = np.load(os.path.join(goldens.HERE, "golden", "synth", name + ".npz"))
    from threadpoolctl import threadpool_limits
    with warnings.catch_warnings(), threadpool_limits(2):  # (many small LAPACK calls: more BLAS threads only contend)
        warnings.simplefilter("ignore")
        mu_arr, Fp, Fm, u0, u = O.pydisort(NFourier=nf, **synthetic.column_kwargs(cfg, 0))
    tau_pts = z["c0.tau_pts"]
    scale = np.max(np.abs(z["c0.u"]))
    assert np.max(np.abs(u(tau_pts, z["phi"]) - z["c0.u"])) / scale < 1e-8
    assert np.allclose(Fp(tau_pts), z["c0.flux_up"], rtol=1e-9, atol=1e-11 * scale)


def test_oracle_against_high_precision_truth():
    """How far the reference's algorithm in float64 (this oracle) is from a 40-digit solution (tools/hp_truth_m0.py):
    rounding level on a benign atmosphere, ~6e-9 when omega = 1 - 1e-6 layers are present.  GPU parity tests against
    the oracle use a looser tolerance for such inputs for this reason."""
    import importlib.util
    import os
    import numpy as np
    from oracle import disort_oracle as O
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("hp_cases", os.path.join(here, "..", "tools", "hp_cases.py"))
    hp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(hp)
    Z = np.load(os.path.join(here, "golden", "hp_truth_m0.npz"))
    for name, kw, tol in (("benign", hp.benign_case(), 1e-12), ("harsh", hp.harsh_case(), 1e-7)):
        _, fu, fd, u0 = O.pydisort(**kw)
        tau = np.concatenate(([0.0], kw["tau_arr"]))
        truth = Z[name]
        err = np.max(np.abs(u0(tau) - truth)) / np.max(np.abs(truth))
        assert err <= tol, (name, err)
        assert np.allclose(u0(tau), Z[name + "_oracle"], rtol=0, atol=1e-11 * np.max(np.abs(truth)))
    kw = hp.intensity_case()
    u = O.pydisort(**kw)[4]
    tau = np.concatenate(([0.0], kw["tau_arr"]))
    truth = Z["intensity"]
    err = np.max(np.abs(u(tau, hp.PHI) - truth)) / np.max(np.abs(truth))
    assert err <= 1e-8, err   # 2e-10 observed: one omega = 1 - 1e-6 layer


def test_oracle_against_high_precision_truth_32_streams():
    """How far the reference's algorithm in float64 (the oracle) is from a 40-digit solution of a 20-layer, 32-stream
    atmosphere with four omega = 1 - 1e-6 layers (tools/hp_truth_q32.py): this is the error budget behind the
    tolerances of the oracle-based GPU tests on near-conservative atmospheres (the HIP path itself is held to 1e-9
    against the same truth, tests/test_gpu_parity.py::test_high_precision_truth_32_streams)."""
    import os
    import numpy as np
    from oracle import disort_oracle as O
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hp_truth_q32.npz"))
    kw = {k[3:]: (z[k] if z[k].ndim else z[k][()]) for k in z.files if k.startswith("in.")}
    kw["NQuad"] = int(kw["NQuad"])
    p = O.prepare(**kw)
    tau = np.concatenate(([0.0], kw["tau_arr"]))
    modes = [int(m) for m in z["modes"]]
    um = O.Solution(p)._um(modes, tau)[0] * p["rescale"]
    worst = max(np.max(np.abs(um[i] - z[f"um{m}"])) / np.max(np.abs(z[f"um{m}"])) for i, m in enumerate(modes))
    assert worst < 1e-6       # the oracle is within the north star's tolerance of the truth ...
    assert worst > 1e-9       # ... but 6.4e-8 off (mode 0): it cannot arbitrate below ~1e-7 on such atmospheres


def test_oracle_against_high_precision_truth_56_streams():
    """56 streams, 8 layers, a thin top layer with omega = 1 - 1e-6 (tools/hp_truth_q32.py --q56): the reference's algorithm
    in float64 is 3.4e-6 of the field scale off the 40-digit solution -- beyond the north star's own 1e-6.  This is why
    oracle-based tolerances of near-conservative many-stream cases are 2e-5, while the HIP path is held to 1e-9 against the
    truth itself (tests/test_gpu_parity.py::test_high_precision_truth_56_streams)."""
    import os
    import warnings
    import numpy as np
    from oracle import disort_oracle as O
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hp_truth_q56.npz"))
    kw = {k[3:]: (z[k] if z[k].ndim else z[k][()]) for k in z.files if k.startswith("in.")}
    kw["NQuad"], kw["only_flux"] = int(kw["NQuad"]), bool(kw["only_flux"])
    tau = np.concatenate(([0.0], kw["tau_arr"]))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        orc = O.pydisort(**kw)[3](tau)
    err = np.max(np.abs(orc - z["um0"])) / np.max(np.abs(z["um0"]))
    assert 1e-7 < err < 2e-5
