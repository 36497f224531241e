"""Parity of the HIP path (through the C ABI) with the reference -- run with ``-m gpu`` on an MI355X.

* every pydisort call and closure evaluation captured from the reference's 42 pytest cases
  (tests/golden/ref) replayed through pydisort_amd.pydisort;
* the Stamnes DISORT 4.0.99 comparisons of those cases with the reference's own thresholds;
* BASELINE.json's synthetic configs: reference-computed goldens (tests/golden/synth) and, at larger
  column counts, the CPU oracle on the same seeded inputs;
* size-independent properties at the full cfg4 size (flux consistency, reciprocity of batching).

Tolerance (written here, north star = 1e-6 relative on intensities): 1e-9 of the radiation-field
scale of the call, 1e-6 for the near-conservative cases (omega = 1 - 1e-6) where two LAPACK orderings
of the reference's own algorithm already differ by ~1e-8.
"""
import os
import warnings
from math import pi

import numpy as np
import pytest

import goldens

pytestmark = pytest.mark.gpu

ILL_CONDITIONED = {"1b", "1e", "2b", "2d", "3a", "3b", "4a", "5a"}
TOL, TOL_ILL = 1e-9, 1e-7   # scale-relative; omega = 1 - 1e-6 cases: measured <= 2.4e-8 (the oracle's own error there, see
#                             test_high_precision_truth_32_streams)
ROOT_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PW_TOL = 1e-6               # north star: intensities within 1e-6 relative of the reference, pointwise
# 8ARTS_A (thermal emission, 20 layers, intensities spanning six decades): the reference and the oracle -- the same
# algorithm, the same LAPACK calls, both float64 on the CPU -- differ by 7.0e-5 pointwise at intensities 1e-6 of the
# largest (9e-11 of the field scale), and the reference's own captured result is 5.0e-5 pointwise off the 40-digit solution
# (tools/hp_truth_case.py golden 8ARTS_A).  Against the reference the pointwise metric is therefore not held for this case
# (None: the report says null); it is held against the 40-digit solution instead, in
# test_golden_case_against_high_precision_truth, at PW_TRUTH_TOL: since round 5 the kernels keep the source polynomials about
# the top of their own layer (csrc/rtd_dd.h) and are 3.7e-6 from the truth there (2.6e-5 before) -- the float64 floor of the
# polynomial particular solution itself: in the 1.7e-6-thin bottom layer its constant term is 1.4e4 times the intensity (1.4e-6 of
# the largest) it leaves after cancelling against the homogeneous part; the absolute error there is 5e-12.
PW_EXCEPT = {"8ARTS_A": None}
WIDE_TOL, WIDE_PW_TOL = 1e-8, 4e-7  # 66 ... 128 streams against the reference: 10 x the measured 7.3e-10 / 3.5e-8 (round 5 held 1e-7 / 1e-6)
PW_TRUTH_TOL = {"8ARTS_A": 5e-6}


@pytest.fixture(scope="module")
def amd():
    import pydisort_amd
    from pydisort_amd import _engine
    assert _engine.device_count() >= 1, "no HIP device visible"
    return pydisort_amd


def _replay(call, solver):
    """-> (scale-relative error, pointwise relative error over the significant points) of every captured evaluation."""
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = solver(**call["kwargs"])
    assert np.allclose(res[0], call["mu_arr"], rtol=0, atol=1e-14)
    fns = dict(zip(["flux_up", "flux_down", "u0", "u"], res[1:]))
    scale = max(max(np.max(np.abs(o), initial=0.0) for o in
                    (ev["out"] if isinstance(ev["out"], tuple) else (ev["out"],))) for ev in call["evals"])
    worst = worst_pw = 0.0
    for ev in call["evals"]:
        got = fns[ev["name"]](*ev["args"], **ev["kwargs"])
        gots = got if isinstance(got, tuple) else (got,)
        wants = ev["out"] if isinstance(ev["out"], tuple) else (ev["out"],)
        assert len(gots) == len(wants)
        for g, w in zip(gots, wants):
            assert np.shape(g) == np.shape(w), (ev["name"], np.shape(g), np.shape(w))
            assert np.all(np.isfinite(g))
            g, w = np.asarray(g, float), np.asarray(w, float)
            worst = max(worst, float(np.max(np.abs(g - w), initial=0.0)) / scale)
            if ev["name"] in ("u", "u0") and not ev["kwargs"].get("return_Fourier_error") and w.size and np.max(np.abs(w)) > 0:
                worst_pw = max(worst_pw, goldens.max_rel_err(g, w)[1])  # the north star's metric: intensities, pointwise
    return worst, worst_pw


@pytest.mark.parametrize("test_id", goldens.list_ids())
def test_reference_golden(amd, test_id):
    """Two metrics per captured call: max |d| over the radiation-field scale of the call (TOL: what two float64
    implementations of the same mathematics reach) and the north star's own -- max |dI| / |I_ref| over the points with
    |I_ref| > 1e-8 max |I_ref| -- against its 1e-6."""
    from conftest import record_parity
    tol = TOL_ILL if test_id in ILL_CONDITIONED else TOL
    worst = worst_pw = 0.0
    for call in goldens.load(test_id):
        a, b = _replay(call, amd.pydisort)
        worst, worst_pw = max(worst, a), max(worst_pw, b)
    pw_tol = PW_EXCEPT.get(test_id, PW_TOL)
    record_parity("golden/" + test_id, worst, worst_pw, tol, pw_tol, against="reference")


# The reference's own ill-conditioned test problems (omega = 1 - 1e-6: 1b, 1e, 2b, 2d, 3a, 3b, 4a, 5a; pydisotest/4_test.py:35-45,
# 5_test.py:53-63) pass test_reference_golden with little margin -- 4a at 9.49e-7 of the 1e-6 pointwise bound -- against a reference
# result that is itself ~1e-8 of the field scale off.  Round 6 arbitrates every one of them with a 40-digit solution
# (tools/hp_truth_case.py golden <id>; calls with NT_cor get the input-only correction terms added): the reference's captured
# result is 9.49e-7 (4a), 2.4e-7 (5a), 4.4e-8 (2d) pointwise from the truth -- the whole of the HIP-vs-reference distance is the
# reference's own error -- and the HIP path is held to the truth at (scale, pointwise) = 10 x what it measures there.
# Measured (profiles/r06_parity_report.json): HIP vs truth 3e-15 ... 9e-13 of the scale and 9e-15 ... 1.7e-11 pointwise on all
# eleven; 4a: 5.0e-14 / 2.0e-12 where the reference is 2.6e-8 / 9.49e-7.  Held at ten times the largest measured.
_ILL_TOL = (1e-11, 2e-10)
HP_TOL = {"8ARTS_A": (1e-9, 5e-6), **{k: _ILL_TOL for k in ("1b", "1e", "2b", "2d", "3a", "3b", "4a", "5a")},
          **{k: _ILL_TOL for k in ("2c", "4b", "5b")}}  # (the last three: well-conditioned siblings, the machinery's check)


@pytest.mark.parametrize("test_id", sorted(HP_TOL))
def test_golden_case_against_high_precision_truth(amd, test_id):
    """The reference-captured cases whose float64 reference result is itself the weak side -- 8ARTS_A (pointwise metric not
    held against the reference at all) and the eight omega = 1 - 1e-6 test problems (held against the reference at 1e-7 of the
    scale instead of 1e-9) -- against the 40-digit solution of the same inputs (tests/golden/hp/golden_<id>.npz,
    tools/hp_truth_case.py): HIP within 1e-9 of the field scale of the truth and within the north star's 1e-6 pointwise (8ARTS_A:
    5e-6, where the reference is 5.0e-5 from the truth; see the comment at PW_EXCEPT).  Where the reference sits is recorded."""
    from conftest import record_parity
    z = np.load(f"{goldens.HERE}/golden/hp/golden_{test_id}.npz")
    worst = worst_pw = ref_pw = ref_scale = 0.0
    for ci, call in enumerate(goldens.load(test_id)):
        if f"c{ci}.u" not in z.files:
            continue
        ev = next(e for e in call["evals"] if e["name"] == "u" and not e["kwargs"] and len(e["args"]) == 2)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            res = amd.pydisort(**call["kwargs"])
        got = res[4](*ev["args"])
        a, b = goldens.max_rel_err(got, z[f"c{ci}.u"])
        worst, worst_pw = max(worst, a), max(worst_pw, b)
        ra, rb = goldens.max_rel_err(ev["out"], z[f"c{ci}.u"])
        ref_scale, ref_pw = max(ref_scale, ra), max(ref_pw, rb)
        if f"c{ci}.flux_up" in z.files:  # (fixtures of round 6 carry the flux of the truth too)
            fu = np.atleast_1d(res[1](ev["args"][0]))
            assert np.max(np.abs(fu - z[f"c{ci}.flux_up"])) <= 1e-9 * np.max(np.abs(z[f"c{ci}.flux_up"])), test_id
    tol, pw_tol = HP_TOL[test_id]
    if test_id in PW_EXCEPT:  # (the exception exists because the reference itself is further than this from the truth)
        assert ref_pw > pw_tol
    record_parity("golden/" + test_id + " vs truth", worst, worst_pw, tol, pw_tol, against="40-digit truth",
                  reference_vs_truth_pointwise_rel=ref_pw, reference_vs_truth_scale_rel=ref_scale)


# ---- the reference's own pass criteria vs Fortran DISORT (pydisotest/*_test.py, e.g. 1_test.py:78-81)
STAMNES_IDS = [t for t in goldens.list_ids() if t not in ("11a", "8ARTS_A", "8ARTS_B", "9corrections", "Ia", "Ib", "Ic")]
NEAR_BEAM_DEG = {"3a": 10, "3b": 10, "4a": 10, "4b": 10, "4c": 10, "5a": 10, "5b": 10}  # e.g. 5_test.py:94-98


@pytest.mark.parametrize("test_id", STAMNES_IDS)
def test_stamnes_disort(amd, test_id):
    call = goldens.load(test_id)[0]
    kw = call["kwargs"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = amd.pydisort(**kw)
    mu_arr, flux_up, flux_down = res[0], res[1], res[2]
    u = res[4] if len(res) > 4 else None
    results = goldens.stamnes(test_id)
    reorder = np.argsort(mu_arr)
    mu_ro = mu_arr[reorder]
    mu0 = kw["mu0"]
    deg = NEAR_BEAM_DEG.get(test_id, 0)
    keep = np.abs(np.arccos(np.abs(mu_ro)) - np.arccos(mu0)) * 180 / pi > deg if deg else np.ones(len(mu_ro), bool)
    has_uu = "uu" in results.files and u is not None
    out = amd.subroutines.compare_with_stamnes(results, keep, reorder, flux_up, flux_down, u if has_uu else None)
    for diff, ratio in zip(out[0:6:2], out[1:6:2]):
        assert np.max(ratio[diff > 1e-3], initial=0) < 1e-3
    if has_uu:
        assert np.max(out[7][out[6] > 1e-3], initial=0) < 1e-2


def test_arts_a_thermal(amd):
    """8ARTS_A: 101 thermal-only 20-layer cases against the ARTS results at 1 % (8_test.py:277)."""
    arts = np.load(goldens.STAMNES_DIR + "/8ARTS_A_test.npy")
    got = np.empty(len(arts))
    for i, call in enumerate(goldens.load("8ARTS_A")):
        kw = call["kwargs"]
        u = amd.pydisort(**kw)[4]
        got[i] = u(np.atleast_1d(kw["tau_arr"]), 0.0).T[-1, -1]
    assert np.max(np.abs(got - arts) / arts) < 1e-2


# ---- BASELINE configs: reference-computed goldens for the first columns
@pytest.mark.parametrize("name,maker,kwargs", [
    ("cfg4", "cfg4_columns", {}),
    ("cfg3_big", "cfg3_columns", {"big": True}),
    ("cfg3_small", "cfg3_columns", {"big": False}),
    ("cfg5", "cfg5_columns", {}),
])
def test_synthetic_config_vs_reference(amd, name, maker, kwargs):
    """Reference-computed goldens of the first columns of every synthetic config (cfg4: 64 columns, cfg5: 8).
    cfg5 (64 streams, 50 layers, 64 modes) is given 5e-9 of the field scale: two float64 implementations differ by
    2e-9 there (it is the reference's own roundoff level at that size: its banded LU works on 3200 x 3200 systems)."""
    from conftest import record_parity
    from pydisort_amd import synthetic
    z = np.load(f"{goldens.HERE}/golden/synth/{name}.npz")
    ncol = int(z["ncol"])
    assert ncol == {"cfg4": 64, "cfg5": 8}.get(name, 4)
    tol = 5e-9 if name == "cfg5" else TOL
    cfg = getattr(synthetic, maker)(ncol, **kwargs)
    mu_arr, sol = amd.pydisort_batch(**cfg)
    tau = np.stack([z[f"c{i}.tau_pts"] for i in range(ncol)])
    u = sol.u(tau, z["phi"])
    fu = sol.flux_up(tau)
    fd, fdir = sol.flux_down(tau)
    u0 = sol.u0(tau)
    worst = worst_pw = 0.0
    for i in range(ncol):
        a, b = goldens.max_rel_err(u[i], z[f"c{i}.u"])
        worst, worst_pw = max(worst, a), max(worst_pw, b)
        scale = np.max(np.abs(z[f"c{i}.u"]))
        assert np.max(np.abs(u0[i] - z[f"c{i}.u0"])) / scale < tol
        fs = max(np.max(np.abs(z[f"c{i}.flux_up"])), np.max(np.abs(z[f"c{i}.flux_down_diffuse"])), 1e-300)
        assert np.max(np.abs(fu[i] - z[f"c{i}.flux_up"])) / fs < tol
        assert np.max(np.abs(fd[i] - z[f"c{i}.flux_down_diffuse"])) / fs < tol
        assert np.allclose(fdir[i], z[f"c{i}.flux_down_direct"], rtol=1e-12, atol=1e-300)
    record_parity("synthetic/" + name, worst, worst_pw, tol, PW_TOL, against="reference")
    assert worst < tol, name
    assert worst_pw < PW_TOL, name


SYNTH_TRUTHS = sorted(f[6:-4] for f in os.listdir(os.path.join(goldens.HERE, "golden", "hp")) if f.startswith("synth_"))


@pytest.mark.parametrize("key", SYNTH_TRUTHS)
def test_synthetic_column_against_high_precision_truth(amd, key):
    """Columns of the benchmark configs against their 40-digit solutions (tools/hp_truth_case.py synth <cfg>_<column>):
    cfg4 column 44 is the one whose beam is nearly resonant with an eigenvalue (mu0 = 0.912: the particular and the
    homogeneous solution cancel; it set the stop rule of the Jacobi sweeps in round 2), cfg5 column 0 is the stress config
    (64 streams, 50 layers, 64 modes, BDRF, thermal source), whose reference-computed golden is held to 5e-9 only because
    the reference's own roundoff is of that size there -- here the HIP path is held to the truth at 1e-9 / 1e-6 like every
    other case, and the reference's golden is measured against the truth too."""
    from conftest import record_parity
    from pydisort_amd import synthetic
    name, col = key.rsplit("_", 1)
    col = int(col)
    z = np.load(f"{goldens.HERE}/golden/hp/synth_{key}.npz")
    if name == "cfg4cloud":  # cfg4 with an omega = 1 - 1e-6 cloud layer in every column (the all-cloud leg of bench.py): the
        #                          mode-0 chain takes the register-resident pivoted elimination (GjPiv) throughout
        whole = synthetic.cfg4_cloud_columns(col + 1)
        cfg = {k: (v[col:col + 1] if isinstance(v, np.ndarray) and v.shape[:1] == (col + 1,) else v) for k, v in whole.items()}
    else:
        cfg = {"cfg4": synthetic.cfg4_columns, "cfg5": synthetic.cfg5_columns}[name](1, first=col)   # column c is always the same atmosphere
    _, sol = amd.pydisort_batch(**cfg)
    got = sol.u(z["tau"][None], z["phi"])[0]
    a, b = goldens.max_rel_err(got, z["u"])
    gpath = f"{goldens.HERE}/golden/synth/{name}.npz"
    g = np.load(gpath) if os.path.exists(gpath) else None
    ra, rb = goldens.max_rel_err(g[f"c{col}.u"], z["u"]) if g is not None and f"c{col}.u" in g.files else (np.nan, np.nan)
    if name == "cfg4cloud":  # (no reference golden: the fixture records the oracle's -- the reference's algorithm -- distance)
        ra, rb = float(z["oracle_u_scale_rel"]), float(z["oracle_u_pointwise_rel"])
    record_parity(f"synthetic/{key} vs truth", a, b, 1e-9, PW_TOL, against="40-digit truth", reference_vs_truth_scale_rel=ra,
                  reference_vs_truth_pointwise_rel=rb)
    sol.plan.close()


@pytest.mark.parametrize("tag", ["a", "b"])
def test_cfg2_literal_cloud_c1_at_32_streams(amd, tag):
    """BASELINE.json configs[1] as worded: Test Problem 5 -- Cloud C.1 phase function (300 moments, read from the
    fixture), one layer of optical depth 64, beam source -- at 32 streams with delta-M and the Nakajima-Tanaka
    corrections, against the reference run at that stream count (a: omega = 1 - 1e-6, b: omega = 0.9)."""
    from conftest import record_parity
    z = np.load(f"{goldens.HERE}/golden/synth/cfg2_q32_cloud_{tag}.npz")
    leg = z["Leg_coeffs_all"]
    assert leg.shape == (1, 300)
    kw = dict(tau_arr=np.array([64.0]), omega_arr=z["omega"], NQuad=32, Leg_coeffs_all=leg.copy(), mu0=1.0, I0=pi, phi0=pi,
              f_arr=np.array([leg[0, 32]]), NT_cor=True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = amd.pydisort(**kw)
    tol = TOL_ILL if tag == "a" else TOL
    a, b = goldens.max_rel_err(res[4](z["tau_pts"], z["phi"]), z["u"])
    record_parity("synthetic/cfg2_q32_cloud_" + tag, a, b, tol, PW_TOL, against="reference")
    assert a < tol and b < PW_TOL
    assert goldens.max_rel_err(res[3](z["tau_pts"]), z["u0"])[0] < tol
    fs = np.max(np.abs(z["flux_down_diffuse"]))
    assert np.max(np.abs(res[1](z["tau_pts"]) - z["flux_up"])) / fs < tol
    assert np.max(np.abs(res[2](z["tau_pts"])[0] - z["flux_down_diffuse"])) / fs < tol


@pytest.mark.parametrize("big", [True, False])
def test_cfg3_at_1024_columns(amd, big):
    """BASELINE.json configs[2] at its full batch size: Test-Problem-9-like atmospheres x 1024 columns (per-column
    perturbed single-scattering albedos, thermal + beam + Dirichlet sources, Lambertian surface), solved in one batch;
    a sample of columns against the oracle, every column against size-independent properties."""
    from conftest import record_parity
    from oracle import disort_oracle as O
    from pydisort_amd import synthetic
    C = 1024
    cfg = synthetic.cfg3_columns(C, big=big)
    _, sol = amd.pydisort_batch(**cfg)
    L = cfg["tau_arr"].shape[1]
    tau = np.concatenate((np.zeros((C, 1)), cfg["tau_arr"], 0.5 * cfg["tau_arr"][:, :1]), axis=1)
    phi = np.array([0.0, pi / 2, pi])
    u, u0 = sol.u(tau, phi), sol.u0(tau)
    fu, (fd, fdir) = sol.flux_up(tau), sol.flux_down(tau)
    assert np.all(np.isfinite(u)) and u.shape == (C, cfg["NQuad"], L + 2, 3)
    worst = worst_pw = 0.0
    for i in range(0, C, 93):
        kw = synthetic.column_kwargs(cfg, i)
        kw["BDRF_Fourier_modes"] = [0.5]
        ref = O.pydisort(**kw)
        a, b = goldens.max_rel_err(u[i], ref[4](tau[i], phi))
        worst, worst_pw = max(worst, a), max(worst_pw, b)
        assert np.allclose(fu[i], ref[1](tau[i]), rtol=1e-9)
    record_parity("synthetic/cfg3_%s_x1024" % ("big" if big else "small"), worst, worst_pw, TOL, PW_TOL)
    assert worst < TOL and worst_pw < PW_TOL
    # properties that hold for every column: Beer's law for the direct beam, fluxes = quadrature of the zeroth mode,
    # and a column's result does not depend on its batch (columns 500..515 solved again on their own, bit for bit)
    assert np.allclose(fdir, (cfg["I0"] * cfg["mu0"])[:, None] * np.exp(-tau / cfg["mu0"][:, None]), rtol=1e-13)
    N = cfg["NQuad"] // 2
    mu, w = sol.prep["mu"], sol.prep["W"]
    assert np.allclose(fu, 2 * pi * np.einsum("cit,i->ct", u0[:, :N], mu * w), rtol=1e-12)
    assert np.allclose(fd, 2 * pi * np.einsum("cit,i->ct", u0[:, N:], mu * w), rtol=1e-12)
    sub = {k: (v[500:516] if isinstance(v, np.ndarray) and v.shape[:1] == (C,) else v) for k, v in cfg.items()}
    _, sol2 = amd.pydisort_batch(**sub)
    assert np.array_equal(sol2.u(tau[500:516], phi), u[500:516])


def test_cfg4_batch_vs_oracle_256_columns(amd):
    """Same seeded inputs through the batched HIP path and the CPU oracle (columns 100..355)."""
    from oracle import disort_oracle as O
    from pydisort_amd import synthetic
    C = 256
    cfg = synthetic.cfg4_columns(C, first=100)
    mu_arr, sol = amd.pydisort_batch(**cfg)
    tau = np.concatenate((np.zeros((C, 1)), cfg["tau_arr"]), axis=1)
    phi = np.array([0.0, pi / 2, pi])
    u = sol.u(tau, phi)
    fu = sol.flux_up(tau)
    for i in range(0, C, 37):
        ref = O.pydisort(**synthetic.column_kwargs(cfg, i))
        want = ref[4](tau[i], phi)
        assert np.max(np.abs(u[i] - want)) / np.max(np.abs(want)) < TOL
        assert np.allclose(fu[i], ref[1](tau[i]), rtol=1e-9, atol=1e-12)
    assert sol.plan.max_sweeps() <= 12


def test_full_size_properties(amd):
    """Size-independent checks at a full-size batch (4096 cfg4 columns): results do not depend on the
    batch a column is solved in, are finite, and the direct beam obeys Beer's law exactly."""
    from pydisort_amd import synthetic
    C = 4096
    cfg = synthetic.cfg4_columns(C)
    _, sol = amd.pydisort_batch(**cfg)
    tau = np.concatenate((np.zeros((C, 1)), cfg["tau_arr"]), axis=1)
    phi = np.array([0.0, pi])
    u = sol.u(tau, phi)
    assert np.all(np.isfinite(u))
    fd, fdir = sol.flux_down(tau)
    assert np.allclose(fdir, (cfg["I0"] * cfg["mu0"])[:, None] * np.exp(-tau / cfg["mu0"][:, None]), rtol=1e-13)
    sub = {k: (v[1000:1016] if isinstance(v, np.ndarray) and v.shape[:1] == (C,) else v) for k, v in cfg.items()}
    _, sol2 = amd.pydisort_batch(**sub)
    u2 = sol2.u(tau[1000:1016], phi)
    assert np.array_equal(u2, u[1000:1016])  # bit-identical: no cross-column coupling, deterministic kernels
    # energy: net flux is non-increasing with depth for absorbing atmospheres (omega < 1, no thermal source)
    net = fd + fdir - sol.flux_up(tau)
    assert np.all(np.diff(net, axis=1) < 1e-9)


def test_baseline_batch_of_100000_columns_properties(amd):
    """BASELINE.json configs[3] at its literal size (10^5 cfg4 columns, NumPy in -> NumPy out through the windowed,
    pipelined path): everything finite, Beer's law for the direct beam, the net flux never grows with depth (omega < 1, no
    thermal source), and the 64 golden columns -- spliced into the batch 1 500 columns apart, so that they fall into
    different windows -- equal the reference's outputs and, bit for bit, what a 64-column call returns for them."""
    from pydisort_amd import synthetic
    C = 100_000
    cfg = synthetic.cfg4_columns_block(C, first=0)
    z = np.load(f"{goldens.HERE}/golden/synth/cfg4.npz")
    ncol = int(z["ncol"])
    gold = synthetic.cfg4_columns(ncol)
    at = 7 + 1500 * np.arange(ncol)
    for k, v in gold.items():
        if isinstance(v, np.ndarray):
            cfg[k][at] = v
    tau = np.concatenate((np.zeros((C, 1)), cfg["tau_arr"]), axis=1)
    phi = z["phi"][:3]
    res = amd.solve_columns_streamed(cfg, tau, phi, chunk_columns=2048)
    for k in ("u", "u0", "flux_up", "flux_down_diffuse", "flux_down_direct"):
        assert np.all(np.isfinite(res[k])), k
    assert np.allclose(res["flux_down_direct"], (cfg["I0"] * cfg["mu0"])[:, None] * np.exp(-tau / cfg["mu0"][:, None]), rtol=1e-13)
    net = res["flux_down_diffuse"] + res["flux_down_direct"] - res["flux_up"]
    assert np.all(np.diff(net, axis=1) < 1e-9)
    assert np.all(res["flux_up"][:, 0] > 0)  # (u itself may dip below zero: delta-M without the NT corrections)
    worst = 0.0
    for i in range(ncol):  # the golden points are the interfaces and the mid-layer points: compare at the interfaces
        pts = np.searchsorted(z[f"c{i}.tau_pts"], tau[at[i]])
        assert np.array_equal(z[f"c{i}.tau_pts"][pts], tau[at[i]])
        worst = max(worst, goldens.max_rel_err(res["u"][at[i]], z[f"c{i}.u"][:, pts, :3])[0])
    assert worst < TOL, worst
    small = amd.solve_columns_streamed(gold, tau[at], phi, chunk_columns=2048)
    assert np.array_equal(small["u"], res["u"][at]) and np.array_equal(small["flux_up"], res["flux_up"][at])


def test_plans_on_concurrent_host_threads(amd):
    """one host thread per plan, plans independent.  Eight threads create, solve, evaluate and
    close their own one-column plans at the same time (ctypes releases the GIL during the calls): every result is
    bit-identical to the single-threaded one, and an error raised in one thread (tau out of range) carries its own text."""
    import threading
    from pydisort_amd import synthetic
    cases = [synthetic.column_kwargs(synthetic.cfg4_columns(1, first=i), 0) for i in range(3)]
    cases += [goldens.load(n)[0]["kwargs"] for n in ("9c", "5a", "6d")]

    def run(kw):
        res = amd.pydisort(**kw)
        tau = np.concatenate(([0.0], np.atleast_1d(kw["tau_arr"])))
        out = res[4](tau, np.array([0.0, 1.0])) if len(res) > 4 else res[3](tau)  # (only_flux cases return no u)
        res[1].__self__.plan.close()
        return out

    bad, errors = [], []

    def worker(tid):
        try:
            for it in range(40):
                i = (it * 5 + tid) % len(cases)
                if not np.array_equal(run(cases[i]), want[i]):
                    bad.append((tid, it, i))
                if it % 10 == 0 and tid % 2 == 0:
                    res = amd.pydisort(**cases[3])
                    with pytest.raises(ValueError, match="tau"):
                        res[1](np.array([99.0]))
        except BaseException as e:  # noqa: BLE001 -- reported by the main thread
            errors.append(repr(e))

    with warnings.catch_warnings():  # (the filter list is process-wide: set once, around all threads)
        warnings.simplefilter("ignore")
        want = [run(kw) for kw in cases]
        threads = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    assert not errors, errors
    assert not bad, bad


def test_failure_in_a_higher_fourier_mode_leaves_fluxes_and_u0(amd):
    """A phase function whose 4-moment truncation is not positive (g = 0.89, 4 streams, no delta-M; found by
    tools/fuzz_parity.py, seed 1153 of the random family): the eigenvalue problem of Fourier mode 1 fails in one layer
    (the reference's sqrt gives NaN, _solve_for_gen_and_part_sols.py:186).  The reference returns valid fluxes and u0 --
    they come from mode 0 -- and a NaN intensity; so does the device path, except that `u` raises LinAlgError instead."""
    from oracle import disort_oracle as O
    import test_gpu_random_parity as R
    kw = R.make_case(1153)
    tau, phi = R.eval_points("random", 1153, kw)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = O.pydisort(**kw)
        got = amd.pydisort(**kw)
        assert not np.all(np.isfinite(ref[4](tau, phi)))  # the reference algorithm: NaN in u
    scale = np.max(np.abs(ref[3](tau)))
    assert np.max(np.abs(got[3](tau) - ref[3](tau))) / scale < 1e-9
    assert np.allclose(got[1](tau), ref[1](tau), rtol=1e-9)
    assert np.allclose(got[2](tau)[0], ref[2](tau)[0], rtol=1e-9, atol=1e-12 * scale)
    with pytest.raises(np.linalg.LinAlgError, match="m > 0"):
        got[4](tau, phi)
    assert np.allclose(got[1](tau), ref[1](tau), rtol=1e-9)  # (and again after the failed call)
    st = int(got[1].__self__.plan.column_status()[0])
    assert st & 0xFF == 0 and st & (4 << 8)  # non-finite eigenvalue, raised by a mode m > 0 only


def test_a_failed_column_does_not_touch_the_rest_of_the_batch(amd):
    """The same atmosphere between benign ones in a batch (4 streams: sixteen (column, mode) chains share a wavefront in
    the boundary-condition kernel -- a chain that has gone NaN used to write outside its group's LDS and spoil its
    neighbours, found by tools/fuzz_parity.py): the benign columns and the failed column's mode 0 equal the oracle,
    `column_status` names the column, numeric_errors="nan" returns the batch with that column's intensity NaN."""
    from oracle import disort_oracle as O
    import test_gpu_random_parity as R
    bad = R.make_case(1153)
    bad.pop("BDRF_Fourier_modes")
    good = dict(bad)
    leg = bad["Leg_coeffs_all"].copy()
    leg[1] = 0.6 ** np.arange(leg.shape[1])
    good["Leg_coeffs_all"] = leg
    tau, phi = R.eval_points("random", 1153, bad)
    order = [good, bad, good, good, bad]
    cfg = dict(tau_arr=np.stack([k["tau_arr"] for k in order]), omega_arr=np.stack([k["omega_arr"] for k in order]), NQuad=bad["NQuad"],
               Leg_coeffs_all=np.stack([k["Leg_coeffs_all"] for k in order]), mu0=np.full(5, bad["mu0"]), I0=np.full(5, bad["I0"]),
               phi0=np.full(5, bad["phi0"]), b_neg=bad["b_neg"])
    taub = np.tile(tau, (5, 1))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        rg, rb = O.pydisort(**good), O.pydisort(**bad)
        _, sol = amd.pydisort_batch(**cfg)
        u0 = sol.u0(taub)
        for i, k in enumerate(order):
            assert np.allclose(u0[i], (rb if k is bad else rg)[3](tau), rtol=1e-9, atol=1e-12), i
        with pytest.raises(np.linalg.LinAlgError, match=r"2 of 5 columns \(1, 4\)"):
            sol.u(taub, phi)
        st = sol.plan.column_status()
        assert np.array_equal(st != 0, [False, True, False, False, True]) and np.all(st[[1, 4]] & 0xFF == 0)
        _, soln = amd.pydisort_batch(numeric_errors="nan", **cfg)
        u = soln.u(taub, phi)
        assert np.all(np.isnan(u[[1, 4]])) and np.allclose(u[[0, 2, 3]], rg[4](tau, phi), rtol=1e-9, atol=1e-12)
        assert np.allclose(soln.flux_up(taub)[1], rb[1](tau), rtol=1e-9)  # fluxes of the failed column: mode 0, valid
        # the last Fourier mode (return_Fourier_error) of the healthy columns survives the failure of the others: it used to be
        # left unwritten -- uninitialised host memory -- when the status check returned first (round-3 advisor finding)
        phi1 = np.atleast_1d(phi)
        ev = soln.plan.evaluate(taub, phi1, want=("u", "ulast"))
        _, solg = amd.pydisort_batch(**{k: (v[[0, 2, 3]] if isinstance(v, np.ndarray) and v.shape[:1] == (5,) else v) for k, v in cfg.items()})
        want_last = solg.plan.evaluate(taub[[0, 2, 3]], phi1, want=("ulast",))["ulast"]
        assert np.array_equal(ev["ulast"][[0, 2, 3]], want_last) and np.all(np.isnan(ev["ulast"][[1, 4]]))


def test_invalidate_tables_and_window_budget(amd):
    """Round 4 API: (1) rtd_plan_invalidate_tables declares the resident inputs new -- the next run recomputes the per-column
    Legendre tables at -mu0 and the attenuations and returns the same bits, on one-window and on pipelined plans, interleaved
    with cached runs; (2) an explicit work_columns is honoured while it fits RTD_WORK_BYTES and shrunk otherwise (subprocess:
    the variable is read at plan creation), with bit-equal results."""
    import subprocess
    import sys
    from pydisort_amd import synthetic
    C = 40
    cfg = synthetic.cfg4_columns(C, L=6, NQuad=16)
    tau = np.concatenate((np.zeros((C, 1)), cfg["tau_arr"]), axis=1)
    phi = np.array([0.0, 1.3])
    ref = None
    for win in (0, 16):
        _, sol = amd.pydisort_batch(work_columns=win, _defer_solve=True, **cfg)
        plan = sol.plan
        plan.set_eval_points(tau, phi)
        plan.run()
        a = plan.fetch()
        plan.invalidate_tables()
        plan.run()
        plan.run()  # (cached again)
        plan.invalidate_tables()
        plan.run()
        b = plan.fetch()
        for k in a:
            assert np.array_equal(a[k], b[k]), (win, k)
        if ref is None:
            ref = a
        for k in a:
            assert np.array_equal(a[k], ref[k]), (win, k)
        plan.close()
    code = ("import sys, numpy as np; sys.path[:0] = [%r, %r]; import pydisort_amd; from pydisort_amd import synthetic\n"
            "cfg = synthetic.cfg4_columns(40, L=6, NQuad=16)\n"
            "_, sol = pydisort_amd.pydisort_batch(work_columns=32, _defer_solve=True, **cfg)\n"
            "tau = np.concatenate((np.zeros((40, 1)), cfg['tau_arr']), axis=1)\n"
            "sol.plan.set_eval_points(tau, np.array([0.0, 1.3])); sol.plan.run(); r = sol.plan.fetch()\n"
            "print(sol.plan.windows()[0], repr(float(r['u'].sum())))\n") % (ROOT_DIR, os.path.join(ROOT_DIR, "pythonic-disort_amd"))
    outs = {}
    for budget in ("", "200000"):  # no variable: 32 columns per window fit; 200 KB: a 16-stream 6-layer column needs ~90 KB -> 1 ... 2 columns
        env = dict(os.environ)
        env.pop("RTD_WORK_BYTES", None)
        if budget:
            env["RTD_WORK_BYTES"] = budget
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[budget] = r.stdout.strip().splitlines()[-1].split()
    assert int(outs[""][0]) == 32 and 1 <= int(outs["200000"][0]) < 32, outs
    assert outs[""][1] == outs["200000"][1]  # same bits whatever the windowing


@pytest.mark.parametrize("tool,count", [("fuzz_batch.py", "120"), ("fuzz_plan_reuse.py", "80")])
def test_random_sweeps_of_the_batch_entry_points_find_nothing(tool, count):
    """tools/fuzz_batch.py (a batch = its columns one by one = its windowed plan = the streamed raw path = the oracle, over
    random stream counts, sources, windows and evaluation points) and tools/fuzz_plan_reuse.py (random call sequences on one
    long-lived plan against fresh plans: cached tables, stream forks, hand-off slots, status words), a few seconds each."""
    import subprocess
    import sys
    root = os.path.dirname(goldens.HERE)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", tool), count], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert ", 0 findings" in r.stdout.strip().splitlines()[-1], r.stdout[-3000:]


def test_a_failed_runtime_call_does_not_poison_the_next_one(amd):
    """A plan on a device that does not exist fails with the HIP error text -- and leaves nothing behind: HIP keeps the code
    of a failed call as the thread's "last error", which the launch check of the next, valid, solve used to report as its own
    ("kernel launch: invalid device ordinal")."""
    kw = goldens.load("9c")[0]["kwargs"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with pytest.raises(RuntimeError, match="invalid device ordinal"):
            amd.pydisort(device=99, **kw)
        res = amd.pydisort(**kw)
        want = amd.pydisort(**kw)[1](0.5)
    assert np.array_equal(res[1](0.5), want)


def test_tensors_match_oracle_invariants(amd):
    """The exported reference-layout tensors: K sorted, B, and the gauge-invariant product GC exp(K dtau)."""
    from oracle import disort_oracle as O
    call = goldens.load("9c")[0]
    kw = call["kwargs"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = amd.pydisort(**kw)
    plan = res[1].__self__.plan
    t = plan.tensors(0)
    p = O.prepare(**kw)
    sol = O.Solution(p)
    assert np.allclose(np.sort(t["K"], axis=-1), np.sort(sol.K, axis=-1), rtol=1e-10)
    assert np.allclose(t["B"], sol.B, rtol=1e-9, atol=1e-13)
    e = np.exp(-np.abs(sol.K) * 0.3)
    assert np.allclose(np.einsum("mlij,mlj->mli", t["GC"], np.exp(-np.abs(t["K"]) * 0.3)),
                       np.einsum("mlij,mlj->mli", sol.GC, e), rtol=1e-8, atol=1e-12)


def test_tau_out_of_range_raises(amd):
    kw = goldens.load("1a")[0]["kwargs"]
    res = amd.pydisort(**kw)
    with pytest.raises(ValueError):
        res[1](np.array([0.0, 99.0]))
    with pytest.raises(ValueError):
        res[4](-0.1, 0.0)


def test_rccl_allgather_single_rank(amd):
    """The one collective of the path (ncclAllGather of the flux results) with a 1-rank communicator."""
    from pydisort_amd import synthetic
    from pydisort_amd._engine import Plan
    Plan.comm_preload()
    cfg = synthetic.cfg4_columns(8, L=5, NQuad=8)
    _, sol = amd.pydisort_batch(**cfg)
    plan = sol.plan
    tau = np.concatenate((np.zeros((8, 1)), cfg["tau_arr"]), axis=1)
    plan.set_eval_points(tau, np.array([0.0, 1.0]))
    plan.run()
    plan.comm_init(Plan.comm_unique_id(), 0, 1)
    plan.allgather_fluxes()
    got = plan.fetch_gathered()
    res = plan.fetch()
    assert got.shape == (1, 3, 8, 6)
    assert np.array_equal(got[0, 0], res["flux_up"]) and np.array_equal(got[0, 1], res["flux_down_diffuse"])
    assert np.array_equal(got[0, 2], res["flux_down_direct"])
    # the collective of the mode-shard partition (ncclAllReduce, sum) with the same 1-rank communicator: identity
    plan.allreduce_results()
    again = plan.fetch()
    for k in ("u", "u0", "flux_up", "flux_down_diffuse", "flux_down_direct"):
        assert np.array_equal(again[k], res[k]), k


def test_batch_nt_corrections_match_single_column_reference_path(amd):
    """pydisort_batch(NT_cor=True) == per-column drop-in pydisort(NT_cor=True) == oracle with NT."""
    from oracle import disort_oracle as O
    from pydisort_amd import synthetic
    C = 3
    cfg = synthetic.cfg4_columns(C, L=5, NQuad=16)
    k = np.arange(40)
    g = cfg["Leg_coeffs_all"][:, :, 1]
    cfg["Leg_coeffs_all"] = g[:, :, None] ** k[None, None, :]   # more moments than NLeg -> corrections active
    _, sol = amd.pydisort_batch(NT_cor=True, **cfg)
    tau = np.concatenate((np.zeros((C, 1)), cfg["tau_arr"] * 0.999), axis=1)
    phi = np.array([0.1, 2.0, 4.0])
    u = sol.u(tau, phi)
    for i in range(C):
        kw = synthetic.column_kwargs(cfg, i)
        want = O.pydisort(NT_cor=True, **kw)[4](tau[i], phi)
        assert np.max(np.abs(u[i] - want)) / np.max(np.abs(want)) < 1e-9
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            got1 = amd.pydisort(NT_cor=True, **kw)[4](tau[i], phi)
        assert np.max(np.abs(got1 - u[i])) / np.max(np.abs(want)) < 1e-12


@pytest.mark.parametrize("name", ["cfg1_q4", "cfg2_q32"])
def test_baseline_literal_configs(amd, name):
    """BASELINE.json configs[0]/[1] as literally worded (4-stream TP1, 32-stream TP5-like with NT): reference goldens."""
    from pydisort_amd import synthetic
    kw, tau_pts = synthetic.literal_cases()[name]
    z = np.load(f"{goldens.HERE}/golden/synth/{name}.npz")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mu_arr, Fp, Fm, u0, u = amd.pydisort(**kw)
    scale = np.max(np.abs(z["u"]))
    assert np.max(np.abs(u(tau_pts, z["phi"]) - z["u"])) / scale < TOL
    assert np.max(np.abs(u0(tau_pts) - z["u0"])) / scale < TOL
    assert np.allclose(Fp(tau_pts), z["flux_up"], rtol=1e-9, atol=1e-12 * scale)
    fd = Fm(tau_pts)
    assert np.allclose(fd[0], z["flux_down_diffuse"], rtol=1e-9, atol=1e-11 * scale)
    assert np.allclose(fd[1], z["flux_down_direct"], rtol=1e-12, atol=1e-300)


@pytest.mark.parametrize("name", ["q72", "q96", "q128"])
def test_beyond_64_streams_vs_reference(amd, name):
    """72 / 96 / 128 streams -- the reference has no cap on NQuad (pydisort.py:258-264); these sizes run on the NP = 64 instances
    (one eigenproblem per wavefront; a boundary-condition chain per workgroup of four wavefronts, csrc/rtd_bc_wide.hip; one chain per
    wavefront on the row-per-lane kernels under RTD_BC_WIDE_V1) -- against the reference's own outputs
    (tests/golden/synth/q*.npz, make_synthetic_goldens.py).  Tolerances (round 6: ten times what is measured, not 1e-7 / 1e-6):
    WIDE_TOL = 1e-8 of the field scale (measured 8e-11 ... 7.3e-10; the reference's float64 algorithm and its restatement in
    oracle/ differ by 3e-10 from each other here), WIDE_PW_TOL = 4e-7 pointwise (measured 3e-10 ... 3.5e-8)."""
    from conftest import record_parity
    from pydisort_amd import synthetic
    kw, tau_pts = synthetic.many_stream_cases()[name]
    z = np.load(f"{goldens.HERE}/golden/synth/{name}.npz")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mu_arr, Fp, Fm, u0, u = amd.pydisort(**kw)
    a, b = goldens.max_rel_err(u(tau_pts, z["phi"]), z["u"])
    record_parity("synthetic/" + name, a, b, WIDE_TOL, WIDE_PW_TOL, against="reference")
    assert a < WIDE_TOL and b < WIDE_PW_TOL
    scale = np.max(np.abs(z["u"]))
    assert np.max(np.abs(u0(tau_pts) - z["u0"])) / scale < WIDE_TOL
    assert np.allclose(Fp(tau_pts), z["flux_up"], rtol=1e-8, atol=1e-10 * scale)
    fd = Fm(tau_pts)
    assert np.allclose(fd[0], z["flux_down_diffuse"], rtol=1e-8, atol=1e-10 * scale)
    assert np.allclose(fd[1], z["flux_down_direct"], rtol=1e-12, atol=1e-300)
    assert Fp.__self__.plan.max_sweeps() <= 14


@pytest.mark.parametrize("name", ["q96_L20", "q72_L50", "q128_L50"])
def test_timed_many_stream_workloads_at_full_depth_vs_reference(amd, name):
    """The 66 ... 128-stream workloads that bench.py / tools/many_stream_timing.py time -- 96 x 20 x 48, 72 x 50 x 36 and
    128 streams x 50 layers x 64 modes -- at their full depth against the reference's own outputs for the first columns of those
    very batches (tests/golden/synth/q*_L*.npz; round-4 verdict: the only 128-stream golden was two layers deep): every
    interface and mid-layer point, four azimuths, u0 and the fluxes; the golden columns sit in a batch with others behind them."""
    from conftest import record_parity
    from pydisort_amd import synthetic
    maker_kw, nf, ncol = synthetic.many_stream_deep_cases()[name]
    z = np.load(f"{goldens.HERE}/golden/synth/{name}.npz")
    C = ncol + 3
    cfg = synthetic.cfg4_columns(C, **maker_kw)
    cfg["NFourier"] = nf
    _, sol = amd.pydisort_batch(**cfg)
    npts = len(z["c0.tau_pts"])
    tau = np.stack([z[f"c{i}.tau_pts"] if i < ncol else np.linspace(0.0, cfg["tau_arr"][i, -1], npts) for i in range(C)])
    u, u0, fu, (fdd, fdir) = sol.u(tau, z["phi"]), sol.u0(tau), sol.flux_up(tau), sol.flux_down(tau)
    worst = worst_pw = 0.0
    for i in range(ncol):
        a, b = goldens.max_rel_err(u[i], z[f"c{i}.u"])
        worst, worst_pw = max(worst, a), max(worst_pw, b)
        scale = np.max(np.abs(z[f"c{i}.u"]))
        assert np.max(np.abs(u0[i] - z[f"c{i}.u0"])) / scale < WIDE_TOL
        assert np.allclose(fu[i], z[f"c{i}.flux_up"], rtol=1e-8, atol=1e-10 * scale)
        assert np.allclose(fdd[i], z[f"c{i}.flux_down_diffuse"], rtol=1e-8, atol=1e-10 * scale)
        assert np.allclose(fdir[i], z[f"c{i}.flux_down_direct"], rtol=1e-12, atol=1e-300)
    record_parity("synthetic/" + name, worst, worst_pw, WIDE_TOL, WIDE_PW_TOL, against="reference")
    assert worst < WIDE_TOL and worst_pw < WIDE_PW_TOL
    assert sol.plan.max_sweeps() <= 14
    sol.plan.close()


def test_beyond_64_streams_feature_paths(amd):
    """The other entry points at more than 64 streams: Nakajima-Tanaka corrections, antiderivatives and the Fourier-error
    output at 80 streams against the oracle; a batch of 96-stream columns against per-column calls (bit-identical), as a
    three-window plan (bit-identical) and through the raw-input streamed path; callable BDRF modes at 72 streams."""
    from oracle import disort_oracle as O
    from pydisort_amd import synthetic
    phi = np.array([0.0, 1.0, 3.0])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        kw = dict(tau_arr=np.array([1.0, 3.0]), omega_arr=np.array([0.9, 0.8]), NQuad=80, Leg_coeffs_all=np.tile(0.85 ** np.arange(200), (2, 1)),
                  mu0=0.5, I0=pi, phi0=0.0, f_arr=np.full(2, 0.85**80), NT_cor=True, NFourier=6)
        got, ref = amd.pydisort(**kw), O.pydisort(**kw)
        tau = np.array([0.0, 0.4, 1.0, 2.0, 3.0])
        scale = np.max(np.abs(ref[4](tau, phi)))
        assert np.max(np.abs(got[4](tau, phi) - ref[4](tau, phi))) / scale < 1e-8
        assert np.max(np.abs(got[4](tau, phi, True) - ref[4](tau, phi, True))) / scale < 1e-8
        assert np.max(np.abs(got[4](tau, phi, False, True)[1] - ref[4](tau, phi, False, True)[1])) < 1e-6
        assert np.max(np.abs(got[3](tau, True) - ref[3](tau, True))) / scale < 1e-8
        assert np.max(np.abs(got[1](tau, True) - ref[1](tau, True))) / scale < 1e-8
        C = 5
        cfg = synthetic.cfg4_columns(C, L=3, NQuad=96)
        cfg["NFourier"] = 4
        _, sol = amd.pydisort_batch(**cfg)
        taub = np.concatenate((np.zeros((C, 1)), cfg["tau_arr"]), axis=1)
        ub = sol.u(taub, phi)
        for i in range(C):
            kwi = synthetic.column_kwargs(cfg, i)
            kwi["NFourier"] = 4
            assert np.array_equal(amd.pydisort(**kwi)[4](taub[i], phi), ub[i])
        ro = O.pydisort(**dict(synthetic.column_kwargs(cfg, 0), NFourier=4))
        assert np.max(np.abs(ro[4](taub[0], phi) - ub[0])) / np.max(np.abs(ub[0])) < 1e-8
        _, sol3 = amd.pydisort_batch(work_columns=2, **cfg)
        assert sol3.plan.windows()[1] == 3 and np.array_equal(sol3.u(taub, phi), ub)
        res = amd.solve_columns_streamed(cfg, taub, phi, chunk_columns=4)
        assert np.max(np.abs(res["u"] - ub)) / np.max(np.abs(ub)) < 1e-12
        kw = dict(tau_arr=np.array([0.5]), omega_arr=np.array([0.8]), NQuad=72, Leg_coeffs_all=np.array([0.7 ** np.arange(73)]), mu0=0.6,
                  I0=1.0, phi0=0.0, NFourier=3,
                  BDRF_Fourier_modes=[lambda mu, nmup: 0.2 + 0.1 * np.outer(mu, nmup), lambda mu, nmup: 0.05 * np.outer(mu, nmup), 0.01])
        got, ref = amd.pydisort(**kw), O.pydisort(**kw)
        tau = np.array([0.0, 0.25, 0.5])
        assert np.max(np.abs(got[4](tau, phi) - ref[4](tau, phi))) / np.max(np.abs(ref[4](tau, phi))) < 1e-8


def test_a_failed_column_at_72_streams_does_not_touch_the_rest_of_the_batch(amd):
    """Beyond 64 streams a boundary-condition chain is a workgroup of four wavefronts (csrc/rtd_bc_wide.hip) whose pivot search
    finds nothing in a chain that has gone NaN: a column whose 73-moment truncation of a g = 0.99 Henyey-Greenstein phase function
    is not positive (no delta-M: the Cholesky factorisation of the eigen stage fails, as the reference's sqrt does,
    _solve_for_gen_and_part_sols.py:186) sits between benign columns; they must equal their own one-column solves bit for bit,
    `column_status` must name the failed ones, numeric_errors="nan" must return the batch with those columns NaN."""
    NQ = 72
    def column(g):
        return dict(tau_arr=np.array([0.5, 1.0, 3.0]), omega_arr=np.array([0.8, 0.9, 0.7]), NQuad=NQ,
                    Leg_coeffs_all=np.stack([0.6 ** np.arange(NQ + 1), g ** np.arange(NQ + 1), 0.5 ** np.arange(NQ + 1)]),
                    mu0=0.6, I0=1.0, phi0=0.0, NFourier=6)
    good, bad = column(0.9), column(0.99)
    order = [good, bad, good, bad, good]
    cfg = dict(tau_arr=np.stack([k["tau_arr"] for k in order]), omega_arr=np.stack([k["omega_arr"] for k in order]), NQuad=NQ,
               Leg_coeffs_all=np.stack([k["Leg_coeffs_all"] for k in order]), mu0=np.full(5, 0.6), I0=np.full(5, 1.0),
               phi0=np.full(5, 0.0), NFourier=6)
    tau, phi = np.array([0.0, 0.7, 3.0]), np.array([0.0, 1.0])
    taub = np.tile(tau, (5, 1))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        one = amd.pydisort(**good)
        want = one[4](tau, phi)
        with pytest.raises(np.linalg.LinAlgError):
            amd.pydisort(**bad)[4](tau, phi)
        _, sol = amd.pydisort_batch(**cfg)
        with pytest.raises(np.linalg.LinAlgError, match=r"2 of 5 columns \(1, 3\)"):
            sol.u(taub, phi)
        st = sol.plan.column_status()
        assert np.array_equal(st != 0, [False, True, False, True, False])
        _, soln = amd.pydisort_batch(numeric_errors="nan", **cfg)
        u = soln.u(taub, phi)
    assert np.all(np.isnan(u[[1, 3]]))
    for i in (0, 2, 4):
        assert np.array_equal(u[i], want), i


EDGE_CASES = {
    # two streams (N = 1, padded to 4 lanes), single layer
    "two_streams": dict(tau_arr=0.7, omega_arr=0.6, NQuad=2, Leg_coeffs_all=np.array([1.0, 0.3, 0.1]), mu0=0.4, I0=2.0, phi0=1.0),
    # fewer Legendre moments / Fourier modes than streams, matrix-form Dirichlet BCs, mixed zero-omega layers
    "nleg_lt_nquad": dict(tau_arr=np.array([0.2, 0.9, 1.0, 4.0]), omega_arr=np.array([0.0, 0.8, 0.0, 0.3]), NQuad=12,
                          Leg_coeffs_all=np.tile(0.6 ** np.arange(13), (4, 1)), mu0=0.9, I0=1.0, phi0=0.0, NLeg=7, NFourier=5,
                          b_pos=np.outer(np.linspace(0.1, 0.6, 6), [1.0, 0.5, 0.25, 0.1, 0.05]),
                          b_neg=np.linspace(0.3, 0.1, 6)),
    # no sources except a thermal cubic; BDRF given as a callable mode plus a scalar mode
    "thermal_cubic_bdrf": dict(tau_arr=np.array([0.5, 1.5]), omega_arr=np.array([0.4, 0.7]), NQuad=8,
                               Leg_coeffs_all=np.tile(0.5 ** np.arange(9), (2, 1)), mu0=0.0, I0=0.0, phi0=0.0,
                               s_poly_coeffs=np.array([[1.0, 0.5, 0.2, 0.05], [2.0, -0.3, 0.1, 0.01]]),
                               BDRF_Fourier_modes=[lambda mu, nmup: 0.2 + 0.1 * np.outer(mu, nmup), 0.05]),
    # very thick and very thin layers next to each other, near-grazing beam
    "thick_thin": dict(tau_arr=np.cumsum([1e-6, 40.0, 1e-4, 5.0, 1e-6]), omega_arr=np.array([0.9, 0.99, 0.2, 0.95, 0.5]),
                       NQuad=16, Leg_coeffs_all=np.tile(0.8 ** np.arange(17), (5, 1)), mu0=0.05, I0=3.0, phi0=2.0,
                       f_arr=np.full(5, 0.8**16), b_pos=0.2),
    "only_flux": dict(tau_arr=np.array([1.0, 2.0, 3.5]), omega_arr=np.array([0.5, 0.9, 0.7]), NQuad=16,
                      Leg_coeffs_all=np.tile(0.7 ** np.arange(17), (3, 1)), mu0=0.6, I0=1.5, phi0=0.3, only_flux=True,
                      BDRF_Fourier_modes=[0.3]),
    # a deep chain for the fused boundary-condition kernel: 150 layers, 32 streams, alternating thick / thin layers,
    # thermal + beam + Lambertian surface (the layer-sorted eigen stage sees 150 / 4 = 38 wavefronts per mode)
    "deep_chain_32": dict(tau_arr=np.cumsum(np.where(np.arange(150) % 3 == 0, 2.0, 0.02)),
                          omega_arr=0.3 + 0.65 * (np.arange(150) % 7) / 6.0, NQuad=32,
                          Leg_coeffs_all=(0.2 + 0.6 * (np.arange(150) % 5) / 4.0)[:, None] ** np.arange(33)[None, :],
                          mu0=0.35, I0=1.0, phi0=0.5, NFourier=6, b_neg=0.1,
                          s_poly_coeffs=np.tile(np.array([[0.5, 0.01]]), (150, 1)), BDRF_Fourier_modes=[0.4]),
    # 30 streams (N = 15 padded to 16 lanes in the MFMA layout), delta-M, two BDRF modes
    "padded_30": dict(tau_arr=np.array([0.1, 1.1, 1.15, 6.0]), omega_arr=np.array([0.99, 0.3, 0.0, 0.9]), NQuad=30,
                      Leg_coeffs_all=np.array([0.85, 0.5, 0.2, 0.7])[:, None] ** np.arange(34)[None, :], mu0=0.8, I0=2.0,
                      phi0=0.0, f_arr=np.array([0.85, 0.5, 0.2, 0.7]) ** 30,
                      BDRF_Fourier_modes=[lambda mu, nmup: 0.3 * (1 + 0.4 * np.outer(mu, nmup)),
                                          lambda mu, nmup: 0.1 * np.outer(np.sqrt(1 - mu**2), np.sqrt(1 - np.asarray(nmup) ** 2))]),
    # 64 streams (the largest size of the tuned kernels): fused eigen kernel at 32 lanes per problem, row-per-lane BC kernels
    "max_streams_64": dict(tau_arr=np.array([0.5, 2.5, 3.0]), omega_arr=np.array([0.9, 0.6, 0.95]), NQuad=64,
                           Leg_coeffs_all=np.array([0.9, 0.4, 0.8])[:, None] ** np.arange(66)[None, :], mu0=0.45, I0=1.0,
                           phi0=1.0, f_arr=np.array([0.9, 0.4, 0.8]) ** 64, NFourier=5, b_pos=0.3,
                           s_poly_coeffs=np.array([[0.2, 0.1], [0.4, 0.0], [0.1, 0.05]])),
}


@pytest.mark.parametrize("name", list(EDGE_CASES))
def test_edge_cases_vs_oracle(amd, name):
    """Shapes and corner sizes the reference accepts: N = 1, NLeg/NFourier < NQuad, matrix BCs, omega = 0 layers,
    high-order thermal polynomials, callable BDRF modes, 1e-6 ... 40 layer thicknesses, only_flux; evaluated at 0,
    every interface, tau_L and interior points (scalar and array tau)."""
    from oracle import disort_oracle as O
    kw = EDGE_CASES[name]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = amd.pydisort(**kw)
        ref = O.pydisort(**kw)
    assert len(got) == len(ref) and np.array_equal(got[0], ref[0])
    tau_arr = np.atleast_1d(kw["tau_arr"])
    tau = np.unique(np.concatenate(([0.0], tau_arr, 0.5 * tau_arr, [0.999999 * tau_arr[-1]])))
    scale = max(np.max(np.abs(ref[3](tau))), 1e-300)
    assert np.max(np.abs(got[3](tau) - ref[3](tau))) / scale < TOL
    assert np.allclose(got[1](tau), ref[1](tau), rtol=1e-8, atol=1e-10 * scale)
    for a, b in zip(got[2](tau), ref[2](tau)):
        assert np.allclose(a, b, rtol=1e-8, atol=1e-10 * scale)
    assert np.shape(got[1](0.3 * tau_arr[-1])) == np.shape(ref[1](0.3 * tau_arr[-1])) == ()
    if len(got) > 4:
        phi = np.array([0.0, 1.3, 5.0])
        assert np.max(np.abs(got[4](tau, phi) - ref[4](tau, phi))) / scale < TOL
        assert np.shape(got[4](tau[1], 0.5)) == np.shape(ref[4](tau[1], 0.5))
        e1 = got[4](tau, phi, False, True)[1]
        e2 = ref[4](tau, phi, False, True)[1]
        assert abs(e1 - e2) < 1e-6 * max(1.0, abs(e2))


def test_streamed_chunks_equal_one_batch(amd):
    """solve_columns_streamed (one plan, several windows incl. a shorter last one, device-side preparation) == one batch."""
    from pydisort_amd import synthetic
    C = 50
    cfg = synthetic.cfg4_columns(C, L=6, NQuad=16)
    tau = np.concatenate((np.zeros((C, 1)), cfg["tau_arr"]), axis=1)
    phi = np.array([0.0, 2.0])
    _, sol = amd.pydisort_batch(**cfg)
    want_u, want_f = sol.u(tau, phi), sol.flux_up(tau)
    got = amd.solve_columns_streamed(cfg, tau, phi, chunk_columns=16)
    # (the streamed form prepares its inputs on the device, the batch above in NumPy: same mathematics, other rounding;
    #  bit-equality of windowed and one-window plans is test_windowed_plan_equals_single_window)
    assert np.max(np.abs(got["u"] - want_u)) <= 1e-12 * np.max(np.abs(want_u))
    assert np.allclose(got["flux_up"], want_f, rtol=1e-12)


# ---- BDRF Fourier modes formed on the device ---------------------------------------
def _hapke_case(test_id):
    call = goldens.load(test_id)[0]
    kw = call["kwargs"]
    return call, kw


@pytest.mark.gpu
@pytest.mark.parametrize("test_id", ["6d", "6e", "6f", "6g", "6h"])
def test_device_bdrf_fourier_integration_hapke(amd, test_id):
    """Hapke surface of DISORT's test problem 6: the reference integrates every Fourier mode with scipy's adaptive
    quad_vec on the host (pydisotest/6_test.py:194-201; its tables are in the golden file); here the reflectance is
    sampled at 4096 azimuths and the device forms the modes.  Modes agree with the captured tables to the trapezoid
    error of the opposition cusp (mu = mu', dphi = pi: < 1e-5 abs), the fluxes of the solve to 1e-5 relative."""
    from pydisort_amd import subroutines as sub
    call, kw = _hapke_case(test_id)
    NQuad, N = kw["NQuad"], kw["NQuad"] // 2
    mu0 = float(kw["mu0"])
    rho = sub.Hapke_BDRF(1.0, 0.06, 0.6)
    rho_qq, rho_q0 = sub.sample_BDRF(rho, NQuad, mu0, nphi=4096)
    nb = len(kw["BDRF_Fourier_modes"])
    tau = np.atleast_1d(np.asarray(kw["tau_arr"], float))[None, :]
    L = tau.shape[1]
    Leg = np.asarray(kw["Leg_coeffs_all"], float)
    Leg = np.broadcast_to(Leg if Leg.ndim == 2 else Leg[None, :], (L, Leg.shape[-1]))[None]
    sp = kw.get("s_poly_coeffs")
    sp = None if sp is None or np.size(sp) == 0 else np.asarray(sp, float).reshape(1, L, -1)
    only_flux = bool(kw.get("only_flux", False))
    mu_arr, sol = amd.pydisort_batch(tau, np.atleast_1d(np.asarray(kw["omega_arr"], float))[None, :], NQuad, Leg,
                                     [mu0], [float(kw["I0"])], [float(kw["phi0"])], only_flux=only_flux,
                                     b_pos=np.asarray(kw.get("b_pos", 0), float) if np.ndim(kw.get("b_pos", 0)) == 0 else np.asarray(kw["b_pos"], float).reshape(1, N, -1)[:, :, 0],
                                     b_neg=np.asarray(kw.get("b_neg", 0), float) if np.ndim(kw.get("b_neg", 0)) == 0 else np.asarray(kw["b_neg"], float).reshape(1, N, -1)[:, :, 0],
                                     s_poly_coeffs=sp, bdrf_samples=(rho_qq[None], rho_q0[None]), NBDRF=min(nb, 1 if only_flux else NQuad))
    # the device tables against the reference's quad_vec tables
    plan = sol.plan
    for m in range(plan.prep["NBDRF"]):
        f = kw["BDRF_Fourier_modes"][m]
        # host restatement of the same trapezoid sum against the reference's adaptive integral (cusp error)
        p = np.arange(4096)
        q_host = (1 if m == 0 else 2) / 4096 * np.einsum("ijp,p->ij", rho_qq, np.cos(2 * np.pi * m * p / 4096))
        assert np.max(np.abs(q_host - f.tab)) < 1e-5
    # fluxes of the full solve against the reference's outputs
    checked = 0
    for ev in call["evals"]:
        if ev["name"] not in ("flux_up", "flux_down") or ev["kwargs"]:
            continue
        t = np.atleast_1d(np.asarray(ev["args"][0], float))[None, :]
        if ev["name"] == "flux_up":
            got = sol.flux_up(t)[0]
            want = np.atleast_1d(ev["out"])
            assert np.max(np.abs(got - want)) <= 1e-5 * max(1.0, np.max(np.abs(want)))
        else:
            gd, gdir = sol.flux_down(t)
            wd, wdir = ev["out"]
            assert np.max(np.abs(gd[0] - np.atleast_1d(wd))) <= 1e-5 * max(1.0, np.max(np.abs(wd)))
            assert np.max(np.abs(gdir[0] - np.atleast_1d(wdir))) <= 1e-9 * max(1.0, np.max(np.abs(wdir)))
        checked += 1
    assert checked > 0
    plan.close()


@pytest.mark.gpu
def test_device_bdrf_modes_equal_host_cosine_sums(amd):
    """The kernel against the same trapezoid sums in NumPy on random smooth samples (several columns, odd nphi):
    identical quadrature, so agreement is at rounding level; a solve with the device-made tables equals a solve with
    the host-made ones."""
    rng = np.random.default_rng(12)
    C, NQuad, L, nphi, nb = 3, 16, 4, 75, 5
    N = NQuad // 2
    p = np.arange(nphi)
    rho_qq = 0.1 + 0.05 * rng.uniform(size=(C, N, N, 1)) * np.cos(2 * np.pi * p / nphi) ** 2 + 0.02 * rng.uniform(size=(C, N, N, nphi))
    rho_q0 = 0.1 + 0.02 * rng.uniform(size=(C, N, nphi))
    wm = np.where(np.arange(nb) == 0, 1.0, 2.0)[:, None]
    cosmp = np.cos(2 * np.pi * np.arange(nb)[:, None] * p[None, :] / nphi)
    q = np.einsum("cijp,mp->cmij", rho_qq, wm * cosmp) / nphi
    q0 = np.einsum("cip,mp->cmi", rho_q0, wm * cosmp) / nphi
    tau = np.cumsum(rng.uniform(0.1, 0.6, (C, L)), axis=1)
    om = rng.uniform(0.3, 0.95, (C, L))
    Leg = (rng.uniform(0.3, 0.8, (C, L, 1)) ** np.arange(NQuad + 1)[None, None, :])
    common = dict(NQuad=NQuad, mu0=rng.uniform(0.3, 0.9, C), I0=np.full(C, 2.0), phi0=np.zeros(C), b_pos=0.2)
    _, sa = amd.pydisort_batch(tau, om, Leg_coeffs_all=Leg, bdrf_q=q, bdrf_q0=q0, **common)
    _, sb = amd.pydisort_batch(tau, om, Leg_coeffs_all=Leg, bdrf_samples=(rho_qq, rho_q0), NBDRF=nb, **common)
    t = np.concatenate((np.zeros((C, 1)), tau), axis=1)
    phi = np.array([0.0, 1.0, 2.5])
    ua, ub = sa.u(t, phi), sb.u(t, phi)
    assert np.max(np.abs(ua - ub)) <= 1e-12 * np.max(np.abs(ua))
    assert np.max(np.abs(sa.flux_up(t) - sb.flux_up(t))) <= 1e-12 * np.max(np.abs(sa.flux_up(t)))
    sa.plan.close()
    sb.plan.close()


@pytest.mark.gpu
@pytest.mark.parametrize("how", ["1", "2"])
def test_fused_bc_kernel_pivoted_path_on_goldens(how):
    """The fused boundary-condition kernel eliminates speculatively (diagonal pivots) and falls back to the fully
    pivoted elimination when a pivot is small.  RTD_BC_FORCE_PIVOT=1 sends EVERY elimination through the fallback
    (the LDS redo and its column un-permutation); =2 sends every chain of the 32-stream kernel through the register-resident
    column-pivoted elimination that near-conservative mode-0 chains take (GjPiv, round 4): the golden replay and the
    random cases with 18..32 streams must still pass."""
    import subprocess
    import sys
    env = dict(os.environ, RTD_BC_FORCE_PIVOT=how)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        os.path.join(os.path.dirname(__file__), "test_gpu_parity.py"),
                        os.path.join(os.path.dirname(__file__), "test_gpu_random_parity.py"),
                        "-k", "reference_golden or stamnes or synthetic_config or random or edge_cases or cfg4_batch"],
                       env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("switch", ["RTD_EIG_MFMA", "RTD_BC_TILED", "RTD_BC_FORCE_HANDOVER", "RTD_NO_PIPELINE", "RTD_SMALL_SPLIT", "RTD_BC_TILE_V1", "RTD_EIG_SMALL_V1", "RTD_BC_WIDE_V1"])
def test_alternative_kernel_paths_stay_correct(switch):
    """The runtime switches that select an alternative path -- RTD_EIG_MFMA=1: the assembly of Pm, Qm as rank-4 MFMA updates
    (32 streams); RTD_BC_TILED=1: the tiled fused kernel (the 64-stream kernel) with one tile, in place of the 32-stream
    kernel it generalises; RTD_BC_FORCE_HANDOVER=1: the tiled kernel hands every third Fourier mode's chain to the pivoted
    row-per-lane kernels (its last resort for singular carry blocks; the window's fused interface evaluation is then
    replaced by the evaluation kernel); RTD_NO_PIPELINE=1: the windows of a plan one after the other on one stream instead
    of the two-stream pipeline; RTD_SMALL_SPLIT=1: 2 ... 16 streams through the separate interface / sweep / evaluation kernels
    of rounds 1-3 instead of the fused rtd_bc_small_kernel (round 4); RTD_BC_TILE_V1=1: 64 streams through rtd_bc_tile_kernel<2>
    (one wavefront per SIMD) instead of the lean rtd_bc_tile2_kernel; RTD_EIG_SMALL_V1=1: 2 ... 8 streams through
    rtd_eigen_kernel<4, 2> instead of the one-lane-per-problem eigen kernel; RTD_BC_WIDE_V1=1: 66 ... 128 streams through the
    row-per-lane kernels (one wavefront per chain) instead of the four-wavefronts-per-chain kernels of rtd_bc_wide.hip -- pass the golden replay (it has 40-, 48- and 64-stream cases), the synthetic configs incl.
    cfg5, the random cases, the windowed plans and the fused-evaluation comparison.  (Round 3 removed the switches whose
    paths had lost every A/B: RTD_BC_SPLIT at 32 streams, RTD_EIG_V1, RTD_BCF_WAVES3.)"""
    import subprocess
    import sys
    env = dict(os.environ, **{switch: "1"})
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        os.path.join(os.path.dirname(__file__), "test_gpu_parity.py"),
                        os.path.join(os.path.dirname(__file__), "test_gpu_random_parity.py"),
                        "-k", "reference_golden or synthetic_config or random_many or edge_cases or fused_interface or windowed or layer_shards"
                              + (" or stamnes or cfg3 or random or mode_shards or failed_column or failure_in" if switch in ("RTD_SMALL_SPLIT", "RTD_EIG_SMALL_V1") else "")
                              + (" or random_64 or cfg5 or high_precision_truth_56" if switch == "RTD_BC_TILE_V1" else "")
                              + (" or beyond_64 or random_128 or random_many" if switch == "RTD_BC_WIDE_V1" else "")],
                       env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    if switch in ("RTD_EIG_MFMA", "RTD_NO_PIPELINE", "RTD_BC_TILE_V1", "RTD_BC_WIDE_V1"):
        # round 6: the retained forms under the switches that change the eigen stage (RTD_EIG_MFMA has its own reading of the lean
        # form's chunk lists), the window pipeline, or the consumers of the hand-off at 64 / 96 streams
        r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                            os.path.join(os.path.dirname(__file__), "test_gpu_retained.py"), "-k", "windowed_plan or lean_plan or auto_retention"],
                           env=env, capture_output=True, text=True, timeout=1200)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_high_precision_truth_m0(amd):
    """The HIP path against a 40-digit (mpmath) solution of the m = 0 discrete-ordinate problem computed straight from
    the ODE system (tools/hp_truth_m0.py; fixture tests/golden/hp_truth_m0.npz): a benign six-layer atmosphere and one
    with two omega = 1 - 1e-6 layers between absorbing ones.  The HIP path stays at rounding level on both; the
    reference's algorithm in float64 (the oracle) is at 1.5e-13 and 6e-9 (test_oracle_against_high_precision_truth)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("hp_cases", os.path.join(os.path.dirname(__file__), "..", "tools", "hp_cases.py"))
    hp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(hp)
    Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "hp_truth_m0.npz"))
    for name, kw in (("benign", hp.benign_case()), ("harsh", hp.harsh_case())):
        _, fu, fd, u0 = amd.pydisort(**kw)
        tau = np.concatenate(([0.0], kw["tau_arr"]))
        truth = Z[name]
        assert np.max(np.abs(u0(tau) - truth)) <= 1e-12 * np.max(np.abs(truth)), name
    # full intensities (8 Fourier modes, one omega = 1 - 1e-6 layer): u(tau, phi) against the 40-digit Fourier sum
    kw = hp.intensity_case()
    _, fu, fd, u0, u = amd.pydisort(**kw)
    tau = np.concatenate(([0.0], kw["tau_arr"]))
    truth = Z["intensity"]
    assert np.max(np.abs(u(tau, hp.PHI) - truth)) <= 1e-12 * np.max(np.abs(truth))


@pytest.mark.gpu
def test_one_call_entry_points_equal_the_plan_api(amd):
    """rtd_solve_batch / rtd_solve_tensors (the one-call names of SURVEY section 8(b)) against the plan API on cfg3
    columns (thermal + beam + BDRF + Dirichlet sources)."""
    from pydisort_amd import synthetic, _engine
    cfg = synthetic.cfg3_columns(5, big=False)
    _, sol = amd.pydisort_batch(**cfg)
    prep = sol.plan.prep
    tau = np.concatenate((np.zeros((5, 1)), cfg["tau_arr"]), axis=1)
    phi = np.array([0.0, 1.1, 3.0])
    once = _engine.solve_batch_once(prep, tau, phi)
    assert np.array_equal(once["u"], sol.u(tau, phi))
    assert np.array_equal(once["flux_up"], sol.flux_up(tau))
    fd = sol.flux_down(tau)
    assert np.array_equal(once["flux_down_diffuse"], fd[0]) and np.array_equal(once["flux_down_direct"], fd[1])
    t1 = _engine.solve_tensors_once(prep, column=3)
    t2 = sol.plan.tensors(3)
    for k in ("GC", "K", "B"):
        assert np.array_equal(t1[k], t2[k]), k
    sol.plan.close()


@pytest.mark.gpu
@pytest.mark.parametrize("G", [2, 3, 8])
def test_fourier_mode_shards_add_up(amd, G):
    """SURVEY section 8(e), secondary partition: G plans, each solving the Fourier modes r, r + G, ... of the same
    columns (what G ranks would do), evaluated separately; the partial intensities, u0 and fluxes must add up to the
    unsharded result.  cfg3 columns: beam + thermal + Dirichlet + BDRF sources, delta-M off; cfg4: delta-M on."""
    from pydisort_amd import synthetic
    for cfg in (synthetic.cfg3_columns(3, big=True), synthetic.cfg4_columns(2)):
        tau = np.concatenate((np.zeros((cfg["tau_arr"].shape[0], 1)), cfg["tau_arr"]), axis=1)
        phi = np.array([0.0, 0.9, 2.5, 4.0])
        _, full = amd.pydisort_batch(**cfg)
        want_u, want_u0 = full.u(tau, phi), full.u0(tau)
        want_fu, want_fd = full.flux_up(tau), full.flux_down(tau)
        full.plan.close()
        got_u = np.zeros_like(want_u)
        got_u0 = np.zeros_like(want_u0)
        got_fu = np.zeros_like(want_fu)
        got_fd = [np.zeros_like(want_fd[0]), np.zeros_like(want_fd[1])]
        for r in range(G):
            _, part = amd.pydisort_batch(mode_shard=(r, G), **cfg)
            got_u += part.u(tau, phi)
            got_u0 += part.u0(tau)
            got_fu += part.flux_up(tau)
            fd = part.flux_down(tau)
            got_fd[0] += fd[0]
            got_fd[1] += fd[1]
            part.plan.close()
        s = np.max(np.abs(want_u))
        assert np.max(np.abs(got_u - want_u)) <= 1e-13 * s
        assert np.max(np.abs(got_u0 - want_u0)) <= 1e-13 * s
        assert np.array_equal(got_fu, want_fu)
        assert np.array_equal(got_fd[0], want_fd[0]) and np.array_equal(got_fd[1], want_fd[1])


# ---- round 2: windowed plans, the u + flux collective, device status --------------------------------------------
@pytest.mark.gpu
def test_rccl_allgather_results_single_rank(amd):
    """The collective of SURVEY 8(e) -- ncclAllGather of u AND the fluxes, on the plan's communication stream, overlapped
    with the next run -- with a 1-rank communicator; two pipelined steps, the gather of the second is checked."""
    from pydisort_amd import synthetic
    from pydisort_amd._engine import Plan
    Plan.comm_preload()
    cfg = synthetic.cfg4_columns(8, L=5, NQuad=8)
    _, sol = amd.pydisort_batch(**cfg)
    plan = sol.plan
    tau = np.concatenate((np.zeros((8, 1)), cfg["tau_arr"]), axis=1)
    plan.set_eval_points(tau, np.array([0.0, 1.0]))
    plan.comm_init(Plan.comm_unique_id(), 0, 1)
    for _ in range(3):  # the evaluation kernel of step i + 1 waits for the gather of step i
        plan.run()
        plan.allgather_results()
    plan.synchronize()
    gu, gf = plan.fetch_gathered_results()
    res = plan.fetch()
    assert gu.shape == (8, 8, 6, 2) and gf.shape == (1, 3, 8, 6)
    assert np.array_equal(gu, res["u"])
    assert np.array_equal(gf[0, 0], res["flux_up"]) and np.array_equal(gf[0, 1], res["flux_down_diffuse"])
    assert np.array_equal(gf[0, 2], res["flux_down_direct"])
    plan.close()


@pytest.mark.gpu
def test_windowed_plan_equals_single_window(amd):
    """A plan whose intermediates cover 16 columns at a time (4 windows, the last one short) gives the results of the
    one-window plan bit for bit through every entry point: closures (re-solve per window), run/fetch, run_fetch, tensors."""
    from pydisort_amd import synthetic
    C = 50
    cfg = synthetic.cfg3_columns(C, big=True)   # thermal source, BDRF, beam, b_pos/b_neg: every input array is windowed
    tau = np.tile(np.linspace(0.0, cfg["tau_arr"][0, -1], 7), (C, 1))
    phi = np.array([0.0, 2.0])
    _, one = amd.pydisort_batch(**cfg)
    _, win = amd.pydisort_batch(work_columns=16, **cfg)
    assert win.plan.windows() == (16, 4) and one.plan.windows() == (50, 1)
    assert np.array_equal(win.u(tau, phi), one.u(tau, phi))
    assert np.array_equal(win.flux_up(tau), one.flux_up(tau))
    assert np.array_equal(win.flux_down(tau, True)[0], one.flux_down(tau, True)[0])   # antiderivative branch
    for col in (0, 17, 49):
        a, b = win.plan.tensors(col), one.plan.tensors(col)
        for k in a:
            assert np.array_equal(a[k], b[k]), (col, k)
    win.plan.set_eval_points(tau, phi)
    got = win.plan.run_fetch()
    one.plan.set_eval_points(tau, phi)
    one.plan.run()
    want = one.plan.fetch()
    for k in want:
        assert np.array_equal(got[k], want[k]), k
    win.plan.run()
    again = win.plan.fetch()
    for k in want:
        assert np.array_equal(again[k], want[k]), k


@pytest.mark.gpu
def test_window_pipeline_equals_serial_windows(amd):
    """The two-stream window pipeline (eigen stage of window w + 1 beside the boundary-condition stage of window w, two sets
    of hand-off buffers, events per slot) against the same plan with its windows one after the other (RTD_NO_PIPELINE=1):
    3 000 cfg4 columns in 12 windows of 256 (the last one short), three runs queued back to back without a host
    synchronisation in between (the pipeline stays full across runs), then run_fetch -- bit for bit."""
    from pydisort_amd import synthetic
    C = 3000
    cfg = synthetic.cfg4_columns_block(C, first=123)
    tau = np.concatenate((np.zeros((C, 1)), cfg["tau_arr"]), axis=1)
    phi = np.array([0.0, np.pi / 2, np.pi])
    results = {}
    for mode in ("pipelined", "serial"):
        if mode == "serial":
            os.environ["RTD_NO_PIPELINE"] = "1"
        try:
            _, sol = amd.pydisort_batch(work_columns=256, _defer_solve=True, **cfg)
        finally:
            os.environ.pop("RTD_NO_PIPELINE", None)
        plan = sol.plan
        assert plan.windows() == (256, 12)
        plan.set_eval_points(tau, phi)
        for _ in range(3):
            plan.run()
        a = plan.fetch()
        b = plan.run_fetch()
        for k in a:
            assert np.array_equal(a[k], b[k]), (mode, k)
        results[mode] = a
        plan.close()
    for k in results["serial"]:
        assert np.array_equal(results["pipelined"][k], results["serial"][k]), k
    assert np.all(np.isfinite(results["pipelined"]["u"])) and np.all(results["pipelined"]["flux_up"][:, 0] > 0)  # (black surface: 0 at the bottom)


@pytest.mark.gpu
def test_cfg5_windowed_plan_equals_single_window(amd):
    """The 64-stream path (eigen kernel at NP = 32, tiled fused boundary-condition kernel, 2-mode BDRF, thermal source) through
    a plan of several windows, the last one short: 22 cfg5 columns in windows of 8 -- interface points (the fused evaluation
    inside the boundary-condition kernel) through run / fetch and run_fetch, general points through the closures -- bit-equal
    to the one-window plan; the first 8 columns are the reference-computed goldens."""
    from pydisort_amd import synthetic
    C = 22
    cfg = synthetic.cfg5_columns(C)
    tau_if = np.concatenate((np.zeros((C, 1)), cfg["tau_arr"]), axis=1)
    phi = np.array([0.0, np.pi / 2, np.pi])
    _, one = amd.pydisort_batch(**cfg)
    _, win = amd.pydisort_batch(work_columns=8, **cfg)
    assert win.plan.windows() == (8, 3) and one.plan.windows() == (C, 1)
    for plan in (one.plan, win.plan):
        plan.set_eval_points(tau_if, phi)
    one.plan.run()
    want = one.plan.fetch()
    got = win.plan.run_fetch()
    for k in want:
        assert np.array_equal(got[k], want[k]), k
    win.plan.run()
    again = win.plan.fetch()
    for k in want:
        assert np.array_equal(again[k], want[k]), k
    z = np.load(f"{goldens.HERE}/golden/synth/cfg5.npz")
    tau_g = np.stack([z[f"c{i}.tau_pts"] for i in range(8)] + [z["c0.tau_pts"] * (cfg["tau_arr"][i, -1] / cfg["tau_arr"][0, -1])
                                                             for i in range(8, C)])
    tau_g = np.minimum(tau_g, cfg["tau_arr"][:, -1:])
    uw, uo = win.u(tau_g, z["phi"]), one.u(tau_g, z["phi"])
    assert np.array_equal(uw, uo)
    for i in range(8):
        assert goldens.max_rel_err(uw[i], z[f"c{i}.u"])[0] < 5e-9
    one.plan.close()
    win.plan.close()


@pytest.mark.gpu
def test_cfg5_windowed_plan_with_forced_handover():
    """The same comparison with RTD_BC_FORCE_HANDOVER=1: every third chain of every window leaves the tiled kernel for the
    pivoted row-per-lane kernels, and the windows in which that happens take the evaluation kernel instead of the fused
    interface evaluation."""
    import subprocess
    import sys
    env = dict(os.environ, RTD_BC_FORCE_HANDOVER="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        os.path.abspath(__file__), "-k", "test_cfg5_windowed_plan_equals_single_window"],
                       env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_cfg5_batch_of_2560_columns_at_occupancy(amd):
    """BASELINE.json configs[4] at occupancy (round-3 verdict: the largest 64-stream batch under test was 22 columns): 2 560 cfg5
    columns -- 64 streams, 50 layers, 64 Fourier modes, 2-mode BDRF surface, thermal source -- through the windowed, pipelined
    path in windows of 128 columns (8 192 chains per launch: one per SIMD x 8), NumPy in -> NumPy out.  The 8 reference-computed
    golden columns and the 4 columns with 40-digit solutions are spliced into the batch 300 columns apart (different windows)
    and held to the reference (5e-9: its own roundoff at this size) and to the truth (1e-9 of the field scale, 1e-6
    pointwise) at the 51 interfaces; every column: finite, Beer's law, fluxes = quadrature of the zeroth mode; the spliced
    columns bit-equal to a 12-column call; and the closures of a windowed plan (general points) against the goldens."""
    from conftest import record_parity
    from pydisort_amd import synthetic
    C, WIN = 2560, 128
    cfg = synthetic.cfg5_columns(C, first=1000)
    z = np.load(f"{goldens.HERE}/golden/synth/cfg5.npz")
    gold = synthetic.cfg5_columns(8)
    at = 5 + 300 * np.arange(8)  # windows 0, 2, 4, 7, 9, 11, 14, 16
    for k, v in gold.items():
        if isinstance(v, np.ndarray) and v.shape[:1] == (8,):
            cfg[k] = cfg[k].copy()
            cfg[k][at] = v
    tau = np.concatenate((np.zeros((C, 1)), cfg["tau_arr"]), axis=1)
    phi = z["phi"]
    res = amd.solve_columns_streamed(cfg, tau, phi, chunk_columns=WIN)
    for k in ("u", "u0", "flux_up", "flux_down_diffuse", "flux_down_direct"):
        assert np.all(np.isfinite(res[k])), k
    assert np.allclose(res["flux_down_direct"], (cfg["I0"] * cfg["mu0"])[:, None] * np.exp(-tau / cfg["mu0"][:, None]), rtol=1e-13)
    from pydisort_amd._prepare import double_gauss
    mu, w = double_gauss(32)
    assert np.allclose(res["flux_up"], 2 * pi * np.einsum("cit,i->ct", res["u0"][:, :32], mu * w), rtol=1e-12)
    # (the downward diffuse flux of a delta-M scaled column also carries the difference of the scaled and the true direct beam,
    #  _assemble_intensity_and_fluxes.py:527-613: not a plain quadrature of u0)
    worst = worst_pw = 0.0
    for i in range(8):
        pts = np.searchsorted(z[f"c{i}.tau_pts"], tau[at[i]])
        assert np.array_equal(z[f"c{i}.tau_pts"][pts], tau[at[i]])
        a, b = goldens.max_rel_err(res["u"][at[i]], z[f"c{i}.u"][:, pts])
        worst, worst_pw = max(worst, a), max(worst_pw, b)
        fs = np.max(np.abs(z[f"c{i}.flux_down_diffuse"]))
        assert np.max(np.abs(res["flux_up"][at[i]] - z[f"c{i}.flux_up"][pts])) / fs < 5e-9
    record_parity("synthetic/cfg5_x2560_goldens", worst, worst_pw, 5e-9, PW_TOL, against="reference")
    assert worst < 5e-9 and worst_pw < PW_TOL, (worst, worst_pw)
    tw = tw_pw = 0.0
    for i in range(4):
        h = np.load(f"{goldens.HERE}/golden/hp/synth_cfg5_{i}.npz")
        pts = np.searchsorted(h["tau"], tau[at[i]])
        assert np.array_equal(h["tau"][pts], tau[at[i]]) and np.array_equal(h["phi"], phi)
        a, b = goldens.max_rel_err(res["u"][at[i]], h["u"][:, pts])
        tw, tw_pw = max(tw, a), max(tw_pw, b)
    # (forced hand-over: a third of the chains take the row-per-lane kernels, 2e-9 from the reference at this size -- the
    #  last-resort path is held to the reference's own budget, the tiled kernel to the truth)
    ttol = 5e-9 if os.environ.get("RTD_BC_FORCE_HANDOVER") else 1e-9
    record_parity("synthetic/cfg5_x2560_truth", tw, tw_pw, ttol, PW_TOL, against="40-digit truth")
    assert tw < ttol and tw_pw < PW_TOL, (tw, tw_pw)
    small = amd.solve_columns_streamed(gold, tau[at], phi, chunk_columns=WIN)
    for k in ("u", "u0", "flux_up", "flux_down_diffuse"):
        assert np.array_equal(small[k], res[k][at]), k
    # general points through the closures of a windowed, pipelined plan (20 windows are solved again per call)
    _, sol = amd.pydisort_batch(work_columns=WIN, **cfg)
    assert sol.plan.windows() == (WIN, C // WIN)
    tau_g = np.tile(z["c0.tau_pts"][None, 1::9], (C, 1)) * (cfg["tau_arr"][:, -1:] / cfg["tau_arr"][at[0], -1])
    tau_g[at] = np.stack([z[f"c{i}.tau_pts"][1::9] for i in range(8)])
    tau_g = np.minimum(tau_g, cfg["tau_arr"][:, -1:])
    ug = sol.u(tau_g, phi)
    assert np.all(np.isfinite(ug))
    for i in range(8):
        assert goldens.max_rel_err(ug[at[i]], z[f"c{i}.u"][:, 1::9])[0] < 5e-9
    sol.plan.close()


@pytest.mark.gpu
def test_cfg5_batch_at_occupancy_with_forced_handover():
    """The same 2 560-column batch with RTD_BC_FORCE_HANDOVER=1: every third chain of every 128-column window leaves the tiled
    kernel for the pivoted row-per-lane kernels (2 731 of 8 192 chains per launch), and those windows take the evaluation
    kernel instead of the fused interface evaluation."""
    import subprocess
    import sys
    env = dict(os.environ, RTD_BC_FORCE_HANDOVER="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        os.path.abspath(__file__), "-k", "test_cfg5_batch_of_2560_columns_at_occupancy"],
                       env=env, capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_streamed_batch_takes_its_sources_from_every_column(amd):
    """Regression (round-1 advisor finding): columns 0..15 have no beam and no thermal source, later columns have both.
    The streamed solver must not drop the sources of the later windows."""
    from pydisort_amd import synthetic
    C = 40
    cfg = synthetic.cfg3_columns(C, big=False)
    cfg["I0"] = cfg["I0"].copy()
    cfg["I0"][:16] = 0.0
    cfg["s_poly_coeffs"] = cfg["s_poly_coeffs"].copy()
    cfg["s_poly_coeffs"][:16] = 0.0
    tau = np.tile(np.linspace(0.0, cfg["tau_arr"][0, -1], 5), (C, 1))
    phi = np.array([0.3])
    got = amd.solve_columns_streamed(cfg, tau, phi, chunk_columns=16)
    _, sol = amd.pydisort_batch(**cfg)
    want = sol.u(tau, phi)
    assert np.max(np.abs(got["u"] - want)) <= 1e-12 * np.max(np.abs(want))
    # and against single-column solves of a sourced column and of a source-free column
    for i in (3, 30):
        kw = synthetic.column_kwargs(cfg, i)
        kw["BDRF_Fourier_modes"] = [0.5]
        res = amd.pydisort(**kw)
        want = res[4](tau[i], phi)
        assert np.max(np.abs(got["u"][i][:, :, 0] - want)) <= 1e-12 * max(1.0, np.max(np.abs(want)))


@pytest.mark.gpu
def test_numerical_failure_is_reported_not_returned(amd):
    """A phase function that is not positive definite after delta-M scaling (moments > 1) breaks the Cholesky
    factorisation on the device: the C ABI returns RTD_ERR_NUMERIC and the host raises (the reference raises
    LinAlgError / returns NaN, _solve_for_gen_and_part_sols.py:186, :226-231)."""
    from pydisort_amd import _lib
    from pydisort_amd._engine import Plan
    from pydisort_amd._prepare import prepare_columns
    C, Lr, NQ = 2, 3, 16
    N = NQ // 2
    Leg = np.tile(np.array([1.0] + [3.0] * NQ), (C, Lr, 1))   # impossible moments: Pm loses positive definiteness
    prep = prepare_columns(np.tile([0.5, 1.0, 2.0], (C, 1)), np.full((C, Lr), 0.99), NQ, Leg, np.full(C, 0.6),
                           np.full(C, np.pi), np.zeros(C), NQ, NQ, np.zeros((C, N, NQ)), np.zeros((C, N, NQ)),
                           np.zeros((C, Lr)), np.zeros((C, Lr, 0)), np.zeros((C, 0, N, N)), np.zeros((C, 0, N)))
    plan = Plan(prep)
    plan.solve()
    with pytest.raises(_lib.NumericalError):
        plan.evaluate(np.tile([0.0, 1.0], (C, 1)), np.array([0.0]))
    with pytest.raises(np.linalg.LinAlgError):   # the reference's exception type is a base class of ours
        plan.evaluate(np.tile([0.0, 1.0], (C, 1)), np.array([0.0]))
    # A failed solve whose results are never fetched must not haunt the next batch on the same plan (round-2 advisor
    # finding: the device status word used to be cleared only after a fetch): solve the bad batch again, do NOT fetch,
    # upload a good batch, run + fetch -> clean.
    plan.solve()
    good = prepare_columns(np.tile([0.5, 1.0, 2.0], (C, 1)), np.full((C, Lr), 0.9), NQ, np.tile(0.7 ** np.arange(NQ + 1), (C, Lr, 1)),
                           np.full(C, 0.6), np.full(C, np.pi), np.zeros(C), NQ, NQ, np.zeros((C, N, NQ)), np.zeros((C, N, NQ)),
                           np.zeros((C, Lr)), np.zeros((C, Lr, 0)), np.zeros((C, 0, N, N)), np.zeros((C, 0, N)))
    plan.set_columns(good)
    plan.set_eval_points(np.tile([0.0, 1.0], (C, 1)), np.array([0.0]))
    out = plan.run_fetch()
    assert np.all(np.isfinite(out["u"])) and np.all(out["flux_up"] > 0)
    plan.close()


@pytest.mark.gpu
def test_raw_upload_checks_every_shape(amd):
    """rtd_plan_set_columns_raw takes bare pointers whose extents the plan implies: the Python layer refuses any array whose
    shape does not match (round-2 advisor finding: only two of them used to be checked), and the plan's view of the batch
    (tau range of the closures) follows a raw upload."""
    from pydisort_amd import synthetic
    cfg = synthetic.cfg4_columns(5, L=4, NQuad=8)
    _, sol = amd.pydisort_batch(device_prepare=True, **cfg)
    plan, raw = sol.plan, dict(sol.prep["raw"])
    for key, bad in (("omega_arr", np.zeros((5, 3))), ("f_arr", np.zeros((4, 4))), ("mu0", np.zeros(6)), ("I0", np.zeros((5, 1))),
                     ("b_pos", np.zeros((5, 8, 3))), ("leg", np.zeros((5, 4, 3))), ("tau_arr", np.ones((5, 5)))):
        with pytest.raises(ValueError):
            plan.set_columns_raw(dict(raw, **{key: bad}))
    other = synthetic.cfg4_columns(5, L=4, NQuad=8, seed=11)
    raw2 = dict(raw, tau_arr=other["tau_arr"], omega_arr=other["omega_arr"])
    plan.set_columns_raw(raw2)
    assert np.array_equal(plan.prep["tau"], other["tau_arr"])
    plan.solve()
    tau = np.concatenate((np.zeros((5, 1)), other["tau_arr"]), axis=1)
    _, want = amd.pydisort_batch(**dict(cfg, tau_arr=other["tau_arr"], omega_arr=other["omega_arr"]))
    got = plan.evaluate(tau, np.array([0.0, 1.0]))
    assert np.max(np.abs(got["u"] - want.u(tau, np.array([0.0, 1.0])))) <= 1e-12 * np.max(np.abs(got["u"]))


@pytest.mark.gpu
def test_default_fourier_count_at_64_streams(amd):
    """pydisort(NQuad=64) with its default NFourier = 64 (the evaluation kernel's largest LDS footprint, 64 KB + change)
    against the oracle on a small atmosphere."""
    from oracle import disort_oracle as O
    k = np.arange(65)
    kw = dict(tau_arr=np.array([0.4, 1.1]), omega_arr=np.array([0.9, 0.7]), NQuad=64,
              Leg_coeffs_all=np.stack((0.7**k, 0.5**k)), mu0=0.55, I0=np.pi, phi0=0.3)
    tau, phi = np.array([0.0, 0.4, 0.9, 1.1]), np.array([0.0, 1.0, 3.0])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got, ref = amd.pydisort(**kw), O.pydisort(**kw)
    want = ref[4](tau, phi)
    assert np.max(np.abs(got[4](tau, phi) - want)) < 1e-9 * np.max(np.abs(want))
    assert np.allclose(got[1](tau), ref[1](tau), rtol=1e-9)


@pytest.mark.gpu
def test_high_precision_truth_32_streams(amd):
    """The HIP path against a 40-digit solution (tools/hp_truth_q32.py: mpmath, the reference's equations, banded
    elimination with partial pivoting) of a 20-layer, 32-stream atmosphere with four omega = 1 - 1e-6 layers -- the
    regime of the fused boundary-condition kernel -- for the Fourier modes 0, 1, 2, 9, 31.  Mode m is the last mode of
    a solve with NFourier = m + 1.  The float64 oracle's own distance to this truth (CPU test
    test_oracle_against_high_precision_truth_32_streams) is what the oracle-based tolerances of this suite allow for."""
    from conftest import record_parity
    z = np.load(f"{goldens.HERE}/golden/hp_truth_q32.npz")
    kw = {k[3:]: (z[k] if z[k].ndim else z[k][()]) for k in z.files if k.startswith("in.")}
    kw["NQuad"] = int(kw["NQuad"])
    tau = np.concatenate(([0.0], kw["tau_arr"]))
    worst = worst_pw = 0.0
    for m in z["modes"]:
        cfg = {k: (np.asarray(v)[None] if k != "NQuad" else v) for k, v in kw.items()}
        _, sol = amd.pydisort_batch(NFourier=int(m) + 1, **cfg)
        got = sol.plan.evaluate(tau[None], np.array([0.0]), want=("ulast",))["ulast"][0]
        a, b = goldens.max_rel_err(got, z[f"um{m}"])
        worst, worst_pw = max(worst, a), max(worst_pw, b)
        sol.plan.close()
    record_parity("hp_truth_q32", worst, worst_pw, 1e-9, 1e-6, against="40-digit truth")
    assert worst < 1e-9       # measured 7.5e-11 of the field scale (mode 0; the oracle: 6.4e-8) -- the conditioning of
    #                           omega = 1 - 1e-6 layers (~1e6) times double rounding
    assert worst_pw < 1e-6    # measured 1.0e-7, pointwise down to intensities 1e-8 of the largest


@pytest.mark.gpu
def test_fused_interface_evaluation_equals_the_evaluation_kernel(amd):
    """Run-path points at the layer interfaces are evaluated inside the boundary-condition kernel's backward sweep (u^m from
    the Y_l, A_l and coefficients in its registers); any other set of points goes through the evaluation kernel.  Same
    columns through both: interfaces + one interior point (general kernel) against interfaces only (fused), for a beam-only
    batch (cfg4), a batch with every source type (cfg3 at 32 streams is not available: cfg4 + thermal + surface), one
    layer, and Fourier-mode shards."""
    from pydisort_amd import synthetic
    from pydisort_amd._engine import Plan
    phi = np.array([0.0, 1.1, pi])
    cases = {"cfg4": synthetic.cfg4_columns(24), "one_layer": synthetic.cfg4_columns(5, L=1),
             "26_streams": synthetic.cfg4_columns(6, L=7, NQuad=26)}
    full = synthetic.cfg4_columns(12, L=9)
    full.update(s_poly_coeffs=np.tile(np.array([[0.3, 0.02, 0.001]]), (12, 9, 1)), b_pos=0.2, b_neg=0.1,
                bdrf_q=np.full((12, 1, 16, 16), 0.4), bdrf_q0=np.full((12, 1, 16), 0.4))
    cases["all_sources"] = full
    # more layers than the 20-layer window of small vectors the kernel keeps in LDS: refills in both sweeps
    cases["47_layers"] = synthetic.cfg4_columns(4, L=47)
    deep = synthetic.cfg4_columns(3, L=45)
    deep.update(s_poly_coeffs=np.tile(np.array([[0.3, 0.02, 0.001]]), (3, 45, 1)), b_pos=0.2, b_neg=0.1,
                bdrf_q=np.full((3, 1, 16, 16), 0.4), bdrf_q0=np.full((3, 1, 16), 0.4))
    cases["all_sources_45_layers"] = deep
    # 64 streams: the tiled kernel's fused evaluation (27 layers: more than its 24-layer window), beam only and every source
    cases["64_streams"] = synthetic.cfg4_columns(3, L=27, NQuad=64)
    s64 = synthetic.cfg4_columns(2, L=5, NQuad=64)
    s64.update(s_poly_coeffs=np.tile(np.array([[0.3, 0.02, 0.001]]), (2, 5, 1)), b_pos=0.2, b_neg=0.1,
               bdrf_q=np.full((2, 1, 32, 32), 0.4), bdrf_q0=np.full((2, 1, 32), 0.4))
    cases["all_sources_64_streams"] = s64
    for name, cfg in cases.items():
        C = cfg["tau_arr"].shape[0]
        if name == "26_streams":  # more azimuths than the Fourier-sum kernel takes in one pass over the modes (four)
            phi = np.array([0.0, 0.4, 1.1, 2.0, pi, 5.5])
        for shard in (None, (1, 3)):
            _, sol = amd.pydisort_batch(mode_shard=shard, **cfg)
            plan = sol.plan
            iface = np.concatenate((np.zeros((C, 1)), cfg["tau_arr"]), axis=1)
            plan.set_eval_points(iface, phi)
            plan.run()
            fused = plan.fetch()
            general = np.concatenate((iface, 0.5 * cfg["tau_arr"][:, :1]), axis=1)  # one more point: not the interfaces
            plan.set_eval_points(general, phi)
            plan.run()
            want = plan.fetch()
            for k in ("u", "u0", "flux_up", "flux_down_diffuse", "flux_down_direct"):
                a, b = fused[k], want[k][..., :-1, :] if k == "u" else want[k][..., :-1]
                scale = max(np.max(np.abs(b)), 1e-300)
                # (the fused path takes an interface from the layer below it, the evaluation kernel -- like the reference --
                #  from the layer above: the two agree to the residual of the boundary-condition solve's continuity rows)
                # (measured: <= 1e-12 at 32 streams; at 64 streams x 27 layers 5.2e-12, and 1.4e-11 of the smaller scale of a
                #  mode shard's partial sums)
                tol = 3e-11 if "64_streams" in name else 2e-12
                assert np.max(np.abs(a - b)) <= tol * scale, (name, shard, k, np.max(np.abs(a - b)) / scale)
            plan.close()


@pytest.mark.gpu
def test_device_side_preparation_equals_host_preparation(amd):
    """pydisort_batch(device_prepare=True): delta-M scaling, thermal-source recentring and source rescaling on the device
    (rtd_prep.hip, pydisort.py:316-372) against the NumPy front end, on batches that exercise every branch: cfg4 (delta-M,
    beam), cfg3 (thermal source of order 1, Dirichlet BCs, Lambertian surface, no delta-M), a batch with a cubic thermal
    source + delta-M + per-column source-free columns, and 10 streams (padding 5 -> 8 lanes)."""
    from pydisort_amd import synthetic
    phi = np.array([0.0, 2.0])
    mixed = synthetic.cfg4_columns(9, L=7, NQuad=16)
    mixed.update(s_poly_coeffs=np.tile(np.array([[0.4, 0.03, -0.002, 1e-4]]), (9, 7, 1)), b_pos=0.3, b_neg=np.linspace(0, 0.2, 9))
    mixed["I0"] = mixed["I0"].copy()
    mixed["I0"][::3] = 0.0
    cases = {"cfg4": synthetic.cfg4_columns(33), "cfg3": synthetic.cfg3_columns(20, big=True), "mixed": mixed,
             "q10": synthetic.cfg4_columns(6, L=4, NQuad=10)}
    for name, cfg in cases.items():
        C = cfg["tau_arr"].shape[0]
        tau = np.concatenate((np.zeros((C, 1)), cfg["tau_arr"], 0.37 * cfg["tau_arr"][:, :1]), axis=1)
        _, host = amd.pydisort_batch(**cfg)
        _, dev = amd.pydisort_batch(device_prepare=True, **cfg)
        for fn in ("u", "u0", "flux_up"):
            a = getattr(dev, fn)(tau, phi) if fn == "u" else getattr(dev, fn)(tau)
            b = getattr(host, fn)(tau, phi) if fn == "u" else getattr(host, fn)(tau)
            scale = np.max(np.abs(b))
            # cubic thermal source: the recentred coefficients come out of a different operation order (2e-13)
            assert np.max(np.abs(a - b)) <= (2e-12 if name == "mixed" else 1e-13) * scale, (name, fn, np.max(np.abs(a - b)) / scale)
        fd_a, fd_b = dev.flux_down(tau), host.flux_down(tau)
        assert np.allclose(fd_a[0], fd_b[0], rtol=1e-12, atol=1e-14) and np.allclose(fd_a[1], fd_b[1], rtol=1e-13)


@pytest.mark.gpu
@pytest.mark.parametrize("G", [1, 2, 5])
def test_layer_shards_stitch_the_boundary_condition_system(amd, G):
    """The north star's layer-sharded variant, emulated on one GPU: the eigen stage runs shard by
    shard over disjoint layer ranges (what G ranks would do at once), then ONE boundary-condition solve over all layers
    and the evaluation -- equal to the unsharded solve of the same plan.  With G = 1 the eigen-stage results also go
    through the collective of the variant (pack -> ncclAllGather with a 1-rank communicator -> unpack)."""
    from pydisort_amd import synthetic
    from pydisort_amd._engine import Plan
    cfg = synthetic.cfg4_columns(3, L=20)
    cfg.update(s_poly_coeffs=np.tile(np.array([[0.3, 0.02]]), (3, 20, 1)), b_pos=0.1)  # thermal vectors travel too
    tau = np.concatenate((np.zeros((3, 1)), cfg["tau_arr"], 0.5 * cfg["tau_arr"][:, :1]), axis=1)
    phi = np.array([0.0, 2.0])
    _, sol = amd.pydisort_batch(**cfg)
    want = sol.plan.evaluate(tau, phi)
    plan = sol.plan
    Lloc = 20 // G
    # wipe the eigen-stage results: solve OTHER inputs on the same plan (every intermediate is overwritten), then put
    # the right inputs back without solving -- from here on only the shards can make the results right
    other = synthetic.cfg4_columns(3, L=20, seed=77)
    other.update(s_poly_coeffs=np.tile(np.array([[0.1, 0.05]]), (3, 20, 1)), b_pos=0.3, tau_arr=cfg["tau_arr"])
    _, sol_other = amd.pydisort_batch(_defer_solve=True, **other)
    plan.set_columns(sol_other.prep)
    plan.solve()
    wiped = plan.evaluate(tau, phi)
    assert np.max(np.abs(wiped["u"] - want["u"])) > 1e-3 * np.max(np.abs(want["u"]))
    sol_other.plan.close()
    plan.set_columns(sol.prep)

    def reshard(skip=None):
        for r in reversed(range(G)):  # any order: the shards are independent
            if r != skip:
                plan.solve_layers(r * Lloc, Lloc)
        if G == 1:
            Plan.comm_preload()
            if not getattr(plan, "_comm_up", False):
                plan.comm_init(Plan.comm_unique_id(), 0, 1)
                plan._comm_up = True
            plan.allgather_layers(Lloc)
        plan.solve_bc()
        return plan.evaluate(tau, phi)

    if G > 1:  # a shard left out must show: its layers still hold the other batch's decomposition
        bad = reshard(skip=G - 1)
        assert not np.max(np.abs(bad["u"] - want["u"])) <= 1e-6 * np.max(np.abs(want["u"]))
    got = reshard()
    for k in ("u", "u0", "flux_up", "flux_down_diffuse"):
        if G == 1:
            assert np.array_equal(got[k], want[k]), k  # same launch geometry: bit-equal, through the collective
        else:  # other layers share a wavefront (a wavefront sweeps until its slowest problem is done): rounding differs
            assert np.max(np.abs(got[k] - want[k])) <= 1e-12 * np.max(np.abs(want[k])), k
    with pytest.raises(RuntimeError):
        plan.solve_layers(15, 10)  # beyond the last layer
    plan.close()


@pytest.mark.gpu
def test_bdrf_samples_with_many_azimuths(amd):
    """rtd_plan_set_bdrf_samples keeps cos(2 pi p / nphi) for all p in LDS: 12 000 azimuths are 96 KB of dynamic LDS (more
    than the 64 KB a workgroup gets by default elsewhere, within gfx950's 160 KB).  A Lambertian + one-harmonic
    reflectance whose Fourier modes are known exactly."""
    from pydisort_amd import synthetic
    from pydisort_amd._engine import Plan
    from pydisort_amd._prepare import double_gauss
    C, NQ, nphi = 2, 8, 12000
    N = NQ // 2
    cfg = synthetic.cfg4_columns(C, L=3, NQuad=NQ)
    mu, _ = double_gauss(N)
    dphi = 2 * pi * np.arange(nphi) / nphi
    a, b = 0.3, 0.1
    rho_qq = a + b * np.cos(dphi)[None, None, None, :] * (mu[None, :, None, None] * mu[None, None, :, None]) * np.ones((C, 1, 1, 1))
    rho_q0 = a + b * np.cos(dphi)[None, None, :] * mu[None, :, None] * cfg["mu0"][:, None, None]
    _, got = amd.pydisort_batch(bdrf_samples=(rho_qq, rho_q0), NBDRF=2, **cfg)
    q = np.stack((np.full((C, N, N), a), b * np.broadcast_to(np.outer(mu, mu), (C, N, N))), axis=1)
    q0 = np.stack((np.full((C, N), a), b * mu[None, :] * cfg["mu0"][:, None]), axis=1)
    _, want = amd.pydisort_batch(bdrf_q=q, bdrf_q0=q0, **cfg)
    tau = np.concatenate((np.zeros((C, 1)), cfg["tau_arr"]), axis=1)
    phi = np.array([0.0, 1.0])
    assert np.max(np.abs(got.u(tau, phi) - want.u(tau, phi))) <= 1e-12 * np.max(np.abs(want.u(tau, phi)))


@pytest.mark.gpu
def test_high_precision_truth_56_streams(amd):
    """The atmosphere on which the random 64-stream cases found the HIP path and the oracle 3.4e-6 apart (56 streams, 8
    layers, a thin top layer with omega = 1 - 1e-6, isotropic illumination from above): against the 40-digit solution
    (tools/hp_truth_q32.py --q56) the HIP path -- eigen kernel at NP = 32, tiled fused boundary-condition kernel -- is
    within 1e-9 (measured 1e-11); the oracle is the one that is 3.4e-6 off
    (test_oracle_against_high_precision_truth_56_streams)."""
    from conftest import record_parity
    z = np.load(f"{goldens.HERE}/golden/hp_truth_q56.npz")
    kw = {k[3:]: (z[k] if z[k].ndim else z[k][()]) for k in z.files if k.startswith("in.")}
    kw["NQuad"], kw["only_flux"] = int(kw["NQuad"]), bool(kw["only_flux"])
    tau = np.concatenate(([0.0], kw["tau_arr"]))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = amd.pydisort(**kw)[3](tau)
    a, b = goldens.max_rel_err(got, z["um0"])
    record_parity("hp_truth_q56", a, b, 1e-9, 1e-6, against="40-digit truth")
    assert a < 1e-9 and b < 1e-6
