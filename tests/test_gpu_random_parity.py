"""Randomised parity sweep of the HIP path against the CPU oracle (pinned to the reference): random layer counts,
stream counts (every even NQuad up to 32, i.e. every padding pattern), optical properties incl. non-scattering layers,
beam / thermal / Dirichlet sources, scalar and tabulated BDRF modes, delta-M with and without NT corrections,
reduced NLeg / NFourier.  Seeds are fixed: the cases are reproducible."""
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def make_case(seed):
    rng = np.random.default_rng([2026, seed])
    L = int(rng.integers(1, 9))
    NQuad = int(rng.choice([2, 4, 6, 8, 10, 12, 16, 20, 24, 32]))
    N = NQuad // 2
    nall = NQuad + int(rng.integers(1, 12))
    tau = np.cumsum(10.0 ** rng.uniform(-3, 1.2, L))
    omega = rng.uniform(0.0, 0.995, L)
    omega[rng.random(L) < 0.15] = 0.0
    g = rng.uniform(0.0, 0.9, L)
    Leg = g[:, None] ** np.arange(nall)[None, :]
    kw = dict(tau_arr=tau, omega_arr=omega, NQuad=NQuad, Leg_coeffs_all=Leg, mu0=0.0, I0=0.0, phi0=0.0)
    beam = rng.random() < 0.7
    if beam:
        kw.update(mu0=float(rng.uniform(0.05, 1.0)), I0=float(rng.uniform(0.1, 5.0)), phi0=float(rng.uniform(0, 6.2)))
    if rng.random() < 0.5:
        kw["f_arr"] = g**NQuad
        if beam and rng.random() < 0.6:
            kw["NT_cor"] = True
    if rng.random() < 0.3 and NQuad > 4:
        kw["NLeg"] = int(rng.integers(max(2, NQuad // 2), NQuad + 1))
        kw["NFourier"] = int(rng.integers(1, kw["NLeg"] + 1))
        kw.pop("f_arr", None)
        kw.pop("NT_cor", None)
    if rng.random() < 0.5:
        ns = int(rng.integers(1, 4))
        kw["s_poly_coeffs"] = rng.uniform(0.0, 1.0, (L, ns)) * 10.0 ** (-np.arange(ns))[None, :]
    if rng.random() < 0.5:
        kw["b_pos"] = float(rng.uniform(0, 1)) if rng.random() < 0.5 else rng.uniform(0, 1, N)
    if rng.random() < 0.5:
        kw["b_neg"] = float(rng.uniform(0, 1))
    if not beam and "s_poly_coeffs" not in kw and "b_pos" not in kw and "b_neg" not in kw:
        kw["b_neg"] = 0.5
    r = rng.random()
    if r < 0.25:
        kw["BDRF_Fourier_modes"] = [float(rng.uniform(0.05, 0.6))]
    elif r < 0.5:
        a, b = rng.uniform(0.05, 0.3), rng.uniform(0.0, 0.3)
        kw["BDRF_Fourier_modes"] = [lambda mu, nmup, a=a, b=b: a * (1 + b * np.outer(mu, nmup)),
                                    lambda mu, nmup, a=a: 0.3 * a * np.outer(np.sqrt(1 - mu**2), np.sqrt(1 - np.asarray(nmup) ** 2))]
    if rng.random() < 0.2:
        kw["only_flux"] = True
        kw.pop("NT_cor", None)
    return kw


@pytest.mark.parametrize("seed", range(60))
def test_random_case_matches_oracle(seed):
    import pydisort_amd
    from oracle import disort_oracle as O
    kw = make_case(seed)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        try:
            ref = O.pydisort(**kw)
        except Exception:
            pytest.skip("oracle rejects this random input")
        got = pydisort_amd.pydisort(**kw)
    tau_arr = kw["tau_arr"]
    rng = np.random.default_rng(seed)
    tau = np.sort(np.concatenate(([0.0, tau_arr[-1]], tau_arr[:-1], rng.uniform(0, tau_arr[-1], 5))))
    phi = np.array([0.0, 0.7, 3.0, 5.5])
    want0 = ref[3](tau)
    scale = max(float(np.max(np.abs(want0))), 1e-300)
    if not np.isfinite(scale) or np.max(np.abs(ref[1](tau))) > 1e8 * scale:
        pytest.skip("oracle result is not finite / ill-conditioned")
    assert np.max(np.abs(got[3](tau) - want0)) / scale < 1e-8
    assert np.allclose(got[1](tau), ref[1](tau), rtol=1e-7, atol=1e-9 * scale)
    for a, b in zip(got[2](tau), ref[2](tau)):
        assert np.allclose(a, b, rtol=1e-7, atol=1e-9 * scale)
    if len(got) > 4:
        wantu = ref[4](tau, phi)
        su = max(float(np.max(np.abs(wantu))), scale)
        assert np.max(np.abs(got[4](tau, phi) - wantu)) / su < 1e-8


def make_case_many_streams(seed):
    """Cases for the fused boundary-condition kernel (18 <= NQuad <= 32, i.e. N = 9..16 padded to 16 lanes): more
    layers, thick / thin / non-scattering / near-conservative layers mixed, every source type."""
    rng = np.random.default_rng([2027, seed])
    L = int(rng.integers(1, 26))
    NQuad = int(rng.choice([18, 20, 22, 24, 26, 28, 30, 32]))
    N = NQuad // 2
    nall = NQuad + int(rng.integers(1, 8))
    tau = np.cumsum(10.0 ** rng.uniform(-4, 1.5, L))
    omega = rng.uniform(0.0, 0.999, L)
    omega[rng.random(L) < 0.1] = 0.0
    omega[rng.random(L) < 0.1] = 1 - 1e-6
    g = rng.uniform(0.0, 0.92, L)
    Leg = g[:, None] ** np.arange(nall)[None, :]
    kw = dict(tau_arr=tau, omega_arr=omega, NQuad=NQuad, Leg_coeffs_all=Leg, mu0=0.0, I0=0.0, phi0=0.0)
    beam = rng.random() < 0.75
    if beam:
        kw.update(mu0=float(rng.uniform(0.03, 1.0)), I0=float(rng.uniform(0.1, 5.0)), phi0=float(rng.uniform(0, 6.2)))
    if rng.random() < 0.6:
        kw["f_arr"] = g**NQuad
    if rng.random() < 0.4:
        ns = int(rng.integers(1, 4))
        kw["s_poly_coeffs"] = rng.uniform(0.0, 1.0, (L, ns)) * 10.0 ** (-np.arange(ns))[None, :]
    if rng.random() < 0.5:
        kw["b_pos"] = float(rng.uniform(0, 1)) if rng.random() < 0.5 else rng.uniform(0, 1, N)
    if rng.random() < 0.5:
        kw["b_neg"] = float(rng.uniform(0, 1))
    if not beam and "s_poly_coeffs" not in kw and "b_pos" not in kw and "b_neg" not in kw:
        kw["b_neg"] = 0.5
    r = rng.random()
    if r < 0.3:
        kw["BDRF_Fourier_modes"] = [float(rng.uniform(0.05, 0.9))]
    elif r < 0.6:
        a, b = rng.uniform(0.05, 0.4), rng.uniform(0.0, 0.5)
        kw["BDRF_Fourier_modes"] = [lambda mu, nmup, a=a, b=b: a * (1 + b * np.outer(mu, nmup)),
                                    lambda mu, nmup, a=a: 0.3 * a * np.outer(np.sqrt(1 - mu**2), np.sqrt(1 - np.asarray(nmup) ** 2))]
    return kw


@pytest.mark.parametrize("seed", range(40))
def test_random_many_stream_case_matches_oracle(seed):
    import pydisort_amd
    from oracle import disort_oracle as O
    kw = make_case_many_streams(seed)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        try:
            ref = O.pydisort(**kw)
        except Exception:
            pytest.skip("oracle rejects this random input")
        got = pydisort_amd.pydisort(**kw)
    tau_arr = kw["tau_arr"]
    rng = np.random.default_rng(seed)
    tau = np.sort(np.concatenate(([0.0, tau_arr[-1]], tau_arr[:-1], rng.uniform(0, tau_arr[-1], 5))))
    phi = np.array([0.0, 0.7, 3.0, 5.5])
    want = ref[4](tau, phi)
    scale = max(float(np.max(np.abs(want))), 1e-300)
    if not np.isfinite(scale) or np.max(np.abs(ref[1](tau))) > 1e8 * scale:
        pytest.skip("oracle result is not finite / ill-conditioned")
    # Layers with omega = 1 - 1e-6 limit the ORACLE, not the HIP path: against a 40-digit solution of a 20-layer,
    # 32-stream atmosphere with four such layers the reference's algorithm in float64 is off by 6.4e-8 (Fourier mode 0;
    # 1e-12 for the other modes) while the HIP path is below 1e-11 (tools/hp_truth_q32.py,
    # test_high_precision_truth_32_streams); random mixes reach 3e-7.  So: the north star's 1e-6 for those, and what two
    # float64 implementations reach (1e-9 of the field scale) for everything else.  Both metrics of SURVEY 8(d).
    from conftest import record_parity
    import goldens
    near_conservative = bool(np.any(kw["omega_arr"] > 1 - 1e-5))
    tol = 1e-6 if near_conservative else 1e-9
    a, b = goldens.max_rel_err(got[4](tau, phi), want)
    record_parity("random32/%d" % seed, a, b, tol, 1e-6 if near_conservative else 1e-6)
    assert a < tol
    if not near_conservative:
        assert b < 1e-6  # pointwise relative over |I| > 1e-8 max |I|
    assert np.allclose(got[1](tau), ref[1](tau), rtol=10 * tol, atol=tol * scale)


def make_case_64_streams(seed):
    """Cases for the 64-stream kernels (34 <= NQuad <= 64, i.e. N = 17..32 padded to 32 lanes: the tiled fused
    boundary-condition kernel and the eigen kernel at NP = 32): the mix of make_case_many_streams with fewer layers."""
    kw = make_case_many_streams(1000 + seed)
    rng = np.random.default_rng([2028, seed])
    NQuad = int(rng.choice([34, 40, 48, 56, 62, 64]))
    L = min(len(kw["tau_arr"]), int(rng.integers(1, 13)))
    N = NQuad // 2
    g = rng.uniform(0.0, 0.9, L)
    nall = NQuad + int(rng.integers(1, 6))
    kw.update(tau_arr=kw["tau_arr"][:L], omega_arr=kw["omega_arr"][:L], NQuad=NQuad,
              Leg_coeffs_all=g[:, None] ** np.arange(nall)[None, :])
    if "f_arr" in kw:
        kw["f_arr"] = g**NQuad
    if "s_poly_coeffs" in kw:
        kw["s_poly_coeffs"] = kw["s_poly_coeffs"][:L]
    if isinstance(kw.get("b_pos"), np.ndarray):
        kw["b_pos"] = rng.uniform(0, 1, N)
    return kw


@pytest.mark.parametrize("seed", range(12))
def test_random_64_stream_case_matches_oracle(seed):
    import pydisort_amd
    from conftest import record_parity
    import goldens
    from oracle import disort_oracle as O
    kw = make_case_64_streams(seed)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        try:
            ref = O.pydisort(**kw)
        except Exception:
            pytest.skip("oracle rejects this random input")
        got = pydisort_amd.pydisort(**kw)
    tau_arr = kw["tau_arr"]
    rng = np.random.default_rng(seed)
    tau = np.sort(np.concatenate(([0.0, tau_arr[-1]], tau_arr[:-1], rng.uniform(0, tau_arr[-1], 3))))
    phi = np.array([0.0, 0.7, 3.0])
    want = ref[4](tau, phi)
    scale = max(float(np.max(np.abs(want))), 1e-300)
    if not np.isfinite(scale) or np.max(np.abs(ref[1](tau))) > 1e8 * scale:
        pytest.skip("oracle result is not finite / ill-conditioned")
    # Near-conservative layers at these stream counts: the ORACLE (the reference's algorithm in float64) is the one that is
    # off -- 3.4e-6 of the field scale against a 40-digit solution on seed 5's atmosphere (56 streams, one thin
    # omega = 1 - 1e-6 layer), where the HIP path is within 1e-11 (tools/hp_truth_q32.py --q56,
    # test_high_precision_truth_56_streams).  Such cases are held to 2e-5 against the oracle; the others to 2e-9
    # (64 streams: the reference's own roundoff is ~1e-9, see the cfg5 goldens).
    near_conservative = bool(np.any(kw["omega_arr"] > 1 - 1e-5))
    tol = 2e-5 if near_conservative else 2e-9
    a, b = goldens.max_rel_err(got[4](tau, phi), want)
    record_parity("random64/%d" % seed, a, b, tol, 1e-6)
    assert a < tol
    assert np.allclose(got[1](tau), ref[1](tau), rtol=10 * tol, atol=tol * scale)
