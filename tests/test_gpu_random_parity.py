"""Randomised parity sweep of the HIP path against the CPU oracle (pinned to the reference): random layer counts,
stream counts (every even NQuad up to 32, i.e. every padding pattern), optical properties incl. non-scattering layers,
beam / thermal / Dirichlet sources, scalar and tabulated BDRF modes, delta-M with and without NT corrections,
reduced NLeg / NFourier.  Seeds are fixed: the cases are reproducible."""
import warnings

import numpy as np
import pytest

import os

pytestmark = pytest.mark.gpu
HP_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hp")

# Seeds whose random input the oracle refuses (raises) or solves to a non-finite / wildly ill-conditioned field: none at
# present.  A seed listed here is skipped by name; any OTHER seed that the oracle cannot solve fails the test -- coverage
# cannot shrink silently.
ORACLE_REJECTS = {"random": set(), "random32": set(), "random64": set(), "random128": set()}


def eval_points(family, seed, kw):
    """The optical depths and azimuths a random case is compared at (tools/hp_truth_case.py builds its 40-digit
    fixtures at the same points)."""
    tau_arr = np.atleast_1d(kw["tau_arr"])
    rng = np.random.default_rng(seed)
    extra = 3 if family in ("random64", "random128") else 5
    tau = np.sort(np.concatenate(([0.0, tau_arr[-1]], tau_arr[:-1], rng.uniform(0, tau_arr[-1], extra))))
    phi = np.array([0.0, 0.7, 3.0]) if family in ("random64", "random128") else np.array([0.0, 0.7, 3.0, 5.5])
    return tau, phi


def oracle_solution(family, seed, kw, tau):
    """The oracle's callables for a random case; a seed the oracle cannot solve must be listed in ORACLE_REJECTS."""
    from oracle import disort_oracle as O
    listed = seed in ORACLE_REJECTS[family]
    try:
        ref = O.pydisort(**kw)
        scale = float(np.max(np.abs(ref[3](tau))))
        ok = np.isfinite(scale) and np.max(np.abs(ref[1](tau))) <= 1e8 * max(scale, 1e-300)
    except Exception:
        ref, ok = None, False
    if listed:
        assert not ok, f"{family}/{seed} is listed in ORACLE_REJECTS but the oracle solves it"
        pytest.skip("listed in ORACLE_REJECTS: the oracle cannot solve this input")
    assert ok, f"the oracle cannot solve {family}/{seed}: list the seed in ORACLE_REJECTS (with the reason) or change the generator"
    return ref


def arbitrated(family, seed, got_u, oracle_u, tol_scale=1e-9, tol_pw=1e-6):
    """A case with an omega > 1 - 1e-5 layer: there the reference's algorithm in float64 (the oracle) loses up to ~1e6 ulp
    and is itself beyond the north star's 1e-6 on some atmospheres, so the case is judged against its committed 40-digit
    solution (tools/hp_truth_case.py): the HIP path within tol_scale / tol_pw of the truth -- the tolerances every
    well-conditioned case is held to against the oracle -- AND the disagreement between HIP and oracle is the oracle's:
    |oracle - truth| >= |HIP - oracle| - tol_scale."""
    from conftest import record_parity
    import goldens
    path = os.path.join(HP_DIR, f"{family}_{seed}.npz")
    assert os.path.exists(path), f"no 40-digit fixture for the near-conservative case {family}/{seed}: run tools/hp_truth_case.py {family} {seed}"
    z = np.load(path)
    a, b = goldens.max_rel_err(got_u, z["u"])
    oa, ob = goldens.max_rel_err(oracle_u, z["u"])
    ha, hb = goldens.max_rel_err(got_u, oracle_u)
    record_parity(f"{family}/{seed}", a, b, tol_scale, tol_pw, against="40-digit truth", oracle_vs_truth_scale_rel=oa,
                  oracle_vs_truth_pointwise_rel=ob, hip_vs_oracle_scale_rel=ha, hip_vs_oracle_pointwise_rel=hb)
    assert oa >= ha - tol_scale, (family, seed, "the oracle is closer to the truth than to the HIP path", oa, ha)


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
    kw = make_case(seed)
    tau, phi = eval_points("random", seed, kw)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = oracle_solution("random", seed, kw, tau)
        got = pydisort_amd.pydisort(**kw)
    want0 = ref[3](tau)
    scale = max(float(np.max(np.abs(want0))), 1e-300)
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
    """Both metrics of SURVEY 8(d), both asserted: 1e-9 of the field scale (what two float64 implementations reach) and the
    north star's 1e-6 pointwise -- against the oracle, or, for the cases with a near-conservative layer, against their
    40-digit solution (see ``arbitrated``)."""
    import pydisort_amd
    from conftest import record_parity
    import goldens
    kw = make_case_many_streams(seed)
    tau, phi = eval_points("random32", seed, kw)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = oracle_solution("random32", seed, kw, tau)
        got = pydisort_amd.pydisort(**kw)
    want, gotu = ref[4](tau, phi), got[4](tau, phi)
    scale = max(float(np.max(np.abs(want))), 1e-300)
    near_conservative = bool(np.any(kw["omega_arr"] > 1 - 1e-5))
    if near_conservative:
        arbitrated("random32", seed, gotu, want)
        z = np.load(os.path.join(HP_DIR, f"random32_{seed}.npz"))
        assert np.allclose(got[1](tau), z["flux_up"], rtol=1e-8, atol=1e-9 * scale)
    else:
        a, b = goldens.max_rel_err(gotu, want)
        record_parity("random32/%d" % seed, a, b, 1e-9, 1e-6)
        assert np.allclose(got[1](tau), ref[1](tau), rtol=1e-8, atol=1e-9 * scale)


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
    """As the 32-stream cases; against the oracle the scale tolerance is 2e-9 (64 streams: the reference's own roundoff is
    ~1e-9, see the cfg5 goldens).  Seed 5 is the atmosphere where the reference's algorithm is 3.4e-6 off the truth
    (tools/hp_truth_q32.py --q56); every near-conservative seed is now judged against its own 40-digit solution."""
    import pydisort_amd
    from conftest import record_parity
    import goldens
    kw = make_case_64_streams(seed)
    tau, phi = eval_points("random64", seed, kw)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = oracle_solution("random64", seed, kw, tau)
        got = pydisort_amd.pydisort(**kw)
    want, gotu = ref[4](tau, phi), got[4](tau, phi)
    scale = max(float(np.max(np.abs(want))), 1e-300)
    near_conservative = bool(np.any(kw["omega_arr"] > 1 - 1e-5))
    if near_conservative:
        arbitrated("random64", seed, gotu, want, tol_scale=2e-9)
        z = np.load(os.path.join(HP_DIR, f"random64_{seed}.npz"))
        assert np.allclose(got[1](tau), z["flux_up"], rtol=2e-8, atol=2e-9 * scale)
    else:
        a, b = goldens.max_rel_err(gotu, want)
        record_parity("random64/%d" % seed, a, b, 2e-9, 1e-6)
        assert np.allclose(got[1](tau), ref[1](tau), rtol=2e-8, atol=2e-9 * scale)


# Near-conservative cases that wide random sweeps (tools/fuzz_parity.py: 7 920 and 33 000 seeds beyond the fixed ranges, round 3)
# found more than 1e-5 away from the oracle, pinned with their 40-digit solutions: in every one of them the distance is the
# reference algorithm's own distance from the truth (1.0e-5 ... 2.5e-5 of the field scale at 56-64 streams with an
# omega = 1 - 1e-6 layer, up to 7e-5 pointwise); the HIP path is within 3e-13 ... 1.4e-9 of the truth.
EXTRA_ARBITRATED = [("random64", s) for s in (130, 321, 400, 410, 425, 502, 513, 531, 666, 792, 1158, 1539, 2104)]  # (the last three: round 5)


@pytest.mark.parametrize("family,seed", EXTRA_ARBITRATED)
def test_near_conservative_case_found_by_the_random_sweep(family, seed):
    import pydisort_amd
    kw = {"random64": make_case_64_streams, "random32": make_case_many_streams}[family](seed)
    tau, phi = eval_points(family, seed, kw)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from oracle import disort_oracle as O
        ref = O.pydisort(**kw)
        got = pydisort_amd.pydisort(**kw)
    assert np.any(kw["omega_arr"] > 1 - 1e-5)
    arbitrated(family, seed, got[4](tau, phi), ref[4](tau, phi), tol_scale=2e-9)


@pytest.mark.parametrize("family,seed", [("random64", 2197), ("random128", 591), ("random128", 1884)])
def test_largest_well_conditioned_disagreements_of_the_random_sweep_are_the_oracles(family, seed):
    """The largest HIP/oracle differences of the 33 000-seed sweep among the cases WITHOUT a layer near omega = 1: 1.2e-8 at
    64 streams (random64/2197), 1.2e-7 at 128 streams (random128/591, /1884).  Their 40-digit solutions put them on the
    oracle's account (the float64 eigen-decomposition of the reference's algorithm at N = 32 / 64); the HIP path is held to
    the tolerance of every other case."""
    import pydisort_amd
    from oracle import disort_oracle as O
    kw = {"random64": make_case_64_streams, "random128": make_case_128_streams}[family](seed)
    tau, phi = eval_points(family, seed, kw)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = O.pydisort(**kw)
        got = pydisort_amd.pydisort(**kw)
    assert not np.any(kw["omega_arr"] > 1 - 1e-5)
    arbitrated(family, seed, got[4](tau, phi), ref[4](tau, phi), tol_scale=2e-9)


def test_thermal_polynomial_in_a_near_conservative_thin_layer_is_as_good_as_the_reference():
    """The one input of the 41 000-seed sweeps on which the HIP path was FURTHER from the truth than the reference (seed 3736 of
    the 32-stream family): a single layer of optical depth 5.6e-4 with omega = 1 - 1e-6 and a quadratic thermal source, no
    beam.  The reference's formulation -- and this one: same mathematics -- represents the field (3e-8) as the difference of a
    polynomial particular solution of size 1e10 (1/k^3, k = 1.4e-3) and a homogeneous part of the same size: 17 orders of
    cancellation, so that the result is as good as the RELATIVE accuracy of the coefficients C (measured on the oracle:
    du/u = 4e11 dC/C).  LAPACK's pivoted LU gives the reference 1.8e-4 of the field scale; the speculative (diagonal-pivot)
    elimination of the fused kernels gave 9e-2.  Chains of that kind -- Fourier mode 0 with a thermal source and an eigenvalue
    below 0.02 -- now take the pivoted elimination (rtd_bc.hip: chain_needs_pivoting): 1.9e-4.  Held: the cancellation is
    what it is said to be, the HIP path is no further from the 40-digit solution than three times the reference's own
    distance, and one to four orders of cancellation less (omega = 0.9999 ... 0.99) give 1e-7 ... 1e-10 against the oracle."""
    import pydisort_amd
    from oracle import disort_oracle as O
    import goldens
    from conftest import record_parity
    kw = make_case_many_streams(3736)
    tau, phi = eval_points("random32", 3736, kw)
    z = np.load(os.path.join(HP_DIR, "random32_3736.npz"))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = O.pydisort(**kw)
        got = pydisort_amd.pydisort(**kw)
        p = O.prepare(**kw)
        G, K, B, Ginv0 = O.gen_and_part_sols(p)
        zvec = Ginv0 @ np.concatenate((1 / p["mu"], -1 / p["mu"]))
        v_top = O.mathscr_v(p, G[0], K[0], zvec, [0.0], [0])[:, 0]
    field = float(np.max(np.abs(z["u"])))
    assert np.max(np.abs(v_top)) / field > 1e16  # particular solution / field: beyond what float64 can subtract
    oa, ob = goldens.max_rel_err(ref[4](tau, phi), z["u"])
    ha, hb = goldens.max_rel_err(got[4](tau, phi), z["u"])
    record_parity("random32/3736 (17 orders of cancellation)", ha, hb, None, None, against="40-digit truth",
                  oracle_vs_truth_scale_rel=oa, oracle_vs_truth_pointwise_rel=ob)
    assert oa > 1e-5 and ha < 3 * oa  # (reference 1.8e-4, HIP 1.9e-4 measured)
    for om, tol in ((0.9999, 1e-7), (0.999, 1e-8), (0.99, 1e-10)):
        kw2 = dict(kw, omega_arr=np.array([om]))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            a, b = pydisort_amd.pydisort(**kw2)[3](tau), O.pydisort(**kw2)[3](tau)
        assert np.max(np.abs(a - b)) / np.max(np.abs(b)) < tol


def make_case_128_streams(seed):
    """Cases beyond 64 streams (66 <= NQuad <= 128, N = 33..64 padded to 64 lanes: the NP = 64 kernels, one
    eigenproblem / one chain per wavefront): 1-4 layers, omega <= 0.995 (no near-conservative layer: a 40-digit arbitration
    at these sizes takes hours), delta-M, beam / thermal / surface / Dirichlet sources mixed, few Fourier modes for most
    seeds and up to 40 for some (the reference's Legendre tables overflow beyond l + m ~ 170)."""
    rng = np.random.default_rng([2029, seed])
    NQuad = int(rng.choice([66, 70, 80, 96, 100, 112, 128]))
    N = NQuad // 2
    L = int(rng.integers(1, 5))
    g = rng.uniform(0.3, 0.9, L)
    nall = NQuad + int(rng.integers(1, 6))
    kw = dict(tau_arr=np.cumsum(rng.choice([0.01, 0.3, 1.0, 6.0], L) * rng.uniform(0.5, 1.5, L)),
              omega_arr=rng.uniform(0.05, 0.995, L), NQuad=NQuad, Leg_coeffs_all=g[:, None] ** np.arange(nall)[None, :],
              mu0=float(rng.uniform(0.1, 1.0)), I0=float(rng.uniform(0.5, 4.0)), phi0=float(rng.uniform(0, 2 * np.pi)),
              NFourier=int(rng.choice([1, 2, 3, 5, 8, 40])))
    if rng.random() < 0.7:
        kw["f_arr"] = g**NQuad
    if rng.random() < 0.4:
        kw["s_poly_coeffs"] = rng.uniform(0.0, 1.0, (L, int(rng.integers(1, 4))))
    if rng.random() < 0.5:
        kw["BDRF_Fourier_modes"] = [float(rng.uniform(0.05, 0.6))]
    if rng.random() < 0.4:
        kw["b_pos"] = rng.uniform(0, 1, N) if rng.random() < 0.5 else float(rng.uniform(0, 1))
    if rng.random() < 0.4:
        kw["b_neg"] = float(rng.uniform(0, 1))
    if rng.random() < 0.2:
        kw["I0"], kw["mu0"] = 0.0, 0.0
        if "s_poly_coeffs" not in kw and "b_neg" not in kw:
            kw["b_neg"] = 0.5
    return kw


@pytest.mark.parametrize("seed", range(10))
def test_random_128_stream_case_matches_oracle(seed):
    """Beyond 64 streams: scale tolerance 2e-8 (round 6: ten times the largest measured, 1.2e-9; 1e-7 until then -- the oracle
    and the reference themselves differ by ~3e-10 at these sizes, tests/test_oracle_vs_reference_goldens.py), pointwise 1e-7
    (measured <= 2.1e-9; the north star asks 1e-6)."""
    import pydisort_amd
    from conftest import record_parity
    import goldens
    kw = make_case_128_streams(seed)
    tau, phi = eval_points("random128", seed, kw)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = oracle_solution("random128", seed, kw, tau)
        got = pydisort_amd.pydisort(**kw)
    want, gotu = ref[4](tau, phi), got[4](tau, phi)
    scale = max(float(np.max(np.abs(want))), 1e-300)
    a, b = goldens.max_rel_err(gotu, want)
    record_parity("random128/%d" % seed, a, b, 2e-8, 1e-7)
    assert np.allclose(got[1](tau), ref[1](tau), rtol=2e-8, atol=2e-8 * scale)
    assert np.allclose(got[2](tau)[0], ref[2](tau)[0], rtol=2e-8, atol=2e-8 * scale)


@pytest.mark.parametrize("nquad", [66, 94, 96, 98, 126])
def test_stream_counts_either_side_of_the_48_stream_instances(nquad):
    """66 ... 96 streams (N <= 48) run on boundary-condition kernels that leave the padding columns 48 ... 63 out
    (rtd_sweep_wide_kernel<12>, rtd_iface_mfma_kernel<48>), 98 ... 128 on the full instances: the same three-layer atmosphere
    with a thermal source, a beam and a Lambertian surface at stream counts on both sides of that switch, and at the ends of the
    range, against the oracle."""
    import pydisort_amd
    import goldens
    from oracle import disort_oracle as O
    kw = dict(tau_arr=np.array([0.3, 1.1, 2.5]), omega_arr=np.array([0.95, 0.6, 0.85]), NQuad=nquad,
              Leg_coeffs_all=np.stack([0.75 ** np.arange(nquad + 1), 0.5 ** np.arange(nquad + 1), 0.8 ** np.arange(nquad + 1)]),
              mu0=0.55, I0=1.3, phi0=0.4, NFourier=5, f_arr=np.array([0.75, 0.5, 0.8]) ** nquad,
              s_poly_coeffs=np.array([[0.2, 0.05], [0.1, 0.0], [0.3, -0.02]]), b_neg=0.1,
              BDRF_Fourier_modes=[0.3])
    tau, phi = np.array([0.0, 0.2, 0.3, 1.0, 2.5]), np.array([0.0, 1.0, 2.5])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = O.pydisort(**kw)
        got = pydisort_amd.pydisort(**kw)
    want, gotu = ref[4](tau, phi), got[4](tau, phi)
    scale = float(np.max(np.abs(want)))
    a, b = goldens.max_rel_err(gotu, want)
    from conftest import record_parity
    record_parity("streams_%d" % nquad, a, b, 1e-8, 3e-8)  # round 6: ten times the measured 9.3e-10 / 3.0e-9 (1e-7 / 1e-6 until then)
    assert np.allclose(got[1](tau), ref[1](tau), rtol=1e-8, atol=1e-8 * scale)
    assert np.allclose(got[2](tau)[0], ref[2](tau)[0], rtol=1e-8, atol=1e-8 * scale)
