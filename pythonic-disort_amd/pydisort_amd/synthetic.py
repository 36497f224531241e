"""Synthetic batched atmospheres of BASELINE.json's configs: inputs for bench.py
and for the parity tests.  Deterministic: column c of a config is always the same atmosphere."""
import numpy as np


def cfg4_columns(C, first=0, L=20, NQuad=32, seed=4, g_hi=0.85):
    """Henyey-Greenstein atmospheres: dtau ~ U(0.05,0.5), omega ~ U(0.5,0.99), g ~ U(0.6,g_hi) (0.85: SURVEY 8(d) cfg4),
    Leg[l,k] = g_l^k (NQuad+1 moments), f = g^NQuad (delta-M on), mu0 ~ U(0.2,1), I0 = pi, phi0 = 0,
    black surface, no thermal source.  Column index c uses its own generator default_rng([seed, c])."""
    tau = np.empty((C, L))
    omega = np.empty((C, L))
    g = np.empty((C, L))
    mu0 = np.empty(C)
    for i in range(C):
        rng = np.random.default_rng([seed, first + i])
        tau[i] = np.cumsum(rng.uniform(0.05, 0.5, L))
        omega[i] = rng.uniform(0.5, 0.99, L)
        g[i] = rng.uniform(0.6, g_hi, L)
        mu0[i] = rng.uniform(0.2, 1.0)
    k = np.arange(NQuad + 1)
    Leg = g[:, :, None] ** k[None, None, :]
    return dict(tau_arr=tau, omega_arr=omega, NQuad=NQuad, Leg_coeffs_all=Leg, mu0=mu0, I0=np.full(C, np.pi),
                phi0=np.zeros(C), f_arr=g**NQuad)


def cfg4_columns_block(C, first=0, L=20, NQuad=32, seed=4):
    """The distributions of ``cfg4_columns`` drawn from ONE generator for the whole block (vectorised: 10^5 columns in
    a fraction of a second instead of a Python loop over per-column generators).  Deterministic in (seed, first, C);
    NOT the same atmospheres as ``cfg4_columns`` -- used where only the workload's shape matters (throughput legs)."""
    rng = np.random.default_rng([seed, 77, first, C])
    tau = np.cumsum(rng.uniform(0.05, 0.5, (C, L)), axis=1)
    omega = rng.uniform(0.5, 0.99, (C, L))
    g = rng.uniform(0.6, 0.85, (C, L))
    mu0 = rng.uniform(0.2, 1.0, C)
    k = np.arange(NQuad + 1)
    Leg = g[:, :, None] ** k[None, None, :]
    return dict(tau_arr=tau, omega_arr=omega, NQuad=NQuad, Leg_coeffs_all=Leg, mu0=mu0, I0=np.full(C, np.pi),
                phi0=np.zeros(C), f_arr=g**NQuad)


def cfg4_cloud_columns(C, first=0, L=20, NQuad=32, cloud_layer=7, omega_cloud=1.0 - 1e-6):
    """cfg4 with a conservative cloud in EVERY column: layer `cloud_layer` has omega = 1 - 1e-6 (its smallest eigenvalue is
    ~1e-3: every Fourier-mode-0 chain of the batch takes the pivoted elimination of the boundary-condition kernel).  The
    near-conservative regime the benchmark distribution (omega <= 0.99) never touches; bench.py reports its rate beside the
    headline so that the cost of the careful path is driver-visible."""
    cfg = cfg4_columns_block(C, first=first, L=L, NQuad=NQuad)
    cfg["omega_arr"] = cfg["omega_arr"].copy()
    cfg["omega_arr"][:, cloud_layer] = omega_cloud
    return cfg


def cfg3_columns(C, first=0, big=True, seed=9):
    """Test-Problem-9-like multi-layer atmospheres with every layer different, replicated with a
    per-column perturbation of omega.  big=True: L=8, NQuad=16 (BASELINE wording); False: L=6, NQuad=8."""
    L, NQuad = (8, 16) if big else (6, 8)
    tau = np.cumsum(np.arange(1, L + 1)).astype(float)
    omega0 = 0.6 + np.arange(1, L + 1) * 0.05 * (6 / L)
    Leg = np.array([[(l + 1) / (L + 1)] for l in range(L)]) ** np.arange(NQuad + 1)[None, :]
    om = np.empty((C, L))
    for i in range(C):
        rng = np.random.default_rng([seed, first + i])
        om[i] = omega0 * (1 - 0.05 * rng.uniform())
    s_poly = np.tile(np.array([[0.3, 0.02]]), (C, L, 1))
    return dict(tau_arr=np.tile(tau, (C, 1)), omega_arr=om, NQuad=NQuad, Leg_coeffs_all=np.tile(Leg, (C, 1, 1)),
                mu0=np.full(C, 0.5), I0=np.full(C, np.pi), phi0=np.zeros(C), f_arr=0.0,
                s_poly_coeffs=s_poly, b_pos=0.1, b_neg=0.05,
                bdrf_q=np.full((C, 1, NQuad // 2, NQuad // 2), 0.5), bdrf_q0=np.full((C, 1, NQuad // 2), 0.5))


def cfg5_columns(C, first=0, L=50, NQuad=64, seed=5):
    """Stress config: 64 streams, 50 layers, 64 Fourier modes, the per-layer distributions of cfg4 with
    g ~ U(0.6, 0.9), f = g^64, 2-mode BDRF surface, linear thermal source."""
    base = cfg4_columns(C, first, L, NQuad, seed, g_hi=0.9)
    N = NQuad // 2
    x, _ = np.polynomial.legendre.leggauss(N)
    mu = 0.5 * (x + 1)
    rho = np.empty(C)
    for i in range(C):
        rho[i] = np.random.default_rng([seed + 100, first + i]).uniform(0.05, 0.4)
    q0 = rho[:, None, None] * (1 + 0.5 * mu[None, :, None] * mu[None, None, :])
    q1 = rho[:, None, None] * 0.3 * (np.sqrt(1 - mu**2)[None, :, None] * np.sqrt(1 - mu**2)[None, None, :])
    mu0 = base["mu0"]
    q00 = rho[:, None] * (1 + 0.5 * mu[None, :] * mu0[:, None])
    q10 = rho[:, None] * 0.3 * np.sqrt(1 - mu**2)[None, :] * np.sqrt(1 - mu0**2)[:, None]
    base.update(bdrf_q=np.stack((q0, q1), axis=1), bdrf_q0=np.stack((q00, q10), axis=1),
                b_pos=0.1 * (1 - rho), s_poly_coeffs=np.tile(np.array([[0.3, 0.02]]), (C, L, 1)))
    return base


def column_kwargs(cfg, i):
    """The i-th column of a batched config as keyword arguments of the single-column ``pydisort``."""
    from .pydisort import pydisort  # noqa: F401  (documented pairing)

    kw = dict(tau_arr=cfg["tau_arr"][i], omega_arr=cfg["omega_arr"][i], NQuad=cfg["NQuad"],
              Leg_coeffs_all=cfg["Leg_coeffs_all"][i], mu0=float(cfg["mu0"][i]), I0=float(cfg["I0"][i]),
              phi0=float(cfg["phi0"][i]))
    f = cfg.get("f_arr", 0.0)
    kw["f_arr"] = f[i] if np.ndim(f) == 2 else f
    if "s_poly_coeffs" in cfg:
        kw["s_poly_coeffs"] = cfg["s_poly_coeffs"][i]
    for b in ("b_pos", "b_neg"):
        if b in cfg:
            v = cfg[b]
            kw[b] = v[i] if np.ndim(v) >= 1 else v
    return kw


# Legendre moments of the Cloud C.1 phase function used by DISORT Test Problem 5 are 300 numbers of tabulated data
# (pydisotest/5_test.py:10-45); the BASELINE-literal 32-stream variant below uses a Henyey-Greenstein stand-in with
# the same asymmetry regime (g = 0.85, 300 moments) so that the case needs no table.
def literal_cases():
    """BASELINE.json configs[0] and configs[1] as literally worded (they have no Stamnes file):
    cfg1_q4  : Test Problem 1a with 4 streams (isotropic scattering, 1 layer);
    cfg2_q32 : a Test-Problem-5-like single thick layer, 32 streams, 300 moments, delta-M + NT corrections."""
    k = np.arange(300)
    cases = {
        "cfg1_q4": (dict(tau_arr=0.03125, omega_arr=0.2, NQuad=4, Leg_coeffs_all=np.array([1.0, 0, 0, 0, 0]), mu0=0.1,
                         I0=np.pi / 0.1, phi0=np.pi), np.array([0.0, 0.01, 0.03125])),
        "cfg2_q32": (dict(tau_arr=64.0, omega_arr=0.9, NQuad=32, Leg_coeffs_all=0.85**k, mu0=1.0, I0=np.pi, phi0=np.pi,
                          f_arr=0.85**32, NT_cor=True), np.array([0.0, 3.2, 32.0, 64.0])),
    }
    return cases


def many_stream_deep_cases():
    """The 66 ... 128-stream workloads that bench.py and tools/many_stream_timing.py TIME, at their full depth (the reference is
    finite there: 2 N_modes <= 170 - streams): name -> (keyword arguments of ``cfg4_columns``, NFourier,
    number of golden columns).  The timed batches start with these very columns.
    q128_L50 : 128 streams x 50 layers x 64 modes (BASELINE configs[4] in its 128-stream reading);
    q96_L20  : 96 streams x 20 layers x 48 modes;   q72_L50 : 72 streams x 50 layers x 36 modes."""
    return {"q128_L50": (dict(L=50, NQuad=128, g_hi=0.9), 64, 2), "q96_L20": (dict(L=20, NQuad=96, g_hi=0.9), 48, 2),
            "q72_L50": (dict(L=50, NQuad=72, g_hi=0.9), 36, 1)}


def many_stream_cases():
    """More than 64 streams (the reference has no cap on NQuad; its associated-Legendre tables overflow when l + m passes
    ~170, so NFourier stays where the reference itself is finite):
    q72  : 72 streams, 4 layers, all 72 modes, delta-M, beam + Dirichlet bottom;
    q96  : 96 streams, 3 layers, 40 modes, thermal source + Lambertian surface + beam;
    q128 : 128 streams, 2 layers, 64 modes, strongly forward-peaked Henyey-Greenstein (g = 0.9), delta-M."""
    def hg(g, n, L):
        return np.tile(g ** np.arange(n), (L, 1))
    cases = {
        "q72": (dict(tau_arr=np.array([0.3, 1.1, 2.0, 5.0]), omega_arr=np.array([0.95, 0.6, 0.99, 0.8]), NQuad=72,
                     Leg_coeffs_all=hg(0.8, 77, 4), mu0=0.55, I0=np.pi, phi0=0.4, f_arr=np.full(4, 0.8**72), b_pos=0.2),
                np.array([0.0, 0.15, 0.3, 1.1, 1.5, 2.0, 4.0, 5.0])),
        "q96": (dict(tau_arr=np.array([0.5, 2.0, 3.0]), omega_arr=np.array([0.7, 0.9, 0.5]), NQuad=96,
                     Leg_coeffs_all=hg(0.7, 100, 3), mu0=0.8, I0=2.0, phi0=1.0, NFourier=40,
                     s_poly_coeffs=np.array([[0.3, 0.05], [0.4, 0.0], [0.2, 0.1]]), BDRF_Fourier_modes=[0.25], b_neg=0.05),
                np.array([0.0, 0.25, 0.5, 1.0, 2.0, 2.5, 3.0])),
        "q128": (dict(tau_arr=np.array([1.0, 4.0]), omega_arr=np.array([0.98, 0.85]), NQuad=128,
                      Leg_coeffs_all=hg(0.9, 140, 2), mu0=0.35, I0=np.pi, phi0=0.0, NFourier=64, f_arr=np.full(2, 0.9**128)),
                 np.array([0.0, 0.5, 1.0, 2.5, 4.0])),
    }
    return cases
