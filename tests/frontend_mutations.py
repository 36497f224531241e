"""Invalid-input cases for the front end (shared by tests/golden/make_frontend_error_goldens.py, which records what the
reference raises for them, and tests/test_host_logic.py, which holds the drop-in to the same exception type and text)."""
import numpy as np


def good(seed):
    rng = np.random.default_rng([606, seed])
    L = int(rng.integers(1, 4))
    NQuad = int(rng.choice([4, 8, 16]))
    g = rng.uniform(0.1, 0.8, L)
    return dict(tau_arr=np.cumsum(rng.uniform(0.1, 1, L)), omega_arr=rng.uniform(0.1, 0.9, L), NQuad=NQuad,
                Leg_coeffs_all=g[:, None] ** np.arange(NQuad + 3)[None, :], mu0=0.5, I0=1.0, phi0=1.0)


def _first_node(NQuad):
    x, _ = np.polynomial.legendre.leggauss(NQuad // 2)
    return float(0.5 * (x[0] + 1.0))


MUTATIONS = {
    "negative tau": lambda k: k.update(tau_arr=-k["tau_arr"]),
    "decreasing tau": lambda k: k.update(tau_arr=k["tau_arr"][::-1] if len(k["tau_arr"]) > 1 else -k["tau_arr"]),
    "omega = 1": lambda k: k.update(omega_arr=np.ones_like(k["omega_arr"])),
    "negative omega": lambda k: k.update(omega_arr=-k["omega_arr"]),
    "omega of the wrong length": lambda k: k.update(omega_arr=np.append(k["omega_arr"], 0.5)),
    "odd NQuad": lambda k: k.update(NQuad=k["NQuad"] + 1),
    "NQuad = 0": lambda k: k.update(NQuad=0),
    "zeroth moment not 1": lambda k: k.update(Leg_coeffs_all=k["Leg_coeffs_all"] * 0.9),
    "too few moments": lambda k: k.update(Leg_coeffs_all=k["Leg_coeffs_all"][:, :2]),
    "moment > 1": lambda k: k.update(Leg_coeffs_all=np.concatenate((k["Leg_coeffs_all"][:, :1], 1.5 + 0 * k["Leg_coeffs_all"][:, 1:]), axis=1)),
    "negative mu0": lambda k: k.update(mu0=-0.2),
    "mu0 > 1": lambda k: k.update(mu0=1.2),
    "negative I0": lambda k: k.update(I0=-1.0),
    "phi0 >= 2 pi": lambda k: k.update(phi0=7.0),
    "negative phi0": lambda k: k.update(phi0=-0.1),
    "NLeg > NQuad": lambda k: k.update(NLeg=k["NQuad"] + 1),
    "NLeg = 0": lambda k: k.update(NLeg=0),
    "NFourier > NLeg": lambda k: k.update(NFourier=k["NQuad"] + 1),
    "NFourier = 0": lambda k: k.update(NFourier=0),
    "negative f": lambda k: k.update(f_arr=-0.1 * np.ones_like(k["omega_arr"])),
    "f > 1": lambda k: k.update(f_arr=1.1 * np.ones_like(k["omega_arr"])),
    "f of the wrong length": lambda k: k.update(f_arr=np.full(len(k["omega_arr"]) + 1, 0.1)),
    "negative b_pos": lambda k: k.update(b_pos=-1.0),
    "negative b_neg": lambda k: k.update(b_neg=-1.0),
    "b_pos of the wrong length": lambda k: k.update(b_pos=np.ones(k["NQuad"] // 2 + 1)),
    "s_poly of the wrong shape": lambda k: k.update(s_poly_coeffs=np.ones((len(k["omega_arr"]) + 1, 2))),
    "negative s_poly": lambda k: k.update(s_poly_coeffs=-np.ones((len(k["omega_arr"]), 2))),
    "too many BDRF modes": lambda k: k.update(BDRF_Fourier_modes=[0.1] * (k["NQuad"] + 2)),
    "banded threshold < 3": lambda k: k.update(use_banded_solver_NLayers=2),
    "NT corrections with mu0 on a quadrature node": lambda k: k.update(NT_cor=True, mu0=_first_node(k["NQuad"]), f_arr=0.1 * np.ones_like(k["omega_arr"])),
    "no source at all": lambda k: k.update(I0=0.0),
}
SEEDS = range(3)


def case(name, seed):
    kw = good(seed)
    MUTATIONS[name](kw)
    return kw
