"""Host-side preparation of the Nakajima-Tanaka corrections (TMS + IMS).

The corrections themselves run on the device (csrc/rtd_nt.hip); this module only forms the column-level
constants the reference computes once per ``pydisort`` call (src/PythonicDISORT/pydisort.py:601-611): the
tau-omega weighted averages and the Legendre series of the IMS term.
"""
import numpy as np


def nt_inputs(prep, omega, f, Leg_all, NLeg, mu0):
    """omega, f [C, L]; Leg_all [C, L, NLeg_all] -> (weighted_leg_all, f, ims_coef [C, NLeg_all], ims_par [C, 2])."""
    omega = np.asarray(omega, float)
    f = np.asarray(f, float)
    Leg_all = np.asarray(Leg_all, float)
    C, L, nall = Leg_all.shape
    tau = prep["tau"]
    w = omega * tau
    omega_avg = w.sum(axis=1) / tau.sum(axis=1)
    fw = (f * w).sum(axis=1)
    f_avg = fw / w.sum(axis=1)
    resid = Leg_all.copy()
    resid[:, :, :NLeg] = f[:, :, None]
    ravg = (resid * w[:, :, None]).sum(axis=1) / fw[:, None]
    ell = 2 * np.arange(nall) + 1
    ims_coef = ell[None, :] * (2 * ravg - ravg**2)
    mu0 = np.asarray(mu0, float).reshape(C)
    of = omega_avg * f_avg
    ims_par = np.stack((mu0 / (1 - of), prep["I0"] / (4 * np.pi) * of**2 / (1 - of)), axis=1)
    return Leg_all * ell[None, None, :], f, ims_coef, ims_par
