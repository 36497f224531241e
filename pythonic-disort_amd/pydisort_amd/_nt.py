"""Nakajima-Tanaka intensity corrections (TMS + IMS) added to the delta-M scaled device solution.

SURVEY section 8(f) row f1 ("next"): mirrors what the reference adds inside ``u_corrected``
(src/PythonicDISORT/pydisort.py:375-698).  These are closed-form single-scattering expressions
evaluated per requested (tau, phi) point on top of the device result u*; they are evaluated on the
host for now (post-processing layer L3 of the reference, not part of the hot path of section 8(a));
moving them into rtd_eval.hip is the next step of this row.
"""
import numpy as np
from numpy.polynomial.legendre import legval


class Corrections:
    def __init__(self, prep, mu_pos, omega, f, Leg_all, NLeg, mu0, phi0):
        self.mu, self.mu0, self.phi0 = mu_pos, float(mu0), float(phi0)
        self.N = len(mu_pos)
        self.mu_all = np.concatenate((mu_pos, -mu_pos))
        self.tau = prep["tau"][0]
        self.ts0 = prep["tau_s0"][0]
        self.sc = prep["scale_tau"][0]
        self.omega_s = prep["omega_s"][0]
        self.wtrunc = prep["wleg"][0]
        self.I0_4pi = prep["I0"][0] / (4 * np.pi)
        self.rescale = prep["rescale"][0]
        self.f = np.asarray(f, float)
        self.L = len(self.tau)
        nall = Leg_all.shape[1]
        self.wfull = Leg_all * (2 * np.arange(nall) + 1)[None, :]
        # IMS constants (pydisort.py:601-611)
        w = omega * self.tau
        self.omega_avg = w.sum() / self.tau.sum()
        self.f_avg = (self.f * w).sum() / w.sum()
        resid = Leg_all.copy()
        resid[:, :NLeg] = self.f[:, None]
        ravg = (resid * w[:, None]).sum(axis=0) / (self.f * w).sum()
        self.ims_series = (2 * np.arange(nall) + 1) * (2 * ravg - ravg**2)
        self.smu0 = self.mu0 / (1 - self.omega_avg * self.f_avg)
        self._layer_tables = {}

    def _nu(self, mu, phi):  # cosine of the scattering angle w.r.t. the beam (-mu0, phi0)
        return (-self.mu0 * mu)[:, None] + (np.sqrt(1 - self.mu0**2) * np.sqrt(1 - mu**2))[:, None] \
            * np.cos(self.phi0 - phi)[None, :]

    def _other_layers(self, antider):
        """Attenuated single-scattering sums from the layers below (up-streams) / above (down-streams)
        of each layer (pydisort.py:489-589)."""
        key = bool(antider)
        if key in self._layer_tables:
            return self._layer_tables[key]
        mu, mu0, ts0, L, N = self.mu, self.mu0, self.ts0, self.L, self.N
        dts = np.diff(ts0)
        intf = (mu[:, None] / self.sc[None, :]) if antider else np.ones((N, L))
        Rpos, Rneg = np.zeros((N, L)), np.zeros((N, L))
        for r in range(L):
            tpos = (1 - np.exp(-dts[r] * (1 / mu + 1 / mu0))) * intf[:, r] * np.exp(-ts0[r] / mu0)
            for ll in range(r):
                Rpos[:, ll] += tpos * np.exp(-(ts0[r] - ts0[ll + 1]) / mu)
            dd = dts[r] * (1 / mu - 1 / mu0)
            em1 = np.expm1(-np.abs(dd))
            tneg = np.where(dd >= 0, -em1 * np.exp(-ts0[r + 1] / mu0), em1 * np.exp(-dts[r] / mu) * np.exp(-ts0[r] / mu0))
            if antider:
                tneg = -intf[:, r] * tneg
            for ll in range(r + 1, L):
                Rneg[:, ll] += tneg * np.exp(-(ts0[ll] - ts0[r + 1]) / mu)
        self._layer_tables[key] = (Rpos, Rneg)
        return Rpos, Rneg

    def __call__(self, tau, phi, antider=False):
        mu, mu0, N = self.mu, self.mu0, self.N
        l = np.argmax(tau[:, None] <= self.tau[None, :], axis=1)
        ts = self.ts0[1:][l] - (self.tau[l] - tau) * self.sc[l]
        tb, tt, scl = self.ts0[1:][l], self.ts0[l], self.sc[l]
        # ---- TMS (pydisort.py:409-596)
        nu = self._nu(self.mu_all, phi)
        calB = np.empty((2 * N, len(tau), len(phi)))
        for ll in np.unique(l):
            b = (self.omega_s[ll] * self.I0_4pi) * (mu0 / (mu0 + self.mu_all))[:, None] \
                * (legval(nu, self.wfull[ll]) / (1 - self.f[ll]) - legval(nu, self.wtrunc[ll]))
            calB[:, l == ll, :] = b[:, None, :]
        att = np.exp(-ts / mu0)
        if antider:
            c0 = att / (-scl / mu0)
            up = c0[None, :] - np.exp((ts - tb)[None, :] / mu[:, None] - tb[None, :] / mu0) / (scl[None, :] / mu[:, None])
            dn = c0[None, :] + np.exp((tt - ts)[None, :] / mu[:, None] - tt[None, :] / mu0) / (scl[None, :] / mu[:, None])
        else:
            up = att[None, :] - np.exp((ts - tb)[None, :] / mu[:, None] - tb[None, :] / mu0)
            dn = att[None, :] - np.exp((tt - ts)[None, :] / mu[:, None] - tt[None, :] / mu0)
        if self.L > 1:
            Rpos, Rneg = self._other_layers(antider)
            up = up + Rpos[:, l] * np.exp((ts - tb)[None, :] / mu[:, None])
            dn = dn + Rneg[:, l] * np.exp((tt - ts)[None, :] / mu[:, None])
        corr = calB * np.concatenate((up, dn), axis=0)[:, :, None]
        # ---- IMS, downward streams only (pydisort.py:613-638)
        x = 1 / mu - 1 / self.smu0
        s0 = self.smu0
        if antider:
            chi = ((s0 - x[:, None] * s0 * (s0 + tau)[None, :]) * np.exp(-tau / s0)[None, :]
                   - mu[:, None] * np.exp(-tau[None, :] / mu[:, None])) / (mu * s0 * x**2)[:, None]
        else:
            chi = ((tau[None, :] - 1 / x[:, None]) * np.exp(-tau / s0)[None, :]
                   + np.exp(-tau[None, :] / mu[:, None]) / x[:, None]) / (mu * s0 * x)[:, None]
        amp = self.I0_4pi * (self.omega_avg * self.f_avg) ** 2 / (1 - self.omega_avg * self.f_avg)
        corr[N:] += (amp * legval(self._nu(-mu, phi), self.ims_series))[:, None, :] * chi[:, :, None]
        return self.rescale * corr
