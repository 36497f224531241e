// rtd_nt.hip -- Nakajima-Tanaka intensity corrections (TMS + IMS) on the device.
//
// Replaces what the reference adds to the delta-M scaled intensity inside u_corrected
// (src/PythonicDISORT/pydisort.py:375-698): TMS (:409-596) with its layer prefix/suffix sums (:489-589)
// and IMS (:601-638).  Two kernels: rtd_nt_tables_kernel builds, per column and stream, the attenuated
// single-scattering sums contributed by the layers below (up-streams) / above (down-streams) of each layer;
// rtd_nt_apply_kernel adds rescale * (TMS + IMS) to u at every requested (tau, phi).
#include "rtd_device.h"

namespace {

// sum_l c[l] P_l(x) by the forward three-term recurrence
__device__ __forceinline__ double legendre_series(const double* c, int n, double x) {
  double pm1 = 1.0, p = x, acc = (n > 0) ? c[0] : 0.0;
  if (n > 1) acc += c[1] * x;
  for (int l = 1; l + 1 < n; ++l) {
    const double pn = ((2.0 * l + 1.0) * x * p - l * pm1) / (l + 1.0);
    pm1 = p;
    p = pn;
    acc += c[l + 1] * p;
  }
  return acc;
}

// tables R[c][antider][pos|neg][n][l]  (pydisort.py:489-589)
__global__ void rtd_nt_tables_kernel(RtdDev d, RtdNt nt) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;  // (c, antider, n)
  const int N = d.N, L = d.L;
  if (idx >= (long)d.C * 2 * N) return;
  const int n = (int)(idx % N), ad = (int)((idx / N) % 2), c = (int)(idx / (2 * N));
  const double mu = d.mu[n], mu0 = d.mu0[c];
  const double* ts0 = d.taus0 + (long)c * (L + 1);
  const double* sc = d.scale + (long)c * L;
  double* Rpos = nt.R + ((((long)c * 2 + ad) * 2 + 0) * d.NP + n) * L;
  double* Rneg = nt.R + ((((long)c * 2 + ad) * 2 + 1) * d.NP + n) * L;
  for (int l = 0; l < L; ++l) Rpos[l] = Rneg[l] = 0.0;
  for (int r = 0; r < L; ++r) {
    const double dt = ts0[r + 1] - ts0[r];
    const double intf = ad ? mu / sc[r] : 1.0;
    const double tpos = -expm1(-dt * (1.0 / mu + 1.0 / mu0)) * intf * exp(-ts0[r] / mu0);
    for (int ll = 0; ll < r; ++ll) Rpos[ll] += tpos * exp(-(ts0[r] - ts0[ll + 1]) / mu);
    const double dd = dt * (1.0 / mu - 1.0 / mu0);
    const double em1 = expm1(-fabs(dd));
    double tneg = (dd >= 0.0) ? -em1 * exp(-ts0[r + 1] / mu0) : em1 * exp(-dt / mu) * exp(-ts0[r] / mu0);
    if (ad) tneg = -intf * tneg;
    for (int ll = r + 1; ll < L; ++ll) Rneg[ll] += tneg * exp(-(ts0[ll] - ts0[r + 1]) / mu);
  }
}

__global__ void rtd_nt_apply_kernel(RtdDev d, RtdNt nt, RtdEval ev) {
  const int t = (int)(blockIdx.x % ev.ntau), c = (int)(blockIdx.x / ev.ntau);
  const int N = d.N, L = d.L, Qr = 2 * N;
  const double tau = ev.tau[(long)c * ev.ntau + t];
  const double* tau_arr = d.tau + (long)c * L;
  const double* ts0 = d.taus0 + (long)c * (L + 1);
  int l = 0;
  while (l < L - 1 && !(tau <= tau_arr[l])) ++l;
  const double sc = d.scale[(long)c * L + l];
  const double ts = ts0[l + 1] - (tau_arr[l] - tau) * sc;
  const double tb = ts0[l + 1], tt = ts0[l];
  const double mu0 = d.mu0[c], phi0 = d.phi0[c];
  const double I0_4pi = d.I0[c] / (4.0 * M_PI);
  const bool ad = ev.antider != 0;
  const double* wfull = nt.wfull + ((long)c * L + l) * nt.nleg_all;
  const double* wtrun = d.wleg + ((long)c * L + l) * d.P;
  const double* ims = nt.ims_coef + (long)c * nt.nleg_all;
  const double smu0 = nt.ims_par[c * 2 + 0], amp = nt.ims_par[c * 2 + 1];
  const double fl = nt.f[(long)c * L + l], oms = d.omega[(long)c * L + l];
  const double att = exp(-ts / mu0);
  for (int idx = threadIdx.x; idx < Qr * ev.nphi; idx += blockDim.x) {
    const int ir = idx / ev.nphi, p = idx % ev.nphi;
    const bool up = ir < N;
    const int i = up ? ir : ir - N;
    const double mu = d.mu[i], mus = up ? mu : -mu;
    const double nu = -mu0 * mus + sqrt(1.0 - mu0 * mu0) * sqrt(1.0 - mus * mus) * cos(phi0 - ev.phi[p]);
    // TMS: mathscr_B of the point's layer (:424-449) times the in-layer and other-layer attenuation factors
    const double calB = oms * I0_4pi * (mu0 / (mu0 + mus)) *
                        (legendre_series(wfull, nt.nleg_all, nu) / (1.0 - fl) - legendre_series(wtrun, d.P, nu));
    const double* R = nt.R + ((((long)c * 2 + (ad ? 1 : 0)) * 2 + (up ? 0 : 1)) * d.NP + i) * L;
    double fac;
    if (up) {
      const double e = exp((ts - tb) / mu - tb / mu0);
      fac = ad ? att / (-sc / mu0) - e / (sc / mu) : att - e;
      if (L > 1) fac += R[l] * exp((ts - tb) / mu);
    } else {
      const double e = exp((tt - ts) / mu - tt / mu0);
      fac = ad ? att / (-sc / mu0) + e / (sc / mu) : att - e;
      if (L > 1) fac += R[l] * exp((tt - ts) / mu);
    }
    double corr = calB * fac;
    if (!up) {  // IMS, downward streams only (:613-638)
      const double x = 1.0 / mu - 1.0 / smu0;
      double chi;
      if (ad)
        chi = ((smu0 - x * smu0 * (smu0 + tau)) * exp(-tau / smu0) - mu * exp(-tau / mu)) / (mu * smu0 * x * x);
      else
        chi = ((tau - 1.0 / x) * exp(-tau / smu0) + exp(-tau / mu) / x) / (mu * smu0 * x);
      corr += amp * legendre_series(ims, nt.nleg_all, nu) * chi;
    }
    ev.u[(((long)c * Qr + ir) * ev.ntau + t) * ev.nphi + p] += d.rescale[c] * corr;
  }
}

}  // namespace

void rtd_launch_nt_tables(const RtdDev& d, const RtdNt& nt, hipStream_t s) {
  const long n = (long)d.C * 2 * d.N;
  hipLaunchKernelGGL(rtd_nt_tables_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, d, nt);
}

void rtd_launch_nt_apply(const RtdDev& d, const RtdNt& nt, const RtdEval& e, hipStream_t s) {
  hipLaunchKernelGGL(rtd_nt_apply_kernel, dim3((unsigned)((long)e.ntau * d.C)), dim3(128), 0, s, d, nt, e);
}
