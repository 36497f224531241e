// rtd_prep.hip -- the front end's preparation of the hot path's arguments, on the device.
//
// Replaces, for batches that arrive as RAW inputs (rtd_plan_set_columns_raw), the host-side part of
// src/PythonicDISORT/pydisort.py:316-372: delta-M scaling of optical depth, single-scattering albedo and phase-function
// moments (:316-338), the thermal source polynomial re-expressed in the scaled optical depth (subroutines.py:574-610,
// pydisort.py:326-329, :338), and the rescaling of every source by the largest one (:351-372).  One thread per column for
// the per-layer scalars (a running sum over the layers) and the layer order of the eigen stage; one thread per (column,
// layer, moment) for the weighted scaled Legendre coefficients.
#include "rtd_device.h"
#include "rtd_dd.h"

namespace {

__global__ void rtd_prepare_columns_kernel(RtdDev d, RtdRaw r) {
  const long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= d.C) return;
  const int L = d.L, M = d.M, N = d.N, NP = d.NP, Ns = d.Ns, P = d.P;
  const double* tau = r.tau + c * L;
  const double* om = r.omega + c * L;
  const double* f = r.f + c * L;
  double* omega_s = const_cast<double*>(d.omega) + c * L;
  double* tau_o = const_cast<double*>(d.tau) + c * L;
  double* ts0 = const_cast<double*>(d.taus0) + c * (L + 1);
  double* scale = const_cast<double*>(d.scale) + c * L;
  double* sp = Ns > 0 ? const_cast<double*>(d.spoly) + c * L * Ns : nullptr;
  int* perm = const_cast<int*>(d.lperm) + c * L;
  // delta-M scaling (:316-329; with f = 0 it is the identity, :331-338)
  double top = 0.0, acc = 0.0;
  ts0[0] = 0.0;
  bool scaled = false;  // "if np.any(f_arr > 0)" (:316); otherwise the scaled optical depth IS tau_arr (:331-338), to the bit
  for (int l = 0; l < L; ++l) scaled |= f[l] > 0.0;
  for (int l = 0; l < L; ++l) {
    const double fl = f[l], sc = 1.0 - om[l] * fl;
    const double shift = acc - sc * top;  // tau* = sc tau + shift inside layer l
    acc = scaled ? acc + sc * (tau[l] - top) : tau[l];
    ts0[l + 1] = acc;
    scale[l] = sc;
    omega_s[l] = (1.0 - fl) / sc * om[l];
    tau_o[l] = tau[l];
    if (Ns > 0) {
      // s(tau) = sum_j a_j tau^j in the layer; the kernels want it about the layer's top in the scaled depth (rtd_dd.h):
      // tau = top + x / sc  ->  b = Taylor shift of a by `top` (double-double), coefficient i divided by sc^i; then / sc, (1 - omega)
      const double* a = r.spoly + (c * L + l) * Ns;
      double* o = sp + (long)l * Ns;
      for (int i = 0; i < Ns; ++i) o[i] = a[i];
      rtd_taylor_shift(o, Ns, top);
      const double rsc = 1.0 / sc;
      double sci = 1.0;  // sc^-i
      for (int i = 0; i < Ns; ++i) {
        o[i] *= sci;
        sci *= rsc;
      }
      const double k = (1.0 - om[l]) * rsc;
      for (int i = 0; i < Ns; ++i) o[i] *= k;
    }
    top = tau[l];
  }
  // rescale factor (:351-366): max(I0, max b_pos, max b_neg[, s*(0) of the top layer, s*(tau*_L) of the bottom layer])
  const double I0 = r.I0[c];
  double big = I0;
  if (r.bpos)
    for (long k = 0; k < (long)M * N; ++k) big = fmax(big, r.bpos[c * M * N + k]);
  else
    big = fmax(big, 0.0);
  if (r.bneg)
    for (long k = 0; k < (long)M * N; ++k) big = fmax(big, r.bneg[c * M * N + k]);
  else
    big = fmax(big, 0.0);
  if (Ns > 0) {
    big = fmax(big, sp[0]);  // s*(0): the top layer's own origin
    double v = 0.0, tp = 1.0;
    const double dlast = ts0[L] - ts0[L - 1];  // s*(tau*_L) of the bottom layer, in its local variable
    for (int j = 0; j < Ns; ++j) {
      v += sp[(long)(L - 1) * Ns + j] * tp;
      tp *= dlast;
    }
    big = fmax(big, v);
  }
  const double div = (Ns == 0 && big == 0.0) ? 1.0 : big;  // nothing to rescale in a source-free column (:367)
  const double rdiv = 1.0 / div;
  const_cast<double*>(d.I0)[c] = I0 / div;
  const_cast<double*>(d.rescale)[c] = big;
  const_cast<double*>(d.mu0)[c] = r.mu0[c];
  const_cast<double*>(d.phi0)[c] = r.phi0[c];
  double* bp = const_cast<double*>(d.bpos) + c * M * NP;
  double* bn = const_cast<double*>(d.bneg) + c * M * NP;
  for (int m = 0; m < M; ++m)
    for (int i = 0; i < NP; ++i) {
      bp[m * NP + i] = (r.bpos && i < N) ? r.bpos[(c * M + m) * N + i] / div : 0.0;
      bn[m * NP + i] = (r.bneg && i < N) ? r.bneg[(c * M + m) * N + i] / div : 0.0;
    }
  if (Ns > 0)
    for (long k = 0; k < (long)L * Ns; ++k) sp[k] *= rdiv;
  // layer order of the eigen stage: ascending omega* / (1 - g*), g* = first scaled moment (see rtd_plan_set_columns)
  for (int l = 0; l < L; ++l) perm[l] = l;
  if (P > 1) {
    const long la = r.nleg_all;
    auto key = [&](int l) {
      const double fl = f[l];
      const double g = (r.leg[(c * L + l) * la + 1] - fl) / (1.0 - fl);
      return omega_s[l] / fmax(1.0 - g, 1e-6);
    };
    for (int i = 1; i < L; ++i) {  // insertion sort: L is a few tens
      const int li = perm[i];
      const double ki = key(li);
      int j = i - 1;
      while (j >= 0 && key(perm[j]) > ki) {
        perm[j + 1] = perm[j];
        --j;
      }
      perm[j + 1] = li;
    }
  }
}

__global__ void rtd_prepare_wleg_kernel(RtdDev d, RtdRaw r) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;  // (c, l, ell)
  const long total = (long)d.C * d.L * d.P;
  if (idx >= total) return;
  const int ell = (int)(idx % d.P);
  const long cl = idx / d.P;
  const double fl = r.f[cl];
  const double g = (r.leg[cl * r.nleg_all + ell] - fl) / (1.0 - fl);  // (:323-324)
  const_cast<double*>(d.wleg)[idx] = (2.0 * ell + 1.0) * g;
}

}  // namespace

void rtd_launch_prepare(const RtdDev& d, const RtdRaw& r, hipStream_t s) {
  hipLaunchKernelGGL(rtd_prepare_columns_kernel, dim3((unsigned)((d.C + 63) / 64)), dim3(64), 0, s, d, r);
  const long total = (long)d.C * d.L * d.P;
  hipLaunchKernelGGL(rtd_prepare_wleg_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, d, r);
}
