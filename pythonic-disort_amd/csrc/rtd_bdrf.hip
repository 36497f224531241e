// rtd_bdrf.hip -- Fourier modes of a bidirectional reflectance function on the device.
//
// The reference takes the surface as a list of callables BDRF_Fourier_modes[m](mu, -mu') (pydisort.py:32-36) that its
// solver evaluates on the quadrature grid (_solve_for_coeffs.py:121-134); for a reflectance given as rho(mu, mu', dphi)
// its tests form each mode with an adaptive host quadrature per (mu, mu') pair (pydisotest/6_test.py:194-201):
//      q^m(mu, mu') = 1 / ((1 + delta_m0) pi)  Int_0^2pi  rho(mu, mu', dphi) cos(m dphi) d dphi .
// Here the caller samples rho on a uniform dphi grid (p = 0 .. nphi-1, dphi_p = 2 pi p / nphi) at the quadrature nodes
// and the device forms all modes with the trapezoid rule -- spectrally accurate for a periodic integrand:
//      q^m = (2 - delta_m0) / nphi  Sum_p rho_p cos(2 pi m p / nphi) ,
// written straight into the plan's bdrf tables [C][NBDRF][NP][NP] and [C][NBDRF][NP] (the mu0 column).
// One wavefront per sample vector; the cosine table lives in LDS; HBM-bound (each sample is read once per mode
// from L2, once from HBM).
#include "rtd_device.h"

namespace {

__global__ __launch_bounds__(64) void rtd_bdrf_modes_kernel(RtdDev d, int nphi, const double* __restrict__ rho_qq,
                                                            const double* __restrict__ rho_q0) {
  extern __shared__ double ctab[];  // cos(2 pi p / nphi)
  const int lane = threadIdx.x;
  for (int p = lane; p < nphi; p += 64) ctab[p] = cospi(2.0 * (double)p / (double)nphi);
  __syncthreads();
  const int N = d.N, NP = d.NP, NB = d.NBDRF;
  const long nqq = (long)d.C * N * N;
  long vec = blockIdx.x;
  const double* rho;
  double* out;
  long mstride;
  if (vec < nqq) {
    const long c = vec / ((long)N * N);
    const int i = (int)((vec / N) % N), j = (int)(vec % N);
    rho = rho_qq + vec * nphi;
    out = const_cast<double*>(d.bdrfq) + ((c * NB) * NP + i) * NP + j;
    mstride = (long)NP * NP;
  } else {
    vec -= nqq;
    const long c = vec / N;
    const int i = (int)(vec % N);
    rho = rho_q0 + vec * nphi;
    out = const_cast<double*>(d.bdrfq0) + (c * NB) * NP + i;
    mstride = NP;
  }
  for (int m = 0; m < NB; ++m) {
    double acc = 0.0;
    for (int p = lane; p < nphi; p += 64) acc += rho[p] * ctab[(int)(((long)m * p) % nphi)];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) out[m * mstride] = acc * ((m == 0) ? 1.0 : 2.0) / (double)nphi;
  }
}

}  // namespace

void rtd_launch_bdrf_modes(const RtdDev& d, int nphi, const double* rho_qq, const double* rho_q0, hipStream_t s) {
  const long nvec = (long)d.C * d.N * d.N + (rho_q0 ? (long)d.C * d.N : 0);
  hipLaunchKernelGGL(rtd_bdrf_modes_kernel, dim3((unsigned)nvec), dim3(64), (size_t)nphi * sizeof(double), s, d, nphi,
                     rho_qq, rho_q0);
}
