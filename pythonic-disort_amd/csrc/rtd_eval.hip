// rtd_eval.hip -- evaluation of the solution at user (tau, phi) points and export of the reference's tensors.
//
// Replaces the closures u, u0, flux_up, flux_down of _assemble_intensity_and_fluxes
// (src/PythonicDISORT/_assemble_intensity_and_fluxes.py:170-330, :334-433, :446-524, :527-613):
// layer lookup (:185), delta-M tau mapping (:190-195), non-positive exponents (:197-203),
// GC exp(K dtau) + B exp(-tau*/mu0) + v(tau*) (:221-254), Fourier sum (:256-260), fluxes (:519, :601),
// rescale (:262), and their is_antiderivative_wrt_tau variants.  One workgroup per (column, tau point);
// the M x Q x Q temporary of the reference is never formed: G rows are streamed once per point.
#include <algorithm>

#include "rtd_device.h"

namespace {

constexpr int EVAL_THREADS = 256;

// Transposed reduction inside an NP-lane group: every lane enters with NP partial terms v[0..NP) (term i belongs to
// row i) and leaves with the complete sum of row `jj` -- NP-1 cross-lane moves instead of NP log2(NP).
template <int NP, int O>
struct TransposeStep {
  static __device__ __forceinline__ void run(double (&v)[NP], int jj) {
    const bool hi = (jj & O) != 0;
#pragma unroll
    for (int i = 0; i < O; ++i) {  // compile-time bounds: v[] stays in registers
      const double keep = hi ? v[i + O] : v[i];
      const double send = hi ? v[i] : v[i + O];
      v[i] = keep + __shfl_xor(send, O, NP);
    }
    TransposeStep<NP, O / 2>::run(v, jj);
  }
};
template <int NP>
struct TransposeStep<NP, 0> {
  static __device__ __forceinline__ void run(double (&)[NP], int) {}
};
template <int NP>
__device__ __forceinline__ double transpose_reduce(double (&v)[NP], int jj) {
  TransposeStep<NP, NP / 2>::run(v, jj);
  return v[0];
}

template <int NP>
__global__ __launch_bounds__(EVAL_THREADS) void rtd_eval_kernel(RtdDev d, RtdEval ev) {
  constexpr int Q = 2 * NP;
  extern __shared__ double smem[];
  const int M = d.M, L = d.L, N = d.N;
  // The Fourier modes are taken MC at a time (MC = M unless 2 M Q doubles exceed 64 KiB of LDS: more than 32 modes at 128
  // streams), the sums over m accumulated from chunk to chunk.
  const int MC = ev.mchunk > 0 && ev.mchunk < M ? ev.mchunk : M;
  double* e_s = smem;           // [MC][Q] scaled exponentials times BC coefficients
  double* um = smem + MC * Q;   // [MC][Q] Fourier modes of the intensity at this point
  __shared__ int s_l;
  __shared__ double s_ts;
  if (ev.run_if_set != nullptr && *ev.run_if_set == 0) return;  // (the Fourier-sum kernel has done this window)
  const int t = (int)(blockIdx.x % ev.ntau), c = (int)(blockIdx.x / ev.ntau), tid = threadIdx.x;
  const double tau = ev.tau[(long)c * ev.ntau + t];
  const double* tau_arr = d.tau + (long)c * L;
  const double* ts0 = d.taus0 + (long)c * (L + 1);
  if (tid == 0) {
    int l = 0;
    // argmax(tau <= tau_arr)  (:185); with the fused evaluation the points are the interfaces [0, tau_arr] (checked by
    // rtd_plan_set_eval_points): point t lies in layer max(t - 1, 0), no search (a chain of dependent loads) needed
    if (ev.um_in != nullptr) l = t > 0 ? t - 1 : 0;
    else
      while (l < L - 1 && !(tau <= tau_arr[l])) ++l;
    if (!(tau >= 0.0) || !(tau <= tau_arr[L - 1])) atomicOr(d.status, 1);
    s_l = l;
    s_ts = ts0[l + 1] - (tau_arr[l] - tau) * d.scale[(long)c * L + l];  // (:190-195)
  }
  __syncthreads();
  const int l = s_l;
  const double ts = s_ts;
  const double sc = d.scale[(long)c * L + l];
  const bool antider = ev.antider != 0;
  const double dtop = ts - ts0[l], dbot = ts0[l + 1] - ts;
  const bool beam = d.beam != 0;
  const double mu0 = beam ? d.mu0[c] : 1.0;
  for (int mb = 0; mb < M; mb += MC) {  // (one pass unless the modes do not fit the LDS)
  const int mc = min(MC, M - mb);
  if (mb > 0) __syncthreads();  // the previous chunk's modes have been summed
  if (ev.um_in != nullptr) {
    // the boundary-condition kernel has formed u^m at this point (a layer interface) already: only the sums are left
    for (int idx = tid; idx < mc * Q; idx += EVAL_THREADS) {
      const int m = idx / Q, i = idx % Q;
      um[idx] = ev.um_in[(((long)c * M + mb + m) * ev.ntau + t) * Q + i];
    }
  } else {
  // exponent * coefficient, both halves non-positive exponents (:197-203)
  for (int idx = tid; idx < mc * NP; idx += EVAL_THREADS) {
    const int m = idx / NP, jj = idx % NP;
    const long ml = ((long)c * M + mb + m) * L + l;
    const double k = d.kk[ml * NP + jj];
    double en = exp(-k * dtop) * d.coef[ml * Q + jj];
    double ep = exp(-k * dbot) * d.coef[ml * Q + NP + jj];
    if (antider) {  // / (scale_tau K), K = -k | +k  (:221-227)
      en /= (-k * sc);
      ep /= (k * sc);
    }
    // Gp en + Gm ep = [Y (en+ep) - A (en-ep)/k]/T and Gm en + Gp ep = [Y (en+ep) + A (en-ep)/k]/T
    e_s[m * Q + jj] = en + ep;
    e_s[m * Q + NP + jj] = (en - ep) / k;
  }
  __syncthreads();
  double bfac = beam ? exp(-ts / mu0) : 0.0;
  if (antider) bfac /= (-sc / mu0);
  // u^m_i = sum_j G_ij e_j + B_i exp(-tau*/mu0) (+ v_i for m = 0): NP lanes per (m, stream i); the up- and the
  // down-stream of a quadrature node share the two row sums  P = Y_i . (en+ep),  Qs = A_i . (en-ep)/k
  const int grp = tid / NP, jj = tid % NP;
  constexpr int NGRP = EVAL_THREADS / NP;
  for (int m = grp; m < mc; m += NGRP) {  // one NP-lane group per Fourier mode; lane jj ends up owning node i = jj
    const long ml = ((long)c * M + mb + m) * L + l;
    const double* Yl = d.Ym + ml * NP * NP + jj;
    const double* Al = d.Am + ml * NP * NP + jj;
    const double e1 = e_s[m * Q + jj], e2 = e_s[m * Q + NP + jj];
    double ps[NP], qs[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      ps[i] = Yl[i * NP] * e1;
      qs[i] = Al[i * NP] * e2;
    }
    const double P = transpose_reduce<NP>(ps, jj), Qs = transpose_reduce<NP>(qs, jj);
    const double it = 1.0 / d.T[jj];
    double vu = (P - Qs) * it, vd = (P + Qs) * it;
    if (beam) {
      vu += d.Bv[ml * Q + jj] * bfac;
      vd += d.Bv[ml * Q + NP + jj] * bfac;
    }
    if (d.m0 + d.mstep * (mb + m) == 0 && d.Ns > 0) {  // isotropic-source particular solution (subroutines.py:786-862)
      // The coefficient vectors dq are about the TOP of the layer (rtd_dd.h): v(x) = sum_q dq[q] x^q, x = ts - ts0[l] = dtop.
      const double* dq = d.dq + ((long)c * L + l) * d.Ns * Q;
      if (!antider) {
        double tp = 1.0;
        for (int q = 0; q < d.Ns; ++q) {
          vu += dq[q * Q + jj] * tp;
          vd += dq[q * Q + NP + jj] * tp;
          tp *= dtop;
        }
      } else {
        // The reference's antiderivative is the one of the ABSOLUTE form, sum_q b_q ts^(q+1) / ((q + 1) scale) (:792-793, :812-815,
        // :855-858): an antiderivative whose constant is fixed by tau = 0, not by the layer.  b_q = sum_{i >= q} dq[i] C(i, q)
        // (-ts0[l])^(i - q) re-expands the local coefficients (the cancellation of the absolute form comes back with it: it is
        // what the reference's closure is defined to return).
        const double t0 = ts0[l];
        double tp = ts;
        for (int q = 0; q < d.Ns; ++q) {
          double bu = 0.0, bd = 0.0, binom = 1.0, pw = 1.0;  // C(i, q) and (-t0)^(i - q) for i = q, q + 1, ...
          for (int i = q; i < d.Ns; ++i) {
            bu += dq[i * Q + jj] * binom * pw;
            bd += dq[i * Q + NP + jj] * binom * pw;
            binom = binom * (double)(i + 1) / (double)(i + 1 - q);
            pw *= -t0;
          }
          const double f = tp / ((q + 1) * sc);
          vu += bu * f;
          vd += bd * f;
          tp *= ts;
        }
      }
    }
    um[m * Q + jj] = vu;
    um[m * Q + NP + jj] = vd;
  }
  }
  __syncthreads();
  const double rescale = d.rescale[c];
  const int Qr = 2 * N;
  // intensity: Fourier sum over m with cos(m (phi0 - phi))  (:256-262)
  if (ev.u != nullptr) {
    const double phi0 = d.phi0[c];
    for (int idx = tid; idx < Qr * ev.nphi; idx += EVAL_THREADS) {
      const int ir = idx / ev.nphi, p = idx % ev.nphi;
      const int i2 = ir < N ? ir : NP + (ir - N);
      const double dl = phi0 - ev.phi[p];
      // sum_k um[k] cos((m0 + k mstep) dl): Chebyshev recurrence in steps of mstep (m0 = 0, mstep = 1 without mode shards)
      const double cs = cos(d.mstep * dl);
      const int mf = d.m0 + d.mstep * mb;  // the chunk's first mode
      double ckm1 = cos((mf - d.mstep) * dl), ck = cos(mf * dl);
      double acc = 0.0;
      for (int m = 0; m < mc; ++m) {
        acc += um[m * Q + i2] * ck;
        const double cn = 2.0 * cs * ck - ckm1;
        ckm1 = ck;
        ck = cn;
      }
      double* out = ev.u + (((long)c * Qr + ir) * ev.ntau + t) * ev.nphi + p;  // (the same thread owns it in every chunk)
      *out = mb == 0 ? rescale * acc : *out + rescale * acc;
    }
  }
  if (tid < Qr) {
    const int i2 = tid < N ? tid : NP + (tid - N);
    // with mode shards the zeroth / last mode belongs to one shard only; the others contribute zeros to the sum
    const bool own0 = d.m0 == 0, ownlast = d.m0 + d.mstep * (M - 1) == d.mtot - 1;
    if (ev.u0 != nullptr && mb == 0) ev.u0[((long)c * Qr + tid) * ev.ntau + t] = own0 ? rescale * um[i2] : 0.0;
    if (ev.ulast != nullptr && mb + mc == M)
      ev.ulast[((long)c * Qr + tid) * ev.ntau + t] = ownlast ? rescale * um[(mc - 1) * Q + i2] : 0.0;
  }
  // fluxes from the zeroth mode (:519, :568-601)
  if (mb == 0 && tid == 0 && (ev.fup != nullptr || ev.fdn != nullptr || ev.fdir != nullptr)) {
    double fu = 0.0, fd = 0.0;
    for (int i = 0; i < N; ++i) {
      const double mw = d.mu[i] * d.w[i];
      fu += mw * um[i];
      fd += mw * um[NP + i];
    }
    double direct = 0.0, direct_s = 0.0;
    if (beam) {
      const double I0 = d.I0[c];
      direct = I0 * mu0 * exp(-tau / mu0);
      direct_s = I0 * mu0 * exp(-ts / mu0);
      if (antider) {
        direct *= -mu0;
        direct_s /= (-sc / mu0);
      }
    }
    const long o = (long)c * ev.ntau + t;
    const double own = (d.m0 == 0) ? 1.0 : 0.0;  // fluxes come from the zeroth mode (mode shards: its owner only)
    if (ev.fup != nullptr) ev.fup[o] = own * rescale * 2.0 * M_PI * fu;
    if (ev.fdn != nullptr) ev.fdn[o] = own * rescale * (2.0 * M_PI * fd + direct_s - direct);
    if (ev.fdir != nullptr) ev.fdir[o] = own * rescale * direct;
  }
  }  // chunks of modes
}

// The throughput path: the boundary-condition kernel has left u^m at the points already (the layer interfaces [0, tau_arr],
// checked by rtd_plan_set_eval_points), only the Fourier sum (:256-262), the zeroth mode and the fluxes (:519, :568-601)
// remain.  One thread per (point, stream): a workgroup takes FT_T consecutive points of one column, so that every mode's
// slice is one contiguous read; no layer search, no LDS staging of the modes.  Same arithmetic, in the same order, as the
// corresponding part of rtd_eval_kernel (which stays the kernel of every other evaluation).
// points of a column per workgroup: 8 at 32 / 64 streams (one / two (point, stream) pairs per thread); at 8 / 16 streams 32 / 16 of
// them, so that the 9 interfaces of a cfg3 column are one workgroup, not a full one and a nearly empty one
__host__ __device__ constexpr int ft_points(int np) { return np >= 16 ? 8 : 256 / (2 * np); }
template <int NP>
__global__ __launch_bounds__(EVAL_THREADS) void rtd_fourier_kernel(RtdDev d, RtdEval ev) {
  constexpr int Q = 2 * NP, FT_T = ft_points(NP);
  static_assert(FT_T * Q <= 2 * EVAL_THREADS, "at most two (point, stream) pairs per thread");
  const int M = d.M, L = d.L, N = d.N, Qr = 2 * N;
  // a chain of this window went to the row-per-lane kernels, which leave no u^m: the evaluation kernel takes the window
  if (d.split_any != nullptr && *d.split_any != 0) return;
  const int nchunk = (ev.ntau + FT_T - 1) / FT_T;
  const int c = (int)(blockIdx.x / nchunk), t0 = (int)(blockIdx.x % nchunk) * FT_T;
  const int nt = min(FT_T, ev.ntau - t0), tid = threadIdx.x;
  __shared__ double s_um0[FT_T][Q];  // zeroth mode of the chunk's points (fluxes)
  const double rescale = d.rescale[c];
  const double phi0 = d.phi0[c];
  const bool own0 = d.m0 == 0, ownlast = d.m0 + d.mstep * (M - 1) == d.mtot - 1;
  for (int idx = tid; idx < nt * Qr; idx += EVAL_THREADS) {
    const int tl = idx / Qr, ir = idx % Qr, t = t0 + tl;
    const int i2 = ir < N ? ir : NP + (ir - N);
    const double* um = ev.um_in + ((long)c * M * ev.ntau + t) * Q + i2;  // mode m at um[m * ntau * Q]
    const long mstride = (long)ev.ntau * Q;
    const double first = um[0], last = um[(long)(M - 1) * mstride];
    s_um0[tl][i2] = first;
    if (ev.u0 != nullptr) ev.u0[((long)c * Qr + ir) * ev.ntau + t] = own0 ? rescale * first : 0.0;
    if (ev.ulast != nullptr) ev.ulast[((long)c * Qr + ir) * ev.ntau + t] = ownlast ? rescale * last : 0.0;
    if (ev.u != nullptr) {
      for (int p0 = 0; p0 < ev.nphi; p0 += 4) {  // four azimuths per pass over the modes
        double cs[4], ckm1[4], ck[4], acc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const double dl = phi0 - ev.phi[min(p0 + q, ev.nphi - 1)];
          // sum_k um[k] cos((m0 + k mstep) dl): Chebyshev recurrence in steps of mstep
          cs[q] = cos(d.mstep * dl);
          ckm1[q] = cos((d.m0 - d.mstep) * dl);
          ck[q] = cos(d.m0 * dl);
          acc[q] = 0.0;
        }
        // (sixteen modes per pass: their loads are independent of the recurrence and go out together -- one load per trip
        //  made the kernel a chain of M memory latencies)
        for (int m0 = 0; m0 < M; m0 += 16) {
          double v[16];
#pragma unroll
          for (int k = 0; k < 16; ++k) v[k] = um[(long)min(m0 + k, M - 1) * mstride];
#pragma unroll
          for (int k = 0; k < 16; ++k)
            if (m0 + k < M) {
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                acc[q] += v[k] * ck[q];
                const double cn = 2.0 * cs[q] * ck[q] - ckm1[q];
                ckm1[q] = ck[q];
                ck[q] = cn;
              }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (p0 + q < ev.nphi) ev.u[(((long)c * Qr + ir) * ev.ntau + t) * ev.nphi + p0 + q] = rescale * acc[q];
      }
    }
  }
  __syncthreads();
  if (tid < nt && (ev.fup != nullptr || ev.fdn != nullptr || ev.fdir != nullptr)) {
    const int t = t0 + tid, l = t > 0 ? t - 1 : 0;
    double fu = 0.0, fd = 0.0;
    for (int i = 0; i < N; ++i) {
      const double mw = d.mu[i] * d.w[i];
      fu += mw * s_um0[tid][i];
      fd += mw * s_um0[tid][NP + i];
    }
    double direct = 0.0, direct_s = 0.0;
    if (d.beam != 0) {
      const double tau = ev.tau[(long)c * ev.ntau + t];
      const double ts = d.taus0[(long)c * (L + 1) + l + 1] - (d.tau[(long)c * L + l] - tau) * d.scale[(long)c * L + l];  // (:190-195)
      const double I0 = d.I0[c], mu0 = d.mu0[c];
      direct = I0 * mu0 * exp(-tau / mu0);
      direct_s = I0 * mu0 * exp(-ts / mu0);
    }
    const long o = (long)c * ev.ntau + t;
    const double own = (d.m0 == 0) ? 1.0 : 0.0;  // fluxes come from the zeroth mode (mode shards: its owner only)
    if (ev.fup != nullptr) ev.fup[o] = own * rescale * 2.0 * M_PI * fu;
    if (ev.fdn != nullptr) ev.fdn[o] = own * rescale * (2.0 * M_PI * fd + direct_s - direct);
    if (ev.fdir != nullptr) ev.fdir[o] = own * rescale * direct;
  }
}

// the reference's tensors of one column, in the reference's layout (unpadded)
__global__ void rtd_export_kernel(RtdDev d, int col, double* GC, double* K, double* B, double* Gim, double* G) {
  const int N = d.N, NP = d.NP, Qr = 2 * N, Q = 2 * NP, M = d.M, L = d.L;
  const long total = (long)M * L * Qr * Qr;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int cj = (int)(idx % Qr), ri = (int)((idx / Qr) % Qr);
    const long ml = idx / ((long)Qr * Qr);
    const long mlg = (long)col * M * L + ml;
    const bool rup = ri < N, cneg = cj < N;
    const int i = rup ? ri : ri - N, j = cneg ? cj : cj - N;
    // G = [[Gp, Gm],[Gm, Gp]],  Gp = (Y - A/k)/T,  Gm = (Y + A/k)/T
    const double yv = d.Ym[(mlg * NP + i) * NP + j], av = d.Am[(mlg * NP + i) * NP + j] / d.kk[mlg * NP + j];
    const double g = ((rup == cneg) ? yv - av : yv + av) / d.T[i];
    const double cf = d.coef[mlg * Q + (cneg ? j : NP + j)];
    if (G) G[idx] = g;
    if (GC) GC[idx] = g * cf;
    if (ri == 0) {
      const double k = d.kk[mlg * NP + j];
      if (K) K[ml * Qr + cj] = cneg ? -k : k;
      if (B) B[ml * Qr + cj] = d.beam ? d.Bv[mlg * Q + (cneg ? j : NP + j)] : 0.0;
      if (Gim && ml < L && d.Ns > 0) {
        const double z = d.zneg[((long)col * L + ml) * NP + j];
        Gim[ml * Qr + cj] = cneg ? z : -z;
      }
    }
  }
}

}  // namespace

void rtd_launch_eval(const RtdDev& d, const RtdEval& e, hipStream_t s) {
  if (e.um_in != nullptr && e.antider == 0 && rtd_bc_fuses_eval(d)) {  // u^m is there already: sums only
    const int ftt = ft_points(d.NP);
    const dim3 g((unsigned)((long)d.C * ((e.ntau + ftt - 1) / ftt)));
    if (d.NP == 4) hipLaunchKernelGGL(rtd_fourier_kernel<4>, g, dim3(EVAL_THREADS), 0, s, d, e);
    else if (d.NP == 8) hipLaunchKernelGGL(rtd_fourier_kernel<8>, g, dim3(EVAL_THREADS), 0, s, d, e);
    else if (d.NP == 16) hipLaunchKernelGGL(rtd_fourier_kernel<16>, g, dim3(EVAL_THREADS), 0, s, d, e);
    else hipLaunchKernelGGL(rtd_fourier_kernel<32>, g, dim3(EVAL_THREADS), 0, s, d, e);
    // the tiled kernel may have handed chains to the row-per-lane kernels (singular carry blocks; a test hook): those leave
    // no u^m, the flag is set, the Fourier-sum kernel has left at once and the evaluation kernel does the window
    static const bool tiled16 = getenv("RTD_BC_TILED") != nullptr;
    if (d.NP != 32 && !tiled16) return;
    RtdEval g2 = e;
    g2.um_in = nullptr;
    g2.run_if_set = d.split_any;
    rtd_launch_eval(d, g2, s);
    return;
  }
  const dim3 grid((unsigned)((long)e.ntau * d.C));  // 1-D: no 65535 limit on the column count
  // modes per pass of the evaluation kernel: all of them unless 2 M 2 NP doubles exceed 64 KiB of LDS
  RtdEval e2 = e;
  e2.mchunk = std::min<int>(d.M, (64 << 10) / (int)(2 * 2 * d.NP * sizeof(double)));
  const size_t shm = (size_t)2 * e2.mchunk * 2 * d.NP * sizeof(double);
  switch (d.NP) {
    case 4: hipLaunchKernelGGL(rtd_eval_kernel<4>, grid, dim3(EVAL_THREADS), shm, s, d, e2); break;
    case 8: hipLaunchKernelGGL(rtd_eval_kernel<8>, grid, dim3(EVAL_THREADS), shm, s, d, e2); break;
    case 16: hipLaunchKernelGGL(rtd_eval_kernel<16>, grid, dim3(EVAL_THREADS), shm, s, d, e2); break;
    case 32: hipLaunchKernelGGL(rtd_eval_kernel<32>, grid, dim3(EVAL_THREADS), shm, s, d, e2); break;
    case 64: hipLaunchKernelGGL(rtd_eval_kernel<64>, grid, dim3(EVAL_THREADS), shm, s, d, e2); break;
    default: break;
  }
}

void rtd_launch_export(const RtdDev& d, int col, double* GC, double* K, double* B, double* Gim, double* G,
                       hipStream_t s) {
  hipLaunchKernelGGL(rtd_export_kernel, dim3(256), dim3(256), 0, s, d, col, GC, K, B, Gim, G);
}
