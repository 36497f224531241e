// TEST INFRASTRUCTURE (see fake_hip/hip/hip_runtime.h): the other side of the host-only sanitizer build of librtd's host code.
//  * the fake runtime's allocator: "device" memory = heap memory, with the bookkeeping hipMemGetInfo needs;
//  * SHADOW LAUNCHERS for the kernels of the other translation units: each touches exactly the extents the real kernel reads
//    and writes for the view (RtdDev / RtdEval / RtdNt) it is given -- the layouts of csrc/rtd_device.h -- so that a window
//    offset, an arena carve or a slot pointer that is wrong in rtd_api.hip is an AddressSanitizer report on the CPU.
// No numerics: what is written is a constant.
#include <atomic>
#include <cstdio>

#include "../../pythonic-disort_amd/csrc/rtd_device.h"

thread_local fake_idx blockIdx, threadIdx, blockDim, gridDim;

namespace {
std::atomic<size_t> g_used{0};
size_t total_bytes() {
  const char* s = getenv("FAKE_HIP_TOTAL");
  return s ? (size_t)atoll(s) : (size_t)8 << 30;
}
struct Header { size_t n, pad; };
volatile double g_sink;

void wr(const double* p, long n, double v = 0.5) {
  double* q = const_cast<double*>(p);
  for (long i = 0; i < n; ++i) q[i] = v;
}
void wri(const int* p, long n, int v = 0) {
  int* q = const_cast<int*>(p);
  for (long i = 0; i < n; ++i) q[i] = v;
}
void rd(const double* p, long n) {
  double s = 0.0;
  for (long i = 0; i < n; ++i) s += p[i];
  g_sink = s;
}
void rdi(const int* p, long n) {
  long s = 0;
  for (long i = 0; i < n; ++i) s += p[i];
  g_sink = (double)s;
}
}  // namespace

extern "C" {
hipError_t hipMalloc(void** p, size_t n) {
  if (g_used.load() + n > total_bytes()) {
    *p = nullptr;
    return hipErrorOutOfMemory;
  }
  Header* h = (Header*)std::malloc(sizeof(Header) + n);
  if (!h) return hipErrorOutOfMemory;
  h->n = n;
  g_used += n;
  *p = h + 1;
  return hipSuccess;
}
hipError_t hipFree(void* p) {
  if (!p) return hipSuccess;
  Header* h = (Header*)p - 1;
  g_used -= h->n;
  std::free(h);
  return hipSuccess;
}
hipError_t hipHostMalloc(void** p, size_t n, unsigned) {
  *p = std::malloc(n);
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipHostFree(void* p) {
  std::free(p);
  return hipSuccess;
}
hipError_t hipMemGetInfo(size_t* free_b, size_t* total_b) {
  const size_t t = total_bytes(), u = g_used.load();
  if (free_b) *free_b = t > u ? t - u : 0;
  if (total_b) *total_b = t;
  return hipSuccess;
}
}

// ---- shadow launchers: extents per csrc/rtd_device.h, for the d.C columns of the view ------------------------------------------
void rtd_launch_prepare(const RtdDev& d, const RtdRaw& r, hipStream_t) {
  const long C = d.C, L = d.L;
  rd(r.tau, C * L); rd(r.omega, C * L); rd(r.f, C * L); rd(r.leg, C * L * r.nleg_all);
  rd(r.mu0, C); rd(r.I0, C); rd(r.phi0, C);
  if (r.bpos) rd(r.bpos, C * d.M * d.N);
  if (r.bneg) rd(r.bneg, C * d.M * d.N);
  if (r.spoly) rd(r.spoly, C * L * d.Ns);
  wr(d.omega, C * L); wr(d.taus0, C * (L + 1)); wr(d.scale, C * L, 1.0); wr(d.wleg, C * L * d.P);
  for (long c = 0; c < C; ++c)
    for (long l = 0; l < L; ++l) {
      const_cast<double*>(d.tau)[c * L + l] = r.tau[c * L + l];
      const_cast<int*>(d.lperm)[c * L + l] = (int)l;
    }
  wr(d.mu0, C); wr(d.I0, C); wr(d.phi0, C); wr(d.rescale, C, 1.0);
  wr(d.bpos, C * d.M * d.NP, 0.0); wr(d.bneg, C * d.M * d.NP, 0.0);
  if (d.Ns > 0) wr(d.spoly, C * L * d.Ns);
}

void rtd_launch_tables(const RtdDev& d, hipStream_t, bool with_quad) {
  if (with_quad) {
    rd(d.mu, d.NP);
    wr(d.Y, (long)d.M * d.P * d.NP);
  }
  if (d.beam) {
    rd(d.mu0, d.C); rd(d.taus0, (long)d.C * (d.L + 1));
    wr(d.Y0, (long)d.C * d.M * d.P); wr(d.att, (long)d.C * (d.L + 1));
  }
}

static void eig_shadow(const RtdDev& d) {
  const long C = d.C, L = d.L, M = d.M, NP = d.NP, Q2 = 2 * NP, CML = C * M * L;
  rd(d.Y, M * d.P * NP); rdi(d.lperm, C * L); rd(d.omega, C * L); rd(d.wleg, C * L * d.P);
  rd(d.invmu, NP); rd(d.S, NP); rd(d.T, NP); rd(d.mu, NP); rd(d.w, NP); rd(d.taus0, C * (L + 1));
  if (d.beam) { rd(d.Y0, C * M * d.P); rd(d.mu0, C); rd(d.I0, C); }
  if (d.Ns > 0) rd(d.spoly, C * L * d.Ns);
  if (d.nsel > 0) {
    const int nchunk = (d.L + 64 / d.NP - 1) / (64 / d.NP);
    for (long i = 0; i < C * d.nsel; ++i)
      if (d.chunk_sel[i] < -1 || d.chunk_sel[i] >= nchunk) { fprintf(stderr, "shadow eigen stage: chunk %d outside [-1, %d)\n", d.chunk_sel[i], nchunk); abort(); }
  }
  wr(d.Ym, CML * NP * NP); wr(d.Am, CML * NP * NP); wr(d.kk, CML * NP, 1.0); wr(d.Ek, CML * NP); wr(d.Bv, CML * Q2);
  wr(d.zneg, C * L * NP);
  if (d.Ns > 0) { wr(d.dq, C * L * d.Ns * Q2); wr(d.vb, C * L * 4 * NP); }
  rdi(d.sweeps, 1); rdi(d.status, 1); rdi(d.col_status, C);
}
void rtd_launch_eig_small(const RtdDev& d, hipStream_t) { eig_shadow(d); }
void rtd_launch_eig(const RtdDev& d, hipStream_t, int part) {
  if (part == 1) eig_shadow(d);
}

bool rtd_small_split() { return false; }
bool rtd_bc_fuses_eval(const RtdDev& d) { return d.NP == 16 || d.NP == 32 || d.NP <= 8; }

void rtd_launch_bc(const RtdDev& d, hipStream_t, int part) {
  if (part != 1) return;
  const long C = d.C, L = d.L, M = d.M, NP = d.NP, Q2 = 2 * NP, CML = C * M * L;
  rd(d.Ym, CML * NP * NP); rd(d.Am, CML * NP * NP); rd(d.kk, CML * NP); rd(d.Ek, CML * NP); rd(d.Bv, CML * Q2);
  rd(d.bpos, C * M * NP); rd(d.bneg, C * M * NP);
  if (d.beam) { rd(d.att, C * (L + 1)); rd(d.mu0, C); rd(d.I0, C); }
  if (d.Ns > 0) rd(d.vb, C * L * 4 * NP);
  if (d.NBDRF > 0) { rd(d.bdrfq, C * d.NBDRF * NP * NP); rd(d.bdrfq0, C * d.NBDRF * NP); }
  wr(d.coef, CML * Q2);
  if (L > 1) wr(d.Fws, C * M * (L - 1) * Q2 * Q2);
  if (d.um) wr(d.um, C * M * (L + 1) * Q2);
  wri(d.need_split, C * M); wri(d.split_any, 1);
  rdi(d.status, 1); rdi(d.col_status, C);
}
void rtd_launch_bc_small(const RtdDev& d, hipStream_t s) { rtd_launch_bc(d, s, 1); }
void rtd_launch_bc_tile2(const RtdDev& d, hipStream_t s) { rtd_launch_bc(d, s, 1); }
void rtd_launch_bc_wide(const RtdDev& d, hipStream_t s, int part) { rtd_launch_bc(d, s, part); }

void rtd_launch_eval(const RtdDev& d, const RtdEval& e, hipStream_t) {
  const long C = d.C, L = d.L, M = d.M, NP = d.NP, Q2 = 2 * NP, CML = C * M * L, Qr = 2 * d.N, nt = e.ntau, np = e.nphi;
  rd(e.tau, C * nt);
  if (np > 0) rd(e.phi, np);
  rd(d.tau, C * L); rd(d.taus0, C * (L + 1)); rd(d.scale, C * L); rd(d.rescale, C); rd(d.phi0, C);
  if (e.run_if_set) rdi(e.run_if_set, 1);
  if (e.um_in) rd(e.um_in, C * M * nt * Q2);
  else {
    rd(d.Ym, CML * NP * NP); rd(d.Am, CML * NP * NP); rd(d.kk, CML * NP); rd(d.coef, CML * Q2); rd(d.Bv, CML * Q2);
    if (d.Ns > 0) rd(d.dq, C * L * d.Ns * Q2);
  }
  // Values that carry the column's identity (its first single-scattering albedo) and the position inside the column's block: a
  // slot of a gathered array that holds another rank's or another column's results is then a VALUE mismatch in the two-rank
  // CPU test (tests/test_host_asan.py), not only an address error
  if (e.u)
    for (long c = 0; c < C; ++c)
      for (long k = 0; k < Qr * nt * np; ++k) e.u[c * Qr * nt * np + k] = d.omega[c * L] + 1e-3 * (double)k;
  // self-test of the harness (tests/test_host_asan.py): 32 KB past the end of the window's u, as a wrong window offset or a
  // short evaluation buffer would produce -- the sanitizer must stop the run
  if (e.u && getenv("FAKE_HIP_FAULT") && std::strcmp(getenv("FAKE_HIP_FAULT"), "eval_u_overrun") == 0) wr(e.u + C * Qr * nt * np, 1 << 12);
  if (e.u0) wr(e.u0, C * Qr * nt);
  if (e.ulast) wr(e.ulast, C * Qr * nt);
  for (long c = 0; c < C; ++c)
    for (long k = 0; k < nt; ++k) {
      if (e.fup) e.fup[c * nt + k] = 2.0 * d.omega[c * L] + 1e-3 * (double)k;
      if (e.fdn) e.fdn[c * nt + k] = 3.0 * d.omega[c * L] + 1e-3 * (double)k;
      if (e.fdir) e.fdir[c * nt + k] = 4.0 * d.omega[c * L] + 1e-3 * (double)k;
    }
  rdi(d.status, 1);
}

void rtd_launch_nt_tables(const RtdDev& d, const RtdNt& nt, hipStream_t) {
  const long C = d.C, L = d.L;
  rd(nt.wfull, C * L * nt.nleg_all); rd(nt.f, C * L); rd(nt.ims_coef, C * nt.nleg_all); rd(nt.ims_par, C * 2);
  wr(nt.R, C * 4 * d.NP * L);
}
void rtd_launch_nt_apply(const RtdDev& d, const RtdNt& nt, const RtdEval& e, hipStream_t) {
  const long C = d.C, L = d.L, Qr = 2 * d.N;
  rd(nt.R, C * 4 * d.NP * L); rd(nt.wfull, C * L * nt.nleg_all); rd(nt.f, C * L); rd(nt.ims_coef, C * nt.nleg_all); rd(nt.ims_par, C * 2);
  rd(e.tau, C * e.ntau);
  if (e.u) { rd(e.u, C * Qr * e.ntau * e.nphi); wr(e.u, C * Qr * e.ntau * e.nphi); }
}

void rtd_launch_bdrf_modes(const RtdDev& d, int nphi, const double* rho_qq, const double* rho_q0, hipStream_t) {
  const long C = d.C, N = d.N, NP = d.NP;
  rd(rho_qq, C * N * N * nphi);
  if (rho_q0) rd(rho_q0, C * N * nphi);
  wr(d.bdrfq, C * d.NBDRF * NP * NP);
  wr(d.bdrfq0, C * d.NBDRF * NP);
}

void rtd_launch_export(const RtdDev& d, int col, double* GC, double* K, double* B, double* Gim, double* G, hipStream_t) {
  const long L = d.L, M = d.M, NP = d.NP, Q2 = 2 * NP, Qr = 2 * d.N, ML = M * L;
  if (col < 0 || col >= d.C) { fprintf(stderr, "shadow export: column %d outside the view of %d\n", col, d.C); abort(); }
  rd(d.Ym + (long)col * ML * NP * NP, ML * NP * NP); rd(d.Am + (long)col * ML * NP * NP, ML * NP * NP);
  rd(d.kk + (long)col * ML * NP, ML * NP); rd(d.coef + (long)col * ML * Q2, ML * Q2); rd(d.Bv + (long)col * ML * Q2, ML * Q2);
  if (d.Ns > 0) rd(d.zneg + (long)col * L * NP, L * NP);
  wr(GC, ML * Qr * Qr); wr(G, ML * Qr * Qr); wr(K, ML * Qr); wr(B, ML * Qr); wr(Gim, L * Qr);
}
