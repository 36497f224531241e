// rtd_device.h -- shared declarations of the HIP kernels behind include/rtd.h (gfx950 only).
//
// Device-side layout (all float64, NP = per-hemisphere stream count padded to a power of two,
// Q2 = 2*NP; padded streams have mu = 2, w = 0 and decouple exactly):
//   Ym, Am   [C][M][L][NP][NP]  Y = L^-T Z and A = L Z (row = stream, column = eigen-index): everything the
//                               later stages need of the reference's eigenvector matrix
//                               G = [[Gp, Gm],[Gm, Gp]]  (_solve_for_gen_and_part_sols.py:192-198):
//                               Gp = (Y - A/k)/T, Gm = (Y + A/k)/T,  V = (Gp+Gm)/2 = Y/T, U = (Gp-Gm)/2 = -A/(kT),
//                               V^-1 = A^T T, U^-1 = -k Y^T T   (T = diag(sqrt(mu w)))
//   kk       [C][M][L][NP]      positive eigenvalues k;   K = [-k, +k]      (:186-187)
//   Bv       [C][M][L][Q2]      beam particular solution  [B+ ; B-]         (:209-231)
//   dq       [C][L][Ns][Q2]     isotropic-source particular solution as polynomial coefficient vectors about the TOP of the
//                               layer: v(tau*) = sum_q dq[q] (tau* - taus0[l])^q  (subroutines.py:746-862 builds it in the
//                               absolute depth; the construction is translation invariant, rtd_dd.h says why the origin moved)
//   spoly    [C][L][Ns]         the source polynomial itself, likewise about the layer's top (shifted in double-double on upload)
//   zneg     [C][L][NP]         first half of G^-1 [1/mu ; -1/mu]  (second half is its negative)
//   coef     [C][M][L][Q2]      BC coefficients [C- ; C+]                   (_solve_for_coeffs.py)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct RtdDev {
  // sizes
  int C, L, N, NP, P, M, Ns, NBDRF, beam;
  // Fourier-mode shard (secondary partition): local mode m stands for the mode m0 + mstep * m of mtot
  int m0, mstep, mtot;
  // layer shard of the eigen stage (the north star's "all-gather to stitch the boundary-condition
  // system"): only the layers [l0, l0 + ln) are decomposed by this launch; l0 = 0, ln = L without shards
  int l0, ln;
  // Chunk selection of the eigen stage (lean retained plans, rtd_api.hip): nsel > 0 -> only the wavefront chunks
  // chunk_sel[c][0 .. nsel) of column c are decomposed (a chunk = the 64/NP consecutive slots of the layer order lperm that share a
  // wavefront in the full launch; -1: nothing) -- the layers an evaluation touches, recomputed in the SAME wavefront composition as
  // the solve had them, hence to the same bits.  nsel = 0: every chunk.
  const int* chunk_sel;
  int nsel;
  int flags;  // bit 2: every chain of rtd_bc_mfma_kernel takes the register-resident pivoted elimination (RTD_BC_FORCE_PIVOT=2); bit 1: the tiled fused BC kernel hands every third chain to the pivoted kernels (RTD_BC_FORCE_HANDOVER); bit 0: the fused BC kernel skips its speculative elimination (test hook, env RTD_BC_FORCE_PIVOT)
  // quadrature (padded to NP)
  const double *mu, *w, *invmu, *S, *T;  // S = sqrt(w/mu), T = sqrt(w*mu) (1 for padding)
  // Legendre tables
  double* Y;   // [M][P][NP]   normalised associated Legendre functions at the quadrature nodes
  double* Y0;  // [C][M][P]    the same at -mu0 of each column
  const int* lperm;  // [C][L]  layer order of the eigen stage: layers of similar Jacobi sweep counts share a wavefront
  double* att;  // [C][L+1]    beam attenuation exp(-tau*_l / mu0) at the scaled layer boundaries (beam only)
  // per-column inputs
  const double *omega, *tau, *taus0, *scale, *wleg, *mu0, *I0, *phi0, *rescale;
  const double *bpos, *bneg;  // [C][M][NP]
  const double* spoly;        // [C][L][Ns]
  const double *bdrfq, *bdrfq0;  // [C][NBDRF][NP][NP], [C][NBDRF][NP]
  // intermediates
  double *Ym, *Am, *kk, *Bv, *dq, *zneg, *coef;
  double* vb;         // [C][L][4][NP]  the thermal particular solution v_l at the layer's own boundaries: up- and down-streams at its
                      //                top, then at its bottom (mode 0 only; read by rtd_bc_small_kernel instead of the polynomials)
  double* Ek;         // [C][M][L][NP]  exp(-k dtau*_l): the Stamnes-Conklin scaling factors
  double* Fws;  // BC workspace: [C][M][L-1][4 NP^2]: Wp, Wq, S, rho_t, rho_b, s per interface (rtd_bc.hip)
  // Fused evaluation (rtd_bc_mfma_kernel): when the evaluation points are the layer interfaces [0, tau_arr] the backward
  // sweep of the boundary-condition kernel forms the Fourier modes of the intensity there itself -- Y_l, A_l and the
  // coefficients are in its registers, the exponentials are E_l or 1 -- and the evaluation kernel only sums over the modes.
  double* um;         // [C][M][L+1][Q2]  u^m at the interfaces (null: not wanted)
  int* need_split;    // [C][M]  chains the tiled fused BC kernel hands to the pivoted row-per-lane kernels (64 streams)
  int* split_any;     // [1]     set when any chain of the window was handed over (then the fused evaluation is incomplete)
  int* sweeps;        // [1] max Jacobi sweeps (diagnostic)
  int* status;        // [1] device-side status flags (RTD_ST_*)
  int* col_status;    // [C] the numerical bits of `status` per column (which columns of a batch failed)
};

// device-side status bits (rtd_api.hip maps bit 0 to RTD_ERR_TAU_RANGE, the others to RTD_ERR_NUMERIC)
enum : int {
  RTD_ST_TAU = 1,     // an evaluation point lies outside [0, tau_arr[-1]] of its column
  RTD_ST_JACOBI = 2,  // the Jacobi iteration of some eigenproblem hit its sweep limit
  RTD_ST_CHOL = 4,    // non-positive pivot in a Cholesky factorisation / non-finite eigenvalue
  RTD_ST_BC = 8,      // non-finite boundary-condition coefficients (singular system)
  RTD_ST_BEAM = 16,   // non-finite beam particular solution (1/mu0 on an eigenvalue)
  // The numerical bits are raised in the low byte by Fourier mode 0 and in the next byte by the modes m > 0: the fluxes
  // and u0 come from mode 0 alone, and the reference returns them unharmed when only a higher mode fails (its u is NaN then).
  RTD_ST_HIGH_MODE_SHIFT = 8
};
#ifdef __HIPCC__
__device__ __forceinline__ void rtd_raise(const RtdDev& d, const int bit, const int mode, const int column) {
  const int b = mode == 0 ? bit : bit << RTD_ST_HIGH_MODE_SHIFT;
  atomicOr(d.status, b);
  atomicOr(d.col_status + column, b);
}
#endif

struct RtdEval {
  int ntau, nphi, antider;
  const double* tau;  // [C][ntau]
  const double* phi;  // [nphi]
  double *u, *u0, *fup, *fdn, *fdir, *ulast;  // device outputs (may be null)
  const double* um_in;  // [C][M][ntau][Q2] Fourier modes already formed by the boundary-condition kernel (else null)
  const int* run_if_set;  // not null: the evaluation kernel leaves at once unless this flag is set (see rtd_launch_eval)
  int mchunk;  // Fourier modes per pass of the evaluation kernel (set by rtd_launch_eval; 0: all)
};

// Nakajima-Tanaka corrections (rtd_nt.hip)
struct RtdNt {
  int nleg_all;
  const double* wfull;     // [C][L][nleg_all]  (2l+1) g_l of the full phase function
  const double* f;         // [C][L]            delta-M truncation fractions
  const double* ims_coef;  // [C][nleg_all]     Legendre series of the IMS correction
  const double* ims_par;   // [C][2]            scaled mu0, amplitude
  double* R;               // [C][2][2][NP][L]  other-layer sums: [antiderivative][up|down]
};

// raw (unprepared) per-column inputs on the device, for rtd_prep.hip
struct RtdRaw {
  int nleg_all;        // phase-function moments given per layer (>= P)
  const double *tau, *omega, *f;  // [C][L]    tau_arr, omega_arr, f_arr
  const double* leg;   // [C][L][nleg_all]     Leg_coeffs_all
  const double *mu0, *I0, *phi0;  // [C]
  const double *bpos, *bneg;      // [C][M][N] or null (zero)
  const double* spoly;            // [C][L][Ns] or null
};

// launchers (one per translation unit)
void rtd_launch_prepare(const RtdDev& d, const RtdRaw& r, hipStream_t s);
void rtd_launch_tables(const RtdDev& d, hipStream_t s, bool with_quad = true);  // with_quad: also the column-independent Y table
void rtd_launch_eig_small(const RtdDev& d, hipStream_t s);  // rtd_eig_small.hip: the one-lane-per-problem eigen kernel (NP = 4, 8)
void rtd_launch_eig(const RtdDev& d, hipStream_t s, int part);  // the fused eigen kernel runs as part 1 (0, 2: empty timing slots)
void rtd_launch_bc(const RtdDev& d, hipStream_t s, int part);   // 0 iface, 1 sweep
void rtd_launch_bc_small(const RtdDev& d, hipStream_t s);  // rtd_bc_small.hip: the fused kernel of the 2 ... 16-stream path
void rtd_launch_bc_tile2(const RtdDev& d, hipStream_t s);  // rtd_bc_tile2.hip: the lean 64-stream kernel (two wavefronts per SIMD)
void rtd_launch_bc_wide(const RtdDev& d, hipStream_t s, int part);  // rtd_bc_wide.hip: 66 ... 128 streams, four wavefronts per chain (0 iface, 1 sweep)
bool rtd_small_split();  // RTD_SMALL_SPLIT is set: NP <= 8 takes the separate interface / sweep / evaluation kernels
bool rtd_bc_fuses_eval(const RtdDev& d);  // the boundary-condition kernel chosen for d can fill d.um
void rtd_launch_eval(const RtdDev& d, const RtdEval& e, hipStream_t s);
void rtd_launch_nt_tables(const RtdDev& d, const RtdNt& nt, hipStream_t s);
void rtd_launch_nt_apply(const RtdDev& d, const RtdNt& nt, const RtdEval& e, hipStream_t s);
void rtd_launch_bdrf_modes(const RtdDev& d, int nphi, const double* rho_qq, const double* rho_q0, hipStream_t s);
void rtd_launch_export(const RtdDev& d, int col, double* GC, double* K, double* B, double* Gim, double* G,
                       hipStream_t s);
