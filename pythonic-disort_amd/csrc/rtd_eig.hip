// rtd_eig.hip -- per (column, Fourier mode, layer) eigen stage on gfx950.
//
// Replaces _solve_for_gen_and_part_sols (src/PythonicDISORT/_solve_for_gen_and_part_sols.py:5-243):
//   Legendre tables (:96-109)  -> rtd_tables_* kernels (normalised three-term recurrences)
//   D+/D-, alpha, beta (:123-135), eig((alpha-beta)(alpha+beta)) (:179-183), G blocks (:186-198),
//   beam particular solution (:143-152, :209-231), G^-1 [1/mu;-1/mu] (:203-205 + _assemble.py:124)
//   and the isotropic-source particular solution coefficients (subroutines.py:746-862)
//   -> rtd_eigen_kernel<NP, JV> (one fused kernel; JV = 2: the default; JV = 3: the same with the assembly of Pm, Qm on the
//      matrix cores, NP = 16, behind RTD_EIG_MFMA -- the north star's wording, measured slower, kept as a tested variant).
//
// Algorithm (own design, not the reference's LAPACK calls): with T = diag(sqrt(mu w)) the matrices
// -(T(alpha+beta)T^-1) = Pm and -(T(alpha-beta)T^-1) = Qm are symmetric positive definite, so with
// the Cholesky factor Pm = L L^T the non-symmetric problem (alpha-beta)(alpha+beta) v = k^2 v becomes
// the symmetric H z = k^2 z, H = L^T Qm L = F F^T with F = L^T R (Qm = R R^T), solved by a one-sided (Hestenes)
// parallel-order (XOR round-robin) cyclic Jacobi iteration on the columns of F.  One problem occupies NP lanes of a
// wavefront (64/NP problems per wave); lane j owns column j of every matrix in registers; columns are exchanged with
// DPP / ds_swizzle cross-lane moves, small vectors and the Cholesky factor go through LDS.
#include <cstdlib>
#include <type_traits>

#include "rtd_device.h"

namespace {

// compiler-only barrier: keeps the scheduler from hoisting a whole unrolled loop's LDS loads
#define RTD_FENCE() asm volatile("" ::: "memory")
#ifndef RTD_XOR_DPP
#define RTD_XOR_DPP 0x818E  /* bit set of the xor masks done by ONE DPP move per dword (1, 2, 3, 7, 8, 15) instead of ds_swizzle:
                               A/B on one box 6.03 -> 5.94 ms (with bound_ctrl moves; with the old copy-then-move form it lost 2 %) */
#endif

// DPP control of a lane permutation "lane ^ MASK" inside a 16-lane row that one DPP move can express, else -1:
// quad permutes for 1, 2, 3; row_half_mirror = ^7; row_ror:8 = ^8; row_mirror = ^15
constexpr __host__ __device__ int dpp_xor_ctrl(int mask) {
  return mask == 1 ? 0xB1 : mask == 2 ? 0x4E : mask == 3 ? 0x1B : mask == 7 ? 0x141 : mask == 8 ? 0x128 : mask == 15 ? 0x140 : -1;
}

template <int MASK>
__device__ __forceinline__ double xor_lane(double v) {
  // value of lane (lane ^ MASK)
  // (lane ^ 16 and lane ^ 32 by v_permlane16_swap / v_permlane32_swap -- VALU, no LDS crossbar -- measured in round 4: the 64-stream
  //  eigen kernel 9.3 -> 9.5 ms per 128 cfg5 columns, the 128-stream one unchanged; not kept)
  if constexpr (MASK >= 32) return __shfl_xor(v, MASK, 64);  // across the halves of the wavefront: ds_bpermute (NP = 64 only)
  int lo = __double2loint(v), hi = __double2hiint(v);
  constexpr int ctrl = dpp_xor_ctrl(MASK);
  if constexpr (ctrl >= 0 && ((RTD_XOR_DPP >> MASK) & 1)) {  // one VALU move per dword, no LDS crossbar (RTD_XOR_DPP = bit set of masks)
    lo = __builtin_amdgcn_update_dpp(0, lo, ctrl, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, ctrl, 0xF, 0xF, true);
  } else {  // ds_swizzle bit-mode: and = 0x1f, or = 0, xor = MASK
    constexpr int pat = (MASK << 10) | 0x1F;
    lo = __builtin_amdgcn_ds_swizzle(lo, pat);
    hi = __builtin_amdgcn_ds_swizzle(hi, pat);
  }
  return __hiloint2double(hi, lo);
}

constexpr __host__ __device__ int high_bit(int t) {
  int b = 1;
  while ((b << 1) <= t) b <<= 1;
  return b;
}

// 1/sqrt(x) and 1/x to full double precision from the hardware seeds (x > 0, normal range)
__device__ __forceinline__ double fast_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  const double hx = 0.5 * x;
  y = y * (1.5 - hx * y * y);
  y = y * (1.5 - hx * y * y);
  return y;
}
__device__ __forceinline__ double fast_rcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = y * (2.0 - x * y);
  y = y * (2.0 - x * y);
  return y;
}
// one Newton step from the ~5e-8 hardware seeds: ~4e-15 relative (tools/hiptests/seed_accuracy.hip), enough for
// quantities that only steer a rotation angle
__device__ __forceinline__ double approx_rsqrt(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  return y * (1.5 - 0.5 * x * y * y);
}
__device__ __forceinline__ double approx_rcp(double x) {
  const double y = __builtin_amdgcn_rcp(x);
  return y * (2.0 - x * y);
}

#ifndef RTD_JAC_F32_ANGLE
#define RTD_JAC_F32_ANGLE 1  /* rotation angle of the pair-layout sweeps from float arithmetic (see PairStep) */
#endif
#ifndef RTD_CHOL_FMAC_DPP
#define RTD_CHOL_FMAC_DPP 1  /* Cholesky trailing updates as ONE v_fmac_f64_dpp (row_newbcast) per element, NP = 16 */
#endif

#ifndef RTD_EIGEN32_WAVES
#define RTD_EIGEN32_WAVES 2  /* waves per SIMD of the 64-stream eigen kernel (256 VGPRs; the spill counts of every kernel: profiles/rNN_kernel_resources.json, tools/kernel_resources.py; 1: 278 VGPRs) */
#endif
#ifndef RTD_EIGEN_WAVES
#define RTD_EIGEN_WAVES 3  /* waves per SIMD the fused eigen kernel is compiled for at NP <= 16 (149 VGPRs, no spills, LDS 10.5 KB per
                              wavefront).  4 (128 VGPRs, 21-41 dwords spilled in the once-per-wavefront stages, packed L): 2 % slower (A/B) */
#endif

// A sweep is the last one when every pair it met had cos^2(angle) <= RTD_JAC_TOL before its rotation.  Quadratic
// convergence leaves about that much (not its square) of non-orthogonality behind: eigenvalues are then good to
// rounding, eigenvectors to ~RTD_JAC_TOL.  The eigenvector error matters where the beam source is nearly resonant with
// an eigenvalue (1/mu0 ~ k: the particular solution's 1/(1/mu0^2 - k^2) and the homogeneous part cancel): on the 64
// golden columns of cfg4 the worst column (mu0 = 0.912) is 1.8e-9 of the field scale off the reference at 1e-11 and
// 1e-13, 7.3e-11 at 1e-14 and 1e-16 (all other columns 2-3e-11 throughout); kernel time 4.86 / 4.94 / 5.06 ms at
// 1e-11 / 1e-14 / 1e-16.
#ifndef RTD_JAC_TOL
#define RTD_JAC_TOL 1e-14
#endif

// One level of a transposed reduction across the lane bit O: the lanes with (lane & O) == 0 keep lo and take the partner's lo, the
// others keep hi and take the partner's hi:  result = (lane & O ? hi : lo) + value of lane ^ O of (lane & O ? hi : lo)... which is
// exactly what ONE row / half swap of the two registers gives, with no select and no LDS crossbar: v_permlane16_swap exchanges the
// odd 16-lane rows of its first operand with the even rows of its second, v_permlane32_swap the upper half of the first with the
// lower half of the second -- afterwards first + second is the level's result in every lane.  (O = 16 and 32 were ds_swizzle /
// ds_bpermute exchanges behind two selects; their latency-hiding schedule held both operand sets of every element at once: the beam
// stage of the 64-stream kernel was where its 68-110 spilled registers came from.)
template <int O, bool SWAP>
__device__ __forceinline__ double level_add(const double lo, const double hi, const bool is_hi) {
  if constexpr (SWAP && (O == 16 || O == 32)) {
    typedef unsigned int u2 __attribute__((ext_vector_type(2)));
    u2 a, b;
    if constexpr (O == 16) {
      a = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(lo), (unsigned)__double2loint(hi), false, false);
      b = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(lo), (unsigned)__double2hiint(hi), false, false);
    } else {
      a = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(lo), (unsigned)__double2loint(hi), false, false);
      b = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(lo), (unsigned)__double2hiint(hi), false, false);
    }
    return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
  } else {
    const double keep = is_hi ? hi : lo, send = is_hi ? lo : hi;
    return keep + xor_lane<O>(send);
  }
}

// Transposed reduction: every lane enters with NP terms v[0..NP) (term i belongs to row i) and leaves with the sum of
// row `j` over the NP lanes of its group -- NP-1 exchange-adds in registers, no LDS memory.
template <int NP, int O>
struct TransposeStep {
  static __device__ __forceinline__ void run(double (&v)[NP], const int j) {
    const bool hi = (j & O) != 0;
#pragma unroll
    for (int i = 0; i < O; ++i) v[i] = level_add<O, NP == 32>(v[i], v[i + O], hi);  // (the 128-stream kernel keeps its ds_bpermute form:
    //                                                                                 with the swaps it spilled 123 registers and ran 3.7 x slower)
    TransposeStep<NP, O / 2>::run(v, j);
  }
};
template <int NP>
struct TransposeStep<NP, 0> {
  static __device__ __forceinline__ void run(double (&)[NP], const int) {}
};
template <int NP>
__device__ __forceinline__ double transpose_reduce(double (&v)[NP], const int j) {
  TransposeStep<NP, NP / 2>::run(v, j);
  return v[0];
}

// sum over the NP lanes of a group (result in every lane)
template <int NP>
__device__ __forceinline__ double group_sum(double v) {
  if (NP > 1) v += xor_lane<1>(v);
  if (NP > 2) v += xor_lane<2>(v);
  if (NP > 4) v += xor_lane<4>(v);
  if (NP > 8) v += xor_lane<8>(v);
  if (NP > 16) v += xor_lane<16>(v);
  if (NP > 32) v += xor_lane<32>(v);
  return v;
}

// compile-time loop: f(std::integral_constant<int, I>) for I in [B, E)
template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

// value of lane K of this lane's NP-group, K a compile-time constant: a DPP row broadcast (VALU, no LDS crossbar)
// when the group is one 16-lane DPP row, a ds_bpermute otherwise
template <int NP, int K>
__device__ __forceinline__ double bcast_lane(double v) {
  if constexpr (NP == 16) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + K, 0xF, 0xF, true);  // row_newbcast:K; bound_ctrl + full masks: no `old` operand, no copy
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + K, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
  } else if constexpr (NP == 64) {
    // the group is the wavefront: lane K by v_readlane (an SGPR pair, used as such by the FMA that follows) -- no LDS crossbar.
    // (ds_bpermute here made a 128-stream Cholesky factorisation 170 k cycles, 2 650 per pivot step: s_memtime stamps, round 4)
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), K), __builtin_amdgcn_readlane(__double2loint(v), K));
  } else {
    return __shfl(v, K, NP);
  }
}

// In-register Cholesky of a symmetric positive definite matrix held one column per lane (col[i] = A[i][j]);
// on exit col[i] = L[i][j] for i >= j and 0 above the diagonal; returns 1 / L[j][j].  The trailing matrix is kept
// symmetric so that A^(K)[j][K] is available in the lane's own registers, and the scaling of a finished column by
// 1/sqrt(pivot) is deferred to the end: a step is one broadcast and one FMA per element,
//   col_j[i] -= A^(K)[i][K] * A^(K)[j][K] / A^(K)[K][K]   for j > K (factor 0 for the finished columns j <= K).
template <int NP, int K, int I, bool ORDERED = false>
struct CholRowDpp {  // col[i] -= bcast_K(col[i]) * f for i = I .. NP - 1, one v_fmac_f64_dpp each
  static __device__ __forceinline__ void run(double (&col)[NP], const double f) {
    // The compiler's hazard recogniser does not see a VALU write inside inline asm, and the next step broadcasts
    // col[K + 1] (its pivot) with a DPP move of its own: the two wait states a DPP read needs after a VALU write of the
    // same VGPR are spent here, behind the rows that DPP reads next (the first and the last of the step).
    // ORDERED (NP = 32): the statements keep their program order (asm volatile) -- left free, the scheduler put the updates
    // of one element by consecutive steps next to each other (the ISA scan of build.py caught it).
    if constexpr (ORDERED) {
      if constexpr (I == K + 1 || I == NP - 1)
        asm volatile("v_fmac_f64_dpp %0, -%0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf\n\ts_nop 1" : "+v"(col[I]) : "v"(f), "n"(K % 16));
      else
        asm volatile("v_fmac_f64_dpp %0, -%0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(col[I]) : "v"(f), "n"(K % 16));
    } else {
      if constexpr (I == K + 1 || I == NP - 1)
        asm("v_fmac_f64_dpp %0, -%0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf\n\ts_nop 1" : "+v"(col[I]) : "v"(f), "n"(K % 16));
      else
        asm("v_fmac_f64_dpp %0, -%0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(col[I]) : "v"(f), "n"(K % 16));
    }
    CholRowDpp<NP, K, I + 1, ORDERED>::run(col, f);
  }
};
template <int NP, int K, bool ORDERED>
struct CholRowDpp<NP, K, NP, ORDERED> {
  static __device__ __forceinline__ void run(double (&)[NP], const double) {}
};

template <int NP, int K>
struct CholStep {
  static __device__ __forceinline__ void run(double (&col)[NP], double& diag, const int j) {
    const double akk = bcast_lane<NP, K>(col[K]);
    const double f = (j > K) ? col[K] * fast_rcp(akk) : 0.0;
    diag = (j == K) ? akk : diag;
    if constexpr (RTD_CHOL_FMAC_DPP && NP == 16) {
      // col[i] -= bcast_K(col[i]) * f in one instruction: the DP-ALU DPP form exists for row_newbcast only (the DPP
      // source is the accumulator itself).  A VALU write of a VGPR needs two wait states before a DPP read of it: the
      // elements were last written by step K - 1's updates, at least NP - K instructions back; `f` is not a DPP operand.
      CholRowDpp<NP, K, K + 1>::run(col, f);
    } else {
#pragma unroll
      for (int i = K + 1; i < NP; ++i) col[i] = fma(-bcast_lane<NP, K>(col[i]), f, col[i]);
    }
    CholStep<NP, K + 1>::run(col, diag, j);
  }
};
template <int NP>
struct CholStep<NP, NP> {
  static __device__ __forceinline__ void run(double (&)[NP], double&, const int) {}
};
// acc[I] += bcast_I(y0) * coef and acc[16 + I] += bcast_I(y1) * coef for I = 0..15: lane I of the caller's DPP row supplies the
// multiplier (row_newbcast), one v_fmac_f64_dpp per term
template <int I>
struct RowFmacDpp {
  static __device__ __forceinline__ void run(double (&acc)[32], const double y0, const double y1, const double coef) {
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc[I]) : "v"(y0), "v"(coef), "n"(I));
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc[16 + I]) : "v"(y1), "v"(coef), "n"(I));
    if constexpr (I + 1 < 16) RowFmacDpp<I + 1>::run(acc, y0, y1, coef);
  }
};

// the same for 64 rows: acc[16 g + I] += bcast_I(y[g]) * coef, g < 4 (128 streams: lane jj of EVERY DPP row holds the table
// elements jj, 16 + jj, 32 + jj, 48 + jj of the moment, so lane I of the caller's own row has the multiplier of row 16 g + I)
template <int I>
struct RowFmacDpp64 {
  static __device__ __forceinline__ void run(double (&acc)[64], const double (&y)[4], const double coef) {
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc[I]) : "v"(y[0]), "v"(coef), "n"(I));
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc[16 + I]) : "v"(y[1]), "v"(coef), "n"(I));
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc[32 + I]) : "v"(y[2]), "v"(coef), "n"(I));
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc[48 + I]) : "v"(y[3]), "v"(coef), "n"(I));
    if constexpr (I + 1 < 16) RowFmacDpp64<I + 1>::run(acc, y, coef);
  }
};

// acc += (lane K of the caller's own 16-lane DPP row of `src`) * mul: one v_fmac_f64_dpp row_newbcast.  `src` comes from an LDS
// load (no VALU write in front of the DPP read: no hazard for tools/check_dpp_hazards.py to find).
template <int K>
__device__ __forceinline__ void fmac_bcast(double& acc, const double src, const double mul) {
  asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(K));
}

// ---- NP = 32, round 5: the triangular products of the eigen stage with the rows of L SPREAD OVER THE LANES ----------------
// F = L^T R, Y = L^-T Z and A = L Z were 528 FMAs per lane each, every one with its multiplier L[r][i] from a broadcast LDS read
// (all 32 lanes of a problem read one address): 2 LDS cycles per read against 1 DP-ALU cycle per FMA with every CU's eight
// wavefronts doing the same -- LDS-bandwidth-bound, 10-12 k cycles per product and wavefront at two wavefronts per SIMD (s_memtime
// stamps, profiles/r05_eigen32_phases.txt).  Now lane t of every DPP row loads L[r][t] and L[r][16 + t] of row r ONCE (48 conflict-
// free reads per product: the packed row is contiguous) and every FMA takes its multiplier as a DPP row broadcast of those two
// registers (NP = 64: four registers per row, the padded square of L in LDS; same three products).  Lanes t > r hold whatever follows row r in the packed triangle: they are never broadcast (i <= r at compile time).
//   F (column j of R in the lane):  w[i]   += L[r][i] qcol[r]            for i <= r      (accumulators independent)
//   Y (back substitution):          ya[r]  *= 1 / L[r][r];  ya[i] -= L[r][i] ya[r]      for i < r, r = 31 ... 0
//   A (row r of L):                 aa[r]   = sum_{i <= r} L[r][i] zc[i]                 (two accumulation chains per row)
template <int NP>
struct LRowN {  // row r of L: lane t of every DPP row holds L[r][16 g + t] in g[g] (NP = 64: four registers per row)
  double g[NP / 16];
  template <int R, bool PACKED>
  static __device__ __forceinline__ LRowN load(const double* L_, const int t) {
    constexpr int off = PACKED ? R * (R + 1) / 2 : R * (NP + 1);  // (the packed triangle at NP = 32, the padded square at NP = 64)
    LRowN x;
#pragma unroll
    for (int gi = 0; gi < NP / 16; ++gi) x.g[gi] = 16 * gi <= R ? L_[off + 16 * gi + t] : 0.0;
    return x;
  }
};
template <int NP, int I, int IEND>
struct AxpyRowN {  // acc[i] += L[r][i] * mul for i = I .. IEND - 1
  static __device__ __forceinline__ void run(double (&acc)[NP], const LRowN<NP>& row, const double mul) {
    if constexpr (I < IEND) {
      fmac_bcast<I % 16>(acc[I], row.g[I / 16], mul);
      AxpyRowN<NP, I + 1, IEND>::run(acc, row, mul);
    }
  }
};
template <int NP, int I, int IEND>
struct DotRowN {  // a[g] += sum_{16 g <= i < 16 g + 16} L[r][i] v[i]: one accumulation chain per 16 columns
  static __device__ __forceinline__ void run(double (&a)[NP / 16], const LRowN<NP>& row, const double (&v)[NP]) {
    if constexpr (I < IEND) {
      fmac_bcast<I % 16>(a[I / 16], row.g[I / 16], v[I]);
      DotRowN<NP, I + 1, IEND>::run(a, row, v);
    }
  }
};

// transposed reduction of x[i] = a[i] * s (and of a[i] * s + b[i] * t) with the products formed inside its first level: 16
// temporaries instead of 32 -- stage 2 of the 64-stream kernel held zc, ya, aa and 32-element temporaries (256 + registers: 68
// spilled, their reloads interleaved with the global stores of Y and A, each reload a wait for a store's acknowledgement)
template <int NP>
__device__ __forceinline__ double transpose_reduce_scaled(const double (&a)[NP], const double s, const int j) {
  constexpr int O = NP / 2;
  const bool hi = (j & O) != 0;
  double v[O];
#pragma unroll
  for (int i = 0; i < O; ++i) v[i] = level_add<O, true>(a[i] * s, a[i + O] * s, hi);
  return transpose_reduce<O>(v, j);
}
template <int NP>
__device__ __forceinline__ double transpose_reduce_scaled2(const double (&a)[NP], const double s, const double (&b)[NP], const double t, const int j) {
  constexpr int O = NP / 2;
  const bool hi = (j & O) != 0;
  double v[O];
#pragma unroll
  for (int i = 0; i < O; ++i) v[i] = level_add<O, true>(fma(b[i], t, a[i] * s), fma(b[i + O], t, a[i + O] * s), hi);
  return transpose_reduce<O>(v, j);
}

// ---- NP = 32: blocked (2 x 2 blocks of 16) in the same one-column-per-lane layout.  A problem is two DPP rows of 16 lanes:
// row 0 holds the columns 0..15, row 1 the columns 16..31, every lane all 32 rows of its column.  The pivot column of a
// step lives in ONE of the two rows, and v_fmac_f64_dpp row_newbcast reaches exactly the lanes of that row:
//   panel A (steps 0..15):  the row-0 lanes eliminate inside their 16 columns, all 32 rows (A11 and A21);
//   block update:           A22 -= sum_K X[.][K] X[.][K]^T / a_KK with X = the finished A21 (16 x 16) -- the only place where
//                           the two rows of lanes exchange data: X goes through LDS once into the operand layout of
//                           v_mfma_f64_16x16x4_f64 (4 MFMA per problem) and the product comes back once;
//   panel B (steps 16..31): the row-1 lanes eliminate inside A22.
// The unblocked form took every multiplier of every step across the two rows with ds_bpermute: 1 056 LDS-crossbar
// instructions per factorisation, 32 dependent LDS round trips (26-50 k cycles per factorisation at two wavefronts per
// SIMD, s_memtime stamps); this form has four.
template <int K>
__device__ __forceinline__ double row_bcast16(double v) {  // lane K of the caller's own 16-lane DPP row
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + K, 0xF, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + K, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
template <int K, int KEND>
struct CholPanel32 {  // steps K .. KEND - 1 with their pivot columns in DPP row PR = K / 16 of every problem
  static __device__ __forceinline__ void run(double (&col)[32], double& diag, double& dmine, const int row, const int jj) {
    constexpr int PR = K / 16, KL = K % 16;
    const double akk = row_bcast16<KL>(col[K]);
    const double r = fast_rcp(akk);
    const bool mine = row == PR;
    const double f = (mine && jj > KL) ? col[K] * r : 0.0;
    diag = (mine && jj == KL) ? akk : diag;
    dmine = (mine && jj == KL) ? r : dmine;
    CholRowDpp<32, K, K + 1, true>::run(col, f);
    if constexpr (K + 1 < KEND) CholPanel32<K + 1, KEND>::run(col, diag, dmine, row, jj);
  }
};
// scratch: 1 024 doubles of LDS (the staging area of the assembly, free while the factorisation runs)
__device__ __forceinline__ double cholesky_columns32(double (&col)[32], const int j, double* scratch) {
  typedef double v4d __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x, g = lane >> 5, row = (lane >> 4) & 1, jj = lane & 15;
  double diag = 1.0, dmine = 0.0;
  CholPanel32<0, 16>::run(col, diag, dmine, row, jj);
  // X[i][K] = col[16 + i] of lane (row 0, K); scaled copy X d_K next to it
  double* sX = scratch;         // [2 problems][16 K][16 i]
  double* sXd = scratch + 512;  // the same times 1 / a_KK
  __syncthreads();
  if (row == 0) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      sX[(g * 16 + jj) * 16 + i] = col[16 + i];
      sXd[(g * 16 + jj) * 16 + i] = col[16 + i] * dmine;
    }
  }
  __syncthreads();
  v4d acc[2];
#pragma unroll
  for (int pg = 0; pg < 2; ++pg) {
    acc[pg] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int t = 0; t < 4; ++t) {  // lane (k, i): A[i][k] = X[i][4 t + k] d, B[k][j] = X[j][4 t + k]: the same element
      const int e = (pg * 16 + 4 * t + (lane >> 4)) * 16 + (lane & 15);
      acc[pg] = __builtin_amdgcn_mfma_f64_16x16x4f64(sXd[e], sX[e], acc[pg], 0, 0, 0);
    }
  }
  __syncthreads();
  // the products back: lane (k, c) holds S[4 q + k][c]; the row-1 lane of column c wants all 16 rows
  double* sS = scratch;  // [2 problems][16 rows][16 columns]
#pragma unroll
  for (int pg = 0; pg < 2; ++pg)
#pragma unroll
    for (int q = 0; q < 4; ++q) sS[(pg * 16 + 4 * q + (lane >> 4)) * 16 + (lane & 15)] = acc[pg][q];
  __syncthreads();
  if (row == 1) {
#pragma unroll
    for (int i = 0; i < 16; ++i) col[16 + i] -= sS[(g * 16 + i) * 16 + jj];
  }
  CholPanel32<16, 32>::run(col, diag, dmine, row, jj);
  const double rinv = fast_rsqrt(diag);
#pragma unroll
  for (int i = 0; i < 32; ++i) col[i] = (i >= j) ? col[i] * rinv : 0.0;
  __syncthreads();  // the scratch area goes back to its owner
  return rinv;
}

template <int NP>
__device__ __forceinline__ double cholesky_columns(double (&col)[NP], const int j) {
  double diag = 1.0;
  CholStep<NP, 0>::run(col, diag, j);
  const double rinv = fast_rsqrt(diag);
#pragma unroll
  for (int i = 0; i < NP; ++i) col[i] = (i >= j) ? col[i] * rinv : 0.0;
  return rinv;
}

// problem index of this lane's group; invalid groups redo the last layer and skip their stores
struct ProbId {
  long pid;
  int c, m, l;
  int mg;  // the Fourier mode the local index m stands for (mode shards: m0 + mstep * m)
  bool valid;
};
template <int NP>
__device__ __forceinline__ ProbId locate(const RtdDev& d, const int tx = threadIdx.x) {
  // One wavefront = the 64/NP consecutive layers of ONE (column, mode): c and m depend on blockIdx only, so
  // they are wave-uniform and the Legendre-table reads (indexed by m and l only) become scalar loads.
  constexpr int GPW = 64 / NP;
  // layer shard (rtd_plan_solve_layers): only the layers [l0, l0 + ln) are decomposed; ln = L without shards.
  // chunk selection (lean retained plans): the wavefront takes entry blockIdx % nsel of its column's chunk list
  const int nchunk = d.nsel > 0 ? d.nsel : (d.ln + GPW - 1) / GPW;
  const long cmi_b = (long)blockIdx.x / nchunk;
  int chunk = (int)((long)blockIdx.x % nchunk);
  ProbId p;
  // Workgroups are handed out mode by mode (all columns of mode 0, then of mode 1, ...): the sweep count falls with the Fourier mode
  // (5.8 sweeps at m = 0, 1.8 at m = 31 on cfg4), so the longest problems start first and the launch ends on its shortest ones
  // instead of on the last column's mode 0; the mode's table rows are shared by everything that runs at the same time.
#ifdef RTD_EIG_COLUMN_MAJOR  /* A/B builds: the order of rounds 1-5 */
  p.m = (int)(cmi_b % d.M);
  p.c = (int)(cmi_b / d.M);
#else
  p.m = (int)(cmi_b / d.C);
  p.c = (int)(cmi_b % d.C);
#endif
  p.mg = d.m0 + d.mstep * p.m;
  const long cmi = (long)p.c * d.M + p.m;
  if (d.nsel > 0) chunk = d.chunk_sel[(long)p.c * d.nsel + chunk];  // (wave-uniform) -1: this entry of the list is empty
  const int slot = chunk < 0 ? d.ln : chunk * GPW + tx / NP;
  p.valid = slot < d.ln;
  const int sl = p.valid ? slot : d.ln - 1;  // invalid groups redo the last slot and skip the stores
  p.l = d.ln == d.L ? d.lperm[(long)p.c * d.L + sl] : d.l0 + sl;
  p.pid = cmi * d.L + p.l;
  return p;
}

// ------------------------------------------------------------------------------------------------
// Jacobi sweeps in the "pair" layout (the default): the NP lanes of a problem are NP/2 pair slots x 2 halves; lane
// (p, u) holds the elements [u NP/2, (u + 1) NP/2) of BOTH columns of the pair in slot p.  Against the column-per-lane
// form of round 1 (one column per lane, removed in round 3): the dot product of a pair is NP/2 FMAs and one cross-lane add (not NP FMAs after NP column moves),
// and after the rotation only ONE of the two columns moves on, as a half column: NP/2 doubles per lane (not NP) to
// lane ^ mask with mask in {1, 2, 3, 7, 8, 15} -- a single DPP move per dword.  Per step a wavefront issues
// ~NP/2 + 25 + 2 NP FP64 instructions and NP + 6 moves instead of NP + 34 + 2 NP and 2 NP + 2.
//
// Ordering: a butterfly on the NP/2 pair slots.  Level g (g = NP/2, NP/4, ..., 1) pairs the kept column X of a slot
// with the g columns Y that circulate inside its group of g slots (g steps, Gray-code walk with the masks above); at
// the end of a level the slots of the upper half of each group hand on their X instead of their Y, which splits the
// group into two independent halves.  Which of the two columns a slot hands on is folded into the rotation itself
// (the lanes write the rotated pair into (X, Y) or into (Y, X): four coefficient selects, no column selects).  Every one
// of the NP (NP - 1) / 2 pairs meets exactly once per sweep whatever the arrangement the sweep starts from, so sweeps
// simply follow each other (tools/jacobi_schedule.py replays the schedule and checks this).  The columns wander; each
// carries the index it started with, and the lanes put them back in that order after the last sweep (the
// boundary-condition kernel's speculative diagonal pivoting relies on eigen-columns that stay next to their diagonal).
// ------------------------------------------------------------------------------------------------
template <int NP>
struct JSched {  // after the rotation of step s: slots with (p & sw[s]) hand on X instead of Y; the move is slot ^ mk[s]
  int sw[NP - 1], mk[NP - 1];
  constexpr JSched() : sw{}, mk{} {
    constexpr int LPP = NP / 2;
    int s = 0;
    for (int g = LPP; g >= 1; g >>= 1) {
      for (int b = 0; b < g; ++b, ++s) {
        if (b < g - 1) {  // walk inside the level: Gray code with 4 replaced by 7 (row_half_mirror)
          const int low = (b + 1) & -(b + 1);
          sw[s] = 0;
          mk[s] = low == 4 ? 7 : low;
        } else if (g > 1) {  // split the groups of g slots into halves
          const int h = g >> 1;
          sw[s] = h;
          mk[s] = h == 4 ? 7 : h;
        } else {  // into the next sweep
          sw[s] = 0;
          mk[s] = LPP > 1 ? LPP - 1 : 0;
        }
      }
    }
  }
};

template <int MASK>
__device__ __forceinline__ int xor_lane_i(int v) {
  constexpr int ctrl = dpp_xor_ctrl(MASK);
  if constexpr (ctrl >= 0) return __builtin_amdgcn_update_dpp(0, v, ctrl, 0xF, 0xF, true);
  else return __builtin_amdgcn_ds_swizzle(v, (MASK << 10) | 0x1F);
}

template <int NP, int S>
struct PairStep {
  static constexpr int H = NP / 2;
  static __device__ __forceinline__ void run(double (&xh)[H], double (&yh)[H], double& ax, double& ay, int& ix, int& iy,
                                             const int p, int& notconv) {
    constexpr JSched<NP> sched{};
    constexpr int SW = sched.sw[S], MK = sched.mk[S];
    double g0 = 0.0, g1 = 0.0;
#pragma unroll
    for (int i = 0; i < H; i += 2) {
      g0 = fma(xh[i], yh[i], g0);
      if (i + 1 < H) g1 = fma(xh[i + 1], yh[i + 1], g1);
    }
    const double gp = g0 + g1;
    const double gamma = gp + xor_lane<H>(gp);  // both halves of the pair: the same bits in both lanes
    // tan(2 theta) = 2 gamma / (|y|^2 - |x|^2);  t = tan(theta) without cancellation.  Branch-free: the tiny term keeps
    // gamma = 0 (decoupled or already orthogonal columns, also with equal norms) at t = 0, c = 1 without a 0/0.
    const double delta = ay - ax;
    const double g2 = 2.0 * gamma;
#if RTD_JAC_F32_ANGLE
    // The angle only steers the iteration: a t that is 1e-7 off leaves 1e-7 of the pair's inner product behind, far below
    // what the sweep test asks for (the test is made BEFORE the rotation).  So t comes from float arithmetic with the
    // hardware's rsq / rcp and no Newton step (eigenvalues here lie between 1e-3 and 1e5: no float range issue; a
    // product that underflows belongs to a pair that has converged).  c and s below are double, from that t: the
    // rotation itself stays orthogonal to double accuracy.
    const float df = (float)delta, gf = (float)g2;
    const float r2f = fmaf(df, df, fmaf(gf, gf, 1e-36f));
    const float rhof = r2f * __builtin_amdgcn_rsqf(r2f);
    const float denf = df + copysignf(rhof, df);
    const double tt = (double)(gf * __builtin_amdgcn_rcpf(denf));
#else
    const double r2 = fma(delta, delta, fma(g2, g2, 1e-280));
    const double rho = r2 * approx_rsqrt(r2);
    const double den = delta + copysign(rho, delta);
    const double tt = g2 * approx_rcp(den);
#endif
    // c from ONE Newton step (4e-15): the error scales BOTH columns of the pair by the same 1 + eps, so orthogonality and
    // the directions z are untouched; only the norms k drift, by ~50 rotations x 4e-15 (parity unchanged)
    const double c = approx_rsqrt(fma(tt, tt, 1.0));
    const double sn = tt * c;
    notconv |= (gamma * gamma > RTD_JAC_TOL * ax * ay) ? 1 : 0;
    const double nax = fma(-tt, gamma, ax), nay = fma(tt, gamma, ay);  // |c x - s y|^2, |s x + c y|^2
    // (x, y) <- (c x - s y, s x + c y), written the other way round in the slots that hand on their x
    double cxx = c, cxy = -sn, cyx = sn, cyy = c;
    double oax = nax, oay = nay;
    if constexpr (SW != 0) {
      const bool swp = (p & SW) != 0;
      cxx = swp ? sn : c;
      cxy = swp ? c : -sn;
      cyx = swp ? c : sn;
      cyy = swp ? -sn : c;
      oax = swp ? nay : nax;
      oay = swp ? nax : nay;
      const int t = swp ? iy : ix;
      iy = swp ? ix : iy;
      ix = t;
    }
#pragma unroll
    for (int i = 0; i < H; ++i) {
      const double xi = xh[i], yi = yh[i];
      xh[i] = fma(cxy, yi, cxx * xi);
      yh[i] = xor_lane<MK>(fma(cyy, yi, cyx * xi));
    }
    ax = oax;
    ay = xor_lane<MK>(oay);
    iy = xor_lane_i<MK>(iy);
    PairStep<NP, S + 1>::run(xh, yh, ax, ay, ix, iy, p, notconv);
  }
};
template <int NP>
struct PairStep<NP, NP - 1> {  // the last step of a sweep is step NP - 2
  static __device__ __forceinline__ void run(double (&)[NP / 2], double (&)[NP / 2], double&, double&, int&, int&, const int, int&) {}
};

#ifndef RTD_JAC_FAST
#define RTD_JAC_FAST 1  /* NP >= 32: scaled ("fast") rotations in the pair-layout sweeps, see FastPairStep */
#endif
// The same step with SCALED rotations, for NP >= 32.  The lane keeps its halves of the two columns as x = sx xs, y = sy ys (xs, ys
// stored, sx, sy per-column scalars) and applies (x, y) <- c (x - t y, y + t x) as
//     xs <- xs - (t sy / sx) ys,   ys <- ys + (t sx / sy) xs,   sx <- c sx,   sy <- c sy :
// TWO FMAs per element pair instead of mul + fma twice -- NP instructions less per step for ~14 of scale bookkeeping (1 / sx,
// 1 / sy are carried along, multiplied by 1 / c = (1 + t^2) c; the scales of the column that moves on move with it).  At NP = 16
// that is no gain (16 against 11 + moves, measured in round 3); at NP = 32 it is 18 of ~150 instructions per step, at NP = 64
// 50 of ~267.  The scales are folded back into the columns after every sweep (2 multiplications per element), so they stay
// within 2^-(NP/2) and sx (1 / sx) drifts by no more than NP - 1 roundings.  The steps that hand on X instead of Y (one per level
// of the butterfly) keep the general two-coefficient form.
template <int NP, int S>
struct FastPairStep {
  static constexpr int H = NP / 2;
  static __device__ __forceinline__ void run(double (&xh)[H], double (&yh)[H], double& ax, double& ay, int& ix, int& iy,
                                             const int p, int& notconv, double& sx, double& sy, double& rx, double& ry) {
    constexpr JSched<NP> sched{};
    constexpr int SW = sched.sw[S], MK = sched.mk[S];
    double g0 = 0.0, g1 = 0.0;
#pragma unroll
    for (int i = 0; i < H; i += 2) {
      g0 = fma(xh[i], yh[i], g0);
      g1 = fma(xh[i + 1], yh[i + 1], g1);
    }
    const double gp = g0 + g1;
    const double gamma = (gp + xor_lane<H>(gp)) * (sx * sy);  // the inner product of the two columns themselves
    const double delta = ay - ax;
    const double g2 = 2.0 * gamma;
    // (the angle from float arithmetic: see PairStep)
    const float df = (float)delta, gf = (float)g2;
    const float r2f = fmaf(df, df, fmaf(gf, gf, 1e-36f));
    const float rhof = r2f * __builtin_amdgcn_rsqf(r2f);
    const float denf = df + copysignf(rhof, df);
    const double tt = (double)(gf * __builtin_amdgcn_rcpf(denf));
    const double t2 = fma(tt, tt, 1.0);
    // c to full precision here (two Newton steps): sx and 1 / sx are carried separately, and an error e of c makes their product
    // drift by 2 e per rotation -- with the one-step c of PairStep (4e-15) the ratio sy / sx was 2e-13 off by the end of a sweep
    // and the rotations that much off orthogonal (the fused / general evaluation paths then differed by 3e-11 instead of 1.4e-11)
    const double c = fast_rsqrt(t2);
    const double cinv = t2 * c;
    notconv |= (gamma * gamma > RTD_JAC_TOL * ax * ay) ? 1 : 0;
    const double nax = fma(-tt, gamma, ax), nay = fma(tt, gamma, ay);
    const double a = -tt * (sy * rx), b = tt * (sx * ry);  // xs <- xs + a ys, ys <- ys + b xs
    const double nsx = c * sx, nsy = c * sy, nrx = cinv * rx, nry = cinv * ry;
    if constexpr (SW == 0) {
#pragma unroll
      for (int i = 0; i < H; ++i) {
        const double xi = xh[i], yi = yh[i];
        xh[i] = fma(a, yi, xi);
        yh[i] = xor_lane<MK>(fma(b, xi, yi));
      }
      ax = nax;
      ay = xor_lane<MK>(nay);
      sx = nsx;
      rx = nrx;
      sy = xor_lane<MK>(nsy);
      ry = xor_lane<MK>(nry);
    } else {
      // the slots with (p & SW) write the rotated pair the other way round: (xs, ys) <- (ys + b xs, xs + a ys)
      const bool swp = (p & SW) != 0;
      const double pxx = swp ? b : 1.0, pxy = swp ? 1.0 : a, pyx = swp ? 1.0 : b, pyy = swp ? a : 1.0;
#pragma unroll
      for (int i = 0; i < H; ++i) {
        const double xi = xh[i], yi = yh[i];
        xh[i] = fma(pxy, yi, pxx * xi);
        yh[i] = xor_lane<MK>(fma(pyy, yi, pyx * xi));
      }
      ax = swp ? nay : nax;
      ay = xor_lane<MK>(swp ? nax : nay);
      sx = swp ? nsy : nsx;
      rx = swp ? nry : nrx;
      sy = xor_lane<MK>(swp ? nsx : nsy);
      ry = xor_lane<MK>(swp ? nrx : nry);
      const int t = swp ? iy : ix;
      iy = swp ? ix : iy;
      ix = t;
    }
    iy = xor_lane_i<MK>(iy);
    FastPairStep<NP, S + 1>::run(xh, yh, ax, ay, ix, iy, p, notconv, sx, sy, rx, ry);
  }
};
template <int NP>
struct FastPairStep<NP, NP - 1> {
  static __device__ __forceinline__ void run(double (&)[NP / 2], double (&)[NP / 2], double&, double&, int&, int&, const int, int&,
                                             double&, double&, double&, double&) {}
};

// ------------------------------------------------------------------------------------------------
// Fused eigen stage: assembly, Cholesky factors, one-sided Jacobi and the eigenvector /
// particular-solution stage in ONE kernel per (c, m, l).  F, L, Qm and k Z never leave the CU (registers + LDS):
// compared with the three-kernel pipeline this removes 10.6 KB of HBM traffic per problem (a third of the path's
// total) and the Lw / Qw workspaces.
// ------------------------------------------------------------------------------------------------
// packed lower triangle: element (r, c), r >= c
__host__ __device__ constexpr int tri(int r, int c) { return r * (r + 1) / 2 + c; }

// Diagnostic build (-DRTD_EIG_STAMPS, never shipped): lane 0 of a few wavefronts records s_memtime at the phase boundaries of
// the fused eigen kernel and prints the differences (cycles) with its sweep count; tools/eig_phase_cycles.py formats them.
#ifdef RTD_EIG_STAMPS
#define RTD_ESTAMP(k)                                      \
  {                                                        \
    __builtin_amdgcn_sched_barrier(0);                     \
    est[k] = (long long)__builtin_amdgcn_s_memtime();      \
    __builtin_amdgcn_sched_barrier(0);                     \
  }
#else
#define RTD_ESTAMP(k)
#endif

template <int NP, int JV>  // JV: 2 = default; 3 = the assembly of Pm, Qm on the matrix cores (RTD_EIG_MFMA, NP = 16)
__global__ __launch_bounds__(64, (NP <= 16 ? RTD_EIGEN_WAVES : NP == 32 ? RTD_EIGEN32_WAVES : 1)) void rtd_eigen_kernel(RtdDev d) {
  constexpr int GPW = 64 / NP;
  // Cholesky factor L of Pm in LDS: a padded square at NP <= 16; at NP = 32 the packed lower triangle (element (r, c), r >= c,
  // at r (r + 1) / 2 + c): half the LDS, which is what lets two wavefronts per SIMD fit there (A/B: 12.4 -> 11.6 ms per 128
  // cfg5 columns; at NP = 16 the selects of the two full-row / full-column products cost more than the LDS buys: 5.18 -> 5.27 ms)
  constexpr bool PACKED = NP == 32;
  constexpr int LD = NP + 1;
  constexpr int LSIZE = PACKED ? NP * (NP + 1) / 2 : NP * LD;
  __shared__ double sL[GPW][LSIZE];
  auto lix = [](const int r, const int c) { return PACKED ? tri(r, c) : r * LD + c; };  // element (r, c) of L, r >= c
  __shared__ double sV[GPW][4][NP];
  // NP = 32: the table rows one parity of the assembly needs ((P - m) / 2 rows of Ybar^m at the quadrature nodes, shared by
  // the wavefront's two problems), their coefficients and the beam factors, staged by coalesced loads that are all in
  // flight together.  Read as scalar loads inside the loop (the form that suits 3 wavefronts per SIMD at NP <= 16) every
  // row cost a full memory latency at 2 wavefronts per SIMD: the two assembly loops were 35 % of the kernel's time at
  // low Fourier modes (s_memtime stamps, -DRTD_EIG_STAMPS, profiles/archive/r03_eigen32_phases.txt).
  constexpr int TROWS = NP == 32 ? 32 : 1;
  __shared__ double sTab[NP == 32 ? TROWS * NP : 1];
  __shared__ double sCoef[NP == 32 ? GPW * TROWS : 1];
  __shared__ double sY0[NP == 32 ? TROWS : 1];
  double w[NP];  // column j of F = L^T R, then of k Z
  // beam source terms of this lane's stream, sum_l (omega w_l Y_l[j]) Ybar_l(-mu0) over the even / odd l - m: they fall out of
  // the assembly loop for one FMA per term (the second pass over the moments that stage 2 used to make is gone)
  double xe_sum = 0.0, xo_sum = 0.0;
#ifdef RTD_EIG_STAMPS
  long long est[12];
  for (int k = 0; k < 12; ++k) est[k] = 0;
  int est_sweeps = 0;
#endif
  RTD_ESTAMP(0);
  // The column's beam direction and strength, read HERE: the wavefront's column is uniform and nothing has been stored yet, so these
  // are scalar loads into SGPRs.  Read in stage 2 behind the stores of Y they were vector loads (the compiler cannot prove that the
  // kernel's own stores leave them alone) that waited for those stores' acknowledgements: loads and stores share one in-order counter.
  double mu0_c = 1.0, I0_c = 0.0;
  if (d.beam) {
    const int c0 = locate<NP>(d).c;
    mu0_c = d.mu0[c0];
    I0_c = d.I0[c0];
  }
  {  // ---- stage 1: assembly, Cholesky factors, F, Jacobi.  Only w, the two sums (and the LDS tile of L) leave this block.
  const int grp = threadIdx.x / NP, j = threadIdx.x % NP;
  const ProbId id = locate<NP>(d);
  const int P = d.P, m = id.m, c = id.c, l = id.l;
  double* L_ = sL[grp];
  double* dinv = sV[grp][3];  // 1 / L[i][i]
  const double* wl = d.wleg + ((long)c * d.L + l) * P;
  const double om = d.omega[(long)c * d.L + l];
  const double* Ym = d.Y + (long)m * P * NP;

  const double invmu_j = d.invmu[j], S_j = d.S[j];
  if constexpr (JV == 3 && NP == 16) {
    // Assembly on the matrix cores (the north star's "MFMA for the batched dense GEMMs in matrix assembly"); RTD_EIG_MFMA=1.
    // Measured (profiles/archive/r02_eigen_mfma_assembly.json, same box): 2.346 G VALU instructions per launch instead of 2.421 G,
    // SQ_VALU_MFMA_BUSY_CYCLES 0.42 G, kernel 5.48 ms instead of 5.17 ms: the four dependent MFMA per accumulator and the
    // lane-row transposes behind them sit on the critical path of a kernel that was not short of issue slots there.  Kept as
    // a selectable, tested variant; not the default.
    //   Pm = M^-1 - sum_{l - m even} (omega w_l) (S Y_l)(S Y_l)^T,  Qm likewise over the odd terms  (:123-135)
    // as rank-4 updates v_mfma_f64_16x16x4_f64 of an accumulator that starts as M^-1: the A operand of a lane (i, k) is
    // -(omega w_l S_i Y_l[i]) for the k-th l of the chunk, the B operand S_j Y_l[j] -- the same table element, so one load
    // feeds both and the four problems of the wavefront (four layers of one (column, mode): the table is shared) differ
    // by their coefficient only.  The accumulators come out in the D layout (lane (kq, col): rows 4 q + kq of column col of
    // ONE problem); the Cholesky wants column col of problem g in lane (g, col): a 4 x 4 transpose of the lane-rows per
    // register, two v_permlane16_swap + two v_permlane32_swap per dword.
    typedef double v4d __attribute__((ext_vector_type(4)));
    typedef unsigned int u2 __attribute__((ext_vector_type(2)));
    const int tx = threadIdx.x, i16 = tx & 15, k4 = tx >> 4;
    int chunk = (int)((long)blockIdx.x % (d.nsel > 0 ? d.nsel : (d.ln + GPW - 1) / GPW));  // as locate(): layer shards decompose [l0, l0 + ln)
    if (d.nsel > 0) chunk = d.chunk_sel[(long)c * d.nsel + chunk];
    if (chunk < 0) chunk = (d.ln + GPW - 1) / GPW;  // (an empty entry: every slot past the end, as in locate())
    // "shortcut" (:119): per problem, multiple scattering is switched off when max_l |omega w_l / 2| <= 1e-8
    double cm_ = 0.0;
    for (int ell = id.mg + j; ell < P; ell += NP) cm_ = fmax(cm_, fabs(0.5 * om * wl[ell]));
    const unsigned long long livemask = __ballot(cm_ > 1e-8);
    if (d.beam && ((livemask >> (NP * grp)) & 0xffffull)) {  // the beam sums of this lane's stream (not a matrix product)
      const double* Y0b = d.Y0 + ((long)c * d.M + m) * P;
      for (int ell = id.mg; ell < P; ell += 2) {
        xe_sum = fma(om * wl[ell] * Ym[(long)ell * NP + j], Y0b[ell], xe_sum);
        if (ell + 1 < P) xo_sum = fma(om * wl[ell + 1] * Ym[(long)(ell + 1) * NP + j], Y0b[ell + 1], xo_sum);
      }
    }
    const double S_i = d.S[i16], invmu_c = d.invmu[i16];
    v4d accP[GPW], accQ[GPW];
#pragma unroll
    for (int g = 0; g < GPW; ++g)
#pragma unroll
      for (int q = 0; q < 4; ++q) accP[g][q] = accQ[g][q] = (4 * q + k4 == i16) ? invmu_c : 0.0;
    const double* wlg[GPW];
    double omg[GPW];
#pragma unroll
    for (int g = 0; g < GPW; ++g) {  // wave-uniform: the layers of the four problems (scalar loads)
      const int slot = chunk * GPW + g;
      const int sl = slot < d.ln ? slot : d.ln - 1;
      const int lg = d.ln == d.L ? d.lperm[(long)c * d.L + sl] : d.l0 + sl;
      wlg[g] = d.wleg + ((long)c * d.L + lg) * P;
      omg[g] = ((livemask >> (NP * g)) & 0xffffull) ? -d.omega[(long)c * d.L + lg] : 0.0;
    }
    for (int base = id.mg; base < P; base += 8) {  // 4 even and 4 odd terms per chunk
      const int le = base + 2 * k4, lo = le + 1;
      const double ye = le < P ? Ym[(long)le * NP + i16] * S_i : 0.0;
      const double yo = lo < P ? Ym[(long)lo * NP + i16] * S_i : 0.0;
#pragma unroll
      for (int g = 0; g < GPW; ++g) {
        const double ce = le < P ? omg[g] * wlg[g][le] : 0.0, co = lo < P ? omg[g] * wlg[g][lo] : 0.0;
        accP[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(ce * ye, ye, accP[g], 0, 0, 0);
        accQ[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(co * yo, yo, accQ[g], 0, 0, 0);
      }
    }
    // D layout -> one column per lane: T_k = rows (R0[k], R1[k], R2[k], R3[k]) of the registers R_g = acc[g][q]
    double pcol[NP], qcol[NP];
    auto transpose4 = [](const v4d (&acc)[GPW], const int q, double (&out)[NP]) {
      unsigned r[4][2];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        r[g][0] = (unsigned)__double2loint(acc[g][q]);
        r[g][1] = (unsigned)__double2hiint(acc[g][q]);
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const u2 s01 = __builtin_amdgcn_permlane16_swap(r[0][h], r[1][h], false, false);  // (R0[0],R1[0],R0[2],R1[2]), (R0[1],R1[1],R0[3],R1[3])
        const u2 s23 = __builtin_amdgcn_permlane16_swap(r[2][h], r[3][h], false, false);
        const u2 t02 = __builtin_amdgcn_permlane32_swap(s01[0], s23[0], false, false);    // T_0, T_2
        const u2 t13 = __builtin_amdgcn_permlane32_swap(s01[1], s23[1], false, false);    // T_1, T_3
        r[0][h] = t02[0];
        r[2][h] = t02[1];
        r[1][h] = t13[0];
        r[3][h] = t13[1];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) out[4 * q + k] = __hiloint2double((int)r[k][1], (int)r[k][0]);
    };
    if constexpr (GPW == 4) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        transpose4(accP, q, pcol);
        transpose4(accQ, q, qcol);
      }
    }
    dinv[j] = cholesky_columns<NP>(pcol, j);  // Pm = L L^T
    cholesky_columns<NP>(qcol, j);            // Qm = R R^T
#pragma unroll
    for (int i = 0; i < NP; ++i)
        if (!PACKED || j <= i) L_[lix(i, 0) + j] = pcol[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      double a = 0.0;
#pragma unroll
      for (int r = i; r < NP; ++r) a += L_[lix(r, i)] * qcol[r];
      w[i] = a;
    }
  } else if constexpr (NP == 32) {
    // One parity at a time: Pm is assembled, factorised and parked in LDS before Qm is touched, so that the accumulator
    // and the Cholesky column of only ONE of the two matrices are alive at once (NP = 32: 128 VGPRs less).
    // this lane's two moments of the layer (terms id.mg + 2 j and id.mg + 2 j + 1): one load each, then LDS
    const int le = id.mg + 2 * j, lo = le + 1;
    const double wle = le < P ? wl[le] : 0.0, wlo = lo < P ? wl[lo] : 0.0;
    const double* Y0b = d.Y0 + ((long)c * d.M + m) * P;  // Ybar_l^m(-mu0): read only with a beam
    const double y0e = (d.beam && le < P) ? Y0b[le] : 0.0, y0o = (d.beam && lo < P) ? Y0b[lo] : 0.0;
    double cmax = fmax(fabs(0.5 * om * wle), fabs(0.5 * om * wlo));
    cmax = fmax(cmax, xor_lane<1>(cmax));
    cmax = fmax(cmax, xor_lane<2>(cmax));
    cmax = fmax(cmax, xor_lane<4>(cmax));
    cmax = fmax(cmax, xor_lane<8>(cmax));
    cmax = fmax(cmax, xor_lane<16>(cmax));
    // "shortcut" of the reference when multiple scattering is insignificant (:119, :162-168): the layer
    // is treated as non-scattering; the general path then gives G = [[0,D],[D,0]], k = 1/mu, B = 0.
    const double live = (cmax > 1e-8) ? 1.0 : 0.0;
    auto assemble = [&](const int first, const double wmine, const double y0mine, double (&col)[NP], double& beam_sum) {
      // M^-1 - S (2 sum_{l = first, first + 2, ...} c_l Y_l Y_l^T) S
      const int nrows = (P - first + 1) / 2;  // rows first, first + 2, ... < P (P <= 2 NP = 64: at most TROWS of them)
      __syncthreads();  // the previous parity's readers are done with the staging area
      {
        constexpr int TRIPS = TROWS * (NP / 2) / 64;  // 16 bytes per lane and trip, rows back to back; every load issued
        double2 v[TRIPS];                             // before the first is stored (indices clamped, not predicated)
#pragma unroll
        for (int t = 0; t < TRIPS; ++t) {
          const int idx = t * 64 + (int)threadIdx.x, row = idx / (NP / 2), q = idx % (NP / 2);
          const int ell = first + 2 * row < P ? first + 2 * row : P - 1;
          v[t] = *reinterpret_cast<const double2*>(Ym + (long)ell * NP + 2 * q);
        }
#pragma unroll
        for (int t = 0; t < TRIPS; ++t) {
          const int idx = t * 64 + (int)threadIdx.x;
          if (idx < nrows * (NP / 2)) *reinterpret_cast<double2*>(&sTab[idx * 2]) = v[t];
        }
      }
      sCoef[grp * TROWS + j] = om * wmine;
      if (grp == 0) sY0[j] = y0mine;
      __syncthreads();
      // acc[i] += coef Yr[i] with the table row spread over the lanes of every DPP row (lane jj holds Yr[jj] and Yr[16 + jj]):
      // the multiplier reaches the FMA as a DPP row broadcast -- one v_fmac_f64_dpp per term and two 8-byte LDS reads per
      // row.  (Broadcast LDS reads of the row, 16 x ds_read_b128 per wavefront and row, made the loop LDS-bandwidth-bound:
      // 8 wavefronts per CU x 1 KiB per read against 128 B per clock.)
      double acc[NP];
#pragma unroll
      for (int i = 0; i < NP; ++i) acc[i] = 0.0;
      const int jj = j & 15;
      for (int k = 0; k < nrows; ++k) {
        const double y0 = sTab[k * NP + jj], y1 = sTab[k * NP + 16 + jj];
        const double coef = sCoef[grp * TROWS + k] * (j < 16 ? y0 : y1);
        beam_sum = fma(live * coef, sY0[k], beam_sum);
        RowFmacDpp<0>::run(acc, y0, y1, coef);
      }
#pragma unroll
      for (int i = 0; i < NP; ++i) col[i] = (i == j ? invmu_j : 0.0) - d.S[i] * (live * acc[i]) * S_j;
    };
    {
      double pcol[NP];
      assemble(id.mg, wle, y0e, pcol, xe_sum);  // Pm = M^-1 - S Ae S  (D+/D- split by parity of l - m, :123-125)
      RTD_ESTAMP(1);
      dinv[j] = cholesky_columns32(pcol, j, sTab);  // Pm = L L^T
      RTD_ESTAMP(2);
#pragma unroll
      for (int i = 0; i < NP; ++i)
        if (!PACKED || j <= i) L_[lix(i, 0) + j] = pcol[i];
    }
    double qcol[NP];
    assemble(id.mg + 1, wlo, y0o, qcol, xo_sum);  // Qm = M^-1 - S Ao S
    RTD_ESTAMP(3);
    cholesky_columns32(qcol, j, sTab);  // Qm = R R^T
    RTD_ESTAMP(4);
    __syncthreads();
    {  // F = L^T R with the rows of L spread over the lanes (LRowN)
      const int t16 = j & 15;
#pragma unroll
      for (int i = 0; i < NP; ++i) w[i] = 0.0;
      static_for<0, NP>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        const LRowN<NP> row = LRowN<NP>::template load<r, PACKED>(L_, t16);
        AxpyRowN<NP, 0, r + 1>::run(w, row, qcol[r]);
        if constexpr ((r & 3) == 3) RTD_FENCE();
      });
    }
  } else if constexpr (NP == 64) {
    // 66 ... 128 streams: one wavefront per SIMD (512 registers), nothing hides a latency, and only the 256 architectural registers
    // can be VALU operands.  (a) One parity at a time, as at NP = 32: Pm is assembled, factorised and parked in LDS before Qm is
    // touched -- with both accumulators and both Cholesky columns alive, half of them sat in AGPRs and every FMA on them paid
    // two v_accvgpr_read and two v_accvgpr_write (970 cycles per moment, s_memtime stamps; 2 300 in the scalar-load form of round
    // 3, one or two memory latencies per moment).  (b) Lane jj of every DPP row loads the elements jj, 16 + jj, 32 + jj, 48 + jj
    // of a moment's table row -- eight moments of the parity requested together, one chunk ahead of their use -- and the
    // multiplier of acc[i] reaches its FMA as a DPP row broadcast of lane i % 16 (one v_fmac_f64_dpp per term, as at 64 streams).
    typedef const double __attribute__((address_space(4))) kdouble;
    kdouble* wk = (kdouble*)wl;    // wave-uniform: s_load
    kdouble* Y0k = (kdouble*)(d.Y0 + ((long)c * d.M + m) * P);  // Ybar_l^m(-mu0) (allocated with or without a beam)
    const int jj = j & 15;
    const bool jr1 = (j & 16) != 0, jr2 = (j & 32) != 0;  // the lane's own DPP row: which of its four elements is Yr[j]
    // "shortcut" of the reference when multiple scattering is insignificant (:119, :162-168): max_l |omega w_l / 2| over all the
    // layer's moments (lane j: the terms mg + j and mg + 64 + j; P <= 2 NP)
    double cmax = 0.0;
    {
      const int l0 = id.mg + j, l1 = l0 + NP;
      if (l0 < P) cmax = fabs(0.5 * om * wl[l0]);
      if (l1 < P) cmax = fmax(cmax, fabs(0.5 * om * wl[l1]));
      cmax = fmax(cmax, xor_lane<1>(cmax));
      cmax = fmax(cmax, xor_lane<2>(cmax));
      cmax = fmax(cmax, xor_lane<4>(cmax));
      cmax = fmax(cmax, xor_lane<8>(cmax));
      cmax = fmax(cmax, xor_lane<16>(cmax));
      cmax = fmax(cmax, xor_lane<32>(cmax));
    }
    const double live = (cmax > 1e-8) ? 1.0 : 0.0;
    auto assemble = [&](const int first, double (&col)[NP], double& beam_sum) {
      // M^-1 - S (2 sum_{l = first, first + 2, ...} c_l Y_l Y_l^T) S
      double acc[NP];
#pragma unroll
      for (int i = 0; i < NP; ++i) acc[i] = 0.0;
      constexpr int CH = 8;
      double yv[CH][4], yn[CH][4], wv[CH], wn[CH], y0v[CH], y0n[CH];  // (wv, y0v: wave-uniform, they live in SGPRs)
#pragma unroll
      for (int e = 0; e < CH; ++e) {
        const int ell = first + 2 * e < P ? first + 2 * e : P - 1;  // (clamped, not predicated: no load waits behind a branch)
#pragma unroll
        for (int g = 0; g < 4; ++g) yv[e][g] = Ym[(long)ell * NP + 16 * g + jj];
        wv[e] = wk[ell];
        y0v[e] = Y0k[ell];
      }
#pragma unroll 1
      for (int base = first; base < P; base += 2 * CH) {
#pragma unroll
        for (int e = 0; e < CH; ++e) {
          const int ell = base + 2 * (CH + e) < P ? base + 2 * (CH + e) : P - 1;
#pragma unroll
          for (int g = 0; g < 4; ++g) yn[e][g] = Ym[(long)ell * NP + 16 * g + jj];
          wn[e] = wk[ell];
          y0n[e] = Y0k[ell];
        }
#pragma unroll
        for (int e = 0; e < CH; ++e) {
          if (base + 2 * e < P) {  // (wave-uniform)
            const double ylo = jr1 ? yv[e][1] : yv[e][0], yhi = jr1 ? yv[e][3] : yv[e][2];
            const double coef = om * wv[e] * (jr2 ? yhi : ylo);  // 2 c_l Yr[j], c_l = omega w_l / 2
            beam_sum = fma(coef, y0v[e], beam_sum);
            RowFmacDpp64<0>::run(acc, yv[e], coef);
          }
        }
#pragma unroll
        for (int e = 0; e < CH; ++e) {
#pragma unroll
          for (int g = 0; g < 4; ++g) yv[e][g] = yn[e][g];
          wv[e] = wn[e];
          y0v[e] = y0n[e];
        }
      }
      beam_sum *= live;
#pragma unroll
      for (int i = 0; i < NP; ++i) col[i] = (i == j ? invmu_j : 0.0) - d.S[i] * (live * acc[i]) * S_j;
    };
    {
      double pcol[NP];
      assemble(id.mg, pcol, xe_sum);  // Pm = M^-1 - S Ae S  (D+/D- split by parity of l - m, :123-125)
      RTD_ESTAMP(1);
      dinv[j] = cholesky_columns<NP>(pcol, j);  // Pm = L L^T
      RTD_ESTAMP(2);
#pragma unroll
      for (int i = 0; i < NP; ++i) L_[lix(i, 0) + j] = pcol[i];
    }
    double qcol[NP];
    assemble(id.mg + 1, qcol, xo_sum);  // Qm = M^-1 - S Ao S
    RTD_ESTAMP(3);
    cholesky_columns<NP>(qcol, j);  // Qm = R R^T
    RTD_ESTAMP(4);
    __syncthreads();
    {  // F = L^T R with the rows of L spread over the lanes (LRowN; round 5: as at NP = 32)
      const int t16 = j & 15;
#pragma unroll
      for (int i = 0; i < NP; ++i) w[i] = 0.0;
      static_for<0, NP>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        const LRowN<NP> row = LRowN<NP>::template load<r, PACKED>(L_, t16);
        AxpyRowN<NP, 0, r + 1>::run(w, row, qcol[r]);
        if constexpr ((r & 3) == 3) RTD_FENCE();
      });
    }
  } else {
  // D+/D- split by parity of (l - m): Ae = 2 sum_even c_l Y_l Y_l^T, Ao likewise (:123-125)
  double acc_e[NP], acc_o[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) acc_e[i] = acc_o[i] = 0.0;
  double cmax = 0.0;
  const double* Y0b = d.Y0 + ((long)c * d.M + m) * P;  // Ybar_l^m(-mu0): wave-uniform (scalar loads); read only with a beam
  for (int ell = id.mg; ell < P; ell += 2) {
    {
      const double cl = 0.5 * om * wl[ell];
      cmax = fmax(cmax, fabs(cl));
      const double* Yr = Ym + (long)ell * NP;
      const double coef = 2.0 * cl * Yr[j];
      if (d.beam) xe_sum = fma(coef, Y0b[ell], xe_sum);
#pragma unroll
      for (int i = 0; i < NP; ++i) acc_e[i] += coef * Yr[i];
    }
    if (ell + 1 < P) {
      const double cl = 0.5 * om * wl[ell + 1];
      cmax = fmax(cmax, fabs(cl));
      const double* Yr = Ym + (long)(ell + 1) * NP;
      const double coef = 2.0 * cl * Yr[j];
      if (d.beam) xo_sum = fma(coef, Y0b[ell + 1], xo_sum);
#pragma unroll
      for (int i = 0; i < NP; ++i) acc_o[i] += coef * Yr[i];
    }
  }
  // "shortcut" of the reference when multiple scattering is insignificant (:119, :162-168): the layer
  // is treated as non-scattering; the general path then gives G = [[0,D],[D,0]], k = 1/mu, B = 0.
  if (!(cmax > 1e-8)) {
#pragma unroll
    for (int i = 0; i < NP; ++i) acc_e[i] = acc_o[i] = 0.0;
    xe_sum = xo_sum = 0.0;
  }
  {
    double pcol[NP], qcol[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const double Si = d.S[i];
      pcol[i] = (i == j ? invmu_j : 0.0) - Si * acc_e[i] * S_j;  // Pm = M^-1 - S Ae S
      qcol[i] = (i == j ? invmu_j : 0.0) - Si * acc_o[i] * S_j;  // Qm = M^-1 - S Ao S
    }
    RTD_ESTAMP(1);
    dinv[j] = cholesky_columns<NP>(pcol, j);  // Pm = L L^T
    RTD_ESTAMP(2);
    RTD_ESTAMP(3);
    cholesky_columns<NP>(qcol, j);  // Qm = R R^T
    RTD_ESTAMP(4);
#pragma unroll
    for (int i = 0; i < NP; ++i)
        if (!PACKED || j <= i) L_[lix(i, 0) + j] = pcol[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      double a = 0.0;
#pragma unroll
      for (int r = i; r < NP; ++r) a += L_[lix(r, i)] * qcol[r];
      w[i] = a;
    }
  }
  }
  RTD_ESTAMP(5);
  // one-sided Jacobi on the columns of F
  int nsweep = 0;
  bool converged = false;
  {
    constexpr int H = NP / 2;
    const int u = j / H, p = j % H;  // lane (p, u): half u of the two columns of pair slot p
    double xh[H], yh[H];
#pragma unroll
    for (int i = 0; i < H; ++i) {  // columns p and p + H start in slot p: trade the halves with lane ^ H
      const double recv = xor_lane<H>(u ? w[i] : w[H + i]);
      xh[i] = u ? recv : w[i];
      yh[i] = u ? w[H + i] : recv;
    }
    int ix = p, iy = p + H;  // the columns' starting indices travel with them
    for (int sweep = 0; sweep < 40; ++sweep) {
      int notconv = 0;
      double ax = 0.0, ay = 0.0;
#pragma unroll
      for (int i = 0; i < H; ++i) {
        ax = fma(xh[i], xh[i], ax);
        ay = fma(yh[i], yh[i], ay);
      }
      ax += xor_lane<H>(ax);
      ay += xor_lane<H>(ay);
      if constexpr (RTD_JAC_FAST && NP >= 32) {
        double sx = 1.0, sy = 1.0, rx = 1.0, ry = 1.0;
        FastPairStep<NP, 0>::run(xh, yh, ax, ay, ix, iy, p, notconv, sx, sy, rx, ry);
#pragma unroll
        for (int i = 0; i < H; ++i) {  // the scales back into the columns
          xh[i] *= sx;
          yh[i] *= sy;
        }
      } else {
        PairStep<NP, 0>::run(xh, yh, ax, ay, ix, iy, p, notconv);
      }
      ++nsweep;
      if (!__any(notconv)) {
        converged = true;
        break;
      }
    }
    // back to one column per lane: lane (p, 0) takes column X of its slot, lane (p, 1) column Y ...
#pragma unroll
    for (int i = 0; i < H; ++i) {
      const double recv = xor_lane<H>(u ? xh[i] : yh[i]);
      w[i] = u ? recv : xh[i];
      w[H + i] = u ? yh[i] : recv;
    }
    // ... and every column returns to the lane it started in (its index j): the source lane of lane j through LDS
    int* where = reinterpret_cast<int*>(sV[grp][0]);
    where[u ? iy : ix] = j;
    __syncthreads();
    const int src = where[j];
#pragma unroll
    for (int i = 0; i < NP; ++i) w[i] = __shfl(w[i], src, NP);
    __syncthreads();
  }
  RTD_ESTAMP(6);
#ifdef RTD_EIG_STAMPS
  est_sweeps = nsweep;
#endif
  if (threadIdx.x == 0 && nsweep > *(volatile int*)d.sweeps) atomicMax(d.sweeps, nsweep);
  if (!converged && threadIdx.x == 0) rtd_raise(d, RTD_ST_JACOBI, id.mg, id.c);  // NaN input (failed Cholesky) also ends here
  }
  // ---- stage 2: eigenvector blocks and particular solutions.  The lane's identifiers are rebuilt from an opaque copy
  //      of the lane index, so that none of them is kept (and spilled) across the Jacobi loop.
  int tx = threadIdx.x;
  asm volatile("" : "+v"(tx));
  const int grp = tx / NP, j = tx % NP;
  const ProbId id = locate<NP>(d, tx);
  const int P = d.P, m = id.m, c = id.c, l = id.l;
  const bool valid = id.valid;
  const long base = id.pid;
  double* L_ = sL[grp];
  double* v0 = sV[grp][0];
  double* v1 = sV[grp][1];
  double* v2 = sV[grp][2];
  double* dinv = sV[grp][3];  // 1 / L[i][i]
  const double* wl = d.wleg + ((long)c * d.L + l) * P;
  const double om = d.omega[(long)c * d.L + l];
  const double* Ym = d.Y + (long)m * P * NP;
  double invmu_j = d.invmu[j], T_j = d.T[j];
  double dts;  // the layer's scaled optical thickness (requested with the quadrature values, ahead of every store of the stage)
  {
    const double* ts0 = d.taus0 + (long)c * (d.L + 1);
    dts = ts0[l + 1] - ts0[l];
  }
  double k2 = 0.0;
#pragma unroll
  for (int i = 0; i < NP; ++i) k2 += w[i] * w[i];
  const double rk0 = fast_rsqrt(k2), kj = k2 * rk0;
  // a non-positive Cholesky pivot (phase function not positive definite after delta-M scaling) or an overflow shows up
  // as a non-finite or non-positive eigenvalue: the reference's eig / sqrt would return NaN here (:186)
  if (valid && !(k2 > 0.0 && k2 < 1e300)) rtd_raise(d, RTD_ST_CHOL, id.mg, id.c);
  double zc[NP];
  {
#pragma unroll
    for (int i = 0; i < NP; ++i) zc[i] = w[i] * rk0;
  }

  // eigenvector blocks (:190-198): V = T^-1 L^-T Z, U = (alpha+beta) V / k = -T^-1 L Z / k; stored as
  // Y = L^-T Z and A = L Z, from which Gp = (Y - A/k)/T, Gm = (Y + A/k)/T, V^-1 = A^T T, U^-1 = -k Y^T T
  double ya[NP];
  if constexpr (NP >= 32) {  // the rows of L spread over the lanes (LRowN): ya[r] final, then its multiples leave the rows above
    const int t16 = j & 15;
#pragma unroll
    for (int i = 0; i < NP; ++i) ya[i] = zc[i];
    static_for<0, NP>([&](auto kc) {
      constexpr int r = NP - 1 - decltype(kc)::value;
      ya[r] *= dinv[r];
      if constexpr (r > 0) {
        const LRowN<NP> row = LRowN<NP>::template load<r, PACKED>(L_, t16);
        AxpyRowN<NP, 0, r>::run(ya, row, -ya[r]);
      }
      if constexpr ((r & 3) == 0) RTD_FENCE();
    });
  } else {
#pragma unroll
  for (int i = NP - 1; i >= 0; --i) {
    double a = zc[i];
#pragma unroll
    for (int r = i + 1; r < NP; ++r) a -= L_[lix(r, i)] * ya[r];
    ya[i] = a * dinv[i];
    RTD_FENCE();
  }
  }
  RTD_ESTAMP(7);
  // exp(-k dtau*): formed here, ahead of every store of the stage -- the load of the layer's boundaries behind them would wait for
  // their acknowledgement (loads, stores and scratch accesses share one in-order counter)
  double ekj = 0.0;
  if constexpr (NP == 32) ekj = exp(-kj * dts);
  // every load of the stage has landed before its first store: used for the first time behind the stores of Y, the quadrature
  // values requested at the top of the stage made the wavefront wait for those stores (one in-order counter for loads and stores)
  asm volatile("" : "+v"(invmu_j), "+v"(T_j), "+v"(dts));
  auto store_Y = [&]() {
    if constexpr (NP == 32) {  // every value in a register BEFORE the first store: a spilled one reloaded between two stores waits
      //                         for the acknowledgement of the stores in front of it
      asm volatile("" : "+v"(ya[0]), "+v"(ya[1]), "+v"(ya[2]), "+v"(ya[3]), "+v"(ya[4]), "+v"(ya[5]), "+v"(ya[6]), "+v"(ya[7]), "+v"(ya[8]),
                        "+v"(ya[9]), "+v"(ya[10]), "+v"(ya[11]), "+v"(ya[12]), "+v"(ya[13]), "+v"(ya[14]), "+v"(ya[15]));
      asm volatile("" : "+v"(ya[16]), "+v"(ya[17]), "+v"(ya[18]), "+v"(ya[19]), "+v"(ya[20]), "+v"(ya[21]), "+v"(ya[22]), "+v"(ya[23]),
                        "+v"(ya[24]), "+v"(ya[25]), "+v"(ya[26]), "+v"(ya[27]), "+v"(ya[28]), "+v"(ya[29]), "+v"(ya[30]), "+v"(ya[31]));
    }
    if (valid) {
      double* Ym = d.Ym + base * NP * NP;
#pragma unroll
      for (int i = 0; i < NP; ++i) Ym[i * NP + j] = ya[i];
      d.kk[base * NP + j] = kj;
      if constexpr (NP != 32) ekj = exp(-kj * dts);
      d.Ek[base * NP + j] = ekj;
    }
  };
  // NP = 32: Y leaves behind the beam stage, in one run of stores.  This kernel reloads spilled registers in the beam stage: stored
  // here, every such reload waited for the acknowledgement of the 34 stores in front of it (the beam stage took 23 k of a
  // wavefront's 178 k cycles: s_memtime stamps, profiles/r05_eigen32_phases.txt)
  if constexpr (NP != 32) store_Y();

  // beam particular solution (:143-152, :226-231) through the spectral decomposition:
  //  s = B+ + B-, dd = B+ - B- ;  (I/mu0^2 - Qm Pm) T s = T(x+ + x-)/mu0 - Qm T (x+ - x-)
  //  T dd = mu0 [ T (x+ - x-) - Pm T s ],  Qm Pm = L^-T Z k^2 Z^T L^T
  double bv_up = 0.0, bv_dn = 0.0;  // (NP = 32: stored with A at the end of the stage, no store in front of the stage's reloads)
  if (d.beam) {
    const double mu0 = mu0_c;
    // X^e_j, X^o_j of this lane's stream (:143-152): I0/(4 pi) (2 - delta_m0) omega sum_l w_l Ybar_l(-mu0) Y_l[j], from stage 1
    const double fac = I0_c * (0.25 / M_PI) * (id.mg == 0 ? 1.0 : 2.0);
    const double xe = fac * xe_sum, xo = fac * xo_sum;
    const double txd = 2.0 * T_j * xe * invmu_j;  // T (x+ - x-)
    v0[j] = txd;
    __syncthreads();
    // Qm v0 through Qm = L^-T H L^-1 = Y k^2 Y^T (exact for the rotated columns whatever their convergence):
    // lane e forms k_e^2 (Y^T v0)_e, the sum over the eigen-index is a transposed reduction in registers
    double qv;
    {
      double tq = 0.0;
#pragma unroll
      for (int i = 0; i < NP; ++i) tq += ya[i] * v0[i];
      tq *= k2;
      if constexpr (NP == 32) {
        qv = transpose_reduce_scaled<NP>(ya, tq, j);
      } else {
        double x[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) x[i] = ya[i] * tq;
        qv = transpose_reduce<NP>(x, j);
      }
    }
    const double rmu0 = fast_rcp(mu0);
    const double rhat = 2.0 * T_j * xo * invmu_j * rmu0 - qv;
    v1[j] = rhat;
    __syncthreads();
    double g = 0.0;  // g = L^T rhat
#pragma unroll
    for (int r = 0; r < NP; ++r) g += ((!PACKED || r >= j) ? L_[lix(r, 0) + j] : 0.0) * v1[r];  // column j of L (zero above the diagonal)
    v2[j] = g;
    __syncthreads();
    double h = 0.0;  // h = Z^T g / (1/mu0^2 - k^2)
#pragma unroll
    for (int i = 0; i < NP; ++i) h += zc[i] * v2[i];
    h *= fast_rcp(rmu0 * rmu0 - k2);
    // shat = L^-T Z h = Y h  and  t = L^T shat = Z h  (sums over the eigen-index = lanes): no triangular solve
    double sh, e;
    if constexpr (NP == 32) {
      sh = transpose_reduce_scaled<NP>(ya, h, j);
      e = transpose_reduce_scaled<NP>(zc, h, j);
    } else {
      double x[NP];
#pragma unroll
      for (int i = 0; i < NP; ++i) x[i] = ya[i] * h;
      sh = transpose_reduce<NP>(x, j);
#pragma unroll
      for (int i = 0; i < NP; ++i) x[i] = zc[i] * h;
      e = transpose_reduce<NP>(x, j);
    }
    __syncthreads();
    v2[j] = e;
    __syncthreads();
    double ps = 0.0;  // Pm shat = L (L^T shat) = L t
#pragma unroll
    for (int r = 0; r < NP; ++r) ps += ((!PACKED || r <= j) ? L_[lix(j, 0) + r] : 0.0) * v2[r];  // row j of L
    const double rT = fast_rcp(T_j);
    const double s_j = sh * rT;
    const double d_j = mu0 * (txd - ps) * rT;
    bv_up = 0.5 * (s_j + d_j);
    bv_dn = 0.5 * (s_j - d_j);
    if constexpr (NP != 32) {
      if (valid) {
        d.Bv[base * 2 * NP + j] = bv_up;
        d.Bv[base * 2 * NP + NP + j] = bv_dn;
      }
    }
    // 1/mu0 on an eigenvalue: the reference's solve (:226-231) meets a singular matrix
    if (valid && !(fabs(s_j) + fabs(d_j) < 1e300)) rtd_raise(d, RTD_ST_BEAM, id.mg, id.c);
  }
  __syncthreads();
  RTD_ESTAMP(8);
  // A = L Z after the beam stage: its 2 NP registers are not live while that stage runs
  double aa[NP];
  if constexpr (NP == 32) {  // the rows of L spread over the lanes (LRowN)
    store_Y();
    const int t16 = j & 15;
    static_for<0, NP>([&](auto rc) {
      constexpr int r = decltype(rc)::value;
      const LRowN<NP> row = LRowN<NP>::template load<r, PACKED>(L_, t16);
      double a[NP / 16] = {};
      DotRowN<NP, 0, r + 1>::run(a, row, zc);
      aa[r] = r >= 16 ? a[0] + a[1] : a[0];
      if constexpr ((r & 3) == 3) RTD_FENCE();
    });
    asm volatile("" : "+v"(aa[0]), "+v"(aa[1]), "+v"(aa[2]), "+v"(aa[3]), "+v"(aa[4]), "+v"(aa[5]), "+v"(aa[6]), "+v"(aa[7]), "+v"(aa[8]),
                      "+v"(aa[9]), "+v"(aa[10]), "+v"(aa[11]), "+v"(aa[12]), "+v"(aa[13]), "+v"(aa[14]), "+v"(aa[15]));
    asm volatile("" : "+v"(aa[16]), "+v"(aa[17]), "+v"(aa[18]), "+v"(aa[19]), "+v"(aa[20]), "+v"(aa[21]), "+v"(aa[22]), "+v"(aa[23]),
                      "+v"(aa[24]), "+v"(aa[25]), "+v"(aa[26]), "+v"(aa[27]), "+v"(aa[28]), "+v"(aa[29]), "+v"(aa[30]), "+v"(aa[31]));
  } else if constexpr (NP == 64) {  // the same product at 128 streams: four registers per spread row, four accumulation chains
    const int t16 = j & 15;
    static_for<0, NP>([&](auto rc) {
      constexpr int r = decltype(rc)::value;
      const LRowN<NP> row = LRowN<NP>::template load<r, PACKED>(L_, t16);
      double a[NP / 16] = {};
      DotRowN<NP, 0, r + 1>::run(a, row, zc);
      aa[r] = (a[0] + a[1]) + (a[2] + a[3]);
      if constexpr ((r & 3) == 3) RTD_FENCE();
    });
  } else {
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    double a = 0.0;
#pragma unroll
    for (int r = 0; r <= i; ++r) a += L_[lix(i, r)] * zc[r];
    aa[i] = a;
    if (NP != 64 || (i & 7) == 7) RTD_FENCE();  // (NP = 64, one wavefront per SIMD: eight rows between fences, so that their LDS reads overlap)
  }
  }
  if (valid) {
    double* Am = d.Am + base * NP * NP;
#pragma unroll
    for (int i = 0; i < NP; ++i) Am[i * NP + j] = aa[i];
    if constexpr (NP == 32) {
      if (d.beam) {
        d.Bv[base * 2 * NP + j] = bv_up;
        d.Bv[base * 2 * NP + NP + j] = bv_dn;
      }
    }
  }

  // isotropic (thermal) source, Fourier mode 0 only (subroutines.py:746-862, _assemble.py:124).
  // A wave holds layers of ONE (c, m), so the branch is wave-uniform.
  if (d.Ns > 0 && id.mg == 0) {
    const bool act = true;
    // NP = 32: this branch (Fourier mode 0 only: one wavefront in M) reads the lane's columns of Y and A back from the arrays it
    // has just stored instead of keeping 128 registers alive across the beam stage for it -- with them the 64-stream kernel
    // spilled 68-110 registers in stage 2, and every reload behind a global store is a wait for that store's acknowledgement
    // (loads, stores and scratch accesses share one in-order counter).  A lane reads only what it wrote itself.
    double yl[NP == 32 ? NP : 1], al[NP == 32 ? NP : 1];
    if constexpr (NP == 32) {
      const double* Yg = d.Ym + base * NP * NP;
      const double* Ag = d.Am + base * NP * NP;
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        yl[i] = Yg[i * NP + j];
        al[i] = Ag[i * NP + j];
      }
    }
    const auto& ya_t = [&]() -> const double(&)[NP] { if constexpr (NP == 32) return yl; else return ya; }();
    const auto& aa_t = [&]() -> const double(&)[NP] { if constexpr (NP == 32) return al; else return aa; }();
    // zneg_j = -k_j/2 [Z^T L^-1 (T/mu)]_j = -k_j/2 sum_i Y[i][j] T_i/mu_i   (Y = L^-T Z: no triangular solve)
    double zn = 0.0;
#pragma unroll
    for (int i = 0; i < NP; ++i) zn += ya_t[i] * (d.T[i] * d.invmu[i]);
    zn *= -0.5 * kj;
    if (valid && act) d.zneg[((long)c * d.L + l) * NP + j] = zn;
    const double* sp = d.spoly + ((long)c * d.L + l) * d.Ns;
    const double rk = rk0;
    // v_l at the layer's own boundaries (vb: what every boundary-condition kernel reads -- none evaluates a polynomial).  The
    // coefficients sp are about the layer's TOP (rtd_dd.h), so v_l(x) = sum_q dq[q] x^q with x the scaled depth below the top:
    // the top is x = 0 (the constant term alone), the bottom x = the layer's scaled thickness.
    const double ts_top = 0.0, ts_bot = d.taus0[(long)c * (d.L + 1) + l + 1] - d.taus0[(long)c * (d.L + 1) + l];
    double vtu = 0.0, vtd = 0.0, vbu = 0.0, vbd = 0.0, tpt = 1.0, tpb = 1.0;
    for (int q = 0; q < d.Ns; ++q) {
      // b_q(K) = sum_{jj>=q} jj!/q! a_jj K^-(jj-q+1), K = -k (first N eigen-columns) and +k
      double bneg = 0.0, bpos = 0.0, ratio = 1.0, pw_pos = rk, pw_neg = -rk;
      for (int jj = q; jj < d.Ns; ++jj) {
        bpos += ratio * sp[jj] * pw_pos;
        bneg += ratio * sp[jj] * pw_neg;
        ratio *= (double)(jj + 1);
        pw_pos *= rk;
        pw_neg *= -rk;
      }
      const double a = zn * bneg, b = -zn * bpos;
      // up-streams: Gp a + Gm b ; down-streams: Gm a + Gp b  (sum over eigen-index = lanes), with
      // Gp = (Y - A/k)/T, Gm = (Y + A/k)/T:  up = [Y (a+b) - A (a-b)/k]/T, down = [Y (a+b) + A (a-b)/k]/T
      const double sab = a + b, dab = (a - b) * rk;
      const double rT = 1.0 / T_j;
      double up, dn;
      if constexpr (NP == 32) {  // one after the other, the products formed inside the first level: 16 temporaries, not 2 x 32
        up = transpose_reduce_scaled2<NP>(ya_t, sab, aa_t, -dab, j) * rT;
        dn = transpose_reduce_scaled2<NP>(ya_t, sab, aa_t, dab, j) * rT;
      } else {
        double xu[NP], xd[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) {
          const double py = ya_t[i] * sab, pa = aa_t[i] * dab;
          xu[i] = py - pa;
          xd[i] = py + pa;
        }
        up = transpose_reduce<NP>(xu, j) * rT;
        dn = transpose_reduce<NP>(xd, j) * rT;
      }
      if (valid && act) {
        double* dq = d.dq + (((long)c * d.L + l) * d.Ns + q) * 2 * NP;
        dq[j] = up;
        dq[NP + j] = dn;
      }
      vtu += up * tpt;
      vtd += dn * tpt;
      vbu += up * tpb;
      vbd += dn * tpb;
      tpt *= ts_top;
      tpb *= ts_bot;
    }
    if (valid && act) {
      double* vb = d.vb + ((long)c * d.L + l) * 4 * NP;
      vb[j] = vtu;
      vb[NP + j] = vtd;
      vb[2 * NP + j] = vbu;
      vb[3 * NP + j] = vbd;
    }
    __syncthreads();
  }
  RTD_ESTAMP(9);
#ifdef RTD_EIG_STAMPS
  if (threadIdx.x == 0 && blockIdx.x % 4099 == 7)
    printf("EIGSTAMP np %d m %d sweeps %d : asmP %lld cholP %lld asmQ %lld cholQ %lld F %lld jacobi %lld order+Y %lld beam %lld A+thermal %lld total %lld\n",
           NP, id.mg, est_sweeps, est[1] - est[0], est[2] - est[1], est[3] - est[2], est[4] - est[3], est[5] - est[4], est[6] - est[5],
           est[7] - est[6], est[8] - est[7], est[9] - est[8], est[9] - est[0]);
#endif

}

// ---- Legendre tables: Ybar_l^m(x) = sqrt((l-m)!/(l+m)!) P_l^m(x) without the Condon-Shortley sign
//      (it cancels in every product the path forms); replaces scipy.special.lpmv/poch (:96-109).
__device__ __forceinline__ void ybar_column(int m, int P, double x, double* out, long stride) {
  double v = 1.0;
  const double sx = sqrt(fmax(0.0, 1.0 - x * x));
  for (int jj = 1; jj <= m; ++jj) v *= sqrt((2.0 * jj - 1.0) / (2.0 * jj)) * sx;
  for (int ell = 0; ell < m && ell < P; ++ell) out[ell * stride] = 0.0;
  if (m >= P) return;
  double ym1 = 0.0, y = v;
  out[m * stride] = y;
  for (int ell = m; ell + 1 < P; ++ell) {
    const double yn = ((2.0 * ell + 1.0) * x * y - sqrt((double)(ell + m) * (double)(ell - m)) * ym1) /
                      sqrt((double)(ell + 1 - m) * (double)(ell + 1 + m));
    ym1 = y;
    y = yn;
    out[(ell + 1) * stride] = y;
  }
}

__global__ void rtd_tables_quad_kernel(RtdDev d) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;  // (m, i)
  if (t >= d.M * d.NP) return;
  const int m = t / d.NP, i = t % d.NP;
  double* out = d.Y + (long)m * d.P * d.NP + i;
  if (i >= d.N) {
    for (int ell = 0; ell < d.P; ++ell) out[(long)ell * d.NP] = 0.0;
    return;
  }
  ybar_column(d.m0 + d.mstep * m, d.P, d.mu[i], out, d.NP);
}

__global__ void rtd_tables_mu0_kernel(RtdDev d) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;  // (c, m)
  if (t >= (long)d.C * d.M) return;
  const int c = (int)(t / d.M), m = (int)(t % d.M);
  ybar_column(d.m0 + d.mstep * m, d.P, -d.mu0[c], d.Y0 + t * d.P, 1);
  if (m == 0) {  // beam attenuation at the scaled layer boundaries, used by the boundary-condition kernel
    const double rmu0 = 1.0 / d.mu0[c];
    for (int l = 0; l <= d.L; ++l) d.att[(long)c * (d.L + 1) + l] = exp(-d.taus0[(long)c * (d.L + 1) + l] * rmu0);
  }
}

}  // namespace

void rtd_launch_tables(const RtdDev& d, hipStream_t s, bool with_quad) {
  if (with_quad) {
    const int n = d.M * d.NP;
    hipLaunchKernelGGL(rtd_tables_quad_kernel, dim3((n + 63) / 64), dim3(64), 0, s, d);
  }
  if (d.beam) {
    const long n = (long)d.C * d.M;
    hipLaunchKernelGGL(rtd_tables_mu0_kernel, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, s, d);
  }
}

void rtd_launch_eig(const RtdDev& d, hipStream_t s, int part) {
  // One fused kernel (launched as part 1; parts 0 and 2 are the empty timing slots of the earlier three-kernel form).
  if (part != 1) return;
  const int gpw = 64 / d.NP;
  // (d.nsel > 0: the chunk lists of a lean retained plan's evaluation; the caller leaves it 0 for the one-lane-per-problem kernel)
  const dim3 grid((unsigned)((long)d.C * d.M * (d.nsel > 0 ? d.nsel : (d.ln + gpw - 1) / gpw)));
  // RTD_EIG_MFMA=1: the assembly of Pm, Qm on the matrix cores (NP = 16; A/B runs and a regression test)
  static const bool mfma = getenv("RTD_EIG_MFMA") != nullptr;
#ifndef RTD_EIG_LDS_PAD
#define RTD_EIG_LDS_PAD 0  /* profiling builds only: dynamic LDS requested per workgroup, caps the eigen kernel's wavefronts per CU */
#endif
#define RTD_EIG_CASE(NPV)                                                                        \
  case NPV:                                                                                      \
    hipLaunchKernelGGL((rtd_eigen_kernel<NPV, 2>), grid, dim3(64), RTD_EIG_LDS_PAD, s, d);       \
    break;
  // 2 ... 8 streams: the one-lane-per-problem kernel (rtd_eig_small.hip) unless RTD_EIG_SMALL_V1 asks for rtd_eigen_kernel<4, 2>
  static const bool small_v1 = getenv("RTD_EIG_SMALL_V1") != nullptr;
  if (d.NP == 4 && !small_v1) {
    rtd_launch_eig_small(d, s);
    return;
  }
  switch (d.NP) {
    RTD_EIG_CASE(4)
    RTD_EIG_CASE(8)
    case 16:
      if (mfma) hipLaunchKernelGGL((rtd_eigen_kernel<16, 3>), grid, dim3(64), 0, s, d);
      else hipLaunchKernelGGL((rtd_eigen_kernel<16, 2>), grid, dim3(64), 0, s, d);
      break;
    RTD_EIG_CASE(32)
    RTD_EIG_CASE(64)  // 66 ... 128 streams: one problem per wavefront (round 4: parity-at-a-time DPP assembly, readlane Cholesky)
    default: break;
  }
#undef RTD_EIG_CASE
}
