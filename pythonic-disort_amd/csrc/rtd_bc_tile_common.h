// rtd_bc_tile_common.h -- helpers of the matrix-core-layout boundary-condition kernels (rtd_bc.hip: rtd_bc_mfma_kernel,
// rtd_bc_tile_kernel<T>; rtd_bc_tile2.hip: the lean 64-stream kernel): 16 x 16 tiles in the operand / accumulator layout of
// v_mfma_f64_16x16x4_f64 ("D layout": lane = 16 kq + col, register q holds element [4 q + kq][col]), products, row / column
// sums, the speculative tiled elimination and the rule for chains that are pivoted throughout.  Included inside each file's
// anonymous namespace, after rtd_bc_common.h.
#pragma once

__device__ __forceinline__ v4f64 mm_t(const v4f64& X, const v4f64& Y) {  // X^T Y
  v4f64 acc = {0.0, 0.0, 0.0, 0.0};
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(X[0], Y[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(X[1], Y[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(X[2], Y[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(X[3], Y[3], acc, 0, 0, 0);
  return acc;
}
// value held by lane-row R (compile-time) in the same column, for every lane-row
template <int R>
__device__ __forceinline__ double bcast_row(double v, const int col) {
  return bperm(((R << 4) | col) << 2, v);
}
// sum over the four lane-rows (kq) of the wavefront; the result is replicated over them
__device__ __forceinline__ double sum_kq(double p) {
  p += xor_lane<16>(p);
  p += __shfl_xor(p, 32, 64);
  return p;
}
// sum_rows X[r][col] v[r]  with v in row form (register q = v[4 q + kq]); result in column form
__device__ __forceinline__ double col_dot(const v4f64& X, const v4f64& vr) {
  return sum_kq(X[0] * vr[0] + X[1] * vr[1] + X[2] * vr[2] + X[3] * vr[3]);
}
template <int CTRL>
__device__ __forceinline__ double dpp_add(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
  return v + __hiloint2double(hi, lo);
}
// sum over the 16 lanes of a lane-row (replicated over them): quad permutes, half mirror, mirror
__device__ __forceinline__ double row_sum16(double v) {
  v = dpp_add<0xB1>(v);
  v = dpp_add<0x4E>(v);
  v = dpp_add<0x141>(v);
  v = dpp_add<0x140>(v);
  return v;
}
// sum_cols X[row][c] u[c]  with u in column form; result in row form
__device__ __forceinline__ v4f64 row_dot(const v4f64& X, const double uc) {
  v4f64 r;
  r[0] = row_sum16(X[0] * uc);
  r[1] = row_sum16(X[1] * uc);
  r[2] = row_sum16(X[2] * uc);
  r[3] = row_sum16(X[3] * uc);
  return r;
}
// column form -> row form of a 16-vector
__device__ __forceinline__ v4f64 col_to_row(const double vc, const int rowbase, const int kq) {
  v4f64 r;
  r[0] = bperm((rowbase | kq) << 2, vc);
  r[1] = bperm((rowbase | (4 + kq)) << 2, vc);
  r[2] = bperm((rowbase | (8 + kq)) << 2, vc);
  r[3] = bperm((rowbase | (12 + kq)) << 2, vc);
  return r;
}
__device__ __forceinline__ double readlane_f64(const double v, const int lane_uniform) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane_uniform), __builtin_amdgcn_readlane(__double2loint(v), lane_uniform));
}

// A chain whose result hangs on the last digits of its coefficients: the thermal (polynomial) particular solution of a layer with
// a tiny eigenvalue k is ~ 1/k^(order + 1) times the source and is cancelled by the homogeneous part -- at k = 1.4e-3 (omega =
// 1 - 1e-6) seventeen orders of magnitude above the field, which is then as good as the RELATIVE accuracy of C.  The speculative
// elimination accepts multipliers up to 64 (1e6 in the tiled kernel) and loses two to three digits against the pivoted one
// there (9e-2 against 3e-4 of the field scale on the case of DESIGN.md section 8, the reference: 2e-4); such chains -- mode 0
// with a thermal source and an eigenvalue below RTD_BC_CAREFUL_K somewhere -- take the pivoted elimination throughout.
#ifndef RTD_BC_CAREFUL_K
#define RTD_BC_CAREFUL_K 0.02
#endif
#ifndef RTD_BC_CAREFUL_ALL_MODE0
#define RTD_BC_CAREFUL_ALL_MODE0 1  /* 32 streams (rtd_bc_mfma_kernel): every mode-0 chain with such an eigenvalue, thermal source or
                                       not.  Round 3 measured what it buys -- the near-conservative beam cases go from <= 1.5e-9 to
                                       <= 5e-12 of their 40-digit solutions -- and what it cost with the LDS redo: 219 k -> 71 k col/s on a
                                       batch with a conservative cloud layer in every column; with GjPiv (registers) it is the default.
                                       The tiled 64-stream kernel keeps the thermal-only rule (its pivoted path is the LDS redo). */
#endif
// Which chain (column c, local mode m: index c M + m) a workgroup takes.  Mode 0 of every column first, then the other modes column
// by column: the chains that are pivoted throughout (above: mode 0 only) take 3 ... 7 times as long as the others, and handed out in
// index order the last of them started when three quarters of the launch were over -- a batch with a conservative cloud layer in
// every column spent a third of its boundary-condition kernel waiting for 1/32 of its chains (0.80 against 0.53 ms per 256 cfg4 columns).
// Longest first costs nothing where mode 0 is a chain like the others.
__device__ __forceinline__ long chain_of_block(const long b, const int C, const int M) {
  if (M == 1 || b < C) return b * M;
  const long r = b - C;
  return (r / (M - 1)) * M + 1 + r % (M - 1);
}

__device__ __forceinline__ int chain_needs_pivoting(const RtdDev& d, const bool iso, const double* kk, const int L, const int np) {
  int careful = 0;
  if (iso) {  // (wave-uniform condition: one chain per wavefront.  The scan is lane-parallel -- one memory latency, then a wave
    //           reduction: a scalar loop over the L np eigenvalues cost a mode-0 chain L np dependent loads, 6 % of its life)
    double kmin = 1e300;
    for (int i = (int)threadIdx.x; i < L * np; i += 64) kmin = fmin(kmin, kk[i]);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) kmin = fmin(kmin, __shfl_xor(kmin, o, 64));
    careful = kmin < RTD_BC_CAREFUL_K ? 1 : 0;
  }
  return careful;
}

#ifndef RTD_GJ_GROWTH_TILED
#define RTD_GJ_GROWTH_TILED 1e6
#endif
#ifndef RTD_BCT_WIN
#define RTD_BCT_WIN 24  // layers of small vectors resident in LDS (T = 2: 18 KB next to the 17 KB save area, 4 wavefronts per CU)
#endif
template <int T> struct MatT { v4f64 t[T][T]; };  // t[I][J][q] at lane (kq, col) = element [16 I + 4 q + kq][16 J + col]
template <int T> struct RowT { v4f64 r[T]; };     // vector in row form:    r[I][q] = v[16 I + 4 q + kq], same in every column
template <int T> struct ColT { double c[T]; };    // vector in column form: c[J] = v[16 J + col], same in every lane-row

template <int T>
__device__ __forceinline__ MatT<T> mmT(const MatT<T>& X, const MatT<T>& Y) {  // X^T Y
  MatT<T> R;
#pragma unroll
  for (int I = 0; I < T; ++I)
#pragma unroll
    for (int J = 0; J < T; ++J) {
      v4f64 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int K = 0; K < T; ++K)
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx)
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(X.t[K][I][sidx], Y.t[K][J][sidx], acc, 0, 0, 0);
      R.t[I][J] = acc;
    }
  return R;
}
template <int T>
__device__ __forceinline__ ColT<T> col_dotT(const MatT<T>& X, const RowT<T>& v) {  // X^T v
  ColT<T> o;
#pragma unroll
  for (int J = 0; J < T; ++J) {
    double a = 0.0;
#pragma unroll
    for (int I = 0; I < T; ++I) a += X.t[I][J][0] * v.r[I][0] + X.t[I][J][1] * v.r[I][1] + X.t[I][J][2] * v.r[I][2] + X.t[I][J][3] * v.r[I][3];
    o.c[J] = sum_kq(a);
  }
  return o;
}
template <int T>
__device__ __forceinline__ RowT<T> row_dotT(const MatT<T>& X, const ColT<T>& v) {  // X v
  RowT<T> o;
#pragma unroll
  for (int I = 0; I < T; ++I)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      double a = 0.0;
#pragma unroll
      for (int J = 0; J < T; ++J) a += X.t[I][J][q] * v.c[J];
      o.r[I][q] = row_sum16(a);
    }
  return o;
}
template <int T>
__device__ __forceinline__ RowT<T> col_to_rowT(const ColT<T>& v, const int rowbase, const int kq) {
  RowT<T> o;
#pragma unroll
  for (int I = 0; I < T; ++I) o.r[I] = col_to_row(v.c[I], rowbase, kq);
  return o;
}

template <int T, int K>
struct GjFastT {
  static __device__ __forceinline__ void run(MatT<T>& ta, MatT<T>& tb, ColT<T>& tv, int& bad, const int col) {
    constexpr int KI = K >> 4, K16 = K & 15, QK = K16 >> 2, RK = K16 & 3;
    double x[T], f[T];
#pragma unroll
    for (int J = 0; J < T; ++J) x[J] = bcast_row<RK>(ta.t[KI][J][QK], col);  // row K of Ta^T, replicated over the lane-rows
    const double xk = bcast16<K16>(x[KI]);
    const double r0 = __builtin_amdgcn_rcp(xk);
    const double rp = r0 * (2.0 - xk * r0);
#pragma unroll
    for (int J = 0; J < T; ++J) {
      f[J] = (J == KI && col == K16) ? 1.0 - rp : x[J] * rp;
      bad |= (16 * J + col > K && fabs(f[J]) > RTD_GJ_GROWTH_TILED) ? 1 : 0;
    }
    // v[J] -= f[J] bcast(v[KI]) for every register row v of [Ta^T ; Tb^T ; t^T] that is not finished, the pivot tile column
    // last (it is the source of the others); one v_fmac_f64_dpp each (see GjFast for the hazard notes)
    // (ext-vector elements cannot be bound to asm operands by reference: copy in, update, copy out -- registers all the way)
#define RTD_UPD(VEC, Q, SRC, FJ)                                                                                              \
  {                                                                                                                           \
    double t_ = VEC[Q];                                                                                                       \
    const double s_ = SRC;                                                                                                    \
    asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(t_) : "v"(s_), "v"(FJ), "n"(K16)); \
    VEC[Q] = t_;                                                                                                              \
  }
#define RTD_UPD_SELF(VEC, Q, FJ)                                                                                              \
  {                                                                                                                           \
    double t_ = VEC[Q];                                                                                                       \
    asm volatile("v_fmac_f64_dpp %0, -%0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(t_) : "v"(FJ), "n"(K16));    \
    VEC[Q] = t_;                                                                                                              \
  }
#pragma unroll
    for (int I = KI; I < T; ++I)
#pragma unroll
      for (int q = (I == KI ? QK : 0); q < 4; ++q) {
#pragma unroll
        for (int J = 0; J < T; ++J)
          if (J != KI) RTD_UPD(ta.t[I][J], q, ta.t[I][KI][q], f[J])
        RTD_UPD_SELF(ta.t[I][KI], q, f[KI])
      }
#pragma unroll
    for (int I = 0; I < T; ++I)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int J = 0; J < T; ++J)
          if (J != KI) RTD_UPD(tb.t[I][J], q, tb.t[I][KI][q], f[J])
        RTD_UPD_SELF(tb.t[I][KI], q, f[KI])
      }
#pragma unroll
    for (int J = 0; J < T; ++J)
      if (J != KI) RTD_UPD(tv.c, J, tv.c[KI], f[J])
    {
      double t_ = tv.c[KI];
      asm volatile("v_fmac_f64_dpp %0, -%0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf\n\ts_nop 1" : "+v"(t_) : "v"(f[KI]), "n"(K16));
      tv.c[KI] = t_;
    }
#undef RTD_UPD
#undef RTD_UPD_SELF
    if constexpr (K + 1 < 16 * T) GjFastT<T, K + 1>::run(ta, tb, tv, bad, col);
  }
};

