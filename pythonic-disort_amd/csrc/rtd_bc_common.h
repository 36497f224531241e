// rtd_bc_common.h -- helpers shared by the boundary-condition kernels (rtd_bc.hip, rtd_bc_small.hip): cross-lane moves, the
// workspace layout of an interface and the row-per-lane Gauss-Jordan step.  Included inside each file's anonymous namespace.
#pragma once

#define RTD_FENCE() asm volatile("" ::: "memory")
#ifndef RTD_GJ_BATCH
#define RTD_GJ_BATCH 3  /* cross-lane fetches in flight per batch - 1 (power of two minus one) */
#endif
#ifndef RTD_SWEEP_WAVES
#define RTD_SWEEP_WAVES 3
#endif

template <int MASK>
__device__ __forceinline__ double xor_lane(double v) {
  if constexpr (MASK >= 32) return __shfl_xor(v, MASK, 64);  // across the halves of the wavefront: ds_bpermute (128 streams only)
  constexpr int pat = (MASK << 10) | 0x1F;
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_ds_swizzle(lo, pat);
  hi = __builtin_amdgcn_ds_swizzle(hi, pat);
  return __hiloint2double(hi, lo);
}

template <int NP>
__device__ __forceinline__ double group_max(double v) {
  if (NP > 1) v = fmax(v, xor_lane<1>(v));
  if (NP > 2) v = fmax(v, xor_lane<2>(v));
  if (NP > 4) v = fmax(v, xor_lane<4>(v));
  if (NP > 8) v = fmax(v, xor_lane<8>(v));
  if (NP > 16) v = fmax(v, xor_lane<16>(v));
  if (NP > 32) v = fmax(v, xor_lane<32>(v));
  return v;
}


// max over the NP lanes of a group for non-negative f32 keys, with DPP row operations (no LDS crossbar):
// xor-1 and xor-2 quad permutes, row_half_mirror, row_mirror; one v_max_f32 each.
template <int CTRL>
__device__ __forceinline__ float dpp_max_f32(float v) {
  const int o = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true);
  return fmaxf(v, __int_as_float(o));
}
template <int NP>
__device__ __forceinline__ float group_max_key(float v) {
  v = dpp_max_f32<0xB1>(v);                // quad_perm [1,0,3,2]
  v = dpp_max_f32<0x4E>(v);                // quad_perm [2,3,0,1]
  if (NP > 4) v = dpp_max_f32<0x141>(v);   // row_half_mirror
  if (NP > 8) v = dpp_max_f32<0x140>(v);   // row_mirror
  if (NP > 16) {
    const int o = __builtin_amdgcn_ds_swizzle(__float_as_int(v), (16 << 10) | 0x1F);
    v = fmaxf(v, __int_as_float(o));
  }
  if (NP > 32) v = fmaxf(v, __shfl_xor(v, 32, 64));
  return v;
}

__device__ __forceinline__ double fast_rcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = y * (2.0 - x * y);
  y = y * (2.0 - x * y);
  return y;
}

// value of `v` in lane `addr/4` (addr precomputed once per pivot step)
__device__ __forceinline__ double bperm(int addr, double v) {
  const int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(v));
  const int hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(v));
  return __hiloint2double(hi, lo);
}

// compile-time loop: f(std::integral_constant<int, I>) for I in [B, E)
template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

// value of lane K (compile-time) of this lane's 16-lane group: DPP row broadcast, VALU only
template <int K>
__device__ __forceinline__ double bcast16(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + K, 0xF, 0xF, true);  // row_newbcast:K; bound_ctrl + full masks: no `old` operand, no copy
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + K, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

// value of lane K (compile-time) of this lane's 8-lane group (two groups per 16-lane DPP row): the row broadcast of lane K for the
// banks of the lower half, of lane 8 + K for the upper half (bank_mask; a masked lane keeps `old`) -- two VALU moves per dword
template <int K>
__device__ __forceinline__ int bcast8_i(int v) {
  int r = __builtin_amdgcn_update_dpp(0, v, 0x150 + K, 0xF, 0x3, false);
  return __builtin_amdgcn_update_dpp(r, v, 0x150 + 8 + K, 0xF, 0xC, false);
}
template <int K>
__device__ __forceinline__ double bcast8(double v) {
  return __hiloint2double(bcast8_i<K>(__double2hiint(v)), bcast8_i<K>(__double2loint(v)));
}
// lane K of the caller's NP-lane group for NP = 8 or 16
template <int NP, int K>
__device__ __forceinline__ double bcast_grp(double v) {
  if constexpr (NP == 16) return bcast16<K>(v);
  else return bcast8<K>(v);
}

// v[c] -= f * (lane K of the caller's 8-lane group's v[c]) for c in [C0, C1), ONE half of a 16-lane DPP row per instruction: the DP
// ALU's DPP form is the row broadcast, so lane K of the row serves the banks of its lower eight lanes (HI = false) and lane 8 + K
// the upper eight (HI = true) -- two v_fmac_f64_dpp per register against four moves and an FMA.  The caller issues all first
// halves, then all second halves: a DPP read needs two wait states after a VALU write of the same register, and the compiler's
// hazard recogniser does not see writes made inside inline asm (build.py scans the generated ISA, tools/check_dpp_hazards.py).
template <int K, bool HI, int C0, int C1, int N>
struct FmacBank8 {
  static __device__ __forceinline__ void run(double (&v)[N > 0 ? N : 1], const double f) {
    if constexpr (C0 < C1) {
      if constexpr (HI)
        asm volatile("v_fmac_f64_dpp %0, -%0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xc" : "+v"(v[C0]) : "v"(f), "n"(8 + K));
      else
        asm volatile("v_fmac_f64_dpp %0, -%0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0x3" : "+v"(v[C0]) : "v"(f), "n"(K));
      FmacBank8<K, HI, C0 + 1, C1, N>::run(v, f);
    }
  }
};

// Workspace layout per (c, m, l), l < L-1, inside d.Fws (4 NP^2 doubles per slot):
//   [0, NP^2) Wp   [NP^2, 2NP^2) Wq   [2NP^2, 3NP^2) S   then rho_t, rho_b, s (NP each)
template <int NP>
struct Ws {
  static constexpr long SLOT = 4L * NP * NP;
  static constexpr int WP = 0, WQ = NP * NP, S = 2 * NP * NP, RT = 3 * NP * NP, RB = RT + NP, SV = RB + NP;
};

typedef double v4f64 __attribute__((ext_vector_type(4)));


// Gauss-Jordan with partial pivoting on [A | B | b] (NP rows, one per lane of the group): on exit the lane
// that owned pivot column `pc` holds row pc of A^-1 B in bm[] and (A^-1 b)[pc] in bv.
template <int NP, int NB, int K>
struct GjStep {
  static __device__ __forceinline__ void run(double (&am)[NP], double (&bm)[NB], double& bv, int& pc, const int grp) {
    // pivot search on f32 keys (a pivot within 2^-24 of the largest candidate is as good as the largest)
    const float key = (pc < 0) ? fabsf((float)am[K]) : -1.0f;
    const float kmax = group_max_key<NP>(key);
    const int j = (int)(threadIdx.x % NP);
    double f, rp;
    bool isp;
    bool fast = false;
    if constexpr (NP == 16 || NP == 8) {
      // threshold pivoting: when the diagonal candidate (lane K, still unused) is within a factor 4 of the largest
      // candidate of its group, it is taken as the pivot: the source lane is then a compile-time
      // constant and the pivot row travels by DPP row broadcasts (VALU) instead of ds_bpermute (LDS crossbar, the
      // pipe that bounds this kernel).  ~95 % of the steps of real atmospheres qualify; the others take the fully
      // pivoted path below.  Growth is bounded as in partial pivoting with threshold 1/4.
      int kd;
      if constexpr (NP == 16) kd = __builtin_amdgcn_update_dpp(0, __float_as_int(key), 0x150 + K, 0xF, 0xF, true);
      else kd = bcast8_i<K>(__float_as_int(key));  // (NP = 8: two groups per DPP row, see bcast8)
      // (decided per group, not per wavefront: a chain's arithmetic must not depend on which chains share its wavefront -- a
      //  windowed plan groups them differently and has to return the same bits; a wavefront whose groups all qualify skips
      //  the pivoted branch altogether)
      fast = __int_as_float(kd) >= 0.25f * kmax && __int_as_float(kd) > 0.0f;
    }
    if (fast) {
      isp = (j == K);
      const double piv = bcast_grp<NP, K>(am[K]);
      rp = fast_rcp(piv);
      f = isp ? 0.0 : am[K] * rp;
      if constexpr (NP == 8) {
        // (see FmacBank8: two bank-masked v_fmac_f64_dpp per register; all first halves, then all second halves)
        FmacBank8<K, false, K + 1, NP, NP>::run(am, f);
        FmacBank8<K, false, 0, NB, NB>::run(bm, f);
        asm volatile("v_fmac_f64_dpp %0, -%0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0x3\n\ts_nop 1" : "+v"(bv) : "v"(f), "n"(K));
        FmacBank8<K, true, K + 1, NP, NP>::run(am, f);
        FmacBank8<K, true, 0, NB, NB>::run(bm, f);
        asm volatile("v_fmac_f64_dpp %0, -%0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xc\n\ts_nop 1" : "+v"(bv) : "v"(f), "n"(8 + K));
      } else {
      static_for<K + 1, NP>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        am[c] -= f * bcast_grp<NP, K>(am[c]);
      });
      static_for<0, NB>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        bm[c] -= f * bcast_grp<NP, K>(bm[c]);
      });
      bv -= f * bcast_grp<NP, K>(bv);
      }
    } else {
      const unsigned long long bal = __ballot(key == kmax);
      const unsigned long long bits = NP == 64 ? bal : (bal >> (grp * NP)) & ((1ull << (NP & 63)) - 1);
      const int src = __ffsll((long long)bits) - 1;  // pivot lane of this group
      isp = (j == src);
      const int addr = (grp * NP + src) << 2;
      const double piv = bperm(addr, am[K]);
      rp = fast_rcp(piv);
      f = isp ? 0.0 : am[K] * rp;
      // (scheduling barriers bound the number of cross-lane results in flight: register pressure)
#pragma unroll
      for (int c = K + 1; c < NP; ++c) {
        am[c] -= f * bperm(addr, am[c]);
        if ((c & RTD_GJ_BATCH) == RTD_GJ_BATCH) __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int c = 0; c < NB; ++c) {
        bm[c] -= f * bperm(addr, bm[c]);
        if ((c & RTD_GJ_BATCH) == RTD_GJ_BATCH) __builtin_amdgcn_sched_barrier(0);
      }
      bv -= f * bperm(addr, bv);
    }
    if (isp) {  // normalise the pivot row now: later steps leave it untouched in column K
      pc = K;
#pragma unroll
      for (int c = K + 1; c < NP; ++c) am[c] *= rp;
#pragma unroll
      for (int c = 0; c < NB; ++c) bm[c] *= rp;
      bv *= rp;
    }
    GjStep<NP, NB, K + 1>::run(am, bm, bv, pc, grp);
  }
};
template <int NP, int NB>
struct GjStep<NP, NB, NP> {
  static __device__ __forceinline__ void run(double (&)[NP], double (&)[NB], double&, int&, const int) {}
};

