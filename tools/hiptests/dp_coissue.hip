// Microbenchmark: do FP64 vector FMAs and FP64 MFMAs of DIFFERENT wavefronts on one SIMD overlap on gfx950, or do they
// share the DP ALUs?  One workgroup of 512 threads = 8 wavefronts = 2 per SIMD of one CU; wave roles by wave id:
//   mode 0: all 8 wavefronts issue v_fma_f64 (8 independent chains each)
//   mode 1: all 8 issue v_mfma_f64_16x16x4_f64 (4 independent accumulators each)
//   mode 2: wavefronts 0-3 (one per SIMD) issue FMAs, 4-7 (their SIMD partners) MFMAs -- the same amount of each as one
//           wavefront does in modes 0 / 1.  If the pipes are separate the time is ~ max of the two, if shared ~ their sum.
// Build: hipcc --offload-arch=gfx950 -O3 -o dp_coissue.bin dp_coissue.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void k(double* out, long long* cyc, int n, int mode) {
  const int wave = threadIdx.x >> 6;
  const bool do_fma = mode == 0 || (mode == 2 && wave < 4), do_mfma = mode == 1 || (mode == 2 && wave >= 4);
  double x[8];
  for (int i = 0; i < 8; ++i) x[i] = 1.0 + threadIdx.x * 1e-6 + i;
  v4f64 acc[4];
  for (int c = 0; c < 4; ++c) acc[c] = {0.0, 0.0, 0.0, 0.0};
  const double a = 1.0 - 1e-9, b = 1e-9 * threadIdx.x;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  if (do_fma) {
    for (int it = 0; it < n; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = __builtin_fma(x[i], a, b);   // 32 FMAs per trip
    }
  }
  if (do_mfma) {
    for (int it = 0; it < n; ++it) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);  // 8 MFMAs per trip
    }
  }
  double s = 0.0;
  for (int i = 0; i < 8; ++i) s += x[i];
  for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[threadIdx.x] = s;
  const long long t1 = __builtin_amdgcn_s_memtime();
  __syncthreads();
  if ((threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
}
int main() {
  double* out;
  long long* cyc;
  hipMalloc(&out, 512 * sizeof(double));
  hipMalloc(&cyc, 8 * sizeof(long long));
  const int n = 2000;
  long long h[8];
  const char* names[3] = {"8 wavefronts x 32 v_fma_f64 per trip", "8 wavefronts x 8 v_mfma_f64_16x16x4 per trip",
                          "4 wavefronts FMA + their 4 SIMD partners MFMA"};
  for (int mode = 0; mode < 3; ++mode)
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, out, cyc, n, mode);
      hipDeviceSynchronize();
      hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
      if (rep)
        printf("%-50s cycles per trip: wave0 %.1f wave3 %.1f wave4 %.1f wave7 %.1f\n", names[mode], (double)h[0] / n, (double)h[3] / n,
               (double)h[4] / n, (double)h[7] / n);
    }
  printf("(one wavefront alone: 32 FMAs = 128 issue cycles, 8 MFMAs = 8 x 64 = 512 cycles of the matrix pipe;\n"
         " two FMA wavefronts per SIMD: 256 per trip each; two MFMA wavefronts per SIMD: 1024 per trip each;\n"
         " mode 2: 128 / 512 if the pipes overlap, ~640 for the slower one if they share the DP ALUs)\n");
  return 0;
}
