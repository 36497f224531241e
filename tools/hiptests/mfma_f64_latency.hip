// Microbenchmark: issue interval and dependent-chain latency of v_mfma_f64_16x16x4_f64 on gfx950 (one wavefront).
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_f64_latency.bin mfma_f64_latency.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));

template <int CHAINS>
__global__ void k(double* out, long long* cyc, int n) {
  v4f64 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) acc[c] = {0.0, 0.0, 0.0, 0.0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
  }
  double s = 0.0;
  for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[threadIdx.x] = s;
  asm volatile("s_waitcnt vmcnt(0)");
  long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
// MFMA -> dependent VALU -> MFMA (the pattern of the carry recursion: products, scale, products)
__global__ void kv(double* out, long long* cyc, int n) {
  v4f64 acc = {0.0, 0.0, 0.0, 0.0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; ++i) {
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    a = acc[0] * 0.5;  // VALU read of the MFMA result, feeding the next MFMA's operand
  }
  out[threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3] + a;
  asm volatile("s_waitcnt vmcnt(0)");
  long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  double* out;
  long long* cyc;
  hipMalloc(&out, 64 * sizeof(double));
  hipMalloc(&cyc, sizeof(long long));
  const int n = 1000;
  long long h;
#define RUN(KERNEL, LABEL, PER)                                                          \
  for (int rep = 0; rep < 2; ++rep) {                                                    \
    hipLaunchKernelGGL(KERNEL, dim3(1), dim3(64), 0, 0, out, cyc, n);                    \
    hipDeviceSynchronize();                                                              \
    hipMemcpy(&h, cyc, sizeof(h), hipMemcpyDeviceToHost);                                \
    if (rep) printf("%-34s %8.1f s_memtime ticks per MFMA\n", LABEL, (double)h / (n * PER)); \
  }
  RUN(k<1>, "1 dependent chain", 1)
  RUN(k<2>, "2 interleaved chains", 2)
  RUN(k<4>, "4 interleaved chains", 4)
  RUN(k<8>, "8 interleaved chains", 8)
  RUN(kv, "MFMA -> VALU -> MFMA operand", 1)
  return 0;
}
