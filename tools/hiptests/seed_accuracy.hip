// Accuracy of the v_rsq_f64 / v_rcp_f64 hardware seeds on gfx950 (how many Newton steps fast_rsqrt/fast_rcp need).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
__global__ void k(const double* x, double* r0, double* r1, double* r2, double* c0, double* c1, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v = x[i];
  double y = __builtin_amdgcn_rsq(v);
  r0[i] = y;
  y = y * (1.5 - 0.5 * v * y * y);
  r1[i] = y;
  y = y * (1.5 - 0.5 * v * y * y);
  r2[i] = y;
  double z = __builtin_amdgcn_rcp(v);
  c0[i] = z;
  z = z * (2.0 - v * z);
  c1[i] = z;
}
int main() {
  const int n = 1 << 16;
  double *x, *r0, *r1, *r2, *c0, *c1;
  hipMallocManaged(&x, n * 8); hipMallocManaged(&r0, n * 8); hipMallocManaged(&r1, n * 8);
  hipMallocManaged(&r2, n * 8); hipMallocManaged(&c0, n * 8); hipMallocManaged(&c1, n * 8);
  for (int i = 0; i < n; ++i) x[i] = std::exp(-20.0 + 40.0 * i / n) * (1.0 + 0.37 * std::sin(i * 1.7));
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, x, r0, r1, r2, c0, c1, n);
  hipDeviceSynchronize();
  double e[5] = {0, 0, 0, 0, 0};
  for (int i = 0; i < n; ++i) {
    const double tr = 1.0 / std::sqrt(x[i]), tc = 1.0 / x[i];
    e[0] = std::fmax(e[0], std::fabs(r0[i] / tr - 1)); e[1] = std::fmax(e[1], std::fabs(r1[i] / tr - 1));
    e[2] = std::fmax(e[2], std::fabs(r2[i] / tr - 1)); e[3] = std::fmax(e[3], std::fabs(c0[i] / tc - 1));
    e[4] = std::fmax(e[4], std::fabs(c1[i] / tc - 1));
  }
  printf("rsq seed %.3e, +1 Newton %.3e, +2 Newton %.3e | rcp seed %.3e, +1 Newton %.3e\n", e[0], e[1], e[2], e[3], e[4]);
  return 0;
}
