// Semantics check of the gfx950 v_permlane16_swap / v_permlane32_swap builtins (used for cross-row moves).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* out) {
  const unsigned lane = threadIdx.x;
  u2 r16 = __builtin_amdgcn_permlane16_swap(lane, lane + 100, false, false);
  u2 r32 = __builtin_amdgcn_permlane32_swap(lane, lane + 100, false, false);
  out[lane] = r16[0];
  out[64 + lane] = r16[1];
  out[128 + lane] = r32[0];
  out[192 + lane] = r32[1];
}
int main() {
  unsigned* d;
  hipMalloc(&d, 256 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  unsigned h[256];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[4] = {"p16.vdst", "p16.src ", "p32.vdst", "p32.src "};
  for (int a = 0; a < 4; ++a) {
    printf("%s:", names[a]);
    for (int i = 0; i < 64; i += 8) printf(" [%d]=%u", i, h[a * 64 + i]);
    printf("\n");
  }
  return 0;
}
