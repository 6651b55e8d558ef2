#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void ks(float* out) {
  const int lane = threadIdx.x;
  float x = lane, y = 100 + lane;
  auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, y), false, false);
  out[lane * 2] = __builtin_bit_cast(float, r[0]); out[lane * 2 + 1] = __builtin_bit_cast(float, r[1]);
}
int main() {
  float* d; float h[128];
  hipMalloc(&d, sizeof h);
  hipLaunchKernelGGL(ks, dim3(1), dim3(64), 0, 0, d); hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  for (int l : {0, 1, 31, 32, 33, 63}) printf("lane %d: r0 = %g r1 = %g\n", l, h[2 * l], h[2 * l + 1]);
  return 0;
}
