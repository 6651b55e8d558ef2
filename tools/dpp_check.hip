// standalone check of the DPP wave reductions in csrc/common.h against a double-precision host sum
// build: hipcc -O3 --offload-arch=gfx950 -I mimrl_amd/csrc tools/dpp_check.hip -o gpurun_out/dpp_check ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "common.h"
using namespace mimrl;
__global__ void k(const float* x, float* o_dpp, float* o_shfl, float* o_max, int n, long R) {
  const int lane = threadIdx.x & 63;
  const long wid = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6, nwv = ((long)gridDim.x * blockDim.x) >> 6;
  for (long r = wid; r < R; r += nwv) {
    float s = 0.f, m = -INFINITY;
    for (int j = lane; j < n; j += 64) { s += x[r * n + j]; m = fmaxf(m, x[r * n + j]); }
    const float a = wave_sum(s), b = wave_sum_shfl(s), c = wave_max(m);
    if (lane == (int)(r % 64)) { o_dpp[r] = a; o_shfl[r] = b; o_max[r] = c; }
  }
}
int main() {
  const int n = 128; const long R = 5000;
  std::vector<float> h(R * n);
  unsigned s = 12345u;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.f - 0.5f; }
  float *x, *a, *b, *c;
  hipMalloc(&x, h.size() * 4); hipMalloc(&a, R * 4); hipMalloc(&b, R * 4); hipMalloc(&c, R * 4);
  hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(40), dim3(256), 0, 0, x, a, b, c, n, R);
  std::vector<float> ha(R), hb(R), hc(R);
  hipMemcpy(ha.data(), a, R * 4, hipMemcpyDeviceToHost); hipMemcpy(hb.data(), b, R * 4, hipMemcpyDeviceToHost); hipMemcpy(hc.data(), c, R * 4, hipMemcpyDeviceToHost);
  double ea = 0, eb = 0, ec = 0;
  for (long r = 0; r < R; ++r) {
    double t = 0, m = -1e30;
    for (int j = 0; j < n; ++j) { t += h[r * n + j]; m = std::fmax(m, (double)h[r * n + j]); }
    ea = std::fmax(ea, std::fabs(ha[r] - t)); eb = std::fmax(eb, std::fabs(hb[r] - t)); ec = std::fmax(ec, std::fabs(hc[r] - m));
  }
  printf("max |err| dpp %.3e  shfl %.3e  max %.3e\n", ea, eb, ec);
  return 0;
}
