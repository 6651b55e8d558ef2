// lane-layout probe of the 16-block 4x4 MFMAs on gfx950: hipcc --offload-arch=gfx950 tools/mfma_probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
__global__ void k1(int la, int lb, float* out) {   // 4x4x1 f32: a one-hot at lane la, b one-hot at lane lb
  const int lane = threadIdx.x;
  f32x4 d = {0.f, 0.f, 0.f, 0.f};
  d = __builtin_amdgcn_mfma_f32_4x4x1f32(la < 0 ? 1.f : (lane == la ? 1.f : 0.f), lb < 0 ? 1.f : (lane == lb ? 1.f : 0.f), d, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[lane * 4 + r] = d[r];
}
__global__ void k4(int la, int ka, int lb, int kb, float* out) {   // 4x4x4 bf16: a one-hot at (lane la, k ka) ...
  const int lane = threadIdx.x;
  bf16x4 a, b;
  for (int q = 0; q < 4; ++q) {
    a[q] = (__bf16)((la < 0 || (lane == la && q == ka)) ? 1.f : 0.f);
    b[q] = (__bf16)((lb < 0 || (lane == lb && q == kb)) ? 1.f : 0.f);
  }
  f32x4 d = {0.f, 0.f, 0.f, 0.f};
  d = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), d, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[lane * 4 + r] = d[r];
}
static void show(const char* tag, float* h) {
  printf("%s:", tag);
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) if (h[l * 4 + r] != 0.f) printf(" (lane %d reg %d = %g)", l, r, h[l * 4 + r]);
  printf("\n");
}
int main() {
  float* d; float h[256];
  hipMalloc(&d, sizeof h);
  const int probes[] = {0, 1, 5, 22, 37, 63};
  for (int p : probes) {
    char t[64];
    hipLaunchKernelGGL(k1, dim3(1), dim3(64), 0, 0, p, -1, d); hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    snprintf(t, sizeof t, "f32 A one-hot lane %d, B ones", p); show(t, h);
    hipLaunchKernelGGL(k1, dim3(1), dim3(64), 0, 0, -1, p, d); hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    snprintf(t, sizeof t, "f32 B one-hot lane %d, A ones", p); show(t, h);
  }
  for (int p : {5, 37}) for (int kk : {0, 3}) {
    char t[64];
    hipLaunchKernelGGL(k4, dim3(1), dim3(64), 0, 0, p, kk, -1, 0, d); hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    snprintf(t, sizeof t, "bf16 A one-hot lane %d k %d, B ones", p, kk); show(t, h);
    hipLaunchKernelGGL(k4, dim3(1), dim3(64), 0, 0, -1, 0, p, kk, d); hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    snprintf(t, sizeof t, "bf16 B one-hot lane %d k %d, A ones", p, kk); show(t, h);
    hipLaunchKernelGGL(k4, dim3(1), dim3(64), 0, 0, p, kk, p, (kk + 1) % 4, d); hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    snprintf(t, sizeof t, "bf16 A(lane %d,k %d) x B(same lane,k %d)", p, kk, (kk + 1) % 4); show(t, h);
  }
  return 0;
}
