// Diagnostic, stand-alone (hipcc --offload-arch=gfx950 tools/hw/pk_hazard.hip -o tools/hw/pk_hazard): does a packed-fp32 add whose
// destination pair is also its second source, with the halves swapped by op_sel -- the instruction the compiler emitted for the one
// accumulator that was not reproducible in kmix_bwd<MODE 2> -- ever return a wrong value while a kernel that keeps
// MFMA accumulators in AGPRs is resident on the same SIMDs?  Integer-valued floats: every sum is exact, so any mismatch is an error.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// kernel under test: NV = number of VGPRs to force (the failing build had 223)
template <int HIGHV>
__global__ __launch_bounds__(256, 1) void pk_kernel(unsigned* bad, unsigned* checks, int iters) {
  const int lane = threadIdx.x & 63;
  if (HIGHV) asm volatile("v_mov_b32 v222, 0" ::: "v222");
  f2 acc = {0.f, 0.f};                 // (ag[2], ag[3]) of the original
  float e0 = 0.f, e1 = 0.f;            // the same recurrence with scalar adds
  unsigned nbad = 0;
  for (int it = 0; it < iters; ++it) {
    f2 x = {(float)((lane + it) & 7), (float)((lane * 3 + it) & 7)};
    f2 y = {(float)((it >> 1) & 3), (float)((lane + 2 * it) & 3)};
    f2 t;
    // t = x * y ; acc' = (acc.lo + t.hi, acc.hi + t.lo) written over t ; acc = t
    asm volatile(
        "v_pk_mul_f32 %0, %2, %3\n\t"
        "s_nop 0\n\t"
        "v_pk_add_f32 %0, %1, %0 op_sel:[0,1] op_sel_hi:[1,0]\n\t"
        : "=&v"(t) : "v"(acc), "v"(x), "v"(y));
    acc = t;
    const float p0 = x.x * y.x, p1 = x.y * y.y;
    e0 += p1; e1 += p0;
    if (e0 > 4.0e6f) { acc = f2{0.f, 0.f}; e0 = e1 = 0.f; }   // stay exact
    if (acc.x != e0 || acc.y != e1) { ++nbad; acc = f2{e0, e1}; }
  }
  if (nbad) atomicAdd(bad, nbad);
  if (threadIdx.x == 0) atomicAdd(checks, 1u);
}

// neighbour: MFMA accumulators in AGPRs, ~200 VGPRs of allocation, runs until told to stop (or for `iters`)
__global__ __launch_bounds__(256, 1) void agpr_kernel(float* out, int iters, int use_agpr) {
  asm volatile("v_mov_b32 v189, 0" ::: "v189");
  float a = (float)(threadIdx.x & 3), b = 1.0f;
  f16v c = {0};
  if (use_agpr) {
    for (int it = 0; it < iters; ++it) {
      asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
      asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(c) : "v"(b), "v"(a));
    }
  } else {
    for (int it = 0; it < iters; ++it) {
      asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
      asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(c) : "v"(b), "v"(a));
    }
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += c[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 20;
  unsigned *bad, *checks; float* out;
  CK(hipMalloc(&bad, 4)); CK(hipMalloc(&checks, 4)); CK(hipMalloc(&out, 4 * 1024 * 256));
  hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
  for (int mode = 0; mode < 3; ++mode) {       // 0: alone, 1: beside the AGPR kernel, 2: beside the same kernel with VGPR accumulators
    for (int highv = 0; highv < 2; ++highv) {
      CK(hipMemset(bad, 0, 4)); CK(hipMemset(checks, 0, 4));
      for (int r = 0; r < reps; ++r) {
        if (mode) hipLaunchKernelGGL(agpr_kernel, dim3(512), dim3(256), 0, s1, out, 40000, mode == 1);
        for (int k = 0; k < 8; ++k) {
          if (highv) hipLaunchKernelGGL(pk_kernel<1>, dim3(512), dim3(256), 0, s2, bad, checks, 4000);
          else hipLaunchKernelGGL(pk_kernel<0>, dim3(512), dim3(256), 0, s2, bad, checks, 4000);
        }
        CK(hipDeviceSynchronize());
      }
      unsigned hb, hc; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hc, checks, 4, hipMemcpyDeviceToHost));
      printf("neighbour %-22s  %s-VGPR build: %u wrong results in %u workgroup runs x 256 lanes x 4000 iterations\n",
             mode == 0 ? "none" : mode == 1 ? "MFMA acc in AGPRs" : "MFMA acc in VGPRs", highv ? "223" : "low", hb, hc);
    }
  }
  return 0;
}
