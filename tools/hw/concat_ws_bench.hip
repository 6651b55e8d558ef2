// Stand-alone timing + cross-check of the concat-critic kernels: weights-stationary (concat_ws.hip) against weight-streaming (concat_fused.hip).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I mimrl_amd/csrc -I include tools/hw/concat_ws_bench.hip mimrl_amd/csrc/concat_fused.hip mimrl_amd/csrc/concat_ws.hip mimrl_amd/csrc/errors.cpp mimrl_amd/csrc/knobs.cpp -o tools/hw/concat_ws_bench
//   tools/hw/concat_ws_bench [B=256] [E=5] [reps=20]
#include "../../mimrl_amd/csrc/concat_fused.h"
#include "../../mimrl_amd/csrc/knobs.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

using namespace mimrl;
#ifdef WS_PHASE
namespace mimrl { int concat_ws_read_phases(long long* out); }
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <class T> static T* dmalloc(size_t n) { T* p = nullptr; if (hipMalloc(&p, n * sizeof(T)) != hipSuccess) { printf("alloc of %zu failed\n", n * sizeof(T)); exit(1); } hipMemset(p, 0, n * sizeof(T)); return p; }
static float* dev_rand(size_t n, float scale, std::mt19937& g, float shift = 0.f) {
  std::vector<float> h(n); std::normal_distribution<float> d(0.f, scale);
  for (auto& v : h) v = d(g) + shift;
  float* p = dmalloc<float>(n); hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice); return p;
}
static __bf16* to_bf(const float* src, size_t n) {
  std::vector<float> h(n); hipMemcpy(h.data(), src, n * 4, hipMemcpyDeviceToHost);
  std::vector<__bf16> b(n); for (size_t i = 0; i < n; ++i) b[i] = (__bf16)h[i];
  __bf16* p = dmalloc<__bf16>(n); hipMemcpy(p, b.data(), n * 2, hipMemcpyHostToDevice); return p;
}
template <class T> static std::vector<T> grab(const T* p, size_t n) { std::vector<T> h(n); hipMemcpy(h.data(), p, n * sizeof(T), hipMemcpyDeviceToHost); return h; }

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 256, E = argc > 2 ? atoi(argv[2]) : 5, reps = argc > 3 ? atoi(argv[3]) : 20;
  std::mt19937 g(3);
  const long rows = (long)B * B, pstride = 3L * 256 * 256 + 4 * 256;   // [W1 | W2 | W3-unused | b1 b2 w3 b3] per estimator
  float* params = dev_rand((size_t)E * pstride, 0.06f, g);
  __bf16* img = to_bf(params, (size_t)E * pstride);
  ConcatFwdArgs a{};
  a.P = dev_rand((size_t)E * B * 256, 0.7f, g); a.Q = dev_rand((size_t)E * B * 256, 0.7f, g);
  a.W1 = img; a.W2 = img + 65536; a.b1 = params + 3 * 65536; a.b2 = a.b1 + 256; a.w3 = a.b1 + 512; a.b3 = a.b1 + 768; a.pstride = pstride;
  a.E = E; a.B = B;
  float* scores[3] = {dmalloc<float>(E * rows), dmalloc<float>(E * rows), dmalloc<float>(E * rows)};
  uint32_t* masks[3][3]; for (auto& m : masks) for (auto& p : m) p = dmalloc<uint32_t>(E * rows * 8);
  __bf16* a0b[3] = {dmalloc<__bf16>(E * rows * 256), dmalloc<__bf16>(E * rows * 256), dmalloc<__bf16>(E * rows * 256)};
  __bf16* a1b[3] = {dmalloc<__bf16>(E * rows * 256), dmalloc<__bf16>(E * rows * 256), dmalloc<__bf16>(E * rows * 256)};
  float* a2[3] = {dmalloc<float>(E * rows * 256), dmalloc<float>(E * rows * 256), dmalloc<float>(E * rows * 256)};
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int save : {3, 2, 0}) {
    float us[3] = {0.f, 0.f, 0.f};
    for (int k = 0; k < 3; ++k) {                      // k = 0: streaming (rounds 2-5), 1: weights-stationary 8 waves, 2: weights-stationary 4 waves
      ConcatFwdArgs x = a; x.save = save; x.scores = scores[k]; x.m0 = masks[k][0]; x.m1 = masks[k][1]; x.m2 = masks[k][2];
      x.a0b = a0b[k]; x.a1b = a1b[k]; x.a2 = a2[k];
      if (k == 0) setenv("MIMRL_CONCAT_STREAMED", "1", 1); else unsetenv("MIMRL_CONCAT_STREAMED");
      auto run = [&]() { return k == 2 ? concat_fwd_ws4(s, x) : concat_fwd_fused(s, x); };
      for (int w = 0; w < 3; ++w) if (run() != 0) { printf("launch failed: %s\n", mimrl::last_error_slot().c_str()); return 1; }
      CK(hipStreamSynchronize(s));
      CK(hipEventRecord(e0, s));
      for (int r = 0; r < reps; ++r) run();
      CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
      float ms = 0.f; CK(hipEventElapsedTime(&ms, e0, e1)); us[k] = 1e3f * ms / reps;
#ifdef WS_PHASE
      if (k == 1) {
        long long ph[18]; concat_ws_read_phases(ph);
        printf("   shader clock during the launch: %.0f MHz\n", ph[17] ? 100.0 * ph[16] / ph[17] : 0.0);
        const char* nm[7] = {"load-issue", "product", "gen", "epilogue", "copy-out", "scores", "barrier"};
        for (int w = 0; w < 2; ++w) { printf("   save %d wave %d (layer %d), us per launch:", save, 4 * w, w + 1); double tot = 0; for (int i = 0; i < 7; ++i) { printf(" %s %.1f", nm[i], ph[w * 8 + i] / 100.0 / (reps + 3)); tot += ph[w * 8 + i] / 100.0 / (reps + 3); } printf("  | sum %.1f, steps %.0f\n", tot, (double)ph[w * 8 + 7] / (reps + 3)); }
      } else { long long ph[18]; concat_ws_read_phases(ph); }
#endif
    }
    // cross-check of both weights-stationary kernels against the streaming one
    const double gflop = 2.0 * 2 * E * rows * 256.0 * 256.0 / 1e9;
    printf("save %d  B %d E %d: streaming %8.1f us (%.0f TFLOP/s)   8-wave %8.1f us (%.0f)   4-wave %8.1f us (%.0f)\n", save, B, E, us[0], gflop / us[0] * 1e3,
           us[1], gflop / us[1] * 1e3, us[2], gflop / us[2] * 1e3);
    auto s0 = grab(scores[0], E * rows);
    for (int k = 1; k < 3; ++k) {
      auto s1 = grab(scores[k], E * rows);
      double dmax = 0, smax = 0; for (size_t i = 0; i < s0.size(); ++i) { dmax = std::max(dmax, (double)std::fabs(s0[i] - s1[i])); smax = std::max(smax, (double)std::fabs(s0[i])); }
      long mdiff[3] = {0, 0, 0};
      if (save >= 2) for (int l = 0; l < 3; ++l) { auto m0 = grab(masks[0][l], E * rows * 8), m1 = grab(masks[k][l], E * rows * 8); for (size_t i = 0; i < m0.size(); ++i) mdiff[l] += __builtin_popcount(m0[i] ^ m1[i]); }
      double adiff = 0; long nb = 0, nz = 0;
      if (save == 2) {
        auto x0 = grab(a2[0], E * rows * 256), x1 = grab(a2[k], E * rows * 256); for (size_t i = 0; i < x0.size(); i += 7) adiff = std::max(adiff, (double)std::fabs(x0[i] - x1[i]));
        auto y0 = grab(a1b[0], E * rows * 256), y1 = grab(a1b[k], E * rows * 256); for (size_t i = 0; i < y0.size(); ++i) nb += (float)y0[i] != (float)y1[i];
        auto z0 = grab(a0b[0], E * rows * 256), z1 = grab(a0b[k], E * rows * 256); for (size_t i = 0; i < z0.size(); ++i) nz += (float)z0[i] != (float)z1[i];
      }
      printf("   %d-wave vs streaming: max |score diff| %.3e (max |score| %.2f)  sign-bit flips m0 %ld m1 %ld m2 %ld  a2 diff %.2e  a1b / a0b entries differing %ld / %ld\n",
             k == 1 ? 8 : 4, dmax, smax, mdiff[0], mdiff[1], mdiff[2], adiff, nb, nz);
    }
  }
  // ---------------------------------------------------------------- backward chain: weight-streaming (concat_fused.hip, RUNS) against weights-stationary
  {
    // transposed bf16 images [in][out] of W1 / W2 per estimator
    std::vector<float> hp = grab(params, (size_t)E * pstride);
    std::vector<__bf16> ht((size_t)E * pstride);
    for (int e = 0; e < E; ++e)
      for (int l = 0; l < 2; ++l)
        for (int o = 0; o < 256; ++o)
          for (int i = 0; i < 256; ++i) ht[(size_t)e * pstride + l * 65536 + (size_t)i * 256 + o] = (__bf16)hp[(size_t)e * pstride + l * 65536 + (size_t)o * 256 + i];
    __bf16* imgT = dmalloc<__bf16>((size_t)E * pstride); hipMemcpy(imgT, ht.data(), ht.size() * 2, hipMemcpyHostToDevice);
    float* ds = dev_rand((size_t)E * rows, 0.01f, g);
    const long scratch = concat_bwd_dq_scratch(E, B);
    if (scratch <= 0) { printf("backward: no dQ plan for B = %d\n", B); return 0; }
    // masks of the streaming forward (save 2 run above left them in masks[0])
    { ConcatFwdArgs x = a; x.save = 2; x.scores = scores[0]; x.m0 = masks[0][0]; x.m1 = masks[0][1]; x.m2 = masks[0][2]; x.a0b = a0b[0]; x.a1b = a1b[0]; x.a2 = a2[0];
      setenv("MIMRL_CONCAT_STREAMED", "1", 1); concat_fwd_fused(s, x); CK(hipStreamSynchronize(s)); }
    float* grads[2] = {dmalloc<float>((size_t)E * pstride), dmalloc<float>((size_t)E * pstride)};
    float* dQ[2] = {dmalloc<float>((size_t)E * B * 256), dmalloc<float>((size_t)E * B * 256)};
    float* dPo[2] = {dmalloc<float>((size_t)E * B * 256), dmalloc<float>((size_t)E * B * 256)};
    float* part[2] = {dmalloc<float>(scratch), dmalloc<float>(scratch)};
    __bf16* dz2[2] = {a0b[1], a0b[2]};          // (reuse: [E][rows][256] bf16 buffers)
    __bf16* dz1[2] = {a1b[1], a1b[2]};
    for (int wgm = 0; wgm < 2; ++wgm) {
      float us[2] = {0.f, 0.f};
      for (int k = 0; k < 2; ++k) {
        ConcatBwdArgs b{};
        b.ds = ds; b.compact = 1; b.m0 = masks[0][0]; b.m1 = masks[0][1]; b.m2 = masks[0][2]; b.w3 = a.w3;
        b.W2T = imgT + 65536; b.W1T = imgT; b.pstride = pstride; b.dQ = dQ[k]; b.dq_part = part[k]; b.dP = dPo[k]; b.E = E; b.B = B;
        if (wgm) { b.dz2 = dz2[k]; b.dz1 = dz1[k]; b.db1 = grads[k] + 3 * 65536; b.db2 = b.db1 + 256; b.dw3 = b.db1 + 512; b.db3 = b.db1 + 768; }
        if (k == 0) setenv("MIMRL_CONCAT_STREAMED", "1", 1); else unsetenv("MIMRL_CONCAT_STREAMED");
        auto run = [&]() { if (B > 128) hipMemsetAsync(dPo[k], 0, sizeof(float) * E * B * 256, s); return concat_bwd_fused(s, b); };
        for (int w = 0; w < 2; ++w) if (run() != 0) { printf("bwd launch failed: %s\n", mimrl::last_error_slot().c_str()); return 1; }
        CK(hipStreamSynchronize(s));
        hipMemset(grads[k], 0, sizeof(float) * E * pstride);
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < reps; ++r) run();
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms = 0.f; CK(hipEventElapsedTime(&ms, e0, e1)); us[k] = 1e3f * ms / reps;
      }
      const double gflop = 2.0 * 2 * E * rows * 256.0 * 256.0 / 1e9;
      printf("backward %s  B %d E %d: streaming %8.1f us (%.0f TFLOP/s)   stationary %8.1f us (%.0f)   [incl. the dQ / dP reduce launch and the dP memset]\n",
             wgm ? "stage 1 (weight-gradient operands)" : "stage 2", B, E, us[0], gflop / us[0] * 1e3, us[1], gflop / us[1] * 1e3);
      auto cmp = [&](const char* nm, const float* x, const float* y, size_t n, double scale_floor) {
        auto hx = grab(x, n), hy = grab(y, n); double d = 0, m = 0; for (size_t i = 0; i < n; ++i) { d = std::max(d, (double)std::fabs(hx[i] - hy[i])); m = std::max(m, (double)std::fabs(hx[i])); }
        printf("   %-4s max |diff| %.3e of max |value| %.3e (%.2e)\n", nm, d, m, d / std::max(m, scale_floor));
      };
      cmp("dQ", dQ[0], dQ[1], (size_t)E * B * 256, 1e-30);
      cmp("dP", dPo[0], dPo[1], (size_t)E * B * 256, 1e-30);
      if (wgm) {
        auto y0 = grab(dz1[0], (size_t)E * rows * 256), y1 = grab(dz1[1], (size_t)E * rows * 256); long nb = 0; for (size_t i = 0; i < y0.size(); ++i) nb += (float)y0[i] != (float)y1[i];
        auto z0 = grab(dz2[0], (size_t)E * rows * 256), z1 = grab(dz2[1], (size_t)E * rows * 256); long nz = 0; for (size_t i = 0; i < z0.size(); ++i) nz += (float)z0[i] != (float)z1[i];
        printf("   dz1 / dz2 entries differing %ld / %ld of %zu\n", nb, nz, y0.size());
        // (per launch: the bucket slots were cleared before the timed launches, so both hold reps x the gradient)
        for (int e = 0; e < E; ++e) if (e == 0 || e == E - 1) {
          cmp("db1", grads[0] + (size_t)e * pstride + 3 * 65536, grads[1] + (size_t)e * pstride + 3 * 65536, 256, 1e-30);
          cmp("db2", grads[0] + (size_t)e * pstride + 3 * 65536 + 256, grads[1] + (size_t)e * pstride + 3 * 65536 + 256, 256, 1e-30);
          cmp("db3", grads[0] + (size_t)e * pstride + 3 * 65536 + 768, grads[1] + (size_t)e * pstride + 3 * 65536 + 768, 1, 1e-30);
        }
      }
    }
  }
  return 0;
}
