// Where does a workgroup of the tall LDS-DMA GEMM (mimrl_amd/csrc/gemm_tall.hip) spend its time?  Stand-alone: compiles the kernel with
// TALL_PROBE (every workgroup leaves 100 MHz ticks at {start, tile 0 visible, k-loop done, stores issued, stores retired} + its XCC / CU id),
// runs the cfg3 shapes of the layer-1 input projection and of dh0, prints the launch duration and the phase distribution.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I mimrl_amd/csrc -I include tools/hw/tall_probe.hip mimrl_amd/csrc/errors.cpp -o tools/hw/tall_probe
#define TALL_PROBE 1
#include "../../mimrl_amd/csrc/gemm_tall.hip"

#include <algorithm>
#include <cstdio>
#include <vector>

using namespace mimrl;
namespace mimrl { void capture_note(hipStream_t) {} }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

static int run(const char* name, int M, int N, int K, int K2, int lda, int nmod, int ndir_inner, bool f16, bool cf16, bool bias) {
  // A: [nmod][seg][M, lda] (two segments when K2 > 0), W: [nmod][inner][N, K + K2], C: [nmod][inner][M, N]
  const int nseg = K2 > 0 ? 2 : 1, ninner = ndir_inner;
  size_t na = (size_t)nmod * nseg * M * lda, nw = (size_t)nmod * ninner * N * (K + K2), nc = (size_t)nmod * ninner * M * N;
  unsigned short *A, *W; void* Cm; float* bv;
  CK(hipMalloc(&A, na * 2)); CK(hipMalloc(&W, nw * 2)); CK(hipMalloc(&Cm, nc * (cf16 ? 2 : 4))); CK(hipMalloc(&bv, (size_t)nmod * ninner * N * 4));
  { std::vector<unsigned short> h(na); unsigned x = 12345; for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (unsigned short)((f16 ? 0x3000 : 0x3e00) | ((x >> 9) & 0x83ff)); } CK(hipMemcpy(A, h.data(), na * 2, hipMemcpyHostToDevice)); }
  { std::vector<unsigned short> h(nw); unsigned x = 777; for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (unsigned short)((f16 ? 0x2800 : 0x3d00) | ((x >> 9) & 0x83ff)); } CK(hipMemcpy(W, h.data(), nw * 2, hipMemcpyHostToDevice)); }
  CK(hipMemset(bv, 0, (size_t)nmod * ninner * N * 4));
  GemmDesc d;
  d.A = (const float*)A; d.B = (const float*)W; d.C = (float*)Cm; d.M = M; d.N = N; d.K = K;
  d.sa_m = lda; d.sa_k = 1; d.sb_k = 1; d.sb_n = K + K2; d.sc_m = N; d.sc_n = 1;
  d.a_bf16 = d.b_bf16 = 1; d.f16 = f16; d.c_f16 = cf16; d.bias_n = bias ? bv : nullptr;
  if (K2 > 0) {
    d.batch = nmod; d.sa_b = (long)2 * M * lda; d.sb_b = (long)N * (K + K2); d.sc_b = (long)M * N;
    d.A2 = (const float*)(A + (size_t)M * lda); d.B2 = (const float*)(W + K); d.K2 = K2; d.sa2_m = lda; d.sa2_k = 1; d.sa2_b = d.sa_b; d.sb2_k = 1; d.sb2_n = K + K2; d.sb2_b = d.sb_b;
  } else {
    d.batch = nmod * ninner; d.batch_in = ninner; d.sa_b = 0; d.sa_bo = (long)M * lda; d.sb_b = (long)N * K; d.sb_bo = (long)ninner * N * K;
    d.sc_b = (long)M * N; d.sc_bo = (long)ninner * M * N; d.bias_n_b = N; d.bias_n_bo = (long)ninner * N;
  }
  if (!gemm_tall_ok(d)) { printf("%s: not eligible\n", name); return 1; }
  const long grid = 256;
  unsigned long long* st; CK(hipMalloc(&st, grid * 64)); CK(hipMemset(st, 0, grid * 64));
  unsigned long long* nullp = nullptr;
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_tall_stamps), &nullp, sizeof nullp));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) if (gemm_tall(0, d)) { printf("launch failed\n"); return 1; }
  CK(hipEventRecord(e0, 0));
  const int iters = 20;
  for (int i = 0; i < iters; ++i) gemm_tall(0, d);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / iters, fl = 2.0 * M * N * (K + K2) * nmod * (K2 > 0 ? 1 : ninner);
  const double bytes = (double)nmod * nseg * M * (K2 > 0 ? K : K) * 2 + nw * 2.0 + nc * (cf16 ? 2.0 : 4.0);
  printf("%-44s %8.1f us  %7.1f TF/s  %5.2f TB/s (alg.)\n", name, us, fl / us / 1e6, bytes / us / 1e6);
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_tall_stamps), &st, sizeof st));
  gemm_tall(0, d); CK(hipDeviceSynchronize());
  std::vector<unsigned long long> h(grid * 8);
  CK(hipMemcpy(h.data(), st, grid * 64, hipMemcpyDeviceToHost));
  std::vector<double> ph[5]; unsigned long long tmin = ~0ull, tmax = 0;
  for (long g = 0; g < grid; ++g) {
    const unsigned long long* q = &h[g * 8];
    if (!q[0] || !q[1] || !q[4]) continue;
    tmin = std::min(tmin, q[0]); tmax = std::max(tmax, q[1]);
    ph[0].push_back((q[1] - q[0]) * 0.01); ph[1].push_back(q[2] * 0.01 / q[4]); ph[2].push_back(q[3] * 0.01 / q[4]); ph[3].push_back((double)q[4]); ph[4].push_back(q[5] * 0.01 / q[4]);
  }
  const char* nm[5] = {"workgroup lifetime", "epilogue per tile", "waits per tile", "tiles", "DMA issue per tile"};
  printf("   stamped launch: %.1f us first start -> last end, %zu workgroups\n", (tmax - tmin) * 0.01, ph[0].size());
  for (int p = 0; p < 5; ++p) {
    auto& v = ph[p]; std::sort(v.begin(), v.end());
    double s = 0; for (double x : v) s += x;
    printf("   %-20s mean %7.2f  p10 %7.2f  p50 %7.2f  p90 %7.2f  max %7.2f\n", nm[p], s / v.size(), v[v.size() / 10], v[v.size() / 2], v[v.size() * 9 / 10], v.back());
  }
  (void)hipFree(A); (void)hipFree(W); (void)hipFree(Cm); (void)hipFree(bv); (void)hipFree(st);
  return 0;
}

static int run_tn(const char* name, int M, int N, int K, int gap, bool shared_b) {
  const int nb = 4;
  size_t na = (size_t)nb * K * 512, nbel = (size_t)(shared_b ? 2 : 4) * K * N, nc = (size_t)nb * M * N;
  unsigned short *A, *Bm; float* Cm;
  CK(hipMalloc(&A, na * 2)); CK(hipMalloc(&Bm, nbel * 2)); CK(hipMalloc(&Cm, nc * 4));
  { std::vector<unsigned short> h(na); unsigned x = 12345; for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (unsigned short)(0x3e00 | ((x >> 9) & 0x83ff)); } CK(hipMemcpy(A, h.data(), na * 2, hipMemcpyHostToDevice)); }
  { std::vector<unsigned short> h(nbel); unsigned x = 777; for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (unsigned short)(0x3d00 | ((x >> 9) & 0x83ff)); } CK(hipMemcpy(Bm, h.data(), nbel * 2, hipMemcpyHostToDevice)); }
  CK(hipMemset(Cm, 0, nc * 4));
  GemmDesc d;
  d.A = (const float*)A; d.B = (const float*)Bm; d.C = Cm; d.M = M; d.N = N; d.K = K;
  d.sa_m = 1; d.sa_k = 512; d.sb_k = N; d.sb_n = 1; d.sc_m = N; d.sc_n = 1;
  d.a_bf16 = d.b_bf16 = 1; d.atomic = 1;
  d.batch = 4; d.batch_in = 2; d.sa_b = (long)K * 512; d.sa_bo = 2L * K * 512; d.sb_b = shared_b ? 0 : (long)K * N; d.sb_bo = (shared_b ? 1L : 2L) * K * N;
  d.sc_b = (long)M * N; d.sc_bo = 2L * M * N;
  if (gap) { d.a_gap_at = 256; d.a_gap_rows = gap; }
  if (!gemm_tall_tn_ok(d)) { printf("%s: not eligible\n", name); return 1; }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) if (gemm_tall_tn(0, d)) { printf("launch failed\n"); return 1; }
  CK(hipEventRecord(e0, 0));
  const int iters = 20;
  for (int i = 0; i < iters; ++i) gemm_tall_tn(0, d);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / iters, fl = 2.0 * M * N * (double)K * nb;
  const double bytes = (double)nb * K * M * 2 + nbel * 2.0;
  printf("%-44s %8.1f us  %7.1f TF/s  %5.2f TB/s (operands once)\n", name, us, fl / us / 1e6, bytes / us / 1e6);
  (void)hipFree(A); (void)hipFree(Bm); (void)hipFree(Cm);
  return 0;
}

int main(int argc, char** argv) {
  const int BT3 = 128000;
  const int dbg = argc > 1 ? atoi(argv[1]) : 0;
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_tall_dbg), &dbg, sizeof dbg));
  printf("ablation mask %d (1 no A, 2 no W, 4 no MFMA, 8 A nt, 16 no stores, 32 W nt)\n", dbg);
  if (argc > 2) {
    if (run_tn("dW_ih l1 TN: 384x256x128000 x4, shared B", 384, 256, BT3, 0, true)) return 1;
    if (run_tn("dW_hh l1 TN (gap): 384x128x128000 x4", 384, 128, BT3, 128, false)) return 1;
    return 0;
  }
  if (run("gx l1 (f16, fp32 out): 128000x384x256 x2x2", BT3, 384, 256, 0, 256, 2, 2, true, false, true)) return 1;
  if (run("gx l1 (f16, fp16 out)", BT3, 384, 256, 0, 256, 2, 2, true, true, true)) return 1;
  if (run("dh0 (bf16): 128000x256x(384+384) x2", BT3, 256, 384, 384, 512, 2, 1, false, false, false)) return 1;
  if (run("W_t-like (bf16): 128000x128x768", BT3, 128, 768, 0, 768, 1, 1, false, false, true)) return 1;
  return 0;
}
