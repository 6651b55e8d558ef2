// Where does the time of the 8-wave estimator MLP kernel go?  Stand-alone: compiles mimrl_amd/csrc/mlp_fused.hip with MLP_PHASE_PROBE
// (workgroup 0 leaves 100 MHz ticks at phase boundaries), runs the CMI-classifier and critic-tower shapes of cfg2 and prints phase
// times plus the average back-to-back launch duration.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I mimrl_amd/csrc -I include tools/hw/mlp_phase.hip mimrl_amd/csrc/errors.cpp -o gpurun_out/mlp_phase
#define MLP_PHASE_PROBE 1
#include "../../mimrl_amd/csrc/mlp_fused.hip"

#include <cstdio>
#include <vector>
#include <random>
#include <cmath>


using namespace mimrl;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

static float* dev_rand(size_t n, float scale, std::mt19937& g) {
  std::vector<float> h(n);
  std::normal_distribution<float> d(0.f, scale);
  for (auto& v : h) v = d(g);
  float* p; hipMalloc(&p, n * 4); hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice);
  return p;
}

struct Result { std::vector<float> v; };
static void grab(std::vector<float>& dst, const float* p, size_t n) { size_t o = dst.size(); dst.resize(o + n); hipMemcpy(dst.data() + o, p, n * 4, hipMemcpyDeviceToHost); }

static int run(const char* name, int nb, int rows, int nl, const int* dims, bool topw, bool want_din) {
  std::mt19937 g(1);
  MlpFusedArgs a{};
  a.nb = nb; a.rows = rows; a.brows = rows; a.nl = nl;
  for (int i = 0; i <= nl; ++i) a.dims[i] = dims[i];
  long off[8], tot = 0;
  for (int l = 0; l < nl; ++l) { off[l] = tot; tot += (long)dims[l] * dims[l + 1]; tot = (tot + 7) & ~7L; }
  long boff[8];
  for (int l = 0; l < nl; ++l) { boff[l] = tot; tot += dims[l + 1]; tot = (tot + 7) & ~7L; }
  a.pstride = tot;
  float* P = dev_rand((size_t)tot * nb, 0.05f, g);
  float* G; hipMalloc(&G, (size_t)tot * nb * 4); hipMemset(G, 0, (size_t)tot * nb * 4);
  __bf16 *Pb, *PbT, *Pf, *PfT; hipMalloc(&Pb, (size_t)tot * nb * 2); hipMalloc(&PbT, (size_t)tot * nb * 2); hipMalloc(&Pf, (size_t)tot * nb * 2); hipMalloc(&PfT, (size_t)tot * nb * 2);
  bf16_image(0, P, Pb, tot * nb);
  TransposeTable t{}; t.n = nl;
  for (int l = 0; l < nl; ++l) { t.off[l] = off[l]; t.N[l] = dims[l + 1]; t.K[l] = dims[l]; t.nb[l] = nb; t.gstride[l] = tot; }
  bf16_transposed_images(0, P, PbT, t);
  FragTable ff{}, ft{};
  for (int l = 0; l < nl; ++l) {
    const int N = dims[l + 1], K = dims[l];
    if (N % 32 == 0 && K % 64 == 0) { int e = ff.n++; ff.off[e] = off[l]; ff.OUT[e] = N; ff.RED[e] = K; ff.nb[e] = nb; ff.tr[e] = 0; ff.gstride[e] = tot; }
    if (K % 32 == 0 && N % 64 == 0) { int e = ft.n++; ft.off[e] = off[l]; ft.OUT[e] = K; ft.RED[e] = N; ft.nb[e] = nb; ft.tr[e] = 1; ft.gstride[e] = tot; }
  }
  if (bf16_frag_images(0, P, Pf, ff) || bf16_frag_images(0, P, PfT, ft)) { printf("frag images: %s\n", last_error_slot().c_str()); return 1; }
  for (int l = 0; l < nl; ++l) { a.W[l] = P + off[l]; a.b[l] = P + boff[l]; a.Wb[l] = Pb + off[l]; a.WbT[l] = PbT + off[l]; }
  const size_t R = (size_t)nb * rows;
  a.in = dev_rand(R * dims[0], 1.f, g);
  for (int l = 0; l < nl - 1; ++l) { hipMalloc(&a.act[l], (R * dims[l + 1] + 4 * MLPF_MAX_WIDTH) * 4); hipMalloc(&a.dz[l + 1], R * dims[l + 1] * 4); a.db[l] = G + boff[l]; }
  hipMalloc(&a.out, R * dims[nl] * 4);
  a.dout = dev_rand(R * dims[nl], 1.f, g);
  if (want_din) hipMalloc(&a.din, R * dims[0] * 4);
  a.db_top = G + boff[nl - 1];
  a.dw_top = topw ? G + off[nl - 1] : nullptr;
  a.act_slack = 1;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<float> res[2];
  for (int variant = 0; variant < 2; ++variant) {
    for (int l = 0; l < nl; ++l) { a.Wf[l] = variant ? Pf + off[l] : nullptr; a.WfT[l] = variant ? PfT + off[l] : nullptr; }
    for (int bwd = 0; bwd < 2; ++bwd) {
      long long ph[64];
      // one clean pass for the comparison
      hipMemset(G, 0, (size_t)tot * nb * 4);
      if (bwd) { if (mlp_stack_bwd_fused(0, a)) { printf("bwd: %s\n", last_error_slot().c_str()); return 1; } }
      else if (mlp_stack_fwd_fused(0, a)) { printf("fwd: %s\n", last_error_slot().c_str()); return 1; }
      CK(hipDeviceSynchronize());
      if (!bwd) { grab(res[variant], a.out, R * dims[nl]); for (int l = 0; l < nl - 1; ++l) grab(res[variant], a.act[l], R * dims[l + 1]); }
      else { for (int l = 1; l < nl; ++l) grab(res[variant], a.dz[l], R * dims[l]); if (want_din) grab(res[variant], a.din, R * dims[0]); grab(res[variant], G, (size_t)tot * nb); }
      for (int i = 0; i < 5; ++i) { if (bwd) mlp_stack_bwd_fused(0, a); else mlp_stack_fwd_fused(0, a); }
      CK(hipDeviceSynchronize());
      const int N = 200;
      hipEventRecord(e0, 0);
      for (int i = 0; i < N; ++i) { if (bwd) mlp_stack_bwd_fused(0, a); else mlp_stack_fwd_fused(0, a); }
      hipEventRecord(e1, 0);
      CK(hipDeviceSynchronize());
      float ms; hipEventElapsedTime(&ms, e0, e1);
      CK(hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_mlp_phase), sizeof(ph)));
      printf("%s %s %s: nb %d rows %d  back-to-back %.2f us/launch;  workgroup 0 phases (us since entry):\n", name, variant ? "FRAG" : "img8", bwd ? "bwd" : "fwd", nb, rows, 1e3 * ms / N);
      printf("   prologue %.2f", (ph[1] - ph[0]) * 0.01);
      for (int s = 0; s < nl; ++s)
        if (ph[2 + 4 * s] >= ph[0]) printf(" | L%d mfma-done %.2f epi-done %.2f", s, (ph[2 + 4 * s] - ph[0]) * 0.01, (ph[3 + 4 * s] - ph[0]) * 0.01);
      printf(" | end %.2f\n", (ph[20] - ph[0]) * 0.01);
      long long z[64] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_mlp_phase), z, sizeof(z));
    }
  }
  if (res[0].size() != res[1].size()) { printf("%s: result sizes differ\n", name); return 1; }
  double worst = 0, scale = 0; size_t wi = 0;
  for (size_t i = 0; i < res[0].size(); ++i) { const double d = fabs((double)res[0][i] - res[1][i]); if (d > worst) { worst = d; wi = i; } scale = fmax(scale, fabs((double)res[0][i])); }
  printf("%s: img8 vs FRAG over %zu values: max |diff| %.3g at %zu (values up to %.3g)\n", name, res[0].size(), worst, wi, scale);
  return 0;
}

int main() {
  const int cmi[5] = {384, 256, 256, 256, 2};
  const int tower[5] = {128, 256, 256, 256, 128};
  const int base[5] = {128, 256, 256, 256, 1};
  if (run("cmi  ", 6, 256, 4, cmi, true, false)) return 1;
  if (run("cmi2 ", 6, 256, 4, cmi, false, true)) return 1;
  if (run("tower", 10, 128, 4, tower, false, true)) return 1;
  if (run("base ", 5, 128, 4, base, true, true)) return 1;
  if (run("cmi-r", 6, 250, 4, cmi, true, true)) return 1;     // ragged last tile
  if (run("tow-r", 10, 45, 4, tower, false, true)) return 1;
  if (run("cmi1 ", 1, 32, 4, cmi, true, false)) return 1;      // one workgroup: the latency floor
  return 0;
}
