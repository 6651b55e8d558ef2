// Does the CAPTURE ORDER of a fork decide which child stays on the parent's hardware queue?  (tools/hw/graph_order.hip, GPU box)
// A chain of N kernels on s0; behind every chain kernel a short side kernel on s1 (s2, s3 round-robin) that depends on it.
//   variant A: fork + side launch captured BEFORE the next chain kernel (the engine's pattern)
//   variant B: the next chain kernel captured first, the side launch behind it (same dependencies)
// Prints the replay time per chain link for both.  hipcc --offload-arch=gfx950 -O2 graph_order.hip -o graph_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <functional>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void spin(long long ticks, int* sink) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) { }
  if (sink && threadIdx.x == 12345) *sink = 1;
}
// re-insert every node's outgoing edges so that the child with the longest path to a sink comes first
static int reorder_edges(hipGraph_t g) {
  size_t nn = 0, ne = 0;
  CK(hipGraphGetNodes(g, nullptr, &nn));
  std::vector<hipGraphNode_t> nodes(nn);
  CK(hipGraphGetNodes(g, nodes.data(), &nn));
  CK(hipGraphGetEdges(g, nullptr, nullptr, &ne));
  std::vector<hipGraphNode_t> from(ne), to(ne);
  CK(hipGraphGetEdges(g, from.data(), to.data(), &ne));
  auto idx = [&](hipGraphNode_t n) { for (size_t i = 0; i < nn; ++i) if (nodes[i] == n) return (int)i; return -1; };
  std::vector<std::vector<int>> out(nn);
  for (size_t e = 0; e < ne; ++e) out[idx(from[e])].push_back(idx(to[e]));
  std::vector<int> h(nn, -1);
  std::function<int(int)> height = [&](int v) { if (h[v] >= 0) return h[v]; int m = 0; for (int c : out[v]) m = std::max(m, 1 + height(c)); return h[v] = m; };
  for (size_t v = 0; v < nn; ++v) height((int)v);
  for (size_t v = 0; v < nn; ++v) {
    if (out[v].size() < 2) continue;
    std::vector<int> o = out[v];
    std::stable_sort(o.begin(), o.end(), [&](int a, int b) { return h[a] > h[b]; });
    if (o == out[v]) continue;
    std::vector<hipGraphNode_t> f(o.size(), nodes[v]), t;
    for (int c : out[v]) t.push_back(nodes[c]);
    CK(hipGraphRemoveDependencies(g, f.data(), t.data(), t.size()));
    t.clear();
    for (int c : o) t.push_back(nodes[c]);
    CK(hipGraphAddDependencies(g, f.data(), t.data(), t.size()));
  }
  return 0;
}
int main(int argc, char** argv) {
  const int N = 40, NS = argc > 1 ? atoi(argv[1]) : 3;
  const long long chain_ticks = 500, side_ticks = 1000;   // 100 MHz: 5 us / 10 us
  hipStream_t s0, side[3];
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  for (auto& s : side) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  std::vector<hipEvent_t> ev(2 * N + 8);
  for (size_t q = 0; q < ev.size(); ++q) CK(hipEventCreateWithFlags(&ev[q], hipEventDisableTiming));
  for (int variant = 0; variant < 8; ++variant) {
    hipGraph_t g; hipGraphExec_t ex;
    CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
    spin<<<1, 64, 0, s0>>>(chain_ticks, nullptr);
    for (int i = 0; i < N; ++i) {
      hipStream_t sd = side[i % NS];
      CK(hipEventRecord(ev[i], s0));                       // after chain kernel i
      if (variant == 4) { spin<<<1, 64, 0, s0>>>(chain_ticks, nullptr); continue; }            // E: whole chain first (events recorded), all side launches behind it
      if (variant == 5) {                                                                        // F: side kernels hang off the chain's FIRST node only
        if (i == 0) for (int j = 0; j < NS; ++j) CK(hipStreamWaitEvent(side[j], ev[0], 0));
        spin<<<64, 64, 0, sd>>>(side_ticks, nullptr); spin<<<1, 64, 0, s0>>>(chain_ticks, nullptr); continue;
      }
      if (variant == 6) {                                                                        // G: fork behind every link, side kernel 1 us
        spin<<<1, 64, 0, s0>>>(chain_ticks, nullptr); CK(hipStreamWaitEvent(sd, ev[i], 0)); spin<<<1, 64, 0, sd>>>(100, nullptr); continue;
      }
      if (variant == 2) { spin<<<1, 64, 0, s0>>>(chain_ticks, nullptr); continue; }   // C: no side work at all
      if (variant == 3 && (i & 3)) { spin<<<1, 64, 0, s0>>>(chain_ticks, nullptr); continue; }   // D: a fork behind every 4th link only (order B)
      if (variant == 0 || variant == 7) {
        CK(hipStreamWaitEvent(sd, ev[i], 0));
        spin<<<64, 64, 0, sd>>>(side_ticks, nullptr);      // side child captured first
        spin<<<1, 64, 0, s0>>>(chain_ticks, nullptr);      // chain child second
      } else {
        spin<<<1, 64, 0, s0>>>(chain_ticks, nullptr);      // chain child first
        CK(hipStreamWaitEvent(sd, ev[i], 0));
        spin<<<64, 64, 0, sd>>>(side_ticks, nullptr);
      }
    }
    if (variant == 4) for (int i = 0; i < N; ++i) { hipStream_t sd = side[i % NS]; CK(hipStreamWaitEvent(sd, ev[i], 0)); spin<<<64, 64, 0, sd>>>(side_ticks, nullptr); }
    for (int j = 0; j < NS; ++j) { CK(hipEventRecord(ev[N + j], side[j])); CK(hipStreamWaitEvent(s0, ev[N + j], 0)); }
    CK(hipStreamEndCapture(s0, &g));
    if (variant == 7 && reorder_edges(g)) return 1;   // H: A's capture, edges re-inserted chain-first
    CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int w = 0; w < 5; ++w) CK(hipGraphLaunch(ex, s0));
    CK(hipStreamSynchronize(s0));
    const int R = 50;
    CK(hipEventRecord(a, s0));
    for (int r = 0; r < R; ++r) CK(hipGraphLaunch(ex, s0));
    CK(hipEventRecord(b, s0));
    CK(hipStreamSynchronize(s0));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    printf("variant %c (%d side streams): %.2f us per chain link (chain kernel 5 us, side kernel 10 us)\n", "ABCDEFGH"[variant], NS, 1e3 * ms / R / (N + 1));
  }
  return 0;
}
