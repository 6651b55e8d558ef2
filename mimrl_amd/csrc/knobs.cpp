// see knobs.h
#include "knobs.h"

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <set>
#include <string>

namespace mimrl {

namespace {
struct KnobDef { const char* name; const char* where; const char* dflt; const char* help; };
const KnobDef kKnobs[] = {
  {"MIMRL_KNOBS", "engine_abi.hip", "", "1: print this table (values in effect) to stderr when a handle is created"},
  {"MIMRL_ADAM_FRAG", "engine_abi.hip", "", "0: combined step with frag_images + stage_boundary as launches behind the critic Adam (round 5a) instead of inside it"},
  {"MIMRL_CONCAT_STREAMED", "concat_fused.hip", "", "1: the concat critic on the weight-streaming kernels of rounds 2-5 (concat_fused.hip) instead of the weights-stationary ones (concat_ws.hip)"},
  {"MIMRL_DBG_DELAY_TAG", "engine_backward.hip", "-1", "critical-path probe: <tag>[:<us>] -- the phase tag behind which a spin kernel of <us> (default 50) microseconds is injected (tools/critical_path.sh)"},
  {"MIMRL_DDP_SPLIT", "engine_abi.hip", "", "0: the main gradient bucket all-reduced in one piece (in-library RCCL and dist.py)"},
  {"MIMRL_DG_FP32", "engine_abi.hip", "", "BPTT outputs dg / h_prev stored as fp32 instead of bf16"},
  {"MIMRL_DWIH_H16", "engine_abi.hip", "", "0: the layer-1 dW_ih product reads the fp32 layer-0 outputs instead of the recurrence's fp16 copy"},
  {"MIMRL_FWD_BF16", "engine_abi.hip", "", "forward products round to bf16 instead of fp16"},
  {"MIMRL_GEMM_TALL_MIN_M", "gemm_tall.hip", "", "row threshold of the tall LDS-DMA GEMM (default 4096)"},
  {"MIMRL_GRAPH_DOT", "engine_step.hip", "", "dump the captured two-stage graph to this file (hipGraphDebugDotPrint)"},
  {"MIMRL_GRU_LDS_PAD", "gru.hip", "-1", "KiB of dynamic LDS a small BPTT launch reserves to keep parked kernels off its CUs (default 144 below 129 workgroups)"},
  {"MIMRL_GRU_SKIP", "gru.hip", "0", "probe build only (make probe): phase-elimination mask of the recurrence kernels; ignored by the default build"},
  {"MIMRL_GRU_WAVES", "gru.hip", "4", "8: one hidden unit per lane, two waves per SIMD, in the bf16 recurrence kernels (default 4)"},
  {"MIMRL_GX_F16", "engine_abi.hip", "", "1: hoisted GRU input projections stored as fp16 (measured slower at cfg3)"},
  {"MIMRL_L0_PACK", "engine_abi.hip", "", "0: no packed layer-0 operands (2 + 4 launches instead of 1 + 2)"},
  {"MIMRL_NO_CONCAT_DW", "engine_estimators.hip", "", "1: the concat critic's hidden-layer weight gradients as two split-K GEMMs instead of the one-launch kernel (concat_dw.hip, round 6)"},
  {"MIMRL_NO_GRU_WGRAD", "engine_backward.hip", "", "1: the layer-0 GRU weight gradients as two batched split-K GEMMs instead of the one-pass kernel (gru_wgrad.hip, round 6)"},
  {"MIMRL_LAXIS_BWD_LONG", "engine_abi.hip", "", "0: the L-axis backward of a long-sequence CubeMLP block (L > 64) as colln_bwd + GEMM chain instead of the LONG instantiation of laxis_bwd_kernel"},
  {"MIMRL_LAXIS_LONG", "engine_abi.hip", "", "0: the L-axis MLP of a long-sequence CubeMLP block (L > 64) as the GEMM chain instead of the one-pass kernel of cube_long.hip"},
  {"MIMRL_LSTM_MFMA_FP32", "lstm.hip", "", "1: fp32 precision mode runs the fp32-MFMA LSTM kernels instead of the scalar ones (measured slower)"},
  {"MIMRL_LSTM_SCALAR", "lstm.hip", "", "1: the scalar fp32 LSTM kernels of round 1 instead of the MFMA ones"},
  {"MIMRL_MLP_NO_FRAG", "mlp_fused.hip", "", "the round-2 kernels (read per call: tests/test_gpu_fused_oracle.py toggles it)"},
  {"MIMRL_NO_FOLD_UNPACK", "engine_abi.hip", "", "packed layer-0 weight gradients through l0_unpack instead of the Adam fold"},
  {"MIMRL_NO_FUSED_CONCAT", "engine_abi.hip", "", "concat critic tail as a GEMM chain instead of concat_fwd / concat_bwd"},
  {"MIMRL_NO_FUSED_CUBE", "engine_abi.hip", "", "bf16 mode: CubeMLP forward as the unfused GEMM / LayerNorm chain"},
  {"MIMRL_NO_FUSED_CUBE_BWD", "engine_abi.hip", "", "bf16 mode: CubeMLP backward as the unfused chain"},
  {"MIMRL_NO_FUSED_MI", "engine_estimators.hip", "", "separable critic: scores / bound / gradients as separate kernels instead of mi_sep_fused / mi_sep_nce"},
  {"MIMRL_NO_FUSED_MLP", "engine_abi.hip", "", "bf16 mode: estimator MLP stacks as grouped GEMMs"},
  {"MIMRL_NO_FUSED_MLP_BWD", "engine_estimators.hip", "", "estimator MLP stacks: data-gradient chain as GEMMs instead of the fused kernel"},
  {"MIMRL_NO_CONCAT_DQ", "engine_estimators.hip", "", "concat critic backward: dZ0 goes out in fp32 and pair_reduce_q sums it over the x rows (rounds 2-4) instead of the in-kernel dQ partial sums"},
  {"MIMRL_NO_DUAL_TAIL_PRE", "engine_step.hip", "", "long sequences (T > 128): each forward tail runs its own text dropout / LN + ReLU + dropout / temporal-mean launches instead of one launch for both"},
  {"MIMRL_NO_FUSED_TAIL_PRE", "engine_forward.hip", "", "forward tail: text dropout, LN + ReLU + dropout and the temporal means as separate launches"},
  {"MIMRL_NO_GEMM_TALL", "gemm_tall.hip", "", "tall LDS-DMA GEMM (gemm_tall.hip) off: the 128x128 register-staged kernels"},
  {"MIMRL_NO_H16", "engine_abi.hip", "", "fp32-stored operands for the layer-1 projection / dh0 (bit-identical results)"},
  {"MIMRL_NO_MI_NCE_TILED", "engine_estimators.hip", "", "separable InfoNCE: the one-workgroup-per-estimator kernel instead of the row-tiled one"},
  {"MIMRL_NO_SHARED_PREFIX", "engine_step.hip", "", "evaluate the prefix twice"},
  {"MIMRL_NO_STEP_GRAPH", "engine_step.hip", "", "Solver.step as two per-stage graphs instead of one combined graph"},
  {"MIMRL_NO_XIN", "engine_abi.hip", "", "layer-0 input projection as its own GEMM instead of fused into the recurrence kernel"},
  {"MIMRL_REC16", "engine_abi.hip", "", "0: the BPTT launches read fp32 dout / h_prev (round 5a): dh0 and ds stored fp32, the fused-projection forward writes its fp32 outputs"},
  {"MIMRL_SINGLE_STREAM", "engine_abi.hip", "", "enqueue everything on one stream (no side streams)"},
};
constexpr int kN = sizeof(kKnobs) / sizeof(kKnobs[0]);
}  // namespace

const char* knob(const char* name) {
  bool known = false;
  for (int i = 0; i < kN && !known; ++i) known = std::strcmp(kKnobs[i].name, name) == 0;
  if (!known) {
    static std::mutex mu;
    static std::set<std::string> told;
    std::lock_guard<std::mutex> g(mu);
    if (told.insert(name).second) std::fprintf(stderr, "[mimrl] knob %s is read but not registered in csrc/knobs.cpp\n", name);
  }
  return std::getenv(name);
}

int knob_int(const char* name, int dflt) {
  const char* v = knob(name);
  return v ? std::atoi(v) : dflt;
}

void knobs_print(FILE* f) {
  std::fprintf(f, "[mimrl] environment knobs (tuning / debugging only; '-' = not set, the default applies)\n");
  for (int i = 0; i < kN; ++i) {
    const char* v = std::getenv(kKnobs[i].name);
    std::fprintf(f, "  %-28s %-10s %-16s %s%s%s\n", kKnobs[i].name, v ? v : "-", kKnobs[i].where, kKnobs[i].help,
                 kKnobs[i].dflt[0] ? " [default " : "", kKnobs[i].dflt[0] ? (std::string(kKnobs[i].dflt) + "]").c_str() : "");
  }
}

}  // namespace mimrl
