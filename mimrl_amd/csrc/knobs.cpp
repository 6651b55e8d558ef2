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
  {"MIMRL_BEGIN_IN_PACK", "engine_step.hip", "", "0: the begin-of-stage bookkeeping as its own launch instead of riding on the layer-0 pack launch"},
  {"MIMRL_BEGIN_ON_SIDE", "engine_step.hip", "", "begin-of-stage bookkeeping on a side stream (older placement)"},
  {"MIMRL_BPTT_FIRST", "engine_backward.hip", "", "capture order: the layer-1 BPTT in front of the LayerNorm-backward side work"},
  {"MIMRL_CUBE_FWD_GROUPS", "cube_fused.hip", "2", "wave groups per workgroup"},
  {"MIMRL_DAXIS_PG_FUSE", "engine_backward.hip", "", "D-axis parameter gradients folded into daxis_bwd (measured slower)"},
  {"MIMRL_DBG_DELAY_TAG", "engine_backward.hip", "-1", "critical-path probe: the phase tag behind which a spin kernel is injected (tools/critical_path.sh)"},
  {"MIMRL_DBG_DELAY_US", "engine_backward.hip", "50", "critical-path probe: spin time in microseconds"},
  {"MIMRL_DDP_SPLIT", "engine_abi.hip", "", "0: the main gradient bucket all-reduced in one piece (in-library RCCL and dist.py)"},
  {"MIMRL_DG_FP32", "engine_abi.hip", "", "BPTT outputs dg / h_prev stored as fp32 instead of bf16"},
  {"MIMRL_DH0_LAST", "engine_backward.hip", "", "capture order of dh0 vs the side-stream weight gradients"},
  {"MIMRL_DWIH_H16", "engine_abi.hip", "", "0: the layer-1 dW_ih product reads the fp32 layer-0 outputs instead of the recurrence's fp16 copy"},
  {"MIMRL_EST_INTERLEAVE", "engine_estimators.hip", "0", "bit mask (1: stage 1, 2: stage 2): capture both estimator branches' forward halves in front of either backward half; + 4: the image launches on side 3 behind both (measured: within process-to-process noise)"},
  {"MIMRL_EST_MI_FIRST", "engine_estimators.hip", "0", "capture order of the MI and the CMI estimator branches"},
  {"MIMRL_FRAG_INLINE", "engine_step.hip", "", "fragment images of the critics on the main stream instead of side 3"},
  {"MIMRL_FUSED_MLP_BIG", "engine_estimators.hip", "", "concat-critic tail through the direct-from-L2 fused MLP variant (round 1)"},
  {"MIMRL_FWD_BF16", "engine_abi.hip", "", "forward products round to bf16 instead of fp16"},
  {"MIMRL_FWD_FP32_SITES", "engine.h", "0", "bit mask of forward product sites that keep fp32 MFMA operands in bf16 mode (accuracy bisection)"},
  {"MIMRL_GEMM_MIN_WGS", "gemm.hip", "700", "fast-path tile choice: smallest grid (workgroups) a larger tile may leave (default 700: ~3 per CU beat ~2 larger ones)"},
  {"MIMRL_GEMM_NO_BK128", "gemm.hip", "", "generic GEMM: no BK = 128 instantiation for small grids"},
  {"MIMRL_GEMM_NO_FAST", "gemm.hip", "", "no 128x128 / 128x64 / 64x64 fast-path kernels: everything on the generic GEMM"},
  {"MIMRL_GEMM_NO_GROUP", "gemm.hip", "", "gemm_group: one launch per problem instead of the grouped launch"},
  {"MIMRL_GEMM_NO_GROUPK", "gemm.hip", "", "gemm_group_splitk: one launch per problem instead of the grouped split-K launch"},
  {"MIMRL_GEMM_NO_LEAN", "gemm.hip", "", "generic GEMM: no 16-byte-load (lean) instantiations"},
  {"MIMRL_GEMM_NO_RAGGED", "gemm.hip", "", "generic GEMM: lean kernels only for tile-aligned M / N"},
  {"MIMRL_GEMM_NO_XCD", "gemm.hip", "", "no XCD-aware tile order in the GEMM kernels"},
  {"MIMRL_GEMM_WIDE_N", "gemm.hip", "", "128 x 256 tiles for long-reduction weight gradients with 16-bit stored operands and N a multiple of 256 (opt-in; 1: all, 2: M a multiple of 256 only, 3: the others)"},
  {"MIMRL_GEMM_TALL_MIN_M", "gemm_tall.hip", "", "row threshold of the tall LDS-DMA GEMM (default 4096)"},
  {"MIMRL_GEMM_TALL_TN", "gemm_tall.hip", "", "1: tall weight-gradient LDS-DMA kernel on (opt-in: ties the split-K kernel)"},
  {"MIMRL_GEMM_TALL_TN_MIN_K", "gemm_tall.hip", "", "reduction-length threshold of the tall weight-gradient kernel (default 16384)"},
  {"MIMRL_GEMM_TRACE", "gemm.hip", "", "diagnostic: which products miss the fast path, and why"},
  {"MIMRL_GRAPH_DOT", "engine_step.hip", "", "dump the captured two-stage graph to this file (hipGraphDebugDotPrint)"},
  {"MIMRL_GRAPH_PAD", "engine_step.hip", "", "pad-node list for MIMRL_GRAPH_REORDER=3 (tools/pad_search.py)"},
  {"MIMRL_GRAPH_PERM", "engine_step.hip", "", "edge permutation for MIMRL_GRAPH_REORDER=4 (tools/perm_search.py)"},
  {"MIMRL_GRAPH_REORDER", "engine_step.hip", "", "re-insert fork edges of the captured graph: 1 by height, 2 captured-on-parent first, 3 pad nodes, 4 MIMRL_GRAPH_PERM"},
  {"MIMRL_GRAPH_VERBOSE", "engine_step.hip", "", "print what the graph post-processing did"},
  {"MIMRL_GRU_BTV", "gru.hip", "0", "batch rows per recurrence workgroup (1..4; default: ~128 workgroups)"},
  {"MIMRL_GRU_LDS_PAD", "gru.hip", "-1", "KiB of dynamic LDS a small BPTT launch reserves to keep parked kernels off its CUs (default 144 below 129 workgroups)"},
  {"MIMRL_GRU_SKIP", "gru.hip", "0", "probe build only: phase-elimination mask of the recurrence kernels"},
  {"MIMRL_GRU_WAVES", "gru.hip", "4", "8: one hidden unit per lane, two waves per SIMD, in the bf16 recurrence kernels (default 4)"},
  {"MIMRL_GX_F16", "engine_abi.hip", "", "1: hoisted GRU input projections stored as fp16 (measured slower at cfg3)"},
  {"MIMRL_IMGT_FIRST", "engine_estimators.hip", "", "capture order: transposed critic images in front of the stage-2 estimators"},
  {"MIMRL_KMIX_BWD_WGS", "model_ops.hip", "512", "workgroups of kmix_bwd (default 512)"},
  {"MIMRL_KMIX_DX_WGS", "model_ops.hip", "4096", "workgroups of the K-axis data-gradient kernel in the split mode (default 4096)"},
  {"MIMRL_KMIX_PG_WGS", "model_ops.hip", "256", "workgroups of the parked K-axis parameter-gradient kernel"},
  {"MIMRL_KNN_BRUTE", "knn_mfma.hip", "", "tuning / cross-check knob: the round-1 exact scan for every call"},
  {"MIMRL_KNN_WGS", "knn_mfma.hip", "512", "workgroups per launch"},
  {"MIMRL_L0_BWD_PACK", "engine_abi.hip", "", "1: pack the layer-0 inputs for the backward products only"},
  {"MIMRL_L0_PACK", "engine_abi.hip", "", "0: no packed layer-0 operands (2 + 4 launches instead of 1 + 2)"},
  {"MIMRL_L0_WG_SIDE", "engine_backward.hip", "1", "side stream of the packed layer-0 W_hh weight gradient"},
  {"MIMRL_L1_WG_SIDE", "engine_backward.hip", "1", "first side stream of the layer-1 weight gradients (4: sides 4-5)"},
  {"MIMRL_LAXIS_BWD_LONG", "engine_abi.hip", "", "0: the L-axis backward of a long-sequence CubeMLP block (L > 64) as colln_bwd + GEMM chain instead of the LONG instantiation of laxis_bwd_kernel"},
  {"MIMRL_LAXIS_LONG", "engine_abi.hip", "", "0: the L-axis MLP of a long-sequence CubeMLP block (L > 64) as the GEMM chain instead of the one-pass kernel of cube_long.hip"},
  {"MIMRL_LENS_SIDE0", "engine_forward.hip", "", "0: sequence-length scan on side 4 in front of the layer-0 projection instead of side 0"},
  {"MIMRL_LN_BWD_BLOCKS", "model_ops.hip", "0", "workgroups per modality of the LayerNorm + ReLU + dropout backward (default: 128 up to 16384 rows, 512 above)"},
  {"MIMRL_LN_BWD_WAVE_ROWS", "model_ops.hip", "", "the one-row-per-wave kernel of round 2"},
  {"MIMRL_LN_TAIL_SPLIT_FLUSH", "engine_backward.hip", "", "0: with the fused LayerNorm tail, block 0's D-axis parked work is flushed behind the L-axis kernel instead of beside it"},
  {"MIMRL_LN_TAIL_LONG", "engine_abi.hip", "", "1: the encoders' LayerNorm + ReLU + dropout backward rides on the LONG L-axis backward kernel of block 0 (long sequences)"},
  {"MIMRL_LN_TAIL_FUSE", "engine_abi.hip", "", "1: the encoders' LayerNorm + ReLU + dropout backward rides on block 0's L-axis backward kernel (opt-in: measured slower at cfg2)"},
  {"MIMRL_LSTM_MFMA_FP32", "lstm.hip", "", "1: fp32 precision mode runs the fp32-MFMA LSTM kernels instead of the scalar ones (measured slower)"},
  {"MIMRL_LSTM_SCALAR", "lstm.hip", "", "1: the scalar fp32 LSTM kernels of round 1 instead of the MFMA ones"},
  {"MIMRL_MLP_IMG_WAVES", "mlp_fused.hip", "0", "4 = never, 8 = both directions"},
  {"MIMRL_MLP_NO_FRAG", "mlp_fused.hip", "", "the round-2 kernels (read per call: tests/test_gpu_fused_oracle.py toggles it)"},
  {"MIMRL_NO_DAXIS_PG_ONE", "engine_backward.hip", "", "D-axis parameter gradients as three launches instead of daxis_param_grads"},
  {"MIMRL_NO_DEFER_WGRAD", "engine_backward.hip", "", "CubeMLP weight gradients in line instead of parked on side streams"},
  {"MIMRL_NO_FOLD_UNPACK", "engine_abi.hip", "", "packed layer-0 weight gradients through l0_unpack instead of the Adam fold"},
  {"MIMRL_NO_FUSED_BOUNDARY", "engine_step.hip", "", "the round-1 stage boundary"},
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
  {"MIMRL_NO_HEAD_GATHER", "engine_estimators.hip", "", "head backward without the fused feature-gradient gather"},
  {"MIMRL_NO_KNN_PREFETCH", "engine_abi.hip", "", "kNN sampler on the main stream instead of beside the recurrence"},
  {"MIMRL_NO_MI_NCE_TILED", "engine_estimators.hip", "", "separable InfoNCE: the one-workgroup-per-estimator kernel instead of the row-tiled one"},
  {"MIMRL_NO_SHARED_PREFIX", "engine_step.hip", "", "evaluate the prefix twice"},
  {"MIMRL_NO_STEP_GRAPH", "engine_step.hip", "", "Solver.step as two per-stage graphs instead of one combined graph"},
  {"MIMRL_NO_TOP1", "engine_estimators.hip", "", "concat critic score head backward as GEMMs instead of top1_bwd"},
  {"MIMRL_NO_TOP_WGRAD_FUSE", "engine_estimators.hip", "", "top-layer weight gradient of the estimator stacks as a GEMM instead of inside the fused backward"},
  {"MIMRL_NO_WG_BIG_SIDE", "engine_estimators.hip", "", "estimator weight gradients: the large group not on its own side stream"},
  {"MIMRL_NO_WG_GROUP", "engine_estimators.hip", "", "the round-1 schedule (helper stream, alternating)"},
  {"MIMRL_NO_WG_GROUPK", "engine_backward.hip", "", "parked CubeMLP weight gradients as single launches instead of gemm_groupk"},
  {"MIMRL_NO_WG_SPLIT", "engine_estimators.hip", "", "estimator weight gradients: no split across two streams"},
  {"MIMRL_NO_XIN", "engine_abi.hip", "", "layer-0 input projection as its own GEMM instead of fused into the recurrence kernel"},
  {"MIMRL_REC16", "engine_abi.hip", "", "0: the BPTT launches read fp32 dout / h_prev (round 5a): dh0 and ds stored fp32, the fused-projection forward writes its fp32 outputs"},
  {"MIMRL_PREFETCH_FIRST", "engine_step.hip", "", "capture order of the two chains"},
  {"MIMRL_SINGLE_STREAM", "engine_abi.hip", "", "enqueue everything on one stream (no side streams)"},
  {"MIMRL_TAIL_STREAMS", "engine_backward.hip", "1", "streams the layer-0 tail weight gradients are dealt over (default 1)"},
  {"MIMRL_TEXT_BWD_FIRST", "engine_backward.hip", "", "capture order: the W_t weight gradient in front of the recurrence BPTT"},
  {"MIMRL_TEXT_LATE", "engine_forward.hip", "0", "(capture order)"},
  {"MIMRL_WG_SIDES", "engine_backward.hip", "3", "side streams the parked weight-gradient GEMMs are dealt over"},
  {"MIMRL_WG_SPLIT_PARITY", "engine_estimators.hip", "0", "which half of the estimator weight gradients goes to the helper stream"},
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
