// Fused ReLU-MLP stacks for the estimators (critic towers VMI.py:13-22,53-57; CMI classifiers Model.py:47-72), bf16 MFMA.
// A stack is tiny (<= 256 rows per group, <= 4 layers, widths <= 384): as separate GEMMs every layer is a 10-20 us
// kernel boundary on the critical path.  Here one workgroup owns 32 rows of one group for the WHOLE stack: the
// activations stay in LDS (bf16), the weights stream from L2 straight into MFMA B-fragments (fp32 -> bf16 in registers;
// with a single 32-row M-tile no other wave would reuse them, so LDS staging buys nothing).
#pragma once
#include "common.h"

namespace mimrl {

constexpr int MLPF_MAX_LAYERS = 4;
constexpr int MLPF_MAX_WIDTH = 384;

struct MlpFusedArgs {
  int nb, rows, brows;            // groups, valid rows per group, row pitch between groups in every activation buffer
  int nl;                         // layers
  int dims[MLPF_MAX_LAYERS + 1];  // widths: dims[0] input ... dims[nl] output
  const float* W[MLPF_MAX_LAYERS];   // [dims[l+1], dims[l]] row-major, group g at + g*pstride
  const float* b[MLPF_MAX_LAYERS];   // [dims[l+1]]
  long pstride;
  // optional bf16 images of the weights at the same group stride: Wb[l] = W[l] as is (forward), WbT[l] = W[l] transposed
  // to [dims[l], dims[l+1]] (data-gradient chain).  With images the kernels never touch the fp32 weights.
  const __bf16* Wb[MLPF_MAX_LAYERS];
  const __bf16* WbT[MLPF_MAX_LAYERS];
  // optional FRAGMENT-ORDER images (bf16_frag_images below) at the same offsets: Wf[l] of the forward product, WfT[l] of the
  // data-gradient product; only layers whose matrix is [32k x 64k] in that product have one (narrow top layers use Wb).  With them,
  // a 4-layer stack of hidden width 256 takes mlp_frag_kernel (weights go global -> VGPR -> MFMA, no LDS staging)
  const __bf16* Wf[MLPF_MAX_LAYERS];
  const __bf16* WfT[MLPF_MAX_LAYERS];
  const float* in;                // [nb, brows, dims[0]]
  float* act[MLPF_MAX_LAYERS];    // post-ReLU outputs of layers 0..nl-2: [nb, brows, dims[l+1]]  (kept for the backward pass)
  float* out;                     // [nb, brows, dims[nl]]  (linear)
  // backward only
  const float* dout;              // [nb, brows, dims[nl]]
  float* dz[MLPF_MAX_LAYERS];     // dz[l] = gradient w.r.t. the pre-activation of layer l-1's output, l = 1..nl-1: [nb, brows, dims[l]]
  float* din;                     // [nb, brows, dims[0]] or null
  float* db[MLPF_MAX_LAYERS];     // optional bias gradients of layers 0..nl-2 (column sums of dz[l+1]), group g at + g*pstride
  float* db_top;                  // optional bias gradient of the TOP layer (column sums of dout), group g at + g*pstride
  float* dw_top;                  // optional WEIGHT gradient of a narrow top layer (dims[nl] * dims[nl-1] <= 512, e.g. the 2-logit CMI
                                  // head: dW = dout^T act[nl-2]), accumulated; only where mlp_bwd_takes_top_wgrad() says so
  int act_slack;                  // backward: >= 4 * MLPF_MAX_WIDTH floats are readable behind the END of every act[] buffer (the engine's arena
                                  // guarantees it; the operator-level ABI cannot and leaves it 0: ragged tiles then take the 4-wave kernel)
  int dbg;                        // timing experiments only (MIMRL_DBG_MLPB): 1 no ReLU mask, 2 no bias atomics, 4 no dz stores
};

bool mlp_fused_supported(int nb, int rows, int nl, const int* dims);
int mlp_stack_fwd_fused(hipStream_t s, const MlpFusedArgs& a);
// data-gradient chain only: fills dz[1..nl-1] (+ din, + db[]); weight gradients stay GEMMs over (dz, act)
int mlp_stack_bwd_fused(hipStream_t s, const MlpFusedArgs& a);
// true if mlp_stack_bwd_fused will also produce dw_top for this stack (8-wave image kernel, narrow top layer)
bool mlp_bwd_takes_top_wgrad(const MlpFusedArgs& a);

// bf16 images of a parameter bucket: dst[i] = bf16(src[i]); and, for a table of strided groups of [N,K] matrices,
// dstT[off + g*gstride + k*N + n] = bf16(src[off + g*gstride + n*K + k])
int bf16_image(hipStream_t s, const float* src, __bf16* dst, long n);
struct TransposeTable { long off[12]; int N[12], K[12], nb[12]; long gstride[12]; int n; };
int bf16_transposed_images(hipStream_t s, const float* src, __bf16* dstT, const TransposeTable& t);
// MFMA-fragment-order images (layout: frag_images_kernel in mlp_fused.hip).  Entry e: nb groups (pitch gstride floats) of a matrix at
// float offset off with OUT output columns (multiple of 32) and reduction length RED (multiple of 64); tr = 0: element (c, r) =
// src[c * RED + r] (forward weight), tr = 1: src[r * OUT + c] (the same matrix in the data-gradient product).  blk0 is filled in.
struct FragTable { long off[24]; int OUT[24], RED[24], nb[24], tr[24]; long gstride[24], dshift[24]; int blk0[25]; int n; };   // dshift: added to the destination index (two images, one launch)
int bf16_frag_images(hipStream_t s, const float* src, __bf16* dst, const FragTable& t);

}  // namespace mimrl
