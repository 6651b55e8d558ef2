// Kernels of the 5 variational-MI and 6 classifier-CMI estimators (VMI.py:53-69,162-166; Model.py:75-225,305-386).
#pragma once
#include "common.h"
#include "stage_boundary.h"

namespace mimrl {

enum Bound : int { BOUND_INFONCE = 0, BOUND_NWJ = 1, BOUND_TUBA = 2, BOUND_DV = 3, BOUND_JS_FGAN = 4, BOUND_JS = 5,
                   BOUND_SMILE = 6, BOUND_MINE = 7, BOUND_INTERP = 8 };

// copy rows:  dst[i][b,:] = src[i][b,:]   for i < n (table of pointers; used to gather tower inputs)
// (+ optional zero-fill jobs done by the same launch: `rep` chunks of `chunk` floats, `stride` apart)
struct ZeroJob { float* p = nullptr; long chunk = 0, stride = 0; int rep = 0; };
struct CopyTable { const float* src[16]; float* dst[16]; int n; ZeroJob z[2]; };
int copy_rows(hipStream_t s, const CopyTable& t, long floats_each);

// scores [E][B][B] -> mi[e] (bound value) and dscores = gscale[e] * d(mi)/d(scores)   (one workgroup / estimator)
// gscale lives in device memory (loss coefficient with sign); dscores may be null (evaluation only).
// separable critic: scores, bound and the gradients w.r.t. both tower outputs in one launch (tout/dtout: [E][2][B][128])
bool mi_sep_fused_supported(int B);
// InfoNCE only, row-tiled: one workgroup per (estimator, 32 rows of the score matrix) -- the bound is a sum of row terms.  mi / mil
// and the g(x) halves of dtout are ACCUMULATED (float atomics): the caller zeroes them first (CopyTable::z in the same stage).
int mi_sep_nce_tiled(hipStream_t s, const float* tout, float* dtout, float* mi, float* mil, const float* gscale, int E, int B, int do_bwd);
int mi_sep_fused(hipStream_t s, const float* tout, float* dtout, float* mi, float* mil, const float* gscale, int E, int B,
                 int bound, unsigned lossform, int do_bwd, const float* lb = nullptr, float* dlb = nullptr, long lb_stride = 0);
// `nce_ws`: NCE_WS_FLOATS zero-initialised floats owned by the caller (one per engine handle): ticket + slots of the row-tiled InfoNCE kernel;
// nullptr = the one-workgroup-per-estimator kernel, which shares no state between launches
constexpr int NCE_WS_FLOATS = 16 + 16 * 64;
int mi_bound_fwd_bwd(hipStream_t s, const float* scores, float* dscores, float* mi, float* mil, const float* gscale, int E,
                     int B, int bound, unsigned lossform, const float* lb = nullptr, float* dlb = nullptr, long lb_stride = 0,
                     float* nce_ws = nullptr);

// concat critic layer 1:  a1[(i*B+j), c] = relu(P[i,c] + Q[j,c])     P = x Wx^T, Q = y Wy^T + b   (VMI.py:59-65)
int pair_expand_fwd(hipStream_t s, const float* P, const float* Q, float* a1, int E, int B, int Hd);
// da1 (in place -> du1 = da1 * (a1>0)), then dP[i,c] = sum_j du1, dQ[j,c] = sum_i du1
// dQ only, from a du1 that already carries the ReLU mask (the fused concat backward writes it)
int pair_reduce_q(hipStream_t s, const float* du1, float* dQ, int E, int B, int Hd);
int pair_expand_bwd(hipStream_t s, const float* a1, float* da1, float* dP, float* dQ, int E, int B, int Hd);
// relu backward in place: g *= (a > 0)
int relu_bwd_inplace(hipStream_t s, const float* a, float* g, long n);
// Backward of an MLP top layer with ONE output (the concat critic's score head, VMI.py:33: Linear(256, 1)), as a streaming
// pass instead of three 64-column GEMM tiles of which one column is real:
//   dz[g][r,c] = dout[g][r] * W[g][c] * (act[g][r,c] > 0)          gradient into the ReLU layer below
//   dW[g][c]  += sum_r dout[g][r] * act[g][r,c] ;  db_top[g] += sum_r dout[g][r] ;  db_below[g][c] += sum_r dz[g][r,c]
// act / dz are [nb][brows, din] (rows valid: `rows`), W / dW / db_* live in a parameter bucket with group stride pstride.
int top1_bwd(hipStream_t s, const float* dout, const float* W, const float* act, float* dz, float* dW, float* db_top, float* db_below,
             int nb, int rows, int brows, int din, long pstride);

// exact kNN product sampler (Model.py:75-106): for call c and anchor a: the k nearest non-anchor rows of Z (knn_mfma.hip).
// 128-column banks: fp32 MFMA distance tiles (every bank slab read once for all anchors of a call) + exact refinement of the k + 2
// survivors, proven complete or re-done by an exact scan; 1-column banks: exact scan.  Ties -> lower row.
constexpr int KNN_MAX_CALLS = 12;   // the six CMI estimators of both stages in one launch (overlap mode samples stage 2 beside stage 1)
struct KnnCall { const float* Z; int dz; const int* anchors; int* idx_x; };   // anchors [m]; idx_x [m*k] nearest first, anchor-major
struct KnnArgs {
  KnnCall call[KNN_MAX_CALLS];
  int N, m, k, ncall;
};
struct KnnPlan { int KP, nab, S, NTW, RP, ppw, nchunks, nlists; size_t scratch_bytes; };
KnnPlan knn_plan(int N, int m, int k);
size_t knn_scratch_bytes(int Ncap, int m, int k);   // candidate-list scratch that serves every bank size up to Ncap
// scratch = nullptr: a process-wide buffer grown on demand (operator-level ABI; not for captured graphs)
int knn_sample(hipStream_t s, const KnnArgs& a, void* scratch = nullptr, size_t scratch_bytes = 0);

// device-side replacement of `np.random.choice(range(N), m, replace=False)` (Model.py:81): every row gets a random
// 32-bit key (counter hash of seed / step / stream id / call / row); the m rows with the smallest (key, row), in that order, are the
// anchors -- a uniformly random m-subset in uniformly random order.  One workgroup per draw: rows under a hash threshold
// are collected and ranked by counting (knn_mfma.hip); any bank size.  Draw e: out[e][m], call index call[e] (0..5), stream id
// stream_id[e] (the engine: 100 + stage), RNG step *step + step_add[e].
struct AnchorDraws { int* out[KNN_MAX_CALLS]; int call[KNN_MAX_CALLS]; uint32_t stream_id[KNN_MAX_CALLS]; int step_add[KNN_MAX_CALLS]; int n; };
int sample_anchors(hipStream_t s, const AnchorDraws& d, int m, int N, uint32_t seed_lo, uint32_t seed_hi, const int* step);

// classifier input batch (Model.py:160-174):
//   rows [0,n): joint = [x | y | z] of the current batch (operand = feature block [B,128] or the label column)
//   rows [n,2n): prod  = [X[idx_x[j]] | Y[anchor[j/k]] | Z[anchor[j/k]]] from the banks (label bank tiled x128)
struct CmiOperand { const float* cur; const float* bank; int is_label; };   // cur: [B,128] or labels [B]; bank: [N,128] or [N,1]
struct CmiAssembleArgs {
  CmiOperand op[6][3];
  const int* anchors;   // [6][m]
  const int* idx_x;     // [6][n]
  float* out;           // [6][2n][384]
  int n, m, k, ncall;
};
int cmi_assemble(hipStream_t s, const CmiAssembleArgs& a);

// logits [E][2n][2] -> bce[e], cmi[e]; dlogits = g_bce[e]*dBCE/dlogit + g_cmi[e]*dCMI/dlogit   (Model.py:65-72,198-219)
int cmi_loss_fwd_bwd(hipStream_t s, const float* logits, float* dlogits, float* bce, float* cmi, const float* g_bce,
                     const float* g_cmi, int E, int n, int hardtanh);

// dst[b,:] (+)= sum_i src_i[b*ld_i + off_i + :]  for rows b < rows_i   (deterministic gather-sum of input gradients)
int gather_sum(hipStream_t s, float* dst, const GatherSum& g, int B, int D, int accumulate);
struct GatherSum4 { GatherSum g[4]; float* dst[4]; };   // four destinations in one launch (F, T, A, V feature gradients)
int gather_sum4(hipStream_t s, const GatherSum4& g, int B, int D, int first = 0);   // destinations first..3

// fused gradient value-clip + Adam over one flat bucket (Solver.py:144-146,211-213; torch.optim.Adam semantics)
struct AdamArgs {
  float* p; float* g; float* m; float* v; long n;
  const float* lr;        // device scalar (schedulers rewrite it between epochs)
  const int* step;        // device counter, already incremented for this update
  float beta1, beta2, eps, weight_decay, clip;
  __bf16* pimg = nullptr; // optional: bf16 image of the updated parameters (the fused estimator kernels read weights from it)
  float gscale = 1.f;     // the gradient is multiplied by this first (1 / world_size after a SUM all-reduce: no separate scaling pass)
  // optional: gradient pieces still parked in scratch (the packed layer-0 GRU weight gradients, engine_step.hip): element j of bucket range
  // [lo[q], hi[q]) also takes src[q][(j / d[q]) * ld[q] + j % d[q]], which is re-zeroed -- the scatter kernel that used to sit between the
  // last weight-gradient GEMM and this launch (one launch + one dependent-launch gap on the chain) rides on the update
  struct Fold { int n = 0; long lo[8], hi[8]; float* src[8]; int d[8], ld[8]; long lo_all = 0, hi_all = 0; } fold;
  // optional (round 5b, critic bucket of a combined step; n % 8 == 0, no fold): the FORWARD fragment-order images of the estimator stacks
  // (mlp_fused.h: bf16_frag_images, entries with tr = 0) written by this launch -- 8 consecutive elements of a row are one 16-byte piece of
  // the image -- so that the stage-2 stacks depend on the update itself and frag_images leaves the chain (its data-gradient entries, needed
  // only by the stage's backward, stay a side-stream launch).  Entry e covers nb[e] matrices [OUT x RED] at lo[e] + g * gstride[e].
  struct Frag { int n = 0; long lo[12], gstride[12]; int nb[12], mat[12], RED[12]; float inv_gs[12], inv_red[12]; __bf16* dst = nullptr; } frag;
  // optional: the stage boundary (stage_boundary.h) is the first thing workgroup 0 does
  int sb_on = 0;
  StageBoundaryArgs sb;
};
int adam_step(hipStream_t s, const AdamArgs& a);

#ifdef MIMRL_PHASE_PROBE
int nce_read_phases(long long* out);   // 8 ticks, see estimator_ops.hip
#endif

}  // namespace mimrl
