/* libmimrl_hip -- C ABI of the MI355X-native MIMRL two-stage training step.
 *
 * The reference (kiva12138/MIMRL) has no FFI: its boundary is the Python class surface
 *   Solver.train loop bodies      Solver.py:204-216 (stage 1), :220-238 (stage 2)
 *   Model.forward                 Model.py:388-519
 *   Model.compute_vmi_loss_stage1 Model.py:305-341,  ..._stage2  Model.py:343-386
 *   clip_grad_value_ + Adam.step  Solver.py:144-146,211-213,233-235
 * This header is what a ctypes / pybind / cgo binding of that path binds to.
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on success or a negative
 * MIMRL_ERR_* code (message via mimrl_last_error(), thread-local); nothing throws or aborts.  All device
 * memory except the private workspace is owned by the caller (PyTorch-ROCm tensors on the Python side) and
 * borrowed for the lifetime of the handle.  All work is enqueued on the stream given at create time;
 * no call synchronises the host.  One handle per (process, GPU); calls on a handle are not re-entrant.
 */
#ifndef MIMRL_H_
#define MIMRL_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIMRL_ABI_VERSION 6
#define MIMRL_MAX_BLOCKS 4

enum { MIMRL_OK = 0, MIMRL_ERR_ARG = -1, MIMRL_ERR_HIP = -2, MIMRL_ERR_STATE = -3, MIMRL_ERR_NODEVICE = -4 };
enum { MIMRL_GROUP_MAIN = 0, MIMRL_GROUP_CRITIC = 1 };
enum { MIMRL_CRITIC_SEPARATE = 0, MIMRL_CRITIC_CONCAT = 1 };                       /* VMI.py:35-45 */
enum { MIMRL_BOUND_INFONCE = 0, MIMRL_BOUND_NWJ, MIMRL_BOUND_TUBA, MIMRL_BOUND_DV, MIMRL_BOUND_JS_FGAN,
       MIMRL_BOUND_JS, MIMRL_BOUND_SMILE, MIMRL_BOUND_MINE, MIMRL_BOUND_INTERPOLATE };                                        /* Model.py:121-146 */
enum { MIMRL_ACT_NONE = 0, MIMRL_ACT_RELU = 1, MIMRL_ACT_GELU = 2, MIMRL_ACT_TANH = 3 };
/* MFMA operand type per section (bit mask); accumulation, recurrent state, statistics and optimizer are always fp32 */
enum { MIMRL_PREC_FP32 = 0, MIMRL_PREC_BF16_GEMM_FWD = 1, MIMRL_PREC_BF16_GEMM_BWD = 2, MIMRL_PREC_BF16_GRU_FWD = 4,
       MIMRL_PREC_BF16_GRU_BWD = 8, MIMRL_PREC_BF16 = 15 };

enum { MIMRL_BASELINE_CONSTANT = 0, MIMRL_BASELINE_GAUSSAIN = 1, MIMRL_BASELINE_UNNORMALIZED = 2 };   /* VMI.py:72-110 */
enum { MIMRL_ENCODER_GRU = 0, MIMRL_ENCODER_CONV = 1, MIMRL_ENCODER_LSTM = 2 };    /* Model.py:247-257 */

/* Hot-path subset of Parameters.py:8-70 (same meaning as the flags of the same name). */
typedef struct mimrl_cfg {
  int32_t batch;                 /* --batch_size (rows per rank) */
  int32_t seq_len;               /* T of the input triples, <= time_len */
  int32_t time_len;              /* --time_len (L axis of CubeMLP) */
  int32_t d_t, d_a, d_v;         /* 768 / 74 / 35 for MOSI-shaped data (Config.py:59-76) */
  int32_t d_common;              /* --d_common; must be 128 (Model.py:285 hard-codes embed_dim) */
  int32_t n_blocks;              /* len(--d_hiddens) */
  int32_t d_hiddens[MIMRL_MAX_BLOCKS][3];
  int32_t d_outs[MIMRL_MAX_BLOCKS][3];
  int32_t res_project[MIMRL_MAX_BLOCKS];
  int32_t bias, ln_first;
  int32_t activation;            /* --activate */
  int32_t compose_t_sum, compose_k_sum; /* --features_compose_t/k: 0 mean, 1 sum */
  int32_t critic_type, bound_type;
  int32_t cmi_hardtanh;          /* --cmi_last_acticate hardtanh */
  int32_t k_neighbor;
  int32_t bank_capacity;         /* max rows of the feature banks (dataset size) */
  float dropout[4];              /* --dropout (t,a,v,classifier) */
  float dropout_mlp[3];          /* --dropout_mlp (l,k,d) */
  float coef1[11];               /* --loss_mi_coefficient1 */
  float coef2[8];                /* --loss_mi_coefficient2 */
  float weight_decay, grad_clip; /* --weight_decay, --gradient_clip */
  float beta1, beta2, adam_eps;  /* torch.optim.Adam defaults 0.9 / 0.999 / 1e-8 */
  int32_t precision;             /* MIMRL_PREC_* */
  int32_t use_graph;             /* capture each stage into a hipGraph on first use */
  int32_t device_anchors;        /* 1: draw the kNN anchors on the device each step (overwrites buffers.anchors, where the caller can read the draws of the last step back); 0: host-provided */
  int32_t baseline_type;         /* --baseline_type: MIMRL_BASELINE_CONSTANT | _GAUSSAIN | _UNNORMALIZED (VMI.py:72-110; read by tuba / interpolate) */
  int32_t encoder;               /* --encoders: MIMRL_ENCODER_GRU (Model.py:253-255), _CONV (:247-249,437-439) or _LSTM (:250-252) */
  uint64_t seed;                 /* dropout stream seed */
} mimrl_cfg;

/* Device pointers (fp32 unless noted).  Flat buckets follow the layout reported by mimrl_layout_entry. */
typedef struct mimrl_buffers {
  float *main_p, *main_g, *main_m, *main_v;       /* [mimrl_bucket_floats(MAIN)]   */
  float *crit_p, *crit_g, *crit_m, *crit_v;       /* [mimrl_bucket_floats(CRITIC)] */
  const float *text, *audio, *video, *labels;     /* [B,T,d_t] [B,T,d_a] [B,T,d_v] [B] */
  const float *bank_c, *bank_f, *bank_t, *bank_a, *bank_v; /* [cap,1] [cap,128] x4 : previous epoch's stage-2 features */
  int32_t* anchors;                               /* [2][6][B/k] kNN anchor rows for stage 1 / stage 2 (Model.py:81) */
  float *lr_main, *lr_critic;                     /* device scalars (schedulers rewrite them) */
  float *pred;                                    /* [B] */
  float *feats;                                   /* [4][B,128] = F_F, T_F, A_F, V_F */
  float *scalars;                                 /* [MIMRL_NSCALARS] see MIMRL_S_* */
  const int32_t* knn_override;                    /* optional [2][6][(B/k)*k]: neighbour rows for mimrl_set_knn_override_mask */
  int32_t* counters;                              /* [4] device ints owned by the caller like m / v: [0] dropout/anchor RNG step,
                                                     [1] Adam step of the main bucket, [2] Adam step of the critic bucket
                                                     (torch.optim.Adam's state['step'], Solver.py:144-146), [3] reserved.
                                                     Optional (NULL: private storage).  Handles bound to the same buckets AND
                                                     counters form one optimizer (e.g. a second handle for the partial last
                                                     batch, Parameters.py:21 drop_last=False); saving m, v and these three
                                                     ints is a complete optimizer checkpoint (Solver.py:57-62). */
} mimrl_buffers;

/* indices into mimrl_buffers.scalars */
enum {
  MIMRL_S1_LOSS = 0,      /* stage-1 total loss                                   */
  MIMRL_S1_MIS = 1,       /* 11 values: mi_f_t..mi_t_v, cmi_ac_t..cmi_tc_v         */
  MIMRL_S1_LOSSES = 12,   /* 11 values: -mi x5, BCE x6                             */
  MIMRL_S2_LOSS = 32,     /* stage-2 total loss                                   */
  MIMRL_S2_TASK = 33,     /* MAE task loss                                        */
  MIMRL_S2_MIS = 34,      /* 8 values: f_t f_a f_v inv spec_t spec_a spec_v comp   */
  MIMRL_S2_LOSSES = 42,   /* 8 values                                             */
  MIMRL_NSCALARS = 64
};

const char* mimrl_last_error(void);
int mimrl_abi_version(void);
/* 1 in the deterministic build of this library (libmimrl_hip_det.so, `make det`; loaded when MIMRL_DETERMINISTIC=1 -- the reference's
 * switch is torch.backends.cudnn.deterministic + the seeds of Main.py:14-20): every floating-point accumulation that the default build does
 * with float atomics is order-independent there (64-bit fixed point, csrc/det.h), the engine runs on one stream, and two runs of a stage
 * on the same inputs give bit-identical gradients.  Flag bits on top of the 1 (sticky since load; the call synchronises): | 2 its
 * accumulation table -- 8 M distinct target addresses per launch -- ran full at some launch; | 4 some contribution was NaN / Inf / >= 2^22
 * in magnitude.  Either way that contribution went through a plain float atomic (so a NaN propagates as in the default build) and the
 * run is not bit-reproducible.  0 in the default build. */
int mimrl_deterministic(void);
int mimrl_device_check(void);   /* 0 iff a gfx950 device is usable by this process */

/* ---- parameter layout (host only; callable without a GPU) ---- */
int mimrl_layout_count(const mimrl_cfg* cfg);
int mimrl_layout_entry(const mimrl_cfg* cfg, int idx, char* name, int name_cap, int* group, int64_t* offset, int* ndim,
                       int* dim0, int* dim1);
int mimrl_layout_entry_dim2(const mimrl_cfg* cfg, int idx);   /* third dimension of a 3-d tensor (Conv1d weights), else 0 */
int64_t mimrl_bucket_floats(const mimrl_cfg* cfg, int group);

/* ---- engine ---- */
typedef struct mimrl_handle mimrl_handle;
int mimrl_create(const mimrl_cfg* cfg, void* hip_stream, mimrl_handle** out);
int mimrl_bind(mimrl_handle* h, const mimrl_buffers* bufs);
int mimrl_set_bank_rows(mimrl_handle* h, int rows);    /* 0 => epoch-0 rule (Customization.py:97-98,105-106) */
/* Double-buffered inputs (host-only call, no device work): make input set `set` (0 | 1) = the given (text, audio, video, labels)
 * device buffers the bound batch.  Captured graphs bake the input addresses in, so they are cached per set: a caller that
 * alternates between two fixed buffer sets uploads batch i+1 into the idle set while the step on batch i runs, and pays
 * nothing to switch (the reference's loop does a synchronous .cuda() per batch, Customization.py:47-50).  mimrl_bind sets set 0. */
int mimrl_set_inputs(mimrl_handle* h, int set, const float* text, const float* audio, const float* video, const float* labels);
/* Epoch-ordered critic pass (round 6; replaces the body of the reference's stage-1 loop, Solver.py:200-216: `stage1_n` passes of critic
 * updates over the whole loader with the main model frozen).  mimrl_stage1_pipe_prime: forward pass (Model.forward, Model.py:388-519, training
 * mode) of the BOUND batch into the primary forward set.  mimrl_stage1_pipe(next_valid): stage-1 loss + critic gradients + clip + Adam
 * (Solver.py:205-214) on the bound batch, whose forward pass a previous call left behind; with next_valid != 0 the forward pass of the batch
 * in the OTHER input set (mimrl_set_inputs) runs beside it -- the caller then switches to that set before the next call.  Losses, MI / CMI
 * values, dropout masks and anchor draws are those of mimrl_stage1_step on the same batches in the same order.  Not in stage-2 prefetch mode,
 * not with a communicator; any other step / forward call in between needs a new mimrl_stage1_pipe_prime. */
int mimrl_stage1_pipe_prime(mimrl_handle* h);
int mimrl_stage1_pipe(mimrl_handle* h, int next_valid);
int mimrl_stage1_step(mimrl_handle* h);                /* Solver.py:205-214 : critics update               */
int mimrl_stage2_step(mimrl_handle* h);                /* Solver.py:221-236 : main-model update            */
int mimrl_two_stage_step(mimrl_handle* h);             /* the new Solver.step(datas): mimrl_stage1_step then mimrl_stage2_step on the bound batch; in overlap mode with graphs ONE captured graph, one launch */
int mimrl_stage_grads(mimrl_handle* h, int stage);     /* forward+backward only (data-parallel: all-reduce follows) */
int mimrl_stage_apply(mimrl_handle* h, int stage);     /* value-clip + Adam on that stage's bucket         */
/* mimrl_stage_grads(h, 2) in two launches, for a data-parallel caller that starts reducing gradients while the rest of the backward pass
 * still runs (north_star: "all-reduce ... overlapped with ... backward"; the reference's nn.DataParallel reduces after backward,
 * Solver.py:33-35): part 0 = everything up to and including the layer-1 GRU weight gradients -- afterwards every main-bucket gradient
 * EXCEPT rnn_*_l0* (the layer-0 GRU tensors) is final --, part 1 = the layer-0 BPTT and its weight gradients.  ABI 4. */
int mimrl_stage_grads_part(mimrl_handle* h, int stage, int part);
int mimrl_forward(mimrl_handle* h, int train_mode, int with_losses);  /* Solver.evaluate body (Solver.py:255-258) */
int mimrl_estimate(mimrl_handle* h, int stage);        /* estimators only, on the features of the last forward (Model.py:305/343) */
/* Overlap mode for Solver.step() (one stage-1 + one stage-2 update on ONE batch).  When on,
 * mimrl_stage1_step / mimrl_stage_grads(1) also run the stage-2 Model.forward (Model.py:388-519) of the bound batch on
 * a second stream -- legal because stage 1 (Solver.py:205-214) only updates critic parameters -- and the next
 * mimrl_stage2_step / mimrl_stage_grads(2) consumes it instead of running its own (Solver.py:221).  Results are the
 * same as in sequential mode (same parameters, inputs and dropout key).  The part of that forward pass in front of the
 * first dropout (W_t projection, encoders) is the same function of the same inputs in both stages and is evaluated once;
 * stage 2's kNN sampler (Model.py:74-105) runs beside it.  Contract: between the two calls the caller must not modify
 * the bound inputs, the banks, host-supplied anchors of EITHER stage (mimrl_set_anchors) or main parameters; a stage-2
 * call without a preceding stage-1 call fails with MIMRL_ERR_STATE.  Ignored while the banks are empty (epoch-0 rule). */
int mimrl_set_stage2_prefetch(mimrl_handle* h, int on);
/* Data-parallel variant (on = 2, "deferred tail"): as above, except that the stage-2 forward TAIL (everything behind the
 * first dropout: LN+ReLU+dropout, CubeMLP, head; Model.py:452-515) is not issued beside stage 1 but by this call, which
 * the caller places between the START of the stage-1 gradient all-reduce and mimrl_stage_apply(1): the collective of the
 * critic bucket (the larger one) then runs under it.  Order per step: mimrl_stage_grads(1) -> all-reduce(crit_g) [async]
 * -> mimrl_stage2_forward_tail -> wait -> mimrl_stage_apply(1) -> mimrl_stage_grads(2) -> all-reduce(main_g) ->
 * mimrl_stage_apply(2).  A no-op in the other modes and while the banks are empty. */
int mimrl_stage2_forward_tail(mimrl_handle* h);
/* The gradient buckets are multiplied by `scale` inside the fused clip+Adam (before clipping): 1/world_size after a SUM
 * all-reduce, so that no separate scaling pass runs over the bucket.  Default 1. */
int mimrl_set_grad_scale(mimrl_handle* h, float scale);
/* Tell the engine that parameter buckets were written from outside (checkpoint load, broadcast): cached bf16 images of
 * the critic parameters are rebuilt at the next call.  mimrl_bind implies it; the engine's own Adam keeps them fresh. */
int mimrl_params_changed(mimrl_handle* h);
int64_t mimrl_workspace_bytes(const mimrl_handle* h);
/* phase profiler: HIP events on the engine's stream around each phase of the eager (non-graph) path */
enum { MIMRL_PH_GEMM_MISC = 0, MIMRL_PH_GRU_FWD, MIMRL_PH_GRU_BWD, MIMRL_PH_CUBE_FWD, MIMRL_PH_CUBE_BWD, MIMRL_PH_EST_FWD,
       MIMRL_PH_EST_BWD, MIMRL_PH_OPT, MIMRL_PH_MODEL_MISC, MIMRL_NPHASES };
int mimrl_profile_enable(mimrl_handle* h, int on);     /* forces eager launches while on */
/* Launch stamps of the two persistent recurrence kernels -- the largest single kernels of the step -- that also work INSIDE a
 * replayed hipGraph, where HIP events cannot bracket one kernel: every workgroup records its start / end time (wall_clock64, 100 MHz
 * ticks) with an atomicMin into ring[((rng_step & (slots-1)) * 4 + id) * 2 + {0: start, 1: ~end}] (device uint64, caller-owned,
 * pre-filled with 0xFF; slots a power of two; id 0/1 = forward layer 0/1, 2/3 = BPTT layer 1/0; rng_step = counters[0], +1 per stage).
 * bench.py derives its roofline block from these over the TIMED region.  ring = NULL: off (default).  Drops captured graphs. */
int mimrl_set_kernel_stamps(mimrl_handle* h, unsigned long long* ring, int slots);
/* ---- data parallel (ABI v5): RCCL inside the library.  Reference counterpart: nn.DataParallel's gradient reduce (Solver.py:33-35).
 * One process per GPU, every rank a full replica with its own batch.  mimrl_comm_unique_id (rank 0) fills 128 bytes (ncclUniqueId) that
 * the caller hands to every rank by its own means (torch.distributed broadcast, MPI, a file); mimrl_set_comm (all ranks, after
 * mimrl_bind, collective) creates the handle's communicator on the current device.  From then on every update pass of the handle --
 * mimrl_stage1_step / mimrl_stage2_step / mimrl_two_stage_step, eager or captured -- all-reduces (SUM) the stage's gradient bucket
 * between the gradient pass and clip + Adam, on the engine's own streams, as part of the same graph; the 1/world of the mean is the
 * caller's mimrl_set_grad_scale.  The main bucket travels in two pieces: floats [0, mimrl_main_late_offset) under the layer-0 BPTT,
 * the layer-0 recurrence tensors behind it (MIMRL_DDP_SPLIT=0: one piece).  world = 1 is a valid (one-rank) communicator.
 * unique_id = NULL removes the communicator.  librccl.so.1 is dlopen'ed on first use; single-GPU runs never load it. */
int mimrl_comm_unique_id(void* out128);
int mimrl_set_comm(mimrl_handle* h, const void* unique_id128, int world, int rank);
int64_t mimrl_main_late_offset(const mimrl_handle* h);
/* on != 0: the CRITIC bucket's all-reduce moves bf16 (half the bytes of the larger collective: SURVEY section 5): every rank rounds its
 * gradients to bf16 once, RCCL sums in bf16, the sum is widened in front of clip + Adam -- two conversion launches inside the same graph.
 * Changes the update at the 2^-9 level of each gradient (Adam normalises the scale away); the main bucket stays fp32.  Default off. */
int mimrl_set_comm_critic_bf16(mimrl_handle* h, int on);
int mimrl_profile_read(mimrl_handle* h, float* ms_sum /*[MIMRL_NPHASES]*/, int32_t* launches /*[MIMRL_NPHASES]*/);  /* syncs; resets */
/* GEMM family of the eager steps since the last read: out = {algorithmic FLOPs, algorithmic bytes (operands and output once),
 * summed launch durations in ms (HIP events on each launch's own stream), launches}; syncs; resets */
int mimrl_profile_read_gemm(mimrl_handle* h, double out[4]);
void mimrl_destroy(mimrl_handle* h);

/* ---- operator-level entry points (used by the parity tests; all asynchronous on `stream`) ---- */
int mimrl_op_gemm(void* stream, const float* A, const float* B, float* C, int M, int N, int K, int batch,
                  const int64_t strides[9] /* sa_m sa_k sa_b sb_k sb_n sb_b sc_m sc_n sc_b */, const float* bias_n,
                  const float* bias_m, float alpha, float beta, int act, int precision);
/* same + the optional parts the engine uses: second product C = epi(A.B + A2.B2) (strides2 = sa2_m sa2_k sa2_b sb2_k sb2_n sb2_b),
 * a gap in A's row axis (rows >= a_gap_at live a_gap_rows further on), act'(u) factor, fused column sums; act bit 8 = atomic */
int mimrl_op_gemm_ex(void* stream, const float* A, const float* B, float* C, int M, int N, int K, int batch, const int64_t strides[9],
                     const float* A2, const float* B2, int K2, const int64_t strides2[6], int a_gap_at, int a_gap_rows,
                     const float* bias_n, const float* gradact_u, float* colsum, int act, int precision);
/* (ABI v5) mimrl_op_gemm_ex with operands STORED in 16 bits behind the float-typed pointers (strides in 16-bit elements) and the two-level
 * batch of the engine's projections: entry b = (b / batch_in, b % batch_in), outer strides strides_bo = {sa_bo, sb_bo, sc_bo, bias_n_bo, bias_n_b}
 * (batch_in = 0: flat batch, only bias_n_b is read; NULL = all zero).  flags: bit 0 A stored 16-bit, bit 1 B stored 16-bit, bit 2 the 16-bit type is fp16 (else bf16), bit 3 C is
 * stored as fp16 (strides in fp16 elements), bit 4 accumulate into C with float atomics (weight gradients: C must be zeroed by the caller),
 * bits 8-19 / 20-31 the A-row gap of mimrl_op_gemm_ex (a_gap_at / a_gap_rows).  Tall k-contiguous products (M >= 4096, both operands stored) and tall
 * row-contiguous reductions (K >= 16384, atomic) take the LDS-DMA kernels of csrc/gemm_tall.hip -- the GRU layer-1 input projection and its data gradient dh0 at cfg3 (Model.py:254-255 and autograd). */
int mimrl_op_gemm16(void* stream, const void* A, const void* B, void* C, int M, int N, int K, int batch, const int64_t strides[9],
                    const void* A2, const void* B2, int K2, const int64_t strides2[6], int batch_in, const int64_t strides_bo[5],
                    const float* bias_n, int flags);
/* n <= 12 weight-gradient products C_i += A_i . B_i (float atomics into caller-zeroed outputs; a batch with sc_b = 0 is reduced too) as
 * ONE grouped split-K launch when all share one operand-layout class, else n launches.  dims = n x {M, N, K, batch}, strides = n x 9 as
 * in mimrl_op_gemm.  (The engine's parked CubeMLP weight gradients, Model.py:150-214 backward.) */
int mimrl_op_gemm_wgrad_group(void* stream, int n, const float* const* A, const float* const* B, float* const* C, const int32_t* dims,
                              const int64_t* strides, int precision);
int64_t mimrl_op_gru_saved_floats(int B, int T);
/* one bidirectional GRU layer (both directions): gx/w_hh/b_hh/saved per direction; out [B,T,256] */
int mimrl_op_gru_forward(void* stream, const float* gx_f, const float* gx_r, const float* whh_f, const float* whh_r,
                         const float* bhh_f, const float* bhh_r, const int32_t* lens, float* out, float* saved_f,
                         float* saved_r, int B, int T, int precision);
/* BPTT of that layer: dg_* [B,T,512] rows = [dr'|dz'|dn'|dn'*r] (dgx = cols 0..383, dgh = cols 0..255 + 384..511) */
int mimrl_op_gru_backward(void* stream, const float* whh_f, const float* whh_r, const float* saved_f,
                          const float* saved_r, const int32_t* lens, const float* out, const float* dout, float* dg_f,
                          float* dg_r, float* hprev_f, float* hprev_r, int B, int T, int precision);
/* (ABI v6) the layer-0 weight gradients of four (modality, direction) sequences in ONE pass over dg (gru_wgrad.hip; autograd of nn.GRU,
 * Model.py:254-255): dw_ih[s] [384, kp] += dgx_s^T x_s, dw_hh[s] [384, 128] += dgh_s^T hp_s with dg[s] [rows, 512], x[s] [rows, kp], hp[s]
 * [rows, 128] all STORED as bf16 (dgx / dgh as in mimrl_op_gru_backward); kp a multiple of 8, <= 96; x[s] of the two directions of a
 * modality may be the same array.  The outputs accumulate (zero them first). */
/* (ABI v6) weight gradients of the concat critic's hidden layers (VMI.py:58-65 under autograd; concat_dw.hip): for each of E estimators
 * dw2[e] [256, 256] += dz2[e]^T a1[e] and (optional second set) dw1[e] += dz1[e]^T a0[e], dz* / a* [E][rows, 256] STORED as bf16,
 * dw*[e] dw_stride floats apart; optional third product on the same launch (all three or none): the score head's
 * dw3[e] [256] += ds[e]^T a2[e], ds [E][rows] fp32, a2h [E][rows, 256] STORED as fp16.  With m2 ([E][rows][8] sign words of the
 * layer-2 activation: bit c of word g = column 32 g + c) and w3 (the score head's weight, estimator e at + e * dw_stride) dz2 is NOT read
 * (may be null) but regenerated as bf16(ds[r] * w3[c]) where the bit is set -- needs ds.  With P, Q ([E][B][256] fp32: the separable first
 * layer's two projections) and B (rows = B * B, B a multiple of 32) a0 is NOT read (may be null) but regenerated as bf16(relu(P[i] + Q[j]))
 * for pair row i B + j.  The outputs accumulate. */
int mimrl_op_concat_dw(void* stream, const void* dz2, const void* a1, float* dw2, const void* dz1, const void* a0, float* dw1, int E, int64_t rows,
                       int64_t dw_stride, const float* ds, const void* a2h, float* dw3, const uint32_t* m2, const float* w3, const float* P,
                       const float* Q, int B);
int mimrl_op_gru_wgrad(void* stream, const void* const* dg, const void* const* x, const void* const* hp, float* const* dw_ih,
                       float* const* dw_hh, int64_t rows, int kp);
int mimrl_op_mi_bound(void* stream, const float* scores, float* dscores, float* mi, const float* gscale, int E, int B,
                      int bound);
/* same + the estimators' loss terms (mi_loss of Model.py:115-148; differs from -mi only for `mine`); bit e of lossform:
 * estimator e enters the objective through its loss term (all in stage 1; f_t,f_a,f_v in stage 2, Model.py:386) */
int mimrl_op_mi_bound_ex(void* stream, const float* scores, float* dscores, float* mi, float* mi_loss, const float* gscale,
                         int E, int B, int bound, uint32_t lossform);
/* tuba / interpolate with a log-baseline log a(y_i) per row (VMI.py:72-110): lb, dlb are [E,B]; scores are modified in place */
int mimrl_op_mi_bound_baseline(void* stream, float* scores, float* dscores, float* mi, const float* gscale, const float* lb,
                               float* dlb, int E, int B, int bound);
/* separable critic + InfoNCE in one launch (VMI.py:55-57, 162-166): tout = [E][2][B][128] tower outputs (g(x), h(y)), scores = h g^T,
 * mi[e] = log B + mean_i(s_ii - logsumexp_j s_ij), mi_loss[e] = -mi[e], dtout (may be null) = gscale[e] * d mi / d tout.  bf16 MFMA.
 * tiled = 0: one workgroup per estimator; 1: one per (estimator, 32 score rows) -- that one ACCUMULATES into mi, mi_loss and the g(x)
 * halves of dtout, which the caller zeroes first.  B in {32, 64, 96, 128}. */
int mimrl_op_mi_sep_infonce(void* stream, const float* tout, float* dtout, float* mi, float* mi_loss, const float* gscale, int E,
                            int B, int tiled);
int mimrl_op_knn(void* stream, const float* Z, int dz, int N, const int32_t* anchors, int m, int k, int32_t* idx_out);
/* The device-side anchor draw of a step (Model.py:81, `np.random.choice(range(N), m, replace=False)` per CMI estimator): anchors_out
 * [ncall][m] = for every call c the m rows with the smallest (hash(seed, *step + step_add, stream_id, c, row), row) keys, in key order.
 * The engine calls it with stream_id = 100 + stage and *step = counters[0] (ABI 4; tests/test_gpu_ops.py checks it against the same hash
 * restated in numpy: distinct rows, the exact selection, uniform inclusion frequencies). */
int mimrl_op_sample_anchors(void* stream, int32_t* anchors_out, int ncall, int m, int N, uint64_t seed, const int32_t* step,
                            uint32_t stream_id, int step_add);
/* HOST routine (no device work): the k nearest non-anchor rows of a 1-column bank Z (the labels) for every anchor, with
 * scikit-learn's KDTree tie order (Model.py:82-86 with sklearn 1.7.2; see csrc/knn_r1.cpp).  idx_out [m*k]: ORIGINAL bank rows,
 * nearest first, anchor-major.  Real labels are discrete, so ties decide the product sample of the ta_c / tv_c estimators. */
int mimrl_knn_r1_host(const float* z, int N, const int32_t* anchors, int m, int k, int32_t* idx_out);
/* Use caller-supplied neighbour rows instead of the device kNN for the calls whose bit is set in `call_mask` (bit e = e-th
 * CMI estimator in the order ac_t ta_c vc_t tv_c tc_a tc_v) of `stage` (1 | 2): the rows are read from
 * mimrl_buffers.knn_override [2][6][(B/k)*k] (device int32, stage-major) at every step. */
int mimrl_set_knn_override_mask(mimrl_handle* h, int stage, unsigned call_mask);
int mimrl_op_cmi_loss(void* stream, const float* logits, float* dlogits, float* bce, float* cmi, const float* g_bce,
                      const float* g_cmi, int E, int n, int hardtanh);
/* fused ReLU-MLP stack (critic towers VMI.py:13-22, CMI classifier Model.py:47-72), bf16 MFMA, fp32 in/out.
 * W[l] is [dims[l+1], dims[l]] row-major, group g at + g*pstride (same for b[l], db[l]); activations are
 * [nb, brows, width].  forward: act[l] (l < nl-1) = post-ReLU outputs, out = linear top layer.
 * backward: data-gradient chain only: dz[l] (1 <= l < nl) = d/d(pre-activation of layer l-1's output), din (optional),
 * db[l] (optional, l < nl-1) += column sums of dz[l+1]. */
int mimrl_op_mlp_stack_forward(void* stream, int nb, int rows, int brows, int nl, const int32_t* dims, const float* const* W,
                               const float* const* b, int64_t pstride, const float* in, float* const* act, float* out);
int mimrl_op_mlp_stack_backward(void* stream, int nb, int rows, int brows, int nl, const int32_t* dims, const float* const* W,
                                int64_t pstride, const float* const* act, const float* dout, float* const* dz, float* din,
                                float* const* db);
/* ---- test probes: ONE sub-block of the step run by the engine's own code path (same kernels, buffers, streams and precision mode as a
 * step of this handle) on caller-supplied operands.  The fused bf16 kernels of the benchmarked mode have no stand-alone entry point;
 * the parity tests reach them through these (tests/test_gpu_fused_oracle.py: float64 autograd of the oracle on rounded operands). */
/* CubeMLP stack, MLPEncoder.forward (MLPProcess.py:124-137) and its autograd: x [B,time_len,3,128] -> out [B,ol,ok,128] of the last block;
 * with dout (same shape as out): dx [B,time_len,3,128] and every mlp_encoder.* gradient in main_g (the bucket is zeroed first). */
int mimrl_probe_cube(mimrl_handle* h, const float* x, float* out, const float* dout, float* dx);
/* The encoders of Model.forward on the BOUND batch (Model.py:395-466: W_t projection, both 2-layer bi-GRUs on packed sequences, LayerNorm +
 * ReLU + dropout, zero padding to time_len, the stack) and their autograd: cube_x [B,time_len,3,128] = the CubeMLP's input; with dcube (same
 * shape; dmean [3][B,128] = gradient of the T_F / A_F / V_F temporal means, or NULL = 0): every W_t / rnn_* / ln_* gradient in main_g (the
 * bucket is zeroed first) through the step's own kernels -- gemm_fast_f16, gru_fwd / gru_bwd_kernel, ln_relu_drop_*, the weight-gradient
 * GEMMs (ABI 4; tests/test_gpu_fused_oracle.py::test_encoders_vs_rounded_oracle). */
int mimrl_probe_encoders(mimrl_handle* h, float* cube_x, const float* dcube, const float* dmean);
/* The five MI estimators of `stage` (Model.py:313-319 / 352-361, VMI.py:53-69) on the CALLER-WRITTEN mimrl_buffers.feats, forward +
 * backward: mi [2][5] = bound values, loss terms; scores [5][B][B] (concat critic only, else NULL); stage 1: every vmi_estimator_*
 * gradient in crit_g (zeroed first), objective sum_e -coef1[e] mi_e; stage 2: dtin_out [5][2][B][128] = gradient of
 * sum_e g2[e] mi_e (g2 = -coef2[0], -coef2[1], -coef2[2], -coef2[3], -coef2[3], Model.py:364-386) w.r.t. the (x, y) operand of each estimator. */
int mimrl_probe_mi(mimrl_handle* h, int stage, float* mi, float* scores, float* dtin_out);
/* The six CMI classifiers of `stage` (MLP_For_CMI, Model.py:47-72; BCE and the NWJ-style CMI value, Model.py:185-219) on a CALLER-assembled
 * batch cmi_in [6][2n][384] (n = (B / k) * k rows [x|y|z] of the joint, then n rows of the kNN product sample): logits [6][2n][2] (before the
 * +-10 clamp), vals [2][6] = BCE, CMI; stage 1: every vcmi_estimator_* gradient in crit_g (zeroed first), objective sum_e coef1[5+e] BCE_e;
 * stage 2: dcin_out [6][2n][384] (rows [0,n): the joint rows; the product rows come from the detached banks) = gradient of
 * sum_e g2[e] CMI_e, g2 = (-c5, c4+c5-c7, -c6, c4+c6-c7, -c4, -c4) with c = coef2 (Model.py:381-386). */
int mimrl_probe_cmi(mimrl_handle* h, int stage, const float* cmi_in, float* logits, float* vals, float* dcin_out);
/* Neighbour rows of the last kNN product sample of `stage` (Model.py:82-93): idx_out [6][(B/k)*k] device int32, anchor-major, nearest
 * first -- what cmi_assemble gathered the product rows from.  Asynchronous copy on the engine's stream (ABI 4; the device-anchor parity
 * test compares it with the oracle's kNN on the anchors read back from mimrl_buffers.anchors). */
int mimrl_probe_knn(mimrl_handle* h, int stage, int32_t* idx_out);
int mimrl_op_adam(void* stream, float* p, float* g, float* m, float* v, int64_t n, const float* lr, const int32_t* step,
                  float beta1, float beta2, float eps, float weight_decay, float clip);

#ifdef __cplusplus
}
#endif
#endif /* MIMRL_H_ */
