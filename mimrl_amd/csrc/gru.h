// Launch descriptors of the persistent bi-GRU recurrence kernels (gru.hip).
#pragma once
#include "common.h"

namespace mimrl {

struct GruSeq {
  const float* gx;     // [B,T,3H]  x W_ih^T + b_ih  (hoisted input projection)
  const float* w_hh;   // [3H,H]
  const float* b_hh;   // [3H]
  float* out;          // [B,T,out_ld]; this direction writes columns [dir*H, dir*H+H)
  float* saved;        // gate slab for BPTT (gru_saved_floats(B,T) floats) or nullptr
  _Float16* out16 = nullptr;   // optional (bf16 mode; all sequences of a launch or none): the same outputs again as fp16, same indexing -- the
                               // operand the next layer's input projection reads (gemm_fast_f16s_kernel: it rounded them to fp16 anyway)
};

// In-kernel launch stamps (bench.py's roofline block: the duration of a launch INSIDE a replayed hipGraph, where HIP events cannot
// bracket a single kernel).  ring[((*step & (slots - 1)) * 4 + id) * 2 + {0, 1}] = {min over workgroups of the start time, min of
// ~(end time)} in wall_clock64 ticks (100 MHz), accumulated with atomicMin into a ring the caller pre-fills with 0xFF.  Null: off.
struct KernelStamp { unsigned long long* ring = nullptr; const int* step = nullptr; int slots = 0, id = 0; };

struct GruFwdArgs {
  KernelStamp stamp;
  GruSeq seq[2][2];        // [modality][direction]
  const int* lens[2];      // [modality][B] valid lengths (packed-sequence semantics)
  int B, T, out_ld, nmod;
  int btv;             // batch rows per workgroup (1..4)
  int no_out32 = 0;    // (fused input projection + out16 only) the fp32 outputs are NOT written: every consumer reads the fp16 copy
  int gx_f16 = 0;      // gx is an FP16 array behind the float-typed pointer (bf16 mode only; written by a GemmDesc::c_f16 projection)
  // Fused input projection (bf16 mode, layer 0 of the packed path): gx = x W_ih^T + b_ih is NOT read; the kernel computes it per cell step
  // from the fp16 packed inputs xin[modality] [B*T, kp] and the fp16 packed weights wih[modality][direction] [3H, kp] (kp % 8 == 0,
  // kp <= 96; columns >= the true width are zero in both) with three more k-steps per gate on a matrix pipe that is idle most of the
  // step -- they do not depend on h.  Runs the 8-wave kernel (one unit per lane: the 4-wave one has no registers left for the 9 extra
  // weight fragments) and writes the saved-gate slab in the 4-wave layout, so the BPTT launch of the layer is the usual one.
  int xin_on = 0, kp = 0;
  const _Float16* xin[2] = {nullptr, nullptr};
  const _Float16* wih[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
  const float* bih[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
};

struct GruSeqBwd {
  const float* w_hh;   // [3H,H]
  const float* saved;  // gate slab written by the forward kernel
  const float* out;    // forward outputs [B,T,out_ld] (source of h_prev)
  const float* dout;   // gradient w.r.t. this direction's outputs, [B,T,dout_ld] (+ dir*dout_off)
  float* dg;           // [B,T,4H] = [dr' | dz' | dn' | dn'*r]:  dgx = columns [0,3H), dgh = columns [0,2H) + [3H,4H)
  float* hprev;        // [B,T,H]  h_{prev} of every step (0 at sequence starts / padded steps): operand of dW_hh
  float* db_ih;        // [3H] += sum_{b,t} dgx   (nullable)
  float* db_hh;        // [3H] += sum_{b,t} dgh   (nullable)
  const _Float16* out16 = nullptr;   // optional (all sequences of a launch or none; bf16 mode, 4-wave kernel): the forward kernel's fp16 copy of `out`
                                     // (GruSeq::out16) -- h_prev is read from it and `out` is not touched (it may never have been written)
};

struct GruBwdArgs {
  KernelStamp stamp;
  GruSeqBwd seq[2][2];
  const int* lens[2];
  int B, T, out_ld, dout_ld, dout_off, nmod;
  int btv;             // must equal the forward launch's value (addresses the saved-gate slab)
  int dg_bf16 = 0;     // dg / hprev are written as bf16 (same element indices): their only consumers are bf16-operand GEMMs
  int dout_bf16 = 0;   // dout is a bf16 array behind the float-typed pointers (same element indices; bf16 mode with dg_bf16, 4-wave kernel)
  // record layout of the saved-gate slab the forward launch of this layer wrote: 0 = whatever gru_upl() says (the forward ran the
  // kernel MIMRL_GRU_WAVES picked), 2 = the 4-wave layout regardless (the fused-projection forward, GruFwdArgs::xin_on, always writes
  // that one): the BPTT kernel is chosen by the slab it has to read, not by the knob
  int slab_upl = 0;
};

int gru_forward(hipStream_t s, const GruFwdArgs& a, bool bf16);
int gru_backward(hipStream_t s, const GruBwdArgs& a, bool bf16);
bool gru_bwd_io16_ok(int slab_upl);
long gru_saved_floats(int B, int T);
// batch rows per workgroup (<= 4): the kernels are bound by the per-step instruction latency of one wave, so the batch
// is spread over as many CUs as possible
int gru_pick_btv(int B, int nmod);
#ifdef MIMRL_PHASE_PROBE
int gru_bwd_read_phases(long long* out);   // 2 layers x 16 slots, see gru.hip
#endif
void gru_probe_setup();   // `make PHASE_PROBE=1` builds: reads MIMRL_GRU_SKIP (gru.hip); a no-op otherwise

}  // namespace mimrl
