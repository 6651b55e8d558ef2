// Parameter layout: see layout.h.  Citations: Model.py:243-303, MLPProcess.py:26-52, VMI.py:32-45, Solver.py:124-133.
#include "layout.h"

#include <cstring>

#include "common.h"

namespace mimrl {

static const char* kVmi[5] = {"f_t", "f_a", "f_v", "t_a", "t_v"};
static const char* kVcmi[6] = {"ac_t", "ta_c", "vc_t", "tv_c", "tc_a", "tc_v"};

int validate_cfg(const mimrl_cfg& c) {
  if (c.batch < 1) return set_error(MIMRL_ERR_ARG, "batch must be >= 1");
  if (c.d_common != 128)
    return set_error(MIMRL_ERR_ARG, "d_common must be 128 (the reference hard-codes embed_dim=128, Model.py:285; got %d)",
                     c.d_common);
  if (c.seq_len < 1 || c.seq_len > c.time_len) return set_error(MIMRL_ERR_ARG, "need 1 <= seq_len <= time_len");
  if (c.n_blocks < 1 || c.n_blocks > MIMRL_MAX_BLOCKS) return set_error(MIMRL_ERR_ARG, "1..%d CubeMLP blocks", MIMRL_MAX_BLOCKS);
  if (c.critic_type != MIMRL_CRITIC_SEPARATE && c.critic_type != MIMRL_CRITIC_CONCAT)
    return set_error(MIMRL_ERR_ARG, "critic_type must be separate|concat (VMI.py:44-45)");
  if (c.bound_type < MIMRL_BOUND_INFONCE || c.bound_type > MIMRL_BOUND_INTERPOLATE) return set_error(MIMRL_ERR_ARG, "bound_type unsupported");
  if (c.k_neighbor < 1 || c.k_neighbor > 8) return set_error(MIMRL_ERR_ARG, "k_neighbor must be in [1,8]");
  if (c.batch / c.k_neighbor < 1) return set_error(MIMRL_ERR_ARG, "batch smaller than k_neighbor");
  if (c.d_t < 1 || c.d_a < 1 || c.d_v < 1) return set_error(MIMRL_ERR_ARG, "feature dims must be positive");
  if (c.baseline_type < MIMRL_BASELINE_CONSTANT || c.baseline_type > MIMRL_BASELINE_UNNORMALIZED)
    return set_error(MIMRL_ERR_ARG, "baseline_type must be constant|gaussain|unnormalized (VMI.py:89-90)");
  if (c.encoder != MIMRL_ENCODER_GRU && c.encoder != MIMRL_ENCODER_CONV && c.encoder != MIMRL_ENCODER_LSTM)
    return set_error(MIMRL_ERR_ARG, "encoder must be gru|conv|lstm");
  int din[3] = {c.time_len, 3, c.d_common};
  for (int i = 0; i < c.n_blocks; ++i) {
    for (int ax = 0; ax < 3; ++ax) {
      if (c.d_hiddens[i][ax] < 1 || c.d_outs[i][ax] < 1) return set_error(MIMRL_ERR_ARG, "CubeMLP dims must be positive");
      if (!c.res_project[i] && din[ax] != c.d_outs[i][ax])
        return set_error(MIMRL_ERR_ARG, "block %d axis %d: identity residual needs d_in == d_out (MLPProcess.py:45-48)", i, ax);
    }
    if (c.d_hiddens[i][1] > 8 || c.d_outs[i][1] > 8) return set_error(MIMRL_ERR_ARG, "K-axis sizes must be <= 8");
    for (int ax = 0; ax < 3; ++ax) din[ax] = c.d_outs[i][ax];
  }
  for (int i = 0; i < 3; ++i)
    if (c.dropout[i] < 0.f || c.dropout[i] >= 1.f) return set_error(MIMRL_ERR_ARG, "dropout must be in [0,1)");
  for (int i = 0; i < 3; ++i)
    if (c.dropout_mlp[i] < 0.f || c.dropout_mlp[i] >= 1.f) return set_error(MIMRL_ERR_ARG, "dropout_mlp must be in [0,1)");
  return MIMRL_OK;
}

int build_layout(const mimrl_cfg& c, Layout* out) {
  MX(validate_cfg(c));
  out->entries.clear();
  out->index.clear();
  out->floats[0] = out->floats[1] = 0;
  auto add = [&](const std::string& name, int d0, int d1, int d2 = 0) {
    LayoutEntry e;
    e.name = name;
    e.ndim = d2 > 0 ? 3 : d1 > 0 ? 2 : 1;
    e.d0 = d0;
    e.d1 = d1 > 0 ? d1 : 0;
    e.d2 = d2 > 0 ? d2 : 0;
    e.group = (name.find("vmi") != std::string::npos || name.find("vcmi") != std::string::npos) ? MIMRL_GROUP_CRITIC
                                                                                                : MIMRL_GROUP_MAIN;
    e.offset = -1;                                   // assigned below, in two passes (the layer-0 recurrence tensors last)
    out->index[name] = (int)out->entries.size();
    out->entries.push_back(e);
  };
  const int D = c.d_common, H = D;
  if (c.encoder == MIMRL_ENCODER_CONV) {   // Model.py:247-249: Conv1d(d, d_common, kernel 3, padding 1), weight [out, in, 3]
    add("conv_a.weight", D, c.d_a, 3); add("conv_a.bias", D, 0);
    add("conv_v.weight", D, c.d_v, 3); add("conv_v.bias", D, 0);
  }
  const struct { const char* nm; int d; } mods[2] = {{"rnn_v", c.d_v}, {"rnn_a", c.d_a}};
  for (auto& m : mods)
    for (int layer = 0; layer < (c.encoder == MIMRL_ENCODER_GRU ? 2 : c.encoder == MIMRL_ENCODER_LSTM ? 1 : 0); ++layer) {
      const int din = layer == 0 ? m.d : 2 * H;
      const int ng = c.encoder == MIMRL_ENCODER_LSTM ? 4 : 3;      // gates: LSTM i,f,g,o / GRU r,z,n
      for (const char* sfx : {"", "_reverse"}) {
        const std::string l = "_l" + std::to_string(layer) + sfx;
        add(std::string(m.nm) + ".weight_ih" + l, ng * H, din);
        add(std::string(m.nm) + ".weight_hh" + l, ng * H, H);
        add(std::string(m.nm) + ".bias_ih" + l, ng * H, 0);
        add(std::string(m.nm) + ".bias_hh" + l, ng * H, 0);
      }
    }
  add("ln_a.weight", D, 0); add("ln_a.bias", D, 0); add("ln_v.weight", D, 0); add("ln_v.bias", D, 0);
  add("W_t.weight", D, c.d_t);
  int din[3] = {c.time_len, 3, D};
  const char ax_name[3] = {'l', 'k', 'd'};
  for (int i = 0; i < c.n_blocks; ++i) {
    const std::string pre = "mlp_encoder.layers_stack." + std::to_string(i);
    for (int ax = 0; ax < 3; ++ax) {
      const std::string m = pre + ".mlp_" + ax_name[ax];
      add(m + ".fc1.weight", c.d_hiddens[i][ax], din[ax]);
      if (c.bias) add(m + ".fc1.bias", c.d_hiddens[i][ax], 0);
      add(m + ".fc2.weight", c.d_outs[i][ax], c.d_hiddens[i][ax]);
      if (c.bias) add(m + ".fc2.bias", c.d_outs[i][ax], 0);
    }
    for (int ax = 0; ax < 3; ++ax) {
      const int n = c.ln_first ? din[ax] : c.d_outs[i][ax];
      add(pre + ".ln_" + ax_name[ax] + ".weight", n, 0);
      add(pre + ".ln_" + ax_name[ax] + ".bias", n, 0);
    }
    if (c.res_project[i])
      for (int ax = 0; ax < 3; ++ax) add(pre + ".res_projection_" + ax_name[ax] + ".weight", c.d_outs[i][ax], din[ax]);
    for (int ax = 0; ax < 3; ++ax) din[ax] = c.d_outs[i][ax];
  }
  add("classifier.0.weight", 1, din[2]);
  add("classifier.0.bias", 1, 0);
  const int hid = 256, emb = 128;   // Model.py:285
  const int idx4[4] = {0, 2, 4, 6};
  for (const char* n : kVmi) {
    const std::string pre = std::string("vmi_estimator_") + n + ".critic_model";
    if (c.critic_type == MIMRL_CRITIC_SEPARATE) {
      const int dims[4][2] = {{hid, D}, {hid, hid}, {hid, hid}, {emb, hid}};
      for (const char* tw : {"MLP_g", "MLP_h"})
        for (int l = 0; l < 4; ++l) {
          const std::string p = pre + "." + tw + "." + std::to_string(idx4[l]);
          add(p + ".weight", dims[l][0], dims[l][1]);
          add(p + ".bias", dims[l][0], 0);
        }
    } else {
      const int dims[4][2] = {{hid, 2 * D}, {hid, hid}, {hid, hid}, {1, hid}};
      for (int l = 0; l < 4; ++l) {
        const std::string p = pre + ".MLP_f." + std::to_string(idx4[l]);
        add(p + ".weight", dims[l][0], dims[l][1]);
        add(p + ".bias", dims[l][0], 0);
      }
    }
  }
  if (c.baseline_type == MIMRL_BASELINE_UNNORMALIZED)   // VMI.py:82-84: mlps(128, 256, 1, 2); behind all critics (towers stay strided)
    for (const char* n : kVmi) {
      const std::string pre = std::string("vmi_estimator_") + n + ".baseline_model.MLP";
      const int dims[4][2] = {{256, D}, {256, 256}, {256, 256}, {1, 256}};
      const int idx[4] = {0, 2, 4, 6};
      for (int l = 0; l < 4; ++l) {
        add(pre + "." + std::to_string(idx[l]) + ".weight", dims[l][0], dims[l][1]);
        add(pre + "." + std::to_string(idx[l]) + ".bias", dims[l][0], 0);
      }
    }
  for (const char* n : kVcmi) {
    const std::string pre = std::string("vcmi_estimator_") + n + ".classifier.mlp";
    const int dims[4][2] = {{hid, 3 * emb}, {hid, hid}, {hid, hid}, {2, hid}};
    for (int l = 0; l < 4; ++l) {
      const std::string p = pre + "." + std::to_string(idx4[l]);
      add(p + ".weight", dims[l][0], dims[l][1]);
      add(p + ".bias", dims[l][0], 0);
    }
  }
  // Offsets.  The ENTRY order above is the reference's registration order (state_dict / optimizer order); the OFFSETS put the layer-0
  // recurrence tensors (rnn_*.*_l0*: the gradients that become final LAST in the backward pass, behind the layer-0 BPTT) at the tail
  // of the main bucket, so that under data parallelism everything in front of `late_offset` is ONE contiguous range that can be
  // all-reduced while the layer-0 BPTT still runs, and the tail a second one (round 5; rounds 3-4 had five ranges).
  for (int pass = 0; pass < 2; ++pass)
    for (auto& e : out->entries) {
      if ((layout_is_late(e.name) ? 1 : 0) != pass) continue;
      e.offset = out->floats[e.group];
      out->floats[e.group] += (e.numel() + 63) / 64 * 64;
    }
  out->late_offset = out->floats[MIMRL_GROUP_MAIN];
  for (const auto& e : out->entries)
    if (layout_is_late(e.name) && e.offset < out->late_offset) out->late_offset = e.offset;
  return MIMRL_OK;
}

bool layout_is_late(const std::string& name) {
  return name.compare(0, 4, "rnn_") == 0 && name.find("_l0") != std::string::npos;
}

}  // namespace mimrl

// ------------------------------------------------------------------------------------------------- C ABI (host only)
extern "C" {

int mimrl_layout_count(const mimrl_cfg* cfg) {
  if (!cfg) return mimrl::set_error(MIMRL_ERR_ARG, "null cfg");
  mimrl::Layout L;
  const int r = mimrl::build_layout(*cfg, &L);
  return r != 0 ? r : (int)L.entries.size();
}

int mimrl_layout_entry(const mimrl_cfg* cfg, int idx, char* name, int name_cap, int* group, int64_t* offset, int* ndim,
                       int* dim0, int* dim1) {
  if (!cfg || !name || name_cap < 2) return mimrl::set_error(MIMRL_ERR_ARG, "bad arguments");
  mimrl::Layout L;
  const int r = mimrl::build_layout(*cfg, &L);
  if (r != 0) return r;
  if (idx < 0 || idx >= (int)L.entries.size()) return mimrl::set_error(MIMRL_ERR_ARG, "layout index out of range");
  const auto& e = L.entries[idx];
  std::strncpy(name, e.name.c_str(), name_cap - 1);
  name[name_cap - 1] = 0;
  if (group) *group = e.group;
  if (offset) *offset = e.offset;
  if (ndim) *ndim = e.ndim;
  if (dim0) *dim0 = e.d0;
  if (dim1) *dim1 = e.d1;
  return MIMRL_OK;
}

int mimrl_layout_entry_dim2(const mimrl_cfg* cfg, int idx) {
  if (!cfg) return mimrl::set_error(MIMRL_ERR_ARG, "bad arguments");
  mimrl::Layout L;
  const int r = mimrl::build_layout(*cfg, &L);
  if (r != 0) return r;
  if (idx < 0 || idx >= (int)L.entries.size()) return mimrl::set_error(MIMRL_ERR_ARG, "layout index out of range");
  return L.entries[idx].d2;
}

int64_t mimrl_bucket_floats(const mimrl_cfg* cfg, int group) {
  if (!cfg || group < 0 || group > 1) return mimrl::set_error(MIMRL_ERR_ARG, "bad arguments");
  mimrl::Layout L;
  const int r = mimrl::build_layout(*cfg, &L);
  return r != 0 ? r : L.floats[group];
}

}  // extern "C"
