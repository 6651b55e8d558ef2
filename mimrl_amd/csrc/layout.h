// Parameter inventory / flat-bucket layout (mirror of mimrl_amd/layout.py; names = reference state_dict keys).
#pragma once
#include <map>
#include <string>
#include <vector>

#include "../../include/mimrl.h"

namespace mimrl {

struct LayoutEntry {
  std::string name;
  int ndim = 1;
  int d0 = 0, d1 = 0, d2 = 0;   // [d0], [d0,d1] or [d0,d1,d2]
  int group = MIMRL_GROUP_MAIN;
  long offset = 0;        // floats, 64-aligned
  long numel() const { return ndim == 3 ? (long)d0 * d1 * d2 : ndim == 2 ? (long)d0 * d1 : d0; }
};

struct Layout {
  std::vector<LayoutEntry> entries;
  std::map<std::string, int> index;
  long floats[2] = {0, 0};
  long late_offset = 0;   // main bucket: floats [late_offset, floats[MAIN]) are the layer-0 recurrence tensors (final last in the backward pass)
  const LayoutEntry* find(const std::string& n) const {
    auto it = index.find(n);
    return it == index.end() ? nullptr : &entries[it->second];
  }
};

// returns 0 or MIMRL_ERR_ARG (message set)
int build_layout(const mimrl_cfg& c, Layout* out);
int validate_cfg(const mimrl_cfg& c);
bool layout_is_late(const std::string& name);   // rnn_*.*_l0*: laid out at the tail of the main bucket

}  // namespace mimrl
