// One table of every environment knob the native library reads (round 5, VERDICT r04 item 8: 61 getenv sites in the then one-file engine alone, 84
// distinct names over the library).  Every site calls knob("NAME") -- getenv for a REGISTERED name (an unregistered one is reported once on
// stderr: a typo in a tuning script no longer silently measures the default) -- and MIMRL_KNOBS=1 prints the table with the values in
// effect when a handle is created.  Knobs are tuning / debugging switches: the defaults are what bench.py measures, none changes a
// result beyond float summation order unless its line says so.  Result-changing DEBUG knobs (MIMRL_DBG_*, common.h: dbg_env) exist only
// in `make DEBUG_KNOBS=1` builds and are listed by kDebugKnobs.  (The Python side reads MIMRL_DETERMINISTIC, MIMRL_LIB_PATH,
// MIMRL_DIST_BACKEND, MIMRL_DDP_TORCH, MIMRL_DDP_DEFERRED_TAIL, MIMRL_DDP_FORCE_COLLECTIVES: mimrl_amd/_lib.py, dist.py.)
#pragma once
#include <cstdio>

namespace mimrl {

const char* knob(const char* name);                 // value in the environment, or nullptr; `name` must be in the table
inline bool knob_on(const char* name) { const char* v = knob(name); return v != nullptr && !(v[0] == '0' && v[1] == 0); }
int knob_int(const char* name, int dflt);
void knobs_print(FILE* f);                          // the table: name, value in effect (or "-"), where it is read, what it does

}  // namespace mimrl
