// The two C-ABI entry points of engine_abi.hip that the HOST-ONLY sanitizer library needs (`make asan`: layout.cpp + errors.cpp + knn_r1.cpp
// + this file, no device code).  Never linked into libmimrl_hip.so, which gets them from engine_abi.hip.
#include "common.h"
extern "C" const char* mimrl_last_error(void) { return mimrl::last_error_slot().c_str(); }
extern "C" int mimrl_abi_version(void) { return MIMRL_ABI_VERSION; }
