#!/bin/bash
# Host-side code under AddressSanitizer + UBSan (CPU only): builds mimrl_amd/libmimrl_host_asan.so (make asan) and runs the CPU tests that
# drive it -- the scikit-learn KDTree tie-order restatement (108 banks up to N = 16326) and the native parameter layout -- with the
# sanitizer runtime preloaded into python.  usage: tools/asan_host.sh   (from the repo root; exit code = pytest's)
set -eu
root=$(cd "$(dirname "$0")/.." && pwd)
make -C "$root/mimrl_amd/csrc" asan > /dev/null
rt=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
cd "$root"
# detect_leaks=0: CPython itself "leaks" at exit; link order: the runtime must come first in the process
LD_PRELOAD="$rt" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
  MIMRL_LIB_PATH="$root/mimrl_amd/libmimrl_host_asan.so" \
  python -m pytest tests/test_knn_ties.py tests/test_layout.py -q -p no:cacheprovider -k "host_knn or native_layout or parameter_counts or bad_config" "$@"
