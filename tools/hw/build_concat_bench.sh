#!/bin/bash
# builds tools/hw/concat_ws_bench[<suffix>] (in-tree, git-ignored: travels to the GPU box with gpurun): build_concat_bench.sh [suffix] [-D flags]
sfx=$1; shift
cd "$(dirname "$0")/../.." && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -I mimrl_amd/csrc -I include tools/hw/concat_ws_bench.hip mimrl_amd/csrc/concat_fused.hip \
  mimrl_amd/csrc/concat_ws.hip mimrl_amd/csrc/concat_ws_bwd.hip tools/hw/concat_ws4.hip mimrl_amd/csrc/errors.cpp mimrl_amd/csrc/knobs.cpp "$@" -o tools/hw/concat_ws_bench$sfx 2>&1 | grep -E "error|undefined" ; ls -la tools/hw/concat_ws_bench$sfx
