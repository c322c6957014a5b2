#!/bin/bash
# Builds kernel variants for A/B timing into build/ (git-ignored, travels to the GPU box).  usage: build_variants.sh name "-DFLAG ..." ...
set -e
cd "$(dirname "$0")/.."
while [ $# -gt 0 ]; do
  name=$1; flags=$2; shift 2
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-fast-math -ffp-contract=off -mllvm -amdgpu-kernarg-preload-count=16 $flags -o build/libsbr_amd_$name.so gym_sbr2_amd/csrc/sbr_amd.hip
  echo "built build/libsbr_amd_$name.so  [$flags]"
done
