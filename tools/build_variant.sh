#!/bin/bash
# Builds a variant of the library for A/B runs: bash tools/build_variant.sh build_ab/name.so [-DMACRO=value ...]
# (the flags of __graft_entry__.build, plus the given defines)
set -e
OUT=$1; shift
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize \
    -fPIC -shared "$@" -Iinclude -Iraytracer_amd/csrc -o "$OUT" raytracer_amd/csrc/hip_executor.hip
