#!/bin/bash
# build_variant.sh <suffix> <source.hip> [-Dflags...]: rejit_amd/librejit_hip_<suffix>.so = the current objects with ONE source
# compiled again under extra flags (A/B builds for tools/probes/*_ab.sh; the variants are scratch, git-ignored, and travel to the GPU box)
set -e
cd "$(dirname "$0")/../.."
suffix=$1; src=$2; shift 2
obj=rejit_amd/build/variant_${suffix}_$(basename "$src" .hip).o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -pthread "$@" -c rejit_amd/csrc/$src -o $obj
objs=$(ls rejit_amd/build/*.o | grep -v "/variant_" | grep -v "/$(basename "$src" .hip).o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o rejit_amd/librejit_hip_${suffix}.so $objs $obj
echo built rejit_amd/librejit_hip_${suffix}.so
