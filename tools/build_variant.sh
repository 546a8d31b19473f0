#!/bin/bash
# Builds a variant of libsfhip.so under build/ for A/B timing on the GPU box (tools/ab.sh):
# usage: tools/build_variant.sh <name> [extra hipcc flags, e.g. -DSFHIP_LANE_SHIFT_BPERMUTE]
set -eu
name=$1; shift
cd "$(dirname "$0")/.."
mkdir -p build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 -Iinclude "$@" \
    sailfish_amd/csrc/sfhip.hip -o "build/$name"
echo "built build/$name"
