#!/bin/bash
# A/B tooling: builds csrc/ble_kernels.hip with extra flags into build_ab/libble_<name>.so (git-ignored,
# travels to the GPU box).  Usage: bash profiles/build_variant.sh <name> [hipcc flags...]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
mkdir -p $ROOT/build_ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on -fPIC -shared -I $ROOT/balloon_learning_environment_amd/csrc "$@" -o $ROOT/build_ab/libble_$NAME.so \
    $ROOT/balloon_learning_environment_amd/csrc/ble_kernels.hip
echo built build_ab/libble_$NAME.so
