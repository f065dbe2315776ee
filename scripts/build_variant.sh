#!/bin/bash
# Build the CURRENT source tree's liblbm_d2q9.so as lib/variants/<name>.so (for scripts/ab_libs.py).
# Usage: bash scripts/build_variant.sh <name> [extra hipcc flags, e.g. -DLBM_EXPERIMENT=1]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
OUT=$ROOT/mpilattice-boltzmann_amd/lib/variants
mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -pthread -I $ROOT/include -I $ROOT/mpilattice-boltzmann_amd/csrc \
  -fPIC -shared "$@" $ROOT/mpilattice-boltzmann_amd/csrc/lbm_kernels.hip $ROOT/mpilattice-boltzmann_amd/csrc/lbm_host.cpp -o $OUT/$NAME.so
echo $OUT/$NAME.so
