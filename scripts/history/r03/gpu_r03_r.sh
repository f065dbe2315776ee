#!/bin/bash
# A/B: base (paired LDS accesses) / single-dword LDS accesses / + one-compare pair flags + sqrt fixed-point select behind a uniform branch
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03r
mkdir -p $OUT
V=mpilattice-boltzmann_amd/lib/variants
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off scripts/experiments/sqrt_exhaustive.hip -o /tmp/sq 2>/dev/null && timeout -k 10 300 /tmp/sq | tee $OUT/sqrt_exhaustive.txt || exit 1
for g in 8192x8192 4096x4096 1024x1024 8192x1024; do
  s=60; [ $g = 1024x1024 ] && s=400; [ $g = 4096x4096 ] && s=120; [ $g = 8192x1024 ] && s=200
  echo "== $g"
  timeout -k 10 300 python scripts/ab_libs.py --grid $g --steps $s --rounds 3 $V/base.so $V/ldssingle.so $V/cur.so $V/base.so $V/ldssingle.so $V/cur.so 2>&1 | tail -6
done | tee $OUT/ab_flags_sqrt.txt
timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "not bench and not exhaustive" 2>&1 | tail -3
