#!/bin/bash
# A/B: split-plane LDS accesses as single dwords on one base (16-bit offsets) instead of ds_read2/write2_b32 with a base per plane
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03q
mkdir -p $OUT
V=mpilattice-boltzmann_amd/lib/variants
for g in 8192x8192 4096x4096 1024x1024 8192x1024; do
  s=60; [ $g = 1024x1024 ] && s=400; [ $g = 4096x4096 ] && s=120; [ $g = 8192x1024 ] && s=200
  echo "== $g"
  timeout -k 10 300 python scripts/ab_libs.py --grid $g --steps $s --rounds 3 $V/base.so $V/ldssingle.so $V/base.so $V/ldssingle.so 2>&1 | tail -4
done | tee $OUT/ab_lds_single.txt
