#!/bin/bash
# ODDX without the spill (double / float terms: 79 VGPRs) against 64 x 23 with the same terms
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03ae
mkdir -p $OUT
V=mpilattice-boltzmann_amd/lib/variants
{
for g in 8192x8192 4096x4096; do
  s=60; [ $g = 4096x4096 ] && s=120
  echo "== $g"
  timeout -k 10 300 python scripts/ab_libs.py --grid $g --steps $s --rounds 3 $V/tall23.so:128 $V/oddx25.so:128 $V/oddx24.so:128 $V/tall23.so:64 $V/oddx25.so:64 $V/oddx24.so:64 $V/tall23.so:128 $V/oddx24.so:128 2>&1 | tail -8
done
} | grep -v amdgpu.ids | tee $OUT/ab_oddx_nospill.txt
