#!/bin/bash
# Sustained and short-run A/B of the sum|u| term forms: double precision (base), compensated float (comp), plain float (flag 64)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03t
mkdir -p $OUT
V=mpilattice-boltzmann_amd/lib/variants
{
for g in 8192x8192 4096x4096 1024x1024 8192x1024; do
  s=60; [ $g = 1024x1024 ] && s=400; [ $g = 4096x4096 ] && s=120; [ $g = 8192x1024 ] && s=200
  echo "== $g, short runs"
  timeout -k 10 300 python scripts/ab_libs.py --grid $g --steps $s --rounds 3 $V/base.so $V/comp.so $V/base.so:64 $V/base.so $V/comp.so $V/base.so:64 2>&1 | tail -6
done
echo "== 8192x8192 sustained (400 steps x 12 rounds each, interleaved)"
timeout -k 10 300 python scripts/ab_libs.py --grid 8192x8192 --steps 400 --rounds 12 $V/base.so $V/comp.so $V/base.so:64 2>&1 | tail -3
} | grep -v amdgpu.ids | tee $OUT/ab_terms.txt
