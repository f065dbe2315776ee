#!/bin/bash
# Tile order of lbm_multi_kernel: row-major vs column strips S tiles wide (L2 reuse of the top / bottom ring rows)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03w
mkdir -p $OUT
V=mpilattice-boltzmann_amd/lib/variants
{
for g in 8192x8192 4096x4096 1024x1024; do
  s=60; [ $g = 1024x1024 ] && s=400; [ $g = 4096x4096 ] && s=120
  echo "== $g"
  timeout -k 10 300 python scripts/ab_libs.py --grid $g --steps $s --rounds 3 $V/strip.so $V/strip.so::LBM_TUNE_MULTI_STRIP=4 $V/strip.so::LBM_TUNE_MULTI_STRIP=8 $V/strip.so::LBM_TUNE_MULTI_STRIP=16 $V/strip.so::LBM_TUNE_MULTI_STRIP=32 $V/strip.so 2>&1 | tail -6
done
} | grep -v amdgpu.ids | tee $OUT/ab_strip.txt
