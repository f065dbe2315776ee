#!/bin/bash
# Tile order: tiles of 2 / 3 / 4 consecutive tile rows taken column by column (the tile below follows at once: its ring rows hit L2)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03al
mkdir -p $OUT
V=mpilattice-boltzmann_amd/lib/variants
{
for g in 8192x8192 4096x4096; do
  s=60; [ $g = 4096x4096 ] && s=120
  echo "== $g: row-major / groups of 2 / 3 / 4 / row-major / 2"
  timeout -k 10 300 python scripts/ab_libs.py --grid $g --steps $s --rounds 3 $V/rowgroup.so $V/rowgroup.so::LBM_TUNE_MULTI_ROWGROUP=2 $V/rowgroup.so::LBM_TUNE_MULTI_ROWGROUP=3 $V/rowgroup.so::LBM_TUNE_MULTI_ROWGROUP=4 $V/rowgroup.so $V/rowgroup.so::LBM_TUNE_MULTI_ROWGROUP=2 2>&1 | tail -6
done
} | grep -v amdgpu.ids | tee $OUT/ab_rowgroup.txt
