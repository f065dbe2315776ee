#!/bin/bash
# 1024 x 1024 (the shipped deck's size): tall (720 tiles on 512 slots) against 64 x 16 tiles, two blocks per CU (1024 tiles = two full rounds)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03ao
mkdir -p $OUT
V=mpilattice-boltzmann_amd/lib/variants
{
for g in 1024x1024 1536x1536 2048x2048 1024x2048; do
  echo "== $g: tall / std 64x13 / 64x16 two per CU (512 lanes) / 64x16 (768 lanes; K<=3 too)"
  timeout -k 10 300 python scripts/ab_libs.py --grid $g --steps 800 --rounds 3 $V/base.so $V/base.so::LBM_TUNE_MULTI_GEOM=0 $V/mid16.so::LBM_TUNE_MULTI_GEOM=0 $V/mid16l768.so::LBM_TUNE_MULTI_GEOM=0 $V/base.so $V/mid16.so::LBM_TUNE_MULTI_GEOM=0 2>&1 | tail -6
done
} | grep -v amdgpu.ids | tee $OUT/ab_mid16.txt
