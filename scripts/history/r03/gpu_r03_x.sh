#!/bin/bash
# K = 4 on 64 x 22 tiles (768 / 1024 lanes, two blocks per CU: ring work 1.25 x instead of 1.36 x) again, SUSTAINED: under the power cap
# a launch that does less work per cell should gain more than it did in short runs
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03x
mkdir -p $OUT
V=mpilattice-boltzmann_amd/lib/variants
{
echo "== 8192x8192 sustained (400 steps x 10 rounds each, interleaved)"
timeout -k 10 300 python scripts/ab_libs.py --grid 8192x8192 --steps 400 --rounds 10 $V/base.so $V/k4_ty22_l768.so $V/k4_ty22_l1024.so 2>&1 | tail -3
echo "== 8192x8192 short"
timeout -k 10 300 python scripts/ab_libs.py --grid 8192x8192 --steps 60 --rounds 3 $V/base.so $V/k4_ty22_l768.so $V/k4_ty22_l1024.so $V/base.so $V/k4_ty22_l768.so $V/k4_ty22_l1024.so 2>&1 | tail -6
echo "== 8192x1024 short"
timeout -k 10 300 python scripts/ab_libs.py --grid 8192x1024 --steps 200 --rounds 3 $V/base.so $V/k4_ty22_l768.so $V/k4_ty22_l1024.so $V/base.so $V/k4_ty22_l768.so $V/k4_ty22_l1024.so 2>&1 | tail -6
echo "== 1024x1024 short"
timeout -k 10 300 python scripts/ab_libs.py --grid 1024x1024 --steps 400 --rounds 3 $V/base.so $V/k4_ty22_l768.so $V/k4_ty22_l1024.so $V/base.so $V/k4_ty22_l768.so $V/k4_ty22_l1024.so 2>&1 | tail -6
} | grep -v amdgpu.ids | tee $OUT/ab_big_blocks_sustained.txt
