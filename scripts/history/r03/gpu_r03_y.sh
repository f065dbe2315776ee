#!/bin/bash
# K = 4 on taller tiles with 768 - 896-lane blocks, two per CU (with the compensated sum|u| terms): which geometry
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03y
mkdir -p $OUT
V=mpilattice-boltzmann_amd/lib/variants
ALL="$V/base.so $V/k4_ty20_l768.so $V/k4_ty21_l768.so $V/k4_ty22_l768.so $V/k4_ty23_l768.so $V/k4_ty19_l832.so $V/k4_ty20_l832.so $V/k4_ty20_l896.so $V/k4_ty22_l896.so $V/k4_ty23_l896.so $V/base.so"
{
echo "== 8192x8192 short"
timeout -k 10 400 python scripts/ab_libs.py --grid 8192x8192 --steps 60 --rounds 3 $ALL 2>&1 | tail -11
echo "== 4096x4096 short"
timeout -k 10 300 python scripts/ab_libs.py --grid 4096x4096 --steps 120 --rounds 3 $ALL 2>&1 | tail -11
echo "== 8192x1024 short"
timeout -k 10 300 python scripts/ab_libs.py --grid 8192x1024 --steps 200 --rounds 3 $ALL 2>&1 | tail -11
echo "== 2048x2048 short"
timeout -k 10 300 python scripts/ab_libs.py --grid 2048x2048 --steps 400 --rounds 3 $ALL 2>&1 | tail -11
echo "== 1024x1024 short"
timeout -k 10 300 python scripts/ab_libs.py --grid 1024x1024 --steps 400 --rounds 3 $ALL 2>&1 | tail -11
} | grep -v amdgpu.ids | tee $OUT/ab_big_blocks_matrix.txt
