#!/bin/bash
# Plane skew scan for K = 4 tall and K = 3 (tails) at 8192 x 8192
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03ai
mkdir -p $OUT
L=mpilattice-boltzmann_amd/lib/liblbm_d2q9.so
SK="0 2 4 8 12 16 17 24 33 34 40 48 68"
{
A=""; for s in $SK; do A="$A $L::LBM_TUNE_SKEW=$s"; done
echo "== 8192x8192 K = 4 tall, skews $SK, then 34 and 0 again"
timeout -k 10 400 python scripts/ab_libs.py --grid 8192x8192 --steps 40 --rounds 3 $A $L::LBM_TUNE_SKEW=34 $L::LBM_TUNE_SKEW=0 2>&1 | tail -15
B=""; for s in $SK; do B="$B $L::LBM_TUNE_SKEW=$s,LBM_TUNE_MULTI_K=3"; done
echo "== 8192x8192 K = 3 (64 x 16), same skews, then 34 and 0 again"
timeout -k 10 400 python scripts/ab_libs.py --grid 8192x8192 --steps 39 --rounds 3 $B $L::LBM_TUNE_SKEW=34,LBM_TUNE_MULTI_K=3 $L::LBM_TUNE_SKEW=0,LBM_TUNE_MULTI_K=3 2>&1 | tail -15
} | grep -v amdgpu.ids | tee $OUT/ab_skew_scan.txt
