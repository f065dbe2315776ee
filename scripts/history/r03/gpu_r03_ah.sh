#!/bin/bash
# Plane skew again, on the tall geometry (and what it does to the other kernels)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03ah
mkdir -p $OUT
L=mpilattice-boltzmann_amd/lib/liblbm_d2q9.so
{
echo "== 8192x8192 tall: skew 34 / 0 / 34 / 0 / 33 / 35 / 34 / 0"
timeout -k 10 300 python scripts/ab_libs.py --grid 8192x8192 --steps 60 --rounds 3 $L $L::LBM_TUNE_SKEW=0 $L $L::LBM_TUNE_SKEW=0 $L::LBM_TUNE_SKEW=33 $L::LBM_TUNE_SKEW=35 $L $L::LBM_TUNE_SKEW=0 2>&1 | tail -8
echo "== 4096x4096 tall: skew 34 / 0 / 34 / 0 / 17 / 68"
timeout -k 10 300 python scripts/ab_libs.py --grid 4096x4096 --steps 120 --rounds 3 $L $L::LBM_TUNE_SKEW=0 $L $L::LBM_TUNE_SKEW=0 $L::LBM_TUNE_SKEW=17 $L::LBM_TUNE_SKEW=68 2>&1 | tail -6
echo "== 8192x1024 (a rank's rows, one periodic launch): skew 34 / 0 / 34 / 0"
timeout -k 10 300 python scripts/ab_libs.py --grid 8192x1024 --steps 200 --rounds 3 $L $L::LBM_TUNE_SKEW=0 $L $L::LBM_TUNE_SKEW=0 2>&1 | tail -4
echo "== 2048x2048: skew 34 / 0 / 34 / 0"
timeout -k 10 300 python scripts/ab_libs.py --grid 2048x2048 --steps 400 --rounds 3 $L $L::LBM_TUNE_SKEW=0 $L $L::LBM_TUNE_SKEW=0 2>&1 | tail -4
echo "== 8192x8192 one-step kernel (LBM_TUNE_MULTI_K=0): skew 34 / 0 / 34 / 0"
timeout -k 10 300 python scripts/ab_libs.py --grid 8192x8192 --steps 30 --rounds 3 --env LBM_TUNE_MULTI_K=0 $L $L::LBM_TUNE_SKEW=0 $L $L::LBM_TUNE_SKEW=0 2>&1 | tail -4
echo "== 8192x8192 K = 3 (64 x 16): skew 34 / 0 / 34 / 0"
timeout -k 10 300 python scripts/ab_libs.py --grid 8192x8192 --steps 60 --rounds 3 --env LBM_TUNE_MULTI_K=3 $L $L::LBM_TUNE_SKEW=0 $L $L::LBM_TUNE_SKEW=0 2>&1 | tail -4
} | grep -v amdgpu.ids | tee $OUT/ab_skew_tall.txt
