#!/bin/bash
# Skew 24 against 34 (old default) and 0 at other sizes and kernels
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03aj
mkdir -p $OUT
L=mpilattice-boltzmann_amd/lib/liblbm_d2q9.so
{
for g in 4096x4096 2048x2048 1024x1024 8192x1024 8192x2048 8192x4096 16384x4096 4096x8192; do
  s=120; [ $g = 1024x1024 ] && s=400; [ $g = 2048x2048 ] && s=400; [ $g = 8192x4096 ] && s=60; [ $g = 16384x4096 ] && s=40; [ $g = 4096x8192 ] && s=60
  echo "== $g K = 4: skew 34 / 24 / 0 / 34 / 24"
  timeout -k 10 300 python scripts/ab_libs.py --grid $g --steps $s --rounds 3 $L::LBM_TUNE_SKEW=34 $L::LBM_TUNE_SKEW=24 $L::LBM_TUNE_SKEW=0 $L::LBM_TUNE_SKEW=34 $L::LBM_TUNE_SKEW=24 2>&1 | tail -5
done
echo "== 8192x8192 one-step kernel: skew 34 / 24 / 0 / 34 / 24 / 0"
timeout -k 10 300 python scripts/ab_libs.py --grid 8192x8192 --steps 30 --rounds 3 --env LBM_TUNE_MULTI_K=0 $L::LBM_TUNE_SKEW=34 $L::LBM_TUNE_SKEW=24 $L::LBM_TUNE_SKEW=0 $L::LBM_TUNE_SKEW=34 $L::LBM_TUNE_SKEW=24 $L::LBM_TUNE_SKEW=0 2>&1 | tail -6
} | grep -v amdgpu.ids | tee $OUT/ab_skew_sizes.txt
