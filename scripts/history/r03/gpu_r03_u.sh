#!/bin/bash
# Three forms of the sum|u| terms in one build (LBM_TUNE_TERMS 0 double / 1 float / 2 compensated = default), then the suite
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03u
mkdir -p $OUT
L=mpilattice-boltzmann_amd/lib/liblbm_d2q9.so
{
for g in 8192x8192 4096x4096 1024x1024 8192x1024; do
  s=60; [ $g = 1024x1024 ] && s=400; [ $g = 4096x4096 ] && s=120; [ $g = 8192x1024 ] && s=200
  echo "== $g, short runs: double / compensated / float, twice"
  timeout -k 10 300 python scripts/ab_libs.py --grid $g --steps $s --rounds 3 $L:128 $L $L:64 $L:128 $L $L:64 2>&1 | tail -6
done
echo "== 8192x8192 sustained (400 steps x 12 rounds each, interleaved)"
timeout -k 10 300 python scripts/ab_libs.py --grid 8192x8192 --steps 400 --rounds 12 $L:128 $L $L:64 2>&1 | tail -3
} | grep -v amdgpu.ids | tee $OUT/ab_terms.txt
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; tail -5 $OUT/pytest_gpu.log
