#!/bin/bash
# The three forms of the sum|u| terms again, now on the tall geometry (and the standard one, forced): is compensated still worth it
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03ac
mkdir -p $OUT
L=mpilattice-boltzmann_amd/lib/liblbm_d2q9.so
{
echo "== 8192x8192 tall geometry, short runs: double / compensated / float, three positions each"
timeout -k 10 300 python scripts/ab_libs.py --grid 8192x8192 --steps 60 --rounds 3 $L:128 $L $L:64 $L:128 $L $L:64 $L:128 $L $L:64 2>&1 | tail -9
echo "== 8192x8192 tall geometry, sustained (400 steps x 10 rounds, interleaved)"
timeout -k 10 300 python scripts/ab_libs.py --grid 8192x8192 --steps 400 --rounds 10 $L:128 $L $L:64 2>&1 | tail -3
echo "== 8192x8192 standard geometry (LBM_TUNE_MULTI_GEOM=0), sustained"
timeout -k 10 300 python scripts/ab_libs.py --grid 8192x8192 --steps 400 --rounds 10 --env LBM_TUNE_MULTI_GEOM=0 $L:128 $L $L:64 2>&1 | tail -3
echo "== 8192x8192 tall vs standard, compensated, sustained"
timeout -k 10 300 python scripts/ab_libs.py --grid 8192x8192 --steps 400 --rounds 10 $L $L::LBM_TUNE_MULTI_GEOM=0 $L:128 $L:128:LBM_TUNE_MULTI_GEOM=0 2>&1 | tail -4
echo "== 4096x4096 short: tall double / comp / float, standard double / comp"
timeout -k 10 300 python scripts/ab_libs.py --grid 4096x4096 --steps 120 --rounds 3 $L:128 $L $L:64 $L:128:LBM_TUNE_MULTI_GEOM=0 $L::LBM_TUNE_MULTI_GEOM=0 $L:128 $L 2>&1 | tail -7
} | grep -v amdgpu.ids | tee $OUT/ab_terms_tall.txt
