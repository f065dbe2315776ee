#!/bin/bash
# Round-3 session H: does a shorter tile (three blocks per CU) help the K = 4 launch ?
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03h
mkdir -p $OUT
V=mpilattice-boltzmann_amd/lib/variants
timeout -k 10 600 python scripts/ab_libs.py --grid 8192x8192 --steps 60 --rounds 3 $V/base.so::LBM_TUNE_MULTI_K=4 $V/ty12w6.so::LBM_TUNE_MULTI_K=4 $V/ty10w8.so::LBM_TUNE_MULTI_K=4 $V/base.so $V/base.so::LBM_TUNE_MULTI_K=4 2>&1 | tail -6 | tee $OUT/ab_k4_short_tiles.txt
