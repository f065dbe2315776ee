#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03n
mkdir -p $OUT
timeout -k 10 500 python scripts/big_grid_check.py > $OUT/big.log 2>&1; echo "rc=$?"; tail -4 $OUT/big.log
V=mpilattice-boltzmann_amd/lib/variants
timeout -k 10 300 python scripts/ab_libs.py --grid 8192x8192 --steps 60 --rounds 3 $V/base.so $V/base.so::LBM_TUNE_MULTI_REMAP=0 $V/base.so:64 $V/base.so 2>&1 | tail -4 | tee $OUT/ab_remap.txt
