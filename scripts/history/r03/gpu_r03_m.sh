#!/bin/bash
# Round-3 session M: randomised cross-check with the K = 4 defaults, two seeds; maximum-size check.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03m
mkdir -p $OUT
timeout -k 10 500 python scripts/fuzz_kernels.py --cases 300 --seed 77 > $OUT/fuzz77.log 2>&1; echo "rc=$?"; tail -2 $OUT/fuzz77.log
timeout -k 10 500 python scripts/fuzz_kernels.py --cases 120 --seed 5 --scale 3 > $OUT/fuzz5.log 2>&1; echo "rc=$?"; tail -2 $OUT/fuzz5.log
timeout -k 10 400 python scripts/big_grid_check.py > $OUT/big.log 2>&1; echo "rc=$?"; tail -3 $OUT/big.log
