#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03p
mkdir -p $OUT
V=mpilattice-boltzmann_amd/lib/variants
timeout -k 10 400 python scripts/ab_libs.py --grid 8192x8192 --steps 60 --rounds 3 $V/base.so $V/k4_ty22_l768.so $V/k4_ty22_l1024.so $V/k4_ty22_l896.so $V/k4_ty21_l768.so $V/k4_ty22_l640.so $V/base.so 2>&1 | tail -7 | tee $OUT/ab_k4_big_blocks2.txt
