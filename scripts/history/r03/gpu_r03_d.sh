#!/bin/bash
# Round-3 session D: short tiles (four blocks per CU) A/B; ring A/B of the polling wait.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03d
mkdir -p $OUT
V=mpilattice-boltzmann_amd/lib/variants
timeout -k 10 600 python scripts/ab_libs.py --grid 8192x8192 --steps 60 --rounds 3 $V/base.so $V/ty10w8.so $V/ty11w8.so $V/ty12w6.so $V/base.so 2>&1 | tail -6 | tee $OUT/ab_short_tiles.txt
timeout -k 10 300 python scripts/ab_ring.py --grid 8192x1024 --steps 20 --rounds 60 LBM_SPIN_WAIT_US=0 LBM_SPIN_WAIT_US=4000 2>&1 | tail -3 | tee $OUT/ab_ring_spin.txt
