#!/bin/bash
# Plane skew on a rank's partition (8192 x 1024 rows + ghost rows), one ring per process
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 LBM_P2P_TIMEOUT_MS=10000
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03ak
mkdir -p $OUT
{
for round in 1 2; do
for sk in 24 0 8 34 48 96; do
  echo "== round $round ring 8192x1024 skew $sk"
  LBM_TUNE_SKEW=$sk timeout -k 10 200 python scripts/ab_ring.py --grid 8192x1024 --steps 200 --rounds 8 - 2>&1 | tail -1
done
done
for sk in 24 0 34 96; do
  echo "== ring 8192x2048 skew $sk"
  LBM_TUNE_SKEW=$sk timeout -k 10 200 python scripts/ab_ring.py --grid 8192x2048 --steps 200 --rounds 6 - 2>&1 | tail -1
done
} | grep -v amdgpu.ids | tee $OUT/ab_skew_ring.txt
