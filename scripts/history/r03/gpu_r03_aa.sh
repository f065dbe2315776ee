#!/bin/bash
# Partitions: standard (64 x 13, 512 lanes) vs tall (64 x 23, 768 lanes) K = 4 geometry on 1-rank rings and as single periodic launches
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03aa
mkdir -p $OUT
{
for g in 8192x1024 8192x2048 8192x4096; do
  for st in 20 200; do
    r=40; [ $st = 200 ] && r=8
    echo "== ring $g, $st steps per run"
    timeout -k 10 200 python scripts/ab_ring.py --per-context --grid $g --steps $st --rounds $r LBM_TUNE_MULTI_GEOM=0 LBM_TUNE_MULTI_GEOM=2 2>&1 | tail -2
  done
  echo "== single periodic $g, 200 steps per run"
  timeout -k 10 200 python scripts/ab_ring.py --per-context --single --grid $g --steps 200 --rounds 8 LBM_TUNE_MULTI_GEOM=0 LBM_TUNE_MULTI_GEOM=2 2>&1 | tail -2
done
echo "== single 8192x8192, 20 steps per run"
timeout -k 10 200 python scripts/ab_ring.py --per-context --single --grid 8192x8192 --steps 20 --rounds 20 LBM_TUNE_MULTI_GEOM=0 LBM_TUNE_MULTI_GEOM=2 2>&1 | tail -2
} | grep -v amdgpu.ids | tee $OUT/ab_ring_geom.txt
