#!/bin/bash
# Experiment: a macro-step as C band launches on alternating streams (LBM_TUNE_CHUNKS): parity, then timing
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03ab
mkdir -p $OUT
timeout -k 10 300 python scripts/experiments/chunk_check.py 2>&1 | grep -v amdgpu.ids | tee $OUT/chunk_check.txt | tail -17 || exit 1
{
for g in 1024x1024 2048x2048 8192x1024 8192x2048 8192x8192; do
  st=200; [ $g = 1024x1024 ] && st=2000; [ $g = 2048x2048 ] && st=800; [ $g = 8192x8192 ] && st=60
  echo "== single periodic $g, $st steps per run"
  timeout -k 10 200 python scripts/ab_ring.py --per-context --single --grid $g --steps $st --rounds 6 - LBM_TUNE_CHUNKS=3 LBM_TUNE_CHUNKS=4 LBM_TUNE_CHUNKS=5 LBM_TUNE_CHUNKS=7 2>&1 | tail -5
done
} | grep -v amdgpu.ids | tee $OUT/ab_chunks.txt
