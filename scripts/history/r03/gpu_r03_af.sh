#!/bin/bash
# Fused p2p schedule (one launch per macro-step, push kernel waits for the edge blocks' count): parity, then timing against the edge-stream schedule
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 LBM_P2P_TIMEOUT_MS=10000
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03af
mkdir -p $OUT
timeout -k 10 300 python scripts/experiments/fused_check.py 2>&1 | grep -v amdgpu.ids | tee $OUT/fused_check.txt | tail -14
grep -q "^bad 0" $OUT/fused_check.txt || exit 1
{
for g in 8192x1024 8192x2048 8192x4096; do
  for st in 20 200; do
    r=40; [ $st = 200 ] && r=8
    echo "== ring $g, $st steps per run"
    timeout -k 10 200 python scripts/ab_ring.py --per-context --grid $g --steps $st --rounds $r LBM_P2P_SCHEDULE=edge LBM_P2P_SCHEDULE=fused 2>&1 | tail -2
  done
  echo "== single periodic $g (standard geometry), 200 steps per run"
  timeout -k 10 200 python scripts/ab_ring.py --per-context --single --grid $g --steps 200 --rounds 8 LBM_TUNE_MULTI_GEOM=0 2>&1 | tail -1
done
} | grep -v amdgpu.ids | tee $OUT/ab_fused.txt
