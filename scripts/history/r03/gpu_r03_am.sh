#!/bin/bash
# Edge-first schedule (edge launch, then interior launch on ONE stream; the push hides behind the interior launch): parity, then timing, one ring per process
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 LBM_P2P_TIMEOUT_MS=10000
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03am
mkdir -p $OUT
timeout -k 10 300 python scripts/experiments/sched_check.py edgefirst 2>&1 | grep -v amdgpu.ids | tee $OUT/sched_check.txt | tail -10
grep -q "^bad 0" $OUT/sched_check.txt || exit 1
{
for round in 1 2; do
for g in 8192x1024 8192x2048; do
  for sc in edge edgefirst; do
    echo "== round $round ring $g schedule $sc: 200 steps per run, then 20"
    LBM_P2P_SCHEDULE=$sc timeout -k 10 200 python scripts/ab_ring.py --grid $g --steps 200 --rounds 8 - 2>&1 | tail -1
    LBM_P2P_SCHEDULE=$sc timeout -k 10 200 python scripts/ab_ring.py --grid $g --steps 20 --rounds 40 - 2>&1 | tail -1
  done
done
done
} | grep -v amdgpu.ids | tee $OUT/ab_edgefirst.txt
