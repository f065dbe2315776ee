#!/bin/bash
# One ring per PROCESS (two rings in one process may share hardware queues): schedules edge / fused, geometry standard / tall
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 LBM_P2P_TIMEOUT_MS=10000
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03ag
mkdir -p $OUT
{
for round in 1 2; do
for g in 8192x1024 8192x2048; do
  for cfg in "edge 0" "fused 0" "edge 2" "fused 2"; do
    set -- $cfg
    echo "== round $round ring $g schedule $1 geometry $2"
    LBM_P2P_SCHEDULE=$1 LBM_TUNE_MULTI_GEOM=$2 timeout -k 10 200 python scripts/ab_ring.py --grid $g --steps 200 --rounds 8 - 2>&1 | tail -1
    LBM_P2P_SCHEDULE=$1 LBM_TUNE_MULTI_GEOM=$2 timeout -k 10 200 python scripts/ab_ring.py --grid $g --steps 20 --rounds 40 - 2>&1 | tail -1
  done
done
done
} | grep -v amdgpu.ids | tee $OUT/ab_fused_procs.txt
