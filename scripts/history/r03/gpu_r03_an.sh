#!/bin/bash
# Band launches on alternating streams again, ONE context per process (the first measurement had several contexts share a process)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03an
mkdir -p $OUT
{
for g in 8192x1024 2048x2048 1024x1024; do
  st=200; r=8; [ $g = 1024x1024 ] && st=2000 && r=5; [ $g = 2048x2048 ] && st=800 && r=5
  for ch in 0 3 4 5 8; do
    echo "== single periodic $g, $st steps per run, LBM_TUNE_CHUNKS=$ch"
    LBM_TUNE_CHUNKS=$ch timeout -k 10 200 python scripts/ab_ring.py --single --grid $g --steps $st --rounds $r - 2>&1 | tail -1
  done
done
echo "== GPU_MAX_HW_QUEUES=8, 8192x1024: chunks 0 / 4"
GPU_MAX_HW_QUEUES=8 LBM_TUNE_CHUNKS=0 timeout -k 10 200 python scripts/ab_ring.py --single --grid 8192x1024 --steps 200 --rounds 8 - 2>&1 | tail -1
GPU_MAX_HW_QUEUES=8 LBM_TUNE_CHUNKS=4 timeout -k 10 200 python scripts/ab_ring.py --single --grid 8192x1024 --steps 200 --rounds 8 - 2>&1 | tail -1
} | grep -v amdgpu.ids | tee $OUT/ab_chunks_procs.txt
