#!/bin/bash
# Round-3 session G: randomised cross-check of every kernel family and both partitioned loops (new: sweep kernel, four ghost rows).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03g
mkdir -p $OUT
timeout -k 10 1000 python scripts/fuzz_kernels.py --cases 350 --seed 31 > $OUT/fuzz.log 2>&1
echo "rc=$?"; tail -5 $OUT/fuzz.log; grep -c MISMATCH $OUT/fuzz.log
