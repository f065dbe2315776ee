#!/bin/bash
# Round-3 session C: the sweep kernel — parity, then same-process A/B against lbm_multi_kernel<3> at 8192^2.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03c
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "sweep" > $OUT/pytest_sweep.log 2>&1
rc=$?
tail -30 $OUT/pytest_sweep.log
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "pytest rc=$rc: stopping"; exit $rc; fi
V=mpilattice-boltzmann_amd/lib/variants
timeout -k 10 600 python scripts/ab_libs.py --grid 8192x8192 --steps 60 --rounds 3 $V/base.so $V/base.so::LBM_TUNE_SWEEP=5 $V/base.so::LBM_TUNE_SWEEP=4 $V/base.so::LBM_TUNE_SWEEP=5,LBM_TUNE_SWEEP_BLOCKS=1536 $V/base.so::LBM_TUNE_SWEEP=5,LBM_TUNE_SWEEP_MODE=1 $V/base.so::LBM_TUNE_SWEEP=5,LBM_TUNE_SWEEP_MODE=0 $V/base.so 2>&1 | tail -8 | tee $OUT/ab_sweep4.txt
