#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "multi or k_step or digest or smoke or partial" 2>&1 | tail -5 &&
timeout -k 10 300 python scripts/sweep.py --grid 8192x8192 --steps 60 --rounds 3 "LBM_TUNE_MULTI_K=2" "LBM_TUNE_MULTI_K=3" "LBM_TUNE_MULTI_K=4" 2>&1 | tail -3 &&
timeout -k 10 300 python scripts/sweep.py --grid 1024x1024 --steps 2000 --rounds 3 "LBM_TUNE_MULTI_K=2" "LBM_TUNE_MULTI_K=3" "LBM_TUNE_MULTI_K=4" 2>&1 | tail -3
