#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
V=mpilattice-boltzmann_amd/lib/variants
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "multi or k_step or digest or smoke or partial" 2>&1 | tail -3 &&
timeout -k 10 300 python scripts/ab_libs.py --grid 8192x8192 --steps 60 --rounds 4 --env LBM_TUNE_MULTI_K=3 $V/base_c1.so $V/waveskip.so $V/flags.so 2>&1 | tail -4 &&
timeout -k 10 300 python scripts/ab_libs.py --grid 1024x1024 --steps 1500 --rounds 4 --env LBM_TUNE_MULTI_K=3 $V/base_c1.so $V/waveskip.so $V/flags.so 2>&1 | tail -4
