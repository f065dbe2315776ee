#!/bin/bash
# GPU parity suite only.  Usage: bash scripts/gpu_tests.sh [pytest -k expression]
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
if [ -n "$1" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "$1" 2>&1 | tail -25
else
  timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25
fi
