cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 900 python scripts/fuzz_kernels.py --cases 400 --seed 5 2>&1 | tail -8
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "fast_av_vels or randomised_kernel" 2>&1 | tail -3
