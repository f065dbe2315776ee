cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
LBM_FORCE_DEVICE=0 LBM_DIST_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 4 --steps 20 --warmup 5 --reps 3 --workload 4096x4096 2> gpurun_out/b4.err | cut -c1-1800
LBM_FORCE_DEVICE=0 LBM_DIST_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 3 --steps 20 --warmup 5 --reps 3 --workload 8192x8192 2> gpurun_out/b3.err | python -c "import json,sys; d=json.load(sys.stdin); print(d['value'], d['config']['loop'], d['config']['p2p'], d['parity_check']['ok'], d['exchange_attempts'], d['launch_attempts'])"
