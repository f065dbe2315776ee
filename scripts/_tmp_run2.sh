export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -m gpu -q -k "separate_processes or partitions_in_one_process or bench_self or several_ranks" > gpurun_out/t4.log 2>&1; tail -8 gpurun_out/t4.log
