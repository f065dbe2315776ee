cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -k "cli_drives or p2p_partitions_in_one" > $OUT/t7.log 2>&1; tail -6 $OUT/t7.log
