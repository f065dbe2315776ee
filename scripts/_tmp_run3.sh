cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "p2p or several_ranks or bench_self or cli_drives" > $OUT/t14.log 2>&1; tail -3 $OUT/t14.log
python bench.py --ring --exchange p2p --workload 1024x128 --steps 3000 --warmup 30 --reps 3 --no-cpu-baseline | python -c "import json,sys; d=json.load(sys.stdin); print('ring 1024x128 p2p us/step %.2f' % (d['ms_per_step']*1e3), d['config']['p2p'], d['parity_check']['ok'])"
python bench.py --ring --exchange p2p --workload 8192x1024 --steps 300 --warmup 30 --reps 3 --no-cpu-baseline | python -c "import json,sys; d=json.load(sys.stdin); print('ring 8192x1024 p2p us/step %.2f' % (d['ms_per_step']*1e3), d['config']['p2p'], d['parity_check']['ok'])"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_ring_p2p_small -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --ring --exchange p2p --workload 1024x128 --steps 3000 --warmup 30 --reps 1 --no-cpu-baseline --no-verify > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/prof_ring_p2p_small.err
cat $GRAFT_REPO_ROOT/$OUT/prof_ring_p2p_small/trace_kernel_stats.csv | cut -c1-150 | head -4
