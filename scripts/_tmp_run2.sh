cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "p2p or several_ranks or bench_self" > $OUT/t5.log 2>&1; tail -4 $OUT/t5.log
for ex in p2p; do
  for wl in 8192x1024 8192x4096 1024x128; do
    steps=300; [ $wl = 1024x128 ] && steps=3000
    python bench.py --ring --exchange $ex --workload $wl --steps $steps --warmup 30 --reps 3 --no-cpu-baseline > $OUT/ring_${wl}_${ex}.json 2> $OUT/ring_${wl}_${ex}.err
    python - <<PY
import json
d=json.load(open("$OUT/ring_${wl}_${ex}.json"))
print("$ex $wl", "us/step %.2f" % (d["ms_per_step"]*1e3), d["config"]["loop"], d["config"]["p2p"], d.get("parity_check",{}).get("ok"))
PY
  done
done
LBM_P2P_SCHEDULE=edge python bench.py --ring --exchange p2p --workload 1024x128 --steps 3000 --warmup 30 --reps 3 --no-cpu-baseline | python -c "import json,sys; d=json.load(sys.stdin); print('p2p edge 1024x128 us/step %.2f' % (d['ms_per_step']*1e3))"
LBM_P2P_SCHEDULE=serial python bench.py --ring --exchange p2p --workload 8192x1024 --steps 300 --warmup 30 --reps 3 --no-cpu-baseline | python -c "import json,sys; d=json.load(sys.stdin); print('p2p serial 8192x1024 us/step %.2f' % (d['ms_per_step']*1e3))"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_ring_p2p -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --ring --exchange p2p --workload 8192x1024 --steps 300 --warmup 30 --reps 1 --no-cpu-baseline --no-verify > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/prof_ring_p2p.err
cat $GRAFT_REPO_ROOT/$OUT/prof_ring_p2p/trace_kernel_stats.csv | cut -c1-160 | head -12
