#!/bin/bash
# Round-3 session B: parity suite, ring timings after the run-boundary changes, the one-XCD hand-off experiment, PMC round.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03b
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1
rc=$?
tail -8 $OUT/pytest_gpu.log
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "pytest rc=$rc: stopping"; exit $rc; fi
short() { python -c "import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], 'us/step %.2f' % (d['ms_per_step']*1e3), d['config']['loop'], d['config'].get('p2p'), (d.get('parity_check') or {}).get('ok'), [round(t,3) for t in d['timing']['ms_per_rep']], {k: round(v,1) for k, v in (d.get('phases') or {}).get('max_over_ranks', {}).items() if k in ('setup','reduce','host_overhead','steps')})" $1; }
echo "== rings 8192x1024"
for steps in 20 300; do
  reps=9; [ $steps = 300 ] && reps=3
  for cfg in "new:" "nospin:LBM_SPIN_WAIT_US=0"; do
    name=${cfg%%:*}; envs=${cfg#*:}
    env $envs timeout -k 10 300 python bench.py --ring --exchange p2p --workload 8192x1024 --steps $steps --warmup 5 --reps $reps --no-cpu-baseline --no-variants --no-secondary > $OUT/ring_8192x1024_p2p_${name}_s$steps.json 2>> $OUT/ring.err || { tail -20 $OUT/ring.err; exit 1; }
    short $OUT/ring_8192x1024_p2p_${name}_s$steps.json
  done
done
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-variants > $OUT/bench_n1_s20.json 2>> $OUT/ring.err; short $OUT/bench_n1_s20.json
echo "== one-XCD neighbour hand-off"
hipcc --offload-arch=gfx950 -O3 scripts/experiments/xcd_handoff.hip -o /tmp/xcd_handoff 2> /dev/null && timeout -k 10 120 /tmp/xcd_handoff | tee $OUT/xcd_handoff.txt
hipcc --offload-arch=gfx950 -O3 scripts/experiments/grid_barrier.hip -o /tmp/grid_barrier 2> /dev/null && timeout -k 10 120 /tmp/grid_barrier | tee $OUT/grid_barrier.txt
echo "== PMC round"
bash scripts/gpu_round.sh r03 skip-tests 2>&1 | tail -40
