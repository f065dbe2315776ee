#!/bin/bash
# Round-3 session F: final-code validation — parity suite, the bench lines (21 repetitions for short regions), 2-rank harvest, ring at 20 steps.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03f
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1
rc=$?
tail -4 $OUT/pytest_gpu.log
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "pytest rc=$rc: stopping"; exit $rc; fi
short() { python -c "import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], 'us/step %.2f' % (d['ms_per_step']*1e3), d['config']['loop'], (d.get('parity_check') or {}).get('ok'), 'frac', (d.get('roofline') or {}).get('frac'), 'wall', d.get('wall_s'))" $1; }
python bench.py > $OUT/bench_n1.json 2> $OUT/bench.err && short $OUT/bench_n1.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_n1_driver_style.json 2>> $OUT/bench.err && short $OUT/bench_n1_driver_style.json
LBM_FORCE_DEVICE=0 python bench.py --gpus 2 --steps 20 --warmup 5 > $OUT/bench_2ranks_one_gpu_8192.json 2>> $OUT/bench.err && short $OUT/bench_2ranks_one_gpu_8192.json
python bench.py --ring --exchange p2p --workload 8192x1024 --steps 20 --warmup 5 --no-cpu-baseline --no-variants --no-secondary > $OUT/ring_8192x1024_p2p_s20.json 2>> $OUT/bench.err && short $OUT/ring_8192x1024_p2p_s20.json
python bench.py --ring --exchange p2p --workload 8192x1024 --steps 300 --warmup 30 --reps 3 --no-cpu-baseline --no-variants --no-secondary > $OUT/ring_8192x1024_p2p.json 2>> $OUT/bench.err && short $OUT/ring_8192x1024_p2p.json
python bench.py --steps 300 --warmup 30 --reps 3 --workload 8192x1024 --no-cpu-baseline --no-variants > $OUT/single_8192x1024.json 2>> $OUT/bench.err && short $OUT/single_8192x1024.json
