#!/bin/bash
# Round-3 session L: the short-region lines with settling repetitions.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03l
mkdir -p $OUT
short() { python -c "import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], 'us/step %.2f' % (d['ms_per_step']*1e3), d['config']['loop'], (d.get('parity_check') or {}).get('ok'), 'settle', d['timing'].get('settle_reps'), [round(t,3) for t in d['timing']['ms_per_rep']][:8], 'wall', d.get('wall_s'))" $1; }
python bench.py > $OUT/bench_n1.json 2> $OUT/bench.err && short $OUT/bench_n1.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_n1_driver_style.json 2>> $OUT/bench.err && short $OUT/bench_n1_driver_style.json
LBM_FORCE_DEVICE=0 python bench.py --gpus 2 --steps 20 --warmup 5 > $OUT/bench_2ranks_one_gpu_8192.json 2>> $OUT/bench.err && short $OUT/bench_2ranks_one_gpu_8192.json
python bench.py --ring --exchange p2p --workload 8192x1024 --steps 20 --warmup 5 --no-cpu-baseline --no-variants --no-secondary > $OUT/ring_8192x1024_p2p_s20.json 2>> $OUT/bench.err && short $OUT/ring_8192x1024_p2p_s20.json
python bench.py --ring --exchange rccl --workload 8192x1024 --steps 20 --warmup 5 --no-cpu-baseline --no-variants --no-secondary > $OUT/ring_8192x1024_rccl_s20.json 2>> $OUT/bench.err && short $OUT/ring_8192x1024_rccl_s20.json
python bench.py --ring --exchange p2p --workload 8192x4096 --steps 20 --warmup 5 --no-cpu-baseline --no-variants --no-secondary > $OUT/ring_8192x4096_p2p_s20.json 2>> $OUT/bench.err && short $OUT/ring_8192x4096_p2p_s20.json
python bench.py --ring --exchange p2p --workload 8192x2048 --steps 20 --warmup 5 --no-cpu-baseline --no-variants --no-secondary > $OUT/ring_8192x2048_p2p_s20.json 2>> $OUT/bench.err && short $OUT/ring_8192x2048_p2p_s20.json
timeout -k 10 300 python -m pytest tests -m gpu -q -x -k "bench_self_launch" 2>&1 | tail -2
