#!/bin/bash
# Round-3 session A: parity suite, headline bench, the 2-rank one-GPU harvest, ring timings (old vs new tail / first push),
# tile-width A/B.  Usage (via gpurun): bash scripts/gpu_r03_a.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03a
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1
rc=$?
tail -15 $OUT/pytest_gpu.log
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "pytest rc=$rc: stopping"; exit $rc; fi
short() { python -c "import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], 'us/step %.2f' % (d['ms_per_step']*1e3), d['config']['loop'], d['config'].get('p2p'), (d.get('parity_check') or {}).get('ok'), 'wall', d.get('wall_s'))" $1; }
echo "== bench driver style"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_n1_driver_style.json 2> $OUT/bench_n1_driver_style.err || { tail -20 $OUT/bench_n1_driver_style.err; exit 1; }
short $OUT/bench_n1_driver_style.json
echo "== 2 ranks on one GPU, driver-style invocation"
t0=$(date +%s)
LBM_FORCE_DEVICE=0 LBM_DIST_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 2 --steps 20 --warmup 5 > $OUT/bench_2ranks_one_gpu.json 2> $OUT/bench_2ranks_one_gpu.err || { tail -30 $OUT/bench_2ranks_one_gpu.err; exit 1; }
echo "seconds: $(( $(date +%s) - t0 ))"
short $OUT/bench_2ranks_one_gpu.json
echo "== rings 8192x1024"
for steps in 20 300; do
  reps=9; [ $steps = 300 ] && reps=3
  for cfg in "new:" "oldtail:LBM_TUNE_MACRO_GHOST=3" "k4:LBM_TUNE_MACRO_K=4"; do
    name=${cfg%%:*}; envs=${cfg#*:}
    env $envs timeout -k 10 300 python bench.py --ring --exchange p2p --workload 8192x1024 --steps $steps --warmup 5 --reps $reps --no-cpu-baseline --no-variants --no-secondary > $OUT/ring_8192x1024_p2p_${name}_s$steps.json 2>> $OUT/ring.err || { tail -20 $OUT/ring.err; exit 1; }
    short $OUT/ring_8192x1024_p2p_${name}_s$steps.json
  done
done
timeout -k 10 300 python bench.py --workload 8192x1024 --steps 300 --warmup 30 --reps 3 --no-cpu-baseline --no-variants > $OUT/single_8192x1024.json 2>> $OUT/ring.err; short $OUT/single_8192x1024.json
echo "== tile width / block size A/B at 8192^2"
V=mpilattice-boltzmann_amd/lib/variants
timeout -k 10 600 python scripts/ab_libs.py --grid 8192x8192 --steps 60 --rounds 3 $V/base.so $V/base.so::LBM_TUNE_MULTI_TILE=32 $V/l256.so::LBM_TUNE_MULTI_TILE=32 $V/l256w6.so::LBM_TUNE_MULTI_TILE=32 $V/base.so 2>&1 | tail -6 | tee $OUT/ab_tile32.txt
