#!/bin/bash
# The secondary measurements quoted in DESIGN.md / profiles/README.md, one GPU-box session.
# Usage (via gpurun): bash scripts/gpu_extras.sh <tag>
set -e
TAG=${1:-r03}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$GRAFT_REPO_ROOT/gpurun_out/profiles_$TAG
mkdir -p $OUT
D=tests/golden/decks
# whole-run wall time of the drop-in CLI on the four shipped decks (its own "Elapsed time" line)
for n in 128x128 128x256 256x256 1024x1024; do
  ( cd /tmp && LBM_NO_OUTPUT=1 $GRAFT_REPO_ROOT/mpilattice-boltzmann_amd/bin/d2q9-bgk $GRAFT_REPO_ROOT/$D/input_$n.params $GRAFT_REPO_ROOT/$D/obstacles_$n.dat | grep -E "Elapsed time" | sed "s/^/$n /" )
done | tee $OUT/cli_decks.txt
short() { python -c "import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], 'us/step %.2f' % (d['ms_per_step']*1e3), d['config']['loop'], d['config'].get('p2p'), (d.get('parity_check') or {}).get('ok'))" $1; }
python bench.py --workload 1024x1024 --steps 3000 --warmup 100 --no-cpu-baseline > $OUT/bench_1024x1024.json; short $OUT/bench_1024x1024.json
# the row-partitioned loops on a 1-rank ring: one rank's share of the 8192^2 deck on 2 / 8 GPUs, of the 1024^2 deck on 8
for wl in 8192x4096 8192x1024 1024x128; do
  steps=300; [ $wl = 1024x128 ] && steps=3000
  for ex in p2p rccl; do
    python bench.py --ring --exchange $ex --workload $wl --steps $steps --warmup 30 --reps 3 --no-cpu-baseline --no-variants --no-secondary > $OUT/ring_${wl}_${ex}.json; short $OUT/ring_${wl}_${ex}.json
  done
done
# the driver's timed region (20 steps) on the 8-GPU share: what a run costs beyond its steps
python bench.py --ring --exchange p2p --workload 8192x1024 --steps 20 --warmup 5 --no-cpu-baseline --no-variants --no-secondary > $OUT/ring_8192x1024_p2p_s20.json; short $OUT/ring_8192x1024_p2p_s20.json
python scripts/ab_ring.py --grid 8192x1024 --steps 20 --rounds 60 LBM_SPIN_WAIT_US=0 LBM_SPIN_WAIT_US=4000 2>&1 | tail -2 | tee $OUT/ab_ring_spin_8192x1024_s20.txt
python bench.py --ring --exchange rccl --step-allreduce --workload 8192x1024 --steps 300 --warmup 30 --reps 3 --no-cpu-baseline --no-variants --no-secondary > $OUT/ring_8192x1024_rccl_step_allreduce.json; short $OUT/ring_8192x1024_rccl_step_allreduce.json
python bench.py --ring --exchange rccl --step-allreduce --workload 1024x128 --steps 3000 --warmup 30 --reps 3 --no-cpu-baseline --no-variants --no-secondary > $OUT/ring_1024x128_rccl_step_allreduce.json; short $OUT/ring_1024x128_rccl_step_allreduce.json
python bench.py --workload 8192x1024 --steps 300 --warmup 30 --reps 3 --no-cpu-baseline > $OUT/single_8192x1024.json; short $OUT/single_8192x1024.json
python bench.py --workload 1024x128 --steps 3000 --warmup 30 --reps 3 --no-cpu-baseline > $OUT/single_1024x128.json; short $OUT/single_1024x128.json
# two rank PROCESSES sharing this GPU (gloo group, IPC-mapped peers): the driver's own invocation, self-launched
# the driver's own N = 2 invocation, self-launched: headline + phases + variants + the shipped 1024 x 1024 deck in one line
LBM_FORCE_DEVICE=0 python bench.py --gpus 2 --steps 20 --warmup 5 > $OUT/bench_2ranks_one_gpu_8192.json; short $OUT/bench_2ranks_one_gpu_8192.json
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_ring_p2p_$TAG -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --ring --exchange p2p --workload 8192x1024 --steps 300 --warmup 30 --reps 1 --no-cpu-baseline --no-verify --no-variants --no-secondary --no-phases > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/prof_ring_p2p_$TAG.err
cp $GRAFT_REPO_ROOT/gpurun_out/prof_ring_p2p_$TAG/trace_kernel_stats.csv $OUT/ring_8192x1024_p2p_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_ring_p2p_small_$TAG -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --ring --exchange p2p --workload 1024x128 --steps 3000 --warmup 30 --reps 1 --no-cpu-baseline --no-verify --no-variants --no-secondary --no-phases > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/prof_ring_p2p_small_$TAG.err
cp $GRAFT_REPO_ROOT/gpurun_out/prof_ring_p2p_small_$TAG/trace_kernel_stats.csv $OUT/ring_1024x128_p2p_kernel_stats.csv
cd $GRAFT_REPO_ROOT
V=mpilattice-boltzmann_amd/lib
python scripts/ab_libs.py --grid 8192x8192 --steps 60 --rounds 4 $V/liblbm_d2q9.so $V/liblbm_d2q9.so:64 2>&1 | tail -2 | tee $OUT/ab_fast_avvels_8192.txt
hipcc --offload-arch=gfx950 -O3 scripts/experiments/xcd_handoff.hip -o /tmp/xcd_handoff 2> /dev/null && /tmp/xcd_handoff | tee $OUT/xcd_handoff.txt
