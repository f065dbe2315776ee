#!/bin/bash
# The secondary measurements quoted in DESIGN.md / profiles/README.md, one GPU-box session.
# Usage (via gpurun): bash scripts/gpu_extras.sh <tag>
set -e
TAG=${1:-r01}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
D=tests/golden/decks
# whole-run wall time of the drop-in CLI on the four shipped decks (its own "Elapsed time" line)
for n in 128x128 128x256 256x256 1024x1024; do
  ( cd /tmp && LBM_NO_OUTPUT=1 $GRAFT_REPO_ROOT/mpilattice-boltzmann_amd/bin/d2q9-bgk $GRAFT_REPO_ROOT/$D/input_$n.params $GRAFT_REPO_ROOT/$D/obstacles_$n.dat | grep -E "Elapsed time" | sed "s/^/$n /" )
done | tee $OUT/cli_decks_$TAG.txt
python bench.py --workload 1024x1024 --steps 3000 --warmup 100 --no-cpu-baseline > $OUT/bench_1024_$TAG.json
cat $OUT/bench_1024_$TAG.json | cut -c1-200
python bench.py --ring --workload 8192x1024 --steps 300 --warmup 30 --no-cpu-baseline > $OUT/ring_8192x1024_bench_$TAG.json
cat $OUT/ring_8192x1024_bench_$TAG.json | cut -c1-200
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_ring_$TAG -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --ring --workload 8192x1024 --steps 300 --warmup 30 --no-cpu-baseline > /dev/null 2> $OUT/prof_ring_$TAG.err
cd $GRAFT_REPO_ROOT
LBM_TUNE_MULTI_K=0 python bench.py --no-cpu-baseline > $OUT/bench_onestep_$TAG.json
cat $OUT/bench_onestep_$TAG.json | cut -c1-200
