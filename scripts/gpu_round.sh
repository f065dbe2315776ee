#!/bin/bash
# One GPU-box session: parity suite, bench line, rocprofv3 kernel trace + HBM counters.
# Usage (from the repo root, via gpurun): bash scripts/gpu_round.sh <tag>
set -e
TAG=${1:-r01}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $OUT/pytest_gpu_$TAG.log; cat $OUT/pytest_gpu_$TAG.log
grep -q "passed" $OUT/pytest_gpu_$TAG.log
python bench.py > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err || (cat $OUT/bench_$TAG.err; exit 1)
cat $OUT/bench_$TAG.json
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 100 --warmup 10 > $OUT/prof_bench_$TAG.json 2> $OUT/prof_$TAG.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$TAG -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 20 --warmup 2 > /dev/null 2> $OUT/pmc_fetch_$TAG.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$TAG -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 20 --warmup 2 > /dev/null 2> $OUT/pmc_write_$TAG.err
find $OUT -name "*.csv" | head -20
