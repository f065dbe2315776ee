#!/bin/bash
# One GPU-box session: parity suite, bench line, rocprofv3 kernel trace + the PMC passes behind the roofline line.
# Usage (from the repo root, via gpurun): bash scripts/gpu_round.sh <tag> [skip-tests]
set -e
TAG=${1:-r03}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
if [ "$2" != "skip-tests" ]; then
  timeout -k 10 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu_$TAG.log 2>&1 || (tail -40 $OUT/pytest_gpu_$TAG.log; exit 1)
  tail -5 $OUT/pytest_gpu_$TAG.log
fi
cd /tmp
B="$GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-variants --reps 1"
# the PMC passes run the DRIVER's step count: 20 steps = 2 x lbm_multi_kernel<4> + 4 x lbm_multi_kernel<3> launches, so that
# both instantiations a driver-style line times are profiled (make_roofline.py keeps every kernel it finds)
P="--steps 20 --warmup 5 --reps 2"
# the kernel trace runs the DEFAULT region (200 steps x 5 repetitions + warm-up: ~280 launches), so that the launches made while
# the part ramps up from idle (up to 1.7 ms each, a dozen of them) weigh as little in its average as in bench.py's median
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-variants > $OUT/prof_bench_$TAG.json 2> $OUT/prof_$TAG.err
echo trace done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_${TAG}_fetch -o pmc -- python3 $B $P > /dev/null 2> $OUT/pmc_${TAG}_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_${TAG}_write -o pmc -- python3 $B $P > /dev/null 2> $OUT/pmc_${TAG}_write.err
echo hbm passes done
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_${TAG}_sq1 -o pmc -- python3 $B $P > /dev/null 2> $OUT/pmc_${TAG}_sq1.err
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/pmc_${TAG}_sq2 -o pmc -- python3 $B $P > /dev/null 2> $OUT/pmc_${TAG}_sq2.err
echo sq passes done
cd $GRAFT_REPO_ROOT
python scripts/make_roofline.py $TAG $OUT/pmc_${TAG}_fetch $OUT/pmc_${TAG}_write $OUT/pmc_${TAG}_sq1 $OUT/pmc_${TAG}_sq2 | tail -30
# the bench lines AFTER the PMC passes of this session: roofline.* then comes from the counters of the same build on the same box
python bench.py > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err || (cat $OUT/bench_$TAG.err; exit 1)
cat $OUT/bench_$TAG.json | cut -c1-400
# the driver's own invocation: short timed region
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_style_$TAG.json 2>> $OUT/bench_$TAG.err
cat $OUT/bench_driver_style_$TAG.json | cut -c1-200
mkdir -p $OUT/profiles_$TAG && cp -r profiles/$TAG/* $OUT/profiles_$TAG/
find $OUT/prof_$TAG -name "*kernel_stats.csv" -exec cp {} $OUT/profiles_$TAG/kernel_stats_bench_8192.csv \;
cp $OUT/bench_$TAG.json $OUT/profiles_$TAG/bench_n1.json
cp $OUT/bench_driver_style_$TAG.json $OUT/profiles_$TAG/bench_n1_driver_style.json
