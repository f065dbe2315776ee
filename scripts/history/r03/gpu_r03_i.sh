#!/bin/bash
# Round-3 session I: K = 4 on 64 x 12 tiles — parity, then K = 3 vs K = 4 across grid sizes and on rings.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03i
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1
rc=$?
tail -6 $OUT/pytest_gpu.log
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "pytest rc=$rc: stopping"; exit $rc; fi
V=mpilattice-boltzmann_amd/lib/variants
for g in 8192x8192 4096x4096 2048x2048 1024x1024 512x512 8192x1024; do
  steps=60; [ $g = 1024x1024 ] && steps=600; [ $g = 512x512 ] && steps=1200; [ $g = 2048x2048 ] && steps=240
  echo "== $g"
  timeout -k 10 300 python scripts/ab_libs.py --grid $g --steps $steps --rounds 3 $V/base.so::LBM_TUNE_MULTI_K=3 $V/base.so::LBM_TUNE_MULTI_K=4 $V/ty4_13.so::LBM_TUNE_MULTI_K=4 $V/ty4_11.so::LBM_TUNE_MULTI_K=4 2>&1 | tail -4 | tee -a $OUT/ab_k3_k4.txt
done
short() { python -c "import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], 'us/step %.2f' % (d['ms_per_step']*1e3), d['config']['macro_k'], (d.get('parity_check') or {}).get('ok'))" $1; }
for wl in 8192x1024 8192x4096 1024x128; do
  for K in 3 4; do
    for steps in 20 300; do
      s=$steps; [ $wl = 1024x128 ] && s=$((steps*10))
      LBM_TUNE_MACRO_K=$K timeout -k 10 300 python bench.py --ring --exchange p2p --workload $wl --steps $s --warmup 5 --no-cpu-baseline --no-variants --no-secondary --no-phases > $OUT/ring_${wl}_K${K}_s$s.json 2>> $OUT/ring.err || { tail -5 $OUT/ring.err; exit 1; }
      short $OUT/ring_${wl}_K${K}_s$s.json
    done
  done
done
