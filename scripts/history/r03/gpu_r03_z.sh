#!/bin/bash
# After the 64 x 23 / 768-lane change: fuzz, where K = 4 takes over from K = 3, rings, bench lines
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03z
mkdir -p $OUT
L=mpilattice-boltzmann_amd/lib/liblbm_d2q9.so
O=mpilattice-boltzmann_amd/lib/variants/old13.so
timeout -k 10 600 python scripts/fuzz_kernels.py --cases 400 --seed 23 > $OUT/fuzz.log 2>&1; tail -2 $OUT/fuzz.log
{
for g in 512x512 640x640 1024x512 512x1024 768x768 1024x768 1536x1536; do
  echo "== $g: K=3 / K=4 new / K=4 old geometry"
  timeout -k 10 200 python scripts/ab_libs.py --grid $g --steps 600 --rounds 3 $L::LBM_TUNE_MULTI_K=3 $L::LBM_TUNE_MULTI_K=4 $O::LBM_TUNE_MULTI_K=4 $L::LBM_TUNE_MULTI_K=3 $L::LBM_TUNE_MULTI_K=4 2>&1 | tail -5
done
} | grep -v amdgpu.ids | tee $OUT/ab_k3_k4_threshold_768lanes.txt
short() { python -c "import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], 'us/step %.2f' % (d['ms_per_step']*1e3), d['config']['loop'], d['config'].get('p2p'), (d.get('parity_check') or {}).get('ok'))" $1; }
for wl in 8192x4096 8192x2048 8192x1024; do
  python bench.py --ring --exchange p2p --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-variants --no-secondary > $OUT/ring_${wl}_p2p_s20.json; short $OUT/ring_${wl}_p2p_s20.json
done
python bench.py --ring --exchange rccl --workload 8192x1024 --steps 20 --warmup 5 --no-cpu-baseline --no-variants --no-secondary > $OUT/ring_8192x1024_rccl_s20.json; short $OUT/ring_8192x1024_rccl_s20.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver.json; cut -c1-200 $OUT/bench_driver.json
