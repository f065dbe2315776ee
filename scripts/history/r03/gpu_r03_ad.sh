#!/bin/bash
# ODDX (regions grow one column per side and step): parity suite on the tall geometry, then A/B against 64 x 23
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03ad
mkdir -p $OUT
V=mpilattice-boltzmann_amd/lib/variants
timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "not exhaustive" > $OUT/pytest_gpu.log 2>&1; tail -4 $OUT/pytest_gpu.log
timeout -k 10 300 python scripts/fuzz_kernels.py --cases 200 --seed 5 --scale 4 > $OUT/fuzz.log 2>&1; tail -2 $OUT/fuzz.log
{
for g in 8192x8192 4096x4096 2048x2048 1024x1024; do
  s=60; [ $g = 1024x1024 ] && s=400; [ $g = 4096x4096 ] && s=120; [ $g = 2048x2048 ] && s=400
  echo "== $g"
  timeout -k 10 300 python scripts/ab_libs.py --grid $g --steps $s --rounds 3 $V/tall23.so $V/oddx25.so $V/oddx24.so $V/tall23.so $V/oddx25.so $V/oddx24.so 2>&1 | tail -6
done
} | grep -v amdgpu.ids | tee $OUT/ab_oddx.txt
