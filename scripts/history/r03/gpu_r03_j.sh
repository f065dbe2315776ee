#!/bin/bash
# Round-3 session J: where K = 4 takes over from K = 3 on whole grids; then the full record with the new defaults.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03j
mkdir -p $OUT
V=mpilattice-boltzmann_amd/lib/variants
for g in 768x768 1024x512 512x1024 640x640 1536x1536; do
  echo "== $g"
  timeout -k 10 300 python scripts/ab_libs.py --grid $g --steps 1200 --rounds 3 --env LBM_TUNE_TILE_MAX=0 $V/base.so::LBM_TUNE_MULTI_K=3 $V/base.so::LBM_TUNE_MULTI_K=4 2>&1 | tail -2 | tee -a $OUT/ab_k3_k4_threshold.txt
done
bash scripts/gpu_round.sh r03 2>&1 | tail -12 && bash scripts/gpu_extras.sh r03 2>&1 | grep -v "amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl\|socket.cpp\|OMP_NUM\|\*\*\*\*\|^$" | tail -40
