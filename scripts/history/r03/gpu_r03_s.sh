#!/bin/bash
# Is the 4-step launch held back by the power limit?  Socket power / shader clock while each kernel family runs for seconds.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03s
mkdir -p $OUT
L=mpilattice-boltzmann_amd/lib/liblbm_d2q9.so
ls /sys/class/drm/ > $OUT/sysfs.txt 2>&1
for d in /sys/class/drm/card*/device; do echo $d; ls $d/hwmon/*/ 2>/dev/null | tr '\n' ' '; echo; done >> $OUT/sysfs.txt 2>&1
(rocm-smi --showpower --showclocks --showmaxpower --showperflevel 2>&1 | head -60) > $OUT/rocm_smi_idle.txt
{
timeout -k 10 200 python scripts/power_trace.py --label "K=4 8192x8192" -- python scripts/ab_libs.py --grid 8192x8192 --steps 400 --rounds 24 $L
timeout -k 10 200 python scripts/power_trace.py --label "K=3 8192x8192" -- python scripts/ab_libs.py --grid 8192x8192 --steps 400 --rounds 24 --env LBM_TUNE_MULTI_K=3 $L
timeout -k 10 200 python scripts/power_trace.py --label "K=1 (one-step kernel) 8192x8192" -- python scripts/ab_libs.py --grid 8192x8192 --steps 200 --rounds 24 --env LBM_TUNE_MULTI_K=0 $L
timeout -k 10 200 python scripts/power_trace.py --label "K=4 fast av_vels 8192x8192" -- python scripts/ab_libs.py --grid 8192x8192 --steps 400 --rounds 24 --flags 64 $L
} 2>&1 | grep -v amdgpu.ids | tee $OUT/power.txt
