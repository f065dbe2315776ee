cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python scripts/measure.py --grid 8192x1024 --mode ring --steps 400 2>&1 | grep mode=
LBM_RCCL_SCHEDULE=edge python scripts/measure.py --grid 8192x1024 --mode ring --steps 400 2>&1 | grep mode=
cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ring_trace -o ring -- python3 $GRAFT_REPO_ROOT/scripts/measure.py --grid 8192x1024 --mode ring --steps 50 --warmup 5 --repeat 1 2>&1 | grep mode=
cd /tmp && LBM_RCCL_SCHEDULE=edge rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ring_trace_edge -o ring -- python3 $GRAFT_REPO_ROOT/scripts/measure.py --grid 8192x1024 --mode ring --steps 50 --warmup 5 --repeat 1 2>&1 | grep mode=
