cd $GRAFT_REPO_ROOT
python scripts/measure.py --grid 8192x1024 --mode single --steps 400 2>&1 | grep mode=
python scripts/measure.py --grid 8192x1024 --mode ring --steps 400 2>&1 | grep mode=
LBM_RCCL_SCHEDULE=edge python scripts/measure.py --grid 8192x1024 --mode ring --steps 400 2>&1 | grep mode=
LBM_RCCL_PRIORITY=1 python scripts/measure.py --grid 8192x1024 --mode ring --steps 400 2>&1 | grep mode=
LBM_RCCL_PRIORITY=1 LBM_RCCL_SCHEDULE=edge python scripts/measure.py --grid 8192x1024 --mode ring --steps 400 2>&1 | grep mode=
