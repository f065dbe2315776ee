cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python scripts/measure.py --grid 8192x8192 --mode single --steps 200 2>&1 | grep mode=
python scripts/measure.py --grid 8192x1024 --mode ring --steps 402 2>&1 | grep mode=
python scripts/measure.py --grid 8192x2048 --mode ring --steps 402 2>&1 | grep mode=
python scripts/measure.py --grid 8192x4096 --mode ring --steps 201 2>&1 | grep mode=
LBM_TUNE_MULTI_K=3 python scripts/measure.py --grid 8192x8192 --mode single --steps 201 2>&1 | grep mode=
