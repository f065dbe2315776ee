cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "rccl or nccl" 2>&1 | tail -3
for g in 8192x8192 8192x1024 1024x128; do
  python scripts/measure.py --grid $g --mode single --steps 400 2>&1 | grep mode=
  python scripts/measure.py --grid $g --mode ring --steps 400 2>&1 | grep mode=
done
