cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for g in 128x128 256x256 128x256 1024x1024; do
  python scripts/measure.py --grid $g --mode single --steps 20000 --flags 16 2>&1 | grep mode=
  python scripts/measure.py --grid $g --mode single --steps 20000 2>&1 | grep mode=
done
