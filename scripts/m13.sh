cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python scripts/sweep.py --rounds 5 "default" "FLAGS=4"
python scripts/sweep.py --grid 1024x1024 --steps 2000 --rounds 3 "default" "FLAGS=4"
python scripts/sweep.py --grid 256x256 --steps 5000 --rounds 3 "default" "FLAGS=4"
