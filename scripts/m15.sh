cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for g in 128x128 128x256 256x256 512x512 1024x1024; do
python scripts/sweep.py --grid $g --steps 4000 --rounds 3 "default" "LBM_TUNE_NARROW_MAX=100000000"
done
