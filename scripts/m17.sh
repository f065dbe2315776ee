cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q -k "bit_exact or partitioned or odd" 2>&1 | tail -3
for g in 128x128 256x256 1024x1024; do
python scripts/sweep.py --grid $g --steps 4000 --rounds 3 "default" "LBM_TUNE_NARROW_MAX=0"
done
