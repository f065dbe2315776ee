cd $GRAFT_REPO_ROOT
python scripts/sweep.py --rounds 5 "default" "LBM_TUNE_VARIANT=2" "LBM_TUNE_VARIANT=4" "LBM_TUNE_VARIANT=6" "LBM_TUNE_MAXBLOCKS=4096"
python scripts/sweep.py --grid 1024x1024 --steps 2000 --rounds 3 "default" "LBM_TUNE_VARIANT=4" "FLAGS=1" "LBM_TUNE_VARIANT=2"
