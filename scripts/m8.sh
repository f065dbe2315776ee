cd $GRAFT_REPO_ROOT
python scripts/measure.py --grid 8192x8192 --steps 200 2>&1 | grep mode=
for mb in 8192 16384 32768; do for sk in 0 1 2 34; do echo "MAXBLOCKS=$mb SKEW=$sk"; LBM_TUNE_SKEW=$sk LBM_TUNE_MAXBLOCKS=$mb python scripts/measure.py --grid 8192x8192 --steps 200 2>&1 | grep mode=; done; done
for mb in 12288 24576; do echo "MAXBLOCKS=$mb SKEW=0"; LBM_TUNE_SKEW=0 LBM_TUNE_MAXBLOCKS=$mb python scripts/measure.py --grid 8192x8192 --steps 200 2>&1 | grep mode=; done
python scripts/measure.py --grid 8192x8192 --steps 200 2>&1 | grep mode=
