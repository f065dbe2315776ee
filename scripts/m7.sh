cd $GRAFT_REPO_ROOT
python scripts/measure.py --grid 8192x8192 --steps 200 2>&1 | grep mode=
for mb in 1024 2048 8192 16384 65536; do echo "MAXBLOCKS=$mb"; LBM_TUNE_MAXBLOCKS=$mb python scripts/measure.py --grid 8192x8192 --steps 200 2>&1 | grep mode=; done
for sk in 0 1 3 17 35 67 131; do echo "SKEW=$sk"; LBM_TUNE_SKEW=$sk python scripts/measure.py --grid 8192x8192 --steps 200 2>&1 | grep mode=; done
python scripts/measure.py --grid 8192x8192 --steps 200 2>&1 | grep mode=
