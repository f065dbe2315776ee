cd $GRAFT_REPO_ROOT
for g in 8192x8192 8192x4096 8192x2048 8192x1024; do
  python scripts/measure.py --grid $g --mode single
  python scripts/measure.py --grid $g --mode ring
  python scripts/measure.py --grid $g --mode ring-torch
done
python scripts/measure.py --grid 1024x1024 --mode single --steps 2000
python scripts/measure.py --grid 1024x1024 --mode ring --steps 2000
python scripts/measure.py --grid 1024x128 --mode ring --steps 2000
python scripts/measure.py --grid 256x256 --mode single --steps 5000
python scripts/measure.py --grid 128x128 --mode single --steps 5000
