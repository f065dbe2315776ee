#!/bin/bash
# Quick GPU check after a kernel change (about 2.5 minutes of box time): the GPU parity tests, the four shipped decks through the
# CLI, a short headline bench, and — when lib/variants/tile_stamps.so exists (scripts/build_variant.sh tile_stamps
# -DLBM_TILE_STAMPS=1) — the phase stamps of the tile kernel.
#   gpurun --timeout 1100 -- 'bash scripts/gpu_quick.sh > gpurun_out/quick.log 2>&1; head -8 gpurun_out/quick.log'
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
D=tests/golden/decks
for n in 128x128 128x256 256x256 1024x1024; do
  ( cd /tmp && LBM_NO_OUTPUT=1 $GRAFT_REPO_ROOT/mpilattice-boltzmann_amd/bin/d2q9-bgk $GRAFT_REPO_ROOT/$D/input_$n.params $GRAFT_REPO_ROOT/$D/obstacles_$n.dat | grep -E "Elapsed time" | sed "s/^/$n /" )
done
python bench.py --steps 60 --warmup 6 --reps 3 --no-cpu-baseline --no-variants | python -c "import json,sys; d=json.load(sys.stdin); print('8192 us/step %.1f' % (d['ms_per_step']*1e3))"
if [ -f mpilattice-boltzmann_amd/lib/variants/tile_stamps.so ]; then
  python scripts/tile_stamps.py --grid 256x256 && python scripts/tile_stamps.py --grid 128x128
fi
