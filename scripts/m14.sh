cd /tmp && export TMPDIR=/tmp
D=$GRAFT_REPO_ROOT/tests/golden/decks
for g in 128x128 256x256 1024x1024; do
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/small_$g -o t -- python3 $GRAFT_REPO_ROOT/scripts/measure.py --grid $g --steps 2000 --warmup 10 --repeat 1 2>&1 | grep mode=
done
