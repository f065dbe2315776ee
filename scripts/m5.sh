cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "rccl or nccl" 2>&1 | tail -3
python scripts/measure.py --grid 1024x128 --mode ring --steps 2000 2>&1 | grep mode=
python scripts/measure.py --grid 8192x8192 --mode ring --steps 200 2>&1 | grep mode=
python scripts/measure.py --grid 8192x4096 --mode ring --steps 200 2>&1 | grep mode=
python scripts/measure.py --grid 8192x2048 --mode ring --steps 200 2>&1 | grep mode=
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 50 --warmup 5 --no-cpu-baseline 2>&1 | tail -2
