#!/bin/bash
# Round-3 session O: rehearsal of the driver's N = 3 and N = 4 invocations with the rank processes sharing this box's one GPU.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03o
mkdir -p $OUT
for n in 3 4; do
  t0=$(date +%s)
  LBM_FORCE_DEVICE=0 timeout -k 10 300 python bench.py --gpus $n --steps 20 --warmup 5 > $OUT/bench_${n}ranks_one_gpu.json 2> $OUT/bench_${n}ranks.err || { tail -30 $OUT/bench_${n}ranks.err; exit 1; }
  echo "N=$n seconds: $(( $(date +%s) - t0 ))"
  python -c "
import json,sys; d=json.load(open(sys.argv[1]))
print(d['n_gpus'], 'us/step %.2f' % (d['ms_per_step']*1e3), d['config']['loop'], d['config']['macro_k'], d['parity_check']['ok'], 'settle', d['timing']['settle_reps'], 'wall', d['wall_s'], d.get('truncated'))
print('  variants', {k: v.get('error', v.get('value')) for k, v in d['variants'].items()})
s=d['secondary']['input_1024x1024']; print('  secondary p2p', {k: s['p2p'].get(k) for k in ('us_per_step','parity_ok','reynolds_line_equals_reference','p2p')}, 'rccl', s['rccl'].get('error'))
print('  phases max', {k: round(v,1) for k,v in d['phases']['max_over_ranks'].items() if k in ('setup','steps','reduce','push_first','push_avg','interior_avg','edge_avg','host_overhead')})
" $OUT/bench_${n}ranks_one_gpu.json
done
