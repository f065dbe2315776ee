#!/bin/bash
# One GPU-box session = a tag and a list of steps:   gpurun -- 'bash scripts/gpu_session.sh <tag> <step> [<step> ...]'
# Every step writes under gpurun_out/<tag>/ (merged back by gpurun) and the steps are joined with &&: after a step that
# fails or times out no further GPU step starts.  A step is `name` or `name:arg` (arg: the rest after the first colon).
#   tests[:<pytest -k expr>]      the GPU parity suite (or a selection), complete log in pytest.log
#   smoke                         __graft_entry__.smoke()
#   rings[:<grids>]               1-rank peer-to-peer rings, ONE PER PROCESS, over the halo depths (LBM_TUNE_MACRO_GHOST) 0 8 12 16 at
#                                 20 and 200 steps per run — a rank's share of a partitioned run, the wire a self-copy (default grids: 8192x1024 1024x128)
#   ringsrccl[:<grids>]           the same over the RCCL loop
#   ring1:<grid>:<steps>:<rounds>:<VAR=val VAR=val ...>   one ring in one process under that environment (wall time per run, as bench.py times it)
#   single1:<grid>:<steps>:<rounds>:<VAR=val@VAR=val>     the same grid as ONE periodic domain (lbm_run), one process
#   ringbench:<grid>[:<steps>]    bench.py --ring on one grid (default 20 steps): the line with `phases` in ring_<grid>_s<steps>.json
#   bench[:<bench.py args>]       bench.py (default: the driver's --steps 20 --warmup 5), the line in bench.json
#   prof[:<bench.py args>]        rocprofv3 --kernel-trace --stats of bench.py, summary in prof/
#   pmc:<counters>:<bench args>   one rocprofv3 --pmc pass of bench.py (its own run, no trace flags)
#   pmcsum:<counters>[:<tag>[:<bench args>]]   the same, then the mean per launch of every counter for lbm_multi_kernel<4 (appended to pmcsum.txt)
#   uselib:<variant>              ON THE BOX: lib/variants/<variant>.so takes the place of lib/liblbm_d2q9.so for the steps that follow
#   round[:<workload>]            the record behind bench.py's `roofline` for profiles/<tag>/: rocprofv3 --kernel-trace --stats of the default bench.py,
#                                 FETCH_SIZE / WRITE_SIZE / two SQ --pmc passes at the driver's 20 steps (each its own run), scripts/make_roofline.py,
#                                 then the default and the driver-style bench lines; everything copied to gpurun_out/<tag>/profiles/
#   extras                        the secondary figures: 1024x1024 bench line, 1-rank rings (p2p / rccl, with phases) of a rank's share of the
#                                 8192^2 deck on 2 / 4 / 8 GPUs and of the 1024^2 deck on 8, kernel traces of the two 8-GPU shares, two rank
#                                 processes on this GPU (the driver's N = 2 line)
#   decks                         the four shipped decks through bin/d2q9-bgk
#   fuzz[:<cases>[:<seed>]]       scripts/fuzz_kernels.py
#   timeline[:<bench args>]       per-dispatch start / end of the last 400 kernels of a short bench.py run (default: the 8-GPU share as a ring, 20 steps)
#   soak                          2000-step runs x 3 of both native loops on a 1-rank ring of 8192x1024 rows, then 4 rank processes on this GPU (400 steps x 5), all parity-checked
#   ab:<libA>,<libB>[:args]       scripts/ab_libs.py on two builds of the library (lib/variants/*.so)
#   tilebench                     the tile decomposition's figures: a rank's block of the 8192^2 deck on 4 x 2 / 2 x 2 / 2 x 1 ranks and of the 1024^2 deck on
#                                 4 x 2 as 1 x 1 tile rings (one per process, 300 / 3000 steps and the driver's 20) beside the row shares of the same cells
#                                 in the same session, kernel traces of the two 8-GPU blocks, then 4 rank processes on this GPU as 2 x 2 tiles and as rows
#   tilewide                      grids much wider than tall on 8 ranks: a rank's share as a row block and as a block of an 8 x 1 tiling, 1-rank rings
#   tiles[:<case>;<case>...]      tests/tile_inprocess_worker.py per case ("nx ny px py K ghost group runs [walls]"; default: a spread of rank grids),
#                                 each a fresh process: the ranks of a tile (2-D) decomposition on this GPU, bit for bit against the oracle
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 LBM_P2P_TIMEOUT_MS=${LBM_P2P_TIMEOUT_MS:-10000}
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"

ring_lines() {   # $1 exchange, $2 file, $3.. grids
  local exch=$1 file=$2; shift 2
  for g in "$@"; do
    for ghost in 0 8 12 16; do
      for steps in 20 200; do
        local rounds=40; [ $steps = 200 ] && rounds=8
        [ "${g#1024x}" != "$g" ] && rounds=$((rounds * 3))
        echo "== $exch ring $g, LBM_TUNE_MACRO_GHOST=$ghost, $steps steps per run"
        LBM_TUNE_MACRO_GHOST=$ghost timeout -k 10 240 python scripts/ab_ring.py --exchange $exch --grid $g --steps $steps --rounds $rounds - 2>&1 | tail -1 || return 1
      done
    done
  done | grep -v amdgpu.ids | tee "$file"
}

step() {
  local name=${1%%:*} arg=""
  [ "$name" != "$1" ] && arg=${1#*:}
  echo "#### $TAG: $name $arg"
  case $name in
    tests)
      if [ -n "$arg" ]; then timeout -k 10 1100 python -m pytest tests -m gpu -x -q -k "$arg" > "$OUT/pytest.log" 2>&1
      else timeout -k 10 1100 python -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1; fi
      local rc=$?; tail -15 "$OUT/pytest.log"; return $rc ;;
    tiles)
      local cases=${arg:-"512 256 1 1 4 - - 20,11;512 256 2 1 4 - - 20,11;512 256 2 2 4 - - 20,11 walls;768 384 3 2 4 - - 33;1024 512 4 2 3 - - 19,7;640 300 2 3 4 7 - 25;2048 1100 2 1 4 - - 17;2048 2048 2 2 4 - - 21"}
      local IFS=';'
      for cs in $cases; do
        echo "== tiles $cs"
        ( IFS=' '; GPU_MAX_HW_QUEUES=16 timeout -k 10 300 python tests/tile_inprocess_worker.py $cs 2>&1 | grep -v amdgpu.ids | tail -4 ) || return 1
      done | tee "$OUT/tiles.txt"
      ! grep -q "Error\|Traceback\|assert" "$OUT/tiles.txt" ;;
    smoke) timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee "$OUT/smoke.log" ;;
    rings) ring_lines p2p "$OUT/rings_p2p.txt" ${arg:-8192x1024 1024x128} ;;
    ringsrccl) ring_lines rccl "$OUT/rings_rccl.txt" ${arg:-8192x1024 1024x128} ;;
    ring1)
      local g st ro ev; IFS=: read -r g st ro ev <<< "$arg"
      echo "== ring $g, $st steps per run, env: $ev" | tee -a "$OUT/ring1.txt"
      env ${ev//@/ } timeout -k 10 240 python scripts/ab_ring.py --grid $g --steps $st --rounds $ro - 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a "$OUT/ring1.txt" ;;
    single1)
      local g st ro ev; IFS=: read -r g st ro ev <<< "$arg"
      echo "== single periodic grid $g, $st steps per run, env: $ev" | tee -a "$OUT/single1.txt"
      env ${ev//@/ } timeout -k 10 240 python scripts/ab_ring.py --single --grid $g --steps $st --rounds $ro - 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a "$OUT/single1.txt" ;;
    ringbench)
      local g=${arg%%:*} st=20; [ "$g" != "$arg" ] && st=${arg#*:}
      timeout -k 10 300 python bench.py --ring --workload $g --steps $st --warmup 5 --no-cpu-baseline --no-variants --no-secondary > "$OUT/ring_${g}_s$st.json" 2> "$OUT/ring_${g}_s$st.err"
      local rc=$?; python - "$OUT/ring_${g}_s$st.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ph = d.get("phases", {}).get("max_over_ranks", {})
print(f"ring {d['config'].get('workload')}: {d['ms_per_step'] * 1e3:.2f} us/step; " + ", ".join(f"{k} {v:.1f}" for k, v in ph.items()))
PY
      return $rc ;;
    bench) timeout -k 10 500 python bench.py ${arg:---steps 20 --warmup 5} > "$OUT/bench.json" 2> "$OUT/bench.err"; local rc=$?; tail -c 1500 "$OUT/bench.json"; return $rc ;;
    prof)
      rm -rf "$OUT/prof"
      timeout -k 10 500 rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o trace --output-format csv -- python3 bench.py ${arg:---steps 20 --warmup 5 --no-cpu-baseline --no-variants} > "$OUT/prof_bench.json" 2> "$OUT/prof.err"
      local rc=$?; find "$OUT/prof" -name '*kernel_stats.csv' | head -1 | xargs -r head -8; find "$OUT/prof" -name '*kernel_trace.csv' -delete; return $rc ;;
    pmc)
      local ctr=${arg%%:*} rest=""; [ "$ctr" != "$arg" ] && rest=${arg#*:}
      local d="$OUT/pmc_$(echo $ctr | tr ' ' '_' | cut -c1-40)"; rm -rf "$d"
      timeout -k 10 500 rocprofv3 --pmc $ctr -d "$d" -o pmc --output-format csv -- python3 bench.py ${rest:---steps 20 --warmup 5 --no-cpu-baseline --no-variants --no-phases --no-power} > "$d.json" 2> "$d.err"
      local rc=$?; tail -3 "$d.err"; return $rc ;;
    pmcsum)
      local ctr lab rest; IFS=: read -r ctr lab rest <<< "$arg"
      local d="$OUT/pmcsum_${lab:-x}"; rm -rf "$d"
      timeout -k 10 500 rocprofv3 --pmc $ctr -d "$d" -o pmc --output-format csv -- python3 bench.py ${rest:---steps 20 --warmup 5 --reps 2} --no-cpu-baseline --no-variants --no-secondary --no-phases --no-power > "$d.json" 2> "$d.err" || { tail -5 "$d.err"; return 1; }
      python - "$d" "$lab" <<'PY' | tee -a "$OUT/pmcsum.txt"
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "lbm_multi_kernel<4" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(f"[{sys.argv[2]}] lbm_multi_kernel<4>: " + ", ".join(f"{k} {sum(v) / len(v):.4g} (n={len(v)})" for k, v in sorted(acc.items())))
PY
      find "$d" -name '*.csv' -size +200k -delete ;;
    pmcab)
      # counters of two BUILDS in one run: rocprofv3 --pmc <counters> over scripts/ab_libs.py <libs...>; means per kernel name (the builds' names differ)
      local ctr=${arg%%:*} rest=${arg#*:}
      case "$ctr" in *FETCH_SIZE*WRITE_SIZE*|*WRITE_SIZE*FETCH_SIZE*) echo "FETCH_SIZE and WRITE_SIZE need separate passes (together the pass hangs)"; return 2 ;; esac
      local d="$OUT/pmcab_$(echo $ctr | tr ' ' '_' | cut -c1-30)"; rm -rf "$d"
      timeout -k 10 500 rocprofv3 --pmc $ctr -d "$d" -o pmc --output-format csv -- python3 scripts/ab_libs.py $rest > "$d.txt" 2> "$d.err" || { tail -5 "$d.err"; return 1; }
      python - "$d" <<'PY' | tee -a "$OUT/pmcab.txt"
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "lbm_multi_kernel<4" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[1] if False else r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items()):
    print(k + ": " + ", ".join(f"{c} {sum(v) / len(v):.5g} (n={len(v)})" for c, v in sorted(cs.items())))
PY
      find "$d" -name '*.csv' -size +200k -delete ;;
    uselib) cp "mpilattice-boltzmann_amd/lib/variants/$arg.so" mpilattice-boltzmann_amd/lib/liblbm_d2q9.so && echo "now running lib/variants/$arg.so" ;;
    round)
      local wl=${arg:-8192x8192} P="$OUT/profiles"; mkdir -p "$P"
      local B="bench.py --workload $wl --no-cpu-baseline --no-variants --no-secondary --reps 1" S="--steps 20 --warmup 5 --reps 2"
      rm -rf "$OUT/prof" "$OUT"/pmc_fetch "$OUT"/pmc_write "$OUT"/pmc_sq1 "$OUT"/pmc_sq2
      # the kernel trace runs the DEFAULT region (200 steps x 5 repetitions + warm-up: ~280 launches): launches made while the part ramps
      # up from idle weigh as little in its average as in bench.py's median
      timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o trace -- python3 bench.py --workload $wl --no-cpu-baseline --no-variants --no-secondary > "$OUT/prof_bench.json" 2> "$OUT/prof.err" || return 1
      # the PMC passes run the DRIVER's step count, each in its own run (no trace flags beside --pmc)
      timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o pmc -- python3 $B $S > /dev/null 2> "$OUT/pmc_fetch.err" || return 1
      timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o pmc -- python3 $B $S > /dev/null 2> "$OUT/pmc_write.err" || return 1
      timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq1" -o pmc -- python3 $B $S > /dev/null 2> "$OUT/pmc_sq1.err" || return 1
      timeout -k 10 400 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d "$OUT/pmc_sq2" -o pmc -- python3 $B $S > /dev/null 2> "$OUT/pmc_sq2.err" || return 1
      local sfx=""; [ "$wl" != 8192x8192 ] && sfx="_$wl"
      python scripts/make_roofline.py "$TAG" "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_sq1" "$OUT/pmc_sq2" --workload $wl --suffix "$sfx" | tail -12 || return 1
      find "$OUT/prof" -name '*kernel_stats.csv' -exec cp {} "$P/kernel_stats_bench${sfx:-_8192}.csv" \;
      find "$OUT/prof" -name '*kernel_trace.csv' -delete; find "$OUT" -path '*pmc_*' -name '*.csv' -size +2000k -delete
      if [ "$wl" = 8192x8192 ]; then
        # the bench lines AFTER the PMC passes of this session: roofline.* rests on the counters of the same build on the same box
        # (bench.py reads profiles/<PROFILE_ROUND>/roofline.json, which make_roofline.py has just written on this box)
        timeout -k 10 500 python bench.py > "$P/bench_n1.json" 2> "$OUT/bench_n1.err" || { tail -5 "$OUT/bench_n1.err"; return 1; }
        timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$P/bench_n1_driver_style.json" 2>> "$OUT/bench_n1.err" || return 1
        cut -c1-300 "$P/bench_n1.json"; echo; cut -c1-200 "$P/bench_n1_driver_style.json"; echo
      fi
      cp profiles/$TAG/roofline*.json profiles/$TAG/pmc_*.csv "$P/" ;;
    extras)
      local P="$OUT/profiles"; mkdir -p "$P"
      short() { python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], 'us/step %.2f' % (d['ms_per_step']*1e3), d['config']['loop'], d['config'].get('p2p'), (d.get('parity_check') or {}).get('ok'))" "$1"; }
      # the whole grid on THIS box, for the ratios (boxes differ by a few per cent): the default region and the driver's
      python bench.py --no-cpu-baseline --no-variants --no-secondary > "$P/whole_8192x8192.json" && short "$P/whole_8192x8192.json" || return 1
      python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-variants --no-secondary > "$P/whole_8192x8192_s20.json" && short "$P/whole_8192x8192_s20.json" || return 1
      python bench.py --workload 1024x1024 --steps 3000 --warmup 100 --no-cpu-baseline --no-secondary > "$P/bench_1024x1024.json" && short "$P/bench_1024x1024.json" || return 1
      for wl in 8192x4096 8192x2048 8192x1024 1024x128; do
        local st=300; [ $wl = 1024x128 ] && st=3000
        for ex in p2p rccl; do
          python bench.py --ring --exchange $ex --workload $wl --steps $st --warmup 30 --reps 3 --no-cpu-baseline --no-variants --no-secondary > "$P/ring_${wl}_${ex}.json" && short "$P/ring_${wl}_${ex}.json" || return 1
        done
        python bench.py --ring --exchange p2p --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-variants --no-secondary > "$P/ring_${wl}_p2p_s20.json" && short "$P/ring_${wl}_p2p_s20.json" || return 1
      done
      for wl in 8192x1024 1024x128; do
        local st=300; [ $wl = 1024x128 ] && st=3000
        python bench.py --ring --exchange rccl --step-allreduce --workload $wl --steps $st --warmup 30 --reps 3 --no-cpu-baseline --no-variants --no-secondary > "$P/ring_${wl}_rccl_step_allreduce.json" && short "$P/ring_${wl}_rccl_step_allreduce.json" || return 1
        python bench.py --workload $wl --steps $st --warmup 30 --reps 3 --no-cpu-baseline --no-secondary > "$P/single_${wl}.json" && short "$P/single_${wl}.json" || return 1
        for ex in p2p rccl; do
          rm -rf "$OUT/prof_ring_${wl}_$ex"
          timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_ring_${wl}_$ex" -o trace -- python3 bench.py --ring --exchange $ex --workload $wl --steps $st --warmup 30 --reps 1 --no-cpu-baseline --no-verify --no-variants --no-secondary --no-phases > /dev/null 2> "$OUT/prof_ring_${wl}_$ex.err" || return 1
          find "$OUT/prof_ring_${wl}_$ex" -name '*kernel_stats.csv' -exec cp {} "$P/ring_${wl}_${ex}_kernel_stats.csv" \;
          find "$OUT/prof_ring_${wl}_$ex" -name '*kernel_trace.csv' -delete
        done
      done
      LBM_FORCE_DEVICE=0 timeout -k 10 500 python bench.py --gpus 2 --steps 20 --warmup 5 > "$P/bench_2ranks_one_gpu_8192.json" && short "$P/bench_2ranks_one_gpu_8192.json" ;;
    tilebench)
      local P="$OUT/profiles"; mkdir -p "$P"
      short() { python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], 'us/step %.2f' % (d['ms_per_step']*1e3), d['config']['loop'], d['config'].get('p2p'), (d.get('parity_check') or {}).get('ok'))" "$1"; }
      # block of a 4 x 2 / 2 x 2 / 2 x 1 tiling of 8192^2 and of a 4 x 2 tiling of 1024^2, each beside the row share of the same cell count
      for pair in 2048x4096:8192x1024 4096x4096:8192x2048 4096x8192:8192x4096 256x512:1024x128; do
        local tl=${pair%%:*} rw=${pair#*:} st=300; [ $tl = 256x512 ] && st=3000
        python bench.py --ring --rank-grid 1x1 --workload $tl --steps $st --warmup 30 --reps 3 --no-cpu-baseline --no-variants --no-secondary > "$P/tile_ring_${tl}.json" && short "$P/tile_ring_${tl}.json" || return 1
        python bench.py --ring --rank-grid 1x1 --workload $tl --steps 20 --warmup 5 --no-cpu-baseline --no-variants --no-secondary > "$P/tile_ring_${tl}_s20.json" && short "$P/tile_ring_${tl}_s20.json" || return 1
        python bench.py --ring --exchange p2p --workload $rw --steps $st --warmup 30 --reps 3 --no-cpu-baseline --no-variants --no-secondary > "$P/tile_session_row_ring_${rw}.json" && short "$P/tile_session_row_ring_${rw}.json" || return 1
      done
      for tl in 2048x4096 256x512; do
        local st=300; [ $tl = 256x512 ] && st=3000
        rm -rf "$OUT/prof_tile_$tl"
        timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_tile_$tl" -o trace -- python3 bench.py --ring --rank-grid 1x1 --workload $tl --steps $st --warmup 30 --reps 1 --no-cpu-baseline --no-verify --no-variants --no-secondary --no-phases > /dev/null 2> "$OUT/prof_tile_$tl.err" || return 1
        find "$OUT/prof_tile_$tl" -name '*kernel_stats.csv' -exec cp {} "$P/tile_ring_${tl}_kernel_stats.csv" \;
        find "$OUT/prof_tile_$tl" -name '*kernel_trace.csv' -delete
      done
      LBM_FORCE_DEVICE=0 timeout -k 10 500 python bench.py --gpus 4 --rank-grid 2x2 --workload 4096x4096 --steps 20 --warmup 5 --no-secondary > "$P/bench_4ranks_one_gpu_4096_tiles_2x2.json" && short "$P/bench_4ranks_one_gpu_4096_tiles_2x2.json" || return 1
      LBM_FORCE_DEVICE=0 timeout -k 10 500 python bench.py --gpus 4 --workload 4096x4096 --steps 20 --warmup 5 --no-secondary > "$P/bench_4ranks_one_gpu_4096_rows.json" && short "$P/bench_4ranks_one_gpu_4096_rows.json" ;;
    tilewide)
      # grids much wider than tall on 8 ranks ("2-D decomposition for grids wider than tall"): a rank's share as a row block
      # (few rows: deep ghost rows cost a large share, 16 rows fall back to the one-step loop) and as a block of an 8 x 1 tiling
      local P="$OUT/profiles"; mkdir -p "$P"
      short() { python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], 'us/step %.2f' % (d['ms_per_step']*1e3), d['config']['loop'], d['config'].get('p2p'), (d.get('parity_check') or {}).get('ok'))" "$1"; }
      for pair in 2048x512:16384x64 4096x256:32768x32 8192x128:65536x16; do
        local tl=${pair%%:*} rw=${pair#*:}
        python bench.py --ring --rank-grid 1x1 --column-block --workload $tl --steps 1000 --warmup 30 --reps 3 --no-cpu-baseline --no-variants --no-secondary > "$P/wide_tile_ring_${tl}.json" && short "$P/wide_tile_ring_${tl}.json" || return 1
        python bench.py --ring --rank-grid 1x1 --workload $tl --steps 1000 --warmup 30 --reps 3 --no-cpu-baseline --no-variants --no-secondary > "$P/wide_tile_ring_${tl}_ghost_rows.json" && short "$P/wide_tile_ring_${tl}_ghost_rows.json" || return 1
        python bench.py --ring --exchange p2p --workload $rw --steps 1000 --warmup 30 --reps 3 --no-cpu-baseline --no-variants --no-secondary > "$P/wide_row_ring_${rw}.json" && short "$P/wide_row_ring_${rw}.json" || return 1
      done ;;
    decks)
      for d in 128x128 128x256 256x256 1024x1024; do
        ( cd /tmp && "$GRAFT_REPO_ROOT/mpilattice-boltzmann_amd/bin/d2q9-bgk" "$GRAFT_REPO_ROOT/tests/golden/decks/input_$d.params" "$GRAFT_REPO_ROOT/tests/golden/decks/obstacles_$d.dat" | sed -n '2,3p;6p' | tr '\n' ' '; echo "  [$d]" )
      done | tee "$OUT/cli_decks.txt" ;;
    fuzz)
      local n=${arg%%:*} seed=4; [ "$n" != "$arg" ] && seed=${arg#*:}
      timeout -k 10 1000 python scripts/fuzz_kernels.py --cases ${n:-150} --seed $seed 2>&1 | grep -v amdgpu.ids > "$OUT/fuzz_$seed.log"
      local rc=$?; tail -14 "$OUT/fuzz_$seed.log"; return $rc ;;
    timeline)
      # kernel-by-kernel timeline of short runs: rocprofv3 --kernel-trace of <bench args>, the dispatch table kept (timeline.csv: name, start, end in ns)
      rm -rf "$OUT/timeline"
      timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/timeline" -o trace -- python3 bench.py ${arg:---ring --workload 8192x1024 --steps 20 --warmup 5 --reps 3 --no-cpu-baseline --no-variants --no-secondary --no-phases --no-verify --no-power} > "$OUT/timeline.json" 2> "$OUT/timeline.err" || return 1
      python - "$OUT/timeline" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
with open(sys.argv[1] + "/../timeline.csv", "w") as out:
    out.write("kernel,queue,start_ns,end_ns\n")
    for r in rows[-400:]:
        out.write(f'{r["Kernel_Name"][:60].replace(",", ";")},{r.get("Queue_Id", "")},{r["Start_Timestamp"]},{r["End_Timestamp"]}\n')
print(len(rows), "dispatches; the last 400 in timeline.csv")
PY
      rm -rf "$OUT/timeline" ;;
    soak)
      # long runs of both native loops on an 8-GPU rank's share, parity-checked against a single-GPU run before and after; then four rank
      # processes sharing this GPU for 400 steps x 5
      local P="$OUT/profiles"; mkdir -p "$P"
      for ex in p2p rccl; do
        timeout -k 10 300 python bench.py --ring --exchange $ex --workload 8192x1024 --steps 2000 --warmup 30 --reps 3 --no-cpu-baseline --no-variants --no-secondary --no-phases > "$P/ring_soak_$ex.json" 2>> "$OUT/soak.err" || return 1
        python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], 'us/step %.2f' % (d['ms_per_step']*1e3), d['parity_check'])" "$P/ring_soak_$ex.json"
      done
      LBM_FORCE_DEVICE=0 timeout -k 10 500 python bench.py --gpus 4 --workload 4096x4096 --steps 400 --warmup 20 --reps 5 --no-secondary > "$P/bench_4ranks_one_gpu_4096_soak.json" 2>> "$OUT/soak.err" || return 1
      python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], 'us/step %.2f' % (d['ms_per_step']*1e3), d['config']['p2p'], d['parity_check']['ok'], {k: v.get('parity_ok', v.get('error')) for k, v in d['variants'].items()})" "$P/bench_4ranks_one_gpu_4096_soak.json" ;;
    ab)
      local libs=${arg%%:*} rest=""; [ "$libs" != "$arg" ] && rest=${arg#*:}
      timeout -k 10 900 python scripts/ab_libs.py $(echo $libs | tr ',' ' ') $rest 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/ab.txt" ;;
    *) echo "unknown step $name"; return 2 ;;
  esac
}

for s in "$@"; do
  step "$s" || { echo "#### step $s failed: stopping"; exit 1; }
done
echo "#### $TAG done"
