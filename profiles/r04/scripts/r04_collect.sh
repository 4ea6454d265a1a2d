#!/bin/bash
# One collection of the round's profile artifacts (run from the repo root on the GPU box):
#   bash profiles/r04/scripts/r04_collect.sh <tag> [parts]      parts: any of  stats pmc sq configs prof rehearse  (default: all)
# -> gpurun_out/profiles/r04/<tag>_*.  Under rocprofv3 the program itself follows "--" and bench.py runs with --no-e2e (no
# child process under the profiler); counters only with --kernel-trace, one counter group per pass.
set -o pipefail
TAG=${1:-mid}
PARTS=${2:-"stats pmc sq configs prof rehearse"}
O=gpurun_out
DST=$O/profiles/r04
COLL=$O/coll_$TAG
mkdir -p $DST $COLL
export TMPDIR=/tmp
ROOT=$(pwd)
has() { [[ " $PARTS " == *" $1 "* ]]; }
QUIET="--no-cpu --no-h2h --no-paralog --no-e2e"
if has stats; then
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$COLL/stats" -o stats -- python3 "$ROOT/bench.py" --steps 10 --warmup 3 $QUIET > "$DST/${TAG}_bench_under_rocprof.json" 2> $COLL/stats.err || { tail -5 $COLL/stats.err; exit 1; }
  cp "$(find "$COLL/stats" -name '*kernel_stats.csv' | head -1)" "$DST/${TAG}_kernel_stats.csv"
  echo "[collect] kernel stats done"
fi
if has pmc; then
  for cfg in 2 3 5; do
    for C in FETCH_SIZE WRITE_SIZE; do
      timeout -k 10 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$COLL/pmc_c${cfg}_$C" -o pmc -- python3 "$ROOT/bench.py" --config $cfg --steps 1 --warmup 0 $QUIET > "$COLL/pmc_c${cfg}_$C.json" 2> "$COLL/pmc_c${cfg}_$C.err" || { echo "pmc $cfg $C failed"; tail -5 "$COLL/pmc_c${cfg}_$C.err"; exit 1; }
    done
    echo "[collect] pmc config $cfg done"
  done
fi
if has sq; then
  i=0
  for C in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$COLL/sq_$i" -o pmc -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 $QUIET > "$COLL/sq_$i.json" 2> "$COLL/sq_$i.err" || { echo "sq pass $i failed"; tail -5 "$COLL/sq_$i.err"; exit 1; }
  done
  echo "[collect] sq passes done"
fi
if has pmc || has sq; then
  python3 profiles/aggregate_counters.py "$COLL" "$DST" "$TAG" || exit 1
  mkdir -p profiles && cp $O/pmc_traffic.json $O/sq_search.json profiles/ 2>/dev/null   # (on the GPU box: so that the bench below sees them)
fi
timeout -k 10 400 python3 bench.py --steps 10 --warmup 3 > "$DST/${TAG}_bench.json" 2> $COLL/bench.err || { tail -20 $COLL/bench.err; exit 1; }
timeout -k 10 400 python3 bench.py > "$DST/${TAG}_bench_default.json" 2> $COLL/bench_default.err || { tail -20 $COLL/bench_default.err; exit 1; }
echo "[collect] bench lines done"
if has configs; then
  for c in 3 4 5; do
    timeout -k 10 500 python3 bench.py --config $c --steps 3 --warmup 1 --no-cpu --no-paralog --no-e2e > $DST/${TAG}_bench_c$c.json 2> $COLL/bench_c$c.err || { tail -5 $COLL/bench_c$c.err; exit 1; }
    echo "[collect] config $c done"
  done
fi
if has prof; then
  TALC_PROF_PRINT=1 TALC_LIB=talc_amd/_build/libtalc_hip_prof.so timeout -k 10 300 python3 tools/search_bench.py --reps 1 --no-paralog 2>&1 | grep "^\[prof\]" | awk '{k=$2" "$3} !seen[k]++' | tail -95 > $DST/${TAG}_category_profile_config2.txt
  TALC_PROF_PRINT=1 TALC_LIB=talc_amd/_build/libtalc_hip_prof.so timeout -k 10 300 python3 tools/search_bench.py --reps 1 --no-main 2>&1 | grep "^\[prof\]" | tail -95 > $DST/${TAG}_category_profile_paralog.txt
  echo "[collect] category profiles done"
fi
if has rehearse; then
  for n in 2 4; do
    TALC_BENCH_REHEARSAL=1 timeout -k 10 400 python3 bench.py --gpus $n --steps 2 --warmup 1 --reads 40000 --kmers 5000000 > $DST/${TAG}_rehearse_n$n.json 2> $COLL/rehearse_n$n.err || { tail -20 $COLL/rehearse_n$n.err; exit 1; }
  done
  echo "[collect] rehearsals done"
fi
echo "[collect] all done"
