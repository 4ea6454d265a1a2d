#!/bin/bash
# The long-gap risk term of the work-queue estimate on one box: (Q + F x forking share) x min(sum g^2, CAP^2), for config 2
# (search ms), config 5 (ms per step) and the branching workload (ms per pass).  Each setting twice, alternating.
O=gpurun_out
for rep in 1 2; do
for s in "14 0 900" "6 200 900" "6 200 100000" "10 200 900" "14 200 900" "8 400 900" "14 0 700"; do
  set -- $s
  export TALC_COST_GAPQ=$1 TALC_COST_GAPFORK=$2 TALC_COST_GAPCAP=$3
  c2=$(python3 tools/cov_bench.py --full --reps 3 2>> $O/sweep.err | sed 's/.*search //')
  pl=$(python3 bench.py --no-cpu --no-e2e --no-h2h --steps 2 --warmup 1 2>&1 >/dev/null | grep paralog | sed 's/.*workload: //; s/ ms.*//')
  c5=$(python3 bench.py --config 5 --no-cpu --no-e2e --no-h2h --no-paralog --steps 2 --warmup 1 2>/dev/null | python3 -c "import sys,json; print(round(json.loads(sys.stdin.read())['ms_per_step'],1))")
  echo "Q=$1 F=$2 CAP=$3 | c2 search $c2 | branching $pl | c5 step $c5" | tee -a $O/cost_sweep.txt
done
done
