#!/bin/bash
# One collection of the round's profile artifacts (run from the repo root on the GPU box):
#   bash profiles/r03/scripts/r03_collect.sh <tag>
# -> gpurun_out/profiles/r03/<tag>_*: bench (default command), kernel stats CSV (rocprofv3 --kernel-trace --stats), PMC traffic
#    (FETCH_SIZE / WRITE_SIZE, separate passes), SQ lane counters, configs 3 / 4 / 5 on one GPU, the category profile.
set -o pipefail
TAG=${1:-mid}
O=gpurun_out
bash profiles/collect.sh r03 $TAG || exit 1
bash profiles/r03/scripts/r03_lanes.sh ${TAG}_lanes > $O/${TAG}_lanes.log 2>&1 && cp $O/${TAG}_lanes_sq.json $O/profiles/r03/${TAG}_sq_counters.json
for c in 3 4 5; do
  python3 bench.py --config $c --steps 3 --warmup 1 --no-cpu --no-paralog --no-e2e > $O/profiles/r03/${TAG}_bench_c$c.json 2> $O/${TAG}_bench_c$c.err || { tail -5 $O/${TAG}_bench_c$c.err; exit 1; }
  echo "[collect] config $c done"
done
for c in 2 5; do
  TALC_PROF_PRINT=1 TALC_LIB=talc_amd/_build/libtalc_hip_prof.so python3 bench.py --config $c --steps 1 --warmup 0 --no-cpu --no-h2h --no-paralog --no-e2e 2>&1 >/dev/null | grep "^\[prof\]" | awk '!seen[$2]++' > $O/profiles/r03/${TAG}_category_profile_config$c.txt
done
echo "[collect] all done"
