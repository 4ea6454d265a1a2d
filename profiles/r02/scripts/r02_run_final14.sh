#!/bin/bash
set -o pipefail
O=gpurun_out
bash profiles/collect.sh r02 final14 > $O/final14_collect.log 2>&1; echo "collect rc=$?"; tail -2 $O/final14_collect.log
python bench.py --config 3 --steps 3 --warmup 1 --no-cpu --no-paralog > $O/final14_bench_c3.json 2> $O/final14_bench_c3.err || exit 1
python bench.py --config 5 --steps 3 --warmup 1 --no-cpu --no-paralog > $O/final14_bench_c5.json 2> $O/final14_bench_c5.err || exit 1
python bench.py --config 4 --steps 3 --warmup 1 --no-cpu --no-paralog --no-h2h > $O/final14_bench_c4.json 2> $O/final14_bench_c4.err || exit 1
for c in 3 4 5; do python -c "import json; d=json.load(open('$O/final14_bench_c$c.json')); print('config $c', d['value'], d['ms_per_step'], d['kernels_ms'], (d.get('host_to_host') or {}).get('value'))"; done
TALC_LIB=$PWD/talc_amd/_build/libtalc_hip_prof.so TALC_PROF_PRINT=1 python bench.py --steps 1 --warmup 1 --no-cpu --no-h2h --no-paralog > $O/final14_prof.json 2> $O/final14_prof.err || exit 1
grep "prof\]" $O/final14_prof.err | tail -41 > $O/final14_category_profile_config2.txt
TALC_LIB=$PWD/talc_amd/_build/libtalc_hip_prof.so TALC_PROF_PRINT=1 TALC_PROF_SLOW=1 python bench.py --config 5 --steps 1 --warmup 1 --no-cpu --no-h2h --no-paralog > $O/final14_prof5.json 2> $O/final14_prof5.err || exit 1
(grep "prof\]" $O/final14_prof5.err | tail -41; grep "slow\]" $O/final14_prof5.err | tail -26) > $O/final14_category_profile_config5.txt
tail -2 $O/final14_category_profile_config2.txt
