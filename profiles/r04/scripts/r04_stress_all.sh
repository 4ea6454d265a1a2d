#!/bin/bash
# the round's closing stress run: every set at 6000 reads, then variants (N bases, truncated reads, extra noise, forced retry
# stages, walk tables / kept rows / fork_step / by-lane children switched off) on a few sets
O=gpurun_out
S=profiles/r04/scripts/r04_stress.sh
STRESS_READS=6000 bash $S 101 102 103 104 105 201 202 203 204 205 206 207 301 302 303 304 || exit 1
STRESS_READS=3000 STRESS_N=0.003 bash $S 101 103 201 203 303 || exit 1
STRESS_READS=4000 STRESS_TRUNC=900 bash $S 101 201 204 || exit 1
STRESS_READS=3000 STRESS_NOISE=0.08 bash $S 102 202 || exit 1
STRESS_READS=2500 TALC_TEST_TINY_CAPS=1 bash $S 101 103 201 304 || exit 1
STRESS_READS=2500 TALC_WALK=0 bash $S 101 201 || exit 1
STRESS_READS=2500 TALC_NO_ROWS=1 bash $S 103 || exit 1
STRESS_READS=2500 TALC_NO_FORKSTEP=1 bash $S 101 201 || exit 1
STRESS_READS=2500 TALC_CHILDREN_SEQ=1 bash $S 101 301 || exit 1
STRESS_READS=300 STRESS_CLEAN=0.5 bash $S 101 207 || exit 1
grep -c "mismatches 0" $O/r04_stress.log; grep "TOTAL" $O/r04_stress.log | sort | uniq -c
