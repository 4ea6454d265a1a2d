#!/bin/bash
# the whole program (talc CLI: text dump + FASTA -> <o>.fa) on configs 3 and 5: bench.py's end_to_end leg alone
# (main.cpp:213-236,311-313 is the reference's own split: loading / building the graph, then the correction)
set -o pipefail
O=gpurun_out
mkdir -p $O
df -h /tmp . | tail -3
for c in "$@"; do
  timeout -k 10 1000 python3 bench.py --config $c --steps 1 --warmup 0 --no-cpu --no-h2h --no-paralog > $O/r04_e2e_c$c.json 2> $O/r04_e2e_c$c.err || { tail -5 $O/r04_e2e_c$c.err; exit 1; }
  python3 -c "
import json; d=json.loads(open('$O/r04_e2e_c$c.json').read().strip().splitlines()[-1]); e=d.get('end_to_end',{})
print('config $c', json.dumps({k:e.get(k) for k in ('wall_s','value','split_s','table_build_detail','dump_bytes','fasta_bytes','output_fa_bytes','inputs_generated_in_s','error')}))"
done
