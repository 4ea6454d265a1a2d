#!/bin/bash
# the N > 1 path of bench.py on the one-GPU box: 2 and 4 ranks on device 0 over gloo (TALC_BENCH_REHEARSAL=1), reduced
# workload; rank 0's stdout must be exactly ONE line (backend banners go to stderr)
set -o pipefail
O=gpurun_out
export TALC_BENCH_REHEARSAL=1
for n in 2 4; do
  timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29510 + n)) \
    bench.py --gpus $n --steps 2 --warmup 1 --reads 40000 --kmers 5000000 > $O/r03_rehearse_n$n.json 2> $O/r03_rehearse_n$n.err || { tail -20 $O/r03_rehearse_n$n.err; exit 1; }
  python -c "
lines=open('$O/r03_rehearse_n$n.json').read().strip().splitlines()
import json; d=json.loads(lines[-1])
print('n=$n stdout_lines', len(lines), d['n_gpus'], d['value'], d['ms_per_step'], d['config']['reads_rank0'], d['config']['gathered_reads_on_rank0'], d['config']['read_deal'], d['config'].get('gather_host_reads_per_step_rank0'), d['data'][:40])"
done
