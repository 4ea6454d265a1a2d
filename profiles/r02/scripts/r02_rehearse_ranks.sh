#!/bin/bash
# the N > 1 path of bench.py on the one-GPU box: 2 and 4 ranks on device 0 over gloo (TALC_BENCH_REHEARSAL=1), reduced workload
set -o pipefail
O=gpurun_out
export TALC_BENCH_REHEARSAL=1
for n in 2 4; do
  timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29510 + n)) \
    bench.py --gpus $n --steps 2 --warmup 1 --reads 40000 --kmers 5000000 > $O/rehearse_n$n.json 2> $O/rehearse_n$n.err || { tail -20 $O/rehearse_n$n.err; exit 1; }
  python -c "import json; d=json.loads(open('$O/rehearse_n$n.json').read().strip().splitlines()[-1]); print('n=$n', d['n_gpus'], d['value'], d['ms_per_step'], d['config']['reads_rank0'], d['config']['gathered_reads_on_rank0'], d['data'])"
done
