"""Development tool: category profile of a handful of reads of a bench workload (the profile build of the library).
    TALC_LIB=talc_amd/_build/libtalc_hip_prof.so TALC_PROF_PRINT=1 python tools/slow_reads.py 5 49157 87728 ...
prints k_search's category profile for a batch that holds only those reads of BASELINE config <n> (or `paralog`), and their lengths,
statuses and the number of regions the structure step found."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench as B
from talc_amd import lib as T
from talc_amd.synth import Synth


def main():
    ids = [int(x) for x in sys.argv[2:]]
    if sys.argv[1] == "paralog":     # bench.py's paralog side workload
        synth = Synth(target_kmers=2_000_000, k=25, seed=77, paralog_frac=0.6, paralog_div=0.05)
        params = T.default_params(k=25)
    else:
        w = dict(B.CONFIGS[int(sys.argv[1])])
        synth = Synth(target_kmers=w["kmers"], k=w["k"], seed=0, mixed_lengths=int(w["mixed"]))
        params = T.default_params(k=w["k"], use_junctions=int(w["junctions"]))
    keys, counts = synth.dump_arrays()
    table = T.Table.from_arrays(keys, counts, params, device=0)
    table.decolour_repeats()
    table.upload(0)
    ctx = T.Context(table, params, 0)
    seqs = []
    for r in ids:
        b, o = synth.reads(r, 1)
        seqs.append(np.asarray(b))
    bases = np.concatenate(seqs)
    offs = np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype(np.uint64)
    batch = ctx.batch(bases, offs)
    for it in range(2):
        t0 = time.time()
        batch.correct()
        print("pass %d: %.1f ms  %s" % (it, 1e3 * (time.time() - t0), {k: round(v, 2) for k, v in ctx.timing().as_dict().items()}), file=sys.stderr)
    out, oo, st = batch.fetch_corrected()
    for i, r in enumerate(ids):
        print("read %d: len %d -> %d status %d" % (r, len(seqs[i]), int(oo[i + 1] - oo[i]), int(st[i])), file=sys.stderr)


if __name__ == "__main__":
    main()
