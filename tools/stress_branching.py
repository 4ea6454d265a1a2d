"""Development tool: parity of the HIP path with the oracle beyond the test suite's seeds (branching graphs 101-105, unique
sequence and parameter variants 201-207).
    python tools/stress_branching.py [case ...]      (TALC_LIB selects the library)
"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import numpy as np
import parity_util as PU

from stress_cases import CASES, half_corrected


def main():
    which = [int(x) for x in sys.argv[1:]] or sorted(CASES)
    n_reads = int(os.environ.get("STRESS_READS", "1500"))
    bad_total = 0
    for c in which:
        kw, pkw = CASES[c]
        t0 = time.time()
        pair = PU.Pair(**kw, **pkw)
        pair.upload(0)
        bases, offs = pair.reads(0, n_reads)
        n_rate = float(os.environ.get("STRESS_N", "0"))
        if n_rate > 0:   # unknown bases in the long reads (Dna5 N: Jellyfish.cpp / Read.cpp treat the k-mers over them as absent)
            rng = np.random.default_rng(kw["seed"])
            bases = np.array(bases, copy=True)
            bases[rng.random(len(bases)) < n_rate] = ord("N")
        trunc = int(os.environ.get("STRESS_TRUNC", "0"))
        noise = float(os.environ.get("STRESS_NOISE", "0"))
        if trunc > 0 or noise > 0:   # reads cut to 0..trunc bases (below K, around the 512-position tiles); extra substitutions
            rng = np.random.default_rng(kw["seed"] + 7)
            seqs = PU.seqs_of(bases, offs)
            if trunc > 0:
                seqs = [q[int(rng.integers(0, max(len(q) - 1, 1))):][:int(rng.integers(0, trunc + 1))] for q in seqs]
            if noise > 0:
                seqs = ["".join(("ACGT"[int(rng.integers(0, 4))] if rng.random() < noise else ch) for ch in q) for q in seqs]
            bases = np.frombuffer("".join(seqs).encode(), dtype=np.uint8) if any(seqs) else np.zeros(0, np.uint8)
            offs = np.zeros(len(seqs) + 1, dtype=np.uint64); offs[1:] = np.cumsum([len(x) for x in seqs])
        err_clean = float(os.environ.get("STRESS_CLEAN", "0"))
        if err_clean > 0:   # a share of the reads replaced by their own corrected form (long clean regions, tiles full of hits)
            bases, offs = half_corrected(pair, bases, offs, err_clean, kw["seed"])
        bad, (so, ost), (sg, gst) = PU.compare_correction(pair, bases, offs, nthreads=16, verbose=False)
        print(c, "k", kw["k"], "reads", len(so), "status", np.bincount(ost, minlength=4).tolist(), "mismatches", len(bad), bad[:6],
              "%.1fs" % (time.time() - t0), "(oracle %.1fs on 16 threads, HIP path %.1fs)" % pair.last_times, "lib", os.environ.get("TALC_LIB", "default"), flush=True)
        if bad and os.environ.get("STRESS_TRACE"):
            print("  first trace difference", PU.first_trace_diff(pair, bases, offs, bad[0]))
        bad_total += len(bad)
    print("TOTAL MISMATCHES", bad_total)
    return 1 if bad_total else 0


if __name__ == "__main__":
    sys.exit(main())
