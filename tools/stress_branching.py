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

CASES = {
    101: (dict(target_kmers=400_000, k=21, seed=101, synth_kw=dict(paralog_frac=0.6, paralog_div=0.02)), dict()),
    102: (dict(target_kmers=400_000, k=25, seed=102, synth_kw=dict(paralog_frac=0.6, paralog_div=0.05)), dict()),
    103: (dict(target_kmers=300_000, k=21, seed=103, synth_kw=dict(paralog_frac=0.8, paralog_div=0.04)), dict(max_nb_competing_paths=8, check_interval=4)),
    104: (dict(target_kmers=300_000, k=23, seed=104, synth_kw=dict(paralog_frac=0.5, paralog_div=0.08)), dict(max_nb_competing_paths=3, window_size=5)),
    105: (dict(target_kmers=500_000, k=31, seed=105, synth_kw=dict(paralog_frac=0.4, paralog_div=0.03, mixed_lengths=1)), dict()),
    # unique-sequence transcriptomes, other seeds / parameters than the suite's
    201: (dict(target_kmers=600_000, k=21, seed=201), dict()),
    202: (dict(target_kmers=600_000, k=25, seed=202), dict(min_count=3)),
    203: (dict(target_kmers=600_000, k=31, seed=203, synth_kw=dict(mixed_lengths=1)), dict()),
    204: (dict(target_kmers=500_000, k=21, seed=204, junctions=True), dict()),
    205: (dict(target_kmers=500_000, k=19, seed=205), dict(reverse=1, window_size=12)),
    206: (dict(target_kmers=500_000, k=27, seed=206), dict(alpha=1.3, sr_error_rate=0.05, check_interval=9, max_border_length=300)),
    207: (dict(target_kmers=400_000, k=21, seed=207, synth_kw=dict(paralog_frac=0.3, paralog_div=0.10)), dict(max_nb_competing_paths=6, max_nb_border_paths=3)),
    301: (dict(target_kmers=350_000, k=21, seed=301, synth_kw=dict(paralog_frac=0.9, paralog_div=0.015)), dict(max_nb_competing_paths=10)),
    302: (dict(target_kmers=350_000, k=29, seed=302, synth_kw=dict(paralog_frac=0.5, paralog_div=0.03, mixed_lengths=1)), dict(check_interval=5)),
    303: (dict(target_kmers=350_000, k=24, seed=303, junctions=True, synth_kw=dict(paralog_frac=0.4, paralog_div=0.06)), dict(min_count=3, max_nb_inner_paths=20)),
}


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
            o_out, o_off, _ = pair.otab.correct_batch(bases, offs, nthreads=16)
            seqs = PU.seqs_of(bases, offs); cor = PU.seqs_of(o_out, o_off)
            rng = np.random.default_rng(kw["seed"] + 1)
            pick = rng.random(len(seqs)) < err_clean
            seqs = [c if p else q for q, c, p in zip(seqs, cor, pick)]
            bases = np.frombuffer("".join(seqs).encode(), dtype=np.uint8)
            offs = np.zeros(len(seqs) + 1, dtype=np.uint64); offs[1:] = np.cumsum([len(x) for x in seqs])
        bad, (so, ost), (sg, gst) = PU.compare_correction(pair, bases, offs, nthreads=16, verbose=False)
        print(c, "k", kw["k"], "reads", len(so), "status", np.bincount(ost, minlength=4).tolist(), "mismatches", len(bad), bad[:6],
              "%.1fs" % (time.time() - t0), "lib", os.environ.get("TALC_LIB", "default"), flush=True)
        if bad and os.environ.get("STRESS_TRACE"):
            print("  first trace difference", PU.first_trace_diff(pair, bases, offs, bad[0]))
        bad_total += len(bad)
    print("TOTAL MISMATCHES", bad_total)
    return 1 if bad_total else 0


if __name__ == "__main__":
    sys.exit(main())
