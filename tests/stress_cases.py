"""Seed / parameter sets of the stress comparison (HIP path against the oracle, read by read) beyond the suite's own
fixtures: branching transcriptomes (1xx, 3xx) and unique-sequence / junction / reverse / parameter variants (2xx).
`tools/stress_branching.py` runs any of them at any size; `tests/test_gpu_stress.py` draws a bounded sample of them in
the -m gpu suite (the one parity bug of round 3 passed every fixed fixture and was found by set 103)."""
import numpy as np

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
    304: (dict(target_kmers=350_000, k=29, seed=304, synth_kw=dict(paralog_frac=0.5, paralog_div=0.04)), dict()),
    # parameters at the ends of their ranges
    401: (dict(target_kmers=300_000, k=21, seed=401, synth_kw=dict(paralog_frac=0.5, paralog_div=0.04)), dict(max_nb_competing_paths=1, max_nb_inner_paths=1)),
    402: (dict(target_kmers=300_000, k=18, seed=402, synth_kw=dict(paralog_frac=0.3, paralog_div=0.05)), dict(check_interval=1, window_size=1)),
    403: (dict(target_kmers=300_000, k=22, seed=403, synth_kw=dict(paralog_frac=0.4, paralog_div=0.03)), dict(sr_error_rate=1.5, alpha=0.5, min_count=5)),
    404: (dict(target_kmers=300_000, k=30, seed=404), dict(max_start_anchors=1, min_start_anchors=1, max_border_length=50, allowed_failure_rate=0.9, max_nb_border_failures=0)),
    405: (dict(target_kmers=300_000, k=21, seed=405, junctions=True, synth_kw=dict(paralog_frac=0.4, paralog_div=0.05)), dict(max_in_count=40, coloured_count_thr=5, max_nb_border_paths=2, min_inner_score=0.95, min_border_score=0.95)),
}


def half_corrected(pair, bases, offs, share, seed, nthreads=16):
    """A share of the reads replaced by their own corrected form (the oracle's): long clean regions over a branching
    graph, tiles full of hits — a second correction pass over already corrected reads (DESIGN: the cost cliff)."""
    import parity_util as PU
    o_out, o_off, _ = pair.otab.correct_batch(bases, offs, nthreads=nthreads)
    seqs, cor = PU.seqs_of(bases, offs), PU.seqs_of(o_out, o_off)
    rng = np.random.default_rng(seed + 1)
    pick = rng.random(len(seqs)) < share
    seqs = [c if p else q for q, c, p in zip(seqs, cor, pick)]
    nb = np.frombuffer("".join(seqs).encode(), dtype=np.uint8)
    no = np.zeros(len(seqs) + 1, dtype=np.uint64)
    no[1:] = np.cumsum([len(x) for x in seqs])
    return nb, no
