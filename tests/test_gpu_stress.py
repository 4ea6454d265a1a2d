"""-m gpu: a bounded draw from the stress comparison (tests/stress_cases.py; the full tool is tools/stress_branching.py)
and BASELINE config 1 at its stated size.  Every read of every case is compared with the oracle, sequence and status."""
import os
import subprocess

import numpy as np
import pytest

import parity_util as PU
from stress_cases import CASES, half_corrected
from talc_amd import build as B
from talc_amd.synth import Synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _pair(case):
    kw, pkw = CASES[case]
    pair = PU.Pair(**kw, **pkw)
    pair.upload(0)
    return pair, kw


@pytest.mark.parametrize("case,n_reads", [(103, 1500), (303, 1500), (304, 1500), (401, 1500), (403, 1500)],
                         ids=["branching-MAXB8-CI4", "branching-junctions-MINC3-k24", "branching-k29", "MAXB1-INNER1", "SR_ERROR_RATE-1.5-MINC5"])
def test_stress_draw_matches_oracle(case, n_reads):
    """1500 reads from five of the tool's sets: MAX_NB_BRANCHES != 7 with CHECK_INTERVAL 4 (the set that found round 3's
    bug), junction colours with MIN_COUNT 3, K = 29, and two with parameters at the ends of their ranges (one competing path
    and one inner path: every fork goes through scoring; an SR error rate above 1: lambda_noise exceeds the count)."""
    pair, kw = _pair(case)
    bases, offs = pair.reads(0, n_reads)
    bad, (so, ost), _ = PU.compare_correction(pair, bases, offs, nthreads=16, verbose=False)
    assert not bad, (case, bad[:5])
    assert int(np.bincount(ost, minlength=4)[0]) > 0.85 * n_reads         # the draw really is corrected reads


def test_edge_anchors_run_by_other_waves_match_oracle(monkeypatch):
    """The start anchors of a head / tail search handed to waves that have run out of reads (talc_kernels_search.h, "edge
    tasks"): 1200 reads over a branching graph (set 102, K = 25) with every edge published from the start of its search,
    with the in-order redo forced for every anchor another wave has run, with the default thresholds, as the batch itself
    decides, with the tasks off, with few wave slots (the queue runs dry while most reads are still in their inner gaps)
    and every wave staying to the end, and with the first round's rule alone — each against the oracle's records."""
    pair, kw = _pair(102)
    bases, offs = pair.reads(0, 1200)
    o_out, o_off, o_st = pair.otab.correct_batch(bases, offs, nthreads=16)
    so = PU.seqs_of(o_out, o_off)
    on = {"TALC_EDGE_TASKS": "1"}
    for env in (dict(on, TALC_EDGE_TASK_MIN="0", TALC_EDGE_TASK_HEAVY="0"), dict(on, TALC_EDGE_TASK_MIN="0", TALC_TEST_EDGE_REDO="1"),
                on, {}, {"TALC_EDGE_TASKS": "0"}, dict(on, TALC_EDGE_TASK_MIN="0", TALC_SEARCH_SLOTS="700", TALC_EDGE_LINGER_MOD="1"),
                dict(on, TALC_EDGE_TASK_ROUNDS="1", TALC_EDGE_TASK_HEAVY="100")):
        for k in ("TALC_EDGE_TASKS", "TALC_EDGE_TASK_MIN", "TALC_EDGE_TASK_HEAVY", "TALC_EDGE_TASK_ROUNDS", "TALC_TEST_EDGE_REDO",
                  "TALC_SEARCH_SLOTS", "TALC_EDGE_LINGER_MOD"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        g_out, g_off, g_st = pair.ctx.correct(bases, offs)
        sg = PU.seqs_of(g_out, g_off)
        bad = [i for i in range(len(so)) if so[i] != sg[i] or int(o_st[i]) != int(g_st[i])]
        assert not bad, (env, bad[:5])


def test_gaps_beyond_2047_bases_over_a_branching_graph_match_oracle():
    """300 reads of set 105 (K = 31, 40 % paralogs, reads to 20 kb): inner gaps of 2-3 kb walked with several Trails, whose
    scoreBridges alignments are continued in the kept rows' wide instances (references of 2048-8191 bases) — the walks
    that took tens of seconds of one wave while every Trail was aligned from scratch."""
    pair, kw = _pair(105)
    bases, offs = pair.reads(0, 300)
    bad, (so, ost), _ = PU.compare_correction(pair, bases, offs, nthreads=16, verbose=False)
    assert not bad, bad[:5]
    assert max(len(q) for q in so) > 10_000 and pair.last_times[1] < 8.0   # (the HIP path with its first allocations: under a second; 14 s before the wide rows)


def test_half_corrected_reads_over_a_branching_graph_match_oracle():
    """100 reads of set 101, every second one replaced by its own corrected form: nearly clean reads over a branching
    graph (anchor lists of a whole kilobase region, Explorer.cpp:493-543; x-drops of several hundred)."""
    pair, kw = _pair(101)
    bases, offs = pair.reads(0, 100)
    bases, offs = half_corrected(pair, bases, offs, 0.5, kw["seed"])
    bad, _, _ = PU.compare_correction(pair, bases, offs, nthreads=16, verbose=False)
    assert not bad, bad[:5]


def test_config1_at_its_stated_size(tmp_path):
    """BASELINE configs[0]: 1 k synthetic ONT-like reads (~2 kb, 12 % error) + a 5 M-entry k = 21 dump, the CPU side with
    -t 8.  Every read through the C ABI against the oracle, and the CLI's four files against the reference driver's
    restatement (oracle/talc_ref_main.cpp), byte for byte."""
    S = Synth(target_kmers=5_000_000, k=21, seed=1)
    dump, fa = str(tmp_path / "sr.dump"), str(tmp_path / "reads.fa")
    S.write_dump(dump)
    S.write_fasta(fa, 0, 1000)
    pair = PU.Pair(target_kmers=5_000_000, k=21, seed=1)
    pair.upload(0)
    bases, offs = pair.reads(0, 1000)
    bad, (so, ost), _ = PU.compare_correction(pair, bases, offs, nthreads=8, verbose=False)
    assert not bad, bad[:5]
    assert int(np.bincount(ost, minlength=4)[0]) > 950
    B.build_cli()
    talc, ref = os.path.join(B.OUT, "talc"), os.path.join(ROOT, "oracle", "_build", "talc_ref")
    a = subprocess.run([talc, fa, "-k", "21", "-SR", dump, "-o", "gpu"], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    b = subprocess.run([ref, fa, "-k", "21", "-SR", dump, "-o", "ref", "-t", "8"], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert a.returncode == 0, a.stderr.decode()[-800:]
    assert b.returncode == 0, b.stderr.decode()[-800:]
    for ext in (".fa", ".log", ".stats_basics.txt"):
        ga, gb = open(str(tmp_path / "gpu") + ext, "rb").read(), open(str(tmp_path / "ref") + ext, "rb").read()
        assert ga == gb, ext
    assert open(str(tmp_path / "gpu.fa"), "rb").read().count(b">") == 1000
    # the CLI's records are the C ABI's
    recs = [l for l in open(str(tmp_path / "gpu.fa")).read().split(">")[1:]]
    assert ["".join(r.split("\n")[1:]) for r in recs] == so
