"""Golden fixtures (tests/golden/*.json.gz, produced by tests/golden/make_golden.py from the
oracle): the oracle must keep reproducing them on CPU, the HIP path must reproduce them on GPU."""
import glob
import gzip
import hashlib
import json
import os

import numpy as np
import pytest

import oracle_lib as O
import parity_util as PU
from talc_amd import lib as T
from talc_amd.synth import Synth

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = sorted(glob.glob(os.path.join(HERE, "golden", "*.json.gz")))
COMP = str.maketrans("ACGTN", "TGCAN")


def load(path):
    with gzip.open(path, "rb") as f:
        return json.loads(f.read().decode())


def inputs(fx):
    S = Synth(**fx["synth"])
    keys, counts = S.dump_arrays()
    assert hashlib.sha256(keys.tobytes() + counts.tobytes()).hexdigest() == fx["dump_sha256"], "generator drifted"
    jk = jc = None
    if fx["params"].get("use_junctions"):
        jk, jc = S.junction_arrays()
        assert hashlib.sha256(jk.tobytes() + jc.tobytes()).hexdigest() == fx["junction_sha256"]
    reads = fx["reads"]
    rb = "".join(reads).encode()
    offs = np.zeros(len(reads) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(x) for x in reads])
    bases = np.frombuffer(rb, dtype=np.uint8) if rb else np.zeros(0, np.uint8)
    return keys, counts, jk, jc, bases, offs


def dna5(s):
    return "".join(ch if ch in "ACGT" else "N" for ch in s.upper())


def test_fixtures_present():
    assert len(FIXTURES) >= 8


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-8] for p in FIXTURES])
def test_oracle_reproduces_golden(path):
    fx = load(path)
    keys, counts, jk, jc, bases, offs = inputs(fx)
    q = O.params(**fx["params"])
    tab = O.OracleTable(q, O.OracleTable.FLAT)       # the other backend than the one that made the fixture
    tab.insert_packed(keys, counts)
    if jk is not None:
        tab.colour_packed(jk, jc)
    tab.decolour()
    assert len(tab) == fx["table_size"]
    out, oo, st = tab.correct_batch(bases, offs, nthreads=4)
    got = PU.seqs_of(out, oo)
    assert [int(x) for x in st] == fx["status"]
    assert got == fx["expected"]
    h = hashlib.sha256()
    for s in fx["reads"]:
        if fx["params"].get("reverse"):
            s = dna5(s).translate(COMP)[::-1]
        cov, j, nin = tab.coverage(s)
        h.update(cov.tobytes()); h.update(j.tobytes())
    assert h.hexdigest() == fx["coverage_sha256"]


def test_embedded_dump_text_roundtrip(tmp_path):
    """g1 carries its dump as text: both table builders read it from a file (buildCDBG surface)."""
    fx = load([p for p in FIXTURES if "g1_" in p][0])
    dump = str(tmp_path / "g1.dump")
    with open(dump, "w") as f:
        f.write(fx["dump_text"])
    p, q = PU.both_params(**fx["params"])
    ot = O.OracleTable(q, O.OracleTable.MAP)
    ot.build_from_files(dump)
    tt = T.Table.from_files(dump, None, p)
    assert len(ot) == len(tt) == fx["table_size"]
    reads = fx["reads"]
    rb = "".join(reads).encode()
    offs = np.zeros(len(reads) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(x) for x in reads])
    out, oo, st = ot.correct_batch(np.frombuffer(rb, dtype=np.uint8), offs, nthreads=4)
    assert PU.seqs_of(out, oo) == fx["expected"]


@pytest.mark.gpu
@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-8] for p in FIXTURES])
def test_gpu_reproduces_golden(path):
    fx = load(path)
    keys, counts, jk, jc, bases, offs = inputs(fx)
    p = T.default_params(**fx["params"])
    tab = T.Table.from_arrays(keys, counts, p)
    if jk is not None:
        tab.colour(jk, jc)
    tab.decolour_repeats()
    assert len(tab) == fx["table_size"]
    tab.upload(0)
    ctx = T.Context(tab, p, 0)
    b = ctx.batch(bases, offs)
    b.coverage()
    c, j, ko, nin = b.fetch_coverage()
    h = hashlib.sha256()
    for i in range(len(fx["reads"])):
        h.update(c[int(ko[i]):int(ko[i + 1])].tobytes()); h.update(j[int(ko[i]):int(ko[i + 1])].tobytes())
    assert h.hexdigest() == fx["coverage_sha256"]
    b.correct()
    out, oo, st = b.fetch_corrected()
    assert [int(x) for x in st] == fx["status"]
    assert PU.seqs_of(out, oo) == fx["expected"]
    b.close()
    ctx.close()
    tab.close()
