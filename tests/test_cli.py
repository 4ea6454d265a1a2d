"""The drop-in `talc` CLI (talc_amd/csrc/talc_main.cpp, C++ host over the C ABI) against the
oracle's restatement of the reference driver (oracle/talc_ref_main.cpp): same option table, same
four output files (main.cpp:83-325, Settings.cpp:160-185, io.cpp:50-111, Read.cpp:394-415)."""
import os
import subprocess

import pytest

import parity_util as PU
from talc_amd import build as B
from talc_amd.synth import Synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TALC = os.path.join(B.OUT, "talc")
TALC_REF = os.path.join(ROOT, "oracle", "_build", "talc_ref")


@pytest.fixture(scope="module")
def cli():
    B.build_cli()
    assert os.path.exists(TALC) and os.path.exists(TALC_REF)
    return TALC


@pytest.fixture(scope="module")
def data(tmp_path_factory):
    d = tmp_path_factory.mktemp("clidata")
    S = Synth(target_kmers=150_000, k=21, seed=77)
    S.write_dump(str(d / "sr.dump"))
    S.write_junctions(str(d / "junc.dump"))
    S.write_fasta(str(d / "reads.fa"), 0, 60)
    # the same reads as FASTQ with multi-line sequences
    with open(d / "reads.fa") as f, open(d / "reads.fq", "w") as g:
        lines = f.read().splitlines()
        for i in range(0, len(lines), 2):
            s = lines[i + 1]
            g.write("@" + lines[i][1:] + "\n" + s + "\n+\n" + "I" * len(s) + "\n")
    return d


def run(exe, args, cwd):
    return subprocess.run([exe] + args, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)


def files(prefix):
    out = {}
    for ext in (".fa", ".log", ".config.txt", ".stats_basics.txt"):
        p = prefix + ext
        out[ext] = open(p, "rb").read() if os.path.exists(p) else None
    return out


def test_cli_parse_errors_and_version(cli, tmp_path):
    assert run(cli, [], tmp_path).returncode == 1
    assert run(cli, ["reads.fa", "-SR", "x"], tmp_path).returncode == 1                     # -k required
    assert run(cli, ["reads.fa", "-k", "21"], tmp_path).returncode == 1                     # -SR required
    assert run(cli, ["reads.fa", "-k", "17", "-SR", "x"], tmp_path).returncode == 1         # range 18..31
    assert run(cli, ["reads.fa", "-k", "21", "-SR", "x", "--MIN_COUNT", "1"], tmp_path).returncode == 1
    assert run(cli, ["reads.fa", "-k", "21", "-SR", "x", "-qm", "kmc"], tmp_path).returncode == 1
    assert run(cli, ["reads.fa", "-k", "21", "-SR", "x", "--bogus"], tmp_path).returncode == 1
    r = run(cli, ["--version"], tmp_path)
    assert r.returncode == 0 and b"1.01" in r.stdout
    assert run(cli, ["--help"], tmp_path).returncode == 0


def test_cli_config_and_stats_files_match_reference_text(cli, data, tmp_path):
    """Both files are written before anything else happens (main.cpp:203-204)."""
    args = [str(data / "missing.fa"), "-k", "21", "-SR", str(data / "sr.dump"), "--MIN_INNER_SCORE", "0.55", "--WINDOW_SIZE", "11",
            "-j", str(data / "junc.dump")]
    a = run(cli, args + ["-o", "gpu"], tmp_path)
    b = run(TALC_REF, args + ["-o", "ref"], tmp_path)
    assert a.returncode == 0 and b.returncode == 0                   # unreadable input: "ISSUE WITH INPUT FILES", exit 0
    assert b"ISSUE WITH INPUT FILES" in a.stdout
    fa, fb = files(str(tmp_path / "gpu")), files(str(tmp_path / "ref"))
    assert fa[".config.txt"].replace(b"gpu", b"ref") == fb[".config.txt"]
    assert fa[".stats_basics.txt"] == fb[".stats_basics.txt"]
    assert b"KmerSize=21" in fa[".config.txt"] and b"MIN_INNER_SCORE=0.55" in fa[".config.txt"]
    assert fa[".fa"] is None


def test_cli_jellyfish2_mode_is_a_dead_path_like_the_reference(cli, data, tmp_path):
    """-qm jellyfish2: the reference's dispatch strings never match (Jellyfish.cpp:302 vs main.cpp:123),
    so every read longer than K logs 'No solid kmer could be found.' and passes through."""
    args = [str(data / "reads.fa"), "-k", "21", "-SR", str(data / "sr.dump"), "-qm", "jellyfish2"]
    a = run(cli, args + ["-o", "gpu"], tmp_path)
    b = run(TALC_REF, args + ["-o", "ref"], tmp_path)
    assert a.returncode == 0 and b.returncode == 0, (a.stderr, b.stderr)
    fa, fb = files(str(tmp_path / "gpu")), files(str(tmp_path / "ref"))
    assert fa[".fa"] == fb[".fa"] and fa[".log"] == fb[".log"]
    assert fa[".log"].count(b"No solid kmer could be found.") == 60


def test_cli_streaming_reader_formats_and_batches(cli, data, tmp_path):
    """The streaming reader and the ordered batch writer, without a GPU (the pass-through of -qm jellyfish2 goes through
    the same pipeline): FASTQ, FASTA wrapped at 60 columns with blank lines and CRLF, batches of 1, 7 and everything —
    the records come out in input order and identical to the one-batch FASTA run."""
    fa = (data / "reads.fa").read_text().splitlines()
    wrapped = tmp_path / "wrapped.fa"
    with open(wrapped, "w", newline="") as g:
        g.write("\r\n")
        for i in range(0, len(fa), 2):
            g.write(fa[i] + "\r\n")
            for p in range(0, len(fa[i + 1]), 60):
                g.write(fa[i + 1][p:p + 60] + "\r\n")
            g.write("\r\n")
    base = ["-k", "21", "-SR", str(data / "sr.dump"), "-qm", "jellyfish2"]
    ref = run(cli, [str(data / "reads.fa")] + base + ["-o", "one"], tmp_path)
    assert ref.returncode == 0, ref.stderr
    want = files(str(tmp_path / "one"))
    assert want[".fa"].count(b">") == 60
    for name, src, extra in (("fq", data / "reads.fq", ["--batch-reads", "7"]), ("wr", wrapped, ["--batch-reads", "1"]),
                             ("b13", data / "reads.fa", ["--batch-reads", "13"])):
        r = run(cli, [str(src)] + base + extra + ["-o", name], tmp_path)
        assert r.returncode == 0, r.stderr
        got = files(str(tmp_path / name))
        assert got[".fa"] == want[".fa"] and got[".log"] == want[".log"], name
    # a file that is neither FASTA nor FASTQ: the reference prints and leaves with 0 (main.cpp:219,323)
    (tmp_path / "bad.txt").write_text("hello\nworld\n")
    r = run(cli, [str(tmp_path / "bad.txt")] + base, tmp_path)
    assert r.returncode == 0 and b"ISSUE WITH INPUT FILES" in r.stdout


def test_cli_empty_table_aborts_with_exit_1(cli, data, tmp_path):
    (tmp_path / "empty.dump").write_text("ACGTACGTACGTACGTACGTA 1\n")     # below MIN_COUNT
    r = run(cli, [str(data / "reads.fa"), "-k", "21", "-SR", str(tmp_path / "empty.dump")], tmp_path)
    assert r.returncode == 1 and b"The de Bruijn Graph is empty" in r.stdout


@pytest.mark.skipif(PU.T.device_count() > 0, reason="only meaningful on a host without a GPU")
def test_cli_fails_loudly_without_gpu(cli, data, tmp_path):
    r = run(cli, [str(data / "reads.fa"), "-k", "21", "-SR", str(data / "sr.dump")], tmp_path)
    assert r.returncode == 2 and b"no CPU fallback" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["-j", "JUNC"], ["-rev"], ["--MIN_COUNT", "3", "--MAX_NB_BRANCHES", "5", "--batch-reads", "7"],
                                   ["--read-stats", "--batch-reads", "7"]],
                         ids=["default", "junctions", "reverse", "params+small-batches", "read-stats"])
def test_cli_end_to_end_files_identical_to_reference_driver(cli, data, tmp_path, extra):
    extra = [str(data / "junc.dump") if x == "JUNC" else x for x in extra]
    reads = str(data / "reads.fq") if "-rev" in extra else str(data / "reads.fa")
    args = [reads, "-k", "21", "-SR", str(data / "sr.dump")] + extra
    ref_args = [x for x in args if x not in ("--batch-reads", "7")]
    a = run(cli, args + ["-o", "gpu"], tmp_path)
    b = run(TALC_REF, ref_args + ["-o", "ref", "-t", "8"], tmp_path)
    assert a.returncode == 0, a.stderr.decode()
    assert b.returncode == 0, b.stderr.decode()
    fa, fb = files(str(tmp_path / "gpu")), files(str(tmp_path / "ref"))
    assert fa[".fa"] == fb[".fa"]
    assert fa[".log"] == fb[".log"]
    assert fa[".config.txt"].replace(b"gpu", b"ref") == fb[".config.txt"]
    assert fa[".stats_basics.txt"] == fb[".stats_basics.txt"]
    if "--read-stats" in extra:      # the rows of Read::outputBasicReadStats (Read.cpp:418-433), one per read longer than K
        rows = fa[".stats_basics.txt"].split(b"\n")[2:]
        assert len(rows) >= 55 and all(len(r.split(b"\t")) == 5 for r in rows)
    lines = fa[".fa"].splitlines()
    assert len([l for l in lines if l.startswith(b">")]) == 60 and max(len(l) for l in lines if not l.startswith(b">")) == 70


@pytest.mark.gpu
def test_cli_two_gpu_sharder_rehearsed_on_one_device_gives_the_one_gpu_files(cli, tmp_path):
    """The in-process `--gpus N` sharder (main.cpp:247-308 replaced): TALC_FAKE_GPUS=2 runs it as on a two-GPU node —
    four workers, their own contexts and streams, batches dealt dynamically, records written in input order — with both
    logical GPUs on device 0.  <o>.fa and <o>.log must be the one-GPU run's, byte for byte, and its timing line says so."""
    S = Synth(target_kmers=150_000, k=21, seed=77)
    S.write_dump(str(tmp_path / "sr.dump"))
    S.write_fasta(str(tmp_path / "reads.fa"), 0, 900)
    args = [str(tmp_path / "reads.fa"), "-k", "21", "-SR", str(tmp_path / "sr.dump"), "--batch-reads", "37"]
    one = run(cli, args + ["-o", "one", "--gpus", "1"], tmp_path)
    assert one.returncode == 0, one.stderr.decode()
    env = dict(os.environ, TALC_FAKE_GPUS="2")
    two = subprocess.run([cli] + args + ["-o", "two", "--gpus", "2"], cwd=tmp_path, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert two.returncode == 0, two.stderr.decode()
    assert b'"gpus": 2' in two.stderr and b'"workers": 4' in two.stderr and b'"gpus": 1' in one.stderr
    assert b"correcting on 2 GPU(s)" in two.stdout
    f1, f2 = files(str(tmp_path / "one")), files(str(tmp_path / "two"))
    assert f1[".fa"] == f2[".fa"] and f1[".log"] == f2[".log"]
    assert f1[".fa"].count(b">") == 900
    # default batch size: at least two batches per worker
    dflt = subprocess.run([cli] + args[:-2] + ["-o", "dflt", "--gpus", "2"], cwd=tmp_path, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert dflt.returncode == 0 and files(str(tmp_path / "dflt"))[".fa"] == f1[".fa"]


# ---------------------------------------------------------------- jellyfish2 query mode / .jf input
FAKE_JELLYFISH = """#!/usr/bin/env python3
# stand-in for `jellyfish dump -c [-L n] -o OUT FILE.jf` (test only): the text dump kept next to FILE.jf, filtered by -L
import os, sys
a = sys.argv[1:]
open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "calls.log"), "a").write(" ".join(a) + "\\n")
if os.environ.get("FAKE_JELLYFISH_EXIT"):
    sys.exit(int(os.environ["FAKE_JELLYFISH_EXIT"]))
assert a[0] == "dump" and "-c" in a and "-o" in a, a
low = int(a[a.index("-L") + 1]) if "-L" in a else 0
with open(a[-1] + ".txt") as f, open(a[a.index("-o") + 1], "w") as g:
    for line in f:
        t = line.split()
        if len(t) >= 2 and int(t[1]) >= low:
            g.write(line)
"""


@pytest.fixture(scope="module")
def jfdata(data, tmp_path_factory):
    """sr.jf / junc.jf in the layout talc_jf.h reads (tests/jf_writer.py), their text dumps next to them under the
    names the stand-in tool opens, and a directory holding that tool as `jellyfish`."""
    import shutil
    import jf_writer as JW
    d = tmp_path_factory.mktemp("jfcli")
    JW.dump_to_jf(str(data / "sr.dump"), str(d / "sr.jf"), 21, seed=8)
    JW.dump_to_jf(str(data / "junc.dump"), str(d / "junc.jf"), 21, seed=9)
    shutil.copy(data / "sr.dump", d / "sr.jf.txt")
    shutil.copy(data / "junc.dump", d / "junc.jf.txt")
    tool = d / "bin"
    tool.mkdir()
    (tool / "jellyfish").write_text(FAKE_JELLYFISH)
    os.chmod(tool / "jellyfish", 0o755)
    return d


def nodes_line(stdout):
    return [l for l in stdout.splitlines() if b"SR-dBG contains" in l]


def test_cli_jellyfish2_mode_builds_the_table_from_the_tool_or_the_jf(cli, data, jfdata, tmp_path):
    """Up to the table (no GPU needed): -qm jellyfish2 -jf2 DIR runs DIR/jellyfish dump -c -L MIN_COUNT once per file
    and builds the table of the text dump; without -jf2 the .jf is read natively; the temporary dumps are removed; a
    failing or missing tool ends the run with exit code 2."""
    base = [str(data / "reads.fa"), "-k", "21"]
    want = nodes_line(run(cli, base + ["-SR", str(data / "sr.dump"), "-j", str(data / "junc.dump"), "-o", "m"], tmp_path).stdout)
    assert len(want) == 1 and not want[0].endswith(b" 0 nodes.")
    jf = ["-SR", str(jfdata / "sr.jf"), "-j", str(jfdata / "junc.jf")]
    r = run(cli, base + jf + ["-qm", "jellyfish2", "-jf2", str(jfdata / "bin"), "--MIN_COUNT", "2", "-o", "t"], tmp_path)
    assert nodes_line(r.stdout) == want, (r.stdout, r.stderr)
    calls = (jfdata / "bin" / "calls.log").read_text().splitlines()
    assert calls[-2] == "dump -c -L 2 -o t.SRCounts.dump.tmp " + str(jfdata / "sr.jf")
    assert calls[-1] == "dump -c -o t.junctions.dump.tmp " + str(jfdata / "junc.jf")
    assert not [f for f in os.listdir(tmp_path) if f.endswith(".tmp")]
    r = run(cli, base + jf + ["-qm", "jellyfish2", "-o", "n"], tmp_path)         # native reader
    assert nodes_line(r.stdout) == want and b"natively" in r.stdout, (r.stdout, r.stderr)
    r = run(cli, base + jf + ["-o", "mm"], tmp_path)                             # memory mode takes a .jf too
    assert nodes_line(r.stdout) == want, (r.stdout, r.stderr)
    env = dict(os.environ, FAKE_JELLYFISH_EXIT="3")
    r = subprocess.run([cli] + base + jf + ["-qm", "jellyfish2", "-jf2", str(jfdata / "bin"), "-o", "f"], cwd=tmp_path, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 2 and b"ended with exit code 3" in r.stderr
    r = run(cli, base + jf + ["-qm", "jellyfish2", "-jf2", str(tmp_path / "nowhere"), "-o", "g"], tmp_path)
    assert r.returncode == 2 and b"cannot run" in r.stderr
    r = run(cli, base + ["-SR", str(jfdata / "sr.jf"), "-k", "25", "-o", "k"], tmp_path)   # -k disagrees with the file
    assert r.returncode == 2 and b"21-mers" in r.stderr


@pytest.mark.gpu
def test_cli_jellyfish2_mode_end_to_end_equals_the_memory_mode_run(cli, data, jfdata, tmp_path):
    """The three routes to the counts of a .jf give the records and the log of the memory-mode run on its text dump
    (which test_cli_end_to_end_files_identical_to_reference_driver ties to the reference driver)."""
    base = [str(data / "reads.fa"), "-k", "21"]
    m = run(cli, base + ["-SR", str(data / "sr.dump"), "-j", str(data / "junc.dump"), "-o", "m"], tmp_path)
    assert m.returncode == 0, m.stderr.decode()
    want = files(str(tmp_path / "m"))
    assert want[".fa"].count(b">") == 60 and b"No solid kmer" not in (want[".log"] or b"")     # a real table, not the empty run
    jf = ["-SR", str(jfdata / "sr.jf"), "-j", str(jfdata / "junc.jf")]
    for name, extra in (("tool", ["-qm", "jellyfish2", "-jf2", str(jfdata / "bin")]), ("native", ["-qm", "jellyfish2"]), ("mem", [])):
        r = run(cli, base + jf + extra + ["-o", name], tmp_path)
        assert r.returncode == 0, (name, r.stderr.decode())
        got = files(str(tmp_path / name))
        assert got[".fa"] == want[".fa"] and got[".log"] == want[".log"], name
