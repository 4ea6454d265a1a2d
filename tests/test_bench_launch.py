"""`bench.py --gpus N` is the path's parallelism switch, as `-t N` is the reference's (main.cpp:242,247): without a
launcher it starts its own N ranks from a process that stays off the GPU, relays rank 0's one JSON line and returns the
children's worst exit code; under a launcher whose WORLD_SIZE disagrees with --gpus it refuses to run."""
import importlib.util
import json
import os
import subprocess
import sys
import time

import pytest

from conftest import has_gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _bench_module():
    spec = importlib.util.spec_from_file_location("bench_mod_launch", BENCH)
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


def test_world_size_that_differs_from_gpus_is_refused():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--steps", "1", "--warmup", "0"], env=_clean_env(WORLD_SIZE="2", RANK="0"),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert p.returncode == 2 and p.stdout == b"" and b"--gpus 4 but WORLD_SIZE=2" in p.stderr
    p = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--steps", "1", "--warmup", "0"], env=_clean_env(WORLD_SIZE="8", RANK="3"),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert p.returncode == 2 and p.stdout == b""


def test_launcher_starts_n_ranks_and_relays_rank0s_line(tmp_path, capfd):
    """The launcher itself, with a stand-in for the rank program: N children with RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR = 127.0.0.1 / one MASTER_PORT, the arguments passed through, only rank 0's stdout relayed."""
    bench = _bench_module()
    script = ("import os, sys, json\n"
              "r = int(os.environ['RANK'])\n"
              "open(os.path.join(%r, 'rank%%d' %% r), 'w').write(json.dumps({k: os.environ.get(k) for k in "
              "('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')} | {'argv': sys.argv[1:]}))\n"
              "print('noise from rank %%d' %% r if r else json.dumps({'n_gpus': int(os.environ['WORLD_SIZE'])}))\n") % str(tmp_path)
    argv = ["--gpus", "3", "--steps", "1"]
    rc = bench.launch_ranks(bench.parse(argv), argv, child=[sys.executable, "-c", script])
    out = capfd.readouterr()
    assert rc == 0
    assert out.out.strip().splitlines() == ['{"n_gpus": 3}']
    seen = [json.load(open(tmp_path / ("rank%d" % r))) for r in range(3)]
    assert [s["RANK"] for s in seen] == ["0", "1", "2"] and [s["LOCAL_RANK"] for s in seen] == ["0", "1", "2"]
    assert all(s["WORLD_SIZE"] == "3" and s["MASTER_ADDR"] == "127.0.0.1" and s["argv"] == argv for s in seen)
    assert len({s["MASTER_PORT"] for s in seen}) == 1


def test_launcher_returns_the_worst_code_and_ends_the_survivors(capfd):
    """A rank that fails must not leave the others waiting in a collective: they are ended, the code is the failure's."""
    bench = _bench_module()
    script = ("import os, sys, time\n"
              "r = int(os.environ['RANK'])\n"
              "if r == 1: sys.exit(7)\n"
              "time.sleep(600)\n")
    argv = ["--gpus", "2"]
    t0 = time.time()
    rc = bench.launch_ranks(bench.parse(argv), argv, child=[sys.executable, "-c", script])
    assert rc == 7 and time.time() - t0 < 60
    assert capfd.readouterr().out == ""


@pytest.mark.gpu
@pytest.mark.skipif(not has_gpu(), reason="needs a GPU")
def test_bench_gpus_2_launches_its_own_ranks_rehearsal():
    """The command of the driver's N > 1 run, without torchrun, on the one-GPU box (TALC_BENCH_REHEARSAL=1: both ranks on
    device 0 over gloo): exactly one stdout line, n_gpus 2, every read of the workload gathered and merged on rank 0."""
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--reads", "20000", "--kmers", "2000000", "--steps", "1", "--warmup", "0"],
                       env=_clean_env(TALC_BENCH_REHEARSAL="1"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-2000:]
    lines = p.stdout.decode().strip().splitlines()
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["gathered_reads_on_rank0"] == 20000 and d["config"]["reads_total"] == 20000
    assert "own child ranks" in d["config"]["launched_by"] and "REHEARSAL" in d["data"]
    assert d["config"]["gather_host_reads_per_step_rank0"] == 2      # one host read per half-shard gather
