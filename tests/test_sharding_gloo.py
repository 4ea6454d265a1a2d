"""The N>1 path on CPU: world_size-2 `gloo` processes shard a read set, "correct" their shard
(a stand-in transform here: the GPU kernels are covered by the -m gpu tests) and gather the
records on rank 0, which must see them in input order."""
import os
import subprocess
import sys
import textwrap

import numpy as np

from talc_amd import sharding as SH

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_balance_and_cover():
    rng = np.random.default_rng(0)
    lengths = rng.integers(500, 20000, 1000)
    for world in (1, 2, 3, 4, 8):
        b = SH.shard_bounds(lengths, world)
        assert b[0] == 0 and b[-1] == 1000 and len(b) == world + 1 and all(b[i] <= b[i + 1] for i in range(world))
        per = [int(lengths[b[r]:b[r + 1]].sum()) for r in range(world)]
        assert max(per) - min(per) <= 2 * int(lengths.max())
    assert SH.shard_bounds([], 4) == [0, 0, 0, 0, 0]
    assert SH.shard_bounds([5], 4)[-1] == 1


def test_pack_unpack_merge_roundtrip():
    a = SH.pack_records(b"ACGTAC", [0, 4, 6], [0, 2])
    b = SH.pack_records(b"", [0], [])
    c = SH.pack_records(b"TTT", [0, 3], [3])
    seq, offs, st = SH.merge_in_order([a, b, c])
    assert seq == b"ACGTACTTT" and offs.tolist() == [0, 4, 6, 9] and st.tolist() == [0, 2, 3]


WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    import torch
    import torch.distributed as dist
    sys.path.insert(0, %r)
    from talc_amd import sharding as SH
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(123)                      # same read set on every rank
    lengths = rng.integers(30, 400, 57)
    reads = [bytes(rng.choice(list(b"ACGT"), int(n)).astype(np.uint8)) for n in lengths]
    b = SH.shard_bounds(lengths, world)
    mine = reads[b[rank]:b[rank + 1]]
    # stand-in for the per-read correction: reverse the read, status = length %% 4
    out = [r[::-1] for r in mine]
    offs = np.concatenate([[0], np.cumsum([len(x) for x in out])]).astype(np.uint64)
    st = np.array([len(x) %% 4 for x in out], dtype=np.int32)
    payload = torch.from_numpy(SH.pack_records(b"".join(out), offs, st).copy())
    got = SH.gather_records(payload, dist, rank, world, dst=0)
    if rank == 0:
        seq, o, s = SH.merge_in_order([g.numpy() for g in got])
        exp = [r[::-1] for r in reads]
        assert seq == b"".join(exp)
        assert o.tolist() == np.concatenate([[0], np.cumsum([len(x) for x in exp])]).tolist()
        assert s.tolist() == [len(x) %% 4 for x in exp]
        print("GATHER_OK", len(exp))
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()
""")


def test_two_rank_gloo_gather_restores_input_order(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2")
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "GATHER_OK 57" in outs[0]
