"""The N>1 path on CPU: world_size-2 `gloo` processes shard a read set exactly as bench.py does (shard_bounds over the
read lengths), take their shard's corrected records — canned here from the CPU oracle, in the (records, offsets,
status) form talc_batch_fetch_corrected returns; the GPU kernels themselves are covered by the -m gpu tests — pack
them with the real payload code, gather them on rank 0 and merge: rank 0 must hold what correcting the whole read
set in one piece gives, in input order."""
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
    sys.path.insert(0, os.path.join(%r, "tests"))
    from talc_amd import sharding as SH
    from talc_amd.synth import Synth
    import oracle_lib as O
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # the same synthetic workload on every rank (every read depends on (seed, index) only)
    S = Synth(target_kmers=60_000, k=21, seed=77)
    keys, counts = S.dump_arrays()
    q = O.params(k=21)
    tab = O.OracleTable(q, O.OracleTable.FLAT)
    tab.insert_packed(keys, counts)
    tab.decolour()
    N = 41
    lengths = S.read_lengths(0, N)
    b = SH.shard_bounds(lengths, world)
    bases, offs = S.reads(b[rank], b[rank + 1] - b[rank])
    out, oo, st = tab.correct_batch(bases, offs, nthreads=2)      # canned fetch_corrected result of this shard
    payload = SH.pack_records_device(torch, torch.from_numpy(out.copy()), oo, st)
    assert bytes(payload.numpy()) == bytes(SH.pack_records(out.tobytes(), oo, st))
    got = SH.gather_records(payload, dist, rank, world, dst=0)
    if rank == 0:
        seq, o, s = SH.merge_in_order([g.numpy() for g in got])
        wb, wo = S.reads(0, N)
        e_out, e_oo, e_st = tab.correct_batch(wb, wo, nthreads=2)  # the whole set in one piece
        assert seq == e_out.tobytes() and o.tolist() == e_oo.tolist() and s.tolist() == e_st.tolist()
        assert (e_st == 0).sum() > N // 2
        print("GATHER_OK", len(s))
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()
""")


def test_two_rank_gloo_gather_restores_input_order(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % (ROOT, ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2")
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "GATHER_OK 41" in outs[0]
