"""The N>1 path on CPU: world_size-2 `gloo` processes shard a read set exactly as bench.py does (small chunks dealt
round-robin, talc_amd.sharding.deal_chunks), take their shard's corrected records — canned here from the CPU oracle, in the (records, offsets,
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


def test_chunk_deal_covers_the_input_and_merges_back_in_order():
    """Chunk c goes to rank c mod N; the merge walks the chunks in input order whatever N and the chunk size are."""
    rng = np.random.default_rng(3)
    for n, world, chunk in ((0, 2, 4), (1, 4, 4), (41, 2, 7), (100, 3, 8), (1000, 8, None), (257, 4, 64)):
        chunks = SH.deal_chunks(n, world, chunk)
        assert sum(c for _, c in chunks) == n and all(chunks[i][0] + chunks[i][1] == chunks[i + 1][0] for i in range(len(chunks) - 1))
        recs = [bytes(rng.integers(65, 70, int(rng.integers(0, 9))).astype(np.uint8)) for _ in range(n)]
        st = rng.integers(0, 4, n).astype(np.int32)
        payloads = []
        for r in range(world):
            idx = [i for lo, cnt in SH.rank_chunks(chunks, world, r) for i in range(lo, lo + cnt)]
            offs = np.concatenate([[0], np.cumsum([len(recs[i]) for i in idx])]).astype(np.uint64)
            payloads.append(SH.pack_records(b"".join(recs[i] for i in idx), offs, st[idx]))
        seq, offs, s2 = SH.merge_in_order(payloads, chunks)
        assert seq == b"".join(recs) and s2.tolist() == st.tolist()
        assert offs.tolist() == np.concatenate([[0], np.cumsum([len(x) for x in recs])]).tolist()


def _model_cost(lengths, rng):
    """DESIGN §8's cost model of a read, in wave-cycles: an inner gap of g bases ~ 870 g + 20 000, an edge of h <= 500
    bases ~ 5 (550 h + 27 h^2), + 50 per base.  Structure drawn like a 12 %-error read's at k = 31: a solid k-mer every
    ~1 / (0.12 * 0.88^31) positions, gaps and edges exponential around that mean."""
    cost = np.zeros(len(lengths))
    mean_gap = 1.0 / (0.12 * 0.88 ** 31)
    for i, L in enumerate(lengths):
        head, tail = rng.exponential(mean_gap, 2)
        inner = max(L - head - tail, 0.0)
        ngaps = int(inner / (mean_gap + 40.0))
        gaps = rng.exponential(mean_gap, ngaps)
        c = 50.0 * L + float(np.sum(870.0 * gaps + 20000.0))
        for h in (head, tail):
            if 0 < h <= 500:
                c += 5.0 * (550.0 * h + 27.0 * h * h)
        cost[i] = c
    return cost


def test_round_robin_chunks_balance_the_model_cost_on_config5_lengths():
    """BASELINE config 5 (100 k reads, log-uniform 500 b - 20 kb) over 8 ranks: the heaviest rank's model cost stays
    within 5 % of the mean — for the input as generated, sorted by length (a common file order) and sorted by cost;
    the contiguous split by bases of rounds 1-2 does not on the sorted inputs."""
    rng = np.random.default_rng(11)
    n, world = 100_000, 8
    lengths = np.exp(rng.uniform(np.log(500), np.log(20000), n)).astype(np.int64)
    cost = _model_cost(lengths, rng)
    chunks = SH.deal_chunks(n, world)
    assert len(chunks) >= 32 * world

    def ratio_chunks(c):
        per = np.zeros(world)
        for i, (lo, cnt) in enumerate(chunks):
            per[i % world] += c[lo:lo + cnt].sum()
        return per.max() / per.mean()

    def ratio_contig(c, ln):
        b = SH.shard_bounds(ln, world)
        per = np.array([c[b[r]:b[r + 1]].sum() for r in range(world)])
        return per.max() / per.mean()

    by_len = np.argsort(lengths, kind="stable")
    by_cost = np.argsort(cost, kind="stable")
    for name, order in (("as generated", np.arange(n)), ("sorted by length", by_len), ("sorted by cost", by_cost)):
        r = ratio_chunks(cost[order])
        assert r <= 1.05, (name, r)
    assert ratio_contig(cost[by_cost], lengths[by_cost]) > 1.05   # what the static split does with an ordered file


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
    chunks = SH.deal_chunks(N, world, 5)                           # 9 chunks, interleaved over the two ranks
    mine = SH.rank_chunks(chunks, world, rank)
    parts = [S.reads(lo, cnt) for lo, cnt in mine]
    bases = np.concatenate([p[0] for p in parts])
    offs = np.concatenate([[0], np.cumsum(np.concatenate([np.diff(p[1].astype(np.int64)) for p in parts]))]).astype(np.uint64)
    out, oo, st = tab.correct_batch(bases, offs, nthreads=2)      # canned fetch_corrected result of this shard
    payload = SH.pack_records_device(torch, torch.from_numpy(out.copy()), oo, st)
    assert bytes(payload.numpy()) == bytes(SH.pack_records(out.tobytes(), oo, st))
    G = SH.RecordGatherer(dist, rank, world, "cpu", dst=0)
    for step in range(3):                                          # kept buffers: every step gives the same payloads
        got = G.gather(payload)
        assert (got is None) == (rank != 0)
    if rank == 0:
        assert G.host_syncs == 3                                    # one host read of the sizes per step, no more
        seq, o, s = SH.merge_in_order([g.numpy() for g in got], chunks)
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
