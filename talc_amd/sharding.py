"""Multi-GPU read sharding — the replacement of the reference's per-read OpenMP loop
(`#pragma omp parallel for schedule(dynamic)`, main.cpp:247-308).

Reads are independent units (the only cross-read state in the reference is a write-only
counter), so they shard with NO data-path collective: every rank corrects its reads against its
own replica of the k-mer table.  The only communication is the gather of the corrected records
to rank 0, which restores the input order (main.cpp:310 writes all records in input order).
Works with any torch.distributed backend: "nccl" (= RCCL over xGMI) on GPUs, "gloo" on CPU for
the tests.

The deal: the input is cut into chunks of a few hundred to a few thousand consecutive reads and
chunk c goes to rank c mod N (`deal_chunks`).  A read's cost is only weakly tied to its length
(an edge of h bases costs ~ 550 h + 27 h^2 wave-cycles per start anchor, an inner gap
~ 870 g + 20 000), it is not known before the read's structure is, and input files are often
ordered (by length, by locus): many small interleaved chunks give every rank the same mixture,
which is what `schedule(dynamic)` gives the reference's threads.  Rank 0 merges by chunk index.
"""
import numpy as np


def shard_bounds(lengths, world):
    """Contiguous blocks of reads with (nearly) equal total bases: bounds[r]..bounds[r+1].
    (The static split of rounds 1-2; kept for callers that need contiguous shards.)"""
    lengths = np.asarray(lengths, dtype=np.int64)
    n = len(lengths)
    csum = np.concatenate([[0], np.cumsum(lengths)])
    total = int(csum[-1])
    bounds = [0]
    for r in range(1, world):
        target = total * r / world
        i = int(np.searchsorted(csum, target, side="left"))
        i = min(max(i, bounds[-1]), n)
        bounds.append(i)
    bounds.append(n)
    return bounds


def chunk_size_for(n_reads, world):
    """Reads per chunk: at least 64 chunks per rank where the input allows, chunks of 64 .. 4096 reads."""
    return int(min(4096, max(64, n_reads // (64 * max(world, 1)))))


def deal_chunks(n_reads, world, chunk=None):
    """[(first, count)] of every chunk in input order; chunk c belongs to rank c % world."""
    if chunk is None:
        chunk = chunk_size_for(n_reads, world)
    return [(lo, min(chunk, n_reads - lo)) for lo in range(0, n_reads, chunk)]


def rank_chunks(chunks, world, rank):
    """The chunks of `rank`, in input order."""
    return [c for i, c in enumerate(chunks) if i % world == rank]


class RecordGatherer:
    """Gathers one uint8 tensor per rank (variable size) on `dst`, every step of a run.

    Per step: ONE small all_gather of the payload sizes into a preallocated tensor, ONE host read of it on dst (the
    other ranks never wait for it), and point-to-point transfers of exactly the payload bytes into receive buffers dst
    keeps from step to step (they only ever grow): no padding to the largest payload, no zero fill, no per-rank
    host round trips.  On RCCL the batched send / recv pairs are one group: every peer's payload travels over its
    own xGMI link to rank 0 at the same time.
    """

    def __init__(self, dist, rank, world, device, dst=0):
        import torch
        self.torch, self.dist, self.rank, self.world, self.dst, self.device = torch, dist, rank, world, dst, device
        self.sizes = torch.zeros(world, dtype=torch.int64, device=device)
        self.mine = torch.zeros(1, dtype=torch.int64, device=device)
        self.recv = [None] * world      # dst only: one buffer per peer
        self.host_syncs = 0             # (diagnostic: host reads of device values per run)

    def gather(self, payload):
        """payload: uint8 tensor on self.device.  Returns the list of payloads in rank order on dst (views into the
        kept receive buffers: valid until the next call), None elsewhere."""
        torch, dist = self.torch, self.dist
        self.mine[0] = payload.numel()
        dist.all_gather_into_tensor(self.sizes, self.mine)
        if self.rank != self.dst:
            if payload.numel():
                for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, payload, self.dst)]):
                    w.wait()
            return None
        sizes = self.sizes.tolist()     # the step's one host read on dst
        self.host_syncs += 1
        ops, out = [], []
        for r in range(self.world):
            if r == self.dst:
                out.append(payload)
                continue
            if self.recv[r] is None or self.recv[r].numel() < sizes[r]:
                self.recv[r] = torch.empty(int(sizes[r] * 1.05) + 4096, dtype=torch.uint8, device=self.device)
            view = self.recv[r][: sizes[r]]
            out.append(view)
            if sizes[r]:
                ops.append(dist.P2POp(dist.irecv, view, r))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return out


def gather_records(payload, dist, rank, world, dst=0):
    """One-shot form of RecordGatherer.gather (tests, single calls)."""
    return RecordGatherer(dist, rank, world, payload.device, dst).gather(payload)


def pack_records(seq_bytes, offsets, status):
    """Serialise one rank's corrected records: [n u64][offsets (n+1) u64][status n i32][bytes]."""
    n = len(status)
    head = np.array([n], dtype=np.uint64).tobytes()
    return np.frombuffer(head + np.asarray(offsets, dtype=np.uint64).tobytes()
                         + np.asarray(status, dtype=np.int32).tobytes() + bytes(seq_bytes), dtype=np.uint8)


def header_bytes(n):
    return 8 + 8 * (n + 1) + 4 * n


def pack_records_device(torch, records, offsets, status, out=None):
    """The same payload assembled on the device: `records` is a uint8 device tensor (the dense corrected records of
    talc_batch_copy_corrected_device), offsets / status the host arrays that call returned.  One small H2D copy for the
    header; the records never leave the GPU.  `out`: a kept uint8 device tensor to assemble into (grown when too small)."""
    n = len(status)
    head = np.concatenate([np.array([n], dtype=np.uint64).view(np.uint8),
                           np.ascontiguousarray(offsets, dtype=np.uint64).view(np.uint8),
                           np.ascontiguousarray(status, dtype=np.int32).view(np.uint8)])
    total = len(head) + records.numel()
    if out is None or out.numel() < total:
        out = torch.empty(int(total * 1.05) + 4096, dtype=torch.uint8, device=records.device)
    out[: len(head)].copy_(torch.from_numpy(head))     # (a pageable temporary: the copy must not outlive it)
    out[len(head): total].copy_(records)
    return out[:total]


def unpack_records(buf):
    b = bytes(buf)
    n = int(np.frombuffer(b[:8], dtype=np.uint64)[0])
    offs = np.frombuffer(b[8:8 + 8 * (n + 1)], dtype=np.uint64)
    st = np.frombuffer(b[8 + 8 * (n + 1):8 + 8 * (n + 1) + 4 * n], dtype=np.int32)
    seq = b[8 + 8 * (n + 1) + 4 * n:]
    return seq, offs, st


def merge_in_order(per_rank, chunks=None):
    """The unpacked per-rank records as one record set in input order.

    chunks = None: rank order is input order (contiguous shards).  chunks = deal_chunks(...): rank r holds the records
    of chunks r, r + N, ... back to back; the merge walks the chunks in input order."""
    world = len(per_rank)
    parts = [unpack_records(buf) for buf in per_rank]
    if chunks is None:
        chunks_of = None
        order = [(r, 0, len(parts[r][2])) for r in range(world)]
    else:
        cursor = [0] * world
        order = []
        for i, (_, cnt) in enumerate(chunks):
            r = i % world
            order.append((r, cursor[r], cnt))
            cursor[r] += cnt
        for r in range(world):
            if cursor[r] != len(parts[r][2]):
                raise ValueError("rank %d sent %d records, its chunks hold %d" % (r, len(parts[r][2]), cursor[r]))
    seqs, sts, offs = [], [], [np.zeros(1, dtype=np.uint64)]
    base = 0
    for r, lo, cnt in order:
        seq, o, st = parts[r]
        o = np.asarray(o, dtype=np.uint64)
        b0, b1 = int(o[lo]), int(o[lo + cnt])
        seqs.append(seq[b0:b1])
        sts.append(st[lo:lo + cnt])
        offs.append(o[lo + 1: lo + cnt + 1] - np.uint64(b0) + np.uint64(base))
        base += b1 - b0
    return (b"".join(seqs), np.concatenate(offs).astype(np.uint64),
            (np.concatenate(sts) if sts else np.zeros(0, np.int32)))
