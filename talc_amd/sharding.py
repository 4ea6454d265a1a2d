"""Multi-GPU read sharding — the replacement of the reference's per-read OpenMP loop
(`#pragma omp parallel for schedule(dynamic)`, main.cpp:247-308).

Reads are independent units (the only cross-read state in the reference is a write-only
counter), so they shard with NO data-path collective: rank r corrects a contiguous block of
the input against its own replica of the k-mer table.  The only communication is the gather of
the corrected records to rank 0, which restores the input order (main.cpp:310 writes all
records in input order).  Works with any torch.distributed backend: "nccl" (= RCCL over xGMI)
on GPUs, "gloo" on CPU for the tests.
"""
import numpy as np


def shard_bounds(lengths, world):
    """Contiguous blocks of reads with (nearly) equal total bases: bounds[r]..bounds[r+1].

    Contiguity keeps the merge trivial (rank order == input order); balancing by bases rather
    than by count evens out the per-rank work for mixed read lengths (BASELINE config 5)."""
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


def gather_records(payload, dist, rank, world, dst=0):
    """Gather one uint8 tensor per rank (variable size) on `dst`; returns the list in rank order
    on dst, None elsewhere.  One all_gather of the sizes + one padded gather of the payloads."""
    import torch
    dev = payload.device
    n = torch.tensor([payload.numel()], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    mx = max(max(sizes), 1)
    pad = torch.zeros(mx, dtype=torch.uint8, device=dev)
    pad[: payload.numel()] = payload
    out = [torch.empty(mx, dtype=torch.uint8, device=dev) for _ in range(world)] if rank == dst else None
    dist.gather(pad, out, dst=dst)
    if rank != dst:
        return None
    return [out[r][: sizes[r]] for r in range(world)]


def pack_records(seq_bytes, offsets, status):
    """Serialise one rank's corrected records: [n u64][offsets (n+1) u64][status n i32][bytes]."""
    n = len(status)
    head = np.array([n], dtype=np.uint64).tobytes()
    return np.frombuffer(head + np.asarray(offsets, dtype=np.uint64).tobytes()
                         + np.asarray(status, dtype=np.int32).tobytes() + bytes(seq_bytes), dtype=np.uint8)


def pack_records_device(torch, records, offsets, status):
    """The same payload assembled on the device: `records` is a uint8 device tensor (the dense corrected records of
    talc_batch_copy_corrected_device), offsets / status the host arrays that call returned.  One small H2D copy for the
    header; the records never leave the GPU."""
    n = len(status)
    head = np.concatenate([np.array([n], dtype=np.uint64).view(np.uint8),
                           np.ascontiguousarray(offsets, dtype=np.uint64).view(np.uint8),
                           np.ascontiguousarray(status, dtype=np.int32).view(np.uint8)])
    return torch.cat([torch.from_numpy(head).to(records.device), records])


def unpack_records(buf):
    b = bytes(buf)
    n = int(np.frombuffer(b[:8], dtype=np.uint64)[0])
    offs = np.frombuffer(b[8:8 + 8 * (n + 1)], dtype=np.uint64)
    st = np.frombuffer(b[8 + 8 * (n + 1):8 + 8 * (n + 1) + 4 * n], dtype=np.int32)
    seq = b[8 + 8 * (n + 1) + 4 * n:]
    return seq, offs, st


def merge_in_order(per_rank):
    """Concatenate the unpacked per-rank records (rank order == input order)."""
    seqs, sts, offs = [], [], [0]
    for buf in per_rank:
        seq, o, st = unpack_records(buf)
        seqs.append(seq)
        sts.append(st)
        base = offs[-1]
        offs.extend(int(base + x) for x in o[1:])
    return b"".join(seqs), np.array(offs, dtype=np.uint64), (np.concatenate(sts) if sts else np.zeros(0, np.int32))
