#!/usr/bin/env python3
"""bench.py — corrected long-read bases/sec of the TALC hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`.  `--gpus N` IS the parallelism, as `-t N` is the
reference's (main.cpp:242,247): run plainly with N > 1 and no WORLD_SIZE in the environment, this process stays off
the GPU, starts the N ranks itself as child processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, 127.0.0.1),
relays rank 0's one JSON line and exits with the children's worst code.  Launched by torch.distributed.run (one rank
per GPU, backend nccl = RCCL) it is one of those ranks; a WORLD_SIZE that differs from `--gpus` is an error (exit 2).
One "step" is one pass of the whole hot path (coverage probe -> structure -> path search -> reassembly) over the
workload's reads, already resident in HBM, plus — for N>1 — the RCCL gather of the corrected records to rank 0.

Workloads (BASELINE.json `configs`, synthetic data):
  N = 1  -> configs[1] ("config2"): 100 k ONT-like reads (~2 kb, 12 % error), 50 M-entry k=21 dump
  N > 1  -> configs[2] ("config3"): 1 M reads, 200 M-entry k=21 dump — the SAME 1 M reads whatever N is (strong
            scaling): the input is cut into small chunks and chunk c goes to half-shard c mod 2N (rank = that / 2:
            talc_amd.sharding.deal_chunks, replaces the OpenMP schedule(dynamic) loop of main.cpp:247-308); a rank
            runs its two half-shards on two contexts (own stream each), so the gather of the first half's records
            rides under the second half's search; the table is built once on rank 0 and its device image broadcast
            to the other ranks over RCCL / xGMI (replicated per GPU, no data-path collective)
  --config 2|3|4|5 forces one of them at any N (4 = config3 + junction colours; 5 = k=31, 500 M entries, 100 k reads
  of 500 b - 20 kb); --reads / --kmers / --k override single figures (the label then says "custom").
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
COV_BYTES_PER_KMER = 25        # 16 B slot + 8 B result + 1 B base
STEP_BYTES = 64                # 64 B per Trail-step (4 slots x 16 B)

CONFIGS = {
    2: dict(reads=100_000, kmers=50_000_000, k=21, junctions=False, mixed=False),
    3: dict(reads=1_000_000, kmers=200_000_000, k=21, junctions=False, mixed=False),
    4: dict(reads=1_000_000, kmers=200_000_000, k=21, junctions=True, mixed=False),
    5: dict(reads=100_000, kmers=500_000_000, k=31, junctions=False, mixed=True),
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None,
                    help="GPUs of the node to shard the reads over (default: WORLD_SIZE, else 1); without WORLD_SIZE in the "
                         "environment and N > 1 this process starts the N ranks itself")
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="auto", help="auto (N=1: 2, N>1: 3) or one of 2, 3, 4, 5")
    ap.add_argument("--reads", type=int, default=None, help="total reads of the workload (override)")
    ap.add_argument("--kmers", type=int, default=None, help="distinct k-mers in the synthetic dump (override)")
    ap.add_argument("--k", type=int, default=None)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cpu-sample", type=int, default=0,
                    help="reads of the same workload timed on the CPU oracle (0 = sized for ~15 s from a pilot run)")
    ap.add_argument("--cpu-backend", choices=["flat", "map"], default="flat",
                    help="oracle table: flat hash (quick to build) or the reference's std::map (SURVEY §8d)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-h2h", action="store_true", help="skip the host-to-host pass")
    ap.add_argument("--no-paralog", action="store_true", help="skip the branching-graph (paralog families) side measurement")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end run of the talc CLI (text dump + FASTA -> <o>.fa)")
    ap.add_argument("--no-map", action="store_true", help="skip the std::map leg of the CPU baseline (its table takes ~1 min to build)")
    ap.add_argument("--chunk-reads", type=int, default=0, help="N > 1: reads per dealt chunk (0 = talc_amd.sharding.chunk_size_for)")
    ap.add_argument("--halves", type=int, default=2, choices=[1, 2],
                    help="N > 1: half-shards (contexts) per rank; 2 = the first half's gather runs under the second half's search")
    return ap.parse_args(argv)


def resolve_workload(a, world):
    """The workload's figures and its label, from the arguments alone (tests/test_abi_and_host.py pins this)."""
    cfg = (2 if world == 1 else 3) if a.config == "auto" else int(a.config)
    if cfg not in CONFIGS:
        raise SystemExit("--config must be auto, 2, 3, 4 or 5")
    w = dict(CONFIGS[cfg])
    custom = []
    for key in ("reads", "kmers", "k"):
        v = getattr(a, key)
        if v is not None and v != w[key]:
            w[key] = v
            custom.append(key)
    w["config"] = cfg
    name = "config%d" % cfg if not custom else "custom (config%d with %s changed)" % (cfg, ", ".join(custom))
    shape = "500 b - 20 kb log-uniform" if w["mixed"] else "~2 kb"
    w["label"] = ("%s: %d synthetic ONT-like reads in all (%s, 12%% error) sharded over %d GPU(s), %d-base synthetic "
                  "transcriptome's k=%d k-mer dump%s, table replicated per GPU"
                  % (name, w["reads"], shape, world, w["kmers"], w["k"], " + junction dump (--junctions)" if w["junctions"] else ""))
    return w


def rehearsal():
    """TALC_BENCH_REHEARSAL=1: the N > 1 code path on a box with ONE GPU — every rank on device 0, backend gloo, collectives
    staged through host memory.  For checking the path's logic only; its numbers mean nothing and the line says so."""
    return os.environ.get("TALC_BENCH_REHEARSAL", "") == "1"


def build_table(T, synth, w, params, dev, rank, world, dist, torch, log):
    """Rank 0 builds the table on its GPU (insertion, colouring, de-colouring: all kernels); for N > 1 its device
    image goes to every other rank by RCCL broadcasts (1 GiB pieces of the two bucket tables) and is imported there."""
    n_dump = 0
    table = None
    if rank == 0:
        keys, counts = synth.dump_arrays()
        n_dump = len(keys)
        table = T.Table.from_arrays(keys, counts, params, device=dev)
        if w["junctions"]:
            jk, jc = synth.junction_arrays()
            table.colour(jk, jc)
        table.decolour_repeats()
    else:
        keys = counts = None
    if world > 1:
        meta = torch.zeros(3, dtype=torch.int64, device="cpu" if rehearsal() else "cuda")
        if rank == 0:
            meta[0], meta[1], meta[2] = table.capacity, len(table), n_dump
        dist.broadcast(meta, src=0)
        cap, nk, n_dump = (int(x) for x in meta.tolist())
        t0 = time.time()
        br = torch.empty(cap * 32, dtype=torch.uint8, device="cuda")
        bl = torch.empty(cap * 32, dtype=torch.uint8, device="cuda")
        if rank == 0:
            table.export_device(dev, br.data_ptr(), bl.data_ptr())
        for buf in (br, bl):      # pieces of 1 GiB: element counts stay far below any 32-bit limit on the way
            for lo in range(0, buf.numel(), 1 << 30):
                piece = buf[lo:lo + (1 << 30)]
                if rehearsal():
                    host = piece.cpu()
                    dist.broadcast(host, src=0)
                    piece.copy_(host)
                else:
                    dist.broadcast(piece, src=0)
        torch.cuda.synchronize()
        if rank != 0:
            table = T.Table.import_device(params, cap, nk, br.data_ptr(), bl.data_ptr(), dev)
        del br, bl
        torch.cuda.empty_cache()
        log("table image (2 x %.1f GB) broadcast over RCCL in %.1f s" % (cap * 32 / 1e9, time.time() - t0))
    table.upload(dev)
    return table, keys, counts, n_dump


def claim_stdout():
    """The contract is ONE JSON line on rank 0's stdout.  Backend banners ("[Gloo] Rank ...", RCCL / HIP notices) are
    written to file descriptor 1 by native code, so the descriptor itself is pointed at stderr for the whole run and the
    line goes to a private duplicate of the original stdout at the end."""
    sys.stdout.flush()
    keep = os.dup(1)
    os.dup2(2, 1)
    return keep


def emit_line(fd, text):
    os.write(fd, (text + "\n").encode())


def launch_ranks(a, argv, child=None):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes.  This process never
    imports torch or touches the GPU (a process that has must not be replaced or forked from, and the children must see
    an untouched runtime; `child`: another command line for the ranks, for the tests); rank 0's stdout is a pipe whose one JSON line is relayed, every other rank's stdout goes to
    stderr.  If a rank fails the others are ended (their exact PIDs) so that nobody waits in a collective for ever.
    Exit code = the worst of the children's."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TALC_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen((child or [sys.executable, os.path.abspath(__file__)]) + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr))
    lines = []

    def drain():
        for raw in procs[0].stdout:
            lines.append(raw)

    th = threading.Thread(target=drain, daemon=True)
    th.start()
    worst, failed_at = 0, None
    ended = set()                     # ranks this launcher ended itself: their signal is not the run's result
    live = set(range(a.gpus))
    while live:
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is None:
                continue
            live.discard(r)
            if rc != 0 and r not in ended:
                worst = max(worst, rc if rc > 0 else 128 - rc)
                if failed_at is None:
                    failed_at = time.time()
                    print("[bench launcher] rank %d exited with %d; ending the other ranks" % (r, rc), file=sys.stderr, flush=True)
        if failed_at is not None and live and time.time() - failed_at > 15.0:
            for r in live - ended:
                procs[r].kill()       # exact PIDs of this launcher's own children
                ended.add(r)
        time.sleep(0.05)
    th.join(timeout=10.0)
    for raw in lines:
        os.write(1, raw)
    return worst


class HalfShard:
    """One context + one resident batch: a rank's reads (N = 1) or one of its two halves (N > 1: virtual rank
    2 * rank + h owns the chunks c with c mod 2N == 2 * rank + h)."""

    def __init__(self, T, SH, torch, table, params, dev, synth, my_chunks, whole=None):
        self.T, self.SH, self.torch, self.dev = T, SH, torch, dev
        self.ctx = T.Context(table, params, dev)
        if whole is not None:
            bases, offs = synth.reads(0, whole)
        else:
            parts = [synth.reads(lo, cnt) for lo, cnt in my_chunks]
            bases = np.concatenate([p[0] for p in parts]) if parts else np.zeros(0, np.uint8)
            offs = np.zeros(sum(len(p[1]) - 1 for p in parts) + 1, dtype=np.uint64)
            pos, k = 0, 0
            for pb, po in parts:
                offs[k + 1: k + len(po)] = po[1:] + np.uint64(pos)
                pos += int(po[-1]); k += len(po) - 1
            del parts
        self.bases, self.offs = bases, offs
        self.n_reads = len(offs) - 1
        self.batch = self.ctx.batch(bases, offs)
        self.n_bases = self.batch.n_bases
        self.records = None     # kept device buffers (they only ever grow)
        self.payload_buf = None
        self.payload = None

    def correct(self):
        self.batch.correct()

    def correct_and_pack(self):
        """The half's hot path, then [n][offsets][status][records] assembled on the device for the gather."""
        torch = self.torch
        torch.cuda.set_device(self.dev)          # (the current device is per host thread)
        self.batch.correct()
        nbytes = self.batch.corrected_bytes
        if self.records is None or self.records.numel() < max(nbytes, 1):
            self.records = torch.empty(int(max(nbytes, 1) * 1.05) + 4096, dtype=torch.uint8, device="cuda")
        oo_, st_ = self.batch.copy_corrected_to_device(self.records.data_ptr(), nbytes)
        total = self.SH.header_bytes(len(st_)) + nbytes
        if self.payload_buf is None or self.payload_buf.numel() < total:
            self.payload_buf = torch.empty(int(total * 1.05) + 4096, dtype=torch.uint8, device="cuda")
        self.payload = self.SH.pack_records_device(torch, self.records[:nbytes], oo_, st_, out=self.payload_buf)
        torch.cuda.current_stream().synchronize()   # the payload is complete before another thread's collective reads it


def main():
    argv = sys.argv[1:]
    a = parse(argv)
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and a.gpus is not None and a.gpus > 1:
        sys.exit(launch_ranks(a, argv))
    rank = int(os.environ.get("RANK", "0"))
    world = int(env_world or "1")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus is not None and a.gpus != world:
        print("bench.py: --gpus %d but WORLD_SIZE=%d: the launcher's rank count and --gpus must agree" % (a.gpus, world),
              file=sys.stderr, flush=True)
        sys.exit(2)
    a.gpus = world
    out_fd = claim_stdout()
    e2e = None
    if world == 1 and not a.no_e2e and not under_profiler():
        # before anything in this process touches the GPU: the CLI is a child process of a GPU-free parent (under
        # rocprofv3 the preloaded tool has initialised the GPU already and would trace the child into the same files)
        e2e = end_to_end_cli(a, resolve_workload(a, 1))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        if rehearsal():
            local_rank = 0
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = local_rank if world > 1 else 0
    torch.cuda.set_device(dev)

    from talc_amd import build as B
    from talc_amd import lib as T
    from talc_amd import sharding as SH
    from talc_amd.synth import Synth

    log = (lambda m: print("[bench r%d] %s" % (rank, m), file=sys.stderr, flush=True))
    w = resolve_workload(a, world)
    t_setup = time.time()
    synth = Synth(target_kmers=w["kmers"], k=w["k"], seed=a.seed, mixed_lengths=int(w["mixed"]))
    params = T.default_params(k=w["k"], use_junctions=int(w["junctions"]))
    table, keys, counts, n_dump = build_table(T, synth, w, params, dev, rank, world, dist, torch, log)
    n_table = len(table)
    # this rank's share of the reads: the input cut into small chunks, chunk c to virtual rank c mod V (V = N half-shards
    # x N ranks; talc_amd/sharding.py: every rank gets the same mixture of reads whatever the order of the input, as
    # schedule(dynamic) gives the reference's threads); a half-shard's chunks back to back are its one resident batch
    halves = 1 if world == 1 else a.halves
    vworld = world * halves
    chunks = SH.deal_chunks(w["reads"], vworld, a.chunk_reads or None)
    if world == 1:
        shards = [HalfShard(T, SH, torch, table, params, dev, synth, None, whole=w["reads"])]
    else:
        shards = [HalfShard(T, SH, torch, table, params, dev, synth, SH.rank_chunks(chunks, vworld, rank * halves + h))
                  for h in range(halves)]
    ctx, batch = shards[0].ctx, shards[0].batch
    bases, offs = shards[0].bases, shards[0].offs
    n_mine = sum(s.n_reads for s in shards)
    n_bases = sum(s.n_bases for s in shards)
    setup_s = time.time() - t_setup
    log("setup %.1fs: table %d k-mers (%.2f GB on device), %d of %d reads in %d half-shard(s) of %d chunks in all / %d bases resident" %
        (setup_s, n_table, table.device_bytes / 1e9, n_mine, w["reads"], len(shards), len(chunks), n_bases))
    # the std::map oracle table (the reference's own backend) takes about a minute to build for 54 M k-mers:
    # one host thread builds it while the GPU legs run (the box has far more cores than those legs use)
    map_job = None
    if world == 1 and rank == 0 and not a.no_cpu and not a.no_map and a.cpu_backend != "map":
        map_job = MapTableJob(w, synth, keys, counts)

    comm_dev = "cpu" if (world > 1 and rehearsal()) else "cuda"
    gatherers = [SH.RecordGatherer(dist, rank, world, comm_dev, dst=0) for _ in shards] if world > 1 else []

    def one_step():
        """N = 1: the hot path over the resident batch.  N > 1: the rank's two half-shards run on their own contexts
        and streams from two host threads (the first one's launch a moment ahead: its search takes the GPU, the second
        one's waves fill the slots it frees); this thread gathers half 0's records on rank 0 — the 'trivial RCCL gather':
        one small all_gather of sizes, one host read of it on rank 0, point-to-point transfers of exactly the payload
        bytes into kept buffers — while half 1 is still searching, then half 1's."""
        if world == 1:
            shards[0].correct()
            return None
        done = [threading.Event() for _ in shards]
        errs = [None] * len(shards)

        def work(i):
            try:
                shards[i].correct_and_pack()
            except BaseException as e:   # noqa: BLE001 — re-raised on the main thread below
                errs[i] = e
            finally:
                done[i].set()

        th = []
        for i in range(len(shards)):
            t = threading.Thread(target=work, args=(i,))
            t.start()
            th.append(t)
            if i + 1 < len(shards):
                time.sleep(0.0005)
        out = []
        for i, s in enumerate(shards):
            done[i].wait()
            # (a failed half still takes part in the collectives below with an empty payload: no rank may hang)
            payload = s.payload if errs[i] is None else torch.zeros(0, dtype=torch.uint8, device="cuda")
            out.append(gatherers[i].gather(payload.cpu() if rehearsal() else payload))
        for t in th:
            t.join()
        for e in errs:
            if e is not None:
                raise e
        return out

    tm_keys = ("encode_ms", "coverage_ms", "structure_ms", "search_ms", "emit_ms", "retry_ms")

    def step_timing():
        """Kernel times of the step just run, summed over the rank's half-shards (N > 1: the halves' kernels share the
        GPU, so a half's event times include waiting for the other's waves: not per-kernel figures, see `roofline` at N = 1)."""
        ts = [s.ctx.timing() for s in shards]
        return ts, {k: sum(getattr(t, k) for t in ts) for k in tm_keys}

    gathered = None
    for i in range(a.warmup):
        tw = time.time()
        gathered = one_step()
        log("warmup %d: %.2fs  %s" % (i, time.time() - tw, {k: round(v, 2) for k, v in step_timing()[1].items()}))
    tm = {k: 0.0 for k in tm_keys}
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last_all = None
    for _ in range(a.steps):
        gathered = one_step()
        last_all, t_step = step_timing()
        for k in tm:
            tm[k] += t_step[k]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal() else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tb = torch.tensor([float(n_bases)], dtype=torch.float64, device="cpu" if rehearsal() else "cuda")
        dist.all_reduce(tb, op=dist.ReduceOp.SUM)
        total_bases = float(tb.item())
    else:
        total_bases = float(n_bases)
    for k in tm:
        tm[k] /= max(a.steps, 1)
    if last_all is None:
        last_all = step_timing()[0]
    n_kmers = sum(t.n_kmers for t in last_all)
    n_trail_steps = sum(t.n_trail_steps for t in last_all)
    n_dp_cells = sum(t.n_dp_cells for t in last_all)
    n_retried = sum(t.n_retried for t in last_all)

    out, oo, st = batch.fetch_corrected()
    status_hist = np.bincount(st, minlength=5)
    for s in shards[1:]:
        status_hist = status_hist + np.bincount(s.batch.fetch_corrected()[2], minlength=5)
    status_hist = status_hist.tolist()
    merged_reads = None
    if world > 1 and rank == 0 and gathered is not None:
        # untimed: the gathered payloads merge into one record set in input order (virtual rank 2 r + h = half h of rank r)
        per_v = [None] * vworld
        for h, per_rank in enumerate(gathered):
            for r, g in enumerate(per_rank):
                per_v[r * halves + h] = g.cpu().numpy()
        seq_all, off_all, st_all = SH.merge_in_order(per_v, chunks)
        merged_reads = len(st_all)
        # rank 0's first half-shard holds the input's first chunk: its records must open the merged set
        n0 = chunks[0][1] if chunks else 0
        if (merged_reads != w["reads"] or int(off_all[-1]) != len(seq_all)
                or seq_all[: int(oo[n0])] != out[: int(oo[n0])].tobytes() or not np.array_equal(st_all[:n0], st[:n0])):
            raise SystemExit("gathered records do not merge into the workload's %d reads" % w["reads"])

    result = None
    if rank == 0:
        value = total_bases * a.steps / elapsed
        cov_s = tm["coverage_ms"] / 1e3
        cov_gbs = (n_kmers * COV_BYTES_PER_KMER / cov_s / 1e9) if cov_s > 0 else 0.0
        search_s = (tm["search_ms"] + tm["retry_ms"]) / 1e3
        search_gbs = (n_trail_steps * STEP_BYTES / search_s / 1e9) if search_s > 0 else 0.0
        lib_hash = B.source_hash()
        pmc = committed_pmc(lib_hash, w, n_mine)
        sq = committed_sq(lib_hash, w, n_mine)
        result = {
            "metric": "corrected long-read bases/sec (whole node); k-mer-probe HBM GB/s",
            "value": value,
            "unit": "bases/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": 1e3 * elapsed / max(a.steps, 1),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic" if not (world > 1 and rehearsal()) else "synthetic; REHEARSAL of the N > 1 path on one GPU over gloo: not a measurement",
            "config": {
                "workload": w["label"],
                "value_is": "reads resident in HBM -> corrected records resident in HBM (the task's bench contract); SURVEY §8d's "
                            "phase (first read submitted from host memory -> last record back in host memory) is `host_to_host`, "
                            "the whole program (text dump + FASTA -> <o>.fa) is `end_to_end`; their figures are repeated in this object",
                "read_deal": ("all reads on the one GPU" if world == 1 else
                              "%d chunks of <= %d reads, chunk c on half-shard c mod %d (rank = half-shard / %d, %d contexts per rank); "
                              "rank 0 merges by chunk index" % (len(chunks), chunks[0][1], vworld, halves, halves)),
                "launched_by": "bench.py --gpus N (own child ranks)" if os.environ.get("TALC_BENCH_SELF_LAUNCHED") == "1"
                               else ("torch.distributed.run" if world > 1 else "single process"),
                "baseline_config": w["config"], "reads_total": w["reads"], "reads_rank0": n_mine, "k": w["k"],
                "dump_entries": n_dump, "table_kmers": n_table, "junctions": bool(w["junctions"]),
                "table_device_bytes": table.device_bytes, "bases_total": total_bases, "bases_rank0": n_bases,
                "read_status_hist_rank0[corrected,short,no_solid,no_structure,error]": status_hist,
                "setup_s": setup_s, "lib_source_hash": lib_hash, "gathered_reads_on_rank0": merged_reads,
            },
            # the kernel the metric names: the k-mer coverage probe (Read::reCoverage); HBM-bound.  `achieved` =
            # algorithmic bytes (25 B per k-mer probed) / this run's launch time (HIP events on the library's stream).
            # `traffic` = FETCH_SIZE + WRITE_SIZE of one launch from the separate rocprofv3 --pmc passes committed under
            # profiles/ (null unless they were taken from this very build and workload): requests that leave the XCD L2s,
            # Infinity-Cache hits included (MI355X_MICROARCH.md §HBM) — an upper bound of the HBM traffic, not HBM bytes.
            "roofline": {
                "kernel": "k_coverage", "bound": "hbm", "achieved": cov_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": cov_gbs / HBM_PEAK_GBS, "traffic": pmc.get("k_coverage"),
                "algorithmic_bytes_per_launch": n_kmers * COV_BYTES_PER_KMER, "launch_ms": tm["coverage_ms"],
                "traffic_note": "L2-miss (fabric) bytes per launch incl. Infinity-Cache hits, separate PMC pass; null = no pass for this build",
            },
            # the kernel that dominates the step time: the path search (integer DP + dependent probes).  It is LATENCY-bound
            # (the SQ counters of the committed pass say where the waves' time goes), which is why its share of the HBM peak is small.
            "roofline_search": dict({
                "kernel": "k_search", "bound": "hbm", "achieved": search_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": search_gbs / HBM_PEAK_GBS, "traffic": pmc.get("k_search"),
                "algorithmic_bytes_per_launch": n_trail_steps * STEP_BYTES, "launch_ms": 1e3 * search_s,
                "trail_steps": n_trail_steps, "dp_cells": n_dp_cells,
                "dp_gcups": (n_dp_cells / search_s / 1e9) if search_s > 0 else 0.0,
            }, **sq),
            "kernels_ms": tm,
            "retried_reads": n_retried,
        }
    if world == 1 and not a.no_h2h:
        h2h = host_to_host(T, table, params, dev, bases, offs, max(2, a.steps), log)
        if rank == 0:
            result["host_to_host"] = h2h
            result["config"]["host_to_host_bases_per_s"] = h2h["value"]
            result["config"]["host_to_host_ms_per_step"] = h2h["ms_per_step"]
    if world == 1 and not a.no_paralog and rank == 0:
        pw = paralog_workload(T, dev, log, with_cpu=not a.no_cpu)
        result["paralog_workload"] = pw
        result["config"]["paralog_bases_per_s"] = pw["value"]
        result["config"]["paralog_ms_per_step"] = pw["ms_per_step"]
        if "cpu_value" in pw:
            result["config"]["paralog_cpu_bases_per_s"] = pw["cpu_value"]
    if rank == 0:
        if e2e is not None:
            result["end_to_end"] = e2e
            if "wall_s" in e2e:
                result["config"]["end_to_end_wall_s"] = e2e["wall_s"]
                result["config"]["end_to_end_bases_per_s"] = e2e["value"]
                result["config"]["end_to_end_split_s"] = e2e["split_s"]
            else:
                result["config"]["end_to_end_error"] = e2e.get("error")
        if not a.no_cpu and world == 1:
            result["cpu_baseline"] = cpu_baseline(a, w, synth, keys, counts, bases, offs, out, oo, st, map_job, log)
        elif not a.no_cpu:
            result["cpu_baseline"] = None   # (N > 1: the CPU baseline is rank 0's N = 1 leg, see BENCH at N = 1)
        if gatherers:
            result["config"]["gather_host_reads_per_step_rank0"] = sum(g.host_syncs for g in gatherers) / max(a.steps + a.warmup, 1)
        emit_line(out_fd, json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def under_profiler():
    """True when a rocprofiler tool library is preloaded into this process (rocprofv3 -- python3 bench.py ...)."""
    blob = " ".join(os.environ.get(k, "") for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD",
                                                     "ROCPROF_OUTPUT_PATH", "ROCPROF_OUTPUT_FILE_NAME"))
    return "rocprof" in blob.lower() or any(k.startswith("ROCPROF") for k in os.environ)


def committed_sq(lib_hash, w, reads_rank):
    """Where `k_search`'s waves spend their time, from the SQ-counter pass committed under profiles/ (same rule as the
    traffic: only when it was taken from this very build on this workload): wait_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES,
    valu_issue_frac = VALU instructions x cycles per wave64 instruction / (SIMDs x launch cycles), waves_per_simd."""
    try:
        with open(os.path.join(ROOT, "profiles", "sq_search.json")) as f:
            sq = json.load(f)
        if sq.get("lib_source_hash") != lib_hash or sq.get("baseline_config") != w["config"] or sq.get("reads") != reads_rank:
            return {}
        return {k: sq[k] for k in ("wait_frac", "wait_inst_frac", "valu_issue_frac", "waves_per_simd", "lanes_per_valu") if k in sq}
    except (OSError, KeyError, ValueError, TypeError):
        return {}


def committed_pmc(lib_hash, w, reads_rank):
    """HBM-side traffic per launch from the PMC passes committed under profiles/ (rocprofv3 cannot collect counters
    from inside this run): only when they were taken from this very build (same source hash) on this workload."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            pmc = json.load(f)
        entry = pmc.get("configs", {}).get(str(w["config"]))
        if pmc.get("lib_source_hash") != lib_hash or not entry or entry.get("reads") != reads_rank or "custom" in w["label"]:
            return {}
        return {k: 1024.0 * (entry[k]["FETCH_SIZE_KB"] + entry[k]["WRITE_SIZE_KB"]) for k in ("k_coverage", "k_search") if k in entry}
    except (OSError, KeyError, ValueError, TypeError):
        return {}


def host_to_host(T, table, params, dev, bases, offs, passes, log):
    """SURVEY §8d's phase as the reference has it — first read submitted from host memory to last corrected record back
    in host memory (the replacement of main.cpp:247-310) — never `value`.  The workload goes through in 2-4 sub-batches on
    two contexts (two host threads, one stream each, pinned buffers from talc_pinned_alloc), so that one sub-batch's
    H2D / D2H copies run under the other's kernels: the streaming form the CLI uses (io.cpp:26-75 replaced)."""
    n = len(offs) - 1
    nsub = 4 if n >= 400_000 else (2 if n >= 4000 else 1)   # (small sub-batches leave k_search's waves a long tail)
    if os.environ.get("TALC_H2H_SUB"):
        nsub = max(1, int(os.environ["TALC_H2H_SUB"]))         # (experiments)
    cuts = [n * i // nsub for i in range(nsub + 1)]
    subs, pins = [], []
    for i in range(nsub):
        lo, hi = cuts[i], cuts[i + 1]
        nb_i = int(offs[hi]) - int(offs[lo])
        pin_in, pin_out = T.PinnedArray(nb_i), T.PinnedArray(2 * nb_i + 4096)   # the reader's / the writer's buffers
        pin_in.array[:] = bases[int(offs[lo]):int(offs[hi])]
        pins += [pin_in, pin_out]
        subs.append((pin_in.array, (offs[lo:hi + 1] - offs[lo]).copy(), pin_out.array))
    ctxs = [T.Context(table, params, dev) for _ in range(2 if nsub > 1 else 1)]
    outs = [None] * nsub

    def worker(ci):
        for i in range(ci, nsub, len(ctxs)):
            outs[i] = ctxs[ci].correct(subs[i][0], subs[i][1], out=subs[i][2])

    def one_pass():
        th = [threading.Thread(target=worker, args=(ci,)) for ci in range(len(ctxs))]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        return time.perf_counter() - t0

    one_pass()   # warm-up: allocations, pinned staging buffers
    times = [one_pass() for _ in range(passes)]
    for c in ctxs:
        c.close()
    outs = None
    for pa in pins:
        pa.close()
    dt = float(np.median(times))
    nb = float(int(offs[n]))
    log("host-to-host: %.1f ms per pass of %d reads (%d sub-batches on %d contexts)" % (1e3 * dt, n, nsub, len(ctxs)))
    return {"value": nb / dt, "unit": "bases/s", "ms_per_step": 1e3 * dt, "sub_batches": nsub, "contexts": len(ctxs),
            "what": "host buffers in -> corrected records back in host buffers (H2D + kernels + D2H, overlapped across "
                    "two contexts); table upload and FASTA parsing/writing not included"}


def paralog_workload(T, dev, log, with_cpu=True):
    """A side figure, never `value`: the same hot path on a transcriptome with paralog families (60 % of the transcripts
    are 5 %-diverged copies of others, K = 25) — forks and bubbles in the graph, so most searches carry several Trails
    and the time goes to scoreBridges / gardening / the generic expansion step (Explorer.cpp:546-612,689-865) instead
    of the single-Trail fast path the headline workload lives on.  `cpu_value`: the oracle on the box's host cores on the
    first reads of the same workload (same form as `cpu_baseline`), whose records are compared with the GPU's."""
    from talc_amd.synth import Synth
    S = Synth(target_kmers=2_000_000, k=25, seed=77, paralog_frac=0.6, paralog_div=0.05)
    keys, counts = S.dump_arrays()
    p = T.default_params(k=25)
    tab = T.Table.from_arrays(keys, counts, p, device=dev)
    tab.decolour_repeats()
    tab.upload(dev)
    ctx = T.Context(tab, p, dev)
    n = 20_000
    bases, offs = S.reads(0, n)
    b = ctx.batch(bases, offs)
    b.correct()                      # warm-up
    t0 = time.perf_counter()
    for _ in range(2):
        b.correct()
    dt = (time.perf_counter() - t0) / 2
    tm = ctx.timing()
    nb = b.n_bases
    res = {"value": nb / dt, "unit": "bases/s", "ms_per_step": 1e3 * dt, "reads": n, "bases": nb, "k": 25,
           "table_kmers": len(tab), "search_ms": tm.search_ms, "retry_ms": tm.retry_ms, "n_retried": tm.n_retried,
           "n_failed": tm.n_failed, "trail_steps": tm.n_trail_steps, "dp_cells": tm.n_dp_cells,
           "what": "20 k reads on a 2 M-base transcriptome with 60 % paralogs at 5 % divergence, K = 25 (branching graph)"}
    log("paralog workload: %.1f ms per pass of %d reads (%.3g bases/s)" % (1e3 * dt, n, nb / dt))
    if with_cpu:
        try:
            g_out, g_off, g_st = b.fetch_corrected()
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib as O
            threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            otab = O.OracleTable(O.params(k=25), O.OracleTable.FLAT)
            otab.insert_packed(keys, counts)
            otab.decolour()
            n0 = min(n, 2 * threads)
            t0 = time.perf_counter()
            otab.correct_batch(bases[: int(offs[n0])], offs[: n0 + 1].copy(), nthreads=threads)
            per_read = (time.perf_counter() - t0) / max(n0, 1)
            m = int(min(n, max(4 * threads, 6.0 / max(per_read, 1e-6))))
            sub = offs[: m + 1].copy()
            t0 = time.perf_counter()
            o_out, o_off, o_st = otab.correct_batch(bases[: int(sub[m])], sub, nthreads=threads)
            cdt = time.perf_counter() - t0
            res["cpu_value"] = float(int(sub[m]) / cdt)
            res["cpu_threads"] = threads
            res["cpu_cores"] = physical_cores() or threads
            res["cpu_sample"] = "first %d reads (%d bases, %.1f s wall), oracle with the flat table, OpenMP schedule(dynamic)" % (m, int(sub[m]), cdt)
            res["cpu_parity_with_gpu_on_sample"] = bool(int(o_off[m]) == int(g_off[m]) and np.array_equal(o_out, g_out[: int(g_off[m])])
                                                        and np.array_equal(o_st, g_st[:m]))
            otab.close()
            log("paralog workload on the CPU oracle: %.3g bases/s (%d threads)" % (res["cpu_value"], threads))
        except Exception as e:   # the side figure must never take the headline down
            res["cpu_error"] = "%s: %s" % (type(e).__name__, e)
    b.close()
    ctx.close()
    tab.close()
    return res


def physical_cores():
    """Physical cores among the CPUs this process may run on (SURVEY §8d asks for physical cores): distinct
    (physical id, core id) pairs of /proc/cpuinfo restricted to the affinity mask; None when it cannot be read."""
    try:
        allowed = os.sched_getaffinity(0)
        seen, cur = set(), {}
        with open("/proc/cpuinfo") as f:
            for line in f.read().split("\n") + [""]:
                if not line.strip():
                    if "processor" in cur and int(cur["processor"]) in allowed:
                        seen.add((cur.get("physical id", "0"), cur.get("core id", cur["processor"])))
                    cur = {}
                    continue
                k, _, v = line.partition(":")
                cur[k.strip()] = v.strip()
        return len(seen) or None
    except (OSError, ValueError):
        return None


class MapTableJob:
    """Builds the oracle's std::map<string, ...> table — the reference's own backend (Jellyfish.hpp:32-34) — on one host
    thread while the GPU legs of the bench run (ctypes releases the GIL for the call)."""

    def __init__(self, w, synth, keys, counts):
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O
        self.O, self.w, self.synth = O, w, synth
        self.q = O.params(k=w["k"], use_junctions=int(w["junctions"]))
        self.tab = O.OracleTable(self.q, O.OracleTable.MAP)
        self.build_s = None
        self.error = None
        self._t = threading.Thread(target=self._run, args=(keys, counts), daemon=True)
        self._t.start()

    def _run(self, keys, counts):
        try:
            t0 = time.time()
            order = np.argsort(keys, kind="stable")
            self.tab.insert_packed(keys[order], counts[order], sorted_hint=True)
            if self.w["junctions"]:
                jk, jc = self.synth.junction_arrays()
                self.tab.colour_packed(jk, jc)
            self.tab.decolour()
            self.build_s = time.time() - t0
        except BaseException as e:   # noqa: BLE001 — handed to table()
            self.error = e

    def table(self):
        """The finished table; raises what the build raised (a partly built table is never timed)."""
        self._t.join()
        if self.error is not None:
            raise self.error
        return self.tab


def cpu_baseline(a, w, synth, keys, counts, bases, offs, g_out, g_off, g_st, map_job=None, log=None):
    """The oracle (kind "port": the CPU restatement of the reference path) timed on this box's host cores on the first
    reads of the same workload, OpenMP schedule(dynamic) over the reads like main.cpp:247; also a parity spot-check of
    the GPU records for those reads.  Two table backends: the flat hash (`value`: the faster one, so GPU / CPU is
    conservative) and — `map_value` — the reference's own std::map<string, ...> on a smaller sample."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    q = O.params(k=w["k"], use_junctions=int(w["junctions"]))

    def timed(tab, seconds, fixed_n):
        n = min(fixed_n, len(offs) - 1) if fixed_n > 0 else 0
        if n <= 0:   # pilot on 2 reads per thread, then size the sample for `seconds` of wall time
            n0 = min(len(offs) - 1, 2 * threads)
            t0 = time.perf_counter()
            tab.correct_batch(bases[: int(offs[n0])], offs[: n0 + 1].copy(), nthreads=threads)
            per_read = (time.perf_counter() - t0) / max(n0, 1)
            n = int(min(len(offs) - 1, max(4 * threads, seconds / max(per_read, 1e-6))))
        sub_off = offs[: n + 1].copy()
        t0 = time.perf_counter()
        o_out, o_off, o_st = tab.correct_batch(bases[: int(sub_off[n])], sub_off, nthreads=threads)
        dt = time.perf_counter() - t0
        same = bool(int(o_off[n]) == int(g_off[n]) and np.array_equal(o_out, g_out[: int(g_off[n])])
                    and np.array_equal(o_st, g_st[:n]))
        return n, int(sub_off[n]), dt, same

    backend = a.cpu_backend
    tab = O.OracleTable(q, O.OracleTable.MAP if backend == "map" else O.OracleTable.FLAT)
    t0 = time.time()
    if backend == "map":
        order = np.argsort(keys, kind="stable")
        tab.insert_packed(keys[order], counts[order], sorted_hint=True)
    else:
        tab.insert_packed(keys, counts)
    if w["junctions"]:
        jk, jc = synth.junction_arrays()
        tab.colour_packed(jk, jc)
    tab.decolour()
    build_s = time.time() - t0
    n, nb, dt, same = timed(tab, 15.0, a.cpu_sample)
    res = {
        "value": float(nb / dt), "unit": "bases/s", "cores": physical_cores() or threads, "threads": threads, "kind": "port",
        "cores_note": "cores = physical cores in the affinity mask (/proc/cpuinfo); threads = OpenMP threads used = logical CPUs",
        "table_backend": "std::map<string,...> (the reference's, SURVEY §8d)" if backend == "map" else "flat open-addressed hash (faster than the reference's std::map: the ratio GPU/CPU is conservative)",
        "sample": "first %d reads of the same workload (%d bases, %.1f s wall), oracle table backend=%s built in %.0f s, "
                  "OpenMP schedule(dynamic) like main.cpp:247" % (n, nb, dt, backend, build_s),
        "parity_with_gpu_on_sample": same,
    }
    tab.close()
    if map_job is not None:
        try:
            mtab = map_job.table()
        except Exception as e:
            res["map_error"] = "%s: %s" % (type(e).__name__, e)
            return res
        n2, nb2, dt2, same2 = timed(mtab, 6.0, 0)
        res["map_value"] = float(nb2 / dt2)
        res["map_sample"] = ("first %d reads (%d bases, %.1f s wall) against the reference's std::map<string, pair<uint,uint>> "
                             "(Jellyfish.hpp:32-34), built in %.0f s on one thread beside the GPU legs" % (n2, nb2, dt2, map_job.build_s or 0.0))
        res["map_parity_with_gpu_on_sample"] = same2
        mtab.close()
        if log:
            log("cpu baseline: flat %.3g bases/s, std::map %.3g bases/s" % (res["value"], res["map_value"]))
    return res


def end_to_end_cli(a, w):
    """The whole program, as the reference times it (main.cpp:213-236: loading, building the graph; :311-313: correction):
    the `talc` command line on this workload from a text k-mer dump (`jellyfish dump -c` format) and a FASTA file to <o>.fa
    and <o>.log.  Files are written to a temporary directory first (untimed) and removed afterwards.  Runs as a child process
    before this process has touched the GPU.  Never `value`."""
    import shutil
    import subprocess
    import tempfile
    from talc_amd import build as B
    from talc_amd.synth import Synth
    exe = os.path.join(B.OUT, "talc")
    if not os.path.exists(exe):
        return {"error": "talc CLI not built"}
    tmp = tempfile.mkdtemp(prefix="talc_e2e_", dir=os.environ.get("TALC_E2E_TMP") or None)
    try:
        t0 = time.time()
        S = Synth(target_kmers=w["kmers"], k=w["k"], seed=a.seed, mixed_lengths=int(w["mixed"]))
        dump, fa, outp = os.path.join(tmp, "sr.dump"), os.path.join(tmp, "reads.fa"), os.path.join(tmp, "out")
        S.write_dump(dump)
        S.write_fasta(fa, 0, w["reads"])
        cmd = [exe, fa, "-k", str(w["k"]), "-SR", dump, "-o", outp]
        if w["junctions"]:
            jd = os.path.join(tmp, "junctions.dump")
            S.write_junctions(jd)
            cmd += ["-j", jd]
        S.close()
        gen_s = time.time() - t0
        sizes = {"dump_bytes": os.path.getsize(dump), "fasta_bytes": os.path.getsize(fa)}
        t0 = time.perf_counter()
        p = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=1500, env=dict(os.environ, TALC_TIMING=os.environ.get("TALC_E2E_TIMING", "1")))
        wall = time.perf_counter() - t0
        timing, lib_notes = None, []
        for line in p.stderr.decode(errors="replace").splitlines():
            if line.startswith("[talc-timing] "):
                timing = json.loads(line[len("[talc-timing] "):])
            elif line.startswith("[talc-lib] "):
                lib_notes.append(line[len("[talc-lib] "):])
            elif line.startswith("[talc-batch] "):
                print(line, file=sys.stderr, flush=True)
        if p.returncode != 0 or timing is None:
            return {"error": "talc exited with %d: %s" % (p.returncode, p.stderr.decode(errors="replace")[-400:])}
        out_bytes = os.path.getsize(outp + ".fa")
        res = {"value": timing["bases"] / wall, "unit": "bases/s", "wall_s": wall,
               "correct_phase_bases_per_s": timing["bases"] / timing["correct_phase_s"] if timing["correct_phase_s"] > 0 else None,
               "split_s": timing, "table_build_detail": lib_notes, "output_fa_bytes": out_bytes, "inputs_generated_in_s": gen_s,
               "what": "talc CLI, child process: read scan + text dump parse + device table build (table_parse_build_s), upload "
                       "incl. filter and walk tables (upload_s), then the pipeline FASTA parse -> H2D -> kernels -> D2H -> 70-column "
                       "FASTA write over two workers per GPU (correct_phase_s; its three busy times overlap); value = read bases / wall_s "
                       "of the whole process (start-up and table included)"}
        res.update(sizes)
        return res
    except Exception as e:   # the side figure must never take the headline down
        return {"error": "%s: %s" % (type(e).__name__, e)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
