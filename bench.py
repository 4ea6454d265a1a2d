#!/usr/bin/env python3
"""bench.py — corrected long-read bases/sec of the TALC hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 it is launched by
torch.distributed.run with one rank per GPU (backend nccl = RCCL).  One "step" is one pass of
the whole hot path (coverage probe -> structure -> path search -> reassembly) over one batch of
synthetic long reads that is already resident in HBM, plus — for N>1 — the RCCL gather of the
corrected records to rank 0.  Reads shard across ranks (no data-path collective), the k-mer
table is replicated per GPU: weak scaling, every rank corrects --reads reads.

Workload at N=1: BASELINE.json configs[1] — 100k synthetic ONT-like reads (~2 kb, 12% error)
against a 50M-entry synthetic k=21 k-mer dump.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
COV_BYTES_PER_KMER = 25        # 16 B slot + 8 B result + 1 B base
STEP_BYTES = 64                # 64 B per Trail-step (4 slots x 16 B)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=100_000, help="reads per GPU (config 2: 100k)")
    ap.add_argument("--kmers", type=int, default=50_000_000, help="distinct k-mers in the synthetic dump")
    ap.add_argument("--k", type=int, default=21)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cpu-sample", type=int, default=0,
                    help="reads of the same workload timed on the CPU oracle (0 = sized for ~15 s from a pilot run)")
    ap.add_argument("--cpu-backend", choices=["flat", "map"], default="flat",
                    help="oracle table: flat hash (quick to build) or the reference's std::map")
    ap.add_argument("--no-cpu", action="store_true")
    return ap.parse_args()


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = local_rank if world > 1 else 0
    torch.cuda.set_device(dev)

    from talc_amd import lib as T
    from talc_amd.synth import Synth

    t_setup = time.time()
    synth = Synth(target_kmers=a.kmers, k=a.k, seed=a.seed)
    keys, counts = synth.dump_arrays()
    params = T.default_params(k=a.k)
    table = T.Table.from_arrays(keys, counts, params, device=dev)   # insertion on the GPU (untimed setup)
    table.decolour_repeats()
    n_table = len(table)
    table.upload(dev)
    ctx = T.Context(table, params, dev)
    # this rank's shard of the read set (weak scaling: --reads per GPU)
    bases, offs = synth.reads(rank * a.reads, a.reads)
    batch = ctx.batch(bases, offs)
    n_bases = batch.n_bases
    setup_s = time.time() - t_setup
    log = (lambda m: print("[bench r%d] %s" % (rank, m), file=sys.stderr, flush=True))
    log("setup %.1fs: table %d k-mers (%.2f GB on device), %d reads / %d bases resident" %
        (setup_s, n_table, table.device_bytes / 1e9, a.reads, n_bases))

    from talc_amd import sharding as SH

    def gather_records():
        """The 'trivial RCCL gather': corrected records of every rank -> rank 0 (device tensors,
        torch.distributed over RCCL/xGMI; rank order == input order)."""
        nbytes = batch.corrected_bytes
        buf = torch.empty(max(nbytes, 1), dtype=torch.uint8, device="cuda")
        batch.copy_corrected_to_device(buf.data_ptr(), nbytes)
        if world == 1:
            return buf
        return SH.gather_records(buf[:nbytes], dist, rank, world, dst=0)

    def one_step():
        batch.correct()
        return gather_records()

    for i in range(a.warmup):
        tw = time.time()
        one_step()
        log("warmup %d: %.2fs  %s" % (i, time.time() - tw, {k: round(v, 2) for k, v in ctx.timing().as_dict().items()}))
    tm = {k: 0.0 for k in ("encode_ms", "coverage_ms", "structure_ms", "search_ms", "emit_ms", "retry_ms")}
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = None
    for _ in range(a.steps):
        one_step()
        last = ctx.timing()
        for k in tm:
            tm[k] += getattr(last, k)
        log("step done at %.2fs" % (time.perf_counter() - t0))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tb = torch.tensor([float(n_bases)], dtype=torch.float64, device="cuda")
        dist.all_reduce(tb, op=dist.ReduceOp.SUM)
        total_bases = float(tb.item())
    else:
        total_bases = float(n_bases)
    for k in tm:
        tm[k] /= max(a.steps, 1)

    out, oo, st = batch.fetch_corrected()
    status_hist = np.bincount(st, minlength=5).tolist()

    result = None
    if rank == 0:
        value = total_bases * a.steps / elapsed
        cov_s = tm["coverage_ms"] / 1e3
        cov_gbs = (last.n_kmers * COV_BYTES_PER_KMER / cov_s / 1e9) if cov_s > 0 else 0.0
        search_s = (tm["search_ms"] + tm["retry_ms"]) / 1e3
        search_gbs = (last.n_trail_steps * STEP_BYTES / search_s / 1e9) if search_s > 0 else 0.0
        # HBM traffic per launch from the committed PMC pass of this same command (rocprofv3 cannot collect counters
        # from inside the run): only reported when the workload is the one that pass measured
        traffic_cov = traffic_search = None
        try:
            with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01", "final_pmc_traffic.json")) as f:
                pmc = json.load(f)
            if pmc.get("reads_per_gpu") == a.reads:
                traffic_cov = 1024.0 * (pmc["k_coverage"]["FETCH_SIZE_KB"] + pmc["k_coverage"]["WRITE_SIZE_KB"])
                traffic_search = 1024.0 * (pmc["k_search"]["FETCH_SIZE_KB"] + pmc["k_search"]["WRITE_SIZE_KB"])
        except (OSError, KeyError, ValueError):
            pass
        result = {
            "metric": "corrected long-read bases/sec (whole node); k-mer-probe HBM GB/s",
            "value": value,
            "unit": "bases/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": 1e3 * elapsed / max(a.steps, 1),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": "config2: %d synthetic ONT-like reads/GPU (~2 kb, 12%% error), %d-entry synthetic k=%d k-mer dump "
                            "(%d k-mers kept), table replicated per GPU" % (a.reads, len(keys), a.k, n_table),
                "reads_per_gpu": a.reads, "k": a.k, "table_kmers": n_table,
                "table_device_bytes": table.device_bytes, "bases_per_gpu": n_bases,
                "read_status_hist[corrected,short,no_solid,no_structure,error]": status_hist,
                "setup_s": setup_s,
            },
            # the kernel the metric names: the k-mer coverage probe (Read::reCoverage); HBM-bound
            "roofline": {
                "kernel": "k_coverage", "bound": "hbm", "achieved": cov_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": cov_gbs / HBM_PEAK_GBS, "traffic": traffic_cov,
                "algorithmic_bytes_per_launch": last.n_kmers * COV_BYTES_PER_KMER, "launch_ms": tm["coverage_ms"],
                # measured bytes (PMC pass) over this run's launch time: the figure BASELINE's ">= 40 % per rocprof" refers to
                "measured_gbs": (traffic_cov / (tm["coverage_ms"] * 1e-3) / 1e9) if (traffic_cov and tm["coverage_ms"] > 0) else None,
                "measured_frac": (traffic_cov / (tm["coverage_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if (traffic_cov and tm["coverage_ms"] > 0) else None,
            },
            # the kernel that dominates the step time: the path search (integer DP + dependent probes)
            "roofline_search": {
                "kernel": "k_search", "bound": "hbm", "achieved": search_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": search_gbs / HBM_PEAK_GBS, "traffic": traffic_search,
                "algorithmic_bytes_per_launch": last.n_trail_steps * STEP_BYTES, "launch_ms": 1e3 * search_s,
                "trail_steps": last.n_trail_steps, "dp_cells": last.n_dp_cells,
                "dp_gcups": (last.n_dp_cells / search_s / 1e9) if search_s > 0 else 0.0,
            },
            "kernels_ms": tm,
            "retried_reads": last.n_retried,
        }
        if not a.no_cpu and world == 1:
            result["cpu_baseline"] = cpu_baseline(a, synth, keys, counts, bases, offs, out, oo, st)
        elif not a.no_cpu:
            result["cpu_baseline"] = None
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(a, synth, keys, counts, bases, offs, g_out, g_off, g_st):
    """The oracle (kind "port": the CPU restatement of the reference path) timed on this box's
    host cores on the first --cpu-sample reads of the same workload; also used as a parity
    spot-check of the GPU records for those reads."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    n = min(a.cpu_sample, len(offs) - 1) if a.cpu_sample > 0 else 0
    q = O.params(k=a.k)
    tab = O.OracleTable(q, O.OracleTable.MAP if a.cpu_backend == "map" else O.OracleTable.FLAT)
    t0 = time.time()
    if a.cpu_backend == "map":
        order = np.argsort(keys, kind="stable")
        tab.insert_packed(keys[order], counts[order], sorted_hint=True)
    else:
        tab.insert_packed(keys, counts)
    tab.decolour()
    build_s = time.time() - t0
    if a.cpu_sample <= 0:
        # pilot on 2 reads per core, then size the sample for ~15 s of wall time
        n0 = min(len(offs) - 1, 2 * cores)
        t0 = time.perf_counter()
        tab.correct_batch(bases[: int(offs[n0])], offs[: n0 + 1].copy(), nthreads=cores)
        per_read = (time.perf_counter() - t0) / max(n0, 1)
        n = int(min(len(offs) - 1, max(4 * cores, 15.0 / max(per_read, 1e-6))))
    sub_off = offs[: n + 1].copy()
    sub_bases = bases[: int(sub_off[n])]
    t0 = time.perf_counter()
    o_out, o_off, o_st = tab.correct_batch(sub_bases, sub_off, nthreads=cores)
    dt = time.perf_counter() - t0
    same = bool(int(o_off[n]) == int(g_off[n]) and np.array_equal(o_out, g_out[: int(g_off[n])])
                and np.array_equal(o_st, g_st[:n]))
    return {
        "value": float(int(sub_off[n]) / dt), "unit": "bases/s", "cores": cores, "kind": "port",
        "sample": "first %d reads of the same workload (%d bases, %.1f s wall), oracle table backend=%s built in %.0f s, "
                  "OpenMP schedule(dynamic) like main.cpp:247" % (n, int(sub_off[n]), dt, a.cpu_backend, build_s),
        "parity_with_gpu_on_sample": same,
    }


if __name__ == "__main__":
    main()
