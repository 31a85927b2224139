#!/usr/bin/env python3
"""bench.py -- input MB/s building the BCR BWT (.rl_bwt image) on MI355X.

A "step" is one pass of the whole parse-then-induce path over the workload,
input cells already resident in HBM, output .rl_bwt image left in HBM.
N=1 workload: BASELINE.json configs[1] (1,000,000 x 100 bp uniform ACGT reads,
101,000,000 bytes).  N>1: records are sharded (weak scaling: one such shard per
GPU, different seeds), one process per GPU; the ranks build ONE BWT of the whole sharded collection
(grlbwt_amd/dist.py), so value = total input bytes of the collection / time.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, algorithmic bytes
/ HIP-event time on the engine's stream) and `cpu_baseline` (the CPU oracle,
"port", single core, on a bounded prefix sample of the same workload).
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def algorithmic_bytes(name, rounds, levels, cell_bytes, idx_bytes):
    """Must-touch bytes of ONE launch-set of kernel `name` summed over its launches."""
    ib = idx_bytes
    if name == "hash_phrases":        # n_r*w_r read + n_{r+1}*4 slot ids written
        return sum(r["n_in"] * (cell_bytes if i == 0 else 4) + r["parse_size"] * 4 for i, r in enumerate(rounds))
    if name == "lms_breaks":          # n_r*w_r read + n_r/8 bits written
        return sum(r["n_in"] * (cell_bytes if i == 0 else 4) + r["n_in"] // 8 for i, r in enumerate(rounds))
    if name == "emit_parse":
        return sum(r["parse_size"] * 8 for r in rounds)
    if name == "induce_split.scatter":  # stable multi-split of the induced cells: key+index in, key+index out (one pass minimum)
        return sum(l["induced_cells"] * (4 + ib) * 2 for l in levels)
    if name == "induce_expand":       # R*(4+ib) runs read + R*4 rewritten + E*(4+4+ib+ib) cells written
        return sum(l["runs_next"] * (8 + ib) + l["induced_cells"] * (8 + 2 * ib) for l in levels)
    if name == "induce_count":
        return sum(l["runs_next"] * (4 + ib) for l in levels)
    if name == "asm.atoms":
        return sum(l["atoms"] * (4 + ib) + l["segments"] * (4 + 2 * ib) for l in levels)
    return None


def pmc_traffic_per_launch(kernel_substr, n_launches):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC summary
    (profiles/<round>/pmc_traffic.json, collected with separate --pmc FETCH_SIZE / WRITE_SIZE passes of this
    same command; FETCH_SIZE doubled per the gfx950 correction).  None when no summary is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "pmc_traffic.json")))
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))
        tot = 0.0
        launches = 0
        for name, e in d["kernels"].items():
            if kernel_substr in name:
                tot += 2.0 * e.get("fetch_kib_total", 0.0) * 1024 + e.get("write_kib_total", 0.0) * 1024
                launches += max(e.get("fetch_launches", 0), e.get("write_launches", 0))
        return round(tot / launches) if launches else None
    except Exception:
        return None


def main():
    # stdout carries exactly one line (the JSON result): anything a library prints to fd 1 on the way
    # (RCCL's version banner at communicator creation) is sent to stderr instead
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=1000000)
    ap.add_argument("--read-len", type=int, default=100)
    ap.add_argument("--workload", default="uniform", choices=["uniform", "illumina"],
                    help="uniform: BASELINE configs[1] shape (i.i.d. ACGT reads); illumina: configs[3] shape "
                         "(--reads x 150 bp sampled from a --genome bp random genome, 0.5 %% substitutions)")
    ap.add_argument("--genome", type=int, default=330000000)
    ap.add_argument("--cpu-sample-reads", type=int, default=600000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import __graft_entry__ as g
    from grlbwt_amd import engine, workloads

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    force_dist = os.environ.get("GRLBWT_BENCH_FORCE_DIST") == "1"   # measure the sharded flow's overhead at world 1
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if os.environ.get("GRLBWT_BENCH_BACKEND") == "gloo":     # functional check of the N > 1 flow on a 1-GPU box
            local_rank = 0
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback)"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    lib = os.environ.get("GRLBWT_HIP_LIB") or g.build_hip()
    # shard of this rank: the named workload (rank 0) / same shape with another seed (other ranks)
    # generated on the device (bit-identical to workloads.uniform_reads on the host; tests/test_gpu_parity.py)
    if args.workload == "illumina":
        args.read_len = 150
        text = workloads.sampled_reads_torch(args.reads, 150, args.genome, seed=20260003 + 10 * rank, device=dev)
    else:
        text = workloads.uniform_reads_torch(args.reads, args.read_len, seed=20260001 + rank, device=dev)
    n_bytes = int(text.numel())
    torch.cuda.synchronize()

    comm = None
    flags = 0
    if world > 1 or force_dist:
        from grlbwt_amd import dist as gdist
        comm = gdist.Communicator(dev)
        if world * n_bytes >= 0xFFFFFF00:
            flags |= engine.FLAG_FORCE_IDX64
    ctx = engine.Context(local_rank, flags, lib)

    def step():
        ctx.attach_device(text.data_ptr(), text.numel(), 1, keepalive=text)
        if comm is None:
            ctx.build()
        else:
            gdist.dist_build(ctx, comm)      # BWT of the whole collection (all shards), identical on every rank

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = world * n_bytes * args.steps / dt / 1e6

    # ---- roofline leg: one more step with HIP-event timing of every kernel on the engine's stream.
    # Every rank takes the step (for N > 1 it contains collectives); only rank 0 records and reports.
    if rank == 0:
        ctx.profile_enable(True)
        if comm is not None and os.environ.get("GRLBWT_BENCH_DETAIL"):
            comm.log = []
    step()
    barrier()
    out = None
    if rank == 0:
        prof_detail = ctx.profile()
        host_syncs = prof_detail.pop("@host_sync", (0, 0.0, 0))[0]
        ctx.profile_enable(False)
        # fold the per-level tags ("name#level") and group launch sites by the kernel FUNCTION they launch, so that
        # "dominant kernel" means the same thing as in the rocprofv3 --stats summary under profiles/
        def kernel_of(site):
            if site.endswith(".scatter"):
                return "k_rs_scatter"
            if site.endswith(".hist"):
                return "k_rs_hist"
            return site
        prof, fam = {}, {}
        for k, (c, ms, nb) in prof_detail.items():
            b = k.split("#")[0]
            pc, pm, pb = prof.get(b, (0, 0.0, 0))
            prof[b] = (pc + c, pm + ms, pb + nb)
            f = kernel_of(b)
            fc, fm, fb = fam.get(f, (0, 0.0, 0))
            fam[f] = (fc + c, fm + ms, fb + nb)
        if os.environ.get("GRLBWT_BENCH_DETAIL") and comm is not None and comm.log is not None:
            for kind, nb, sec in comm.log:
                print("  collective %-10s %12d bytes %8.3f ms" % (kind, nb, sec * 1e3), file=sys.stderr)
        if os.environ.get("GRLBWT_BENCH_DETAIL"):
            for k, (c, ms, nb) in sorted(prof_detail.items(), key=lambda kv: -kv[1][1])[:90]:
                print("  %-32s %4d launches %9.3f ms" % (k, c, ms), file=sys.stderr)
        nr = 0
        rounds = []
        while True:
            try:
                rounds.append(ctx.round_info(nr))
                nr += 1
            except engine.GrlbwtError:
                break
        levels = [ctx.level_info(l) for l in range(nr)]
        cnt = ctx.counters()
        ib = cnt["idx_bytes"]

        def algo_bytes(kern):
            c, ms, nb = fam[kern]
            if nb:                       # stated at the launch site (radix sort passes: pairs read once + written once)
                return nb
            return algorithmic_bytes(kern, rounds, levels, 1, ib)

        total_kernel_ms = sum(ms for _, ms, _ in fam.values())
        ranked = sorted(fam.items(), key=lambda kv: -kv[1][1])
        dom = None
        for name, (launches, ms, nb) in ranked:
            ab = algo_bytes(name)
            if ab:
                dom = (name, launches, ms, ab)
                break
        roofline = None
        if dom:
            name, launches, ms, ab = dom
            achieved = ab / (ms * 1e-3) / 1e9
            rocprof_name = {"hash_phrases": "HashInsertFn", "k_rs_scatter": "k_rs_scatter", "k_rs_hist": "k_rs_hist"}.get(name, name)
            roofline = {"bound": "hbm", "kernel": name, "launches": launches,
                        "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 5),
                        "traffic": pmc_traffic_per_launch(rocprof_name, launches),
                        "algorithmic_bytes": ab, "algorithmic_bytes_per_launch": round(ab / max(launches, 1)),
                        "kernel_ms_total": round(ms, 4), "avg_launch_ms": round(ms / max(launches, 1), 5),
                        "share_of_kernel_time": round(ms / max(total_kernel_ms, 1e-9), 4)}
        top = [{"kernel": k, "launches": c, "ms": round(ms, 3)} for k, (c, ms, _) in ranked[:12]]
        # the same accounting for every kernel with a stated algorithmic byte count
        others = []
        for kname, (launches, ms, nb) in sorted(list(fam.items()) + [(k, v) for k, v in prof.items() if k not in fam],
                                                key=lambda kv: -kv[1][1]):
            ab = nb if nb else algorithmic_bytes(kname, rounds, levels, 1, ib)
            if ab and ms > 0:
                others.append({"kernel": kname, "launches": launches, "ms": round(ms, 4),
                               "achieved_GBps": round(ab / (ms * 1e-3) / 1e9, 2), "frac": round(ab / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)})

        # ---- CPU baseline leg (reported, not the target): the oracle ("port"), 1 core, bounded sample
        cpu = None
        if not args.no_cpu_baseline and world == 1:      # reported at N = 1 only
            from oracle import oracle
            oracle.build()
            m = min(args.cpu_sample_reads, args.reads)
            sample = text[: m * (args.read_len + 1)].cpu().numpy()
            with tempfile.TemporaryDirectory() as td:
                fi, fo = os.path.join(td, "in.txt"), os.path.join(td, "out.rl_bwt")
                sample.tofile(fi)
                tc = time.perf_counter()
                subprocess.check_call([oracle.CLI, fi, fo, "-q"])
                tcpu = time.perf_counter() - tc
            cpu = {"value": round(sample.size / 1e6 / tcpu, 3), "unit": "MB/s", "cores": 1, "kind": "port",
                   "sample": "first %d reads (%d bytes) of the same workload, oracle/oracle_cli, %.1f s"
                             % (m, sample.size, tcpu),
                   "host_cpus": os.cpu_count()}
        out = {
            "metric": "input MB/s building BCR BWT (.rl_bwt) of DNA reads", "value": round(value, 3), "unit": "MB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": ("%d x %d bp uniform ACGT reads per GPU (%d bytes), sigma=5, byte alphabet; BASELINE configs[1]"
                                    % (args.reads, args.read_len, n_bytes)) if args.workload == "uniform" else
                                   ("%d x 150 bp Illumina-style reads per GPU (%d bytes) from a %d bp genome, 0.5%% substitutions; "
                                    "BASELINE configs[3] shape" % (args.reads, n_bytes, args.genome)),
                       "input_resident": "HBM", "output": ".rl_bwt image in HBM", "parallelism": ("1 GPU" if world == 1 else "ONE collection of %d record shards, one per GPU: local hashing/emission, RCCL all-gather "
                                       "dictionary merge + key-range-sharded dictionary stage per round, induction sharded by BWT position "
                                       "(per-bucket rank-count exchange, all-to-all atom routing)" % world)},
            "roofline": roofline, "cpu_baseline": cpu,
            "stage_seconds": {k: round(v, 5) for k, v in cnt.items() if k.startswith("t_")},
            "top_kernels": top, "roofline_by_kernel": others, "rounds": nr,
            "kernel_launches": sum(c for c, _, _ in prof_detail.values()), "host_syncs": host_syncs,
        }
        if comm is not None:     # totals over warmup + timed + profile steps on rank 0
            nsteps = args.warmup + args.steps + 1
            out["collectives_per_step"] = {"allgather": comm.n_allgather // nsteps, "alltoallv": comm.n_alltoall // nsteps,
                                           "bytes": comm.bytes_moved // nsteps, "ms_in_callbacks": round(comm.seconds / nsteps * 1e3, 3)}
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        os.write(result_fd, (json.dumps(out) + "\n").encode())


if __name__ == "__main__":
    main()
