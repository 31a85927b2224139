#!/usr/bin/env python3
"""What can be known about N = 2/4/8 on a ONE-GPU box: the N-rank collection-level flow with the ranks
TIME-SHARING cuda:0 -- gloo transport, and a cross-process lock that lets only one rank have kernels in flight at a time
(taken when a rank leaves a collective, released when it enters the next), so that the HIP-event time of every launch is
that rank's own kernel time (waiting excluded, no other rank's kernels in between).

Per N it records, per rank: kernel ms by stage, peak device memory; per exchange: bytes.  From these:
  critical path  = sum over stages of the slowest rank's kernel ms      (the ranks run a stage concurrently on N GPUs)
  transfer       = bytes a rank sends / receives, priced at the xGMI point-to-point rate (7 links x ~50 GB/s usable per direction
                   when all peers exchange at once: an all-to-all block crosses ONE link; 153 GB/s per link is the raw figure)
  projected step = critical path + transfer + collectives x latency
and the implied speed-up against the single-GPU build of the same collection, i.e. the Amdahl ceiling of the replicated work.

  python tools/gpu_scale_projection.py [--reads R --genome G] [--ranks 2,4,8]      (parent: spawns torchrun per N)
Results -> gpurun_out/scale_projection.json
"""
import argparse
import fcntl
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

STAGES = [
    ("parse.local (LMS breaks, hashing, table compaction)", lambda s: s in ("lms_breaks", "phrase_ordinals", "hash_sample", "hash_sample_count", "hash_phrases", "hash_nocount", "hash_hot", "byte_hist", "table_compact", "dict_freq_check", "dict_maxlen", "dict_syms", "dict_offsets", "hash_long_list", "hash_long_phrases", "hash_giant_phrases") or s.startswith(("phrase_part", "phrase_dedupe"))),
    ("parse.dictionary exchange + merge", lambda s: s.startswith("dist.") and not any(k in s for k in ("Tpos", "Ppos", "pre_scan", "owners", "cell_bounds", "take_sums", "window", "merge_cells", "piece_maps", "sample_keys", "mark_", "full_", "apply_phrase"))),
    ("parse.dictionary stage (sort, groups, grammar)", lambda s: s.startswith(("dict_build", "suffix_", "group_", "prebwt_", "grammar", "phrase_values", "merge_runs")) or s in ("dist.sample_keys", "dist.mark_scan", "dist.mark_pairs", "dist.full_scan", "dist.full_pairs", "dist.apply_phrase_ranks")),
    ("parse.emit", lambda s: s in ("slot_values", "emit_parse", "dist.list_values", "dist.local_values") or s.startswith("emit_part")),
    ("induce (A+B, exchange prep, C)", lambda s: s.startswith(("induce", "asm.", "parse2bwt", "stat.")) or any(k in s for k in ("dist.Tpos", "dist.Ppos", "dist.pre_scan", "dist.owners", "dist.cell_bounds", "dist.take_sums", "dist.window", "dist.merge_cells", "dist.piece_maps"))),
    ("image", lambda s: s in ("pack_rl_bwt",)),
]


def stage_of(site, phase):
    for name, pred in STAGES:
        if pred(site):
            if name.startswith("parse.dictionary stage") and phase == "i":
                return "induce (A+B, exchange prep, C)"          # merge_runs of pass C
            return name
    return "other"


def worker(args):
    import torch
    import torch.distributed as dist
    import __graft_entry__ as g
    from grlbwt_amd import dist as gdist
    from grlbwt_amd import engine, workloads
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    dev = torch.device("cuda", 0)
    lib = g.build_hip()
    lo, hi = args.reads * rank // world, args.reads * (rank + 1) // world
    lockf = open(args.lock, "r+")

    class SerialComm(gdist.Communicator):          # one rank computes at a time; everybody may sit in a collective
        held = False

        def take(self):
            fcntl.flock(lockf, fcntl.LOCK_EX)
            self.held = True

        def give(self):
            torch.cuda.synchronize()
            if self.held:
                fcntl.flock(lockf, fcntl.LOCK_UN)
            self.held = False

        def _allgather(self, *a):
            self.give()
            rc = super()._allgather(*a)
            self.take()
            return rc

        def _alltoallv(self, *a):
            self.give()
            rc = super()._alltoallv(*a)
            self.take()
            return rc

    comm = SerialComm(dev)
    comm._ag = gdist._AG(comm._allgather)
    comm._a2a = gdist._A2A(comm._alltoallv)
    comm.struct = gdist.CommStruct(comm.rank, comm.size, None, comm._ag, comm._a2a, 0)
    comm.log = []
    comm.take()
    text = workloads.sampled_reads_torch(args.reads, 150, args.genome, seed=20260003, device=dev, read_lo=lo, read_hi=hi)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()                     # (the generator's temporaries: N ranks share ONE device here, what torch caches is lost to the others)
    flags = engine.FLAG_FORCE_IDX64 if args.reads * 151 >= 0xFFFFFF00 else 0
    out = {"rank": rank, "shard_bytes": int(text.numel())}
    with engine.Context(0, flags, lib) as ctx:
        def one():
            ctx.attach_device(text.data_ptr(), text.numel(), 1, keepalive=text)
            if world == 1 and not args.force_dist:
                ctx.build()
            else:
                gdist.dist_build(ctx, comm, keep_parts)
            torch.cuda.synchronize()
        # (the image stays in parts on the ranks that induced them, as in bench.py's timed steps at N > 1 and the executable's --gpus;
        # --gather-image: the all-gather to every rank as well)
        keep_parts = comm is not None and world > 1 and not args.gather_image
        one()                                    # warm-up: the first build of a process pays for the arena, code loading, page faults
        # (--reps profiled builds; a stage's kernel time is its smallest over them: N processes time-sharing one GPU are noisy,
        # and the critical path takes the MAX over the ranks of every stage)
        st_min = {}
        for _ in range(max(1, args.reps)):
            comm.log = []
            ctx.profile_enable(True)
            t0 = time.time()
            one()
            out["wall_s_serialised"] = round(time.time() - t0, 3)
            prof = ctx.profile()
            hs = prof.pop("@host_sync", (0, 0.0, 0))[0]
            out["host_syncs"] = int(hs)
            out["kernel_launches"] = int(sum(c for k, (c, _, _) in prof.items() if not k.startswith("@")))
            st1 = {}
            for k, (c, ms, nb) in prof.items():
                if k.startswith("@xfer:"):
                    continue
                site, _, tag = k.partition("#")
                name = stage_of(site, tag[:1])
                st1[name] = st1.get(name, 0.0) + ms
            for k, v in st1.items():
                st_min[k] = min(st_min.get(k, v), v)
        xfer = {}                                    # bytes this rank sends to other ranks, by exchange site (engine-side accounting)
        for k in [k for k in prof if k.startswith("@xfer:")]:
            c, ms, nb = prof.pop(k)
            site = k[6:].partition("#")[0]
            xfer[site] = xfer.get(site, 0) + nb
        out["sent_bytes_by_site"] = sorted(xfer.items(), key=lambda kv: -kv[1])
        st = st_min
        out["kernel_ms_by_stage"] = {k: round(v, 2) for k, v in st.items()}
        out["kernel_ms_total"] = round(sum(st.values()), 2)
        out["top_sites"] = [[k, round(v[1], 2)] for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])[:12]]
        allsites = {}
        for k, (c, ms, nb) in prof.items():
            site, _, tag = k.partition("#")
            allsites[site] = allsites.get(site, 0.0) + ms
        out["sites"] = [[k, round(v, 2)] for k, v in sorted(allsites.items(), key=lambda kv: -kv[1]) if v >= 0.5]      # (all levels of a site together)
        out["exchanges"] = [[k, b] for k, b, _ in comm.log if b >= (64 << 20)]
        ds = {}
        for k, (c, ms, nb) in prof.items():
            site, _, tag = k.partition("#")
            if stage_of(site, tag[:1]).startswith(("parse.dictionary")):
                ds[site] = ds.get(site, 0.0) + ms
        out["dictionary_sites"] = [[k, round(v, 2)] for k, v in sorted(ds.items(), key=lambda kv: -kv[1])[:16]]
        out["peak_bytes"] = ctx.memory_usage()["peak_live_bytes"]
        nb, nr = ctx.result_size()
        out["image_part"] = list(ctx.result_part())
        if keep_parts:                           # (outside the profile: the whole image on rank 0 for the md5)
            ctx.profile_enable(False)
            keep_parts = False
            saved, comm.log = comm.log, []
            one()
            comm.log = saved
        out["image_md5"] = workloads.md5_device(gdist._view(ctx.result_device_ptr(), nb, dev)) if rank == 0 else None
        out["image_bytes"] = nb
    comm.give()
    ag = [b for k, b, _ in comm.log if k == "allgather"]
    aa = [b for k, b, _ in comm.log if k == "alltoallv"]
    out["collectives"] = {"allgather_calls": len(ag), "allgather_bytes_received": sum(ag), "alltoallv_calls": len(aa), "alltoallv_bytes_sent": sum(aa),
                          "largest_alltoallv_bytes": max(aa) if aa else 0}
    with open(os.path.join(args.out, "rank%d.json" % rank), "w") as f:
        json.dump(out, f)
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=6622517)
    ap.add_argument("--genome", type=int, default=33000000)
    ap.add_argument("--ranks", default="1,2,4,8")
    ap.add_argument("--worker", action="store_true")
    ap.add_argument("--force-dist", action="store_true")
    ap.add_argument("--gather-image", action="store_true")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--lock", default="")
    ap.add_argument("--out", default="")
    ap.add_argument("--tag", default="")
    args = ap.parse_args()
    if args.worker:
        return worker(args)
    import tempfile
    res = {"collection": {"reads": args.reads, "genome": args.genome, "bytes": args.reads * 151},
           "method": "ranks time-share one MI355X, gloo transport, one rank computes at a time (flock): HIP-event kernel ms per rank and stage, the smallest of %d profiled builds; the image stays in parts%s" % (args.reps, " and is all-gathered" if args.gather_image else ""), "runs": []}
    # xGMI: 7 links per GPU, ~153 GB/s raw per link; an all-to-all block to one peer crosses one link.  Two prices: 0.7 x the raw
    # figure per direction (if 153 GB/s is what one direction carries), and 0.4 x (if it is both directions together: ~61 GB/s,
    # about what RCCL point-to-point copies reach on the previous generation's 64 GB/s-per-direction links)
    link = 0.7 * 153e9
    link_slow = 0.4 * 153e9
    base_ms = None
    for n in [int(x) for x in args.ranks.split(",")]:
        with tempfile.TemporaryDirectory() as td:
            lock = os.path.join(td, "lock")
            open(lock, "w").close()
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
                   "--master-port", str(29700 + n), os.path.abspath(__file__), "--worker", "--reads", str(args.reads), "--genome", str(args.genome),
                   "--lock", lock, "--out", td] + (["--gather-image"] if args.gather_image else []) + ["--reps", str(args.reps)]
            t0 = time.time()
            p = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT)
            if os.environ.get("GRLBWT_MEM_TRACE"):      # per-stage peak memory of every rank (stderr of the workers)
                os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                with open(os.path.join(ROOT, "gpurun_out", "scale_projection%s_n%d.stderr" % (args.tag, n)), "w") as f:
                    f.write("\n".join(l for l in p.stderr.splitlines() if "[grlbwt] stage" in l or "rror" in l))
            if p.returncode != 0:
                err = [l for l in p.stderr.splitlines() if "rror" in l and "Signal 15" not in l and "error_file" not in l]
                res["runs"].append({"ranks": n, "failed": p.returncode, "errors": err[:12], "stderr": p.stderr[-1500:]})
                print(json.dumps(res["runs"][-1]), flush=True)
                continue
            ranks = [json.load(open(os.path.join(td, "rank%d.json" % r))) for r in range(n)]
        stages = sorted({k for r in ranks for k in r["kernel_ms_by_stage"]})
        by_stage = {s: {"max_rank_ms": round(max(r["kernel_ms_by_stage"].get(s, 0.0) for r in ranks), 2),
                        "min_rank_ms": round(min(r["kernel_ms_by_stage"].get(s, 0.0) for r in ranks), 2)} for s in stages}
        crit = sum(v["max_rank_ms"] for v in by_stage.values())
        sent = max(r["collectives"]["alltoallv_bytes_sent"] for r in ranks)
        recv_ag = max(r["collectives"]["allgather_bytes_received"] for r in ranks)
        ncoll = max(r["collectives"]["allgather_calls"] + r["collectives"]["alltoallv_calls"] for r in ranks)
        # all-to-all: a rank's bytes leave over its 7 links in parallel when the peers are distinct (N-1 of them); all-gather of B
        # bytes in total: every rank receives B*(N-1)/N over its links
        links = max(1, min(7, n - 1))
        # (`sent` = bytes to OTHER ranks: a rank's own block of an exchange is a local copy and never reaches the callback.  Rounds
        # 3-4 multiplied by (n - 1) / n once more -- right when the own block still went through the callback, too optimistic since:
        # by 2 x at N = 2, 4/3 x at N = 4)
        xfer_ms = 0.0 if n == 1 else (sent / (links * link) + recv_ag * (n - 1) / n / (links * link)) * 1e3
        xfer_slow_ms = xfer_ms * link / link_slow
        # What the kernels and the wire leave out (round 6, measured with ONE rank over RCCL on the 10 GB collection --
        # profiles/r06/rccl_world1_overhead.md: +293 ms of wall against the plain build, +224 ms of them kernels that this tool sees
        # per rank and stage): the host side of the sharded flow -- a synchronisation in front of every counter exchange and size
        # readback (420 against 190), the slab pool of the RCCL path -- came to 69 ms.  It does not shrink with N (the same exchanges
        # at any N); it is added as a CONSTANT, scaled with the collection (the runs here synchronise once more per exchange because
        # gloo is not stream-ordered: their own count would overstate it).
        # A collective: the 81 small all-gathers of that run took 37 us each through torch; a grouped send/recv with N - 1 peers
        # is ASSUMED to cost that plus 10 us per peer (no multi-GPU node has run it: tools/gpu_rccl_sizes.py measures payload sizes
        # at world 1 only).
        nsync = max(r.get("host_syncs", 0) for r in ranks)
        host_ms = 0.0 if n == 1 else 69.0 * min(1.0, (args.reads * 151) / 1e10)
        lat_ms = 0.0 if n == 1 else ncoll * (0.037 + 0.010 * (n - 1))
        run = {"ranks": n, "wall_s": round(time.time() - t0, 1), "kernel_ms_by_stage": by_stage, "critical_path_kernel_ms": round(crit, 2),
               "max_rank_total_kernel_ms": max(r["kernel_ms_total"] for r in ranks), "peak_bytes_max_rank": max(r["peak_bytes"] for r in ranks),
               "bytes_sent_alltoallv_max_rank": sent, "bytes_received_allgather_max_rank": recv_ag, "collective_calls": ncoll,
               "projected_transfer_ms": round(xfer_ms, 2), "projected_latency_ms": round(lat_ms, 2), "projected_host_ms": round(host_ms, 2),
               "host_syncs_max_rank": nsync, "kernel_launches_max_rank": max(r.get("kernel_launches", 0) for r in ranks),
               "projected_step_ms": round(crit + xfer_ms + lat_ms + host_ms, 2),
               "projected_transfer_ms_slow_links": round(xfer_slow_ms, 2), "projected_step_ms_slow_links": round(crit + xfer_slow_ms + lat_ms + host_ms, 2),
               "image_md5": ranks[0].get("image_md5"), "image_bytes": ranks[0]["image_bytes"],
               "top_sites_rank0": ranks[0]["top_sites"], "dictionary_sites_rank0": ranks[0]["dictionary_sites"],
               "sites_rank0": ranks[0].get("sites"), "large_exchanges_rank0": ranks[0].get("exchanges"),
               "sent_bytes_by_site_max_rank": sorted({k: max(dict(r.get("sent_bytes_by_site", [])).get(k, 0) for r in ranks) for k in
                                                      {k for r in ranks for k, _ in r.get("sent_bytes_by_site", [])}}.items(), key=lambda kv: -kv[1])}
        if n == 1:
            base_ms = run["projected_step_ms"]
        if base_ms:
            run["projected_speedup_vs_1"] = round(base_ms / run["projected_step_ms"], 2)
            run["projected_speedup_vs_1_slow_links"] = round(base_ms / run["projected_step_ms_slow_links"], 2)
        res["runs"].append(run)
        print(json.dumps({k: run[k] for k in ("ranks", "critical_path_kernel_ms", "projected_transfer_ms", "projected_step_ms", "projected_step_ms_slow_links", "peak_bytes_max_rank", "image_md5")} |
                         ({"speedup": run.get("projected_speedup_vs_1")})), flush=True)
    md5s = {r.get("image_md5") for r in res["runs"] if r.get("image_md5")}
    res["same_image_for_every_N"] = len(md5s) <= 1
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "scale_projection%s.json" % args.tag), "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
