#!/usr/bin/env python3
"""bench.py -- input MB/s building the BCR BWT (.rl_bwt image) on MI355X.

A "step" is one pass of the whole parse-then-induce path (grl_bwt_algo: par_phase + ind_phase) over the
workload, input cells already resident in HBM, output .rl_bwt image left in HBM.

Workload (default): BASELINE.json's metric configuration -- configs[3], 66,225,166 x 150 bp Illumina-style
reads from a 330 Mbp genome, 0.5 % substitutions, 10,000,000,066 bytes.  At N = 1 the whole collection is on
one GPU.  At N > 1 the SAME collection is sharded by record (rank g holds reads [g*R/N, (g+1)*R/N)), one
process per GPU, and the ranks build ONE BWT of the whole collection (grlbwt_dist_build with grlbwt_amd/dist.py's
RCCL callbacks: hash-partitioned dictionary merge + key-range-sharded dictionary stage per round; per induction level
the cells and the BWT_{r+1} windows go once to the owners of the output pieces): strong scaling,
value = collection bytes / time.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself
(`python -m torch.distributed.run` as a child process, before this process touches torch or the GPU) and
relays rank 0's JSON line; under torchrun (WORLD_SIZE set) it is one of the ranks.

Prints ONE JSON line (rank 0) with
  roofline      the induction pass A+B group (the kernels of exact_ind_phase.cpp:42-109,143-258): SURVEY.md 8(d)'s
                algorithmic bytes / the summed HIP-event time of the group's launches on the engine's stream
  cpu_baseline  the CPU oracle ("port", 1 core) on a bounded prefix sample of the same workload
and the same accounting for the other groups (roofline_groups), per-pass scatter efficiency (pass_efficiency),
and the 101 MB configs[1] figure as an extra key.
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


def bitlen(v):
    return int(v).bit_length()


def cdiv8(bits):
    return max(1, (bits + 7) // 8)


# ---------------------------------------------------------------------------------------------
# algorithmic (must-touch) bytes per level in the REFERENCE's own cell widths, so that the
# figure does not move with this engine's layout choices.  r = level, sigma_r its alphabet, M_r its metasymbols.
#   sbN, fbN   bytes per run symbol / run length of bwt_lev_{r+1}: symbols are metasymbols < M_r, rewritten in place to
#              symbols < sigma_r+3 (exact_ind_phase.cpp:257), lengths are bounded by n_{r+1}
#   al_b, b    hocc cell: ceil(sym_width(sigma_r+3)/8) symbol bytes + the -b run-length bytes (default 1)
#   c_r        grammar cell: ceil(sym_width(metasym_dummy)/8), metasym_dummy = sigma_r+3+M_r+1 (exact_par_phase.cpp:19-20)
#   pass A+B   R_{r+1}*(sbN+fbN) read + R_{r+1}*sbN rewritten + E_r*(al_b+b) written + 2*c_r*E'_r gathered
#   pass C     P_r*(sbP+fbP) + E_r*(al_b+b) + R_{r+1}*(sbN+fbN) read + R_r*(sbR+fbR) written
#   E_r = hocc cells after the in-bucket merge, E'_r = grammar-chain steps (both counted by the engine while profiling)
def induction_bytes(rounds, levels, r, run_len_bytes=1):
    rd, lv = rounds[r], levels[r]
    sigma3 = rd["sigma"] + 3
    M = rd["n_metasyms"]
    n_next = levels[r + 1]["n"]
    sbN, fbN = cdiv8(bitlen(max(M, sigma3))), cdiv8(bitlen(n_next))
    cell = cdiv8(bitlen(sigma3)) + run_len_bytes
    c_r = cdiv8(bitlen(sigma3 + M + 1))
    E = lv["merged_cells"] or lv["induced_cells"]
    Es = lv["chain_steps"] or lv["induced_cells"]
    sigma_prev = rounds[r - 1]["sigma"] + 3 if r > 0 else 0
    sbR, fbR = cdiv8(bitlen(max(sigma3, sigma_prev))), cdiv8(bitlen(lv["n"]))
    sbP, fbP = cdiv8(bitlen(sigma3)), cdiv8(bitlen(lv["n"]))
    ab = lv["runs_next"] * (sbN + fbN) + lv["runs_next"] * sbN + E * cell + 2 * c_r * Es
    c = lv["prebwt_runs"] * (sbP + fbP) + E * cell + lv["runs_next"] * (sbN + fbN) + lv["n_runs"] * (sbR + fbR)
    return ab, c


def dict_bytes(rounds, levels):
    """dictionary stage (outside SURVEY 8(d)'s pricing): per round every dictionary suffix as a
    (key, position) record read and written ONCE -- S_r * (8 + 4) * 2 -- + the grammar written, 2 * c_r * M_r, + the
    pre-BWT written, P_r * (sbP + fbP), in the reference's cell widths."""
    tot = 0
    for r, rd in enumerate(rounds):
        sigma3 = rd["sigma"] + 3
        c_r = cdiv8(bitlen(sigma3 + rd["n_metasyms"] + 1))
        sbP, fbP = cdiv8(bitlen(sigma3)), cdiv8(bitlen(rd["n_in"]))
        tot += rd["dict_syms"] * 12 * 2 + 2 * c_r * rd["n_metasyms"] + levels[r]["prebwt_runs"] * (sbP + fbP)
    return tot


def parse_bytes(rounds, cell_bytes):
    """classify+hash pass: n_r*w_r read; lookup-emit pass: n_r*w_r read + n_{r+1}*w_{r+1} written (w = 4 above level 0)."""
    hash_b = emit_b = 0
    for i, r in enumerate(rounds):
        w = cell_bytes if i == 0 else 4
        hash_b += r["n_in"] * w
        emit_b += r["n_in"] * w + r["parse_size"] * 4
    return hash_b, emit_b


# launch sites (prim::prof names, "#<phase><level>" stripped) of each accounting group
def _is_hash_emit(s):
    # (round 3: the partitioned naming of the levels above 0 -- record pass, partition sort, LDS de-duplication, the values' way back)
    return s in ("lms_breaks", "phrase_ordinals", "hash_sample", "hash_sample_count", "hash_phrases", "slot_values", "emit_parse",
                 "hash_prepare", "hash_lookup", "hash_hot", "hash_long_list", "hash_long_phrases", "hash_giant_phrases") or \
        s.startswith(("phrase_part", "phrase_dedupe", "emit_part"))


GROUP_SITES = {
    "induce_AB": lambda s, ph: ph == "i" and (s == "induce" or s.startswith("induce_") or s.startswith("induce.")),
    "induce_C": lambda s, ph: ph == "i" and (s.startswith("asm.") or s.startswith("merge_runs")),
    "hash_emit": lambda s, ph: ph == "p" and _is_hash_emit(s),
    # everything else of a parsing round: dictionary compaction and build, suffix sort + refinement, groups -> pre-BWT and
    # ranks, grammar (reference: dictionary ctor, suffix_induction, produce_pre_bwt, produce_grammar -- rows a5-a9)
    "dict_stage": lambda s, ph: ph == "p" and not _is_hash_emit(s),
}
# rocprofv3 kernel-name fragments of the kernels a group launches (PMC traffic lookup)
GROUP_KERNELS = {
    "induce_AB": ["k_xs_count", "k_xs_scatter", "PackGrammarFn", "ChainCountFn", "ChainExpandFn", "NoVal, 1, ", "k_rs_hist<unsigned long, 1,", "k_rs_hist<unsigned int, 1,"],
    "induce_C": ["k_sm_sums", "k_sm_merge", "k_sm_wide", "SmHeadsIn", "SmWideCountIn", "TermHeadIn", "RankCellPopcIn", "PrePlaceFn", "BucketEdgesFn",
                 "BucketSizeIn", "NotCodeIn", "BuildBitsFn"],
    "hash_emit": ["HashInsertFn", "k_giant", "k_start_bits", "::MapFn>", "ScatterValFn", "PhraseRecordFn", "BitPositionsFn", "k_rec_dedupe", "k_rs_unscatter",
                  "k_rs_scatter<unsigned long, unsigned long, 2,", "k_rs_hist<unsigned long, 2,", "RecBoundsFn", "PartPhraseFn", "PartValFn", "PartCombineFn"],
    "dict_stage": ["ClaimCompactFn", "DictBuildFn", "Key0KeepFn", "unsigned int, 0, ", "k_rs_hist<unsigned long, 0,", "k_rs_hist<unsigned int, 0,", "HeadFlagFn", "FirstUnresolvedFn",
                   "ExtKeyFn", "SegStartFn", "SegSortSmallFn", "SegBig", "GroupStartsFn", "DenseGidFn", "SuffixRecFn", "GroupAccum", "GroupDecideFn",
                   "GroupEmitFn", "PackGroupInfoFn", "MetaPosFn", "::GrammarFn>", "PhraseValFn", "ComposeMapFn", "PreToMetaFn", "BuildBits32Fn"],
}


# kernels of fall-back branches the headline workload never launches (unfused chain expansion, giant phrases, phrase values by table
# slot): their fragments are looked up when they run, and are no evidence of a stale summary when they do not
OPTIONAL_KERNELS = {"ChainCountFn", "ChainExpandFn", "k_giant", "ScatterValFn"}


def pmc_traffic_files():
    import glob
    return sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "pmc_traffic.json")))


def pmc_traffic_source():
    """Where `traffic` comes from: NOT measured in this run -- the newest committed rocprofv3 PMC summary of this same command."""
    files = pmc_traffic_files()
    return os.path.relpath(files[-1], ROOT) if files else None


def pmc_traffic(kernel_fragments):
    """HBM bytes (FETCH_SIZE + WRITE_SIZE, corrected as the microarchitecture guide prescribes) summed over the
    kernels whose names contain one of `kernel_fragments`, per build, from the committed rocprofv3 PMC passes of
    this same command (profiles/<round>/pmc_traffic.json).  None when no summary is committed."""
    files = pmc_traffic_files()
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))
        steps = max(1, int(d.get("builds_profiled", 1)))
        tot = 0.0
        hit = False
        for name, e in d["kernels"].items():
            if any(f in name for f in kernel_fragments):
                hit = True
                tot += e.get("hbm_bytes_total", 0.0)
        return round(tot / steps) if (hit and tot > 0) else None
    except Exception:
        return None


def pmc_traffic_stale(kernel_fragments):
    """The name fragments of a group that match NO kernel of the committed PMC summary: a kernel that was renamed or re-templated
    since the summary was taken silently drops out of `traffic` otherwise (rounds 3-4 reported A+B without its later radix passes
    that way).  A non-empty list goes into the line as `traffic_stale`."""
    files = pmc_traffic_files()
    if not files:
        return []
    try:
        names = list(json.load(open(files[-1]))["kernels"])
    except Exception:
        return list(kernel_fragments)
    return [f for f in kernel_fragments if f not in OPTIONAL_KERNELS and not any(f in n for n in names)]


def spawn_ranks(args, argv):
    """--gpus N without a launcher: start N ranks as a CHILD (this process has not imported torch or touched the
    GPU), relay rank 0's line, exit with the child's code."""
    port = int(os.environ.get("MASTER_PORT", "29517"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, cwd=ROOT)
    out, _ = p.communicate()
    line = None
    for ln in out.decode(errors="replace").splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    if line is not None:
        sys.stdout.write(line + "\n")
        sys.stdout.flush()
    if p.returncode == 0 and line is None:
        sys.stderr.write("bench.py: the ranks printed no result line\n")
        sys.exit(3)
    sys.exit(p.returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="illumina", choices=["illumina", "uniform"],
                    help="illumina: BASELINE configs[3], the metric's configuration (--reads x 150 bp sampled from a --genome bp "
                         "random genome, 0.5 %% substitutions); uniform: configs[1] shape (i.i.d. ACGT reads of --read-len)")
    ap.add_argument("--reads", type=int, default=None, help="reads of the WHOLE collection (default 66225166 / 1000000)")
    ap.add_argument("--read-len", type=int, default=100)
    ap.add_argument("--genome", type=int, default=330000000)
    ap.add_argument("--cpu-sample-reads", type=int, default=400000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the 101 MB configs[1] side measurement")
    ap.add_argument("--no-cli", action="store_true", help="skip the end-to-end leg through the grlbwt executable (file in -> .rl_bwt file closed)")
    args = ap.parse_args()
    if args.reads is None:
        args.reads = 66225166 if args.workload == "illumina" else 1000000
    if args.workload == "illumina":
        args.read_len = 150

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args, sys.argv[1:])        # never returns

    # stdout carries exactly one line (the JSON result): anything a library prints to fd 1 on the way
    # (RCCL's version banner at communicator creation) is sent to stderr instead
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d: refusing to mislabel the run\n" % (args.gpus, world))
        sys.exit(2)

    import torch
    import __graft_entry__ as g
    from grlbwt_amd import engine, workloads

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

    # this rank's record shard of the ONE collection, generated on the device (bit-identical to the host generators
    # of grlbwt_amd/workloads.py; tests/test_gpu_parity.py)
    lo, hi = args.reads * rank // world, args.reads * (rank + 1) // world

    def make_text(workload, reads, read_len, lo, hi):
        if workload == "illumina":
            return workloads.sampled_reads_torch(reads, 150, args.genome, seed=20260003, device=dev, read_lo=lo, read_hi=hi)
        full = workloads.uniform_reads_torch(reads, read_len, seed=20260001, device=dev)     # small: slice the whole
        return full[lo * (read_len + 1): hi * (read_len + 1)].clone()

    text = make_text(args.workload, args.reads, args.read_len, lo, hi)
    n_bytes = int(text.numel())
    total_bytes = args.reads * (args.read_len + 1)
    torch.cuda.synchronize()

    cli_e2e = None

    comm = None
    flags = 0
    if world > 1 or force_dist:
        from grlbwt_amd import dist as gdist
        comm = gdist.Communicator(dev)
        if total_bytes >= 0xFFFFFF00:
            flags |= engine.FLAG_FORCE_IDX64
        flags |= gdist.pool_flags(comm)
    ctx = engine.Context(local_rank, flags, lib)

    # N > 1: the image of the timed steps stays in parts on the ranks that induced them (GRLBWT_COMM_KEEP_PARTS: what the grlbwt
    # executable's --gpus does -- N ranks write one file at N offsets); GRLBWT_BENCH_GATHER_IMAGE=1 times the all-gather to every
    # rank as well.  The steps after the timed region gather it, for the md5 and the comparison with a single-GPU build.
    keep_parts = comm is not None and world > 1 and os.environ.get("GRLBWT_BENCH_GATHER_IMAGE") != "1"

    def step(c=ctx, t=text, parts=None):
        c.attach_device(t.data_ptr(), t.numel(), 1, keepalive=t)
        if comm is None:
            c.build()
        else:
            # BWT of the whole collection (all shards): the same image for every N
            gdist.dist_build(c, comm, keep_parts if parts is None else parts)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, steps, warmup):
        for _ in range(warmup):
            fn()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    # which device every rank runs on (a SCALE record must show N distinct GPUs: VERDICT r4 13c)
    def device_id():
        pr = torch.cuda.get_device_properties(dev)
        d = {"rank": rank, "local_rank": local_rank, "hip_device": torch.cuda.current_device(), "name": pr.name,
             "total_memory": int(pr.total_memory)}
        for k in ("uuid", "pci_bus_id", "pci_device_id", "pci_domain_id"):
            if hasattr(pr, k):
                d[k] = str(getattr(pr, k))
        return d
    devices = [device_id()]
    if dist is not None and world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, devices[0])
        devices = gathered

    dt = timed(step, args.steps, args.warmup)
    # (the collectives of the warm-up and timed steps only: the steps after the timed region -- the gathered image, the profiled
    # build -- do not count into collectives_per_step)
    comm_timed = (comm.n_allgather, comm.n_alltoall, comm.bytes_moved, comm.seconds) if comm is not None else None
    ms_per_step = dt / args.steps * 1e3
    value = total_bytes * args.steps / dt / 1e6

    # the image the timed steps produced (outside the timed region): size, runs, md5 -- so that a driver run and a lease run
    # can be compared, and tests/test_gpu_parity.py::test_headline_10GB_round_trip decodes exactly this image
    parts_span = None
    if keep_parts:
        parts_span = ctx.result_part()
        step(parts=False)                 # (untimed, every rank: the whole image on every rank from here on)
        barrier()
    image = None
    if rank == 0 and os.environ.get("GRLBWT_BENCH_MD5", "1") != "0":
        from grlbwt_amd import dist as _gd
        nb_img, nr_img = ctx.result_size()
        image = {"bytes": nb_img, "runs": nr_img, "md5": workloads.md5_device(_gd._view(ctx.result_device_ptr(), nb_img, dev))}

    # ---- SURVEY 8(d)'s metric as worded: wall time of the whole CLI run, file read -> .rl_bwt closed (main.cpp:98-154,
    # grl_bwt.hpp:77), through grlbwt_amd/bin/grlbwt on the SAME bytes (file in the page cache), a warm-up run and three timed ones, best reported;
    # outside the timed region and AFTER this process's own builds: device memory that no process has touched since the box
    # came up is slow to back (the first 90 GB cost a build 2.7-6 s on a fresh box, tens of milliseconds afterwards), so the
    # first child run still pays for the pages this process does not hold -- the later ones show the steady state.  This
    # process keeps its arena (~90 GB) and the text while the child runs; the device has room for both.
    if rank == 0 and world == 1 and not force_dist and not args.no_cli:
        import hashlib
        import re
        import shutil
        torch.cuda.empty_cache()                 # (the generator's temporaries: the child process needs the device's memory)
        tmpdir = os.environ.get("GRLBWT_E2E_TMP") or tempfile.gettempdir()
        need = 2 * n_bytes + (1 << 30)
        if shutil.disk_usage(tmpdir).free < need:
            cli_e2e = {"skipped": "less than %d bytes free under %s" % (need, tmpdir)}
        else:
            cli = g.build_cli()
            fin, fout = os.path.join(tmpdir, "grlbwt_bench_in.txt"), os.path.join(tmpdir, "grlbwt_bench_out.rl_bwt")
            try:
                stage = torch.empty(min(1 << 28, n_bytes), dtype=torch.uint8, pin_memory=True)
                with open(fin, "wb") as f:
                    for a in range(0, n_bytes, 1 << 28):
                        m = min(1 << 28, n_bytes - a)
                        stage[:m].copy_(text[a:a + m])
                        torch.cuda.synchronize()
                        f.write(memoryview(stage[:m].numpy()))
                runs = []
                for rep in range(4):
                    # (run 0 is a warm-up and is dropped: it backs ~90 GB of device memory this process has not touched.  A few
                    # seconds between runs: the driver scrubs a process's device memory after its exit, and a process that starts
                    # into that waits for it in its first allocations -- measured 2-3 s on some boxes, none on others)
                    try:
                        os.remove(fout)              # a fresh output file every run (replacing an existing 8 GB output costs the release
                    except OSError:                  # of its cached pages inside rename(): ~0.6 s that belong to the previous run)
                        pass
                    time.sleep(3.0)
                    tc = time.perf_counter()
                    p = subprocess.run([cli, fin, "-o", fout], capture_output=True, text=True)
                    wall = time.perf_counter() - tc
                    mt = re.search(r"grlbwt-timing: read\+upload ([\d.]+) s, build ([\d.]+) s, write ([\d.]+) s, total ([\d.]+) s", p.stdout)
                    if p.returncode != 0 or not mt:
                        runs.append({"failed": p.returncode, "stderr": p.stderr[-300:]})
                        break
                    runs.append({"wall_s": round(wall, 3), "read_upload_s": float(mt.group(1)), "build_s": float(mt.group(2)),
                                 "write_s": float(mt.group(3)), "MBps_wall": round(n_bytes / 1e6 / wall, 1)})
                hh = hashlib.md5()
                if os.path.exists(fout):
                    with open(fout, "rb") as f:
                        for blk in iter(lambda: f.read(1 << 26), b""):
                            hh.update(blk)
                if len(runs) > 1 and "wall_s" in runs[0]:
                    runs[0]["warm_up"] = True
                # (the MEDIAN of the timed runs is what is reported -- VERDICT r3: not the best of three)
                timed = sorted((r for r in runs if "wall_s" in r and not r.get("warm_up")), key=lambda r: r["wall_s"])
                best = timed[len(timed) // 2] if timed else None
                cli_e2e = {"command": "grlbwt_amd/bin/grlbwt FILE -o OUT (file in the page cache, output to %s)" % tmpdir, "runs": runs,
                           "output_md5": hh.hexdigest(), "md5_equals_hbm_image": None}
                if best:
                    cli_e2e.update(best)
            finally:
                for pth in (fin, fout):
                    try:
                        os.remove(pth)
                    except OSError:
                        pass
    if cli_e2e and image and cli_e2e.get("output_md5"):
        cli_e2e["md5_equals_hbm_image"] = cli_e2e["output_md5"] == image["md5"]

    # ---- roofline leg: one more step with HIP-event timing of every kernel on the engine's stream.
    # Every rank takes the step (for N > 1 it contains collectives); only rank 0 records and reports.
    if rank == 0:
        ctx.profile_enable(True)
        if comm is not None and os.environ.get("GRLBWT_BENCH_DETAIL"):
            comm.log = []
    step(parts=False)
    barrier()
    out = None
    if rank == 0:
        prof_detail = ctx.profile()
        host_syncs = prof_detail.pop("@host_sync", (0, 0.0, 0))[0]
        ctx.profile_enable(False)
        rounds = []
        while True:
            try:
                rounds.append(ctx.round_info(len(rounds)))
            except engine.GrlbwtError:
                break
        nr = len(rounds)
        levels = [ctx.level_info(l) for l in range(nr + 1)]
        cnt = ctx.counters()
        mem = ctx.memory_usage()

        # launch sites -> (site, phase, level); groups of SURVEY 8(d)
        def split(k):
            site, _, tag = k.partition("#")
            ph = tag[:1] if tag[:1].isalpha() else ""
            return site, ph
        grp = {gname: [0, 0.0] for gname in GROUP_SITES}
        sites = {}
        for k, (c, ms, nb) in prof_detail.items():
            site, ph = split(k)
            e = sites.setdefault(site, [0, 0.0, 0])
            e[0] += c; e[1] += ms; e[2] += nb
            for gname, pred in GROUP_SITES.items():
                if pred(site, ph):
                    grp[gname][0] += c
                    grp[gname][1] += ms
        ab_bytes = c_bytes = 0
        per_level = []
        if comm is None:
            for r in range(nr):
                ab, cb = induction_bytes(rounds, levels, r)
                ab_bytes += ab
                c_bytes += cb
                per_level.append({"level": r, "AB_bytes": ab, "C_bytes": cb})
        hash_b, emit_b = parse_bytes(rounds, 1)
        gbytes = {"induce_AB": ab_bytes, "induce_C": c_bytes, "hash_emit": hash_b + emit_b, "dict_stage": dict_bytes(rounds, levels)}
        total_kernel_ms = sum(ms for _, ms, _ in prof_detail.values())

        def group_obj(gname):
            launches, ms = grp[gname]
            nb = gbytes[gname]
            if not nb or ms <= 0:
                return None
            ach = nb / (ms * 1e-3) / 1e9
            tr = pmc_traffic(GROUP_KERNELS[gname])
            return {"bound": "hbm", "kernel": gname, "launches": launches, "achieved": round(ach, 3), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": tr,
                    "traffic_source": (pmc_traffic_source() + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, committed; not measured in this run)") if tr else None,
                    "traffic_stale": pmc_traffic_stale(GROUP_KERNELS[gname]) or None,
                    "algorithmic_bytes": nb, "kernel_ms_total": round(ms, 4),
                    "share_of_kernel_time": round(ms / max(total_kernel_ms, 1e-9), 4)}

        roofline = group_obj("induce_AB")
        if roofline:
            roofline["kernel"] = ("induction pass A+B (compute_hocc_size + infer_lvl_bwt scatter, exact_ind_phase.cpp:42-109,143-258): "
                                  "launch sites induce_*; bytes = SURVEY 8(d) formula in the reference's cell widths (DESIGN.md section 4)")
            roofline["per_launch"] = {"algorithmic_bytes": round(roofline["algorithmic_bytes"] / max(1, roofline["launches"])),
                                      "avg_launch_ms": round(roofline["kernel_ms_total"] / max(1, roofline["launches"]), 5)}
        groups = {gname: group_obj(gname) for gname in GROUP_SITES}
        if gbytes["induce_AB"] or gbytes["hash_emit"]:
            whole = (ab_bytes + c_bytes + hash_b + emit_b)
            groups["whole_job"] = {"algorithmic_bytes": whole, "ms": round(ms_per_step, 3),
                                   "achieved": round(whole / (ms_per_step * 1e-3) / 1e9, 3),
                                   "frac": round(whole / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
        # per-pass efficiency of the streaming kernels whose launch sites state their own read+write bytes
        pass_eff = []
        for site, (c, ms, nb) in sorted(sites.items(), key=lambda kv: -kv[1][1]):
            if nb and ms > 0:
                pass_eff.append({"site": site, "launches": c, "ms": round(ms, 4), "GBps": round(nb / (ms * 1e-3) / 1e9, 1),
                                 "frac": round(nb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)})
        top = [{"site": k, "launches": c, "ms": round(ms, 3)} for k, (c, ms, _) in
               sorted(sites.items(), key=lambda kv: -kv[1][1])[:16]]
        if os.environ.get("GRLBWT_BENCH_DETAIL"):
            if comm is not None and comm.log is not None:
                for kind, nb, sec in comm.log:
                    print("  collective %-10s %12d bytes %8.3f ms" % (kind, nb, sec * 1e3), file=sys.stderr)
            lim = 120 if os.environ.get("GRLBWT_BENCH_DETAIL") == "1" else None        # (any other value: every site)
            for k, (c, ms, nb) in sorted(prof_detail.items(), key=lambda kv: -kv[1][1])[:lim]:
                print("  %-32s %4d launches %9.3f ms" % (k, c, ms), file=sys.stderr)

    # ---- extra: the 101 MB configs[1] workload on one GPU (round-1's headline), a few steps
    extra = None
    if rank == 0 and world == 1 and not force_dist and not args.no_extra and not (args.workload == "uniform" and args.reads == 1000000):
        ctx.close()
        t1 = workloads.uniform_reads_torch(1000000, 100, seed=20260001, device=dev)
        torch.cuda.synchronize()
        c1 = engine.Context(local_rank, 0, lib)

        def step1():
            c1.attach_device(t1.data_ptr(), t1.numel(), 1, keepalive=t1)
            c1.build()
        for _ in range(2):
            step1()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            step1()
        torch.cuda.synchronize()
        d1 = (time.perf_counter() - t0) / 10
        c1.profile_enable(True)                  # one more build with per-launch events: kernel time vs wall, launches, host waits
        step1()
        p1 = c1.profile()
        syncs1 = p1.pop("@host_sync", (0, 0.0, 0))[0]
        c1.profile_enable(False)
        extra = {"workload": "1000000 x 100 bp uniform ACGT reads (101000000 bytes), BASELINE configs[1]",
                 "ms_per_step": round(d1 * 1e3, 3), "value_MBps": round(101.0 / d1, 1), "steps": 10,
                 "kernel_ms_total": round(sum(ms for _, ms, _ in p1.values()), 3), "kernel_launches": sum(c for c, _, _ in p1.values()),
                 "host_syncs": syncs1}
        c1.close()
        del t1

    # ---- CPU baseline leg (reported, not the target): the oracle ("port"), 1 core, bounded sample
    cpu = None
    if rank == 0 and not args.no_cpu_baseline and world == 1 and not force_dist:
        from oracle import oracle
        oracle.build()
        m = min(args.cpu_sample_reads, hi - lo)
        sample = text[: m * (args.read_len + 1)].cpu().numpy()
        with tempfile.TemporaryDirectory() as td:
            fi, fo = os.path.join(td, "in.txt"), os.path.join(td, "out.rl_bwt")
            sample.tofile(fi)
            tc = time.perf_counter()
            subprocess.check_call([oracle.CLI, fi, fo, "-q"])
            tcpu = time.perf_counter() - tc
        cpu = {"value": round(sample.size / 1e6 / tcpu, 3), "unit": "MB/s", "cores": 1, "kind": "port",
               "sample": "first %d reads (%d bytes) of the same workload, oracle/oracle_cli (CPU restatement of the path; the "
                         "reference binary needs SDSL-lite and cannot be built in this image), %.1f s" % (m, sample.size, tcpu),
               "host_cpus": os.cpu_count()}

    if rank == 0:
        if args.workload == "illumina":
            wl = ("%d x 150 bp Illumina-style reads (%d bytes) from a %d bp genome, 0.5%% substitutions; BASELINE configs[3]%s"
                  % (args.reads, total_bytes, args.genome, "" if args.reads == 66225166 else " shape (reduced read count)"))
        else:
            wl = ("%d x %d bp uniform ACGT reads (%d bytes), sigma=5, byte alphabet; BASELINE configs[1]%s"
                  % (args.reads, args.read_len, total_bytes, "" if (args.reads, args.read_len) == (1000000, 100) else " shape"))
        out = {
            "metric": "input MB/s building BCR BWT on 10 GB DNA reads, 1/2/4/8 MI355X", "value": round(value, 3), "unit": "MB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            # (scaling: ONE collection of fixed size -- strong scaling at N > 1; at N = 1 neither word applies)
            "higher_is_better": True, "scaling": "strong" if world > 1 else "single", "vs_baseline": None, "dtype": "u8",
            "data": "synthetic",
            # value = the build with the input resident in HBM (the contract of this bench); value_cli = the same bytes through the
            # grlbwt executable, file in the page cache -> .rl_bwt file closed (SURVEY 8(d)'s wording of the metric; median run)
            "value_cli": (cli_e2e or {}).get("MBps_wall"),
            "config": {"workload": wl, "input_resident": "HBM",
                       "output": (".rl_bwt image in HBM" if not keep_parts else
                                  ".rl_bwt image in HBM, in %d parts on the ranks that induced them (rank 0: bytes [%d, %d)); gathered after the "
                                  "timed region for the md5 / single-GPU comparison" % (world, parts_span[0], parts_span[0] + parts_span[1])),
                       "value_is": "HBM-resident build (value); value_cli = wall time of the CLI, file in -> .rl_bwt file closed, median of the timed runs",
                       "parallelism": ("1 GPU" if world == 1 else
                                       "the ONE collection sharded by record over %d GPUs (%d reads each): local LMS parsing/hashing/emission, "
                                       "hash-partitioned dictionary merge (all-to-all) + key-range-sharded dictionary stage per round, induction "
                                       "sharded by output piece (cells and BWT_{r+1} windows cross the fabric once per level), RCCL" % (world, hi - lo))},
            # what the TIMED steps leave behind: the whole image on one device (N = 1, and N > 1 with GRLBWT_BENCH_GATHER_IMAGE=1), or -- the
            # definition of the metric at N > 1 since round 5 -- the image in parts on the ranks that induced them (rounds 1-4 timed the
            # all-gather to every rank as well: SCALE records of r01-r04 and of r05 on are not comparable)
            "image_layout": "parts" if keep_parts else "whole",
            "roofline": roofline, "cpu_baseline": cpu, "image": image, "cli_end_to_end": cli_e2e,
            "roofline_groups": groups, "pass_efficiency": pass_eff[:12],
            "stage_seconds": {k: round(v, 5) for k, v in cnt.items() if k.startswith("t_")},
            "top_sites": top, "rounds": nr,
            "levels": [{k: lv[k] for k in ("n", "n_runs", "runs_next", "induced_cells", "chain_steps", "merged_cells", "prebwt_runs")}
                       for lv in levels],
            "round_shapes": [{k: rd[k] for k in ("n_in", "n_phrases", "dict_syms", "n_metasyms", "parse_size")} for rd in rounds],
            "induction_bytes_per_level": per_level,
            "kernel_launches": sum(c for c, _, _ in prof_detail.values()), "host_syncs": host_syncs,
            "kernel_ms_total": round(total_kernel_ms, 3),
            "device_memory": {"peak_live_bytes": mem["peak_live_bytes"], "reserved_bytes": mem["reserved_bytes"]},
            "idx_bytes": cnt["idx_bytes"], "extra_configs1_101MB": extra,
            # per-GPU view of the same run (the collection is fixed, so this is NOT a weak-scaling measurement: a rank's work
            # includes the replicated parts of the dictionary stage, which do not shrink with N)
            "per_gpu": {"shard_bytes_rank0": n_bytes, "value_per_gpu": round(value / world, 3)},
        }
        out["devices"] = devices
        out["distinct_devices"] = len({d.get("uuid") or d.get("pci_bus_id") or (d["local_rank"], d["hip_device"]) for d in devices})
        if comm is not None:     # totals over warmup + timed + profile steps on rank 0
            nsteps = max(1, args.warmup + args.steps)
            out["collectives_per_step"] = {"allgather": comm_timed[0] // nsteps, "alltoallv": comm_timed[1] // nsteps,
                                           "bytes": comm_timed[2] // nsteps, "ms_in_callbacks": round(comm_timed[3] / nsteps * 1e3, 3)}
            # the sharded result against a single-GPU build of the WHOLE collection on this rank (outside the timed region;
            # the other ranks wait at the final barrier): the .rl_bwt images must be the same bytes
            if os.environ.get("GRLBWT_BENCH_VERIFY", "1") != "0":
                try:
                    nb, _ = ctx.result_size()
                    mine = gdist._view(ctx.result_device_ptr(), nb, dev)
                    full = make_text(args.workload, args.reads, args.read_len, 0, args.reads)
                    torch.cuda.synchronize()          # (torch's stream wrote the text; the engine has its own)
                    with engine.Context(local_rank, flags, lib) as c2:
                        c2.attach_device(full.data_ptr(), full.numel(), 1, keepalive=full)
                        c2.build()
                        nb2, _ = c2.result_size()
                        ref = gdist._view(c2.result_device_ptr(), nb2, dev)
                        out["sharded_image_equals_single_gpu"] = bool(nb == nb2 and torch.equal(mine, ref))
                    del full
                except Exception as e:        # (memory on a smaller device, ...): reported, never fatal for the measurement
                    out["sharded_image_equals_single_gpu"] = "not checked: %r" % (e,)
    try:
        ctx.close()
    except Exception:
        pass
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        os.write(result_fd, (json.dumps(out) + "\n").encode())


if __name__ == "__main__":
    main()
