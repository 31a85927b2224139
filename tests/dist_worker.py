"""Worker for the world_size>1 tests (gloo on CPU over the serial test stand-in of the
device primitives; nccl/RCCL with the HIP library on a GPU box)."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    from grlbwt_amd import dist as gdist
    from grlbwt_amd import engine, workloads
    lib, backend, case, out_dir = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if backend == "gloo":               # the CPU tests hand this worker the serial stand-in (tests/hostsim)
        engine._test_allow_standin(lib)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl", device_id=torch.device("cuda", torch.cuda.current_device()))
        device = "cuda:%d" % torch.cuda.current_device()
    elif backend == "gloo-cuda":       # two ranks sharing one GPU, gloo transport staged through the host
        dist.init_process_group("gloo")
        device = "cuda:0"
    else:
        dist.init_process_group("gloo")
        device = "cpu"
    w = 1
    if case.startswith("illumina_dev:"):
        # record shard of ONE Illumina-style collection generated on the device (bench.py's N > 1 workload shape):
        # illumina_dev:<reads>:<genome>; every rank writes the md5 of the collection's .rl_bwt it computed
        _, reads, genome = case.split(":")
        reads, genome = int(reads), int(genome)
        lo, hi = reads * rank // world, reads * (rank + 1) // world
        dev = torch.device(device)
        text = workloads.sampled_reads_torch(reads, 150, genome, seed=20260003, device=dev, read_lo=lo, read_hi=hi)
        torch.cuda.synchronize()
        comm = gdist.Communicator(dev)
        flags = engine.FLAG_FORCE_IDX64 if reads * 151 >= 0xFFFFFF00 else 0
        with engine.Context(dev.index or 0, flags | gdist.pool_flags(comm), lib) as ctx:
            ctx.attach_device(text.data_ptr(), text.numel(), 1, keepalive=text)
            gdist.dist_build(ctx, comm)
            out = ctx.result_bytes()
        with open(os.path.join(out_dir, "illumina_dev.rank%d.md5" % rank), "w") as f:
            f.write("%s %d" % (hashlib.md5(out).hexdigest(), len(out)))
        dist.barrier()
        dist.destroy_process_group()
        return
    if case.startswith("gaps_dev:"):
        # gaps_dev:<Mbp per string>:<gap Mbp>: every rank holds two strings with a long N gap each (lease script: the sharded flow on
        # very long phrases); rank 0 also builds the whole collection alone and compares
        import time
        _, mbp, gap = case.split(":")
        mbp, gap = int(mbp), int(gap)
        dev = torch.device(device)

        def string(k):
            rng = np.random.default_rng(100 + k)
            s_ = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=mbp * 1000000 + 31 * k)
            s_[len(s_) // 3: len(s_) // 3 + gap * 1000000 + 7 * k] = ord("N")
            return np.concatenate([s_, np.array([10], dtype=np.uint8)])
        mine = np.concatenate([string(2 * rank), string(2 * rank + 1)])
        text = torch.from_numpy(mine).to(dev)
        comm = gdist.Communicator(dev)
        with engine.Context(dev.index or 0, gdist.pool_flags(comm), lib) as ctx:
            for rep in range(2):
                ctx.profile_enable(rep == 1)
                torch.cuda.synchronize()
                t0 = time.time()
                ctx.attach_device(text.data_ptr(), text.numel(), 1, keepalive=text)
                gdist.dist_build(ctx, comm)
                torch.cuda.synchronize()
                dt = time.time() - t0
            out = ctx.result_bytes()
            if rank == 0:
                for k, (c, ms, nb) in sorted(ctx.profile().items(), key=lambda kv: -kv[1][1])[:8]:
                    print("  %-32s %4d %10.2f ms" % (k, c, ms), flush=True)
        msg = "rank %d: sharded build %.2f s, image md5 %s" % (rank, dt, hashlib.md5(out).hexdigest())
        if rank == 0:
            whole = torch.from_numpy(np.concatenate([string(k) for k in range(2 * world)])).to(dev)
            with engine.Context(dev.index or 0, 0, lib) as ctx:
                for rep in range(2):
                    t0 = time.time()
                    ctx.attach_device(whole.data_ptr(), whole.numel(), 1, keepalive=whole)
                    ctx.build()
                    torch.cuda.synchronize()
                    d1 = time.time() - t0
                msg += "; single-GPU build %.2f s, md5 %s" % (d1, hashlib.md5(ctx.result_bytes()).hexdigest())
        print(msg, flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return
    if case == "illformed":
        # rank 1 holds a shard that does not end with the separator: EVERY rank must raise (no rank may be left waiting
        # in the first collective of the build)
        good = workloads.sampled_reads(400, 50, 3000, seed=3).tobytes()
        shard = good if rank != 1 else good[:-1]
        try:
            gdist.grl_bwt_algo_sharded(shard, 1, device, lib, 0)
            verdict = "returned"
        except engine.GrlbwtError as e:
            verdict = "raised %d" % e.code
        with open(os.path.join(out_dir, "illformed.rank%d" % rank), "w") as f:
            f.write(verdict)
        dist.barrier()
        dist.destroy_process_group()
        return
    if case == "injected":
        # one rank fails inside its parsing round (a failure only that shard can have): every rank must raise
        good = workloads.sampled_reads(400, 50, 3000, seed=3)
        shard = gdist.shard_records(good, rank, world)
        try:
            gdist.grl_bwt_algo_sharded(shard.tobytes(), 1, device, lib, 0)
            verdict = "returned"
        except engine.GrlbwtError as e:
            verdict = "raised %d" % e.code
        with open(os.path.join(out_dir, "injected.rank%d" % rank), "w") as f:
            f.write(verdict)
        dist.barrier()
        dist.destroy_process_group()
        return
    if case == "reads":
        data = workloads.sampled_reads(3001, 100, 20000, seed=11)
    elif case == "uniform":
        data = workloads.uniform_reads(2000, 100, seed=5)
    elif case == "repetitive":
        data = workloads.repetitive_copies(31, 8000, seed=3)
    elif case == "tokens":
        data = workloads.zipf_tokens(30000, doc_len=100, vocab=3000)
        w = 2
    elif case == "reads_big":
        data = workloads.sampled_reads(60001, 100, 400000, seed=13)
    elif case == "longruns":       # long equal runs and long phrases: many refinement passes, empty key ranges
        parts = [b"A" * 3000 + b"C" * 2000 + b"\n", b"ACGT" * 900 + b"\n", b"\n" * 40, b"T" * 5000 + b"\n",
                 b"A" * 3000 + b"C" * 2000 + b"\n", b"G" * 100 + b"ACGT" * 50 + b"\n"] * 3
        data = np.frombuffer(b"".join(parts), dtype=np.uint8)
    elif case == "samechar":       # two runs in the whole BWT: slices that are a single run continuing the previous slice's run
        data = np.frombuffer(b"A\n" * 12, dtype=np.uint8)
    elif case == "tiny":
        data = np.frombuffer(b"A\n\nA\nGATTACA\nGATTACA\nT\n", dtype=np.uint8)
    else:
        raise SystemExit("unknown case " + case)
    shard = gdist.shard_records(data, rank, world)
    flags = engine.FLAG_FORCE_IDX64 if case == "repetitive" else 0
    comms = []
    if os.environ.get("GRLBWT_TEST_KEEP_PARTS"):
        # the image stays in parts: every rank writes its part at its offset of ONE file (grlbwt_result_write_part); the parts
        # must tile the image
        path = os.path.join(out_dir, case + ".rl_bwt")
        if rank == 0 and os.path.exists(path):
            os.remove(path)
        dist.barrier()
        off, part, total = gdist.grl_bwt_algo_sharded(shard.tobytes(), w, device, lib, flags, comm_out=comms, keep_parts=True, part_file=path)
        spans = [None] * world
        dist.all_gather_object(spans, (off, len(part)))
        if rank == 0:
            pos = 0
            for o, n in spans:
                assert o == pos or n == 0, spans
                pos += n
            assert pos == total and os.path.getsize(path) == total, (spans, total)
            with open(os.path.join(out_dir, case + ".input"), "wb") as f:
                f.write(data.tobytes())
            with open(os.path.join(out_dir, case + ".a2a"), "w") as f:
                f.write("%d %d" % (comms[0].n_allgather, comms[0].bytes_moved))
        dist.barrier()
        dist.destroy_process_group()
        return
    out = gdist.grl_bwt_algo_sharded(shard.tobytes(), w, device, lib, flags, comm_out=comms)
    with open(os.path.join(out_dir, "%s.rank%d.md5" % (case, rank)), "w") as f:
        f.write(hashlib.md5(out).hexdigest())
    with open(os.path.join(out_dir, "%s.rank%d.comm" % (case, rank)), "w") as f:
        f.write("%d %d %d" % (comms[0].n_allgather, comms[0].n_alltoall, comms[0].bytes_moved))
    if rank == 0:
        with open(os.path.join(out_dir, case + ".rl_bwt"), "wb") as f:
            f.write(out)
        with open(os.path.join(out_dir, case + ".input"), "wb") as f:
            f.write(data.tobytes())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
