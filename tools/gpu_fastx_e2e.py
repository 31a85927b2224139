#!/usr/bin/env python3
"""Lease script: f3 ingestion at size.  Writes an Illumina-style FASTQ (150 bp reads, the bench workload's distribution) to
/tmp, then times (a) the library's loader with the per-kernel profile, (b) the CLI end to end (FASTQ file -> .rl_bwt file),
and checks the result against the build of the same reads given as plain text.  usage: gpu_fastx_e2e.py <reads> <genome> [gz]"""
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    import __graft_entry__ as g
    from grlbwt_amd import engine, workloads
    reads = int(sys.argv[1]) if len(sys.argv) > 1 else 6622517
    genome = int(sys.argv[2]) if len(sys.argv) > 2 else 33000000
    gz = len(sys.argv) > 3 and sys.argv[3] == "gz"
    lib, cli = g.build_hip(), g.build_cli()
    dev = torch.device("cuda", 0)
    L = 150
    text = workloads.sampled_reads_torch(reads, L, genome, seed=20260003, device=dev)
    rows = text.view(reads, L + 1)
    hdr = torch.tensor(list(b"@SRR000000.1 1/1\n"), dtype=torch.uint8, device=dev).expand(reads, -1)
    plus = torch.tensor(list(b"+\n"), dtype=torch.uint8, device=dev).expand(reads, -1)
    qual = torch.full((reads, L), ord("I"), dtype=torch.uint8, device=dev)
    nl = torch.full((reads, 1), 10, dtype=torch.uint8, device=dev)
    fq = torch.cat([hdr, rows, plus, qual, nl], dim=1).reshape(-1)
    path = "/tmp/reads.fq"
    t0 = time.time()
    blob = fq.cpu().numpy()
    with open(path, "wb") as f:
        f.write(memoryview(blob))
    if gz:
        subprocess.check_call(["gzip", "-1", "-f", path])
        path += ".gz"
    fsize = os.path.getsize(path)
    print("wrote %s: %d bytes in %.1f s" % (path, fsize, time.time() - t0), file=sys.stderr)
    del fq, blob, hdr, plus, qual
    out = {"reads": reads, "fastq_bytes": fsize, "gz": gz, "text_bytes": int(text.numel())}
    # (a) library loader with the kernel profile
    with engine.Context(0, 0, lib) as ctx:
        ctx.load_fastx(path, False)           # warm: page cache, pool
        ctx.profile_enable(True)
        t0 = time.time()
        ns = ctx.load_fastx(path, False)
        torch.cuda.synchronize()
        out["load_fastx_s"] = round(time.time() - t0, 4)
        prof = ctx.profile()
        out["fastx_kernels_ms"] = {k: round(v[1], 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1]) if k.startswith("fastx")}
        out["fastx_kernel_ms_total"] = round(sum(v[1] for k, v in prof.items() if k.startswith("fastx")), 3)
        ctx.profile_enable(False)
        assert ns == reads
        ctx.build()
        md5_fx = hashlib.md5(ctx.result_bytes()).hexdigest() if reads <= 8000000 else None
    with engine.Context(0, 0, lib) as ctx:
        ctx.attach_device(text.data_ptr(), text.numel(), 1, keepalive=text)
        ctx.build()
        md5_plain = hashlib.md5(ctx.result_bytes()).hexdigest() if reads <= 8000000 else None
    out["image_md5_equal"] = (md5_fx == md5_plain) if md5_fx else None
    del text
    torch.cuda.empty_cache()
    # (b) CLI end to end, file in page cache
    for rep in range(2):
        # (--fastx: the conversion is opt-in, as in the reference; without it the file would be read as one-string-per-line cells)
        p = subprocess.run([cli, path, "--fastx", "-o", "/tmp/fx_out"], capture_output=True, text=True)
        line = [l for l in p.stdout.splitlines() if l.startswith("grlbwt-timing:")]
        out["cli_rc"] = p.returncode
        assert p.returncode == 0, p.stderr[-500:]
        assert "The input is in FASTA/Q format" in p.stdout, p.stdout[-500:]
        out["cli_timing_%d" % rep] = line[0] if line else p.stderr[-300:]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "fastx_e2e_%d%s.json" % (reads, "_gz" if gz else "")), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
