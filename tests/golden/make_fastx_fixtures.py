#!/usr/bin/env python3
"""Generates tests/golden/fastx/* and tests/golden/fastx_ref.json with the REFERENCE's own FASTA/Q converter.

Run in the build container (needs /root/reference): `python tests/golden/make_fastx_fixtures.py`.  oracle/Makefile's
`ref` target compiles oracle/_ref/fastx2plain from the reference's sources (external/bioparsers/lib/fastx_handler.cpp,
dna_string.cpp, external/cdt/lib/utils.cpp) behind oracle/ref_fastx_driver.cpp; this script writes small FASTA/FASTQ
inputs (plain and gzip), runs that program on each with and without reverse complements, and records what it produced:
md5 and size of the plain text, the number of strings, and for inputs it rejects ("The input seems not to be DNA") the
exit code and the message.  The committed inputs + expectations pin oracle/fastx_oracle.c and the device path."""
import gzip
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402

OUT_DIR = os.path.join(HERE, "fastx")


def fasta(rng, n, lo, hi, width, alphabet=b"ACGT", crlf=False, blank_every=0):
    nl = b"\r\n" if crlf else b"\n"
    parts = []
    for i in range(n):
        L = int(rng.integers(lo, hi + 1))
        seq = bytes(rng.choice(np.frombuffer(alphabet, dtype=np.uint8), size=L))
        parts.append(b">r%d some comment %d" % (i, L) + nl)
        for k in range(0, L, width):
            parts.append(seq[k:k + width] + nl)
        if blank_every and i % blank_every == 0:
            parts.append(nl)
    return b"".join(parts)


def fastq(rng, n, lo, hi, alphabet=b"ACGT", crlf=False, plus_name=False):
    nl = b"\r\n" if crlf else b"\n"
    parts = []
    for i in range(n):
        L = int(rng.integers(lo, hi + 1))
        seq = bytes(rng.choice(np.frombuffer(alphabet, dtype=np.uint8), size=L))
        qual = bytes(rng.integers(33, 74, size=L).astype(np.uint8))      # includes '@', '+' and '>' as quality characters
        parts += [b"@q%d/1 len=%d" % (i, L) + nl, seq + nl, (b"+q%d/1" % i if plus_name else b"+") + nl, qual + nl]
    return b"".join(parts)


def cases():
    rng = np.random.default_rng(20260003)
    yield "fa_single_line.fa", fasta(rng, 40, 1, 120, 1000)
    yield "fa_wrapped60.fa", fasta(rng, 25, 50, 400, 60)
    yield "fa_wrapped_blank_lines.fa", fasta(rng, 12, 0, 200, 70, blank_every=3)
    yield "fa_crlf.fa", fasta(rng, 10, 1, 150, 50, crlf=True, blank_every=4)
    yield "fa_with_N_lower.fa", fasta(rng, 10, 20, 80, 60, alphabet=b"ACGTNacgt")
    yield "fa_empty_records.fa", b">e1\n>e2\nACGT\n>e3\n\n>e4\nTT\nGG\n"
    yield "fa_no_final_newline.fa", b">x y z\nACGTAC\nGGT"
    yield "fa_header_only.fa", b">only"
    yield "fa_at_headers.fa", b"@r1\nACGT\nAC\n@r2\nGG\n>r3\nTTA\n"
    yield "fq_regular.fq", fastq(rng, 60, 30, 150)
    yield "fq_plus_names.fq", fastq(rng, 20, 10, 100, plus_name=True)
    yield "fq_crlf.fq", fastq(rng, 12, 10, 60, crlf=True)
    yield "fq_with_N.fq", fastq(rng, 15, 20, 60, alphabet=b"ACGTN")
    yield "fq_empty_read.fq", b"@a\nACGT\n+\nIIII\n@b\n\n+\n\n@c\nGG\n+\n@@\n"
    yield "fq_multiline.fq", b"@m1\nACGT\nTTGA\n+\nIIII\nJJJJ\n@m2\nGGC\n+m2\n@+>\n"
    yield "fq_truncated_quality.fq", b"@t1\nACGT\n+\nIIII\n@t2\nACGTAC\n+\nIII\n"
    yield "fq_trailing_blank.fq", fastq(rng, 5, 10, 30) + b"\n\n"
    yield "fa_gz.fa.gz", gzip.compress(fasta(rng, 30, 10, 300, 80), mtime=0)
    yield "fq_gz.fq.gz", gzip.compress(fastq(rng, 50, 50, 150), mtime=0)
    yield "fq_gz_two_members.fq.gz", gzip.compress(fastq(rng, 10, 20, 40), mtime=0) + gzip.compress(fastq(rng, 10, 20, 40), mtime=0)


def run_ref(prog, path, rc):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "plain")
        p = subprocess.run([prog, path, out, "1" if rc else "0"], capture_output=True, text=True)
        info = {"exit": p.returncode}
        for line in p.stdout.splitlines():
            k, _, v = line.partition(" ")
            if k in ("is_fastx", "n_strings"):
                info[k] = int(v)
        if p.returncode == 0:
            blob = open(out, "rb").read()
            info["size"] = len(blob)
            info["md5"] = hashlib.md5(blob).hexdigest()
        else:
            info["stderr"] = p.stderr.strip()
        return info


def main():
    assert oracle.build_ref(), "needs /root/reference (build container)"
    prog = oracle.ref_prog("fastx2plain")
    os.makedirs(OUT_DIR, exist_ok=True)
    table = []
    for name, data in cases():
        path = os.path.join(OUT_DIR, name)
        with open(path, "wb") as f:
            f.write(data)
        table.append({"name": name, "input_md5": hashlib.md5(data).hexdigest(), "plain": run_ref(prog, path, False), "revcomp": run_ref(prog, path, True)})
    with open(os.path.join(HERE, "fastx_ref.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_fastx_fixtures.py", "program": "oracle/_ref/fastx2plain (reference sources, oracle/Makefile ref)",
                   "cases": table}, f, indent=1)
    print("wrote %d cases" % len(table))


if __name__ == "__main__":
    main()
