"""Deterministic synthetic inputs: splitmix64 streams.

The same bytes are produced on every host so CPU-baseline and GPU runs see
identical inputs.  numpy-vectorised.
"""
import numpy as np

_M64 = (1 << 64) - 1


def splitmix64(seed, n):
    """n successive splitmix64 outputs for `seed` (uint64 array)."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, n + 1, dtype=np.uint64)
        s = np.uint64(seed & _M64) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = s
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def uniform_reads(n_reads, read_len, seed=20260001):
    """config 2: n_reads x read_len i.i.d. uniform ACGT, '\\n' terminated."""
    z = splitmix64(seed, n_reads * read_len)
    bases = _ACGT[(z >> np.uint64(62)).astype(np.int64)].reshape(n_reads, read_len)
    out = np.empty((n_reads, read_len + 1), dtype=np.uint8)
    out[:, :read_len] = bases
    out[:, read_len] = 10
    return out.reshape(-1)


def genome(length, seed):
    z = splitmix64(seed, length)
    return _ACGT[(z >> np.uint64(62)).astype(np.int64)]


def sampled_reads(n_reads, read_len, genome_len, seed=20260003, err=0.005):
    """config 4 style: reads sampled (forward strand) from a random genome with
    substitution errors, '\\n' terminated."""
    g = genome(genome_len, seed)
    z = splitmix64(seed + 1, n_reads)
    starts = (z % np.uint64(genome_len - read_len)).astype(np.int64)
    idx = starts[:, None] + np.arange(read_len, dtype=np.int64)[None, :]
    bases = g[idx]
    if err > 0:
        e = splitmix64(seed + 2, n_reads * read_len).reshape(n_reads, read_len)
        hit = (e >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53)) < err
        shift = ((e & np.uint64(3)) % np.uint64(3) + np.uint64(1)).astype(np.int64)
        code = np.zeros(256, dtype=np.int64)
        code[_ACGT] = np.arange(4)
        sub = _ACGT[(code[bases] + shift) % 4]
        bases = np.where(hit, sub, bases)
    out = np.empty((n_reads, read_len + 1), dtype=np.uint8)
    out[:, :read_len] = bases
    out[:, read_len] = 10
    return out.reshape(-1)


def repetitive_copies(n_copies, length, seed=20260002, rate=1e-3):
    """config 3 style: n_copies of a pseudo-chromosome, copy k with i.i.d.
    substitutions at `rate`, one copy per line."""
    g = genome(length, seed)
    code = np.zeros(256, dtype=np.int64)
    code[_ACGT] = np.arange(4)
    out = np.empty((n_copies, length + 1), dtype=np.uint8)
    for k in range(n_copies):
        e = splitmix64(seed + 1 + k, length)
        hit = (e >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53)) < rate
        shift = ((e & np.uint64(3)) % np.uint64(3) + np.uint64(1)).astype(np.int64)
        out[k, :length] = np.where(hit, _ACGT[(code[g] + shift) % 4], g)
        out[k, length] = 10
    return out.reshape(-1)


def zipf_tokens(n_cells, doc_len=1000, vocab=65000, s=1.1, seed=20260005):
    """config 5 style: uint16 documents of doc_len tokens ~ Zipf(s) over [1, vocab],
    separator 0 after each document."""
    n_docs = n_cells // (doc_len + 1)
    w = np.arange(1, vocab + 1, dtype=np.float64) ** (-s)
    cdf = np.cumsum(w)
    cdf /= cdf[-1]
    z = splitmix64(seed, n_docs * doc_len)
    u = (z >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
    tok = (np.searchsorted(cdf, u) + 1).astype(np.uint16).reshape(n_docs, doc_len)
    out = np.zeros((n_docs, doc_len + 1), dtype=np.uint16)
    out[:, :doc_len] = tok
    return out.reshape(-1)


def uniform_reads_torch(n_reads, read_len, seed=20260001, device="cuda", chunk_reads=4_000_000):
    """Same bytes as uniform_reads(), generated on `device` with torch (for multi-GB inputs that
    should never exist on the host).  int64 arithmetic wraps like uint64; shifts are masked to be logical."""
    import torch
    out = torch.empty(n_reads * (read_len + 1), dtype=torch.uint8, device=device)
    view = out.view(n_reads, read_len + 1)
    acgt = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=device)
    golden = -7046029254386353131          # 0x9E3779B97F4A7C15 as int64
    m1 = -4658895280553007687              # 0xBF58476D1CE4E5B9
    m2 = -7723592293110705685              # 0x94D049BB133111EB
    s0 = seed if seed < (1 << 63) else seed - (1 << 64)
    for r0 in range(0, n_reads, chunk_reads):
        r1 = min(n_reads, r0 + chunk_reads)
        idx = torch.arange(r0 * read_len + 1, r1 * read_len + 1, dtype=torch.int64, device=device)
        z = s0 + idx * golden
        z = (z ^ ((z >> 30) & ((1 << 34) - 1))) * m1
        z = (z ^ ((z >> 27) & ((1 << 37) - 1))) * m2
        z = z ^ ((z >> 31) & ((1 << 33) - 1))
        base = acgt[((z >> 62) & 3)]
        view[r0:r1, :read_len] = base.view(r1 - r0, read_len)
        view[r0:r1, read_len] = 10
        del idx, z, base
    return out


def _splitmix64_torch(seed, i0, i1, device):
    """splitmix64 outputs number i0+1 .. i1 for `seed` as int64 (bit pattern of the uint64 value)."""
    import torch
    golden, m1, m2 = -7046029254386353131, -4658895280553007687, -7723592293110705685
    s0 = seed if seed < (1 << 63) else seed - (1 << 64)
    idx = torch.arange(i0 + 1, i1 + 1, dtype=torch.int64, device=device)
    z = s0 + idx * golden
    z = (z ^ ((z >> 30) & ((1 << 34) - 1))) * m1
    z = (z ^ ((z >> 27) & ((1 << 37) - 1))) * m2
    return z ^ ((z >> 31) & ((1 << 33) - 1))


def sampled_reads_torch(n_reads, read_len, genome_len, seed=20260003, err=0.005, device="cuda", chunk_reads=2_000_000,
                        read_lo=0, read_hi=None):
    """Same bytes as sampled_reads() (config 4: Illumina-style reads from a random genome, substitution errors),
    generated on `device`.  read_lo/read_hi select the contiguous record range [read_lo, read_hi) of the n_reads-read
    collection (a record shard of the SAME collection: the bytes equal that slice of the full output)."""
    import torch
    read_hi = n_reads if read_hi is None else read_hi
    acgt = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=device)
    g = torch.empty(genome_len, dtype=torch.uint8, device=device)
    for a in range(0, genome_len, 1 << 26):
        b = min(genome_len, a + (1 << 26))
        g[a:b] = acgt[(_splitmix64_torch(seed, a, b, device) >> 62) & 3]
    code = torch.zeros(256, dtype=torch.int64, device=device)
    code[acgt.long()] = torch.arange(4, device=device)
    out = torch.empty((read_hi - read_lo) * (read_len + 1), dtype=torch.uint8, device=device)
    view = out.view(read_hi - read_lo, read_len + 1)
    ar = torch.arange(read_len, dtype=torch.int64, device=device)
    span = genome_len - read_len
    for r0 in range(read_lo, read_hi, chunk_reads):
        r1 = min(read_hi, r0 + chunk_reads)
        z = _splitmix64_torch(seed + 1, r0, r1, device)
        # uint64 modulo on int64 bit patterns: split off the sign bit
        hi = (z >> 63) & 1
        lo = z & ((1 << 63) - 1)
        starts = (lo % span + hi * (pow(2, 63, span))) % span
        bases = g[starts[:, None] + ar[None, :]]
        if err > 0:
            e = _splitmix64_torch(seed + 2, r0 * read_len, r1 * read_len, device).view(r1 - r0, read_len)
            u = ((e >> 11) & ((1 << 53) - 1)).to(torch.float64) * (1.0 / (1 << 53))
            hit = u < err
            shift = (e & 3) % 3 + 1
            sub = acgt[(code[bases.long()] + shift) % 4]
            bases = torch.where(hit, sub, bases)
            del e, u, hit, shift, sub
        view[r0 - read_lo:r1 - read_lo, :read_len] = bases
        view[r0 - read_lo:r1 - read_lo, read_len] = 10
        del z, hi, lo, starts, bases
    return out


def repetitive_copies_torch(n_copies, length, seed=20260002, rate=1e-3, device="cuda"):
    """Same bytes as repetitive_copies() (config 3: n_copies of a pseudo-chromosome with substitutions), on `device`."""
    import torch
    acgt = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=device)
    g = torch.empty(length, dtype=torch.uint8, device=device)
    step = 1 << 26
    for a in range(0, length, step):
        b = min(length, a + step)
        g[a:b] = acgt[(_splitmix64_torch(seed, a, b, device) >> 62) & 3]
    code = torch.zeros(256, dtype=torch.int64, device=device)
    code[acgt.long()] = torch.arange(4, device=device)
    out = torch.empty(n_copies * (length + 1), dtype=torch.uint8, device=device)
    view = out.view(n_copies, length + 1)
    for k in range(n_copies):
        for a in range(0, length, step):
            b = min(length, a + step)
            e = _splitmix64_torch(seed + 1 + k, a, b, device)
            hit = ((e >> 11) & ((1 << 53) - 1)).to(torch.float64) * (1.0 / (1 << 53)) < rate
            shift = (e & 3) % 3 + 1
            gs = g[a:b]
            view[k, a:b] = torch.where(hit, acgt[(code[gs.long()] + shift) % 4], gs)
            del e, hit, shift
        view[k, length] = 10
    return out


def zipf_tokens_torch(n_cells, doc_len=1000, vocab=65000, s=1.1, seed=20260005, device="cuda", chunk_docs=20000):
    """Same cells as zipf_tokens() (config 5: uint16 Zipf documents, separator 0), as an int16 tensor on `device`
    (torch has no uint16 arithmetic; the bit patterns are the uint16 values)."""
    import torch
    n_docs = n_cells // (doc_len + 1)
    w = np.arange(1, vocab + 1, dtype=np.float64) ** (-s)
    cdf = np.cumsum(w)
    cdf /= cdf[-1]
    cdf_t = torch.from_numpy(cdf).to(device)
    out = torch.zeros(n_docs * (doc_len + 1), dtype=torch.int16, device=device)
    view = out.view(n_docs, doc_len + 1)
    for d0 in range(0, n_docs, chunk_docs):
        d1 = min(n_docs, d0 + chunk_docs)
        z = _splitmix64_torch(seed, d0 * doc_len, d1 * doc_len, device)
        u = ((z >> 11) & ((1 << 53) - 1)).to(torch.float64) * (1.0 / (1 << 53))
        tok = torch.searchsorted(cdf_t, u) + 1                     # np.searchsorted(side="left") semantics
        view[d0:d1, :doc_len] = tok.view(d1 - d0, doc_len).to(torch.int16)     # wraps to the uint16 bit pattern
        del z, u, tok
    return out


def md5_device(t, chunk=1 << 28):
    """md5 of a uint8 device tensor, downloaded through one pinned staging buffer (bench.py prints it for the image it
    times; tests/test_gpu_parity.py compares the 10 GB headline image with the committed value)."""
    import hashlib
    import torch
    h = hashlib.md5()
    n = t.numel()
    stage = torch.empty(min(chunk, max(n, 1)), dtype=torch.uint8, pin_memory=True)
    for o in range(0, n, chunk):
        m = min(chunk, n - o)
        stage[:m].copy_(t[o:o + m])
        torch.cuda.synchronize()
        h.update(stage[:m].numpy().tobytes() if m < 1 << 20 else memoryview(stage[:m].numpy()))
    return h.hexdigest()
