"""The .rl_bwt consumers of the reference's scripts/ (grl2plain, grlbwt2rle, bwt_stats) against a numpy
restatement of what those scripts compute.  CPU: over the serial stand-in; -m gpu: the HIP kernels on device memory."""
import math
import os
import subprocess

import numpy as np
import pytest

from grlbwt_amd import engine, workloads
from tests import bcr_check as bc

HERE = os.path.dirname(os.path.abspath(__file__))


def expected(blob):
    """What scripts/grl2plain.cpp, grlbwt2rle.cpp and bwt_stats.cpp produce for this image."""
    sb, fb, sym, ln = bc.parse_rl_bwt(blob)
    sym = np.asarray(sym, dtype=np.uint64)
    ln = np.asarray(ln, dtype=np.uint64)
    r = len(sym)
    st = {"n_runs": r, "text_size": int(ln.sum()), "min_run": int(ln.min()), "max_run": int(ln.max()),
          "fit1": int((ln <= 255).sum()), "fit2": int(((ln > 255) & (ln <= 65535)).sum())}
    st["fit3"] = r - st["fit1"] - st["fit2"]
    runs_of = np.bincount(sym.astype(np.int64), minlength=256)
    freq_of = np.bincount(sym.astype(np.int64), weights=ln.astype(np.float64), minlength=256).astype(np.uint64)
    st["runs_of"] = [int(x) for x in runs_of[:256]]
    st["freq_of"] = [int(x) for x in freq_of[:256]]
    st["sigma"] = int(sum(1 for x in runs_of[:256] if (int(x) & 0xFF) != 0))     # bwt_stats.cpp:57-62 iterates as unsigned char
    srt = np.sort(ln.astype(np.uint32))
    prop, dec = 0.1, []
    for _ in range(9):                                                            # bwt_stats.cpp:83-89
        q = min(int(math.ceil(float(r) * prop)), r - 1)
        dec.append(int(srt[q]))
        prop += 0.1
    st["deciles"] = dec
    st["non_maximal"] = int((sym[1:] == sym[:-1]).sum())
    return sym, ln, st


def run_consumers(lib, data, on_gpu):
    if on_gpu:
        import torch
        torch.zeros(1, device="cuda:0")      # torch's HIP runtime has to be the one the process initialises first
    with engine.Context(0, 0, lib) as ctx:
        ctx.upload(data, 1)
        ctx.build()
        blob = ctx.result_bytes()
        nb, nr = ctx.result_size()
        img = ctx.result_device_ptr()
        sym, ln, st = expected(blob)
        n = st["text_size"]
        if on_gpu:
            plain = torch.zeros(n, dtype=torch.uint8, device="cuda:0")
            syms = torch.zeros(nr, dtype=torch.uint8, device="cuda:0")
            lens = torch.zeros(nr, dtype=torch.int32, device="cuda:0")
            pp, sp, lp = plain.data_ptr(), syms.data_ptr(), lens.data_ptr()
            get = lambda t, dt: t.cpu().numpy().view(dt)
        else:
            plain = np.zeros(n, dtype=np.uint8); syms = np.zeros(nr, dtype=np.uint8); lens = np.zeros(nr, dtype=np.uint32)
            pp, sp, lp = plain.ctypes.data, syms.ctypes.data, lens.ctypes.data
            get = lambda t, dt: t.view(dt)
        assert ctx.image_plain(img, nb, pp, n) == n
        assert np.array_equal(get(plain, np.uint8), np.repeat(sym.astype(np.uint8), ln.astype(np.int64)))
        assert ctx.image_plain(img, nb, pp, n, null_char=ord("#")) == n      # no symbol 0 in these inputs: same bytes
        assert np.array_equal(get(plain, np.uint8), np.repeat(sym.astype(np.uint8), ln.astype(np.int64)))
        assert ctx.image_rle(img, nb, sp, lp, nr) == nr
        assert np.array_equal(get(syms, np.uint8), sym.astype(np.uint8)) and np.array_equal(get(lens, np.uint32), ln.astype(np.uint32))
        got = ctx.image_stats(img, nb)
        for k, v in st.items():
            assert got[k] == v, (k, got[k], v)
        with pytest.raises(engine.GrlbwtError):
            ctx.image_plain(img, nb, pp, n - 1)                              # output buffer too small


CASES = {
    "reads": lambda: workloads.sampled_reads(4000, 100, 30000, seed=3).tobytes(),
    "repetitive": lambda: workloads.repetitive_copies(20, 6000, seed=9).tobytes(),
    "runs": lambda: (b"A" * 70000 + b"C" * 300 + b"\n" + b"ACGT" * 500 + b"\n") * 3,
    "tiny": lambda: b"A\n\nA\n",
}


@pytest.fixture(scope="module")
def sim():
    d = os.path.join(HERE, "hostsim")
    subprocess.check_call(["make", "-s", "-C", d], stdout=subprocess.DEVNULL)
    os.environ["GRLBWT_ALLOW_TEST_STANDIN"] = "1"
    return os.path.join(d, "_build", "libgrlbwt_sim.so")


@pytest.mark.parametrize("case", sorted(CASES))
def test_consumers_logic_on_stand_in(sim, case):
    run_consumers(sim, CASES[case](), on_gpu=False)


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(CASES))
def test_consumers_hip(case):
    import __graft_entry__ as g
    run_consumers(g.build_hip(), CASES[case](), on_gpu=True)


@pytest.mark.gpu
def test_consumers_hip_golden_file():
    import __graft_entry__ as g
    from tests import parity
    run_consumers(g.build_hip(), open(os.path.join(parity.GOLD, "test_byte_alphabet.txt"), "rb").read(), on_gpu=True)
