"""bench.py's host side (no GPU): the --gpus / WORLD_SIZE contract and the SURVEY 8(d) byte accounting."""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_world_size_mismatch_is_refused():
    """A run whose launcher started another number of ranks than --gpus says must not print a mislabelled line."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env, timeout=120)
    assert p.returncode == 2 and "refusing to mislabel" in p.stderr and p.stdout.strip() == ""


def test_survey_8d_byte_formulas():
    b = load_bench()
    assert (b.cdiv8(0), b.cdiv8(8), b.cdiv8(9), b.bitlen(255), b.bitlen(256)) == (1, 1, 2, 8, 9)
    # two levels of a DNA-like build: level 0 (sigma 85, 20,000 metasymbols), level 1 above it
    rounds = [{"sigma": 85, "n_metasyms": 20000, "n_in": 1000, "parse_size": 300}, {"sigma": 20000, "n_metasyms": 500, "n_in": 300, "parse_size": 90}]
    levels = [{"n": 1000, "n_runs": 700, "runs_next": 250, "induced_cells": 900, "merged_cells": 600, "chain_steps": 650, "prebwt_runs": 40},
              {"n": 300, "n_runs": 250, "runs_next": 80, "induced_cells": 200, "merged_cells": 0, "chain_steps": 0, "prebwt_runs": 30},
              {"n": 90, "n_runs": 80, "runs_next": 0, "induced_cells": 0, "merged_cells": 0, "chain_steps": 0, "prebwt_runs": 0}]
    ab, c = b.induction_bytes(rounds, levels, 0)
    # sbN = 2 (20000 metasymbols), fbN = 2 (n_1 = 300), hocc cell = 1 + 1, grammar cell = 2 (88 + 20000 + 1)
    assert ab == 250 * (2 + 2) + 250 * 2 + 600 * 2 + 2 * 2 * 650
    # pre-BWT and BWT_0 records: 1-byte symbols (88), 2-byte lengths (n_0 = 1000)
    assert c == 40 * (1 + 2) + 600 * 2 + 250 * (2 + 2) + 700 * (1 + 2)
    ab1, _ = b.induction_bytes(rounds, levels, 1)       # no profile counters at that level: the raw cell counts stand in
    assert ab1 == 80 * (2 + 1) + 80 * 2 + 200 * (2 + 1) + 2 * 2 * 200
    assert b.parse_bytes(rounds, 1) == (1000 + 300 * 4, 1000 + 300 * 4 + 300 * 4 + 90 * 4)


def test_launch_site_groups():
    b = load_bench()
    g = b.GROUP_SITES
    assert g["induce_AB"]("induce.xscatter", "i") and g["induce_AB"]("induce_pack_grammar", "i") and not g["induce_AB"]("asm.cell_atoms", "i")
    assert g["induce_C"]("asm.take_scan", "i") and g["induce_C"]("merge_runs.scan", "i") and not g["induce_C"]("merge_runs.scan", "p")
    assert g["hash_emit"]("hash_phrases", "p") and g["hash_emit"]("emit_parse", "p") and not g["hash_emit"]("suffix_sort0.scatter", "p")


def test_traffic_groups_against_the_committed_pmc_summary():
    """`traffic` of a group = the HBM bytes of the kernels whose rocprofv3 names hold one of the group's fragments, from the newest
    committed PMC summary.  No kernel of that summary may be counted in two groups (round 6 found `unsigned long, 2,` -- meant for
    the record sort -- matching the one-walk kernel of pass C, and `MapFn` matching ComposeMapFn), no fragment of a kernel the
    headline workload launches may match nothing (`traffic_stale`: rounds 3-4 lost the later radix passes of A+B that way), and the
    big kernels must all belong to a group."""
    import json
    b = load_bench()
    files = b.pmc_traffic_files()
    assert files, "no profiles/*/pmc_traffic.json committed"
    d = json.load(open(files[-1]))
    names = list(d["kernels"])
    for n in names:
        groups = [g for g, fr in b.GROUP_KERNELS.items() if any(f in n for f in fr)]
        assert len(groups) <= 1, (n[:100], groups)
    for g, fr in b.GROUP_KERNELS.items():
        assert b.pmc_traffic_stale(fr) == [], (g, b.pmc_traffic_stale(fr))
        assert b.pmc_traffic(fr) > 0
    builds = max(1, int(d.get("builds_profiled", 1)))
    loose = [(e["hbm_bytes_total"] / builds, n) for n, e in d["kernels"].items()
             if "prim::k_" in n and not any(f in n for fr in b.GROUP_KERNELS.values() for f in fr)
             and not any(s in n for s in ("k_reduce", "k_scan_tile", "k_pack_records", "k_byte_hist", "k_rs_chunk_sums", "k_rs_tile_offsets", "DictMetaInitFn", "DiffFn", "PackPhraseInfoFn", "ClaimSlotsFn"))]
    assert all(bytes_ < 5e9 for bytes_, _ in loose), sorted(loose, reverse=True)[:5]
