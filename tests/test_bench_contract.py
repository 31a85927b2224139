"""The JSON line bench.py prints: every key of the driver's contract must be in the dict literal it is built from (a comment
once swallowed two of them).  Static check: bench.py itself needs a GPU."""
import ast
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "roofline", "cpu_baseline"}


def test_bench_line_has_every_contract_key():
    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    best = set()
    for node in ast.walk(tree):
        if isinstance(node, ast.Dict):
            keys = {k.value for k in node.keys if isinstance(k, ast.Constant) and isinstance(k.value, str)}
            if "metric" in keys and "ms_per_step" in keys:
                best = keys
    assert best, "bench.py builds no result line"
    assert REQUIRED <= best, sorted(REQUIRED - best)


def test_roofline_objects_carry_their_fields():
    src = open(os.path.join(ROOT, "bench.py")).read()
    for key in ('"bound"', '"achieved"', '"peak"', '"frac"', '"traffic"', '"traffic_source"', '"cores"', '"kind"', '"sample"', '"devices"'):
        assert key in src, key
