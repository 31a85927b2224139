"""CPU: the C-ABI library loads and exports every symbol include/grlbwt_hip.h declares
(no compute calls without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hip_lib():
    import __graft_entry__ as g
    return g.build_hip()


def _declared():
    txt = open(os.path.join(ROOT, "include", "grlbwt_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(grlbwt_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported(hip_lib):
    lib = ctypes.CDLL(hip_lib)
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "missing export %s" % n
    assert lib.grlbwt_abi_version() == 3


def test_python_binding_covers_header():
    from grlbwt_amd import engine
    assert sorted(engine.ABI_SYMBOLS) == _declared()


def test_missing_library_fails_loudly(tmp_path):
    from grlbwt_amd import engine
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        engine.load_library(str(tmp_path / "nope.so"))


def test_no_device_is_an_error(hip_lib):
    """Without a GPU the product refuses to run instead of falling back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from grlbwt_amd import engine
    with pytest.raises(engine.GrlbwtError):
        engine.Context(0, 0, hip_lib)


def test_product_does_not_touch_oracle():
    """The product tree never references oracle/ or the serial test stand-in."""
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "grlbwt_amd")):
        for f in files:
            if f.endswith((".py", ".hpp", ".hip", ".cpp", ".h")):
                s = open(os.path.join(base, f), errors="ignore").read()
                if re.search(r"^\s*(from|import)\s+oracle|liboracle|libgrlbwt_sim|prim_sim\.hpp\"", s, flags=re.M):
                    if "#include" in s and "prim_sim" in s and f == "engine_impl.hpp":
                        continue
                    bad.append(f)
    assert not bad, bad


def test_test_standin_is_refused_by_the_product(monkeypatch):
    """The serial stand-in of the device primitives (tests/hostsim) identifies itself; the host mirror only
    accepts it when a test explicitly opts in, so the product cannot end up on a CPU path."""
    import subprocess
    d = os.path.join(ROOT, "tests", "hostsim")
    subprocess.check_call(["make", "-s", "-C", d], stdout=subprocess.DEVNULL)
    sim = os.path.join(d, "_build", "libgrlbwt_sim.so")
    from grlbwt_amd import engine
    monkeypatch.setattr(engine, "_standin_paths", set())
    monkeypatch.setenv("GRLBWT_ALLOW_TEST_STANDIN", "1")       # the environment is not a way in
    with pytest.raises(RuntimeError, match="not the HIP library"):
        engine.Context(0, 0, sim)
    assert "GRLBWT_ALLOW_TEST_STANDIN" not in open(engine.__file__).read()
    lib = ctypes.CDLL(sim)
    lib.grlbwt_backend_name.restype = ctypes.c_char_p
    assert lib.grlbwt_backend_name() == b"serial-test-standin"


def test_hip_library_identifies_itself(hip_lib):
    lib = ctypes.CDLL(hip_lib)
    lib.grlbwt_backend_name.restype = ctypes.c_char_p
    assert lib.grlbwt_backend_name() == b"hip-gfx950"


def test_fault_injection_is_not_in_the_product(hip_lib):
    """The failure-agreement tests inject faults through GRLBWT_TEST_FAIL_RANK* -- in the serial test stand-in only: the product
    library never reads those variables (an environment variable could make a production rank throw)."""
    blob = open(hip_lib, "rb").read()
    assert b"GRLBWT_TEST_" not in blob              # no test hook of any kind (the dictionary-part padding of round 5 included)
    assert b"injected by the test" not in blob


def test_environment_switches_of_the_product_are_few(hip_lib):
    """Every GRLBWT_* variable the device library reads is a diagnostic (traces), a debugging aid of the allocator, or a limit the
    GPU tests lower to send small inputs down a large-input branch.  Experiment switches live in development builds
    (prim::dev_env, tools/build_dev.sh), the switches of the CPU suites in the test stand-in (prim::test_env): neither reaches
    the product binary (56 names, a dozen of them rejected experiments)."""
    blob = open(hip_lib, "rb").read()
    names = set(m.decode() for m in re.findall(rb"GRLBWT_[A-Z0-9_]+", blob))
    names -= {"GRLBWT_FLAG_CLASSIC_POOL", "GRLBWT_FLAG_FORCE_IDX64", "GRLBWT_BENCH_", "GRLBWT_HIP_LIB", "GRLBWT_E2E_TMP", "GRLBWT_SIM_LIB", "GRLBWT_"}
    assert not [n for n in names if n.startswith("GRLBWT_DEV_")], names
    assert len(names) <= 35, sorted(names)
    src = ""
    for f in ("engine_impl.hpp", "prim_hip.hpp", "capi_impl.hpp"):
        src += open(os.path.join(ROOT, "grlbwt_amd", "csrc", f)).read()
    read = set(re.findall(r'(?<![a-z_])getenv\("(GRLBWT_[A-Z0-9_]+)"\)', src))
    assert len(read) <= 35, sorted(read)
