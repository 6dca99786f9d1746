"""The C-ABI library loads on a box without a GPU and exports every entry point include/teaser_hip.h
declares (no compute calls here); creating a solver without a device fails loudly."""
import ctypes as C
import importlib
import os
import re

from util import ROOT

tp = importlib.import_module("teaser-plusplus_amd")


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "teaser_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"TEASER_HIP_API\s+[A-Za-z0-9_\s\*]*?\b(teaser_hip_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_entry_point_is_exported():
    names = declared_symbols()
    assert len(names) >= 25
    L = tp.lib()
    for name in names:
        assert getattr(L, name) is not None, name
    assert sorted(tp.EXPORTED_SYMBOLS) == names
    assert L.teaser_hip_abi_version() == 1


def test_no_device_is_a_loud_error():
    if tp.device_count() > 0:
        return  # on the GPU box the parity tests cover creation
    h = C.c_void_p()
    rc = tp.lib().teaser_hip_solver_create(None, 0, C.byref(h))
    assert rc == 3 and not h  # TEASER_HIP_ERR_NO_DEVICE: never a CPU fallback
    try:
        tp.RobustRegistrationSolver()
    except tp.TeaserHipError as e:
        assert "NO_DEVICE" in str(e) or "no MI355X" in str(e)
    else:
        raise AssertionError("constructing a solver without a GPU must raise")


def test_product_does_not_import_the_oracle():
    """oracle/ is test infrastructure: nothing under teaser-plusplus_amd/ or include/ may reference it."""
    bad = []
    for base in ("teaser-plusplus_amd", "include", "teaserpp_python"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".cpp")):
                    txt = open(os.path.join(dirpath, f), errors="ignore").read()
                    if re.search(r"^\s*(from\s+oracle\b|import\s+oracle\b)|#\s*include[^\n]*oracle|"
                                 r"libteaser_oracle|dlopen[^\n]*oracle", txt, flags=re.M):
                        bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_bench_cli_parses_without_gpu():
    """bench.py --help must work (argparse help strings) and, without a GPU, the run must stop with the
    explicit no-device message rather than a traceback from deep inside."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True)
    assert r.returncode == 0 and "--streams" in r.stdout and "--batch" in r.stdout
    if tp.device_count() == 0:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")], capture_output=True, text=True)
        assert r.returncode != 0 and "needs an MI355X" in (r.stderr + r.stdout)
