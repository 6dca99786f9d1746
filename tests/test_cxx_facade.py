"""The drop-in C++ facade (include/teaser/registration.h: teaser::RobustRegistrationSolver over the C
ABI) compiled with g++ against libteaser_hip.so and used like the reference's own example
(examples/teaser_cpp_ply/teaser_cpp_ply.cc:76-100)."""
import os
import subprocess

import pytest

from util import ROOT

SRC = os.path.join(ROOT, "tests", "cxx", "facade_example.cpp")
EXE = os.path.join(ROOT, "tests", "cxx", "facade_example")
LIBDIR = os.path.join(ROOT, "teaser-plusplus_amd")


def build_example():
    if not os.path.exists(os.path.join(LIBDIR, "libteaser_hip.so")):
        pytest.skip("libteaser_hip.so not built (run __graft_entry__.build())")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), SRC, "-o", EXE,
           "-L" + LIBDIR, "-lteaser_hip", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return EXE


def test_facade_compiles_and_fails_loudly_without_gpu():
    """No Eigen in this image: the header's Eigen-less value types are used.  Without a GPU the
    constructor must throw (exit code 77 of the example) -- there is no CPU path to fall back to."""
    exe = build_example()
    rc = subprocess.call([exe], stdout=subprocess.DEVNULL)
    import importlib
    tp = importlib.import_module("teaser-plusplus_amd")
    assert rc == (0 if tp.device_count() > 0 else 77)


@pytest.mark.gpu
def test_facade_solves_on_gpu():
    exe = build_example()
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
