"""The HOST side of the asynchronous batch API under ThreadSanitizer and AddressSanitizer + UBSan (CPU only).

csrc/solver.hip -- lanes, tickets, the staging slot of host batches, the per-lane finisher threads, the speculative
bound stage's bookkeeping, destruction with batches in flight -- is compiled by g++ against a host-only stand-in for the
HIP runtime (tests/hip_stub/hip/hip_runtime.h: device memory = host memory, streams and events complete at once,
the three copy kernels solver.hip launches itself are run on the calling thread, every other launch does nothing;
tests/host_stub_launchers.cpp: the launchers of the other .hip files as stand-ins -- the "peel" closes a problem or
leaves it open as the driver says, the "colouring bound" leaves roots for every third open problem, the "exact search"
now and then finds a larger clique, overflows its arena once, hits the time limit: pools, retries and the second
estimator pass all run).  The numbers are
meaningless, the bookkeeping and the threads are the real ones.  tests/host_threads_driver.cpp drives random
interleavings of submit / wait (any order) / getters / depth changes / synchronous solves and checks the API's contract;
any sanitizer report fails the run.  ("solver.hip compiled host-only against a stub device layer".)

The first run of this test found a real contract bug: submit answered BUSY while a lane was free (it only looked at
the lane whose turn it was; tickets may be waited for in any order)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "teaser-plusplus_amd", "csrc")
SOURCES = [os.path.join(CSRC, "solver.hip"), os.path.join(ROOT, "tests", "host_stub_launchers.cpp"),
           os.path.join(ROOT, "tests", "host_threads_driver.cpp")]


def build(out, sanitizers):
    cmd = ["g++", "-x", "c++", "-std=c++17", "-O1", "-g", "-pthread", "-fsanitize=" + sanitizers, "-fno-omit-frame-pointer",
           "-I" + os.path.join(ROOT, "tests", "hip_stub"), "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, *SOURCES,
           # launchers of paths the driver does not take (features, scale stage, certifier ...) stay unresolved
           "-Wl,--unresolved-symbols=ignore-all", "-o", out]
    subprocess.check_call(cmd)
    return out


@pytest.mark.parametrize("sanitizers,tag", [("thread", "tsan"), ("address,undefined", "asan")])
def test_async_host_side_is_clean_under_sanitizers(tmp_path, sanitizers, tag):
    exe = build(str(tmp_path / ("host_threads_" + tag)), sanitizers)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1:exitcode=66", ASAN_OPTIONS="detect_leaks=1:exitcode=66",
               UBSAN_OPTIONS="halt_on_error=1:exitcode=66:print_stacktrace=1")
    for finisher in ("1", "0"):
        for seed in (1, 2, 3):
            p = subprocess.run([exe, "500", str(seed)], capture_output=True, text=True, timeout=300,
                               env=dict(env, TEASER_HIP_FINISHER=finisher))
            assert p.returncode == 0, (finisher, seed, p.stdout[-500:], p.stderr[-3000:])
            assert "Sanitizer" not in p.stderr and "runtime error" not in p.stderr, p.stderr[-3000:]
            words = p.stdout.split()
            assert int(words[1]) > 100 and int(words[3]) > 100  # submitted, waited
            assert int(words[13]) > 10 and int(words[15]) > 10  # exact searches, speculative bound stages: both paths ran
            # the heuristic stage ran for some batches only (the others were decided by the degree closure)
            assert 10 < int(words[17]) < int(words[1]) + 3 * int(words[3])
