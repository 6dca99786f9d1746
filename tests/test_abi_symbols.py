"""The C-ABI library loads on a box without a GPU and exports every entry point include/teaser_hip.h
declares (no compute calls here); creating a solver without a device fails loudly."""
import ctypes as C
import importlib
import os
import re

import pytest

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
    assert r.returncode == 0 and "--depth" in r.stdout and "--batch" in r.stdout and "--gpus" in r.stdout
    if tp.device_count() == 0:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")], capture_output=True, text=True)
        assert r.returncode != 0 and "needs an MI355X" in (r.stderr + r.stdout)


def test_bench_roofline_object():
    """The roofline arithmetic of bench.py (64 x N = 10 k per launch, 0.80 ms): executed-pipe fractions, the largest
    one at the top level, none above 1; the FP64-equivalent ratio nested (it MAY exceed 1)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    pairs = 64 * 10000 * 9999 // 2
    byts = 64 * (48 * 10000 + 8 * 10000 * 157)
    issue = dict(valu_insts_per_1024_pairs=77.6, source="profiles/x/k1_sq_counters.json")
    r = bench.roofline_object(k1_ms=20 * 0.80, k1_launches=20, k1_bytes=20 * byts, k1_pairs=20 * pairs,
                              k1_aux_ms=20 * 0.15, traffic=1.78e9, traffic_src="profiles/x", issue=issue)
    assert set(r["pipes"]) == {"valu", "mfma", "hbm"}
    assert r["bound"] == "valu" and r["unit"] == "G wave-instructions/s" and r["peak"] == 1024 * 2.4 / 4
    assert abs(r["avg_launch_ms"] - 0.80) < 1e-9 and r["launches"] == 20
    assert abs(r["achieved"] - 77.6 * pairs / 1024 / 0.80e-3 / 1e9) < 1e-6 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["frac"] == max(p["frac"] for p in r["pipes"].values()) and all(0 <= p["frac"] <= 1 for p in r["pipes"].values())
    assert abs(r["pipes"]["mfma"]["achieved"] - 128 * pairs / 0.80e-3 / 1e12) < 1e-6 and r["pipes"]["mfma"]["peak"] == 2500.0
    assert abs(r["pipes"]["hbm"]["achieved"] - byts / 0.80e-3 / 1e9) < 1e-6 and r["traffic"] == 1.78e9
    f = r["fp64_equivalent"]
    assert abs(f["achieved"] - 20 * pairs / 0.80e-3 / 1e12) < 1e-6 and f["peak"] == 78.6 and f["ratio"] > 1.0
    # a counter pass that also holds the vector-memory instruction count adds the vector-L1 pipe (1 KB per instruction)
    r1 = bench.roofline_object(20 * 0.80, 20, 20 * byts, 20 * pairs, 0.0, None, None,
                               issue=dict(issue, vmem_insts_per_1024_pairs=3.04, ta_busy_frac=0.65))
    assert set(r1["pipes"]) == {"valu", "mfma", "hbm", "l1"} and r1["pipes"]["l1"]["peak"] == 256 * 64 * 2.4
    assert abs(r1["pipes"]["l1"]["achieved"] - 3.04 * pairs / 0.80e-3 / 1e9) < 1e-6 and 0 < r1["pipes"]["l1"]["frac"] < 1
    r0 = bench.roofline_object(20 * 0.80, 20, 20 * byts, 20 * pairs, 0.0, None, None)  # no counter pass committed
    assert r0["pipes"]["valu"]["valu_insts_per_1024_pairs"] == bench.K1_VALU_PER_1024_STATIC
    z = bench.roofline_object(0.0, 0, 0, 0, 0.0, None, None)  # no launches: no division by zero
    assert z["achieved"] == 0.0 and z["traffic"] is None


def test_roofline_flat_copies_and_measured_valu_peak():
    """A record that keeps scalars only must still carry the pipe fractions, the traffic ratio and BOTH VALU peaks:
    the assumed one (4 cycles per wave64 instruction at 2.4 GHz) and the one measured for K1's instruction mix
    (scripts/probe/valu_rate weighted by scripts/k1_isa_stats.py, both committed under profiles/)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    pairs = 64 * 10000 * 9999 // 2
    byts = 64 * (48 * 10000 + 8 * 10000 * 157)
    r = bench.roofline_object(20 * 0.75, 20, 20 * byts, 20 * pairs, 0.0, 1.7e9, "profiles/x",
                              issue=dict(valu_insts_per_1024_pairs=69.4, vmem_insts_per_1024_pairs=2.5, source="profiles/x"))
    for k in ("frac_valu", "frac_mfma", "frac_hbm", "frac_l1"):
        assert r[k] == r["pipes"][k[5:]]["frac"]
    assert abs(r["traffic_over_algorithmic"] - 1.7e9 / byts) < 1e-9
    m = r["valu_peak_measurement"]
    assert m is not None and m["timed_share_of_mix"] > 0.8 and 2.0 < m["cycles_per_valu_inst_at_assumed_ghz"] < 6.0
    assert r["valu_peak_assumed"] == 1024 * 2.4 / 4 and abs(r["valu_peak_measured"] - 1024 * m["assumed_ghz"] / m["cycles_per_valu_inst_at_assumed_ghz"]) < 0.1
    assert abs(r["frac_valu_at_measured_peak"] - r["pipes"]["valu"]["achieved"] / r["valu_peak_measured"]) < 1e-12


def test_set_option_validates_names_and_ranges():
    """teaser_hip_set_option needs no device: unknown names and values outside an option's range answer BAD_ARG and
    leave the table untouched (round-5 advice: a negative k4_lds_stack used to be masked and used)."""
    L = tp.lib()
    assert L.teaser_hip_set_option(None, b"no_such_option", 1) != 0
    assert L.teaser_hip_set_option(None, b"k4_lds_stack", -16) != 0
    assert L.teaser_hip_set_option(None, b"depth", 0) != 0 and L.teaser_hip_set_option(None, b"depth", 17) != 0
    assert L.teaser_hip_set_option(None, b"deg_closure", 2) != 0
    for name, v in ((b"depth", 2), (b"deg_closure", 1), (b"greedy_small", 1), (b"k4_lds_stack", 16384)):
        assert L.teaser_hip_set_option(None, name, v) == 0
    with pytest.raises(tp.TeaserHipError):
        tp.set_option("fused_estimators", 7)
    # the keyword-only device: Params followed by a positional argument is refused loudly (round-5 advice)
    with pytest.raises(TypeError, match="device"):
        tp.RobustRegistrationSolver._params_from_ctor_args((tp.RobustRegistrationSolver.Params(), 0), {})


def test_every_documented_option_exists_with_its_default():
    """INTEGRATION.md's settings table against the library's: every option a row names is accepted with the default the
    row states (a renamed or dropped option fails here; the call also leaves the table at its defaults), and the options
    this round added check their ranges."""
    L = tp.lib()
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    table = text[text.index("| option | env | default | meaning |"):]
    table = table[:table.index("\n\n")]
    seen = set()
    for line in table.splitlines()[2:]:
        cols = [c.strip() for c in line.strip().strip("|").split("|")]
        names = re.findall(r"`([a-z0-9_]+)`", cols[0])
        defaults = [int(x) for x in re.findall(r"-?\d+", cols[2].replace("\u2212", "-").replace(" ", ""))]
        assert names and defaults, line[:80]
        if len(defaults) == 1:
            defaults = defaults * len(names)
        assert len(defaults) == len(names), (names, defaults)
        for name, default in zip(names, defaults):
            assert L.teaser_hip_set_option(None, name.encode(), default) == 0, (name, default)
            seen.add(name)
    assert len(seen) >= 30 and {"colour_mis", "colour_mis_any", "colour_persistent", "deg_closure", "scale_hull"} <= seen, sorted(seen)
    assert L.teaser_hip_set_option(None, b"colour_mis", 65537) != 0 and L.teaser_hip_set_option(None, b"colour_mis_any", 2) != 0
