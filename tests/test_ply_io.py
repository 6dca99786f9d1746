"""include/teaser/ply_io.h (PLYReader / PLYWriter of the C++ facade; the reference's teaser/ply_io.h
interface, its tinyply dependency replaced by a from-scratch parser) on PLY files built from the
reference's own bunny vertices (golden `bunny` = examples/example_data/bun_zipper_res3.ply)."""
import os
import struct
import subprocess

import numpy as np
import pytest

from util import ROOT, golden

SRC = os.path.join(ROOT, "tests", "cxx", "ply_example.cpp")
EXE = os.path.join(ROOT, "tests", "cxx", "ply_example")


@pytest.fixture(scope="module")
def exe():
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           SRC, "-o", EXE])
    return EXE


def write_ply(path, pts, fmt, with_extras):
    """Like the Stanford file: optional extra vertex properties and a face element (list property)."""
    n = len(pts)
    faces = [(0, 1, 2), (2, 3, 4)] if with_extras else []
    hdr = ["ply", "format %s 1.0" % fmt, "comment test", "element vertex %d" % n,
           "property float x", "property float y", "property float z"]
    if with_extras:
        hdr += ["property float confidence", "property uchar red"]
        hdr = hdr[:3] + ["element camera 1", "property double focal", "property list uchar int dummy"] + hdr[3:]
        hdr += ["element face %d" % len(faces), "property list uchar int vertex_indices"]
    hdr.append("end_header")
    with open(path, "wb") as f:
        f.write(("\n".join(hdr) + "\n").encode())
        if fmt == "ascii":
            if with_extras:
                f.write(b"35.5 2 7 9\n")
            for p in pts:
                row = " ".join(repr(float(np.float32(v))) for v in p)
                f.write((row + (" 0.5 200" if with_extras else "") + "\n").encode())
            for fc in faces:
                f.write(("3 %d %d %d\n" % fc).encode())
        else:
            e = "<" if fmt == "binary_little_endian" else ">"
            if with_extras:
                f.write(struct.pack(e + "dBii", 35.5, 2, 7, 9))
            for p in pts:
                f.write(struct.pack(e + "fff", *[float(np.float32(v)) for v in p]))
                if with_extras:
                    f.write(struct.pack(e + "fB", 0.5, 200))
            for fc in faces:
                f.write(struct.pack(e + "Biii", 3, *fc))


@pytest.mark.parametrize("fmt", ["ascii", "binary_little_endian", "binary_big_endian"])
@pytest.mark.parametrize("extras", [False, True])
def test_reader_matches_reference_vertices(exe, tmp_path, fmt, extras):
    pts = golden()["bunny"].astype(np.float32)  # 1889 x 3, the reference's example cloud
    path = str(tmp_path / "in.ply")
    write_ply(path, pts, fmt, extras)
    out = subprocess.run([exe, path, str(tmp_path / "re.ply"), "1" if fmt != "ascii" else "0"],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    tok = out.stdout.split()
    assert int(tok[0]) == 1889
    sums = pts.astype(np.float64).sum(0)
    assert np.allclose([float(t) for t in tok[1:4]], sums, rtol=0, atol=1e-9)
    assert np.float32(tok[4]) == pts[0, 0] and np.float32(tok[5]) == pts[-1, 1] and np.float32(tok[6]) == pts[1889 // 2, 2]


def test_reader_rejects_garbage(exe, tmp_path):
    for name, content in (("missing.ply", None), ("noply.ply", b"plx\nformat ascii 1.0\nend_header\n"),
                          ("novertex.ply", b"ply\nformat ascii 1.0\nelement face 0\nend_header\n"),
                          ("trunc.ply", b"ply\nformat binary_little_endian 1.0\nelement vertex 3\nproperty float x\n"
                                        b"property float y\nproperty float z\nend_header\n\x00\x00")):
        path = str(tmp_path / name)
        if content is not None:
            open(path, "wb").write(content)
        assert subprocess.run([exe, path], capture_output=True).returncode == 1


def test_reads_the_reference_file_when_present(exe):
    ref = "/root/reference/examples/example_data/bun_zipper_res3.ply"
    if not os.path.exists(ref):
        pytest.skip("reference checkout not present (GPU box)")
    out = subprocess.run([exe, ref], capture_output=True, text=True)
    assert out.returncode == 0
    pts = golden()["bunny"].astype(np.float32)
    tok = out.stdout.split()
    assert int(tok[0]) == 1889 and np.allclose([float(t) for t in tok[1:4]], pts.astype(np.float64).sum(0), atol=1e-9)
