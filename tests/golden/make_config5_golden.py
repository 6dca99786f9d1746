#!/usr/bin/env python3
"""Generates tests/golden/config5_clouds.npz: BASELINE config 5's input, the 3DMatch pair of the reference's
examples/teaser_python_fpfh_icp (data/cloud_bin_0.ply, cloud_bin_4.ply: binary little-endian PLY, float
xyz + more properties, 258 342 / 313 395 points), voxel-down-sampled at VOXEL_SIZE = 0.05 as example.py:19-20
does (Open3D voxel_down_sample: voxel index = floor((p - (min_bound - voxel/2)) / voxel), output = mean of
the voxel's points; Open3D emits the voxels in hash-map order, here they are sorted by voxel index so that
the fixture is reproducible).  The down-sampled clouds (a few thousand float32 points each) are what the GPU
front-end (FPFH + matcher) and the registration consume in tests/test_gpu_features.py::test_config5_*.

Run from the repo root (needs /root/reference):  python tests/golden/make_config5_golden.py
"""
import os

import numpy as np

REF = "/root/reference/examples/teaser_python_fpfh_icp/data/"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
VOXEL = 0.05

PLY_TYPES = {"float": "<f4", "float32": "<f4", "double": "<f8", "float64": "<f8", "uchar": "u1", "uint8": "u1",
             "char": "i1", "int8": "i1", "short": "<i2", "int16": "<i2", "ushort": "<u2", "uint16": "<u2",
             "int": "<i4", "int32": "<i4", "uint": "<u4", "uint32": "<u4"}


def read_binary_ply_xyz(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"ply"
        fmt, n, props, in_vertex = None, 0, [], False
        while True:
            line = f.readline().decode().strip()
            if line == "end_header":
                break
            tok = line.split()
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    n = int(tok[2])
            elif tok[0] == "property" and in_vertex:
                assert tok[1] != "list"
                props.append((tok[2], PLY_TYPES[tok[1]]))
        assert fmt == "binary_little_endian", fmt
        data = np.frombuffer(f.read(n * np.dtype(props).itemsize), dtype=np.dtype(props), count=n)
    return np.stack([data["x"], data["y"], data["z"]], axis=1).astype(np.float64)


def voxel_down_sample(p, voxel):
    lo = p.min(0) - voxel * 0.5
    idx = np.floor((p - lo) / voxel).astype(np.int64)
    key = (idx[:, 0] * (1 << 42)) + (idx[:, 1] * (1 << 21)) + idx[:, 2]
    order = np.argsort(key, kind="stable")
    key, p = key[order], p[order]
    start = np.flatnonzero(np.concatenate([[True], key[1:] != key[:-1]]))
    sums = np.add.reduceat(p, start, axis=0)
    cnt = np.diff(np.concatenate([start, [len(p)]]))[:, None]
    return (sums / cnt).astype(np.float32)


out = {}
for name in ("cloud_bin_0", "cloud_bin_4"):
    raw = read_binary_ply_xyz(REF + name + ".ply")
    ds = voxel_down_sample(raw, VOXEL)
    print(name, raw.shape, "->", ds.shape)
    out[name] = ds
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "config5_clouds.npz"), voxel_size=np.float32(VOXEL), **out)
