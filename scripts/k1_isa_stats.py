#!/usr/bin/env python3
"""Static instruction statistics of the K1 kernel from the compiler's own assembly (CPU only, no GPU):
  hipcc --offload-arch=gfx950 ... --save-temps -c csrc/kernels_graph.hip   ->   *-gfx950.s
For the default instantiation tim_graph_mfma3_kernel<0, false, 3, 1, 1, false>: registers, LDS, and per BASIC BLOCK the
instruction mix (VALU / MFMA / SALU / LDS / VMEM / waits); the block with the most MFMAs is the column-tile loop body.
A loop iteration covers one 64 x 64 column tile of a wave = 4096 pairs, so VALU per 1024 pairs = VALU / 4.
usage: k1_isa_stats.py [<file.s>]   (without an argument the file is produced in a temporary directory)"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "_ZN4thip22tim_graph_mfma3_kernelILi0ELb0ELi3ELi1ELi1ELb0EE"


def assembly():
    d = tempfile.mkdtemp(prefix="k1isa")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "teaser-plusplus_amd", "csrc"),
                           "-mllvm", "-amdgpu-mfma-vgpr-form", "--save-temps", "-c",
                           os.path.join(ROOT, "teaser-plusplus_amd", "csrc", "kernels_graph.hip"), "-o", os.path.join(d, "kg.o")],
                          cwd=d, stderr=subprocess.DEVNULL)
    return [os.path.join(d, f) for f in os.listdir(d) if f.endswith("gfx950.s")][0]


def classify(op):
    if "mfma" in op:
        return "mfma"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_barrier") or op.startswith("s_sleep"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("buffer_") or op.startswith("global_") or op.startswith("flat_") or op.startswith("scratch_"):
        return "vmem"
    return "other"


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else assembly()
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith(KERNEL) and l.rstrip().endswith(":") or (l.startswith(KERNEL) and ": " in l))
    blocks, cur, name = [], collections.Counter(), "entry"
    meta = {}
    for l in lines[start + 1:]:
        t = l.strip()
        if t.startswith(".Lfunc_end") or t.startswith(".section"):
            break
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            blocks.append((name, cur))
            cur, name = collections.Counter(), m.group(1)
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        op = t.split()[0]
        cur[classify(op)] += 1
        cur["ops:" + op] += 1
    blocks.append((name, cur))
    for l in lines[start:]:
        for key in ("NumVgprs", "NumAgprs", "ScratchSize", "Occupancy", "LDSByteSize", "NumSgprs"):
            m = re.match(r"^;\s*%s:\s*(\d+)" % key, l.strip())
            if m and key not in meta:
                meta[key] = int(m.group(1))
        if len(meta) == 6:
            break
    total = collections.Counter()
    for _, c in blocks:
        total.update({k: v for k, v in c.items() if not k.startswith("ops:")})
    loops = sorted(blocks, key=lambda b: -b[1]["mfma"])[:2]
    out = {"kernel": "tim_graph_mfma3_kernel<0, false, 3, 1, 1, false>", "resources": meta, "basic_blocks": len(blocks),
           "static_total": dict(total)}
    for rank, (nm, c) in enumerate(loops):
        mix = {k: v for k, v in c.items() if not k.startswith("ops:")}
        top = sorted(((k[4:], v) for k, v in c.items() if k.startswith("ops:")), key=lambda kv: -kv[1])[:14]
        out["loop_body_%d" % rank] = {"label": nm, "mix": mix, "valu_per_1024_pairs": round(mix.get("valu", 0) / 4.0, 1),
                                      "mfma_per_1024_pairs": round(mix.get("mfma", 0) / 4.0, 2), "most_frequent": dict(top)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
