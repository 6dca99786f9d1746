#!/usr/bin/env python3
"""Static instruction statistics of the K1 kernel from the compiler's own assembly (CPU only, no GPU):
  hipcc --offload-arch=gfx950 <the Makefile's flags for kernels_graph.o> --save-temps -c csrc/kernels_graph.hip
For EVERY instantiation of tim_graph_mfma3_kernel<PIPE, PLAIN, OCC, CHUNKS> in the object (one in the product build,
twelve with --lab = -DTEASER_K1_LAB): registers, scratch, LDS, occupancy, and per BASIC BLOCK the instruction mix
(VALU / MFMA / SALU / LDS / VMEM / waits).  The block with the most MFMAs is the steady-state column-tile loop body:
one iteration covers one 64 x 64 column tile of a wave = 4096 pairs, so VALU per 1024 pairs = VALU / 4.
usage: k1_isa_stats.py [--lab] [--brief] [<file.s>]"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PREFIX = "_ZN4thip22tim_graph_mfma3_kernelI"


def assembly(lab):
    d = tempfile.mkdtemp(prefix="k1isa")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
           "-fvisibility=hidden", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "teaser-plusplus_amd", "csrc"),
           "-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-slp-vectorize", "--save-temps", "-c",
           os.path.join(ROOT, "teaser-plusplus_amd", "csrc", "kernels_graph.hip"), "-o", os.path.join(d, "kg.o")]
    if lab:
        cmd.insert(1, "-DTEASER_K1_LAB")
    subprocess.check_call(cmd, cwd=d, stderr=subprocess.DEVNULL)
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
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith("buffer_") or op.startswith("global_") or op.startswith("flat_"):
        return "vmem"
    return "other"


def template_args(sym):
    m = re.match(re.escape(PREFIX) + r"Lb([01])ELb([01])ELi(\d+)ELi(\d+)EE", sym)
    return "<PIPE=%s, PLAIN=%s, OCC=%s, CHUNKS=%s>" % m.groups() if m else sym


def one_kernel(lines, start):
    blocks, cur, name = [], collections.Counter(), "entry"
    meta = {}
    end = start
    for off, l in enumerate(lines[start + 1:]):
        t = l.strip()
        if t.startswith(".Lfunc_end"):
            end = start + 1 + off
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
    for l in lines[end:end + 400]:
        for key in ("NumVgprs", "NumAgprs", "ScratchSize", "Occupancy", "LDSByteSize", "NumSgprs"):
            m = re.match(r"^;\s*%s:\s*(\d+)" % key, l.strip())
            if m and key not in meta:
                meta[key] = int(m.group(1))
        if len(meta) == 6:
            break
    total = collections.Counter()
    for _, c in blocks:
        total.update({k: v for k, v in c.items() if not k.startswith("ops:")})
    # the two blocks with the most MFMAs: the steady-state body (fewer VALU: rank 0) and the diagonal tile's copy
    loops = sorted(sorted(blocks, key=lambda b: -b[1]["mfma"])[:2], key=lambda b: b[1]["valu"])
    out = {"resources": meta, "basic_blocks": len(blocks), "static_total": dict(total)}
    for rank, (nm, c) in enumerate(loops):
        mix = {k: v for k, v in c.items() if not k.startswith("ops:")}
        top = sorted(((k[4:], v) for k, v in c.items() if k.startswith("ops:")), key=lambda kv: -kv[1])[:14]
        per = 4.0 * max(1, mix.get("mfma", 16) // 16)  # (a block that holds two tiles' MFMAs covers 8192 pairs)
        out["loop_body_%d" % rank] = {"label": nm, "mix": mix, "valu_per_1024_pairs": round(mix.get("valu", 0) / per, 1),
                                      "mfma_per_1024_pairs": round(mix.get("mfma", 0) / per, 2), "most_frequent": dict(top)}
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    lab, brief = "--lab" in sys.argv, "--brief" in sys.argv
    path = args[0] if args else assembly(lab)
    lines = open(path).read().splitlines()
    out = {}
    for i, l in enumerate(lines):
        if l.startswith(PREFIX) and ": " in l:
            sym = l.split(":")[0]
            out["tim_graph_mfma3_kernel" + template_args(sym)] = one_kernel(lines, i)
    if brief:
        for k, v in out.items():
            lb, dg = v["loop_body_0"], v["loop_body_1"]
            print("%-58s vgpr %3d scratch %3d occ %d lds %5d | loop body: valu %3d (%.1f / 1024 pairs; diagonal copy %3d) mfma %2d salu %3d lds %2d vmem %2d scratch %d | static valu %d"
                  % (k, v["resources"].get("NumVgprs", -1), v["resources"].get("ScratchSize", -1), v["resources"].get("Occupancy", -1),
                     v["resources"].get("LDSByteSize", -1), lb["mix"].get("valu", 0), lb["valu_per_1024_pairs"], dg["mix"].get("valu", 0), lb["mix"].get("mfma", 0),
                     lb["mix"].get("salu", 0), lb["mix"].get("lds", 0), lb["mix"].get("vmem", 0), lb["mix"].get("scratch", 0),
                     v["static_total"].get("valu", 0)))
    else:
        print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
