#!/usr/bin/env python3
"""Knock-out variants of the product's K1 kernel for A/B timing on ONE box (diagnostics, not the product).
Writes gpurun_ab/kernels_graph_<name>.hip = csrc/kernels_graph.hip with one ingredient of the column-tile loop removed.
Variants other than `old` / `touch` produce WRONG bits, so their outputs are forced to zero and their flags dropped
behind an opaque use (empty graph, no worklist overflow, no fallback: the events time the kernel alone).
  zero      control: outputs zeroed, nothing removed
  nomfma    no MFMAs (accumulators defined by an empty asm)        halfmfma  the u chain's first link and the w MFMA only
  noepi     no epilogue (sign collection, min |d|, flags)           nofinish  no word assembly / transposes / LDS parking
  noload    no column-operand loads in the loop (operands of the first tile reused)
  sameload  the loop's loads always fetch the wave's first column tile
  load32    buffer_load_dword instead of dwordx4 (same instruction count, a quarter of the bytes)
  ldsload   the loop's operands come from LDS (ds_read_b128 of a constant buffer) instead of global memory
  touch     (correct results) one extra dword load per 64-byte line of the half tile after the next, as a prefetch
  seq3/seq4 (correct results) ONE accumulator set live at a time (row half 0: 4 MFMAs + epilogue, then row half 1 with the
            next half tile's loads behind its MFMAs) at 3 / 4 waves per SIMD: 151 VGPRs / 128 with no scratch access in
            the steady-state loop.  Round 6: 0.606 - 0.612 ms alone against 0.596 - 0.611 for the product (two sets,
            168 VGPRs, 3 waves): the fourth wave buys nothing (profiles/r6b/k1_sequential_accumulators.txt)
usage: make_variants.py [name ...]   then  bash scripts/probe/k1_ab/build.sh <name> ...   (see README.md)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
SRC = open(os.path.join(ROOT, "teaser-plusplus_amd", "csrc", "kernels_graph.hip")).read()
OUT = os.path.join(ROOT, "gpurun_ab")
LOADS = ["      bX[%d] = load_op(Jn, 1, gn, %d);" % (m, m) for m in range(3)]  # the three column operands of the next half tile
EPI = "      for (int rt = 0; rt < 2; ++rt) tr[ct][rt] = epi(acc[rt], DIAG && ct == rt);"


def once(s, old, new):
    assert s.count(old) == 1, old
    return s.replace(old, new)


def each_load(bd, f):
    """replace the three operand-load lines of the flat body: f(m) -> replacement line"""
    for m, l in enumerate(LOADS):
        bd = once(bd, l, f(m))
    return bd


def zeroed(s):
    s = once(s, "    lds_own[wave][lane][(J - Jbase) & (kMfmaColTiles - 1)] = ownw;\n",
             '    asm volatile("" :: "v"(ownw)); ownw = 0;\n    lds_own[wave][lane][(J - Jbase) & (kMfmaColTiles - 1)] = ownw;\n')
    s = once(s, "    lds_tr[(J - Jbase) & (kMfmaColTiles - 1)][lane][wave] = trw_out;\n",
             '    asm volatile("" :: "v"(trw_out));\n    lds_tr[(J - Jbase) & (kMfmaColTiles - 1)][lane][wave] = 0ull;\n')
    return once(s, "      unsigned int vf = flags;\n", '      asm volatile("" :: "v"(flags));\n      unsigned int vf = 0;\n')


def flat_body(s):
    a = s.index("  auto body_flat = [&](const int J, auto diag_tag) {")
    b = s.index("  // pipelined schedule.  Invariant at the top")
    return a, b


def in_body(s, f):
    a, b = flat_body(s)
    nb = f(s[a:b])
    assert nb != s[a:b]
    return s[:a] + nb + s[b:]


def no_mfma(bd, keep=lambda l: False):
    out = [l for l in bd.split("\n") if not ("__builtin_amdgcn_mfma_f32_32x32x16_bf16" in l and "acc[" in l) or keep(l)]
    return "\n".join(out)



SEQ_BODY = '''  auto body_flat = [&](const int J, auto diag_tag) {
    constexpr bool DIAG = decltype(diag_tag)::value;
    unsigned int tr[2][2];  // [ct][rt]: this lane's 16 column bits
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const bf16x8 b0 = __builtin_bit_cast(bf16x8, bX[0]), b1 = __builtin_bit_cast(bf16x8, bX[1]);
      const bf16x8 b2 = __builtin_bit_cast(bf16x8, bX[2]);
      const int Jn = (ct == 0 || J + 1 >= Jend) ? J : J + 1, gn = ct ^ 1;
      f32x16 z;
      for (int k = 0; k < 16; ++k) z[k] = 0.f;
      {
        Acc acc;
        __builtin_amdgcn_s_setprio(2);
        acc.U = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[0][0], b0, z, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        acc.W = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[0][3], b0, z, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        acc.U = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[0][1], b1, acc.U, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        acc.U = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[0][2], b2, acc.U, 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        tr[ct][0] = epi(acc, DIAG && ct == 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      {
        Acc acc;
        __builtin_amdgcn_s_setprio(2);
        acc.U = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[1][0], b0, z, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        acc.W = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[1][3], b0, z, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        bX[0] = load_op(Jn, 1, gn, 0);
        __builtin_amdgcn_sched_barrier(0);
        acc.U = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[1][1], b1, acc.U, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        bX[1] = load_op(Jn, 1, gn, 1);
        __builtin_amdgcn_sched_barrier(0);
        acc.U = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[1][2], b2, acc.U, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        bX[2] = load_op(Jn, 1, gn, 2);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        tr[ct][1] = epi(acc, DIAG && ct == 1);
      }
    }
    finish_tile(J, DIAG, tr);
  };

'''


def sequential(occ):
    a, b = flat_body(SRC)
    return once(SRC[:a] + SEQ_BODY + SRC[b:], "constexpr int kK1Chunks = 1, kK1Occ = 3;", "constexpr int kK1Chunks = 1, kK1Occ = %d;" % occ)


VARIANTS = {
    "seq3": lambda: sequential(3),
    "seq4": lambda: sequential(4),
    "zero": lambda: zeroed(SRC),
    "nomfma": lambda: in_body(zeroed(SRC), lambda bd: once(no_mfma(bd), LOADS[0],
        '      asm volatile("" : "=v"(acc[0].U), "=v"(acc[0].W), "=v"(acc[1].U), "=v"(acc[1].W) : "v"(b0), "v"(b1), "v"(b2));\n' + LOADS[0])),
    "halfmfma": lambda: in_body(zeroed(SRC), lambda bd: no_mfma(bd, keep=lambda l: not ("b1, acc" in l or "b2, acc" in l))),
    "noepi": lambda: in_body(zeroed(SRC), lambda bd: once(bd, EPI,
        '      for (int rt = 0; rt < 2; ++rt) { asm volatile("" :: "v"(acc[rt].U), "v"(acc[rt].W)); tr[ct][rt] = (unsigned int)lane * 2654435761u + J; }')),
    "nofinish": lambda: in_body(zeroed(SRC), lambda bd: once(bd, "    finish_tile(J, DIAG, tr);",
        '    asm volatile("" :: "v"(tr[0][0]), "v"(tr[0][1]), "v"(tr[1][0]), "v"(tr[1][1]));')),
    "noload": lambda: in_body(zeroed(SRC), lambda bd: each_load(bd, lambda m: "      (void)Jn; (void)gn;")),
    "sameload": lambda: in_body(zeroed(SRC), lambda bd: each_load(bd, lambda m:
        "      (void)Jn; bX[%d] = load_op(Jfirst < T ? Jfirst : T - 1, 1, gn, %d);" % (m, m))),
    "load32": lambda: in_body(zeroed(SRC), lambda bd: each_load(bd, lambda m:
        "      bX[%d].x = __builtin_amdgcn_raw_buffer_load_b32(q_rsrc, lane * 4, Jn * (int)sizeof(TimOperandTile2) + (int)offsetof(TimOperandTile2, b) + (gn * kTimColOperands + %d) * 1024, 0);" % (m, m))),
    "ldsload": lambda: once(in_body(zeroed(SRC), lambda bd: each_load(bd, lambda m:
        "      (void)Jn; bX[%d] = lds_b[(2 * J + gn) & 1][%d][lane];" % (m, m))),
        "  uint4 bX[kTimColOperands], bY[kTimColOperands];",
        "  __shared__ uint4 lds_b[2][4][64];\n  lds_b[0][wave][lane] = make_uint4(lane, wave, 1, 2); lds_b[1][wave][lane] = make_uint4(wave, lane, 3, 4);\n  __syncthreads();\n  uint4 bX[kTimColOperands], bY[kTimColOperands];"),
    "touch": lambda: once(once(in_body(SRC, lambda bd: once(bd, LOADS[2], LOADS[2] + """
      {
        const int half2 = 2 * J + ct + 2, J2 = min(half2 >> 1, Jend - 1), g2 = half2 & 1;
        asm volatile("" :: "v"(touch[ct]));
        touch[ct] = __builtin_amdgcn_raw_buffer_load_b32(q_rsrc, lane * 64, J2 * (int)sizeof(TimOperandTile2) + (int)offsetof(TimOperandTile2, b) + g2 * kTimColOperands * 1024, 0);
      }""")), "  uint4 bX[kTimColOperands], bY[kTimColOperands];", "  unsigned int touch[2] = {0u, 0u};\n  uint4 bX[kTimColOperands], bY[kTimColOperands];"),
        "  if (rowvalid)\n    __builtin_amdgcn_raw_ptr_buffer_atomic_add_i32(degacc,",
        '  asm volatile("" :: "v"(touch[0]), "v"(touch[1]));\n  if (rowvalid)\n    __builtin_amdgcn_raw_ptr_buffer_atomic_add_i32(degacc,'),
}

if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    for name in (sys.argv[1:] or list(VARIANTS)):
        open(os.path.join(OUT, "kernels_graph_%s.hip" % name), "w").write(VARIANTS[name]())
        print("wrote gpurun_ab/kernels_graph_%s.hip" % name)
