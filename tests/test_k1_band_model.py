"""CPU check of the K1 filter's error-band algebra (csrc/kernels_graph.hip).

The matrix-core kernel decides a pair only when both band edges d - band, d + band have the same sign.
This test re-implements the operand preparation (centring, f32 rounding, exact three-way bf16 split,
K layout), an f32 accumulation under the documented hardware model (every addition rounds to f32;
several accumulation orders are tried) and the f32 epilogue with the SAME constants, and checks on
millions of pairs -- random, and engineered to sit at ulps-to-1e-3 relative distance from the decision
boundary -- that a pair the filter trusts always gets the reference's answer.  (The GPU parity tests
check the real hardware against the oracle; this one pins the constants and the formulas.)"""
import numpy as np

U = np.float32(2.0 ** -24)
K_EPS_U = np.float32(300.0)


def bf16_rne(v):
    b = v.astype(np.float32).view(np.uint32)
    r = ((b + np.uint32(0x7FFF) + ((b >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)) << np.uint32(16)
    return r.view(np.float32)


def split3(v):
    h = bf16_rne(v)
    r1 = (v - h).astype(np.float32)
    m = bf16_rne(r1)
    r2 = (r1 - m).astype(np.float32)
    return h, m, bf16_rne(r2)


def operands(pts64):
    """pts64 [n,3] -> (A-side [n,24], B-side [n,24]) f32 arrays of bf16-representable values, R^2."""
    lo, hi = pts64.astype(np.float32).min(0), pts64.astype(np.float32).max(0)
    c = 0.5 * (lo.astype(np.float64) + hi.astype(np.float64))
    x = (pts64 - c).astype(np.float32)
    n = ((x.astype(np.float64) ** 2).sum(1)).astype(np.float32)
    A, B = np.zeros((len(x), 24), np.float32), np.zeros((len(x), 24), np.float32)
    for k in range(3):
        h, m, l = split3(x[:, k])
        A[:, 6 * k:6 * k + 6] = np.stack([h, h, m, h, l, m], 1)
        B[:, 6 * k:6 * k + 6] = np.float32(-2) * np.stack([h, m, h, l, h, m], 1)
    h, m, l = split3(n)
    A[:, 18:21] = np.stack([h, m, l], 1)
    B[:, 18:21] = 1
    A[:, 21:24] = 1
    B[:, 21:24] = np.stack([h, m, l], 1)
    return A, B, float(n.max())


def accumulate(Ai, Bj, order):
    """sum_k A[i,k] B[j,k] with every product exact (bf16 x bf16) and every addition rounded to f32."""
    acc = np.zeros(Ai.shape[0], np.float32)
    for k in order:
        acc = (acc + (Ai[:, k].astype(np.float64) * Bj[:, k].astype(np.float64)).astype(np.float32)).astype(np.float32)
    return acc


def consts(beta, r2max):
    f = np.float32
    up = f(1.000001)
    b = f(beta) * up
    R2 = f(r2max) * up
    R = np.sqrt(R2) * up
    b2 = b * b * up
    b4 = b2 * b2 * up
    eps = K_EPS_U * U * R2 * up
    lam_lo = f(beta) * np.sqrt(f(r2max)) * f(0.999999)
    lam_hi = b * R * up
    eol = eps / lam_lo * up
    kappa = f(3.04) * U + f(2.03) * eol
    K2 = (f(15.0) * U * b2 + f(4.04) * eol * b2) * up
    K0 = (f(4) * eps * eps + f(4) * b2 * eps + f(7) * U * b4 + f(2.02) * eol * b4 + f(2.02) * eps * lam_hi +
          f(1.3e-13) * b * R2 * R + f(8e-15) * b2 * R2) * up
    assert kappa <= 0.25
    sc = f(1.001) / (f(1) - kappa)
    k2, k0 = K2 * sc * up, K0 * sc * up
    m2b2, fb4 = f(-2.0 * beta * beta), f(beta ** 4)
    tau = (b2 * (f(1) + f(8) * U) + f(2.1) * eps) * up
    return dict(c1lo=m2b2 - k2, c2lo=fb4 - k0, c1hi=m2b2 + k2, c2hi=fb4 + k0, tau=tau)


def reference_edge(src, dst, i, j, beta):
    a, b = src[j] - src[i], dst[j] - dst[i]
    v1 = np.sqrt((a[:, 0] * a[:, 0] + a[:, 1] * a[:, 1]) + a[:, 2] * a[:, 2])
    v2 = np.sqrt((b[:, 0] * b[:, 0] + b[:, 1] * b[:, 1]) + b[:, 2] * b[:, 2])
    return np.abs(v1 - v2) <= beta


def check(src, dst, beta, rng, npairs):
    n = len(src)
    As, Bs, r2s = operands(src)
    Ad, Bd, r2d = operands(dst)
    kc = consts(beta, max(r2s, r2d))
    i = rng.integers(0, n, size=npairs)
    j = rng.integers(0, n, size=npairs)
    keep = i != j
    i, j = i[keep], j[keep]
    ref = reference_edge(src, dst, i, j, beta)
    trusted_total = 0
    for order in (range(24), range(23, -1, -1), rng.permutation(24)):
        A = accumulate(As[i], Bs[j], list(order))
        B = accumulate(Ad[i], Bd[j], list(order))
        D, t = (A - B).astype(np.float32), (A + B).astype(np.float32)
        fma = lambda x, y, z: (x.astype(np.float64) * np.float64(y) + np.float64(z)).astype(np.float32)
        elo, ehi = fma(t, kc["c1lo"], kc["c2lo"]), fma(t, kc["c1hi"], kc["c2hi"])
        dlo = (D.astype(np.float64) * D.astype(np.float64) + elo.astype(np.float64)).astype(np.float32)
        dhi = (D.astype(np.float64) * D.astype(np.float64) + ehi.astype(np.float64)).astype(np.float32)
        slo, shi = np.signbit(dlo), np.signbit(dhi)
        short = ~(t > kc["tau"])
        trusted = (slo == shi) & ~short
        # a trusted pair: edge <=> sign bit of d_hi
        assert (shi[trusted] == ref[trusted]).all(), "filter trusted a wrong sign"
        trusted_total += int(trusted.sum())
    return trusted_total / (3.0 * len(i))


def test_band_random_and_adversarial():
    rng = np.random.default_rng(2025)
    # (a) the benchmark geometry: unit cube, beta = 0.02, 95 % outliers
    n = 3000
    src = rng.uniform(size=(n, 3))
    Rm = np.linalg.qr(rng.normal(size=(3, 3)))[0]
    dst = src @ Rm.T + rng.uniform(-1, 1, size=3)
    out = rng.uniform(size=n) < 0.95
    dst[out] = rng.uniform(-1, 1, size=(int(out.sum()), 3))
    dst[~out] += rng.uniform(-0.0057, 0.0057, size=(int((~out).sum()), 3))
    frac = check(src, dst, 0.02, rng, 1_000_000)
    assert frac > 0.995  # the filter decides almost everything
    # (b) pairs engineered onto the boundary: dst lengths = src lengths +- beta (1 + delta)
    for scale, beta in ((1.0, 0.02), (300.0, 0.1), (0.05, 2e-4)):
        n = 1500
        src = rng.uniform(-1, 1, size=(n, 3)) * scale
        Rm = np.linalg.qr(rng.normal(size=(3, 3)))[0]
        dst = src @ Rm.T
        off = rng.choice([0.0, 1.0, -1.0], size=n) * beta * (
            1 + rng.choice([0, 1e-15, 1e-12, 1e-9, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3], size=n))
        d = dst / np.linalg.norm(dst, axis=1, keepdims=True)
        dst = dst + d * (off * rng.uniform(0.3, 1.0, size=n))[:, None]
        check(src, dst, beta, rng, 600_000)
    # (c) large offsets are absorbed by the centring
    src = rng.uniform(size=(1000, 3)) + np.array([1e4, -2e4, 3e4])
    dst = src @ Rm.T + 50.0
    check(src, dst, 0.02, rng, 300_000)


# ---------------------------------------------------------------------------------------------
# second formulation (csrc/kernels_graph.hip, "u / w"): u = B - A - beta^2 over 42 K slots in three chained
# MFMAs, w = -4 beta^2 A over 13 slots in one, points scaled so that 4 beta^2 is a power of two; epilogue
# d = fma(u, u, w), -band = fma(w, K2, -K0), edges d -+ band; a pair is trusted iff both edges have one sign.
# ---------------------------------------------------------------------------------------------
K_EPS_U2 = np.float32(680.0)
K_EPS_A2 = np.float32(1300.0)


def scale2(beta):
    x = 4.0 * beta * beta
    kexp = int(np.floor(np.log2(x))) + 1
    return np.sqrt(2.0 ** kexp / x), kexp


def operands2(src64, dst64, beta):
    """per point: A-side / B-side f32 arrays [n, 64] of bf16-representable values (slots 0..47 the u chain,
    48..63 the w MFMA) and the largest squared norm of the scaled centred f32 points."""
    g, kexp = scale2(beta)
    kappa = np.float32(2.0 ** kexp)

    def centred(p):
        lo, hi = p.astype(np.float32).min(0), p.astype(np.float32).max(0)
        c = 0.5 * (lo.astype(np.float64) + hi.astype(np.float64))
        return ((p - c) * g).astype(np.float32)

    s, d = centred(src64), centred(dst64)
    na = (s.astype(np.float64) ** 2).sum(1)
    nb = (d.astype(np.float64) ** 2).sum(1)
    n = len(s)
    # the slot table of tim_prep_pack2_kernel: the w MFMA (row slots 48..63) runs over the u chain's FIRST column
    # operand again, so the column side has 48 slots and B[:, 48:64] below is a copy of B[:, 0:16]
    A, B = np.zeros((n, 64), np.float32), np.zeros((n, 64), np.float32)
    for k in range(3):
        h, m, l = split3(s[:, k])
        A[:, 3 * k:3 * k + 3] = np.stack([h, h, m], 1)
        B[:, 3 * k:3 * k + 3] = np.float32(2) * np.stack([h, m, h], 1)
        A[:, 48 + 3 * k:48 + 3 * k + 3] = kappa * np.stack([h, h, m], 1)
        A[:, 16 + 3 * k:16 + 3 * k + 3] = np.stack([h, l, m], 1)
        B[:, 16 + 3 * k:16 + 3 * k + 3] = np.float32(2) * np.stack([l, h, m], 1)
        h, m, l = split3(d[:, k])
        A[:, 28 + 6 * k:28 + 6 * k + 6] = np.stack([h, h, m, h, l, m], 1)
        B[:, 28 + 6 * k:28 + 6 * k + 6] = np.float32(-2) * np.stack([h, m, h, l, h, m], 1)
    beta2s = 2.0 ** (kexp - 2)
    h, m, l = split3(na.astype(np.float32))
    B[:, 9:11] = -np.stack([h, m], 1)
    A[:, 48 + 9:48 + 11] = kappa
    B[:, 11:14] = 1
    A[:, 48 + 11:48 + 13] = -kappa * np.stack([h, m], 1)
    h, m, l = split3((nb - na - beta2s).astype(np.float32))
    A[:, 11:14] = np.stack([h, m, l], 1)
    h, m, l = split3((nb - na).astype(np.float32))
    A[:, 25:28] = 1
    B[:, 25:28] = np.stack([h, m, l], 1)
    assert not B[:, 46:].any()
    B[:, 48:64] = B[:, 0:16]
    r2 = float(max(na.astype(np.float32).max(), nb.astype(np.float32).max()))
    return A, B, r2, g, kexp


def consts2(beta, r2max):
    f = np.float32
    g, kexp = scale2(beta)
    up = f(1.000001)
    b = f(beta * g) * up
    kappa = f(2.0 ** kexp)
    R2 = f(r2max) * up
    R = np.sqrt(R2) * up
    b2 = f(0.25) * kappa
    eps_u = K_EPS_U2 * U * R2 * up
    eps_a = K_EPS_A2 * U * R2 * up
    lam_lo = f(2) * f(beta * g) * np.sqrt(f(r2max)) * f(0.999999)
    lam_hi = f(2) * b * R * up
    eta = eps_u / lam_lo * up
    assert eta <= 0.125 and b2 * f(5.76) <= R2
    den = f(1) - f(2) * eta - f(2) * U
    K2 = eta / den * up
    G = (f(1.3e-13) * b * R2 * R + f(8e-15) * b2 * R2) * up
    K0p = (eps_u * lam_hi + eps_u * eps_u + kappa * eps_a * (f(1) + eta) + G) * up
    K0 = (K0p / den + f(2) * K2 * kappa * eps_a) * up
    short_d = (f(4) * b2 * b2 * (f(1) + f(16) * U) + f(4) * b2 * eps_u + eps_u * eps_u + kappa * eps_a) * f(1.001) * up
    K0e = K0 * f(1.001) * up
    return dict(K2=K2 * f(1.001) * up, K0=max(K0e, short_d) * f(1.00001))


def check2(src, dst, beta, rng, npairs):
    n = len(src)
    A, B, r2, g, kexp = operands2(src, dst, beta)
    kc = consts2(beta, r2)
    i = rng.integers(0, n, size=npairs)
    j = rng.integers(0, n, size=npairs)
    keep = i != j
    i, j = i[keep], j[keep]
    ref = reference_edge(src, dst, i, j, beta)
    fma = lambda x, y, z: (np.asarray(x, np.float64) * np.asarray(y, np.float64) + np.asarray(z, np.float64)).astype(np.float32)
    trusted_total = 0
    for order_u, order_w in ((list(range(48)), list(range(48, 64))),
                             (list(range(47, -1, -1)), list(range(63, 47, -1))),
                             (list(rng.permutation(48)), list(48 + rng.permutation(16)))):
        u = accumulate(A[i], B[j], order_u)
        w = accumulate(A[i], B[j], order_w)
        d = fma(u, u, w)
        nb = fma(w, kc["K2"], -kc["K0"])
        dlo, dhi = (d + nb).astype(np.float32), (d - nb).astype(np.float32)
        slo, shi = np.signbit(dlo), np.signbit(dhi)
        trusted = slo == shi
        # a trusted pair: edge <=> sign bit of d + band (both edges negative); short pairs must never be trusted
        assert (shi[trusted] == ref[trusted]).all(), "u / w filter trusted a wrong sign"
        trusted_total += int(trusted.sum())
    return trusted_total / (3.0 * len(i))


def test_band2_random_and_adversarial():
    rng = np.random.default_rng(2026)
    n = 3000
    src = rng.uniform(size=(n, 3))
    Rm = np.linalg.qr(rng.normal(size=(3, 3)))[0]
    dst = src @ Rm.T + rng.uniform(-1, 1, size=3)
    out = rng.uniform(size=n) < 0.95
    dst[out] = rng.uniform(-1, 1, size=(int(out.sum()), 3))
    dst[~out] += rng.uniform(-0.0057, 0.0057, size=(int((~out).sum()), 3))
    frac = check2(src, dst, 0.02, rng, 1_000_000)
    assert frac > 0.9995  # the filter decides almost everything (fewer pairs in the band than the first formulation)
    # pairs engineered onto the boundary: dst lengths = src lengths +- beta (1 + delta)
    for scale, beta in ((1.0, 0.02), (300.0, 0.1), (0.05, 2e-4)):
        n = 1500
        src = rng.uniform(-1, 1, size=(n, 3)) * scale
        Rm = np.linalg.qr(rng.normal(size=(3, 3)))[0]
        dst = src @ Rm.T
        off = rng.choice([0.0, 1.0, -1.0], size=n) * beta * (
            1 + rng.choice([0, 1e-15, 1e-12, 1e-9, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3], size=n))
        d = dst / np.linalg.norm(dst, axis=1, keepdims=True)
        dst = dst + d * (off * rng.uniform(0.3, 1.0, size=n))[:, None]
        check2(src, dst, beta, rng, 600_000)
    # large offsets are absorbed by the centring
    src = rng.uniform(size=(1000, 3)) + np.array([1e4, -2e4, 3e4])
    dst = src @ Rm.T + 50.0
    check2(src, dst, 0.02, rng, 300_000)


def test_band2_short_pairs_are_never_trusted():
    """S = |a| + |b| <= beta is the one region where the sign of d misleads: every such pair must land inside the
    band (K0 >= 4 beta^4 + margins).  Clusters of near-coincident correspondences at several beta / size ratios."""
    rng = np.random.default_rng(7)
    for scale, beta in ((1.0, 0.02), (1.0, 0.2), (10.0, 0.05), (0.05, 2e-4)):
        n = 1200
        centres = rng.uniform(-1, 1, size=(40, 3)) * scale
        which = rng.integers(0, 40, size=n)
        src = centres[which] + rng.uniform(-1, 1, size=(n, 3)) * beta * rng.choice([0.05, 0.3, 0.6], size=(n, 1))
        Rm = np.linalg.qr(rng.normal(size=(3, 3)))[0]
        dst = src @ Rm.T + rng.uniform(-1, 1, size=(n, 3)) * beta * rng.choice([0.0, 0.05, 0.3], size=(n, 1))
        A, B, r2, g, kexp = operands2(src, dst, beta)
        kc = consts2(beta, r2)
        i = rng.integers(0, n, size=400_000)
        j = rng.integers(0, n, size=400_000)
        keep = (i != j) & (which[i] == which[j])
        i, j = i[keep], j[keep]
        a = np.linalg.norm(src[j] - src[i], axis=1)
        b = np.linalg.norm(dst[j] - dst[i], axis=1)
        short = a + b <= beta * (1 + 1e-9)
        assert short.sum() > 1000
        fma = lambda x, y, z: (np.asarray(x, np.float64) * np.asarray(y, np.float64) + np.asarray(z, np.float64)).astype(np.float32)
        u = accumulate(A[i], B[j], list(range(48)))
        w = accumulate(A[i], B[j], list(range(48, 64)))
        d = fma(u, u, w)
        nb = fma(w, kc["K2"], -kc["K0"])
        trusted = np.signbit((d + nb).astype(np.float32)) == np.signbit((d - nb).astype(np.float32))
        assert not trusted[short].any()
        ref = reference_edge(src, dst, i, j, beta)
        assert (np.signbit((d - nb).astype(np.float32))[trusted] == ref[trusted]).all()


# ---------------------------------------------------------------------------------------------
# third formulation (tim_graph_mfma3_kernel): same operands and accumulation, CONSTANT band C per problem; the
# provisional bit is sign(d~), a pair is trusted iff |d~| > C (in the kernel: iff the minimum of |d~| over the 16
# pairs of its lane-tile exceeds C -- a subset of this condition)
# ---------------------------------------------------------------------------------------------
def consts3(beta, r2max):
    f = np.float32
    g, kexp = scale2(beta)
    up = f(1.000001)
    b = f(beta * g) * up
    kappa = f(2.0 ** kexp)
    R2 = f(r2max) * up
    R = np.sqrt(R2) * up
    b2 = f(0.25) * kappa
    eps_u = K_EPS_U2 * U * R2 * up
    eps_w = kappa * (K_EPS_A2 * U * R2 * up) * up
    U0 = (f(4) * b * R * f(1.001) + f(2) * eps_u) * up
    G = (f(1.3e-13) * b * R2 * R + f(8e-15) * b2 * R2) * up
    E = (f(2) * U0 * eps_u + eps_u * eps_u + eps_w + G) * up
    C0 = E / (f(1) - f(4) * U) * f(1.001) * up
    short_d = (f(4) * b2 * b2 * (f(1) + f(16) * U) + f(4) * b2 * eps_u + eps_u * eps_u + eps_w) * f(1.001) * up
    return dict(C=max(C0, short_d) * f(1.00001), C0=C0, short_d=short_d)


def check3(src, dst, beta, rng, npairs, pairs=None):
    n = len(src)
    A, B, r2, g, kexp = operands2(src, dst, beta)
    C = consts3(beta, r2)["C"]
    if pairs is None:
        i = rng.integers(0, n, size=npairs)
        j = rng.integers(0, n, size=npairs)
    else:
        i, j = pairs
    keep = i != j
    i, j = i[keep], j[keep]
    ref = reference_edge(src, dst, i, j, beta)
    fma = lambda x, y, z: (np.asarray(x, np.float64) * np.asarray(y, np.float64) + np.asarray(z, np.float64)).astype(np.float32)
    trusted_total = 0
    for order_u, order_w in ((list(range(48)), list(range(48, 64))),
                             (list(range(47, -1, -1)), list(range(63, 47, -1))),
                             (list(rng.permutation(48)), list(48 + rng.permutation(16)))):
        u = accumulate(A[i], B[j], order_u)
        w = accumulate(A[i], B[j], order_w)
        d = fma(u, u, w)
        trusted = np.abs(d) > C
        assert (np.signbit(d)[trusted] == ref[trusted]).all(), "min |d| filter trusted a wrong sign"
        trusted_total += int(trusted.sum())
    return trusted_total / (3.0 * len(i)), ref, (i, j)


def test_band3_random_and_adversarial():
    rng = np.random.default_rng(2027)
    n = 3000
    src = rng.uniform(size=(n, 3))
    Rm = np.linalg.qr(rng.normal(size=(3, 3)))[0]
    dst = src @ Rm.T + rng.uniform(-1, 1, size=3)
    out = rng.uniform(size=n) < 0.95
    dst[out] = rng.uniform(-1, 1, size=(int(out.sum()), 3))
    dst[~out] += rng.uniform(-0.0057, 0.0057, size=(int((~out).sum()), 3))
    frac, _, _ = check3(src, dst, 0.02, rng, 1_000_000)
    # constant band: here (dst outliers spread over [-1, 1]^3, R = sqrt 3) 1.3e-3 of the pairs fall inside it, about
    # four times the w-dependent band's share; ~7e-4 at the bench geometry
    assert frac > 0.998
    for scale, beta in ((1.0, 0.02), (300.0, 0.1), (0.05, 2e-4)):
        n = 1500
        src = rng.uniform(-1, 1, size=(n, 3)) * scale
        Rm = np.linalg.qr(rng.normal(size=(3, 3)))[0]
        dst = src @ Rm.T
        off = rng.choice([0.0, 1.0, -1.0], size=n) * beta * (
            1 + rng.choice([0, 1e-15, 1e-12, 1e-9, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3], size=n))
        d = dst / np.linalg.norm(dst, axis=1, keepdims=True)
        dst = dst + d * (off * rng.uniform(0.3, 1.0, size=n))[:, None]
        check3(src, dst, beta, rng, 600_000)
    # large offsets are absorbed by the centring; far pairs (|u*| > U0, case B of the derivation) dominate here
    src = rng.uniform(size=(1000, 3)) + np.array([1e4, -2e4, 3e4])
    dst = src @ Rm.T + 50.0
    check3(src, dst, 0.02, rng, 300_000)
    # clouds of very different extent (|u*| far beyond U0 for most pairs)
    src = rng.uniform(-1, 1, size=(1500, 3))
    dst = rng.uniform(-1, 1, size=(1500, 3)) * 0.05
    check3(src, dst, 0.02, rng, 300_000)
    check3(dst, src, 0.02, rng, 300_000)


def test_band3_short_pairs_are_never_trusted():
    rng = np.random.default_rng(8)
    for scale, beta in ((1.0, 0.02), (1.0, 0.2), (10.0, 0.05), (0.05, 2e-4)):
        n = 1200
        centres = rng.uniform(-1, 1, size=(40, 3)) * scale
        which = rng.integers(0, 40, size=n)
        src = centres[which] + rng.uniform(-1, 1, size=(n, 3)) * beta * rng.choice([0.05, 0.3, 0.6], size=(n, 1))
        Rm = np.linalg.qr(rng.normal(size=(3, 3)))[0]
        dst = src @ Rm.T + rng.uniform(-1, 1, size=(n, 3)) * beta * rng.choice([0.0, 0.05, 0.3], size=(n, 1))
        A, B, r2, g, kexp = operands2(src, dst, beta)
        C = consts3(beta, r2)["C"]
        i = rng.integers(0, n, size=400_000)
        j = rng.integers(0, n, size=400_000)
        keep = (i != j) & (which[i] == which[j])
        i, j = i[keep], j[keep]
        a = np.linalg.norm(src[j] - src[i], axis=1)
        b = np.linalg.norm(dst[j] - dst[i], axis=1)
        short = a + b <= beta * (1 + 1e-9)
        assert short.sum() > 1000
        fma = lambda x, y, z: (np.asarray(x, np.float64) * np.asarray(y, np.float64) + np.asarray(z, np.float64)).astype(np.float32)
        u = accumulate(A[i], B[j], list(range(48)))
        w = accumulate(A[i], B[j], list(range(48, 64)))
        d = fma(u, u, w)
        trusted = np.abs(d) > C
        assert not trusted[short].any()
        ref = reference_edge(src, dst, i, j, beta)
        assert (np.signbit(d)[trusted] == ref[trusted]).all()
