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
