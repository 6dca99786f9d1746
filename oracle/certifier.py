"""CPU restatement of the reference's DRS rotation certifier (teaser/src/certification.cc) -- TEST
INFRASTRUCTURE ONLY (the checker for a future GPU certifier; nothing in the product imports it).

numpy, FP64.  Each function cites the reference lines it follows; the dense / sparse distinction of the
reference (Eigen::SparseMatrix for A_inv and lambda_guess) is dropped: values are identical, only storage
differs.  Pinned by tests/test_certifier_oracle.py to the reference's own fixtures
(test/teaser/data/certification_{small,large}_instances, committed as tests/golden/certifier_golden.npz by
tests/golden/make_certifier_golden.py): omega, block_diag_omega, Q_cost, lambda_guess, A_inv, W_dual and
M_affine of the first iteration, the sub-optimality of the first iteration and the whole sub-optimality
trajectories of certify(), at the reference tests' own tolerance (1e-7, certification-test.cc:29).
"""
import numpy as np

# coefficient matrix that maps vec(q q^T) to vec(R)  (certification.cc:241-252)
_P = np.array([
    [1, 0, 0, 0, 0, -1, 0, 0, 0, 0, -1, 0, 0, 0, 0, 1],
    [0, 1, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 1, 0],
    [0, 0, 1, 0, 0, 0, 0, -1, 1, 0, 0, 0, 0, -1, 0, 0],
    [0, 1, 0, 0, 1, 0, 0, 0, 0, 0, 0, -1, 0, 0, -1, 0],
    [-1, 0, 0, 0, 0, 1, 0, 0, 0, 0, -1, 0, 0, 0, 0, 1],
    [0, 0, 0, 1, 0, 0, 1, 0, 0, 1, 0, 0, 1, 0, 0, 0],
    [0, 0, 1, 0, 0, 0, 0, 1, 1, 0, 0, 0, 0, 1, 0, 0],
    [0, 0, 0, -1, 0, 0, 1, 0, 0, 1, 0, 0, -1, 0, 0, 0],
    [-1, 0, 0, 0, 0, -1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1]], dtype=np.float64)


def hatmap(v):
    """teaser::hatmap (linalg.h): the cross-product matrix."""
    return np.array([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]])


def rotation_to_quaternion(R):
    """Eigen::Quaterniond(R) followed by normalize() (certification.cc:67-68): Eigen's branchy conversion
    (Quaternion.h, quaternionbase_assign_impl for a 3x3 matrix).  Returns (x, y, z, w)."""
    R = np.asarray(R, dtype=np.float64)
    t = R[0, 0] + R[1, 1] + R[2, 2]
    q = np.zeros(4)  # x y z w
    if t > 0:
        t = np.sqrt(t + 1.0)
        q[3] = 0.5 * t
        t = 0.5 / t
        q[0] = (R[2, 1] - R[1, 2]) * t
        q[1] = (R[0, 2] - R[2, 0]) * t
        q[2] = (R[1, 0] - R[0, 1]) * t
    else:
        i = 0
        if R[1, 1] > R[0, 0]:
            i = 1
        if R[2, 2] > R[i, i]:
            i = 2
        j, k = (i + 1) % 3, (i + 2) % 3
        t = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
        q[i] = 0.5 * t
        t = 0.5 / t
        q[3] = (R[k, j] - R[j, k]) * t
        q[j] = (R[j, i] + R[i, j]) * t
        q[k] = (R[k, i] + R[i, k]) * t
    return q / np.linalg.norm(q)


def omega1(q):
    """getOmega1 (certification.cc:301-310); q = (x, y, z, w)."""
    x, y, z, w = q
    return np.array([[w, -z, y, x], [z, w, -x, y], [-y, x, w, z], [-x, -y, -z, w]], dtype=np.float64)


def block_diag_omega(npm, q):
    """getBlockDiagOmega (certification.cc:312-321)."""
    D = np.zeros((npm, npm))
    O = omega1(q)
    for i in range(npm // 4):
        D[4 * i:4 * i + 4, 4 * i:4 * i + 4] = O
    return D


def q_cost(v1, v2, noise_bound, cbar2):
    """getQCost (certification.cc:233-299); v1, v2 are 3 x N."""
    N = v1.shape[1]
    npm = 4 + 4 * N
    nbs = cbar2 * noise_bound ** 2
    Q1 = np.zeros((npm, npm))
    Q2 = np.zeros((npm, npm))
    for k in range(N):
        s = 4 * k + 4
        A = np.outer(v2[:, k], v1[:, k])
        Pk = (_P.T @ A.reshape(9, order="F")).reshape(4, 4, order="F")  # column-major maps, as Eigen::Map
        nn = v1[:, k] @ v1[:, k] + v2[:, k] @ v2[:, k]
        ck = 0.5 * (nn - nbs)
        Q1[0:4, s:s + 4] += -0.5 * Pk + ck / 2 * np.eye(4)
        Q1[s:s + 4, 0:4] += -0.5 * Pk + ck / 2 * np.eye(4)
        ck2 = 0.5 * (nn + nbs)
        Q2[s:s + 4, s:s + 4] += -Pk + ck2 * np.eye(4)
    return Q1 + Q2


def lambda_guess(R, theta, src, dst, noise_bound, cbar2):
    """getLambdaGuess (certification.cc:454-536), dense."""
    K = theta.shape[0]
    npm = 4 * K + 4
    nbs = cbar2 * noise_bound ** 2
    L = np.zeros((npm, npm))
    top = np.zeros((4, 4))
    I3 = np.eye(3)
    for i in range(K):
        s = src[:, i]
        sh = hatmap(s)
        xi = R.T @ (dst[:, i] - R @ s)
        xh = hatmap(xi)
        cur = np.zeros((4, 4))
        n2 = xi @ xi
        if theta[i] > 0:
            cur[3, 3] = -0.75 * n2 - 0.25 * nbs
            cur[:3, :3] = (sh @ sh - 0.5 * (s @ xi) * I3 + 0.5 * xh @ sh + 0.5 * np.outer(xi, s)
                           - 0.75 * n2 * I3 - 0.25 * nbs * I3)
            cur[:3, 3] = -1.5 * xh @ s
        else:
            cur[3, 3] = -0.25 * n2 - 0.75 * nbs
            cur[:3, :3] = (sh @ sh - 0.5 * (s @ xi) * I3 + 0.5 * xh @ sh + 0.5 * np.outer(xi, s)
                           - 0.25 * n2 * I3 - 0.25 * nbs * I3)
            cur[:3, 3] = -0.5 * xh @ s
        cur[3, :3] = cur[:3, 3]
        L[4 * i + 4:4 * i + 8, 4 * i + 4:4 * i + 8] = -cur
        top += cur
    L[:4, :4] += top
    return L


def linear_projection(theta_prepended):
    """getLinearProjection (certification.cc:538-657): the inverse map A_inv, dense nr_vals x nr_vals."""
    th = np.asarray(theta_prepended, dtype=np.float64)
    N0 = th.shape[0] - 1
    y = 1.0 / (2 * float(N0) + 6)
    x = (float(N0) + 1.0) * y
    N = N0 + 1
    nr = N * (N - 1) // 2
    m2v = np.zeros((N, N), dtype=np.int64)
    c = 0
    for i in range(N - 1):
        for j in range(i + 1, N):
            m2v[i, j] = c
            c += 1
    A = np.zeros((nr, nr))
    for i in range(N - 1):
        for j in range(i + 1, N):
            col = m2v[i, j]
            for p in range(N):
                if p != j and p != i:
                    if p < i:
                        A[m2v[p, i], col] += y * th[j] * th[p]
                    else:
                        A[m2v[i, p], col] += -y * th[j] * th[p]
            for p in range(N):
                if p != i and p != j:
                    if p < j:
                        A[m2v[p, j], col] += -y * th[i] * th[p]
                    else:
                        A[m2v[j, p], col] += y * th[i] * th[p]
            A[col, col] += x
    return A


def optimal_dual_projection(W, theta_prepended, A_inv):
    """getOptimalDualProjection (certification.cc:323-452)."""
    th = np.asarray(theta_prepended, dtype=np.float64)
    npm = W.shape[0]
    N = npm // 4 - 1
    nr = A_inv.shape[0]
    bW = np.zeros((nr, 3))
    c = 0
    for i in range(N):
        r0, r3 = 4 * i, 4 * i + 3
        for j in range(i + 1, N + 1):
            c0, c3 = 4 * j, 4 * j + 3
            tij = th[i] * th[j]
            C_, D_ = W[r3, r0:r0 + 3], W[c3, r0:r0 + 3]
            E_, F_ = W[r3, c0:c0 + 3], W[c3, c0:c0 + 3]
            bW[c] = (-tij * C_ + D_) + (-E_ + tij * F_)
            c += 1
    bWd = A_inv @ bW
    Wd = np.zeros((npm, npm))
    c = 0
    for i in range(N):
        r0 = 4 * i
        for j in range(i + 1, N + 1):
            c0 = 4 * j
            Wij = W[r0:r0 + 4, c0:c0 + 4]
            blk = (Wij - Wij.T) / 2
            blk[:3, 3] = bWd[c]
            blk[3, :3] = -bWd[c]
            Wd[r0:r0 + 4, c0:c0 + 4] = blk
            c += 1
    Wd = Wd + Wd.T
    vec = np.zeros(npm)
    vec[3::4] = th  # kron(theta, [0 0 0 1]^T)  (getBlockRowSum, certification.cc:659-671)
    diag_sum = np.zeros((3, 3))
    for i in range(N + 1):
        s = 4 * i
        rs = Wd[s:s + 4, :] @ vec
        Wii = W[s:s + 4, s:s + 4].copy()
        Wii[:, 3] = -th[i] * rs
        Wii[3, :] = -th[i] * rs
        Wd[s:s + 4, s:s + 4] = Wii
        diag_sum += Wii[:3, :3]
    mean = diag_sum / (N + 1)
    for i in range(N + 1):
        Wd[4 * i:4 * i + 3, 4 * i:4 * i + 3] -= mean
    return Wd


def nearest_psd(A):
    """teaser::getNearestPSD (linalg.h:85-99)."""
    B = (A + A.T) / 2
    w, V = np.linalg.eigh(B)
    w = np.where(w < 0, 0.0, w)
    return (V * w) @ V.T


def suboptimality_gap(M, mu, N):
    """computeSubOptimalityGap (certification.cc:192-231), EIGEN solver branch."""
    w = np.linalg.eigvalsh((M + M.T) / 2)
    m = w.min()
    if m > 0:
        return 0.0
    return (-m * (N + 1)) / mu


def certify(R, src, dst, theta, noise_bound=0.01, cbar2=1.0, sub_optimality=1e-3, max_iterations=200,
            gamma_tau=1.999999, return_first_iteration=False):
    """DRSCertifier::certify (certification.cc:39-190).  Returns dict(is_optimal, best_suboptimality,
    suboptimality_traj[, first-iteration intermediates])."""
    R = np.asarray(R, dtype=np.float64)
    src = np.asarray(src, dtype=np.float64)
    dst = np.asarray(dst, dtype=np.float64)
    theta = np.asarray(theta, dtype=np.float64).reshape(-1)
    N = src.shape[1]
    npm = 4 + 4 * N
    thp = np.concatenate([[1.0], theta])
    A_inv = linear_projection(thp)
    Q = q_cost(src, dst, noise_bound, cbar2)
    q = rotation_to_quaternion(R)
    x = np.kron(thp, q)  # vectorKron(theta_prepended, q)
    D = block_diag_omega(npm, q)
    Q_bar = D.T @ (Q @ D)
    J = np.zeros((npm, npm))
    J[:4, :4] = np.eye(4)
    mu = float(x @ (Q @ x))
    lam = lambda_guess(R, theta, src, dst, noise_bound, cbar2)
    M_init = Q_bar - mu * J - lam
    M = M_init.copy()
    traj = []
    best = np.inf
    first = None
    for it in range(int(max_iterations)):
        M_psd = nearest_psd(M)
        W = 2 * M_psd - M - M_init
        W_dual = optimal_dual_projection(W, thp, A_inv)
        M_aff = M_init + W_dual
        gap = suboptimality_gap(M_aff, mu, N)
        if it == 0 and return_first_iteration:
            first = dict(W=W, W_dual=W_dual, M_affine=M_aff, mu=mu, lambda_guess=lam, A_inv=A_inv, Q_cost=Q)
        traj.append(gap)
        best = min(best, gap)
        if gap < sub_optimality:
            break
        M = M + gamma_tau * (M_aff - M_psd)
    out = dict(is_optimal=bool(best < sub_optimality), best_suboptimality=float(best),
               suboptimality_traj=np.array(traj))
    if first is not None:
        out["first_iteration"] = first
    return out
