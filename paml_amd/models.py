"""Host-side model set-up in numpy: builds the small inputs of the likelihood hot path
(pi, U/V/Root or Cijk, rate classes) the way the reference's model layer does.

This is the Python mirror of the C host library (paml_amd/host/); it exists so that bench.py,
the synthetic-data generator and the tests can build engine inputs without a C driver.  It is
*not* on the device path — the engine only ever sees the arrays produced here.

Reference behaviour restated (never copied): eigenQcodon codeml.c:3229-3321 (codon Q, mean-rate
scaling), eigenQREV tools.c:5023-5110 (sqrt(pi) symmetrisation, descending roots),
eigenQREVbase treesub.c:2488-2540 (GTR -> Cijk), DiscreteGamma tools.c:2601-2627,
F3x4 frequencies codeml.c:3772-3873, state orders tools.c:15-84.
"""
from __future__ import annotations

import numpy as np

BASES = "TCAG"   # tools.c:15
AAS = "ARNDCQEGHILKMFPSTWYV"  # tools.c:17
# standard code (icode 0), codon index = 16*b1 + 4*b2 + b3 with T,C,A,G = 0..3 (tools.c:23-84)
_STD_CODE = "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"


def sense_codons(code: str = _STD_CODE):
    """FROM61-style table: indices (0..63) of sense codons in order (treesub.c:2344-2349)."""
    return [i for i in range(64) if code[i] != "*"]


def codon_tables(code: str = _STD_CODE):
    from61 = sense_codons(code)
    aa = [code[i] for i in from61]
    return from61, aa


def f3x4(fb3x4: np.ndarray, code: str = _STD_CODE) -> np.ndarray:
    """Codon frequencies from 3x4 position-specific base frequencies, normalised over sense codons."""
    from61, _ = codon_tables(code)
    pi = np.array([fb3x4[0, c // 16] * fb3x4[1, (c // 4) % 4] * fb3x4[2, c % 4] for c in from61])
    return pi / pi.sum()


def codon_q(kappa: float, omega: float, pi: np.ndarray, code: str = _STD_CODE):
    """Q (n x n, rows sum to 0) and mean rate mr = -sum pi_i Q_ii, as eigenQcodon builds them
    (codeml.c:3274-3315; HKY-style kappa, one omega)."""
    from61, aa = codon_tables(code)
    n = len(from61)
    Q = np.zeros((n, n))
    for i in range(1, n):
        c1 = from61[i]
        f = (c1 // 16, (c1 // 4) % 4, c1 % 4)
        for j in range(i):
            c2 = from61[j]
            t = (c2 // 16, (c2 // 4) % 4, c2 % 4)
            diff = [k for k in range(3) if f[k] != t[k]]
            if len(diff) != 1:
                continue
            p = diff[0]
            q = kappa if (f[p] + t[p]) in (1, 5) else 1.0
            if aa[i] != aa[j]:
                q *= omega
            Q[i, j] = Q[j, i] = q
    Q = Q * pi[None, :]
    Q[np.diag_indices(n)] = -Q.sum(axis=1)
    mr = -float(np.dot(pi, np.diag(Q)))
    return Q, mr


def eigen_rev(Q: np.ndarray, pi: np.ndarray):
    """U, V, Root with Q = U diag(Root) V for a reversible Q (tools.c:5023-5110): symmetrise with
    sqrt(pi), symmetric eigen-solve, roots sorted descending (Root[0] ~ 0)."""
    sp = np.sqrt(pi)
    A = Q * sp[:, None] / sp[None, :]
    A = 0.5 * (A + A.T)
    w, R = np.linalg.eigh(A)
    order = np.argsort(-w)
    w, R = w[order], R[:, order]
    U = R / sp[:, None]
    V = R.T * sp[None, :]
    return np.ascontiguousarray(U), np.ascontiguousarray(V), np.ascontiguousarray(w)


def codon_m0_eigen(kappa: float, omega: float, pi: np.ndarray, scale: float | None = None):
    """U, V, Root for M0 with Root divided by the mean rate (codeml.c:3316-3321), or by `scale`
    (= 1/Qfactor_NS under NSsites, treesub.c:7675-7685)."""
    Q, mr = codon_q(kappa, omega, pi)
    U, V, root = eigen_rev(Q, pi)
    return U, V, root / (mr if scale is None else scale), mr


def gtr_q(rates5, pi: np.ndarray):
    """GTR Q in baseml's REV parametrisation: (TC, TA, TG, CA, CG) relative to AG = 1, state order
    T,C,A,G, scaled so the mean rate is 1 (treesub.c:2499-2518)."""
    a, b, c, d, e = rates5
    S = np.array([[0, a, b, c], [a, 0, d, e], [b, d, 0, 1.0], [c, e, 1.0, 0]])
    Q = S * pi[None, :]
    Q[np.diag_indices(4)] = -Q.sum(axis=1)
    mr = -float(np.dot(pi, np.diag(Q)))
    return Q / mr


def cijk_from_uvroot(U, V, root):
    """Cijk[i][j][k] = U[i,k] V[k,j] with nR = n (treesub.c:2526-2535)."""
    n = U.shape[0]
    C = np.einsum("ik,kj->ijk", U, V)
    return np.ascontiguousarray(C), np.ascontiguousarray(root), n


def hky_q(kappa: float, pi: np.ndarray):
    S = np.ones((4, 4))
    S[0, 1] = S[1, 0] = S[2, 3] = S[3, 2] = kappa
    np.fill_diagonal(S, 0)
    Q = S * pi[None, :]
    Q[np.diag_indices(4)] = -Q.sum(axis=1)
    mr = -float(np.dot(pi, np.diag(Q)))
    return Q / mr


def discrete_gamma(alpha: float, K: int):
    """Mean-of-category discrete gamma with beta = alpha (tools.c:2601-2627, UseMedian = 0)."""
    from scipy.special import gammainc, gammaincinv
    cuts = gammaincinv(alpha, np.arange(1, K) / K) / alpha          # quantiles of G(alpha, beta=alpha)
    cdf1 = gammainc(alpha + 1, cuts * alpha)                          # Eq. 10
    edges = np.concatenate(([0.0], cdf1, [1.0]))
    rK = np.diff(edges) * K
    return np.full(K, 1.0 / K), rK


def expm_rev(U, V, root, t):
    """P(t) = U exp(root t) V (plain numpy; generator use only)."""
    return (U * np.exp(root * t)[None, :]) @ V


def read_aa_ratefile(path: str):
    """Empirical amino-acid model file (dat/*.dat): lower-triangle exchangeabilities then 20 frequencies, amino acids in
    the order ARNDCQEGHILKMFPSTWYV (GetDaa codeml.c:3967-4009).  Returns (S symmetric 20x20, pi)."""
    toks = []
    with open(path) as f:
        for line in f:
            for t in line.split():
                try:
                    toks.append(float(t))
                except ValueError:
                    break
            if len(toks) >= 190 + 20:
                break
    S = np.zeros((20, 20))
    k = 0
    for i in range(20):
        for j in range(i):
            S[i, j] = S[j, i] = toks[k]
            k += 1
    pi = np.array(toks[190:210])
    return S, pi      # used as read, NOT renormalised — the reference only checks |1 - sum| < 1e-5 (codeml.c:4004-4008)


def aa_empirical_eigen(S: np.ndarray, pi: np.ndarray):
    """U, V, Root of the empirical aa model scaled to mean rate 1 (eigenQaa codeml.c:3400-3484)."""
    Q = S * pi[None, :]
    np.fill_diagonal(Q, 0.0)
    Q[np.diag_indices(20)] = -Q.sum(axis=1)
    mr = -float(np.dot(pi, np.diag(Q)))
    U, V, root = eigen_rev(Q, pi)
    return U, V, root / mr


def aa_code_map():
    """nChara / CharaMap for amino acids with cleandata = 0 (SetMapAmbiguity treesub.c:1218-1247): the 20 residues map
    to themselves, '-', '*', '?', 'X' to all 20 states.  Codes index the string AAs + '-*?X' (tools.c:19)."""
    n_codes = 24
    n_chara = np.ones(n_codes, dtype=np.int32)
    cmap = np.zeros((n_codes, 20), dtype=np.uint8)
    cmap[:20, 0] = np.arange(20)
    n_chara[20:] = 20
    cmap[20:] = np.arange(20)
    return n_chara, cmap


def nssites_classes(ns, par, ncatG):
    """freqK, omega per class for NSsites = 0, 1, 2, 3, 7, 8 at untransformed parameters (SetParametersNSsites
    codeml.c:2483-2578 with LASTROUND = 1; DiscreteNSsites codeml.c:2846: M7 / M8 take the beta quantiles at the K bin
    mid-points)."""
    par = [float(v) for v in par]
    if ns == 0:
        return np.ones(1), np.array([par[0]])
    if ns == 1:
        p0, w0 = par[:2]
        return np.array([p0, 1 - p0]), np.array([w0, 1.0])
    if ns == 2:
        p0, p1, w0, w2 = par[:4]
        return np.array([p0, p1, 1 - p0 - p1]), np.array([w0, 1.0, w2])
    if ns == 3:
        K = ncatG
        p = np.array(par[:K - 1])
        return np.concatenate((p, [1 - p.sum()])), np.array(par[K - 1:2 * K - 1])
    from scipy.special import betaincinv
    if ns == 7:
        p, q = par[:2]
        K = ncatG
        return np.full(K, 1.0 / K), betaincinv(p, q, (2 * np.arange(K) + 1) / (2.0 * K))
    if ns == 8:
        p0, p, q, ws = par[:4]
        K = ncatG
        w = betaincinv(p, q, (2 * np.arange(K) + 1) / (2.0 * K))
        return np.concatenate((np.full(K, p0 / K), [1 - p0])), np.concatenate((w, [ws]))
    raise ValueError(ns)
