"""Seeded synthetic inputs for the BASELINE.json configurations.

Columns are simulated down the tree with the standard Markov-chain recipe (root state by inverse CDF
from pi, each child by inverse CDF from its parent's row of P(t)), written fresh with numpy's PCG64
so the same seed gives the same data here and on the GPU box.  Weights are all 1 (the reference's
`P` pattern format, treesub.c:954-963).
"""
from __future__ import annotations

import numpy as np

from . import models
from .problem import (EIGEN_CIJK, EIGEN_UVROOT, MODE_LFUN, MODE_LFUNDG, Problem, Tree, balanced_tree)


def simulate_tips(tree: Tree, pi: np.ndarray, pmat_of_node, n_patt: int, seed: int, chunk: int = 1 << 18):
    """uint8 [n_tips, n_patt] tip states.  pmat_of_node(node) -> P(t) (n x n) for the branch above node."""
    rng = np.random.default_rng(seed)
    n = len(pi)
    z = np.zeros((tree.n_tips, n_patt), dtype=np.uint8)
    cdf_pi = np.cumsum(pi)
    cdf_pi[-1] = 1.0
    cdfs = {}
    order = []

    def pre(i):
        for c in tree.sons[i]:
            P = pmat_of_node(c)
            cd = np.cumsum(P, axis=1)
            cd[:, -1] = 1.0
            cdfs[c] = cd
            order.append((i, c))
            pre(c)
    pre(tree.root)
    for lo in range(0, n_patt, chunk):
        hi = min(n_patt, lo + chunk)
        m = hi - lo
        state = {tree.root: np.minimum(np.searchsorted(cdf_pi, rng.random(m), side="right"), n - 1)}
        for parent, child in order:
            u = rng.random(m)
            cd = cdfs[child][state[parent]]            # [m, n]
            s = np.minimum((u[:, None] >= cd).sum(axis=1), n - 1)
            state[child] = s
            if child < tree.n_tips:
                z[child, lo:hi] = s
        if tree.root < tree.n_tips:
            z[tree.root, lo:hi] = state[tree.root]
    return z


def f3x4_from_codon_tips(z: np.ndarray, weights: np.ndarray) -> np.ndarray:
    """3x4 position-specific base frequencies counted over all sequences (clean data), pattern-weighted
    — the F3x4 estimator of InitializeCodon (codeml.c:3772-3873)."""
    from61 = np.array(models.sense_codons())
    c = from61[z]                                    # [n_tips, n_patt] codon index 0..63
    fb = np.zeros((3, 4))
    for pos, b in enumerate((c // 16, (c // 4) % 4, c % 4)):
        for k in range(4):
            fb[pos, k] = ((b == k) * weights[None, :]).sum()
    return fb / fb.sum(axis=1, keepdims=True)


FB3X4_DEFAULT = np.array([[0.20, 0.25, 0.30, 0.25], [0.30, 0.20, 0.30, 0.20], [0.25, 0.30, 0.20, 0.25]])


def codon_m0_problem(n_tips=16, n_patt=1000, kappa=2.0, omega=0.4, seed=20260926, pi=None,
                     estimate_pi=False) -> Problem:
    """C4-shaped problem: codeml M0 (61 states), balanced-ish unrooted tree, tip branches 0.1,
    internal 0.05, evaluated at the generating parameters (lfun path, one class)."""
    tree = balanced_tree(n_tips)
    pi_gen = models.f3x4(FB3X4_DEFAULT) if pi is None else np.asarray(pi)
    U, V, root, _ = models.codon_m0_eigen(kappa, omega, pi_gen)
    z = simulate_tips(tree, pi_gen, lambda nd: np.clip(models.expm_rev(U, V, root, tree.branch[nd]), 0, None),
                      n_patt, seed)
    w = np.ones(n_patt)
    if estimate_pi:       # CodonFreq = 2 in the reference: F3x4 from the data
        pi_use = models.f3x4(f3x4_from_codon_tips(z, w))
        U, V, root, _ = models.codon_m0_eigen(kappa, omega, pi_use)
    else:
        pi_use = pi_gen
    return Problem(n=61, tree=tree, z=z, weights=w, pi=pi_use,
                   eigen=[dict(kind=EIGEN_UVROOT, U=U, V=V, Root=root)], mode=MODE_LFUN)


def with_ambiguous_codons(base: Problem, missing_rate=0.015, partial_rate=0.0005, sizes=(2, 2, 2, 3, 4, 4, 4, 4, 4, 16, 16), seed=7) -> Problem:
    """`base` (clean codon data) as a cleandata = 0 alignment: the 61 sense codons, then one code per distinct ambiguous triplet the way
    SetMapAmbiguity numbers them (treesub.c:1218-1286) — here len(sizes) partially resolved triplets (state sets of 2 .. 16 codons, each
    in `partial_rate` of the cells) and, LAST, the fully missing one (every codon, `missing_rate` of the cells): 61 + 12 = 73 codes."""
    import dataclasses
    rng = np.random.default_rng(seed)
    n = base.n
    n_codes = n + len(sizes) + 1
    n_chara = np.ones(n_codes, dtype=np.int32)
    cmap = np.zeros((n_codes, n), dtype=np.uint8)
    cmap[:n, 0] = np.arange(n)
    for k, sz in enumerate(sizes):
        first = int(rng.integers(0, n - sz + 1))
        n_chara[n + k] = sz
        cmap[n + k, :sz] = np.arange(first, first + sz)      # (neighbouring codons: what an unresolved third or second position gives)
    n_chara[n_codes - 1] = n
    cmap[n_codes - 1] = np.arange(n)
    z = base.z.copy()
    u = rng.random(z.shape)
    z[u < missing_rate] = n_codes - 1
    lo = missing_rate
    for k in range(len(sizes)):
        z[(u >= lo) & (u < lo + partial_rate)] = n + k
        lo += partial_rate
    return dataclasses.replace(base, z=z, cleandata=0, n_chara=n_chara, chara_map=cmap, eigen_of=None, qfactor=None)


def codon_nssites_problem(base: Problem, kappa: float, omegas, freqs) -> Problem:
    """Same data/tree as `base`, K omega classes sharing one Qfactor_NS scale (codeml.c:2590-2600,
    treesub.c:7675-7685): class ir uses U,V,Root of Q(omega_ir) with Root / (1/Qfactor_NS)."""
    import copy
    pi = base.pi[0]
    omegas = np.asarray(omegas, dtype=float)
    freqs = np.asarray(freqs, dtype=float)
    # Qfactor_NS = 1 / mr(Q at the mean omega)  (codeml.c:2586-2605)
    qfactor_ns = 1.0 / models.codon_q(kappa, float(np.dot(freqs, omegas)), pi)[1]
    eig = []
    for w in omegas:
        U, V, root, _ = models.codon_m0_eigen(kappa, w, pi, scale=1.0 / qfactor_ns)
        eig.append(dict(kind=EIGEN_UVROOT, U=U, V=V, Root=root))
    p = copy.copy(base)
    K = len(omegas)
    p.eigen = eig
    p.mode = MODE_LFUNDG
    p.freqK = freqs.copy()
    p.rate = np.ones(K)
    p.eigen_of = np.arange(K, dtype=np.int32).reshape(1, K, 1)
    p.qfactor = np.ones((K, 1))
    return p


GTR_RATES = (1.1, 0.4, 0.3, 0.5, 0.45)   # TC, TA, TG, CA, CG relative to AG = 1


def nuc_gtr_gamma_problem(n_tips=32, n_patt=1000, alpha=0.5, K=4, seed=20260927, estimate_pi=True) -> Problem:
    """C2-shaped problem: baseml GTR(REV)+Gamma_K on a balanced-ish tree; P(t) through Cijk
    (baseml.c:1572) exactly as baseml's REV path does."""
    tree = balanced_tree(n_tips)
    pi_gen = np.array([0.30, 0.22, 0.28, 0.20])
    Q = models.gtr_q(GTR_RATES, pi_gen)
    U, V, root = models.eigen_rev(Q, pi_gen)
    freqK, rK = models.discrete_gamma(alpha, K)
    rng = np.random.default_rng(seed + 1)
    # per-site rate classes: simulate each class's share of columns with its own rate
    cls = rng.integers(0, K, size=n_patt)
    z = np.zeros((n_tips, n_patt), dtype=np.uint8)
    for k in range(K):
        idx = np.nonzero(cls == k)[0]
        if len(idx) == 0:
            continue
        zk = simulate_tips(tree, pi_gen, lambda nd: np.clip(models.expm_rev(U, V, root, tree.branch[nd] * rK[k]), 0, None),
                           len(idx), seed + 17 * (k + 1))
        z[:, idx] = zk
    w = np.ones(n_patt)
    if estimate_pi:   # InitializeBaseAA (treesub.c:1548): observed base frequencies
        cnt = np.array([((z == b) * w[None, :]).sum() for b in range(4)])
        pi_use = cnt / cnt.sum()
    else:
        pi_use = pi_gen
    Q = models.gtr_q(GTR_RATES, pi_use)
    U, V, root = models.eigen_rev(Q, pi_use)
    Cijk, rootc, nR = models.cijk_from_uvroot(U, V, root)
    return Problem(n=4, tree=tree, z=z, weights=w, pi=pi_use,
                   eigen=[dict(kind=EIGEN_CIJK, Cijk=Cijk, Root=rootc, nR=nR)], mode=MODE_LFUNDG,
                   freqK=freqK, rate=rK)


def aa_model_tables(seed=20260928):
    """Exchangeabilities S (symmetric, zero diagonal) and frequencies pi of the synthetic 20-state model, rounded to 8 decimals so that
    a rate file written with 8 decimals (write_aa_ratefile) gives the reference exactly these numbers."""
    rng = np.random.default_rng(seed)
    pi = np.round(rng.dirichlet(np.full(20, 5.0)), 8)
    pi[-1] = np.round(1 - pi[:-1].sum(), 8)
    S = np.round(np.triu(rng.gamma(0.8, 1.0, size=(20, 20)) + 0.01, 1), 8)
    return S + S.T, pi


def write_aa_ratefile(path: str, S: np.ndarray, pi: np.ndarray):
    """An amino-acid rate file in the layout of dat/*.dat (GetDaa codeml.c:3967-4009): the lower triangle of the exchangeabilities row by
    row, then the 20 frequencies; amino acids in the order ARNDCQEGHILKMFPSTWYV."""
    with open(path, "w") as f:
        for i in range(1, 20):
            f.write(" ".join("%.8f" % S[i, j] for j in range(i)) + "\n")
        f.write("\n" + " ".join("%.8f" % v for v in pi) + "\n")


def aa_gamma_problem(n_tips=32, n_patt=1000, alpha=0.5, K=4, seed=20260928) -> Problem:
    """C3-shaped problem at scale: codeml seqtype 2, model 2 (an amino-acid rate file) + Gamma_K on a balanced-ish tree; P(t) through
    U, V, Root as for the empirical amino-acid models (codeml.c:4001-4009).  The exchangeabilities are seeded gamma draws and the
    frequencies a seeded Dirichlet draw — no empirical matrix is shipped in the package — and the columns are simulated down the tree
    class by class like the other generators."""
    tree = balanced_tree(n_tips)
    S, pi = aa_model_tables(seed)
    U, V, root = models.aa_empirical_eigen(S, pi)
    freqK, rK = models.discrete_gamma(alpha, K)
    rng = np.random.default_rng(seed + 1)
    cls = rng.integers(0, K, size=n_patt)
    z = np.zeros((n_tips, n_patt), dtype=np.uint8)
    for k in range(K):
        idx = np.nonzero(cls == k)[0]
        if len(idx):
            z[:, idx] = simulate_tips(tree, pi, lambda nd: np.clip(models.expm_rev(U, V, root, tree.branch[nd] * rK[k]), 0, None),
                                      len(idx), seed + 17 * (k + 1))
    return Problem(n=20, tree=tree, z=z, weights=np.ones(n_patt), pi=pi, eigen=[dict(kind=EIGEN_UVROOT, U=U, V=V, Root=root)],
                   mode=MODE_LFUNDG, freqK=freqK, rate=rK)


def write_pattern_file(path: str, z: np.ndarray, weights: np.ndarray, seqtype: str):
    """The reference's `P` (pre-compressed patterns) sequence format (treesub.c:549, 951-983)."""
    n_tips, n_patt = z.shape
    if seqtype == "codon":
        from61 = models.sense_codons()
        trip = ["".join(models.BASES[(c >> s) & 3] for s in (4, 2, 0)) for c in range(64)]
        table = np.array([trip[c] for c in from61])
        nchar = 3 * n_patt
    elif seqtype == "nuc":
        table = np.array(list(models.BASES))
        nchar = n_patt
    else:
        table = np.array(list(models.AAS))
        nchar = n_patt
    with open(path, "w") as f:
        f.write("%d %d P\n" % (n_tips, nchar))
        for i in range(n_tips):
            f.write("t%d  %s\n" % (i + 1, "".join(table[z[i]])))
        ws = weights.astype(int)
        for lo in range(0, n_patt, 4000):
            f.write(" ".join(map(str, ws[lo:lo + 4000])) + "\n")
