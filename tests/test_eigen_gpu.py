"""The batched eigen-decomposition on the device (paml_amd_set_eigen_qrev_batch, eigen_kernels.h) against the host path it
replaces on the hot loop of small-data optimisation (eigenQREV tools.c:5023-5110 under eigenQcodon codeml.c:3229; here
numpy's LAPACK symmetric solver through models.eigen_rev): roots to 1e-12, U diag(Root) V = Q and U V = I, P(t) of an evaluation
to 1e-13, lnL unchanged to 1e-12 relative.  Eigenvectors themselves are not compared: they are unique only up to sign and to
rotations inside an eigenspace, and P(t) does not depend on the choice."""
import numpy as np
import pytest

import helpers
from paml_amd import models, synth
from paml_amd.engine import engine_for

pytestmark = pytest.mark.gpu


def random_f3x4(rng, zero=False):
    fb = rng.dirichlet(np.ones(4) * 4, size=3)
    if zero:                       # a nucleotide never seen at a position: 16 (minus stops) codons of frequency zero
        fb[2, 3] = 0
        fb[2] /= fb[2].sum()
    return models.f3x4(fb)


def test_codon_matrices_decomposed_on_the_device():
    rng = np.random.default_rng(7)
    pb = synth.codon_m0_problem(n_tips=6, n_patt=300)
    eng = engine_for(pb)
    cases = [(2.0, 0.4, False), (3.5, 1e-4, False), (1.0, 1.0, False), (8.0, 7.5, False), (2.0, 0.0, False), (2.5, 0.3, True), (0.7, 2.2, True)]
    cases += [(float(rng.uniform(0.5, 10)), float(rng.uniform(0, 3)), False) for _ in range(40)]
    Qs, pis, mrs = [], [], []
    for kappa, omega, zero in cases:
        pi = random_f3x4(rng, zero)
        Q, mr = models.codon_q(kappa, omega, pi)
        Qs.append(Q); pis.append(pi); mrs.append(mr)
    ids = np.arange(len(cases)) + 3            # (not from 0: the table grows, set 0 stays the problem's own)
    eng.set_eigen_qrev_batch(ids, np.array(Qs), np.array(pis), np.array(mrs))
    cnt = eng.eigen_counters()
    assert cnt["n_decomposed"] == len(cases) and len(cnt["sweeps"]) == len(cases)
    assert cnt["sweeps"].max() <= 13 and cnt["sweeps"].min() >= 2, cnt["sweeps"]
    for k, (Q, pi, mr) in enumerate(zip(Qs, pis, mrs)):
        U, V, R = eng.get_eigen(int(ids[k]))
        live = pi > 1e-100
        sp = np.sqrt(pi[live])
        A = (Q[np.ix_(live, live)] * sp[:, None] / sp[None, :])
        A = np.tril(A) + np.tril(A, -1).T
        w = np.sort(np.concatenate([np.linalg.eigvalsh(A), np.zeros((~live).sum())]))[::-1] / mr
        scale = np.abs(A).max() / mr
        assert np.all(np.diff(R) <= 0), "roots descending"
        assert np.max(np.abs(R - w)) <= 1e-12 * scale, (k, np.max(np.abs(R - w)))
        assert abs(R[0]) <= 1e-13 * scale                                   # the zero root first (what pmat_deriv_kernel relies on)
        Qs_ = Q.copy()
        Qs_[~live, :] = 0; Qs_[:, ~live] = 0                                # the chain never enters or leaves a left-out state
        assert np.max(np.abs(U @ np.diag(R) @ V - Qs_ / mr)) <= 2e-13 * scale, k
        assert np.max(np.abs(U @ V - np.eye(61))) <= 2e-13, k


@pytest.mark.parametrize("K", [1, 3])
def test_pmat_and_lnl_with_device_eigen_match_the_host_path(K):
    base = synth.codon_m0_problem(n_tips=10, n_patt=1500)
    kappa = 2.3
    omegas, freqs = ([0.4], [1.0]) if K == 1 else ([0.05, 1.0, 3.1], [0.6, 0.3, 0.1])
    pb = synth.codon_nssites_problem(base, kappa, omegas, freqs) if K > 1 else base
    if K == 1:
        U, V, root, mr = models.codon_m0_eigen(kappa, omegas[0], base.pi[0])
        pb.eigen = [dict(kind=helpers.EIGEN_UVROOT, U=U, V=V, Root=root)]
    eng = engine_for(pb)
    host = eng.eval(pb.tree.branch, want_lnf=True)
    node = pb.tree.n_tips + 1
    P_host = [eng.get_pmat(0, k, node) for k in range(K)] + [eng.get_pmat(0, k, 0) for k in range(K)]
    pi = base.pi[0]
    scale = models.codon_q(kappa, float(np.dot(freqs, omegas)), pi)[1]
    Qs = [models.codon_q(kappa, w, pi)[0] for w in omegas]
    eng.set_eigen_qrev_batch(np.arange(K), np.array(Qs), np.array([pi] * K), np.array([scale] * K))
    dev = eng.eval(pb.tree.branch, want_lnf=True)
    P_dev = [eng.get_pmat(0, k, node) for k in range(K)] + [eng.get_pmat(0, k, 0) for k in range(K)]
    for a, b in zip(P_host, P_dev):
        assert np.max(np.abs(a - b)) <= 1e-13
    assert abs(dev["lnL"] - host["lnL"]) <= 1e-12 * abs(host["lnL"])
    assert np.max(np.abs(dev["lnf"] - host["lnf"])) <= 1e-10
    # the branch-local derivatives read the same sets (plain exp, first root forced to zero)
    b = pb.tree.n_tips + 2
    ts = np.array([pb.tree.branch[b], 0.3])
    l1, d1, dd1 = eng.eval_branch(b, ts, pb.tree.branch)
    eng2 = engine_for(pb)
    l0, d0, dd0 = eng2.eval_branch(b, ts, pb.tree.branch)
    assert np.allclose(l1, l0, rtol=1e-12, atol=0) and np.allclose(d1, d0, rtol=1e-9, atol=1e-8) and np.allclose(dd1, dd0, rtol=1e-9, atol=1e-7)


@pytest.mark.parametrize("n", [4, 20, 33, 60])
def test_small_reversible_matrices(n):
    """n = 4 (GTR), n = 20 (a random reversible amino-acid matrix), n = 60 (the sense codons of a mitochondrial code) and an odd order in
    between: orders 20, 60 (and 61) run the kernel's register form (R^T in one wave's registers, the rounds unrolled), the others its
    any-order form."""
    rng = np.random.default_rng(n)
    pb = helpers.random_problem(n, 6, 200, K=1, seed=5)
    eng = engine_for(pb)
    m = 9
    Qs, pis = [], []
    for _ in range(m):
        pi = rng.dirichlet(np.ones(n) * 3)
        S = rng.uniform(0.05, 2.0, size=(n, n))
        S = np.tril(S, -1) + np.tril(S, -1).T
        Q = S * pi[None, :]
        Q[np.diag_indices(n)] = -Q.sum(axis=1)
        Qs.append(Q); pis.append(pi)
    eng.set_eigen_qrev_batch(np.arange(m) + 1, np.array(Qs), np.array(pis))
    for k in range(m):
        U, V, R = eng.get_eigen(k + 1)
        assert np.max(np.abs(U @ np.diag(R) @ V - Qs[k])) <= 1e-13 * np.abs(Qs[k]).max() * n
        assert np.max(np.abs(U @ V - np.eye(n))) <= 1e-13
        assert abs(R[0]) < 1e-13 and np.all(np.diff(R) <= 0)


@pytest.mark.parametrize("n", [4, 20, 33, 60, 61])
def test_degenerate_spectra_and_scales(n):
    """What a Jacobi iteration can trip over, in both forms of the kernel: one eigenvalue of multiplicity n - 1 (equal rates, equal
    frequencies) at three scales (x 1, 1e-150, 1e150: the angle formula's squares must neither underflow nor overflow), two groups of
    states that never exchange (a block-diagonal matrix: half of the off-diagonal elements are zero from the start), the zero matrix
    (converged before the first sweep)."""
    rng = np.random.default_rng(100 + n)
    pb = helpers.random_problem(n, 6, 200, K=1, seed=5)
    eng = engine_for(pb)
    flat = np.full(n, 1.0 / n)
    J = np.full((n, n), 1.0 / n)
    J[np.diag_indices(n)] = 0
    J[np.diag_indices(n)] = -J.sum(axis=1)
    h = n // 2
    S = rng.uniform(0.1, 2.0, size=(n, n))
    S = np.tril(S, -1) + np.tril(S, -1).T
    S[:h, h:] = 0; S[h:, :h] = 0
    pib = rng.dirichlet(np.ones(n) * 3)
    B = S * pib[None, :]
    B[np.diag_indices(n)] = -B.sum(axis=1)
    Qs = [J, J * 1e-150, J * 1e150, B, np.zeros((n, n))]
    pis = [flat, flat, flat, pib, flat]
    eng.set_eigen_qrev_batch(np.arange(len(Qs)) + 1, np.array(Qs), np.array(pis))
    sw = eng.eigen_counters()["sweeps"]
    assert sw.min() >= 0 and sw.max() <= 10 and sw[4] == 0, sw
    for k, Q in enumerate(Qs):
        U, V, R = eng.get_eigen(k + 1)
        scale = max(np.abs(Q).max(), 1e-300)
        assert np.all(np.diff(R) <= 0), k
        assert np.max(np.abs(U @ np.diag(R) @ V - Q)) <= 1e-13 * n * scale, (k, np.max(np.abs(U @ np.diag(R) @ V - Q)) / scale)
        assert np.max(np.abs(U @ V - np.eye(n))) <= 1e-13, k
    assert abs(eng.get_eigen(1)[2][0]) <= 1e-15 and np.allclose(eng.get_eigen(1)[2][1:], -1.0, rtol=0, atol=1e-13)      # roots 0, -1 (n - 1 times)


def test_warm_started_decompositions_along_an_optimisers_path():
    """paml_amd_set_eigen_warm_start: three sets (omega classes) decomposed again and again while kappa and the omegas move the way an
    optimiser moves them — finite-difference steps of 1e-6 relative, line-search steps of a few per cent.  From the second call on the
    sweeps start from the previous eigenvectors: fewer sweeps, the same accuracy as the cold start (roots against LAPACK, U diag(Root) V
    = Q, U V = I), through the periodic cold restarts and a change of the set of states with frequency zero; off by default."""
    rng = np.random.default_rng(3)
    pb = synth.codon_m0_problem(n_tips=6, n_patt=300)
    eng = engine_for(pb)
    assert eng.set_eigen_warm_start() == 0
    pi = random_f3x4(rng)
    kappa, om = 2.0, np.array([0.1, 1.0, 2.5])
    ids = np.array([1, 2, 3])

    def send(pi_):
        Qs, mrs = zip(*[models.codon_q(kappa, w, pi_) for w in om])
        eng.set_eigen_qrev_batch(ids, np.array(Qs), np.array([pi_] * 3), np.array(mrs))
        return Qs, mrs, eng.eigen_counters()["sweeps"].copy()

    def check(Qs, mrs, pi_):
        for k in range(3):
            U, V, R = eng.get_eigen(int(ids[k]))
            live = pi_ > 1e-100
            sp = np.sqrt(pi_[live])
            A = Qs[k][np.ix_(live, live)] * sp[:, None] / sp[None, :]
            A = np.tril(A) + np.tril(A, -1).T
            w = np.sort(np.concatenate([np.linalg.eigvalsh(A), np.zeros((~live).sum())]))[::-1] / mrs[k]
            scale = np.abs(A).max() / mrs[k]
            Qz = Qs[k].copy()
            Qz[~live, :] = 0; Qz[:, ~live] = 0
            assert np.max(np.abs(R - w)) <= 1e-12 * scale and np.all(np.diff(R) <= 0) and abs(R[0]) <= 1e-13 * scale
            assert np.max(np.abs(U @ np.diag(R) @ V - Qz / mrs[k])) <= 2e-13 * scale
            assert np.max(np.abs(U @ V - np.eye(61))) <= 2e-13

    Qs, mrs, cold = send(pi)
    check(Qs, mrs, pi)
    eng.set_eigen_warm_start(1)
    Qs, mrs, sw = send(pi)                               # first call after switching on: nothing to start from yet
    assert (sw == cold).all() and eng.set_eigen_warm_start() == 0
    small, large = [], []
    for it in range(70):
        fd = it % 5 != 4
        step = 1e-6 if fd else 0.05
        kappa *= 1 + step * rng.choice([-1, 1])
        om[[0, 2]] *= 1 + step * rng.choice([-1, 1], size=2)
        Qs, mrs, sw = send(pi)
        (small if fd else large).append(sw)
        if it % 7 == 0 or it > 60:
            check(Qs, mrs, pi)
    small, large = np.array(small), np.array(large)
    n_warm = eng.set_eigen_warm_start()
    assert 3 * 60 <= n_warm < 3 * 70                      # all but the periodic cold restarts
    # (the set with omega = 1 does not move on the omega steps: kappa moves it)
    assert np.median(small) <= 3 and np.median(large) <= 5 and small.min() >= 1, (np.median(small), np.median(large))
    assert (small >= cold.min()).sum() <= 3 * 5           # the cold restarts (every 16th decomposition of a set) are the only slow ones
    # a codon position loses a nucleotide: other states are left out now, the next decomposition starts cold and is right
    pi0 = random_f3x4(rng, zero=True)
    Qs, mrs, sw = send(pi0)
    assert eng.set_eigen_warm_start() == n_warm and (sw >= cold.min() - 2).all()
    check(Qs, mrs, pi0)
    Qs, mrs, sw = send(pi0)                               # the same matrices again, warm: converged at once
    assert eng.set_eigen_warm_start() == n_warm + 3 and sw.max() <= 2
    check(Qs, mrs, pi0)
    eng.set_eigen_warm_start(0)
    Qs, mrs, sw = send(pi0)
    assert eng.set_eigen_warm_start() == n_warm + 3 and (sw >= cold.min() - 2).all()


def test_a_gradients_perturbed_points_start_from_the_base_points_sets():
    """Round 6: a warm start comes from the NEAREST matrix any set was last decomposed for.  The base point's three omega classes are
    decomposed into sets 1-3; the perturbed points of a gradient (kappa or one omega moved by 1e-6 relative) go to sets the engine has
    never seen (4 ...) in one batch: each starts from the base point's set of the same class — two sweeps, the right answer — and the
    next iterate's base point (5 % away) from the nearest of them all."""
    rng = np.random.default_rng(9)
    pb = synth.codon_m0_problem(n_tips=6, n_patt=300)
    eng = engine_for(pb)
    pi = random_f3x4(rng)
    kappa, om = 2.0, np.array([0.1, 1.0, 2.5])

    def mats(k, w):
        return zip(*[models.codon_q(k, x, pi) for x in w])

    def check(ids, Qs, mrs):
        for sid, Q, mr in zip(ids, Qs, mrs):
            U, V, R = eng.get_eigen(int(sid))
            scale = np.abs(Q).max() / mr
            assert np.max(np.abs(U @ np.diag(R) @ V - Q / mr)) <= 2e-13 * scale and np.max(np.abs(U @ V - np.eye(61))) <= 2e-13

    eng.set_eigen_warm_start(1)
    Qs, mrs = mats(kappa, om)
    eng.set_eigen_qrev_batch(np.array([1, 2, 3]), np.array(Qs), np.array([pi] * 3), np.array(mrs))
    cold = eng.eigen_counters()["sweeps"].copy()
    assert cold.min() >= 6 and eng.set_eigen_warm_start() == 0
    pert = [(kappa * (1 + 1e-6), om)] + [(kappa, om * (1 + 1e-6 * (np.arange(3) == j))) for j in range(3)]
    allQ, allmr = [], []
    for k, w in pert:
        q, m = mats(k, w)
        allQ += q; allmr += m
    ids = np.arange(len(allQ)) + 4
    eng.set_eigen_qrev_batch(ids, np.array(allQ), np.array([pi] * len(allQ)), np.array(allmr))
    sw = eng.eigen_counters()["sweeps"]
    assert eng.set_eigen_warm_start() == len(allQ) and sw.max() <= 3, sw      # (the unmoved classes: the same matrix again, 0 sweeps)
    check(ids, allQ, allmr)
    Qn, mrn = mats(kappa * 1.05, om * 0.95)
    eng.set_eigen_qrev_batch(np.array([1, 2, 3]), np.array(Qn), np.array([pi] * 3), np.array(mrn))
    sw = eng.eigen_counters()["sweeps"]
    assert sw.max() < cold.min() and eng.set_eigen_warm_start() == len(allQ) + 3, (sw, cold)
    check([1, 2, 3], Qn, mrn)


def test_matrices_handed_over_as_their_elements():
    """paml_amd_set_eigen_qrev_batch_sparse: codon matrices as the 263 + 61 elements a single nucleotide change can put at or below the
    diagonal (what the C host sends: 2.6 KB per matrix instead of 30 KB) against the same matrices handed over whole: the same roots to
    1e-13 of the scale, U diag(Root) V = Q and U V = I; a state of frequency zero; an element above the diagonal is refused."""
    rng = np.random.default_rng(21)
    pb = synth.codon_m0_problem(n_tips=6, n_patt=300)
    eng = engine_for(pb)
    pis = [random_f3x4(rng), random_f3x4(rng, zero=True), random_f3x4(rng)]
    Qs, mrs = zip(*[models.codon_q(k, w, pi) for (k, w), pi in zip(((2.0, 0.4), (5.0, 0.01), (0.8, 3.0)), pis)])
    may = np.zeros((61, 61), dtype=bool)
    for Q in Qs:
        may |= Q != 0
    may |= may.T
    row, col = np.nonzero(np.tril(may))
    assert len(row) <= 263 + 61 and (row >= col).all()
    vals = np.array([Q[row, col] for Q in Qs])
    eng.set_eigen_qrev_batch(np.array([1, 2, 3]), np.array(Qs), np.array(pis), np.array(mrs))
    dense = [eng.get_eigen(k) for k in (1, 2, 3)]
    eng.set_eigen_qrev_batch_sparse(np.array([4, 5, 6]), row, col, vals, np.array(pis), np.array(mrs))
    for k in range(3):
        U, V, R = eng.get_eigen(4 + k)
        live = pis[k] > 1e-100
        Qz = Qs[k].copy()
        Qz[~live, :] = 0; Qz[:, ~live] = 0
        scale = np.abs(Qz).max() / mrs[k]
        assert np.max(np.abs(R - dense[k][2])) <= 1e-13 * scale and np.all(np.diff(R) <= 0)
        assert np.max(np.abs(U @ np.diag(R) @ V - Qz / mrs[k])) <= 2e-13 * scale and np.max(np.abs(U @ V - np.eye(61))) <= 2e-13
    with pytest.raises(Exception, match="lower triangle"):
        eng.set_eigen_qrev_batch_sparse(np.array([7]), col[:5], row[:5] + 1, vals[:1, :5], np.array(pis[:1]), np.array(mrs[:1]))
    with pytest.raises(Exception, match="appears twice"):
        eng.set_eigen_qrev_batch_sparse(np.array([7]), np.array([3, 3]), np.array([1, 1]), vals[:1, :2], np.array(pis[:1]), np.array(mrs[:1]))


def test_warm_starts_from_other_sets_stay_right_through_a_random_schedule():
    """The bookkeeping behind the nearest-matrix warm start (two eigenvector buffers per set, chains that restart cold, sets of states
    left out): 40 batches of random size over 14 set ids, matrices drawn from a small family (so that near and far neighbours, exact
    repeats and other zero patterns of pi all occur), dense and sparse calls mixed — every decomposition is checked."""
    rng = np.random.default_rng(33)
    pb = synth.codon_m0_problem(n_tips=6, n_patt=300)
    eng = engine_for(pb)
    pis = [random_f3x4(rng), random_f3x4(rng, zero=True)]
    eng.set_eigen_warm_start(1)
    may = np.zeros((61, 61), dtype=bool)
    for pi in pis:
        may |= models.codon_q(2.0, 0.5, pi)[0] != 0
    may |= may.T
    row, col = np.nonzero(np.tril(may))
    n_cold = 0
    for it in range(40):
        ids = rng.choice(np.arange(1, 15), size=int(rng.integers(1, 13)), replace=False)
        which = rng.integers(0, 2, size=len(ids)) * (it % 3 == 0)
        kap = 2.0 * (1 + 0.05 * rng.integers(-2, 3, size=len(ids))) * (1 + 1e-6 * rng.integers(-1, 2, size=len(ids)))
        om = 0.4 * (1 + 0.1 * rng.integers(-2, 3, size=len(ids)))
        Qs, mrs = zip(*[models.codon_q(k, w, pis[p_]) for k, w, p_ in zip(kap, om, which)])
        P = np.array([pis[p_] for p_ in which])
        if it % 2:
            eng.set_eigen_qrev_batch_sparse(ids, row, col, np.array([Q[row, col] for Q in Qs]), P, np.array(mrs))
        else:
            eng.set_eigen_qrev_batch(ids, np.array(Qs), P, np.array(mrs))
        sw = eng.eigen_counters()["sweeps"]
        assert sw.min() >= 0 and sw.max() <= 11, sw
        n_cold += int((sw >= 7).sum())
        for sid, Q, mr, p_ in zip(ids, Qs, mrs, which):
            U, V, R = eng.get_eigen(int(sid))
            live = pis[p_] > 1e-100
            Qz = Q.copy()
            Qz[~live, :] = 0; Qz[:, ~live] = 0
            scale = np.abs(Qz).max() / mr
            assert np.all(np.diff(R) <= 0) and abs(R[0]) <= 1e-13 * scale
            assert np.max(np.abs(U @ np.diag(R) @ V - Qz / mr)) <= 3e-13 * scale, (it, sid)
            assert np.max(np.abs(U @ V - np.eye(61))) <= 3e-13, (it, sid)
    assert 0 < n_cold < 120 and eng.set_eigen_warm_start() > 150      # most started warm, some chains restarted cold


NOCONV_SCRIPT = r"""
import json, os, sys
import torch  # noqa: F401
import numpy as np
sys.path.insert(0, %(repo)r)
sys.path.insert(0, os.path.join(%(repo)r, "tests"))
import helpers
from paml_amd import engine, hostlib, models, synth
pb = synth.codon_m0_problem(n_tips=6, n_patt=300)
eng = engine.engine_for(pb)
ok = eng.eval(pb.tree.branch)["lnL"]
pi = np.asarray(pb.pi, dtype=np.float64).reshape(-1)[:61]
Q, mr = models.codon_q(2.0, 0.4, pi)
eng.set_eigen_qrev_batch([0], np.array([Q]), np.array([pi]), np.array([mr]))      # one sweep allowed: not converged
try:
    eng.eval(pb.tree.branch)
    code = None
except engine.EngineError as ex:
    code = str(ex)
sweeps = eng.eigen_counters()["sweeps"].tolist()
again = None
try:      # the flag was consumed: with a host decomposition in the set's place the engine evaluates again
    eng.set_eigen(0, pb.eigen[0])
    again = eng.eval(pb.tree.branch)["lnL"]
except engine.EngineError as ex:
    again = str(ex)
# the other entry points that end with a host synchronisation report it too (round 6): node_posterior, eval_adg (a two-class
# mixture of the same model: lfundG mode), the BEB grid; and paml_amd_eigen_status for callers of eval_device, which never synchronises
others = {}
def attempt(name, f):
    eng.set_eigen_qrev_batch([0], np.array([Q]), np.array([pi]), np.array([mr]))
    try:
        f()
        others[name] = None
    except engine.EngineError as ex:
        others[name] = str(ex)
    eng.set_eigen(0, pb.eigen[0])
    eng.eval(pb.tree.branch)      # (and nothing stays latched for the evaluation after it)
attempt("node_posterior", lambda: eng.node_posterior(pb.tree.root, pb.tree.branch))
d_out = torch.zeros(1, dtype=torch.float64, device="cuda")
def dev():
    eng.eval_device(pb.tree.branch, d_out.data_ptr())
    eng.eigen_status()
attempt("eigen_status", dev)
pb2 = synth.codon_nssites_problem(pb, 2.0, [0.2, 1.0], [0.6, 0.4])
eng2 = engine.engine_for(pb2)
eng2.eval(pb2.tree.branch)
pi2 = np.asarray(pb2.pi, dtype=np.float64).reshape(-1)[:61]
def adg():
    eng2.set_eigen_qrev_batch([0], np.array([Q]), np.array([pi2]), np.array([mr]))
    eng2.eval_adg(pb2.tree.branch, np.full((2, 2), 0.5), np.arange(pb2.n_patt, dtype=np.int32))
try:
    adg()
    others["eval_adg"] = None
except engine.EngineError as ex:
    others["eval_adg"] = str(ex)
# the C host on the same engine library: HIV M2a at the golden's parameters — device decomposition fails, host takes over, same lnL
g = helpers.load_golden("hiv_m2a")
a = hostlib.Analysis(os.path.join(helpers.GOLDEN, "ctl", "hiv_ns2.ctl"), "codeml")
l1 = a.eval_gpu(np.array(g["x"]), want_lnf=False)[0]
xs = np.stack([np.array(g["x"]), np.array(g["x"]) * 1.01])
lb = a.eval_batch_gpu(xs)
print(json.dumps({"first": ok, "code": code, "sweeps": sweeps, "again": again, "host_lnL": l1, "golden": g["lnL"], "batch": lb.tolist(), "others": others}))
"""


def test_a_device_decomposition_that_does_not_converge_is_reported_and_the_host_takes_over(tmp_path):
    """The sweep limit of the device Jacobi (40) lowered to 1 through PAML_AMD_EIGEN_SWEEP_LIMIT, in a process of its own: the evaluation behind
    an unconverged decomposition returns PAML_AMD_ENOCONV (-5) instead of a likelihood, paml_amd_eigen_counters names the set (-1), and the C
    host (pamlh_eval_gpu, the batched evaluation) falls back to its own eigen-decomposition and returns the reference's lnL."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", NOCONV_SCRIPT % {"repo": repo}], env=dict(os.environ, PAML_AMD_EIGEN_SWEEP_LIMIT="1"),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    out = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert out["code"] is not None and "(code -5)" in out["code"] and "sweep limit" in out["code"]
    assert out["sweeps"] == [-1]
    assert isinstance(out["again"], float) and abs(out["again"] - out["first"]) <= 1e-9 * abs(out["first"])
    assert abs(out["host_lnL"] - out["golden"]) <= 2e-6
    assert abs(out["batch"][0] - out["golden"]) <= 2e-6 and out["batch"][1] < out["batch"][0]
    assert b"decomposed on the host from here on" in r.stderr
    for name in ("node_posterior", "eigen_status", "eval_adg"):
        assert out["others"][name] is not None and "(code -5)" in out["others"][name], (name, out["others"][name])
