"""Pin the oracle (oracle/cpu_ref.c) against golden vectors produced by the unmodified reference
binaries (tests/golden/make_golden.py): lnL and every per-pattern log f_h from the `lnf` file."""
import numpy as np
import pytest

import helpers
import oracle

CASES = ["hiv_m0", "hiv_m1a", "hiv_m2a", "hiv_m7", "hiv_m8", "syn_codon_m0", "syn_nuc_gtr_g4", "brown_hky85",
         "stewart_lg_g4", "mhc_m0_scaled", "syn_aa_g4",
         # 4000-pattern versions of the north_star sweep's class tables (M2a's and M8's run through NSsites = 3, M7)
         "syn_codon_m2a_as_m3_4000", "syn_codon_m8_as_m3_4000", "syn_codon_m7_4000"]


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference(name):
    g = helpers.load_golden(name)
    pb = helpers.problem_from_golden(g)
    r = oracle.evaluate(pb)
    # the reference prints lnL with 6 decimals and log f_h with 10
    assert abs(r["lnL"] - g["lnL"]) <= 2e-6 + 1e-9 * abs(g["lnL"]), (r["lnL"], g["lnL"])
    assert np.max(np.abs(r["lnf"] - np.array(g["logf"]))) < 2e-8
    if g.get("counters"):
        assert r["npmat"] == g["counters"][2]          # same number of P(t) constructions as the reference
    if g.get("published_lnL") is not None:             # MHC: the value printed in the reference's own README
        assert abs(r["lnL"] - g["published_lnL"]) < 5e-6


def _closed_form(pb, kind):
    """Swap the random eigen system for one of the closed-form kinds (K80 with kappa = 2.5, or JC69-like)."""
    from paml_amd.problem import EIGEN_JC69LIKE, EIGEN_K80
    if kind == "k80":
        pb.eigen = [dict(kind=EIGEN_K80, kappa=2.5)]
        pb.pi = np.full((1, 4), 0.25) if pb.pi.ndim == 2 else np.full(4, 0.25)
    elif kind == "jc":
        pb.eigen = [dict(kind=EIGEN_JC69LIKE)]
        pb.pi = np.full_like(pb.pi, 1.0 / pb.n)
    return pb


@pytest.mark.parametrize("n,K,kind", [(4, 1, None), (4, 3, None), (20, 2, None), (61, 1, None), (4, 2, "k80"), (20, 1, "jc")])
def test_oracle_branch_derivatives_match_pinned_lnl(n, K, kind):
    """orc_eval_branch (lfuntdd restated) against the golden-pinned full evaluation: l(t) equals lnL with that branch
    length to rounding, and dl / ddl equal its central finite differences."""
    pb = _closed_form(helpers.random_problem(n, 8, 50, K=K, seed=40 + n, ambiguity=(n == 4)), kind)
    for b in (1, pb.tree.n_tips + 1):
        t0 = float(pb.tree.branch[b])

        def f(t):
            pb.tree.branch[b] = t
            v = oracle.evaluate(pb, want_lnf=False)["lnL"]
            pb.tree.branch[b] = t0
            return v
        ts = np.array([t0, 0.5 * t0 + 0.01, 0.3])
        l, dl, ddl = oracle.eval_branch(pb, b, ts)
        for i, t in enumerate(ts):
            e = 1e-4
            assert abs(l[i] - f(t)) <= 1e-11 * abs(l[i])
            assert abs(dl[i] - (f(t + e) - f(t - e)) / (2 * e)) <= 2e-5 * max(1.0, abs(dl[i]))
            assert abs(ddl[i] - (f(t + e) - 2 * f(t) + f(t - e)) / e ** 2) <= 2e-3 * max(1.0, abs(ddl[i]))


@pytest.mark.parametrize("name", ["syn_nuc_gtr_g4_full", "syn_aa_g4_full", "syn_codon_m0_full"])
def test_oracle_matches_reference_full_size(name):
    """BASELINE configs[1] / configs[3] at full size (32 taxa x 1e5 nucleotide, 16 taxa x 1e6 codon patterns): the
    reference binary was run once on the seeded generator's data (make_golden.py); its lnL, the sum and a strided
    sample of its per-pattern log f_h are committed.  The oracle is checked on the sample (and, for the cheap
    nucleotide case, on the whole lnL)."""
    g = helpers.load_golden(name)
    pb = helpers.problem_from_golden(g)
    assert pb.n_patt == g["n_patt"]
    idx = np.arange(0, pb.n_patt, g["sample_stride"])
    if pb.n == 4:
        r = oracle.evaluate(pb, nthreads=4)
        assert abs(r["lnL"] - g["lnL"]) <= 2e-6 + 1e-12 * abs(g["lnL"])
        assert abs(r["lnf"].sum() - g["logf_sum"]) < 1e-4
        lnf = r["lnf"][idx]
    else:
        sub = pb.slice_patterns(0, pb.n_patt)
        sub.z = np.ascontiguousarray(pb.z[:, idx])
        sub.weights = np.ascontiguousarray(pb.weights[idx])
        sub.gene_off = np.array([0, len(idx)], dtype=np.int32)
        lnf = oracle.evaluate(sub)["lnf"]
    assert np.max(np.abs(lnf - np.array(g["logf_sample"]))) < 2e-8


def _sites(pb, rng):
    """com.pose for a problem whose weights are site counts: every pattern repeated `weight` times, in random order."""
    pose = np.repeat(np.arange(pb.n_patt), pb.weights.astype(int))
    rng.shuffle(pose)
    return pose.astype(np.int32)


@pytest.mark.parametrize("scaled", [False, True])
def test_oracle_adg_reduces_to_lfundg_for_independent_sites(scaled):
    """lfunAdG restated (orc_eval_adg): with a rate chain whose rows all equal freqK the sites are independent and the
    likelihood must equal lfundG's (pinned by the golden vectors); the site order then does not matter either."""
    pb = helpers.random_problem(4, 10, 60, K=4, seed=11, scale_every=3 if scaled else None)
    rng = np.random.default_rng(2)
    pb.weights = rng.integers(1, 4, pb.n_patt).astype(float)
    MK = np.tile(pb.freqK, (pb.K, 1))
    ref = oracle.evaluate(pb, want_lnf=False)["lnL"]
    a = oracle.evaluate_adg(pb, MK, _sites(pb, rng))
    b = oracle.evaluate_adg(pb, MK, _sites(pb, rng))
    assert abs(a - ref) <= 1e-11 * abs(ref) and abs(b - ref) <= 1e-11 * abs(ref)
    # a persistent chain (rates of neighbouring sites correlated) gives a different value that depends on the order
    MK2 = 0.7 * np.eye(pb.K) + 0.3 * MK
    assert abs(oracle.evaluate_adg(pb, MK2, _sites(pb, rng)) - ref) > 1e-6


def test_oracle_unrest_matexp_matches_scipy():
    """orc_pmat_qmat (matexp with 7 Taylor terms and 5 squarings, tools.c:4879) against scipy's expm."""
    from scipy.linalg import expm
    from paml_amd.problem import EIGEN_QMAT
    rng = np.random.default_rng(5)
    pb = helpers.random_problem(4, 6, 30, seed=8)
    Q = rng.gamma(1.0, 1.0, size=(4, 4))
    Q[np.diag_indices(4)] = 0
    Q[np.diag_indices(4)] = -Q.sum(axis=1)
    pb.eigen = [dict(kind=EIGEN_QMAT, Q=Q)]
    for node in range(pb.tree.n_nodes):
        if node == pb.tree.root:
            continue
        P = oracle.pmat_branch(pb, 0, 0, node)
        assert np.allclose(P, expm(Q * pb.tree.branch[node] * pb.rate[0] * pb.gene_rate[0]), rtol=0, atol=1e-9)
    assert np.isfinite(oracle.evaluate(pb)["lnL"])


def _brown_anc():
    g = helpers.load_golden("brown_hky85_anc")
    gb = helpers.load_golden("brown_hky85")
    pb = helpers.problem_from_golden(gb)
    raw = ["".join("TCAG"[c] for c in pb.z[:, h]) for h in range(pb.n_patt)]      # baseml's state order T, C, A, G
    return g, pb, raw


def test_oracle_node_posterior_matches_reference_reconstruction():
    """Marginal ancestral reconstruction: for every site pattern of brown.nuc the reference (RateAncestor = 1, same fixed
    parameters as the lnL golden) lists the most probable base and its posterior probability at internal nodes 6, 7, 8."""
    g, pb, raw = _brown_anc()
    posts = [oracle.node_posterior(pb, node - 1) for node in g["nodes_1based"]]
    seen = 0
    for h, patt in enumerate(raw):
        row = g["patterns"][patt]
        for post, best, prob in zip(posts, row["best"], row["prob"]):
            i = int(np.argmax(post[h]))
            assert "TCAG"[i] == best and abs(post[h, i] - prob) < 6e-4, (patt, post[h], best, prob)
        seen += 1
    assert seen == len(g["patterns"]) == pb.n_patt
    assert np.allclose(posts[0].sum(axis=1), 1)
