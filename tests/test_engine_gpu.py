"""GPU parity tests proper: the HIP engine (through the C ABI) against the oracle on the same seeded
inputs and against the reference-generated golden vectors.  FP64 throughout; tolerance 1e-10 relative
on lnL (north_star asks for 1e-6), 1e-9 absolute on per-pattern log f_h."""
import copy

import numpy as np
import pytest

import helpers
import oracle
from paml_amd import synth
from paml_amd.engine import JIT, KEEP_PARTIALS, SHARD, engine_for
from paml_amd.problem import Tree

pytestmark = pytest.mark.gpu

GOLDEN = ["hiv_m0", "hiv_m1a", "hiv_m2a", "hiv_m7", "hiv_m8", "syn_codon_m0", "syn_nuc_gtr_g4", "brown_hky85",
          "stewart_lg_g4", "mhc_m0_scaled", "syn_aa_g4"]


def check(pb, lnl_rtol=1e-10, lnf_atol=1e-9, flags=0):
    ref = oracle.evaluate(pb, want_fhk=True)
    eng = engine_for(pb, flags=flags)
    out = eng.eval(pb.tree.branch, pb.gene_rate, want_lnf=True, want_fhk=True)
    assert np.isfinite(out["lnL"])
    assert abs(out["lnL"] - ref["lnL"]) <= lnl_rtol * abs(ref["lnL"]) + 1e-9, (out["lnL"], ref["lnL"])
    assert np.max(np.abs(out["lnf"] - ref["lnf"])) < lnf_atol
    m = pb.weights > 0
    fk, rk = out["fhK"][:, m], ref["fhK"][:, m]
    # a class whose f(x|class) is many orders below the dominant class (M7/M8 omega ~ 0) is a sum with heavy
    # cancellation: compare it relative to the largest class of the same pattern
    tol = 1e-9 * np.abs(rk) + 1e-12 * np.abs(rk).max(axis=0, keepdims=True)
    assert (np.abs(fk - rk) <= tol).all(), float(np.max(np.abs(fk - rk) / np.abs(rk).max(axis=0, keepdims=True)))
    return eng, out, ref


@pytest.mark.parametrize("name", GOLDEN)
def test_golden(name):
    g = helpers.load_golden(name)
    pb = helpers.problem_from_golden(g)
    eng, out, ref = check(pb)
    assert abs(out["lnL"] - g["lnL"]) <= 2e-6 + 1e-9 * abs(g["lnL"])
    assert np.max(np.abs(out["lnf"] - np.array(g["logf"]))) < 2e-8
    if g.get("counters"):
        assert eng.counters()["n_pmat"] == g["counters"][2]


@pytest.mark.parametrize("n,n_tips,n_patt,K", [(4, 5, 100, 1), (4, 32, 3000, 4), (5, 9, 257, 3), (20, 6, 130, 4),
                                              (20, 24, 1000, 1), (61, 16, 1000, 1), (61, 13, 79, 11), (64, 7, 65, 2),
                                              (60, 40, 200, 1), (21, 5, 16, 1), (2, 4, 50, 1)])
def test_random_models(n, n_tips, n_patt, K):
    pb = helpers.random_problem(n, n_tips, n_patt, K=K, seed=n * 1000 + n_tips)
    check(pb)


@pytest.mark.parametrize("n_tips,n_patt,K,kw", [(6, 98, 4, {}), (24, 1000, 1, {}), (40, 700, 3, dict(ambiguity=True)), (30, 300, 2, dict(scale_every=4)),
                                                 (11, 513, 2, dict(polytomy=True)), (49, 260, 1, {}), (3, 40, 2, {}),
                                                 (12, 20011, 2, {}), (9, 70001, 1, dict(ambiguity=True))])      # many units per workgroup: ranges, tickets, a ragged last unit
def test_20_state_kernel_on_4x4x4_mfma(n_tips, n_patt, K, kw):
    """The per-tree 20-state kernel (jit_generate_m20: five 4-state blocks per partial, rows 0-15 of every product on
    v_mfma_f64_16x16x4 and rows 16-19 on v_mfma_f64_4x4x4, P(t) of every internal branch in LDS, units of 32 patterns handed out by
    an LDS ticket) against the oracle: clean and ambiguous tips, scaling nodes, polytomies, the largest tree it takes."""
    pb = helpers.random_problem(20, n_tips, n_patt, K=K, seed=500 + n_tips, **kw)
    eng, out, ref = check(pb, flags=JIT)
    assert eng.kernel_name == ("mfma4x20_jit" if n_tips > 3 else eng.kernel_name)


@pytest.mark.parametrize("n", [4, 20, 61])
def test_pmat_matches_oracle(n):
    pb = helpers.random_problem(n, 8, 40, K=3, seed=5)
    pb.tree.branch[2] = 0.0            # t < 1e-100 -> identity (tools.c:525)
    pb.tree.branch[3] = 1e-120
    eng = engine_for(pb)
    eng.eval(pb.tree.branch)
    for node in range(pb.tree.n_nodes):
        if node == pb.tree.root:
            continue
        for ic in range(pb.K):
            P = eng.get_pmat(0, ic, node)
            Pr = oracle.pmat_branch(pb, 0, ic, node)
            assert np.max(np.abs(P - Pr)) < 1e-13
    assert np.array_equal(eng.get_pmat(0, 0, 2), np.eye(n))


@pytest.mark.parametrize("n", [4, 20, 61])
def test_ambiguity_codes(n):
    pb = helpers.random_problem(n, 10, 300, K=2, seed=77, ambiguity=True)
    check(pb)


@pytest.mark.parametrize("n,every", [(4, 3), (20, 4), (61, 3)])
def test_node_scaling(n, every):
    pb = helpers.random_problem(n, 30, 200, K=1, seed=9, scale_every=every)
    assert pb.scale_node.sum() >= 2
    check(pb)
    pbk = helpers.random_problem(n, 30, 200, K=3, seed=10, scale_every=every)   # lfundG log-sum-exp branch
    check(pbk)


@pytest.mark.parametrize("n", [4, 61])
def test_scaling_underflow_branch(n):
    """Force max < 1e-300 at a scaled node (treesub.c:7218-7221: partials := 1, factor := -800)."""
    pb = helpers.random_problem(n, 12, 64, K=1, seed=3, scale_every=2)
    for e in pb.eigen:
        e["U"] = e["U"] * 1e-200          # off-diagonal P entries ~1e-200: partials of mismatching tips underflow
    rng = np.random.default_rng(0)
    pb.z[:] = rng.integers(0, n, size=pb.z.shape)
    ref = oracle.evaluate(pb, want_partials=True)
    assert (ref["scalef"] == -800).any()
    check(pb, lnl_rtol=1e-9)


@pytest.mark.parametrize("n", [4, 20, 61])
def test_multigene_and_polytomy(n):
    pb = helpers.random_problem(n, 11, 500, K=2, seed=21, n_genes=3, polytomy=True)
    check(pb)


@pytest.mark.parametrize("n", [4, 61])
def test_zero_weight_patterns_skipped(n):
    pb = helpers.random_problem(n, 6, 90, K=1, seed=4)
    pb.weights[::7] = 0
    check(pb)


@pytest.mark.parametrize("n", [4, 20, 61])
def test_keep_partials_and_dirty_eval(n):
    pb = helpers.random_problem(n, 14, 150, K=2, seed=31, scale_every=4)
    ref = oracle.evaluate(pb, want_partials=True)
    eng, out, _ = check(pb, flags=KEEP_PARTIALS)
    t = pb.tree
    for node in range(t.n_tips, t.n_nodes):
        for ic in range(pb.K):
            got = eng.get_partials(node, ic)
            assert np.allclose(got, ref["partials"][ic, node - t.n_tips], rtol=1e-11, atol=1e-300)
    k = 0
    for node in range(t.n_nodes):
        if pb.scale_node[node]:
            assert np.allclose(eng.get_scale(node, 1), ref["scalef"][1, k], rtol=1e-12)
            k += 1
    # change one tip branch: every node off the path to the root stays clean (com.oldconP)
    father = t.father()
    tip = 3
    br = t.branch.copy()
    br[tip] *= 1.7
    clean = np.ones(t.n_nodes, dtype=np.uint8)
    node = tip
    while node != -1:
        clean[node] = 0
        node = father[node]
    lnl_dirty = eng.eval_dirty(br, clean)
    pb2 = helpers.random_problem(n, 14, 150, K=2, seed=31, scale_every=4)
    pb2.tree.branch[:] = br
    ref2 = oracle.evaluate(pb2)
    assert abs(lnl_dirty - ref2["lnL"]) <= 1e-10 * abs(ref2["lnL"])


@pytest.mark.parametrize("n_genes,scale_every", [(1, None), (3, 4)])
def test_keep_partials_on_the_per_tree_kernel(n_genes, scale_every):
    """PAML_AMD_KEEP_PARTIALS on the per-tree MFMA kernel (jit.h: OP_STORE as coalesced 1 KB wave stores; OP_LOAD when the kernel is asked
    for): every internal node's partial (com.conP, codeml.c:3531-3582) against the oracle, read back through the ONE resident layout the
    128-pattern per-tree kernel and the 64-pattern interpreter share — then a dirty evaluation by the interpreter (LOAD of what the per-tree
    kernel stored), one by a per-tree kernel with LOADs, and a full one by the interpreter, all the same bits per pattern."""
    pb = helpers.random_problem(61, 12, 1500, K=2, seed=77, scale_every=scale_every, n_genes=n_genes)      # (genes: 128-pattern tiles that start off the 64-pattern grid)
    ref = oracle.evaluate(pb, want_partials=True)
    eng, out, _ = check(pb, flags=KEEP_PARTIALS | JIT)
    assert eng.kernel_name == "mfma64_jit"
    t = pb.tree
    got = {(node, ic): eng.get_partials(node, ic) for node in range(t.n_tips, t.n_nodes) for ic in range(pb.K)}
    for (node, ic), g in got.items():
        assert np.allclose(g, ref["partials"][ic, node - t.n_tips], rtol=1e-11, atol=1e-300), (node, ic)
    father = t.father()
    br = t.branch.copy()
    br[2] *= 1.3
    clean = np.ones(t.n_nodes, dtype=np.uint8)
    node = 2
    while node != -1:
        clean[node] = 0
        node = father[node]
    pb2 = helpers.random_problem(61, 12, 1500, K=2, seed=77, scale_every=scale_every, n_genes=n_genes)
    pb2.tree.branch[:] = br
    ref2 = oracle.evaluate(pb2, want_partials=True)
    lnl_jit_load = eng.eval_dirty(br, clean, pb.gene_rate)                      # forced per-tree kernel: LOADs inside it
    assert eng.kernel_name == "mfma64_jit" and abs(lnl_jit_load - ref2["lnL"]) <= 1e-10 * abs(ref2["lnL"])
    for node in range(t.n_tips, t.n_nodes):
        assert np.allclose(eng.get_partials(node, 1), ref2["partials"][1, node - t.n_tips], rtol=1e-11, atol=1e-300), node
    eng.close()
    eng3 = engine_for(pb, flags=KEEP_PARTIALS)      # small data, no per-tree kernel: the interpreter stores
    eng3.eval(t.branch, pb.gene_rate)
    assert eng3.kernel_name != "mfma64_jit"
    for (node, ic), g in got.items():
        assert np.allclose(eng3.get_partials(node, ic), g, rtol=1e-12, atol=1e-300), (node, ic)      # (the per-tree kernel adds column 60's term first: rounding differs)
    l3 = eng3.eval_dirty(br, clean, pb.gene_rate)
    assert abs(l3 - ref2["lnL"]) <= 1e-10 * abs(ref2["lnL"])
    eng3.close()


def test_dirty_evaluation_by_the_interpreter_after_a_full_one_by_the_per_tree_kernel():
    """A data set large enough for the per-tree kernel to be chosen by size (not asked for): the full keep-partials evaluation runs on it
    (128-pattern tiles), eval_dirty — a program with LOADs, different for every set of clean nodes — at first on the interpreter (64-pattern
    tiles), reading what the per-tree kernel stored: one resident layout, nothing invalidated by the change of kernel."""
    pb = helpers.random_problem(61, 10, 33000, K=2, seed=78)
    t = pb.tree
    eng = engine_for(pb, flags=KEEP_PARTIALS)
    out = eng.eval(t.branch, pb.gene_rate)
    assert eng.kernel_name == "mfma64_jit"
    ref = oracle.evaluate(pb, want_partials=True)
    assert abs(out["lnL"] - ref["lnL"]) <= 1e-10 * abs(ref["lnL"])
    for node in (t.n_tips, t.n_nodes - 1):
        assert np.allclose(eng.get_partials(node, 1), ref["partials"][1, node - t.n_tips], rtol=1e-11, atol=1e-300), node
    father = t.father()
    br = t.branch.copy()
    br[1] *= 0.6
    clean = np.ones(t.n_nodes, dtype=np.uint8)
    node = 1
    while node != -1:
        clean[node] = 0
        node = father[node]
    lnl_dirty = eng.eval_dirty(br, clean, pb.gene_rate)
    assert eng.kernel_name in ("mfma64_gather", "mfma64_coop", "mfma64_coopjit")
    pb.tree.branch[:] = br
    ref2 = oracle.evaluate(pb)
    assert abs(lnl_dirty - ref2["lnL"]) <= 1e-10 * abs(ref2["lnL"])
    # ... and back: a full evaluation (per-tree kernel again), the same value as a fresh engine's
    full = engine_for(pb).eval(br, pb.gene_rate)["lnL"]
    assert eng.eval(br, pb.gene_rate)["lnL"] == full and eng.kernel_name == "mfma64_jit"
    # Round 6: the SECOND time the same set of clean nodes is asked for, its LOAD program gets a per-tree kernel of its own (compiled on the
    # worker thread, the interpreter serving until it is there); full and dirty evaluations then alternate between two kernels that both
    # stay loaded, and between two tile tables that both stay built
    import time
    t0 = time.time()
    names = set()
    while time.time() - t0 < 120:
        v = eng.eval_dirty(br, clean, pb.gene_rate)
        names.add(eng.kernel_name)
        assert abs(v - ref2["lnL"]) <= 1e-10 * abs(ref2["lnL"])
        if eng.kernel_name == "mfma64_jit":
            break
        time.sleep(0.2)
    assert eng.kernel_name == "mfma64_jit", names
    for _ in range(3):
        assert eng.eval(br, pb.gene_rate)["lnL"] == full and eng.kernel_name == "mfma64_jit"
        assert abs(eng.eval_dirty(br, clean, pb.gene_rate) - ref2["lnL"]) <= 1e-10 * abs(ref2["lnL"]) and eng.kernel_name == "mfma64_jit"
    t1 = time.time()
    for _ in range(20):
        eng.eval(br, pb.gene_rate)
        eng.eval_dirty(br, clean, pb.gene_rate)
    assert time.time() - t1 < 2.0      # (no compilation, no module load, no tile rebuild inside the alternation)


def test_young_ancestor_root_is_tip():
    """Rooted tree whose root is an observed sequence (codeml.c:3535-3543)."""
    from paml_amd.problem import Tree
    done = 0
    for n in (4, 61):
        for seed in range(8, 40):
            pb = helpers.random_problem(n, 6, 120, K=1, seed=seed)
            t = pb.tree
            tip_sons = [s for s in t.sons[t.root] if s < t.n_tips]
            if not tip_sons:
                continue
            tip = tip_sons[0]
            sons = [list(s) for s in t.sons]
            sons[t.root].remove(tip)
            sons[tip] = [t.root]
            br = t.branch.copy()
            br[t.root] = br[tip]
            br[tip] = 0
            pb.tree = Tree(t.n_tips, t.n_nodes, tip, sons, br, t.label)
            check(pb)
            done += 1
            break
    assert done == 2


def test_full_size_properties_c4():
    """BASELINE configs[3] at full size (16 taxa x 1e6 codon patterns): size-independent checks —
    lnL equals the sum of per-pattern log f_h, pattern order does not matter, and a strided sample
    of patterns agrees with the oracle."""
    pb = synth.codon_m0_problem(n_tips=16, n_patt=1_000_000)
    eng = engine_for(pb)
    out = eng.eval(pb.tree.branch, want_lnf=True)
    assert abs(out["lnf"].sum() - out["lnL"]) < 1e-9 * abs(out["lnL"])
    idx = np.arange(0, pb.n_patt, 997)
    sub = pb.slice_patterns(0, pb.n_patt)
    sub.z = np.ascontiguousarray(pb.z[:, idx])
    sub.weights = np.ascontiguousarray(pb.weights[idx])
    sub.gene_off = np.array([0, len(idx)], dtype=np.int32)
    ref = oracle.evaluate(sub)
    assert np.max(np.abs(out["lnf"][idx] - ref["lnf"])) < 1e-9
    perm = np.random.default_rng(1).permutation(pb.n_patt)
    pb2 = pb.slice_patterns(0, pb.n_patt)
    pb2.z = np.ascontiguousarray(pb.z[:, perm])
    out2 = engine_for(pb2).eval(pb.tree.branch, want_lnf=True)
    assert np.max(np.abs(out2["lnf"] - out["lnf"][perm])) == 0.0      # per-pattern results are order-independent, bitwise
    assert abs(out2["lnL"] - out["lnL"]) < 1e-9 * abs(out["lnL"])


def test_large_size_properties_20_states():
    """The 20-state kernel at 16 taxa x 400 000 patterns x 4 classes (units per workgroup in the dozens, the LDS tickets and the
    contiguous ranges at work): lnL = sum of log f_h, a strided sample against the oracle, and per-pattern results that do not
    depend on the order of the patterns — bitwise."""
    pb = helpers.random_problem(20, 16, 400_000, K=4, seed=2020)
    eng = engine_for(pb)
    out = eng.eval(pb.tree.branch, want_lnf=True)
    assert eng.kernel_name == "mfma4x20_jit"
    assert abs(float(np.dot(out["lnf"], pb.weights)) - out["lnL"]) < 1e-9 * abs(out["lnL"])
    idx = np.arange(0, pb.n_patt, 1999)
    sub = pb.slice_patterns(0, pb.n_patt)
    sub.z = np.ascontiguousarray(pb.z[:, idx])
    sub.weights = np.ascontiguousarray(pb.weights[idx])
    sub.gene_off = np.array([0, len(idx)], dtype=np.int32)
    ref = oracle.evaluate(sub)
    assert np.max(np.abs(out["lnf"][idx] - ref["lnf"])) < 1e-10
    perm = np.random.default_rng(2).permutation(pb.n_patt)
    pb2 = pb.slice_patterns(0, pb.n_patt)
    pb2.z = np.ascontiguousarray(pb.z[:, perm])
    pb2.weights = np.ascontiguousarray(pb.weights[perm])
    out2 = engine_for(pb2).eval(pb.tree.branch, want_lnf=True)
    assert np.max(np.abs(out2["lnf"] - out["lnf"][perm])) == 0.0
    assert abs(out2["lnL"] - out["lnL"]) < 1e-9 * abs(out["lnL"])


@pytest.mark.parametrize("name", ["syn_nuc_gtr_g4_full", "syn_aa_g4_full", "syn_codon_m0_full"])
def test_full_size_against_reference(name):
    """BASELINE configs[1] and configs[3] at full size against the reference binary's own numbers for the same seeded
    data (tests/golden/*_full.json: lnL, sum and a strided sample of per-pattern log f_h)."""
    g = helpers.load_golden(name)
    pb = helpers.problem_from_golden(g)
    out = engine_for(pb).eval(pb.tree.branch, want_lnf=True)
    assert abs(out["lnL"] - g["lnL"]) <= 2e-6 + 1e-12 * abs(g["lnL"]), (out["lnL"], g["lnL"])      # 6 printed decimals
    assert abs(out["lnf"].sum() - g["logf_sum"]) < 1e-4
    idx = np.arange(0, pb.n_patt, g["sample_stride"])
    assert np.max(np.abs(out["lnf"][idx] - np.array(g["logf_sample"]))) < 2e-8                   # 10 printed decimals


@pytest.mark.parametrize("name", ["syn_codon_m1a_full", "syn_codon_m2a_as_m3_full", "syn_codon_m7_full", "syn_codon_m8_as_m3_full",
                                  "syn_codon_m2a_full", "syn_codon_m8_full"])
def test_nssites_sweep_full_size_against_reference(name):
    """The north_star's target workload pinned at scale: the NSsites class tables (K = 2, 3, 10, 11) on the C4 data, 16 taxa x
    10^6 codon patterns, against the unmodified reference program's lnL (and, where it wrote its lnf file, a strided sample
    of per-pattern log f_h).  M2a / M8 proper hold the lnL only — the reference was stopped before its hours of BEB — and
    their class tables are also run through NSsites = 3 (the `_as_m3` goldens), which must give the same lnL."""
    import os
    if not os.path.exists(os.path.join(helpers.GOLDEN, name + ".json")):
        pytest.skip("golden not generated")
    g = helpers.load_golden(name)
    pb = helpers.problem_from_golden(g)
    assert pb.n_patt == 1_000_000 and pb.K == {"m1a": 2, "m2a": 3, "m7": 10, "m8": 11}[name.split("_")[2]]
    out = engine_for(pb).eval(pb.tree.branch, want_lnf="logf_sample" in g)
    assert abs(out["lnL"] - g["lnL"]) <= 2e-6 + 1e-12 * abs(g["lnL"]), (out["lnL"], g["lnL"])      # 6 printed decimals
    if "logf_sample" in g:
        assert abs(out["lnf"].sum() - g["logf_sum"]) < 1e-4
        idx = np.arange(0, pb.n_patt, g["sample_stride"])
        assert np.max(np.abs(out["lnf"][idx] - np.array(g["logf_sample"]))) < 2e-8               # 10 printed decimals
    else:
        twin = helpers.load_golden(name.replace("_full", "_as_m3_full"))
        if name == "syn_codon_m2a_full":      # the same table through NSsites = 3: the reference agrees with itself
            assert twin["lnL"] == g["lnL"]


@pytest.mark.parametrize("n,K,flags", [(4, 1, 0), (4, 3, 0), (4, 3, JIT), (5, 2, JIT), (20, 1, 0), (20, 2, JIT), (61, 1, 0), (61, 3, 0),
                                       (61, 1, JIT), (61, 3, JIT)])
def test_numeric_floors(n, K, flags):
    """lfun's `fh <= 0 -> 1e-80` (treesub.c:7794) and fx_r's `fh <= 0 -> 1e-300` (treesub.c:7741) on the HIP path: with every
    branch length 0, P(t) is the identity (tools.c:525), so a pattern whose tips disagree has a root sum of exactly 0 in
    every class; the lnL must carry log(1e-80) (one class) or log(sum_k freqK_k 1e-300) (lfundG) for it, as the oracle's
    restatement of the floors does."""
    pb = helpers.random_problem(n, 8, 300, K=K, seed=900 + n + K)
    pb.tree.branch[:] = 0.0
    pb.z[:, 1::2] = pb.z[0:1, 1::2]              # odd patterns: all tips equal -> f = pi_state
    assert (pb.z[:, 0::2].min(axis=0) != pb.z[:, 0::2].max(axis=0)).any()
    ref = oracle.evaluate(pb)
    floor = np.log(1e-80) if K == 1 else np.log(1e-300)
    hit = np.isclose(ref["lnf"], floor, rtol=0, atol=1e-9)
    assert hit.any() and not hit.all()
    eng = engine_for(pb, flags=flags)
    out = eng.eval(pb.tree.branch, want_lnf=True, want_fhk=True)
    assert np.max(np.abs(out["lnf"] - ref["lnf"])) < 1e-9
    assert abs(out["lnL"] - ref["lnL"]) <= 1e-10 * abs(ref["lnL"])
    if K > 1:
        assert (out["fhK"][:, hit] == 1e-300).all()      # fx_r stores the floored class likelihoods


@pytest.mark.parametrize("n,K,amb,genes", [(4, 1, False, 1), (4, 4, True, 2), (20, 2, False, 1), (61, 1, False, 1), (61, 3, True, 1)])
def test_eval_branch_matches_oracle(n, K, amb, genes):
    """paml_amd_eval_branch (lfuntdd / lfuntdd_SiteClass) against the oracle for tip and internal branches, several trial
    lengths per call; the ordinary evaluation must still work afterwards."""
    pb = helpers.random_problem(n, 9, 150, K=K, seed=60 + n + K, ambiguity=amb, n_genes=genes)
    eng = engine_for(pb)
    base = eng.eval(pb.tree.branch, pb.gene_rate)["lnL"]
    t = pb.tree
    for b in (0, 4, t.n_tips + 1, t.n_nodes - 1):
        if b == t.root:
            continue
        ts = np.array([t.branch[b], 0.02, 0.7])
        l, dl, ddl = eng.eval_branch(b, ts, t.branch, pb.gene_rate)
        rl, rdl, rddl = oracle.eval_branch(pb, b, ts)
        assert np.allclose(l, rl, rtol=1e-11, atol=0), (b, l, rl)
        assert np.allclose(dl, rdl, rtol=1e-9, atol=1e-9)
        assert np.allclose(ddl, rddl, rtol=1e-9, atol=1e-8)
        assert abs(l[0] - base) <= 1e-11 * abs(base)           # l(t_current) is the tree's lnL
    again = eng.eval(pb.tree.branch, pb.gene_rate)["lnL"]
    assert again == base


@pytest.mark.parametrize("n,K,kind", [(4, 2, "k80"), (20, 1, "jc")])
def test_eval_branch_closed_form_models(n, K, kind):
    """JC69 / K80 (baseml) and the Poisson amino-acid model: P, dP, ddP from the closed forms."""
    from test_oracle_golden import _closed_form
    pb = _closed_form(helpers.random_problem(n, 9, 150, K=K, seed=33 + n), kind)
    eng = engine_for(pb)
    base = eng.eval(pb.tree.branch, pb.gene_rate)["lnL"]
    assert abs(base - oracle.evaluate(pb)["lnL"]) <= 1e-10 * abs(base)
    for b in (2, pb.tree.n_tips + 2):
        ts = np.array([pb.tree.branch[b], 0.03, 0.6])
        l, dl, ddl = eng.eval_branch(b, ts, pb.tree.branch, pb.gene_rate)
        rl, rdl, rddl = oracle.eval_branch(pb, b, ts)
        assert np.allclose(l, rl, rtol=1e-11, atol=0) and abs(l[0] - base) <= 1e-11 * abs(base)
        assert np.allclose(dl, rdl, rtol=1e-9, atol=1e-9) and np.allclose(ddl, rddl, rtol=1e-9, atol=1e-8)


@pytest.mark.parametrize("n,K", [(4, 3), (61, 2)])
def test_eval_branch_with_scaling_nodes(n, K):
    """Trees with NodeScale flags: the exported partials carry their summed scale factors into the branch kernel."""
    pb = helpers.random_problem(n, 14, 120, K=K, seed=70 + n, scale_every=3)
    assert pb.scale_node is not None and pb.scale_node.sum() >= 2
    eng = engine_for(pb)
    base = eng.eval(pb.tree.branch, pb.gene_rate)["lnL"]
    t = pb.tree
    scaled = list(np.nonzero(pb.scale_node)[0])
    for b in [1, scaled[0], scaled[-1], t.n_nodes - 1]:
        if b == t.root:
            continue
        ts = np.array([t.branch[b], 0.05, 0.9])
        l, dl, ddl = eng.eval_branch(b, ts, t.branch, pb.gene_rate)
        rl, rdl, rddl = oracle.eval_branch(pb, b, ts)
        assert np.allclose(l, rl, rtol=1e-11, atol=0), (b, l, rl)
        assert np.allclose(dl, rdl, rtol=1e-9, atol=1e-9)
        assert np.allclose(ddl, rddl, rtol=1e-9, atol=1e-8)
        assert abs(l[0] - base) <= 1e-11 * abs(base)


@pytest.mark.parametrize("n,K,amb,genes", [(4, 4, False, 2), (20, 2, True, 1), (61, 1, False, 1), (61, 3, True, 1)])
def test_eval_batch_matches_single_evals(n, K, amb, genes):
    """paml_amd_eval_batch: every element of a batch (own branch lengths, gene rates, class frequencies / rates) gives
    exactly the lnL of the corresponding single evaluation, and the oracle's."""
    pb = helpers.random_problem(n, 10, 300, K=K, seed=80 + n + K, ambiguity=amb, n_genes=genes)
    eng = engine_for(pb)
    rng = np.random.default_rng(5)
    B = 7
    br = np.abs(pb.tree.branch[None, :] * (1 + 0.3 * rng.standard_normal((B, pb.tree.n_nodes))))
    br[0] = pb.tree.branch
    gr = pb.gene_rate[None, :] * (1 + 0.1 * rng.random((B, pb.n_genes)))
    fk = rng.dirichlet(np.ones(pb.K), size=B)
    rt = pb.rate[None, :] * (1 + 0.2 * rng.random((B, pb.K)))
    fk[0], rt[0], gr[0] = pb.freqK, pb.rate, pb.gene_rate
    got = eng.eval_batch(br, gene_rate=gr, freqK=fk, rate=rt)
    assert abs(got[0] - oracle.evaluate(pb)["lnL"]) <= 1e-10 * abs(got[0])
    for b in range(B):
        q = copy.copy(pb)
        q.freqK, q.rate, q.gene_rate = fk[b].copy(), rt[b].copy(), gr[b].copy()
        q.tree = Tree(pb.tree.n_tips, pb.tree.n_nodes, pb.tree.root, pb.tree.sons, br[b].copy(), pb.tree.label)
        ref = oracle.evaluate(q)["lnL"]
        assert abs(got[b] - ref) <= 1e-10 * abs(ref), (b, got[b], ref)
    # shared tables (NULL) and a plain evaluation afterwards
    got2, lnf2 = eng.eval_batch(br, gene_rate=np.tile(pb.gene_rate, (B, 1)), want_lnf=True)
    one = eng.eval(br[3], pb.gene_rate, want_lnf=True)
    assert got2[3] == one["lnL"]
    assert np.array_equal(lnf2[3], one["lnf"])                  # per-pattern log f_h of every element (HessianSKT2004's input)


def test_eval_batch_per_element_eigen_sets():
    """Batch elements that point at different eigen systems (a nudged kappa/omega is another set id)."""
    pb = helpers.random_problem(61, 8, 200, K=1, seed=91)
    pb2 = helpers.random_problem(61, 8, 200, K=1, seed=92)          # same shapes, different Q
    both = copy.copy(pb)
    both.eigen = [pb.eigen[0], pb2.eigen[0]]
    eng = engine_for(both)
    br = np.stack([pb.tree.branch, pb.tree.branch * 1.1, pb.tree.branch])
    eo = np.array([0, 1, 1], dtype=np.int32).reshape(3, 1, 1, 1)
    got = eng.eval_batch(br, eigen_of=eo)
    for b in range(3):
        q = copy.copy(both)
        q.eigen_of = np.full_like(both.eigen_of, eo[b].ravel()[0])
        q.tree = Tree(pb.tree.n_tips, pb.tree.n_nodes, pb.tree.root, pb.tree.sons, br[b].copy(), pb.tree.label)
        ref = oracle.evaluate(q)["lnL"]
        assert abs(got[b] - ref) <= 1e-10 * abs(ref), (b, got[b], ref)


def test_large_tree_kernel_is_compiled_in_the_background(monkeypatch, tmp_path):
    """A 150-tip tree's kernel takes ten seconds to compile.  When the specialised kernels are switched on by the problem's size (not
    forced), the compile runs on a worker thread: the first evaluations come from the interpreter kernel, later ones from the
    per-tree kernel, and both agree with the oracle.  (A code-object cache of its own: on a box that has run the suite before, the
    user's cache would hand the finished kernel to the first evaluation.)"""
    import time
    from paml_amd.engine import Engine
    monkeypatch.delenv("PAML_AMD_JIT", raising=False)
    monkeypatch.setenv("PAML_AMD_JIT_CACHE", str(tmp_path))
    pb = helpers.random_problem(61, 150, 1100, K=1, seed=91)
    eng = Engine(pb.n, pb.tree.n_tips, pb.n_patt, max_classes=64).load(pb)      # 1100 x 64 >= 65536: on by size
    ref = oracle.evaluate(pb)["lnL"]
    first = eng.eval(pb.tree.branch, pb.gene_rate)["lnL"]
    assert eng.kernel_name in ("mfma64_gather", "mfma64_coop", "mfma64_coopjit"), eng.kernel_name      # (the interpreter kernels: 69 groups of 16 patterns get a CU each)
    assert abs(first - ref) <= 1e-10 * abs(ref)
    # two builds on the worker thread: the quick one (the compiler without the passes that are quadratic on a basic block this large:
    # "mfma64_jit_quick"), then the full one that replaces it ("mfma64_jit"); with a warm code-object cache the engine may skip stages
    t0 = time.time()
    seen = []
    while eng.kernel_name != "mfma64_jit" and time.time() - t0 < 180:
        time.sleep(0.5)
        later = eng.eval(pb.tree.branch, pb.gene_rate)["lnL"]
        assert abs(later - ref) <= 1e-10 * abs(ref)
        if eng.kernel_name not in seen:
            seen.append(eng.kernel_name)
    assert eng.kernel_name == "mfma64_jit", "the background compile never finished (%r)" % (seen,)
    assert all(k in ("mfma64_gather", "mfma64_coop", "mfma64_coopjit", "mfma64_jit_quick", "mfma64_jit") for k in seen), seen
    assert abs(eng.eval(pb.tree.branch, pb.gene_rate)["lnL"] - ref) <= 1e-10 * abs(ref)
    eng.close()
    # an engine destroyed while its compile is still running waits for the worker
    eng2 = Engine(pb.n, pb.tree.n_tips, pb.n_patt, max_classes=64).load(pb)
    eng2.eval(pb.tree.branch, pb.gene_rate)
    eng2.close()


def test_compiled_kernels_are_cached_on_disk(monkeypatch, tmp_path):
    """PAML_AMD_JIT_CACHE: the code object of a tree's kernel is written once and read back by the next engine (same lnL)."""
    import time
    monkeypatch.setenv("PAML_AMD_JIT", "1")
    monkeypatch.setenv("PAML_AMD_JIT_CACHE", str(tmp_path))
    pb = helpers.random_problem(61, 40, 200, K=1, seed=93)
    t0 = time.time(); eng, out, ref = check(pb); t_cold = time.time() - t0
    assert eng.kernel_name == "mfma64_jit"
    files = list(tmp_path.glob("*.hsaco"))
    assert len(files) == 1 and files[0].stat().st_size > 1000
    eng.close()
    t0 = time.time(); eng2, out2, _ = check(pb); t_warm = time.time() - t0
    assert eng2.kernel_name == "mfma64_jit" and out2["lnL"] == out["lnL"]
    assert len(list(tmp_path.glob("*.hsaco"))) == 1 and t_warm < t_cold


def test_per_tree_kernel_spills_deep_stacks(monkeypatch):
    """A balanced 128-tip tree needs six partial-stack slots; the per-tree 61-state kernel keeps four in registers and moves
    the deeper ones through global scratch (jit_spill / jit_mul_mem) — same lnL and log f_h as the oracle."""
    from paml_amd.problem import balanced_tree
    monkeypatch.setenv("PAML_AMD_JIT", "1")
    pb = helpers.random_problem(61, 128, 300, K=2, seed=77)
    t = balanced_tree(128)
    rng = np.random.default_rng(3)
    t.branch = rng.uniform(0.01, 0.2, t.n_nodes); t.branch[t.root] = 0
    pb.tree = t
    eng, out, ref = check(pb)
    assert eng.kernel_name == "mfma64_jit"
    from paml_amd import engine as _engine
    assert "jit_spill(" in _engine.debug_jit(t, compile=False)


@pytest.mark.parametrize("n_tips,K,scale_every", [(50, 2, None), (60, 4, None), (64, 1, 30)])
def test_20_state_matrix_core_kernel_beyond_the_lds_capacity(n_tips, K, scale_every, monkeypatch):
    """20 states on v_mfma_f64_16x16x4 + 4x4x4 (jit_generate_m20) keeps the P(t) of 46 internal branches in LDS; larger trees (more than 49
    taxa) read the other branches' operands from the operand-order copy the P(t) kernel leaves in global memory (m20h_matvec2x), one
    product ahead.  lnL and every per-pattern value against the oracle; the same values as the 16x16x4 kernel on padded matrices."""
    monkeypatch.setenv("PAML_AMD_JIT", "1")
    pb = helpers.random_problem(20, n_tips, 1300, K=K, seed=900 + n_tips, scale_every=scale_every)
    eng, out, ref = check(pb)
    assert eng.kernel_name == "mfma4x20_jit", eng.kernel_name
    monkeypatch.setenv("PAML_AMD_NO_M20", "1")
    eng2, out2, _ = check(pb)
    assert eng2.kernel_name != "mfma4x20_jit"
    assert abs(out2["lnL"] - out["lnL"]) <= 1e-11 * abs(out["lnL"])


@pytest.mark.parametrize("n,n_tips,n_patt,K,jit", [(61, 90, 300, 1, True), (61, 130, 140, 1, True), (61, 200, 300, 2, True), (61, 230, 130, 1, True), (61, 208, 129, 1, True),
                                                   (61, 410, 140, 2, True), (61, 620, 150, 2, True), (61, 1000, 130, 1, True), (61, 7, 1, 2, False),
                                                   (61, 12, 129, 20, True), (33, 9, 200, 2, False), (4, 150, 1000, 1, True),
                                                   (5, 40, 777, 2, True), (20, 60, 500, 1, False)])
def test_size_limits_and_kernel_fallbacks(n, n_tips, n_patt, K, jit, monkeypatch):
    """Edges of the kernel selection: many tips (the specialised 61-state kernel keeps two tip-code blocks in LDS up to 95 tips,
    one — replaced between tiles — up to 207, and beyond that two HALVES per tile in the order the walk consumes the tips, the second
    replacing the first at the crossing: jit_zplan), one pattern, many classes, a state count between the
    specialised sizes (padded MFMA path), ragged last tiles — each with the specialised kernels forced on or off."""
    monkeypatch.setenv("PAML_AMD_JIT", "1" if jit else "0")
    pb = helpers.random_problem(n, n_tips, n_patt, K=K, seed=500 + n + n_tips, ambiguity=(n_patt % 2 == 1), scale_every=40 if n_tips > 90 else None)
    eng, out, ref = check(pb)
    if n == 61 and n_tips in (130, 200):
        assert eng.kernel_name == "mfma64_jit"             # one tip-code block
    if n == 61 and n_tips > 207:      # (208: the first size in pieces; 410: two pieces' limit; 620 / 1000 (round 6): three / five pieces; random topologies, ambiguity codes, scaling nodes)
        assert eng.kernel_name == "mfma64_jit"             # a tile's tip codes in pieces, one in LDS at a time
        # ... the same values from the interpreter kernels
        monkeypatch.setenv("PAML_AMD_JIT", "0")
        eng0, out0, _ = check(pb)
        assert eng0.kernel_name in ("mfma64_gather", "mfma64_coop") and abs(out0["lnL"] - out["lnL"]) <= 1e-11 * abs(out["lnL"])
    if n == 61 and n_tips == 90 and jit:
        assert eng.kernel_name == "mfma64_jit"


@pytest.mark.parametrize("n,n_tips,n_patt,K,genes,amb,every", [(4, 12, 700, 4, 1, True, None), (5, 9, 300, 2, 1, False, None),
                                                             (20, 14, 400, 3, 2, True, None), (20, 33, 260, 1, 1, False, 9),
                                                             (33, 10, 200, 2, 1, False, None), (61, 13, 300, 2, 2, True, None),
                                                             (61, 40, 150, 1, 1, False, 12), (61, 6, 129, 3, 1, False, None)])
def test_specialised_kernels_via_flag(n, n_tips, n_patt, K, genes, amb, every):
    """The per-tree specialised kernels (PAML_AMD_JIT flag, independent of the size threshold that turns them on by
    default): 4 / 5 states unrolled scalar-operand walk, 20 / 33 states the MFMA kernel trimmed to the model's size,
    61 states the full one — with several genes, ambiguity codes, scaling nodes, and an evaluation batch."""
    from paml_amd.engine import JIT
    pb = helpers.random_problem(n, n_tips, n_patt, K=K, seed=900 + n + n_tips, ambiguity=amb, n_genes=genes, scale_every=every)
    eng, out, ref = check(pb, flags=JIT)
    assert eng.kernel_name.endswith("_jit"), eng.kernel_name
    br = np.stack([pb.tree.branch, pb.tree.branch * 1.07])
    got = eng.eval_batch(br, gene_rate=np.tile(pb.gene_rate, (2, 1)))
    assert got[0] == out["lnL"]
    q = copy.copy(pb)
    q.tree = Tree(pb.tree.n_tips, pb.tree.n_nodes, pb.tree.root, pb.tree.sons, br[1].copy(), pb.tree.label)
    r1 = oracle.evaluate(q)["lnL"]
    assert abs(got[1] - r1) <= 1e-10 * abs(r1)


@pytest.mark.parametrize("n,n_tips,n_patt,K,n_amb,kw", [(61, 13, 700, 2, 20, {}), (61, 40, 300, 1, 60, dict(scale_every=12)), (64, 9, 400, 1, 9, {}),
                                                       (61, 16, 2000, 1, 150, dict(amb_rate=0.3)), (33, 10, 300, 2, 40, {}), (61, 12, 500, 3, 12, dict(n_genes=2))])
def test_more_than_64_character_codes_on_the_per_tree_kernel(n, n_tips, n_patt, K, n_amb, kw, monkeypatch):
    """61 sense codons + more than three distinct ambiguous triplets (SetMapAmbiguity treesub.c:1218-1286; the ambiguous-tip branch of
    ConditionalPNode, codeml.c:3560-3567): up to round 5 such data fell to the gather interpreter.  The per-tree kernel's ring block has
    a tip's rows of 64 codes; lanes whose code lies beyond add up the rows of the code's states themselves (jit_tip_overflow), and the engine
    numbers the ambiguous codes by frequency x set size so that those are the rare ones.  lnL, every log f_h and fhK against the oracle
    to 1e-10 — check() crosses the 64 codes — on the per-tree kernel, equal to the interpreter's values, and a keep-partials engine too."""
    pb = helpers.random_problem(n, n_tips, n_patt, K=K, seed=6400 + n_tips, ambiguity=True, n_amb=n_amb, **kw)
    assert pb.n_codes > 64 and len(np.unique(pb.z)) > 64
    eng, out, ref = check(pb, flags=JIT)
    assert eng.kernel_name == "mfma64_jit", eng.kernel_name
    monkeypatch.setenv("PAML_AMD_JIT", "0")
    eng0, out0, _ = check(pb)
    assert eng0.kernel_name in ("mfma64_gather", "mfma64_coop", "mfma64_coopjit")
    assert abs(out0["lnL"] - out["lnL"]) <= 1e-12 * abs(out["lnL"]) and np.max(np.abs(out0["lnf"] - out["lnf"])) < 1e-11
    monkeypatch.setenv("PAML_AMD_JIT", "1")
    engk, outk, _ = check(pb, flags=KEEP_PARTIALS)
    assert engk.kernel_name == "mfma64_jit" and outk["lnL"] == out["lnL"]


def test_more_than_64_codes_with_unordered_state_sets_stay_on_the_interpreter():
    """The per-tree kernel's overflow path adds a code's rows in ascending state order — the order of SetMapAmbiguity's CharaMap; a caller
    whose map lists a set's states in another order gets the interpreter (whose table rows are summed in the caller's order), same values."""
    pb = helpers.random_problem(61, 10, 300, K=1, seed=77, ambiguity=True, n_amb=10)
    for c in range(pb.n, pb.n_codes):      # (every ambiguous set: whichever of them the engine numbers from 64 on is then out of order)
        k = pb.n_chara[c]
        pb.chara_map[c, :k] = pb.chara_map[c, :k][::-1]
    eng, out, ref = check(pb, flags=JIT)
    assert eng.kernel_name != "mfma64_jit"


@pytest.mark.parametrize("n,n_tips,n_patt,K,cuts,own,kw", [
    (4, 12, 3000, 4, [0, 700, 1536, 3000], False, {}),                        # a boundary inside a 256-pattern sub-tile, one on a chunk edge
    (4, 32, 5000, 4, [0, 100, 130, 2600, 5000], True, {}),                    # two boundaries inside one sub-tile; the genes' own models (Mgene 2-4)
    (4, 9, 2000, 1, [0, 999, 2000], True, dict(ambiguity=True)),
    (4, 20, 1500, 3, [0, 400, 1500], False, dict(scale_every=4)),             # scaling nodes: the mixture by log-sum-exp from the stored class values
    (5, 9, 1200, 2, [0, 513, 1200], True, {}),
    (20, 14, 3000, 3, [0, 1000, 1031, 3000], True, dict(ambiguity=True)),     # a gene of 31 patterns: one ragged unit
    (20, 33, 2600, 1, [0, 1300, 2600], False, dict(scale_every=9)),
    (20, 60, 1400, 2, [0, 500, 1400], True, {}),                              # beyond the LDS capacity and several genes
])
def test_several_genes_on_the_per_tree_kernels(n, n_tips, n_patt, K, cuts, own, kw, monkeypatch):
    """Option G (com.posG / com.rgene / com.piG, treesub.c:487) on the per-tree kernels (round 6): the 20-state matrix-core kernel serves
    several genes (a workgroup = one (gene, class)); the 4- / 5-state fused kernel has a several-genes form behind PAML_AMD_VF_GENES=1 (not
    the default: it measures no faster than the unfused per-tree kernel, which stays the default and is what the second half checks).
    lnL, every log f_h and fhK against the oracle, the kernel names, and for the fused kernel the bits of the unfused kernel +
    reduce_stage1 (same lane and turn per pattern whatever the gene boundaries)."""
    pb = helpers.random_problem(n, n_tips, n_patt, K=K, seed=4000 + n + n_tips, n_genes=len(cuts) - 1, **kw)
    pb.gene_off = np.array(cuts, dtype=np.int32)
    if own:
        helpers.give_genes_their_own_models(pb, seed=n_tips)
    monkeypatch.setenv("PAML_AMD_VF_GENES", "1")
    eng, out, ref = check(pb, flags=JIT)
    want = {4: "valu4_fused_jit", 5: "valu5_fused_jit", 20: "mfma4x20_jit"}[n]
    assert eng.kernel_name == want, eng.kernel_name
    br = np.stack([pb.tree.branch, pb.tree.branch * 1.07, pb.tree.branch * 0.9])
    got = eng.eval_batch(br, gene_rate=np.tile(pb.gene_rate, (3, 1)))
    assert got[0] == out["lnL"]
    q = copy.copy(pb)
    q.tree = Tree(pb.tree.n_tips, pb.tree.n_nodes, pb.tree.root, pb.tree.sons, br[2].copy(), pb.tree.label)
    r2 = oracle.evaluate(q)["lnL"]
    assert abs(got[2] - r2) <= 1e-10 * abs(r2)
    if n <= 5:
        monkeypatch.delenv("PAML_AMD_VF_GENES")
        eng0, out0, _ = check(pb, flags=JIT)
        assert eng0.kernel_name == "valu%d_jit" % n
        assert out0["lnL"] == out["lnL"] and np.array_equal(out0["lnf"], out["lnf"])


@pytest.mark.parametrize("n,K,every", [(4, 4, None), (4, 3, 3), (61, 2, None)])
def test_eval_adg_matches_oracle(n, K, every):
    """paml_amd_eval_adg (lfunAdG: fx_r on the device, the rate chain over the sites on the host) against the oracle."""
    from test_oracle_golden import _sites
    pb = helpers.random_problem(n, 9, 80, K=K, seed=21 + n, scale_every=every)
    rng = np.random.default_rng(4)
    pb.weights = rng.integers(1, 4, pb.n_patt).astype(float)
    pose = _sites(pb, rng)
    MK = 0.6 * np.eye(K) + 0.4 * rng.dirichlet(np.ones(K), size=K)
    eng = engine_for(pb)
    got = eng.eval_adg(pb.tree.branch, MK, pose, pb.gene_rate)
    ref = oracle.evaluate_adg(pb, MK, pose)
    assert abs(got - ref) <= 1e-10 * abs(ref), (got, ref)
    assert abs(eng.eval_adg(pb.tree.branch, np.tile(pb.freqK, (K, 1)), pose, pb.gene_rate) - eng.eval(pb.tree.branch, pb.gene_rate)["lnL"]) <= 1e-10 * abs(ref)


def _unrest_problem(seed, n_tips=7, n_patt=120, K=1):
    """baseml UNREST-like: a general (non-reversible) 4 x 4 rate matrix, arbitrary root frequencies, rooted tree."""
    from paml_amd.problem import EIGEN_QMAT
    pb = helpers.random_problem(4, n_tips, n_patt, K=K, seed=seed)
    rng = np.random.default_rng(seed)
    Q = rng.gamma(1.0, 1.0, size=(4, 4))
    Q[np.diag_indices(4)] = 0
    Q[np.diag_indices(4)] = -Q.sum(axis=1)
    Q /= np.abs(np.diag(Q)).mean()
    pb.eigen = [dict(kind=EIGEN_QMAT, Q=Q)]
    pb.pi = rng.dirichlet(np.ones(4) * 3)
    return pb, Q


@pytest.mark.parametrize("K", [1, 3])
def test_rate_matrix_kind_unrest(K):
    """PAML_AMD_EIGEN_QMAT: P(t) = matexp(Qt, n, 7, 5) as GetPMatBranch builds it for UNREST — against the oracle's
    restatement and (to the method's own accuracy) scipy's expm."""
    from scipy.linalg import expm
    pb, Q = _unrest_problem(77, K=K)
    eng, out, ref = check(pb)
    node = pb.tree.n_tips + 1
    P = eng.get_pmat(0, 0, node)
    assert np.allclose(P, oracle.pmat_branch(pb, 0, 0, node), rtol=0, atol=1e-14)
    t = pb.tree.branch[node] * pb.rate[0] * pb.gene_rate[0]
    assert np.allclose(P, expm(Q * t), rtol=0, atol=1e-10) and np.allclose(P.sum(axis=1), 1, atol=1e-12)
    with pytest.raises(Exception):
        eng.eval_branch(1, np.array([0.1]), pb.tree.branch, pb.gene_rate)


def test_abi_error_behaviour():
    """The C ABI reports misuse through return codes + paml_amd_last_error (never exit(), unlike the reference's zerror)."""
    from paml_amd.engine import Engine, EngineError
    pb = helpers.random_problem(4, 6, 50, K=2, seed=1)
    eng = Engine(pb.n, pb.tree.n_tips, pb.n_patt, max_classes=2)
    with pytest.raises(EngineError, match="before set_tips"):
        eng.eval(pb.tree.branch)
    with pytest.raises(EngineError, match="code >= n_codes"):
        bad = pb.z.copy()
        bad[0, 0] = 200
        eng.set_tips(bad, pb.weights, cleandata=True)
    eng.load(pb)
    assert np.isfinite(eng.eval(pb.tree.branch)["lnL"])
    with pytest.raises(EngineError, match="KEEP_PARTIALS"):
        eng.eval_dirty(pb.tree.branch, np.zeros(pb.tree.n_nodes, dtype=np.uint8))
    with pytest.raises(EngineError, match="KEEP_PARTIALS"):
        eng.get_partials(pb.tree.n_tips)
    with pytest.raises(EngineError, match="no branch"):
        eng.eval_branch(pb.tree.root, np.array([0.1]), pb.tree.branch)
    with pytest.raises(EngineError):
        eng.set_classes(pb.mode, np.ones(5) / 5, np.ones(5), np.zeros((1, 5, 1), dtype=np.int32))      # K > max_classes
    with pytest.raises(EngineError, match="out of range"):
        eng.eval_batch(np.tile(pb.tree.branch, (2, 1)), eigen_of=np.full((2, 1, 2, 1), 7, dtype=np.int32))
    assert np.isfinite(eng.eval(pb.tree.branch)["lnL"])          # the engine is still usable after the errors


@pytest.mark.parametrize("every", [None, 2])
def test_beb_grid_matches_numpy_restatement(every):
    """paml_amd_beb_grid (the BEB grid integral as device kernels) against a direct numpy restatement of
    lfunNSsites_M2M8's sums (codeml.c:6482-6580) on a synthetic K-class problem with several thousand patterns; the
    reference's own BEB tables pin the same code through the C host (test_host_c.py).  With scaling nodes fhK holds
    log f + scale factors and the classes are compared through exp(fhK - max) (codeml.c:6286-6294)."""
    K, ncls, ngrid = 7, 3, 500
    pb = helpers.random_problem(61, 8, 3000, K=K, seed=31, scale_every=every)
    rng = np.random.default_rng(8)
    pb.weights = rng.integers(0, 4, pb.n_patt).astype(float)          # some zero-weight patterns too
    eng = engine_for(pb)
    out = eng.eval(pb.tree.branch, pb.gene_rate, want_fhk=True)
    pcl = rng.dirichlet(np.ones(ncls), size=ngrid)
    iw = rng.integers(0, K, size=(ngrid, ncls)).astype(np.int32)
    wc = np.linspace(0.1, 4.0, K)
    got = eng.beb_grid(pcl, iw, wc)
    m = pb.weights > 0                                            # fx_r leaves fhK = 0 for patterns that do not count
    if every:
        assert (out["fhK"][:, m] < 0).all()                       # logarithms
        f = np.exp(out["fhK"][:, m] - out["fhK"][:, m].max(axis=0, keepdims=True))
    else:
        f = out["fhK"][:, m] / out["fhK"][:, m].max(axis=0, keepdims=True)                # [K][patterns with weight]
    mix = np.einsum("gc,gch->gh", pcl, f[iw])                                             # [ngrid][...]
    lnfxs = (np.log(mix) * pb.weights[m]).sum(axis=1)
    fx = np.log(np.exp(lnfxs - lnfxs.max()).sum()) + lnfxs.max()
    wg = np.exp(lnfxs - fx)
    t = pcl[:, :, None] * f[iw] / mix[:, None, :] * wg[:, None, None]                    # [ngrid][ncls][n_patt]
    pr = t[:, -1, :].sum(axis=0)
    m1 = (t * wc[iw][:, :, None]).sum(axis=(0, 1))
    m2 = (t * (wc[iw] ** 2)[:, :, None]).sum(axis=(0, 1))
    assert abs(got["ln_fx"] - fx) <= 1e-9 * abs(fx)
    assert np.allclose(got["pr_last"][m], pr, rtol=1e-9, atol=1e-12) and np.allclose(got["mean_w"][m], m1, rtol=1e-9, atol=1e-12)
    assert np.allclose(got["sd_w"][m], np.sqrt(np.maximum(m2 - m1 * m1, 0)), rtol=1e-6, atol=1e-9)
    assert (got["pr_last"][~m] == 0).all()


def test_beb_grid_classes_matches_numpy_restatement():
    """paml_amd_beb_grid_classes (posterior of every mixture class, lfunNSsites_ACD codeml.c:6970-6985) with more evaluated
    classes than the register-resident kernel takes (K = 40 > 32) against numpy; class posteriors sum to 1 per pattern."""
    K, ncls, ngrid = 40, 4, 300
    pb = helpers.random_problem(20, 7, 1500, K=K, seed=33)
    rng = np.random.default_rng(9)
    pb.weights = rng.integers(0, 3, pb.n_patt).astype(float)
    eng = engine_for(pb)
    out = eng.eval(pb.tree.branch, pb.gene_rate, want_fhk=True)
    pcl = rng.dirichlet(np.ones(ncls), size=ngrid)
    iw = rng.integers(0, K, size=(ngrid, ncls)).astype(np.int32)
    got = eng.beb_grid_classes(pcl, iw)
    m = pb.weights > 0
    f = out["fhK"][:, m] / out["fhK"][:, m].max(axis=0, keepdims=True)
    mix = np.einsum("gc,gch->gh", pcl, f[iw])
    lnfxs = (np.log(mix) * pb.weights[m]).sum(axis=1)
    fx = np.log(np.exp(lnfxs - lnfxs.max()).sum()) + lnfxs.max()
    post = (pcl[:, :, None] * f[iw] / mix[:, None, :] * np.exp(lnfxs - fx)[:, None, None]).sum(axis=0)
    assert abs(got["ln_fx"] - fx) <= 1e-9 * abs(fx)
    assert np.allclose(got["post"][:, m], post, rtol=1e-9, atol=1e-12)
    assert np.allclose(got["post"][:, m].sum(axis=0), 1, atol=1e-9)
    with pytest.raises(RuntimeError):
        eng.beb_grid_classes(np.full((2, 9), 1 / 9), np.zeros((2, 9), dtype=np.int32))      # more than 8 mixture classes


@pytest.mark.parametrize("n,K,amb,every", [(4, 1, False, None), (4, 3, True, 3), (20, 2, False, None), (61, 2, True, None)])
def test_node_posterior_matches_oracle(n, K, amb, every):
    """paml_amd_node_posterior (marginal ancestral reconstruction: one fused walk of the tree rooted at the node) against the
    oracle's message-passing restatement, for the root, a deep node and a node next to tips; evaluation still fine after."""
    pb = helpers.random_problem(n, 9, 140, K=K, seed=51 + n, ambiguity=amb, scale_every=every)
    eng = engine_for(pb)
    base = eng.eval(pb.tree.branch, pb.gene_rate)["lnL"]
    for node in (pb.tree.root, pb.tree.n_tips + 1, pb.tree.n_nodes - 1):
        got = eng.node_posterior(node, pb.tree.branch, pb.gene_rate)
        ref = oracle.node_posterior(pb, node)
        assert np.allclose(got, ref, rtol=1e-9, atol=1e-13), (node, float(np.max(np.abs(got - ref))))
        assert np.allclose(got.sum(axis=1), 1)
    assert eng.eval(pb.tree.branch, pb.gene_rate)["lnL"] == base


def test_node_posterior_reproduces_the_reference_reconstruction():
    """... and against the reference's own marginal reconstruction of brown.nuc (tests/golden/brown_hky85_anc.json)."""
    from test_oracle_golden import _brown_anc
    g, pb, raw = _brown_anc()
    eng = engine_for(pb)
    for k, node in enumerate(g["nodes_1based"]):
        post = eng.node_posterior(node - 1, pb.tree.branch)
        for h, patt in enumerate(raw):
            row = g["patterns"][patt]
            i = int(np.argmax(post[h]))
            assert "TCAG"[i] == row["best"][k] and abs(post[h, i] - row["prob"][k]) < 6e-4


def _branch_model_problem(n, K, seed, n_labels=3, n_genes=1):
    """Branch / branch-site-like set-up (Set_UVR_BranchSite codeml.c:2663, Qfactor_NS_branch treesub.c:7549): every branch
    carries a label, and (gene, class, label) selects one of several eigen systems plus a rate factor."""
    pb = helpers.random_problem(n, 10, 160, K=K, seed=seed, n_genes=n_genes)
    rng = np.random.default_rng(seed)
    pb.tree.label = rng.integers(0, n_labels, pb.tree.n_nodes).astype(np.int32)
    pb.tree.label[:3] = np.arange(3) % n_labels                       # every label occurs
    eig = [helpers.random_problem(n, 4, 4, seed=seed + 100 + i).eigen[0] for i in range(K * n_labels)]
    pb.eigen = [pb.eigen[0]] + eig
    pb.eigen_of = rng.integers(0, len(pb.eigen), size=(pb.n_genes, K, n_labels)).astype(np.int32)
    pb.qfactor = 0.5 + rng.random((K, n_labels))
    return pb


@pytest.mark.parametrize("n,K,genes,jit", [(4, 2, 1, False), (61, 3, 1, False), (61, 2, 2, True), (20, 1, 1, False)])
def test_branch_labels_select_eigen_systems(n, K, genes, jit):
    """eigen_of[gene][class][label] and qfactor[class][label]: full evaluation, a batch, the branch-local derivatives (the
    branch's own label picks its eigen system) against the oracle."""
    from paml_amd.engine import JIT
    pb = _branch_model_problem(n, K, 300 + n + K, n_genes=genes)
    eng, out, ref = check(pb, flags=JIT if jit else 0)
    t = pb.tree
    for b in (1, t.n_tips + 2):
        ts = np.array([t.branch[b], 0.3])
        l, dl, ddl = eng.eval_branch(b, ts, t.branch, pb.gene_rate)
        rl, rdl, rddl = oracle.eval_branch(pb, b, ts)
        assert np.allclose(l, rl, rtol=1e-11, atol=0) and np.allclose(dl, rdl, rtol=1e-9, atol=1e-9) and np.allclose(ddl, rddl, rtol=1e-9, atol=1e-8)
    got = eng.eval_batch(np.stack([t.branch, t.branch]), gene_rate=np.tile(pb.gene_rate, (2, 1)))
    assert got[0] == got[1] and abs(got[0] - ref["lnL"]) <= 1e-10 * abs(ref["lnL"])


def _lexsort_compress(chars, gene=None):
    """numpy restatement of PatternWeight's result: patterns in sorted order of (gene, column bytes), first site, count, pose."""
    n_seq, n_sites = chars.shape[0], chars.shape[1]
    cols = chars.reshape(n_seq, n_sites, -1).transpose(1, 0, 2).reshape(n_sites, -1)          # [site][key bytes]
    keys = [cols[:, k] for k in range(cols.shape[1] - 1, -1, -1)]
    if gene is not None:
        keys.append(gene)
    order = np.lexsort(keys)                                                                     # stable: ties stay in site order
    sc = cols[order]
    head = np.ones(n_sites, dtype=bool)
    head[1:] = (sc[1:] != sc[:-1]).any(axis=1)
    if gene is not None:
        head[1:] |= gene[order][1:] != gene[order][:-1]
    pid = np.cumsum(head) - 1
    pose = np.empty(n_sites, dtype=np.int64)
    pose[order] = pid
    return order[head], np.bincount(pid).astype(float), pose


@pytest.mark.parametrize("n_seq,n_sites,width,alphabet,genes", [(5, 1000, 1, 4, 0), (16, 200_000, 3, 4, 0), (7, 50_001, 1, 20, 3),
                                                                 (32, 300_000, 1, 2, 0), (3, 2049, 3, 17, 2), (2, 1, 1, 4, 0)])
def test_compress_patterns_matches_lexsort(n_seq, n_sites, width, alphabet, genes):
    """paml_amd_compress_patterns (radix sort of the alignment columns on the device) against numpy's lexsort: the same patterns
    in the same order, the same first site per pattern, counts and site -> pattern map; few-symbol alignments give heavy
    duplication, 16 x 3 random bases almost none; tile boundaries (2049 sites), one site, and gene partitions."""
    rng = np.random.default_rng(n_sites + n_seq)
    symbols = rng.choice(np.arange(33, 127), size=alphabet, replace=False).astype(np.uint8)
    base = symbols[rng.integers(0, alphabet, size=(1, n_sites, width))]
    chars = np.where(rng.random((n_seq, n_sites, width)) < 0.15, symbols[rng.integers(0, alphabet, size=(n_seq, n_sites, width))], base)
    chars = np.ascontiguousarray(chars.astype(np.uint8))
    gene = rng.integers(0, genes, n_sites).astype(np.int32) if genes else None
    from paml_amd.engine import compress_patterns
    got = compress_patterns(chars if width > 1 else chars[:, :, 0], gene)
    first, w, pose = _lexsort_compress(chars, gene)
    assert len(got["first_site"]) == len(first)
    assert np.array_equal(got["first_site"], first) and np.array_equal(got["weights"], w) and np.array_equal(got["pose"], pose)
    assert got["weights"].sum() == n_sites


def test_eval_device_pipeline_matches_in_order_evaluations():
    """Consecutive eval_device calls overlap the next evaluation's P(t) kernel (side stream, alternate P buffers) with the
    previous pruning kernel.  Twelve evaluations with different branch lengths queued without any host synchronisation, other
    entry points interleaved (they drop the engine back to in-order execution), must each equal the plain evaluation."""
    import torch
    pb = helpers.random_problem(61, 12, 3000, K=2, seed=123)
    eng = engine_for(pb)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    rng = np.random.default_rng(5)
    brs = [pb.tree.branch * rng.uniform(0.5, 1.5, pb.tree.n_nodes) for _ in range(12)]
    want = [eng.eval(b, pb.gene_rate)["lnL"] for b in brs]
    out = torch.zeros(12, dtype=torch.float64, device="cuda")
    for i, b in enumerate(brs):
        eng.eval_device(b, out.data_ptr() + 8 * i, pb.gene_rate)
        if i == 4:
            eng.get_pmat(0, 0, 3)                       # another entry point in between: back to in-order for one call
        if i == 8:
            assert abs(eng.eval(brs[2], pb.gene_rate)["lnL"] - want[2]) <= 1e-12 * abs(want[2])
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert np.max(np.abs(got - np.array(want)) / np.abs(want)) <= 1e-13
    # the P(t) a caller reads back is the last evaluation's, whichever buffer set it landed in
    P = eng.get_pmat(0, 0, 3)
    eng.eval(brs[-1], pb.gene_rate)
    assert np.array_equal(P, eng.get_pmat(0, 0, 3))


@pytest.mark.parametrize("n,K,n_patt", [(61, 2, 60000), (20, 2, 60000)])
def test_eval_device_reduction_on_the_side_stream(n, K, n_patt):
    """Large problems on the matrix-core kernels, runs of eval_device calls: ten queued evaluations with different branch lengths,
    paml_amd_flush, a synchronisation of the STREAM only — every value has the bits of the plain evaluation, whether the pruning
    kernels of consecutive evaluations alternate between two streams (the default: two slots of class likelihoods / partial sums),
    stay on one, or the reduction of evaluation i moves to the engine's side stream (PAML_AMD_OFFLOAD=1)."""
    import os
    import torch
    pb = helpers.random_problem(n, 8, n_patt, K=K, seed=77)
    rng = np.random.default_rng(9)
    brs = [pb.tree.branch * rng.uniform(0.7, 1.3, pb.tree.n_nodes) for _ in range(10)]
    ref = engine_for(pb)
    want = [ref.eval(b, pb.gene_rate)["lnL"] for b in brs]
    sub = pb.slice_patterns(0, 2000)
    assert abs(engine_for(sub).eval(brs[0], pb.gene_rate)["lnL"] - oracle.evaluate(_with_branches(sub, brs[0]))["lnL"]) <= 1e-9 * abs(want[0])
    # default: the evaluations of the run alternate between two pruning streams; PAML_AMD_DUAL=0: one pruning stream;
    # PAML_AMD_OFFLOAD: one pruning stream, the whole reduction on the side stream (an experiment kept behind the switch)
    # "comm": the same inside a one-rank RCCL communicator (the exchange step of every evaluation on the side stream as well)
    dual0, off1 = ("PAML_AMD_DUAL", "0"), ("PAML_AMD_OFFLOAD", "1")
    lanes3 = ("PAML_AMD_LANES", "3")      # (an experiment switch: three evaluations in flight, three slots)
    for comm, env in ((False, None), (False, dual0), (False, off1), (True, None), (True, dual0), (False, lanes3), (True, lanes3)):
        if env:
            os.environ[env[0]] = env[1]
        try:
            eng = engine_for(pb)
        finally:
            if env:
                os.environ.pop(env[0], None)
        if comm:
            from paml_amd import engine as E
            eng.comm_init(0, 1, E.comm_unique_id(), pb.n_patt, 0)
        st = torch.cuda.Stream()
        eng.set_stream(st.cuda_stream)
        out = torch.zeros(10, dtype=torch.float64, device="cuda")
        for i, b in enumerate(brs):
            eng.eval_device(b, out.data_ptr() + 8 * i, pb.gene_rate)
        eng.flush()
        st.synchronize()
        assert out.cpu().numpy().tolist() == want, (comm, env)
        # another entry point after the run joins the side stream by itself
        for i, b in enumerate(brs[:3]):
            eng.eval_device(b, out.data_ptr() + 8 * i, pb.gene_rate)
        assert eng.eval(brs[5], pb.gene_rate)["lnL"] == want[5]
        assert len(eng.partial_sums()) >= 1
        eng.close()


def _with_branches(pb, br):
    q = copy.copy(pb)
    q.tree = copy.deepcopy(pb.tree)
    q.tree.branch = np.asarray(br, dtype=np.float64).copy()
    return q


# ---- pattern shards over several GPUs: the engine's own exchange step (paml_amd_comm_*) ------------------------------------
def _stage2(partials):
    """reduce_stage2's fixed-order total (pure additions: exactly reproducible on the host)."""
    from paml_amd import distributed
    return distributed.total_fixed_order(partials)


@pytest.mark.parametrize("n,K,genes", [(61, 1, 1), (4, 4, 1), (20, 2, 1), (4, 2, 3), (61, 2, 2)])
def test_sharded_partial_sums_are_world_size_invariant(n, K, genes):
    """Shard engines that know the global pattern range leave their partial sums at global positions; added up (disjoint,
    zero elsewhere: exact) and totalled in the fixed order they give the one-engine lnL bit for bit, for 2, 3 and 5 shards.
    With several genes (option G) the gene boundaries stay where they are in the global range: a shard holds the part of every
    gene inside it, for some shards nothing of a gene."""
    from paml_amd import distributed
    pb = helpers.random_problem(n, 10, 3000, K=K, seed=400 + n, n_genes=genes)
    if genes > 1:
        pb.gene_off = np.array([0, 450, 3000] if genes == 2 else [0, 450, 1100, 3000], dtype=np.int32)
    full = engine_for(pb)
    lnl = full.eval(pb.tree.branch, pb.gene_rate)["lnL"]
    assert abs(lnl - oracle.evaluate(pb, want_lnf=False)["lnL"]) <= 1e-10 * abs(lnl)
    pf = full.partial_sums()
    assert len(pf) == -(-pb.n_patt // distributed.red_chunk(pb.n_patt)) and _stage2(pf) == lnl
    for world in (2, 3, 5):
        tot = np.zeros_like(pf)
        for r in range(world):
            lo, hi = distributed.shard_bounds(pb.n_patt, world, r)
            sub = pb.slice_patterns(lo, hi)
            e = engine_for(sub, flags=SHARD)
            e.comm_init(0, 1, None, pb.n_patt, lo)      # global chunking only: no communicator on a one-GPU box
            local = e.eval(sub.tree.branch, sub.gene_rate)["lnL"]
            ps = e.partial_sums()
            assert _stage2(ps) == local
            assert np.count_nonzero(ps) <= -(-(hi - lo) // distributed.red_chunk(pb.n_patt))
            tot += ps
            e.close()
        assert np.array_equal(tot, pf)
        assert _stage2(tot) == lnl


@pytest.mark.parametrize("n,n_tips,n_patt,K", [(61, 10, 9000, 2), (4, 14, 40000, 3), (20, 9, 12000, 1)])
def test_branch_local_sums_are_world_size_invariant(n, n_tips, n_patt, K):
    """eval_branch forms lnL, dlnL, ddlnL from per-block partial sums at global block positions: the arrays of the shards of a
    2- and a 3-way split add up (adding zeros is exact) to the one-GPU array, so the fixed-order total has the same bits for
    every number of ranks; and the one-rank RCCL path returns those bits."""
    from paml_amd import distributed, engine as E
    pb = helpers.random_problem(n, n_tips, n_patt, K=K, seed=910 + n)
    b = pb.tree.n_tips + 2
    ts = np.array([pb.tree.branch[b], pb.tree.branch[b] * 1.7 + 0.01])
    e0 = engine_for(pb)
    want = e0.eval_branch(b, ts, pb.tree.branch)
    full = e0.branch_partials()
    assert full.shape[1] == 6
    for o in range(6):
        assert distributed.total_fixed_order(full[:, o]) == (want[o % 3][o // 3])
    for world in (2, 3):
        tot = np.zeros_like(full)
        for rank in range(world):
            lo, hi = distributed.shard_bounds(pb.n_patt, world, rank)
            if hi == lo:
                continue
            sub = pb.slice_patterns(lo, hi)
            e = engine_for(sub, flags=SHARD)
            e.comm_init(0, 1, None, pb.n_patt, lo)      # shard geometry only
            e.eval_branch(b, ts, sub.tree.branch)
            tot += e.branch_partials()
            e.close()
        assert np.array_equal(tot, full)
    e1 = engine_for(pb)
    e1.comm_init(0, 1, E.comm_unique_id(), pb.n_patt, 0)
    got = e1.eval_branch(b, ts, pb.tree.branch)
    for o in range(3):
        assert np.array_equal(got[o], want[o])


def test_one_rank_rccl_communicator_matches_golden():
    """The RCCL path in a one-rank communicator: ncclCommInitRank + ncclAllReduce on the engine's stream, same bits as the
    plain engine, and the reference's lnL for the golden data."""
    from paml_amd import engine as E
    g = helpers.load_golden("syn_codon_m0")
    pb = helpers.problem_from_golden(g)
    plain = engine_for(pb).eval(pb.tree.branch)["lnL"]
    eng = engine_for(pb)
    eng.comm_init(0, 1, E.comm_unique_id(), pb.n_patt, 0)
    out = eng.eval(pb.tree.branch, want_lnf=True)
    assert out["lnL"] == plain
    assert abs(out["lnL"] - g["lnL"]) <= 2e-6 + 1e-9 * abs(g["lnL"])
    # batched evaluations and the branch-local evaluation go through the collective too
    B = np.stack([pb.tree.branch, pb.tree.branch * 1.1])
    lb = eng.eval_batch(B)
    assert lb[0] == plain and lb[1] != plain
    b = pb.tree.n_tips + 1
    l, dl, ddl = eng.eval_branch(b, np.array([pb.tree.branch[b]]), pb.tree.branch)
    assert abs(l[0] - plain) <= 1e-11 * abs(plain)
    info = eng.comm_info()
    assert info["world"] == 1 and info["n_patt_global"] == pb.n_patt
    eng.comm_destroy()
    assert eng.eval(pb.tree.branch)["lnL"] == plain
    # misaligned shards are refused
    e2 = engine_for(pb.slice_patterns(0, 300))
    assert e2._L.paml_amd_comm_init(e2._h, 0, 2, None, pb.n_patt, 0) != 0


def test_two_engines_on_two_threads_and_streams():
    """include/paml_amd.h: different engines are independent — two host threads, each with its own engine and HIP stream,
    evaluating concurrently (ctypes releases the GIL) get the values a single thread gets."""
    import threading
    import torch
    pbs = [helpers.random_problem(61, 12, 20000, K=1, seed=71), helpers.random_problem(4, 20, 50000, K=4, seed=72)]
    want = [engine_for(p).eval(p.tree.branch)["lnL"] for p in pbs]
    got = [[], []]
    errs = []

    def work(i):
        try:
            st = torch.cuda.Stream()
            e = engine_for(pbs[i])
            e.set_stream(st.cuda_stream)
            for _ in range(30):
                got[i].append(e.eval(pbs[i].tree.branch)["lnL"])
            e.close()
        except Exception as ex:      # noqa: BLE001
            errs.append(ex)
    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for i in range(2):
        assert got[i] == [want[i]] * 30


@pytest.mark.parametrize("n,K,scale", [(4, 3, None), (61, 2, None), (61, 1, 3), (20, 2, None)])
def test_eval_branch_partial_cache_cycles_like_minbranches(n, K, scale):
    """The resident partials of paml_amd_eval_branch (what updateconP + com.oldconP save the reference, treesub.c:7982,
    treespace.c:250): cycling through the branches as minbranches does (treesub.c:8081-8095), changing one length after
    another, every l / dl / ddl equals the oracle's for the current lengths, and after the first call only the nodes on the
    path between consecutive branches are recomputed — far fewer than a full tree per call."""
    pb = helpers.random_problem(n, 14, 200, K=K, seed=300 + n + K, scale_every=scale)
    eng = engine_for(pb)
    t = pb.tree
    order = []

    def pre(i):                       # tree.branches order: first appearance in the Newick string = pre-order
        for c in t.sons[i]:
            order.append(c)
            pre(c)
    pre(t.root)
    rng = np.random.default_rng(1)
    n_int = t.n_nodes - t.n_tips
    first = None
    for cycle in range(2):
        for b in order:
            ts = np.array([t.branch[b], t.branch[b] * 1.3 + 0.01])
            l, dl, ddl = eng.eval_branch(b, ts, t.branch, pb.gene_rate)
            rl, rdl, rddl = oracle.eval_branch(pb, b, ts)
            assert np.allclose(l, rl, rtol=1e-11, atol=0), (cycle, b, l, rl)
            assert np.allclose(dl, rdl, rtol=1e-9, atol=1e-9)
            assert np.allclose(ddl, rddl, rtol=1e-9, atol=1e-8)
            if first is None:
                first = eng.branch_counters()["n_nodes"]
                assert first == n_int                      # the first call forms every partial once
            t.branch[b] = float(ts[1]) if rng.random() < 0.7 else t.branch[b]      # "Newton step": the length moves on
    c = eng.branch_counters()
    calls_after_first = c["n_calls"] - 1
    assert c["n_nodes"] - first <= 2.5 * calls_after_first < n_int * calls_after_first / 2, c
    # an ordinary evaluation still works and agrees with the branch-local value at the current lengths
    base = eng.eval(t.branch, pb.gene_rate)["lnL"]
    b = order[3]
    l, _, _ = eng.eval_branch(b, np.array([t.branch[b]]), t.branch, pb.gene_rate)
    assert abs(l[0] - base) <= 1e-11 * abs(base)
    # changing a far-away length invalidates only the partials that look across that branch
    far = order[-1]
    t.branch[far] *= 1.5
    before = eng.branch_counters()["n_nodes"]
    l, _, _ = eng.eval_branch(b, np.array([t.branch[b]]), t.branch, pb.gene_rate)
    assert abs(l[0] - oracle.eval_branch(pb, b, np.array([t.branch[b]]))[0][0]) <= 1e-11 * abs(base)
    assert 0 < eng.branch_counters()["n_nodes"] - before < n_int


@pytest.mark.parametrize("K,amb,scale", [(1, False, None), (1, True, None), (3, False, None), (2, True, 3)])
def test_eval_branch_eigen_basis_walk_cache_and_many_trial_lengths(K, amb, scale, monkeypatch):
    """61 states: the contraction in the eigen basis (kernels_branch.h).  A minbranches-style walk with several calls per branch —
    the first forms the coefficients (A from its sons inside the kernel when only its orientation changed), the later ones are
    served from them — with 1, 4 and 7 trial lengths (more than one launch handles), against the oracle and against the
    P / dP / ddP form of the same engine build (PAML_AMD_NO_BRANCH_EIG)."""
    pb = helpers.random_problem(61, 11, 700, K=K, seed=900 + K, ambiguity=amb, scale_every=scale)
    eng = engine_for(pb)
    monkeypatch.setenv("PAML_AMD_NO_BRANCH_EIG", "1")
    old = engine_for(pb)
    monkeypatch.delenv("PAML_AMD_NO_BRANCH_EIG")
    t = pb.tree
    order = []

    def pre(i):
        for c in t.sons[i]:
            order.append(c)
            pre(c)
    pre(t.root)
    rng = np.random.default_rng(3)
    calls = 0
    for b in order:
        for nt in (1, 4, 7):
            ts = np.concatenate([[t.branch[b]], t.branch[b] * rng.uniform(0.3, 2.0, nt - 1) + 1e-3])
            l, dl, ddl = eng.eval_branch(b, ts, t.branch, pb.gene_rate)
            rl, rdl, rddl = oracle.eval_branch(pb, b, ts)
            ol, odl, oddl = old.eval_branch(b, ts, t.branch, pb.gene_rate)
            calls += 1
            assert np.allclose(l, rl, rtol=1e-11, atol=0), (b, nt, l, rl)
            assert np.allclose(dl, rdl, rtol=1e-9, atol=1e-9), (b, nt, dl, rdl)
            assert np.allclose(ddl, rddl, rtol=1e-9, atol=1e-8), (b, nt, ddl, rddl)
            assert np.allclose(l, ol, rtol=1e-12, atol=0) and np.allclose(dl, odl, rtol=1e-9, atol=1e-9) and np.allclose(ddl, oddl, rtol=1e-9, atol=1e-8)
        # the same trial lengths again: served from the coefficients, bit for bit what the forming call returned for them
        l2, dl2, ddl2 = eng.eval_branch(b, ts[:4], t.branch, pb.gene_rate)
        calls += 1
        l4, dl4, ddl4 = eng.eval_branch(b, ts[:4], t.branch, pb.gene_rate)
        calls += 1
        assert (l2 == l4).all() and (dl2 == dl4).all() and (ddl2 == ddl4).all()
        assert np.allclose(l2, l[:4], rtol=1e-13, atol=0)
        t.branch[b] = float(ts[-1]) if rng.random() < 0.6 else t.branch[b]
    c = eng.branch_counters()
    assert c["n_calls"] == calls and c["coef_hits"] == calls - len(order), c      # one forming call per branch
    assert old.branch_counters()["coef_hits"] == 0
    base = eng.eval(t.branch, pb.gene_rate)["lnL"]
    l, _, _ = eng.eval_branch(order[2], np.array([t.branch[order[2]]]), t.branch, pb.gene_rate)
    assert abs(l[0] - base) <= 1e-11 * abs(base)


@pytest.mark.parametrize("K,amb,scale", [(1, False, None), (2, True, None), (1, False, 4)])
def test_eval_branch_refill_on_a_per_tree_kernel(K, amb, scale):
    """Round 6: when every branch length has moved since the resident partials were formed (minB's round after ming2 moved kappa / omega;
    updateconP treesub.c:7982 then recomputes every node) the branch-local evaluation's forest of dirty subtrees runs on a per-tree kernel of
    its own — STOREs in the resident layout from 128-pattern tiles — instead of the interpreter.  Forced here (PAML_AMD_JIT flag: compiled
    while the caller waits); l, l', l'' against the oracle at two branches after two refills, equal to the interpreter engine's values."""
    pb = helpers.random_problem(61, 12, 900, K=K, seed=1300 + K, ambiguity=amb, scale_every=scale)
    t = pb.tree
    eng, ref_eng = engine_for(pb, flags=JIT), engine_for(pb)
    rng = np.random.default_rng(9)
    internal = [v for v in range(t.n_tips, t.n_nodes) if v != t.root]
    for rnd in range(2):
        t.branch[:] = np.where(np.arange(t.n_nodes) == t.root, 0.0, t.branch * rng.uniform(0.7, 1.4, t.n_nodes))      # every length moves: a refill
        for b in (internal[1], 2):
            ts = np.array([t.branch[b], 0.04, 0.6])
            l, dl, ddl = eng.eval_branch(b, ts, t.branch, pb.gene_rate)
            if b == internal[1]:
                assert eng.kernel_name == "mfma64_jit" and eng.branch_counters()["refill_kernels"] == rnd + 1, eng.kernel_name      # (the refill's pruning kernel)
            rl, rdl, rddl = oracle.eval_branch(pb, b, ts)
            il, idl, iddl = ref_eng.eval_branch(b, ts, t.branch, pb.gene_rate)
            assert np.allclose(l, rl, rtol=1e-11, atol=0), (rnd, b, l, rl)
            assert np.allclose(dl, rdl, rtol=1e-9, atol=1e-9) and np.allclose(ddl, rddl, rtol=1e-9, atol=1e-8)
            assert np.allclose(l, il, rtol=1e-12, atol=0) and np.allclose(dl, idl, rtol=1e-9, atol=1e-9)
    assert abs(eng.eval(t.branch, pb.gene_rate)["lnL"] - oracle.evaluate(pb)["lnL"]) <= 1e-10 * abs(oracle.evaluate(pb)["lnL"])


def test_eval_branch_at_1e5_patterns_against_the_oracle():
    """The branch-local evaluation at a size where every workgroup walks several chunks: 16 taxa x 131 072 codon patterns, a tip
    branch, an internal branch and a move to its neighbour (A re-formed inside the kernel), four trial lengths, against the oracle."""
    pb = synth.codon_m0_problem(n_tips=16, n_patt=131_072)
    eng = engine_for(pb)
    t = pb.tree
    base = eng.eval(t.branch)["lnL"]
    internal = [v for v in range(t.n_tips, t.n_nodes) if v != t.root]
    for b in (3, internal[0], internal[1], internal[-1]):
        ts = np.array([t.branch[b], 0.5 * t.branch[b], 0.05, 0.8])
        l, dl, ddl = eng.eval_branch(b, ts, t.branch)
        rl, rdl, rddl = oracle.eval_branch(pb, b, ts)
        assert np.allclose(l, rl, rtol=1e-12, atol=0), (b, l, rl)
        assert np.allclose(dl, rdl, rtol=1e-9, atol=1e-7) and np.allclose(ddl, rddl, rtol=1e-9, atol=1e-6), (b, dl, rdl, ddl, rddl)
        assert abs(l[0] - base) <= 1e-12 * abs(base)
    assert eng.branch_counters()["n_nodes"] < 2 * (t.n_nodes - t.n_tips)


def test_eval_branch_full_size_c4_reproduces_the_reference_lnl():
    """BASELINE configs[3] at full size through the branch-local path: l(t_current) on any branch is the tree's lnL, which the
    unmodified reference printed for this data (-15822122.733473, golden syn_codon_m0_full); 7.2 GB of partials resident, the walk
    re-forms one or two nodes per call, the derivatives agree with central differences of l along the branch."""
    g = helpers.load_golden("syn_codon_m0_full")
    pb = helpers.problem_from_golden(g)
    assert pb.n_patt == 1_000_000
    eng = engine_for(pb)
    t = pb.tree
    internal = [v for v in range(t.n_tips, t.n_nodes) if v != t.root]
    for b in (0, internal[0], internal[1], 7, internal[-1]):
        t0 = t.branch[b]
        hstep = 1e-4
        ts = np.array([t0, t0 - hstep, t0 + hstep])
        l, dl, ddl = eng.eval_branch(b, ts, t.branch)
        assert abs(l[0] - g["lnL"]) <= 2e-6 + 1e-12 * abs(g["lnL"]), (b, l[0], g["lnL"])
        fd1 = (l[2] - l[1]) / (2 * hstep)
        fd2 = (l[2] - 2 * l[0] + l[1]) / hstep ** 2
        # (central differences over 2e-4: their own truncation error, third derivative x h^2 / 6 with a third derivative of ~1e8 on
        #  the short branches, sets the tolerance; the derivatives are pinned by the oracle at 131 072 patterns above)
        assert abs(dl[0] - fd1) <= 1e-3 * abs(fd1) + 1.0, (b, dl[0], fd1)
        assert abs(ddl[0] - fd2) <= 2e-2 * abs(fd2) + 100.0, (b, ddl[0], fd2)
    c = eng.branch_counters()
    assert c["n_nodes"] <= (t.n_nodes - t.n_tips) + 3 * 4


@pytest.mark.parametrize("n,n_tips,n_patt,K,kw", [(61, 13, 79, 1, {}), (61, 10, 300, 3, dict(scale_every=3)), (61, 24, 150, 2, dict(ambiguity=True)),
                                                  (40, 9, 200, 2, {}), (61, 60, 40, 1, dict(scale_every=7, ambiguity=True))])
def test_small_data_cooperative_kernel_has_the_bits_of_the_gather_kernel(n, n_tips, n_patt, K, kw, monkeypatch):
    """Small data sets (every 16-pattern group can have a CU) run prune_mfma64_coop — four waves per group, a row block of every
    product each — instead of one wave per group; per row block the k-blocks accumulate in the same order and the root sum is the
    gather kernel's code, so the two kernels give the SAME bits (the engine changes between them with the size of a launch: an
    evaluation, then a batched gradient), and both agree with the oracle."""
    monkeypatch.setenv("PAML_AMD_JIT", "0")
    pb = helpers.random_problem(n, n_tips, n_patt, K=K, seed=700 + n_tips, **kw)
    eng, out, ref = check(pb)
    assert eng.kernel_name == "mfma64_coop"
    monkeypatch.setenv("PAML_AMD_COOP", "0")
    eng0 = engine_for(pb)
    out0 = eng0.eval(pb.tree.branch, pb.gene_rate, want_lnf=True, want_fhk=True)
    assert eng0.kernel_name == "mfma64_gather"
    assert out0["lnL"] == out["lnL"] and np.array_equal(out0["lnf"], out["lnf"]) and np.array_equal(out0["fhK"], out["fhK"])
    # batched evaluations: two parameter sets still fit the cooperative form, a hundred do not (the same engine changes kernel)
    for nb in (2, 100):
        br = np.stack([pb.tree.branch * (1 + 0.01 * i) for i in range(nb)])
        assert np.array_equal(eng.eval_batch(br), eng0.eval_batch(br))
    # ... and the per-tree form of the cooperative kernel (jit.h: jit_generate_coop — every operand requested straight into registers ahead
    # of its use, mixture + log + the fixed-order sums formed by the workgroup that finishes last: ONE launch after P(t)): the same bits
    # again — lnL, per-pattern values, class likelihoods, batched evaluations (per batch element its own last workgroup)
    monkeypatch.delenv("PAML_AMD_JIT")
    monkeypatch.delenv("PAML_AMD_COOP")
    monkeypatch.setenv("PAML_AMD_JIT_SYNC", "1")      # (compile before the first evaluation instead of beside it)
    engj = engine_for(pb)
    outj = engj.eval(pb.tree.branch, pb.gene_rate, want_lnf=True, want_fhk=True)
    assert engj.kernel_name == "mfma64_coopjit", engj.kernel_name
    assert outj["lnL"] == out0["lnL"] and np.array_equal(outj["lnf"], out0["lnf"]) and np.array_equal(outj["fhK"], out0["fhK"])
    assert engj.eval(pb.tree.branch, pb.gene_rate)["lnL"] == out0["lnL"]              # (no per-pattern outputs asked for)
    for nb in (2, 3, 100):
        br = np.stack([pb.tree.branch * (1 + 0.01 * i) for i in range(nb)])
        gr = None if pb.gene_rate is None else np.stack([pb.gene_rate] * nb)
        vj, lj = engj.eval_batch(br, gene_rate=gr, want_lnf=True)
        v0, l0 = eng0.eval_batch(br, gene_rate=gr, want_lnf=True)
        assert np.array_equal(vj, v0) and np.array_equal(lj, l0), nb
    # runs of evaluations left on the device (what a benchmark loop or an optimiser's independent evaluations queue up)
    import torch
    d = torch.zeros(5, dtype=torch.float64, device="cuda")
    engj.set_stream(torch.cuda.current_stream().cuda_stream)
    for i in range(5):
        engj.eval_device(pb.tree.branch * (1 + 0.01 * i), d.data_ptr() + 8 * i, pb.gene_rate)
    engj.flush()
    torch.cuda.synchronize()
    assert np.array_equal(d.cpu().numpy(), eng0.eval_batch(np.stack([pb.tree.branch * (1 + 0.01 * i) for i in range(5)]), gene_rate=None if pb.gene_rate is None else np.stack([pb.gene_rate] * 5)))


def test_the_cooperative_per_tree_kernel_is_compiled_beside_the_evaluations():
    """Without PAML_AMD_JIT_SYNC the tree's own cooperative kernel is compiled on a worker thread (or read from the code-object cache) while
    the interpreter form serves; the engine changes over when it is there — same value before and after."""
    import time
    pb = helpers.random_problem(61, 11, 120, K=2, seed=4242)      # (a tree no other test compiles: not in the cache the first time)
    eng = engine_for(pb)
    first = eng.eval(pb.tree.branch)["lnL"]
    name0 = eng.kernel_name
    t0 = time.time()
    while eng.kernel_name != "mfma64_coopjit" and time.time() - t0 < 60:
        time.sleep(0.1)
        assert eng.eval(pb.tree.branch)["lnL"] == first
    assert name0 in ("mfma64_coop", "mfma64_coopjit") and eng.kernel_name == "mfma64_coopjit"
    assert eng.eval(pb.tree.branch)["lnL"] == first


def test_small_20_state_data_take_the_matrix_core_interpreter_unless_they_are_a_shard():
    """20 states, at most 4096 patterns: the cooperative MFMA interpreter on zero-padded matrices instead of the scalar-operand kernel
    (a branch in a sixth of the time); its sums are ordered differently, so an engine that holds a SHARD of a larger alignment must
    say so (PAML_AMD_SHARD) — paml_amd_comm_init refuses one that chose by its own size."""
    pb = helpers.random_problem(20, 7, 300, K=4, seed=41)
    eng, out, ref = check(pb)
    assert eng.kernel_name in ("mfma64_coop", "mfma64_coopjit")      # (the tree's own kernel once it is compiled)
    assert eng._L.paml_amd_comm_init(eng._h, 0, 1, None, 100000, 0) != 0 and b"PAML_AMD_SHARD" in eng._L.paml_amd_last_error(eng._h)
    shard, outs, _ = check(pb, flags=SHARD)
    assert shard.kernel_name == "valu20"
    shard.comm_init(0, 1, None, 100000 // 256 * 256 + 300, 100000 // 256 * 256)
    assert abs(outs["lnL"] - out["lnL"]) <= 1e-11 * abs(out["lnL"])
