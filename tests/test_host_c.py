"""The C host (paml_amd/host -> libpamlh.so, pamlh_lnl) reads the reference's own example files — control file,
PHYLIP sequences, Newick tree, dat/lg.dat — and must reproduce the reference's single-evaluation lnL and per-pattern
lnf (same pattern order as the reference's `lnf` file).  CPU: C host + oracle.  GPU: C host + engine, and the
pamlh_lnl driver end to end."""
import os
import re
import subprocess

import numpy as np
import pytest

import helpers
import oracle
from paml_amd import hostlib

CTL = os.path.join(helpers.GOLDEN, "ctl")
CASES = [("brown_hky85", "baseml", "brown_hky85.ctl"), ("stewart_lg_g4", "codeml", "stewart_lg_g4.ctl"),
         ("hiv_m0", "codeml", "hiv_ns0.ctl"), ("hiv_m1a", "codeml", "hiv_ns1.ctl"), ("hiv_m2a", "codeml", "hiv_ns2.ctl"),
         ("hiv_m7", "codeml", "hiv_ns7.ctl"), ("hiv_m8", "codeml", "hiv_ns8.ctl"), ("mhc_m0_scaled", "codeml", "mhc_m0.ctl"),
         ("mhc_m2a", "codeml", "mhc_ns2.ctl"), ("mhc_m8", "codeml", "mhc_ns8.ctl"),      # site classes on a tree with ten scaling nodes
         ("mtcdna_branch", "codeml", "mtcdna_branch.ctl"),
         ("lysos_free", "codeml", "lysos_free.ctl"), ("lysos_branch_fix", "codeml", "lysos_branch_fix.ctl"),
         ("lysos_clade_label", "codeml", "lysos_clade_label.ctl"),      # '$' clade labels, nested, with a '#' inside      # free-ratio model (model = 1): an omega and an eigen system for each of the 11 branches
         # other genetic codes: invertebrate mt (62 sense codons), ciliate nuclear (63), the reference's "regularised" code (64)
         ("hiv_m0_icode4", "codeml", "hiv_ns0_icode4.ctl"), ("hiv_m0_icode5", "codeml", "hiv_ns0_icode5.ctl"), ("hiv_m0_icode11", "codeml", "hiv_ns0_icode11.ctl"),
         # aaDist = 7 (AAClasses): omega by class of amino-acid pair (ctl/OmegaAA.dat), alone and per branch label
         ("mtcdna_aaclass_m0", "codeml", "mtcdna_aaclass_m0.ctl"), ("mtcdna_aaclass_branch", "codeml", "mtcdna_aaclass_branch.ctl"),
         # omega as a function of an amino-acid distance: geometric on Grantham's (aaDist = 1), linear on Miyata's (-2; its slope ends on the bound 1)
         ("mtcdnapri_aadist1", "codeml", "mtcdnapri_aadist1.ctl"), ("mtcdnapri_aadist_m2", "codeml", "mtcdnapri_aadist_m2.ctl"),
         # codon-based amino-acid models: 6 = FromCodon (20 states), 5 = FromCodon0 (60 codon states, amino acids as codon sets)
         ("mtcdnapri_fromcodon", "codeml", "mtcdnapri_fromcodon.ctl"), ("mtcdnapri_fromcodon0", "codeml", "mtcdnapri_fromcodon0.ctl"),
         # general reversible amino-acid models: REVaa_0 (69 exchangeabilities under the mt code) and REVaa (189)
         ("mtcdnapri_revaa0", "codeml", "mtcdnapri_revaa0.ctl"), ("mtcdnapri_revaa", "codeml", "mtcdnapri_revaa.ctl"),
         # branch-site A (alternative and null), B; clade C, D; M3 — goldens at the reference's own 6-decimal MLEs
         ("lyso_bsa", "codeml", "lyso_bsa.ctl"), ("lyso_bsa_null", "codeml", "lyso_bsa_null.ctl"), ("lyso_bsb", "codeml", "lyso_bsb.ctl"),
         ("ecp_cmc", "codeml", "ecp_cmc.ctl"), ("ecp_cmd", "codeml", "ecp_cmd.ctl"), ("hiv_m3", "codeml", "hiv_ns3.ctl"), ("hiv_m4", "codeml", "hiv_ns4.ctl"), ("hiv_m5", "codeml", "hiv_ns5.ctl"), ("hiv_m6", "codeml", "hiv_ns6.ctl"), ("hiv_m9", "codeml", "hiv_ns9.ctl"),
         ("hiv_m10", "codeml", "hiv_ns10.ctl"), ("hiv_m11", "codeml", "hiv_ns11.ctl"), ("hiv_m12", "codeml", "hiv_ns12.ctl"),
         ("hiv_m13", "codeml", "hiv_ns13.ctl"), ("ecp_m2arel", "codeml", "ecp_m2arel.ctl"),
         ("brown_hky85_clock", "baseml", "brown_hky85_clock.ctl"),      # global clock: x holds the node ages
         ("brown_hky85_clock2", "baseml", "brown_hky85_clock2.ctl"),    # local clocks: ages, then the rates of the '#' branch classes
         ("hiv2_tipdate", "baseml", "hiv2_tipdate.ctl"),                # TipDate: dated tips, ages in time units, then the mutation rate
         ("hiv2_tipdate_clock2", "baseml", "hiv2_tipdate_clock2.ctl"),  # ... and a second (absolute) rate for a labelled clade
         ("brown_f84", "baseml", "brown_f84.ctl"), ("brown_t92_g4", "baseml", "brown_t92_g4.ctl"), ("brown_unrest", "baseml", "brown_unrest.ctl"), ("brown_hky85_nhomo1", "baseml", "brown_hky85_nhomo1.ctl"),
         # non-homogeneous models: a kappa per branch (2); frequency sets per branch (3: tips / internal / root, 4: every node), every
         # branch with its own eigen system (one label per node); the nhomo3 estimate has a frequency on the boundary (0.000000)
         ("brown_hky85_nhomo2", "baseml", "brown_hky85_nhomo2.ctl"), ("brown_hky85_nhomo3", "baseml", "brown_hky85_nhomo3.ctl"),
         ("brown_f84_nhomo4", "baseml", "brown_f84_nhomo4.ctl"), ("brown_t92_nhomo3_g4", "baseml", "brown_t92_nhomo3_g4.ctl"),
         ("brown_hky85_nhomo5", "baseml", "brown_hky85_nhomo5.ctl"),      # frequency sets and kappas by the tree's '#' labels, the root a set of its own ("mhc_m0_prop", "codeml", "mhc_m0_prop.ctl"), ("stewart_eqinput", "codeml", "stewart_eqinput.ctl"),
         ("hiv_m0_f3x4mg", "codeml", "hiv_ns0_cf5.ctl"), ("hiv_m0_f1x4mg", "codeml", "hiv_ns0_cf4.ctl"),      # Muse-Gaut style rates
         # mutation-selection models FMutSel0 / FMutSel (mutation bias + amino-acid / codon fitnesses, implied by the observed frequencies or
         # estimated) and codon frequencies as parameters (estFreq = 1); the observed codon table of these 91 codons has 12 zeros
         ("hiv_fmutsel0", "codeml", "hiv_fmutsel0.ctl"), ("hiv_fmutsel0_est", "codeml", "hiv_fmutsel0_est.ctl"), ("hiv_fmutsel", "codeml", "hiv_fmutsel.ctl"),
         ("hiv_fmutsel_est", "codeml", "hiv_fmutsel_est.ctl"), ("hiv_f3x4_est", "codeml", "hiv_f3x4_est.ctl"), ("hiv_f1x4mg_est", "codeml", "hiv_f1x4mg_est.ctl"),
         ("hiv_fmutsel0_m2a", "codeml", "hiv_fmutsel0_m2a.ctl"), ("hiv_f3x4_est_m7", "codeml", "hiv_f3x4_est_m7.ctl"),      # ... under site models
         # option G (several genes): rates only (Mgene 0), + frequencies (2), + kappa / omega (3), both (4); one with gamma
         ("horai_mg0", "baseml", "horai_mg0.ctl"), ("horai_mg2", "baseml", "horai_mg2.ctl"), ("horai_mg3", "baseml", "horai_mg3.ctl"),
         ("horai_mg4", "baseml", "horai_mg4.ctl"), ("horai_mg0_g5", "baseml", "horai_mg0_g5.ctl"),
         ("horai_mg0_malpha", "baseml", "horai_mg0_malpha.ctl"), ("horai_mg4_malpha", "baseml", "horai_mg4_malpha.ctl"),      # Malpha: a gamma shape per gene
         ("lysin_mg0", "codeml", "lysin_mg0.ctl"), ("lysin_mg2", "codeml", "lysin_mg2.ctl"), ("lysin_mg3", "codeml", "lysin_mg3.ctl"),
         ("lysin_mg4", "codeml", "lysin_mg4.ctl")]


def _x(g, a):
    return np.array(g.get("x", []), dtype=float) if a.np else np.zeros(0)


@pytest.mark.parametrize("gname,prog,ctl", CASES)
def test_c_host_reproduces_reference_on_cpu(gname, prog, ctl):
    g = helpers.load_golden(gname)
    a = hostlib.Analysis(os.path.join(CTL, ctl), prog)
    assert (a.n_patt, a.ls, a.n_tips) == (g["n_patt"], g["ls"], g["n_tips"])
    if "x" in g and a.np:
        assert a.np == len(g["x"]) and a.ntime == g.get("ntime", a.ntime)
    assert len(a.default_x()) == a.np      # initial values, bounds and names cover every parameter
    lo, hi = a.bounds()
    assert len(lo) == len(hi) == a.np and (lo <= hi).all()
    pb = a.problem(_x(g, a))
    assert np.array_equal(pb.weights, np.array(g["counts"]))          # same patterns, same order, same counts
    r = oracle.evaluate(pb)
    assert abs(r["lnL"] - g["lnL"]) <= 2e-6
    assert np.max(np.abs(r["lnf"] - np.array(g["logf"]))) < 5e-8
    if g.get("scale_nodes"):
        assert sorted(np.nonzero(pb.scale_node)[0] + 1) == sorted(g["scale_nodes"])   # SetNodeScale picks the same nodes
    if g.get("published_lnL") is not None:
        assert abs(r["lnL"] - g["published_lnL"]) < 5e-6


HORAI_SEPARATE = [(1367, 91, -3355.227556, "0.029574 0.015898 0.019903 0.007940 0.007923 0.006142 0.027475 0.070614 0.052709 11.268489"),
                  (1367, 53, -2459.021781, "0.010783 0.003663 0.007261 0.003622 0.001699 0.003451 0.008605 0.030564 0.022597 9.272028"),
                  (1367, 203, -5637.976052, "0.151172 0.089850 0.194491 0.092983 0.065257 0.039448 0.206379 0.467630 0.746578 29.275418"),
                  (759, 62, -1794.493564, "0.015755 0.008194 0.023548 0.007091 0.008929 0.007257 0.032252 0.061925 0.073694 27.574639")]


def test_c_host_separate_gene_analyses_on_cpu(tmp_path):
    """Mgene = 1 on examples/horai.nuc (HKY85): every gene as an analysis of its own — its sites, patterns, frequencies and ten
    parameters.  The reference's per-gene output (ls, npatt, lnL, estimates) is reproduced by one evaluation at its estimates."""
    ctl = tmp_path / "horai_mg1.ctl"
    ctl.write_text(open(os.path.join(CTL, "horai_mg0.ctl")).read().replace("../data/", os.path.join(helpers.GOLDEN, "data") + "/").replace("Mgene = 0", "Mgene = 1"))
    a = hostlib.Analysis(str(ctl), "baseml")
    assert a.n_genes() == 4
    with pytest.raises(RuntimeError, match="separately"):
        a.set_x(np.ones(a.np))
    for g, (ls, npatt, lnl, xs) in enumerate(HORAI_SEPARATE):
        b = a.gene_subset(g)
        assert (b.ls, b.n_patt, b.np, b.ntime) == (ls, npatt, 10, 9)
        x = np.array([float(v) for v in xs.split()])
        assert abs(oracle.evaluate(b.problem(x), want_lnf=False)["lnL"] - lnl) < 5e-5
        b.close()


def test_c_host_auto_discrete_gamma_on_cpu():
    """lfunAdG pinned to the reference binary: brown.nuc, HKY85 + auto-discrete-gamma (alpha and rho free, 4 classes), one
    evaluation at the reference's printed estimates gives its lnL -2620.901026.  The host builds MK from rho the way AutodGamma
    does (AS 70 normal quantile, AS 66 normal integral, Genz's bivariate normal tail); the oracle runs the site chain."""
    g = helpers.load_golden("brown_hky85_adg")
    a = hostlib.Analysis(os.path.join(CTL, "brown_hky85_adg.ctl"), "baseml")
    assert (a.np, a.ntime, a.n_patt, a.ls) == (len(g["x"]), g["ntime"], g["n_patt"], g["ls"])
    x = np.array(g["x"])
    MK = a.adg_matrix(x)
    assert MK.shape == (4, 4) and np.allclose(MK.sum(axis=1), 1, atol=1e-6) and np.allclose(MK, MK.T, atol=1e-7)
    assert abs(oracle.evaluate_adg(a.problem(x), MK, a.pose()) - g["lnL"]) <= 2e-6
    x0 = x.copy(); x0[-1] = 0.0                       # rho = 0: the chain forgets, lfundG's value
    b = hostlib.Analysis(os.path.join(CTL, "brown_hky85_adg.ctl"), "baseml")
    assert b.adg_matrix(x0) is None


@pytest.mark.gpu
def test_c_host_auto_discrete_gamma_on_gpu():
    """... and through the engine (fx_r on the device, the chain over the 895 sites on the host), including the optimiser."""
    g = helpers.load_golden("brown_hky85_adg")
    a = hostlib.Analysis(os.path.join(CTL, "brown_hky85_adg.ctl"), "baseml")
    lnl, _ = a.eval_gpu(np.array(g["x"]), want_lnf=False)
    assert abs(lnl - g["lnL"]) <= 2e-6
    r = a.optimize(a.default_x())
    assert r["converged"] and abs(r["lnL"] - g["mle_lnL"]) < 2e-5, (r["lnL"], g["mle_lnL"])


def _brown_sequences():
    txt = open(os.path.join(helpers.GOLDEN, "data", "brown.nuc")).read().split("\n")
    ns, ls = [int(v) for v in txt[0].split()[:2]]
    names, seqs = [], []
    for ln in txt[1:]:
        if not ln.strip():
            continue
        if len(names) == ns and len(seqs[-1]) >= ls:
            break
        toks = ln.split()
        if not names or len(seqs[-1]) >= ls:
            names.append(toks[0]); seqs.append("".join(toks[1:]))
        else:
            seqs[-1] += "".join(toks)
    return names, seqs, ls


@pytest.mark.parametrize("fmt", ["fasta", "nexus"])
def test_c_host_reads_fasta_and_nexus(tmp_path, fmt):
    """Aligned FASTA and NEXUS sequence files (GetSeqFileType treesub.c:367; the reference binary gives -2665.422858 on these
    renderings of brown.nuc as on the original): the same patterns, counts and lnL as the native file."""
    names, seqs, ls = _brown_sequences()
    f = tmp_path / ("brown." + fmt)
    if fmt == "fasta":
        f.write_text("".join(">%s\n%s\n" % (n, "\n".join(s[i:i + 70] for i in range(0, ls, 70))) for n, s in zip(names, seqs)))
    else:
        f.write_text("#NEXUS\n\nBegin Data;\n  Dimensions ntax=%d nchar=%d;\n  Format datatype=dna missing=? gap=-;\n  Matrix\n" % (len(names), ls) +
                     "".join("%s   %s [a comment]\n" % (n, s) for n, s in zip(names, seqs)) + ";\nEnd;\n")
    ctl = tmp_path / "baseml.ctl"
    ctl.write_text(open(os.path.join(CTL, "brown_hky85.ctl")).read().replace("../data/brown.nuc", str(f))
                   .replace("../data/brown.trees", os.path.join(helpers.GOLDEN, "data", "brown.trees")))
    g = helpers.load_golden("brown_hky85")
    a = hostlib.Analysis(str(ctl), "baseml")
    b = hostlib.Analysis(os.path.join(CTL, "brown_hky85.ctl"), "baseml")
    pa, pb = a.problem(np.array(g["x"])), b.problem(np.array(g["x"]))
    assert np.array_equal(pa.z, pb.z) and np.array_equal(pa.weights, pb.weights) and np.array_equal(pa.pi, pb.pi)
    assert abs(oracle.evaluate(pa)["lnL"] - g["lnL"]) <= 2e-6


def test_c_host_picks_a_tree_of_a_file_with_several(tmp_path):
    """The reference evaluates the trees of the tree file one after the other (Forestry codeml.c:635); here every tree is an
    analysis of its own (pamlh_load_tree).  stewart.trees holds two trees behind a header that announces one: the header wins, as
    in the reference.  With the header corrected the second tree loads and is the same problem as a file that holds only it."""
    data = os.path.join(helpers.GOLDEN, "data")
    base = open(os.path.join(CTL, "stewart_lg_g4.ctl")).read().replace("../data/stewart.aa", os.path.join(data, "stewart.aa")).replace("../data/lg.dat", os.path.join(data, "lg.dat"))
    a = hostlib.Analysis(os.path.join(CTL, "stewart_lg_g4.ctl"), "codeml")
    assert a.n_trees() == 1
    with pytest.raises(RuntimeError, match="holds 1"):
        hostlib.Analysis(os.path.join(CTL, "stewart_lg_g4.ctl"), "codeml", tree_index=1)
    two = open(os.path.join(data, "stewart.trees")).read().replace("6  1", "6  2", 1)
    (tmp_path / "two.trees").write_text(two)
    (tmp_path / "second.trees").write_text(" 6 1\n(((Rat, Horse), Human), Baboon, (Cow, Langur));\n")
    (tmp_path / "two.ctl").write_text(base.replace("../data/stewart.trees", str(tmp_path / "two.trees")))
    (tmp_path / "second.ctl").write_text(base.replace("../data/stewart.trees", str(tmp_path / "second.trees")))
    t0 = hostlib.Analysis(str(tmp_path / "two.ctl"), "codeml")
    t1 = hostlib.Analysis(str(tmp_path / "two.ctl"), "codeml", tree_index=1)
    only = hostlib.Analysis(str(tmp_path / "second.ctl"), "codeml")
    assert t0.n_trees() == t1.n_trees() == 2 and only.n_trees() == 1
    g = helpers.load_golden("stewart_lg_g4")
    x = np.array(g["x"])
    assert abs(oracle.evaluate(t0.problem(x))["lnL"] - g["lnL"]) <= 2e-6          # the first tree is the golden's
    p1, p2 = t1.problem(x), only.problem(x)
    assert [[int(c) for c in ss] for ss in p1.tree.sons] == [[int(c) for c in ss] for ss in p2.tree.sons]
    l1, l2 = oracle.evaluate(p1)["lnL"], oracle.evaluate(p2)["lnL"]
    assert l1 == l2 and abs(l1 - g["lnL"]) > 1e-3                                     # a different topology


@pytest.mark.parametrize("gname,ctl", [("lysos_free", "lysos_free.ctl"), ("lysos_branch_fix", "lysos_branch_fix.ctl"), ("hiv_m0_icode4", "hiv_ns0_icode4.ctl")])
def test_c_host_dn_ds_per_branch_match_the_reference_table(gname, ctl):
    """"dN & dS for each branch" of the reference's main result file (DetailOutput codeml.c:1349-1404: t, N, S, dN/dS, dN, dS printed with
    3 / 1 / 1 / 4 / 4 / 4 decimals) at its estimates: the free-ratio model (eleven omegas, two of them on the bound 999), the three-ratio
    model with the last omega fixed at 1, and M0 under the invertebrate mitochondrial code."""
    g = helpers.load_golden(gname)
    a = hostlib.Analysis(os.path.join(CTL, ctl), "codeml")
    got = a.dnds(np.array(g["x"]))
    ref = np.array(g["dnds"])[:, 2:]
    assert got.shape == ref.shape == (a.n_nodes - 1, 6)
    for col, tol in enumerate((6e-4, 0.06, 0.06, 6e-5, 6e-5, 6e-5)):
        assert np.max(np.abs(got[:, col] - ref[:, col])) < tol, (col, got[:, col], ref[:, col])
    b = hostlib.Analysis(os.path.join(CTL, "hiv_ns2.ctl"), "codeml")
    with pytest.raises(RuntimeError, match="without site classes"):
        b.dnds(b.default_x())


def test_c_host_shards_an_alignment_with_several_genes_on_cpu():
    """pamlh_set_shard on option-G data (horai.nuc: four genes, Mgene = 4: rates, frequencies and kappa per gene): every rank keeps
    its block of the global pattern range with the gene boundaries clipped to it — frequencies and everything else
    estimated from the data were computed from the whole alignment before — and the ranks' per-chunk partial sums add up to the
    one-rank total bit for bit."""
    from paml_amd import distributed
    g = helpers.load_golden("horai_mg4")
    x = np.array(g["x"])
    full = hostlib.Analysis(os.path.join(CTL, "horai_mg4.ctl"), "baseml").problem(x)
    lnf = oracle.evaluate(full)["lnf"]
    ref = distributed.total_fixed_order(distributed.chunk_partials(lnf, full.weights, 0, full.n_patt))
    assert abs(ref - g["lnL"]) <= 2e-6
    for world in (2,):      # 409 patterns are two reduction chunks of 256
        tot = np.zeros(-(-full.n_patt // distributed.red_chunk(full.n_patt)))
        for r in range(world):
            lo, hi = distributed.shard_bounds(full.n_patt, world, r)
            a = hostlib.Analysis(os.path.join(CTL, "horai_mg4.ctl"), "baseml")
            a.set_shard(r, world)
            assert a.n_patt == hi - lo
            pb = a.problem(x)
            want = full.slice_patterns(lo, hi)
            assert np.array_equal(pb.z, want.z) and np.array_equal(pb.weights, want.weights) and np.array_equal(pb.gene_off, want.gene_off)
            assert np.array_equal(pb.pi, full.pi)
            tot += distributed.chunk_partials(oracle.evaluate(pb)["lnf"], pb.weights, lo, full.n_patt)
        assert distributed.total_fixed_order(tot) == ref


def test_tree_comparison_matches_the_reference_table():
    """rell() (treesub.c:5844-6009) restated as pamlh_tree_comparison: fed with the per-pattern log f_h the reference wrote to `lnf` for the
    two trees of stewart.trees (LG + G4, parameters estimated on each), it gives the reference's table — li, Dli, SE and the
    Kishino-Hasegawa P value to the printed digits; the bootstrap columns (10 000 replicates from another generator) to Monte-Carlo accuracy."""
    g = helpers.load_golden("stewart_two_trees")
    r = hostlib.tree_comparison(np.array(g["logf"]), np.array(g["counts"]), seed=7)
    assert r["best"] == 1
    for t, row in enumerate(g["table"]):
        assert abs(r["li"][t] - row["li"]) < 6e-4 and abs(r["dli"][t] - row["dli"]) < 6e-4 and abs(r["se"][t] - row["se"]) < 6e-4
        assert abs(r["pKH"][t] - row["pKH"]) < 6e-4
        assert abs(r["pSH"][t] - row["pSH"]) < 0.012 and abs(r["pRELL"][t] - row["pRELL"]) < 0.012      # 3 sigma of 10 000 draws at p = 0.08
    assert abs(r["pRELL"].sum() - 1) < 1e-9
    r2 = hostlib.tree_comparison(np.array(g["logf"]), np.array(g["counts"]), seed=7)
    assert np.array_equal(r2["pRELL"], r["pRELL"]) and np.array_equal(r2["pSH"], r["pSH"])      # seeded
    # stratified resampling: with the patterns cut into two "genes" the deterministic columns do not move
    r3 = hostlib.tree_comparison(np.array(g["logf"]), np.array(g["counts"]), gene_off=[0, 40, len(g["counts"])], seed=7)
    assert np.array_equal(r3["se"], r["se"]) and abs(r3["pRELL"][0] - r["pRELL"][0]) < 0.02


def test_c_host_control_file_overrides_pick_a_model_of_a_list(tmp_path):
    """A control file listing several site models ("NSsites = 0 1 2 7 8", examples/HIVNSsites/codeml.ctl — the reference runs them in turn)
    loads as its first model; pamlh_load_with replaces options, so every model of the list is one analysis: the same problem as the
    single-model control files of the goldens."""
    ctl = open(os.path.join(CTL, "hiv_ns0.ctl")).read().replace("NSsites = 0", "NSsites = 0 1 2 7 8").replace("../data/", os.path.join(helpers.GOLDEN, "data") + "/")
    (tmp_path / "list.ctl").write_text(ctl)
    a = hostlib.Analysis(str(tmp_path / "list.ctl"), "codeml")
    assert a.ctl_option("NSsites") == "0 1 2 7 8" and a.np == helpers.load_golden("hiv_m0")["x"].__len__()
    for ns, gname in ((1, "hiv_m1a"), (2, "hiv_m2a"), (8, "hiv_m8")):
        g = helpers.load_golden(gname)
        b = hostlib.Analysis(str(tmp_path / "list.ctl"), "codeml", overrides="NSsites = %d; ncatG = 10" % ns)
        assert b.np == len(g["x"]) and b.ctl_option("NSsites") == str(ns)
        assert abs(oracle.evaluate(b.problem(np.array(g["x"])))["lnL"] - g["lnL"]) <= 2e-6
    with pytest.raises(RuntimeError, match="without '='"):
        hostlib.Analysis(str(tmp_path / "list.ctl"), "codeml", overrides="NSsites 2")


def test_c_host_clade_labels_spread_down_the_tree():
    """'$k' after a clade labels all its branches that carry no '#' of their own, inner clade labels winning over outer ones
    (DownTreeCladeLabel treesub.c:2960-2976): ((1, 2 #2) $1, ((3, 4) $2, 5), (6, 7)) is three branch types."""
    a = hostlib.Analysis(os.path.join(CTL, "lysos_clade_label.ctl"), "codeml")
    g = helpers.load_golden("lysos_clade_label")
    pb = a.problem(np.array(g["x"]))
    lab = {tuple(sorted(int(c) for c in pb.tree.sons[i])): int(pb.tree.label[i]) for i in range(pb.tree.n_tips, pb.tree.n_nodes)}
    assert [int(v) for v in pb.tree.label[:7]] == [1, 2, 2, 2, 0, 0, 0]
    assert lab[(0, 1)] == 1 and lab[(2, 3)] == 2 and lab[(5, 6)] == 0
    assert a.np == 11 + 1 + 3


def test_c_host_rejects_what_it_does_not_support(tmp_path):
    ctl = tmp_path / "x.ctl"
    ctl.write_text("seqfile = %s\ntreefile = %s\nseqtype = 1\nmodel = 1\nNSsites = 2\n" %
                   (os.path.join(helpers.GOLDEN, "data", "HIVenvSweden.txt"), os.path.join(helpers.GOLDEN, "data", "HIVenvSweden.trees")))
    with pytest.raises(RuntimeError, match="not supported"):
        hostlib.Analysis(str(ctl), "codeml")


def test_numerics_against_scipy():
    """discrete gamma / beta classes of the C host vs scipy (independent implementation)."""
    from paml_amd import models
    a = hostlib.Analysis(os.path.join(CTL, "hiv_ns7.ctl"), "codeml")
    g = helpers.load_golden("hiv_m7")
    pb = a.problem(np.array(g["x"]))
    f, w = helpers.nssites_classes(7, g["x"][g["ntime"] + 1:], 10)
    ref = helpers.problem_from_golden(g)
    for ic in (0, 3, 9):                      # same class-specific P(t) as the numpy/scipy model layer
        P1 = oracle.pmat_branch(pb, 0, ic, 3)
        P2 = oracle.pmat_branch(ref, 0, ic, 3)
        assert np.max(np.abs(P1 - P2)) < 1e-12
    a2 = hostlib.Analysis(os.path.join(CTL, "stewart_lg_g4.ctl"), "codeml")
    pb2 = a2.problem(np.array(helpers.load_golden("stewart_lg_g4")["x"]))
    fk, rk = models.discrete_gamma(1.064411, 4)
    assert np.allclose(pb2.rate, rk, rtol=1e-12) and np.allclose(pb2.freqK, fk)


@pytest.mark.gpu
@pytest.mark.parametrize("gname,prog,ctl", CASES)
def test_c_host_on_gpu(gname, prog, ctl):
    g = helpers.load_golden(gname)
    a = hostlib.Analysis(os.path.join(CTL, ctl), prog)
    lnl, lnf = a.eval_gpu(_x(g, a))
    assert abs(lnl - g["lnL"]) <= 2e-6
    assert np.max(np.abs(lnf - np.array(g["logf"]))) < 5e-8


@pytest.mark.gpu
def test_driver_binary_end_to_end(tmp_path):
    """pamlh_lnl <program> <ctl> x...: prints lnL like the reference and writes an `lnf` file in its layout."""
    g = helpers.load_golden("hiv_m2a")
    out = subprocess.run([hostlib.DRIVER_PATH, "codeml", os.path.join(CTL, "hiv_ns2.ctl")] + ["%.6f" % v for v in g["x"]],
                         cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    lnl = float(out.stdout.split("lnL  =")[1].split()[0])
    assert abs(lnl - g["lnL"]) <= 2e-6
    rows = [ln.split() for ln in open(tmp_path / "lnf") if len(ln.split()) > 5]
    assert len(rows) == g["n_patt"]
    assert np.max(np.abs(np.array([float(r[2]) for r in rows]) - np.array(g["logf"]))) < 5e-8
    assert [int(float(r[1])) for r in rows] == [int(c) for c in g["counts"]]


@pytest.mark.gpu
def test_c_host_optimiser_with_frequency_parameters_and_site_classes():
    """x holds kappa, then the npi codon-frequency parameters, then the site-class proportions: the optimiser's log-ratio transform of
    (p0, p1) must sit behind the frequency parameters (pamlh_simplex_groups).  FMutSel0 + M2a on the HIV data: the search ends at the
    lnL the reference's optimiser printed, with the mutation-bias ratios untouched by the transform (they are not a simplex)."""
    g = helpers.load_golden("hiv_fmutsel0_m2a")
    a = hostlib.Analysis(os.path.join(CTL, "hiv_fmutsel0_m2a.ctl"), "codeml")
    r = a.optimize(a.default_x())
    assert r["converged"]
    assert abs(r["lnL"] - g["mle_lnL"]) <= 2e-5, (r["lnL"], g["mle_lnL"])
    lo, hi = a.bounds()
    assert ((r["x"] >= lo) & (r["x"] <= hi)).all()
    xg = np.array(g["x"])
    assert np.allclose(r["x"][a.ntime:], xg[a.ntime:], rtol=2e-2, atol=2e-3), (r["x"][a.ntime:], xg[a.ntime:])


@pytest.mark.gpu
@pytest.mark.parametrize("gname,prog,ctl", [CASES[0], CASES[1], CASES[2], CASES[4]])
def test_c_host_optimiser_finds_the_reference_mle(gname, prog, ctl):
    """pamlh_optimize (BFGS, gradients and line searches as batches on the GPU) started from the control file's initial
    values reaches the lnL the reference's own optimiser reports for the data set (brown HKY85 -2665.422858,
    stewart LG+G4 -1038.351723, HIV M0 -1137.688190, HIV M2a -1106.445004) — the golden x are those MLEs printed with 6 decimals."""
    g = helpers.load_golden(gname)
    a = hostlib.Analysis(os.path.join(CTL, ctl), prog)
    r = a.optimize(a.default_x())
    assert r["converged"]
    assert abs(r["lnL"] - g["lnL"]) <= 5e-6, (r["lnL"], g["lnL"])
    lo, hi = a.bounds()
    assert ((r["x"] >= lo) & (r["x"] <= hi)).all()
    gx = np.array(g["x"])
    free = gx > 1e-5                                   # a zero-length branch sits on the boundary in both programs
    assert np.max(np.abs(r["x"][free] - gx[free]) / (np.abs(gx[free]) + 0.01)) < 2e-2


@pytest.mark.gpu
@pytest.mark.parametrize("gname,prog,ctl", [CASES[2], CASES[0], CASES[4]])
def test_c_host_method1_minbranches(gname, prog, ctl):
    """method = 1 (minB / minbranches, treesub.c:7826-8117) on the engine's branch-local path: from the control file's
    initial values the reference's MLE lnL is reached (HIV M0 -1137.688190, brown HKY85 -2665.422858, HIV M2a -1106.445004),
    and — the point of the method — with few full-tree evaluations: the branch steps recompute only the nodes on the path
    between consecutive branches.  BASELINE.md section 3: the reference spends 122 lfun on HIV M0 with method = 1."""
    g = helpers.load_golden(gname)
    a = hostlib.Analysis(os.path.join(CTL, ctl), prog)
    r = a.optimize_minb(a.default_x())
    assert r["converged"], r
    assert abs(r["lnL"] - g["lnL"]) <= 5e-6, (r["lnL"], g["lnL"])
    n_int = a.n_nodes - a.n_tips
    # work in full-tree equivalents: the parameter steps' evaluations + the partials recomputed by the branch steps
    # (a full tree = n_int internal-node partials) + one matrix product per trial length evaluated (1 / n_int of a tree each)
    equiv = r["n_eval"] + r["nodes_recomputed"] / n_int
    print("%s: %d full evaluations + %d branch calls recomputing %d node partials = %.1f full-tree equivalents"
          % (gname, r["n_eval"], r["branch_calls"], r["nodes_recomputed"], equiv))
    assert r["nodes_recomputed"] < 3 * r["branch_calls"] + n_int
    if gname == "hiv_m0":
        assert equiv <= 200, equiv


@pytest.mark.gpu
@pytest.mark.parametrize("gname,ctl", [("lyso_bsa", "lyso_bsa.ctl"), ("lyso_bsa_null", "lyso_bsa_null.ctl"), ("ecp_cmc", "ecp_cmc.ctl")])
def test_c_host_optimiser_on_branch_site_and_clade_models(gname, ctl):
    """Branch-site model A (alternative and null of the branch-site test, lysozyme data of examples/lysozyme) and clade model C
    (examples/CladeModelCD): four / three site classes x two branch types, eigen systems picked per (class, label) and one
    time scale per branch type (Qfactor_NS_branch).  These surfaces have local optima (examples/lysozyme/README.txt: "run the
    program multiple times, using different initial values"), so the search is started near the reference's optimum — its
    estimates with the substitution parameters moved by 10 % — and must come back to the reference's maximum; from the host's
    own initial values it must converge to a stationary point that is not better than that."""
    g = helpers.load_golden(gname)
    a = hostlib.Analysis(os.path.join(CTL, ctl), "codeml")
    x0 = np.array(g["x"])
    x0[a.ntime:] *= 1.1
    lo, hi = a.bounds()
    r1 = a.optimize(np.clip(x0, lo, hi))
    assert r1["converged"] and abs(r1["lnL"] - g["mle_lnL"]) < 5e-5, (r1["lnL"], g["mle_lnL"])
    r = a.optimize(a.default_x())
    assert r["converged"] and g["mle_lnL"] - 2.0 < r["lnL"] < g["mle_lnL"] + 5e-5, (r["lnL"], g["mle_lnL"])


@pytest.mark.gpu
@pytest.mark.parametrize("gname,np_,tol", [("lysos_free", 23, 1e-3), ("lysos_branch_fix", 14, 2e-5)])
def test_c_host_optimiser_on_the_free_ratio_and_fixed_omega_branch_models(gname, np_, tol):
    """model = 1 (Yang 1998): one omega per branch of the small lysozyme tree — 11 branch lengths, kappa, 11 omegas, 11 eigen systems
    selected through the engine's branch labels.  From the control file's initial values the search reaches the reference's maximum
    (-896.412472; two of its omegas sit on the bound 999 — no synonymous change on the branch — where the surface is flat).
    model = 2 with fix_omega = 1: three branch types, the last omega fixed at 1 (the likelihood-ratio test of omega = 1 on a branch,
    examples/lysozyme/README.txt "table 1E&J"): -903.482633."""
    g = helpers.load_golden(gname)
    a = hostlib.Analysis(os.path.join(CTL, gname + ".ctl"), "codeml")
    assert (a.np, a.ntime) == (np_, 11)
    r = a.optimize(a.default_x())
    assert r["converged"] and r["lnL"] > g["mle_lnL"] - tol and r["lnL"] < g["mle_lnL"] + 0.05, (r["lnL"], g["mle_lnL"])


@pytest.mark.gpu
def test_c_host_branch_site_neb_and_beb_match_the_reference_rst():
    """Branch-site model A on the lysozyme data: the per-site posteriors of the four site classes (0, 1, 2a, 2b) the reference
    writes to `rst` — naive empirical Bayes at the estimates, and Bayes empirical Bayes over the 10^4-point grid of
    (p0, p1, w0, w2) with 121 (background, foreground) omega pairs evaluated in one launch (lfunNSsites_ACD) — printed with
    5 decimals; and the mlc table "Positive sites for foreground lineages" (sites 14 21 23 37 41 50 62 87 126)."""
    g = helpers.load_golden("lyso_bsa")
    a = hostlib.Analysis(os.path.join(CTL, "lyso_bsa.ctl"), "codeml")
    x = np.array(g["x"])
    post, _ = a.neb(x)
    assert np.max(np.abs(post.T - np.array(g["neb_post"]))) < 2e-5
    beb = a.beb_acd(x)
    assert np.max(np.abs(beb.T - np.array(g["beb_post"]))) < 2e-5
    pos = beb[2] + beb[3]
    assert [int(i) + 1 for i in np.nonzero(pos > 0.5)[0]] == [14, 21, 23, 37, 41, 50, 62, 87, 126]
    assert abs(pos[13] - 0.859) < 6e-4 and abs(pos[86] - 0.869) < 6e-4


@pytest.mark.gpu
@pytest.mark.parametrize("gname,ctl", [("ecp_cmc", "ecp_cmc.ctl"), ("ecp_cmd", "ecp_cmd.ctl")])
def test_c_host_clade_model_neb_and_beb_match_the_reference_rst(gname, ctl):
    """Clade models C and D on examples/CladeModelCD (two branch types): NEB and BEB posteriors of the three site classes at
    all 161 sites as the reference writes them to `rst`.  BEB: 111 / 120 (type-0 omega, type-1 omega) pairs in one evaluation,
    then the 10^5- (C) or 10^6-point (D) grid over (p0, p1, w0, [w1,] w2, w3)."""
    g = helpers.load_golden(gname)
    a = hostlib.Analysis(os.path.join(CTL, ctl), "codeml")
    x = np.array(g["x"])
    post, _ = a.neb(x)
    assert np.max(np.abs(post.T - np.array(g["neb_post"]))) < 2e-5
    beb = a.beb_acd(x)
    assert beb.shape == (3, g["ls"])
    assert np.max(np.abs(beb.T - np.array(g["beb_post"]))) < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("gname", ["brown_hky85_nhomo1", "brown_hky85_nhomo2", "brown_t92_nhomo3_g4", "brown_f84_nhomo4", "brown_hky85_nhomo5"])
def test_c_host_optimiser_on_nonhomogeneous_models(gname):
    """nhomo = 2 (seven kappas), 3 with T92 + gamma (seven GC contents, one kappa, alpha) and 4 with F84 (eight frequency sets):
    one eigen system per branch, selected through the engine's branch labels; from the control file's initial values the
    optimiser reaches the reference's maximum (or a slightly better point: these surfaces are flat in the frequency sets)."""
    g = helpers.load_golden(gname)
    a = hostlib.Analysis(os.path.join(CTL, gname + ".ctl"), "baseml")
    r = a.optimize(a.default_x(), max_iter=2000)
    assert r["lnL"] - g["mle_lnL"] > -2e-3, (r["lnL"], g["mle_lnL"], r["converged"])
    assert r["lnL"] - g["mle_lnL"] < 0.5


@pytest.mark.gpu
@pytest.mark.parametrize("gname,prog,ctl", [("horai_mg0_g5", "baseml", "horai_mg0_g5.ctl"), ("horai_mg4", "baseml", "horai_mg4.ctl"),
                                            ("horai_mg0_malpha", "baseml", "horai_mg0_malpha.ctl"), ("horai_mg4_malpha", "baseml", "horai_mg4_malpha.ctl"),
                                            ("lysin_mg3", "codeml", "lysin_mg3.ctl")])
def test_c_host_optimiser_with_several_genes(gname, prog, ctl):
    """Option G data (examples/horai.nuc: four genes by site marks; lysinYangSwanson2002.nuc: two site partitions): gene rates
    (rgene), per-gene frequencies and per-gene kappa / omega go through the engine's gene tables (pattern offsets, one pi and
    one eigen system per gene, gene rates in the batched evaluations); from the control file's initial values the optimiser
    reaches the reference's maximum (the reference's own runs on horai.nuc stop between -13260.093893 and -13260.093876 depending on
    their random starting values, so: not lower than its value, and not higher by more than that spread allows)."""
    g = helpers.load_golden(gname)
    a = hostlib.Analysis(os.path.join(CTL, ctl), prog)
    r = a.optimize(a.default_x())
    assert r["converged"] and -2e-5 < r["lnL"] - g["mle_lnL"] < 2e-4, (r["lnL"], g["mle_lnL"])


@pytest.mark.gpu
@pytest.mark.parametrize("prog,ctl", [("baseml", "brown_hky85.ctl"), ("codeml", "hiv_ns0.ctl"), ("codeml", "mhc_m0.ctl"), ("codeml", "stewart_lg_g4.ctl"),
                                      ("baseml", "horai_mg4.ctl"), ("codeml", "lysin_mg2.ctl"), ("codeml", "lyso_bsa.ctl")])
def test_c_host_device_pattern_compression_gives_the_same_data(prog, ctl, monkeypatch):
    """PAMLH_GPU_COMPRESS=1 routes PatternWeight through paml_amd_compress_patterns: patterns, their order, counts, the
    site -> pattern map and everything derived (frequencies, lnL) must equal the host sort's — nucleotide, codon (3 characters
    per site), amino-acid data with ambiguity characters, 192 taxa, and genes by marks."""
    a = hostlib.Analysis(os.path.join(CTL, ctl), prog)
    pa = a.problem(a.default_x())
    monkeypatch.setenv("PAMLH_GPU_COMPRESS", "1")
    b = hostlib.Analysis(os.path.join(CTL, ctl), prog)
    monkeypatch.delenv("PAMLH_GPU_COMPRESS")
    pb = b.problem(b.default_x())
    assert (a.n_patt, a.ls) == (b.n_patt, b.ls)
    assert np.array_equal(pa.z, pb.z) and np.array_equal(pa.weights, pb.weights) and np.array_equal(pa.gene_off, pb.gene_off)
    assert np.array_equal(pa.pi, pb.pi)
    la, _ = a.eval_gpu(a.default_x(), want_lnf=False)
    lb, _ = b.eval_gpu(b.default_x(), want_lnf=False)
    assert la == lb


@pytest.mark.gpu
def test_c_host_marginal_reconstruction_matches_the_reference_rst():
    """RateAncestor = 1 through the C host: most probable base and its probability at the three internal nodes of the brown.nuc
    tree for every pattern, as the reference's `rst` lists them (tests/golden/brown_hky85_anc.json)."""
    g = helpers.load_golden("brown_hky85_anc")
    a = hostlib.Analysis(os.path.join(CTL, "brown_hky85.ctl"), "baseml")
    raw = ["".join(chr(c) for c in col) for col in a.raw_patterns().T] if hasattr(a, "raw_patterns") else None
    pb = a.problem(np.array(g["x"]))
    for k, node in enumerate(g["nodes_1based"]):
        post = a.node_posterior(np.array(g["x"]), node - 1)
        assert np.allclose(post.sum(axis=1), 1)
        for h in range(a.n_patt):
            patt = "".join("TCAG"[c] for c in pb.z[:, h])
            row = g["patterns"][patt]
            i = int(np.argmax(post[h]))
            assert "TCAG"[i] == row["best"][k] and abs(post[h, i] - row["prob"][k]) < 6e-4


@pytest.mark.gpu
def test_c_host_optimiser_under_the_global_clock():
    """clock = 1 on the rooted brown tree: the optimiser iterates on (root age, age ratios), reports ages, and reaches the
    reference's -2666.330289 with its node ages and kappa; the likelihood-ratio statistic against the unrooted fit
    (-2665.422858) is the molecular-clock test's 2 x 0.907."""
    g = helpers.load_golden("brown_hky85_clock")
    a = hostlib.Analysis(os.path.join(CTL, "brown_hky85_clock.ctl"), "baseml")
    assert (a.np, a.ntime) == (5, 4)
    r = a.optimize(a.default_x())
    assert r["converged"] and abs(r["lnL"] - g["mle_lnL"]) < 5e-6, (r["lnL"], g["mle_lnL"])
    assert np.max(np.abs(r["x"] - np.array(g["x"])) / np.array(g["x"])) < 5e-3
    assert np.all(np.diff(r["x"][:4]) < 0)               # ages decrease from the root down this ladder tree


@pytest.mark.gpu
def test_c_host_optimiser_under_local_clocks():
    """clock = 2: '#1' on the (1,2) clade's stem and '#2' on tip 4 of the rooted brown tree — four node ages, two branch-class rates
    and kappa; from the host's initial values the optimiser reaches the reference's -2665.863666 and its estimates (the second rate
    is poorly determined: a single tip branch)."""
    g = helpers.load_golden("brown_hky85_clock2")
    a = hostlib.Analysis(os.path.join(CTL, "brown_hky85_clock2.ctl"), "baseml")
    assert (a.np, a.ntime) == (7, 6)
    r = a.optimize(a.default_x())
    assert r["converged"] and abs(r["lnL"] - g["mle_lnL"]) < 2e-5, (r["lnL"], g["mle_lnL"])
    gx = np.array(g["x"])
    assert np.max(np.abs(r["x"][[0, 1, 2, 3, 6]] - gx[[0, 1, 2, 3, 6]]) / gx[[0, 1, 2, 3, 6]]) < 2e-2


def _ymd_analysis(tmp_path):
    data = os.path.join(helpers.GOLDEN, "data")
    (tmp_path / "HIV2ge.ymd.txt").write_text(helpers.ymd_names(open(os.path.join(data, "HIV2ge.txt")).read()))
    (tmp_path / "HIV2ge.ymd.tree").write_text(helpers.ymd_names(open(os.path.join(data, "HIV2ge.tree1")).read()))
    ctl = open(os.path.join(CTL, "hiv2_tipdate.ctl")).read().replace("../data/HIV2ge.txt", str(tmp_path / "HIV2ge.ymd.txt"))
    ctl = ctl.replace("../data/HIV2ge.tree1", str(tmp_path / "HIV2ge.ymd.tree")).replace("TipDate = 1 100", "TipDate = 1 36500")
    (tmp_path / "ymd.ctl").write_text(ctl)
    return hostlib.Analysis(str(tmp_path / "ymd.ctl"), "baseml")


def test_c_host_dated_tips_as_calendar_dates_on_cpu(tmp_path):
    """TipDate with yyyy-mm-dd at the end of the names (GetTipDate treesub.c:3573-3582: days since 1970-01-01, here by calendar
    arithmetic — the reference's mktime() under TZ=UTC): the HIV-2 data with every sampling year turned into a date (helpers.ymd_names),
    time unit 36 500 days.  One evaluation at the reference's estimates gives its lnL (-12352.454195)."""
    g = helpers.load_golden("hiv2_tipdate_ymd")
    a = _ymd_analysis(tmp_path)
    assert (a.np, a.ntime, a.n_patt) == (35, 33, g["n_patt"])
    pb = a.problem(np.array(g["x"]))
    assert np.array_equal(pb.weights, np.array(g["counts"]))
    r = oracle.evaluate(pb)
    assert abs(r["lnL"] - g["lnL"]) <= 2e-6          # (the reference's lnf file of this run holds nan: see the golden's note)
    bad = tmp_path / "bad.txt"
    bad.write_text(open(tmp_path / "HIV2ge.ymd.txt").read().replace("_1995-08-04", "_1995-13-04"))
    (tmp_path / "bad.tree").write_text(open(tmp_path / "HIV2ge.ymd.tree").read().replace("_1995-08-04", "_1995-13-04"))
    (tmp_path / "bad.ctl").write_text(open(tmp_path / "ymd.ctl").read().replace("HIV2ge.ymd.txt", "bad.txt").replace("HIV2ge.ymd.tree", "bad.tree"))
    with pytest.raises(RuntimeError, match="date format"):
        hostlib.Analysis(str(tmp_path / "bad.ctl"), "baseml")


@pytest.mark.gpu
def test_c_host_optimiser_with_dated_tips():
    """TipDate (examples/TipDate.HIV2, Stadler & Yang 2012): 33 sequences sampled 1982-1995, global clock, HKY85 + G5.  The
    optimiser iterates on (root age, position of every other node between the oldest tip below it and its father) and reaches
    the reference's -12352.105674, its mutation rate (0.2329 per site per 100 years) and root age."""
    g = helpers.load_golden("hiv2_tipdate")
    a = hostlib.Analysis(os.path.join(CTL, "hiv2_tipdate.ctl"), "baseml")
    assert (a.np, a.ntime) == (35, 33)
    r = a.optimize(a.default_x(), max_iter=2000)
    assert r["converged"] and abs(r["lnL"] - g["mle_lnL"]) < 5e-4, (r["lnL"], g["mle_lnL"])
    gx = np.array(g["x"])
    assert abs(r["x"][32] - gx[32]) / gx[32] < 2e-2 and abs(r["x"][0] - gx[0]) / gx[0] < 2e-2


@pytest.mark.gpu
def test_c_host_site_rates_match_the_reference_rates_file():
    """RateAncestor = 1 under HKY85 + G4: posterior mean rate and most probable rate class of every site of brown.nuc as the
    reference writes them to `rates` (3 decimals) — class posteriors from the device's fhK (lfunRates)."""
    g = helpers.load_golden("brown_hky85_g4_rates")
    a = hostlib.Analysis(os.path.join(CTL, "brown_hky85_g4.ctl"), "baseml")
    x = np.array(g["x"])
    lnl, _ = a.eval_gpu(x, want_lnf=False)
    assert abs(lnl - g["lnL"]) < 5e-6
    post, mean = a.neb(x)
    assert post.shape == (4, 895)
    assert np.max(np.abs(mean - np.array(g["rate_mean"]))) < 6e-4
    assert np.array_equal(np.argmax(post, axis=0) + 1, np.array(g["rate_class"]))


@pytest.mark.gpu
def test_c_host_joint_reconstruction_matches_the_reference_rst():
    """The best joint reconstruction of the three internal nodes of the brown.nuc tree and its probability for every pattern, as
    in "(2) Joint reconstruction of ancestral sequences" of the reference's rst (Pupko's algorithm; probabilities to 3 decimals;
    patterns whose two best reconstructions are within the printed precision of each other may swap)."""
    g = helpers.load_golden("brown_hky85_joint")
    a = hostlib.Analysis(os.path.join(CTL, "brown_hky85.ctl"), "baseml")
    x = np.array(g["x"])
    st, pr = a.joint_reconstruction(x)
    pb = a.problem(x)
    swaps = 0
    for h in range(a.n_patt):
        row = g["patterns"]["".join("TCAG"[c] for c in pb.z[:, h])]
        mine = "".join("TCAG"[c] for c in st[h])
        assert abs(pr[h] - row["prob"]) < 6e-4, (h, pr[h], row)
        swaps += mine != row["best"]
        assert mine == row["best"] or row["prob"] < 0.51
    assert swaps <= 1


@pytest.mark.gpu
def test_driver_separate_gene_analyses(tmp_path):
    """pamlh_lnl with Mgene = 1 and --optimize: four independent searches on horai.nuc end at the reference's per-gene maxima."""
    ctl = tmp_path / "horai_mg1.ctl"
    ctl.write_text(open(os.path.join(CTL, "horai_mg0.ctl")).read().replace("../data/", os.path.join(helpers.GOLDEN, "data") + "/").replace("Mgene = 0", "Mgene = 1"))
    out = subprocess.run([hostlib.DRIVER_PATH, "baseml", str(ctl), "--optimize"], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    got = [float(v) for v in re.findall(r"npatt:\s*\d+\s+lnL = (-[0-9.]+)", out.stdout)]
    assert len(got) == 4
    for v, (_, _, lnl, _) in zip(got, HORAI_SEPARATE):
        assert abs(v - lnl) < 5e-5, (v, lnl)
    assert abs(float(re.search(r"Sum of lnL over the 4 genes = (-[0-9.]+)", out.stdout).group(1)) - sum(r[2] for r in HORAI_SEPARATE)) < 2e-4


@pytest.mark.gpu
def test_c_host_plfun_seam():
    """pamlh_plfun has com.plfun's convention: x in, MINUS lnL out; a vector the model rejects gives +1e300, not an exit."""
    g = helpers.load_golden("hiv_m2a")
    a = hostlib.Analysis(os.path.join(CTL, "hiv_ns2.ctl"), "codeml")
    x = np.array(g["x"])
    assert abs(a.plfun(x) + g["lnL"]) <= 2e-6
    x[a.ntime + 1] = 0.9          # p0 + p1 > 1
    assert a.plfun(x) == 1e300
    assert a.plfun(x[:-1]) == 1e300


@pytest.mark.gpu
def test_driver_all_trees_and_their_comparison(tmp_path):
    """`pamlh_lnl codeml <ctl> --all-trees`: both trees of stewart.trees (header corrected to two) maximised from the control file's
    initial values — the reference's -1038.351723 and -1028.130…  — and the comparison table of rell() computed from the per-pattern
    values the engine returned: li, Dli, SE, pKH as the reference printed them (golden stewart_two_trees)."""
    g = helpers.load_golden("stewart_two_trees")
    data = os.path.join(helpers.GOLDEN, "data")
    (tmp_path / "two.trees").write_text(open(os.path.join(data, "stewart.trees")).read().replace("6  1", "6  2", 1))
    ctl = open(os.path.join(CTL, "stewart_lg_g4.ctl")).read().replace("../data/stewart.aa", os.path.join(data, "stewart.aa"))
    ctl = ctl.replace("../data/lg.dat", os.path.join(data, "lg.dat")).replace("../data/stewart.trees", str(tmp_path / "two.trees"))
    (tmp_path / "two.ctl").write_text(ctl)
    out = subprocess.run([hostlib.DRIVER_PATH, "codeml", str(tmp_path / "two.ctl"), "--all-trees"], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    lnls = [float(v) for v in re.findall(r"TREE #\s*\d+:\s+lnL\([^)]*\):\s*(-?[0-9.]+)", out.stdout)]
    assert len(lnls) == 2 and all(abs(a - b) < 5e-5 for a, b in zip(lnls, g["mle_lnL"])), (lnls, g["mle_lnL"])
    rows = re.findall(r"^\s*(\d+)(\*?)\s+(-?[0-9.]+)\s+(-?[0-9.]+)\s+(-?[0-9.]+)\s+(-?[0-9.]+)\s+(-?[0-9.]+)\s+(-?[0-9.]+)\s*$", out.stdout, re.M)
    assert len(rows) == 2 and rows[1][1] == "*"
    for r, ref in zip(rows, g["table"]):
        assert abs(float(r[2]) - ref["li"]) < 2e-3 and abs(float(r[3]) - ref["dli"]) < 2e-3 and abs(float(r[4]) - ref["se"]) < 2e-3
        assert abs(float(r[5]) - ref["pKH"]) < 2e-3 and abs(float(r[7]) - ref["pRELL"]) < 0.012


@pytest.mark.gpu
def test_driver_gpus_flag_one_rank(tmp_path):
    """`pamlh_lnl --gpus 1`: the multi-GPU driver path (rank set-up, pattern shard, RCCL communicator joined inside the engine,
    all-reduced lnL) with a single rank — the only size a one-GPU box can run — gives the single-process lnL; --optimize on it
    reaches the reference's MLE."""
    g = helpers.load_golden("hiv_m0")
    exe = hostlib.DRIVER_PATH
    x = " ".join("%.6f" % v for v in g["x"]).split()
    out = subprocess.run([exe, "codeml", os.path.join(CTL, "hiv_ns0.ctl"), "--gpus", "1"] + x, cwd=tmp_path, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, timeout=300)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    txt = out.stdout.decode()
    assert "sharded over 1 GPUs" in txt
    assert abs(float(txt.split("lnL  =")[1].split()[0]) - g["lnL"]) <= 2e-6
    out = subprocess.run([exe, "codeml", os.path.join(CTL, "hiv_ns0.ctl"), "--gpus", "1", "--optimize"], cwd=tmp_path, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    assert abs(float(out.stdout.decode().split("lnL  =")[1].split()[0]) - g["lnL"]) <= 5e-6


def _mcmctree_ages(a, sample):
    """node ages [n_nodes] for one mcmc.txt sample: tips 0, internal nodes in the reference's numbering (root first)."""
    age = np.zeros(a.n_nodes)
    age[a.n_tips:] = sample["age"]
    return age


def test_mcmctree_exact_likelihood_seam_on_cpu():
    """The mcmctree consumer's arithmetic (lnpD_locus mcmctree.c:1130-1166: branch = (age of father - age) x mu, then one plfun),
    restated by the host's clock = 1 parameterisation + the oracle, against the lnL column of the reference's own MCMC samples
    (usedata = 1, JC69; 3 printed decimals, ages printed with 7)."""
    g = helpers.load_golden("mcmctree_jc_clock1")
    a = hostlib.Analysis(os.path.join(CTL, "mcmctree_locus1.ctl"), "baseml")
    assert a.n_tips == 7 and a.np == a.ntime == 6          # clock = 1, JC69: x = the six internal node ages
    for smp in g["samples"]:
        pb = a.problem(np.array(smp["age"]) * smp["mu"])   # ages x rate: the branch lengths lnpD_locus forms
        assert abs(oracle.evaluate(pb, want_lnf=False)["lnL"] - smp["lnL"]) < 2e-3, smp


@pytest.mark.gpu
def test_mcmctree_exact_likelihood_seam_on_gpu(tmp_path):
    """pamlh_lnpd_locus — ages and rate in, lnL out, only branch lengths re-sent after the first call — reproduces the lnL
    column of the reference's mcmc.txt; so does the C example a maintainer would start from (mcmctree_seam.c), and per-branch
    rates (clock = 2 / 3) equal to the locus rate give the same value."""
    g = helpers.load_golden("mcmctree_jc_clock1")
    a = hostlib.Analysis(os.path.join(CTL, "mcmctree_locus1.ctl"), "baseml")
    a.set_x(a.default_x())
    got = [a.lnpd_locus(_mcmctree_ages(a, s), rgene=s["mu"], model_changed=(i == 0)) for i, s in enumerate(g["samples"])]
    want = [s["lnL"] for s in g["samples"]]
    assert np.max(np.abs(np.array(got) - np.array(want))) < 2e-3, (got[:3], want[:3])
    s0 = g["samples"][0]
    assert abs(a.lnpd_locus(_mcmctree_ages(a, s0), rate=np.full(a.n_nodes, s0["mu"]), model_changed=False) - got[0]) < 1e-9
    with pytest.raises(RuntimeError):                       # a son older than its father: "blength < 0" (mcmctree.c:1150)
        bad = _mcmctree_ages(a, s0)
        bad[a.n_tips + 1] = bad[a.n_tips] * 1.1
        a.lnpd_locus(bad, rgene=s0["mu"])
    exe = tmp_path / "mcmctree_seam"
    inc, libdir = os.path.join(helpers.REPO, "include"), os.path.join(helpers.REPO, "paml_amd", "lib")
    subprocess.check_call(["gcc", "-O2", "-I", inc, os.path.join(helpers.REPO, "paml_amd", "host", "examples", "mcmctree_seam.c"), "-o", str(exe),
                           "-L", libdir, "-lpamlh", "-lpaml_amd", "-lm", "-Wl,-rpath," + libdir])
    states = "".join(" ".join("%.7f" % v for v in s["age"]) + " %.7f\n" % s["mu"] for s in g["samples"])
    out = subprocess.run([str(exe), os.path.join(CTL, "mcmctree_locus1.ctl")], input=states.encode(), stdout=subprocess.PIPE, check=True)
    vals = [float(v) for v in out.stdout.split()]
    assert len(vals) == len(want) and np.max(np.abs(np.array(vals) - np.array(want))) < 2e-3


@pytest.mark.gpu
def test_c_host_batch_matches_single_evaluations():
    g = helpers.load_golden("hiv_m2a")
    a = hostlib.Analysis(os.path.join(CTL, "hiv_ns2.ctl"), "codeml")
    x = np.array(g["x"])
    xs = np.tile(x, (6, 1))
    xs[1, 3] *= 1.01            # a branch length
    xs[2, a.ntime] *= 1.01      # kappa: a different set of eigen systems
    xs[3, a.ntime + 1] = 0.9    # p0 + p1 > 1: rejected
    xs[4, a.ntime + 4] *= 1.1   # omega_2
    xs[5, 7] *= 0.5
    got = a.eval_batch_gpu(xs)
    assert abs(got[0] - g["lnL"]) <= 2e-6
    assert got[3] == -1e300
    for b in (1, 2, 4, 5):
        one, _ = a.eval_gpu(xs[b], want_lnf=False)
        assert abs(got[b] - one) <= 1e-9 * abs(one)


@pytest.mark.gpu
def test_c_host_standard_errors_match_the_reference():
    """getSE = 1 on brown.nuc / HKY85: the reference binary (run in the build container) prints these SEs for the seven
    branch lengths and kappa; it gets them from HessianSKT2004's outer product of per-pattern scores (method 0 here, the
    2 np perturbed evaluations with their per-pattern log f_h being one batch).  Method 1 (second differences of lnL) is
    the observed information: positive definite at the maximum and within a few per cent of method 0."""
    ref_se = np.array([0.010924, 0.006479, 0.008301, 0.009329, 0.010454, 0.013492, 0.016238, 1.301409])
    a = hostlib.Analysis(os.path.join(CTL, "brown_hky85.ctl"), "baseml")
    r = a.optimize(a.default_x())
    se0, I0 = a.standard_errors(r["x"], method=0)
    assert np.max(np.abs(se0 - ref_se) / ref_se) < 2e-4, se0
    se1, H = a.standard_errors(r["x"], method=1)
    assert np.allclose(H, H.T) and (np.linalg.eigvalsh(H) > 0).all()
    assert np.max(np.abs(se1 - se0) / se0) < 0.06


def test_c_host_reads_interleaved_phylip(tmp_path):
    """Option I (ReadSeq treesub.c:487): the same alignment written in interleaved blocks gives the same patterns."""
    src = os.path.join(helpers.GOLDEN, "data", "brown.nuc")
    toks = open(src).read().split()
    ns, ls = int(toks[0]), int(toks[1])
    names, seqs, k = [], [], 2
    for _ in range(ns):
        names.append(toks[k])
        k += 1
        s = ""
        while len(s) < ls:
            s += toks[k]
            k += 1
        seqs.append(s)
    with open(tmp_path / "brown_i.nuc", "w") as f:
        f.write(" %d %d  I\n" % (ns, ls))
        for b in range(0, ls, 60):
            for j in range(ns):
                f.write(("%-12s  " % names[j] if b == 0 else "") + " ".join(seqs[j][b + c:b + c + 10] for c in range(0, min(60, ls - b), 10)) + "\n")
            f.write("\n")
    ctl = open(os.path.join(CTL, "brown_hky85.ctl")).read()
    ctl = ctl.replace("../data/brown.nuc", str(tmp_path / "brown_i.nuc")).replace("../data/brown.trees", os.path.join(helpers.GOLDEN, "data", "brown.trees"))
    (tmp_path / "b.ctl").write_text(ctl)
    a = hostlib.Analysis(os.path.join(CTL, "brown_hky85.ctl"), "baseml")
    b = hostlib.Analysis(str(tmp_path / "b.ctl"), "baseml")
    g = helpers.load_golden("brown_hky85")
    x = np.array(g["x"])
    pa, pb = a.problem(x), b.problem(x)
    assert np.array_equal(pa.z, pb.z) and np.array_equal(pa.weights, pb.weights) and b.ls == a.ls
    assert abs(oracle.evaluate(pb)["lnL"] - g["lnL"]) <= 2e-6


@pytest.mark.gpu
def test_c_host_neb_matches_the_reference_table():
    """NEB under M2a on the HIV data at the MLE: Pr(w > 1) and the posterior mean omega of the sites the reference lists in
    its main output ("Naive Empirical Bayes (NEB) analysis", printed with 3 decimals)."""
    ref = {9: (0.720, 2.889), 22: (0.796, 3.089), 24: (0.578, 2.517), 26: (0.905, 3.376), 28: (0.999, 3.624), 31: (0.566, 2.486),
           39: (0.640, 2.681), 51: (0.883, 3.319), 66: (0.998, 3.621), 68: (0.601, 2.578), 69: (0.830, 3.179), 76: (0.671, 2.761),
           83: (0.811, 3.128), 87: (0.985, 3.587)}
    g = helpers.load_golden("hiv_m2a")
    a = hostlib.Analysis(os.path.join(CTL, "hiv_ns2.ctl"), "codeml")
    post, mw = a.neb(np.array(g["x"]))
    assert post.shape == (3, 91) and np.allclose(post.sum(axis=0), 1)
    for site, (pr, m) in ref.items():
        assert abs(post[2, site - 1] - pr) < 1.5e-3 and abs(mw[site - 1] - m) < 2.5e-3, (site, post[2, site - 1], mw[site - 1])
    assert sorted(np.nonzero(post[2] > 0.5)[0] + 1) == sorted(ref)          # exactly the sites the reference reports


BEB_M2A = {1: (0.548, 2.293, 1.292), 9: (0.750, 2.975, 1.428), 22: (0.812, 3.142, 1.329), 24: (0.607, 2.453, 1.297),
           26: (0.904, 3.408, 1.184), 28: (0.999, 3.729, 1.024), 31: (0.604, 2.481, 1.356), 39: (0.668, 2.674, 1.364),
           40: (0.512, 2.212, 1.318), 51: (0.872, 3.277, 1.190), 66: (0.998, 3.727, 1.026), 68: (0.632, 2.547, 1.334),
           69: (0.825, 3.133, 1.247), 76: (0.686, 2.695, 1.313), 83: (0.808, 3.078, 1.263), 87: (0.987, 3.696, 1.062)}
BEB_M8 = {1: (0.796, 2.627, 1.064), 9: (0.857, 2.819, 1.038), 18: (0.590, 2.093, 1.314), 21: (0.592, 2.090, 1.169),
          22: (0.917, 2.973, 0.899), 24: (0.850, 2.771, 0.986), 26: (0.972, 3.112, 0.744), 28: (1.000, 3.183, 0.653),
          31: (0.801, 2.654, 1.086), 39: (0.843, 2.768, 1.032), 40: (0.733, 2.468, 1.146), 46: (0.660, 2.279, 1.226),
          51: (0.969, 3.100, 0.747), 59: (0.575, 2.049, 1.239), 66: (1.000, 3.183, 0.654), 68: (0.842, 2.758, 1.016),
          69: (0.949, 3.047, 0.804), 75: (0.551, 1.984, 1.237), 76: (0.889, 2.881, 0.929), 83: (0.941, 3.027, 0.823),
          84: (0.611, 2.139, 1.173), 87: (0.995, 3.173, 0.670)}


@pytest.mark.gpu
@pytest.mark.parametrize("gname,ctl,table", [("hiv_m2a", "hiv_ns2.ctl", BEB_M2A), ("hiv_m8", "hiv_ns8.ctl", BEB_M8)])
def test_c_host_beb_matches_the_reference_table(gname, ctl, table):
    """Bayes empirical Bayes under M2a and M8 on the HIV data: the reference's "BEB analysis" table (Pr(w>1), posterior mean
    +- SE of omega, 3 printed decimals; the runs were made with the unmodified binary in the build container) — the
    f(x_h|w) for the grid omegas come from the device, the 10^4-point grid sums from the C host."""
    g = helpers.load_golden(gname)
    a = hostlib.Analysis(os.path.join(CTL, ctl), "codeml")
    pr, mw, se = a.beb(np.array(g["x"]))
    for site, (p_, m_, s_) in table.items():
        got = (pr[site - 1], mw[site - 1], se[site - 1])
        assert abs(got[0] - p_) < 2e-3 and abs(got[1] - m_) < 4e-3 and abs(got[2] - s_) < 4e-3, (site, got)
    assert sorted(np.nonzero(pr > 0.5)[0] + 1) == sorted(table)


@pytest.mark.gpu
@pytest.mark.parametrize("gname,ctl", [("mhc_m2a", "mhc_ns2.ctl"), ("mhc_m8", "mhc_ns8.ctl")])
def test_c_host_neb_and_beb_on_a_tree_with_scaling_nodes(gname, ctl):
    """NEB and BEB on the 192-taxon MHC data (examples/MHC.Swanson2002MBE; ten scaling nodes): with NodeScale fx_r leaves
    log f(x_h | class) + the scale factors in fhK, and both analyses work with exp(fhK - max over classes)
    (lfunNSsites_rate codeml.c:5269-5276, get_grid_para_like_M2M8 codeml.c:6286-6294).  Compared with the class posteriors the
    reference writes to `rst` at all 270 sites (5 decimals): NEB every class, BEB the w > 1 class."""
    g = helpers.load_golden(gname)
    a = hostlib.Analysis(os.path.join(CTL, ctl), "codeml")
    x = np.array(g["x"])
    post, mw = a.neb(x)
    ref = np.array(g["neb_post"])
    assert post.shape == ref.T.shape and np.max(np.abs(post.T - ref)) < 2e-5
    pr, mwb, se = a.beb(x)
    refb = np.array(g["beb_post"])
    assert np.max(np.abs(pr - refb[:, -1])) < 2e-5
    assert (pr > 0.95).sum() == (refb[:, -1] > 0.95).sum() > 0          # the MHC peptide-binding sites
    assert np.isfinite(mwb).all() and (se >= 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("ctl,lnl,est", [("mtcdna_m0.ctl", -20486.034301, {-2: 20.74839, -1: 0.04414}),
                                          ("mtcdna_branch.ctl", -20444.099676, {-3: 21.59077, -2: 0.28638, -1: 0.03693}),
                                          # aaDist = 7: radical (class 0) / conserved (class 1) omegas, README.txt:19-21
                                          ("mtcdna_aaclass_m0.ctl", -20482.229434, {-3: 20.52018, -2: 0.02745, -1: 0.04658}),
                                          ("mtcdna_aaclass_branch.ctl", -20440.382774, {-5: 21.36004, -4: 0.15012, -3: 0.30470, -2: 0.02380, -1: 0.03885})])
def test_c_host_reaches_the_published_mtcdnaape_values(ctl, lnl, est):
    """examples/mtCDNAape/README.txt:15-17 publishes the maximised lnL and the estimates of kappa and omega for the ape
    mitochondrial data under M0 and under the two-ratio branch model (model = 2: within- / between-species branches labelled in
    the tree file; vertebrate mitochondrial code, 60 sense codons).  Control file, PHYLIP reader, '#' labels, F3x4, the
    per-label eigen systems, the batched optimiser and the 60-state kernels all sit between the files and these numbers."""
    a = hostlib.Analysis(os.path.join(CTL, ctl), "codeml")
    assert a.n == 60 and a.n_tips == 6
    r = a.optimize(a.default_x())
    assert r["converged"] and abs(r["lnL"] - lnl) < 2e-4, r["lnL"]
    for k, v in est.items():
        assert abs(r["x"][k] - v) / v < 5e-3, (k, r["x"][k], v)


@pytest.mark.gpu
@pytest.mark.parametrize("ctl,lnl,n,est", [("mtcdnapri_fromcodon.ctl", -14718.224885, 20, {-1: 9.156815}),
                                            ("mtcdnapri_fromcodon0.ctl", -14707.663779, 60, {-2: 9.246897, -1: 0.031208})])
def test_c_host_reaches_the_published_codon_based_aa_values(ctl, lnl, n, est):
    """examples/mtCDNA/AAcodon.result.txt:75-97: the maxima of the two codon-based amino-acid models on the 7-ape
    mitochondrial proteins (vertebrate mt code) — model 6 on the 20-state kernels, model 5 on the 60-state kernels with 21
    ambiguity codes (an amino acid = the set of its codons)."""
    a = hostlib.Analysis(os.path.join(CTL, ctl), "codeml")
    assert a.n == n and a.n_tips == 7
    r = a.optimize(a.default_x())
    assert r["converged"] and abs(r["lnL"] - lnl) < 3e-4, r["lnL"]
    for k, v in est.items():
        assert abs(r["x"][k] - v) / v < 1e-2, (k, r["x"][k], v)


@pytest.mark.gpu
@pytest.mark.parametrize("gname", ["mtcdnapri_aadist1", "mtcdnapri_aadist_m2"])
def test_c_host_optimiser_with_amino_acid_distances(gname):
    """aaDist = 1 / -2 on the 7-ape mitochondrial genes (60-state kernels, vertebrate mt code): kappa and the two parameters of
    omega(d) from the host's initial values to the reference's maximum."""
    g = helpers.load_golden(gname)
    a = hostlib.Analysis(os.path.join(CTL, gname + ".ctl"), "codeml")
    r = a.optimize(a.default_x())
    assert r["converged"] and abs(r["lnL"] - g["mle_lnL"]) < 2e-4, (r["lnL"], g["mle_lnL"])


@pytest.mark.gpu
@pytest.mark.parametrize("gname", ["hiv_fmutsel0", "hiv_fmutsel", "hiv_f3x4_est", "hiv_fmutsel0_est"])
def test_c_host_optimiser_with_frequency_parameters(gname):
    """FMutSel0 / FMutSel with three mutation-bias parameters, F3x4 with its nine frequency ratios estimated, FMutSel0 with the 19
    amino-acid fitnesses as well: every trial point has its own codon frequencies (root distribution, rates and mutation
    multipliers), so the batches are evaluated in groups of equal frequencies; from the host's initial values the optimiser
    reaches the reference's maximum (flat directions in the fitnesses: not lower, possibly slightly higher)."""
    g = helpers.load_golden(gname)
    a = hostlib.Analysis(os.path.join(CTL, gname + ".ctl"), "codeml")
    r = a.optimize(a.default_x(), max_iter=3000)
    assert -2e-3 < r["lnL"] - g["mle_lnL"] < 0.05, (r["lnL"], g["mle_lnL"], r["converged"])


@pytest.mark.gpu
def test_c_host_optimiser_with_69_exchangeabilities():
    """REVaa_0 on the 7-ape mitochondrial proteins: 11 branch lengths + the exchangeabilities of the amino-acid pairs one nucleotide
    change apart, from jones.dat's values to the reference's maximum (a gradient is one batch of 2 x 80 evaluations)."""
    g = helpers.load_golden("mtcdnapri_revaa0")
    a = hostlib.Analysis(os.path.join(CTL, "mtcdnapri_revaa0.ctl"), "codeml")
    assert a.np == 80
    r = a.optimize(a.default_x(), max_iter=3000)
    assert -5e-3 < r["lnL"] - g["mle_lnL"] < 0.5, (r["lnL"], g["mle_lnL"], r["converged"])


def test_c_host_rejects_option_combinations_it_does_not_cover(tmp_path):
    """Options outside what the host implements are refused with a message, not evaluated under another model."""
    data = os.path.join(helpers.GOLDEN, "data") + "/"

    def load(base, prog, **over):
        txt = open(os.path.join(CTL, base)).read().replace("../data/", data)
        for k, v in over.items():
            txt = re.sub(r"^\s*%s\s*=.*$" % k, "", txt, flags=re.M) + "\n %s = %s\n" % (k, v)
        f = tmp_path / "x.ctl"
        f.write_text(txt)
        return hostlib.Analysis(str(f), prog)

    with pytest.raises(RuntimeError, match="sampling date"):
        load("brown_hky85_clock.ctl", "baseml", TipDate="1 100")             # names without dates
    with pytest.raises(RuntimeError, match="branch rate labels"):
        load("brown_hky85_clock.ctl", "baseml", clock=2)                     # local clocks without '#' labels
    assert load("mtcdna_branch.ctl", "codeml", CodonFreq=6).np == 12 + 3      # FMutSel0 + branch model: three mutation-bias parameters more
    with pytest.raises(RuntimeError, match="Malpha"):
        load("brown_hky85_g4.ctl", "baseml", Malpha=1)                        # one gene
    # options that change the analysis are refused, never silently ignored
    for kw, msg in ((dict(runmode=2), "runmode"), (dict(runmode=-2), "runmode"), (dict(ndata=5), "ndata"), (dict(nparK=1), "nparK"), (dict(bootstrap=100), "bootstrap")):
        with pytest.raises(RuntimeError, match=msg):
            load("brown_hky85_g4.ctl", "baseml", **kw)
    with pytest.raises(RuntimeError, match="hkyREV"):
        load("hiv_ns0.ctl", "codeml", hkyREV=1)
    with pytest.raises(RuntimeError, match="nhomo"):
        load("brown_hky85.ctl", "baseml", nhomo=2, model=7)                   # a kappa per branch needs K80 / F84 / HKY85
    with pytest.raises(RuntimeError, match="grantham.dat"):
        load("hiv_ns0.ctl", "codeml", aaDist=1)                               # distance file not beside the control file
    a = load("hiv_ns0.ctl", "codeml", CodonFreq=7, estFreq=1)                 # and one that is covered: 23 + kappa + 3 + 60 + omega
    assert a.np == 23 + 1 + 3 + 60 + 1


def test_c_host_aaclasses_needs_its_class_file(tmp_path):
    """aaDist = 7 reads OmegaAA.dat from the control file's directory; a missing file, a pair listed twice and a bad class count are
    errors, a pair that cannot change in one step under the genetic code is ignored (as the reference does)."""
    data = os.path.join(helpers.GOLDEN, "data") + "/"
    ctl = tmp_path / "a.ctl"
    ctl.write_text(open(os.path.join(CTL, "mtcdna_aaclass_m0.ctl")).read().replace("../data/", data))
    with pytest.raises(RuntimeError, match="OmegaAA.dat"):
        hostlib.Analysis(str(ctl), "codeml")
    (tmp_path / "OmegaAA.dat").write_text("2\n1: RH RK HR\n0: all others\n")
    with pytest.raises(RuntimeError, match="listed twice"):
        hostlib.Analysis(str(ctl), "codeml")
    (tmp_path / "OmegaAA.dat").write_text("3\n1: RH RK AW\n2: DE\n")      # A <-> W needs two changes: ignored
    a = hostlib.Analysis(str(ctl), "codeml")
    assert a.np == 9 + 1 + 3


MTCDNAPRI = {0: (-31744.953377, "0.120080 0.058139 0.175107 0.098210 0.072451 0.060123 0.206933 0.224103 0.115862 0.114939 0.408898 7.417478 0.112776"),
             2: (-29967.856102, "0.261697 0.100524 0.242894 0.120349 0.078171 0.066639 0.286667 0.506947 0.158670 0.142655 0.829442 14.249448 0.041084")}


@pytest.mark.parametrize("cf", [0, 2])
def test_c_host_published_mtcdnapri_values_on_cpu(cf):
    """examples/mtCDNA/AAcodon.result.txt:15-17, 38-40: lnL and the 13 estimates (branch lengths in tree.branches order, kappa,
    omega) of codon model M0 on the primate mitochondrial data, Fequal and F3x4 — one evaluation at the printed estimates must
    give the printed lnL (C host + oracle; the engine does the same on the GPU below)."""
    lnl, xs = MTCDNAPRI[cf]
    a = hostlib.Analysis(os.path.join(CTL, "mtcdnapri_cf%d.ctl" % cf), "codeml")
    assert (a.n, a.n_tips, a.np, a.ntime) == (60, 7, 13, 11)
    x = np.array([float(v) for v in xs.split()])
    assert abs(oracle.evaluate(a.problem(x), want_lnf=False)["lnL"] - lnl) < 5e-5


@pytest.mark.gpu
@pytest.mark.parametrize("cf", [0, 2])
def test_c_host_published_mtcdnapri_values_on_gpu(cf):
    lnl, xs = MTCDNAPRI[cf]
    a = hostlib.Analysis(os.path.join(CTL, "mtcdnapri_cf%d.ctl" % cf), "codeml")
    x = np.array([float(v) for v in xs.split()])
    assert abs(a.eval_gpu(x, want_lnf=False)[0] - lnl) < 5e-5
    r = a.optimize(a.default_x())                          # ... and the optimiser finds them from the control file's start
    assert r["converged"] and abs(r["lnL"] - lnl) < 2e-4
    assert np.max(np.abs(r["x"] - x) / (np.abs(x) + 0.01)) < 1e-2


JTT_LNL, JTT_X = -14717.981418, "0.025579 0.009654 0.022079 0.012328 0.011335 0.010281 0.026980 0.052187 0.026525 0.019651 0.062350"


def test_c_host_published_jtt_value_on_cpu():
    """examples/mtCDNA/AAcodon.result.txt:57-62: AAML with JTT + F (seqtype = 2, model = 3, dat/jones.dat) on the primate
    proteins: the printed lnL at the printed branch lengths."""
    a = hostlib.Analysis(os.path.join(CTL, "mtcdnapri_jtt.ctl"), "codeml")
    assert (a.n, a.n_tips, a.np) == (20, 7, 11)
    x = np.array([float(v) for v in JTT_X.split()])
    assert abs(oracle.evaluate(a.problem(x), want_lnf=False)["lnL"] - JTT_LNL) < 5e-5


@pytest.mark.gpu
def test_c_host_published_jtt_value_on_gpu():
    a = hostlib.Analysis(os.path.join(CTL, "mtcdnapri_jtt.ctl"), "codeml")
    x = np.array([float(v) for v in JTT_X.split()])
    assert abs(a.eval_gpu(x, want_lnf=False)[0] - JTT_LNL) < 5e-5
    r = a.optimize(a.default_x())
    assert r["converged"] and abs(r["lnL"] - JTT_LNL) < 2e-4 and np.max(np.abs(r["x"] - x)) < 5e-5


@pytest.mark.gpu
@pytest.mark.parametrize("ns,lnl", [(1, -7490.993363), (2, -7231.154540), (7, -7502.792534), (8, -7238.014961)])
def test_c_host_reaches_the_published_mhc_site_model_values(ns, lnl):
    """examples/MHC.Swanson2002MBE/README.txt:26-30: 192 MHC alleles x 270 codons, branch lengths fixed at the tree file's
    (fix_blength = 2), kappa and the site-class parameters of M1a / M2a / M7 / M8 estimated.  192 taxa means scaling nodes
    (log-sum-exp class mixing), ambiguity codes, a tree beyond the specialised kernel's tip limit, and batched evaluations
    through all of that."""
    a = hostlib.Analysis(os.path.join(CTL, "mhc_ns%d.ctl" % ns), "codeml")
    assert a.ntime == 0 and a.n_tips == 192
    r = a.optimize(a.default_x(), max_iter=300)
    assert r["converged"] and abs(r["lnL"] - lnl) < 5e-4, (ns, r["lnL"], r["x"])


def test_host_bivariate_normal_and_autod_gamma_numerics():
    """The numerics behind AutodGamma, restated from the published approximations the reference uses: L(h, k, r) (Genz 2004) against
    scipy's bivariate normal over both branches of the algorithm (|r| < 0.925 and above), the AS 70 quantile against scipy to its
    stated accuracy, and the transition matrix: rows sum to 1, symmetric, uniform stationary distribution, identity-like for rho -> 1."""
    import ctypes as C
    from scipy.stats import multivariate_normal, norm
    L = hostlib.lib()
    L.pamlh_lbinormal.restype = C.c_double
    L.pamlh_lbinormal.argtypes = [C.c_double] * 3
    L.pamlh_quantile_normal.restype = C.c_double
    L.pamlh_quantile_normal.argtypes = [C.c_double]
    L.pamlh_autod_gamma.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_int]
    for r in (-0.6, -0.1, 0.0, 0.25, 0.7, 0.93, 0.99, -0.95):
        for h, k in ((-1.0, 0.5), (0.3, 0.3), (1.5, -0.7), (-2.0, -1.0), (0.0, 2.0)):
            want = multivariate_normal(mean=[0, 0], cov=[[1, r], [r, 1]]).cdf([-h, -k])        # Pr(X > h, Y > k) by symmetry
            assert abs(L.pamlh_lbinormal(h, k, r) - want) < 5e-7, (h, k, r)
    for p in (0.001, 0.1, 0.25, 0.5, 0.9, 0.999):
        assert abs(L.pamlh_quantile_normal(p) - norm.ppf(p)) < 5e-7 * max(1, abs(norm.ppf(p)) ** 2) + 5e-7
    for K, rho in ((4, 0.3), (5, -0.15), (8, 0.8), (3, 0.98)):
        M, f, rk = np.zeros((K, K)), np.zeros(K), np.zeros(K)
        L.pamlh_autod_gamma(M.ctypes.data_as(C.c_void_p), f.ctypes.data_as(C.c_void_p), rk.ctypes.data_as(C.c_void_p), 0.6, rho, K)
        assert np.allclose(M.sum(axis=1), 1, atol=2e-6) and np.allclose(M, M.T, atol=1e-7) and (M > -1e-9).all()
        assert np.allclose(f, 1.0 / K) and abs(np.dot(f, rk) - 1) < 1e-9
        if rho > 0.9:
            assert np.all(np.diag(M) > 0.75)


@pytest.mark.gpu
@pytest.mark.parametrize("ctl,prog,seed", [("hiv_ns2.ctl", "codeml", 1), ("hiv_ns8.ctl", "codeml", 2), ("lyso_bsa.ctl", "codeml", 3),
                                           ("horai_mg4.ctl", "baseml", 4), ("brown_hky85_g4.ctl", "baseml", 5),
                                           # the model options of round 2: free-ratio and fixed-omega branch models, other genetic codes, AAClasses per
                                           # branch label, the mutation-selection model, omega from amino-acid distances, gamma shapes per gene,
                                           # frequency sets per branch, site classes on the 192-taxon tree with scaling nodes
                                           ("lysos_free.ctl", "codeml", 6), ("lysos_branch_fix.ctl", "codeml", 7), ("hiv_ns0_icode5.ctl", "codeml", 8),
                                           ("mtcdna_aaclass_branch.ctl", "codeml", 9), ("hiv_fmutsel.ctl", "codeml", 10), ("mtcdnapri_aadist1.ctl", "codeml", 11),
                                           ("horai_mg4_malpha.ctl", "baseml", 12), ("brown_hky85_nhomo3.ctl", "baseml", 13), ("mhc_ns2.ctl", "codeml", 14)])
def test_differential_against_the_reference_binary_at_random_parameters(ctl, prog, seed, tmp_path):
    _differential(ctl, prog, seed, tmp_path)


# control files without a committed golden: checked only against the live reference binary, at random parameter vectors
# (codon frequencies as parameters / the mutation-selection model together with branch models)
DIFF_ONLY = [("-", "codeml", "lysos_branch_f3x4est.ctl"), ("-", "codeml", "lysos_free_fmutsel.ctl"),
             ("-", "codeml", "lyso_bsa_f3x4est.ctl"), ("-", "codeml", "ecp_cmc_fmutsel0.ctl"),      # ... and with branch-site model A, clade model C
             ("-", "codeml", "lysin_mg2_f3x4mg.ctl"), ("-", "codeml", "lysin_mg4_f3x4mg.ctl"),      # F3x4MG with the genes' own frequency tables (Mgene = 2, 4)
             ("-", "codeml", "lysos_seqtype3.ctl")]      # seqtype = 3: codons translated on reading, JTT + G4 on the amino acids
DIFF_CASES = [c for c in CASES if "clock" not in c[2] and "tipdate" not in c[2]] + DIFF_ONLY


@pytest.mark.gpu
@pytest.mark.parametrize("gname,prog,ctl", DIFF_CASES)
def test_differential_over_every_golden_control_file(gname, prog, ctl, tmp_path):
    """... and the same for every control file of the golden cases (the clock models aside: their parameters are ordered ages)."""
    _differential(ctl, prog, 100 + sum(map(ord, ctl)), tmp_path)


@pytest.mark.parametrize("gname,prog,ctl", [c for c in DIFF_CASES if not c[2].startswith("mhc_ns")])
def test_differential_of_the_oracle_on_cpu(gname, prog, ctl, tmp_path):
    """The CPU twin of the test above, for the container that holds the reference: the C host's problem at a random parameter vector
    through the oracle against the reference binary's lnL for the same control file and vector (skipped where oracle/_ref is absent)."""
    _differential(ctl, prog, 500 + sum(map(ord, ctl)), tmp_path, on_gpu=False)


@pytest.mark.parametrize("ctl", ["horai_mg0.ctl", "horai_mg2.ctl", "horai_mg4.ctl"])
def test_pattern_format_with_genes(ctl, tmp_path):
    """Option G with the P format (treesub.c:640-665, 954-983: the numbers on the G line are PATTERNS per gene, the site counts follow the
    sequences): horai.nuc's four genes written as their compressed patterns.  The host reads it to the same problem as the plain file
    (same lnL through the oracle at a random parameter vector), and where the reference binary is here it prints that lnL for the
    pattern file too."""
    a = hostlib.Analysis(os.path.join(CTL, ctl), "baseml")
    pb = a.problem(a.default_x())
    assert pb.n_genes == 4
    go = np.asarray(pb.gene_off)
    with open(tmp_path / "horai_pg.nuc", "w") as f:
        f.write("%d %d GP\nG 4 %s\n" % (pb.tree.n_tips, pb.n_patt, " ".join(str(int(go[g + 1] - go[g])) for g in range(4))))
        for i in range(pb.tree.n_tips):
            f.write("s%d  %s\n" % (i + 1, "".join("TCAG"[c] for c in pb.z[i])))
        f.write(" ".join(str(int(w)) for w in pb.weights) + "\n")
    import shutil
    shutil.copy(os.path.join(helpers.GOLDEN, "data", "horai.trees"), tmp_path / "horai.trees")
    text = open(os.path.join(CTL, ctl)).read().replace("../data/horai.nuc", "horai_pg.nuc").replace("../data/", "")
    (tmp_path / "baseml.ctl").write_text(text + "\noutfile = mlb\nnoisy = 0\nverbose = 0\nrunmode = 0\ngetSE = 0\nRateAncestor = 0\n")
    b = hostlib.Analysis(str(tmp_path / "baseml.ctl"), "baseml")
    assert (b.n_patt, b.np, b.ls) == (a.n_patt, a.np, a.ls)
    rng = np.random.default_rng(11)
    lo, hi = a.bounds()
    x = np.round(np.clip(a.default_x() * rng.uniform(0.7, 1.4, a.np), lo * 1.5, np.minimum(hi * 0.9, 50)), 6)
    pa, pbb = a.problem(x), b.problem(x)
    assert np.array_equal(pa.z, pbb.z) and np.array_equal(pa.weights, pbb.weights) and np.array_equal(pa.gene_off, pbb.gene_off)
    la, lb = oracle.evaluate(pa, want_lnf=False)["lnL"], oracle.evaluate(pbb, want_lnf=False)["lnL"]
    assert abs(la - lb) <= 1e-9 * abs(la)
    exe = os.path.join(helpers.REPO, "oracle", "_ref", "baseml")
    if os.access(exe, os.X_OK):
        (tmp_path / "in.baseml").write_text("-1 " + " ".join("%.6f" % v for v in x) + "\n")
        r = subprocess.run([exe, "baseml.ctl"], cwd=tmp_path, capture_output=True, text=True, input="\n" * 50, timeout=600)
        m = re.findall(r"lnL\(ntime:[^\n]*?(-[0-9]+\.[0-9]+)", open(tmp_path / "mlb").read())
        assert m, r.stdout[-2000:]
        assert abs(lb - float(m[-1])) <= 2e-6 * max(1.0, abs(lb) / 1000), (lb, m[-1])


@pytest.mark.parametrize("prog,ctl", [("baseml", "brown_hky85_clock.ctl"), ("baseml", "brown_hky85_clock2.ctl"), ("baseml", "hiv2_tipdate.ctl"),
                                      ("baseml", "hiv2_tipdate_clock2.ctl"), ("codeml", "lysos_m0_clock.ctl")])
def test_differential_of_the_clock_models_on_cpu(prog, ctl, tmp_path):
    """The clock models (their parameters are node ages: a random vector would put nodes above their ancestors) at the host's own initial
    values against the reference binary: global and local clocks, dated tips, and the codon model M0 under a global clock."""
    _differential(ctl, prog, None, tmp_path, on_gpu=False)


VARIANTS = {      # name: (control file, program, (text, replacement) ...)
    "rooted tree, no clock": ("lysos_m0_clock.ctl", "codeml", ("clock = 1", "clock = 0")),
    "stewart cleandata = 1": ("stewart_lg_g4.ctl", "codeml", ("cleandata = 0", "cleandata = 1")),
    "lysozyme cleandata = 1": ("lysos_free.ctl", "codeml", ("cleandata = 0", "cleandata = 1")),
    "M0 kappa fixed": ("hiv_ns0.ctl", "codeml", ("fix_kappa = 0", "fix_kappa = 1")),
    "M0 omega fixed": ("hiv_ns0.ctl", "codeml", ("fix_omega = 0", "fix_omega = 1")),
    "M2a w2 fixed at 1": ("hiv_ns2.ctl", "codeml", ("fix_omega = 0", "fix_omega = 1"), ("omega = 1.3", "omega = 1")),      # the null of the M2a test with w2 = 1
    "M8 ws fixed at 1": ("hiv_ns8.ctl", "codeml", ("fix_omega = 0", "fix_omega = 1"), ("omega = 1.3", "omega = 1")),       # M8a
    "M1a fix_omega ignored": ("hiv_ns1.ctl", "codeml", ("fix_omega = 0", "fix_omega = 1")),
    "M7 fix_omega ignored": ("hiv_ns7.ctl", "codeml", ("fix_omega = 0", "fix_omega = 1")),
}
VARIANTS.update({
    "clade model C, last omega fixed": ("ecp_cmc.ctl", "codeml", ("fix_omega = 0", "fix_omega = 1")),
    "M3 with 5 classes": ("hiv_ns3.ctl", "codeml", ("ncatG = 3", "ncatG = 5")),
    "M7 with 5 classes": ("hiv_ns7.ctl", "codeml", ("ncatG = 10", "ncatG = 5")),
    "M0 + gamma": ("hiv_ns0.ctl", "codeml", ("ncatG = 10", "ncatG = 4\n fix_alpha = 0\n alpha = 0.7")),
    "baseml alpha fixed": ("brown_hky85_g4.ctl", "baseml", ("fix_alpha = 0.5", "fix_alpha = 1")),
    "mt code, M8 with 5 classes": ("mtcdna_m0.ctl", "codeml", ("NSsites = 0", "NSsites = 8\n ncatG = 5")),
    "aa model without gamma": ("stewart_lg_g4.ctl", "codeml", ("fix_alpha = 0", "fix_alpha = 1"), ("alpha = 0.5", "alpha = 0")),
    "branch-site model B": ("lyso_bsa.ctl", "codeml", ("NSsites = 2", "NSsites = 3")),
    "F3x4MG + M2a": ("hiv_ns2.ctl", "codeml", ("CodonFreq = 2", "CodonFreq = 5")),
    "F1x4MG + M7": ("hiv_ns7.ctl", "codeml", ("CodonFreq = 2", "CodonFreq = 4")),
    "aaDist = 2 (Miyata, geometric)": ("mtcdnapri_aadist1.ctl", "codeml", ("aaDist = 1", "aaDist = 2")),
    "aaDist = -1 (Grantham, linear)": ("mtcdnapri_aadist1.ctl", "codeml", ("aaDist = 1", "aaDist = -1")),
    "nhomo = 1 with REV": ("brown_hky85_nhomo1.ctl", "baseml", ("model = 4", "model = 7")),
    "nhomo = 2 with F84": ("brown_hky85_nhomo2.ctl", "baseml", ("model = 4", "model = 3")),
    "Mgene = 4 with REV": ("horai_mg4.ctl", "baseml", ("model = 4", "model = 7")),
    "Mgene = 2 with TN93": ("horai_mg2.ctl", "baseml", ("model = 4", "model = 6")),
    "Mgene = 3, codons, F1x4": ("lysin_mg3.ctl", "codeml", ("CodonFreq = 2", "CodonFreq = 1")),
    # an option given twice: the later line wins (the reference reads the control file line by line)
    "Mgene = 3 + gamma, options repeated": ("horai_mg3.ctl", "baseml", ("Mgene = 3", "Mgene = 3\n fix_alpha = 0\n alpha = 0.5\n ncatG = 4")),
})
VARIANTS.update({"baseml model = %d" % m: ("brown_hky85_g4.ctl", "baseml", ("model = 4", "model = %d" % m)) for m in range(9)})
VARIANTS.update({"aa model = %d" % m: ("stewart_lg_g4.ctl", "codeml", ("model = 2", "model = %d" % m)) for m in range(4)})
VARIANTS.update({"CodonFreq = %d, M0" % c: ("hiv_ns0.ctl", "codeml", ("CodonFreq = 2", "CodonFreq = %d" % c)) for c in (0, 1, 3)})
VARIANTS.update({"CodonFreq = %d, M8" % c: ("hiv_ns8.ctl", "codeml", ("CodonFreq = 2", "CodonFreq = %d" % c)) for c in (0, 1, 3)})


@pytest.mark.parametrize("variant", sorted(VARIANTS))
def test_differential_of_option_variants_on_cpu(variant, tmp_path):
    """Variants of the golden control files, written on the fly: every baseml substitution model (JC69, K80, F81, F84, HKY85, T92, TN93, REV,
    UNREST) with gamma rates on brown.nuc, the amino-acid models 0-3, CodonFreq 0 / 1 / 3, fixed kappa / omega (M2a with w2 = 1 and M8a among
    them), a rooted tree analysed without a clock (one branch more than the unrooted tree: the reference warns and goes on), and
    cleandata = 1 on alignments with ambiguity characters (sites removed) — random parameters, live reference."""
    base, prog = VARIANTS[variant][:2]
    txt = open(os.path.join(CTL, base)).read()
    for a_, b_ in VARIANTS[variant][2:]:
        assert a_ in txt, (base, a_)
        txt = txt.replace(a_, b_)
    name = "variant_%d.ctl" % (abs(hash(variant)) % 10 ** 8)
    path = os.path.join(CTL, name)      # beside the others: the ../data/ paths stay valid
    open(path, "w").write(txt)
    try:
        _differential(name, prog, 40 + sum(map(ord, variant)), tmp_path, on_gpu=False)
    finally:
        os.remove(path)


# NSsites 9 .. 13 (M9 - M13) in the differential tests: the reference places the omega classes by inverting the mixture's CDF with a line search on
# (CDF - p)^2 that starts from the PREVIOUS call's classes and stops at ~1e-5 in omega (Quantile(CDFdN_dS, ...) in DiscreteNSsites,
# codeml.c:2877): its printed lnL moves by a few 1e-3 with the starting point (see the hiv_m11 golden's note), so for these five models the
# live binary pins the engine to 5e-3 only; every other model to the printed digits (2e-6).  Tighter pins for M9 - M13 are the goldens with
# fixed class tables (tests/golden/hiv_m9 .. hiv_m13: the classes the reference itself used).
TOL_LNL_PRINTED_DIGITS = 2e-6
TOL_LNL_M9_TO_M13_AGAINST_THE_LIVE_BINARY = 5e-3


def _differential(ctl, prog, seed, tmp_path, on_gpu=True):
    """Beyond the committed vectors: the unmodified reference binary (oracle/_ref, when it travelled with the repository) and the
    engine evaluate the same control file at a RANDOM parameter vector inside the bounds; lnL must agree to the printed digits
    (TOL_LNL_PRINTED_DIGITS) — to TOL_LNL_M9_TO_M13_AGAINST_THE_LIVE_BINARY = 5e-3 for NSsites 9 .. 13, see above."""
    import shutil
    exe = os.path.join(helpers.REPO, "oracle", "_ref", prog)
    if not os.access(exe, os.X_OK):
        pytest.skip("oracle/_ref/%s is not here (it is built from /root/reference in the build container)" % prog)
    a = hostlib.Analysis(os.path.join(CTL, ctl), prog)
    lo, hi = a.bounds()
    if seed is None:
        x = np.round(a.default_x(), 6)
    else:
        rng = np.random.default_rng(seed)
        x = a.default_x() * np.where(hi <= 1, rng.uniform(0.7, 1.0, a.np), rng.uniform(0.7, 1.4, a.np))     # (proportions only shrink: they stay feasible)
        x = np.round(np.clip(x, lo * 1.5, np.minimum(hi * 0.9, 50)), 6)           # in.codeml carries six decimals
    text = open(os.path.join(CTL, ctl)).read()
    for name in re.findall(r"\.\./data/(\S+)", text):
        shutil.copy(os.path.join(helpers.GOLDEN, "data", name), tmp_path / name)
    for name in ("OmegaAA.dat", "grantham.dat", "miyata.dat"):      # read by fixed names from the working directory (aaDist models)
        for d in (CTL, os.path.join(helpers.GOLDEN, "data")):
            if os.path.exists(os.path.join(d, name)):
                shutil.copy(os.path.join(d, name), tmp_path / name)
    main = "mlc" if prog == "codeml" else "mlb"
    (tmp_path / (prog + ".ctl")).write_text(text.replace("../data/", "") + "\noutfile = %s\nnoisy = 0\nverbose = 0\nrunmode = 0\ngetSE = 0\nRateAncestor = 0\n" % main)
    (tmp_path / ("in." + prog)).write_text("-1 " + " ".join("%.6f" % v for v in x) + "\n")
    r = subprocess.run([exe, prog + ".ctl"], cwd=tmp_path, capture_output=True, text=True, input="\n" * 50, timeout=600)
    m = re.findall(r"lnL\(ntime:[^\n]*?(-[0-9]+\.[0-9]+)", open(tmp_path / main).read())
    assert m, r.stdout[-2000:]
    ref = float(m[-1])
    got = a.eval_gpu(x, want_lnf=False)[0] if on_gpu else oracle.evaluate(a.problem(x), want_lnf=False)["lnL"]
    tol = TOL_LNL_M9_TO_M13_AGAINST_THE_LIVE_BINARY if re.search(r"hiv_ns(9|1[0-3])\.ctl", ctl) else TOL_LNL_PRINTED_DIGITS * max(1.0, abs(ref) / 1000)
    assert abs(got - ref) <= tol, (got, ref)
