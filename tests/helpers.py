"""Test glue: rebuild engine/oracle inputs (paml_amd.problem.Problem) from the golden JSON fixtures
written by tests/golden/make_golden.py, using the numpy host-side model set-up (paml_amd.models)."""
from __future__ import annotations

import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
if os.path.join(REPO, "oracle") not in sys.path:
    sys.path.insert(0, os.path.join(REPO, "oracle"))

from paml_amd import models, synth  # noqa: E402
from paml_amd.problem import (EIGEN_CIJK, EIGEN_UVROOT, MODE_LFUN, MODE_LFUNDG, Problem, parse_newick,  # noqa: E402
                              set_node_scale)

GOLDEN = os.path.join(REPO, "tests", "golden")


def load_golden(name):
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        return json.load(f)


nssites_classes = models.nssites_classes


EQUATE_BASE = dict(zip("TCAGUYRMKSWHBVD-N?", ["T", "C", "A", "G", "T", "TC", "AG", "CA", "TG", "CG", "TA", "TCA", "TCG", "CAG", "TAG",
                                             "TCAG", "TCAG", "TCAG"]))   # tools.c:15-17


def codon_codes_with_ambiguity(patterns_raw):
    """z, n_chara, chara_map for codon patterns that may hold ambiguous codons (EncodeSeqs treesub.c:1145-1172 +
    SetMapAmbiguity treesub.c:1253-1272): sense codons are codes 0..60, every distinct ambiguous triplet gets the next
    code and maps to its compatible sense codons in the reference's i0,i1,i2 expansion order."""
    from61 = models.sense_codons()
    from64 = {c: i for i, c in enumerate(from61)}
    sense = {"".join(models.BASES[(c >> s) & 3] for s in (4, 2, 0)): i for i, c in enumerate(from61)}
    amb = {}
    rows = []
    for pat in patterns_raw:
        row = []
        for tok in pat.split():
            if tok in sense:
                row.append(sense[tok])
            else:
                if tok not in amb:
                    amb[tok] = 61 + len(amb)
                row.append(amb[tok])
        rows.append(row)
    z = np.array(rows, dtype=np.uint8).T
    n_codes = 61 + len(amb)
    n_chara = np.ones(n_codes, dtype=np.int32)
    cmap = np.zeros((n_codes, 61), dtype=np.uint8)
    cmap[:61, 0] = np.arange(61)
    for tok, code in amb.items():
        lst = []
        for b0 in EQUATE_BASE[tok[0]]:
            for b1 in EQUATE_BASE[tok[1]]:
                for b2 in EQUATE_BASE[tok[2]]:
                    ic = models.BASES.index(b0) * 16 + models.BASES.index(b1) * 4 + models.BASES.index(b2)
                    if ic in from64:
                        lst.append(from64[ic])
        n_chara[code] = len(lst)
        cmap[code, :len(lst)] = lst
    return z, n_chara, cmap


def f3x4_with_ambiguity(patterns_raw, w, iters=20):
    """F3x4 when ambiguous codons are present, as InitializeCodon does it (codeml.c:3772-3850): start from the fully
    resolved codons only (CountCodons 3671-3690), then up to 20 rounds in which every nucleotide position of every
    codon adds fpatt * fb3x4_old[pos][b] / sum over its ambiguity set (AddCodonFreqSeqGene 3726-3750, per position,
    stop codons not considered), renormalise, stop when the change is < 1e-8."""
    bidx = {b: i for i, b in enumerate(models.BASES)}
    sets = {c: [bidx[x] for x in EQUATE_BASE[c]] for c in EQUATE_BASE}
    toks = [p.split() for p in patterns_raw]
    fb0 = np.zeros((3, 4))
    for pat, wt in zip(toks, w):
        for tok in pat:
            s = [sets[ch] for ch in tok]
            if len(s[0]) * len(s[1]) * len(s[2]) > 1:
                continue
            for k in range(3):
                fb0[k, s[k][0]] += wt
    fb0 /= fb0.sum(axis=1, keepdims=True)
    for _ in range(iters):
        fb = np.zeros((3, 4))
        for pat, wt in zip(toks, w):
            for tok in pat:
                for k in range(3):
                    s = sets[tok[k]]
                    t = fb0[k, s].sum()
                    fb[k, s] += wt * fb0[k, s] / t
        fb /= fb.sum(axis=1, keepdims=True)
        d = np.sqrt(((fb - fb0) ** 2).sum())
        fb0 = fb
        if d < 1e-8:
            break
    return fb0


def problem_from_golden(g) -> Problem:
    amb = None
    if "patterns_raw" in g and g["seqtype"] == "codon":
        z, nch, cm = codon_codes_with_ambiguity(g["patterns_raw"])
        w = np.array(g["counts"], dtype=float)
        amb = (nch, cm)
    elif "z" in g:
        z = np.array(g["z"], dtype=np.uint8)
        w = np.array(g["counts"], dtype=float)
    else:
        gen = dict(g["generator"])
        fn = getattr(synth, gen.pop("fn"))
        base = fn(**gen)
        z, w = base.z, base.weights
    tree = parse_newick(g["tree"], names=g.get("names"))
    m = g["model"]
    kind = m["kind"]
    if kind == "aa_synth_gamma":          # the synthetic 20-state model: everything comes from the generator
        gen = dict(g["generator"])
        return getattr(synth, gen.pop("fn"))(**gen)
    if kind == "codon_m0":
        kw = {}
        if amb is not None:
            pi = models.f3x4(f3x4_with_ambiguity(g["patterns_raw"], w))
            kw = dict(cleandata=0, n_chara=amb[0], chara_map=amb[1])
        else:
            pi = models.f3x4(synth.f3x4_from_codon_tips(z, w))
        if g.get("scale_nodes"):
            sc = np.zeros(tree.n_nodes, dtype=np.uint8)
            sc[np.array(g["scale_nodes"]) - 1] = 1          # the reference prints 1-based node numbers
            kw["scale_node"] = sc
        U, V, root, _ = models.codon_m0_eigen(m["kappa"], m["omega"], pi)
        return Problem(n=61, tree=tree, z=z, weights=w, pi=pi, eigen=[dict(kind=EIGEN_UVROOT, U=U, V=V, Root=root)],
                       mode=MODE_LFUN, **kw)
    if kind == "aa_empirical_gamma":
        S, pi = models.read_aa_ratefile(os.path.join(GOLDEN, "data", m["ratefile"]))
        U, V, root = models.aa_empirical_eigen(S, pi)
        n_chara, cmap = models.aa_code_map()
        freqK, rK = models.discrete_gamma(m["alpha"], m["ncatG"])
        return Problem(n=20, tree=tree, z=z, weights=w, pi=pi, eigen=[dict(kind=EIGEN_UVROOT, U=U, V=V, Root=root)],
                       mode=MODE_LFUNDG, freqK=freqK, rate=rK, cleandata=0, n_chara=n_chara, chara_map=cmap)
    if kind == "codon_nssites":
        x = g["x"]
        nt = g["ntime"]
        if "kappa" in m:                      # kappa fixed in the control file: x holds the class parameters only
            kappa, par = m["kappa"], x[nt:]
        else:
            kappa, par = x[nt], x[nt + 1:]
        pi = models.f3x4(synth.f3x4_from_codon_tips(z, w))
        freqs, omegas = nssites_classes(m["NSsites"], par, m["ncatG"])
        if m["NSsites"] == 0:
            U, V, root, _ = models.codon_m0_eigen(kappa, omegas[0], pi)
            return Problem(n=61, tree=tree, z=z, weights=w, pi=pi,
                           eigen=[dict(kind=EIGEN_UVROOT, U=U, V=V, Root=root)], mode=MODE_LFUN)
        base = Problem(n=61, tree=tree, z=z, weights=w, pi=pi, eigen=[], mode=MODE_LFUNDG)
        return synth.codon_nssites_problem(base, kappa, omegas, freqs)
    if kind == "nuc_rev_gamma":
        cnt = np.array([((z == b) * w[None, :]).sum() for b in range(4)])
        pi = cnt / cnt.sum()
        Q = models.gtr_q(m["rates"], pi)
        U, V, root = models.eigen_rev(Q, pi)
        Cijk, rootc, nR = models.cijk_from_uvroot(U, V, root)
        freqK, rK = models.discrete_gamma(m["alpha"], m["ncatG"])
        return Problem(n=4, tree=tree, z=z, weights=w, pi=pi, eigen=[dict(kind=EIGEN_CIJK, Cijk=Cijk, Root=rootc, nR=nR)],
                       mode=MODE_LFUNDG, freqK=freqK, rate=rK)
    if kind == "nuc_hky85":
        cnt = np.array([((z == b) * w[None, :]).sum() for b in range(4)])
        pi = cnt / cnt.sum()
        Q = models.hky_q(m["kappa"], pi)
        U, V, root = models.eigen_rev(Q, pi)
        Cijk, rootc, nR = models.cijk_from_uvroot(U, V, root)
        return Problem(n=4, tree=tree, z=z, weights=w, pi=pi, eigen=[dict(kind=EIGEN_CIJK, Cijk=Cijk, Root=rootc, nR=nR)],
                       mode=MODE_LFUN)
    raise ValueError(kind)


def random_problem(n, n_tips, n_patt, K=1, seed=0, ambiguity=False, scale_every=None, n_genes=1, polytomy=False,
                   mode=None, n_amb=0, amb_rate=0.08):
    """Random reversible model + random tree + random tips: a parity case with no biological meaning."""
    rng = np.random.default_rng(seed)
    pi = rng.dirichlet(np.full(n, 3.0))
    S = rng.gamma(1.0, 1.0, size=(n, n))
    S = np.triu(S, 1)
    S = S + S.T
    Q = S * pi[None, :]
    Q[np.diag_indices(n)] = -Q.sum(axis=1)
    Q /= -np.dot(pi, np.diag(Q))
    U, V, root = models.eigen_rev(Q, pi)
    # random binary tree with trifurcating (or polytomous) root, reference numbering
    sons = [[] for _ in range(2 * n_tips)]
    free = list(range(n_tips))
    rng.shuffle(free)
    nxt = n_tips + 1
    nroot = 3 if not polytomy else 4
    while len(free) > nroot:
        i = int(rng.integers(len(free)))
        a = free.pop(i)
        j = int(rng.integers(len(free)))
        b = free.pop(j)
        node = nxt
        nxt += 1
        sons[node] = [a, b]
        free.append(node)
    root_id = n_tips
    sons[root_id] = free
    n_nodes = nxt
    # renumber internal nodes in '(' order like ReadTreeN
    order = []

    def pre(i):
        if i >= n_tips:
            order.append(i)
        for c in sons[i]:
            pre(c)
    pre(root_id)
    remap = {old: n_tips + k for k, old in enumerate(order)}
    new_sons = [[] for _ in range(n_nodes)]
    for old, new in remap.items():
        new_sons[new] = [remap.get(c, c) for c in sons[old]]
    from paml_amd.problem import Tree
    branch = rng.uniform(0.01, 0.4, size=n_nodes)
    branch[n_tips] = 0
    tree = Tree(n_tips, n_nodes, n_tips, new_sons, branch, np.zeros(n_nodes, dtype=np.int32))
    z = rng.integers(0, n, size=(n_tips, n_patt)).astype(np.uint8)
    # make neighbouring tips agree often so likelihoods are not absurdly small
    base = rng.integers(0, n, size=n_patt)
    keep = rng.random((n_tips, n_patt)) < 0.6
    z = np.where(keep, base[None, :], z).astype(np.uint8)
    w = rng.integers(1, 5, size=n_patt).astype(float)
    kw = {}
    if ambiguity:
        n_codes = n + 3
        n_chara = np.ones(n_codes, dtype=np.int32)
        cmap = np.zeros((n_codes, n), dtype=np.uint8)
        cmap[:n, 0] = np.arange(n)
        n_chara[n] = n
        cmap[n] = np.arange(n)                    # fully missing
        n_chara[n + 1] = 2
        cmap[n + 1, :2] = [1, 3 % n]
        n_chara[n + 2] = 3
        cmap[n + 2, :3] = [0, 2 % n, (n - 1)]
        # n_amb further ambiguity codes (SetMapAmbiguity treesub.c:1218-1286 makes one of every distinct ambiguous triplet): random state
        # sets of 2 .. 16 states in ascending order; with 61 states, n + 3 + n_amb > 64 codes cross the per-tree kernel's ring block
        if n_amb:
            n_chara = np.concatenate([n_chara, np.zeros(n_amb, dtype=np.int32)])
            cmap = np.concatenate([cmap, np.zeros((n_amb, n), dtype=np.uint8)])
            for c in range(n_codes, n_codes + n_amb):
                size = int(rng.integers(2, min(n, 16) + 1))
                n_chara[c] = size
                cmap[c, :size] = np.sort(rng.choice(n, size=size, replace=False))
            n_codes += n_amb
        amb = rng.random((n_tips, n_patt)) < amb_rate
        z = np.where(amb, rng.integers(n, n_codes, size=(n_tips, n_patt)), z).astype(np.uint8)
        kw.update(cleandata=0, n_chara=n_chara, chara_map=cmap)
    if K > 1:
        freqK = rng.dirichlet(np.full(K, 5.0))
        rate = rng.gamma(2.0, 0.5, size=K)
        kw.update(freqK=freqK, rate=rate)
    if n_genes > 1:
        cuts = np.sort(rng.choice(np.arange(1, n_patt), size=n_genes - 1, replace=False))
        kw.update(gene_off=np.concatenate(([0], cuts, [n_patt])).astype(np.int32),
                  gene_rate=rng.uniform(0.5, 1.5, size=n_genes))
    if scale_every:
        kw.update(scale_node=set_node_scale(tree, scale_every))
    if mode is None:
        mode = MODE_LFUNDG if K > 1 else MODE_LFUN
    return Problem(n=n, tree=tree, z=z, weights=w, pi=pi, eigen=[dict(kind=EIGEN_UVROOT, U=U, V=V, Root=root)],
                   mode=mode, **kw)


def give_genes_their_own_models(pb, seed=0):
    """Mgene = 2 .. 4 (com.piG, a rate matrix per gene; treesub.c:487 option G): every gene of `pb` gets its own reversible model —
    frequencies pi[g] and eigen system g, eigen_of[g][class][label] = g."""
    rng = np.random.default_rng(seed)
    n, G = pb.n, pb.n_genes
    pis, eig = [], []
    for g in range(G):
        pi = rng.dirichlet(np.full(n, 3.0))
        S = np.triu(rng.gamma(1.0, 1.0, size=(n, n)), 1)
        S = S + S.T
        Q = S * pi[None, :]
        Q[np.diag_indices(n)] = -Q.sum(axis=1)
        Q /= -np.dot(pi, np.diag(Q))
        U, V, root = models.eigen_rev(Q, pi)
        pis.append(pi)
        eig.append(dict(kind=EIGEN_UVROOT, U=U, V=V, Root=root))
    pb.pi = np.ascontiguousarray(np.stack(pis))
    pb.eigen = eig
    pb.eigen_of = np.ascontiguousarray(np.broadcast_to(np.arange(G, dtype=np.int32)[:, None, None], (G, pb.K, pb.n_labels)))
    return pb


def ymd_names(text):
    """The HIV-2 TipDate data (tests/golden/data/HIV2ge.*) name their sequences by isolate + sampling year (P03h1995): turn the year into a
    calendar date, P03h_1995-MM-DD, month and day drawn from the name — the yyyy-mm-dd flavour of TipDate (golden hiv2_tipdate_ymd was
    generated by the reference from files renamed by this same function)."""
    import re

    def sub(m):
        name, year = m.group(1), m.group(2)
        k = sum(ord(c) for c in name)
        return "%s_%s-%02d-%02d" % (name, year, 1 + k % 12, 1 + k % 28)
    return re.sub(r"\b([A-Za-z0-9]*?[A-Za-z])((?:19|20)\d\d)\b", sub, text)
