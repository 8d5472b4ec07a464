#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the UNMODIFIED reference programs
(oracle/_ref/codeml, oracle/_ref/baseml — built by oracle/Makefile from /root/reference/src) in the
deterministic single-evaluation mode (`in.codeml` / `in.baseml` starting with -1,
or fix_blength=2 with every parameter fixed).

Only runs in the build container (needs /root/reference for the example data files and oracle/_ref).
What is committed is data: each JSON holds the model settings, the parameter vector, the tree with
branch lengths as the reference printed it, the site patterns with their counts as written to the
reference's `lnf` file, and the reference's lnL / per-pattern log f_h.

usage: python tests/golden/make_golden.py [case ...]
"""
from __future__ import annotations

import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
from paml_amd import models, synth  # noqa: E402

REF = os.path.join(REPO, "oracle", "_ref")
EX = "/root/reference/examples"
DAT = "/root/reference/dat"

CODEML_BASE = dict(noisy=3, verbose=0, runmode=0, seqtype=1, CodonFreq=2, clock=0, aaDist=0, model=0, NSsites=0,
                   icode=0, Mgene=0, fix_kappa=0, kappa=2, fix_omega=0, omega=0.4, fix_alpha=1, alpha=0, Malpha=0,
                   ncatG=10, getSE=0, RateAncestor=0, Small_Diff=".5e-6", cleandata=1, fix_blength=0, method=0)
BASEML_BASE = dict(noisy=3, verbose=0, runmode=0, model=4, Mgene=0, clock=0, fix_kappa=0, kappa=5, fix_alpha=1,
                   alpha=0, Malpha=0, ncatG=4, nparK=0, nhomo=0, getSE=0, RateAncestor=0, Small_Diff="7e-6",
                   cleandata=1, method=0)


def run_ref(prog, ctl, files, x=None, timeout=3600):
    d = tempfile.mkdtemp(prefix="golden_")
    try:
        for dst, src in files.items():
            if os.path.exists(str(src)):
                shutil.copy(src, os.path.join(d, dst))
            else:
                with open(os.path.join(d, dst), "w") as f:
                    f.write(src)
        with open(os.path.join(d, prog + ".ctl"), "w") as f:
            for k, v in ctl.items():
                f.write("%s = %s\n" % (k, v))
        if x is not None:
            with open(os.path.join(d, "in." + prog), "w") as f:
                f.write("-1 " + " ".join("%.6f" % v for v in x) + "\n")
        out = subprocess.run([os.path.join(REF, prog), prog + ".ctl"], cwd=d, stdout=subprocess.PIPE,
                             stderr=subprocess.STDOUT, timeout=timeout, input=b"\n" * 50).stdout.decode(errors="replace")
        m = re.findall(r"lnL\s*=\s*(-?[0-9.]+)", out)
        if not m:
            raise RuntimeError("no lnL in reference output:\n" + out[-3000:])
        lnL = float(m[-1])
        cnt = re.search(r"(\d+) lfun, (\d+) eigenQcodon, (\d+) P\(t\)", out)
        qf = re.search(r"Qfactor_NS = ([0-9.]+)", out)
        with open(os.path.join(d, "lnf")) as f:
            lnf_lines = f.read().splitlines()
        main = "mlc" if prog == "codeml" else "mlb"
        with open(os.path.join(d, main)) as f:
            mtxt = f.read()
        rst = ""
        if os.path.exists(os.path.join(d, "rst")):
            with open(os.path.join(d, "rst")) as f:
                rst = f.read()
        return dict(lnL=lnL, stdout=out, lnf=lnf_lines, main=mtxt, rst=rst,
                    counters=[int(g) for g in cnt.groups()] if cnt else None,
                    qfactor_ns=float(qf.group(1)) if qf else None)
    finally:
        shutil.rmtree(d, ignore_errors=True)


AA3 = dict(zip("Ala Arg Asn Asp Cys Gln Glu Gly His Ile Leu Lys Met Phe Pro Ser Thr Trp Tyr Val".split(), "ARNDCQEGHILKMFPSTWYV"))


def parse_lnf(lines, seqtype, n_tips):
    hdr = None
    pats = []
    for ln in lines:
        t = ln.split()
        if not t:
            continue
        if hdr is None:
            if len(t) == 3:
                hdr = [int(v) for v in t]
            continue
        if len(t) < 6:
            continue
        idx, cnt, logf = int(t[0]), float(t[1]), float(t[2])
        rest = ln.split(None, 5)[5]
        if seqtype == "aa3":      # aa model 5: the reference prints the data as three-letter amino-acid names
            toks = [AA3[t3] for t3 in rest.split()]
        elif seqtype == "codon":
            toks = re.findall(r"([A-Z\-\?]{3}) \(.\)", rest)
        else:
            toks = list(rest.split()[0])
        assert len(toks) == n_tips, (ln, toks)
        pats.append((cnt, logf, toks))
    assert hdr and len(pats) == hdr[2], (hdr, len(pats))
    return hdr, pats


def tree_from_main(mtxt):
    """First Newick string after 'tree length =' that carries branch lengths (codeml prints the
    numbered tree with lengths first; baseml prints the bare topology first, then names + lengths)."""
    if "tree length =" not in mtxt:      # (TipDate output has no such line)
        return None
    tail = mtxt[mtxt.index("tree length ="):]
    for m in re.finditer(r"^\(.*;\s*$", tail, re.M):
        if ":" in m.group(0):
            return m.group(0).strip()
    raise RuntimeError("no tree with branch lengths in main output")


def encode(pats, seqtype):
    if seqtype == "codon":
        from61 = models.sense_codons()
        lut = {"".join(models.BASES[(c >> s) & 3] for s in (4, 2, 0)): i for i, c in enumerate(from61)}
    elif seqtype == "nuc":
        lut = {b: i for i, b in enumerate(models.BASES)}
    else:
        lut = {a: i for i, a in enumerate(models.AAS + "-*?X")}
    z = np.array([[lut.get(tok, 255) for tok in p[2]] for p in pats], dtype=int).T
    return z


def finish(name, res, seqtype, n_tips, extra, keep_raw_patterns=False, sample=None):
    hdr, pats = parse_lnf(res["lnf"], seqtype, n_tips)
    z = encode(pats, seqtype)
    g = dict(name=name, seqtype=seqtype, n_tips=n_tips, ls=hdr[1], n_patt=hdr[2], lnL=res["lnL"],
             counters=res["counters"], qfactor_ns=res["qfactor_ns"], tree=tree_from_main(res["main"]))
    g.update(extra)
    counts = [p[0] for p in pats]
    logf = [p[1] for p in pats]
    if sample is None:
        g["counts"] = counts
        g["logf"] = logf
        if (z == 255).any() or keep_raw_patterns:
            g["patterns_raw"] = ["".join(p[2]) if seqtype != "codon" else " ".join(p[2]) for p in pats]
        g["z"] = z.tolist()
    else:   # large synthetic: data are regenerated from the seeded generator; keep a strided sample + checksum
        idx = list(range(0, hdr[2], sample))
        g["sample_stride"] = sample
        g["logf_sample"] = [logf[i] for i in idx]
        g["logf_sum"] = float(np.sum(logf))
        g["z_crc"] = int(np.bitwise_xor.reduce((z.astype(np.uint64) + 1).ravel() * np.arange(1, z.size + 1, dtype=np.uint64) % np.uint64(1000003)))
    path = os.path.join(HERE, name + ".json")
    with open(path, "w") as f:
        json.dump(g, f, separators=(",", ":"))
    print("%-22s lnL %.6f  npatt %d  -> %s (%d bytes)" % (name, g["lnL"], g["n_patt"], os.path.basename(path), os.path.getsize(path)))
    return g


# ---------------------------------------------------------------- cases
HIV_X = {
    "m0": (0, "0.023746 0.079178 0.152813 0.023647 0.050407 0.083460 0.028429 0.099890 0.048101 0.131681 0.033695 0.191150 0.064013 0.193336 0.050591 0.034773 0.036688 0.055603 0.018984 0.070370 0.030301 0.061120 0.198346 2.471747 0.901293"),
    "m1a": (1, "0.024595 0.086392 0.169984 0.024829 0.052083 0.092797 0.029325 0.107730 0.052269 0.144164 0.025220 0.218323 0.067142 0.220681 0.061023 0.022418 0.040795 0.058372 0.019831 0.073759 0.032272 0.063152 0.232592 2.594607 0.484176 0.078848"),
    "m2a": (2, "0.025590 0.088873 0.175160 0.026711 0.046977 0.105490 0.026404 0.112617 0.055482 0.148326 0.025435 0.246061 0.071540 0.250031 0.071600 0.000645 0.038818 0.058692 0.020604 0.073256 0.033954 0.067258 0.273578 2.785547 0.377120 0.441686 0.059978 3.625638"),
    "m7": (7, "0.024444 0.085984 0.169100 0.024862 0.052297 0.091846 0.029496 0.107231 0.051525 0.143550 0.027371 0.215468 0.066975 0.219325 0.058436 0.025357 0.040518 0.058555 0.019942 0.074112 0.032282 0.063201 0.229527 2.560660 0.147502 0.118171"),
    "m8": (8, "0.025623 0.088848 0.175537 0.026537 0.046922 0.104911 0.026895 0.112264 0.055032 0.148482 0.025965 0.245015 0.070339 0.249760 0.071055 0.001661 0.039215 0.058750 0.020579 0.073327 0.033893 0.066500 0.272705 2.786897 0.799504 0.167185 0.148826 3.470359"),
}


def case_hiv(which):
    ns, xs = HIV_X[which]
    x = [float(v) for v in xs.split()]
    ctl = dict(CODEML_BASE, seqfile="HIVenvSweden.txt", treefile="HIVenvSweden.trees", outfile="mlc", NSsites=ns,
               kappa=.3, omega=1.3, ncatG=10)
    res = run_ref("codeml", ctl, {"HIVenvSweden.txt": EX + "/HIVNSsites/HIVenvSweden.txt",
                                  "HIVenvSweden.trees": EX + "/HIVNSsites/HIVenvSweden.trees"}, x=x)
    classes = None
    m = re.search(r"\np:\s+([0-9. ]+)\nw:\s+([0-9. ]+)", res["main"])
    if m:
        classes = dict(p=[float(v) for v in m.group(1).split()], w=[float(v) for v in m.group(2).split()])
    finish("hiv_" + which, res, "codon", 13,
           dict(program="codeml", model=dict(kind="codon_nssites", NSsites=ns, codonfreq="F3x4", ncatG=10), x=x,
                ntime=23, classes_printed=classes))


def case_syn_codon(n_patt=2000, name="syn_codon_m0", sample=None):
    pb = synth.codon_m0_problem(n_tips=16, n_patt=n_patt, estimate_pi=True)
    d = tempfile.mkdtemp()
    synth.write_pattern_file(os.path.join(d, "seq.txt"), pb.z, pb.weights, "codon")
    tree = " 16 1\n" + pb.tree.newick() + "\n"
    ctl = dict(CODEML_BASE, seqfile="seq.txt", treefile="tree.txt", outfile="mlc", fix_kappa=1, kappa=2, fix_omega=1,
               omega=0.4, fix_blength=2)
    res = run_ref("codeml", ctl, {"seq.txt": os.path.join(d, "seq.txt"), "tree.txt": tree})
    shutil.rmtree(d)
    finish(name, res, "codon", 16,
           dict(program="codeml", model=dict(kind="codon_m0", kappa=2.0, omega=0.4, codonfreq="F3x4"),
                generator=dict(fn="codon_m0_problem", n_tips=16, n_patt=n_patt, seed=20260926, estimate_pi=True)),
           sample=sample)


def case_syn_nuc(n_patt=5000, name="syn_nuc_gtr_g4", sample=None):
    pb = synth.nuc_gtr_gamma_problem(n_tips=32, n_patt=n_patt)
    d = tempfile.mkdtemp()
    synth.write_pattern_file(os.path.join(d, "seq.txt"), pb.z, pb.weights, "nuc")
    tree = " 32 1\n" + pb.tree.newick() + "\n"
    ctl = dict(BASEML_BASE, seqfile="seq.txt", treefile="tree.txt", outfile="mlb", model=7, fix_alpha=0, alpha=0.5,
               ncatG=4, fix_blength=2)
    x = list(synth.GTR_RATES) + [0.5]
    res = run_ref("baseml", ctl, {"seq.txt": os.path.join(d, "seq.txt"), "tree.txt": tree}, x=x)
    shutil.rmtree(d)
    finish(name, res, "nuc", 32,
           dict(program="baseml", model=dict(kind="nuc_rev_gamma", rates=list(synth.GTR_RATES), alpha=0.5, ncatG=4), x=x,
                generator=dict(fn="nuc_gtr_gamma_problem", n_tips=32, n_patt=n_patt, seed=20260927)),
           sample=sample)


def case_syn_aa(n_patt=100_000, name="syn_aa_g4_full", sample=97):
    """The 20-state configuration at scale (C3's model class on 32 taxa x 10^5 patterns): codeml seqtype 2, model 2 with the synthetic rate
    file, gamma with four classes, alpha fixed at the generating value, branch lengths fixed at the tree file's."""
    pb = synth.aa_gamma_problem(n_tips=32, n_patt=n_patt)
    d = tempfile.mkdtemp()
    synth.write_pattern_file(os.path.join(d, "seq.txt"), pb.z, pb.weights, "aa")
    S, pi = synth.aa_model_tables()
    synth.write_aa_ratefile(os.path.join(d, "rand.dat"), S, pi)
    tree = " 32 1\n" + pb.tree.newick() + "\n"
    ctl = dict(CODEML_BASE, seqfile="seq.txt", treefile="tree.txt", outfile="mlc", seqtype=2, model=2, aaRatefile="rand.dat",
               fix_alpha=1, alpha=0.5, ncatG=4, fix_blength=2, cleandata=1)
    for k in ("CodonFreq", "NSsites", "fix_kappa", "kappa", "fix_omega", "omega", "icode"):
        ctl.pop(k, None)
    res = run_ref("codeml", ctl, {"seq.txt": os.path.join(d, "seq.txt"), "tree.txt": tree, "rand.dat": os.path.join(d, "rand.dat")})
    shutil.rmtree(d)
    finish(name, res, "aa", 32,
           dict(program="codeml", model=dict(kind="aa_synth_gamma", alpha=0.5, ncatG=4), x=[],
                generator=dict(fn="aa_gamma_problem", n_tips=32, n_patt=n_patt, seed=20260928)),
           sample=sample)


def case_tree_comparison():
    """Two trees in one tree file (examples/stewart.trees with its header corrected to "6 2"), LG + G4 with alpha estimated on each: the
    reference evaluates both, writes both sets of per-pattern log f_h to `lnf` and prints the comparison table of rell()."""
    trees = open(EX + "/stewart.trees").read().replace("6  1", "6  2", 1)
    ctl = dict(CODEML_BASE, seqfile="stewart.aa", treefile="two.trees", outfile="mlc", seqtype=2, model=2, aaRatefile="lg.dat",
               fix_alpha=0, alpha=0.5, ncatG=4, cleandata=0)
    res = run_ref("codeml", ctl, {"stewart.aa": EX + "/stewart.aa", "two.trees": trees, "lg.dat": DAT + "/lg.dat"})
    hdr = [int(v) for v in [ln for ln in res["lnf"] if ln.split()][0].split()]
    blocks, cur = [], None
    for ln in res["lnf"][1:]:
        t = ln.split()
        if len(t) == 1 and t[0].isdigit():
            cur = []
            blocks.append(cur)
        elif len(t) >= 6 and cur is not None:
            cur.append((float(t[1]), float(t[2])))
    assert len(blocks) == 2 and all(len(b) == hdr[2] for b in blocks), (hdr, [len(b) for b in blocks])
    tab = re.findall(r"^\s*(\d+)(\*?)\s+(-?[0-9.]+)\s+(-?[0-9.]+)\s+(-?[0-9.]+)\s+(-?[0-9.]+)\s+(-?[0-9.]+)\s+(-?[0-9.]+)\s*$",
                     res["main"][res["main"].index("Tree comparisons"):], re.M)
    g = dict(name="stewart_two_trees", program="codeml", ls=hdr[1], n_patt=hdr[2], lnL=float(tab[0][2]),
             counts=[c for c, _ in blocks[0]], logf=[[v for _, v in b] for b in blocks],
             table=[dict(tree=int(r[0]), best=bool(r[1]), li=float(r[2]), dli=float(r[3]), se=float(r[4]), pKH=float(r[5]), pSH=float(r[6]), pRELL=float(r[7])) for r in tab],
             n_rep=int(re.search(r"Number of replicates: (\d+)", res["main"]).group(1)),
             mle_lnL=[float(v) for v in re.findall(r"lnL\(ntime:[^)]*\):\s*(-?[0-9.]+)", res["main"])])
    with open(os.path.join(HERE, g["name"] + ".json"), "w") as f:
        json.dump(g, f, separators=(",", ":"))
    print("   stewart_two_trees:", g["table"])


def case_brown():
    x = [float(v) for v in "0.053057 0.017471 0.041370 0.053761 0.057580 0.100159 0.138990 9.389630".split()]
    ctl = dict(BASEML_BASE, seqfile="brown.nuc", treefile="brown.trees", outfile="mlb", model=4, ncatG=1)
    res = run_ref("baseml", ctl, {"brown.nuc": EX + "/brown.nuc", "brown.trees": EX + "/brown.trees"}, x=x)
    finish("brown_hky85", res, "nuc", 5, dict(program="baseml", model=dict(kind="nuc_hky85", kappa=x[-1]), x=x, ntime=7,
                                               names=["Human", "Chimpanzee", "Gorilla", "Orangutan", "Gibbon"]))


def case_brown_anc():
    """Marginal ancestral reconstruction (RateAncestor = 1) at the same fixed parameters: the reference's rst file lists,
    for every site, the most probable state and its posterior probability at each internal node."""
    x = [float(v) for v in "0.053057 0.017471 0.041370 0.053761 0.057580 0.100159 0.138990 9.389630".split()]
    ctl = dict(BASEML_BASE, seqfile="brown.nuc", treefile="brown.trees", outfile="mlb", model=4, ncatG=1, RateAncestor=1)
    res = run_ref("baseml", ctl, {"brown.nuc": EX + "/brown.nuc", "brown.trees": EX + "/brown.trees"}, x=x)
    rows = {}
    blk = res["rst"].split("Prob of best state at each node, listed by site")[1].split("Summary of changes")[0]
    for ln in blk.splitlines():
        m = re.match(r"\s*(\d+)\s+(\d+)\s+([A-Z?-]+):\s+(.*)$", ln)
        if not m:
            continue
        best = re.findall(r"([A-Z])\(([0-9.]+)\)", m.group(4))
        rows[m.group(3)] = dict(count=int(m.group(2)), best="".join(b for b, _ in best), prob=[float(v) for _, v in best])
    nodes = [int(v) for v in re.findall(r"node #(\d+)", res["rst"])]
    g = dict(name="brown_hky85_anc", program="baseml", x=x, lnL=res["lnL"], nodes_1based=sorted(set(nodes)), patterns=rows,
             note="nodes in the order of the reference's table (node ns+1 = root first)")
    path = os.path.join(HERE, "brown_hky85_anc.json")
    with open(path, "w") as f:
        json.dump(g, f, separators=(",", ":"))
    print("%-22s lnL %.6f  %d distinct patterns, nodes %s -> %s" % (g["name"], g["lnL"], len(rows), g["nodes_1based"], os.path.basename(path)))


MTPRI_NUC = {"mtCDNApri.nuc": EX + "/mtCDNA/mtCDNApri.nuc", "mtCDNApri.trees": EX + "/mtCDNA/mtCDNApri.trees"}
MTPRI_AA = {"mtCDNApri.aa": EX + "/mtCDNA/mtCDNApri.aa", "mtCDNApri.trees": EX + "/mtCDNA/mtCDNApri.trees"}
MTAPE = {"mtCDNAape.txt": EX + "/mtCDNAape/mtCDNAape.txt", "mtCDNAape.trees": EX + "/mtCDNAape/mtCDNAape.trees"}


def case_mtcdna_branch():
    """Two-ratio branch model (model = 2, '#1' labels in the tree file) under the vertebrate mitochondrial code on the
    reference's mtCDNAape example, single evaluation at round parameter values."""
    x = [0.02, 0.03, 0.05, 0.04, 0.06, 0.08, 0.07, 0.09, 0.1, 20.0, 0.3, 0.04]
    ctl = dict(CODEML_BASE, seqfile="mtCDNAape.txt", treefile="mtCDNAape.trees", outfile="mlc", model=2, icode=1, cleandata=0,
               kappa=1.234567, omega=1.414)
    res = run_ref("codeml", ctl, {"mtCDNAape.txt": EX + "/mtCDNAape/mtCDNAape.txt", "mtCDNAape.trees": EX + "/mtCDNAape/mtCDNAape.trees"}, x=x)
    finish("mtcdna_branch", res, "codon", 6, dict(program="codeml", model=dict(kind="codon_branch", icode=1, codonfreq="F3x4"), x=x, ntime=9,
                                                 published=dict(m0_lnL=-20486.034301, branch_lnL=-20444.099676)), keep_raw_patterns=True)


def case_stewart():
    x = [float(v) for v in "0.000004 0.019085 0.083331 0.034683 0.067995 0.339072 0.104868 0.276662 0.861606 1.064411".split()]
    ctl = dict(CODEML_BASE, seqfile="stewart.aa", treefile="stewart.trees", outfile="mlc", seqtype=2, model=2,
               aaRatefile="lg.dat", fix_alpha=0, alpha=0.5, ncatG=4, cleandata=0)
    tree1 = " 6 1\n(((Langur, Baboon), Human), Rat, (Cow, Horse));\n"
    res = run_ref("codeml", ctl, {"stewart.aa": EX + "/stewart.aa", "stewart.trees": tree1, "lg.dat": DAT + "/lg.dat"}, x=x)
    finish("stewart_lg_g4", res, "aa", 6,
           dict(program="codeml", model=dict(kind="aa_empirical_gamma", ratefile="lg.dat", alpha=x[-1], ncatG=4), x=x, ntime=9))


def case_mhc():
    """192 taxa: exercises NodeScale (10 scaling nodes).  First the reference optimises kappa/omega with the published
    fix_blength=2 set-up (README lnL -8225.154790), then the golden is the single evaluation at the 6-decimal MLEs."""
    files = {"bigmhc.phy": EX + "/MHC.Swanson2002MBE/bigmhc.phy", "bigmhc.trees": EX + "/MHC.Swanson2002MBE/bigmhc.trees"}
    ctl = dict(CODEML_BASE, seqfile="bigmhc.phy", treefile="bigmhc.trees", outfile="mlc", NSsites=0, kappa=1.6, omega=.9,
               fix_blength=2, Small_Diff=".1e-6", method=0, cleandata=0)   # cleandata=0 as in the example: ambiguous codons kept
    res = run_ref("codeml", ctl, files)
    kap = float(re.search(r"kappa \(ts/tv\) =\s*([0-9.]+)", res["main"]).group(1))
    omg = float(re.search(r"omega \(dN/dS\) =\s*([0-9.]+)", res["main"]).group(1))
    xs = re.search(r"lnL\(ntime:[^\n]*\n([^\n]+)\n", res["main"])
    x = [float(v) for v in xs.group(1).split()]
    print("   MHC MLE lnL %.6f  x = %s (printed kappa %.5f omega %.5f)" % (res["lnL"], x, kap, omg))
    ctl1 = dict(ctl, fix_kappa=1, kappa="%.6f" % x[0], fix_omega=1, omega="%.6f" % x[1])
    res1 = run_ref("codeml", ctl1, files)
    scal = re.search(r"(\d+) node\(s\) used for scaling.*?\n([ 0-9]+)\n", res1["stdout"])
    finish("mhc_m0_scaled", res1, "codon", 192,
           dict(program="codeml", model=dict(kind="codon_m0", kappa=x[0], omega=x[1], codonfreq="F3x4"),
                published_lnL=-8225.154790, mle_lnL=res["lnL"],
                scale_nodes=[int(v) for v in scal.group(2).split()] if scal else None))



sys.path.insert(0, os.path.dirname(HERE))
from helpers import ymd_names  # noqa: E402  (the tests rebuild the same files)


def case_mle(name, ctl_over, files, n_tips, kind, x0=None, prog="codeml", seqtype="codon"):
    """Branch-site / clade / discrete models: the reference first maximises the likelihood from its own initial values (or x0),
    then the golden is the single evaluation at the printed 6-decimal estimates (the -1 recipe), whose lnL must agree with
    the maximised value to ~1e-5."""
    ctl = dict(CODEML_BASE, outfile="mlc", **ctl_over) if prog == "codeml" else dict(BASEML_BASE, outfile="mlb", **ctl_over)
    res = run_ref(prog, ctl, files)
    xs = re.search(r"lnL\(ntime:\s*(\d+)[^\n]*\n(?:[^\n]*\.\.[^\n]*\n)?([^\n]+)\n", res["main"])      # (no branch header line under fix_blength = 3)
    ntime = int(xs.group(1))
    x = [float(v) for v in xs.group(2).split()]
    print("   %s: reference MLE lnL %.6f, np %d" % (name, res["lnL"], len(x)))
    res1 = run_ref(prog, ctl, files, x=x)
    tables = {}
    for key, head in (("neb_post", "Naive Empirical Bayes (NEB) probabilities for"), ("beb_post", "Bayes Empirical Bayes (BEB) probabilities for")):
        if head in res1["rst"]:      # per-site class posteriors as the reference writes them to `rst` (5 decimals)
            blk = res1["rst"][res1["rst"].index(head):]
            rows = re.findall(r"^\s*\d+ \S\s+((?:[01]\.\d{5}\s+)+)\(\s*\d+\)", blk, re.M)
            ls = int([ln for ln in res1["lnf"] if ln.split()][0].split()[1])
            tables[key] = [[float(v) for v in r.split()] for r in rows[:ls]]
    if "dN & dS for each branch" in res1["main"]:      # branch, t, N, S, dN/dS, dN, dS (+ N*dN, S*dS) as printed in the main result file
        blk = res1["main"][res1["main"].index("dN & dS for each branch"):]
        rows = re.findall(r"^\s*(\d+)\.\.(\d+)\s+([-0-9.]+)\s+([-0-9.]+)\s+([-0-9.]+)\s+([-0-9.]+)\s+([-0-9.]+)\s+([-0-9.]+)", blk, re.M)
        tables["dnds"] = [[int(r[0]), int(r[1])] + [float(v) for v in r[2:]] for r in rows]
    finish(name, res1, seqtype, n_tips, dict(tables, program=prog, model=dict(kind=kind, **{k: ctl_over[k] for k in ("model", "NSsites", "fix_omega", "omega", "ncatG", "Mgene", "alpha", "nhomo", "fix_kappa", "Malpha", "clock", "TipDate", "aaDist", "CodonFreq", "estFreq", "icode") if k in ctl_over}),
                                             x=x, ntime=ntime, mle_lnL=res["lnL"]), keep_raw_patterns=True)


def case_at(name, ctl_over, files, n_tips, kind, x, ntime):
    """Single evaluation at chosen parameter values (the -1 recipe): for models whose maximum sits at degenerate values where the
    reference's own discretisation (a line search on (CDF - p)^2) is only accurate to ~1e-5."""
    ctl = dict(CODEML_BASE, outfile="mlc", **ctl_over)
    res1 = run_ref("codeml", ctl, files, x=x)
    g = finish(name, res1, "codon", n_tips, dict(program="codeml", model=dict(kind=kind, **{k: ctl_over[k] for k in ("NSsites", "ncatG") if k in ctl_over}),
                                                 x=x, ntime=ntime), keep_raw_patterns=True)
    # The reference evaluates the likelihood twice (once for the printed lnL, once for the lnf file), and its discretisation of the
    # omega distribution starts each line search from the previous call's classes: under M11 the two calls differ by ~3e-3.  The
    # per-pattern values are the ones a second program can be compared with; lnL is their weighted sum.
    s_lnf = float(np.dot(g["counts"], g["logf"]))
    if abs(s_lnf - g["lnL"]) > 1e-5:
        g["lnL_printed"] = g["lnL"]
        g["lnL"] = round(s_lnf, 6)
        g["note"] = "lnL = sum of counts x log f_h of the lnf file; the reference's printed lnL (lnL_printed) comes from a separate call with a different discretisation"
        with open(os.path.join(HERE, name + ".json"), "w") as f:
            json.dump(g, f, separators=(",", ":"))
        print("   %s: printed lnL %.6f, lnf-file sum %.6f" % (name, g["lnL_printed"], g["lnL"]))



# ---- the north_star workload: the NSsites sweep on the C4 data (16 taxa x 10^6 codon patterns) -----------------------------
# Class tables at fixed values: M1a (K = 2), M2a (K = 3), M7 (K = 10), M8 (K = 11).  The reference writes its `lnf`
# file only AFTER the NEB / BEB post-processing (codeml.c:895-904), and BEB for M2a / M8 (lfunNSsites_M2M8, 10^4 grid points x
# npatt) would take hours at 10^6 patterns.  So for M2a / M8 proper the reference is stopped once it has printed the lnL of its
# single evaluation (golden = lnL only), and the SAME class tables are run to completion through NSsites = 3 (M3 "discrete" with
# K = 3 / 11 classes given as untransformed p's and w's, no BEB) for lnL + the per-pattern log f_h sample.
NS_FULL = {
    # name: (NSsites, ncatG, x after the fixed kappa, stop once lnL is printed)
    "m1a": (1, 2, [0.7, 0.1], False),
    "m2a": (2, 3, [0.6, 0.3, 0.1, 2.5], True),
    "m7": (7, 10, [0.5, 1.2], False),
    "m8": (8, 10, [0.9, 0.5, 1.2, 2.5], True),
}


def _beta_medians(p, q, K):
    from scipy.stats import beta
    return [round(float(beta.ppf((i + 0.5) / K, p, q)), 6) for i in range(K)]


def run_ref_until_lnl(prog, ctl, files, x, timeout=7200):
    """Start the reference, return the lnL it prints after its single evaluation ("lnL  = ..."), then stop it."""
    import time
    d = tempfile.mkdtemp(prefix="golden_")
    try:
        for dst, src in files.items():
            if os.path.exists(str(src)):
                os.symlink(src, os.path.join(d, dst))
            else:
                with open(os.path.join(d, dst), "w") as f:
                    f.write(src)
        with open(os.path.join(d, prog + ".ctl"), "w") as f:
            for k, v in ctl.items():
                f.write("%s = %s\n" % (k, v))
        with open(os.path.join(d, "in." + prog), "w") as f:
            f.write("-1 " + " ".join("%.6f" % v for v in x) + "\n")
        log = open(os.path.join(d, "stdout.txt"), "wb")
        pr = subprocess.Popen([os.path.join(REF, prog), prog + ".ctl"], cwd=d, stdout=log, stderr=subprocess.STDOUT, stdin=subprocess.DEVNULL)
        t0 = time.time()
        try:
            while True:
                time.sleep(5)
                out = open(os.path.join(d, "stdout.txt"), errors="replace").read()
                m = re.findall(r"^lnL\s+=\s*(-?[0-9.]+)", out, re.M)
                if m:
                    return dict(lnL=float(m[-1]), stdout=out)
                if pr.poll() is not None:
                    raise RuntimeError("reference ended without lnL:\n" + out[-2000:])
                if time.time() - t0 > timeout:
                    raise RuntimeError("timeout")
        finally:
            if pr.poll() is None:
                pr.kill()
            pr.wait()
    finally:
        shutil.rmtree(d, ignore_errors=True)


def case_syn_codon_ns(tag, n_patt=1_000_000, sample=997):
    pb = synth.codon_m0_problem(n_tips=16, n_patt=n_patt, estimate_pi=True)
    d = tempfile.mkdtemp()
    seq = os.path.join(d, "seq.txt")
    synth.write_pattern_file(seq, pb.z, pb.weights, "codon")
    tree = " 16 1\n" + pb.tree.newick() + "\n"
    suffix = "_full" if n_patt == 1_000_000 else "_%d" % n_patt
    gen = dict(fn="codon_m0_problem", n_tips=16, n_patt=n_patt, seed=20260926, estimate_pi=True)
    try:
        if tag in NS_FULL:
            ns, ncat, x, stop = NS_FULL[tag]
            ctl = dict(CODEML_BASE, seqfile="seq.txt", treefile="tree.txt", outfile="mlc", fix_kappa=1, kappa=2, NSsites=ns, ncatG=ncat,
                       omega=1.3, fix_blength=2)
            model = dict(kind="codon_nssites", NSsites=ns, ncatG=ncat, kappa=2.0, codonfreq="F3x4")
            if stop:
                res = run_ref_until_lnl("codeml", ctl, {"seq.txt": seq, "tree.txt": tree}, x)
                g = dict(name="syn_codon_%s%s" % (tag, suffix), seqtype="codon", n_tips=16, n_patt=n_patt, lnL=res["lnL"], program="codeml",
                         model=model, x=x, ntime=0, generator=gen, tree=pb.tree.newick(),
                         note="lnL of the reference's single evaluation; the program was stopped before its BEB post-processing, so no lnf sample "
                              "(see syn_codon_%s_as_m3%s for the same class table run to completion through NSsites = 3)" % (tag, suffix))
                path = os.path.join(HERE, g["name"] + ".json")
                with open(path, "w") as f:
                    json.dump(g, f, separators=(",", ":"))
                print("%-22s lnL %.6f (lnL only) -> %s" % (g["name"], g["lnL"], os.path.basename(path)))
            else:
                res = run_ref("codeml", ctl, {"seq.txt": seq, "tree.txt": tree}, x=x, timeout=4 * 3600)
                finish("syn_codon_%s%s" % (tag, suffix), res, "codon", 16, dict(program="codeml", model=model, x=x, ntime=0, generator=gen), sample=sample)
        else:      # "m2a_as_m3" / "m8_as_m3"
            src = tag.split("_")[0]
            ns, ncat, xs, _ = NS_FULL[src]
            if src == "m2a":
                p, w = [xs[0], xs[1]], [xs[2], 1.0, xs[3]]
            else:
                p, w = [round(xs[0] / 10, 6)] * 10, _beta_medians(xs[1], xs[2], 10) + [xs[3]]
            K = len(w)
            x = p + w
            ctl = dict(CODEML_BASE, seqfile="seq.txt", treefile="tree.txt", outfile="mlc", fix_kappa=1, kappa=2, NSsites=3, ncatG=K, omega=1.3,
                       fix_blength=2)
            res = run_ref("codeml", ctl, {"seq.txt": seq, "tree.txt": tree}, x=x, timeout=4 * 3600)
            finish("syn_codon_%s%s" % (tag, suffix), res, "codon", 16,
                   dict(program="codeml", model=dict(kind="codon_nssites", NSsites=3, ncatG=K, kappa=2.0, codonfreq="F3x4"), x=x, ntime=0,
                        generator=gen, classes=dict(p=p + [round(1 - sum(p), 6)], w=w)), sample=sample)
    finally:
        shutil.rmtree(d, ignore_errors=True)


def case_mcmctree():
    """The mcmctree exact-likelihood consumer (usedata = 1; lnpD_locus mcmctree.c:1130-1166 -> com.plfun(NULL, -1)): a short chain
    of the reference on the first locus of examples/DatingSoftBound (7 primates, 1st codon positions), global clock, JC69, no
    gamma — so the sampled lnL is a function of the sampled node ages and rate alone.  mcmc.txt holds, per sample, the ages
    t_n8 .. t_n13 (7 decimals), the rate mu and lnL (3 decimals): the golden."""
    d = tempfile.mkdtemp(prefix="golden_")
    try:
        lines = open(EX + "/DatingSoftBound/mtCDNApri123.txt").read().splitlines()
        first = [ln for ln in lines if ln.strip()][:8]          # header + 7 sequences = the first locus
        with open(os.path.join(d, "locus1.txt"), "w") as f:
            f.write("\n".join(first) + "\n")
        shutil.copy(os.path.join(d, "locus1.txt"), os.path.join(HERE, "data", "mtCDNApri_locus1.txt"))
        tree = "((((human, (chimpanzee, bonobo)) '>.06<.08', gorilla), (orangutan, sumatran)) '>.12<.16', gibbon);"
        with open(os.path.join(d, "tree.txt"), "w") as f:
            f.write(" 7 1\n" + tree + "\n")
        ctl = dict(seed=12345, seqfile="locus1.txt", treefile="tree.txt", mcmcfile="mcmc.txt", outfile="out.txt", ndata=1, seqtype=0, usedata=1,
                   clock=1, RootAge="'<1.0'", model=0, alpha=0, ncatG=5, cleandata=0, BDparas="1 1 0.1 multiplicative", kappa_gamma="6 2",
                   alpha_gamma="1 1", rgene_gamma="2 20 1", sigma2_gamma="1 10 1", finetune="1: .1 .1 .1 .1 .1 .1", print=1, burnin=200,
                   sampfreq=5, nsample=30)
        with open(os.path.join(d, "mcmctree.ctl"), "w") as f:
            for k, v in ctl.items():
                f.write("%s = %s\n" % (k, v))
        subprocess.run([os.path.join(REF, "mcmctree"), "mcmctree.ctl"], cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.STDOUT, timeout=600, check=True)
        rows = [ln.split() for ln in open(os.path.join(d, "mcmc.txt")).read().splitlines()]
        hdr, rows = rows[0], rows[1:]
        assert hdr[1:7] == ["t_n8", "t_n9", "t_n10", "t_n11", "t_n12", "t_n13"] and hdr[7] == "mu" and hdr[8] == "lnL", hdr
        g = dict(name="mcmctree_jc_clock1", program="mcmctree", tree="((((human,(chimpanzee,bonobo)),gorilla),(orangutan,sumatran)),gibbon);",
                 names=["human", "chimpanzee", "bonobo", "gorilla", "orangutan", "sumatran", "gibbon"], seqfile="mtCDNApri_locus1.txt",
                 node_labels=hdr[1:7], samples=[dict(age=[float(v) for v in r[1:7]], mu=float(r[7]), lnL=float(r[8])) for r in rows],
                 note="node t_n<k> = node k (1-based) of the reference's tree numbering: 8 = root, then the internal nodes in the order of their "
                      "opening brackets; branch length of node i = (age of its father - age of i) x mu (lnpD_locus, global clock)")
        with open(os.path.join(HERE, "mcmctree_jc_clock1.json"), "w") as f:
            json.dump(g, f, separators=(",", ":"))
        print("mcmctree_jc_clock1     %d samples, lnL %.3f .. %.3f" % (len(rows), min(s["lnL"] for s in g["samples"]), max(s["lnL"] for s in g["samples"])))
    finally:
        shutil.rmtree(d, ignore_errors=True)


LYSO = {"lysozymeLarge.nuc": EX + "/lysozyme/lysozymeLarge.nuc", "lysozymeLarge.trees": EX + "/lysozyme/lysozymeLarge.trees"}
LYSO_CTL = dict(seqfile="lysozymeLarge.nuc", treefile="lysozymeLarge.trees", kappa=3, cleandata=0)
ECP = {"ECP_EDN_15.nuc": EX + "/CladeModelCD/ECP_EDN_15.nuc", "tree.txt": EX + "/CladeModelCD/tree.txt"}
ECP_CTL = dict(seqfile="ECP_EDN_15.nuc", treefile="tree.txt", kappa=2.5, omega=0.13579, ncatG=3, cleandata=0, Small_Diff=".2e-6")
HIVF = {"HIVenvSweden.txt": EX + "/HIVNSsites/HIVenvSweden.txt", "HIVenvSweden.trees": EX + "/HIVNSsites/HIVenvSweden.trees"}

HORAI = {"horai.nuc": EX + "/horai.nuc", "horai.trees": EX + "/horai.trees"}
LYSIN = {"lysinYangSwanson2002.nuc": EX + "/lysin/lysinYangSwanson2002.nuc", "lysin.trees": EX + "/lysin/lysin.trees"}


def case_horai(mgene, alpha=0):
    """Option G (4 genes: the three codon positions and a tRNA part) in baseml, HKY85, Mgene = 0 / 2 / 3 / 4."""
    over = dict(seqfile="horai.nuc", treefile="horai.trees", model=4, Mgene=mgene, kappa=5)
    if alpha:
        over.update(fix_alpha=0, alpha=alpha, ncatG=5)
    case_mle("horai_mg%d%s" % (mgene, "_g5" if alpha else ""), over, HORAI, 6, "nuc_genes", prog="baseml", seqtype="nuc")


def case_lysin(mgene):
    """Option G in codeml (two site partitions of the sperm lysin, Yang & Swanson 2002), M0, Mgene = 0 / 2 / 3 / 4."""
    over = dict(seqfile="lysinYangSwanson2002.nuc", treefile="lysin.trees", Mgene=mgene, kappa=1.6, omega=.8, cleandata=0, Small_Diff="3e-7")
    case_mle("lysin_mg%d" % mgene, over, LYSIN, 25, "codon_genes")


def case_brown_clock():
    """Global clock (clock = 1) on the first rooted tree of examples/brown.rooted.trees, HKY85: x holds the four node ages."""
    tree1 = "  5  1\n\n((((1,2),3),4),5);\n"
    case_mle("brown_hky85_clock", dict(seqfile="brown.nuc", treefile="brown.rooted.trees", model=4, clock=1, kappa=5),
             {"brown.nuc": EX + "/brown.nuc", "brown.rooted.trees": tree1}, 5, "nuc_clock", prog="baseml", seqtype="nuc")


BROWN = {"brown.nuc": EX + "/brown.nuc", "brown.trees": EX + "/brown.trees"}


def case_brown_rates():
    """Posterior mean rates per site under HKY85 + G4 (RateAncestor = 1: the reference's `rates` file) at its own estimates."""
    d = tempfile.mkdtemp(prefix="golden_")
    try:
        for f in ("brown.nuc", "brown.trees"):
            shutil.copy(EX + "/" + f, os.path.join(d, f))
        ctl = dict(BASEML_BASE, outfile="mlb", seqfile="brown.nuc", treefile="brown.trees", model=4, kappa=5, fix_alpha=0, alpha=0.5, ncatG=4, RateAncestor=1, noisy=0)
        with open(os.path.join(d, "baseml.ctl"), "w") as f:
            for k, v in ctl.items():
                f.write("%s = %s\n" % (k, v))
        subprocess.run([os.path.join(REF, "baseml"), "baseml.ctl"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, input=b"\n" * 50, timeout=600)
        main = open(os.path.join(d, "mlb")).read()
        xs = re.search(r"lnL\(ntime:\s*(\d+)[^\n]*:\s*(-[0-9.]+)[^\n]*\n[^\n]*\n([^\n]+)\n", main)
        rows = re.findall(r"^\s*(\d+)\s+\d+\s+[A-Z\-\?]+\s+([0-9.]+)\s+(\d+)\s*$", open(os.path.join(d, "rates")).read(), re.M)
        g = dict(name="brown_hky85_g4_rates", program="baseml", lnL=float(xs.group(2)), ntime=int(xs.group(1)), x=[float(v) for v in xs.group(3).split()],
                 rate_mean=[float(r[1]) for r in rows], rate_class=[int(r[2]) for r in rows])
        assert len(rows) == 895
        with open(os.path.join(HERE, "brown_hky85_g4_rates.json"), "w") as f:
            json.dump(g, f, separators=(",", ":"))
        print("brown_hky85_g4_rates   lnL %.6f  %d sites" % (g["lnL"], len(rows)))
    finally:
        shutil.rmtree(d, ignore_errors=True)


def case_brown_joint():
    """"(2) Joint reconstruction of ancestral sequences" of the reference's rst for brown.nuc under HKY85 (no rate classes): per
    pattern the best reconstruction of nodes 6-8 and its probability, at the reference's own estimates."""
    d = tempfile.mkdtemp(prefix="golden_")
    try:
        for f in ("brown.nuc", "brown.trees"):
            shutil.copy(EX + "/" + f, os.path.join(d, f))
        ctl = dict(BASEML_BASE, outfile="mlb", seqfile="brown.nuc", treefile="brown.trees", model=4, kappa=5, RateAncestor=1, noisy=0)
        with open(os.path.join(d, "baseml.ctl"), "w") as f:
            for k, v in ctl.items():
                f.write("%s = %s\n" % (k, v))
        subprocess.run([os.path.join(REF, "baseml"), "baseml.ctl"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, input=b"\n" * 50, timeout=600)
        main = open(os.path.join(d, "mlb")).read()
        xs = re.search(r"lnL\(ntime:\s*(\d+)[^\n]*:\s*(-[0-9.]+)[^\n]*\n[^\n]*\n([^\n]+)\n", main)
        rst = open(os.path.join(d, "rst")).read()
        blk = rst[rst.index("(2) Joint reconstruction"):]
        rows = re.findall(r"^\s*\d+\s+(\d+)\s+([TCAG]{5}): ([TCAG]{3}) \(([0-9.]+)\)", blk, re.M)
        g = dict(name="brown_hky85_joint", program="baseml", lnL=float(xs.group(2)), ntime=int(xs.group(1)), x=[float(v) for v in xs.group(3).split()],
                 patterns={r[1]: dict(count=int(r[0]), best=r[2], prob=float(r[3])) for r in rows})
        assert len(rows) == 85
        with open(os.path.join(HERE, "brown_hky85_joint.json"), "w") as f:
            json.dump(g, f, separators=(",", ":"))
        print("brown_hky85_joint      lnL %.6f  %d patterns" % (g["lnL"], len(rows)))
    finally:
        shutil.rmtree(d, ignore_errors=True)


def case_brown_adg():
    """Auto-discrete-gamma (lfunAdG: alpha and rho free, 4 rate classes) on brown.nuc, HKY85.  Sites are not independent under this
    model, so the reference writes no per-pattern values: the golden is lnL at the printed estimates (and the maximised value)."""
    ctl = dict(BASEML_BASE, outfile="mlb", seqfile="brown.nuc", treefile="brown.trees", model=4, kappa=5, fix_alpha=0, alpha=0.5, ncatG=4,
               fix_rho=0, rho=0.3)
    res = run_ref("baseml", ctl, BROWN)
    xs = re.search(r"lnL\(ntime:\s*(\d+)[^\n]*\n[^\n]*\n([^\n]+)\n", res["main"])
    x = [float(v) for v in xs.group(2).split()]
    res1 = run_ref("baseml", ctl, BROWN, x=x)
    hdr = [int(v) for v in [ln for ln in res1["lnf"] if ln.split()][0].split()]
    g = dict(name="brown_hky85_adg", seqtype="nuc", program="baseml", n_tips=5, ls=hdr[1], n_patt=hdr[2], lnL=res1["lnL"], mle_lnL=res["lnL"], x=x,
             ntime=int(xs.group(1)), model=dict(kind="nuc_adg", ncatG=4))
    with open(os.path.join(HERE, "brown_hky85_adg.json"), "w") as f:
        json.dump(g, f, separators=(",", ":"))
    print("brown_hky85_adg        lnL %.6f (maximised %.6f)  x = %s" % (g["lnL"], g["mle_lnL"], x))

CASES = {
    "brown_f84": lambda: case_mle("brown_f84", dict(seqfile="brown.nuc", treefile="brown.trees", model=3, kappa=5), BROWN, 5, "nuc", prog="baseml", seqtype="nuc"),
    "brown_t92_g4": lambda: case_mle("brown_t92_g4", dict(seqfile="brown.nuc", treefile="brown.trees", model=5, kappa=5, fix_alpha=0, alpha=0.5, ncatG=4), BROWN, 5, "nuc", prog="baseml", seqtype="nuc"),
    "brown_unrest": lambda: case_mle("brown_unrest", dict(seqfile="brown.nuc", treefile="brown.trees", model=8), BROWN, 5, "nuc", prog="baseml", seqtype="nuc"),
    "stewart_eqinput": lambda: case_mle("stewart_eqinput", dict(seqfile="stewart.aa", treefile="stewart.trees", seqtype=2, model=1, cleandata=0),
                                        {"stewart.aa": EX + "/stewart.aa", "stewart.trees": " 6 1\n(((Langur, Baboon), Human), Rat, (Cow, Horse));\n"}, 6, "aa", seqtype="aa"),
    "brown_hky85_adg": lambda: case_brown_adg(),
    "brown_hky85_g4_rates": case_brown_rates,
    "brown_hky85_joint": case_brown_joint,
    "brown_hky85_nhomo1": lambda: case_mle("brown_hky85_nhomo1", dict(seqfile="brown.nuc", treefile="brown.trees", model=4, kappa=5, nhomo=1), BROWN, 5, "nuc", prog="baseml", seqtype="nuc"),
    # non-homogeneous models (Yang & Roberts 1995): a kappa per branch (2); frequency sets per tip branch / internal branches / root
    # with a kappa per branch (3, N1); a frequency set at every node with one kappa (4, N2 with fix_kappa = 1)
    "brown_hky85_nhomo2": lambda: case_mle("brown_hky85_nhomo2", dict(seqfile="brown.nuc", treefile="brown.trees", model=4, kappa=5, nhomo=2), BROWN, 5, "nuc", prog="baseml", seqtype="nuc"),
    "brown_hky85_nhomo3": lambda: case_mle("brown_hky85_nhomo3", dict(seqfile="brown.nuc", treefile="brown.trees", model=4, kappa=5, nhomo=3), BROWN, 5, "nuc", prog="baseml", seqtype="nuc"),
    "brown_f84_nhomo4": lambda: case_mle("brown_f84_nhomo4", dict(seqfile="brown.nuc", treefile="brown.trees", model=3, kappa=5, nhomo=4, fix_kappa=1), BROWN, 5, "nuc", prog="baseml", seqtype="nuc"),
    "brown_t92_nhomo3_g4": lambda: case_mle("brown_t92_nhomo3_g4", dict(seqfile="brown.nuc", treefile="brown.trees", model=5, kappa=5, nhomo=3, fix_kappa=1, fix_alpha=0, alpha=0.5, ncatG=4), BROWN, 5, "nuc", prog="baseml", seqtype="nuc"),
    # Malpha: a gamma shape per gene (class rates per gene), with shared kappa (Mgene 0) and with everything per gene (Mgene 4)
    "horai_mg0_malpha": lambda: case_mle("horai_mg0_malpha", dict(seqfile="horai.nuc", treefile="horai.trees", model=4, kappa=5, Mgene=0, fix_alpha=0, alpha=0.5, Malpha=1, ncatG=5), HORAI, 6, "nuc_genes", prog="baseml", seqtype="nuc"),
    "horai_mg4_malpha": lambda: case_mle("horai_mg4_malpha", dict(seqfile="horai.nuc", treefile="horai.trees", model=4, kappa=5, Mgene=4, fix_alpha=0, alpha=0.5, Malpha=1, ncatG=4), HORAI, 6, "nuc_genes", prog="baseml", seqtype="nuc"),
    # local clocks (clock = 2): the '#' labels of a rooted tree are rate classes, the rates of classes 1, 2 follow the node ages in x
    "brown_hky85_clock2": lambda: case_mle("brown_hky85_clock2", dict(seqfile="brown.nuc", treefile="brown.clock2.trees", model=4, clock=2, kappa=5),
                                           {"brown.nuc": EX + "/brown.nuc", "brown.clock2.trees": "  5  1\n\n((((1,2) #1,3),4 #2),5);\n"}, 5, "nuc_clock", prog="baseml", seqtype="nuc"),
    # TipDate (Stadler & Yang 2012, examples/TipDate.HIV2): 33 dated HIV-2 / SIV sequences, global clock, HKY85 + G5; x = 32 node ages in
    # units of 100 years before the youngest sample, the mutation rate per unit, kappa, alpha
    "hiv2_tipdate": lambda: case_mle("hiv2_tipdate", dict(seqfile="HIV2ge.txt", treefile="HIV2ge.tree1", model=4, clock=1, TipDate="1 100", kappa=2, fix_alpha=0, alpha=0.5, ncatG=5, cleandata=0),
                                     {"HIV2ge.txt": EX + "/TipDate.HIV2/HIV2ge.txt", "HIV2ge.tree1": os.path.join(HERE, "data", "HIV2ge.tree1")}, 33, "nuc_tipdate", prog="baseml", seqtype="nuc"),
    # omega as a function of an amino-acid distance (Yang, Nielsen & Hasegawa 1998, table 4): geometric on Grantham's distances, linear on Miyata's
    "mtcdnapri_aadist1": lambda: case_mle("mtcdnapri_aadist1", dict(seqfile="mtCDNApri.nuc", treefile="mtCDNApri.trees", model=0, NSsites=0, icode=1, CodonFreq=2, aaDist=1, kappa=3, omega=.4),
                                          dict(MTPRI_NUC, **{"grantham.dat": DAT + "/grantham.dat"}), 7, "codon_aadist"),
    "mtcdnapri_aadist_m2": lambda: case_mle("mtcdnapri_aadist_m2", dict(seqfile="mtCDNApri.nuc", treefile="mtCDNApri.trees", model=0, NSsites=0, icode=1, CodonFreq=2, aaDist=-2, kappa=3, omega=.4),
                                            dict(MTPRI_NUC, **{"miyata.dat": DAT + "/miyata.dat"}), 7, "codon_aadist"),
    # codon frequencies as parameters (estFreq = 1) and the mutation-selection models FMutSel0 / FMutSel (CodonFreq 6, 7; Yang & Nielsen 2008)
    "hiv_fmutsel0": lambda: case_mle("hiv_fmutsel0", dict(seqfile="HIVenvSweden.txt", treefile="HIVenvSweden.trees", NSsites=0, CodonFreq=6, estFreq=0, kappa=.3, omega=1.3), HIVF, 13, "codon_m0"),
    "hiv_fmutsel0_est": lambda: case_mle("hiv_fmutsel0_est", dict(seqfile="HIVenvSweden.txt", treefile="HIVenvSweden.trees", NSsites=0, CodonFreq=6, estFreq=1, kappa=.3, omega=1.3), HIVF, 13, "codon_m0"),
    "hiv_fmutsel": lambda: case_mle("hiv_fmutsel", dict(seqfile="HIVenvSweden.txt", treefile="HIVenvSweden.trees", NSsites=0, CodonFreq=7, estFreq=0, kappa=.3, omega=1.3), HIVF, 13, "codon_m0"),
    "hiv_fmutsel_est": lambda: case_mle("hiv_fmutsel_est", dict(seqfile="HIVenvSweden.txt", treefile="HIVenvSweden.trees", NSsites=0, CodonFreq=7, estFreq=1, kappa=.3, omega=1.3), HIVF, 13, "codon_m0"),
    "hiv_f3x4_est": lambda: case_mle("hiv_f3x4_est", dict(seqfile="HIVenvSweden.txt", treefile="HIVenvSweden.trees", NSsites=0, CodonFreq=2, estFreq=1, kappa=.3, omega=1.3), HIVF, 13, "codon_m0"),
    "hiv_f1x4mg_est": lambda: case_mle("hiv_f1x4mg_est", dict(seqfile="HIVenvSweden.txt", treefile="HIVenvSweden.trees", NSsites=0, CodonFreq=4, estFreq=1, kappa=.3, omega=1.3), HIVF, 13, "codon_m0"),
    "hiv_fmutsel0_m2a": lambda: case_mle("hiv_fmutsel0_m2a", dict(seqfile="HIVenvSweden.txt", treefile="HIVenvSweden.trees", NSsites=2, CodonFreq=6, estFreq=0, kappa=.3, omega=1.3), HIVF, 13, "codon_nssites"),
    "hiv_f3x4_est_m7": lambda: case_mle("hiv_f3x4_est_m7", dict(seqfile="HIVenvSweden.txt", treefile="HIVenvSweden.trees", NSsites=7, ncatG=10, CodonFreq=2, estFreq=1, kappa=.3, omega=1.3), HIVF, 13, "codon_nssites"),
    # nhomo = 5: frequency sets by the tree's '#' labels (two branch types, the root a third set), with a kappa per label (fix_kappa = 2)
    "brown_hky85_nhomo5": lambda: case_mle("brown_hky85_nhomo5", dict(seqfile="brown.nuc", treefile="brown.nhomo5.trees", model=4, kappa=5, nhomo=5, fix_kappa=2),
                                           {"brown.nuc": EX + "/brown.nuc", "brown.nhomo5.trees": "  5  1\n\n((1,2) #1, 3 #1, (4,5)) #2;\n"}, 5, "nuc", prog="baseml", seqtype="nuc"),
    # general reversible amino-acid models: REVaa_0 (exchangeabilities of the amino-acid pairs one nucleotide change apart: 74 under the
    # universal code) and REVaa (all 189), initial values from jones.dat
    "mtcdnapri_revaa0": lambda: case_mle("mtcdnapri_revaa0", dict(seqfile="mtCDNApri.aa", treefile="mtCDNApri.trees", seqtype=2, model=8, aaRatefile="jones.dat", icode=1, cleandata=1),
                                         dict(MTPRI_AA, **{"jones.dat": DAT + "/jones.dat"}), 7, "aa_revaa", seqtype="aa"),
    "mtcdnapri_revaa": lambda: case_mle("mtcdnapri_revaa", dict(seqfile="mtCDNApri.aa", treefile="mtCDNApri.trees", seqtype=2, model=9, aaRatefile="jones.dat", icode=1, cleandata=1),
                                        dict(MTPRI_AA, **{"jones.dat": DAT + "/jones.dat"}), 7, "aa_revaa", seqtype="aa"),
    # TipDate with local clocks: a second rate for the (Lib1sm, FO784h, SIVMNE) clade
    "hiv2_tipdate_clock2": lambda: case_mle("hiv2_tipdate_clock2", dict(seqfile="HIV2ge.txt", treefile="HIV2ge.clock2.tree", model=4, clock=2, TipDate="1 100", kappa=2, fix_alpha=0, alpha=0.5, ncatG=5, cleandata=0),
                                            {"HIV2ge.txt": EX + "/TipDate.HIV2/HIV2ge.txt", "HIV2ge.clock2.tree": os.path.join(HERE, "data", "HIV2ge.clock2.tree")}, 33, "nuc_tipdate", prog="baseml", seqtype="nuc"),
    "mhc_m0_prop": lambda: case_mle("mhc_m0_prop", dict(seqfile="bigmhc.phy", treefile="bigmhc.trees", NSsites=0, kappa=1.6, omega=.9, fix_blength=3, cleandata=0, Small_Diff=".1e-6"),
                                    {"bigmhc.phy": EX + "/MHC.Swanson2002MBE/bigmhc.phy", "bigmhc.trees": EX + "/MHC.Swanson2002MBE/bigmhc.trees"}, 192, "codon_m0"),
    # the published MHC site-model runs (examples/MHC.Swanson2002MBE/README.txt:25-30: M1a -7490.993363, M2a -7231.154540, M7 -7502.792534,
    # M8 -7238.014961; 192 taxa, branch lengths fixed at the tree file's, ten scaling nodes) with their NEB / BEB tables
    "mhc_m1a": lambda: case_mle("mhc_m1a", dict(seqfile="bigmhc.phy", treefile="bigmhc.trees", NSsites=1, ncatG=2, kappa=1.6, omega=.9, fix_blength=2, cleandata=0, Small_Diff=".1e-6"),
                                 {"bigmhc.phy": EX + "/MHC.Swanson2002MBE/bigmhc.phy", "bigmhc.trees": EX + "/MHC.Swanson2002MBE/bigmhc.trees"}, 192, "codon_nssites"),
    "mhc_m2a": lambda: case_mle("mhc_m2a", dict(seqfile="bigmhc.phy", treefile="bigmhc.trees", NSsites=2, ncatG=3, kappa=1.6, omega=.9, fix_blength=2, cleandata=0, Small_Diff=".1e-6"),
                                 {"bigmhc.phy": EX + "/MHC.Swanson2002MBE/bigmhc.phy", "bigmhc.trees": EX + "/MHC.Swanson2002MBE/bigmhc.trees"}, 192, "codon_nssites"),
    "mhc_m7": lambda: case_mle("mhc_m7", dict(seqfile="bigmhc.phy", treefile="bigmhc.trees", NSsites=7, ncatG=10, kappa=1.6, omega=.9, fix_blength=2, cleandata=0, Small_Diff=".1e-6"),
                                 {"bigmhc.phy": EX + "/MHC.Swanson2002MBE/bigmhc.phy", "bigmhc.trees": EX + "/MHC.Swanson2002MBE/bigmhc.trees"}, 192, "codon_nssites"),
    "mhc_m8": lambda: case_mle("mhc_m8", dict(seqfile="bigmhc.phy", treefile="bigmhc.trees", NSsites=8, ncatG=10, kappa=1.6, omega=.9, fix_blength=2, cleandata=0, Small_Diff=".1e-6"),
                                 {"bigmhc.phy": EX + "/MHC.Swanson2002MBE/bigmhc.phy", "bigmhc.trees": EX + "/MHC.Swanson2002MBE/bigmhc.trees"}, 192, "codon_nssites"),
    # the other genetic codes: invertebrate mt (icode 4, 62 sense codons), ciliate nuclear (5, 63) and the "regularised" code (11, 64) on the
    # HIV data (no TAA / TAG / TGA in it, so the alignment is legal under each)
    "hiv_m0_icode4": lambda: case_mle("hiv_m0_icode4", dict(seqfile="HIVenvSweden.txt", treefile="HIVenvSweden.trees", NSsites=0, icode=4, kappa=.3, omega=1.3), HIVF, 13, "codon_m0"),
    "hiv_m0_icode5": lambda: case_mle("hiv_m0_icode5", dict(seqfile="HIVenvSweden.txt", treefile="HIVenvSweden.trees", NSsites=0, icode=5, kappa=.3, omega=1.3), HIVF, 13, "codon_m0"),
    "hiv_m0_icode11": lambda: case_mle("hiv_m0_icode11", dict(seqfile="HIVenvSweden.txt", treefile="HIVenvSweden.trees", NSsites=0, icode=11, kappa=.3, omega=1.3), HIVF, 13, "codon_m0"),
    # the free-ratio model (model = 1: an omega for every branch; Yang 1998) on the small lysozyme data set, and the one-ratio model next to it
    "lysos_free": lambda: case_mle("lysos_free", dict(seqfile="lysozymeSmall.txt", treefile="lysozymeSmall.free.trees", model=1, NSsites=0, kappa=2, omega=.4, cleandata=0),
                                   {"lysozymeSmall.txt": EX + "/lysozyme/lysozymeSmall.txt", "lysozymeSmall.free.trees": os.path.join(HERE, "data", "lysozymeSmall.free.trees")}, 7, "codon_branch"),
    # ... and the three-ratio branch model with the last omega fixed at 1 (examples/lysozyme/README.txt: table 1 E & J of Yang 1998)
    "lysos_branch_fix": lambda: case_mle("lysos_branch_fix", dict(seqfile="lysozymeSmall.txt", treefile="lysozymeSmall.EJ.trees", model=2, NSsites=0, kappa=2, fix_omega=1, omega=1, cleandata=0),
                                         {"lysozymeSmall.txt": EX + "/lysozyme/lysozymeSmall.txt", "lysozymeSmall.EJ.trees": os.path.join(HERE, "data", "lysozymeSmall.EJ.trees")}, 7, "codon_branch"),
    # TipDate with yyyy-mm-dd dates: the HIV-2 data with every name's year turned into a calendar date (ymd_names below; the tests rebuild the
    # same files); dates become days since 1970-01-01 (mktime — run with TZ=UTC), so the time unit is 36 500 days
    "hiv2_tipdate_ymd": lambda: case_mle("hiv2_tipdate_ymd", dict(seqfile="HIV2ge.ymd.txt", treefile="HIV2ge.ymd.tree", model=4, clock=1, TipDate="1 36500", kappa=2, fix_alpha=0, alpha=0.5, ncatG=5, cleandata=0),
                                         {"HIV2ge.ymd.txt": ymd_names(open(EX + "/TipDate.HIV2/HIV2ge.txt").read()), "HIV2ge.ymd.tree": ymd_names(open(os.path.join(HERE, "data", "HIV2ge.tree1")).read())},
                                         33, "nuc_tipdate", prog="baseml", seqtype="nuc"),
    # clade labels: '$k' labels every branch of a clade that has no '#' of its own (nested: the inner label wins)
    "lysos_clade_label": lambda: case_mle("lysos_clade_label", dict(seqfile="lysozymeSmall.txt", treefile="lysozymeSmall.clade.trees", model=2, NSsites=0, kappa=2, omega=.4, cleandata=0),
                                          {"lysozymeSmall.txt": EX + "/lysozyme/lysozymeSmall.txt", "lysozymeSmall.clade.trees": os.path.join(HERE, "data", "lysozymeSmall.clade.trees")}, 7, "codon_branch"),
    "brown_hky85_clock": case_brown_clock,
    "hiv_m0_f3x4mg": lambda: case_mle("hiv_m0_f3x4mg", dict(seqfile="HIVenvSweden.txt", treefile="HIVenvSweden.trees", NSsites=0, CodonFreq=5, kappa=.3, omega=1.3), HIVF, 13, "codon_m0"),
    "hiv_m0_f1x4mg": lambda: case_mle("hiv_m0_f1x4mg", dict(seqfile="HIVenvSweden.txt", treefile="HIVenvSweden.trees", NSsites=0, CodonFreq=4, kappa=.3, omega=1.3), HIVF, 13, "codon_m0"),
    "horai_mg0": lambda: case_horai(0), "horai_mg2": lambda: case_horai(2), "horai_mg3": lambda: case_horai(3), "horai_mg4": lambda: case_horai(4),
    "horai_mg0_g5": lambda: case_horai(0, 0.5),
    "lysin_mg0": lambda: case_lysin(0), "lysin_mg2": lambda: case_lysin(2), "lysin_mg3": lambda: case_lysin(3), "lysin_mg4": lambda: case_lysin(4),
    "lyso_bsa": lambda: case_mle("lyso_bsa", dict(LYSO_CTL, model=2, NSsites=2, omega=1.5), LYSO, 19, "codon_branchsite"),
    "lyso_bsa_null": lambda: case_mle("lyso_bsa_null", dict(LYSO_CTL, model=2, NSsites=2, fix_omega=1, omega=1), LYSO, 19, "codon_branchsite"),
    "lyso_bsb": lambda: case_mle("lyso_bsb", dict(LYSO_CTL, model=2, NSsites=3, omega=1.5), LYSO, 19, "codon_branchsite"),
    "ecp_cmc": lambda: case_mle("ecp_cmc", dict(ECP_CTL, model=3, NSsites=2), ECP, 15, "codon_clade"),
    "ecp_cmd": lambda: case_mle("ecp_cmd", dict(ECP_CTL, model=3, NSsites=3), ECP, 15, "codon_clade"),
    "ecp_m2arel": lambda: case_mle("ecp_m2arel", dict(ECP_CTL, model=0, NSsites=22), ECP, 15, "codon_nssites"),
    # aaDist = 7 (AAClasses, OmegaAA.dat: radical / conserved changes) under M0 and under the two-ratio branch model
    # (examples/mtCDNAape/README.txt:19-21 publishes both maxima)
    "mtcdna_aaclass_m0": lambda: case_mle("mtcdna_aaclass_m0", dict(seqfile="mtCDNAape.txt", treefile="mtCDNAape.trees", model=0, NSsites=0, icode=1, aaDist=7, cleandata=0, kappa=2, omega=.4),
                                          dict(MTAPE, **{"OmegaAA.dat": EX + "/mtCDNAape/OmegaAA.dat"}), 6, "codon_aaclasses"),
    "mtcdna_aaclass_branch": lambda: case_mle("mtcdna_aaclass_branch", dict(seqfile="mtCDNAape.txt", treefile="mtCDNAape.trees", model=2, NSsites=0, icode=1, aaDist=7, cleandata=0, kappa=2, omega=.4),
                                              dict(MTAPE, **{"OmegaAA.dat": EX + "/mtCDNAape/OmegaAA.dat"}), 6, "codon_aaclasses"),
    # codon-based amino-acid models on the 7-ape mitochondrial proteins (examples/mtCDNA/AAcodon.result.txt:75-97 publishes both
    # maxima): 6 = FromCodon (20 states, rates aggregated from the codon chain), 5 = FromCodon0 (codon chain, amino acids as codon sets)
    "mtcdnapri_fromcodon": lambda: case_mle("mtcdnapri_fromcodon", dict(seqfile="mtCDNApri.aa", treefile="mtCDNApri.trees", seqtype=2, model=6, icode=1, CodonFreq=0, kappa=3, omega=1.5, cleandata=1),
                                            MTPRI_AA, 7, "aa_fromcodon", seqtype="aa"),
    "mtcdnapri_fromcodon0": lambda: case_mle("mtcdnapri_fromcodon0", dict(seqfile="mtCDNApri.aa", treefile="mtCDNApri.trees", seqtype=2, model=5, icode=1, CodonFreq=0, kappa=3, omega=1.5, cleandata=1),
                                             MTPRI_AA, 7, "aa_fromcodon0", seqtype="aa3"),
    "hiv_m3": lambda: case_mle("hiv_m3", dict(seqfile="HIVenvSweden.txt", treefile="HIVenvSweden.trees", NSsites=3, ncatG=3, kappa=.3, omega=1.3), HIVF, 13, "codon_nssites"),
    "hiv_m4": lambda: case_mle("hiv_m4", dict(seqfile="HIVenvSweden.txt", treefile="HIVenvSweden.trees", NSsites=4, ncatG=5, kappa=.3, omega=1.3), HIVF, 13, "codon_nssites"),
    "hiv_m5": lambda: case_mle("hiv_m5", dict(seqfile="HIVenvSweden.txt", treefile="HIVenvSweden.trees", NSsites=5, ncatG=10, kappa=.3, omega=1.3), HIVF, 13, "codon_nssites"),
    "hiv_m6": lambda: case_mle("hiv_m6", dict(seqfile="HIVenvSweden.txt", treefile="HIVenvSweden.trees", NSsites=6, ncatG=10, kappa=.3, omega=1.3), HIVF, 13, "codon_nssites"),
    "hiv_m9": lambda: case_mle("hiv_m9", dict(seqfile="HIVenvSweden.txt", treefile="HIVenvSweden.trees", NSsites=9, ncatG=10, kappa=.3, omega=1.3), HIVF, 13, "codon_nssites"),
    "hiv_m10": lambda: case_mle("hiv_m10", dict(seqfile="HIVenvSweden.txt", treefile="HIVenvSweden.trees", NSsites=10, ncatG=10, kappa=.3, omega=1.3), HIVF, 13, "codon_nssites"),
    "hiv_m11": lambda: case_at("hiv_m11", dict(seqfile="HIVenvSweden.txt", treefile="HIVenvSweden.trees", NSsites=11, ncatG=10, kappa=.3, omega=1.3), HIVF, 13, "codon_nssites",
                               [float(v) for v in HIV_X["m7"][1].split()[:24]] + [0.8, 0.4, 1.2, 2.0, 0.9], 23),
    "hiv_m12": lambda: case_mle("hiv_m12", dict(seqfile="HIVenvSweden.txt", treefile="HIVenvSweden.trees", NSsites=12, ncatG=10, kappa=.3, omega=1.3), HIVF, 13, "codon_nssites"),
    "hiv_m13": lambda: case_at("hiv_m13", dict(seqfile="HIVenvSweden.txt", treefile="HIVenvSweden.trees", NSsites=13, ncatG=10, kappa=.3, omega=1.3), HIVF, 13, "codon_nssites",
                               [float(v) for v in HIV_X["m7"][1].split()[:24]] + [0.3, 0.4, 2.5, 0.3, 0.5, 1.2], 23),
    "hiv_m0": lambda: case_hiv("m0"), "hiv_m1a": lambda: case_hiv("m1a"), "hiv_m2a": lambda: case_hiv("m2a"),
    "hiv_m7": lambda: case_hiv("m7"), "hiv_m8": lambda: case_hiv("m8"),
    "stewart_lg_g4": case_stewart, "mhc_m0_scaled": case_mhc,
    "syn_codon_m0": case_syn_codon, "syn_nuc_gtr_g4": case_syn_nuc, "brown_hky85": case_brown, "brown_hky85_anc": case_brown_anc, "mtcdna_branch": case_mtcdna_branch,
    # BASELINE configs[3] / configs[1] at full size (reference: ~2 min and 6.8 GB / ~2 s): lnL + strided log f_h sample
    "syn_codon_m0_full": lambda: case_syn_codon(1_000_000, "syn_codon_m0_full", sample=997),
    "syn_nuc_gtr_g4_full": lambda: case_syn_nuc(100_000, "syn_nuc_gtr_g4_full", sample=97),
    "stewart_two_trees": case_tree_comparison,
    "syn_aa_g4_full": case_syn_aa, "syn_aa_g4": lambda: case_syn_aa(3000, "syn_aa_g4", sample=None),
    # the NSsites sweep on the C4 data at full size (reference: 3 min ... 20 min each, 7 GB)
    "syn_codon_m1a_full": lambda: case_syn_codon_ns("m1a"), "syn_codon_m2a_full": lambda: case_syn_codon_ns("m2a"),
    "syn_codon_m7_full": lambda: case_syn_codon_ns("m7"), "syn_codon_m8_full": lambda: case_syn_codon_ns("m8"),
    "syn_codon_m2a_as_m3_full": lambda: case_syn_codon_ns("m2a_as_m3"), "syn_codon_m8_as_m3_full": lambda: case_syn_codon_ns("m8_as_m3"),
    "mcmctree_jc_clock1": case_mcmctree,
    # small versions of the same (CPU oracle test)
    "syn_codon_m2a_as_m3_4000": lambda: case_syn_codon_ns("m2a_as_m3", 4000, None), "syn_codon_m8_as_m3_4000": lambda: case_syn_codon_ns("m8_as_m3", 4000, None),
    "syn_codon_m7_4000": lambda: case_syn_codon_ns("m7", 4000, None),
}

if __name__ == "__main__":
    which = sys.argv[1:] or list(CASES)
    for c in which:
        CASES[c]()
