"""The drop-in boundary seen from PAML's side: the REFERENCE's own codeml, compiled from its sources where they lie with
integration/codeml_plfun.patch applied (oracle/Makefile -> oracle/_ref/codeml_gpu; the patch makes com.plfun — codeml.c:125,
assigned in GetInitials codeml.c:2338-2340 — call libpaml_amd.so), run on the reference's example data.  Everything above the
likelihood function is the reference's code: control-file parsing, ReadSeq / PatternWeight / EncodeSeqs, SetParameters, eigenQcodon,
and the optimiser ming2 (tools.c:6595).  The engine is fed com.z, nChara / CharaMap, com.fpatt, nodes[], com.nodeScale straight
from the reference's globals.  Checked: the lnL values the reference's ming2 ends at and prints — HIV NSsites 0 and 2 (-1137.688190,
-1106.445004: examples/HIVNSsites), MHC M0 (-8225.154790: examples/MHC.Swanson2002MBE/README.txt, 192 taxa, ambiguity codes, ten
scaling nodes) — and the per-pattern `lnf` file against the unmodified binary's on the same control file.
codeml / baseml perturb their starting values with a clock-seeded generator (SetSeed(-1, 0) in main), so every run of the optimiser
takes another path and ends within its convergence tolerance of the optimum, not at identical digits: lnL is compared to 2e-5 (the
published values have six decimals), per-pattern values to 1e-3.
Test infrastructure: nothing in the product depends on oracle/_ref."""
import os
import re
import subprocess
import warnings
import time

import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_GPU = os.path.join(REPO, "oracle", "_ref", "codeml_gpu")
REF_CPU = os.path.join(REPO, "oracle", "_ref", "codeml")
BASEML_GPU = os.path.join(REPO, "oracle", "_ref", "baseml_gpu")
BASEML_CPU = os.path.join(REPO, "oracle", "_ref", "baseml")
DATA = os.path.join(helpers.GOLDEN, "data")
TOL = 2e-5      # |lnL - published|: what a run of the reference's optimiser from its randomised starting values reproduces

HIV_CTL = """seqfile = %(data)s/HIVenvSweden.txt
treefile = %(data)s/HIVenvSweden.trees
outfile = mlc
noisy = 3
verbose = 0
runmode = 0
seqtype = 1
CodonFreq = 2
clock = 0
aaDist = 0
model = 0
NSsites = 0 2
icode = 0
Mgene = 0
fix_kappa = 0
kappa = .3
fix_omega = 0
omega = 1.3
ncatG = 10
getSE = 0
RateAncestor = 0
Small_Diff = .45e-6
cleandata = 1
fix_blength = 0
"""

MHC_CTL = """seqfile = %(data)s/bigmhc.phy
treefile = %(data)s/bigmhc.trees
outfile = mlc
noisy = 3
verbose = 0
runmode = 0
seqtype = 1
CodonFreq = 2
model = 0
NSsites = 0
icode = 0
Mgene = 0
fix_kappa = 0
kappa = 1.6
fix_omega = 0
omega = .9
fix_alpha = 1
alpha = 0
ncatG = 10
clock = 0
getSE = 0
RateAncestor = 0
Small_Diff = .1e-6
method = 1
fix_blength = 2
"""


def run_program(argv, cwd, newlines, env=None, limit=int(os.environ.get("PAML_AMD_TEST_RUN_LIMIT_S", "300"))):
    """One run of a reference binary (patched or not), its prompts answered with empty lines.  A run that does not end within `limit` seconds
    is described — what its threads were waiting in (/proc), the tail of what it printed — in a warning and started ONCE more (round 6: two
    of ten runs of the whole tier had one such run, of a search that takes under a second, and nothing reproduced it: 160 + 100 repetitions
    alone and inside the tier all ended); the second time it fails the test with that description instead of holding the suite."""
    for attempt in (0, 1):
        p = subprocess.Popen(argv, cwd=cwd, stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env)
        try:
            out, _ = p.communicate(b"\n" * newlines, timeout=limit)
            return subprocess.CompletedProcess(argv, p.returncode, out, None)
        except subprocess.TimeoutExpired:
            where = []
            for tid in sorted(os.listdir("/proc/%d/task" % p.pid)):
                row = [tid]
                for f in ("comm", "wchan", "syscall"):
                    try:
                        with open("/proc/%d/task/%s/%s" % (p.pid, tid, f)) as fh:
                            row.append(fh.read().strip()[:60])
                    except OSError as e:
                        row.append("?%s" % e.errno)
                try:
                    with open("/proc/%d/task/%s/status" % (p.pid, tid)) as fh:
                        row.append([l.split(":")[1].strip() for l in fh if l.startswith("State")][0])
                except (OSError, IndexError):
                    pass
                where.append(" ".join(row))
            p.kill()
            out, _ = p.communicate()
            what = ("%s did not end within %d s; threads (tid comm wchan syscall state):\n%s\nits output ends with:\n%s"
                    % (argv[0], limit, "\n".join(where), (out or b"").decode(errors="replace")[-2500:]))
            if attempt:
                raise AssertionError(what)
            warnings.warn("started again after: " + what)


def run(exe, ctl, d, env=None, ctl_name="codeml.ctl"):
    d.mkdir()
    (d / ctl_name).write_text(ctl % {"data": DATA})
    t0 = time.perf_counter()
    r = run_program([exe, ctl_name], d, 20, env=env)
    dt = time.perf_counter() - t0
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0, out[-3000:]
    mlc = (d / "mlc").read_text()
    lnl = [float(m.group(1)) for m in re.finditer(r"lnL\(ntime:\s*\d+\s+np:\s*\d+\):\s+(-?\d+\.\d+)", mlc)]
    nfun = [int(m.group(1)) for m in re.finditer(r"(\d+) lfun,", out)]
    lnf = np.array([float(ln.split()[2]) for ln in (d / "lnf").read_text().splitlines() if re.match(r"\s*\d+\s+\d+\s+-\d", ln)])
    return lnl, lnf, nfun, dt, out


def need_binaries():
    if not (os.path.isfile(REF_GPU) and os.access(REF_GPU, os.X_OK)):
        pytest.skip("oracle/_ref/codeml_gpu is not built (make -C oracle, needs /root/reference)")


def test_patched_reference_codeml_reaches_the_published_lnl_on_hiv(tmp_path):
    need_binaries()
    lnl, lnf, nfun, dt, out = run(REF_GPU, HIV_CTL, tmp_path / "gpu")
    assert len(lnl) == 2, out[-2000:]
    assert abs(lnl[0] - (-1137.688190)) <= TOL and abs(lnl[1] - (-1106.445004)) <= TOL, lnl
    assert len(lnf) == 2 * 79                             # the lnf file holds both models' per-pattern values, one after the other
    # the unmodified program on the same control file: the same optimum, the same per-pattern values at it
    cl, clnf, cnfun, cdt, _ = run(REF_CPU, HIV_CTL, tmp_path / "cpu")
    assert np.allclose(lnl, cl, rtol=0, atol=2 * TOL)
    assert np.max(np.abs(lnf - clnf)) < 1e-3          # (two ming2 runs end within their convergence tolerance of each other)
    print("\nHIV NSsites 0 2 through the reference's own ming2: engine %.2f s (%s lfun), unmodified CPU program %.2f s (%s lfun)" % (dt, nfun, cdt, cnfun))
    # PAML_AMD_OFF=1: the same binary leaves com.plfun alone (the reference's own lfun / lfundG)
    ol, _, _, _, _ = run(REF_GPU, HIV_CTL.replace("NSsites = 0 2", "NSsites = 0"), tmp_path / "off", env=dict(os.environ, PAML_AMD_OFF="1"))
    assert abs(ol[0] - (-1137.688190)) <= TOL


def test_patched_reference_codeml_on_mhc_with_ambiguities_and_scaling_nodes(tmp_path):
    need_binaries()
    lnl, lnf, nfun, dt, out = run(REF_GPU, MHC_CTL, tmp_path / "gpu")
    assert len(lnl) == 1 and abs(lnl[0] - (-8225.154790)) <= TOL, (lnl, out[-1500:])
    g = helpers.load_golden("mhc_m0_scaled")
    assert len(lnf) == g["n_patt"]
    assert np.max(np.abs(lnf - np.array(g["logf"]))) < 1e-3      # the golden's kappa, omega are these MLEs printed with 6 decimals
    print("\nMHC M0 (192 taxa, fix_blength = 2) through the reference's own ming2: %.2f s, %s lfun" % (dt, nfun))


BASEML_CTL = """seqfile = %%(data)s/brown.nuc
treefile = %%(data)s/brown.trees
outfile = mlc
noisy = 2
verbose = 0
runmode = 0
model = %d
Mgene = 0
fix_kappa = 0
kappa = 5
fix_alpha = %d
alpha = %s
Malpha = 0
ncatG = 5
fix_rho = 1
rho = 0.
nparK = 0
clock = 0
nhomo = 0
getSE = 0
RateAncestor = 0
Small_Diff = 7e-6
cleandata = 1
method = 0
"""


@pytest.mark.parametrize("model,fix_alpha,alpha,published", [(4, 1, "0", -2665.422858), (0, 1, "0", None), (7, 0, "0.5", None), (6, 0, "0.5", None)])
def test_patched_reference_baseml_matches_the_unmodified_program(model, fix_alpha, alpha, published, tmp_path):
    """integration/baseml_plfun.patch (oracle/_ref/baseml_gpu): the reference's baseml with com.plfun on the engine — HKY85 (Cijk form;
    brown.nuc's published -2665.422858), JC69 (the closed form of PMatK80), REV + gamma and TN93 + gamma (lfundG: class rates through
    SetPSiteClass, RootTN93 / eigenQREVbase per call) — against the unmodified program on the same control file, its own ming2 in both."""
    if not (os.path.isfile(BASEML_GPU) and os.access(BASEML_GPU, os.X_OK)):
        pytest.skip("oracle/_ref/baseml_gpu is not built (make -C oracle, needs /root/reference)")
    ctl = BASEML_CTL % (model, fix_alpha, alpha)
    lnl, lnf, _, dt, out = run(BASEML_GPU, ctl, tmp_path / "gpu", ctl_name="baseml.ctl")
    cl, clnf, _, cdt, _ = run(BASEML_CPU, ctl, tmp_path / "cpu", ctl_name="baseml.ctl")
    assert len(lnl) == 1 and len(cl) == 1, out[-1500:]
    assert abs(lnl[0] - cl[0]) <= 2 * TOL, (lnl, cl)
    if published is not None:
        assert abs(lnl[0] - published) <= TOL
    assert len(lnf) == len(clnf) > 20 and np.max(np.abs(lnf - clnf)) < 1e-3


# ---- every model family the binding takes over, pinned to the reference's printed digits -------------------------------------------------
# The deterministic single-evaluation mode (`in.codeml` / `in.baseml` starting with -1 = "these are the parameters, do not
# iterate") through the PATCHED program, on the control files of the golden cases: the lnL it prints (the first com.plfun call: eigen
# systems decomposed on the device from the rate matrices eigenQcodon built) against the unmodified program's to 2e-6 (six printed
# decimals), and the `lnf` file (the second call, com.print < 0: host eigen systems, uploaded) to 2e-8 (ten decimals).
CTL_OF = {"hiv_m0_f1x4mg": "hiv_ns0_cf4", "hiv_m0_f3x4mg": "hiv_ns0_cf5", "hiv_m0": "hiv_ns0", "hiv_m1a": "hiv_ns1", "hiv_m2a": "hiv_ns2", "hiv_m3": "hiv_ns3", "hiv_m7": "hiv_ns7", "hiv_m8": "hiv_ns8"}
SINGLE = [("codeml", n) for n in ("hiv_m0", "hiv_m2a", "hiv_m3", "hiv_m8", "lysos_branch_fix", "lysos_clade_label", "mtcdna_branch", "lyso_bsa", "lyso_bsa_null",
                                  "lyso_bsb", "ecp_cmc", "ecp_cmd", "ecp_m2arel", "lysin_mg0", "lysin_mg2", "lysin_mg3", "lysin_mg4", "stewart_lg_g4")] + \
         [("baseml", n) for n in ("brown_hky85", "brown_t92_g4", "horai_mg0", "horai_mg0_g5")]
# round 6: the codeml binding takes the mutation-selection models, aaDist / AAClasses, codon frequencies as parameters, REVaa and the codon-based
# amino-acid models (what libpamlh already served on the engine; goldens of round 2)
SINGLE += [("codeml", n) for n in ("hiv_fmutsel", "hiv_fmutsel0", "hiv_fmutsel_est", "hiv_fmutsel0_est", "hiv_fmutsel0_m2a", "hiv_f3x4_est", "hiv_f1x4mg_est", "hiv_f3x4_est_m7",
                                   "hiv_m0_f1x4mg", "hiv_m0_f3x4mg", "mtcdnapri_aadist1", "mtcdnapri_aadist_m2", "mtcdna_aaclass_m0", "mtcdna_aaclass_branch",
                                   "mtcdnapri_revaa", "mtcdnapri_revaa0", "mtcdnapri_fromcodon", "mtcdnapri_fromcodon0", "stewart_eqinput")]
# round 6: the baseml binding takes nhomo 1 .. 5 (an eigen system per branch: treesub.c:7512-7519), Mgene 2 .. 4 (SetPGene), clock 1 / 2
# (GetBranchRate folded into lengths and gene rates; TipDate), UNREST (Q + matexp), rho != 0 (lfunAdG: paml_amd_eval_adg)
SINGLE += [("baseml", n) for n in ("brown_f84", "brown_hky85_nhomo1", "brown_hky85_nhomo2", "brown_hky85_nhomo3", "brown_hky85_nhomo5", "brown_f84_nhomo4",
                                  "brown_t92_nhomo3_g4", "brown_hky85_clock", "brown_hky85_clock2", "hiv2_tipdate", "hiv2_tipdate_clock2", "brown_unrest",
                                  "horai_mg2", "horai_mg3", "horai_mg4", "brown_hky85_adg")]


def single_evaluation(prog, name, d, exe, extra_ctl="", env=None):
    g = helpers.load_golden(name)
    ctl = open(os.path.join(helpers.GOLDEN, "ctl", CTL_OF.get(name, name) + ".ctl")).read()
    ctl = ctl.replace("../data/", DATA + "/").replace("../ctl/", os.path.join(helpers.GOLDEN, "ctl") + "/")
    ctl += "\noutfile = mlc\nnoisy = 3\ngetSE = 0\nRateAncestor = 0\n" + extra_ctl
    d.mkdir()
    (d / (prog + ".ctl")).write_text(ctl)
    for dat in ("grantham.dat", "miyata.dat", "OmegaAA.dat"):      # (aaDist: the reference opens these by name in the working directory)
        (d / dat).write_text(open(os.path.join(helpers.GOLDEN, "ctl", dat)).read())
    (d / ("in." + prog)).write_text("-1 " + " ".join("%.6f" % v for v in g["x"]) + "\n")
    r = run_program([exe, prog + ".ctl"], d, 50, env=env)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0, out[-3000:]
    m = re.findall(r"lnL\s*=\s*(-?[0-9.]+)", out)
    assert m, out[-3000:]
    lnf = np.array([float(ln.split()[2]) for ln in (d / "lnf").read_text().splitlines() if re.match(r"\s*\d+\s+\d+(\.\d+)?\s+-\d", ln)])
    return g, float(m[-1]), lnf, out


@pytest.mark.parametrize("prog,name", SINGLE)
def test_single_evaluation_through_the_patched_reference_matches_the_printed_digits(prog, name, tmp_path):
    exe = REF_GPU if prog == "codeml" else BASEML_GPU
    if not (os.path.isfile(exe) and os.access(exe, os.X_OK)):
        pytest.skip("oracle/_ref/%s_gpu is not built (make -C oracle, needs /root/reference)" % prog)
    g, lnl, lnf, out = single_evaluation(prog, name, tmp_path / "gpu", exe, env=dict(os.environ, PAML_AMD_ANNOUNCE="1"))
    assert "paml_amd" not in out, out[-2000:]
    assert "engine behind com.plfun" in out, out[-2000:]      # (the engine answered, not the reference's own function the binding falls back to for what it does not take)
    assert abs(lnl - g["lnL"]) <= 2e-6, (name, lnl, g["lnL"])
    if g.get("logf"):      # (lfunAdG's sites are not independent: no per-pattern values)
        assert len(lnf) == g["n_patt"] and np.max(np.abs(lnf - np.array(g["logf"]))) <= 2e-8, (name, float(np.max(np.abs(lnf - np.array(g["logf"])))))
    # the engine really was behind com.plfun: with PAML_AMD_OFF the same binary prints the same digits from the reference's own functions
    if name in ("lyso_bsa", "lysin_mg2", "horai_mg0", "brown_hky85_nhomo3", "brown_hky85_clock2", "horai_mg4"):
        g2, lnl2, lnf2, out2 = single_evaluation(prog, name, tmp_path / "off", exe, env=dict(os.environ, PAML_AMD_OFF="1"))
        assert abs(lnl2 - g["lnL"]) <= 2e-6


def test_a_run_shorter_than_its_background_compilation_exits_cleanly(tmp_path):
    """A small data set's cooperative per-tree kernel is compiled on a worker thread while the interpreter serves; a single evaluation
    is over long before (0.3 s against ~1 s), and the reference never destroys its engine: exit() used to run hiprtc's static
    destructors under the compiling thread (SIGSEGV after the results were written — seen only where no earlier run had left the code
    object in the cache).  The library now joins its worker threads at exit.  PAML_AMD_JIT_CACHE=0: nothing cached, every run compiles."""
    need_binaries()
    for i in range(2):
        g, lnl, lnf, out = single_evaluation("codeml", "lysos_branch_fix", tmp_path / ("run%d" % i), REF_GPU, env=dict(os.environ, PAML_AMD_JIT_CACHE="0"))
        assert abs(lnl - g["lnL"]) <= 2e-6


OPT_CASES = [("lyso_bsa", "", -1035.533916), ("ecp_cmc", "", None), ("lysin_mg2", "", None), ("lysos_branch_fix", "", None),
             ("hiv_m0", "method = 1\n", -1137.688190), ("hiv_m2a", "method = 1\n", -1106.445004),
             # (branch-site A under method = 1 stops at -1035.530508 — the UNMODIFIED program does exactly that on this control file, 12 s on
             #  a core here; the published -1035.533916 is the method = 0 optimum above)
             ("lyso_bsa", "method = 1\n", -1035.530508)]


@pytest.mark.parametrize("name,extra,published", OPT_CASES)
def test_patched_reference_optimises_the_wider_model_families(name, extra, published, tmp_path):
    """The reference's own optimisers on the engine: ming2 (method = 0) and minB / minbranches (method = 1: every lfunt / lfuntdd call is
    paml_amd_eval_branch) for branch-site model A on the lysozyme data (examples/lysozyme: -1035.533916), clade model C (examples/CladeModelCD),
    the two partitions of examples/lysin with Mgene = 2, a two-ratio branch model, and the HIV site models with method = 1; the optimum
    is the golden's `mle_lnL` (the unmodified program's) or the published value."""
    need_binaries()
    g = helpers.load_golden(name)
    ctl = open(os.path.join(helpers.GOLDEN, "ctl", CTL_OF.get(name, name) + ".ctl")).read()
    ctl = ctl.replace("../data/", DATA + "/").replace("../ctl/", os.path.join(helpers.GOLDEN, "ctl") + "/")
    ctl += "\noutfile = mlc\nnoisy = 3\ngetSE = 0\nRateAncestor = 0\n" + extra
    d = tmp_path / "gpu"
    d.mkdir()
    (d / "codeml.ctl").write_text(ctl)
    t0 = time.perf_counter()
    r = run_program([REF_GPU, "codeml.ctl"], d, 50)
    dt = time.perf_counter() - t0
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0, out[-3000:]
    lnl = [float(m.group(1)) for m in re.finditer(r"lnL\(ntime:\s*\d+\s+np:\s*\d+\):\s+(-?\d+\.\d+)", (d / "mlc").read_text())]
    want = published if published is not None else g.get("mle_lnL", g["lnL"])
    assert len(lnl) == 1 and abs(lnl[0] - want) <= 5e-5, (name, extra, lnl, want, out[-1500:])
    print("\n%s %s through the reference's own optimiser on the engine: lnL %.6f in %.2f s" % (name, extra.strip(), lnl[0], dt))


def test_patched_baseml_with_method_1_matches_the_unmodified_program(tmp_path):
    """baseml through minB / minbranches on the engine (4 states: the P / dP / ddP form of paml_amd_eval_branch), HKY85 + gamma on
    brown.nuc, against the unmodified program with the same control file."""
    if not (os.path.isfile(BASEML_GPU) and os.access(BASEML_GPU, os.X_OK)):
        pytest.skip("oracle/_ref/baseml_gpu is not built (make -C oracle, needs /root/reference)")
    ctl = (BASEML_CTL % (4, 0, "0.5")).replace("method = 0", "method = 1")
    lnl, lnf, _, dt, out = run(BASEML_GPU, ctl, tmp_path / "gpu", ctl_name="baseml.ctl")
    cl, clnf, _, cdt, _ = run(BASEML_CPU, ctl, tmp_path / "cpu", ctl_name="baseml.ctl")
    assert len(lnl) == 1 and len(cl) == 1 and abs(lnl[0] - cl[0]) <= 2 * TOL, (lnl, cl, out[-1500:])
    assert np.max(np.abs(lnf - clnf)) < 1e-3


def _num_tokens(text):
    out = []
    for tok in text.split():
        try:
            out.append(float(tok.strip("(),")))
        except ValueError:
            out.append(tok)
    return out


@pytest.mark.parametrize("prog,name,extra", [("codeml", "hiv_m0", ""), ("codeml", "stewart_lg_g4", ""), ("codeml", "stewart_lg_g4", "method = 1\n"), ("baseml", "brown_hky85", "")])
def test_rate_ancestor_through_the_patched_reference_reads_host_partials(prog, name, extra, tmp_path):
    """RateAncestor = 1: AncestralSeqs (treesub.c:7071) -> ProbSitePattern / PostProbNode -> updateconP read the HOST's nodes[].conP, which the
    engine behind com.plfun never fills.  The binding steps aside for the reconstruction (gpu_suspend: one evaluation by the reference's own
    function, then updateconP is the reference's again): the `rst` file of the patched program equals the unmodified program's, number by
    number, with one site class, with gamma rates, and with gamma rates under method = 1 (conditional probabilities kept per site class;
    NSsites models with method = 1 are left out: after the BEB the unmodified program's own reconstruction is inconsistent there)."""
    exe, cpu = (REF_GPU, REF_CPU) if prog == "codeml" else (BASEML_GPU, BASEML_CPU)
    if not (os.path.isfile(exe) and os.access(exe, os.X_OK)):
        pytest.skip("oracle/_ref/%s_gpu is not built (make -C oracle, needs /root/reference)" % prog)
    g, lnl, lnf, out = single_evaluation(prog, name, tmp_path / "gpu", exe, extra_ctl="RateAncestor = 1\n" + extra)
    assert "sum!=1" not in out and abs(lnl - g["lnL"]) <= 2e-6
    g2, lnl2, lnf2, out2 = single_evaluation(prog, name, tmp_path / "cpu", cpu, extra_ctl="RateAncestor = 1\n" + extra)
    a, b = _num_tokens((tmp_path / "gpu" / "rst").read_text()), _num_tokens((tmp_path / "cpu" / "rst").read_text())
    assert len(a) == len(b) > 500
    for x, y in zip(a, b):
        if isinstance(x, float) and isinstance(y, float):
            assert abs(x - y) <= 2e-4 + 1e-6 * abs(y), (x, y)
        else:
            assert x == y, (x, y)


@pytest.mark.parametrize("name,extra", [("hiv_m0", ""), ("hiv_m2a", ""), ("hiv_m8", ""), ("lyso_bsa", ""), ("ecp_cmc", ""), ("lysin_mg2", ""), ("stewart_lg_g4", ""),
                                        ("hiv_m2a", "method = 1\n")])
def test_batched_gradient_equals_the_serial_one(name, extra, tmp_path):
    """integration/tools_gradient_seam.patch: gradientB (tools.c:6561) hands its np .. 2 np perturbed vectors to the binding, which evaluates them
    in ONE paml_amd_eval_batch (one paml_amd_set_eigen_qrev_batch for the rate matrices the perturbed kappa / omega / proportions change).
    PAML_AMD_GRADIENT_CHECK=1 makes the binding evaluate the same vectors one by one afterwards and print the largest difference: the
    batched values are those of the serial calls (same kernels, same eigen systems) to 1e-9, for every gradient of an optimisation."""
    need_binaries()
    ctl = open(os.path.join(helpers.GOLDEN, "ctl", CTL_OF.get(name, name) + ".ctl")).read()
    ctl = ctl.replace("../data/", DATA + "/").replace("../ctl/", os.path.join(helpers.GOLDEN, "ctl") + "/")
    ctl += "\noutfile = mlc\nnoisy = 3\ngetSE = 0\nRateAncestor = 0\n" + extra
    d = tmp_path / "gpu"
    d.mkdir()
    (d / "codeml.ctl").write_text(ctl)
    r = run_program([REF_GPU, "codeml.ctl"], d, 50,
                       env=dict(os.environ, PAML_AMD_GRADIENT_CHECK="1"))
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0, out[-3000:]
    checks = [(int(m.group(1)), int(m.group(2)), float(m.group(3))) for m in re.finditer(r"gradient check: (\d+) vectors, (\d+) model parts, max \|batched - serial\| = ([0-9.e+-]+)", out)]
    assert len(checks) >= 3, out[-2000:]                         # every gradient of the run went through the batch
    assert max(c[2] for c in checks) <= 1e-9, max(checks, key=lambda c: c[2])
    assert max(c[1] for c in checks) >= 2                         # ... with perturbed substitution parameters among the vectors
    g = helpers.load_golden(name)
    lnl = [float(m.group(1)) for m in re.finditer(r"lnL\(ntime:\s*\d+\s+np:\s*\d+\):\s+(-?\d+\.\d+)", (d / "mlc").read_text())]
    assert len(lnl) == 1 and abs(lnl[0] - g.get("mle_lnL", g["lnL"])) <= 5e-5, (lnl, g.get("mle_lnL", g["lnL"]))


@pytest.mark.parametrize("name,what", [("hiv_m2a", "fx_r"), ("hiv_m8", "fx_r"), ("lyso_bsa", "121 omega sets"), ("ecp_cmc", "omega sets"), ("ecp_cmd", "omega sets")])
def test_neb_and_beb_tables_through_the_engine_equal_the_reference_code_paths(name, what, tmp_path):
    """After the iteration the reference calls fx_r directly for the NEB table (lfunNSsites_rate, codeml.c:5268) and for the BEB grid of M2a / M8
    (get_grid_para_like_M2M8, codeml.c:6280), and for branch-site model A / the clade models runs one ConditionalPNode pass per omega set of
    the grid with an eigen-decomposition per branch (get_grid_para_like_ACD; 1.3 s of the 1.7 s lysozyme run).  The binding takes those over
    (gpu_fx_r; gpu_beb_collect / gpu_beb_flush: 21 decompositions and ONE paml_amd_eval_batch of 121 elements): the same binary with
    PAML_AMD_NO_BEB=1 leaves them to the reference's own code.  Both at the golden's estimates (`in.codeml` starting with -1: no iteration,
    so the two runs stand at the same point): the `rst` files (every NEB / BEB posterior the program prints) and the main file's BEB section
    agree number by number, to one unit of the last printed digit."""
    need_binaries()
    res = {}
    for tag, env in (("engine", {}), ("host", {"PAML_AMD_NO_BEB": "1"})):
        d = tmp_path / tag
        t0 = time.perf_counter()
        g, lnl, lnf, out = single_evaluation("codeml", name, d, REF_GPU, env=dict(os.environ, PAML_AMD_TIMING="1", **env))
        dt = time.perf_counter() - t0
        assert abs(lnl - g["lnL"]) <= 2e-6
        mlc = (d / "mlc").read_text()
        assert "Bayes Empirical Bayes" in mlc
        res[tag] = ((d / "rst").read_text(), mlc[mlc.index("Bayes Empirical Bayes"):].split("Time used")[0], out, dt)
    timing = [ln for ln in res["engine"][2].splitlines() if ln.startswith("paml_amd timing")][-1]
    host_timing = [ln for ln in res["host"][2].splitlines() if ln.startswith("paml_amd timing")][-1]
    if what == "fx_r":      # NEB + the BEB grid, one evaluation each
        assert re.search(r"fx_r \(NEB, BEB grid of M2a / M8\) 2 calls", timing), timing
    else:
        n_sets = int(re.search(r"clade models (\d+) omega sets", timing).group(1))
        assert n_sets >= 111 and (what != "121 omega sets" or n_sets == 121), timing
        assert re.search(r"fx_r \(NEB, BEB grid of M2a / M8\) 1 calls", timing), timing
    assert re.search(r"fx_r \(NEB, BEB grid of M2a / M8\) 0 calls", host_timing) and " 0 omega sets" in host_timing, host_timing
    for k in (0, 1):      # (0.2565 prints as 0.256 or 0.257 on a difference of 1e-13)
        a, b = res["engine"][k].split(), res["host"][k].split()
        assert len(a) == len(b) > 100
        for x, y in zip(a, b):
            xs, ys = x.strip("(),*+"), y.strip("(),*+")
            try:
                fx, fy = float(xs), float(ys)
            except ValueError:
                assert x == y, (name, x, y)
                continue
            decimals = len(ys.split(".")[1]) if "." in ys and "e" not in ys.lower() else 0
            assert abs(fx - fy) <= 1.01 * 10.0 ** (-decimals) + 1e-6 * abs(fy), (name, x, y)
    print("\n%s at its estimates through codeml_gpu: %.2f s with the NEB / BEB evaluations on the engine, %.2f s with the reference's own" % (name, res["engine"][3], res["host"][3]))


@pytest.mark.parametrize("name,extra", [("brown_hky85", ""), ("brown_t92_g4", ""), ("horai_mg0_g5", ""), ("horai_mg3", ""), ("brown_hky85_clock", ""), ("hiv2_tipdate_clock2", "")])
def test_baseml_batched_gradient_equals_the_serial_one(name, extra, tmp_path):
    """Round 6: baseml's gradientB through the seam too (lfun_gpu_batch of integration/baseml_plfun.patch): one paml_amd_eval_batch per gradient —
    kappa / rate parameters / alpha perturbed (an eigen set per distinct model part), several genes with their own systems, node ages and
    local-clock rates perturbed under the clock models (every vector its own branch lengths through SetBranch + GetBranchRate).  The batched
    values are those of the serial calls to 1e-9 for every gradient of a run, and the run ends at the unmodified program's optimum."""
    if not (os.path.isfile(BASEML_GPU) and os.access(BASEML_GPU, os.X_OK)):
        pytest.skip("oracle/_ref/baseml_gpu is not built (make -C oracle, needs /root/reference)")
    ctl = open(os.path.join(helpers.GOLDEN, "ctl", name + ".ctl")).read()
    ctl = ctl.replace("../data/", DATA + "/").replace("../ctl/", os.path.join(helpers.GOLDEN, "ctl") + "/")
    ctl += "\noutfile = mlc\nnoisy = 3\ngetSE = 0\nRateAncestor = 0\n" + extra
    res = {}
    for tag, exe, env in (("gpu", BASEML_GPU, dict(os.environ, PAML_AMD_GRADIENT_CHECK="1")), ("cpu", BASEML_CPU, None)):
        d = tmp_path / tag
        d.mkdir()
        (d / "baseml.ctl").write_text(ctl)
        r = run_program([exe, "baseml.ctl"], d, 50, env=env)
        out = r.stdout.decode(errors="replace")
        assert r.returncode == 0, out[-3000:]
        lnl = [float(m.group(1)) for m in re.finditer(r"lnL\(ntime:\s*\d+\s+np:\s*\d+\):\s+(-?\d+\.\d+)", (d / "mlc").read_text())]
        res[tag] = (lnl, out)
    out = res["gpu"][1]
    checks = [(int(m.group(1)), int(m.group(2)), float(m.group(3))) for m in re.finditer(r"gradient check: (\d+) vectors, (\d+) model parts, max \|batched - serial\| = ([0-9.e+-]+)", out)]
    assert len(checks) >= 3, out[-2000:]
    assert max(c[2] for c in checks) <= 1e-9, max(checks, key=lambda c: c[2])
    assert len(res["gpu"][0]) == len(res["cpu"][0]) >= 1 and abs(res["gpu"][0][0] - res["cpu"][0][0]) <= 5e-5, (res["gpu"][0], res["cpu"][0])


@pytest.mark.parametrize("prog,ctl_name,extra", [("codeml", "lysos_m0_clock", ""), ("codeml", "lysos_m0_clock", "clock = 2\n"), ("codeml", "stewart_lg_g4", "fix_rho = 0\nrho = 0.2\n"),
                                                 ("baseml", "brown_hky85_adg", "")])
def test_clock_models_and_correlated_rates_through_the_patched_reference(prog, ctl_name, extra, tmp_path):
    """Round 6: codon M0 under the global and the local clock (SetBranch's node ages, GetBranchRate folded into the lengths handed over; the
    batched gradient perturbs ages and rates), and rho != 0 (lfunAdG: fx_r on the engine, the chain over the sites in paml_amd_eval_adg) for
    amino-acid and nucleotide data — the reference's own ming2 on the engine against the unmodified program on the same control file."""
    exe, cpu = (REF_GPU, REF_CPU) if prog == "codeml" else (BASEML_GPU, BASEML_CPU)
    if not (os.path.isfile(exe) and os.access(exe, os.X_OK)):
        pytest.skip("oracle/_ref/%s_gpu is not built (make -C oracle, needs /root/reference)" % prog)
    ctl = open(os.path.join(helpers.GOLDEN, "ctl", ctl_name + ".ctl")).read()
    ctl = ctl.replace("../data/", DATA + "/").replace("../ctl/", os.path.join(helpers.GOLDEN, "ctl") + "/")
    ctl += "\noutfile = mlc\nnoisy = 3\ngetSE = 0\nRateAncestor = 0\n" + extra
    if "clock = 2" in extra:      # the local clock wants its rate classes marked in the tree: one clade with a rate of its own
        tree = open(os.path.join(DATA, "lysozymeSmall.rooted.trees")).read().replace("((3,4),5)", "((3,4) #1,5) #1")
        assert "#1" in tree
        ctl += "treefile = local.trees\n"
    res = {}
    for tag, e, env in (("gpu", exe, dict(os.environ, PAML_AMD_ANNOUNCE="1")), ("cpu", cpu, None)):
        d = tmp_path / tag
        d.mkdir()
        (d / (prog + ".ctl")).write_text(ctl)
        if "clock = 2" in extra:
            (d / "local.trees").write_text(tree)
        t0 = time.perf_counter()
        for _ in range(int(os.environ.get("PAML_AMD_TEST_REPEAT", "1")) if tag == "gpu" else 1):      # (stress runs: the same search many times)
            r = run_program([e, prog + ".ctl"], d, 50, env=env)
        out = r.stdout.decode(errors="replace")
        assert r.returncode == 0, out[-3000:]
        lnl = [float(m.group(1)) for m in re.finditer(r"lnL\(ntime:\s*\d+\s+np:\s*\d+\):\s+(-?\d+\.\d+)", (d / "mlc").read_text())]
        res[tag] = (lnl, out, time.perf_counter() - t0)
    assert "engine behind com.plfun" in res["gpu"][1], res["gpu"][1][-2000:]
    assert len(res["gpu"][0]) == len(res["cpu"][0]) == 1 and abs(res["gpu"][0][0] - res["cpu"][0][0]) <= 5e-5, (res["gpu"][0], res["cpu"][0])
    print("\n%s %s %s: lnL %.6f, engine %.2f s, unmodified program %.2f s" % (prog, ctl_name, extra.strip(), res["gpu"][0][0], res["gpu"][2], res["cpu"][2]))


def test_batched_gradient_can_be_switched_off(tmp_path):
    need_binaries()
    lnl, lnf, nfun, dt, out = run(REF_GPU, HIV_CTL.replace("NSsites = 0 2", "NSsites = 0"), tmp_path / "gpu", env=dict(os.environ, PAML_AMD_NO_BATCH_GRADIENT="1", PAML_AMD_GRADIENT_CHECK="1"))
    assert abs(lnl[0] - (-1137.688190)) <= TOL and "gradient check" not in out


def test_hiv_site_models_through_the_patched_reference_are_fast(tmp_path):
    """HIV NSsites = 0 2 through codeml_gpu: with the rate matrices decomposed on the device only when they changed, and the class table
    / frequencies / per-pattern values moved only when needed, the reference's own ming2 spends its time in its own code."""
    need_binaries()
    lnl, lnf, nfun, dt, out = run(REF_GPU, HIV_CTL, tmp_path / "gpu")
    assert abs(lnl[0] - (-1137.688190)) <= TOL and abs(lnl[1] - (-1106.445004)) <= TOL, lnl
    print("\nHIV NSsites 0 2 through codeml_gpu: %.2f s (%s lfun)" % (dt, nfun))
    assert dt < 1.5          # (gradients batched through integration/tools_gradient_seam.patch; the starting values are the reference's random ones)
