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


def run(exe, ctl, d, env=None, ctl_name="codeml.ctl"):
    d.mkdir()
    (d / ctl_name).write_text(ctl % {"data": DATA})
    t0 = time.perf_counter()
    r = subprocess.run([exe, ctl_name], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, input=b"\n" * 20, timeout=1500, env=env)
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
