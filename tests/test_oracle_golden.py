"""Pin the oracle (oracle/cpu_ref.c) against golden vectors produced by the unmodified reference
binaries (tests/golden/make_golden.py): lnL and every per-pattern log f_h from the `lnf` file."""
import numpy as np
import pytest

import helpers
import oracle

CASES = ["hiv_m0", "hiv_m1a", "hiv_m2a", "hiv_m7", "hiv_m8", "syn_codon_m0", "syn_nuc_gtr_g4", "brown_hky85",
         "stewart_lg_g4", "mhc_m0_scaled"]


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
