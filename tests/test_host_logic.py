"""CPU-side tests (no GPU): the C-ABI library loads and exports every symbol the header declares, the
engine refuses to run without a device (no CPU fallback), and the host-side tree flattening
(paml_amd/csrc/program.h) — interpreted here op by op in numpy — reproduces the oracle's recursion."""
import ctypes
import os
import re

import numpy as np
import pytest

import helpers
import oracle
from paml_amd import engine
from paml_amd.problem import set_node_scale

OP = dict(INIT_ONES=0, INIT_TIP=1, MUL_TIP=2, PUSH=3, MATMUL=4, MATMUL_POP=5, SCALE=6, STORE=7, LOAD=8, ROOT=9, END=10,
          SET_TIP=11, SET_TIP2=12, MUL_TIP2=13)


@pytest.fixture(scope="module")
def lib_path():
    return engine.build()


def test_abi_exports_match_header(lib_path):
    L = ctypes.CDLL(lib_path)
    hdr = open(os.path.join(helpers.REPO, "include", "paml_amd.h")).read()
    declared = sorted(set(re.findall(r"\b(paml_amd_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(L, name), "libpaml_amd.so does not export %s" % name
    assert sorted(engine.EXPORTS) == [d for d in declared]


def _gpu_visible():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_gpu_visible(), reason="a GPU is visible")
def test_no_cpu_fallback(lib_path):
    with pytest.raises(engine.EngineError):
        engine.Engine(4, 5, 10)


def interpret(pb, ops, iclass=0, clean_partials=None):
    """Run the flattened program in numpy for one class; returns (f_h, lnscale, stored partials)."""
    n, npatt = pb.n, pb.n_patt
    t = pb.tree
    cur = None
    stack = {}
    lnscale = np.zeros(npatt)
    stored = {}

    def P(node):
        return oracle.pmat_branch(pb, 0, iclass, node)

    def tipfac(node):
        Pm = P(node)
        cols = np.stack([Pm[:, pb.chara_map[c, :pb.n_chara[c]]].sum(axis=1) for c in range(pb.n_codes)], axis=0)
        return cols[pb.z[node]]                         # [npatt, n]

    for code, a, b, c in ops:
        if code == OP["INIT_ONES"]:
            cur = np.ones((npatt, n))
        elif code == OP["INIT_TIP"]:
            cur = np.zeros((npatt, n))
            if pb.cleandata:
                cur[np.arange(npatt), pb.z[a]] = 1
        elif code == OP["MUL_TIP"]:
            cur = cur * tipfac(a)
        elif code == OP["SET_TIP"]:
            cur = tipfac(a)
        elif code == OP["SET_TIP2"]:
            cur = tipfac(a) * tipfac(b)
        elif code == OP["MUL_TIP2"]:
            cur = (cur * tipfac(a)) * tipfac(b)
        elif code == OP["SET_TIP"]:
            cur = tipfac(a)
        elif code == OP["SET_TIP2"]:
            cur = tipfac(a) * tipfac(b)
        elif code == OP["MUL_TIP2"]:
            cur = (cur * tipfac(a)) * tipfac(b)
        elif code == OP["PUSH"]:
            stack[b] = cur
        elif code in (OP["MATMUL"], OP["MATMUL_POP"]):
            pop, push = (b & 0xff) - 1, ((b >> 8) & 0xff) - 1
            assert (pop >= 0) == (code == OP["MATMUL_POP"])
            val = cur @ P(a).T
            if pop >= 0:
                val = stack.pop(pop) * val
            if push >= 0:
                stack[push] = val
                cur = None
            else:
                cur = val
        elif code == OP["SCALE"]:
            mx = cur.max(axis=1)
            small = mx < 1e-300
            fac = np.where(small, -800.0, np.log(np.where(small, 1.0, mx)))
            cur = np.where(small[:, None], 1.0, cur / np.where(small, 1.0, mx)[:, None])
            lnscale += fac
        elif code == OP["STORE"]:
            stored[a] = cur.copy()
        elif code == OP["LOAD"]:
            cur = clean_partials[a].copy()
        elif code == OP["ROOT"]:
            f = cur @ pb.pi[0]
        elif code == OP["END"]:
            break
    assert not stack
    return f, lnscale, stored


@pytest.mark.parametrize("n_tips,seed,every,poly", [(5, 1, None, False), (16, 2, None, False), (33, 3, 4, False),
                                                     (12, 4, 3, True), (64, 5, 6, False)])
def test_program_reproduces_recursion(lib_path, n_tips, seed, every, poly):
    pb = helpers.random_problem(4, n_tips, 40, K=1, seed=seed, scale_every=every, polytomy=poly, ambiguity=(seed % 2 == 0))
    ops, depth = engine.debug_program(pb.tree, pb.scale_node, keep=True)
    assert depth <= int(np.ceil(np.log2(n_tips))) + 1
    # every MATMUL links to the next one (prefetch chain)
    mm = [o for o in ops if o[0] in (OP["MATMUL"], OP["MATMUL_POP"])]
    assert len(mm) == pb.tree.n_nodes - pb.tree.n_tips - 1 or poly
    for cur, nxt in zip(mm, mm[1:]):
        assert cur[3] == nxt[1]
    assert mm[-1][3] == -1
    # tip prefetch chain: every tip-consuming op names the first tip of the next tip-consuming op
    tipops = [o for o in ops if o[0] in (OP["MUL_TIP"], OP["SET_TIP"], OP["SET_TIP2"], OP["MUL_TIP2"])]
    for cur, nxt in zip(tipops, tipops[1:]):
        assert cur[3] == nxt[1]
    assert tipops[-1][3] == -1
    assert not any(o[0] == OP["PUSH"] and i > 0 and ops[i - 1][0] in (OP["MATMUL"], OP["MATMUL_POP"]) for i, o in enumerate(ops))
    f, lnscale, stored = interpret(pb, ops)
    ref = oracle.evaluate(pb, want_fhk=True, want_partials=True)
    assert np.allclose(np.log(f) + lnscale, ref["fhK"][0], rtol=1e-12, atol=1e-12)
    for node, part in stored.items():
        assert np.allclose(part, ref["partials"][0, node - pb.tree.n_tips], rtol=1e-12, atol=0)


def test_program_with_clean_subtrees(lib_path):
    pb = helpers.random_problem(4, 20, 30, K=1, seed=11, scale_every=5)
    t = pb.tree
    ops, _ = engine.debug_program(t, pb.scale_node, keep=True)
    _, _, stored = interpret(pb, ops)
    father = t.father()
    clean = np.ones(t.n_nodes, dtype=np.uint8)
    node = 7
    while node != -1:
        clean[node] = 0
        node = father[node]
    ops2, _ = engine.debug_program(t, pb.scale_node, keep=True, clean=clean)
    assert any(o[0] == OP["LOAD"] for o in ops2)
    assert len(ops2) < len(ops)
    br = t.branch.copy()
    t.branch[7] *= 2.5
    f2, _, _ = interpret(pb, ops2, clean_partials=stored)
    # scale factors of clean scaled nodes come from storage on the device; here compare unscaled-equivalent lnL
    ref = oracle.evaluate(pb, want_fhk=True, want_partials=True)
    lns = np.zeros(pb.n_patt)
    if ref["scalef"] is not None:
        lns = ref["scalef"][0].sum(axis=0)
    assert np.allclose(np.log(f2) + lns, ref["fhK"][0], rtol=1e-11, atol=1e-11)
    t.branch[:] = br


def test_set_node_scale_matches_reference_rule():
    # caterpillar of 40 tips: marks appear once the running tip count exceeds `every`, never on the root
    pb = helpers.random_problem(4, 40, 4, seed=2)
    flags = set_node_scale(pb.tree, 15)
    assert flags[pb.tree.root] == 0
    assert flags.sum() >= 1
