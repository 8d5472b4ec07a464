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


def test_host_library_exports_match_its_header():
    """libpamlh.so (the C host above the ABI) exports every function include/pamlh.h declares."""
    from paml_amd import hostlib
    L = hostlib.lib()
    hdr = open(os.path.join(helpers.REPO, "include", "pamlh.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(pamlh_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 50
    missing = [n for n in declared if not hasattr(L, n)]
    assert not missing, missing


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


# ---------------------------------------------------------------------------------------------------------------------
# Static check of the specialised kernel's schedule (jit.h): the generated source is a straight line of building-block
# calls whose correctness rests on bookkeeping done at generation time — which ring buffer a block sits in, how many
# DMA pieces may still be in flight at each s_waitcnt, which barrier separates the last reader of a buffer from the
# DMA that overwrites it.  The checker below replays the emitted statements (prologue + three trips round the tile
# loop) against an explicit model of the in-order vector-memory queue, the ring and the two tip-code buffers.
# ---------------------------------------------------------------------------------------------------------------------
class _Sched:
    def __init__(self, src):
        self.zp = int(re.search(r"JIT2_PROLOGUE\((\d+)\)", src).group(1))
        self.zb = 1 if "#define JIT_ZB 1" in src else 2      # one code block: replaced between tiles
        self.nblk = int(re.search(r"JIT2_ADVANCE\((\d+)\)", src).group(1))
        body = src[src.index("JIT2_PROLOGUE"):]
        head, loop = body.split("for (;; ptile = 0) {", 1)
        loop = loop[:loop.rindex("if (!has_next) break;")]
        self.head, self.loop = head, loop
        self.fifo = []                 # in-flight DMA pieces of this thread, oldest first: ("b", G) or ("z", tile)
        self.landed = {}               # item -> pieces complete (this thread); visible to the others after a barrier
        self.visible = set()           # items every wave may read
        self.pending_vis = set()
        self.epoch = 0                 # barriers passed
        self.buf_owner = {}            # ring buffer -> global block number of the newest DMA into it
        self.last_read = {}            # global block / ("z", buffer) -> epoch of its latest read
        self.zbuf_owner = {}
        self.tile = -1                 # tile the loop body is working on; blocks are numbered G = tile * nblk + J
        self.zsel = self.zb - 1
        self.pieces = {}

    def G(self, j):
        return self.tile * self.nblk + j

    def issue_piece(self, j):
        g = self.G(j)
        b = g & 3
        if self.buf_owner.get(b) != g:      # first piece of a new block into this buffer
            old = self.buf_owner.get(b)
            if old is not None:
                assert old == g - 4, "ring slot %d: block %d overwrites %d" % (b, g, old)
                assert self.last_read.get(old, -1) < self.epoch, "block %d overwritten while block %d may still be read" % (g, old)
            self.buf_owner[b] = g
            self.pieces[g] = 0
        self.pieces[g] += 1
        self.fifo.append(("b", g))

    def issue_z(self):
        zb = (self.zsel ^ 1) & (self.zb - 1)
        assert self.last_read.get(("z", zb), -1) < self.epoch, "tip codes overwritten while still being read"
        self.zbuf_owner[zb] = self.tile + 1
        for _ in range(self.zp):
            self.fifo.append(("z", self.tile + 1))
        self.pieces[("z", self.tile + 1)] = self.zp

    def wait(self, n):
        while len(self.fifo) > n:
            it = self.fifo.pop(0)
            key = it[1] if it[0] == "b" else it
            self.landed[key] = self.landed.get(key, 0) + 1
            if self.landed[key] == self.pieces[key]:
                self.pending_vis.add(key)

    def barrier(self):
        self.visible |= self.pending_vis
        self.pending_vis = set()
        self.epoch += 1

    def read_block(self, j, what):
        g = self.G(j)
        assert g in self.visible, "%s reads block %d before it is known complete" % (what, g)
        assert self.buf_owner.get(g & 3) == g, "%s reads block %d but its ring slot holds %s" % (what, g, self.buf_owner.get(g & 3))
        self.last_read[g] = self.epoch

    def read_codes(self, text):
        for kind in re.findall(r"JIT2_(N?CODE)\(", text):
            zb = (self.zsel ^ 1) & (self.zb - 1) if kind == "NCODE" else self.zsel
            t = self.tile + 1 if kind == "NCODE" else self.tile
            assert self.zbuf_owner.get(zb) == t and ("z", t) in self.visible, "tip codes of tile %d read before they arrived" % t
            self.last_read[("z", zb)] = self.epoch

    def side(self, text, reading):
        for m in re.finditer(r"JIT2_PIECE_N?(?:[TP]\((\d+), \d+, \d\)|PC\((\d+), \d+\))", text):
            j = int(m.group(1) or m.group(2))
            assert (self.G(j) & 3) not in [(self.G(r) & 3) for r in reading if r != j], "refill of block %d lands in a buffer this step reads" % j
            self.issue_piece(j)

    def run(self, text):
        for line in text.split("\n"):
            line = line.strip()
            if line.startswith("JIT2_ADVANCE"):
                self.tile += 1
                self.zsel ^= self.zb - 1
                continue
            m = re.match(r"jit_matvec_tip2<(\d+), (?:true|false), \d+, \d+>\(JIT2_BUF\((\d+)\), lane, \w+, \w+, JIT2_BUF\((\d+)\), (\S+ \S+), JIT2_BUF\((\d+)\), (\S+ \S+), q, \w+, (.*)\);(?: \})?$", line)
            if m:
                mid, jp, ja, jb = int(m.group(1)), int(m.group(2)), int(m.group(3)), int(m.group(5))
                self.read_block(jp, "fused matmul")
                self.side(m.group(7), [jp, ja, jb])
                self.wait(mid)
                self.barrier()
                self.read_block(jp, "fused matmul")
                self.read_block(ja, "fused tip gather")
                self.read_block(jb, "fused tip gather")
                self.read_codes(m.group(4) + m.group(6))
                continue
            m = re.match(r"jit_matvec<(?:true|false), \d+, \d+>\(JIT2_BUF\((\d+)\), lane, \w+, \w+, (.*)\);(?: \})?$", line)
            if m:
                self.read_block(int(m.group(1)), "matmul")
                self.side(m.group(2), [int(m.group(1))])
                continue
            if line.startswith("jit_tip") or line.startswith("jit_init_tip") or line.startswith("jit_root_lds"):
                for j in re.findall(r"JIT2_BUF\((\d+)\)", line):
                    self.read_block(int(j), "tip gather")
                self.read_codes(line)
                continue
            # plain statements, possibly several per line
            for tok in re.finditer(r"JIT_WAIT\((\d+)\)|__syncthreads\(\)|JIT_SYNC\(\)|JIT2_ISSUE_Z\(\d+\)|JIT2_PIECE_N?(?:[TP]\((\d+), \d+, \d\)|PC\((\d+), \d+\))", line):
                t = tok.group(0)
                if t.startswith("JIT_WAIT"):
                    self.wait(int(tok.group(1)))
                elif t.startswith("__sync") or t.startswith("JIT_SYNC"):
                    self.barrier()
                elif t.startswith("JIT2_ISSUE_Z"):
                    self.issue_z()
                else:
                    self.issue_piece(int(tok.group(2) or tok.group(3)))

    def check(self, trips=3):
        self.run(self.head)
        for _ in range(trips):
            self.run(self.loop)
        assert self.tile == trips - 1
        return self


@pytest.mark.parametrize("shape", ["balanced16", "balanced8", "hiv", "caterpillar", "random23", "scaled", "random120", "random200", "balanced128"])
def test_jit_schedule_is_consistent(lib_path, shape):
    from paml_amd.problem import balanced_tree
    scale = None
    if shape.startswith("balanced"):
        tree = balanced_tree(int(shape[8:]))
    elif shape == "hiv":
        tree = helpers.problem_from_golden(helpers.load_golden("hiv_m0")).tree
    elif shape == "caterpillar":
        from paml_amd.problem import parse_newick
        s = "(t1:0.1,t2:0.1)"
        for i in range(3, 12):
            s = "(%s:0.05,t%d:0.1)" % (s, i)
        tree = parse_newick("(%s:0.05,t12:0.1,t13:0.1);" % s)
    elif shape in ("random120", "random200"):      # beyond 95 tips: one tip-code block, replaced between tiles
        tree = helpers.random_problem(61, int(shape[6:]), 10, seed=7).tree
    else:
        pb = helpers.random_problem(61, 23, 10, seed=5, scale_every=6 if shape == "scaled" else None)
        tree, scale = pb.tree, pb.scale_node
    src = engine.debug_jit(tree, scale_node=scale, compile=False)
    assert "prune_jit" in src and "#error" not in src
    assert ("#define JIT_ZB 1" in src) == (tree.n_tips > 95)
    # the straight-line walk is cut into basic blocks, a never-taken branch every eighth op (jit_split_mode: what large trees need) — and one full build
    n_ops = len(engine.debug_program(tree, scale_node=scale)[0])
    assert "// JIT_BIG" not in src and src.count("JIT_SPLIT()") == (n_ops - 1) // 8      # (small programs too: the shorter blocks schedule slightly better)
    assert ("jit_spill(" in src) == (shape == "balanced128") and src.count("jit_spill(") == src.count("jit_mul_mem(")   # deep stacks spill
    s = _Sched(src).check()
    assert s.nblk >= 4
    # every operand block of a tile is consumed exactly once per trip: 2 tips + branches below internal nodes
    n_int_branches = tree.n_nodes - tree.n_tips - 1
    assert s.nblk == tree.n_tips + n_int_branches


def test_jit_schedule_checker_catches_broken_schedules(lib_path):
    """The checker itself: a wait that lets one more DMA piece fly, or a missing barrier, must be reported."""
    from paml_amd.problem import balanced_tree
    src = engine.debug_jit(balanced_tree(16), compile=False)
    _Sched(src).check()
    m = next(m for m in re.finditer(r"JIT_WAIT\((\d+)\)", src[src.index("for (;; ptile = 0)"):]) if int(m.group(1)) >= 3)      # a wait in the loop that leaves pieces in flight
    i, n = src.index("for (;; ptile = 0)") + m.start(), int(m.group(1))
    with pytest.raises(AssertionError):      # ... relaxed by more than a block's pieces
        _Sched(src[:i] + "JIT_WAIT(%d)" % (n + 5) + src[i + len(m.group(0)):]).check()
    j = src.index("JIT_SYNC();", src.index("for (;; ptile = 0)"))
    k = src.index("JIT_SYNC();", j + 1)
    with pytest.raises(AssertionError):
        _Sched(src[:k] + src[k + len("JIT_SYNC();"):]).check()


@pytest.mark.parametrize("n_states", [4, 5])
def test_valu_jit_source_compiles_for_gfx950(lib_path, n_states):
    """The specialised one-pattern-per-lane kernel (jit_generate_valu) is generated and hiprtc-compiled without a GPU."""
    pb = helpers.random_problem(4, 19, 10, seed=3, scale_every=5)
    src = engine.debug_jit(pb.tree, scale_node=pb.scale_node, compile=True, n_states=n_states)
    assert "JV_PROLOGUE(%d)" % n_states in src and src.count("jv_matvec<N>") == pb.tree.n_nodes - pb.tree.n_tips - 1
    assert src.count("jv_scale<N>") == int(pb.scale_node.sum())


@pytest.mark.parametrize("n_tips", [6, 16, 32])
def test_m20_jit_source_compiles_for_gfx950(lib_path, n_tips):
    """The 20-state matrix-core kernel (jit_generate_m20) is generated and hiprtc-compiled without a GPU: the walk over a unit appears
    twice — two pattern groups sharing every operand fetch, and the single-group copy the last units of a workgroup's range run as half
    units — with one product per internal branch in each."""
    from paml_amd.problem import balanced_tree
    t = balanced_tree(n_tips)
    src = engine.debug_jit(t, compile=True, n_states=20, fused=(4, 20))
    n_mm = n_tips - 3
    assert src.count("m20h_matvec2x<") == n_mm and src.count("m20h_matvec1x<") == n_mm      # (<current / next operands from global memory>)
    assert "M20_HALF_OF" in src and "if (half < 0) {" in src
    assert src.count("m20_root(") == 3      # two groups + the single-group copy


def test_product_paths_fail_loudly_without_a_gpu(lib_path):
    """No CPU fallback anywhere in the product: on a host without a HIP device (this test's container) engine creation, the C
    host's evaluation and the device pattern compression all return an error instead of computing something else."""
    import ctypes as C
    import subprocess
    from paml_amd import hostlib
    from paml_amd.engine import Engine, EngineError, compress_patterns
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("a GPU is visible: nothing to check here")
    except ImportError:
        pass
    with pytest.raises(EngineError, match="paml_amd_create failed"):
        Engine(61, 8, 100)
    with pytest.raises(EngineError, match="compress_patterns failed"):
        compress_patterns(np.zeros((3, 10), dtype=np.uint8))
    a = hostlib.Analysis(os.path.join(helpers.GOLDEN, "ctl", "hiv_ns0.ctl"), "codeml")
    with pytest.raises(RuntimeError, match="no GPU"):
        a.eval_gpu(a.default_x())
    assert a.plfun(a.default_x()) == 1e300
    r = subprocess.run([hostlib.DRIVER_PATH, "codeml", os.path.join(helpers.GOLDEN, "ctl", "hiv_ns0.ctl")], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "no GPU" in r.stderr


def test_plfun_seam_example_compiles_and_links(tmp_path):
    """paml_amd/host/examples/plfun_seam.c — the assignment to com.plfun a maintainer would make — builds against include/pamlh.h and
    the shared libraries; without a GPU its one call of the objective function reports the missing device."""
    import subprocess
    from paml_amd import hostlib
    lib = os.path.dirname(hostlib.LIB_PATH)
    exe = str(tmp_path / "plfun_seam")
    src = os.path.join(os.path.dirname(lib), "host", "examples", "plfun_seam.c")
    r = subprocess.run(["cc", "-O2", "-Wall", "-I" + os.path.join(helpers.REPO, "include"), src, "-L" + lib, "-lpamlh", "-lpaml_amd", "-lm",
                        "-Wl,-rpath," + lib, "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    run = subprocess.run([exe, "codeml", os.path.join(helpers.GOLDEN, "ctl", "hiv_ns0.ctl")], capture_output=True, text=True, timeout=120)
    try:
        import torch
        gpu = torch.cuda.is_available()
    except ImportError:
        gpu = False
    if gpu:
        assert run.returncode == 0 and "-lnL =" in run.stdout
    else:
        assert run.returncode == 1 and "no GPU" in run.stderr


def test_reference_binding_patches_apply_and_link():
    """integration/{codeml,baseml}_plfun.patch against the reference sources where they lie (build container only): they apply without
    fuzz or rejects, touch nothing outside `#ifdef PAML_AMD`, and the recipe of oracle/Makefile links the patched programs against
    libpaml_amd.so (running them needs the GPU: tests/test_reference_binding_gpu.py)."""
    import shutil
    import subprocess
    ref = "/root/reference/src"
    if not os.path.isdir(ref) or not shutil.which("patch"):
        pytest.skip("the reference sources are not here")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for prog in ("codeml", "baseml"):
        patch = os.path.join(repo, "integration", prog + "_plfun.patch")
        r = subprocess.run(["patch", "--dry-run", "-o", "/dev/null", os.path.join(ref, prog + ".c"), patch], capture_output=True, text=True)
        assert r.returncode == 0 and "fuzz" not in r.stdout and "FAILED" not in r.stdout, r.stdout + r.stderr
        added = [ln[1:] for ln in open(patch) if ln.startswith("+") and not ln.startswith("+++")]
        removed = [ln for ln in open(patch) if ln.startswith("-") and not ln.startswith("---")]
        assert not removed, "the patch only adds lines"
        depth, outside = 0, []
        for ln in added:
            t = ln.strip()
            if t.startswith("#ifdef PAML_AMD"):
                depth += 1
            elif t.startswith("#endif") and depth:
                depth -= 1
            elif t.startswith("#else") and depth:
                pass
            elif depth == 0 and t:
                outside.append(ln)
        assert not outside, outside[:3]
        exe = os.path.join(repo, "oracle", "_ref", prog + "_gpu")
        assert subprocess.run(["make", "-C", os.path.join(repo, "oracle"), "_ref/" + prog + "_gpu"], capture_output=True).returncode == 0
        assert os.access(exe, os.X_OK)
        ldd = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
        assert "libpaml_amd.so" in ldd and "not found" not in ldd.split("libpaml_amd.so")[1].split("\n")[0]


def test_create_flags_of_the_python_mirror_are_the_headers():
    """paml_amd/engine.py restates the create flags of include/paml_amd.h (KEEP_PARTIALS, JIT, SHARD)."""
    import re
    from paml_amd import engine
    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "paml_amd.h")).read()
    for name, value in (("PAML_AMD_KEEP_PARTIALS", engine.KEEP_PARTIALS), ("PAML_AMD_JIT", engine.JIT), ("PAML_AMD_SHARD", engine.SHARD)):
        m = re.search(r"\b%s\s*=\s*(\d+)" % name, text)
        assert m and int(m.group(1)) == value, name


def test_cooperative_per_tree_kernel_is_generated_and_compiles_for_gfx950(tmp_path, monkeypatch):
    """jit.h: jit_generate_coop — the small-data cooperative kernel unrolled for a tree, reduction inside.  hiprtc cross-compiles without a
    GPU: trees with scaling nodes and a polytomy, 61 and 20 states (the 20-state source leaves the zero-padded k-blocks out), each gives
    one code object; the numerics are the GPU tests' (bit-equality with the interpreter kernels)."""
    import glob
    monkeypatch.setenv("PAML_AMD_PREBUILD_COOP", "1")
    for n, kw in ((61, dict(scale_every=3)), (61, dict(polytomy=True)), (20, {})):
        pb = helpers.random_problem(n, 9, 40, K=1, seed=5 + n, **kw)
        d = tmp_path / ("n%d_%s" % (n, "_".join(kw) or "plain"))
        monkeypatch.setenv("PAML_AMD_JIT_DUMP", str(d) + ".hip")
        engine.jit_prebuild(pb.tree, n, n, K=1, n_patt_global=40, scale_node=pb.scale_node, directory=str(d))
        assert len(glob.glob(str(d / "*.hsaco"))) == 1
        src = open(str(d) + ".hip").read()
        assert "coopj_finish(a," in src and src.count("COOPJ_MATVEC(") == sum(1 for o in engine.debug_program(pb.tree, pb.scale_node)[0] if o[0] in (4, 5))
        assert ("#define COOPJ_NP 3" in src) == (n == 20)
