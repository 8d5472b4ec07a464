"""Plain-array description of one likelihood problem (tree + tips + model inputs), the Python mirror
of what the C host hands to the engine ABI (include/paml_amd.h).  Field names follow the reference's
globals: com.z, com.fpatt, com.posG, com.rgene, com.pi, com.freqK, com.rK, nodes[].sons/branch/label,
com.nodeScale (codeml.c:109-147, treesub.c:7177).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

EIGEN_UVROOT, EIGEN_CIJK, EIGEN_K80, EIGEN_JC69LIKE, EIGEN_QMAT = 0, 1, 2, 3, 4
MODE_LFUN, MODE_LFUNDG = 0, 1


@dataclass
class Tree:
    n_tips: int
    n_nodes: int
    root: int
    sons: list            # list of lists, nodes[i].sons order
    branch: np.ndarray    # [n_nodes] branch length above each node
    label: np.ndarray     # [n_nodes] int branch label
    names: list | None = None

    def csr(self):
        ptr = np.zeros(self.n_nodes + 1, dtype=np.int32)
        flat = []
        for i, s in enumerate(self.sons):
            ptr[i + 1] = ptr[i] + len(s)
            flat += list(s)
        return ptr, np.asarray(flat, dtype=np.int32)

    def father(self):
        f = np.full(self.n_nodes, -1, dtype=np.int32)
        for i, s in enumerate(self.sons):
            for c in s:
                f[c] = i
        return f

    def newick(self, names=None, lengths=True):
        names = names or self.names or [str(i + 1) for i in range(self.n_tips)]

        def rec(i):
            s = names[i] if i < self.n_tips and not self.sons[i] else "(" + ", ".join(rec(c) for c in self.sons[i]) + ")"
            if i != self.root and lengths:
                s += ": %.6f" % self.branch[i]
            return s
        return rec(self.root) + ";"


def parse_newick(text: str, names: list | None = None) -> Tree:
    """Newick -> Tree with the reference's numbering (ReadTreeN treesub.c:3048-3216): tips keep their
    sequence order (names looked up in `names`, or 1-based integers), internal nodes are numbered from
    n_tips upward in order of '(' appearance; '#k' after a node sets its label."""
    s = text.strip()
    s = s[: s.index(";")] if ";" in s else s
    # first pass: count tips
    toks = []
    i = 0
    while i < len(s):
        c = s[i]
        if c in "(),":
            toks.append(c)
            i += 1
        elif c.isspace():
            i += 1
        else:
            j = i
            while j < len(s) and s[j] not in "(),":
                j += 1
            toks.append(s[i:j].strip())
            i = j
    tipnames = []
    prev = None
    for t in toks:
        if t not in "(),":
            if prev in ("(", ",", None):
                tipnames.append(t)
        prev = t
    n_tips = len(tipnames)
    n_nodes_max = 2 * n_tips
    sons = [[] for _ in range(n_nodes_max)]
    branch = np.zeros(n_nodes_max)
    label = np.zeros(n_nodes_max, dtype=np.int32)
    next_internal = [n_tips]
    stack = []
    root = None
    last = None    # node whose attributes may follow

    def attrs(node, txt):
        # "name: 0.1 #1" pieces
        if ":" in txt:
            txt, rest = txt.split(":", 1)
            rest = rest.strip()
            num = rest.split("#")[0].split("$")[0].strip()
            if num:
                branch[node] = float(num)
            if "#" in rest:
                label[node] = int(float(rest.split("#")[1].split()[0]))
        if "#" in txt:
            label[node] = int(float(txt.split("#")[1].split()[0]))
        return txt.split("#")[0].strip()

    def tip_index(nm):
        if names is not None and nm in names:
            return names.index(nm)
        return int(nm.lstrip("tT")) - 1      # synthetic data use names t1..tN

    prev = None
    for t in toks:
        if t == "(":
            node = next_internal[0]
            next_internal[0] += 1
            if stack:
                sons[stack[-1]].append(node)
            else:
                root = node
            stack.append(node)
        elif t == ")":
            last = stack.pop()
        elif t == ",":
            pass
        else:
            if prev == ")":
                attrs(last, t)
            else:
                nm = t.split(":")[0].split("#")[0].strip()
                node = tip_index(nm)
                attrs(node, t)
                sons[stack[-1]].append(node)
        prev = t
    n_nodes = next_internal[0]
    return Tree(n_tips, n_nodes, root, sons[:n_nodes], branch[:n_nodes].copy(), label[:n_nodes].copy(),
                names=list(names) if names is not None else None)


def balanced_tree(n_tips: int, tip_len=0.1, int_len=0.05) -> Tree:
    """Unrooted 'balanced-ish' tree with a trifurcating root (SURVEY §8d recipe): the tips are split
    ~1/2, 1/4, 1/4 under the root and each part is a balanced binary tree."""
    a = n_tips // 2
    b = (n_tips - a) // 2
    parts = [a, b, n_tips - a - b]
    sons = [[] for _ in range(2 * n_tips)]
    nxt = [n_tips]
    tip = [0]

    def build(k):
        if k == 1:
            t = tip[0]
            tip[0] += 1
            return t
        node = nxt[0]
        nxt[0] += 1
        l = build((k + 1) // 2)
        r = build(k // 2)
        sons[node] = [l, r]
        return node

    root = nxt[0]
    nxt[0] += 1
    sons[root] = [build(p) for p in parts if p > 0]
    n_nodes = nxt[0]
    branch = np.full(n_nodes, int_len)
    branch[:n_tips] = tip_len
    branch[root] = 0
    return Tree(n_tips, n_nodes, root, sons[:n_nodes], branch, np.zeros(n_nodes, dtype=np.int32))


def set_node_scale(tree: Tree, every: int) -> np.ndarray:
    """com.nodeScale as SetNodeScale marks it (treesub.c:7177-7197): post-order tip count, mark a
    non-root node when the running count exceeds `every` (100 nuc / 15 codon / 50 aa)."""
    flags = np.zeros(tree.n_nodes, dtype=np.uint8)

    def rec(i):
        d = 0
        for c in tree.sons[i]:
            d += rec(c) if tree.sons[c] else 1
        if i != tree.root and d > every:
            flags[i] = 1
            d = 1
        return d
    rec(tree.root)
    return flags


@dataclass
class Problem:
    n: int
    tree: Tree
    z: np.ndarray                 # uint8 [n_tips, n_patt]
    weights: np.ndarray           # float64 [n_patt]
    pi: np.ndarray                # [n_pi, n]
    eigen: list                   # list of dicts: {"kind":..., "U","V","Root"} | {"kind":CIJK,"Cijk","Root","nR"} | {"kind":K80,"kappa"}
    mode: int = MODE_LFUN
    freqK: np.ndarray = field(default_factory=lambda: np.ones(1))
    rate: np.ndarray = field(default_factory=lambda: np.ones(1))
    eigen_of: np.ndarray | None = None   # int32 [n_genes, K, n_labels]
    qfactor: np.ndarray | None = None    # [K, n_labels]
    cleandata: int = 1
    n_chara: np.ndarray | None = None    # int32 [n_codes]
    chara_map: np.ndarray | None = None  # uint8 [n_codes, n]
    gene_off: np.ndarray | None = None   # int32 [n_genes+1]
    gene_rate: np.ndarray | None = None  # [n_genes]
    scale_node: np.ndarray | None = None # uint8 [n_nodes]
    rate_per_gene: bool = False          # Malpha: rate is [n_genes, K] (a gamma shape per gene)

    def __post_init__(self):
        self.z = np.ascontiguousarray(self.z, dtype=np.uint8)
        self.weights = np.ascontiguousarray(self.weights, dtype=np.float64)
        self.pi = np.ascontiguousarray(np.atleast_2d(self.pi), dtype=np.float64)
        self.freqK = np.ascontiguousarray(self.freqK, dtype=np.float64)
        self.rate = np.ascontiguousarray(self.rate, dtype=np.float64)
        K = self.K
        if self.gene_off is None:
            self.gene_off = np.array([0, self.n_patt], dtype=np.int32)
        self.gene_off = np.ascontiguousarray(self.gene_off, dtype=np.int32)
        G = self.n_genes
        if self.gene_rate is None:
            self.gene_rate = np.ones(G)
        self.gene_rate = np.ascontiguousarray(self.gene_rate, dtype=np.float64)
        n_labels = int(self.tree.label.max()) + 1
        if self.eigen_of is None:
            self.eigen_of = np.zeros((G, K, n_labels), dtype=np.int32)
        self.eigen_of = np.ascontiguousarray(self.eigen_of, dtype=np.int32).reshape(G, K, -1)
        if self.qfactor is None:
            self.qfactor = np.ones((K, self.eigen_of.shape[2]))
        self.qfactor = np.ascontiguousarray(self.qfactor, dtype=np.float64).reshape(K, -1)
        if self.n_chara is None:
            self.n_chara = np.ones(self.n, dtype=np.int32)
            self.chara_map = np.zeros((self.n, self.n), dtype=np.uint8)
            self.chara_map[:, 0] = np.arange(self.n)
        self.n_chara = np.ascontiguousarray(self.n_chara, dtype=np.int32)
        self.chara_map = np.ascontiguousarray(self.chara_map, dtype=np.uint8)
        if self.scale_node is not None:
            self.scale_node = np.ascontiguousarray(self.scale_node, dtype=np.uint8)

    @property
    def n_patt(self):
        return self.z.shape[1]

    @property
    def K(self):
        return len(self.freqK)

    @property
    def n_genes(self):
        return len(self.gene_off) - 1

    @property
    def n_labels(self):
        return self.eigen_of.shape[2]

    @property
    def n_codes(self):
        return len(self.n_chara)

    def slice_patterns(self, lo: int, hi: int) -> "Problem":
        """Contiguous pattern shard [lo, hi) (multi-GPU sharding).  Gene boundaries stay where they are in the global
        range: the shard holds the part of every gene inside it (possibly none of it)."""
        import copy
        p = copy.copy(self)
        p.z = np.ascontiguousarray(self.z[:, lo:hi])
        p.weights = np.ascontiguousarray(self.weights[lo:hi])
        p.gene_off = np.clip(np.asarray(self.gene_off, dtype=np.int64) - lo, 0, hi - lo).astype(np.int32)
        return p
