"""ctypes binding of oracle/liboracle.so — TEST INFRASTRUCTURE ONLY.

Imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing in the
paml_amd package imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class _Eigen(C.Structure):
    _fields_ = [("kind", C.c_int), ("nR", C.c_int), ("kappa", C.c_double),
                ("U", C.c_void_p), ("V", C.c_void_p), ("Root", C.c_void_p), ("Cijk", C.c_void_p)]


class _Problem(C.Structure):
    _fields_ = [
        ("n", C.c_int), ("n_tips", C.c_int), ("n_nodes", C.c_int), ("root", C.c_int), ("n_patt", C.c_int),
        ("sons_ptr", C.c_void_p), ("sons", C.c_void_p), ("label", C.c_void_p), ("scale_node", C.c_void_p),
        ("z", C.c_void_p), ("cleandata", C.c_int), ("n_codes", C.c_int), ("n_chara", C.c_void_p),
        ("chara_map", C.c_void_p), ("weights", C.c_void_p), ("n_genes", C.c_int), ("gene_off", C.c_void_p),
        ("gene_rate", C.c_void_p), ("n_pi", C.c_int), ("pi", C.c_void_p), ("n_eigen", C.c_int),
        ("eigen", C.POINTER(_Eigen)), ("mode", C.c_int), ("K", C.c_int), ("freqK", C.c_void_p),
        ("rate", C.c_void_p), ("n_labels", C.c_int), ("eigen_of", C.c_void_p), ("qfactor", C.c_void_p),
        ("branch", C.c_void_p), ("z_stride", C.c_long), ("rate_gs", C.c_int),
    ]


def build(force: bool = False) -> str:
    """Compile liboracle.so (and, when /root/reference exists, oracle/_ref) via oracle/Makefile."""
    so = os.path.join(_HERE, "liboracle.so")
    src = [os.path.join(_HERE, f) for f in ("cpu_ref.c", "cpu_ref.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_eval.restype = C.c_double
        _LIB.orc_eval.argtypes = [C.POINTER(_Problem), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        _LIB.orc_pmat_branch.argtypes = [C.POINTER(_Problem), C.c_int, C.c_int, C.c_int, C.c_void_p]
        _LIB.orc_last_npmat.restype = C.c_long
        _LIB.orc_node_posterior.argtypes = [C.POINTER(_Problem), C.c_int, C.c_void_p]
        _LIB.orc_eval_adg.restype = C.c_double
        _LIB.orc_eval_adg.argtypes = [C.POINTER(_Problem), C.c_void_p, C.c_void_p, C.c_int]
        _LIB.orc_eval_blocked.restype = C.c_double
        _LIB.orc_eval_blocked.argtypes = [C.POINTER(_Problem), C.c_int, C.c_int]
        _LIB.orc_eval_branch.argtypes = [C.POINTER(_Problem), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    return _LIB


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class _Packed:
    """Keeps the numpy buffers alive for the lifetime of the C struct."""

    def __init__(self, pb):
        t = pb.tree
        self.keep = []

        def k(a, dt):
            a = np.ascontiguousarray(a, dtype=dt)
            self.keep.append(a)
            return a
        ptr, flat = t.csr()
        ptr, flat = k(ptr, np.int32), k(flat, np.int32)
        label = k(t.label, np.int32)
        branch = k(t.branch, np.float64)
        eig = (_Eigen * len(pb.eigen))()
        for i, e in enumerate(pb.eigen):
            eig[i].kind = int(e["kind"])
            eig[i].nR = int(e.get("nR", 0))
            eig[i].kappa = float(e.get("kappa", 0.0))
            for name in ("U", "V", "Root", "Cijk"):
                if name in e:
                    setattr(eig[i], name, _ptr(k(e[name], np.float64)))
            if "Q" in e:                      # rate-matrix kind (UNREST): the U slot carries Q
                eig[i].U = _ptr(k(e["Q"], np.float64))
        self.eig = eig
        s = _Problem()
        s.n, s.n_tips, s.n_nodes, s.root, s.n_patt = pb.n, t.n_tips, t.n_nodes, t.root, pb.n_patt
        s.sons_ptr, s.sons, s.label = _ptr(ptr), _ptr(flat), _ptr(label)
        s.scale_node = _ptr(pb.scale_node) if pb.scale_node is not None and pb.scale_node.any() else None
        s.z, s.cleandata = _ptr(pb.z), int(pb.cleandata)
        s.n_codes, s.n_chara, s.chara_map = pb.n_codes, _ptr(pb.n_chara), _ptr(pb.chara_map)
        s.weights = _ptr(pb.weights)
        s.n_genes, s.gene_off, s.gene_rate = pb.n_genes, _ptr(pb.gene_off), _ptr(pb.gene_rate)
        s.n_pi, s.pi = pb.pi.shape[0], _ptr(pb.pi)
        s.n_eigen, s.eigen = len(pb.eigen), eig
        s.mode, s.K, s.freqK, s.rate = int(pb.mode), pb.K, _ptr(pb.freqK), _ptr(pb.rate)
        s.n_labels, s.eigen_of, s.qfactor = pb.n_labels, _ptr(pb.eigen_of), _ptr(pb.qfactor)
        s.branch = _ptr(branch)
        s.rate_gs = pb.K if getattr(pb, "rate_per_gene", False) else 0
        self.s = s


def evaluate(pb, want_lnf=True, want_fhk=False, want_partials=False, nthreads=1):
    """One com.plfun call on the CPU.  Returns dict(lnL, lnf, fhK, partials, scalef, npmat)."""
    L = lib()
    pk = _Packed(pb)
    np_, K, n = pb.n_patt, pb.K, pb.n
    nint = pb.tree.n_nodes - pb.tree.n_tips
    n_scale = int(pb.scale_node.sum()) if pb.scale_node is not None else 0
    lnf = np.zeros(np_) if want_lnf else None
    fhk = np.zeros((K, np_)) if want_fhk else None
    part = np.zeros((K, nint, np_, n)) if want_partials else None
    scalef = np.zeros((K, n_scale, np_)) if (want_partials and n_scale) else None
    lnL = L.orc_eval(C.byref(pk.s), _ptr(lnf), _ptr(fhk), _ptr(part), _ptr(scalef), int(nthreads))
    return dict(lnL=lnL, lnf=lnf, fhK=fhk, partials=part, scalef=scalef, npmat=L.orc_last_npmat())


def node_posterior(pb, node):
    """Marginal posterior probabilities of the states at an internal node, [n_patt][n]."""
    pk = _Packed(pb)
    post = np.zeros((pb.n_patt, pb.n))
    rc = lib().orc_node_posterior(C.byref(pk.s), int(node), _ptr(post))
    if rc != 0:
        raise RuntimeError("orc_node_posterior failed (%d)" % rc)
    return post


def evaluate_adg(pb, MK, pose):
    """lfunAdG: +lnL of the auto-discrete-gamma chain MK[K][K] over the sites pose[ls] (site -> pattern)."""
    pk = _Packed(pb)
    MK = np.ascontiguousarray(MK, dtype=np.float64)
    pose = np.ascontiguousarray(pose, dtype=np.int32)
    return lib().orc_eval_adg(C.byref(pk.s), _ptr(MK), _ptr(pose), len(pose))


def evaluate_blocked(pb, nthreads, block=512):
    """lnL with the patterns cut into blocks spread over `nthreads` host cores (each thread walks the whole tree)."""
    pk = _Packed(pb)
    return lib().orc_eval_blocked(C.byref(pk.s), int(nthreads), int(block))


def pmat_branch(pb, gene, iclass, node):
    L = lib()
    pk = _Packed(pb)
    P = np.zeros((pb.n, pb.n))
    L.orc_pmat_branch(C.byref(pk.s), gene, iclass, node, _ptr(P))
    return P


def eval_branch(pb, node_b, t):
    """lnL(t), dlnL/dt, d2lnL/dt2 for the branch above node_b at the trial lengths t (lfuntdd restated)."""
    L = lib()
    pk = _Packed(pb)
    t = np.ascontiguousarray(np.atleast_1d(t), dtype=np.float64)
    l, dl, ddl = np.zeros(len(t)), np.zeros(len(t)), np.zeros(len(t))
    rc = L.orc_eval_branch(C.byref(pk.s), int(node_b), len(t), _ptr(t), _ptr(l), _ptr(dl), _ptr(ddl))
    if rc != 0:
        raise RuntimeError("orc_eval_branch failed (%d)" % rc)
    return l, dl, ddl
