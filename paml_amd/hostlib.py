"""ctypes binding of libpamlh.so (include/pamlh.h): the C host that reads .ctl / sequence / tree files and turns a
parameter vector into engine inputs.  `load(ctl, program).problem(x)` returns the same plain-array Problem the tests
and the engine binding use, so the C host can be checked against the golden vectors on CPU (through the oracle, in the
tests) and on the GPU (through the engine)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from .problem import EIGEN_CIJK, EIGEN_JC69LIKE, EIGEN_K80, EIGEN_QMAT, EIGEN_UVROOT, Problem, Tree

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libpamlh.so")
DRIVER_PATH = os.path.join(_HERE, "lib", "pamlh_lnl")


def build(force=False):
    from . import engine
    engine.build()
    srcs = [os.path.join(_HERE, "host", f) for f in ("pamlh_num.c", "pamlh_io.c", "pamlh_model.c", "pamlh_opt.c", "pamlh_lnl.c", "pamlh_internal.h", "Makefile")]
    srcs.append(os.path.join(os.path.dirname(_HERE), "include", "pamlh.h"))
    if force or not (os.path.exists(LIB_PATH) and os.path.exists(DRIVER_PATH)) or \
            any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs):
        subprocess.check_call(["make", "-C", os.path.join(_HERE, "host"), "-B"], stdout=subprocess.DEVNULL)
    return LIB_PATH


_L = None


def lib():
    global _L
    if _L is None:
        C.CDLL(os.path.join(_HERE, "lib", "libpaml_amd.so"), mode=C.RTLD_GLOBAL)
        L = C.CDLL(build())
        for name, rt in [("pamlh_tips", C.c_void_p), ("pamlh_weights", C.c_void_p), ("pamlh_n_chara", C.c_void_p),
                         ("pamlh_chara_map", C.c_void_p), ("pamlh_sons_ptr", C.c_void_p), ("pamlh_sons", C.c_void_p),
                         ("pamlh_labels", C.c_void_p), ("pamlh_scale_nodes", C.c_void_p), ("pamlh_branch_order", C.c_void_p),
                         ("pamlh_branch", C.c_void_p), ("pamlh_pi", C.c_void_p), ("pamlh_freqK", C.c_void_p),
                         ("pamlh_rate", C.c_void_p), ("pamlh_eigen_of", C.c_void_p), ("pamlh_error", C.c_char_p)]:
            getattr(L, name).restype = rt
            getattr(L, name).argtypes = [C.c_void_p]
        L.pamlh_load.argtypes = [C.POINTER(C.c_void_p), C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]
        L.pamlh_load_tree.argtypes = [C.POINTER(C.c_void_p), C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.c_int]
        L.pamlh_n_trees.argtypes = [C.c_void_p]
        L.pamlh_load_with.argtypes = [C.POINTER(C.c_void_p), C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.c_char_p, C.c_int]
        L.pamlh_ctl_option.argtypes = [C.c_void_p, C.c_char_p]
        L.pamlh_ctl_option.restype = C.c_char_p
        L.pamlh_set_shard.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.pamlh_dnds.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.pamlh_free.argtypes = [C.c_void_p]
        L.pamlh_dims.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 10
        L.pamlh_default_x.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.pamlh_read_inx.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.pamlh_set_x.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.pamlh_model.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 4
        L.pamlh_eigen.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double)] + [C.POINTER(C.c_void_p)] * 4
        L.pamlh_eval_gpu.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_void_p]
        L.pamlh_eval_batch_gpu.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.pamlh_bounds.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.pamlh_standard_errors.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.pamlh_optimize.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_double, C.c_int, C.POINTER(C.c_int)]
        _L = L
    return _L


def _arr(ptr, dtype, n):
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(np.ctypeslib.as_ctypes_type(dtype))), shape=(n,)).copy()


def tree_comparison(lnf, w, gene_off=None, n_rep=0, seed=1):
    """pamlh_tree_comparison: lnf [n_trees][n_patt], w pattern counts -> dict of li, dli, se, pKH, pSH, pRELL (arrays) and best."""
    L = lib()
    lnf = np.ascontiguousarray(lnf, dtype=np.float64)
    w = np.ascontiguousarray(w, dtype=np.float64)
    nt, npatt = lnf.shape
    go = np.ascontiguousarray(gene_off, dtype=np.int32) if gene_off is not None else None
    out = [np.zeros(nt) for _ in range(6)]
    best = C.c_int()
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))      # noqa: E731
    L.pamlh_tree_comparison.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, C.c_void_p, C.c_int, C.c_ulonglong] + [C.POINTER(C.c_double)] * 6 + [C.POINTER(C.c_int)]
    rc = L.pamlh_tree_comparison(nt, npatt, dp(w), dp(lnf), (len(go) - 1) if go is not None else 1, go.ctypes.data if go is not None else None, n_rep, seed,
                                 *[dp(a) for a in out], C.byref(best))
    if rc != 0:
        raise RuntimeError("pamlh_tree_comparison: bad arguments")
    return dict(zip(("li", "dli", "se", "pKH", "pSH", "pRELL"), out), best=best.value)


class Analysis:
    def __init__(self, ctl_path, program="codeml", tree_index=0, overrides=None):
        L = lib()
        h = C.c_void_p()
        err = C.create_string_buffer(512)
        if L.pamlh_load_with(C.byref(h), os.fsencode(ctl_path), program.encode(), tree_index, overrides.encode() if overrides else None, err, 512) != 0:
            raise RuntimeError("pamlh_load: " + err.value.decode())
        self._h, self._L = h, L
        d = [C.c_int() for _ in range(10)]
        L.pamlh_dims(h, *[C.byref(v) for v in d])
        (self.n, self.n_tips, self.n_patt, self.n_nodes, self.root, self.n_codes, self.cleandata, self.ls, self.np,
         self.ntime) = [v.value for v in d]

    def dnds(self, x):
        """[n_branches][6] = t, N, S, omega, dN, dS (the reference's "dN & dS for each branch" table)."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        out = np.zeros((self.n_nodes - 1, 6))
        if self._L.pamlh_dnds(self._h, x.ctypes.data_as(C.POINTER(C.c_double)), out.ctypes.data_as(C.POINTER(C.c_double))) != 0:
            raise RuntimeError("pamlh_dnds: " + self._L.pamlh_error(self._h).decode())
        return out

    def set_shard(self, rank, world, comm_id=None):
        """Keep this rank's block of site patterns (pamlh_set_shard; before the first evaluation).  comm_id: the 128-byte RCCL id, or
        None for the sharding alone (CPU tests; one-GPU emulation of the ranks)."""
        buf = (C.c_ubyte * 128).from_buffer_copy(comm_id) if comm_id is not None else None
        if self._L.pamlh_set_shard(self._h, rank, world, buf) != 0:
            raise RuntimeError("pamlh_set_shard: " + self._L.pamlh_error(self._h).decode())
        d = [C.c_int() for _ in range(10)]
        self._L.pamlh_dims(self._h, *[C.byref(v) for v in d])
        self.n_patt = d[2].value

    def ctl_option(self, key):
        v = self._L.pamlh_ctl_option(self._h, key.encode())
        return v.decode() if v is not None else None

    def n_trees(self):
        return self._L.pamlh_n_trees(self._h)

    def gene_subset(self, g):
        """Mgene = 1: gene g as an analysis of its own (pamlh_gene_subset)."""
        h = C.c_void_p()
        self._L.pamlh_gene_subset.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        if self._L.pamlh_gene_subset(self._h, int(g), C.byref(h)) != 0:
            raise RuntimeError("pamlh_gene_subset failed")
        a = Analysis.__new__(Analysis)
        a._h, a._L = h, self._L
        d = [C.c_int() for _ in range(10)]
        self._L.pamlh_dims(h, *[C.byref(v) for v in d])
        (a.n, a.n_tips, a.n_patt, a.n_nodes, a.root, a.n_codes, a.cleandata, a.ls, a.np, a.ntime) = [v.value for v in d]
        return a

    def n_genes(self):
        self._L.pamlh_genes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        return self._L.pamlh_genes(self._h, None, None, None, None)

    def close(self):
        if getattr(self, "_h", None):
            self._L.pamlh_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def default_x(self):
        x = np.zeros(max(1, self.np))
        k = self._L.pamlh_default_x(self._h, x.ctypes.data_as(C.c_void_p), len(x))
        return x[:k]

    def read_inx(self):
        x = np.zeros(4096)
        k = self._L.pamlh_read_inx(self._h, x.ctypes.data_as(C.c_void_p), len(x))
        return x[:k]

    def set_x(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        if self._L.pamlh_set_x(self._h, x.ctypes.data_as(C.c_void_p), len(x)) != 0:
            raise RuntimeError("pamlh_set_x: " + self._L.pamlh_error(self._h).decode())

    def problem(self, x) -> Problem:
        """SetParameters(x) in the C host, then everything as plain arrays."""
        self.set_x(x)
        L, h = self._L, self._h
        n, nn = self.n, self.n_nodes
        m = [C.c_int() for _ in range(4)]
        L.pamlh_model(h, *[C.byref(v) for v in m])
        mode, K, n_eigen, n_labels = [v.value for v in m]
        ptr = _arr(L.pamlh_sons_ptr(h), np.int32, nn + 1)
        sons_flat = _arr(L.pamlh_sons(h), np.int32, int(ptr[-1]))
        sons = [list(sons_flat[ptr[i]:ptr[i + 1]]) for i in range(nn)]
        tree = Tree(self.n_tips, nn, self.root, sons, _arr(L.pamlh_branch(h), np.float64, nn), _arr(L.pamlh_labels(h), np.int32, nn))
        eig = []
        for i in range(n_eigen):
            kind, nR, kappa = C.c_int(), C.c_int(), C.c_double()
            ps = [C.c_void_p() for _ in range(4)]
            L.pamlh_eigen(h, i, C.byref(kind), C.byref(nR), C.byref(kappa), *[C.byref(v) for v in ps])
            e = dict(kind=kind.value)
            if kind.value == EIGEN_UVROOT:
                e.update(U=_arr(ps[0].value, np.float64, n * n).reshape(n, n), V=_arr(ps[1].value, np.float64, n * n).reshape(n, n),
                         Root=_arr(ps[2].value, np.float64, n))
            elif kind.value == EIGEN_CIJK:
                e.update(nR=nR.value, Cijk=_arr(ps[3].value, np.float64, n * n * nR.value), Root=_arr(ps[2].value, np.float64, nR.value))
            elif kind.value == EIGEN_K80:
                e.update(kappa=kappa.value)
            elif kind.value == EIGEN_QMAT:
                e.update(Q=_arr(ps[0].value, np.float64, n * n).reshape(n, n))
            eig.append(e)
        scale = _arr(L.pamlh_scale_nodes(h), np.uint8, nn)
        L.pamlh_qfactor.restype = C.c_void_p
        L.pamlh_qfactor.argtypes = [C.c_void_p]
        qf = L.pamlh_qfactor(h)
        # several genes (option G): pattern offsets, rates, one frequency vector / eigen system per gene where the model has them
        go, gr, geo, npi = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int()
        L.pamlh_genes.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_void_p)]
        G = L.pamlh_genes(h, C.byref(go), C.byref(gr), C.byref(npi), C.byref(geo))
        malpha = bool(L.pamlh_malpha(h)) and G > 1
        genes = {}
        if G > 1:
            genes = dict(gene_off=_arr(go.value, np.int32, G + 1), gene_rate=_arr(gr.value, np.float64, G))
        return Problem(qfactor=None if not qf else _arr(qf, np.float64, K * n_labels).reshape(K, n_labels), **genes,
                       n=n, tree=tree, z=_arr(L.pamlh_tips(h), np.uint8, self.n_tips * self.n_patt).reshape(self.n_tips, self.n_patt),
                       weights=_arr(L.pamlh_weights(h), np.float64, self.n_patt), pi=_arr(L.pamlh_pi(h), np.float64, n * npi.value).reshape(npi.value, n), eigen=eig,
                       mode=mode, freqK=_arr(L.pamlh_freqK(h), np.float64, K), rate=_arr(L.pamlh_rate(h), np.float64, K * (G if malpha else 1)), rate_per_gene=malpha,
                       eigen_of=(_arr(geo.value, np.int32, G * K).reshape(G, K, 1) if G > 1 else
                                 _arr(L.pamlh_eigen_of(h), np.int32, K * n_labels).reshape(1, K, n_labels)), cleandata=self.cleandata,
                       n_chara=_arr(L.pamlh_n_chara(h), np.int32, self.n_codes),
                       chara_map=_arr(L.pamlh_chara_map(h), np.uint8, self.n_codes * n).reshape(self.n_codes, n),
                       scale_node=scale if scale.any() else None)

    def eval_gpu(self, x, want_lnf=True):
        self.set_x(x)
        lnl = C.c_double()
        lnf = np.zeros(self.n_patt) if want_lnf else None
        if self._L.pamlh_eval_gpu(self._h, C.byref(lnl), None if lnf is None else lnf.ctypes.data_as(C.c_void_p)) != 0:
            raise RuntimeError("pamlh_eval_gpu: " + self._L.pamlh_error(self._h).decode())
        return lnl.value, lnf

    def eval_batch_gpu(self, xs):
        """lnL at every row of xs[n_batch][np] in one launch (pamlh_eval_batch_gpu)."""
        xs = np.ascontiguousarray(xs, dtype=np.float64)
        assert xs.ndim == 2 and xs.shape[1] == self.np
        out = np.zeros(xs.shape[0])
        if self._L.pamlh_eval_batch_gpu(self._h, xs.shape[0], xs.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)) != 0:
            raise RuntimeError("pamlh_eval_batch_gpu: " + self._L.pamlh_error(self._h).decode())
        return out

    def bounds(self):
        lo, hi = np.zeros(self.np), np.zeros(self.np)
        if self._L.pamlh_bounds(self._h, lo.ctypes.data_as(C.c_void_p), hi.ctypes.data_as(C.c_void_p)) != 0:
            raise RuntimeError("pamlh_bounds failed")
        return lo, hi

    def optimize(self, x0, max_iter=500, tol=1e-10, verbose=False):
        """Maximum-likelihood estimation from x0 (pamlh_optimize).  Returns dict(x, lnL, converged, n_eval)."""
        x = np.ascontiguousarray(x0, dtype=np.float64).copy()
        lnl, nev = C.c_double(), C.c_int()
        rc = self._L.pamlh_optimize(self._h, x.ctypes.data_as(C.c_void_p), C.byref(lnl), max_iter, tol, int(verbose), C.byref(nev))
        if rc < 0:
            raise RuntimeError("pamlh_optimize: " + self._L.pamlh_error(self._h).decode())
        return dict(x=x, lnL=lnl.value, converged=rc == 0, n_eval=nev.value)

    def optimize_minb(self, x0, e0=1e-6, verbose=0):
        """method = 1 (pamlh_optimize_minb): branch lengths one at a time by Newton steps on the branch-local derivatives, the
        other parameters by BFGS in between.  Returns dict(x, lnL, converged, n_eval, branch_calls, nodes_recomputed)."""
        from . import engine
        x = np.ascontiguousarray(x0, dtype=np.float64).copy()
        lnl, nev = C.c_double(), C.c_int()
        self._L.pamlh_optimize_minb.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.c_double, C.c_int, C.POINTER(C.c_int)]
        rc = self._L.pamlh_optimize_minb(self._h, x.ctypes.data_as(C.c_void_p), C.byref(lnl), float(e0), int(verbose), C.byref(nev))
        if rc < 0:
            raise RuntimeError("pamlh_optimize_minb: " + self._L.pamlh_error(self._h).decode())
        a, b = C.c_long(), C.c_long()
        self._L.pamlh_engine_handle.restype = C.c_void_p
        self._L.pamlh_engine_handle.argtypes = [C.c_void_p]
        L = engine.lib()
        L.paml_amd_branch_counters.argtypes = [C.c_void_p, C.POINTER(C.c_long), C.POINTER(C.c_long)]
        L.paml_amd_branch_counters(C.c_void_p(self._L.pamlh_engine_handle(self._h)), C.byref(a), C.byref(b))
        return dict(x=x, lnL=lnl.value, converged=rc == 0, n_eval=nev.value, branch_calls=a.value, nodes_recomputed=b.value)

    def lnpd_locus(self, age, rgene=1.0, rate=None, model_changed=True):
        """mcmctree's lnpD_locus on the GPU (pamlh_lnpd_locus): lnL for node ages age[n_nodes] and a locus rate / branch rates."""
        a = np.ascontiguousarray(age, dtype=np.float64)
        assert a.shape == (self.n_nodes,)
        r = None if rate is None else np.ascontiguousarray(rate, dtype=np.float64)
        lnl = C.c_double()
        self._L.pamlh_lnpd_locus.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_int, C.POINTER(C.c_double)]
        if self._L.pamlh_lnpd_locus(self._h, a.ctypes.data_as(C.c_void_p), float(rgene), None if r is None else r.ctypes.data_as(C.c_void_p),
                                    int(model_changed), C.byref(lnl)) != 0:
            raise RuntimeError("pamlh_lnpd_locus: " + self._L.pamlh_error(self._h).decode())
        return lnl.value

    def standard_errors(self, x, method=0):
        """Standard errors at the estimate x (pamlh_standard_errors): method 0 = the reference's HessianSKT2004 outer
        product of scores, method 1 = second differences of lnL; -1 marks a parameter without an estimate."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        se, H = np.zeros(self.np), np.zeros((self.np, self.np))
        if self._L.pamlh_standard_errors(self._h, x.ctypes.data_as(C.c_void_p), int(method), se.ctypes.data_as(C.c_void_p), H.ctypes.data_as(C.c_void_p)) != 0:
            raise RuntimeError("pamlh_standard_errors: " + self._L.pamlh_error(self._h).decode())
        return se, H

    def neb(self, x):
        """NEB site-class posteriors at x: (post[K][n_sites], mean_omega[n_sites]) in the order of the (cleaned) sites."""
        self.set_x(x)
        mode, K, ne, nl = (C.c_int(), C.c_int(), C.c_int(), C.c_int())
        self._L.pamlh_model(self._h, C.byref(mode), C.byref(K), C.byref(ne), C.byref(nl))
        post, mw = np.zeros((K.value, self.n_patt)), np.zeros(self.n_patt)
        self._L.pamlh_neb.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        if self._L.pamlh_neb(self._h, post.ctypes.data_as(C.c_void_p), mw.ctypes.data_as(C.c_void_p)) != 0:
            raise RuntimeError("pamlh_neb: " + self._L.pamlh_error(self._h).decode())
        ns = C.c_int()
        self._L.pamlh_pose.restype = C.c_void_p
        self._L.pamlh_pose.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        ptr = self._L.pamlh_pose(self._h, C.byref(ns))
        pose = _arr(ptr, np.int32, ns.value)
        return post[:, pose], mw[pose]

    def _is_branchsite(self):
        return self._L.pamlh_positive_classes(self._h) == 2

    def pose(self):
        """com.pose: pattern of every (cleaned) site, in site order."""
        ns = C.c_int()
        self._L.pamlh_pose.restype = C.c_void_p
        self._L.pamlh_pose.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        return _arr(self._L.pamlh_pose(self._h, C.byref(ns)), np.int32, ns.value)

    def adg_matrix(self, x):
        """Auto-discrete-gamma transition matrix MK[K][K] of the model at x (None when the model has no rho)."""
        self.set_x(x)
        m = [C.c_int() for _ in range(4)]
        self._L.pamlh_model(self._h, *[C.byref(v) for v in m])
        K = m[1].value
        self._L.pamlh_adg_matrix.restype = C.c_void_p
        self._L.pamlh_adg_matrix.argtypes = [C.c_void_p]
        ptr = self._L.pamlh_adg_matrix(self._h)
        return _arr(ptr, np.float64, K * K).reshape(K, K) if ptr else None

    def plfun(self, x):
        """-lnL at x with com.plfun's convention (pamlh_plfun)."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        self._L.pamlh_plfun.restype = C.c_double
        self._L.pamlh_plfun.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        return self._L.pamlh_plfun(self._h, x.ctypes.data_as(C.c_void_p), len(x))

    def node_posterior(self, x, node):
        """Marginal reconstruction at internal node `node` (0-based) at x: post[n_patt][n] (pamlh_node_posterior)."""
        self.set_x(x)
        post = np.zeros((self.n_patt, self.n))
        self._L.pamlh_node_posterior.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        if self._L.pamlh_node_posterior(self._h, int(node), post.ctypes.data_as(C.c_void_p)) != 0:
            raise RuntimeError("pamlh_node_posterior: " + self._L.pamlh_error(self._h).decode())
        return post

    def joint_reconstruction(self, x):
        """Best joint assignment of states to the internal nodes per pattern and its probability (pamlh_joint_reconstruction)."""
        self.set_x(x)
        ni = self.n_nodes - self.n_tips
        st, pr = np.zeros((self.n_patt, ni), dtype=np.int32), np.zeros(self.n_patt)
        self._L.pamlh_joint_reconstruction.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        if self._L.pamlh_joint_reconstruction(self._h, st.ctypes.data_as(C.c_void_p), pr.ctypes.data_as(C.c_void_p)) != 0:
            raise RuntimeError("pamlh_joint_reconstruction: " + self._L.pamlh_error(self._h).decode())
        return st, pr

    def beb_acd(self, x):
        """BEB under branch-site model A (4 site classes: 0, 1, 2a, 2b) or clade model C / D (3) at x: class posteriors per site,
        [nc][n_sites] (pamlh_beb_acd)."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        self.set_x(x)
        post = np.zeros((4 if self._is_branchsite() else 3, self.n_patt))
        self._L.pamlh_beb_acd.argtypes = [C.c_void_p] * 3
        if self._L.pamlh_beb_acd(self._h, x.ctypes.data_as(C.c_void_p), post.ctypes.data_as(C.c_void_p)) != 0:
            raise RuntimeError("pamlh_beb_acd: " + self._L.pamlh_error(self._h).decode())
        ns = C.c_int()
        self._L.pamlh_pose.restype = C.c_void_p
        self._L.pamlh_pose.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        pose = _arr(self._L.pamlh_pose(self._h, C.byref(ns)), np.int32, ns.value)
        return post[:, pose]

    def beb(self, x):
        """BEB under M2a / M8 at x: (Pr(w>1), mean omega, sd omega) per site (pamlh_beb)."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        pr, mw, se = np.zeros(self.n_patt), np.zeros(self.n_patt), np.zeros(self.n_patt)
        self._L.pamlh_beb.argtypes = [C.c_void_p] + [C.c_void_p] * 4
        if self._L.pamlh_beb(self._h, *[a.ctypes.data_as(C.c_void_p) for a in (x, pr, mw, se)]) != 0:
            raise RuntimeError("pamlh_beb: " + self._L.pamlh_error(self._h).decode())
        ns = C.c_int()
        self._L.pamlh_pose.restype = C.c_void_p
        self._L.pamlh_pose.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        pose = _arr(self._L.pamlh_pose(self._h, C.byref(ns)), np.int32, ns.value)
        return pr[pose], mw[pose], se[pose]
