"""ctypes binding of libpaml_amd.so (include/paml_amd.h) — thin, no torch types in the ABI.

`Engine` mirrors the call sequence a PAML driver performs around com.plfun: create once per data
set, upload tips/tree, then per evaluation set eigen systems / classes / branch lengths and evaluate.
There is no CPU fallback: constructing an Engine without the built HIP library or without a GPU
raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from .problem import EIGEN_CIJK, EIGEN_JC69LIKE, EIGEN_K80, EIGEN_QMAT, EIGEN_UVROOT, Problem

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PAML_AMD_LIB") or os.path.join(_HERE, "lib", "libpaml_amd.so")   # override: kernel experiments
CSRC = os.path.join(_HERE, "csrc")
KEEP_PARTIALS = 1
JIT = 2
SHARD = 4      # the engine holds one rank's shard of a larger alignment (include/paml_amd.h)

EXPORTS = [
    "paml_amd_set_gene_class_rates", "paml_amd_get_branch_partials", "paml_amd_create", "paml_amd_destroy", "paml_amd_last_error", "paml_amd_set_stream", "paml_amd_set_tips",
    "paml_amd_set_tree", "paml_amd_set_pi", "paml_amd_set_eigen_uvroot", "paml_amd_set_eigen_cijk",
    "paml_amd_set_eigen_qrev_batch", "paml_amd_set_eigen_qrev_batch_sparse", "paml_amd_set_eigen_warm_start", "paml_amd_get_eigen", "paml_amd_eigen_counters", "paml_amd_set_eigen_k80", "paml_amd_set_eigen_jc69like", "paml_amd_set_eigen_qmat", "paml_amd_set_classes", "paml_amd_eval",
    "paml_amd_eval_batch", "paml_amd_eval_adg", "paml_amd_beb_grid", "paml_amd_beb_grid_classes", "paml_amd_compress_patterns", "paml_amd_eval_device", "paml_amd_eval_dirty", "paml_amd_eval_branch", "paml_amd_node_posterior", "paml_amd_get_pmat", "paml_amd_get_partials", "paml_amd_get_scale",
    "paml_amd_device_count", "paml_amd_set_device", "paml_amd_shard_bounds", "paml_amd_max_ranks", "paml_amd_flush", "paml_amd_eigen_status", "paml_amd_comm_unique_id", "paml_amd_comm_init", "paml_amd_comm_destroy", "paml_amd_comm_info", "paml_amd_comm_library", "paml_amd_comm_stats", "paml_amd_get_partial_sums", "paml_amd_branch_counters", "paml_amd_branch_coef_hits", "paml_amd_branch_refill_kernels", "paml_amd_branch_kernel_ms",
    "paml_amd_jit_prebuild", "paml_amd_profile", "paml_amd_profile_read", "paml_amd_counters", "paml_amd_kernel_name", "paml_amd_debug_program", "paml_amd_debug_jit",
]


class EngineError(RuntimeError):
    pass


# per-unit compiler flags (none at present; -mllvm -disable-machine-licm on engine_branch lowers its kernels' register counts from
# 220 - 256 to 186 - 224 — the logarithm's constants are no longer hoisted out of the main loops — and changes no timing: measured
# A / B on one box, profiles/r04_branch.txt)
UNIT_FLAGS = {}
# kernel experiments: a variant library beside the default one — PAML_AMD_LIB=<dir>/libpaml_amd.so PAML_AMD_EXTRA_FLAGS="-DX=1" python -c
# "from paml_amd import engine; engine.build()" compiles every unit with the extra flags into <dir> (objects in <dir>/obj)
EXTRA_FLAGS = os.environ.get("PAML_AMD_EXTRA_FLAGS", "").split()
UNITS = ("engine_core", "engine_comm", "engine_eval", "engine_branch", "engine_beb", "engine_jitdbg", "engine_compress")


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP engine for gfx950 in-tree (hipcc cross-compiles without a GPU): one object per translation unit, in
    parallel, then the shared library."""
    import glob
    from concurrent.futures import ThreadPoolExecutor
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(os.path.dirname(_HERE), "include", "paml_amd.h")]
    newest_hdr = max(os.path.getmtime(h) for h in hdrs)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(os.path.dirname(LIB_PATH), "obj")
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    for u in UNITS:
        src, obj = os.path.join(CSRC, u + ".hip"), os.path.join(objdir, u + ".o")
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(newest_hdr, os.path.getmtime(src)):
            jobs.append([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + UNIT_FLAGS.get(u, []) + EXTRA_FLAGS + ["-c", src, "-o", obj])
    objs = [os.path.join(objdir, u + ".o") for u in UNITS]
    if not jobs and os.path.exists(LIB_PATH) and all(os.path.getmtime(o) <= os.path.getmtime(LIB_PATH) for o in objs):
        return LIB_PATH

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
        list(ex.map(run, jobs))
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs + ["-lhiprtc", "-ldl"])
    return LIB_PATH


_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise EngineError("libpaml_amd.so is not built (run python -c 'import __graft_entry__ as g; g.build()')")
        L = C.CDLL(LIB_PATH)
        L.paml_amd_last_error.restype = C.c_char_p
        L.paml_amd_kernel_name.restype = C.c_char_p
        L.paml_amd_destroy.restype = None
        L.paml_amd_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint]
        L.paml_amd_destroy.argtypes = [C.c_void_p]
        L.paml_amd_last_error.argtypes = [C.c_void_p]
        L.paml_amd_kernel_name.argtypes = [C.c_void_p]
        L.paml_amd_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        L.paml_amd_set_tips.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.paml_amd_set_tree.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.paml_amd_set_pi.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.paml_amd_set_eigen_uvroot.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.paml_amd_set_eigen_cijk.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.paml_amd_set_eigen_qrev_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.paml_amd_set_eigen_qrev_batch_sparse.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.paml_amd_get_eigen.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.paml_amd_eigen_counters.argtypes = [C.c_void_p, C.POINTER(C.c_long), C.c_void_p, C.c_int]
        L.paml_amd_set_eigen_k80.argtypes = [C.c_void_p, C.c_int, C.c_double]
        L.paml_amd_set_eigen_jc69like.argtypes = [C.c_void_p, C.c_int]
        L.paml_amd_set_classes.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.paml_amd_eval.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.c_void_p, C.c_void_p]
        L.paml_amd_eval_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.paml_amd_eval_dirty.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
        L.paml_amd_eval_branch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.paml_amd_get_pmat.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.paml_amd_get_partials.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.paml_amd_get_scale.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.paml_amd_profile.argtypes = [C.c_void_p, C.c_int]
        L.paml_amd_profile_read.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_long)]
        L.paml_amd_counters.argtypes = [C.c_void_p, C.POINTER(C.c_long), C.POINTER(C.c_long)]
        L.paml_amd_shard_bounds.argtypes = [C.c_long, C.c_int, C.c_int, C.POINTER(C.c_long), C.POINTER(C.c_long)]
        L.paml_amd_max_ranks.argtypes = [C.c_long]
        L.paml_amd_flush.argtypes = [C.c_void_p]
        L.paml_amd_comm_unique_id.argtypes = [C.c_void_p]
        L.paml_amd_comm_init.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_long, C.c_long]
        L.paml_amd_comm_destroy.argtypes = [C.c_void_p]
        L.paml_amd_comm_info.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_long), C.POINTER(C.c_long), C.POINTER(C.c_int)]
        _LIB = L
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Engine:
    def __init__(self, n_states, n_tips, n_patt, max_classes=1, n_genes=1, flags=0):
        self._L = lib()
        h = C.c_void_p()
        rc = self._L.paml_amd_create(C.byref(h), n_states, n_tips, n_patt, max_classes, n_genes, flags)
        if rc != 0:
            raise EngineError("paml_amd_create failed (%d): no MI355X/HIP device visible or bad sizes" % rc)
        self._h = h
        self.n, self.n_tips, self.n_patt, self.n_genes = n_states, n_tips, n_patt, n_genes
        self.K = 1
        self.n_nodes = 0

    def close(self):
        if getattr(self, "_h", None):
            self._L.paml_amd_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise EngineError("%s (code %d)" % (self._L.paml_amd_last_error(self._h).decode(), rc))

    @property
    def kernel_name(self):
        return self._L.paml_amd_kernel_name(self._h).decode()

    # ---- pattern shards over several GPUs (paml_amd_comm_*) ----
    def comm_init(self, rank, world, unique_id, n_patt_global, first_pattern):
        """Collective: join the RCCL communicator of the ranks' engines.  `unique_id` = the 128 bytes rank 0 got from
        comm_unique_id() (None with world = 1: global chunking only, no communicator)."""
        buf = None if unique_id is None else C.create_string_buffer(bytes(unique_id), COMM_ID_BYTES)
        self._chk(self._L.paml_amd_comm_init(self._h, int(rank), int(world), buf, int(n_patt_global), int(first_pattern)))

    def comm_destroy(self):
        self._chk(self._L.paml_amd_comm_destroy(self._h))

    def comm_info(self):
        r, w, ch = C.c_int(), C.c_int(), C.c_int()
        ng, fp = C.c_long(), C.c_long()
        self._L.paml_amd_comm_info(self._h, C.byref(r), C.byref(w), C.byref(ng), C.byref(fp), C.byref(ch))
        return dict(rank=r.value, world=w.value, n_patt_global=ng.value, first_pattern=fp.value, chunk=ch.value)

    def comm_stats(self, enable=True, read=False):
        """Timed events around the exchange step (paml_amd_comm_stats).  read=True returns the figures of the last <= 64
        evaluations (flush and synchronise first): dict(n, exchange_us, exchange_us_max, lane_wait_us, lane_wait_us_max)."""
        self._L.paml_amd_comm_stats.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)] + [C.POINTER(C.c_double)] * 4
        if not read:
            self._chk(self._L.paml_amd_comm_stats(self._h, int(bool(enable)), None, None, None, None, None))
            return None
        n = C.c_int()
        v = [C.c_double() for _ in range(4)]
        self._chk(self._L.paml_amd_comm_stats(self._h, int(bool(enable)), C.byref(n), *[C.byref(x) for x in v]))
        return dict(n=n.value, exchange_us=v[0].value, exchange_us_max=v[1].value, lane_wait_us=v[2].value, lane_wait_us_max=v[3].value)

    def partial_sums(self):
        """Per-chunk partial sums of the last evaluation at their global positions (paml_amd_get_partial_sums)."""
        cap = 1 << 14
        out = np.zeros(cap)
        self._L.paml_amd_get_partial_sums.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        n = self._L.paml_amd_get_partial_sums(self._h, _p(out), cap)
        if n < 0:
            self._chk(n)
        return out[:n].copy()

    def set_stream(self, stream_ptr):
        self._chk(self._L.paml_amd_set_stream(self._h, C.c_void_p(stream_ptr)))

    def set_tips(self, z, weights, cleandata=1, n_chara=None, chara_map=None, gene_off=None):
        z = np.ascontiguousarray(z, dtype=np.uint8)
        w = np.ascontiguousarray(weights, dtype=np.float64)
        assert z.shape == (self.n_tips, self.n_patt) and w.shape == (self.n_patt,)
        nch = None if n_chara is None else np.ascontiguousarray(n_chara, dtype=np.int32)
        cm = None if chara_map is None else np.ascontiguousarray(chara_map, dtype=np.uint8)
        go = None if gene_off is None else np.ascontiguousarray(gene_off, dtype=np.int32)
        self._chk(self._L.paml_amd_set_tips(self._h, _p(z), int(cleandata), 0 if nch is None else len(nch), _p(nch), _p(cm),
                                            _p(w), _p(go)))

    def set_tree(self, tree, scale_node=None):
        ptr, flat = tree.csr()
        lab = np.ascontiguousarray(tree.label, dtype=np.int32)
        sc = None if scale_node is None else np.ascontiguousarray(scale_node, dtype=np.uint8)
        self.n_nodes = tree.n_nodes
        self._chk(self._L.paml_amd_set_tree(self._h, tree.n_nodes, tree.root, _p(ptr), _p(flat), _p(lab), _p(sc)))

    def set_pi(self, pi):
        pi = np.ascontiguousarray(np.atleast_2d(pi), dtype=np.float64)
        self._chk(self._L.paml_amd_set_pi(self._h, pi.shape[0], _p(pi)))

    def set_eigen(self, set_id, e):
        k = e["kind"]
        if k == EIGEN_UVROOT:
            U, V, R = (np.ascontiguousarray(e[x], dtype=np.float64) for x in ("U", "V", "Root"))
            self._chk(self._L.paml_amd_set_eigen_uvroot(self._h, set_id, _p(U), _p(V), _p(R)))
        elif k == EIGEN_CIJK:
            Cj, R = (np.ascontiguousarray(e[x], dtype=np.float64) for x in ("Cijk", "Root"))
            self._chk(self._L.paml_amd_set_eigen_cijk(self._h, set_id, int(e["nR"]), _p(Cj), _p(R)))
        elif k == EIGEN_K80:
            self._chk(self._L.paml_amd_set_eigen_k80(self._h, set_id, float(e["kappa"])))
        elif k == EIGEN_JC69LIKE:
            self._chk(self._L.paml_amd_set_eigen_jc69like(self._h, set_id))
        elif k == EIGEN_QMAT:
            Q = np.ascontiguousarray(e["Q"], dtype=np.float64)
            self._L.paml_amd_set_eigen_qmat.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
            self._chk(self._L.paml_amd_set_eigen_qmat(self._h, set_id, _p(Q)))
        else:
            raise ValueError(k)

    def set_classes(self, mode, freqK, rate, eigen_of, qfactor=None):
        f = np.ascontiguousarray(freqK, dtype=np.float64)
        r = np.ascontiguousarray(rate, dtype=np.float64)
        eo = np.ascontiguousarray(eigen_of, dtype=np.int32)
        K = len(f)
        n_labels = eo.size // (self.n_genes * K)
        q = None if qfactor is None else np.ascontiguousarray(qfactor, dtype=np.float64)
        self.K = K
        self._chk(self._L.paml_amd_set_classes(self._h, int(mode), K, _p(f), _p(r), n_labels, _p(eo), _p(q)))

    def load(self, pb: Problem):
        """Upload everything a Problem holds (tips, tree, pi, eigen systems, classes)."""
        self.set_tips(pb.z, pb.weights, pb.cleandata, None if pb.cleandata else pb.n_chara,
                      None if pb.cleandata else pb.chara_map, pb.gene_off if pb.n_genes > 1 else None)
        self.set_tree(pb.tree, pb.scale_node)
        self.set_pi(pb.pi)
        for i, e in enumerate(pb.eigen):
            self.set_eigen(i, e)
        self.set_classes(pb.mode, pb.freqK, pb.rate[:pb.K] if getattr(pb, "rate_per_gene", False) else pb.rate, pb.eigen_of, pb.qfactor)
        if getattr(pb, "rate_per_gene", False):
            r = np.ascontiguousarray(pb.rate, dtype=np.float64).reshape(pb.n_genes * pb.K)
            self._L.paml_amd_set_gene_class_rates.argtypes = [C.c_void_p, C.c_void_p]
            self._chk(self._L.paml_amd_set_gene_class_rates(self._h, _p(r)))
        return self

    def eval(self, branch, gene_rate=None, want_lnf=False, want_fhk=False):
        b = np.ascontiguousarray(branch, dtype=np.float64)
        g = None if gene_rate is None else np.ascontiguousarray(gene_rate, dtype=np.float64)
        lnL = C.c_double()
        lnf = np.zeros(self.n_patt) if want_lnf else None
        fhk = np.zeros((self.K, self.n_patt)) if want_fhk else None
        self._chk(self._L.paml_amd_eval(self._h, _p(b), _p(g), C.byref(lnL), _p(lnf), _p(fhk)))
        return dict(lnL=lnL.value, lnf=lnf, fhK=fhk)

    def eval_batch(self, branch, gene_rate=None, eigen_of=None, qfactor=None, freqK=None, rate=None, want_lnf=False):
        """lnL of every row of branch[n_batch][n_nodes] in one launch (paml_amd_eval_batch); the optional tables carry
        one leading batch axis over the layouts of set_classes."""
        b = np.ascontiguousarray(branch, dtype=np.float64)
        assert b.ndim == 2
        nb = b.shape[0]

        def opt(a, dt):
            if a is None:
                return None
            a = np.ascontiguousarray(a, dtype=dt)
            assert a.shape[0] == nb
            return a
        g, eo = opt(gene_rate, np.float64), opt(eigen_of, np.int32)
        qf, fk, rt = opt(qfactor, np.float64), opt(freqK, np.float64), opt(rate, np.float64)
        out = np.zeros(nb)
        lnf = np.zeros((nb, self.n_patt)) if want_lnf else None
        self._L.paml_amd_eval_batch.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 8
        self._chk(self._L.paml_amd_eval_batch(self._h, nb, _p(b), _p(g), _p(eo), _p(qf), _p(fk), _p(rt), _p(out), _p(lnf)))
        return (out, lnf) if want_lnf else out

    def eval_adg(self, branch, MK, pose, gene_rate=None):
        """lfunAdG: +lnL of the rate chain MK[K][K] over the sites pose[ls] (site -> pattern), fx_r on the device."""
        b = np.ascontiguousarray(branch, dtype=np.float64)
        g = None if gene_rate is None else np.ascontiguousarray(gene_rate, dtype=np.float64)
        MK = np.ascontiguousarray(MK, dtype=np.float64)
        pose = np.ascontiguousarray(pose, dtype=np.int32)
        lnL = C.c_double()
        self._L.paml_amd_eval_adg.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_double)]
        self._chk(self._L.paml_amd_eval_adg(self._h, _p(b), _p(g), _p(MK), _p(pose), len(pose), C.byref(lnL)))
        return lnL.value

    def beb_grid(self, pcl, iw, w_class):
        """BEB grid integral over the fhK of the last evaluation (paml_amd_beb_grid): pcl[n_grid][n_cls], iw[n_grid][n_cls],
        w_class[K] -> dict(ln_fx, pr_last[n_patt], mean_w[n_patt], sd_w[n_patt])."""
        pcl = np.ascontiguousarray(pcl, dtype=np.float64)
        iw = np.ascontiguousarray(iw, dtype=np.int32)
        wc = np.ascontiguousarray(w_class, dtype=np.float64)
        assert pcl.shape == iw.shape and pcl.ndim == 2
        lnfx = C.c_double()
        pr, mw, sd = np.zeros(self.n_patt), np.zeros(self.n_patt), np.zeros(self.n_patt)
        self._L.paml_amd_beb_grid.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_double),
                                              C.c_void_p, C.c_void_p, C.c_void_p]
        self._chk(self._L.paml_amd_beb_grid(self._h, pcl.shape[0], pcl.shape[1], _p(pcl), _p(iw), _p(wc), C.byref(lnfx), _p(pr), _p(mw), _p(sd)))
        return dict(ln_fx=lnfx.value, pr_last=pr, mean_w=mw, sd_w=sd)

    def beb_grid_classes(self, pcl, iw):
        """Posterior of every mixture class over the grid (paml_amd_beb_grid_classes) -> dict(ln_fx, post[n_cls][n_patt])."""
        pcl = np.ascontiguousarray(pcl, dtype=np.float64)
        iw = np.ascontiguousarray(iw, dtype=np.int32)
        assert pcl.shape == iw.shape and pcl.ndim == 2
        lnfx = C.c_double()
        post = np.zeros((pcl.shape[1], self.n_patt))
        self._L.paml_amd_beb_grid_classes.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.c_void_p]
        self._chk(self._L.paml_amd_beb_grid_classes(self._h, pcl.shape[0], pcl.shape[1], _p(pcl), _p(iw), C.byref(lnfx), _p(post)))
        return dict(ln_fx=lnfx.value, post=post)

    def eval_device(self, branch, d_lnL_ptr, gene_rate=None):
        b = np.ascontiguousarray(branch, dtype=np.float64)
        g = None if gene_rate is None else np.ascontiguousarray(gene_rate, dtype=np.float64)
        self._chk(self._L.paml_amd_eval_device(self._h, _p(b), _p(g), C.c_void_p(d_lnL_ptr)))

    def set_eigen_warm_start(self, on=-1):
        """paml_amd_set_eigen_warm_start: switch (1 / 0; -1 leaves it), returns the number of warm-started decompositions so far."""
        n = C.c_long()
        self._L.paml_amd_set_eigen_warm_start.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_long)]
        self._chk(self._L.paml_amd_set_eigen_warm_start(self._h, int(on), C.byref(n)))
        return n.value

    def set_eigen_qrev_batch(self, set_ids, Q, pi, scale=None):
        """Decompose the reversible rate matrices Q[k] (frequencies pi[k]) on the device into eigen sets set_ids[k]; Root is divided by scale[k]."""
        ids = np.ascontiguousarray(set_ids, dtype=np.int32)
        Q = np.ascontiguousarray(Q, dtype=np.float64).reshape(len(ids), self.n, self.n)
        pi = np.ascontiguousarray(pi, dtype=np.float64).reshape(len(ids), self.n)
        sc = None if scale is None else np.ascontiguousarray(scale, dtype=np.float64).reshape(len(ids))
        self._chk(self._L.paml_amd_set_eigen_qrev_batch(self._h, len(ids), _p(ids), _p(Q), _p(pi), _p(sc)))

    def set_eigen_qrev_batch_sparse(self, set_ids, row, col, vals, pi, scale=None):
        """As set_eigen_qrev_batch, the matrices given by their elements at (row[k] >= col[k]), vals[set][k]; everything else zero."""
        ids = np.ascontiguousarray(set_ids, dtype=np.int32)
        row = np.ascontiguousarray(row, dtype=np.int32); col = np.ascontiguousarray(col, dtype=np.int32)
        vals = np.ascontiguousarray(vals, dtype=np.float64).reshape(len(ids), len(row))
        pi = np.ascontiguousarray(pi, dtype=np.float64).reshape(len(ids), self.n)
        sc = None if scale is None else np.ascontiguousarray(scale, dtype=np.float64).reshape(len(ids))
        self._chk(self._L.paml_amd_set_eigen_qrev_batch_sparse(self._h, len(ids), _p(ids), len(row), _p(row), _p(col), _p(vals), _p(pi), _p(sc)))

    def get_eigen(self, set_id):
        U, V, R = np.empty((self.n, self.n)), np.empty((self.n, self.n)), np.empty(self.n)
        self._chk(self._L.paml_amd_get_eigen(self._h, int(set_id), _p(U), _p(V), _p(R)))
        return U, V, R

    def eigen_counters(self, cap=4096):
        n, sw = C.c_long(), np.zeros(cap, dtype=np.int32)
        m = self._L.paml_amd_eigen_counters(self._h, C.byref(n), _p(sw), cap)
        return {"n_decomposed": n.value, "sweeps": sw[:max(m, 0)].copy()}

    def flush(self):
        """After a run of eval_device calls: the engine's stream waits for the totals still on the collective stream."""
        self._chk(self._L.paml_amd_flush(self._h))

    def eigen_status(self):
        """paml_amd_eigen_status: waits for the engine's stream; raises (code -5) if a queued device eigen-decomposition did not converge."""
        self._L.paml_amd_eigen_status.argtypes = [C.c_void_p]
        self._chk(self._L.paml_amd_eigen_status(self._h))

    def eval_dirty(self, branch, clean, gene_rate=None):
        b = np.ascontiguousarray(branch, dtype=np.float64)
        c = np.ascontiguousarray(clean, dtype=np.uint8)
        g = None if gene_rate is None else np.ascontiguousarray(gene_rate, dtype=np.float64)
        lnL = C.c_double()
        self._chk(self._L.paml_amd_eval_dirty(self._h, _p(b), _p(g), _p(c), C.byref(lnL)))
        return lnL.value

    def eval_branch(self, node_b, t, branch, gene_rate=None):
        """lnL(t), dlnL/dt, d2lnL/dt2 of the branch above node_b at the trial lengths t (lfuntdd)."""
        tt = np.ascontiguousarray(np.atleast_1d(t), dtype=np.float64)
        b = np.ascontiguousarray(branch, dtype=np.float64)
        g = None if gene_rate is None else np.ascontiguousarray(gene_rate, dtype=np.float64)
        l, dl, ddl = np.zeros(len(tt)), np.zeros(len(tt)), np.zeros(len(tt))
        self._chk(self._L.paml_amd_eval_branch(self._h, int(node_b), len(tt), _p(tt), _p(b), _p(g), _p(l), _p(dl), _p(ddl)))
        return l, dl, ddl

    def branch_partials(self):
        """Per-block partial sums of the last eval_branch: array [rows, 3 n_t] (paml_amd_get_branch_partials)."""
        rows, cols = C.c_long(), C.c_int()
        self._L.paml_amd_get_branch_partials.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.POINTER(C.c_long), C.POINTER(C.c_int)]
        self._chk(self._L.paml_amd_get_branch_partials(self._h, None, 0, C.byref(rows), C.byref(cols)))
        out = np.zeros((rows.value, cols.value))
        self._chk(self._L.paml_amd_get_branch_partials(self._h, _p(out), out.size, C.byref(rows), C.byref(cols)))
        return out

    def branch_counters(self):
        a, b = C.c_long(), C.c_long()
        self._L.paml_amd_branch_counters.argtypes = [C.c_void_p, C.POINTER(C.c_long), C.POINTER(C.c_long)]
        self._L.paml_amd_branch_counters(self._h, C.byref(a), C.byref(b))
        self._L.paml_amd_branch_coef_hits.restype = C.c_long
        self._L.paml_amd_branch_coef_hits.argtypes = [C.c_void_p]
        self._L.paml_amd_branch_refill_kernels.restype = C.c_long
        self._L.paml_amd_branch_refill_kernels.argtypes = [C.c_void_p]
        return dict(n_calls=a.value, n_nodes=b.value, coef_hits=self._L.paml_amd_branch_coef_hits(self._h),
                    refill_kernels=self._L.paml_amd_branch_refill_kernels(self._h))

    def branch_kernel_ms(self):
        """Duration of the last eval_branch's contraction kernels by HIP events (needs profile(True)); < 0: not timed."""
        self._L.paml_amd_branch_kernel_ms.restype = C.c_double
        self._L.paml_amd_branch_kernel_ms.argtypes = [C.c_void_p]
        return float(self._L.paml_amd_branch_kernel_ms(self._h))

    def node_posterior(self, node, branch, gene_rate=None):
        """Posterior probabilities of the states at an internal node, [n_patt][n] (paml_amd_node_posterior)."""
        b = np.ascontiguousarray(branch, dtype=np.float64)
        g = None if gene_rate is None else np.ascontiguousarray(gene_rate, dtype=np.float64)
        post = np.zeros((self.n_patt, self.n))
        self._L.paml_amd_node_posterior.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        self._chk(self._L.paml_amd_node_posterior(self._h, int(node), _p(b), _p(g), _p(post)))
        return post

    def get_pmat(self, gene, iclass, node):
        P = np.zeros((self.n, self.n))
        self._chk(self._L.paml_amd_get_pmat(self._h, gene, iclass, node, _p(P)))
        return P

    def get_partials(self, node, iclass=0):
        a = np.zeros((self.n_patt, self.n))
        self._chk(self._L.paml_amd_get_partials(self._h, node, iclass, _p(a)))
        return a

    def get_scale(self, node, iclass=0):
        a = np.zeros(self.n_patt)
        self._chk(self._L.paml_amd_get_scale(self._h, node, iclass, _p(a)))
        return a

    def profile(self, on=True):
        self._chk(self._L.paml_amd_profile(self._h, int(on)))

    def profile_read(self):
        a, b, c, n = C.c_double(), C.c_double(), C.c_double(), C.c_long()
        self._chk(self._L.paml_amd_profile_read(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(n)))
        return dict(ms_pmat=a.value, ms_prune=b.value, ms_reduce=c.value, n_evals=n.value)

    def counters(self):
        a, b = C.c_long(), C.c_long()
        self._L.paml_amd_counters(self._h, C.byref(a), C.byref(b))
        return dict(n_eval=a.value, n_pmat=b.value)


COMM_ID_BYTES = 128


def shard_bounds(n_patt_global, world, rank):
    """(first, count) of rank's contiguous pattern shard — paml_amd_shard_bounds, host only (no GPU needed)."""
    first, count = C.c_long(), C.c_long()
    rc = lib().paml_amd_shard_bounds(int(n_patt_global), int(world), int(rank), C.byref(first), C.byref(count))
    if rc != 0:
        raise EngineError("paml_amd_shard_bounds(%d patterns, world %d) failed (%d): at most %d ranks for this many patterns"
                          % (n_patt_global, world, rc, lib().paml_amd_max_ranks(int(n_patt_global))))
    return first.value, count.value


def max_ranks(n_patt_global):
    """The largest number of ranks the patterns can be sharded over (= the number of reduction chunks)."""
    return lib().paml_amd_max_ranks(int(n_patt_global))


def comm_unique_id():
    """The 128-byte RCCL id rank 0 creates and passes to the other ranks (paml_amd_comm_unique_id)."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    rc = lib().paml_amd_comm_unique_id(buf)
    if rc != 0:
        raise EngineError("paml_amd_comm_unique_id failed (%d): librccl.so.1 not found?" % rc)
    return buf.raw


def comm_library():
    """Path of the collective library the exchange step is bound to (paml_amd_comm_library: dladdr of ncclAllReduce)."""
    buf = C.create_string_buffer(4096)
    L = lib()
    L.paml_amd_comm_library.argtypes = [C.c_char_p, C.c_int]
    rc = L.paml_amd_comm_library(buf, len(buf))
    if rc < 0:
        raise EngineError("paml_amd_comm_library failed (%d): librccl.so.1 not found?" % rc)
    return buf.value.decode()


def device_count():
    """GPUs visible to this process (paml_amd_device_count)."""
    return int(lib().paml_amd_device_count())


def debug_program(tree, scale_node=None, keep=False, clean=None):
    """Host-only: the flattened tree program (list of (code, a, b, c)) and its stack depth."""
    L = lib()
    ptr, flat = tree.csr()
    sc = None if scale_node is None else np.ascontiguousarray(scale_node, dtype=np.uint8)
    cl = None if clean is None else np.ascontiguousarray(clean, dtype=np.uint8)
    cap = 8 * tree.n_nodes + 8
    ops = np.zeros((cap, 4), dtype=np.int32)
    ms = C.c_int()
    L.paml_amd_debug_program.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                         C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    nops = L.paml_amd_debug_program(tree.n_tips, tree.n_nodes, tree.root, _p(ptr), _p(flat), _p(sc), int(keep), _p(cl),
                                    _p(ops), cap, C.byref(ms))
    if nops < 0:
        raise EngineError("debug_program failed (%d)" % nops)
    return [tuple(int(v) for v in r) for r in ops[:nops]], ms.value


def debug_jit(tree, scale_node=None, compile=True, n_states=0, fused=None):
    """Host-only: source of the kernel specialised for `tree` (hiprtc-compiled for gfx950 when compile=True);
    n_states 4 / 5 / 20 selects the one-pattern-per-lane kernels, anything else the 61-state MFMA kernel."""
    L = lib()
    ptr, flat = tree.csr()
    sc = None if scale_node is None else np.ascontiguousarray(scale_node, dtype=np.uint8)
    cap = 1 << 20
    buf = C.create_string_buffer(cap)
    L.paml_amd_debug_jit.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_int, C.c_int]
    flags = int(bool(compile)) | (int(n_states) << 8)
    if fused:      # (K classes, n_codes[, reduction chunk]): the fused 4 / 5-state kernel
        flags |= 2 | (int(fused[0]) << 16) | (int(fused[1]) << 24) | ((int(fused[2]) // 256 if len(fused) > 2 else 1) << 2)
    rc = L.paml_amd_debug_jit(tree.n_tips, tree.n_nodes, tree.root, _p(ptr), _p(flat), _p(sc), buf, cap, flags)
    if rc < 0:
        raise EngineError("debug_jit failed (%d): %s" % (rc, buf.value.decode(errors="replace")[-3000:]))
    return buf.value.decode()


JIT_SHIPPED_DIR = os.path.join(_HERE, "lib", "jit")


def jit_prebuild(tree, n_states, n_codes, K=1, n_patt_global=1_000_000, scale_node=None, directory=None):
    """Host-only: compile the per-tree kernel for (tree, sizes) into `directory` (default: the library's lib/jit)."""
    L = lib()
    ptr, flat = tree.csr()
    sc = None if scale_node is None else np.ascontiguousarray(scale_node, dtype=np.uint8)
    d = directory or JIT_SHIPPED_DIR
    os.makedirs(d, exist_ok=True)
    log = C.create_string_buffer(1 << 16)
    L.paml_amd_jit_prebuild.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_long, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_char_p, C.c_char_p, C.c_int]
    rc = L.paml_amd_jit_prebuild(int(n_states), tree.n_tips, int(n_codes), int(K), int(n_patt_global), tree.n_nodes, tree.root, _p(ptr), _p(flat),
                                 _p(sc), os.fsencode(d), log, len(log))
    if rc != 0:
        raise EngineError("jit_prebuild failed (%d): %s" % (rc, log.value.decode(errors="replace")[-2000:]))


def compress_patterns(chars, gene=None):
    """PatternWeight on the device (paml_amd_compress_patterns): chars uint8 [n_seq][n_sites * width] or [n_seq][n_sites][width].
    Returns dict(first_site[n_patt], weights[n_patt], pose[n_sites])."""
    chars = np.ascontiguousarray(chars, dtype=np.uint8)
    width = 1 if chars.ndim == 2 else chars.shape[2]
    n_seq, n_sites = chars.shape[0], chars.shape[1]
    g = None if gene is None else np.ascontiguousarray(gene, dtype=np.int32)
    npatt = C.c_int()
    first, w, pose = np.zeros(n_sites, dtype=np.int32), np.zeros(n_sites), np.zeros(n_sites, dtype=np.int32)
    L = lib()
    L.paml_amd_compress_patterns.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_void_p, C.c_void_p, C.c_void_p]
    rc = L.paml_amd_compress_patterns(n_seq, n_sites, width, _p(chars), _p(g), C.byref(npatt), _p(first), _p(w), _p(pose))
    if rc != 0:
        raise EngineError("paml_amd_compress_patterns failed (%d)" % rc)
    return dict(first_site=first[:npatt.value].copy(), weights=w[:npatt.value].copy(), pose=pose)


def engine_for(pb: Problem, flags=0) -> Engine:
    return Engine(pb.n, pb.tree.n_tips, pb.n_patt, max_classes=pb.K, n_genes=pb.n_genes, flags=flags).load(pb)
