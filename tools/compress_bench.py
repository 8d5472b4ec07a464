"""paml_amd_compress_patterns at scale, next to numpy's lexsort on the host cores' single thread.
python tools/compress_bench.py [n_seq n_sites width alphabet]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from paml_amd.engine import compress_patterns

cases = [(16, 1_000_000, 3, 4), (32, 1_000_000, 1, 4), (16, 10_000_000, 3, 4), (6, 10_000_000, 1, 20)]
if len(sys.argv) > 4:
    cases = [tuple(int(v) for v in sys.argv[1:5])]
for n_seq, n_sites, width, alphabet in cases:
    rng = np.random.default_rng(1)
    symbols = np.frombuffer(b"TCAGYRMKSWHBVDN-?EFILPQ"[:alphabet], dtype=np.uint8)
    base = symbols[rng.integers(0, alphabet, size=(1, n_sites, width))]
    chars = np.where(rng.random((n_seq, n_sites, width)) < 0.02, symbols[rng.integers(0, alphabet, size=(n_seq, n_sites, width))], base).astype(np.uint8)
    chars = np.ascontiguousarray(chars)
    arg = chars if width > 1 else chars[:, :, 0]
    compress_patterns(arg[:, :1000])                      # context / first-launch costs outside the clock
    t0 = time.perf_counter(); got = compress_patterns(arg); t_gpu = time.perf_counter() - t0
    t0 = time.perf_counter()
    cols = chars.transpose(1, 0, 2).reshape(n_sites, -1)
    order = np.lexsort([cols[:, k] for k in range(cols.shape[1] - 1, -1, -1)])
    sc = cols[order]
    head = np.ones(n_sites, dtype=bool); head[1:] = (sc[1:] != sc[:-1]).any(axis=1)
    t_cpu = time.perf_counter() - t0
    assert len(got["weights"]) == int(head.sum()) and np.array_equal(got["first_site"], order[head])
    print("%2d seq x %8d sites x %d chars, %2d symbols: %8d patterns  device %.3f s (%.1f M sites/s, incl. H2D/D2H)  numpy lexsort %.2f s  -> %.0fx"
          % (n_seq, n_sites, width, alphabet, len(got["weights"]), t_gpu, n_sites / t_gpu / 1e6, t_cpu, t_cpu / t_gpu), flush=True)
