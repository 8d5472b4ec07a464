"""paml_amd — MI355X-native likelihood engine for PAML's codeml/baseml hot path.

Only the hot path lives here: batched P(t) construction and Felsenstein pruning as hand-written
HIP kernels behind the C ABI in include/paml_amd.h (paml_amd/csrc), the host-side mirror of the
reference's model set-up (paml_amd/host, paml_amd/models.py) and a thin ctypes binding (engine.py).
"""
__version__ = "0.1.0"
