"""ipopt_amd -- host-side Python mirror of the MI355X KKT linear-solver backend for Ipopt.

The product is the C-ABI shared library ``ipopt_amd/lib/libmi355x_kkt.so`` (hand-written HIP for
gfx950 + host C++ symbolic analysis, declared in ``include/mi355x_kkt.h``).  This package only binds
that ABI with ctypes for tests, the benchmark and the multi-GPU launcher; it contains no numerical
code of its own and never imports anything from ``oracle/``.
"""
from .kkt import KKTSolver, KKTInfo, KKTError, load_library, library_path, STATUS  # noqa: F401
