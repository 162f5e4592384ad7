"""hypre_amd — MI355X-native BoomerAMG solve phase behind hypre's C interface.

The product is the C-ABI shared library ``hypre_amd/lib/libhypre_amd.so`` (HIP
kernels + host runtime, see ``include/*.h``).  This package is the thin Python
host layer used by the tests, ``bench.py`` and ``__graft_entry__.py``: it loads
the library through ctypes and mirrors hypre's object names.
"""
from .binding import lib, load_library, HypreAmdError  # noqa: F401
