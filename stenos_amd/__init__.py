"""stenos_amd -- MI355X-native Stenos block codec (libstenos.so) and its Python plumbing.

The product is the C-ABI shared library built from stenos_amd/csrc (hand-written HIP for gfx950 behind
the unchanged stenos_compress / stenos_decompress / stenos_compress_generic ABI).  This package only
loads it through ctypes and offers torch-tensor conveniences for tests and bench.py; there is no
Python or CPU implementation of the codec here, and importing `stenos_amd.api` fails loudly when the
library has not been built.
"""
from .api import LIB_PATH, StenosError, Stenos, build_library, load_library  # noqa: F401
