"""topdogspectrumanalyser_amd - MI355X-native "sample method" IQ -> spectrum path.

Hand-written HIP (gfx950) kernels behind a ctypes C-ABI (include/tdsa_hip.h), kept behind the
reference's own SampleDataSource / DataProcessor Python API.  Importing the package requires
libtdsa_hip.so (built by __graft_entry__.build()); there is no CPU fallback.
"""
from .engine import SpectrumEngine  # noqa: F401

__all__ = ["SpectrumEngine"]
