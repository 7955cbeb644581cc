"""topdogspectrumanalyser_amd - MI355X-native "sample method" IQ -> spectrum path.

Hand-written HIP (gfx950) kernels behind a ctypes C-ABI (include/tdsa_hip.h), kept behind the
reference's own SampleDataSource / DataProcessor Python API.  Importing the package requires
libtdsa_hip.so (built by __graft_entry__.build()); there is no CPU fallback.
"""
from .engine import HostPipe, SpectrumEngine, TraceState  # noqa: F401
from .utils.signal_processing import TraceAverager  # noqa: F401
from .datasources import (SOURCE_CLASSES, HackrfSamplesDataSource, MicrophoneSamplesDataSource,  # noqa: F401
                          RtlSamplesDataSource, SampleDataSource, SweepDataSource)
from .core.display_data_processor import DataProcessor  # noqa: F401
from .core.tare_state import TareState  # noqa: F401

__all__ = ["SpectrumEngine", "HostPipe", "TraceState", "TraceAverager", "SampleDataSource", "SweepDataSource",
           "HackrfSamplesDataSource", "RtlSamplesDataSource", "MicrophoneSamplesDataSource",
           "SOURCE_CLASSES", "DataProcessor", "TareState"]
