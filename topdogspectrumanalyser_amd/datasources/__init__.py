"""Data sources.  `from . import SampleDataSource` works as in the reference (datasources/__init__.py)."""
from .base import SampleDataSource, SweepDataSource  # noqa: F401
from .hackrf_samples import HackrfSamplesDataSource  # noqa: F401
from .rtl_samples import RtlSamplesDataSource  # noqa: F401
from .audio_samples import MicrophoneSamplesDataSource  # noqa: F401

# the reference's plugin table (core/source_manager.py:33-39), sample sources only
SOURCE_CLASSES = {
    "rtl_samples": RtlSamplesDataSource,
    "microphone_samples": MicrophoneSamplesDataSource,
    "hackrf_samples": HackrfSamplesDataSource,
}
