"""Abstract data-source interfaces - the drop-in boundary of the sample-method path.

Same classes, method names, argument meaning and defaults as the reference's datasources/base.py
(SweepDataSource :15-39, SampleDataSource :43-169); code that drives a reference source drives these.
"""
import threading
import time
from abc import ABC, abstractmethod
from typing import Optional, Tuple

import numpy as np

from ..utils.signal_processing import TraceAverager


class SweepDataSource(ABC):
    """Sweep sources wrap external CLI tools in the reference (out of scope here); interface only."""

    @abstractmethod
    def start(self, frequency=None):
        ...

    @abstractmethod
    def stop(self):
        ...

    @abstractmethod
    def get_data(self):
        ...


class SampleDataSource(ABC):
    """FFT-analysis source: get_power_levels() -> (power_db[N], frequency_bins[N])."""

    def __init__(self, sample_rate: Optional[int] = None, centre_freq: Optional[int] = None):
        self.sample_rate = sample_rate
        self.centre_freq = centre_freq
        self._averager = TraceAverager()
        self._last_raw_samples: Optional[np.ndarray] = None
        self.last_data_time: float = 0.0
        self._raw_lock = threading.Lock()

    @abstractmethod
    def start(self, frequency=None):
        ...

    @abstractmethod
    def stop(self):
        ...

    @abstractmethod
    def get_power_levels(self) -> Tuple[np.ndarray, np.ndarray]:
        ...

    @property
    @abstractmethod
    def sample_count(self) -> int:
        ...

    @sample_count.setter
    @abstractmethod
    def sample_count(self, value: int):
        ...

    @abstractmethod
    def update_frequency(self, sample_rate: float, centre_freq: float):
        ...

    @abstractmethod
    def update_centre_frequency(self, centre_freq: float):
        ...

    def get_raw_samples(self) -> Optional[np.ndarray]:
        with self._raw_lock:
            return self._last_raw_samples

    def read_samples_only(self) -> Optional[np.ndarray]:
        return None

    def _store_raw(self, samples: np.ndarray) -> None:
        with self._raw_lock:
            self._last_raw_samples = samples
        self.last_data_time = time.monotonic()

    def set_psd_mode(self, enabled: bool):
        pass

    def set_averaging(self, mode: str, n: int) -> None:
        self._averager.set_mode(mode, n)

    def reset_averaging(self) -> None:
        self._averager.reset()
