"""MicrophoneSamplesDataSource - interface of the reference's datasources/audio_samples.py (:23-208).

The real-input (rfft) path is row f-1 of SURVEY.md 8(f) ("next"): the GPU kernels of this round are the
complex-IQ path the headline benchmark uses.  The class keeps the reference's constructor, knobs and
frequency axis so that callers can be wired up, and fails loudly at start(): there is deliberately no
numpy stand-in for the missing kernel.
"""
import numpy as np

from .base import SampleDataSource

AUDIO_CHANNELS = ("mono", "left", "right", "stereo")


class MicrophoneSamplesDataSource(SampleDataSource):
    def __init__(self, sample_rate: int = 44100, centre_freq: int = 0):
        super().__init__(sample_rate, centre_freq)
        self.fft_size = 1024
        self.window_type = "hanning"
        self.channel_mode = "mono"
        self.stream = None
        self.running = False
        self.use_psd = False
        self.set_window()

    def set_window(self):
        funcs = {"hanning": np.hanning, "hamming": np.hamming}
        self.window = funcs.get(self.window_type, np.hanning)(self.fft_size)

    def set_fft_size(self, fft_size: int):
        self.fft_size = fft_size
        self.set_window()
        self._averager.reset()

    @property
    def sample_count(self) -> int:
        return self.fft_size

    @sample_count.setter
    def sample_count(self, value: int):
        self.set_fft_size(value)

    def set_window_type(self, window_type: str):
        self.window_type = window_type
        self.set_window()

    def set_channel_mode(self, mode: str) -> None:
        if mode in AUDIO_CHANNELS:
            self.channel_mode = mode

    @property
    def _rfft_bins(self) -> int:
        return self.fft_size // 2 + 1

    def _freq_bins(self) -> np.ndarray:
        return np.linspace(0, self.sample_rate / 2, self._rfft_bins)

    def start(self, frequency=None):
        raise RuntimeError("Microphone initialisation failed: the real-input (rfft) GPU path is not built in "
                           "this release (SURVEY.md 8(f) row f-1); no CPU fallback is provided")

    def stop(self):
        self.running = False

    def get_power_levels(self):
        return np.full(self._rfft_bins, -120.0), self._freq_bins()    # not running (audio_samples.py:137-138)

    def update_frequency(self, sample_rate: float, centre_freq: float):
        self.sample_rate = int(sample_rate)
        self.centre_freq = int(centre_freq)

    def update_centre_frequency(self, centre_freq: float):
        self.centre_freq = int(centre_freq)

    def set_psd_mode(self, enabled: bool):
        self.use_psd = enabled
