"""MicrophoneSamplesDataSource - stereo audio source whose DSP runs on the MI355X.

Surface and semantics follow the reference's datasources/audio_samples.py (class :23-208): stereo float32
stream, rolling buffer at low sample rates (:149-156), per-channel mean removal + Hann/Hamming window +
real FFT + one-sided power with the non-DC/non-Nyquist bins doubled (:121-132), mono / left / right /
stereo selection (:158-180: in stereo the left channel goes through the averager, the right does not),
floors 1e-10 (power) / 1e-12 (PSD), -120 dB row when not running or on error (:137-138,182-184).

The two real channels ride ONE complex FFT on the GPU (z = left + i*right, separated afterwards by
real_fold_kernel): tdsa_process_real2 in include/tdsa_hip.h.  The audio stream object comes from
`sounddevice` when installed, or from `stream_factory` (anything with .start/.stop/.close/.read(n)).
"""
import logging
from typing import Callable, Optional

import numpy as np

from .base import SampleDataSource
from ._gpu import GpuSpectrumMixin
from ..utils.constants import DSPConstants

try:
    import sounddevice as sd  # type: ignore
    _SD_AVAILABLE = True
except (ImportError, OSError):
    sd = None
    _SD_AVAILABLE = False

logger = logging.getLogger(__name__)

AUDIO_CHANNELS = ("mono", "left", "right", "stereo")
_MAX_READ_MS = 30      # target blocking time of one stream read


class MicrophoneSamplesDataSource(GpuSpectrumMixin, SampleDataSource):
    def __init__(self, sample_rate: int = 44100, centre_freq: int = 0,
                 stream_factory: Optional[Callable] = None, gpu_device: int = 0):
        super().__init__(sample_rate, centre_freq)
        self.fft_size = 1024
        self.window_type = "hanning"
        self.channel_mode = "mono"
        self.stream = None
        self.running = False
        self.use_psd = False
        self._stream_factory = stream_factory
        self._gpu_device = gpu_device
        self._audio_buffer = np.zeros((self.fft_size, 2), dtype=np.float32)
        self._audio_block = self.fft_size
        self._engine_dirty = True
        self._averager._on_change = lambda mode, n: setattr(self, "_engine_dirty", True)
        self._averager._on_reset = self._gpu_reset_averager
        self.set_window()

    def set_window(self):
        funcs = {"hanning": np.hanning, "hamming": np.hamming}
        self.window = funcs.get(self.window_type, np.hanning)(self.fft_size)
        self._engine_dirty = True

    def set_fft_size(self, fft_size: int):
        self.fft_size = fft_size
        self.set_window()
        self._averager.reset()
        self._audio_buffer = np.zeros((fft_size, 2), dtype=np.float32)
        self._audio_block = fft_size
        if self.running:
            self.stop()
            self.start(None)

    @property
    def sample_count(self) -> int:
        return self.fft_size

    @sample_count.setter
    def sample_count(self, value: int):
        self.set_fft_size(value)

    def set_window_type(self, window_type: str):
        self.window_type = window_type
        self.set_window()
        if self.running:
            self.stop()
            self.start(None)

    def set_channel_mode(self, mode: str) -> None:
        if mode not in AUDIO_CHANNELS:
            logger.warning("Unknown channel mode: %s", mode)
            return
        self.channel_mode = mode

    def start(self, frequency=None):
        if self.running:
            return
        try:
            target = max(64, int(self.sample_rate * _MAX_READ_MS / 1000))
            self._audio_block = min(self.fft_size, target)
            if self._stream_factory is not None:
                self.stream = self._stream_factory(self.sample_rate, self._audio_block)
            else:
                if not _SD_AVAILABLE:
                    raise RuntimeError("sounddevice (PortAudio) is not available on this system")
                info = sd.query_devices(kind="input")
                max_ch = info.get("max_input_channels", 1) if isinstance(info, dict) else 1
                self.stream = sd.InputStream(samplerate=self.sample_rate, channels=2 if max_ch >= 2 else 1,
                                             blocksize=self._audio_block, dtype=np.float32)
            self.stream.start()
            self._audio_buffer = np.zeros((self.fft_size, 2), dtype=np.float32)
            self.running = True
        except Exception as e:
            self.running = False
            raise RuntimeError(f"Microphone initialisation failed: {e}")

    def stop(self):
        if self.stream is not None:
            try:
                self.stream.stop()
                self.stream.close()
            except Exception as e:
                logger.error("Error stopping microphone: %s", e)
            self.stream = None
        self.running = False

    @property
    def _rfft_bins(self) -> int:
        return self.fft_size // 2 + 1

    def _freq_bins(self) -> np.ndarray:
        return np.linspace(0, self.sample_rate / 2, self._rfft_bins)

    def _ready_engine(self):
        e = self._gpu_engine(self.fft_size)
        if self._engine_dirty or getattr(self, "_engine_cfg", None) != (self.use_psd, self.sample_rate):
            av = self._averager
            e.set_window(self.window.astype(np.float32))
            e.configure(db_mode="pow",
                        power_scale=1.0 / (float(self.sample_rate) * self.fft_size) if self.use_psd else 1.0,
                        log_floor=DSPConstants.LOG_FLOOR if self.use_psd else DSPConstants.POWER_LOG_FLOOR,
                        avg=(av.mode if av.is_active else "off", av.n), dc_alpha=1.0)
            self._engine_cfg = (self.use_psd, self.sample_rate)
            self._engine_dirty = False
        return e

    def _right_engine(self):
        """Second plan for the un-averaged right trace of stereo mode (audio_samples.py:161)."""
        from ..engine import SpectrumEngine
        r = getattr(self, "_engine_right", None)
        if r is None or r.nfft != self.fft_size:
            if r is not None:
                r.close()
            r = self._engine_right = SpectrumEngine(self.fft_size, max_frames=1, device=self._gpu_device)
        r.set_window(self.window.astype(np.float32))
        r.configure(db_mode="pow",
                    power_scale=1.0 / (float(self.sample_rate) * self.fft_size) if self.use_psd else 1.0,
                    log_floor=DSPConstants.LOG_FLOOR if self.use_psd else DSPConstants.POWER_LOG_FLOOR,
                    avg=("off", 1), dc_alpha=1.0)
        return r

    def get_power_levels(self):
        freq_bins = self._freq_bins()
        if not self.running:
            return np.full(self._rfft_bins, -120.0), freq_bins
        try:
            raw, _ = self.stream.read(self._audio_block)
            self._store_raw(np.array(raw, copy=True))
            raw = np.asarray(raw, dtype=np.float32)
            if raw.ndim == 1 or raw.shape[1] == 1:          # mono device: duplicate into both channels
                raw = np.repeat(raw.reshape(-1, 1), 2, axis=1)
            if self._audio_block < self.fft_size:           # low sample rate: slide the FFT window
                self._audio_buffer = np.concatenate([self._audio_buffer[len(raw):], raw], axis=0)
                frame = self._audio_buffer
            else:
                frame = raw
            e = self._ready_engine()
            averaged = self._averager.is_active
            if self.channel_mode == "stereo":
                if averaged:                                # left through the averager, right without:
                    left = e.process_real2(frame, "left", n_frames=1)[0]        # two plans, one per trace
                    right = self._right_engine().process_real2(frame, "right", n_frames=1)[0]
                    return (left.astype(np.float64), right), freq_bins
                both = e.process_real2(frame, "stereo", n_frames=1)[0]
                return (both[0], both[1]), freq_bins
            out = e.process_real2(frame, self.channel_mode, n_frames=1)[0]
            return (out.astype(np.float64) if averaged else out), freq_bins
        except Exception as e:
            logger.error("Error computing power levels: %s", e)
            return np.full(self._rfft_bins, -120.0), freq_bins

    def read_samples_only(self):
        if not self.running or self.stream is None:
            return None
        try:
            raw, _ = self.stream.read(self.fft_size)
            self._store_raw(np.array(raw, copy=True))
            return self._last_raw_samples
        except Exception as e:
            logger.error("Error reading audio samples: %s", e)
            return None

    def update_frequency(self, sample_rate: float, centre_freq: float):
        self.sample_rate = int(sample_rate)
        self.centre_freq = int(centre_freq)
        if self.running:
            self.stop()
            self.start(None)

    def update_centre_frequency(self, centre_freq: float):
        self.centre_freq = int(centre_freq)

    def set_psd_mode(self, enabled: bool):
        self.use_psd = enabled
        self._engine_dirty = True

    def __del__(self):
        try:
            self._gpu_release()
        except Exception:
            pass
