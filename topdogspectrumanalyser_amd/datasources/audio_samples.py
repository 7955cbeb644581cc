"""Sound-card sample source with the DSP on the MI355X.

API and observable behaviour are those of the reference's MicrophoneSamplesDataSource
(datasources/audio_samples.py): a stereo float32 input stream read in ~30 ms blocks, a sliding
fft_size-long window over those blocks when the block is shorter than the FFT (low sample rates), per
channel: mean removal, Hann or Hamming window, real FFT, one-sided power with every bin but DC and
Nyquist doubled, `10*log10(P + 1e-10)` or PSD `10*log10(P/(fs*N) + 1e-12)`; channel selection
mono / left / right / stereo, where stereo returns a (left, right) pair and only the LEFT trace goes
through the averager; a source that is not running, or a block that fails, yields a flat -120 dB row.

On the GPU every real signal a tick needs (mono mix, left, right) is transformed on its own as signal + 0i
(`real_select_kernel` / `real_fold_kernel`, `tdsa_process_real2` in include/tdsa_hip.h): a loud channel leaves
nothing in a quiet one.  The stream object comes from
`sounddevice` when it is installed, else from `stream_factory(sample_rate, blocksize)` - anything with
start / stop / close / read(n) -> (frames, overflowed).
"""
import logging
from typing import Callable, Optional, Tuple

import numpy as np

from ..engine import SpectrumEngine
from ..utils.constants import DSPConstants, gpu_real_input_size_supported
from ._gpu import GpuSpectrumMixin
from .base import SampleDataSource

try:
    import sounddevice as sd  # type: ignore
except (ImportError, OSError):              # PortAudio missing
    sd = None

log = logging.getLogger(__name__)

CHANNEL_MODES = ("mono", "left", "right", "stereo")
READ_BLOCK_MS = 30                          # how long one blocking stream read may take
SILENCE_DB = -120.0
_TAPERS = {"hanning": np.hanning, "hamming": np.hamming}


class MicrophoneSamplesDataSource(GpuSpectrumMixin, SampleDataSource):
    def __init__(self, sample_rate: int = 44100, centre_freq: int = 0,
                 stream_factory: Optional[Callable] = None, gpu_device: int = 0):
        super().__init__(sample_rate, centre_freq)
        self.stream = None
        self.running = False
        self.use_psd = False
        self.fft_size = 1024
        self.window_type = "hanning"
        self.channel_mode = "mono"
        self._stream_factory = stream_factory
        self._gpu_device = gpu_device
        self._engine_dirty = True
        self._engine_right: Optional[SpectrumEngine] = None
        self._averager._on_change = lambda mode, n: setattr(self, "_engine_dirty", True)
        self._averager._on_reset = self._gpu_reset_averager
        self._reset_history()
        self.set_window()

    # ------------------------------------------------------------------ geometry
    @property
    def sample_count(self) -> int:
        return self.fft_size

    @sample_count.setter
    def sample_count(self, value: int):
        self.set_fft_size(value)

    @property
    def _n_out(self) -> int:                 # length of a one-sided spectrum
        return self.fft_size // 2 + 1

    def _axis(self) -> np.ndarray:
        return np.linspace(0, self.sample_rate / 2, self._n_out)

    # the reference's names for the two (audio_samples.py:113-119), for code that reaches for them
    _rfft_bins = _n_out
    _freq_bins = _axis

    def _reset_history(self) -> None:
        self._audio_buffer = np.zeros((self.fft_size, 2), dtype=np.float32)
        self._audio_block = self.fft_size

    def _restart_if_running(self) -> None:
        if self.running:
            self.stop()
            self.start(None)

    # ------------------------------------------------------------------ settings
    def set_window(self):
        self.window = _TAPERS.get(self.window_type, np.hanning)(self.fft_size)
        self._engine_dirty = True

    def set_window_type(self, window_type: str):
        self.window_type = window_type
        self.set_window()
        self._restart_if_running()

    def set_fft_size(self, fft_size: int):
        self.fft_size = fft_size
        self.set_window()
        self._averager.reset()
        self._reset_history()
        self._restart_if_running()

    def set_channel_mode(self, mode: str) -> None:
        if mode in CHANNEL_MODES:
            self.channel_mode = mode
        else:
            log.warning("ignoring unknown channel mode %r", mode)

    def set_psd_mode(self, enabled: bool):
        self.use_psd = enabled
        self._engine_dirty = True

    def update_centre_frequency(self, centre_freq: float):
        self.centre_freq = int(centre_freq)

    def update_frequency(self, sample_rate: float, centre_freq: float):
        self.sample_rate, self.centre_freq = int(sample_rate), int(centre_freq)
        self._restart_if_running()

    # ------------------------------------------------------------------ stream lifetime
    def _open_stream(self):
        if self._stream_factory is not None:
            return self._stream_factory(self.sample_rate, self._audio_block)
        if sd is None:
            raise RuntimeError("sounddevice (PortAudio) is not available on this system")
        dev = sd.query_devices(kind="input")
        inputs = dev.get("max_input_channels", 1) if isinstance(dev, dict) else 1
        return sd.InputStream(samplerate=self.sample_rate, channels=2 if inputs >= 2 else 1,
                              blocksize=self._audio_block, dtype=np.float32)

    def start(self, frequency=None):
        if self.running:
            return
        try:
            per_read = max(64, self.sample_rate * READ_BLOCK_MS // 1000)
            self._audio_block = min(self.fft_size, int(per_read))
            self.stream = self._open_stream()
            self.stream.start()
        except Exception as exc:
            self.running = False
            raise RuntimeError(f"Microphone initialisation failed: {exc}")
        self._audio_buffer = np.zeros((self.fft_size, 2), dtype=np.float32)
        self.running = True

    def stop(self):
        stream, self.stream, self.running = self.stream, None, False
        if stream is not None:
            try:
                stream.stop()
                stream.close()
            except Exception as exc:
                log.error("stopping the audio stream failed: %s", exc)

    # ------------------------------------------------------------------ GPU plans
    def _db_settings(self) -> dict:
        if self.use_psd:
            return dict(db_mode="pow", power_scale=1.0 / (float(self.sample_rate) * self.fft_size),
                        log_floor=DSPConstants.LOG_FLOOR, dc_alpha=1.0)
        return dict(db_mode="pow", power_scale=1.0, log_floor=DSPConstants.POWER_LOG_FLOOR, dc_alpha=1.0)

    def _main_plan(self) -> SpectrumEngine:
        if not gpu_real_input_size_supported(self.fft_size):      # at plan time, not as a device error on every frame
            raise ValueError(f"FFT size {self.fft_size}: the real-input path plans powers of two up to 16384 and any other "
                             "size up to 2^19")
        eng = self._gpu_engine(self.fft_size)
        key = (self.use_psd, self.sample_rate)
        if self._engine_dirty or getattr(self, "_engine_cfg", None) != key:
            avg = self._averager
            eng.set_window(self.window.astype(np.float32))
            eng.configure(avg=(avg.mode if avg.is_active else "off", avg.n), **self._db_settings())
            self._engine_cfg, self._engine_dirty = key, False
        return eng

    def _plain_plan(self) -> SpectrumEngine:
        """A second plan without averaging: the right trace of stereo mode never passes the averager."""
        eng = self._engine_right
        if eng is None or eng.nfft != self.fft_size:
            if eng is not None:
                eng.close()
            eng = self._engine_right = SpectrumEngine(self.fft_size, max_frames=1, device=self._gpu_device)
        eng.set_window(self.window.astype(np.float32))
        eng.configure(avg=("off", 1), **self._db_settings())
        return eng

    # ------------------------------------------------------------------ frames
    def _next_frame(self) -> np.ndarray:
        """[fft_size, 2] float32: the newest block, slid into the history when blocks are short."""
        block, _overflow = self.stream.read(self._audio_block)
        self._store_raw(np.array(block, copy=True))
        block = np.asarray(block, dtype=np.float32)
        if block.ndim == 1 or block.shape[1] == 1:          # mono device: same signal on both channels
            block = np.repeat(block.reshape(-1, 1), 2, axis=1)
        if self._audio_block >= self.fft_size:
            return block
        self._audio_buffer = np.concatenate([self._audio_buffer[len(block):], block], axis=0)
        return self._audio_buffer

    def _traces(self, frame: np.ndarray):
        plan, averaging = self._main_plan(), self._averager.is_active
        if self.channel_mode != "stereo":
            trace = plan.process_real2(frame, self.channel_mode, n_frames=1)[0]
            return trace.astype(np.float64) if averaging else trace
        if not averaging:
            left, right = plan.process_real2(frame, "stereo", n_frames=1)[0]
            return left, right
        left = plan.process_real2(frame, "left", n_frames=1)[0]
        right = self._plain_plan().process_real2(frame, "right", n_frames=1)[0]
        return left.astype(np.float64), right

    def get_power_levels(self) -> Tuple[object, np.ndarray]:
        axis = self._axis()
        if self.running:
            try:
                return self._traces(self._next_frame()), axis
            except Exception as exc:
                log.error("audio frame failed: %s", exc)
        return np.full(self._n_out, SILENCE_DB), axis

    def read_samples_only(self):
        if self.stream is None or not self.running:
            return None
        try:
            block, _overflow = self.stream.read(self.fft_size)
            self._store_raw(np.array(block, copy=True))
        except Exception as exc:
            log.error("audio read failed: %s", exc)
            return None
        return self._last_raw_samples

    def __del__(self):
        try:
            if self._engine_right is not None:
                self._engine_right.close()
            self._gpu_release()
        except Exception:
            pass
