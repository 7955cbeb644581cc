"""Shared GPU plumbing of the sample sources: one SpectrumEngine per (source, FFT size)."""
from typing import Optional

import numpy as np

from .. import _native as nat
from ..engine import SpectrumEngine
from ..utils.constants import DSPConstants, gpu_fft_size_supported


class GpuSpectrumMixin:
    """Mix-in for SampleDataSource subclasses.  The subclass sets, before the first frame:
       self._gpu_device, and calls _gpu_configure(...) whenever a DSP knob changes."""

    _engine: Optional[SpectrumEngine] = None
    _engine_n: int = 0
    _gpu_device: int = 0

    def _gpu_release(self) -> None:
        if self._engine is not None:
            self._engine.close()
        self._engine = None
        self._engine_n = 0

    def _gpu_engine(self, nfft: int) -> SpectrumEngine:
        if not gpu_fft_size_supported(nfft):
            raise ValueError(f"FFT size {nfft}: the device library plans any size from 2 to 2^20")
        if self._engine is None or self._engine_n != nfft:
            # the DC estimate is a property of the source, not of an FFT size (the reference keeps
            # self._dc_estimate across set_num_samples): it moves to the new plan
            carried = self._engine.dc_estimate if self._engine is not None else None
            self._gpu_release()
            self._engine = SpectrumEngine(nfft, max_frames=1, device=self._gpu_device)
            self._engine_n = nfft
            self._engine_dirty = True
            if carried:
                self._engine.dc_estimate = carried
        return self._engine

    def _gpu_configure(self, nfft: int, window: np.ndarray, *, branch: str, use_psd: bool,
                       sample_rate: float, dc_alpha: float) -> SpectrumEngine:
        """branch: 'hackrf' (mag dB unless PSD/averaging) or 'rtl' (always power dB).
        Floors and dB forms follow hackrf_samples.py:374-383 / rtl_samples.py:175-184."""
        e = self._gpu_engine(nfft)
        av = self._averager
        if use_psd:
            db_mode, scale, floor = "pow", 1.0 / (float(sample_rate) * nfft), DSPConstants.LOG_FLOOR
        elif branch == "rtl" or av.is_active:
            db_mode, scale, floor = "pow", 1.0, DSPConstants.POWER_LOG_FLOOR
        else:
            db_mode, scale, floor = "mag", 1.0, DSPConstants.LOG_FLOOR
        e.set_window(window)
        e.configure(db_mode=db_mode, power_scale=scale, log_floor=floor,
                    avg=(av.mode if av.is_active else "off", av.n), dc_alpha=dc_alpha)
        self._engine_dirty = False
        return e

    def _gpu_frame(self, samples: np.ndarray) -> np.ndarray:
        """One frame through the HIP path.  Averaged traces come back float64 like the reference's."""
        e = self._engine
        x = np.ascontiguousarray(samples, dtype=np.complex64)
        out = e.process(x, hop=e.nfft, n_frames=1)[0]
        if self._averager.is_active:
            return out.astype(np.float64)
        return out

    def _gpu_reset_averager(self) -> None:
        if self._engine is not None:
            self._engine.reset(nat.RESET_AVG)

    def _gpu_reset_dc(self) -> None:
        if self._engine is not None:
            self._engine.reset(nat.RESET_DC)
