"""TraceAverager - same interface as the reference's utils/signal_processing.py:5-73, with the buffer
and the recurrence living on the GPU (tdsa_trace_avg_* in include/tdsa_hip.h).

    off / n <= 1 : process() returns its argument unchanged (pass-through, no device work)
    exp          : buf <- buf*(1 - 1/n) + x/n          (first frame: buf <- x)
    lin          : running mean capped at n frames, then EMA with alpha = 1/n

The GPU sample sources do NOT call process(): they hand (mode, n) to their SpectrumEngine so the
averaging stays fused with the FFT batch.  This class serves the host-array users of the reference
API: DataProcessor's sweep averager and third-party sources.
"""
from typing import Optional

import numpy as np


class TraceAverager:
    def __init__(self, device: int = 0):
        self._mode: str = "off"
        self._n: int = 1
        self._device = device
        self._state = None            # engine.TraceState, created on first use / shape change
        self._shape = None
        self._count: int = 0
        self._on_change = None        # hooks: GPU sources mirror (mode, n) / resets into their engine
        self._on_reset = None

    def set_mode(self, mode: str, n: int) -> None:
        if mode not in ("off", "exp", "lin"):
            raise ValueError(f"unknown averaging mode {mode!r}")
        self._mode = mode
        self._n = max(1, int(n))
        if self._on_change is not None:
            self._on_change(self._mode, self._n)
        self.reset()

    def reset(self) -> None:
        self._count = 0
        if self._state is not None:
            self._state.avg_set_mode(self._mode, self._n)   # set_mode on the device object resets it
        if self._on_reset is not None:
            self._on_reset()

    def process(self, linear_power: np.ndarray) -> np.ndarray:
        if self._mode == "off" or self._n <= 1:
            return linear_power
        x = np.asarray(linear_power)
        if self._state is None or self._shape != x.shape:
            from ..engine import TraceState
            if self._state is not None:
                self._state.close()
            self._state = TraceState(int(x.size), device=self._device)
            self._state.avg_set_mode(self._mode, self._n)
            self._shape = x.shape
            self._count = 0
        out = self._state.avg_process(x.ravel()).reshape(x.shape)
        if self._count == 0:
            self._count = 1
        elif self._mode == "lin" and self._count < self._n:
            self._count += 1
        return out

    @property
    def is_active(self) -> bool:
        return self._mode != "off" and self._n > 1

    @property
    def mode(self) -> str:
        return self._mode

    @property
    def n(self) -> int:
        return self._n
