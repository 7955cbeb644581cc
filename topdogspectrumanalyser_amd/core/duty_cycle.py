"""Duty cycle of a pulsed signal over the last ~2 s of frames - the analyser the display processor feeds when
"duty cycle" is switched on (reference: core/duty_cycle.py:10-80, called from
core/display_data_processor.py:397-405).

One value per frame goes into a 100-deep envelope: the frame's peak dB (update_from_power, what the spectrum
path uses) or the mean power of its raw samples (update).  Frames at or above the threshold are "on"; the duty
cycle is their share, on/off power the mean of each side.  Same attributes, same numbers and the same readout
string as the reference's class; update_from_rows() is the addition of this build: the per-frame peaks of a whole
batch of dB rows come from the device (tdsa_rows_stats) without the rows being read back.
"""
from collections import deque
from typing import Optional

import numpy as np

_ON_COLOUR, _OFF_COLOUR = "#00ff88", "#888888"


def _label(text: str) -> str:
    return f'<span style="color:white;font-weight:bold;">{text}</span>'


def _value(text: str, colour: str) -> str:
    return f'<span style="color:{colour};">{text}</span>'


class DutyCycleAnalyser:
    BUFFER_FRAMES = 100          # ~2 s at the GUI's 20 ms timer

    def __init__(self):
        self._envelope = deque(maxlen=self.BUFFER_FRAMES)
        self.threshold_dbm = -60.0
        self.duty_pct = 0.0
        self.on_power_dbm: Optional[float] = None
        self.off_power_dbm: Optional[float] = None

    # -- feeding -----------------------------------------------------------------------------------------
    def update(self, samples, threshold_dbm: float) -> None:
        """One frame of raw samples (complex IQ or real): its mean power in dB is the envelope value."""
        if samples is None or len(samples) == 0:
            return
        self.threshold_dbm = threshold_dbm
        if np.iscomplexobj(samples):
            mean_power = np.mean(np.abs(samples) ** 2)
        else:
            mean_power = np.mean(samples.ravel() ** 2)
        self._push(float(10.0 * np.log10(mean_power + 1e-30)))

    def update_from_power(self, power_levels_db, threshold_dbm: Optional[float] = None) -> None:
        """One dB spectrum: its peak is the envelope value."""
        if power_levels_db is None or len(power_levels_db) == 0:
            return
        if threshold_dbm is not None:
            self.threshold_dbm = threshold_dbm
        self._push(float(np.max(power_levels_db)))

    def update_from_rows(self, engine, rows_dev: int, n_rows: int, threshold_dbm: Optional[float] = None) -> None:
        """update_from_power once per dB row of a device-resident batch; only the per-row peaks leave the GPU."""
        from ..analytics import rows_stats
        if threshold_dbm is not None:
            self.threshold_dbm = threshold_dbm
        peaks, _, _ = rows_stats(engine, rows_dev, n_rows)
        for peak in peaks:
            self._push(float(peak))

    # -- state -------------------------------------------------------------------------------------------
    def _push(self, value_db: float) -> None:
        self._envelope.append(value_db)
        self._recompute(self.threshold_dbm)

    def _recompute(self, threshold_dbm: float) -> None:
        if not self._envelope:
            return
        env = np.array(self._envelope)
        on = env >= threshold_dbm
        n_on = int(np.sum(on))
        self.duty_pct = 100.0 * n_on / len(env)
        self.on_power_dbm = float(np.mean(env[on])) if n_on > 0 else None
        self.off_power_dbm = float(np.mean(env[~on])) if n_on < len(env) else None

    def reset(self) -> None:
        self._envelope.clear()
        self.duty_pct = 0.0
        self.on_power_dbm = None
        self.off_power_dbm = None

    # -- readout -----------------------------------------------------------------------------------------
    def get_readout(self) -> str:
        """The marker-readout fragment the GUI shows (empty until a frame has been seen)."""
        if not self._envelope:
            return ""
        on = "—" if self.on_power_dbm is None else f"{self.on_power_dbm:.1f} dBm"
        off = "—" if self.off_power_dbm is None else f"{self.off_power_dbm:.1f} dBm"
        return (f"{_label('Duty:')} {_value(f'{self.duty_pct:.1f}%', _ON_COLOUR)}  "
                f"{_label('On:')} {_value(on, _ON_COLOUR)}  "
                f"{_label('Off:')} {_value(off, _OFF_COLOUR)}")
