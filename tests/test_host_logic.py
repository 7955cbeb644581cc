"""Host-side logic of the path's callers (no GPU): the display processor's pass-through of an untouched frame and
the sharding front end's refusal of order-dependent modes.  (Round 4: the zero-span trigger, EVM and peak-list read-out
checks that lived here until round 3 were GUI feeds - SURVEY.md section 2 marks them out of scope - and are gone, with
their fixture; tests/test_reference_differential.py still compares those host helpers with the imported reference.)"""
import types

import numpy as np
import pytest

from topdogspectrumanalyser_amd.core.display_data_processor import DataProcessor


class Label:
    text = None

    def setText(self, s):
        self.text = s


def _bare(mw, dm):
    dp = DataProcessor.__new__(DataProcessor)         # no device objects: these paths never touch the GPU
    dp.mw, dp.dm = mw, dm
    dp._fused = None
    dp._sweeps_since_axis_refresh = 0
    dp.reference_hold_alias = False
    return dp


def test_sharding_rejects_order_dependent_modes():
    from topdogspectrumanalyser_amd.sharding import process_sharded
    iq = np.zeros(4096, dtype=np.int8)
    w = np.ones(1024, dtype=np.float32)
    with pytest.raises(ValueError, match="averaging"):
        process_sharded(iq, 1024, 1024, [0, 0], w, avg=("exp", 4))
    with pytest.raises(ValueError, match="DC remover"):
        process_sharded(iq, 1024, 1024, [0, 0], w, dc_alpha=0.25)


def test_plain_display_frame_makes_no_device_call():
    """with no calibration offset, no tare run / baseline and no hold wanted, a displayed frame passes through
    _process_sample_data untouched, as in the reference (display_data_processor.py:153-183) - and WITHOUT a device call:
    on this GPU-less box any tdsa_* call would raise, so completing at all proves it; the frame object itself becomes
    the live trace."""
    from topdogspectrumanalyser_amd.core.tare_state import TareState
    trace = np.linspace(-90.0, -20.0, 512).astype(np.float32)
    axis = np.linspace(88e6, 108e6, 512)
    src = types.SimpleNamespace(get_power_levels=lambda: (trace, axis))
    cal = types.SimpleNamespace(get_offset=lambda kind: 0.0)
    mw = types.SimpleNamespace(current_source=src, live_power_levels=None, max_power_levels=None, min_power_levels=None,
                               frequency_bins=None, min_hold_enabled=False, tare_active=False, baseline_power_levels=None,
                               calibration_manager=cal, source_manager=types.SimpleNamespace(last_source_type="hackrf_samples"),
                               status_label=Label())
    dm = types.SimpleNamespace(tare_state=TareState(), max_peak_search_enabled=False, duty_cycle_enabled=False,
                               peak_list_enabled=False)
    dp = _bare(mw, dm)
    dp._state, dp._device = None, 0
    for _ in range(3):
        dp._process_sample_data()
    assert mw.live_power_levels is trace and mw.frequency_bins is axis
    assert mw.max_power_levels is None and mw.min_power_levels is None and dp._state is None
