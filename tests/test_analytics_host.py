"""Host-side logic of topdogspectrumanalyser_amd.analytics that needs no GPU: band -> bin range, the
duty-cycle bookkeeping (against the sequence captured from the reference) and the result adapters."""
import os

import numpy as np
import pytest

from oracle import analytics_oracle as ao


@pytest.fixture(scope="module")
def an():
    from topdogspectrumanalyser_amd import analytics
    return analytics


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "analytics.npz"))


def test_band_bin_range_equals_the_reference_mask(an, gold):
    bins = gold["band_bins"]
    rng = np.random.default_rng(0)
    edges = [tuple(e) for e in gold["band_edges"]]
    edges += [(float(bins[10]), float(bins[10])), (float(bins[0]) - 1.0, float(bins[0])), (float(bins[-1]), 1e12)]
    edges += [tuple(rng.uniform(bins[0] - 1e5, bins[-1] + 1e5, size=2)) for _ in range(200)]
    for a, b in edges:
        assert an.band_bin_range(bins, a, b) == ao.band_bin_range(bins, a, b), (a, b)


def test_duty_cycle_host_entry_matches_reference_sequence(an, gold):
    d = an.DutyCycle()
    for i, fr in enumerate(gold["duty_frames"]):
        d.update_from_power(fr, threshold_dbm=-60.0 if i < 130 else -45.0)
        assert d.duty_pct == gold["duty_pct"][i]
        on, off = gold["duty_on"][i], gold["duty_off"][i]
        assert (d.on_power_dbm is None and np.isnan(on)) or d.on_power_dbm == on
        assert (d.off_power_dbm is None and np.isnan(off)) or d.off_power_dbm == off
    d.update_from_power(None)
    d.update_from_power(np.array([]))
    assert len(d._envelope) == an.DutyCycle.BUFFER_FRAMES
    d.reset()
    assert d.duty_pct == 0.0 and d.on_power_dbm is None and d.off_power_dbm is None


def test_peaks_as_reference_adapter(an):
    fb = np.linspace(1e6, 2e6, 8)
    bins = np.array([3, 6, -1, -1, -1], dtype=np.int32)
    db = np.array([-10.5, -20.25, np.nan, np.nan, np.nan], dtype=np.float32)
    assert an.peaks_as_reference(fb, bins, db) == [(float(fb[3]), -10.5), (float(fb[6]), -20.25)]


def test_amplitude_axis_constants(an):
    assert (an.AMP_BINS, an.AMP_MIN, an.AMP_RNG) == (ao.AMP_BINS, ao.AMP_MIN, ao.AMP_RNG)
