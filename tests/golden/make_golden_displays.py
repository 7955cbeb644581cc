#!/usr/bin/env python3
"""Golden vectors for the display accumulators from the *imported reference*.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_displays.py

displays/density_display.py and displays/waterfall.py need PyQt6 / pyqtgraph, which are absent here: they
are imported against stub modules made of plain do-nothing classes (no mocks in the class hierarchy), so
the real DensityDisplay / Waterfall objects are constructed and their own numpy code runs:
DensityDisplay._update_hist (density_display.py:306-320) and Waterfall.update_widget_data ->
_add_row / _display_view (waterfall.py:163-180, 305-356).  Writes tests/golden/displays.npz: seeded dB
rows in, histogram / ring-buffer contents out.  DATA only.
"""
import os
import sys
import types

sys.dont_write_bytecode = True
REF = os.environ.get("TDSA_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
HERE = os.path.dirname(os.path.abspath(__file__))

import numpy as np  # noqa: E402


class Stub:
    """A do-nothing GUI object: any constructor arguments, any method, chained calls."""

    def __init__(self, *a, **k):
        pass

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return lambda *a, **k: Stub()

    def __call__(self, *a, **k):
        return Stub()


def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Qt:
    class PenStyle:
        DashLine, SolidLine = 2, 1

    class AlignmentFlag:
        AlignCenter = 0


qtcore = _module("PyQt6.QtCore", Qt=_Qt, QRectF=type("QRectF", (Stub,), {}), QTimer=Stub, pyqtSignal=Stub)
qtwidgets = _module("PyQt6.QtWidgets", QWidget=type("QWidget", (Stub,), {}),
                    QVBoxLayout=type("QVBoxLayout", (Stub,), {}), QLabel=Stub)
qtgui = _module("PyQt6.QtGui", QColor=Stub, QFont=Stub)
_module("PyQt6", QtCore=qtcore, QtWidgets=qtwidgets, QtGui=qtgui)
_module("pyqtgraph", AxisItem=type("AxisItem", (Stub,), {}), PlotWidget=Stub, ImageItem=Stub, PlotCurveItem=Stub,
        InfiniteLine=Stub, GraphicsLayoutWidget=Stub, ColorMap=Stub, colormap=Stub(), mkPen=Stub(), mkBrush=Stub(),
        TextItem=Stub, ScatterPlotItem=Stub)

from displays.density_display import DensityDisplay  # noqa: E402
from displays.waterfall import Waterfall  # noqa: E402


def db_rows(rng, n_rows, n):
    """dB rows with periodogram statistics, a slowly moving tone, and the awkward values the histogram has
    to deal with: NaN, -inf, values below -200 dBm and above +100 dBm."""
    k = np.arange(n)
    rows = []
    for r in range(n_rows):
        p = rng.exponential(1.0, size=n) * 1e-9
        p += 1e-3 * np.sinc((k - n / 3 - 0.21 * r) / 1.4) ** 2
        row = (10 * np.log10(p + 1e-12)).astype(np.float32)
        if r % 5 == 1:
            row[rng.integers(0, n, 3)] = np.nan
        if r % 7 == 2:
            row[rng.integers(0, n, 2)] = -250.0
            row[rng.integers(0, n, 2)] = 150.0
            row[rng.integers(0, n, 1)] = -np.inf
        if r % 9 == 4:
            row[rng.integers(0, n, 4)] = np.float32(-200.0)     # exactly on the lower edge
            row[rng.integers(0, n, 4)] = np.float32(100.0)      # exactly on the upper edge (excluded)
        rows.append(row)
    return np.stack(rows)


def main():
    rng = np.random.default_rng(20240919)
    out = {}

    # ---- density histogram: three decay settings, same 48 rows ------------------------------------
    n = 256
    rows = db_rows(rng, 48, n)
    fb = np.linspace(99e6, 101e6, n)
    out["density_rows"] = rows
    for mode in ("medium", "fast", "off"):
        d = DensityDisplay()
        d.set_decay(mode)
        snaps = []
        for r, row in enumerate(rows):
            with np.errstate(invalid="ignore"):
                d._update_hist(row, fb)
            if r in (0, 7, 47):
                snaps.append(d._hist.copy())
        out[f"density_{mode}_decay"] = np.float64(d._decay)
        out[f"density_{mode}_hist"] = np.stack(snaps)          # after rows 0, 7, 47
    out["density_snap_rows"] = np.array([0, 7, 47])

    # ---- waterfall ring: 40 updates with repeated rows (the 20 ms timer outruns the source) --------
    w = Waterfall()
    w.wf_time_span, w.seconds_per_row = 1.2, 0.1                # -> 12 history lines
    nb = 64
    fbw = np.linspace(2.40e9, 2.48e9, nb)
    src_rows = db_rows(rng, 16, nb)
    src_rows = np.nan_to_num(src_rows, nan=-120.0, neginf=-300.0)
    order = [0, 0, 1, 2, 2, 2, 3, 4, 4, 5, 6, 7, 7, 8, 9, 9, 9, 10, 11, 12, 12, 13, 14, 15, 15, 0, 1, 1, 2, 3,
             3, 4, 5, 6, 6, 7, 8, 9, 10, 10]
    views, ptrs, added = [], [], []
    for step, idx in enumerate(order):
        before = None if w._last_row is None else w._last_row.copy()
        w.update_widget_data(src_rows[idx], None, fbw)
        added.append(before is None or not np.array_equal(before, src_rows[idx]))
        ptrs.append(w._ptr)
        if step in (0, 5, 12, 26, 39):
            views.append(w._display_view().copy())
    out["wf_rows"] = src_rows.astype(np.float32)
    out["wf_order"] = np.array(order)
    out["wf_history_lines"] = np.int64(w.history_lines)
    out["wf_min_db"] = np.float64(w.wf_min_db)
    out["wf_ptr"] = np.array(ptrs)
    out["wf_added"] = np.array(added)
    out["wf_view_steps"] = np.array([0, 5, 12, 26, 39])
    out["wf_views"] = np.stack(views)

    np.savez_compressed(os.path.join(HERE, "displays.npz"), **out)
    print("wrote displays.npz: density", rows.shape, "x3 decays; waterfall", len(order), "updates, H =", w.history_lines)


if __name__ == "__main__":
    main()
