#!/usr/bin/env python3
"""Developer tool: C3-shaped step with the per-frame mean (dc_alpha = 1) against the tracked DC remover (dc_alpha < 1)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from topdogspectrumanalyser_amd import SpectrumEngine, _native as nat  # noqa: E402

n, hop, F = 16384, 8192, 2440
ns = hop * (F - 1) + n
iq = np.random.default_rng(0).integers(-100, 100, size=2 * ns, dtype=np.int8)
di, do = C.c_void_p(), C.c_void_p()
nat.check(nat.lib.tdsa_dev_alloc(0, iq.nbytes, C.byref(di)))
nat.check(nat.lib.tdsa_dev_alloc(0, F * n * 4, C.byref(do)))
nat.check(nat.lib.tdsa_memcpy_h2d(0, di, iq.ctypes.data_as(C.c_void_p), iq.nbytes))
e = SpectrumEngine(n, max_frames=F)
e.set_window(np.hanning(n).astype(np.float32))
for alpha in (1.0, 0.05):
    e.configure(db_mode="mag", log_floor=1e-12, dc_alpha=alpha, hold_max=True)
    for _ in range(300):
        e.process_device(nat.IN_I8, di.value, ns, hop, F, do.value)
    e.synchronize()
    e.timer_begin()
    for _ in range(1000):
        e.process_device(nat.IN_I8, di.value, ns, hop, F, do.value)
    print("dc_alpha", alpha, "step %.1f us" % (e.timer_end() / 1000 * 1e3))
