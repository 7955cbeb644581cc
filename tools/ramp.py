#!/usr/bin/env python3
"""Developer tool: C3-shaped launches in batches of 600, time per step against elapsed time - shows the ~50 ms an
idle MI355X needs before its clocks settle (profiles/r02_c3_experiments.txt), and the same after a 2 s pause."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from topdogspectrumanalyser_amd import SpectrumEngine, _native as nat
n, hop, F = 16384, 8192, 2440
ns = hop * (F - 1) + n
iq = np.random.default_rng(0).integers(-100, 100, size=2 * ns, dtype=np.int8)
dev_in, dev_out = C.c_void_p(), C.c_void_p()
nat.check(nat.lib.tdsa_dev_alloc(0, iq.nbytes, C.byref(dev_in)))
nat.check(nat.lib.tdsa_dev_alloc(0, F * n * 4, C.byref(dev_out)))
nat.check(nat.lib.tdsa_memcpy_h2d(0, dev_in, iq.ctypes.data_as(C.c_void_p), iq.nbytes))
e = SpectrumEngine(n, max_frames=F)
e.set_window(np.hanning(n).astype(np.float32))
e.configure(db_mode="mag", log_floor=1e-12, dc_alpha=1.0, hold_max=True)
t00 = time.perf_counter()
for b in range(60):
    e.timer_begin()
    for _ in range(600):
        e.process_device(nat.IN_I8, dev_in.value, ns, hop, F, dev_out.value)
    ms = e.timer_end()
    print(f"t={time.perf_counter()-t00:6.2f}s  batch {b:2d}: {ms/600*1e3:.1f} us/step")
time.sleep(2.0)
for b in range(5):
    e.timer_begin()
    for _ in range(600):
        e.process_device(nat.IN_I8, dev_in.value, ns, hop, F, dev_out.value)
    ms = e.timer_end()
    print(f"after 2 s idle: batch {b}: {ms/600*1e3:.1f} us/step")
