#!/usr/bin/env python3
"""Developer experiment: do consecutive frame-kernel launches overlap usefully when issued on two streams?
Two plans (each with its own stream) take alternate steps; compares wall time per step with one plan."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from topdogspectrumanalyser_amd import SpectrumEngine, _native as nat  # noqa: E402


def main():
    n, hop, F, steps = 16384, 8192, 2440, 200
    ns = hop * (F - 1) + n
    rng = np.random.default_rng(0)
    ring = 4
    ins, outs = [], []
    for r in range(ring):
        iq = rng.integers(-100, 100, size=2 * ns, dtype=np.int8)
        di, do = C.c_void_p(), C.c_void_p()
        nat.check(nat.lib.tdsa_dev_alloc(0, iq.nbytes, C.byref(di)))
        nat.check(nat.lib.tdsa_dev_alloc(0, F * n * 4, C.byref(do)))
        nat.check(nat.lib.tdsa_memcpy_h2d(0, di, iq.ctypes.data_as(C.c_void_p), iq.nbytes))
        ins.append(di.value); outs.append(do.value)
    engs = []
    for _ in range(3):
        e = SpectrumEngine(n, max_frames=F)
        e.set_window(np.hanning(n).astype(np.float32))
        e.configure(db_mode="mag", log_floor=1e-12, dc_alpha=1.0, hold_max=True)
        engs.append(e)
    for k in (1, 2, 3):
        use = engs[:k]
        for i in range(10):
            use[i % k].process_device(nat.IN_I8, ins[i % ring], ns, hop, F, outs[i % ring])
        for e in use:
            e.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            use[i % k].process_device(nat.IN_I8, ins[i % ring], ns, hop, F, outs[i % ring])
        for e in use:
            e.synchronize()
        dt = (time.perf_counter() - t0) / steps
        print(f"{k} stream(s): {dt*1e6:.1f} us per step -> {F/dt/1e6:.2f} Mframes/s")


if __name__ == "__main__":
    main()
