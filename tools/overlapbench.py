#!/usr/bin/env python3
"""Developer experiment: do consecutive frame-kernel launches overlap usefully when issued on two streams?
Two plans (each with its own stream) take alternate steps; compares wall time per step with one plan."""
import ctypes as C
import os as _os
if _os.environ.get("TDSA_TORCH_FIRST"):
    import torch  # noqa: F401  (same HIP runtime as bench.py)
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
        if os.environ.get("AVG"):             # AVG=exp,4: the batched TraceAverager path (frame kernel -> chain -> re-scan)
            mode, cnt = os.environ["AVG"].split(",")
            e.configure(db_mode="pow", power_scale=1.0, log_floor=1e-10, dc_alpha=1.0, avg=(mode, int(cnt)))
        else:
            e.configure(db_mode="mag", log_floor=1e-12, dc_alpha=1.0, hold_max=bool(int(os.environ.get("HOLD", "1"))))
        engs.append(e)
    def run_plans(k):
        use = engs[:k]
        for e in use:
            e.set_overlap(1)
        for i in range(steps):
            use[i % k].process_device(nat.IN_I8, ins[i % ring], ns, hop, F, outs[i % ring])
        for e in use:
            e.synchronize()

    def run_overlap(k):
        e = engs[0]
        e.set_overlap(k)
        for i in range(steps):
            e.process_device(nat.IN_I8, ins[i % ring], ns, hop, F, outs[i % ring])
        e.synchronize()

    if os.environ.get("AVG"):
        configs = [("1 plan", lambda: run_plans(1)), ("2 plans", lambda: run_plans(2)), ("3 plans", lambda: run_plans(3))]
    else:
      configs = [("1 plan, 1 stream", lambda: run_overlap(1)), ("1 plan, set_overlap(2)", lambda: run_overlap(2)),
                 ("1 plan, set_overlap(3)", lambda: run_overlap(3)), ("1 plan, set_overlap(4)", lambda: run_overlap(4)),
                 ("2 plans", lambda: run_plans(2)), ("3 plans", lambda: run_plans(3))]
    for _ in range(5):                       # warm the clocks up
        run_overlap(1)
    acc = {name: [] for name, _ in configs}
    for rep in range(4):                     # interleaved repetitions: drift hits every configuration alike
        for name, fn in configs:
            t0 = time.perf_counter()
            fn()
            acc[name].append((time.perf_counter() - t0) / steps)
    for name, _ in configs:
        v = np.array(acc[name]) * 1e6
        print(f"{name:24s}: median {np.median(v):6.1f} us per step (min {v.min():.1f}, max {v.max():.1f}) -> "
              f"{F/np.median(v):.2f} Mframes/s")


if __name__ == "__main__":
    main()
