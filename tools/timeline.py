#!/usr/bin/env python3
"""Developer tool: per-phase s_memtime timeline of workgroup 0 (needs libtdsa_hip_tl.so, a -DTDSA_TIMELINE build)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("TDSA_HIP_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..",
                                                   "topdogspectrumanalyser_amd", "libtdsa_hip_tl.so"))
from topdogspectrumanalyser_amd import SpectrumEngine, _native as nat  # noqa: E402

NAMES = ["top", "pre-B0", "post-B0", "converted", "p1 stored", "post-B1", "gathered(issue)", "p2 stored",
         "post-B2", "gather2 issued", "dif3 done", "epilogue done"]


def main():
    hold = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n, hop, F = 16384, 8192, 2440
    ns = hop * (F - 1) + n
    rng = np.random.default_rng(0)
    iq = rng.integers(-100, 100, size=2 * ns, dtype=np.int8)
    dev_in, dev_out = C.c_void_p(), C.c_void_p()
    nat.check(nat.lib.tdsa_dev_alloc(0, iq.nbytes, C.byref(dev_in)))
    nat.check(nat.lib.tdsa_dev_alloc(0, F * n * 4, C.byref(dev_out)))
    nat.check(nat.lib.tdsa_memcpy_h2d(0, dev_in, iq.ctypes.data_as(C.c_void_p), iq.nbytes))
    e = SpectrumEngine(n, max_frames=F)
    e.set_window(np.hanning(n).astype(np.float32))
    e.configure(db_mode="mag", log_floor=1e-12, dc_alpha=1.0, hold_max=bool(hold & 1), hold_min=bool(hold & 2))
    fn = nat.lib._handle  # noqa
    dbg = C.CDLL(nat.LIB_PATH).tdsa_debug_timeline
    dbg.argtypes = [C.c_void_p, C.c_void_p]
    dbg.restype = C.c_int
    for _ in range(3):
        e.process_device(nat.IN_I8, dev_in.value, ns, hop, F, dev_out.value)
    e.synchronize()
    nat.check(dbg(e._h, None))
    e.process_device(nat.IN_I8, dev_in.value, ns, hop, F, dev_out.value)
    out = np.zeros(2048, dtype=np.uint64)
    nat.check(dbg(e._h, out.ctypes.data_as(C.c_void_p)))
    raw = out.reshape(8, 16, 16).astype(np.int64)
    t = raw[:, :, :12]                                        # [frame][wave][stamp]
    entry, ready, left, end = (raw[2, :, k] for k in (12, 13, 14, 15))
    if entry.min() > 0:
        e0 = entry.min()
        print("launch level, workgroup 0 (cycles since its first wave started):")
        print(f"  waves start        {int((entry - e0).min()):8d} .. {int((entry - e0).max())}")
        print(f"  prologue done      {int((ready - e0).min()):8d} .. {int((ready - e0).max())}   (twiddle table + seeds + first frame's bytes fetched)")
        print(f"  third frame top    {int((raw[2, :, 0] - e0).min()):8d} .. {int((raw[2, :, 0] - e0).max())}")
        print(f"  frame loop left    {int((left - e0).min()):8d} .. {int((left - e0).max())}")
        print(f"  wave ends          {int((end - e0).min()):8d} .. {int((end - e0).max())}   (hold traces merged)")
        per = np.diff(raw[:, 0, 0])
        print(f"  frame periods of wave 0: {per.tolist()}")
    t0 = t[0, :, 0].min()
    print("frame-to-frame period (wave 0, cycles):", np.diff(t[:, 0, 0]))
    for f in (2, 3):
        print(f"-- frame {f}: cycles since frame top (rows: waves 0..15)")
        print("   " + " ".join(f"{s:>9.9s}" for s in NAMES))
        for w in range(16):
            print(f"w{w} " + " ".join(f"{int(x - t[f, w, 0]):9d}" for x in t[f, w]))
    d = np.diff(t[2:7], axis=2).mean(axis=(0, 1))
    print("mean phase durations (frames 2..6, all waves):")
    for i, x in enumerate(d):
        print(f"  {NAMES[i]:>16s} -> {NAMES[i+1]:<16s} {x:9.0f}")
    print("  sum", d.sum(), " period", np.diff(t[2:7, :, 0], axis=0).mean())


if __name__ == "__main__":
    main()
