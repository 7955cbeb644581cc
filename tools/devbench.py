#!/usr/bin/env python3
"""Developer micro-benchmark for the frame kernel (not the contract bench: see bench.py).

python tools/devbench.py [--nfft 16384] [--hop 8192] [--frames 2440] [--steps 20] [--hold 1] [--fmt i8]
"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from topdogspectrumanalyser_amd import SpectrumEngine, _native as nat  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nfft", type=int, default=16384)
    ap.add_argument("--hop", type=int, default=0)
    ap.add_argument("--frames", type=int, default=2440)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--hold", type=int, default=1)
    ap.add_argument("--nodb", type=int, default=0)
    ap.add_argument("--mode", default="mag")
    ap.add_argument("--batch", type=int, default=1, help="captures per call (tdsa_process_dev_batch), distinct buffers")
    ap.add_argument("--streams", type=int, default=1)
    ap.add_argument("--fmt", default="i8", choices=["i8", "c64"])
    ap.add_argument("--knob", action="append", default=[], help="name=value for tdsa_debug_knob (repeatable)")
    ap.add_argument("--stats", default="", help="per-frame scalars from the epilogue: 'none' (peak / argmax only) or lo:hi (band bins)")
    ap.add_argument("--rows-stats", type=int, default=0, help="1: tdsa_rows_stats on the rows after every call instead (the second pass)")
    a = ap.parse_args()
    n, hop, F = a.nfft, (a.hop or a.nfft // 2), a.frames
    ns = hop * (F - 1) + n
    rng = np.random.default_rng(0)
    iq = rng.integers(-100, 100, size=2 * ns, dtype=np.int8)
    if a.fmt == "c64":
        iq = (iq.astype(np.float32) / 128.0)
    fmt = nat.IN_I8 if a.fmt == "i8" else nat.IN_C64
    sb = 2 if a.fmt == "i8" else 8
    dev_in, dev_out = C.c_void_p(), C.c_void_p()
    B = max(1, a.batch)
    nat.check(nat.lib.tdsa_dev_alloc(0, iq.nbytes * B, C.byref(dev_in)))
    nat.check(nat.lib.tdsa_dev_alloc(0, F * n * 4 * B, C.byref(dev_out)))
    for b in range(B):
        nat.check(nat.lib.tdsa_memcpy_h2d(0, C.c_void_p(dev_in.value + b * iq.nbytes), iq.ctypes.data_as(C.c_void_p), iq.nbytes))
    e = SpectrumEngine(n, max_frames=F)
    w = np.hanning(n).astype(np.float32)
    e.set_window(w)
    e.configure(db_mode=a.mode, log_floor=1e-12, dc_alpha=1.0, hold_max=bool(a.hold & 1), hold_min=bool(a.hold & 2))
    for kv in a.knob:
        k, v = kv.split("=")
        e.debug_knob(k, int(v))
    out_ptr = None if a.nodb else dev_out.value
    band = None
    if a.stats:
        band = None if a.stats == "none" else tuple(int(x) for x in a.stats.split(":"))
        if not a.rows_stats:
            e.set_frame_stats(True, band)
    pk, pb, bd = np.empty(F * B, np.float32), np.empty(F * B, np.int32), np.empty(F * B, np.float64)
    e.set_overlap(a.streams)

    def call():
        if B == 1:
            e.process_device(fmt, dev_in.value, ns, hop, F, out_ptr)
        else:
            e.process_device_batch(fmt, dev_in.value, iq.nbytes, B, ns, hop, F, out_ptr, F * n)
        if a.rows_stats:      # the second pass over the rows (synchronous: launch + read-back)
            lo, hi = band if band else (0, -1)
            nat.check(nat.lib.tdsa_rows_stats(e._h, C.c_void_p(dev_out.value), F * B, n, lo, hi, 1.0,
                                              pk.ctypes.data_as(C.c_void_p), pb.ctypes.data_as(C.c_void_p),
                                              bd.ctypes.data_as(C.c_void_p) if band else None))
    for _ in range(max(1, a.warmup // B)):
        call()
    e.synchronize()
    calls = max(1, a.steps // B)
    if a.streams == 1:
        e.timer_begin()
    t0 = time.perf_counter()
    for _ in range(calls):
        call()
    if a.streams == 1:
        ms = e.timer_end()
    else:
        e.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
    t1 = time.perf_counter()
    per = ms / (calls * B)
    fps = F / (per * 1e-3)
    bytes_per_frame = sb * hop + 4 * n
    inf = e.info()
    print(f"lib={os.path.basename(nat.LIB_PATH)} {' '.join(a.knob)} stats={a.stats or '-'}{' (rows pass)' if a.rows_stats else ''} fmt={a.fmt} N={n} hop={hop} F={F} batch={B} streams={a.streams} hold={a.hold} grid={inf.grid}x{inf.block} lds={inf.lds_bytes} "
          f"step={per*1e3:.1f} us  {fps/1e6:.3f} Mframes/s  {fps*bytes_per_frame/1e12:.3f} TB/s algorithmic "
          f"({fps*bytes_per_frame/8e12*100:.1f}% of 8 TB/s)  host wall {((t1-t0)/(calls*B))*1e6:.1f} us/step")


if __name__ == "__main__":
    main()
