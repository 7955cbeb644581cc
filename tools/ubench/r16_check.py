#!/usr/bin/env python3
"""Parity + timing of the r16 prototype kernel (tools/ubench/r16_proto.hip) against numpy and against the
production frame kernel, same input, C3 shape."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import spectrum_oracle as so  # noqa: E402
from topdogspectrumanalyser_amd import SpectrumEngine, _native as nat  # noqa: E402


def dev(arr):
    p = C.c_void_p()
    nat.check(nat.lib.tdsa_dev_alloc(0, arr.nbytes, C.byref(p)))
    nat.check(nat.lib.tdsa_memcpy_h2d(0, p, arr.ctypes.data_as(C.c_void_p), arr.nbytes))
    return p


def main():
    lib = C.CDLL(os.path.join(HERE, "libr16.so"))
    lib.r16_run.argtypes = [C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                            C.POINTER(C.c_float)]
    n, hop, F = 16384, 8192, 2440
    ns = hop * (F - 1) + n
    iq = so.synth_iq_int8(ns, n, seed=3)
    w = so.hackrf_window(n)
    tw = np.exp(-2j * np.pi * np.arange(n) / n).astype(np.complex64)
    d_in, d_w, d_tw = dev(iq), dev((w / 128.0).astype(np.float32)), dev(tw)
    out = np.zeros((F, n), dtype=np.float32)
    hold = np.full(n, -np.inf, dtype=np.float32)
    d_out, d_hold = dev(out), dev(hold)
    ms = C.c_float()
    rc = lib.r16_run(d_in, 2 * hop, F, d_w, d_tw, d_out, d_hold, 200, C.byref(ms))
    assert rc == 0, rc
    nat.check(nat.lib.tdsa_memcpy_d2h(0, out.ctypes.data_as(C.c_void_p), d_out, out.nbytes))
    nat.check(nat.lib.tdsa_memcpy_d2h(0, hold.ctypes.data_as(C.c_void_p), d_hold, hold.nbytes))
    picks = [0, 1, 7, F // 2, F - 1]
    x = so.unpack_iq_int8(iq)
    gold = so.HackrfBranchOracle(n, 20e6, precision="gold")
    worst = (0.0, 0.0)
    for k in picks:
        g = np.asarray(gold.power_levels(x[k * hop: k * hop + n]))
        rel, ddb = so.parity_metrics(out[k], g)
        worst = (max(worst[0], rel), max(worst[1], ddb))
    print(f"r16 prototype: {ms.value * 1e3:.1f} us per launch of {F} frames; parity rel {worst[0]:.2e} "
          f"dB/allowance {worst[1] / 1e-3:.2f}; hold == column max: {np.array_equal(hold, out.max(axis=0))}")
    e = SpectrumEngine(n, max_frames=F)
    e.set_window(w)
    e.configure(db_mode="mag", log_floor=1e-12, dc_alpha=1.0, hold_max=True)
    for _ in range(20):
        e.process_device(nat.IN_I8, d_in.value, ns, hop, F, d_out.value)
    e.synchronize()
    e.timer_begin()
    for _ in range(200):
        e.process_device(nat.IN_I8, d_in.value, ns, hop, F, d_out.value)
    print(f"production kernel: {e.timer_end() / 200 * 1e3:.1f} us per launch")
    ref = np.empty_like(out)
    nat.check(nat.lib.tdsa_memcpy_d2h(0, ref.ctypes.data_as(C.c_void_p), d_out, ref.nbytes))
    print("max |dB| difference prototype vs production, strong bins:", float(np.abs(ref - out)[ref > ref.max() - 60].max()))


if __name__ == "__main__":
    main()
