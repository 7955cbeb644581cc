import ctypes as C, sys, time, numpy as np
sys.path.insert(0, ".")
from topdogspectrumanalyser_amd import SpectrumEngine, _native as nat, analytics as an
from topdogspectrumanalyser_amd.utils.synthetic import synth_iq_int8
n, hop, ns = 16384, 8192, 20_000_000
F = (ns - n) // hop + 1
iq = synth_iq_int8(ns, n, seed=3)
d_in, d_out = C.c_void_p(), C.c_void_p()
nat.check(nat.lib.tdsa_dev_alloc(0, iq.nbytes, C.byref(d_in))); nat.check(nat.lib.tdsa_dev_alloc(0, F * n * 4, C.byref(d_out)))
nat.check(nat.lib.tdsa_memcpy_h2d(0, d_in, iq.ctypes.data_as(C.c_void_p), iq.nbytes))
w = np.hanning(n).astype(np.float32); w /= np.sqrt(np.mean(w ** 2))
with SpectrumEngine(n, max_frames=F) as e:
    e.set_window(w); e.configure(db_mode="mag", log_floor=1e-12, dc_alpha=1.0)
    e.process_device(nat.IN_I8, d_in.value, ns, hop, F, d_out.value); e.synchronize()
    def t(f, reps=5):
        f(); t0 = time.perf_counter()
        for _ in range(reps): f()
        return (time.perf_counter() - t0) / reps * 1e3
    for npk in (1, 2, 3, 5, 8):
        print("n_peaks", npk, "%.3f ms" % t(lambda: an.rows_top_peaks(e, d_out.value, F, n=npk)))
    for rows in (256, 512, 1024, 2440):
        print("rows", rows, "%.3f ms" % t(lambda: an.rows_top_peaks(e, d_out.value, rows, n=5)))
    print("marker", "%.3f ms" % t(lambda: an.rows_marker_peaks(e, d_out.value, F)))
    print("stats", "%.3f ms" % t(lambda: an.rows_stats(e, d_out.value, F)))
    for d in (1, 3):
        for thr in (-200.0, -20.0):
            print("marker distance", d, "threshold", thr, "%.3f ms" % t(lambda: an.rows_marker_peaks(e, d_out.value, F, distance=d, peak_threshold=thr)))
    print("marker 256 rows", "%.3f ms" % t(lambda: an.rows_marker_peaks(e, d_out.value, 256)))
