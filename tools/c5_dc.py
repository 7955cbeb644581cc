import ctypes as C, os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from topdogspectrumanalyser_amd import SpectrumEngine, _native as nat
n, K = 1 << 20, 64
iq = np.random.default_rng(0).integers(-100, 100, size=2 * n * K, dtype=np.int8)
di, do = C.c_void_p(), C.c_void_p()
nat.check(nat.lib.tdsa_dev_alloc(0, iq.nbytes, C.byref(di)))
nat.check(nat.lib.tdsa_dev_alloc(0, n * 4, C.byref(do)))
nat.check(nat.lib.tdsa_memcpy_h2d(0, di, iq.ctypes.data_as(C.c_void_p), iq.nbytes))
for dc in (-1.0, 1.0, 0.25):
    e = SpectrumEngine(n, max_frames=K)
    e.set_window(np.hanning(n).astype(np.float32))
    e.configure(db_mode="pow", power_scale=1.0, log_floor=1e-12, dc_alpha=dc, avg=("lin", K))
    def step():
        e.reset(nat.RESET_AVG); e.process_device(nat.IN_I8, di.value, n * K, n, K, do.value)
    for _ in range(20): step()
    e.synchronize(); e.timer_begin()
    for _ in range(200): step()
    print("dc_alpha", dc, "%.1f us per capture" % (e.timer_end() / 200 * 1e3), flush=True)
    e.close()
