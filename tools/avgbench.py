import sys, os, time, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from topdogspectrumanalyser_amd import SpectrumEngine, _native as nat
n, hop, F = 16384, 8192, 2440
ns = hop*(F-1)+n
iq = np.random.default_rng(0).integers(-100,100,size=2*ns,dtype=np.int8)
di, do = C.c_void_p(), C.c_void_p()
nat.check(nat.lib.tdsa_dev_alloc(0, iq.nbytes, C.byref(di))); nat.check(nat.lib.tdsa_dev_alloc(0, F*n*4, C.byref(do)))
nat.check(nat.lib.tdsa_memcpy_h2d(0, di, iq.ctypes.data_as(C.c_void_p), iq.nbytes))
e = SpectrumEngine(n, max_frames=F); e.set_window(np.hanning(n).astype(np.float32))
for avg in (("exp", 8), ("lin", 64)):
    e.configure(db_mode="pow", power_scale=1.0, log_floor=1e-10, dc_alpha=1.0, avg=avg, hold_max=True)
    for _ in range(2): e.process_device(nat.IN_I8, di.value, ns, hop, F, do.value)
    e.synchronize(); e.timer_begin()
    for _ in range(5): e.process_device(nat.IN_I8, di.value, ns, hop, F, do.value)
    print(avg, "step %.1f us" % (e.timer_end()/5*1e3))
