import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import spectrum_oracle as so
from topdogspectrumanalyser_amd import SpectrumEngine
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 17
nfft = 1 << log2n
w = so.rtl_window("hanning", nfft).astype(np.float64)
for k in (2, 3, 4, 5, 6, 7, 8, 12):
    iq = so.synth_iq_int8(nfft * k, nfft, seed=5 + k)
    x = so.unpack_iq_int8(iq).astype(np.complex128)
    acc = np.zeros(nfft)
    per = []
    for s in range(k):
        p = np.abs(np.fft.fftshift(np.fft.fft(x[s * nfft:(s + 1) * nfft] * w))) ** 2
        per.append(p); acc += p
    with SpectrumEngine(nfft, max_frames=k) as e:
        e.set_window(w.astype(np.float32))
        e.configure(db_mode="pow", power_scale=1.0, log_floor=so.POWER_LOG_FLOOR, dc_alpha=-1.0, avg=("lin", k))
        e.process(iq, hop=nfft)
        mean, cnt = e.averaged()
    rel = np.max(np.abs(mean - acc / k)) / (acc / k).max()
    print(f"N=2^{log2n} k={k}: count {cnt} rel err {rel:.3e}")
