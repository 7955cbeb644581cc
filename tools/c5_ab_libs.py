#!/usr/bin/env python3
"""Developer A/B of library variants INSIDE one process (the long-frame chain trades a few us between its passes from
process to process: profiles/r04_c5_experiments.txt): every library given is loaded on its own (ctypes, RTLD_LOCAL), gets
its own plan, and the plans take turns - `--rounds` rounds of `--steps` captures each, per K.
python tools/c5_ab_libs.py --ks 8,64 name=path [name=path ...]"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
from topdogspectrumanalyser_amd import _native as nat  # noqa: E402  (the default library: allocations + signatures)


def load(path):
    lib = C.CDLL(path, mode=os.RTLD_LOCAL)
    for name, (res, args) in nat._SIGNATURES.items():
        fn = getattr(lib, name, None)       # (an older library lacks the newer entry points)
        if fn is not None:
            fn.restype, fn.argtypes = res, args
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--ks", default="8,64")
    ap.add_argument("--steps", type=int, default=150)
    ap.add_argument("--rounds", type=int, default=5)
    a = ap.parse_args()
    ks = [int(k) for k in a.ks.split(",")]
    n, kmax = 1 << 20, max(ks)
    iq = np.random.default_rng(0).integers(-100, 100, size=2 * n * kmax, dtype=np.int8)
    di, do = C.c_void_p(), C.c_void_p()
    nat.check(nat.lib.tdsa_dev_alloc(0, iq.nbytes * 2, C.byref(di)))
    nat.check(nat.lib.tdsa_dev_alloc(0, n * 4 * 2, C.byref(do)))
    for r in range(2):
        nat.check(nat.lib.tdsa_memcpy_h2d(0, C.c_void_p(di.value + r * iq.nbytes), iq.ctypes.data_as(C.c_void_p), iq.nbytes))
    libs = [(s.split("=")[0], load(os.path.abspath(s.split("=")[1]))) for s in a.libs]
    w = np.hanning(n).astype(np.float32)
    for K in ks:
        plans = []
        for name, lib in libs:
            h = C.c_void_p()
            assert lib.tdsa_create(0, n, K, C.byref(h)) == 0
            assert lib.tdsa_set_window(h, w.ctypes.data_as(C.c_void_p), n) == 0
            m = nat.Mode(nat.DB_POW, 1.0, 1e-12, nat.AVG_LIN, K, -1.0, -0.8087, 0)
            assert lib.tdsa_set_mode(h, C.byref(m)) == 0
            plans.append((name, lib, h))

        def run(lib, h, steps):
            for i in range(steps):
                r = i % 2
                lib.tdsa_reset_state(h, nat.RESET_AVG)
                rc = lib.tdsa_process_dev(h, nat.IN_I8, C.c_void_p(di.value + r * iq.nbytes), n * K, n, K,
                                          C.c_void_p(do.value + r * n * 4))
                assert rc == 0, lib.tdsa_last_error_string()
            lib.tdsa_synchronize(h)
        res = {name: [] for name, _, _ in plans}
        steps = max(20, a.steps * 64 // K // 4)
        for name, lib, h in plans:
            run(lib, h, 20)
        for _ in range(a.rounds):
            for name, lib, h in plans:
                t0 = time.perf_counter()
                run(lib, h, steps)
                res[name].append((time.perf_counter() - t0) / steps * 1e6)
        for name, lib, h in plans:
            v = res[name]
            print(f"K={K:3d} {name:12s} median {np.median(v):7.1f} us  " + " ".join(f"{x:6.1f}" for x in v), flush=True)
            lib.tdsa_destroy(h)


if __name__ == "__main__":
    main()
