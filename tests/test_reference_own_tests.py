"""The reference's OWN tests, run against this package.

The reference ships script-style tests for the path's host surface: test_fft_size_changes.py,
test_fft_size_detection.py, test_rbw_calculation.py (the three sample sources: constructors, size setters, window
length, RBW = fs / N), test_duty_cycle.py (DutyCycleAnalyser) and the TraceAverager / DataProcessor / TareState
checks inside test_smoke.py (test_smoke.py:137-175, 222-272, 299-310).  Here those files are executed where they
lie under /root/reference - nothing of them is copied - with the reference's module names (`datasources.*`,
`utils.signal_processing`, `core.display_data_processor`, `core.duty_cycle`, `core.display_manager.TareState`)
bound to this package's modules, i.e. exactly what a maintainer who swaps the package in would run.

CPU only; skipped where the reference tree does not exist (the GPU box).  None of these tests computes a spectrum
(the reference mocks its hardware and FFT libraries in them), so no GPU is needed.
"""
import ast
import contextlib
import importlib
import importlib.util
import os
import sys
import types

import numpy as np
import pytest

REF = os.environ.get("TDSA_REFERENCE", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "test_smoke.py")),
                                reason="reference tree not present")

PKG = "topdogspectrumanalyser_amd"
# reference module name -> module of this package
ALIASES = {
    "datasources": f"{PKG}.datasources",
    "datasources.base": f"{PKG}.datasources.base",
    "datasources.hackrf_samples": f"{PKG}.datasources.hackrf_samples",
    "datasources.rtl_samples": f"{PKG}.datasources.rtl_samples",
    "datasources.audio_samples": f"{PKG}.datasources.audio_samples",
    "utils": f"{PKG}.utils",
    "utils.signal_processing": f"{PKG}.utils.signal_processing",
    "utils.constants": f"{PKG}.utils.constants",
    "core": f"{PKG}.core",
    "core.display_data_processor": f"{PKG}.core.display_data_processor",
    "core.duty_cycle": f"{PKG}.core.duty_cycle",
    "core.tare_state": f"{PKG}.core.tare_state",
}


class _SweepStandIn:
    """test_rbw_calculation.py also touches the two SWEEP sources (out of scope here): all it reads of
    them is the bin size they were constructed with."""

    def __init__(self, start_freq, stop_freq, bin_size):
        self.start_freq, self.stop_freq, self.bin_size = start_freq, stop_freq, bin_size


@contextlib.contextmanager
def reference_names():
    """Bind the reference's module names to this package for the duration of one test; sys.modules is put back
    exactly as it was (the reference's test files also plant MagicMocks for libraries they find missing)."""
    import scipy.fft      # noqa: F401  real scipy first: the files only mock what is not imported yet
    import scipy.signal   # noqa: F401
    saved = dict(sys.modules)
    old_flag, sys.dont_write_bytecode = sys.dont_write_bytecode, True
    try:
        for name, target in ALIASES.items():
            sys.modules[name] = importlib.import_module(target)
        shim = types.ModuleType("core.display_manager")
        shim.TareState = sys.modules["core.tare_state"].TareState        # reference: re-exported there
        sys.modules["core.display_manager"] = shim
        for name, cls in (("datasources.hackrf_sweep", "HackRFSweepDataSource"),
                          ("datasources.rtl_sweep", "RtlSweepDataSource")):
            m = types.ModuleType(name)
            setattr(m, cls, type(cls, (_SweepStandIn,), {}))
            sys.modules[name] = m
        yield
    finally:
        sys.dont_write_bytecode = old_flag
        for k in list(sys.modules):
            if k not in saved:
                del sys.modules[k]
        sys.modules.update(saved)


def _run_test_file(filename: str) -> int:
    """Execute a reference test file in place and call every test_* function it defines."""
    path = os.path.join(REF, filename)
    spec = importlib.util.spec_from_file_location("_ref_" + filename[:-3], path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    names = [n for n, v in vars(mod).items() if n.startswith("test_") and callable(v)]
    for n in names:
        getattr(mod, n)()
    return len(names)


@pytest.mark.parametrize("filename", ["test_fft_size_changes.py", "test_fft_size_detection.py",
                                      "test_rbw_calculation.py", "test_duty_cycle.py"])
def test_reference_test_file_passes_on_this_package(filename, capsys):
    with reference_names():
        assert _run_test_file(filename) >= 1
    capsys.readouterr()       # the scripts narrate what they do


# test_smoke.py runs its checks at import time and covers the whole application; the checks of THIS path are
# taken out of it by name (function definitions compiled from the file's own syntax tree, executed here)
# (_test_trace_averager_exp / _reset average on the device - the product has no CPU path - and the reference tree
#  does not exist on the GPU box: tests/test_gpu_parity.py::test_trace_averager_as_the_reference_smoke_checks_it
#  restates those two there)
SMOKE_CHECKS = ["_test_trace_averager_passthrough", "_test_data_processor_methods", "_test_find_top_peaks",
                "_test_nan_safe", "_test_tare_state"]


@pytest.mark.parametrize("check", SMOKE_CHECKS)
def test_reference_smoke_check_passes_on_this_package(check):
    path = os.path.join(REF, "test_smoke.py")
    with open(path) as f:
        tree = ast.parse(f.read(), filename=path)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == check]
    assert len(fn) == 1, f"{check} not found in the reference's test_smoke.py"
    code = compile(ast.Module(body=fn, type_ignores=[]), path, "exec")
    ns = {"np": np}
    with reference_names():
        exec(code, ns)
        ns[check]()
