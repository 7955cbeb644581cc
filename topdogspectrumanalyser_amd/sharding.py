"""Frame sharding across the GPUs of one node.

Frames are independent, so rank r of W gets the contiguous frame range [r*F/W, (r+1)*F/W) and reads
the matching slice of the IQ stream (extended by nfft - hop samples of halo when frames overlap).
There is NO collective in the data path.  Only the tiny per-GPU trace state is combined afterwards, on
the host: max/min hold with fmax/fmin (associative), Welch / uncapped "lin" averages from per-GPU
float64 means and frame counts.  The "exp" and capped "lin" recurrences are order dependent: they run
on one GPU per trace ("replicas only").
"""
import time
from typing import Optional, Sequence, Tuple

import numpy as np


def shard_frames(n_frames: int, rank: int, world: int) -> Tuple[int, int]:
    """[f0, f1) of `rank`; ranges are contiguous, disjoint, cover [0, n_frames), sizes differ by <= 1."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"rank {rank} of world {world}")
    return (rank * n_frames) // world, ((rank + 1) * n_frames) // world


def shard_samples(f0: int, f1: int, nfft: int, hop: int) -> Tuple[int, int]:
    """Complex-sample range [s0, s1) that frames f0..f1-1 read (empty shard -> (s, s))."""
    if f1 <= f0:
        return f0 * hop, f0 * hop
    return f0 * hop, (f1 - 1) * hop + nfft


def combine_hold(parts: Sequence[np.ndarray], kind: str) -> np.ndarray:
    """Per-GPU hold traces -> one trace (np.fmax / np.fmin: NaN ignored, as the reference's hold)."""
    parts = [p for p in parts if p is not None]
    if not parts:
        raise ValueError("no hold traces to combine")
    op = {"max": np.fmax, "min": np.fmin}[kind]
    return op.reduce(np.stack(parts), axis=0)


def combine_welch(means: Sequence[np.ndarray], counts: Sequence[int]) -> Tuple[np.ndarray, int]:
    """Per-GPU running means of linear power (float64) + frame counts -> overall mean, total count."""
    total = int(sum(counts))
    if total == 0:
        raise ValueError("no frames averaged")
    acc = np.zeros_like(np.asarray(means[0], dtype=np.float64))
    for m, c in zip(means, counts):
        if c:
            acc += np.asarray(m, dtype=np.float64) * c
    return acc / total, total



def _wait_for(cond, timeout_s: float, what: str) -> None:
    """spin on a shared-memory condition: a few thousand polls flat out (the wait is normally microseconds), then yielding
    the GIL and the core; TimeoutError after timeout_s"""
    t0 = time.perf_counter()
    polls = 0
    while not cond():
        polls += 1
        if polls > 2000:
            time.sleep(0)
            if time.perf_counter() - t0 > timeout_s:
                raise TimeoutError(what)

class WelchSlab:
    """Host side of the cross-GPU Welch combine when every GPU has its own PROCESS (bench.py --config c5 --gpus N): a
    POSIX shared-memory slab every rank maps - `slots` x world partial means of n float32 / float64 values plus the
    ranks' progress counters - pinned in every process (tdsa_host_register) so that the export and the combine's
    upload are plain DMA.  No collective and no pickling: rank r's plan writes its partial into slot (step mod slots),
    publishes the step number, rank 0 waits for all of them, has ITS plan combine the slot on its device
    (SpectrumEngine.welch_combine) and publishes `done`, which frees the slot for step + slots.

    The protocol is plain stores and loads on x86 (total store order); the payload is complete before the counter moves
    because tdsa_welch_export returns after its copy has finished."""

    def __init__(self, name: Optional[str], world: int, rank: int, n: int, dtype=np.float32, slots: int = 2,
                 pin: bool = True):
        import os
        import tempfile
        import uuid
        self.world, self.rank, self.n, self.slots = int(world), int(rank), int(n), int(slots)
        self.dtype = np.dtype(dtype)
        self._hdr = 64 * (self.world + 1)                       # one cache line per counter: ready[rank], then done
        size = self._hdr + self.slots * self.world * self.n * self.dtype.itemsize
        self.owner = name is None
        if self.owner:                                          # a file in shared memory every rank maps
            base = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
            name = os.path.join(base, f"tdsa_welch_{os.getpid()}_{uuid.uuid4().hex[:8]}")
            with open(name, "wb") as fh:
                fh.truncate(size)
        self.name = name
        self._map = np.memmap(name, dtype=np.uint8, mode="r+", shape=(size,))
        self._ctr = self._map[: self._hdr].view(np.int64)
        self.parts = self._map[self._hdr:].view(self.dtype).reshape(self.slots, self.world, self.n)
        self.pinned = False
        if pin:
            try:
                from . import _native as nat
                nat.check(nat.lib.tdsa_host_register(self.parts.ctypes.data, self.parts.nbytes))
                self.pinned = True
            except Exception:
                self.pinned = False

    def part(self, step: int, timeout_s: float = 60.0) -> np.ndarray:
        """this rank's partial of `step` (1-based); waits until the slot's previous use has been combined - not for ever:
        a rank 0 that died or gave up (its wait_all timed out) must not leave the others spinning at 100 % CPU"""
        _wait_for(lambda: int(self._ctr[8 * self.world]) >= step - self.slots, timeout_s,
                  f"rank 0 did not combine step {step - self.slots}: slot of step {step} still in use")
        return self.parts[step % self.slots, self.rank]

    def publish(self, step: int) -> None:
        self._ctr[8 * self.rank] = step

    def wait_all(self, step: int, timeout_s: float = 60.0) -> np.ndarray:
        """rank 0: every rank's partial of `step` -> the slot [world, n]"""
        t0 = time.perf_counter()
        for r in range(self.world):
            while int(self._ctr[8 * r]) < step:
                if time.perf_counter() - t0 > timeout_s:
                    raise TimeoutError(f"rank {r} did not deliver its Welch partial of step {step}")
        return self.parts[step % self.slots]

    def done(self, step: int) -> None:
        self._ctr[8 * self.world] = step

    def reset(self) -> None:
        """between timed regions (all ranks at a barrier): counters back to zero"""
        if self.rank == 0:
            self._ctr[:] = 0

    def close(self) -> None:
        import os
        if self.pinned:
            try:
                from . import _native as nat
                nat.lib.tdsa_host_unregister(self.parts.ctypes.data)
            except Exception:
                pass
            self.pinned = False
        self._ctr = self.parts = self._map = None
        if self.owner:
            try:
                os.unlink(self.name)
            except OSError:
                pass
            self.owner = False


class WelchPeerSlab:
    """The same exchange with the partial means left in DEVICE memory (ranks on GPUs of one node): every rank keeps
    `slots` partials of n values in a buffer of its own GPU that other processes can map (tdsa_peer_alloc: HIP IPC),
    the 64-byte handles and the progress counters travel through a small shared-memory file, rank 0 maps every other
    rank's buffer on its device (tdsa_peer_open: refused without peer access) and its combine kernel reads them in
    place - over xGMI, each rank's over its own link - instead of W uploads through rank 0's one PCIe link.  No
    collective, no RCCL.  `ok` is False on every rank when any rank could not allocate or rank 0 could not map a
    buffer: the caller then falls back to WelchSlab (host memory).

    Header of the file, one 64-byte line each: ready[rank] ..., done, verdict, then per rank: status (int64), device
    (int64), handle (64 bytes)."""

    def __init__(self, name: Optional[str], world: int, rank: int, n: int, device: int, dtype=np.float32,
                 slots: int = 2, timeout_s: float = 60.0):
        import os
        import tempfile
        import uuid
        from . import _native as nat
        self._nat = nat
        self.world, self.rank, self.n, self.slots, self.device = int(world), int(rank), int(n), int(slots), int(device)
        self.dtype = np.dtype(dtype)
        self._ctr_lines = self.world + 2
        size = 64 * self._ctr_lines + 128 * self.world
        self.owner = name is None
        if self.owner:
            base = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
            name = os.path.join(base, f"tdsa_welch_peer_{os.getpid()}_{uuid.uuid4().hex[:8]}")
            with open(name, "wb") as fh:
                fh.truncate(size)
        self.name = name
        self._map = np.memmap(name, dtype=np.uint8, mode="r+", shape=(size,))
        self._ctr = self._map[: 64 * self._ctr_lines].view(np.int64)
        self._info = self._map[64 * self._ctr_lines:].reshape(self.world, 128)
        self._mine = None                      # this rank's device buffer [slots][n]
        self._peers = [None] * self.world      # rank 0: every rank's buffer as mapped on ITS device
        self.ok = False
        self._timeout = float(timeout_s)

    # -- set-up: two phases with the caller's barrier between them ------------------------------------------------
    def allocate(self) -> None:
        """phase 1, every rank: its buffer and the handle into the file"""
        import ctypes as C
        nat = self._nat
        ptr, handle = C.c_void_p(), (C.c_ubyte * 64)()
        line = self._info[self.rank]
        try:
            nat.check(nat.lib.tdsa_peer_alloc(self.device, self.slots * self.n * self.dtype.itemsize, C.byref(ptr), handle))
            self._mine = int(ptr.value)
            line[16:80] = np.frombuffer(bytes(handle), dtype=np.uint8)
            line[8:16].view(np.int64)[0] = self.device
            line[0:8].view(np.int64)[0] = 1
        except Exception:
            line[0:8].view(np.int64)[0] = -1

    def connect(self) -> None:
        """phase 2 (after a barrier), rank 0: map every other rank's buffer; the verdict for everybody"""
        import ctypes as C
        nat = self._nat
        verdict = self._ctr[8 * (self.world + 1): 8 * (self.world + 1) + 1]
        if self.rank == 0:
            good = all(int(self._info[r][0:8].view(np.int64)[0]) == 1 for r in range(self.world))
            if good:
                self._peers[0] = self._mine
                for r in range(1, self.world):
                    line = self._info[r]
                    handle = (C.c_ubyte * 64).from_buffer_copy(bytes(line[16:80]))
                    ptr = C.c_void_p()
                    owner = int(line[8:16].view(np.int64)[0])
                    try:
                        nat.check(nat.lib.tdsa_peer_open(self.device, handle, owner, C.byref(ptr)))
                        self._peers[r] = int(ptr.value)
                    except Exception:
                        good = False
                        break
            verdict[0] = 1 if good else -1
        t0 = time.perf_counter()
        while int(verdict[0]) == 0:
            if time.perf_counter() - t0 > self._timeout:
                raise TimeoutError("rank 0 did not report on the peer buffers")
        self.ok = int(verdict[0]) == 1

    # -- per step ---------------------------------------------------------------------------------------------------
    def part_ptr(self, step: int) -> int:
        """device pointer of this rank's partial of `step` (1-based); waits until the slot's previous use is combined"""
        _wait_for(lambda: int(self._ctr[8 * self.world]) >= step - self.slots, self._timeout,
                  f"rank 0 did not combine step {step - self.slots}: slot of step {step} still in use")
        return self._mine + (step % self.slots) * self.n * self.dtype.itemsize

    def publish(self, step: int) -> None:
        self._ctr[8 * self.rank] = step

    def wait_all(self, step: int):
        """rank 0: every rank's partial of `step` as device pointers on rank 0's GPU"""
        t0 = time.perf_counter()
        for r in range(self.world):
            while int(self._ctr[8 * r]) < step:
                if time.perf_counter() - t0 > self._timeout:
                    raise TimeoutError(f"rank {r} did not deliver its Welch partial of step {step}")
        off = (step % self.slots) * self.n * self.dtype.itemsize
        return [p + off for p in self._peers]

    def done(self, step: int) -> None:
        self._ctr[8 * self.world] = step

    def reset(self) -> None:
        if self.rank == 0:
            self._ctr[: 8 * (self.world + 1)] = 0

    def close(self) -> None:
        import os
        nat = self._nat
        for r in range(1, self.world):
            if self._peers[r]:
                nat.lib.tdsa_peer_close(self.device, self._peers[r])
        self._peers = [None] * self.world
        if self._mine:
            nat.lib.tdsa_peer_free(self.device, self._mine)
            self._mine = None
        self._ctr = self._info = self._map = None
        if self.owner:
            try:
                os.unlink(self.name)
            except OSError:
                pass
            self.owner = False


def process_sharded(iq: np.ndarray, nfft: int, hop: int, devices: Sequence[int], window: np.ndarray,
                    hold: str = "", **configure):
    """One capture over several GPUs from ONE process: a thread and a SpectrumEngine per entry of `devices`
    (the C-ABI calls release the GIL), each taking its contiguous frame range plus halo; returns
    (dB rows [frames, nfft] in capture order, combined max-hold trace or None, combined min-hold or None).

    Only order-independent modes can be sharded: no "exp" / capped "lin" averaging and no tracked DC remover
    (dc_alpha in [0, 1) carries dc_state across frames) - those raise ValueError: replicas only.  A
    device may appear more than once - two plans then share that GPU - which is also how this is tested
    on a single-GPU box.  `configure` goes to SpectrumEngine.configure; `hold` is "", "max", "min" or "maxmin".
    """
    import threading
    from .engine import SpectrumEngine

    if configure.get("avg", ("off", 1))[0] != "off":
        raise ValueError("order-dependent averaging cannot be sharded (SURVEY.md 8(e): replicas only)")
    if 0.0 <= float(configure.get("dc_alpha", 1.0)) < 1.0:
        raise ValueError("the tracked DC remover (0 <= dc_alpha < 1) carries state from frame to frame and cannot "
                         "be sharded: use dc_alpha >= 1 (per-frame mean) or < 0 (off)")
    iq = np.ascontiguousarray(iq)
    per_sample = 1 if np.iscomplexobj(iq) else 2           # interleaved bytes: two array elements per sample
    n_samples = iq.size // per_sample
    n_frames = 0 if n_samples < nfft else (n_samples - nfft) // hop + 1
    world = len(devices)
    rows = np.empty((n_frames, nfft), dtype=np.float32)
    holds = [None] * world
    errors = [None] * world

    def work(rank: int) -> None:
        try:
            f0, f1 = shard_frames(n_frames, rank, world)
            if f1 <= f0:
                return
            s0, s1 = shard_samples(f0, f1, nfft, hop)
            with SpectrumEngine(nfft, max_frames=f1 - f0, device=devices[rank]) as eng:
                eng.set_window(window)
                eng.configure(hold_max="max" in hold, hold_min="min" in hold, **configure)
                rows[f0:f1] = eng.process(iq[per_sample * s0: per_sample * s1], hop=hop, n_frames=f1 - f0)
                holds[rank] = eng.hold()
        except Exception as exc:                          # surfaced by the caller's thread
            errors[rank] = exc

    threads = [threading.Thread(target=work, args=(r,), name=f"tdsa-shard-{r}") for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for exc in errors:
        if exc is not None:
            raise exc
    got = [h for h in holds if h is not None]
    mx = combine_hold([h[0] for h in got], "max") if "max" in hold and got else None
    mn = combine_hold([h[1] for h in got], "min") if "min" in hold and got else None
    return rows, mx, mn
