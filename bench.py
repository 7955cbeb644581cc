#!/usr/bin/env python3
"""bench.py - PSD frames/s of the MI355X-native IQ -> spectrum path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]        (any N: spawns one worker per GPU itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W        (same result)

Workload (config C3 of BASELINE.json / SURVEY.md 8(d), the one the metric is quoted on):
HackRF-shaped 20 Msps int8 IQ, N = 16384, hop = N/2, one second of IQ per step = 20e6 samples
-> 2440 frames, HackRF-branch semantics (per-frame DC removal, power-normalised Hann,
20*log10(|X| + 1e-12)), every frame's dB row written + a max-hold trace.  One "step" = one pass of
the hot path over that second; consecutive steps walk a ring of distinct seconds (input + output of the ring
are several times the 256 MiB Infinity Cache) so the traffic really goes to HBM.

Three ways of submitting the same K steps are timed, all in the line:
  value          `--batch` queued seconds per call (tdsa_process_dev_batch: one persistent launch consumes them;
                 the per-launch costs - cold fetch of window / twiddles, hold merge, ragged last round of frames
                 over the CUs - are paid once per launch instead of once per second), calls rotating over
                 `--streams` plan-owned HIP streams
  value_streams  one second per launch, launches rotating over the streams (round 2's `value`)
  value_serial   one second per launch, strictly serial full-chip launches
Results are bit-identical in all three (tests/test_gpu_parity.py::test_batched_captures_*).

The default run (C3) then appends SHORT legs of the other three GPU configurations of BASELINE.json - C2, C4 as named
(65 536 frames x 8192 points) and C5 - each with its own parity block, under `roofline.other_configs`; at N > 1 they are
sharded over the same ranks (C4's frames and C5's segments split over the world - with the cross-rank Welch combine inside
every step - plus C5 with whole captures per rank): one driver run per N carries every configuration's curve.  Also the
shader clock the kernels ran at (`roofline.shader_clock_mhz`: a millisecond of saturated v_add_f32 right after the
kernel-alone pass - the frame kernel is VALU-issue bound, so kernel time x clock is what compares across boxes).

Multi-GPU: frames are independent, every rank processes its own seconds on its own
GPU with its own plan; there is NO collective in the data path and no RCCL anywhere: ranks meet at a
host-side (gloo/TCP) barrier around the timed region and the per-GPU hold traces are combined on the
host with np.fmax.  Aggregate = all ranks' frames / slowest rank's time (weak scaling).
--config c4 shards the 65 536 frames of the ONE named waterfall over the ranks (strong scaling, no combine: every rank
writes its own rows).
--config c5 (2^20-point Welch average of K = 64 segments):
  --c5-shard segments (default; SURVEY.md 8(e) read literally, strong scaling): the SEGMENTS of one capture are sharded
      over the ranks; every step every rank averages its share, exports its partial mean (float32) into a pinned
      shared-memory slab, rank 0's plan combines the partials ON ITS DEVICE (tdsa_welch_combine) and writes the dB row -
      all of it INSIDE the timed region: `value` is end to end, `value_compute_only` leaves the combine out,
      `welch.combine_ms` is what a step pays for it;
  --c5-shard captures (weak scaling): every rank averages whole captures of its own; nothing to combine.

Timing: after W warm-up steps the step loop of K x `inner` steps is timed `--reps` times, each time
fenced (device synchronize + host barrier) on both sides; `inner` is chosen so that one timed region
lasts >= --min-region-s (0.5 s) whatever K is; the MEDIAN repetition (max over ranks) gives ms_per_step and value.

Prints ONE JSON line on rank 0.
"""
import argparse
import copy
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

WORKLOADS = {
    # name: (nfft, hop, samples per step, branch)
    "c3": dict(nfft=16384, hop=8192, n_samples=20_000_000, fs=20e6, branch="hackrf",
               desc="HackRF-shaped: 20 Msps int8 IQ, 16384-pt FFT, 50% overlap + peak-hold trace"),
    "c2": dict(nfft=4096, hop=4096, n_samples=4096 * 4096, fs=2e6, branch="rtl",
               desc="RTL-SDR-shaped: 2 Msps int8 IQ, 4096-pt Hann-windowed FFT"),
    "c4": dict(nfft=8192, hop=8192, n_samples=65536 * 8192, fs=20e6, branch="hackrf",
               desc="Batched waterfall: 64k frames x 8192-pt FFT, frames sharded over the GPUs (no collective)"),
    "c5": dict(nfft=1 << 20, hop=1 << 20, n_samples=64 << 20, fs=2e6, branch="welch",
               desc="Wideband stitch: 1M-pt FFT, Welch average of 64 segments + calibration offset"),
}
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak (MI355X_MICROARCH.md)
CAL_DB = -0.8087054556396822
WELCH_FLOOR = 1e-10


def cpu_all_cores(wl: dict, workers: int, seconds: float) -> dict:
    """SURVEY.md 8(d) CPU baseline (ii): the same numpy restatement in `workers` independent processes
    (one frame stream each, started together), frames/s = sum of frames / slowest worker's wall time."""
    cmd = [sys.executable, "-m", "oracle.cpu_worker", wl["branch"] if wl["branch"] != "welch" else "rtl",
           str(wl["nfft"]), str(wl["hop"]), str(wl["fs"]), str(seconds)]
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    procs = [subprocess.Popen(cmd + [str(100 + i)], cwd=ROOT, env=env, stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                              text=True) for i in range(workers)]
    try:
        for p in procs:
            if p.stdout.readline().strip() != "ready":
                raise RuntimeError("cpu worker failed to start")
        for p in procs:
            p.stdin.write("go\n")
            p.stdin.flush()
        frames, slowest = 0, 0.0
        for p in procs:
            n, dt = p.stdout.readline().split()
            frames += int(n)
            slowest = max(slowest, float(dt))
            p.wait(timeout=60)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    avail = os.cpu_count() or workers
    return {"value": frames / slowest, "unit": "frames/s", "cores": workers, "kind": "port",
            "policy": f"{workers} single-thread worker processes (one per core used) of {avail} host cores; "
                      f"--cpu-workers N picks another count",
            "sample": f"{workers} processes x {seconds:.0f} s of the numpy restatement, one synthetic frame stream each",
            "host_cores_available": avail}


def physical_cores() -> int:
    """Distinct (package, core) pairs of /proc/cpuinfo - the host's physical cores; logical CPU count where that fails."""
    try:
        seen, pkg = set(), None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("physical id"):
                pkg = ln.split(":")[1].strip()
            elif ln.startswith("core id"):
                seen.add((pkg, ln.split(":")[1].strip()))
        if seen:
            return len(seen)
    except OSError:
        pass
    return os.cpu_count() or 1


def synth_iq_int8_device(n_samples: int, nfft: int, seed: int, dev, torch):
    """SURVEY.md 8(d)'s signal model evaluated ON THE DEVICE (float64): three complex tones at bins {N/8, -N/5 + 0.3,
    3N/7 + 0.5} with 40 / 12 / 3 LSB, a (2 + 1j) LSB DC offset, complex Gaussian noise sigma = 4 LSB, rounded and
    clipped to interleaved int8.  Same model as utils/synthetic.synth_iq_int8, whose numpy version needs 25 s for the
    2^26 samples of C5 and minutes for the 2^29 of C4; the noise comes from torch.Generator(seed) instead of
    np.random.default_rng(seed).  Used for the short legs of the default run and for C4; --synth picks."""
    gen = torch.Generator(device=dev)
    gen.manual_seed(1000 + seed)
    out = torch.empty(2 * n_samples, dtype=torch.int8, device=dev)
    chunk = 1 << 24
    bins = (nfft / 8 + 0.0, -nfft / 5 + 0.3, 3 * nfft / 7 + 0.5)
    amps = (40.0, 12.0, 3.0)
    for s0 in range(0, n_samples, chunk):
        m = min(chunk, n_samples - s0)
        n = torch.arange(s0, s0 + m, dtype=torch.float64, device=dev)
        re = torch.full((m,), 2.0, dtype=torch.float64, device=dev)
        im = torch.full((m,), 1.0, dtype=torch.float64, device=dev)
        for b, a in zip(bins, amps):
            ph = torch.remainder(n * (b / nfft), 1.0) * (2.0 * np.pi)       # the phase reduced before the sine
            re += a * torch.cos(ph)
            im += a * torch.sin(ph)
        nz = torch.randn((2, m), dtype=torch.float64, device=dev, generator=gen) * (4.0 / np.sqrt(2.0))
        re += nz[0]
        im += nz[1]
        pair = torch.stack((re, im), dim=1).round_().clamp_(-128, 127).to(torch.int8)
        out[2 * s0: 2 * (s0 + m)] = pair.reshape(-1)
        del n, re, im, nz, pair
    return out


class _DryEngine:
    """--dry-run only: stands in for the GPU plan so that the launch / barrier / aggregation plumbing of
    the multi-GPU leg can be exercised in a container without a GPU.  Does no arithmetic; the line it
    produces is marked as a dry run and carries no performance meaning."""

    def __init__(self, nfft, rank=0):
        self.nfft, self.rank, self._n = nfft, rank, 0

    def step(self, captures=1):
        time.sleep(2e-5 * captures)
        self._n += captures

    def synchronize(self):
        pass

    def hold_stub(self):
        """a recognisable per-rank hold trace: rank r 'held' r in every bin except bin r, where it held 100 + r"""
        h = np.full(64, float(self.rank), dtype=np.float32)
        h[self.rank % 64] = 100.0 + self.rank
        return h

    # the Welch partial of the stand-in: rank r 'measured' the constant r + 1 in every bin (1024 bins)
    DRY_BINS = 1024

    def welch_export(self, dst):
        dst[:] = float(self.rank + 1)

    @staticmethod
    def welch_combine(parts, counts):
        """what tdsa_welch_combine does on the device, for the stand-in: count-weighted mean in float64, rank order"""
        acc = np.zeros(parts.shape[1], dtype=np.float64)
        for m, c in zip(parts, counts):
            if c:
                acc += np.asarray(m, dtype=np.float64) * c
        return acc / float(sum(counts))


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _spawn_workers(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: one worker process per GPU, rendezvous on 127.0.0.1.
    Rank 0's stdout (the JSON line) is passed through.  Every worker is waited for; when one fails the others
    (who would sit in the gloo barrier for ever) are terminated and the first non-zero exit code is returned."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), TDSA_BENCH_WORKER="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    first_bad = 0
    pending = list(procs)
    while pending:
        for p in list(pending):
            rc = p.poll()
            if rc is None:
                continue
            pending.remove(p)
            if rc != 0 and first_bad == 0:
                first_bad = rc
                for q in pending:                      # the rest can only hang at the next barrier
                    q.terminate()
        if pending:
            time.sleep(0.05)
            if first_bad:
                deadline = time.time() + 5.0
                for q in pending:
                    try:
                        q.wait(timeout=max(0.1, deadline - time.time()))
                    except subprocess.TimeoutExpired:
                        q.kill()
                        q.wait()
                pending = []
    return first_bad


class Comm:
    """host-side rendezvous of the ranks: barrier + gather of small Python objects over gloo; nothing else"""

    def __init__(self, world, rank, local_rank, dist):
        self.world, self.rank, self.local_rank, self.dist = world, rank, local_rank, dist

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()

    def gather(self, obj):
        if self.world == 1:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def bcast(self, obj):
        if self.world == 1:
            return obj
        box = [obj]
        self.dist.broadcast_object_list(box, src=0)
        return box[0]


def run_config(args, comm: Comm, torch) -> dict:
    """One configuration, all ranks: returns the result dict on rank 0 (None elsewhere)."""
    world, rank, local_rank = comm.world, comm.rank, comm.local_rank
    host_barrier, gather = comm.barrier, comm.gather
    MIN_REGION_S = max(0.05, args.min_region_s)

    wl = WORKLOADS[args.config]
    nfft, hop, ns = wl["nfft"], wl["hop"], wl["n_samples"]
    frames = (ns - nfft) // hop + 1                   # frames (C5: segments) of one capture
    welch = wl["branch"] == "welch"
    c4 = args.config == "c4"
    shard_captures = welch and args.c5_shard == "captures"
    strong = (welch and not shard_captures) or c4    # ONE capture's units sharded over the ranks
    # C4 as named is one call of 65 536 frames: nothing left to amortise over queued steps
    batch = 1 if (welch or c4) else max(1, args.batch)
    ring = args.ring if args.ring > 0 else (2 if c4 else max(8, 2 * batch))
    ring = max(batch, (ring // batch) * batch)        # whole batches
    # C5 (--c5-shard segments) and C4: the units of ONE capture are sharded over the ranks; everything
    # else: every rank its own captures
    if strong:
        from topdogspectrumanalyser_amd.sharding import shard_frames
        seg0, seg1 = shard_frames(frames, rank, world)
        my_frames = seg1 - seg0
    else:
        seg0, my_frames = 0, frames
    my_ns = ((my_frames - 1) * hop + nfft if my_frames > 0 else 0) if strong else ns   # samples of this rank's capture
    counts = gather(my_frames)
    combine = welch and strong and world > 1          # a cross-rank Welch combine belongs to every step
    slab = None
    synth = args.synth if args.synth != "auto" else ("device" if (c4 or args.short_leg) else "numpy")

    if args.dry_run:
        eng = _DryEngine(nfft, rank)
        visible, dev_index = 0, -1
        step = lambda i: eng.step()                   # noqa: E731
        step_batch = lambda j: eng.step(batch)        # noqa: E731
        dev_sync = lambda: None                       # noqa: E731
        streams = 1
        base = None
    else:
        if not torch.cuda.is_available():
            sys.exit("bench.py needs a GPU (torch.cuda.is_available() is False); --dry-run exercises the plumbing")
        visible = torch.cuda.device_count()
        dev_index = local_rank % visible              # fewer GPUs than ranks: ranks share (stated in config)
        torch.cuda.set_device(dev_index)
        dev = torch.device("cuda", dev_index)

        from topdogspectrumanalyser_amd import SpectrumEngine, _native as nat
        from topdogspectrumanalyser_amd.utils.synthetic import synth_iq_int8

        # ---- synthetic input, resident in HBM before the timed region ---------------------------
        # (strong sharding: every rank synthesises the same capture and keeps its own units)
        seed = 3 + (0 if strong else rank)
        out_rows = 1 if welch else my_frames
        in_ring = torch.empty((ring, max(2, 2 * my_ns)), dtype=torch.int8, device=dev)
        out_ring = torch.empty((ring, max(1, out_rows), nfft), dtype=torch.float32, device=dev)
        if synth == "device":
            if c4:
                # 2^29 samples: one eighth is synthesised (8192 frames), the capture is eight copies of it, each rotated
                # by another odd number of samples - distinct frames, same statistics
                part = synth_iq_int8_device(ns // 8, nfft, seed, dev, torch)
                whole = torch.cat([part if k == 0 else torch.roll(part, 2 * 977 * k) for k in range(8)])
                del part
            else:
                whole = synth_iq_int8_device(ns, nfft, seed, dev, torch)
            mine_dev = whole[2 * seg0 * hop: 2 * seg0 * hop + 2 * my_ns] if strong else whole
            for r in range(ring):
                if my_ns:
                    in_ring[r].copy_(mine_dev if r == 0 else torch.roll(mine_dev, 2 * 977 * r))
            # the parity legs read frames of slot 0 back from the device
            base = None
            base_dev = whole
        else:
            base = synth_iq_int8(ns, nfft, seed=seed)
            mine = base[2 * seg0 * hop: 2 * seg0 * hop + 2 * my_ns] if strong else base
            for r in range(ring):
                host = mine if r == 0 else np.roll(mine, 2 * 977 * r)      # distinct captures, same statistics
                if my_ns:
                    in_ring[r].copy_(torch.from_numpy(np.ascontiguousarray(host)))
            base_dev = None
        torch.cuda.synchronize()
        in_stride_b, out_stride_f = in_ring.stride(0), out_ring.stride(0)

        eng = SpectrumEngine(nfft, max_frames=max(1, my_frames), device=dev_index)
        if wl["branch"] == "hackrf":
            w = np.hanning(nfft).astype(np.float32)
            w /= np.sqrt(np.mean(w ** 2))
            eng.set_window(w)
            eng.configure(db_mode="mag", log_floor=1e-12, dc_alpha=1.0, hold_max=True)
        elif welch:
            eng.set_window(np.hanning(nfft).astype(np.float32))
            eng.configure(db_mode="pow", power_scale=1.0, log_floor=WELCH_FLOOR, dc_alpha=-1.0, avg=("lin", frames),
                          cal_offset_db=CAL_DB)
        else:
            eng.set_window(np.hanning(nfft).astype(np.float32))
            eng.configure(db_mode="pow", power_scale=1.0, log_floor=1e-10, dc_alpha=-1.0, hold_max=True)
        streams = max(1, min(4, args.streams)) if not (welch or c4) else 1

        def step(i: int) -> None:
            r = i % ring
            if welch:
                eng.reset(nat.RESET_AVG)               # every step is one complete Welch average (of this rank's share)
                if my_frames == 0:
                    return
            if my_frames:
                eng.process_device(nat.IN_I8, in_ring[r].data_ptr(), my_ns, hop, my_frames, out_ring[r].data_ptr())

        def step_batch(j: int) -> None:                # `batch` queued steps in one call
            r = (j * batch) % ring
            eng.process_device_batch(nat.IN_I8, in_ring[r].data_ptr(), in_stride_b, batch, my_ns, hop, my_frames,
                                     out_ring[r].data_ptr(), out_stride_f)

        def dev_sync() -> None:
            eng.synchronize()
            torch.cuda.synchronize()

    def frame_iq(k: int) -> np.ndarray:
        """interleaved int8 samples of frame (segment) k of the capture in ring slot 0"""
        if base is not None:
            return base[2 * k * hop: 2 * (k * hop + nfft)]
        return base_dev[2 * k * hop: 2 * (k * hop + nfft)].cpu().numpy()

    # ---- cross-rank Welch combine (C5, --c5-shard segments, more than one rank): part of every step ----------------
    combine_state = {"n": 0, "t_wait": 0.0, "t_comb": 0.0}
    peer = False                                      # the partial means stay in device memory, read in place by rank 0
    if combine:
        from topdogspectrumanalyser_amd.sharding import WelchPeerSlab, WelchSlab
        bins = _DryEngine.DRY_BINS if args.dry_run else nfft
        part_dtype = np.float64 if args.c5_partials == "f64" else np.float32
        if args.c5_combine in ("auto", "peer") and not args.dry_run:
            # device-resident partials (HIP IPC + peer reads over xGMI); any rank that cannot -> everybody falls back
            try:
                pslab = WelchPeerSlab(None, world, 0, bins, dev_index, dtype=part_dtype) if rank == 0 else None
                name = comm.bcast(pslab.name if rank == 0 else None)
                if rank != 0:
                    pslab = WelchPeerSlab(name, world, rank, bins, dev_index, dtype=part_dtype)
                pslab.allocate()
                host_barrier()
                pslab.connect()
                host_barrier()
                if pslab.ok:
                    slab, peer = pslab, True
                else:
                    pslab.close()
            except Exception as exc:                   # (same on every rank: the verdict is rank 0's)
                if args.c5_combine == "peer":
                    raise
                sys.stderr.write(f"[bench] rank {rank}: no peer buffers ({exc}); Welch partials go through host memory\n")
            if not peer and args.c5_combine == "peer":
                sys.exit("--c5-combine peer: the ranks' buffers could not be mapped on rank 0's device")
        if not peer:
            if rank == 0:
                slab = WelchSlab(None, world, 0, bins, dtype=part_dtype, pin=not args.dry_run)
            name = comm.bcast(slab.name if rank == 0 else None)
            if rank != 0:
                slab = WelchSlab(name, world, rank, bins, dtype=part_dtype, pin=not args.dry_run)
        host_barrier()
        dry_combined = {}

        def step_combined(i: int) -> None:
            """one END-TO-END step: this rank's share of the capture, its partial mean into the slab, and - rank 0 - the
            combine of all ranks' partials on its device into the dB row"""
            c = combine_state["n"] + 1
            combine_state["n"] = c
            step(i)
            if counts[rank]:                               # (synchronous: the payload is complete when this returns)
                if peer:
                    eng.welch_export_dev(slab.part_ptr(c), as_f32=part_dtype == np.float32)
                else:
                    eng.welch_export(slab.part(c))
            slab.publish(c)
            if rank == 0:
                t0 = time.perf_counter()
                parts = slab.wait_all(c)
                t1 = time.perf_counter()
                if args.dry_run:
                    dry_combined["mean"] = _DryEngine.welch_combine(parts, counts)
                elif peer:
                    eng.welch_combine_dev(parts, counts, as_f32=part_dtype == np.float32,
                                          out_db_dev=out_ring[i % ring].data_ptr())
                    eng.synchronize()                      # the kernel has read the slots: they may be written again
                else:
                    eng.welch_combine(parts, counts, out_db_dev=out_ring[i % ring].data_ptr())
                    eng.synchronize()                      # the upload has read the slot: it may be written again
                slab.done(c)
                combine_state["t_wait"] += t1 - t0
                combine_state["t_comb"] += time.perf_counter() - t1

    def fence() -> None:
        dev_sync()
        host_barrier()
        dev_sync()

    rank_times = []                                   # per timed region: every rank's own seconds

    def timed_loop(n_calls: int, call) -> float:
        """one timed region: fence, n_calls calls, device synchronize, fence; seconds of the slowest rank"""
        fence()
        if slab is not None:
            combine_state["n"] = 0
            slab.reset()
            host_barrier()
        t0 = time.perf_counter()
        for i in range(n_calls):
            call(i)
        dev_sync()
        own = time.perf_counter() - t0                # this rank's own work, before it waits for the others
        fence()
        every = gather(own)
        rank_times.append(every)
        return max(every)

    def measure(n_streams: int, per_call: int, call=None):
        """K x inner steps per region, `per_call` steps per call -> (median seconds per region, all regions, inner, per rank)"""
        if call is None:
            call = step if per_call == 1 else step_batch
        calls_k = max(1, args.steps // per_call)      # K steps = K / per_call calls (K is rounded to whole calls)
        if not args.dry_run:
            eng.set_overlap(n_streams)
        if slab is not None:
            fence()
            combine_state["n"] = 0
            slab.reset()
            host_barrier()
        for i in range(max(1, args.warmup // per_call)):
            call(i)
        est = timed_loop(calls_k, call)               # calibration region (also warm-up)
        inner = max(1, int(np.ceil(1.2 * MIN_REGION_S / max(est, 1e-9))))
        for _ in range(4):                            # a region that came out short is re-timed longer
            inner = max(gather(inner))                # same loop count on every rank
            del rank_times[:]
            combine_state["t_wait"] = combine_state["t_comb"] = 0.0
            times = [timed_loop(calls_k * inner, call) for _ in range(max(1, args.reps))]
            if min(times) >= MIN_REGION_S:
                break
            inner = int(np.ceil(inner * 1.3 * MIN_REGION_S / max(min(times), 1e-9)))
        per_rank = [statistics.median(t[r] for t in rank_times) for r in range(world)]
        return dict(med=statistics.median(times), times=times, inner=inner, per_rank=per_rank,
                    steps=calls_k * inner * per_call, per_call=per_call, streams=n_streams,
                    comb_wait_s=combine_state["t_wait"], comb_s=combine_state["t_comb"],
                    steps_all_regions=calls_k * inner * per_call * len(times))

    # clocks: an idle MI355X needs a few hundred ms of load before shader/fabric clocks settle; this
    # untimed pre-roll keeps short --steps/--warmup runs from measuring the ramp
    only = args.legs
    pre_call = step_batch if (only == "value" and batch > 1) else step
    t_pre = time.perf_counter()
    i_pre = 0
    while time.perf_counter() - t_pre < args.preroll_seconds:
        for _ in range(50 if pre_call is step else max(1, 50 // batch)):
            pre_call(i_pre)
            i_pre += 1
        dev_sync()

    legs = {}
    compute_only = None
    if combine:
        compute_only = measure(1, 1)                   # every rank's share alone (what round 4 called `value`)
        legs["value"] = measure(1, 1, call=step_combined)
        legs["serial"] = legs["streams"] = legs["value"]
    elif welch or c4 or args.dry_run and batch == 1:
        legs["value"] = measure(1, 1)
        legs["serial"] = legs["streams"] = legs["value"]
    elif only != "all":                                # one submission mode only (rocprofv3 runs)
        one = measure(streams, batch) if only == "value" else (measure(streams, 1) if only == "streams" else measure(1, 1))
        legs["value"] = legs["streams"] = legs["serial"] = one
    else:
        legs["value"] = measure(streams, batch) if batch > 1 else None
        legs["streams"] = measure(streams, 1) if streams > 1 else None
        legs["serial"] = measure(1, 1)
        if legs["streams"] is None:
            legs["streams"] = legs["serial"]
        if legs["value"] is None:
            legs["value"] = legs["streams"]
    head = legs["value"]
    # ---- per-frame scalars from the frame kernel's epilogue (tdsa_set_frame_stats): the same submission
    #      mode as `value` with peak / argmax / band power of every frame left beside the rows, and the scalars of one
    #      launch checked against tdsa_rows_stats of the rows that launch wrote
    frame_stats_block = None
    if (args.config == "c3" and only == "all" and world == 1 and not args.dry_run and not getattr(args, "short_leg", False)
            and not args.no_frame_stats):
        from topdogspectrumanalyser_amd import analytics as _an
        band = (nfft // 4, 3 * nfft // 4)
        eng.set_frame_stats(True, None)                      # peak + argmax only
        leg_pk = measure(head["streams"], head["per_call"])
        eng.set_frame_stats(True, band)
        leg_fs = measure(head["streams"], head["per_call"])
        eng.set_overlap(1)
        step(0)
        pk, pb, bd = eng.frame_stats(bin_width=1.0)
        rp, rb, rband = _an.rows_stats(eng, out_ring[0].data_ptr(), my_frames, freq_bins=np.arange(nfft, dtype=np.float64),
                                       band=(float(band[0]), float(band[1])))
        eng.set_frame_stats(False)
        frame_stats_block = {
            "ms_per_step": leg_fs["med"] / leg_fs["steps"] * 1e3, "ms_per_step_without": head["med"] / head["steps"] * 1e3,
            "overhead": leg_fs["med"] / leg_fs["steps"] / (head["med"] / head["steps"]) - 1.0,
            "ms_per_step_peak_and_bin_only": leg_pk["med"] / leg_pk["steps"] * 1e3,
            "overhead_peak_and_bin_only": leg_pk["med"] / leg_pk["steps"] / (head["med"] / head["steps"]) - 1.0,
            "band_bins": list(band), "steps_per_call": leg_fs["per_call"], "streams": leg_fs["streams"],
            "peak_and_bin_equal_rows_stats": bool(np.array_equal(pk, rp) and np.array_equal(pb, rb)),
            "band_db_max_abs_diff_vs_rows_stats": float(np.max(np.abs(bd - rband))),
            "is": "the `value` leg again with tdsa_set_frame_stats on: every wave of the frame kernel also leaves the maximum of "
                  "its bins, its first position and the band's linear power (16 bytes per wave and frame), folded per frame when "
                  "read; the alternative - tdsa_rows_stats, a second pass over the 160 MB of rows per step - costs more than the "
                  "spectra themselves (profiles/r06_frame_stats.txt)"}
    steps_timed = head["steps"]
    total_frames = frames if strong else world * frames          # frames (segments) one step covers, all ranks together
    per_rank_fps = [my_f * steps_timed / t for my_f, t in zip(counts, head["per_rank"])]

    # ---- dominant kernel alone: HIP events on the plan's stream around every frame-kernel launch,
    #      launches strictly serial so that one kernel owns the GPU while it is timed ---------------
    def kernel_alone(per_call: int):
        """-> (launches, mean seconds per launch) of the frame kernel in the launch shape of `per_call` steps per call"""
        if args.dry_run:
            return 0, 0.0
        eng.set_overlap(1)
        eng.profile_enable(True)
        n_calls = max(1, min(legs["serial"]["steps"], 2000) // per_call)
        for i in range(n_calls):
            (step if per_call == 1 else step_batch)(i)
        launches, kern_ms = eng.profile_read()
        eng.profile_enable(False)
        return launches, (kern_ms * 1e-3 / launches if launches else 0.0)

    launches_b, kern_b = kernel_alone(head["per_call"])
    launches_1, kern_1 = (launches_b, kern_b) if (head["per_call"] == 1 or only != "all") else kernel_alone(1)
    # the shader clock the kernels just ran at: a millisecond of saturated v_add_f32 right behind the kernel-alone pass
    shader_mhz = valu_ns = None
    if not args.dry_run:
        try:
            shader_mhz, valu_ns = eng.shader_clock()
        except Exception:                              # a reported extra only
            shader_mhz = valu_ns = None
    if welch:                                          # a chain of kernels: price the whole serial step (compute only)
        launches_b = launches_1 = 0
        src = compute_only if compute_only is not None else legs["serial"]
        kern_b = kern_1 = src["per_rank"][rank] / src["steps"] if rank < len(src["per_rank"]) else 0.0
        bytes_per_frame = 2 * hop + 4 * nfft // frames      # one dB row per K segments: 2N + 4N/K per segment
    else:
        bytes_per_frame = 2 * hop + 4 * nfft           # every input byte read once, every
    algo_step = my_frames * bytes_per_frame            # output byte written once (this rank's share of a step)
    algo_launch = algo_step * head["per_call"]
    achieved_gbs = algo_launch / kern_b / 1e9 if kern_b else 0.0
    achieved_1 = algo_step / kern_1 / 1e9 if kern_1 else 0.0
    per_gpu_frac = gather(achieved_gbs / HBM_PEAK_GBS)

    # secondary denominator: the device-to-device copy bandwidth this box reaches, read + write bytes
    copy_gbs = None
    if not args.dry_run and rank == 0 and not args.short_leg:
        try:
            nb = 1 << 30                                  # 1 GiB each way: well past the 256 MiB Infinity Cache
            a = torch.empty(nb, dtype=torch.uint8, device=dev)
            b = torch.empty(nb, dtype=torch.uint8, device=dev)
            b.copy_(a)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                b.copy_(a)
            e1.record()
            torch.cuda.synchronize()
            copy_gbs = 10 * 2 * nb / (e0.elapsed_time(e1) * 1e-3) / 1e9
            del a, b
        except Exception:                                 # a reported extra only
            copy_gbs = None

    # per-GPU state combined on the HOST; the Welch combine ran inside the timed region (above)
    hold_combined, welch_block, combined_db = None, None, None
    if welch:
        if combine:
            # one more end-to-end step into ring slot 0: the row the parity block checks
            fence()
            combine_state["n"] = 0
            slab.reset()
            host_barrier()
            step_combined(0)
            dev_sync()
            host_barrier()
            if rank == 0:
                n_steps = max(1, head["steps_all_regions"])
                e2e_ms = head["med"] / steps_timed * 1e3
                comp_ms = compute_only["med"] / compute_only["steps"] * 1e3
                welch_block = {"shard": "segments", "segments_per_rank": counts, "segments_total": int(sum(counts)),
                               "exchange": "peer" if peer else "host",
                               "partials": (f"{slab.dtype.name} means left in device buffers of the ranks' own GPUs "
                                            f"({slab.slots} slots x {slab.n} bins each), mapped on rank 0's device through HIP IPC "
                                            "handles and read in place by its combine kernel (other GPUs': over xGMI); no "
                                            "host copy, no pickling, no collective, no RCCL") if peer else
                                           (f"{slab.dtype.name} means through a {'pinned ' if slab.pinned else ''}shared-memory "
                                            f"slab ({slab.slots} slots x {world} ranks x {slab.n} bins), no pickling, no collective"),
                               "combined_on": f"rank 0's device: tdsa_welch_combine{'_dev' if peer else ''} (count-weighted mean in "
                                              "float64, rank order, then 10*log10(mean + floor) + calibration offset), inside "
                                              "the timed region",
                               "combine_ms": max(0.0, e2e_ms - comp_ms),
                               "combine_ms_is": "ms_per_step (end to end) - ms_per_step_compute_only",
                               "rank0_wait_for_partials_ms": head["comb_wait_s"] / n_steps * 1e3,
                               "rank0_upload_combine_ms": head["comb_s"] / n_steps * 1e3,
                               "ms_per_step_compute_only": comp_ms, "pinned": bool(getattr(slab, "pinned", False))}
                if args.dry_run:
                    welch_block["mean_of_means"] = float(np.mean(dry_combined["mean"]))
                else:
                    combined_db = out_ring[0][0].cpu().numpy()
        else:
            if not args.dry_run:
                step(0)
                eng.synchronize()
            if rank == 0:
                welch_block = {"shard": "captures" if shard_captures else "segments",
                               "segments_per_rank": counts, "segments_total": int(sum(counts)) if strong else frames,
                               "combined_on": "nothing to combine: every rank averages whole captures" if shard_captures
                                              else "one rank: the plan's own gather + finish",
                               "combine_ms": 0.0}
                if not args.dry_run:
                    combined_db = out_ring[0][0].cpu().numpy()
                if args.dry_run:
                    welch_block["mean_of_means"] = 1.0
    elif not args.dry_run:
        from topdogspectrumanalyser_amd.sharding import combine_hold
        mx, _ = eng.hold()
        parts = [p for p in gather(mx) if p is not None]
        if parts:
            hold_combined = combine_hold(parts, "max")

    # HBM bytes per launch from the committed rocprofv3 PMC passes of this same kernel and shape
    # (counters cannot be read from inside the process)
    traffic, traffic_src = None, None
    quoted = {}                                        # committed files the line quotes: sha + mtime, so a stale one shows

    def _quote(path):
        import hashlib
        with open(path, "rb") as fh:
            blob = fh.read()
        quoted[os.path.relpath(path, ROOT)] = {"sha256_16": hashlib.sha256(blob).hexdigest()[:16],
                                               "mtime_utc": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime(os.path.getmtime(path)))}
        return json.loads(blob)

    for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):
        pmc_path = os.path.join(ROOT, "profiles", f"{rnd}_{args.config}_pmc.json")
        if os.path.exists(pmc_path):
            pmc = _quote(pmc_path)
            traffic = pmc["fetch_bytes_upper"] + pmc["write_bytes"]
            traffic_src = (f"profiles/{rnd}_{args.config}_pmc.json (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, per one-step "
                           f"launch of {pmc.get('algorithmic_bytes', 0) / 1e6:.1f} MB algorithmic"
                           + (f"; collected by {pmc['collected_by']}" if "collected_by" in pmc else "") + ")")
            break
    valu_issue = None
    for rnd in ("r06", "r05", "r04", "r03"):
        vi_path = os.path.join(ROOT, "profiles", f"{rnd}_{args.config}_valu_issue.json")
        if os.path.exists(vi_path):
            valu_issue = _quote(vi_path)
            break

    # N > 1: what the node says about GPU-to-GPU reads (hipDeviceCanAccessPeer for every ordered pair of the devices the
    # ranks run on), so that the line explains which Welch exchange it took and why
    peer_matrix = None
    if rank == 0 and world > 1 and not args.dry_run:
        import ctypes as _C
        devs = sorted({r % max(1, visible) for r in range(world)})
        can = {}
        for a_ in devs:
            for b_ in devs:
                if a_ != b_:
                    v = _C.c_int(0)
                    rc = nat.lib.tdsa_peer_can_access(a_, b_, _C.byref(v))
                    can[f"{a_}->{b_}"] = bool(v.value) if rc == 0 else None
        peer_matrix = {"devices_used": devs, "hipDeviceCanAccessPeer": can,
                       "all_pairs": bool(can) and all(bool(x) for x in can.values()),
                       "ranks_share_devices": world > len(devs),
                       "welch_exchange": ("peer (device buffers mapped through HIP IPC, read in place)" if (welch and peer) else
                                          "host (pinned shared memory)" if (welch and combine) else
                                          "none needed by this configuration (rows and hold traces stay per rank; hold traces "
                                          "meet on the host with np.fmax)")}

    result = None
    if rank == 0:
        def rate(leg):
            return total_frames * leg["steps"] / leg["med"]
        value = rate(head)
        if welch:
            par = (f"the {frames} Welch segments of one capture sharded over {world} GPU(s), one process + plan per GPU, no "
                   f"collective: float32 partial means through pinned shared memory, combined on rank 0's device inside "
                   f"every timed step (host gloo barrier around the timed region only)") if not shard_captures else \
                  (f"whole captures of {frames} segments per GPU, {world} GPU(s), one process + plan per GPU, no collective "
                   f"and nothing to combine (host gloo barrier around the timed region)")
        elif c4:
            par = (f"the {frames} frames of the one waterfall sharded over {world} GPU(s) (contiguous ranges), one process + "
                   f"plan per GPU, no collective: every rank writes its own rows (host gloo barrier around the timed region, "
                   f"hold traces combined with np.fmax)")
        else:
            par = (f"frames sharded over {world} GPU(s), one process + plan per GPU, no collective "
                   f"(host gloo barrier around the timed region, hold traces combined with np.fmax)")
        result = {
            "metric": "PSD frames/sec at 16384-pt FFT on synthetic 20 Msps IQ" if args.config == "c3"
                      else f"PSD frames/sec ({args.config})",
            "value": value,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": head["med"] / steps_timed * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": ("synthetic" + (" (SURVEY 8(d) signal model evaluated on the device, noise from torch.Generator)"
                                    if synth == "device" else ""))
                    if not args.dry_run else "dry-run (stand-in engine, no GPU work, no perf meaning)",
            "value_streams": rate(legs["streams"]),
            "ms_per_step_streams": legs["streams"]["med"] / legs["streams"]["steps"] * 1e3,
            "value_serial": rate(legs["serial"]),
            "ms_per_step_serial": legs["serial"]["med"] / legs["serial"]["steps"] * 1e3,
            "per_gpu_frames_per_s": per_rank_fps,
            "timing": {"repetitions": len(head["times"]), "inner_repeats": head["inner"], "steps_per_region": steps_timed,
                       "steps_per_call": head["per_call"],
                       "region_ms": [t * 1e3 for t in head["times"]],
                       "region_ms_streams": [t * 1e3 for t in legs["streams"]["times"]],
                       "region_ms_serial": [t * 1e3 for t in legs["serial"]["times"]],
                       "min_region_s": MIN_REGION_S,
                       "statistic": "median over repetitions of (max over ranks)",
                       "value_is": (f"{head['per_call']} queued step(s) per call (one persistent launch each), calls rotating "
                                    f"over {head['streams']} stream(s) per GPU"
                                    + (", launches sized for half the CUs (two side by side)" if head["streams"] >= 3 else "")
                                    + ("; END TO END: every step includes the export of each rank's partial mean and the "
                                       "combine on rank 0's device" if combine else "")),
                       "value_streams_is": f"one step per launch, {legs['streams']['streams']} stream(s) per GPU (round 2's value)",
                       "value_serial_is": "one step per launch, strictly serial full-chip launches (1 stream)"},
            "config": {"workload": f"{args.config}: {wl['desc']}", "nfft": nfft, "hop": hop,
                       "frames_per_step_per_gpu": my_frames, "frames_per_step_all_gpus": total_frames,
                       "frames_per_step_per_rank": counts,
                       "input": "int8 IQ resident in HBM", "input_ring": ring, "streams_per_gpu": head["streams"],
                       "steps_per_call": head["per_call"], "inner_repeats": head["inner"],
                       "gpus_visible_per_process": visible,
                       **({"peer_access": peer_matrix} if peer_matrix is not None else {}),
                       "launcher": "torch.distributed.run" if not os.environ.get("TDSA_BENCH_WORKER") and world > 1
                                   else ("self-spawned workers" if world > 1 else "single process"),
                       "parallelism": par},
            "roofline": {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved_gbs / HBM_PEAK_GBS, "frac_per_gpu": per_gpu_frac, "traffic": traffic,
                         "traffic_unit": "bytes per one-step launch", "traffic_source": traffic_src,
                         "quoted_files": quoted,
                         "algorithmic_bytes_per_launch": algo_launch,
                         "frames_per_launch": my_frames * head["per_call"],
                         "kernel": "spectrum_kernel" if launches_b else "column pass + row pass + gather + finish (whole serial step, "
                                                                        "this rank's share, no cross-rank combine)",
                         "kernel_avg_us": kern_b * 1e6, "launches_timed": launches_b,
                         "kernel_avg_from": "HIP events on the plan's stream around every launch of a further, strictly "
                                            "serial pass (one launch on the whole chip at a time), in the launch shape of "
                                            "`value` (steps_per_call queued steps per launch)",
                         "algorithmic_bytes_per_frame": bytes_per_frame,
                         "shader_clock_mhz": shader_mhz, "valu_ns_per_wave_instr_per_simd": valu_ns,
                         "shader_clock_is": "2 clocks / (ns per wave-instruction per SIMD) of a millisecond of independent "
                                            "v_add_f32 chains, four waves per SIMD on every CU, timed with HIP events right after "
                                            "the kernel-alone pass (tdsa_shader_clock)",
                         "kernel_mcycles_per_step": (kern_b / head["per_call"] * 1e6 * shader_mhz * 1e-6
                                                     if (shader_mhz and kern_b) else None),
                         "measured_copy_gbs": copy_gbs,
                         "frac_of_measured_copy": (achieved_gbs / copy_gbs) if copy_gbs else None,
                         "measured_copy_is": "torch device-to-device copy of 1 GiB on this GPU, read + write bytes per second "
                                             "(BASELINE.md 4: secondary denominator next to the 8 TB/s peak)",
                         "single_step_launch": {"achieved": achieved_1, "frac": achieved_1 / HBM_PEAK_GBS,
                                                "kernel_avg_us": kern_1 * 1e6, "launches_timed": launches_1,
                                                "algorithmic_bytes_per_launch": algo_step,
                                                "frames_per_launch": my_frames}},
        }
        if frame_stats_block is not None:
            result["frame_stats"] = frame_stats_block
        if welch and kern_b and not args.dry_run:
            # The spec roof (8 TB/s, algorithmic bytes) next to the design's own ceiling: a two-pass transform moves 18 N bytes
            # per segment (2 N samples in, Z = 8 N out and in again, + 4 N / K of sums) through the links between the XCDs and
            # the fabric, which carry reads and writes TOGETHER at the rate measured for them on this chip
            # (profiles/r04_c5_experiments.txt: the column pass's stores alone saturate at 6.7 TB/s, 537 MB in 79.6 us, with Z in
            # HBM or in the Infinity Cache alike; tools/ubench/stream_mix.hip)
            link_gbs = 6700.0
            impl_seg = 18 * nfft + 4 * nfft // frames
            impl_launch = impl_seg * my_frames * head["per_call"]
            result["roofline"]["fabric_roof"] = {
                "implementation_bytes_per_segment": impl_seg, "implementation_bytes_per_launch": impl_launch,
                "link_rate_gbs": link_gbs, "link_rate_from": "profiles/r04_c5_experiments.txt (column pass with stores only: 537 MB "
                                                              "of Z in 79.6 us), tools/ubench/stream_mix.hip",
                "roof_us_per_launch": impl_launch / (link_gbs * 1e9) * 1e6,
                "achieved_implementation_gbs": impl_launch / kern_b / 1e9,
                "frac_of_fabric_roof": impl_launch / kern_b / 1e9 / link_gbs,
                "is": "implementation bytes of the two-pass transform / the XCD <-> fabric link rate: what this design could reach "
                      "at best; `frac` above prices the same time against the algorithmic bytes and the 8 TB/s spec peak"}
        if not welch and launches_b and not args.dry_run:
            # a cheap in-run check of the traffic figure quoted from the committed counter passes: what the plan itself knows
            # it read and wrote per launch (every sample byte of the capture once - overlapping frames share theirs -, every
            # dB row once) against the counters' bytes for the same shape
            known = (((my_frames - 1) * hop + nfft) * 2 + my_frames * nfft * 4)
            chk = {"plan_known_bytes_per_one_step_launch": known,
                   "is": "samples of the capture x 2 bytes (each read once: neighbouring frames share theirs through L2) + rows x 4 N"}
            if traffic:
                chk["counters_over_known"] = traffic / known
                chk["consistent"] = bool(0.9 <= traffic / known <= 1.35)
                chk["note"] = ("FETCH_SIZE is an upper figure (x2 of the 32-byte unit, gfx950 correction of the guide): the shared "
                               "halves of overlapping frames that miss L2 and the window / twiddle tables are in it")
            result["roofline"]["traffic_check"] = chk
        if welch and compute_only is None:             # one rank, or whole captures per rank: nothing to combine
            result["value_compute_only"] = value
            result["ms_per_step_compute_only"] = result["ms_per_step"]
        if compute_only is not None:
            result["value_compute_only"] = total_frames * compute_only["steps"] / compute_only["med"]
            result["ms_per_step_compute_only"] = compute_only["med"] / compute_only["steps"] * 1e3
            result["value_is"] = ("end to end: every rank's share of the capture + export of its partial mean + the combine on "
                                  "rank 0's device, all inside the timed region; value_compute_only leaves export and combine out")
        if only != "all":
            result["config"]["legs"] = f"--legs {only}: only that submission mode was timed; value_streams / value_serial repeat it"
            if head["per_call"] > 1:
                result["roofline"].pop("single_step_launch", None)
        if valu_issue is not None and not args.dry_run and kern_b:
            # secondary roofline: the kernel's VALU issue time at the SIMDs' saturated per-class rates (committed
            # counters + microbenchmarks, tools/valu_issue.py) against this run's own kernel time
            live_us = kern_b / head["per_call"] * 1e6
            vi = {k: valu_issue[k] for k in ("bound", "kernel", "insts_valu_per_wave_frame", "issue_ns_per_wave_frame",
                                             "waves_per_simd", "issue_us_per_frame_slot", "measured_valu_only_us_per_frame_slot",
                                             "floor_us_per_step", "unit", "hbm_frac_if_only_valu_issue_remained", "sources")}
            vi.update({"achieved": live_us, "peak": valu_issue["floor_us_per_step"], "frac": valu_issue["floor_us_per_step"] / live_us,
                       "reading": "frac = VALU issue time of one step (instruction counts x saturated issue cost per class, perfectly "
                                  "balanced over 256 CUs) / this run's kernel time per step: the share of the launch the SIMDs need "
                                  "just to issue the kernel's VALU instructions"})
            if shader_mhz:
                # the same floor from this run's own clock: instructions x 2 clocks at the measured rate (no per-class costs)
                per_wave = valu_issue["insts_valu_per_wave_frame"] * valu_ns * 1e-3             # us per wave and frame at the add rate
                vi["floor_us_per_step_at_this_clock_all_at_add_rate"] = per_wave * valu_issue["waves_per_simd"] * \
                    (my_frames / 256.0)
            result["roofline"]["valu_issue"] = vi
        if args.dry_run:                               # nothing was computed: no performance figures
            for k in ("achieved", "frac", "frac_per_gpu", "traffic", "kernel_avg_us"):
                result["roofline"][k] = None
            result["roofline"]["kernel"] = "none (dry run)"
            result["roofline"].pop("single_step_launch", None)
        if hold_combined is not None:
            result["hold_trace"] = {"combined_on": "host (np.fmax over ranks)", "max_db": float(np.max(hold_combined)),
                                    "argmax_bin": int(np.argmax(hold_combined))}
        if welch_block is not None:
            result["welch"] = welch_block

    # ---- parity (rows of rank 0's launch + the COMBINED hold trace of all ranks; C5: the combined Welch row) and the
    #      CPU baseline (rank 0; a short leg when several ranks run) - at every world size
    multi = world > 1
    want_cpu = rank == 0 and not args.no_cpu_baseline and not args.dry_run
    want_parity = not args.no_parity and not args.dry_run          # every rank takes part in the hold-trace check
    cpu_budget = min(args.cpu_seconds, args.cpu_seconds_multi) if multi else args.cpu_seconds
    if args.dry_run and not welch:
        # plumbing only: the per-rank stand-in traces travel the same gather + np.fmax combine
        from topdogspectrumanalyser_amd.sharding import combine_hold
        stubs = gather(eng.hold_stub())
        if rank == 0:
            comb = combine_hold(stubs, "max")
            result["hold_trace"] = {"combined_on": "host (np.fmax over ranks)", "ranks_combined": len(stubs),
                                    "max_db": float(np.max(comb)), "argmax_bin": int(np.argmax(comb)),
                                    "checked_positions": 0, "max_db_err_vs_gold": None, "pass": None, "dry_run": True}
            result["parity"] = {"dry_run": True, "pass": None, "hold_trace_pass": None, "checked": "nothing (dry run)"}
            result["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": 1, "kind": "port", "dry_run": True,
                                      "sample": f"none (dry run; a real run times {cpu_budget:.0f} s on rank 0)"}
    if want_parity or want_cpu:
        from oracle import spectrum_oracle as so   # checker / reported baseline only

        def parity_block(pairs, checked):
            worst_rel = worst_db = raw60 = raw100 = 0.0
            for got, g in pairs:
                rel, ddb = so.parity_metrics(got, g, floor_rel_db=100.0, amp_floor=2 * so.AMP_FLOOR)
                worst_rel, worst_db = max(worst_rel, rel), max(worst_db, ddb)
                raw60, raw100 = max(raw60, so.parity_raw_db(got, g, 60.0)), max(raw100, so.parity_raw_db(got, g, 100.0))
            return {"max_rel_power_err": worst_rel, "max_db_err_top60dB": raw60, "max_db_err_top100dB": raw100,
                    "db_err_over_allowance_x1e-3": worst_db, "checked": checked, "against": "float64 gold oracle",
                    "bounds": "north_star: rel <= 1e-4 of the frame maximum.  SURVEY 8(d): |dB| <= 1e-3 within 100 dB of it - "
                              "`survey_8d_strict_pass` applies that literally, with no allowance; `pass` allows two float32 "
                              "rounding units (2^-23) of the frame's largest amplitude where that is worth more than 1e-3 dB "
                              "of the bin (bins deeper than 60 dB): db_err_over_allowance_x1e-3 <= 1e-3",
                    "north_star_pass": bool(worst_rel <= 1e-4),
                    "survey_8d_strict_pass": bool(worst_rel <= 1e-4 and raw100 <= 1e-3),
                    "pass": bool(worst_rel <= 1e-4 and worst_db <= 1e-3)}

        done, cpu_s, sample = 0, 0.0, ""
        if welch:
            seg = lambda k: so.unpack_iq_int8(frame_iq(k))   # noqa: E731
            if want_cpu:
                br = so.RtlBranchOracle(nfft, wl["fs"], precision="ref")
                br.averager.set_mode("lin", frames)
                t_cpu0 = time.perf_counter()
                while done < 3 or (time.perf_counter() - t_cpu0 < cpu_budget and done < frames):
                    br.power_levels(seg(done % frames))
                    done += 1
                cpu_s = time.perf_counter() - t_cpu0
                sample = f"{done} segments of 2^20 points, single thread, numpy {np.__version__} restatement incl. int8 unpack"
            if rank == 0 and want_parity:
                # parity: the whole Welch average (all K segments, combined over the ranks) against the float64 gold; the
                # segments' spectra are independent: a few host threads form them, the running mean takes them in order
                g = so.welch_gold(seg, frames, nfft, wl["fs"], threads=min(16, os.cpu_count() or 1))
                g = np.asarray(g, dtype=np.float64) + CAL_DB
                pairs = [(np.asarray(combined_db, dtype=np.float32), g)]
                checked = (f"Welch mean of all {frames} segments, " +
                           (f"combined on rank 0's device from {world} ranks' partial means" if combine
                            else "the device's own dB row" + (" (rank 0's capture)" if shard_captures else "")))
                result["parity"] = parity_block(pairs, checked)
        else:
            branch = "hackrf" if wl["branch"] == "hackrf" else "rtl"
            if want_cpu:
                br = (so.HackrfBranchOracle if branch == "hackrf" else so.RtlBranchOracle)(nfft, wl["fs"], precision="ref")
                # (a capture synthesised on the device: the CPU leg cycles through 256 of its frames, read back once)
                pool = [frame_iq(k) for k in range(min(my_frames, 256))] if base is None else None
                t_cpu0 = time.perf_counter()
                while time.perf_counter() - t_cpu0 < cpu_budget:            # the same second of IQ, over and over
                    if pool is not None:
                        br.power_levels(so.unpack_iq_int8(pool[done % len(pool)]))
                    else:
                        k = done % frames
                        br.power_levels(so.unpack_iq_int8(base[2 * k * hop: 2 * (k * hop + nfft)]))
                    done += 1
                cpu_s = time.perf_counter() - t_cpu0
                sample = (f"{done} frames ({cpu_budget:.0f} s) cycling through the same second of IQ, single "
                          f"thread, numpy {np.__version__} restatement of get_power_levels incl. int8 unpack")
            if want_parity:
                # (1) the hold trace, every rank: ONE step over this rank's own capture (ring slot 0) from a
                #     fresh state; the trace must equal the column maximum of the rows the same launch wrote (bit for
                #     bit) and - combined over the ranks with np.fmax - the float64 gold of
                #     core/display_data_processor.py:371-382 at a sample of positions (tone bins, their neighbours, DC,
                #     the edges, 16 others), which every rank evaluates for its own frames from the DFT definition
                from topdogspectrumanalyser_amd.sharding import combine_hold
                eng.reset()
                eng.set_overlap(1)
                step(0)
                eng.synchronize()
                mine_hold, _ = eng.hold()
                colmax = out_ring[0][:max(1, my_frames)].amax(dim=0).cpu().numpy()
                eq_own = bool(mine_hold is None or np.array_equal(mine_hold, colmax))     # at the capture's full size
                tone_k = [nfft // 8, int(round(-nfft / 5 + 0.3)), int(round(3 * nfft / 7 + 0.5)), 0]
                pos = sorted({(k + d + nfft // 2) % nfft for k in tone_k for d in (-1, 0, 1)} | {0, nfft - 1}
                             | {int(v) for v in np.random.default_rng(1234).integers(0, nfft, 16)})
                # (the gold hold costs one DFT-definition product per frame and position: on captures of more than 4096
                #  frames it is evaluated - like the device trace it is compared with - on the first 4096 frames only)
                hold_frames = min(my_frames, 4096)
                if hold_frames < my_frames:
                    eng.reset()
                    eng.process_device(nat.IN_I8, in_ring[0].data_ptr(), (hold_frames - 1) * hop + nfft, hop, hold_frames,
                                       out_ring[0].data_ptr())
                    eng.synchronize()
                    mine_hold, _ = eng.hold()
                my_iq = (in_ring[0][: 2 * ((hold_frames - 1) * hop + nfft)].cpu().numpy() if hold_frames else
                         np.zeros(0, dtype=np.int8))
                # the trace's allowance is what follows from the rows' (oracle.HoldAllowance: |max a - max b| <= max |a - b|
                # bin by bin): per position the largest allowance any held frame had there - 1e-3 dB on everything within
                # ~60 dB of its frame's maximum, two float32 rounding units of that maximum's amplitude below
                gold_pos, allow_pos = (so.max_hold_at_positions(my_iq, nfft, hop, pos, branch=branch, allowance_units=2.0)
                                       if hold_frames else (None, None))
                per_rank = gather((mine_hold[pos] if mine_hold is not None else None, gold_pos, eq_own, allow_pos))
                if rank == 0:
                    have = [(h, g, e, a) for h, g, e, a in per_rank if h is not None]
                    comb = combine_hold([h for h, _, _, _ in have], "max")
                    gold_comb = np.fmax.reduce(np.stack([g for _, g, _, _ in have]), axis=0)
                    allow_comb = np.max(np.stack([a for _, _, _, a in have]), axis=0)
                    diff = np.abs(comb.astype(np.float64) - gold_comb)
                    err = float(np.max(diff))
                    err_over_allowance = float(np.max(diff / allow_comb))
                    eq = [e for _, _, e, _ in per_rank]
                    hold_pass = bool(err_over_allowance <= 1.0 and all(eq))
                    ht = result.setdefault("hold_trace", {"combined_on": "host (np.fmax over ranks)"})
                    ht.update({"ranks_combined": len(have), "checked_positions": len(pos), "max_db_err_vs_gold": err,
                               "max_db_err_over_allowance": err_over_allowance,
                               "allowance": "per position the largest allowance any held frame had there (1e-3 dB, or two float32 "
                                            "rounding units of that frame's largest amplitude where the bin lies deeper than ~60 dB): "
                                            "|max a - max b| <= max |a - b|, so the trace inherits the rows' bound",
                               "frames_per_rank_checked": hold_frames,
                               "gold": "np.fmax over the ranks of each rank's float64 gold hold over its own frames "
                                       "(oracle.max_hold_at_positions: the branch's arithmetic from the DFT definition at the "
                                       "sampled fftshift-ed positions)",
                               "equals_column_max_of_own_rows": eq, "pass": hold_pass})
                # (2) rows of rank 0: a sampled subset of the frames of a BATCHED launch (ring slot 0)
                if rank == 0:
                    gold = (so.HackrfBranchOracle if branch == "hackrf" else so.RtlBranchOracle)(nfft, wl["fs"], precision="gold")
                    eng.reset()
                    if batch > 1:
                        step_batch(0)
                    else:
                        step(0)
                    eng.synchronize()
                    picks = (0, 1, my_frames // 2, my_frames - 1)
                    pairs = []
                    for k in picks:
                        x = so.unpack_iq_int8(in_ring[0][2 * k * hop: 2 * (k * hop + nfft)].cpu().numpy())
                        pairs.append((out_ring[0][k].cpu().numpy(), np.asarray(gold.power_levels(x))))
                    pb = parity_block(pairs, f"{len(picks)} frames of a {batch}-step launch of rank 0 + the hold trace "
                                             f"combined over {world} rank(s) at {len(pos)} positions")
                    pb["rows_pass"] = pb["pass"]
                    pb["hold_trace_pass"] = hold_pass
                    pb["hold_trace_max_db_err"] = err
                    pb["pass"] = bool(pb["rows_pass"] and hold_pass)
                    result["parity"] = pb
        if want_cpu:
            result["cpu_baseline"] = {"value": done / cpu_s, "unit": "frames/s", "cores": 1,
                                      "kind": "port", "sample": sample, "host_cores_available": os.cpu_count()}
            workers = args.cpu_workers if args.cpu_workers > 0 else physical_cores()
            result["cores_policy"] = (f"cpu_baseline: 1 core; cpu_baseline_pool: {workers} single-thread processes = one per "
                                      f"physical core by default ({physical_cores()} cores, {os.cpu_count()} logical CPUs)"
                                      + ("; not run with several ranks (the ranks' host threads share those cores)" if multi else ""))
            if not args.no_cpu_pool and not welch and not multi and not args.short_leg:
                try:
                    result["cpu_baseline_pool"] = cpu_all_cores(wl, workers, args.cpu_pool_seconds)
                except Exception as exc:               # a reported extra, never a reason to lose the bench line
                    result["cpu_baseline_pool"] = {"value": None, "error": str(exc)}
    if slab is not None:
        host_barrier()
        if peer and rank != 0:                        # rank 0 unmaps the other ranks' buffers before their owners free them
            host_barrier()
        slab.close()
        if peer and rank == 0:
            host_barrier()
    if not args.dry_run:
        eng.close()
        del in_ring, out_ring
        torch.cuda.empty_cache()
    return result


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=600)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--reps", type=int, default=7, help="timed repetitions of the step loop (median reported)")
    ap.add_argument("--config", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--ring", type=int, default=0, help="distinct input/output buffers cycled through "
                    "(0 = 2 x batch, at least 8; C4: 2)")
    ap.add_argument("--streams", type=int, default=3,
                    help="HIP streams consecutive calls rotate over (tdsa_set_overlap); 1 = strictly serial")
    ap.add_argument("--batch", type=int, default=8,
                    help="queued steps handed over per call in the `value` leg (tdsa_process_dev_batch); 1 = one launch per step")
    ap.add_argument("--min-region-s", type=float, default=0.5, help="every timed region lasts at least this long")
    ap.add_argument("--legs", default="all", choices=["all", "value", "streams", "serial"],
                    help="profiling aid: time only this submission mode (pre-roll and kernel-alone pass in its launch shape too), "
                         "so that a rocprofv3 trace of the run holds launches of one shape; the other figures of the line repeat it")
    ap.add_argument("--preroll-seconds", type=float, default=0.4, help="untimed load before the warm-up steps")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="budget of the single-thread CPU baseline leg")
    ap.add_argument("--cpu-seconds-multi", type=float, default=3.0,
                    help="budget of the single-thread CPU baseline leg when more than one rank runs (rank 0 only)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="no CPU legs (the parity block stays)")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--cpu-workers", type=int, default=0,
                    help="processes of the all-cores CPU leg: 0 = one per physical host core (BASELINE.md 3(ii)), N = N processes")
    ap.add_argument("--no-cpu-pool", action="store_true")
    ap.add_argument("--cpu-pool-seconds", type=float, default=6.0)
    ap.add_argument("--c5-shard", default="segments", choices=["segments", "captures"],
                    help="--config c5 on several GPUs: shard the 64 segments of ONE capture (strong scaling; every step pays "
                         "the cross-GPU combine) or give every GPU whole captures (weak scaling, no combine)")
    ap.add_argument("--no-frame-stats", action="store_true",
                    help="C3, one GPU: skip the extra leg with the per-frame scalars of the frame kernel's epilogue switched on")
    ap.add_argument("--c5-combine", default="auto", choices=["auto", "peer", "host"],
                    help="C5 at N > 1, segments sharded: where the ranks' partial means meet - peer = they stay in device "
                         "buffers rank 0 reads in place (HIP IPC, over xGMI), host = pinned shared memory; auto = peer when "
                         "every buffer can be mapped, else host")
    ap.add_argument("--c5-partials", default="f32", choices=["f32", "f64"],
                    help="precision the ranks' partial Welch means travel in (f32: 4 MiB per rank and step; the row moves < 1e-6 dB)")
    ap.add_argument("--synth", default="auto", choices=["auto", "numpy", "device"],
                    help="where the synthetic IQ is made: numpy (np.random.default_rng, SURVEY 8(d) to the letter; minutes for "
                         "C4), device (the same model in float64 torch ops, noise from torch.Generator); auto = numpy except "
                         "for C4 and the short legs of the default run")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="default run only: skip the short C2 / C4 / C5 legs appended under roofline.other_configs")
    ap.add_argument("--dry-other-configs", action="store_true",
                    help="with --dry-run: also walk the short C2 / C4 / C5 legs (tests of the multi-rank plumbing)")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU: exercise spawn / barrier / aggregation with a stand-in engine (no perf meaning)")
    args = ap.parse_args()
    args.short_leg = False

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and args.gpus > 1:
        sys.exit(_spawn_workers(args.gpus))          # no launcher: be our own
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    if os.environ.get("TDSA_BENCH_FAIL_RANK") == str(rank) and world > 1:     # tests/test_bench_launch.py: a worker dies
        sys.exit(3)

    import torch  # first: one HIP runtime per process (torch's bundled libamdhip64.so.7)
    import torch.distributed as dist

    if world > 1:                                    # host-side rendezvous only: barrier + gather of scalars
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # gloo announces its connections on stdout; the contract is ONE JSON line there: park fd 1 meanwhile
        sys.stdout.flush()
        saved_fd, null_fd = os.dup(1), os.open(os.devnull, os.O_WRONLY)
        os.dup2(null_fd, 1)
        try:
            import datetime
            dist.init_process_group(backend="gloo", timeout=datetime.timedelta(minutes=15))
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
            os.close(null_fd)
    comm = Comm(world, rank, local_rank, dist)

    t_run0 = time.perf_counter()
    result = run_config(args, comm, torch)

    # ---- the default run: short legs of the other GPU configurations, one driver run for all four (round-4 verdict) ----
    # At N > 1 the same legs run sharded over the ranks - C4 as named (its 65 536 frames split over the world), C5 both ways
    # (the segments of one capture over the ranks with the cross-rank combine inside each step = the strong-scaling curve
    # north_star names, and whole captures per rank) - so that one driver run per N yields every configuration's curve.
    want_others = (args.config == "c3" and not args.no_other_configs and args.legs == "all"
                   and (not args.dry_run or args.dry_other_configs))
    if want_others:
        others = {}
        leg_list = [("c2", "c2", None), ("c4", "c4", None), ("c5", "c5", "segments")]
        if world > 1:
            leg_list.append(("c5_captures", "c5", "captures"))
        for key, name, shard in leg_list:
            a = copy.copy(args)
            a.config, a.short_leg = name, True
            if shard is not None:
                a.c5_shard = shard
            a.reps, a.min_region_s, a.legs = 3, 0.2, "value"
            a.steps, a.warmup, a.preroll_seconds = 40, 8, 0.1
            a.no_cpu_baseline, a.no_cpu_pool = True, True
            t0 = time.perf_counter()
            try:
                r = run_config(a, comm, torch)
                if r is None:                          # (ranks other than 0 took part; rank 0 holds the result)
                    continue
                pb, rf, cf, tm = r.get("parity", {}), r["roofline"], r["config"], r.get("timing", {})
                others[key] = {"workload": cf["workload"], "value": r["value"], "unit": r["unit"], "n_gpus": r["n_gpus"],
                               "scaling": r["scaling"], "ms_per_step": r["ms_per_step"],
                               "frames_per_step": cf["frames_per_step_all_gpus"],
                               "steps_per_call": cf["steps_per_call"], "streams": cf["streams_per_gpu"],
                               "frac": rf.get("frac"), "achieved": rf.get("achieved"),
                               "kernel": rf.get("kernel"), "kernel_avg_us": rf.get("kernel_avg_us"),
                               "algorithmic_bytes_per_launch": rf.get("algorithmic_bytes_per_launch"),
                               "shader_clock_mhz": rf.get("shader_clock_mhz"),
                               "traffic": rf.get("traffic"), "traffic_source": rf.get("traffic_source"),
                               **({"fabric_roof": rf["fabric_roof"]} if "fabric_roof" in rf else {}),
                               **({"traffic_check": rf["traffic_check"]} if "traffic_check" in rf else {}),
                               "parity": {k: pb.get(k) for k in ("pass", "north_star_pass", "survey_8d_strict_pass",
                                                                  "max_rel_power_err", "max_db_err_top100dB", "hold_trace_pass",
                                                                  "checked")},
                               "timing": {"repetitions": tm.get("repetitions"), "min_region_s": tm.get("min_region_s"),
                                          "region_ms": tm.get("region_ms")},
                               "data": r["data"], "leg_wall_s": time.perf_counter() - t0}
                for k in ("value_compute_only", "ms_per_step_compute_only", "welch"):
                    if k in r:
                        others[key][k] = r[k]
                if "frames_per_step_per_rank" in cf:
                    others[key]["frames_per_step_per_rank"] = cf["frames_per_step_per_rank"]
                result["roofline"].setdefault("quoted_files", {}).update(rf.get("quoted_files") or {})
            except Exception as exc:                   # a reported extra, never a reason to lose the C3 line
                others[key] = {"error": f"{type(exc).__name__}: {exc}"}
        if result is not None:
            result["roofline"]["other_configs"] = others
            result["roofline"]["other_configs_are"] = ("short legs of the same bench (`value` submission mode, 3 repetitions of "
                                                       ">= 0.2 s, parity block each, no CPU legs; at N > 1 sharded over the "
                                                       "ranks like the main line): python bench.py --config cX is the full "
                                                       "run of each")
    if rank == 0:
        result["wall_s"] = time.perf_counter() - t_run0
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
