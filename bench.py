#!/usr/bin/env python3
"""bench.py - PSD frames/s of the MI355X-native IQ -> spectrum path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]        (any N: spawns one worker per GPU itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W        (same result)

Workload (config C3 of BASELINE.json / SURVEY.md 8(d), the one the metric is quoted on):
HackRF-shaped 20 Msps int8 IQ, N = 16384, hop = N/2, one second of IQ per step = 20e6 samples
-> 2440 frames, HackRF-branch semantics (per-frame DC removal, power-normalised Hann,
20*log10(|X| + 1e-12)), every frame's dB row written + a max-hold trace.  One "step" = one pass of
the hot path over that second; consecutive steps walk a ring of distinct seconds (320 MB of input,
larger than the 256 MiB Infinity Cache) so the reads really come from HBM.

Multi-GPU: frames are independent, every rank processes its own seconds on its own
GPU with its own plan; there is NO collective in the data path and no RCCL anywhere: ranks meet at a
host-side (gloo/TCP) barrier around the timed region and the per-GPU hold traces are combined on the
host with np.fmax.  Aggregate = all ranks' frames / slowest rank's time (weak scaling).

Timing: after W warm-up steps the step loop of K x `inner` steps is timed `--reps` times, each time
fenced (device synchronize + host barrier) on both sides; `inner` is chosen so that one timed region
lasts >= 50 ms whatever K is; the MEDIAN repetition (max over ranks) gives ms_per_step and value.

Prints ONE JSON line on rank 0.
"""
import argparse
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
    "c4": dict(nfft=8192, hop=8192, n_samples=8192 * 8192, fs=20e6, branch="hackrf",
               desc="Batched waterfall: 8192 frames x 8192-pt FFT per GPU"),
    "c5": dict(nfft=1 << 20, hop=1 << 20, n_samples=64 << 20, fs=2e6, branch="welch",
               desc="Wideband stitch: 1M-pt FFT, Welch average of 64 segments + calibration offset"),
}
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak (MI355X_MICROARCH.md)
CAL_DB = -0.8087054556396822
MIN_REGION_S = 0.050


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
            p.wait(timeout=30)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    avail = os.cpu_count() or workers
    return {"value": frames / slowest, "unit": "frames/s", "cores": workers, "kind": "port",
            "policy": f"{workers} single-thread worker processes (one per core used) of {avail} host cores; "
                      f"--cpu-workers 0 = one per host core",
            "sample": f"{workers} processes x {seconds:.0f} s of the numpy restatement, one synthetic frame stream each",
            "host_cores_available": avail}


class _DryEngine:
    """--dry-run only: stands in for the GPU plan so that the launch / barrier / aggregation plumbing of
    the multi-GPU leg can be exercised in a container without a GPU.  Does no arithmetic; the line it
    produces is marked as a dry run and carries no performance meaning."""

    def __init__(self, nfft):
        self.nfft, self._n, self._ms = nfft, 0, 0.0

    def step(self):
        time.sleep(2e-5)
        self._n += 1

    def synchronize(self):
        pass


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _spawn_workers(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: one worker process per GPU, rendezvous on 127.0.0.1.
    Rank 0's stdout (the JSON line) is passed through."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), TDSA_BENCH_WORKER="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        rc = rc or p.wait()
    return rc


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=600)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--reps", type=int, default=7, help="timed repetitions of the step loop (median reported)")
    ap.add_argument("--config", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--ring", type=int, default=8, help="distinct input/output buffers cycled through")
    ap.add_argument("--streams", type=int, default=3,
                    help="HIP streams consecutive steps rotate over (tdsa_set_overlap); 1 = strictly serial")
    ap.add_argument("--preroll-seconds", type=float, default=0.4, help="untimed load before the warm-up steps")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="budget of the single-thread CPU baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-workers", type=int, default=-1,
                    help="processes of the all-cores CPU leg: -1 = min(32, host cores), 0 = one per host core")
    ap.add_argument("--no-cpu-pool", action="store_true")
    ap.add_argument("--cpu-pool-seconds", type=float, default=6.0)
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU: exercise spawn / barrier / aggregation with a stand-in engine (no perf meaning)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and args.gpus > 1:
        sys.exit(_spawn_workers(args.gpus))          # no launcher: be our own
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world

    import torch  # first: one HIP runtime per process (torch's bundled libamdhip64.so.7)
    import torch.distributed as dist

    if world > 1:                                    # host-side rendezvous only: barrier + gather of scalars
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # gloo announces its connections on stdout; the contract is ONE JSON line there: park fd 1 meanwhile
        sys.stdout.flush()
        saved_fd, null_fd = os.dup(1), os.open(os.devnull, os.O_WRONLY)
        os.dup2(null_fd, 1)
        try:
            dist.init_process_group(backend="gloo")
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
            os.close(null_fd)

    def host_barrier() -> None:
        if world > 1:
            dist.barrier()

    def gather(obj):
        if world == 1:
            return [obj]
        out = [None] * world
        dist.all_gather_object(out, obj)
        return out

    wl = WORKLOADS[args.config]
    nfft, hop, ns = wl["nfft"], wl["hop"], wl["n_samples"]
    frames = (ns - nfft) // hop + 1
    ring = max(1, args.ring)
    welch = wl["branch"] == "welch"

    if args.dry_run:
        eng = _DryEngine(nfft)
        visible, dev_index = 0, -1
        step = lambda i: eng.step()                   # noqa: E731
        dev_sync = lambda: None                       # noqa: E731
        streams = 1
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
        base = synth_iq_int8(ns, nfft, seed=3 + rank)
        ins, outs = [], []
        out_rows = 1 if welch else frames
        for r in range(ring):
            host = base if r == 0 else np.roll(base, 2 * 977 * r)      # distinct seconds, same statistics
            ins.append(torch.from_numpy(host).to(dev))
            outs.append(torch.empty((out_rows, nfft), dtype=torch.float32, device=dev))
        torch.cuda.synchronize()

        eng = SpectrumEngine(nfft, max_frames=frames, device=dev_index)
        if wl["branch"] == "hackrf":
            w = np.hanning(nfft).astype(np.float32)
            w /= np.sqrt(np.mean(w ** 2))
            eng.set_window(w)
            eng.configure(db_mode="mag", log_floor=1e-12, dc_alpha=1.0, hold_max=True)
        elif welch:
            eng.set_window(np.hanning(nfft).astype(np.float32))
            eng.configure(db_mode="pow", power_scale=1.0, log_floor=1e-10, dc_alpha=-1.0, avg=("lin", frames),
                          cal_offset_db=CAL_DB)
        else:
            eng.set_window(np.hanning(nfft).astype(np.float32))
            eng.configure(db_mode="pow", power_scale=1.0, log_floor=1e-10, dc_alpha=-1.0, hold_max=True)
        streams = max(1, min(4, args.streams)) if not welch else 1

        def step(i: int) -> None:
            r = i % ring
            if welch:
                eng.reset(nat.RESET_AVG)               # every step is one complete Welch average
            eng.process_device(nat.IN_I8, ins[r].data_ptr(), ns, hop, frames, outs[r].data_ptr())

        def dev_sync() -> None:
            eng.synchronize()
            torch.cuda.synchronize()

    def fence() -> None:
        dev_sync()
        host_barrier()
        dev_sync()

    rank_times = []                                   # per timed region: every rank's own seconds

    def timed_loop(n_steps: int) -> float:
        """one timed region: fence, n_steps steps, device synchronize, fence; seconds of the slowest rank"""
        fence()
        t0 = time.perf_counter()
        for i in range(n_steps):
            step(i)
        dev_sync()
        own = time.perf_counter() - t0                # this rank's own work, before it waits for the others
        fence()
        every = gather(own)
        rank_times.append(every)
        return max(every)

    def measure(n_streams: int):
        """-> (median seconds per region, all region times, inner)"""
        if not args.dry_run:
            eng.set_overlap(n_streams)
        for i in range(args.warmup):
            step(i)
        est = timed_loop(args.steps) / max(1, args.steps)          # calibration region (also warm-up)
        inner = max(1, int(np.ceil(1.2 * MIN_REGION_S / max(est * args.steps, 1e-9))))
        for _ in range(4):                                         # a region that came out short is re-timed longer
            inner = max(gather(inner))                             # same loop count on every rank
            del rank_times[:]
            times = [timed_loop(args.steps * inner) for _ in range(max(1, args.reps))]
            if min(times) >= MIN_REGION_S:
                break
            inner = int(np.ceil(inner * 1.3 * MIN_REGION_S / max(min(times), 1e-9)))
        per_rank = [statistics.median(t[r] for t in rank_times) for r in range(world)]
        return statistics.median(times), times, inner, per_rank

    # clocks: an idle MI355X needs a few hundred ms of load before shader/fabric clocks settle; this
    # untimed pre-roll keeps short --steps/--warmup runs from measuring the ramp
    t_pre = time.perf_counter()
    i_pre = 0
    while time.perf_counter() - t_pre < args.preroll_seconds:
        for _ in range(50):
            step(i_pre)
            i_pre += 1
        dev_sync()

    med, times, inner, per_rank_s = measure(streams)
    if streams > 1:
        med_serial, times_serial, inner_serial, _ = measure(1)
    else:
        med_serial, times_serial, inner_serial = med, times, inner
    steps_timed = args.steps * inner
    per_rank_fps = [frames * steps_timed / t for t in per_rank_s]  # each rank's own rate (median region)

    # ---- dominant kernel alone: HIP events on the plan's stream around every frame-kernel launch,
    #      launches strictly serial so that one kernel owns the GPU while it is timed ---------------
    launches, kern_ms = 0, 0.0
    if not args.dry_run:
        eng.set_overlap(1)
        eng.profile_enable(True)
        n_prof = min(steps_timed, 2000)
        for i in range(n_prof):
            step(i)
        launches, kern_ms = eng.profile_read()
        eng.profile_enable(False)
        if welch:                                      # a chain of kernels: price the whole serial step
            launches = 0
    # (2^20-point plans run a chain of kernels: price the whole serial step instead)
    kern_s = kern_ms * 1e-3 / launches if launches else med_serial / (args.steps * inner_serial)
    if welch:                                          # one dB row per K segments: 2N + 4N/K per segment
        bytes_per_frame = 2 * hop + 4 * nfft // frames
    else:
        bytes_per_frame = 2 * hop + 4 * nfft           # every input byte read once, every
    algo_bytes = frames * bytes_per_frame              # output byte written once
    achieved_gbs = algo_bytes / kern_s / 1e9
    per_gpu_frac = gather(achieved_gbs / HBM_PEAK_GBS)

    # per-GPU hold traces combined on the HOST; outside the timed region
    hold_combined = None
    if not args.dry_run:
        from topdogspectrumanalyser_amd.sharding import combine_hold
        mx, _ = eng.hold()
        parts = [p for p in gather(mx) if p is not None]
        if parts:
            hold_combined = combine_hold(parts, "max")

    # HBM bytes per launch from the committed rocprofv3 PMC passes of this same kernel and shape
    # (counters cannot be read from inside the process)
    traffic, traffic_src = None, None
    for rnd in ("r02", "r01"):
        pmc_path = os.path.join(ROOT, "profiles", f"{rnd}_{args.config}_pmc.json")
        if os.path.exists(pmc_path):
            with open(pmc_path) as fh:
                pmc = json.load(fh)
            traffic = pmc["fetch_bytes_upper"] + pmc["write_bytes"]
            traffic_src = f"profiles/{rnd}_{args.config}_pmc.json (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, per launch)"
            break

    result = None
    if rank == 0:
        value = world * frames * steps_timed / med
        value_serial = world * frames * args.steps * inner_serial / med_serial
        result = {
            "metric": "PSD frames/sec at 16384-pt FFT on synthetic 20 Msps IQ" if args.config == "c3"
                      else f"PSD frames/sec ({args.config})",
            "value": value,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": med / steps_timed * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" if not args.dry_run else "dry-run (stand-in engine, no GPU work, no perf meaning)",
            "value_serial": value_serial,
            "ms_per_step_serial": med_serial / (args.steps * inner_serial) * 1e3,
            "per_gpu_frames_per_s": per_rank_fps,
            "timing": {"repetitions": len(times), "inner_repeats": inner, "steps_per_region": steps_timed,
                       "region_ms": [t * 1e3 for t in times], "region_ms_serial": [t * 1e3 for t in times_serial],
                       "statistic": "median over repetitions of (max over ranks)",
                       "value_is": (f"{streams} stream(s) per GPU" + (", launches sized for half the CUs (two side by side)"
                                                                         if streams >= 3 else ""))
                       if streams > 1 else "strictly serial launches",
                       "value_serial_is": "strictly serial launches (1 stream)"},
            "config": {"workload": f"{args.config}: {wl['desc']}", "nfft": nfft, "hop": hop,
                       "frames_per_step_per_gpu": frames, "input": "int8 IQ resident in HBM",
                       "input_ring": ring, "streams_per_gpu": streams, "inner_repeats": inner,
                       "gpus_visible_per_process": visible,
                       "launcher": "torch.distributed.run" if not os.environ.get("TDSA_BENCH_WORKER") and world > 1
                                   else ("self-spawned workers" if world > 1 else "single process"),
                       "parallelism": f"frames sharded over {world} GPU(s), one process + plan per GPU, no collective "
                                      f"(host gloo barrier around the timed region, hold traces combined with np.fmax)"},
            "roofline": {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved_gbs / HBM_PEAK_GBS, "frac_per_gpu": per_gpu_frac, "traffic": traffic,
                         "traffic_unit": "bytes per launch", "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": algo_bytes,
                         "kernel": "spectrum_kernel" if launches else "column pass + row pass + gather + finish (whole serial step)",
                         "kernel_avg_us": kern_s * 1e6,
                         "kernel_avg_from": "HIP events on the plan's stream around every launch of a further, strictly "
                                            "serial pass (one launch on the whole chip at a time)",
                         "algorithmic_bytes_per_frame": bytes_per_frame},
        }
        if args.dry_run:                               # nothing was computed: no performance figures
            for k in ("achieved", "frac", "frac_per_gpu", "traffic", "kernel_avg_us"):
                result["roofline"][k] = None
            result["roofline"]["kernel"] = "none (dry run)"
        if hold_combined is not None:
            result["hold_trace"] = {"combined_on": "host (np.fmax over ranks)", "max_db": float(np.max(hold_combined)),
                                    "argmax_bin": int(np.argmax(hold_combined))}

    # ---- CPU baseline + parity spot check: rank 0, single GPU runs only ---------------------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.dry_run:
        from oracle import spectrum_oracle as so   # checker / reported baseline only
        if welch:
            seg = lambda k: so.unpack_iq_int8(base[2 * k * hop: 2 * (k * hop + nfft)])   # noqa: E731
            br = so.RtlBranchOracle(nfft, wl["fs"], precision="ref")
            br.averager.set_mode("lin", frames)
            t_cpu0 = time.perf_counter()
            done = 0
            while done < 3 or (time.perf_counter() - t_cpu0 < args.cpu_seconds and done < frames):
                br.power_levels(seg(done % frames))
                done += 1
            cpu_s = time.perf_counter() - t_cpu0
            # parity: the whole Welch average (all K segments) against the float64 gold
            gold = so.RtlBranchOracle(nfft, wl["fs"], precision="gold")
            gold.averager.set_mode("lin", frames)
            g = None
            for k in range(frames):
                g = gold.power_levels(seg(k))
            g = np.asarray(g, dtype=np.float64) + CAL_DB
            eng.reset()
            eng.process_device(nat.IN_I8, ins[0].data_ptr(), ns, hop, frames, outs[0].data_ptr())
            eng.synchronize()
            got = outs[0][0].cpu().numpy()
            rel, ddb = so.parity_metrics(got, g, floor_rel_db=100.0, amp_floor=2 * so.AMP_FLOOR)
            checked, sample = f"Welch mean of all {frames} segments", \
                f"{done} segments of 2^20 points, single thread, numpy {np.__version__} restatement incl. int8 unpack"
            worst_rel, worst_db = rel, ddb
            raw60, raw100 = so.parity_raw_db(got, g, 60.0), so.parity_raw_db(got, g, 100.0)
        else:
            if wl["branch"] == "hackrf":
                br = so.HackrfBranchOracle(nfft, wl["fs"], precision="ref")
                gold = so.HackrfBranchOracle(nfft, wl["fs"], precision="gold")
            else:
                br = so.RtlBranchOracle(nfft, wl["fs"], precision="ref")
                gold = so.RtlBranchOracle(nfft, wl["fs"], precision="gold")
            t_cpu0 = time.perf_counter()
            done = 0
            while time.perf_counter() - t_cpu0 < args.cpu_seconds:      # the same second of IQ, over and over
                k = done % frames
                x = so.unpack_iq_int8(base[2 * k * hop: 2 * (k * hop + nfft)])
                br.power_levels(x)
                done += 1
            cpu_s = time.perf_counter() - t_cpu0
            # parity of a sampled subset of the GPU frames (ring slot 0 holds `base`)
            eng.reset()
            eng.process_device(nat.IN_I8, ins[0].data_ptr(), ns, hop, frames, outs[0].data_ptr())
            eng.synchronize()
            worst_rel, worst_db, raw60, raw100 = 0.0, 0.0, 0.0, 0.0
            picks = (0, 1, frames // 2, frames - 1)
            for k in picks:
                x = so.unpack_iq_int8(base[2 * k * hop: 2 * (k * hop + nfft)])
                g = np.asarray(gold.power_levels(x))
                got = outs[0][k].cpu().numpy()
                rel, ddb = so.parity_metrics(got, g, floor_rel_db=100.0, amp_floor=2 * so.AMP_FLOOR)
                worst_rel, worst_db = max(worst_rel, rel), max(worst_db, ddb)
                raw60, raw100 = max(raw60, so.parity_raw_db(got, g, 60.0)), max(raw100, so.parity_raw_db(got, g, 100.0))
            checked = f"{len(picks)} frames"
            sample = (f"{done} frames ({args.cpu_seconds:.0f} s) cycling through the same second of IQ, single "
                      f"thread, numpy {np.__version__} restatement of get_power_levels incl. int8 unpack")
        result["cpu_baseline"] = {"value": done / cpu_s, "unit": "frames/s", "cores": 1, "kind": "port",
                                  "sample": sample, "host_cores_available": os.cpu_count()}
        result["parity"] = {"max_rel_power_err": worst_rel, "max_db_err_top60dB": raw60, "max_db_err_top100dB": raw100,
                            "db_err_over_allowance_x1e-3": worst_db, "checked": checked, "against": "float64 gold oracle",
                            "bounds": "rel <= 1e-4 of the frame maximum; |dB| <= 1e-3 within 100 dB of it, or two "
                                      "float32 rounding units (2^-23) of the frame's largest amplitude where that is "
                                      "worth more (bins deeper than 60 dB): db_err_over_allowance_x1e-3 <= 1e-3",
                            "pass": bool(worst_rel <= 1e-4 and worst_db <= 1e-3)}
        workers = args.cpu_workers
        if workers < 0:
            workers = min(32, os.cpu_count() or 1)
        elif workers == 0:
            workers = os.cpu_count() or 1
        if not args.no_cpu_pool and not welch:
            try:
                result["cpu_baseline_pool"] = cpu_all_cores(wl, workers, args.cpu_pool_seconds)
            except Exception as exc:               # a reported extra, never a reason to lose the bench line
                result["cpu_baseline_pool"] = {"value": None, "error": str(exc)}
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
