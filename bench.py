#!/usr/bin/env python3
"""bench.py - PSD frames/s of the MI355X-native IQ -> spectrum path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N == 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (config C3 of BASELINE.json / SURVEY.md 8(d), the one the metric is quoted on):
HackRF-shaped 20 Msps int8 IQ, N = 16384, hop = N/2, one second of IQ per step = 20e6 samples
-> 2440 frames, HackRF-branch semantics (per-frame DC removal, power-normalised Hann,
20*log10(|X| + 1e-12)), every frame's dB row written + a max-hold trace.  One "step" = one pass of
the hot path over that second; consecutive steps walk a ring of distinct seconds (320 MB of input,
larger than the 256 MiB Infinity Cache) so the reads really come from HBM.  Frames are independent:
with N GPUs every rank processes its own seconds, no collective in the data path (weak scaling).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
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


def cpu_all_cores(wl: dict, workers: int, seconds: float) -> dict:
    """SURVEY.md 8(d) CPU baseline (ii): the same numpy restatement in `workers` independent processes
    (one frame stream each, started together), frames/s = sum of frames / slowest worker's wall time."""
    import subprocess
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
    return {"value": frames / slowest, "unit": "frames/s", "cores": workers, "kind": "port",
            "sample": f"{workers} processes x {seconds:.0f} s of the numpy restatement, one synthetic frame stream each",
            "host_cores_available": os.cpu_count()}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=500)
    ap.add_argument("--config", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--ring", type=int, default=8, help="distinct input/output buffers cycled through")
    ap.add_argument("--streams", type=int, default=3,
                    help="HIP streams consecutive steps rotate over (tdsa_set_overlap); 1 = strictly serial")
    ap.add_argument("--preroll-seconds", type=float, default=0.4, help="untimed load before the warm-up steps")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="budget of the single-thread CPU baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-workers", type=int, default=min(32, os.cpu_count() or 1),
                    help="processes of the all-cores CPU baseline leg (0 = skip)")
    ap.add_argument("--cpu-pool-seconds", type=float, default=6.0)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py: launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
        args.gpus = world

    import torch  # first: one HIP runtime per process (torch's bundled libamdhip64.so.7)
    import torch.distributed as dist

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (torch.cuda.is_available() is False)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)

    from topdogspectrumanalyser_amd import SpectrumEngine, _native as nat
    from topdogspectrumanalyser_amd.utils.synthetic import synth_iq_int8

    wl = WORKLOADS[args.config]
    nfft, hop, ns = wl["nfft"], wl["hop"], wl["n_samples"]
    frames = (ns - nfft) // hop + 1
    ring = max(1, args.ring)

    # ---- synthetic input, resident in HBM before the timed region -------------------------------
    base = synth_iq_int8(ns, nfft, seed=3 + rank)
    ins, outs = [], []
    out_rows = 1 if wl["branch"] == "welch" else frames
    for r in range(ring):
        host = base if r == 0 else np.roll(base, 2 * 977 * r)      # distinct seconds, same statistics
        ins.append(torch.from_numpy(host).to(dev))
        outs.append(torch.empty((out_rows, nfft), dtype=torch.float32, device=dev))
    torch.cuda.synchronize()

    eng = SpectrumEngine(nfft, max_frames=frames, device=local_rank)
    if wl["branch"] == "hackrf":
        w = np.hanning(nfft).astype(np.float32)
        w /= np.sqrt(np.mean(w ** 2))
        eng.set_window(w)
        eng.configure(db_mode="mag", log_floor=1e-12, dc_alpha=1.0, hold_max=True)
    elif wl["branch"] == "welch":
        eng.set_window(np.hanning(nfft).astype(np.float32))
        eng.configure(db_mode="pow", power_scale=1.0, log_floor=1e-10, dc_alpha=-1.0, avg=("lin", frames),
                      cal_offset_db=-0.8087054556396822)
    else:
        eng.set_window(np.hanning(nfft).astype(np.float32))
        eng.configure(db_mode="pow", power_scale=1.0, log_floor=1e-10, dc_alpha=-1.0, hold_max=True)

    # consecutive seconds are independent (per-frame DC removal, no averaging): let the head of step i+1
    # fill the ragged tail of step i's persistent launch (9 or 10 frames per workgroup at C3)
    streams = max(1, min(4, args.streams)) if wl["branch"] != "welch" else 1
    eng.set_overlap(streams)

    def step(i: int) -> None:
        r = i % ring
        if wl["branch"] == "welch":
            eng.reset(nat.RESET_AVG)               # every step is one complete Welch average
        eng.process_device(nat.IN_I8, ins[r].data_ptr(), ns, hop, frames, outs[r].data_ptr())

    def fence() -> None:
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # clocks: an idle MI355X needs a few hundred ms of load before shader/fabric clocks settle; this
    # untimed pre-roll keeps short --steps/--warmup runs from measuring the ramp
    t_pre = time.perf_counter()
    i_pre = 0
    while time.perf_counter() - t_pre < args.preroll_seconds:
        for _ in range(50):
            step(i_pre)
            i_pre += 1
        eng.synchronize()
    for i in range(args.warmup):
        step(i)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- dominant kernel alone: HIP events on the plan's stream around every frame-kernel launch,
    #      launches strictly serial so that one kernel owns the GPU while it is timed ---------------
    eng.set_overlap(1)
    eng.profile_enable(True)
    for i in range(args.steps):
        step(i)
    launches, kern_ms = eng.profile_read()
    eng.profile_enable(False)
    # (2^20-point plans run a chain of kernels: price the whole step instead)
    kern_s = kern_ms * 1e-3 / launches if launches else elapsed / args.steps
    if wl["branch"] == "welch":                   # one dB row per K segments: 2N + 4N/K per segment
        bytes_per_frame = 2 * hop + 4 * nfft // frames
    else:
        bytes_per_frame = 2 * hop + 4 * nfft      # every input byte read once, every
    algo_bytes = frames * bytes_per_frame          # output byte written once
    achieved_gbs = algo_bytes / kern_s / 1e9

    # per-GPU hold traces combined on the host; outside the timed region
    mx, _ = eng.hold()
    if world > 1 and mx is not None:
        gathered = [torch.empty(nfft, dtype=torch.float32, device=dev) for _ in range(world)]
        dist.all_gather(gathered, torch.from_numpy(mx).to(dev))
        mx = np.fmax.reduce([g.cpu().numpy() for g in gathered])

    # HBM bytes per launch from the committed rocprofv3 PMC passes of this same kernel and shape
    # (profiles/r01_c3_pmc.json; counters cannot be read from inside the process)
    traffic, traffic_src = None, None
    pmc_path = os.path.join(ROOT, "profiles", f"r01_{args.config}_pmc.json")
    if os.path.exists(pmc_path):
        with open(pmc_path) as fh:
            pmc = json.load(fh)
        traffic = pmc["fetch_bytes_upper"] + pmc["write_bytes"]
        traffic_src = f"profiles/r01_{args.config}_pmc.json (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, per launch)"

    result = None
    if rank == 0:
        value = world * frames * args.steps / elapsed
        result = {
            "metric": "PSD frames/sec at 16384-pt FFT on synthetic 20 Msps IQ" if args.config == "c3"
                      else f"PSD frames/sec ({args.config})",
            "value": value,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.config}: {wl['desc']}", "nfft": nfft, "hop": hop,
                       "frames_per_step_per_gpu": frames, "input": "int8 IQ resident in HBM",
                       "input_ring": ring, "streams_per_gpu": streams,
                       "parallelism": f"frames sharded over {world} GPU(s), no collective"},
            "roofline": {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_unit": "bytes per launch", "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": algo_bytes,
                         "kernel": "spectrum_kernel" if launches else "four-step chain (whole step)",
                         "kernel_avg_us": kern_s * 1e6,
                         "algorithmic_bytes_per_frame": bytes_per_frame},
        }

    # ---- CPU baseline + parity spot check: rank 0, single GPU runs only ---------------------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline and wl["branch"] != "welch":
        from oracle import spectrum_oracle as so   # checker / reported baseline only
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
        worst_rel, worst_db = 0.0, 0.0
        for k in (0, 1, frames // 2, frames - 1):
            x = so.unpack_iq_int8(base[2 * k * hop: 2 * (k * hop + nfft)])
            g = np.asarray(gold.power_levels(x))
            rel, ddb = so.parity_metrics(outs[0][k].cpu().numpy(), g)
            worst_rel, worst_db = max(worst_rel, rel), max(worst_db, ddb)
        result["cpu_baseline"] = {"value": done / cpu_s, "unit": "frames/s", "cores": 1, "kind": "port",
                                  "sample": f"{done} frames ({args.cpu_seconds:.0f} s) cycling through the same second "
                                            f"of IQ, single thread, numpy {np.__version__} restatement of "
                                            f"get_power_levels incl. int8 unpack",
                                  "host_cores_available": os.cpu_count()}
        result["parity"] = {"max_rel_power_err": worst_rel, "max_db_err_top60dB": worst_db,
                            "frames_checked": 4, "against": "float64 gold oracle"}
        if args.cpu_workers > 0:
            try:
                result["cpu_baseline_all_cores"] = cpu_all_cores(wl, args.cpu_workers, args.cpu_pool_seconds)
            except Exception as exc:               # a reported extra, never a reason to lose the bench line
                result["cpu_baseline_all_cores"] = {"value": None, "error": str(exc)}
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
