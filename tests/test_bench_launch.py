"""bench.py launch conventions (no GPU): `python bench.py --gpus N` spawns its own workers, ranks meet over
gloo on 127.0.0.1, rank 0 prints ONE well-formed JSON line; the same under torch.distributed.run.  The
stand-in engine of --dry-run does no arithmetic: only the plumbing of the multi-GPU leg is exercised."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _run(cmd):
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    env.pop("RANK", None), env.pop("WORLD_SIZE", None), env.pop("LOCAL_RANK", None)
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), out.stdout      # ONE JSON line on stdout, nothing else
    return json.loads(lines[0])


def _check_line(d, n):
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "value_serial", "per_gpu_frames_per_s"):
        assert key in d, key
    assert d["n_gpus"] == n and len(d["per_gpu_frames_per_s"]) == n and d["scaling"] == "weak"
    assert d["value_streams"] > 0 and d["timing"]["steps_per_call"] == 8     # `value`: 8 queued steps per call
    assert d["metric"].startswith("PSD frames/sec at 16384-pt FFT") and d["unit"] == "frames/s"
    assert d["steps"] == 40 and d["warmup"] == 3 and d["value"] > 0 and d["ms_per_step"] > 0
    assert "dry-run" in d["data"] and d["roofline"]["frac"] is None
    assert d["timing"]["repetitions"] == 3 and d["timing"]["steps_per_region"] == 40 * d["timing"]["inner_repeats"]
    assert min(d["timing"]["region_ms"]) >= 50.0          # every timed region lasts >= 50 ms
    assert "no collective" in d["config"]["parallelism"]
    # the N > 1 line carries its own evidence (round-3 verdict): a parity block, the hold trace combined over ALL ranks
    # (core/display_data_processor.py:371-382 is what it must equal) and a CPU baseline leg - in --dry-run the per-rank
    # stand-in traces (rank r holds 100 + r in bin r) travel the same gather + np.fmax combine
    ht = d["hold_trace"]
    assert ht["ranks_combined"] == n and ht["max_db"] == 100.0 + (n - 1) and ht["argmax_bin"] == n - 1
    assert "pass" in d["parity"] and "hold_trace_pass" in d["parity"]
    assert d["cpu_baseline"]["cores"] == 1 and d["cpu_baseline"]["kind"] == "port"
    assert ("3 s" if n > 1 else "10 s") in d["cpu_baseline"]["sample"]
    assert all("sha256_16" in v for v in d["roofline"]["quoted_files"].values())


@pytest.mark.parametrize("n", [1, 2])
def test_bench_spawns_its_own_workers(n):
    d = _run([sys.executable, "bench.py", "--gpus", str(n), "--steps", "40", "--warmup", "3", "--reps", "3", "--dry-run",
              "--min-region-s", "0.05"])
    _check_line(d, n)
    assert d["config"]["launcher"] == ("self-spawned workers" if n > 1 else "single process")


def test_bench_under_torch_distributed_run():
    d = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
              "127.0.0.1", "--master-port", "29517", "bench.py", "--gpus", "2", "--steps", "40", "--warmup", "3",
              "--reps", "3", "--dry-run", "--min-region-s", "0.05"])
    _check_line(d, 2)
    assert d["config"]["launcher"] == "torch.distributed.run"


@pytest.mark.parametrize("n", [1, 2, 3, 8])
def test_bench_c5_shards_the_welch_segments_over_the_ranks(n):
    """--config c5 --gpus N: the 64 segments of one capture are split over the ranks (strong scaling), the float64
    partial means + counts are gathered over gloo and combined with sharding.combine_welch on rank 0 (SURVEY.md 8(e);
    utils/signal_processing.py:35-61 of the reference is the running mean being reassembled).  The stand-in engine of
    --dry-run reports the constant rank + 1 as its partial mean: the combined mean is the count-weighted average."""
    d = _run([sys.executable, "bench.py", "--gpus", str(n), "--config", "c5", "--steps", "4", "--warmup", "1", "--reps", "2",
              "--dry-run", "--min-region-s", "0.05"])
    assert d["n_gpus"] == n and d["scaling"] == "strong" and d["metric"] == "PSD frames/sec (c5)"
    w = d["welch"]
    counts = w["segments_per_rank"]
    assert len(counts) == n and sum(counts) == 64 == w["segments_total"] and max(counts) - min(counts) <= 1
    expect = sum((r + 1) * c for r, c in enumerate(counts)) / 64.0
    assert abs(w["mean_of_means"] - expect) < 1e-12
    assert d["config"]["frames_per_step_all_gpus"] == 64 and "sharded over" in d["config"]["parallelism"]
    assert len(d["per_gpu_frames_per_s"]) == n
    # round-4 verdict: the cross-rank combine is part of every step - `value` is end to end (the ranks' partial means
    # travel through the shared-memory slab and are combined inside the timed region: in --dry-run by the stand-in's
    # count-weighted mean, which is where mean_of_means above comes from), `value_compute_only` leaves it out
    assert w["shard"] == "segments" and d["value_compute_only"] > 0 and d["ms_per_step_compute_only"] > 0
    assert w["combine_ms"] >= 0.0
    if n > 1:
        assert "END TO END" in d["timing"]["value_is"] and d["value"] <= d["value_compute_only"] * 1.5
        assert w["rank0_upload_combine_ms"] >= 0.0 and w["rank0_wait_for_partials_ms"] >= 0.0
        assert "shared-memory slab" in w["partials"] and "no collective" in w["partials"]
        assert abs(d["ms_per_step"] - d["ms_per_step_compute_only"] - w["combine_ms"]) < 1e-9 or w["combine_ms"] == 0.0
    else:
        assert d["value_compute_only"] == d["value"] and w["combine_ms"] == 0.0


@pytest.mark.parametrize("n", [2, 8])
def test_bench_c5_whole_captures_per_rank(n):
    """--c5-shard captures: every rank averages whole captures of 64 segments - weak scaling, nothing to combine."""
    d = _run([sys.executable, "bench.py", "--gpus", str(n), "--config", "c5", "--c5-shard", "captures", "--steps", "4",
              "--warmup", "1", "--reps", "2", "--dry-run", "--min-region-s", "0.05"])
    assert d["n_gpus"] == n and d["scaling"] == "weak"
    assert d["welch"]["shard"] == "captures" and d["welch"]["combine_ms"] == 0.0
    assert d["welch"]["segments_per_rank"] == [64] * n and d["config"]["frames_per_step_all_gpus"] == 64 * n
    assert d["value_compute_only"] == d["value"] and "nothing to combine" in d["config"]["parallelism"]


@pytest.mark.parametrize("n", [1, 2])
def test_bench_c4_is_the_named_waterfall_sharded_over_the_ranks(n):
    """--config c4: BASELINE config 4 as named - ONE waterfall of 65 536 frames x 8192 points (displays/waterfall.py:163-180
    is the layout its rows have), its frames sharded over the ranks in contiguous ranges: strong scaling, no combine."""
    d = _run([sys.executable, "bench.py", "--gpus", str(n), "--config", "c4", "--steps", "4", "--warmup", "1", "--reps", "2",
              "--dry-run", "--min-region-s", "0.05"])
    assert d["scaling"] == "strong" and d["config"]["frames_per_step_all_gpus"] == 65536
    assert d["config"]["frames_per_step_per_rank"] == [65536 // n] * n and d["config"]["steps_per_call"] == 1
    assert "sharded over" in d["config"]["parallelism"] and "no collective" in d["config"]["parallelism"]


@pytest.mark.parametrize("n", [1, 2, 8])
def test_default_run_carries_the_other_configurations_at_every_world_size(n):
    """The default run (C3) appends short legs of the other GPU configurations under roofline.other_configs - at N > 1
    sharded over the same ranks: C2 (every rank its own captures), C4 as named (65 536 frames split over the world), C5
    with the segments of one capture split (strong; cross-rank combine inside the step) and with whole captures per rank
    (weak) - so that one driver run per N yields every configuration's curve."""
    d = _run([sys.executable, "bench.py", "--gpus", str(n), "--steps", "40", "--warmup", "3", "--reps", "3", "--dry-run",
              "--dry-other-configs", "--min-region-s", "0.05"])
    _check_line(d, n)
    oc = d["roofline"]["other_configs"]
    assert set(oc) == ({"c2", "c4", "c5"} | ({"c5_captures"} if n > 1 else set())), oc.keys()
    for k, v in oc.items():
        assert "error" not in v, (k, v)
        assert v["n_gpus"] == n and v["value"] > 0
    assert oc["c2"]["scaling"] == "weak" and oc["c2"]["frames_per_step"] == 4096 * n
    assert oc["c4"]["scaling"] == "strong" and oc["c4"]["frames_per_step"] == 65536
    assert oc["c4"]["frames_per_step_per_rank"] == [65536 // n] * n
    assert oc["c5"]["scaling"] == "strong" and oc["c5"]["frames_per_step"] == 64
    assert oc["c5"]["welch"]["shard"] == "segments" and sum(oc["c5"]["welch"]["segments_per_rank"]) == 64
    if n > 1:
        assert oc["c5"]["welch"]["combine_ms"] >= 0.0 and "value_compute_only" in oc["c5"]
        assert oc["c5_captures"]["scaling"] == "weak" and oc["c5_captures"]["frames_per_step"] == 64 * n
        assert oc["c5_captures"]["welch"]["shard"] == "captures"


def test_bench_workers_are_all_reaped_when_one_fails(tmp_path):
    """a failing worker must not leave the others parked in the gloo barrier - every worker is waited
    for (or terminated) and the first non-zero exit code comes back."""
    import time
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", TDSA_BENCH_FAIL_RANK="1")
    env.pop("RANK", None), env.pop("WORLD_SIZE", None), env.pop("LOCAL_RANK", None)
    t0 = time.time()
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "4", "--warmup", "1", "--reps", "1",
                          "--dry-run", "--min-region-s", "0.05"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and time.time() - t0 < 60


def test_bench_has_no_rccl_in_it():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "nccl" not in src.lower() and 'backend="gloo"' in src
