"""N > 1 path on CPU: two gloo processes shard the frames of one IQ stream (contiguous ranges + halo),
each computes its shard with the CPU oracle (no GPU here), and the per-rank hold traces / Welch partials
are combined on the host exactly as bench.py / a multi-GPU caller does: the result
must equal the single-process answer.  No collective is used for the data path itself."""
import os
import socket

import numpy as np
import pytest

from oracle import spectrum_oracle as so
from topdogspectrumanalyser_amd import sharding

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

NFFT, HOP, NF = 1024, 512, 37


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, iq, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    f0, f1 = sharding.shard_frames(NF, rank, world)
    s0, s1 = sharding.shard_samples(f0, f1, NFFT, HOP)
    my_iq = iq[2 * s0: 2 * s1]                                   # this rank's slice incl. halo
    db, mx, mn = so.hackrf_batch(my_iq, NFFT, HOP, 20e6, n_frames=f1 - f0, precision="gold")
    # Welch partial of this shard: mean of linear power + count
    lin = 10.0 ** (db / 10.0)
    mean = lin.mean(axis=0)
    # gather the small per-rank state on every rank (what the host does after the GPUs finish)
    g_mx = [torch.zeros(NFFT, dtype=torch.float64) for _ in range(world)]
    g_mn = [torch.zeros(NFFT, dtype=torch.float64) for _ in range(world)]
    g_mean = [torch.zeros(NFFT, dtype=torch.float64) for _ in range(world)]
    g_cnt = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(g_mx, torch.from_numpy(mx))
    dist.all_gather(g_mn, torch.from_numpy(mn))
    dist.all_gather(g_mean, torch.from_numpy(mean))
    dist.all_gather(g_cnt, torch.tensor([f1 - f0]))
    if rank == 0:
        np.savez(os.path.join(out_dir, "combined.npz"),
                 mx=sharding.combine_hold([t.numpy() for t in g_mx], "max"),
                 mn=sharding.combine_hold([t.numpy() for t in g_mn], "min"),
                 welch=sharding.combine_welch([t.numpy() for t in g_mean], [int(c) for c in g_cnt])[0],
                 total=sum(int(c) for c in g_cnt))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process(tmp_path):
    iq = so.synth_iq_int8(HOP * (NF - 1) + NFFT, NFFT, seed=5)
    mp.spawn(_worker, args=(2, _free_port(), iq, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "combined.npz")
    db, mx, mn = so.hackrf_batch(iq, NFFT, HOP, 20e6, precision="gold")
    assert int(got["total"]) == NF
    assert np.array_equal(got["mx"], mx) and np.array_equal(got["mn"], mn)
    assert np.allclose(got["welch"], (10.0 ** (db / 10.0)).mean(axis=0), rtol=1e-12, atol=0)


@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
@pytest.mark.parametrize("nf", [0, 1, 7, 64, 2440])
def test_shard_ranges_partition_the_frames(world, nf):
    ranges = [sharding.shard_frames(nf, r, world) for r in range(world)]
    assert ranges[0][0] == 0 and ranges[-1][1] == nf
    for (a0, a1), (b0, b1) in zip(ranges, ranges[1:]):
        assert a1 == b0 and a0 <= a1
    sizes = [b - a for a, b in ranges]
    assert max(sizes) - min(sizes) <= 1
    for f0, f1 in ranges:
        s0, s1 = sharding.shard_samples(f0, f1, 16384, 8192)
        assert (s1 - s0) == (0 if f1 == f0 else (f1 - f0 - 1) * 8192 + 16384)


def test_combine_helpers():
    a = np.array([1.0, np.nan, 3.0])
    b = np.array([2.0, 5.0, np.nan])
    assert np.array_equal(sharding.combine_hold([a, b], "max"), [2.0, 5.0, 3.0])
    assert np.array_equal(sharding.combine_hold([a, b], "min"), [1.0, 5.0, 3.0])
    m, c = sharding.combine_welch([np.array([1.0, 2.0]), np.array([4.0, 8.0])], [1, 3])
    assert c == 4 and np.allclose(m, [3.25, 6.5])
    with pytest.raises(ValueError):
        sharding.combine_welch([np.zeros(2)], [0])


def test_welch_peer_slab_reports_failure_instead_of_raising():
    """sharding.WelchPeerSlab on a box where the device buffers cannot be had (no GPU here): allocate() records the
    failure, connect() turns it into ok == False on every rank - bench.py then takes the host-memory slab."""
    from topdogspectrumanalyser_amd.sharding import WelchPeerSlab
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the buffers can be had (tests/test_gpu_parity.py covers that side)")
    slab = WelchPeerSlab(None, 1, 0, 1024, device=0)
    try:
        slab.allocate()
        slab.connect()
        assert slab.ok is False
    finally:
        name = slab.name
        slab.close()
    assert not os.path.exists(name)
