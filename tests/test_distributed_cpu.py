"""N>1 path on CPU: two gloo ranks each sample their chain shard (with the oracle standing in for the GPU
engine — same chain-id keyed RNG contract) and the trace gather reproduces the single-process job."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, num_chains, out_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    import oracle
    from nutpie_amd.distributed import chain_moments, gather_arrays, shard_chains

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        offset, n_local = shard_chains(num_chains, world, rank)
        s = oracle.default_settings(seed=77, num_chains=n_local, chain_offset=offset, num_tune=40, num_draws=30)
        tr = oracle.sample_tridiag(s, np.linspace(0.5, 2.0, 6))
        local = {"draws": tr.draws[:, ::2], "n_steps": tr.stats["n_steps"], "diverging": tr.stats["diverging"]}
        local["draw_mean"], local["draw_var"] = chain_moments(torch.from_numpy(tr.draws), 40)
        got = gather_arrays(local, n_local)
        # value = total leapfrogs over all ranks (what bench.py aggregates) via an all-reduce
        tot = torch.tensor([float(tr.stats["n_steps"].sum())], dtype=torch.float64)
        dist.all_reduce(tot)
        if rank == 0:
            np.savez(out_path, total=tot.numpy(), **{k: v.numpy() for k, v in got.items()})
        else:
            assert got is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("num_chains", [4, 5])
def test_two_rank_sharding_and_gather(tmp_path, num_chains):
    import oracle

    out = str(tmp_path / "gathered.npz")
    mp.spawn(_worker, args=(2, _free_port(), num_chains, out), nprocs=2, join=True)
    got = np.load(out)
    s = oracle.default_settings(seed=77, num_chains=num_chains, num_tune=40, num_draws=30)
    full = oracle.sample_tridiag(s, np.linspace(0.5, 2.0, 6))
    assert np.array_equal(got["draws"], full.draws[:, ::2])
    assert np.array_equal(got["n_steps"], full.stats["n_steps"])
    assert np.array_equal(got["diverging"], full.stats["diverging"])
    assert got["total"][0] == full.stats["n_steps"].sum()
    np.testing.assert_allclose(got["draw_mean"], full.draws[:, 40:].mean(1), rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(got["draw_var"], full.draws[:, 40:].var(1, ddof=1), rtol=1e-10)


class _OracleSampler:
    """Stands in for PySampler in the CPU test of sample_sharded: the oracle's trace behind the same accessors."""

    def __init__(self, trace, n_local, T, D):
        self.tr, self.n, self.total_draws, self.dim = trace, n_local, T, D

    def wait(self):
        pass

    def device_ptr(self, name):
        return 1

    def _copy(self, name, dtype, vec=False):
        a = self.tr.draws if name == "draws" else self.tr.stats[name]
        return np.ascontiguousarray(a).astype(dtype)


def _sharded_worker(rank, world, port, num_chains, out_path):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    import oracle
    from nutpie_amd.distributed import sample_sharded

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        def make(offset, n_local):
            assert n_local > 0, "ranks without chains must not create a sampler"
            s = oracle.default_settings(seed=5, num_chains=n_local, chain_offset=offset, num_tune=20, num_draws=10)
            return _OracleSampler(oracle.sample_tridiag(s, np.ones(4)), n_local, 30, 4)

        smp, got = sample_sharded(make, num_chains, stats=("n_steps", "diverging"), moments_after=20)
        assert (smp is None) == (rank >= num_chains)
        if rank == 0:
            np.savez(out_path, **{k: v.numpy() for k, v in got.items()})
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_more_ranks_than_chains(tmp_path):
    """world_size 3, two chains: the third rank owns no chain, creates no sampler (n_local_chains = 0 would mean "all" to
    the engine) and still takes part in the gather."""
    import oracle

    out = str(tmp_path / "g.npz")
    mp.spawn(_sharded_worker, args=(3, _free_port(), 2, out), nprocs=3, join=True)
    got = np.load(out)
    full = oracle.sample_tridiag(oracle.default_settings(seed=5, num_chains=2, num_tune=20, num_draws=10), np.ones(4))
    assert np.array_equal(got["draws"], full.draws) and np.array_equal(got["n_steps"], full.stats["n_steps"])
    assert got["draw_mean"].shape == (2, 4)


def test_eight_rank_layout_of_config_5(tmp_path):
    """BASELINE.json config 5's layout at toy size: 8 ranks, contiguous blocks of chains (8192 = 8 x 1024 there, 24 = 8 x 3 here),
    RNG streams keyed by the GLOBAL chain id, one gather of the thinned draws + statistics to rank 0, leapfrogs summed by an
    all-reduce: the result is the single-process job, bit for bit."""
    import oracle

    out = str(tmp_path / "gathered8.npz")
    mp.spawn(_worker, args=(8, _free_port(), 24, out), nprocs=8, join=True)
    got = np.load(out)
    full = oracle.sample_tridiag(oracle.default_settings(seed=77, num_chains=24, num_tune=40, num_draws=30), np.linspace(0.5, 2.0, 6))
    assert np.array_equal(got["draws"], full.draws[:, ::2])
    assert np.array_equal(got["n_steps"], full.stats["n_steps"]) and np.array_equal(got["diverging"], full.stats["diverging"])
    assert got["total"][0] == full.stats["n_steps"].sum()


def _ragged_gather_worker(rank, world, port, out_path):
    import torch
    import torch.distributed as dist

    from nutpie_amd.distributed import gather_arrays, shard_chains

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    total = 11                                            # 11 chains over 3 ranks: 4 + 4 + 3 (ragged: padded for the collective, closed up on the root)
    off, n = shard_chains(total, world, rank)
    local = {"a": torch.arange(off, off + n, dtype=torch.float64)[:, None].repeat(1, 5), "b": torch.arange(off, off + n, dtype=torch.int64)}
    stats = {}
    g = gather_arrays(local, n, stats=stats)
    if rank == 0:
        ok = bool((g["a"][:, 0] == torch.arange(total, dtype=torch.float64)).all() and (g["b"] == torch.arange(total)).all() and g["a"].shape == (total, 5))
        nmax = 4
        ok = ok and stats["root_bytes_allocated"] == world * nmax * (5 * 8 + 8) and stats["payload_bytes"] == total * (5 * 8 + 8)
        open(out_path, "w").write("ok" if ok else f"bad {g} {stats}")
    dist.destroy_process_group()


def test_gather_into_one_buffer_handles_ragged_shards(tmp_path):
    import socket

    import torch.multiprocessing as mp

    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    out = str(tmp_path / "r.txt")
    mp.spawn(_ragged_gather_worker, args=(3, port, out), nprocs=3, join=True)
    assert open(out).read() == "ok"
