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
