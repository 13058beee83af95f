"""Chain sharding across GPUs: one process per GPU, no collective while sampling.

Chains are independent in the reference (each ``ChainProgress`` carries its own step size,
``src/wrapper.rs:90-93``; the mass matrix is per chain, ``docs/sample-stats.qmd:67-83``) and every
random number in the engine is keyed by the GLOBAL chain id (``include/nphip_spec.h``), so a
chain's draws do not depend on how many GPUs share the job.  The only communication is one
gather of the (optionally thinned) trace to rank 0 at the end — ``torch.distributed.gather``,
which RCCL executes as direct sends to the root and therefore uses all of the root's inbound
xGMI links at once instead of a per-link-bound ring.
"""

from __future__ import annotations

import numpy as np


def shard_chains(num_chains: int, world_size: int, rank: int) -> tuple[int, int]:
    """Contiguous block partition: returns (chain_offset, n_local)."""
    base, rem = divmod(int(num_chains), int(world_size))
    n_local = base + (1 if rank < rem else 0)
    offset = rank * base + min(rank, rem)
    return offset, n_local


def device_tensor(ptr: int, shape, dtype="float64", device=0):
    """Zero-copy torch view of an engine-owned device buffer (``PySampler.device_ptr``)."""
    import torch

    typestr = {"float64": "<f8", "int64": "<i8", "uint8": "|u1"}[dtype]

    class _Arr:
        __cuda_array_interface__ = {"shape": tuple(int(s) for s in shape), "typestr": typestr, "data": (int(ptr), False), "version": 2, "strides": None}

    return torch.as_tensor(_Arr(), device=torch.device("cuda", device))


def gather_arrays(local: dict, n_local: int, group=None, dst: int = 0, stats: dict | None = None):
    """Gather per-rank arrays whose leading axis is the local chain axis.

    ``local``: name -> torch.Tensor (CPU for gloo, GPU for nccl/RCCL) or numpy array, leading dim
    ``n_local``.  Returns name -> tensor of all chains (global chain order) on ``dst``, None elsewhere.

    The root receives every array into ONE pre-allocated buffer ``[world * nmax, ...]`` — the receive list handed to ``gather`` are
    views of it — and returns (a view of) that buffer: root memory is 1 x the payload, not ``world`` receive buffers plus their
    concatenation (config 5 at 8 ranks: 12 GB of thinned draws instead of 24).  Ragged shards (``num_chains`` not a multiple of
    the world size) are padded to ``nmax`` rows for the collective and closed up in place.  ``stats`` (optional dict) receives
    ``root_bytes_allocated`` and ``payload_bytes`` on the root."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    counts = [None] * world
    dist.all_gather_object(counts, int(n_local), group=group)
    nmax = max(counts)
    total = int(sum(counts))
    out = {} if rank == dst else None
    allocated = payload = 0
    for name in sorted(local):
        t = local[name]
        if isinstance(t, np.ndarray):
            t = torch.from_numpy(np.ascontiguousarray(t))
        t = t.contiguous()
        if t.shape[0] != nmax:  # pad ragged shards so every rank sends the same shape
            pad = torch.zeros((nmax - t.shape[0], *t.shape[1:]), dtype=t.dtype, device=t.device)
            t = torch.cat([t, pad], 0)
        if rank == dst:
            big = torch.empty((world * nmax, *t.shape[1:]), dtype=t.dtype, device=t.device)
            allocated += big.numel() * big.element_size()
            dist.gather(t, [big[r * nmax:(r + 1) * nmax] for r in range(world)], dst=dst, group=group)
            if total != world * nmax:   # close the gaps the padding left (shards move towards the front: never over unread rows)
                at = 0
                for r, cnt in enumerate(counts):
                    if at != r * nmax and cnt:
                        big[at:at + cnt] = big[r * nmax:r * nmax + cnt].clone()
                    at += cnt
            out[name] = big[:total]
            payload += total * (big[0].numel() if big.shape[0] else 0) * big.element_size()
        else:
            dist.gather(t, None, dst=dst, group=group)
    if stats is not None and rank == dst:
        stats["root_bytes_allocated"], stats["payload_bytes"] = int(allocated), int(payload)
    return out


def chain_moments(draws, n_tune: int):
    """Per-chain posterior mean and variance of every dimension from ``draws[chain, draw, dim]`` (a torch tensor, on
    whatever device it lives): the on-device summary SURVEY.md §8e asks for instead of shipping a config-5 trace."""
    post = draws[:, n_tune:]
    return post.mean(1), post.var(1, unbiased=True)


def sample_sharded(make_sampler, num_chains: int, *, group=None, thin: int = 1, dims=None, gather_draws: bool = True,
                   stats=("depth", "n_steps", "diverging", "tuning", "step_size", "energy", "logp"), device=None,
                   moments_after: int | None = None, timing: dict | None = None):
    """Run this rank's shard to completion and gather the trace on rank 0.

    ``make_sampler(chain_offset, n_local) -> PySampler``.  Draws are thinned by ``thin`` and restricted to
    ``dims`` ON DEVICE before the gather (a full config-5 trace is 82 GB per GPU).
    ``moments_after=n_tune`` additionally gathers ``draw_mean`` / ``draw_var`` ``[chains, D]`` (per-chain moments of the
    post-warm-up draws, reduced on the device; megabytes instead of the trace).
    Returns ``(sampler, gathered)``; ``gathered`` is a dict of tensors on rank 0 and None elsewhere.
    ``timing`` (optional dict) receives ``sample_s`` and what :func:`gather_trace` reports.
    """
    import time

    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    offset, n_local = shard_chains(num_chains, world, rank)
    if n_local == 0:
        # more ranks than chains: this rank owns nothing (n_local_chains = 0 would mean "all chains" to the engine); it still
        # takes part in the gather with empty shards of the shapes rank 0 announces
        return None, _gather_empty(group)
    sampler = make_sampler(offset, n_local)
    t0 = time.perf_counter()
    sampler.wait()
    if timing is not None:
        timing["sample_s"] = time.perf_counter() - t0
    gathered = gather_trace(sampler, n_local, num_chains, group=group, thin=thin, dims=dims, gather_draws=gather_draws, stats=stats,
                            device=device, moments_after=moments_after, timing=timing)
    return sampler, gathered


def gather_trace(sampler, n_local: int, num_chains: int, *, group=None, thin: int = 1, dims=None, gather_draws: bool = True,
                 stats=("depth", "n_steps", "diverging", "tuning", "step_size", "energy", "logp"), device=None,
                 moments_after: int | None = None, timing: dict | None = None):
    """The collective half of :func:`sample_sharded`: the finished sampler's trace (thinned / reduced on the device) to rank 0 in
    ONE ``gather`` per array.  With ``timing`` the on-device reduction and the collective are timed apart (a barrier in between
    keeps a straggling rank's sampling time out of the collective's figure) and the payload is counted:
    ``reduce_s``, ``gather_s``, ``bytes_local`` (this rank's shard), ``bytes_gathered`` (rank 0: all shards), ``arrays``."""
    import time

    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    T, D = sampler.total_draws, sampler.dim
    on_gpu = dist.get_backend(group) == "nccl"
    dev = device if device is not None else (torch.cuda.current_device() if on_gpu else None)

    def sync():
        if on_gpu:
            torch.cuda.synchronize()

    t0 = time.perf_counter()
    local = {}
    dtypes = {"depth": "int64", "n_steps": "int64", "index_in_trajectory": "int64", "diverging": "uint8", "maxdepth_reached": "uint8", "tuning": "uint8"}
    for name in stats:
        ptr = sampler.device_ptr(name)
        if on_gpu:
            local[name] = device_tensor(ptr, (n_local, T), dtypes.get(name, "float64"), dev)
        else:
            local[name] = sampler._copy(name, np.dtype(dtypes.get(name, "float64")))
    if (gather_draws or moments_after is not None) and sampler.device_ptr("draws"):
        if on_gpu:
            d = device_tensor(sampler.device_ptr("draws"), (n_local, T, D), "float64", dev)
        else:
            d = torch.from_numpy(sampler._copy("draws", np.float64, vec=True))
        if moments_after is not None:
            local["draw_mean"], local["draw_var"] = chain_moments(d, int(moments_after))
        if gather_draws:
            d = d[:, ::thin]
            if dims is not None:
                d = d[:, :, torch.as_tensor(list(dims), device=d.device)]
            local["draws"] = d.contiguous()
    if world > num_chains:   # some ranks are empty: tell them what is being gathered (names, trailing shapes, dtypes)
        spec = [{k: (tuple(v.shape[1:]), str(v.dtype).replace("torch.", "")) for k, v in sorted(local.items())}] if rank == 0 else [None]
        dist.broadcast_object_list(spec, src=0, group=group)
    if timing is not None:
        sync()
        timing["reduce_s"] = time.perf_counter() - t0
        timing["bytes_local"] = int(sum(int(np.prod(v.shape)) * (v.element_size() if hasattr(v, "element_size") else v.itemsize) for v in local.values()))
        timing["arrays"] = {k: list(v.shape) for k, v in sorted(local.items())}
        dist.barrier(group=group)
        sync()
        t0 = time.perf_counter()
    gathered = gather_arrays(local, n_local, group=group, stats=timing)
    if timing is not None:
        sync()
        timing["gather_s"] = time.perf_counter() - t0
        timing["collective_ranks"] = world
        if gathered is not None:
            timing["bytes_gathered"] = int(sum(v.numel() * v.element_size() for v in gathered.values()))
    return gathered


def _gather_empty(group=None):
    import torch
    import torch.distributed as dist

    spec = [None]
    dist.broadcast_object_list(spec, src=0, group=group)
    on_gpu = dist.get_backend(group) == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")
    local = {k: torch.empty((0, *shape), dtype=getattr(torch, dt), device=dev) for k, (shape, dt) in spec[0].items()}
    return gather_arrays(local, 0, group=group)
