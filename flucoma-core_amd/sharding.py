"""Corpus sharding across ranks: one process per GPU, buffers are independent jobs
(clients/nrt/NMFClient.hpp:233 loop body has no cross-buffer state), so the corpus shards with no
data-path collective; the only exchange is the final gather of dictionaries / activations.

Pure index arithmetic + a torch.distributed gather; no device code here.  Covered by the
world_size-2 gloo test (tests/test_sharding.py); bench.py uses the same functions over RCCL.
"""
from __future__ import annotations


def shard_range(n_items: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous block partition, remainder spread over the first ranks: [begin, end)."""
    base, rem = divmod(n_items, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def balanced_assignment(costs, world: int):
    """Greedy longest-processing-time deal for ragged corpora (cost ~ T*F*K per buffer):
    returns a list of index lists, one per rank."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    loads = [0.0] * world
    out = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda j: (loads[j], j))
        out[r].append(i)
        loads[r] += costs[i]
    for lst in out:
        lst.sort()
    return out


def gather_buffer(local, world: int):
    """The receive side of gather_results, allocated ONCE by a caller that gathers every step (bench.py): 31 MB per rank
    for the config-4 shard, i.e. a world x 31 MB allocation that has no place inside a timed step."""
    import torch
    return torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)


def gather_results(local, dist_module, world: int, out=None):
    """All-gather equal-shape per-rank result tensors so that every rank holds the whole corpus'
    dictionaries / activations in global buffer order (rank-major == contiguous-block order).
    `local` is [B_local, ...]; returns [world * B_local, ...] (`out` when the caller supplies the buffer)."""
    if world == 1 and not (dist_module is not None and dist_module.is_initialized()):
        return local
    # with an initialised process group the collective runs for ANY world size, also a one-rank group: bench.py with
    # FLUHIP_BENCH_BACKEND set and the 1-GPU RCCL test execute exactly what the N > 1 job executes
    if out is None:
        out = gather_buffer(local, world)
    assert out.shape[0] == world * local.shape[0] and out.dtype == local.dtype and out.device == local.device
    dist_module.all_gather_into_tensor(out, local.contiguous())
    return out
