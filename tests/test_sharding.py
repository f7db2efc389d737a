"""N > 1 path on CPU: world_size-2 gloo processes shard a corpus, each runs its shard with the
oracle standing in for the device (the arithmetic under test is the sharding / gather plumbing,
not the kernels), and the gathered result equals the single-process result in global order."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_covers_everything_once():
    import sharding
    for n in (0, 1, 7, 128, 1024, 1025):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                b, e = sharding.shard_range(n, world, r)
                assert 0 <= b <= e <= n
                seen += list(range(b, e))
            assert seen == list(range(n))
            sizes = [sharding.shard_range(n, world, r) for r in range(world)]
            assert max(e - b for b, e in sizes) - min(e - b for b, e in sizes) <= 1
    assert sharding.shard_range(1024, 8, 3) == (384, 512)   # BASELINE config 4: 128 buffers per GPU


def test_balanced_assignment_for_ragged_corpora():
    import sharding
    rs = np.random.RandomState(0)
    costs = list(rs.randint(1, 100, 57).astype(float))
    parts = sharding.balanced_assignment(costs, 8)
    assert sorted(i for p in parts for i in p) == list(range(57))
    loads = [sum(costs[i] for i in p) for p in parts]
    assert max(loads) - min(loads) <= max(costs)


def test_c_abi_dealing_equals_the_python_one(fluhip_lib_path):
    """fluhip_shard_range / fluhip_balanced_assignment (what a C++ host deals its corpus with) against sharding.py"""
    import ctypes
    import fluhip
    import sharding
    lib = fluhip.load_library(fluhip_lib_path)
    b, e = ctypes.c_int64(), ctypes.c_int64()
    for n in (0, 1, 7, 128, 1024, 1025):
        for world in (1, 2, 3, 8):
            for r in range(world):
                lib.fluhip_shard_range(n, world, r, ctypes.byref(b), ctypes.byref(e))
                assert (b.value, e.value) == sharding.shard_range(n, world, r)
    rs = np.random.RandomState(3)
    for n, world in ((57, 8), (5, 2), (1, 4), (200, 3)):
        costs = np.ascontiguousarray(rs.randint(1, 50, n).astype(np.float64))   # ties on purpose
        out = (ctypes.c_int32 * n)()
        assert lib.fluhip_balanced_assignment(costs.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), n, world, out) == 0
        parts = sharding.balanced_assignment(list(costs), world)
        want = np.empty(n, dtype=np.int32)
        for r, items in enumerate(parts):
            want[items] = r
        assert list(out) == list(want)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "flucoma-core_amd"))
    import torch
    import torch.distributed as dist
    import oracle_np
    import sharding
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_buf, n, win, fft, hop, K, iters = 6, 4096, 256, 256, 64, 3, 5
    b, e = sharding.shard_range(n_buf, world, rank)
    bases, acts = [], []
    for g in range(b, e):  # audio seed 1000 + global index, NMF seed 42
        x = oracle_np.synth_audio(n, 1000 + g)
        bb, aa, *_ = oracle_np.bufnmf_channel(x, win, fft, hop, K, iters, 42)
        bases.append(bb)
        acts.append(aa)
    lb, la = torch.from_numpy(np.stack(bases)), torch.from_numpy(np.stack(acts))
    gb = sharding.gather_results(lb, dist, world)
    # the form bench.py uses: the receive buffer allocated once, the collective run every step into it
    buf = sharding.gather_buffer(la, world)
    for _ in range(2):
        ga = sharding.gather_results(la, dist, world, out=buf)
    assert ga.data_ptr() == buf.data_ptr()
    dist.barrier()
    q.put((rank, gb.numpy(), ga.numpy()))
    dist.destroy_process_group()


def test_two_rank_gloo_shard_and_gather(onp):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n_buf, n, win, fft, hop, K, iters = 6, 4096, 256, 256, 64, 3, 5
    ref_b, ref_a = [], []
    for g in range(n_buf):
        bb, aa, *_ = onp.bufnmf_channel(onp.synth_audio(n, 1000 + g), win, fft, hop, K, iters, 42)
        ref_b.append(bb)
        ref_a.append(aa)
    ref_b, ref_a = np.stack(ref_b), np.stack(ref_a)
    for rank, gb, ga in results:
        assert np.array_equal(gb, ref_b) and np.array_equal(ga, ref_a)  # every rank holds the whole corpus
