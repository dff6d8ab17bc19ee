"""CPU: replica sharding over 2 and 4 processes (gloo).  Each rank computes its block
with the ops (oracle-backed on CPU) and the blocks are all-gathered; every mode
must reproduce the single-process result bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import make_csr


def test_local_range_partitions_everything():
    from torch_sputnik_amd.sharding import local_range
    for replicas in (0, 1, 5, 16, 128, 131):
        for world in (1, 2, 3, 8):
            ranges = [local_range(replicas, world, r) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == replicas
            assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
            sizes = [b - a for a, b in ranges]
            assert max(sizes) - min(sizes) <= 1
    assert local_range(128, 8, 3) == (48, 64)     # config 4: 16 replicas per GPU


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, replicas, results):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import torch_cpu_backend
        torch_cpu_backend.install()
        from torch_sputnik_amd import ops, sharding

        m, k, n = 14, 10, 6
        _, vals, ri, ro, ci = make_csr(m, k, 0.6, seed=1)
        rng = np.random.default_rng(2)
        v = torch.from_numpy(rng.uniform(-1, 1, (replicas, len(vals))).astype(np.float32))
        b = torch.from_numpy(rng.uniform(-1, 1, (replicas, k, n)).astype(np.float32))
        topo = [torch.from_numpy(x) for x in (ri, ro, ci)]
        full = ops.spmm(m, k, v, *topo, b).reshape(replicas, m, n)

        out = {}
        out["collective"] = sharding.spmm(m, k, v, *topo, b)
        out["p2p"] = sharding.spmm(m, k, v, *topo, b, gather_mode="p2p")
        if replicas % world == 0:
            out["overlap"] = sharding.spmm(m, k, v, *topo, b, overlap_chunks=2)
        local = sharding.spmm(m, k, v, *topo, b, gather_output=False)
        a, z = sharding.local_range(replicas, world, rank)
        ok = all(torch.equal(t, full) for t in out.values()) and torch.equal(local, full[a:z])

        # shard-at-origin: every rank passes ONLY its own block of the operands
        for mode in ("collective", "p2p"):
            got = sharding.spmm(m, k, v[a:z].contiguous(), *topo, b[a:z].contiguous(),
                                gather_mode=mode, local_operands=True, replicas=replicas)
            ok = ok and torch.equal(got, full)
        mine = sharding.spmm(m, k, v[a:z].contiguous(), *topo, b[a:z].contiguous(),
                             gather_output=False, local_operands=True, replicas=replicas)
        ok = ok and torch.equal(mine, full[a:z])
        if replicas % world == 0:
            got = sharding.spmm(m, k, v[a:z].contiguous(), *topo, b[a:z].contiguous(),
                                overlap_chunks=2, local_operands=True)
            ok = ok and torch.equal(got, full)

        left = sharding.left_spmm(m, k, v[0].contiguous(), *topo, b)
        ok = ok and torch.equal(left, ops.left_spmm(m, k, v[0].contiguous(), *topo, b))
        lhs = torch.from_numpy(rng.uniform(-1, 1, (replicas, m, 4)).astype(np.float32))
        rhs = torch.from_numpy(rng.uniform(-1, 1, (replicas, k, 4)).astype(np.float32))
        sd = sharding.sddmm(m, k, *topo, lhs, rhs)
        ok = ok and torch.equal(sd, ops.sddmm(m, k, *topo, lhs, rhs).reshape(replicas, -1))
        sm = sharding.sparse_softmax(sd, *topo)
        ok = ok and torch.equal(sm, ops.sparse_softmax(sd, *topo))
        q = torch.from_numpy(rng.uniform(-1, 1, (replicas, m, 8)).astype(np.float32))
        kv = torch.from_numpy(rng.uniform(-1, 1, (replicas, k, 8)).astype(np.float32))
        att = sharding.sparse_attention(q, kv, kv, *topo, 0.35)
        ok = ok and torch.equal(att, ops.sparse_attention(q, kv, kv, *topo, 0.35))
        results[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("replicas", [4, 5])
def test_two_rank_sharding_gloo(replicas):
    world = 2
    ctx = mp.get_context("spawn")
    results = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, replicas, results)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0, f"rank exited with {p.exitcode}"
    assert dict(results) == {0: True, 1: True}


def _worker_c4_shape(rank, world, port, replicas, results):
    """Config 4's partitioning in small: `replicas` / `world` replicas per rank, every
    rank holds ONLY its own block of the operands (shard at origin), and the whole C is
    exchanged -- collective, peer to peer, and overlapped with 8 and 16 chunks per rank
    (the schedules bench.py times on the GPUs)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import torch_cpu_backend
        torch_cpu_backend.install()
        from torch_sputnik_amd import ops, sharding

        m, k, n = 12, 9, 5
        _, vals, ri, ro, ci = make_csr(m, k, 0.6, seed=3)
        rng = np.random.default_rng(4)      # same stream on every rank: the global operands
        v = torch.from_numpy(rng.uniform(-1, 1, (replicas, len(vals))).astype(np.float32))
        b = torch.from_numpy(rng.uniform(-1, 1, (replicas, k, n)).astype(np.float32))
        topo = [torch.from_numpy(x) for x in (ri, ro, ci)]
        full = ops.spmm(m, k, v, *topo, b).reshape(replicas, m, n)   # one process, all replicas
        a, z = sharding.local_range(replicas, world, rank)
        assert z - a == replicas // world
        mine_v, mine_b = v[a:z].contiguous(), b[a:z].contiguous()
        ok = True
        for mode in ("collective", "p2p"):
            got = sharding.spmm(m, k, mine_v, *topo, mine_b, gather_mode=mode, local_operands=True)
            ok = ok and torch.equal(got, full)
        for chunks in (8, 16):
            for mode in ("collective", "p2p"):
                got = sharding.spmm(m, k, mine_v, *topo, mine_b, gather_mode=mode,
                                    overlap_chunks=chunks, local_operands=True)
                ok = ok and torch.equal(got, full)
        # a KEPT exchange (what bench.py --gpus N runs on): the same buffers step after step,
        # every schedule, other operands in between
        ex = sharding.make_exchange(z - a, (m, n), mine_b)
        flat_ptr = ex.flat.data_ptr()
        for step in range(3):
            scale = float(step + 1)
            for mode, chunks in (("collective", 1), ("p2p", 1), ("p2p", 8), ("collective", 16)):
                got = sharding.spmm(m, k, mine_v * scale, *topo, mine_b, gather_mode=mode,
                                    overlap_chunks=chunks, local_operands=True, exchange=ex)
                ok = ok and torch.equal(got, ops.spmm(m, k, v * scale, *topo, b).reshape(replicas, m, n))
        ok = ok and ex.flat.data_ptr() == flat_ptr
        results[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_four_rank_sharding_gloo_local_operands_overlapped():
    """4 ranks, 16 replicas each, rank-local operands, chunked
    overlap with 8 and 16 chunks, both transports -- bit-identical to one process."""
    world, replicas = 4, 64
    ctx = mp.get_context("spawn")
    results = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker_c4_shape, args=(r, world, port, replicas, results))
             for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0, f"rank exited with {p.exitcode}"
    assert dict(results) == {r: True for r in range(world)}
