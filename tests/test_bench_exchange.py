"""CPU: bench.py's multi-GPU exchange schedules (the all-gather of C)
over 2 and 3 gloo ranks on small CPU tensors: every schedule must leave every
rank's block where the schedule's layout puts it (the class's own verify(),
which compares with fingerprints the ranks exchange separately), including after
a schedule with a different layout has used the same buffer.  The 8-GPU run is
the driver's; this is the part of it that can be wrong without a GPU."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class _Problem:
    """Stand-in for bench.SpmmProblem: `replicas` output matrices that each step
    (re)writes with values only this rank and step produce."""

    def __init__(self, rank, replicas, m, n):
        self.replicas, self.rank, self.flops = replicas, rank, 1.0e9
        self.out = torch.zeros(replicas, m, n)
        self.calls = 0

    def step_range(self, a, b):
        self.calls += 1
        base = torch.arange(a, b, dtype=torch.float32).view(-1, 1, 1)
        self.out[a:b] = (self.rank * 1000 + base * 10 + self.calls % 7 +
                         torch.arange(self.out.shape[-1], dtype=torch.float32) * 0.001)

    def step(self):
        self.step_range(0, self.replicas)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, replicas, chunks, results):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        problem = _Problem(rank, replicas, 6, 5)
        ex = bench.Exchange(problem, world, rank, torch.device("cpu"), chunks)
        schedules = {
            "allgather_collective": lambda: (problem.step(), ex.collective()),
            "allgather_overlapped_collective": lambda: ex.overlapped(ex.collective_chunk),
            "allgather_p2p": lambda: (problem.step(), ex.p2p()),
            "allgather_overlapped_p2p": lambda: ex.overlapped(ex.p2p_chunk),
        }
        for name, fn in list(schedules.items()) + list(schedules.items())[::-1]:
            ex.poison()
            for _ in range(2):
                fn()
            ex.verify(name)
        # the chunk-count sweep of the benchmark line: the layout is switched between
        # schedules on the same buffer (16 chunks = one replica each when there are 16)
        for c in (4, 16, 2):
            got = ex.set_chunks(c)
            assert got == len(ex.bounds) <= min(c, replicas) and ex.bounds[-1][1] == replicas
            for name, chunk_fn in ((f"allgather_overlapped_collective_c{c}", ex.collective_chunk),
                                   (f"allgather_overlapped_p2p_c{c}", ex.p2p_chunk)):
                ex.poison()
                ex.overlapped(chunk_fn)
                ex.verify(name)
        ex.set_chunks(chunks)
        # the global replica order of the rank-major layout
        problem.step()
        ex.collective()
        want = torch.arange(world, dtype=torch.float32).view(-1, 1) * 1000
        got = ex.rank_major[:, :, 0, 0] - ex.rank_major[:, :, 0, 0] % 10
        assert torch.equal(got - torch.arange(replicas, dtype=torch.float32) * 10, want.expand(-1, replicas))
        report = ex.report({"compute_only": 1.0, "allgather_collective": 2.0,
                            "allgather_overlapped_p2p_c4": 1.5, "allgather_overlapped_p2p_c16": 1.25,
                            "allgather_overlapped_collective_c4": 1.75}, world)
        assert report["bytes_received_per_rank"] == replicas * 6 * 5 * 4.0 * (world - 1)
        assert report["allgather_overlapped"]["chunks"] == {"p2p": 16, "collective": 4}
        assert report["allgather_overlapped"]["p2p"]["ms_per_step"] == 1.25
        assert set(report["allgather_overlapped"]["by_chunks"]) == {"4", "16"}
        assert report["per_gpu_share"]["replicas"] == replicas
        floors = report["xgmi_model"]["allgather_overlapped_floor_ms"]
        assert floors["16"] <= floors["4"]
        results[rank] = True
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,replicas,chunks", [(2, 4, 2), (3, 5, 4), (2, 3, 1), (2, 16, 16), (3, 16, 8)])
def test_exchange_schedules_over_gloo(world, replicas, chunks):
    ctx = mp.get_context("spawn")
    results = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, replicas, chunks, results))
             for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0, f"rank exited with {p.exitcode}"
    assert dict(results) == {r: True for r in range(world)}
