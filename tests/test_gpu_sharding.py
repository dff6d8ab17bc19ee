"""GPU: the RCCL code paths of the replica sharding, executed on hardware with
ONE rank (a one-GPU box cannot do more; world size 2 is covered on gloo by
tests/test_sharding.py, 8 GPUs by the driver's scaling run).

  * sharding.spmm with every gather mode, the exchange forced through the
    communicator (all_gather_into_tensor of one block / grouped send+recv to
    self) -- results must equal the plain op bit for bit;
  * bench.py under BENCH_FORCE_DIST=1: the one JSON line must carry the three
    figures SURVEY.md 8e asks for (compute only, + all-gather, overlapped) for
    both transports, with bytes per rank and GB/s per link.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from helpers import make_csr

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture(scope="module")
def one_rank_rccl():
    import torch.distributed as dist
    assert torch.cuda.is_available()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dev = torch.device("cuda:0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    os.environ["SPUTNIK_SHARDING_FORCE_COLLECTIVE"] = "1"
    try:
        yield dev
    finally:
        os.environ.pop("SPUTNIK_SHARDING_FORCE_COLLECTIVE", None)
        dist.destroy_process_group()


@pytest.mark.parametrize("mode,chunks", [("collective", 1), ("p2p", 1), ("collective", 2)])
def test_gather_modes_through_rccl(one_rank_rccl, mode, chunks):
    from torch_sputnik_amd import ops, sharding
    dev = one_rank_rccl
    m, k, n, replicas = 256, 192, 128, 4
    _, vals, ri, ro, ci = make_csr(m, k, 0.8, seed=5)
    rng = np.random.default_rng(6)
    v = torch.from_numpy(rng.uniform(-1, 1, (replicas, len(vals))).astype(np.float32)).to(dev)
    b = torch.from_numpy(rng.uniform(-1, 1, (replicas, k, n)).astype(np.float32)).to(dev)
    topo = [torch.from_numpy(x).to(dev) for x in (ri, ro, ci)]
    want = ops.spmm(m, k, v, *topo, b)
    got = sharding.spmm(m, k, v, *topo, b, gather_mode=mode, overlap_chunks=chunks,
                        local_operands=True)
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    sd = sharding.sddmm(m, k, *topo, want[..., :64].contiguous(), b[..., :64].contiguous(),
                        gather_mode=mode)
    assert torch.equal(sd, ops.sddmm(m, k, *topo, want[..., :64].contiguous(),
                                     b[..., :64].contiguous()))


def test_kept_exchange_writes_in_place_through_rccl(one_rank_rccl):
    """A kept `ReplicaExchange` (what bench.py --gpus N runs on): the kernels write straight
    into its local block through the C ABI -- no output tensor, no copy -- and every
    schedule gathers on the same buffers, step after step."""
    from torch_sputnik_amd import ops, sharding
    dev = one_rank_rccl
    m, k, n, replicas = 256, 192, 128, 4
    _, vals, ri, ro, ci = make_csr(m, k, 0.8, seed=7)
    rng = np.random.default_rng(8)
    v = torch.from_numpy(rng.uniform(-1, 1, (replicas, len(vals))).astype(np.float32)).to(dev)
    b = torch.from_numpy(rng.uniform(-1, 1, (replicas, k, n)).astype(np.float32)).to(dev)
    topo = [torch.from_numpy(x).to(dev) for x in (ri, ro, ci)]
    ex = sharding.make_exchange(replicas, (m, n), b)
    flat_ptr = ex.flat.data_ptr()
    for step in range(2):
        scale = float(step + 1)
        want = ops.spmm(m, k, v * scale, *topo, b)
        for mode, chunks in (("collective", 1), ("p2p", 1), ("p2p", 4), ("collective", 2)):
            before = torch.cuda.memory_allocated(dev)
            got = sharding.spmm(m, k, v * scale, *topo, b, gather_mode=mode, overlap_chunks=chunks,
                                local_operands=True, exchange=ex)
            torch.cuda.synchronize()
            assert torch.equal(got, want), (mode, chunks)
            if not (mode == "collective" and chunks > 1):   # (that layout is re-ordered by one copy)
                assert got.data_ptr() == flat_ptr
            del got
    assert ex.flat.data_ptr() == flat_ptr


def test_bench_line_carries_the_three_multi_gpu_figures():
    env = dict(os.environ, BENCH_FORCE_DIST="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("SPUTNIK_SHARDING_FORCE_COLLECTIVE", None)
    proc = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1",
                           "--steps", "3", "--warmup", "1", "--replicas-per-gpu", "4",
                           "--device-warmup-s", "0.05"],
                          env=env, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, proc.stdout
    line = json.loads(lines[0])
    assert line["steps"] == 3 and line["scaling"] == "weak" and line["value"] > 0
    assert line["headline_schedule"].startswith("allgather")
    assert line["compute_only"]["ms_per_step"] > 0
    for transport in ("collective", "p2p"):
        assert line["allgather"][transport]["ms_per_step"] >= line["compute_only"]["ms_per_step"] * 0.9
        assert line["allgather_overlapped"][transport]["ms_per_step"] > 0
        assert line["exchange"]["exchange_only"][transport]["gbs_per_link_per_direction"] > 0
    assert line["exchange"]["bytes_sent_per_rank_per_peer"] == 4 * 4096 * 4096 * 4
    assert line["roofline"]["frac"] > 0


def test_bench_watchdog_prints_what_was_measured():
    """A schedule that never returns: rank 0 still prints the line, with the
    schedules measured before it and a note -- and the run FAILS (non-zero exit on
    every rank): a hung transport must not be reported as a success."""
    env = dict(os.environ, BENCH_FORCE_DIST="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY="0", BENCH_TEST_HANG="allgather_p2p")
    proc = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "3", "--warmup", "1",
                           "--replicas-per-gpu", "2", "--device-warmup-s", "0.05",
                           "--schedule-timeout-s", "5"],
                          env=env, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 3, (proc.returncode, proc.stderr[-2000:])
    line = json.loads([ln for ln in proc.stdout.splitlines() if ln.strip()][-1])
    assert "allgather_p2p did not finish" in line["watchdog"]
    assert line["headline_schedule"].startswith("allgather") and line["value"] > 0
    assert line["allgather"]["collective"]["ms_per_step"] > 0 and line["allgather"]["p2p"] is None
