#!/usr/bin/env python3
"""Headline benchmark: SpMM effective GFLOP/s (+ algorithmic HBM GB/s) at
M=N=K=4096, fp32, on MI355X -- BASELINE.json's metric.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path (the C-ABI entry sputnik_hip_spmm_batched:
topology pre-pass + LDS-tiled SpMM kernel) over one batch of synthetic input
that is already resident in HBM.

  N = 1  workload "C2": 2-D SpMM 4096^3, density 0.1 (BASELINE.json configs[1];
         the other densities of the sweep are reported in `sweep`).
  N > 1  workload "C4": batched SpMM, 16 replicas of the same problem per GPU
         (128 replicas at 8 GPUs), replica dimension sharded over the ranks
         (operands rank-local: no GPU holds another rank's B), one launch per
         rank per step, THEN the all-gather of C over RCCL/xGMI that the north
         star names, inside the timed step.  Per-GPU work is fixed -> "weak".
         Seven schedules are timed with the same W + K steps each and printed
         in the one JSON line: `compute_only`, `allgather` (collective / p2p),
         `allgather_overlapped` (collective / p2p, chunked on a side stream),
         `exchange` (the all-gather alone: bytes per rank, GB/s per link).
         `value` is the fastest schedule that includes the all-gather.

value = 2*nnz*N*replicas_total / time  (effective GFLOP/s of the whole job).
Rank 0 prints ONE JSON line.  Extra keys: `roofline` (HBM roof, algorithmic
bytes / dominant kernel time), `roofline_valu` (the roof that actually binds
fp32 SpMM at this size), `cpu_baseline`, `sweep`, `other_ops`.
"""
import argparse
import json
import os
import sys
import time

# dmabuf IPC is the only mode the host driver supports: without this RCCL's
# cross-process buffer sharing fails (hipIpcGetMemHandle: invalid argument).
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
VALU_PEAK_TFLOPS = 157.3     # fp32 vector FMA peak (spec); MFMA is excluded by the north star
M = K = N = 4096
DENSITIES = [0.5, 0.25, 0.2, 0.15, 0.1, 0.05]
HEADLINE_DENSITY = 0.1
REPLICAS_PER_GPU_MULTI = 16


def spmm_bytes(nnz, m, k, n, replicas=1):
    """Algorithmic bytes: values + column ids, B once, C once per
    replica; offsets + row_indices once."""
    return replicas * (8.0 * nnz + 4.0 * k * n + 4.0 * m * n) + 4.0 * (2 * m + 1)


def event_time_ms(fn, iters, warmup=3, with_min=False):
    """Median (and minimum) GPU time of one call: one HIP event pair per call on
    torch's current stream (= the stream the C ABI launches on), so host launch
    gaps are not counted."""
    for _ in range(warmup):
        fn()
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(iters)]
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(iters)]
    torch.cuda.synchronize()
    for s, e in zip(starts, ends):
        s.record()
        fn()
        e.record()
    torch.cuda.synchronize()
    times = sorted(s.elapsed_time(e) for s, e in zip(starts, ends))
    return (times[len(times) // 2], times[0]) if with_min else times[len(times) // 2]


class SpmmProblem:
    def __init__(self, dev, density, replicas, seed):
        from torch_sputnik_amd import capi
        from torch_sputnik_amd.synthetic import random_csr, uniform
        self.capi = capi
        self.replicas = replicas
        self.ri, self.ro, self.ci, self.nnz = random_csr(M, K, density, dev, seed=seed)
        vshape = (replicas, self.nnz) if replicas > 1 else (self.nnz,)
        bshape = (replicas, K, N) if replicas > 1 else (K, N)
        self.values = uniform(vshape, dev, seed + 1)
        self.dense = uniform(bshape, dev, seed + 2)
        self.out = torch.empty((replicas, M, N) if replicas > 1 else (M, N), device=dev)
        self.ws = torch.empty(capi.spmm_workspace_bytes(M, K, N, self.nnz) + 16,
                              dtype=torch.uint8, device=dev)
        self.flops = 2.0 * self.nnz * N * replicas
        self.bytes = spmm_bytes(self.nnz, M, K, N, replicas)

    def step(self):
        """The whole hot path: pre-pass + kernel (what the torch op does per call)."""
        self.capi.spmm_batched(M, K, N, self.replicas, self.ri, self.values,
                               self.nnz if self.replicas > 1 else 0, self.ro, self.ci,
                               self.dense, self.out, self.ws)

    def step_range(self, a, b):
        """The hot path for replicas a..b-1 only (pipelined exchange)."""
        vals = self.values[a:b] if self.replicas > 1 else self.values
        self.capi.spmm_batched(M, K, N, b - a, self.ri, vals,
                               self.nnz if self.replicas > 1 else 0, self.ro, self.ci,
                               self.dense.reshape(self.replicas, K, N)[a:b],
                               self.out.reshape(self.replicas, M, N)[a:b], self.ws)

    def kernel_only(self):
        """Dominant kernel alone, on an already planned workspace."""
        self.capi.spmm_batched_planned(M, K, N, self.replicas, self.ri, self.values,
                                       self.nnz if self.replicas > 1 else 0, self.ro, self.ci,
                                       self.dense, self.out, self.ws)


def _exchange_base():
    from torch_sputnik_amd.sharding import ReplicaExchange
    return ReplicaExchange


class Exchange(_exchange_base()):
    """The all-gather of C for the replica-sharded product: the PRODUCT's
    exchange -- torch_sputnik_amd/sharding.py, `ReplicaExchange`: preallocated buffers,
    RCCL's all_gather_into_tensor or direct grouped send / recv to every peer, after the
    launch or chunk by chunk on a side stream -- bound to the benchmark's problem (the
    kernels write straight into its local block).  What this class adds is bookkeeping:
    the poison / verify pair around every timed schedule and the report."""

    def __init__(self, problem, world, rank, dev, chunks):
        r = problem.replicas
        m, n = problem.out.shape[-2:]            # (4096 x 4096 in the benchmark)
        self.problem, self.m, self.n = problem, m, n
        super().__init__(problem.out.reshape(r, m, n), world, rank, chunks,
                         compute_range=problem.step_range)

    def poison(self):
        """Before a schedule runs: the first row of every block of the gathered buffer
        is overwritten, so that verify() cannot pass on what an earlier schedule left."""
        self.flat.view(-1, self.n)[::self.m].fill_(float("nan"))

    def verify(self, schedule):
        """After a schedule's timed steps: EVERY rank's block must sit where the schedule
        puts it.  Checked through a fingerprint of each block (its first row), which
        all ranks exchange separately with a small all_gather."""
        if self.dev.type == "cuda":
            torch.cuda.synchronize()
        if not schedule.startswith("allgather"):
            return
        r, m, n = self.problem.replicas, self.m, self.n
        mine = self.local[:, 0, :].contiguous()                       # [r, n]
        prints = torch.empty((self.world, r, n), device=self.dev)
        self.dist.all_gather_into_tensor(prints.view(-1, n), mine)
        if schedule.startswith("allgather_overlapped_collective"):
            for c, (a, b) in enumerate(self.bounds):
                got = self.chunk_major[c].view(self.world, b - a, m, n)[:, :, 0, :]
                assert torch.equal(got, prints[:, a:b]), f"{schedule}: gathered chunk {c} is wrong"
        else:
            assert torch.equal(self.rank_major[:, :, 0, :], prints), f"{schedule}: gathered blocks are wrong"

    def report(self, timings, n_gpus):
        flops = self.problem.flops * n_gpus

        def line(name):
            ms = timings.get(name)
            return None if not ms else {"ms_per_step": ms, "gflops": flops / ms / 1e6}

        def link(name):
            ms = timings.get(name)
            if not ms:
                return None
            return {"ms": ms, "gbs_per_link_per_direction": self.bytes_per_peer / ms / 1e6,
                    "gbs_received_per_gpu": self.bytes_per_peer * max(1, self.world - 1) / ms / 1e6}

        # the overlapped schedules were timed for several chunk counts
        # ("allgather_overlapped_<transport>_c<chunks>"): the best per transport is the
        # figure, all of them are listed
        by_chunks, best = {}, {}
        for transport in ("collective", "p2p"):
            prefix = f"allgather_overlapped_{transport}_c"
            for name, ms in timings.items():
                if name.startswith(prefix) and ms:
                    c = int(name[len(prefix):])
                    by_chunks.setdefault(str(c), {})[transport] = line(name)
                    if transport not in best or ms < timings[best[transport]]:
                        best[transport] = name
        compute_ms = timings.get("compute_only") or 0.0
        # what the links allow (xGMI full mesh, ~153 GB/s per link and direction; a direct
        # exchange sends one block over each of the 7 links at once)
        floor_ms = self.bytes_per_peer / 153.0e6 if self.world > 1 else 0.0
        chunk_counts = sorted(int(c) for c in by_chunks) or [len(self.bounds)]
        model = {
            "links_per_gpu": 7, "gbs_per_link": 153.0,
            "direct_exchange_floor_ms": floor_ms,
            "allgather_floor_ms": compute_ms + floor_ms,
            # chunks of equal size: the longer of the two streams, plus one chunk of the
            # other (the first chunk's compute / the last chunk's exchange is exposed)
            "allgather_overlapped_floor_ms": {
                str(c): max(compute_ms, floor_ms) + min(compute_ms, floor_ms) / c for c in chunk_counts},
            "ring_collective_floor_ms": floor_ms * max(1, self.world - 1),
        }
        return {
            "compute_only": line("compute_only"),
            "per_gpu_share": {"replicas": self.problem.replicas, "ms": compute_ms or None,
                              "gflops_per_gpu": (self.problem.flops / compute_ms / 1e6) if compute_ms else None,
                              "note": "what ONE GPU does per step in this run; the one-GPU line's "
                                      "per_gpu_share_of_multi_gpu_runs measures the same workload"},
            "allgather": {"collective": line("allgather_collective"), "p2p": line("allgather_p2p")},
            "allgather_overlapped": {"collective": line(best["collective"]) if "collective" in best else None,
                                     "p2p": line(best["p2p"]) if "p2p" in best else None,
                                     "chunks": {t: int(best[t].rsplit("_c", 1)[1]) for t in best},
                                     "by_chunks": by_chunks},
            "exchange_only": {"collective": link("exchange_only_collective"),
                              "p2p": link("exchange_only_p2p")},
            "bytes_sent_per_rank_per_peer": self.bytes_per_peer,
            "bytes_received_per_rank": self.bytes_per_peer * (self.world - 1),
            "gathered_bytes_per_rank": self.bytes_per_peer * self.world,
            "xgmi_model": model,
        }


WATCHDOG_EXIT_CODE = 3


class Watchdog:
    """arm(make_line, fd): unless disarm() comes within `seconds`, the process writes
    make_line() (bytes or None) to fd and exits with WATCHDOG_EXIT_CODE -- from a
    daemon thread, because the main thread of a hung run sits inside a blocking
    runtime call.  A hung schedule is a FAILED run on every rank: the partial line
    (it carries a "watchdog" key and only the schedules that finished) is there for
    diagnosis, the exit code says that the headline is not to be trusted.  Nothing is
    restarted or re-executed from a process that has touched the GPU."""

    def __init__(self, seconds, printer=True):
        import threading
        self.seconds = seconds
        self.printer = printer   # rank 0: the rank whose line is the result
        self._lock = threading.Lock()
        self._deadline = None
        self._make_line = None
        self._fd = None
        threading.Thread(target=self._run, daemon=True).start()

    def arm(self, make_line, fd):
        with self._lock:
            self._deadline, self._make_line, self._fd = time.time() + self.seconds, make_line, fd

    def disarm(self):
        with self._lock:
            self._deadline = None

    def _run(self):
        while True:
            time.sleep(1.0)
            with self._lock:
                expired = self._deadline is not None and time.time() > self._deadline
                make_line, fd = self._make_line, self._fd
            if expired:
                line = None
                try:
                    line = make_line()
                    if line:
                        os.write(fd, line)
                finally:
                    os._exit(WATCHDOG_EXIT_CODE)


def dense_gpu_ms(problem):
    """The reference's README table sets Sputnik beside cuSPARSE and cuBLAS
    (/root/reference README.md:46-55); this is the cuBLAS column on THIS GPU: the
    densified A against B with torch.matmul in float32 (the vendor library, used as a
    yardstick only -- the product never calls it)."""
    try:
        rows = torch.repeat_interleave(torch.arange(M, device=problem.ro.device),
                                       (problem.ro[1:] - problem.ro[:-1]).long())
        a = torch.zeros(M, K, device=problem.ro.device)
        a[rows, problem.ci.long()] = problem.values.reshape(-1)[:problem.nnz]
        b = problem.dense.reshape(-1, K, N)[0]
        return event_time_ms(lambda: torch.matmul(a, b), 20, warmup=5)
    except Exception:  # noqa: BLE001 - an extra column, best effort
        return None


def dense_crossover(sweep):
    """Density at which the sparse kernel and the dense float32 GEMM of the same GPU take
    the same time (linear between the two sweep points around it)."""
    pts = sorted((e["density"], e["ms"], e["dense_gpu_ms"]) for e in sweep if e.get("dense_gpu_ms"))
    if not pts:
        return None
    out = {"dense_fp32_matmul_ms": sorted(p[2] for p in pts)[len(pts) // 2],
           "library": "torch.matmul float32 (hipBLASLt / rocBLAS): yardstick only",
           "crossover_density": None}
    for (d0, s0, g0), (d1, s1, g1) in zip(pts, pts[1:]):
        if (s0 - g0) <= 0 <= (s1 - g1):
            t = (g0 - s0) / ((s1 - g1) - (s0 - g0)) if (s1 - g1) != (s0 - g0) else 0.0
            out["crossover_density"] = d0 + t * (d1 - d0)
    if out["crossover_density"] is None:
        out["note"] = ("sparse faster at every density of the sweep" if pts[-1][1] < pts[-1][2]
                       else "dense faster at every density of the sweep")
    return out


def cpu_baseline(problem):
    """Oracle (C restatement, CSR, OpenMP) timed on the host cores; bounded sample."""
    import numpy as np
    from oracle import c_oracle
    if not c_oracle.available():
        return None
    vals = problem.values.reshape(-1)[:problem.nnz].cpu().numpy()
    ro, ci = problem.ro.cpu().numpy(), problem.ci.cpu().numpy()
    dense = problem.dense.reshape(-1, K, N)[0].cpu().numpy()
    # Threads = the cores this process may run on (the GPU box gives one GPU's
    # share of the host, not all of its cores).
    threads = max(1, min(c_oracle.num_threads(), len(os.sched_getaffinity(0))))
    c_oracle.set_num_threads(threads)
    c_oracle.spmm(M, K, vals, ro, ci, dense, f32_accumulate=True)  # warm-up
    times = []
    t_all = time.perf_counter()
    while len(times) < 5 and time.perf_counter() - t_all < 25.0:
        t0 = time.perf_counter()
        c_oracle.spmm(M, K, vals, ro, ci, dense, f32_accumulate=True)
        times.append(time.perf_counter() - t0)
    med = sorted(times)[len(times) // 2]
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")),
                             cpu_model)
    except OSError:
        pass
    base = {"value": 2.0 * problem.nnz * N / med / 1e9, "unit": "GFLOP/s", "cores": threads,
            "kind": "port", "cpu_model": cpu_model,
            "sample": f"{len(times)} runs of the full C2 d={HEADLINE_DENSITY} SpMM "
                      f"(oracle/csrc/sputnik_oracle.c oracle_spmm_f32, OpenMP over rows), median",
            "ms": med * 1e3}
    # The reference's own "CPU path" is the dense matmul its tests compare with
    # (tests/test_spmm.py:9-10): time that too, same A (zeros included) and B.
    try:
        torch.set_num_threads(threads)
        a = torch.zeros(M, K)
        rows = torch.repeat_interleave(torch.arange(M), torch.from_numpy(np.diff(ro)).long())
        a[rows, torch.from_numpy(ci).long()] = torch.from_numpy(vals)
        b = torch.from_numpy(dense)
        torch.matmul(a, b)
        dt = []
        for _ in range(3):
            t0 = time.perf_counter()
            torch.matmul(a, b)
            dt.append(time.perf_counter() - t0)
        dmed = sorted(dt)[1]
        base["dense_torch_matmul"] = {"ms": dmed * 1e3, "effective_gflops": 2.0 * problem.nnz * N / dmed / 1e9,
                                      "dense_gflops": 2.0 * M * K * N / dmed / 1e9, "threads": threads}
    except Exception as e:  # noqa: BLE001 - baseline is best effort
        base["dense_torch_matmul"] = {"error": str(e)}
    return base


def other_ops(dev):
    """Kernel times of the rest of the path at BASELINE.json configs 3 and 5."""
    from torch_sputnik_amd import capi
    from torch_sputnik_amd.synthetic import random_csr, uniform
    res = {}
    s, d, reps = 1024, 64, 64  # config 3: seq 1024, 8 heads x batch 8, head_dim 64
    ri, ro, ci, nnz = random_csr(s, s, 0.1, dev, seed=7)
    q, kk, v = (uniform((reps, s, d), dev, 11 + i) for i in range(3))
    scores = torch.empty(reps, nnz, device=dev)
    sd_ws = torch.empty(capi.sddmm_workspace_bytes(s, d, s, nnz) + 16, dtype=torch.uint8, device=dev)
    probs = torch.empty_like(scores)
    ctx = torch.empty(reps, s, d, device=dev)
    t = event_time_ms(lambda: capi.sddmm_batched(s, d, s, reps, ri, ro, ci, q, kk, scores, sd_ws), 20)
    by = reps * (8.0 * s * d + 4.0 * nnz) + 4.0 * nnz + 4.0 * (2 * s + 1)
    res["sddmm_c3"] = {"ms": t, "gflops": 2.0 * nnz * d * reps / t / 1e6, "alg_gbs": by / t / 1e6}
    try:   # the same on a planned workspace (static mask: kernel only, no pre-pass launch)
        capi.sddmm_plan(s, d, s, ri, ro, ci, sd_ws)
        tp = event_time_ms(lambda: capi.sddmm_batched_planned(s, d, s, reps, ri, ro, ci, q, kk, scores,
                                                              sd_ws), 20)
        res["sddmm_c3"]["planned_ms"] = tp
        res["sddmm_c3"]["planned_hbm_frac"] = by / tp / 1e6 / HBM_PEAK_GBS
        res["sddmm_c3"]["planned_kernel"] = capi.sddmm_kernel_name(s, d, s, nnz, reps, 4, planned=True)
        # the rhs-stationary quad kernel of round 3 on the same plan (SPUTNIK_HIP_SDDMM_FLAT=0)
        os.environ["SPUTNIK_HIP_SDDMM_FLAT"] = "0"
        capi.reload_options()
        try:
            res["sddmm_c3"]["planned_quad_kernel_ms"] = event_time_ms(
                lambda: capi.sddmm_batched_planned(s, d, s, reps, ri, ro, ci, q, kk, scores, sd_ws), 20)
        finally:
            os.environ.pop("SPUTNIK_HIP_SDDMM_FLAT", None)
            capi.reload_options()
    except Exception as e0:  # noqa: BLE001 - extra metric, best effort
        res["sddmm_c3"]["planned_error"] = str(e0)[:200]
    try:   # half operands read as they are (round 3): LDS slab in half, v_dot2 products
        for name, dt in (("fp16", torch.float16), ("bf16", torch.bfloat16)):
            qh, kh = q.to(dt), kk.to(dt)
            capi.sddmm_plan(s, d, s, ri, ro, ci, sd_ws)
            tp = event_time_ms(lambda: capi.sddmm_typed(s, d, s, reps, ri, ro, ci, qh, kh, scores, sd_ws,
                                                        planned=True), 20)
            sh = torch.empty(reps, nnz, device=dev, dtype=dt)
            th = event_time_ms(lambda: capi.sddmm_typed(s, d, s, reps, ri, ro, ci, qh, kh, sh, sd_ws,
                                                        planned=True), 20)
            byh = reps * (4.0 * s * d + 4.0 * nnz) + 4.0 * nnz + 4.0 * (2 * s + 1)
            res["sddmm_c3_" + name] = {"planned_ms": tp, "gflops": 2.0 * nnz * d * reps / tp / 1e6,
                                       "planned_hbm_frac": byh / tp / 1e6 / HBM_PEAK_GBS,
                                       "half_output_planned_ms": th}
        del qh, kh, sh
    except Exception as e0:  # noqa: BLE001 - extra metric, best effort
        res["sddmm_c3_fp16"] = {"error": str(e0)[:200]}
    t = event_time_ms(lambda: capi.sparse_softmax_batched(s, reps, scores, ri, ro, ci, probs), 50)
    by = reps * 8.0 * nnz + 4.0 * (2 * s + 1)
    res["softmax_c3"] = {"ms": t, "alg_gbs": by / t / 1e6, "hbm_frac": by / t / 1e6 / HBM_PEAK_GBS,
                         "replicas": reps}
    # the same bytes through a device copy: what the memory system gives a perfectly
    # streaming kernel of this size (53.7 MB is a 10 us kernel: the launch ramp counts)
    t_copy = event_time_ms(lambda: probs.copy_(scores), 50)
    res["softmax_c3"]["device_copy_same_bytes_ms"] = t_copy
    res["softmax_c3"]["device_copy_hbm_frac"] = by / t_copy / 1e6 / HBM_PEAK_GBS
    grad = torch.empty_like(scores)
    t = event_time_ms(lambda: capi.sparse_softmax_backward_batched(s, reps, probs, scores, ro, 1.0, grad), 50)
    res["softmax_backward_c3"] = {"ms": t, "alg_gbs": reps * 12.0 * nnz / t / 1e6,
                                  "hbm_frac": reps * 12.0 * nnz / t / 1e6 / HBM_PEAK_GBS}
    try:   # native half storage (round 3): 2 + 2 bytes per entry forward, 6 backward
        for name, dt in (("fp16", torch.float16), ("bf16", torch.bfloat16)):
            sh, ph, gh = scores.to(dt), probs.to(dt), torch.empty(reps, nnz, device=dev, dtype=dt)
            t = event_time_ms(lambda: capi.sparse_softmax_typed(s, reps, sh, ri, ro, ci, 1.0, ph), 50)
            byh = reps * 4.0 * nnz + 4.0 * (2 * s + 1)
            res["softmax_c3_" + name] = {"ms": t, "alg_gbs": byh / t / 1e6,
                                         "hbm_frac": byh / t / 1e6 / HBM_PEAK_GBS}
            t = event_time_ms(lambda: capi.sparse_softmax_backward_typed(s, reps, ph, sh, ro, 1.0, gh), 50)
            res["softmax_backward_c3_" + name] = {"ms": t, "hbm_frac": reps * 6.0 * nnz / t / 1e6 / HBM_PEAK_GBS}
        bigh = uniform((512, nnz), dev, 32).half()
        bigh_out = torch.empty_like(bigh)
        t = event_time_ms(lambda: capi.sparse_softmax_typed(s, 512, bigh, ri, ro, ci, 1.0, bigh_out), 20)
        res["softmax_c3_r512_fp16"] = {"ms": t, "hbm_frac": 512 * 4.0 * nnz / t / 1e6 / HBM_PEAK_GBS}
        t = event_time_ms(lambda: capi.sparse_softmax_backward_typed(s, 512, bigh_out, bigh, ro, 1.0, bigh), 20)
        res["softmax_backward_c3_r512_fp16"] = {"ms": t, "hbm_frac": 512 * 6.0 * nnz / t / 1e6 / HBM_PEAK_GBS}
        del bigh, bigh_out, sh, ph, gh
    except Exception as e:  # noqa: BLE001
        res["softmax_c3_fp16"] = {"error": str(e)[:200]}
    try:   # 512 replicas: the launch ramp amortised
        big = uniform((512, nnz), dev, 31)
        big_out = torch.empty_like(big)
        t = event_time_ms(lambda: capi.sparse_softmax_batched(s, 512, big, ri, ro, ci, big_out), 20)
        res["softmax_c3_r512"] = {"ms": t, "hbm_frac": 512 * 8.0 * nnz / t / 1e6 / HBM_PEAK_GBS}
        t = event_time_ms(lambda: capi.sparse_softmax_backward_batched(s, 512, big_out, big, ro, 1.0, big), 20)
        res["softmax_backward_c3_r512"] = {"ms": t, "hbm_frac": 512 * 12.0 * nnz / t / 1e6 / HBM_PEAK_GBS}
        del big, big_out
    except Exception as e:  # noqa: BLE001
        res["softmax_c3_r512"] = {"error": str(e)[:200]}
    ws3 = torch.empty(capi.spmm_workspace_bytes(s, s, d, nnz) + 16, dtype=torch.uint8, device=dev)
    t = event_time_ms(lambda: capi.spmm_batched(s, s, d, reps, ri, probs, nnz, ro, ci, v, ctx, ws3), 20)
    by = reps * (4.0 * nnz + 8.0 * s * d) + 4.0 * nnz + 4.0 * (2 * s + 1)
    res["spmm_c3"] = {"ms": t, "gflops": 2.0 * nnz * d * reps / t / 1e6, "alg_gbs": by / t / 1e6}
    try:   # half weights and half V, widened on the way into LDS (no pass of their own)
        for name, dt in (("fp16", torch.float16), ("bf16", torch.bfloat16)):
            ph, vh = probs.to(dt), v.to(dt)
            t = event_time_ms(lambda: capi.spmm_typed(s, s, d, reps, ri, ph, nnz, ro, ci, vh, ctx, ws3), 20)
            res["spmm_c3_" + name] = {"ms": t, "gflops": 2.0 * nnz * d * reps / t / 1e6}
        del ph, vh
    except Exception as e0:  # noqa: BLE001 - extra metric, best effort
        res["spmm_c3_fp16"] = {"error": str(e0)[:200]}
    try:
        # config 3's projections (modules/sparse_attention.py:108-126): a 512 x 512 weight at
        # density 0.1 against [512, 1024] x batch 8 (panel-resident kernel), its transposed
        # product through the permutation, and the weight gradient summed over the batch
        e, b3 = 512, 8
        wri, wro, wci, wnnz = random_csr(e, e, 0.1, dev, seed=17)
        wv = uniform((wnnz,), dev, 18)
        xs = uniform((b3, e, s), dev, 19)
        ys = torch.empty(b3, e, s, device=dev)
        pws = torch.empty(capi.spmm_workspace_bytes(e, e, s, wnnz) + 16, dtype=torch.uint8, device=dev)
        t = event_time_ms(lambda: capi.spmm_batched(e, e, s, b3, wri, wv, 0, wro, wci, xs, ys, pws), 20)
        res["spmm_projection_c3"] = {"ms": t, "gflops": 2.0 * wnnz * s * b3 / t / 1e6,
                                     "alg_gbs": (8.0 * b3 * e * s + 8.0 * wnnz) / t / 1e6}
        gw = torch.empty(wnnz, device=dev)
        sws = torch.empty(capi.sddmm_sum_workspace_bytes(e, s, e, wnnz) + 16, dtype=torch.uint8, device=dev)
        scr = torch.empty(capi.sddmm_sum_scratch_bytes(e, s, e, wnnz, b3) + 16, dtype=torch.uint8, device=dev)
        t = event_time_ms(lambda: capi.sddmm_sum_batched(e, s, e, b3, wri, wro, wci, ys, xs, gw, sws, scr), 20)
        res["sddmm_sum_weight_gradient_c3"] = {"ms": t, "gflops": 2.0 * wnnz * s * b3 / t / 1e6}
        big = uniform((reps, nnz), dev, 33)
        from torch_sputnik_amd import ops as _ops
        perm = _ops.csr_transpose_with_permutation(s, s, big[0].contiguous(), ro, ci)[3]
        lists = _ops.banded_lists(perm)
        t_plain = event_time_ms(lambda: _ops.permute_last(big, perm), 20)
        t_band = event_time_ms(lambda: _ops.permute_last_banded(big, *lists), 20)
        res["permute_values_c3"] = {"plain_ms": t_plain, "banded_ms": t_band,
                                    "note": "64 x nnz attention weights into the order of the mask's transpose"}
        del big
    except Exception as e2:  # noqa: BLE001 - extra metric, best effort
        res["spmm_projection_c3"] = {"error": str(e2)[:200]}
    # "many mask" family (one mask per batch element, shared by its 8 heads; mixed
    # sparsity as tests/test_attention_many_masks.py:26-36 draws it): b = 8, S = 1024
    try:
        b_mm, h_mm = 8, 8
        dens = (0.1, 0.2, 0.05, 0.5)
        topo = [random_csr(s, s, dens[i % len(dens)], dev, seed=70 + i) for i in range(b_mm)]
        nn = torch.tensor([t4[3] for t4 in topo], dtype=torch.int32)          # host, like utils.py:36
        width = int(nn.max())
        mri = torch.cat([t4[0] for t4 in topo])
        mro = torch.cat([t4[1] for t4 in topo])
        mci = torch.cat([t4[2] for t4 in topo])
        r_mm = b_mm * h_mm
        mscores = torch.zeros(r_mm, width, device=dev)
        mprobs = torch.zeros_like(mscores)
        mws = torch.empty(max(capi.sddmm_many_mask_workspace_bytes(b_mm, s, d, s, width),
                              capi.spmm_workspace_bytes(s, s, d, width)) + 16,
                          dtype=torch.uint8, device=dev)
        total = float(nn.sum()) * h_mm
        t = event_time_ms(lambda: capi.sddmm_many_mask(b_mm, s, d, s, nn, r_mm, mri, mro, mci, q, kk,
                                                       mscores, mws), 20)
        res["many_mask_sddmm"] = {"ms": t, "gflops": 2.0 * total * d / t / 1e6}
        t = event_time_ms(lambda: capi.sparse_softmax_many_mask(b_mm, s, nn, r_mm, mscores, mri, mro,
                                                                mci, d ** -0.5, mprobs), 20)
        res["many_mask_softmax"] = {"ms": t, "alg_gbs": 8.0 * total / t / 1e6,
                                    "hbm_frac": 8.0 * total / t / 1e6 / HBM_PEAK_GBS}
        t = event_time_ms(lambda: capi.spmm_many_mask(b_mm, s, s, d, nn, r_mm, mri, mprobs, mro, mci,
                                                      v, ctx, mws), 20)
        res["many_mask_spmm"] = {"ms": t, "gflops": 2.0 * total * d / t / 1e6}
        # the transposes of the backward pass: all masks in the same three launches against
        # mask after mask (the single-mask workspace)
        mvt = torch.zeros_like(mprobs)
        mrot = torch.empty(b_mm, s + 1, dtype=torch.int32, device=dev)
        mcit = torch.empty_like(mci)
        tws = {name: torch.empty(nbytes, dtype=torch.uint8, device=dev) for name, nbytes in (
            ("ms", capi.csr_transpose_many_mask_workspace_bytes(b_mm, s, s, width)),
            ("mask_after_mask_ms", capi.csr_transpose_workspace_bytes(s, s, width)))}
        res["many_mask_transpose"] = {
            name: event_time_ms(lambda w=w: capi.csr_transpose_many_mask(
                b_mm, s, s, nn, r_mm, mprobs, mro, mci, mvt, mrot, mcit, None, w), 20)
            for name, w in tws.items()}
        res["many_mask_note"] = ("b 8 x 8 heads, S 1024, head_dim 64, mask densities 0.1/0.2/0.05/0.5 "
                                 "repeated: one launch per operator (per phase of the transpose) "
                                 "for all masks")
        del mscores, mprobs, mws, mvt, tws
    except Exception as e3:  # noqa: BLE001 - extra metric, best effort
        res["many_mask_sddmm"] = {"error": str(e3)[:200]}
    # the same chain as ONE kernel (online softmax; scores / weights never reach HBM)
    aws = torch.empty(capi.sparse_attention_workspace_bytes(s, s, d, nnz), dtype=torch.uint8,
                      device=dev)
    t = event_time_ms(lambda: capi.sparse_attention_forward(s, s, d, reps, ri, ro, ci, q, kk, v,
                                                            d ** -0.5, ctx, None, aws), 20)
    by = reps * 16.0 * s * d + 4.0 * nnz + 4.0 * (2 * s + 1)
    res["attention_fused_c3"] = {"ms": t, "gflops": 4.0 * nnz * d * reps / t / 1e6,
                                 "alg_gbs": by / t / 1e6}
    # Whole SparseAttention.forward (modules/sparse_attention.py:105-128) through the
    # torch ops: 4 SparseLinear (left_spmm) + SDDMM + scale + softmax + SpMM, B=8, H=8, D=64.
    try:
        import numpy as np
        from torch_sputnik_amd import SparseAttention
        torch.manual_seed(0)
        emb, heads, batch = 512, 8, 8
        attn = SparseAttention(heads, emb, max_sequence_length=s, device=dev, sparsity=0.9,
                               mask_generator=np.random.default_rng(0))
        for lin in attn.linears:
            w = torch.randn(emb, emb, device=dev) * (torch.rand(emb, emb, device=dev) < 0.1)
            lin.weight = torch.nn.Parameter(w)
            lin.setup_sparse_tensors()
        x = torch.randn(batch, s, emb, device=dev)
        with torch.no_grad():
            t = event_time_ms(lambda: attn(x, x, x, None), 10)
        try:  # the same forward replayed from one hipGraph (torch_sputnik_amd/graphs.py)
            from torch_sputnik_amd.graphs import capture_forward
            fast = capture_forward(attn, x, x, x)
            # `ms`: a replay that first copies the caller's input into the graph's static
            # buffer (16.8 MB here: what a caller with its own tensor pays); `in_place_ms`: the
            # caller filled fast.static_inputs itself (the graph alone)
            res["sparse_attention_forward_c3_hip_graph"] = {
                "ms": event_time_ms(lambda: fast(x, x, x), 10),
                "in_place_ms": event_time_ms(lambda: fast(*fast.static_inputs), 10)}
        except Exception as e:  # noqa: BLE001
            res["sparse_attention_forward_c3_hip_graph"] = {"error": str(e)[:200]}
        res["sparse_attention_forward_c3"] = {"ms": t, "batch": batch, "heads": heads, "seq": s,
                                              "head_dim": emb // heads, "mask_density": 0.1,
                                              "projection_density": 0.1}
        # forward + backward of the whole module (gradients to the inputs and to every
        # projection's values), through the one-kernel attention forward with the
        # recomputing backward (low_memory_training: keeps no [B*H, nnz] tensor between the
        # passes -- a memory option, named `fused_training` in rounds 1-3) and through the
        # separate operators
        for key, flags in (("sparse_attention_fwd_bwd_c3_low_memory_training", {"low_memory_training": True}),
                           ("sparse_attention_fwd_bwd_c3_separate_ops", {"differentiable_softmax": True})):
            for name, value in {"low_memory_training": False, "differentiable_softmax": False, **flags}.items():
                setattr(attn, name, value)
            xg = x.clone().requires_grad_(True)
            gout = torch.randn(batch, s, emb, device=dev)

            def fwd_bwd():
                xg.grad = None
                for lin in attn.linears:
                    lin.values.grad = None
                attn(xg, xg, xg, None).backward(gout)

            res[key] = {"ms": event_time_ms(fwd_bwd, 10)}
            if key.endswith("separate_ops"):
                # the same step replayed from ONE hipGraph (torch_sputnik_amd/graphs.py): no
                # Python, no autograd walk, no launch gaps between its ~27 kernels --
                # `in_place_ms`: the caller wrote the static input / gradient buffers itself
                try:
                    from torch_sputnik_amd.graphs import capture_training_step
                    step = capture_training_step(attn, xg, xg, xg, grad_output=gout)
                    sx, sg = step.static_inputs[0], step.static_grad_output
                    res[key + "_graph"] = {
                        "ms": event_time_ms(lambda: step(xg, xg, xg, grad_output=gout), 10),
                        "in_place_ms": event_time_ms(lambda: step(sx, sx, sx, grad_output=sg), 10)}
                    del step
                except Exception as e:  # noqa: BLE001
                    res[key + "_graph"] = {"error": str(e)[:200]}
    except Exception as e:  # noqa: BLE001 - extra metric, best effort
        res["sparse_attention_forward_c3"] = {"error": str(e)[:200]}
    # dense widths that are no multiple of a tile width (the reference takes any n,
    # src/spmm_cuda.cu:32): 4096 x 4096 at density 0.1 against n columns
    try:
        mk = 4096
        ri, ro, ci, nnz = random_csr(mk, mk, 0.1, dev, seed=21)
        vals = uniform((nnz,), dev, 22)
        sweep_n = []
        for nn in (72, 200, 1000, 4000, 4096):
            bmat = uniform((mk, nn), dev, 23)
            cmat = torch.empty(mk, nn, device=dev)
            wsn = torch.empty(capi.spmm_workspace_bytes(mk, mk, nn, nnz) + 16, dtype=torch.uint8, device=dev)
            t = event_time_ms(lambda: capi.spmm_batched(mk, mk, nn, 1, ri, vals, 0, ro, ci, bmat, cmat, wsn), 20)
            # (`planned_ms`: the topology pre-pass done once, as the modules' plan cache does
            # for a static pattern -- the per-call form above re-derives it like the reference)
            capi.spmm_plan(mk, mk, nn, ri, ro, ci, wsn)
            tp = event_time_ms(lambda: capi.spmm_batched_planned(mk, mk, nn, 1, ri, vals, 0, ro, ci, bmat,
                                                                 cmat, wsn), 20)
            sweep_n.append({"n": nn, "ms": t, "gflops": 2.0 * nnz * nn / t / 1e6, "planned_ms": tp,
                            "planned_gflops": 2.0 * nnz * nn / tp / 1e6})
        res["spmm_4096x4096_d010_by_n"] = sweep_n
    except Exception as e:  # noqa: BLE001 - extra metric, best effort
        res["spmm_4096x4096_d010_by_n"] = {"error": str(e)[:200]}
    # Config 2's product on HALF-stored operands (this library's extension; the reference is
    # float32 only): the matrix-core route of round 5 (csrc/spmm_mfma.hip) by density --
    # one densified image of A per call + the dense tiles, so the time hardly depends on the
    # density; `float32_values`: A's values float32 (two half planes), B float16
    try:
        from torch_sputnik_amd.synthetic import random_csr as _rc
        half_sweep = []
        for dd in (0.5, 0.25, 0.1, 0.05):
            ri2, ro2, ci2, nnz2 = _rc(4096, 4096, dd, dev, seed=31)
            v2 = uniform((nnz2,), dev, 32)
            b2 = uniform((1, 4096, 4096), dev, 33).half()
            c2 = torch.empty(1, 4096, 4096, device=dev)
            row = {"density": dd, "nnz": nnz2}
            for key, vv in (("half_values", v2.half()), ("float32_values", v2)):
                need = capi.left_spmm_half_tiles_workspace_bytes(4096, 4096, 4096, nnz2, 1, vv, b2, torch.float16)
                if need == 0:
                    row[key] = None
                    continue
                wsh = torch.empty(need, dtype=torch.uint8, device=dev)
                t = event_time_ms(lambda: capi.left_spmm_half_tiles(4096, 4096, 4096, 1, ro2, ci2, vv, b2,
                                                                    torch.float16, c2, wsh), 10)
                row[key] = {"ms": t, "gflops": 2.0 * nnz2 * 4096 / t / 1e6}
            half_sweep.append(row)
            del b2, c2
        res["spmm_c2_f16_tiles_by_density"] = half_sweep
    except Exception as e:  # noqa: BLE001 - extra metric, best effort
        res["spmm_c2_f16_tiles_by_density"] = {"error": str(e)[:200]}
    m = n = 2048  # config 5: transpose of a 2048^2, density 0.2 weight
    ri, ro, ci, nnz = random_csr(m, n, 0.2, dev, seed=9)
    vals = uniform((nnz,), dev, 10)
    ov, oro = torch.empty_like(vals), torch.empty(n + 1, dtype=torch.int32, device=dev)
    oci = torch.empty(nnz, dtype=torch.int32, device=dev)
    ws = torch.empty(capi.csr_transpose_workspace_bytes(m, n, nnz), dtype=torch.uint8, device=dev)
    t = event_time_ms(lambda: capi.csr_transpose(m, n, 1, vals, ro, ci, ov, oro, oci, None, ws), 20)
    by = 16.0 * nnz + 4.0 * (m + n + 2)
    res["csr_transpose_c5"] = {"ms": t, "alg_gbs": by / t / 1e6, "hbm_frac": by / t / 1e6 / HBM_PEAK_GBS}
    # config 5, SparseLinear forward / backward operators (modules/sparse_linear.py:18-67):
    # 2048 x 2048 weight at density 0.2; batch 8 x seq 512 (BASELINE.json gives neither)
    batch, seq = 8, 512
    x = uniform((batch, n, seq), dev, 21)
    gy = uniform((batch, m, seq), dev, 22)
    y = torch.empty(batch, m, seq, device=dev)
    ws5 = torch.empty(capi.spmm_workspace_bytes(m, n, seq, nnz) + 16, dtype=torch.uint8, device=dev)
    t = event_time_ms(lambda: capi.spmm_batched(m, n, seq, batch, ri, vals, 0, ro, ci, x, y, ws5), 10)
    res["left_spmm_c5"] = {"ms": t, "gflops": 2.0 * nnz * seq * batch / t / 1e6, "batch": batch,
                           "seq": seq}
    gw = torch.empty(batch, nnz, device=dev)
    sws = torch.empty(capi.sddmm_workspace_bytes(m, seq, n, nnz) + 16, dtype=torch.uint8, device=dev)
    t = event_time_ms(lambda: capi.sddmm_batched(m, seq, n, batch, ri, ro, ci, gy, x, gw, sws), 10)
    res["sddmm_grad_values_c5"] = {"ms": t, "gflops": 2.0 * nnz * seq * batch / t / 1e6}
    # the two products on half-stored operands (BASELINE config 5 says fp16): the matrix-core
    # routes of round 5 (csrc/spmm_mfma.hip, csrc/sddmm_mfma.hip).  `float32_*`: that operand
    # is handed over as float32 and enters the tiles as half planes, not rounded.
    try:
        half = {}
        xh, gyh, vh = x.half(), gy.half(), vals.half()
        for key, v, d in (("half_values_half_dense", vh, xh), ("float32_values_half_dense", vals, xh),
                          ("half_values_float32_dense", vh, x), ("float32_values_float32_dense", vals, x)):
            need = capi.left_spmm_half_tiles_workspace_bytes(m, n, seq, nnz, batch, v, d, torch.float16)
            wst = torch.empty(need, dtype=torch.uint8, device=dev)
            t = event_time_ms(lambda: capi.left_spmm_half_tiles(m, n, seq, batch, ro, ci, v, d, torch.float16,
                                                                y, wst), 10)
            half[key] = {"ms": t, "gflops": 2.0 * nnz * seq * batch / t / 1e6}
        res["left_spmm_c5_f16_tiles"] = half
        grads = {}
        gsum = torch.empty(nnz, device=dev)
        wsum = torch.empty(capi.sddmm_sum_workspace_bytes(m, seq, n, nnz) + 16, dtype=torch.uint8, device=dev)
        capi.sddmm_sum_plan(m, seq, n, ri, ro, ci, wsum)
        for key, a, b in (("half_half", gyh, xh), ("float32_gradient_half_activations", gy, xh),
                          ("float32_float32_vector_kernels", gy, x)):
            mixed = a.dtype != b.dtype
            sc = torch.empty((capi.sddmm_sum_mixed_scratch_bytes(m, seq, n, nnz, batch, a, b) if mixed
                              else capi.sddmm_sum_scratch_bytes(m, seq, n, nnz, batch)) + 16,
                             dtype=torch.uint8, device=dev)
            call = capi.sddmm_sum_mixed if mixed else capi.sddmm_sum_typed
            t = event_time_ms(lambda: call(m, seq, n, batch, ri, ro, ci, a, b, gsum, wsum, sc, planned=True), 10)
            grads[key] = {"ms": t, "gflops": 2.0 * nnz * seq * batch / t / 1e6}
        res["sddmm_sum_grad_values_c5"] = grads
        del xh, gyh, vh
    except Exception as e:  # noqa: BLE001 - extra metric, best effort
        res["left_spmm_c5_f16_tiles"] = {"error": str(e)[:200]}
    # the same operators at config 5's STATED size M = N = K = 2048 (N of left_spmm is the
    # sequence length, modules/sparse_linear.py:28,89): batch 8 x seq 2048
    try:
        seq2 = 2048
        x2 = uniform((batch, n, seq2), dev, 24)
        gy2 = uniform((batch, m, seq2), dev, 25)
        y2 = torch.empty(batch, m, seq2, device=dev)
        ws52 = torch.empty(capi.spmm_workspace_bytes(m, n, seq2, nnz) + 16, dtype=torch.uint8, device=dev)
        t = event_time_ms(lambda: capi.spmm_batched(m, n, seq2, batch, ri, vals, 0, ro, ci, x2, y2, ws52), 10)
        res["left_spmm_c5_n2048"] = {"ms": t, "gflops": 2.0 * nnz * seq2 * batch / t / 1e6, "batch": batch,
                                     "seq": seq2,
                                     "kernel": capi.spmm_kernel_name(m, n, seq2, nnz, batch)}
        gw2 = torch.empty(batch, nnz, device=dev)
        sws2 = torch.empty(capi.sddmm_workspace_bytes(m, seq2, n, nnz) + 16, dtype=torch.uint8, device=dev)
        t = event_time_ms(lambda: capi.sddmm_batched(m, seq2, n, batch, ri, ro, ci, gy2, x2, gw2, sws2), 10)
        res["sddmm_grad_values_c5_n2048"] = {"ms": t, "gflops": 2.0 * nnz * seq2 * batch / t / 1e6}
        del x2, gy2, y2, gw2, ws52, sws2
    except Exception as e:  # noqa: BLE001 - extra metric, best effort
        res["left_spmm_c5_n2048"] = {"error": str(e)[:200]}
    # config 5 end to end: SparseLinear forward + backward through the torch ops and the
    # autograd Function (left_spmm; sddmm + transposed topology + left_spmm), fp32 and with
    # the input stored in fp16.  Default = transposed topology and kernel plans cached per
    # static topology; `_per_call` = the reference's behaviour (csr_transpose + diffsort and
    # the pre-passes in every call, modules/sparse_linear.py:52-57).
    try:
        from torch_sputnik_amd import SparseLinear
        layer = SparseLinear(n, m).to(dev)
        w = torch.randn(m, n, device=dev) * (torch.rand(m, n, device=dev) < 0.2)
        layer.weight = torch.nn.Parameter(w)
        layer.setup_sparse_tensors()
        values32 = layer.values
        for name, dt, sq in (("fp32", torch.float32, seq), ("fp16_storage", torch.float16, seq),
                             ("fp16_storage_and_weights", torch.float16, seq),
                             ("n2048_fp32", torch.float32, 2048), ("n2048_fp16_storage", torch.float16, 2048),
                             ("n2048_fp16_storage_and_weights", torch.float16, 2048)):
            # (`_and_weights`: the layer's values are stored in half precision too)
            layer.values = (torch.nn.Parameter(values32.detach().to(dt)) if name.endswith("and_weights")
                            else values32)
            xin = torch.randn(batch, sq, n, device=dev).to(dt).requires_grad_(True)
            gout = torch.randn(batch, m, sq, device=dev)

            def fwd_bwd():
                layer.values.grad = None
                xin.grad = None
                layer(xin).backward(gout)

            res[f"sparse_linear_fwd_bwd_c5_{name}"] = {"ms": event_time_ms(fwd_bwd, 10), "batch": batch,
                                                       "seq": sq}
            if name == "fp32":
                from torch_sputnik_amd import functional
                functional.enable_transpose_cache(False)
                functional.enable_plan_cache(False)
                try:
                    fwd_bwd()
                    res["sparse_linear_fwd_bwd_c5_fp32_per_call"] = {
                        "ms": event_time_ms(fwd_bwd, 10)}
                finally:
                    functional.enable_transpose_cache(functional.TRANSPOSE_CACHE_DEFAULT)
                    functional.enable_plan_cache(functional.PLAN_CACHE_DEFAULT)
    except Exception as e:  # noqa: BLE001 - extra metric, best effort
        res["sparse_linear_fwd_bwd_c5"] = {"error": str(e)[:200]}
    return res


def load_pmc_traffic(build_id, kernel):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3
    PMC passes (profiles/spmm_c2_d010_traffic.json, collected as
    MI355X_MICROARCH.md prescribes by tools/collect_profiles.sh).  The file
    records the build it was collected on and the kernel it describes: a figure
    from another build or for another kernel is not quoted (None)."""
    path = os.path.join(REPO, "profiles", "spmm_c2_d010_traffic.json")
    try:
        with open(path) as f:
            rec = json.load(f)
    except (OSError, ValueError):
        return None
    if rec.get("build_id") != build_id or not kernel.startswith(rec.get("dominant_kernel", "?")):
        return None
    return rec.get("traffic_bytes_per_launch")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--device-warmup-s", type=float, default=0.25,
                    help="seconds of untimed steps before the W warm-up steps (clock ramp)")
    ap.add_argument("--compute-only", action="store_true",
                    help="N>1: time the local launches only (no all-gather of C)")
    ap.add_argument("--overlap-chunks", type=str, default="4,8,16",
                    help="N>1: the overlapped schedules split a rank's replicas into this many "
                         "chunks and exchange chunk i (side stream) while chunk i+1 is computed; "
                         "a comma list is swept and the best chunk count per transport reported")
    ap.add_argument("--replicas-per-gpu", type=int, default=0,
                    help="override (default 1 at --gpus 1, 16 otherwise)")
    ap.add_argument("--no-extras", action="store_true", help="skip sweep / cpu baseline / other ops")
    ap.add_argument("--schedule-timeout-s", type=float, default=240.0,
                    help="N>1: a schedule that has not finished after this long ends the run; "
                         "rank 0 prints the line with the schedules measured so far")
    args = ap.parse_args()
    # Contract: stdout carries ONE JSON line.  Libraries print there too (RCCL's
    # version banner at communicator creation, for one), so stdout is pointed at
    # stderr for the whole run and the result goes to the saved descriptor.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    import __graft_entry__
    # fresh checkout: compile the native libraries first (one rank per node builds,
    # the others wait for the files)
    if int(os.environ.get("LOCAL_RANK", "0")) == 0:
        __graft_entry__.ensure_built()
    else:
        deadline = time.time() + 900
        waited = False
        while not __graft_entry__.is_built() and time.time() < deadline:
            time.sleep(2)
            waited = True
        if waited:
            time.sleep(10)  # the last link step may still be writing

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # BENCH_FORCE_DIST=1 exercises the torch.distributed path with one rank.
    distributed = world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (the hot path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:   # BENCH_FORCE_DIST without a launcher
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
            os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", device_id=dev)
    n_gpus = world if distributed else 1
    if args.gpus != n_gpus and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using {n_gpus}", file=sys.stderr)

    replicas = args.replicas_per_gpu or (1 if n_gpus == 1 else REPLICAS_PER_GPU_MULTI)
    problem = SpmmProblem(dev, HEADLINE_DENSITY, replicas, seed=1234 + 1000 * DENSITIES.index(HEADLINE_DENSITY) + rank)

    def run_timed(step_fn):
        """W untimed steps, then EXACTLY K steps between barrier + synchronize on
        both sides; the maximum over the ranks, in ms per step."""
        for _ in range(args.warmup):
            step_fn()
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step_fn()
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if distributed:
            t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed / args.steps * 1e3

    # The GPU needs a few tens of milliseconds of load before its clocks settle
    # (tools/sustain_bench.py: the first 50 back-to-back steps average 0.433 ms,
    # every later batch 0.397-0.405 ms): bring it to its sustained state first,
    # untimed, so that K timed steps measure the steady rate whatever W is.
    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < args.device_warmup_s:
        for _ in range(10):
            problem.step()  # compute only: a time-based loop must not contain collectives
        torch.cuda.synchronize()

    # Dominant kernel alone (HIP events on the launch stream = torch's current stream);
    # before the schedules so that a partial result can carry it (see the watchdog).
    problem.step()
    kernel_ms = event_time_ms(problem.kernel_only, max(10, args.steps))
    achieved_gbs = problem.bytes / (kernel_ms * 1e-3) / 1e9
    from torch_sputnik_amd import capi as _capi
    build_id = _capi.build_id()
    dominant_kernel = _capi.spmm_kernel_name(M, K, N, problem.nnz, replicas)

    def core_result(headline, ms_per_step, multi):
        value = problem.flops * n_gpus / (ms_per_step * 1e-3) / 1e9
        result = {
            "metric": "SpMM effective GFLOP/s (2*nnz*N/t), M=N=K=4096 fp32, density 0.1",
            "value": value, "unit": "GFLOP/s", "n_gpus": n_gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": ("C2: 2-D SpMM M=N=K=4096 density 0.1 (nnz=%d), %d replica(s)" % (problem.nnz, replicas))
                if n_gpus == 1 else
                ("C4: batched SpMM M=N=K=4096 density 0.1, %d replicas per GPU, %d total, replica-sharded"
                 % (replicas, replicas * n_gpus)),
                "replicas_per_gpu": replicas, "nnz": problem.nnz,
                "device_warmup_s": args.device_warmup_s,
                "step": "sputnik_hip_spmm_batched (pre-pass + kernel) via the C ABI",
                "collective": ("none (one GPU)" if multi is None else
                               "all-gather of C over RCCL inside the step; schedule timed as `value`: "
                               + headline),
                "inputs": "uniform-random sparsity (tests/connectors.py distribution), U[0,1) values, resident in HBM",
            },
            "algorithmic_gbs": problem.bytes * n_gpus / (ms_per_step * 1e-3) / 1e9,
            "roofline": {
                "bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved_gbs / HBM_PEAK_GBS,
                # the PMC passes were taken at one replica per launch
                "traffic": load_pmc_traffic(build_id, dominant_kernel) if replicas == 1 else None,
                "kernel": dominant_kernel, "kernel_ms": kernel_ms,
                "algorithmic_bytes_per_launch": problem.bytes,
                "note": "fp32 SpMM at this size is above the HBM ridge (93 flop/B vs 19.7): see roofline_valu",
            },
            "roofline_valu": {
                "bound": "valu_fp32+lds", "achieved": problem.flops / (kernel_ms * 1e-3) / 1e12,
                "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": problem.flops / (kernel_ms * 1e-3) / 1e12 / VALU_PEAK_TFLOPS,
                "lds_operand_bound_tflops": 78.6,
            },
            "library": {"version": _capi.version(), "build_id": build_id},
            "a100_reference_sputnik_gflops": 3416.0,  # README.md:54 of the reference (other hardware)
        }
        if multi is not None:
            # SURVEY.md 8e's three figures (and both transports), each timed over the
            # same K steps: compute only / compute + all-gather / overlapped
            result["headline_schedule"] = headline
            result.update({k: multi[k] for k in ("compute_only", "allgather", "allgather_overlapped")})
            result["exchange"] = {k: multi[k] for k in multi
                                  if k not in ("compute_only", "allgather", "allgather_overlapped")}
        return result

    def pick_headline(timings):
        gathered = [k for k, v in timings.items() if k.startswith("allgather") and v]
        return min(gathered, key=lambda k: timings[k]) if gathered else "compute_only"

    multi = None
    if not distributed:
        ms_per_step = run_timed(problem.step)
        headline = "compute (one GPU: nothing to exchange)"
    else:
        # North star / SURVEY.md 8e: the replica-sharded product WITH the all-gather of
        # C over RCCL.  Every schedule is timed the same way (W + K steps each):
        #   compute_only                      the local launch, no exchange
        #   exchange_only_{collective,p2p}    the all-gather alone (link bandwidth)
        #   allgather_{collective,p2p}        launch, then exchange (unoverlapped)
        #   allgather_overlapped_{...}        chunk i exchanged while chunk i+1 computes
        # `value` is the best schedule that includes the all-gather.
        chunk_counts = sorted({max(1, min(int(c), replicas)) for c in str(args.overlap_chunks).split(",")})
        ex = Exchange(problem, world, rank, dev, chunk_counts[0])
        p2p = os.environ.get("BENCH_NO_P2P") != "1"

        def overlapped(chunks, exchange_chunk):
            def run():
                ex.overlapped(exchange_chunk)
            run.chunks = chunks     # (the chunk layout is switched before the schedule's warm-up)
            return run

        variants = {"compute_only": problem.step,
                    "exchange_only_collective": ex.collective,
                    "allgather_collective": lambda: (problem.step(), ex.collective())}
        if p2p:
            variants.update({"exchange_only_p2p": ex.p2p,
                             "allgather_p2p": lambda: (problem.step(), ex.p2p())})
        for c in chunk_counts:
            variants[f"allgather_overlapped_collective_c{c}"] = overlapped(c, ex.collective_chunk)
            if p2p:
                variants[f"allgather_overlapped_p2p_c{c}"] = overlapped(c, ex.p2p_chunk)
        if args.compute_only:
            variants = {"compute_only": problem.step}
        timings = {}
        # A schedule that never returns (a transport that hangs on some fabric) must not
        # lose the schedules already measured: past the deadline rank 0 prints the line
        # with what it has and every rank leaves.  (A thread, not a signal: the main
        # thread would be inside a blocking runtime call.)
        watchdog = Watchdog(args.schedule_timeout_s, printer=(rank == 0))
        for name, fn in variants.items():
            def partial_line(name=name):
                if rank != 0 or not timings.get("compute_only"):
                    return None
                done = dict(timings)
                head = pick_headline(done)
                line = core_result(head, done[head], ex.report(done, n_gpus))
                line["watchdog"] = f"schedule {name} did not finish in {args.schedule_timeout_s:.0f} s"
                return (json.dumps(line) + "\n").encode()
            watchdog.arm(partial_line, result_fd)
            try:
                if os.environ.get("BENCH_TEST_HANG") == name:   # (test of the watchdog)
                    time.sleep(1e6)
                if hasattr(fn, "chunks"):
                    ex.set_chunks(fn.chunks)
                ex.poison()
                timings[name] = run_timed(fn)
                # (the schedules share one gathered buffer in different layouts: each is
                # checked right after its own run)
                ex.verify(name)
            except Exception as e:  # noqa: BLE001 - one schedule failing must not lose the others
                timings[name] = None
                print(f"[bench] rank {rank}: schedule {name} failed: {e}", file=sys.stderr)
            watchdog.disarm()
        headline = pick_headline(timings)
        ms_per_step = timings[headline]
        multi = ex.report(timings, n_gpus)

    result = None
    if rank == 0:
        result = core_result(headline, ms_per_step, multi)
        if not args.no_extras and n_gpus == 1:
            sweep = []
            for d in DENSITIES:
                p = problem if d == HEADLINE_DENSITY else SpmmProblem(dev, d, 1, seed=1234 + 1000 * DENSITIES.index(d))
                p.step()
                # 20 warm-up + 100 timed launches, median and minimum
                ms_full, ms_full_min = event_time_ms(p.step, 100, warmup=20, with_min=True)
                ms_kern = event_time_ms(p.kernel_only, 100, warmup=20)
                sweep.append({"density": d, "nnz": p.nnz, "ms": ms_full, "ms_min": ms_full_min,
                              "kernel_ms": ms_kern,
                              "gflops": p.flops / ms_full / 1e6, "alg_gbs": p.bytes / ms_full / 1e6,
                              "hbm_frac": p.bytes / ms_kern / 1e6 / HBM_PEAK_GBS,
                              "valu_frac": p.flops / ms_kern / 1e9 / VALU_PEAK_TFLOPS,
                              "dense_gpu_ms": dense_gpu_ms(p)})
                if p is not problem:
                    del p
            result["sweep"] = sweep
            result["dense_gpu"] = dense_crossover(sweep)
            result["other_ops"] = other_ops(dev)
            # the same headline call WITH output / workspace allocation, i.e.
            # through the reference-compatible torch op (src/spmm_cuda.cu:9-60 semantics)
            import torch_sputnik
            ms_op = event_time_ms(lambda: torch_sputnik.spmm(M, K, problem.values, problem.ri,
                                                             problem.ro, problem.ci, problem.dense), 20)
            result["other_ops"]["spmm_c2_d010_via_torch_op"] = {
                "ms": ms_op, "gflops": problem.flops / ms_op / 1e6,
                "note": "allocates C and the workspace per call (caching allocator)"}
            # the same product with the rows in the order the reference's modules
            # produce (`diffsort`: ASCENDING length, modules/spmm.py:4-6) instead of
            # tests/sparse_matrix.py:22's descending order -- results and speed must not
            # depend on the permutation (rows are dealt to workgroups)
            try:
                from torch_sputnik_amd.topology import diffsort
                ri_up = diffsort(problem.ro)
                ms_up = event_time_ms(lambda: problem.capi.spmm_batched(
                    M, K, N, 1, ri_up, problem.values, 0, problem.ro, problem.ci, problem.dense,
                    problem.out, problem.ws), 20)
                result["other_ops"]["spmm_c2_d010_rows_in_diffsort_order"] = {
                    "ms": ms_up, "gflops": problem.flops / ms_up / 1e6}
            except Exception as e:  # noqa: BLE001 - extra metric, best effort
                result["other_ops"]["spmm_c2_d010_rows_in_diffsort_order"] = {"error": str(e)[:200]}
            # What one GPU does in the N > 1 runs (config 4's share: 16 replicas in one
            # launch), so that weak-scaling efficiency can be taken against the SAME
            # per-GPU workload rather than against the single product above.
            try:
                p16 = SpmmProblem(dev, HEADLINE_DENSITY, REPLICAS_PER_GPU_MULTI, seed=4234)
                p16.step()
                ms16 = event_time_ms(p16.step, 10)
                result["per_gpu_share_of_multi_gpu_runs"] = {
                    "replicas": REPLICAS_PER_GPU_MULTI, "ms": ms16, "gflops": p16.flops / ms16 / 1e6}
                del p16
            except Exception as e:  # noqa: BLE001 - extra metric, best effort
                result["per_gpu_share_of_multi_gpu_runs"] = {"error": str(e)[:200]}
            result["cpu_baseline"] = cpu_baseline(problem)
        elif n_gpus == 1:
            result["cpu_baseline"] = None
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(result) + "\n").encode())
    os.close(result_fd)


if __name__ == "__main__":
    main()
