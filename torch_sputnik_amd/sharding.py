"""Replica-dimension sharding of the batched ops over the GPUs of one node.

The reference runs its "replication" dimension as a serial host loop on one
GPU (src/spmm_cuda.cu:48-57, src/sddmm_cuda.cu:45-54, src/softmax_cuda.cu:35-43);
the replicas share only the small int32 topology and are otherwise independent.
Here that dimension is the unit of distribution: one process per GPU
(``torch.distributed``, backend "nccl" = RCCL over xGMI; "gloo" in the CPU
tests), rank r owns the contiguous replica block ``local_range(R)``, runs ONE
batched launch on it, and -- only if the caller needs the whole result on every
rank -- the blocks are all-gathered.

xGMI is a point-to-point full mesh (7 links per GPU), so a ring all-gather is
bound by one link.  ``all_gather_replicas(..., mode="p2p")`` therefore has every
rank send its block to all peers at once (grouped isend/irecv, received
straight into the peer's slice of the output: no staging copy); the default
"collective" mode leaves the algorithm to RCCL.  ``overlap_chunks > 1`` cuts the
local block into chunks and starts the exchange of a finished chunk while the
next one is being computed (communication on RCCL's own stream).
"""
import os

import torch
import torch.distributed as dist

from . import ops

# Test knob: with one rank there is nothing to exchange and every gather is a
# local copy.  SPUTNIK_SHARDING_FORCE_COLLECTIVE=1 sends that copy through the
# communicator anyway (all_gather_into_tensor of one block / a send+recv to
# self), so that a one-GPU box executes the RCCL code path
# (tests/test_gpu_sharding.py).
def _force_collective():
    return os.environ.get("SPUTNIK_SHARDING_FORCE_COLLECTIVE") == "1"


def _world(group):
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def local_range(replicas, world_size, rank):
    """[start, stop) of the replicas rank `rank` owns: contiguous blocks, the
    first ``replicas % world_size`` ranks take one extra."""
    base, extra = divmod(replicas, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard(tensor, group=None, dim=0):
    """This rank's block of a tensor that is replicated along `dim`."""
    world, rank = _world(group)
    start, stop = local_range(tensor.size(dim), world, rank)
    return tensor.narrow(dim, start, stop - start)


def all_gather_replicas(local, replicas, group=None, mode="collective", out=None):
    """Blocks [L_r, ...] of every rank -> [replicas, ...] on every rank, in
    replica order.  Returns (out, pending) where `pending` is a list of work
    handles to ``wait()`` on (empty once complete / for world size 1)."""
    world, rank = _world(group)
    if out is None:
        out = local.new_empty((replicas,) + tuple(local.shape[1:]))
    if world == 1 and not (_force_collective() and dist.is_initialized()):
        out.copy_(local)
        return out, []
    ranges = [local_range(replicas, world, r) for r in range(world)]
    even = replicas % world == 0
    local = local.contiguous()
    if mode == "collective" and even:
        return out, [dist.all_gather_into_tensor(out, local, group=group, async_op=True)]
    if mode == "collective":
        # uneven blocks: list form (c10d stages through a flat buffer)
        pieces = [out[a:b] for a, b in ranges]
        if pieces[rank].shape != local.shape:
            raise ValueError("local block does not match this rank's replica range")
        longest = max(b - a for a, b in ranges)
        padded = local.new_zeros((longest,) + tuple(local.shape[1:]))
        padded[: local.shape[0]] = local
        bufs = [local.new_empty(padded.shape) for _ in range(world)]
        dist.all_gather(bufs, padded, group=group)
        for (a, b), buf in zip(ranges, bufs):
            out[a:b] = buf[: b - a]
        return out, []
    if mode != "p2p":
        raise ValueError(f"unknown all-gather mode {mode!r}")
    a, b = ranges[rank]
    p2p = []
    if world == 1:      # forced (test knob): this rank's block to itself through RCCL
        p2p = [dist.P2POp(dist.isend, local, rank, group=group),
               dist.P2POp(dist.irecv, out[a:b], rank, group=group)]
    else:
        out[a:b].copy_(local)
    for step in range(1, world):
        dst = (rank + step) % world
        src = (rank - step) % world
        sa, sb = ranges[src]
        if local.shape[0]:
            p2p.append(dist.P2POp(dist.isend, local, dist.get_global_rank(group, dst) if group else dst,
                                  group=group))
        if sb > sa:
            p2p.append(dist.P2POp(dist.irecv, out[sa:sb], dist.get_global_rank(group, src) if group else src,
                                  group=group))
    return out, (dist.batch_isend_irecv(p2p) if p2p else [])


def _wait_all(pending):
    for w in pending:
        w.wait()


class ReplicaExchange:
    """The all-gather of a replica-sharded result on PREALLOCATED buffers:
    this rank computes its `r` replicas into ``local`` ([r, ...], any trailing shape) and
    every rank ends with all ``world * r`` of them in ``rank_major`` ([world, r, ...] =
    global replica order).  Nothing is allocated per step and nothing is copied twice.
    Two transports -- RCCL's all_gather_into_tensor, and direct grouped send / recv to
    every peer (xGMI is a full mesh: all 7 links carry one block each, no ring) -- each
    either after the whole block is computed or chunk by chunk on a side stream while the
    next chunk is computed (``overlapped``).  `replica_parallel` below runs on it, and so
    does ``bench.py --gpus N`` (its Exchange class adds only the report).

    ``compute_range(a, b)`` writes replicas [a, b) of the local block into ``local[a:b]``.
    """

    def __init__(self, local, world, rank, chunks=1, compute_range=None, group=None):
        self.dist, self.world, self.rank, self.group = dist, world, rank, group
        self.local = local
        self.dev = local.device
        self.compute_range = compute_range
        r = local.shape[0]
        self.replicas = r
        self.tail = tuple(local.shape[1:])
        self.per_replica = int(local[0].numel()) if r else 0
        self.flat = local.new_empty(world * r * self.per_replica)
        self.rank_major = self.flat.view((world, r) + self.tail)          # = global replica order
        self.set_chunks(chunks)
        # (tests/test_bench_exchange.py drives this class on CPU tensors over gloo)
        self.side = torch.cuda.Stream(device=self.dev) if self.dev.type == "cuda" else None
        self.bytes_per_peer = float(r * self.per_replica * local.element_size())

    def _peer(self, r):
        return dist.get_global_rank(self.group, r) if self.group is not None else r

    def set_chunks(self, chunks):
        """The overlapped schedules exchange the local block in `chunks` pieces (at
        most one replica each): chunk i travels while chunk i + 1 is computed."""
        r, world = self.replicas, self.world
        chunks = max(1, min(chunks, r)) if r else 1
        per = (r + chunks - 1) // chunks if r else 1
        self.bounds = [(a, min(a + per, r)) for a in range(0, r, per)] or [(0, 0)]
        # collective chunks land chunk-major ([chunk][rank][replicas of the chunk]):
        # one contiguous all_gather_into_tensor each, in the same storage
        self.chunk_major, off = [], 0
        for a, b in self.bounds:
            size = world * (b - a) * self.per_replica
            self.chunk_major.append(self.flat[off:off + size].view((world * (b - a),) + self.tail))
            off += size
        return len(self.bounds)

    def collective(self):
        dist.all_gather_into_tensor(self.rank_major.view((-1,) + self.tail), self.local, group=self.group)

    def collective_chunk(self, c):
        a, b = self.bounds[c]
        dist.all_gather_into_tensor(self.chunk_major[c], self.local[a:b], group=self.group)

    def p2p_chunk(self, c):
        a, b = self.bounds[c]
        self.p2p(a, b)

    def p2p(self, a=0, b=None):
        b = self.replicas if b is None else b
        send = self.local[a:b]
        p2p_ops = []
        if self.world == 1:   # forced (test knob / BENCH_FORCE_DIST): the block to itself, through RCCL
            p2p_ops = [dist.P2POp(dist.isend, send, self._peer(0), group=self.group),
                       dist.P2POp(dist.irecv, self.rank_major[0, a:b], self._peer(0), group=self.group)]
        else:
            self.rank_major[self.rank, a:b].copy_(send)
            for step in range(1, self.world):
                dst, src = (self.rank + step) % self.world, (self.rank - step) % self.world
                p2p_ops.append(dist.P2POp(dist.isend, send, self._peer(dst), group=self.group))
                p2p_ops.append(dist.P2POp(dist.irecv, self.rank_major[src, a:b], self._peer(src),
                                          group=self.group))
        for work in dist.batch_isend_irecv(p2p_ops):
            work.wait()

    def chunks_in_replica_order(self):
        """After the overlapped COLLECTIVE schedule the gathered data lies chunk-major;
        this returns it as [world * r, ...] in global replica order (one gather-copy --
        the p2p transport and the unchunked collective need none: ``rank_major``)."""
        parts = [cm.view((self.world, b - a) + self.tail) for cm, (a, b) in zip(self.chunk_major, self.bounds)]
        return torch.cat(parts, dim=1).reshape((self.world * self.replicas,) + self.tail)

    def overlapped(self, exchange_chunk):
        """compute chunk i on the main stream; its exchange runs on the side stream
        behind an event while the main stream computes chunk i + 1."""
        if self.side is None:          # CPU (tests): the same order, no streams
            for c, (a, b) in enumerate(self.bounds):
                self.compute_range(a, b)
                exchange_chunk(c)
            return
        main = torch.cuda.current_stream(self.dev)
        for c, (a, b) in enumerate(self.bounds):
            self.compute_range(a, b)
            done = torch.cuda.Event()
            done.record(main)
            with torch.cuda.stream(self.side):
                self.side.wait_event(done)
                exchange_chunk(c)
        main.wait_stream(self.side)


def replica_parallel(op, replicated_args, replicas, group=None, gather_output=True,
                     gather_mode="collective", overlap_chunks=1, local_operands=False,
                     exchange=None):
    """Generic driver: ``op(*local_args) -> [L, ...]`` is run on this rank's
    block of every tensor in `replicated_args` (tensors replicated along dim 0;
    anything else is passed through), then optionally all-gathered.

    ``local_operands=True`` is the shard-at-origin form: the tensors passed ARE
    this rank's blocks (``local_range(replicas, world, rank)`` replicas each), so
    no rank ever holds another rank's operands; `replicas` is the global count
    (default: local count x world size).

    ``exchange``: a `ReplicaExchange` kept by the caller (even shards: its ``local`` has
    this rank's replica count).  The gather then runs on its preallocated buffers -- no
    allocation per call or per chunk -- and an `op` that takes ``out=`` (``op.writes_out``)
    computes straight into them, so nothing is copied twice.  The result is a view of the
    exchange's buffer: valid until the exchange is used again."""
    world, rank = _world(group)
    start, stop = local_range(replicas, world, rank)
    count = stop - start

    def block(a, b):
        if local_operands:   # indices are global replica numbers; operands start at `start`
            a, b = a - start, b - start
            return [x[a:b] if (torch.is_tensor(x) and x.dim() > 0 and x.size(0) == count and flag)
                    else x for x, flag in replicated_args]
        return [x[a:b] if (torch.is_tensor(x) and x.dim() > 0 and x.size(0) == replicas and flag) else x
                for x, flag in replicated_args]

    if local_operands:
        for x, flag in replicated_args:
            if flag and torch.is_tensor(x) and x.size(0) != count:
                raise ValueError(f"local_operands: this rank owns {count} of {replicas} replicas, "
                                 f"got a block of {x.size(0)}")
    if not gather_output or (world == 1 and not _force_collective()):
        return op(*block(start, stop))
    even = replicas % world == 0
    chunks = max(1, min(overlap_chunks, count)) if count else 1
    if chunks > 1 and not even:
        raise ValueError("overlap_chunks > 1 needs replicas divisible by the world size")
    if not even:    # ragged shards: the list form of all_gather_replicas
        local_out = op(*block(start, stop))
        out, pending = all_gather_replicas(local_out, replicas, group, gather_mode)
        _wait_all(pending)
        return out
    if gather_mode not in ("collective", "p2p"):
        raise ValueError(f"unknown all-gather mode {gather_mode!r}")

    in_place = getattr(op, "writes_out", False)
    ex = exchange
    first = None
    if ex is None:
        # (the first chunk's result tells shape, type and device of the buffers)
        step = (count + chunks - 1) // chunks
        first = op(*block(start, start + min(step, count)))
        local = first.new_empty((count,) + tuple(first.shape[1:]))
        ex = ReplicaExchange(local, world, rank, chunks, group=group)
    elif ex.replicas != count or ex.world != world:
        raise ValueError("exchange was made for another shard")
    else:
        ex.set_chunks(chunks)

    def compute_range(a, b):
        nonlocal first
        if first is not None and a == 0:
            ex.local[a:b].copy_(first)
            first = None
        elif in_place:
            op(*block(start + a, start + b), out=ex.local[a:b])
        else:
            ex.local[a:b].copy_(op(*block(start + a, start + b)))

    ex.compute_range = compute_range
    if chunks == 1:
        compute_range(0, count)
        (ex.collective if gather_mode == "collective" else ex.p2p)()
        return ex.rank_major.view((replicas,) + ex.tail)
    ex.overlapped(ex.collective_chunk if gather_mode == "collective" else ex.p2p_chunk)
    if gather_mode == "collective":
        return ex.chunks_in_replica_order()
    return ex.rank_major.view((replicas,) + ex.tail)


def _global_count(tensor, local_operands, replicas, group):
    """Global replica count: given, or the operand's dim 0 (x world size when the
    operand is this rank's block; that shortcut needs even shards)."""
    if replicas is not None:
        return int(replicas)
    return tensor.size(0) * (_world(group)[0] if local_operands else 1)


def make_exchange(local_replicas, tail_shape, like, group=None, chunks=1):
    """A `ReplicaExchange` for a sharded op whose local result is [local_replicas, *tail_shape]
    (dtype / device of `like`): keep it and pass it as ``exchange=`` to the ops below to
    gather on the same buffers in every step."""
    world, rank = _world(group)
    local = like.new_empty((local_replicas,) + tuple(tail_shape), dtype=torch.float32)
    return ReplicaExchange(local, world, rank, chunks, group=group)


def spmm(m, k, values, row_indices, row_offsets, column_indices, dense, group=None,
         gather_output=True, gather_mode="collective", overlap_chunks=1, local_operands=False,
         replicas=None, exchange=None):
    """Replica-parallel batched SpMM.  `values` [R,nnz] and `dense` [R,k,n] are the
    global operands (every rank passes the same tensors and uses its block) or,
    with ``local_operands=True``, this rank's blocks only -- the shard-at-origin
    form, in which no GPU ever holds the other ranks' B (config 4: 1.07 GB of the
    8.6 GB).  Each rank computes its block of C and, with `gather_output`, every
    rank returns the whole [R,m,n].  ``exchange`` (`make_exchange`): gather on
    preallocated buffers, the kernels writing straight into them."""
    count = _global_count(dense, local_operands, replicas, group)

    def op(v, d, out=None):
        if out is not None:
            _spmm_into(m, k, v, row_indices, row_offsets, column_indices, d, out, exchange)
            return out
        res = ops.spmm(m, k, v, row_indices, row_offsets, column_indices, d)
        return res.reshape((d.size(0), m, d.size(-1)))

    # (the in-place form goes through the C ABI: float32 GPU operands, int32 topology)
    op.writes_out = (exchange is not None and dense.is_cuda and dense.dtype == torch.float32 and
                     values.dtype == torch.float32 and
                     all(t.dtype == torch.int32 for t in (row_indices, row_offsets, column_indices)))
    return replica_parallel(op, [(values, True), (dense, True)], count, group, gather_output,
                            gather_mode, overlap_chunks, local_operands, exchange)


def _spmm_into(m, k, values, row_indices, row_offsets, column_indices, dense, out, exchange):
    """spmm of a block of replicas written into `out` ([L, m, n], a slice of the exchange's
    buffer) through the C ABI -- no output allocation; the workspace lives on the exchange."""
    from . import capi
    n, count = dense.size(-1), dense.size(0)
    nnz = column_indices.numel()
    ws = getattr(exchange, "_spmm_workspace", None)
    need = capi.spmm_workspace_bytes(m, k, n, nnz) + 16
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=dense.device)
        exchange._spmm_workspace = ws
    capi.spmm_batched(m, k, n, count, row_indices, values.contiguous(), nnz if values.dim() == 2 else 0,
                      row_offsets, column_indices, dense.contiguous(), out, ws)


def left_spmm(m, k, values, row_indices, row_offsets, column_indices, dense, group=None,
              gather_output=True, gather_mode="collective", overlap_chunks=1,
              local_operands=False, replicas=None):
    """Replica-parallel left_spmm: one sparse matrix (replicated), dense [R,k,n] sharded."""
    count = _global_count(dense, local_operands, replicas, group)

    def op(d):
        return ops.left_spmm(m, k, values, row_indices, row_offsets, column_indices, d)

    return replica_parallel(op, [(dense, True)], count, group, gather_output, gather_mode,
                            overlap_chunks, local_operands)


def sddmm(m, n, row_indices, row_offsets, column_indices, lhs_matrix, rhs_matrix, group=None,
          gather_output=True, gather_mode="collective", local_operands=False, replicas=None):
    """Replica-parallel batched SDDMM -> [R,nnz]."""
    count = _global_count(lhs_matrix, local_operands, replicas, group)

    def op(l, r):
        out = ops.sddmm(m, n, row_indices, row_offsets, column_indices, l, r)
        return out.reshape((l.size(0), -1))

    return replica_parallel(op, [(lhs_matrix, True), (rhs_matrix, True)], count, group,
                            gather_output, gather_mode, 1, local_operands)


def sparse_softmax(values, row_indices, row_offsets, column_indices, group=None,
                   gather_output=True, gather_mode="collective", local_operands=False,
                   replicas=None):
    """Replica-parallel sparse softmax over [R,nnz] values."""
    count = _global_count(values, local_operands, replicas, group)

    def op(v):
        return ops.sparse_softmax(v, row_indices, row_offsets, column_indices)

    return replica_parallel(op, [(values, True)], count, group, gather_output, gather_mode, 1,
                            local_operands)


def sparse_attention(query, key, value, row_indices, row_offsets, column_indices, scale,
                     group=None, gather_output=True, gather_mode="collective",
                     local_operands=False, replicas=None):
    """Replica-parallel fused attention: query [R,S,D], key / value [R,S',D] sharded
    along R (batch x heads), the mask replicated."""
    count = _global_count(query, local_operands, replicas, group)

    def op(q, k, v):
        return ops.sparse_attention(q, k, v, row_indices, row_offsets, column_indices, scale)

    return replica_parallel(op, [(query, True), (key, True), (value, True)], count, group,
                            gather_output, gather_mode, 1, local_operands)
