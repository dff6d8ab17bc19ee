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


def replica_parallel(op, replicated_args, replicas, group=None, gather_output=True,
                     gather_mode="collective", overlap_chunks=1, local_operands=False):
    """Generic driver: ``op(*local_args) -> [L, ...]`` is run on this rank's
    block of every tensor in `replicated_args` (tensors replicated along dim 0;
    anything else is passed through), then optionally all-gathered.

    ``local_operands=True`` is the shard-at-origin form: the tensors passed ARE
    this rank's blocks (``local_range(replicas, world, rank)`` replicas each), so
    no rank ever holds another rank's operands; `replicas` is the global count
    (default: local count x world size)."""
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
    chunks = max(1, min(overlap_chunks, count)) if count else 1
    if chunks == 1:
        local_out = op(*block(start, stop))
        out, pending = all_gather_replicas(local_out, replicas, group, gather_mode)
        _wait_all(pending)
        return out
    # Pipelined: exchange chunk i while chunk i+1 is being computed.  Chunk
    # boundaries are the same on every rank only for even shards.
    if replicas % world != 0:
        raise ValueError("overlap_chunks > 1 needs replicas divisible by the world size")
    out = None
    pending = []
    step = (count + chunks - 1) // chunks
    for c0 in range(0, count, step):
        c1 = min(count, c0 + step)
        part = op(*block(start + c0, start + c1))
        if out is None:
            out = part.new_empty((replicas,) + tuple(part.shape[1:]))
        # a chunk is a strided set of slices of `out`: gather it peer to peer
        out[start + c0:start + c1].copy_(part)
        ops_list = []
        for peer in range(world):
            if peer == rank:
                continue
            pa, _ = local_range(replicas, world, peer)
            gpeer = dist.get_global_rank(group, peer) if group else peer
            ops_list.append(dist.P2POp(dist.isend, part, gpeer, group=group))
            ops_list.append(dist.P2POp(dist.irecv, out[pa + c0:pa + c1], gpeer, group=group))
        if ops_list:
            pending += dist.batch_isend_irecv(ops_list)
    _wait_all(pending)
    return out


def _global_count(tensor, local_operands, replicas, group):
    """Global replica count: given, or the operand's dim 0 (x world size when the
    operand is this rank's block; that shortcut needs even shards)."""
    if replicas is not None:
        return int(replicas)
    return tensor.size(0) * (_world(group)[0] if local_operands else 1)


def spmm(m, k, values, row_indices, row_offsets, column_indices, dense, group=None,
         gather_output=True, gather_mode="collective", overlap_chunks=1, local_operands=False,
         replicas=None):
    """Replica-parallel batched SpMM.  `values` [R,nnz] and `dense` [R,k,n] are the
    global operands (every rank passes the same tensors and uses its block) or,
    with ``local_operands=True``, this rank's blocks only -- the shard-at-origin
    form, in which no GPU ever holds the other ranks' B (config 4: 1.07 GB of the
    8.6 GB).  Each rank computes its block of C and, with `gather_output`, every
    rank returns the whole [R,m,n]."""
    count = _global_count(dense, local_operands, replicas, group)

    def op(v, d):
        out = ops.spmm(m, k, v, row_indices, row_offsets, column_indices, d)
        return out.reshape((d.size(0), m, d.size(-1)))

    return replica_parallel(op, [(values, True), (dense, True)], count, group, gather_output,
                            gather_mode, overlap_chunks, local_operands)


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
