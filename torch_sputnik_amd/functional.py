"""autograd.Functions over the five ops, mirroring the reference's Python
layer: ``Spmm`` (modules/spmm.py:8-74), ``Sddmm`` (modules/sddmm.py:9-74),
``SparseLinearFunction`` (modules/sparse_linear.py:18-67).  Same ``apply``
signatures, same gradients returned in the same positions.

Beyond the reference:
  * 3-D (batched) backward works, because csr_transpose here accepts [R,nnz]
    values (the reference's is 1-D only, src/transpose_cuda.cu:50);
  * ``SparseSoftmax`` gives sparse_softmax a backward (the reference calls the
    raw op, modules/sparse_attention.py:76, which cuts the gradient);
  * the transposed topology of a static mask can be cached
    (``TransposeCache``), so a backward is one gather instead of a transpose.
"""
import collections

import torch

from . import ops
from .topology import diffsort


def _identity(*tensors):
    """Key part naming these tensors' storage and contents: address, version
    counter (bumped by every in-place write through autograd-visible ops), size,
    dtype, device."""
    return tuple((t.data_ptr(), t._version, t.numel(), t.dtype, str(t.device)) for t in tensors)


def _storage_key(t):
    return (t.data_ptr(), t.numel(), t.dtype, str(t.device))


def _tensor_bytes(obj):
    if isinstance(obj, torch.Tensor):
        return obj.numel() * obj.element_size()
    if isinstance(obj, (tuple, list)):
        return sum(_tensor_bytes(o) for o in obj)
    return 0


# ---------------------------------------------------------------------------
# Static topologies.
#
# The caches below serve, by default, only index tensors that their OWNER has
# declared static (`register_static_topology`: SparseLinear.setup_sparse_tensors
# and SparseAttention do, for the patterns they create and never write to).  An
# arbitrary tensor handed to Spmm / Sddmm gets the reference's per-call behaviour
# (modules/spmm.py:59-64): nothing is remembered about it.
#
# A registration lives as long as the registered tensor object (weakref): when
# the owner drops it, the registration and every cache entry made for it go too,
# so a recycled address can never hit an old entry and no topology is pinned in
# memory by the cache alone.
#
# What the key cannot see: writes that bypass the version counter --
# `index.data.copy_(...)`, `set_`, a kernel or C-ABI call writing through the raw
# pointer.  Whoever does that to a registered topology calls
# `unregister_static_topology` / `clear_caches()` (or simply builds new tensors).
# ---------------------------------------------------------------------------
_static = {}        # storage key -> number of live registrations
_listeners = []     # caches to tell when a registration dies


def _drop_registration(key):
    left = _static.get(key, 0) - 1
    if left > 0:
        _static[key] = left
        return
    _static.pop(key, None)
    for cache in list(_listeners):
        cache.forget(key)


def register_static_topology(*tensors):
    """Declares index tensors (row_indices / row_offsets / column_indices) as
    static: transposed topologies and kernel plans derived from them are cached
    until the tensors are written to (in place) or die."""
    import weakref
    for t in tensors:
        key = _storage_key(t)
        _static[key] = _static.get(key, 0) + 1
        weakref.finalize(t, _drop_registration, key)
    return tensors


def unregister_static_topology(*tensors):
    for t in tensors:
        key = _storage_key(t)
        _static.pop(key, None)
        for cache in list(_listeners):
            cache.forget(key)


def _is_static(*tensors):
    return all(_storage_key(t) in _static for t in tensors)


class _Lru:
    """Bounded map: at most `capacity` entries and `max_bytes` of device memory
    (static topologies are few -- a model's layers and masks -- but a 4096^2
    pattern at density 0.1 comes to 27 MB of transposed indices and plan)."""

    def __init__(self, capacity, max_bytes=1 << 30):
        self.capacity = capacity
        self.max_bytes = max_bytes
        self.bytes = 0
        self._entries = collections.OrderedDict()   # key -> (entry, bytes, storage keys it was made for)

    def get(self, key):
        slot = self._entries.get(key)
        if slot is None:
            return None
        self._entries.move_to_end(key)
        return slot[0]

    def put(self, key, entry, made_for=()):
        size = _tensor_bytes(entry)
        old = self._entries.pop(key, None)
        if old is not None:
            self.bytes -= old[1]
        self._entries[key] = (entry, size, tuple(made_for))
        self.bytes += size
        while len(self._entries) > 1 and (len(self._entries) > self.capacity or self.bytes > self.max_bytes):
            _, (_, dropped, _) = self._entries.popitem(last=False)
            self.bytes -= dropped
        return entry

    def forget(self, storage_key):
        for key in [k for k, (_, _, made_for) in self._entries.items() if storage_key in made_for]:
            self.bytes -= self._entries.pop(key)[1]

    def clear(self):
        self._entries.clear()
        self.bytes = 0

    def __len__(self):
        return len(self._entries)


class _TopologyCache:
    """Common part of the two caches: whom they serve.  scope "static": index
    tensors registered with `register_static_topology` only (entries hold no
    reference to them: they die with their owner); scope "all": any tensor --
    the caller's promise that the index tensors it passes are not written to
    behind the version counter; entries then keep the tensors alive so that an
    address cannot be recycled under a live entry."""

    def __init__(self, capacity, scope):
        self.scope = scope
        self._entries = _Lru(capacity)
        _listeners.append(self)

    def serves(self, *tensors):
        return self.scope == "all" or _is_static(*tensors)

    def _keep(self, *tensors):
        return tensors if self.scope == "all" else ()

    def forget(self, storage_key):
        self._entries.forget(storage_key)

    def clear(self):
        self._entries.clear()

    def __len__(self):
        return len(self._entries)


class TransposeCache(_TopologyCache):
    """Memoises the transposed topology (and the value permutation) of static
    CSR patterns.  The reference re-runs csr_transpose + diffsort in every
    backward although the topology never changes (modules/spmm.py:59-64,
    modules/sddmm.py:60-65, modules/sparse_linear.py:52-57); with the cache a
    backward is one gather of the values through the stored permutation."""

    def __init__(self, capacity=128, scope="static"):
        super().__init__(capacity, scope)

    def lookup(self, m, n, row_offsets, column_indices, probe_values):
        """(row_indices_t, row_offsets_t, column_indices_t, permutation), cached
        when this cache serves the pattern, computed for this call otherwise."""
        cached = self.serves(row_offsets, column_indices)
        key = (m, n) + _identity(row_offsets, column_indices)
        entry = self._entries.get(key) if cached else None
        if entry is None:
            _, row_offsets_t, column_indices_t, permutation = ops.csr_transpose_with_permutation(
                m, n, probe_values.detach().reshape(-1, probe_values.shape[-1])[0].contiguous(),
                row_offsets, column_indices, checked=cached)   # the wait only for a result that is kept
            entry = (diffsort(row_offsets_t), row_offsets_t, column_indices_t,
                     permutation.contiguous()) + self._keep(row_offsets, column_indices)
            if cached:
                # the derived pattern is as static as the one it was made from: plans
                # for it (the transposed products of every backward) are cached too,
                # for as long as this entry holds its tensors
                register_static_topology(*entry[:3])
                self._entries.put(key, entry, (_storage_key(row_offsets), _storage_key(column_indices)))
        return entry[:4]


class PlanCache(_TopologyCache):
    """Memoises the topology-only pre-pass of the LDS-tiled kernels
    (ops.spmm_plan / ops.sddmm_plan) per static topology and operand width, so
    that training steps launch kernels only.  Every method returns None for a
    pattern this cache does not serve (the caller then takes the per-call op)."""

    def __init__(self, capacity=256, scope="static"):
        super().__init__(capacity, scope)

    def _plan(self, kind, shape, make, *index):
        if not self.serves(*index):
            return None
        key = (kind,) + shape + _identity(*index)
        entry = self._entries.get(key)
        if entry is None:
            entry = self._entries.put(key, (make(),) + self._keep(*index),
                                      tuple(_storage_key(t) for t in index))
        return entry[0]

    def spmm(self, m, k, n, row_indices, row_offsets, column_indices):
        return self._plan("spmm", (m, k, n),
                          lambda: ops.spmm_plan(m, k, n, row_indices, row_offsets, column_indices),
                          row_indices, row_offsets, column_indices)

    def sddmm(self, m, n, k, row_indices, row_offsets, column_indices, summed=False):
        return self._plan("sddmm_sum" if summed else "sddmm", (m, n, k),
                          lambda: (ops.sddmm_sum_plan if summed else ops.sddmm_plan)(
                              m, n, k, row_indices, row_offsets, column_indices),
                          row_indices, row_offsets, column_indices)

    def half_linear(self, m, k, row_offsets, column_indices):
        """Plan of the half-storage layer's weight gradient (ops.half_linear_plan)."""
        return self._plan("half_linear", (m, k),
                          lambda: ops.half_linear_plan(m, k, row_offsets, column_indices),
                          row_offsets, column_indices)

    def attention(self, m, n, d, row_indices, row_offsets, column_indices):
        return self._plan("attention", (m, n, d),
                          lambda: ops.sparse_attention_plan(m, n, d, row_indices, row_offsets,
                                                            column_indices),
                          row_indices, row_offsets, column_indices)


# Both caches exist by default with scope "static": they serve the patterns that
# modules register (see above) and nothing else.  `enable_*_cache(True)` widens
# a cache to every tensor (scope "all": the caller vouches for the tensors),
# `enable_*_cache(False)` gives the reference's per-call behaviour everywhere.
TRANSPOSE_CACHE_DEFAULT = "static"
PLAN_CACHE_DEFAULT = "static"


def _make_cache(cls, enabled):
    if not enabled:
        return None
    return cls(scope="all" if enabled is True or enabled == "all" else "static")


_cache = _make_cache(TransposeCache, TRANSPOSE_CACHE_DEFAULT)
_plans = _make_cache(PlanCache, PLAN_CACHE_DEFAULT)


def enable_transpose_cache(enabled=True):
    """False: off; "static" (the default): registered topologies only; True: all tensors."""
    global _cache
    if _cache is not None and _cache in _listeners:
        _listeners.remove(_cache)
    _cache = _make_cache(TransposeCache, enabled)
    return _cache


def enable_plan_cache(enabled=True):
    """False: off; "static" (the default): registered topologies only; True: all tensors."""
    global _plans
    if _plans is not None and _plans in _listeners:
        _listeners.remove(_plans)
    _plans = _make_cache(PlanCache, enabled)
    return _plans


def clear_caches():
    for cache in (_cache, _plans):
        if cache is not None:
            cache.clear()


def _contiguous(x):
    """``x.contiguous()``; a transposed view of a contiguous tensor (the gradient
    that reaches a module which returned ``out.transpose(1, 2)``,
    modules/sparse_attention.py:126) goes through the tiled transpose kernel
    instead of a strided elementwise copy (7 vs 23 us at config 3)."""
    if x.is_contiguous():
        return x
    if x.is_cuda and x.dim() >= 2 and x.dtype == torch.float32:
        view = x.transpose(-1, -2)
        if view.is_contiguous():
            return ops.transpose_last2(view)
    return x.contiguous()


def _as_transposed(x):
    """``x.transpose(-1, -2)`` as a contiguous tensor.  Free when `x` is itself the
    transposed view of a contiguous tensor -- how the Functions below hand each
    other gradients that a kernel's store phase already wrote in the layout the
    receiver needs (`transposed_grads`) -- and one tiled-transpose launch otherwise."""
    view = x.transpose(-1, -2)
    if view.is_contiguous():
        return view
    return ops.transpose_last2(_contiguous(x))


# From this many values on (several rows of them) a cached permutation goes
# through the LDS-banded kernel; its two index lists are made once per cached
# permutation and kept on the permutation tensor (the cache entry owns it).
BANDED_PERMUTE_FROM = 1 << 21


def _permute_cached(values, perm):
    """values[..., perm] for a permutation held by the transpose cache."""
    if values.dim() == 2 and values.size(0) > 1 and values.numel() >= BANDED_PERMUTE_FROM \
            and values.is_cuda:
        lists = getattr(perm, "_sputnik_banded_lists", None)
        if lists is None:
            lists = ops.banded_lists(perm)
            perm._sputnik_banded_lists = lists
        return ops.permute_last_banded(values, *lists)
    return ops.permute_last(values, perm)


def _transpose(m, n, values, row_offsets, column_indices):
    """(values_t, row_indices_t, row_offsets_t, column_indices_t)."""
    values = values.contiguous()
    if _cache is not None and _cache.serves(row_offsets, column_indices):
        row_indices_t, row_offsets_t, column_indices_t, perm = _cache.lookup(
            m, n, row_offsets, column_indices, values)
        return _permute_cached(values, perm), row_indices_t, row_offsets_t, column_indices_t
    # a pattern no cache serves: the reference's per-call transpose (asynchronous, one pass)
    values_t, row_offsets_t, column_indices_t = ops.csr_transpose(
        m, n, values, row_offsets, column_indices)
    return values_t, diffsort(row_offsets_t), row_offsets_t, column_indices_t


def _spmm_transposed(m, n, values, row_offsets, column_indices, dense, left=False,
                     block_rows=0):
    """(A^T) @ dense for the m x n CSR matrix A: with the transposed-topology cache
    the values stay in A's order and the kernel gathers them through the cached
    permutation (ops.spmm_permuted); without it the reference's per-call
    csr_transpose (modules/spmm.py:59-62).  ``block_rows``: the product comes
    back as ops.spmm_transposed_out stores it, [R * n / block_rows, width, block_rows]."""
    if _cache is None or not _cache.serves(row_offsets, column_indices):
        values_t, row_indices_t, row_offsets_t, column_indices_t = _transpose(
            m, n, values, row_offsets, column_indices)
        perm = None
    else:
        values = values.contiguous()
        row_indices_t, row_offsets_t, column_indices_t, perm = _cache.lookup(
            m, n, row_offsets, column_indices, values)
        if ops.spmm_permuted_fused(n, m, dense.size(-1), perm.numel()):
            values_t = values          # gathered through `perm` inside the kernel
        else:
            values_t, perm = _permute_cached(values, perm), None
    plan = None if _plans is None else _plans.spmm(n, m, dense.size(-1), row_indices_t,
                                                   row_offsets_t, column_indices_t)
    if block_rows:
        return ops.spmm_transposed_out(n, m, values_t, row_indices_t, row_offsets_t,
                                       column_indices_t, dense, block_rows, permutation=perm,
                                       plan=plan, left=left)
    if perm is not None:
        return ops.spmm_permuted(n, m, values_t, perm, row_indices_t, row_offsets_t,
                                 column_indices_t, dense, plan, left=left)
    return _spmm(n, m, values_t, row_indices_t, row_offsets_t, column_indices_t, dense, left=left)


def _spmm(m, k, values, row_indices, row_offsets, column_indices, dense, left=False):
    """spmm / left_spmm, through the cached plan of the topology when enabled."""
    plan = None if _plans is None else _plans.spmm(m, k, dense.size(-1), row_indices, row_offsets,
                                                   column_indices)
    if plan is None:
        return (ops.left_spmm if left else ops.spmm)(m, k, values, row_indices, row_offsets,
                                                     column_indices, dense)
    return (ops.left_spmm_planned if left else ops.spmm_planned)(
        m, k, values, row_indices, row_offsets, column_indices, dense, plan)


def _linear(m, k, values, row_indices, row_offsets, column_indices, dense, split_rows=0):
    """left_spmm, optionally with the head-split store (ops.spmm_transposed_out)."""
    if not split_rows:
        return _spmm(m, k, values, row_indices, row_offsets, column_indices, dense, left=True)
    plan = None if _plans is None else _plans.spmm(m, k, dense.size(-1), row_indices,
                                                   row_offsets, column_indices)
    return ops.spmm_transposed_out(m, k, values, row_indices, row_offsets, column_indices, dense,
                                   block_rows=split_rows, plan=plan, left=True)


def _sddmm(m, n, row_indices, row_offsets, column_indices, lhs_matrix, rhs_matrix,
           sum_replicas=False):
    plan = None if _plans is None else _plans.sddmm(m, n, lhs_matrix.size(-1), row_indices,
                                                    row_offsets, column_indices, summed=sum_replicas)
    if plan is None:
        return (ops.sddmm_sum if sum_replicas else ops.sddmm)(
            m, n, row_indices, row_offsets, column_indices, lhs_matrix, rhs_matrix)
    return (ops.sddmm_sum_planned if sum_replicas else ops.sddmm_planned)(
        m, n, row_indices, row_offsets, column_indices, lhs_matrix, rhs_matrix, plan)


def _attention(query, key, value, row_indices, row_offsets, column_indices, scale):
    """Fused attention forward, through the cached plan of the mask when enabled."""
    plan = None if _plans is None else _plans.attention(query.size(-2), key.size(-2), query.size(-1),
                                                        row_indices, row_offsets, column_indices)
    if plan is None:
        return ops.sparse_attention(query, key, value, row_indices, row_offsets, column_indices,
                                    scale)
    return ops.sparse_attention_planned(query, key, value, row_indices, row_offsets,
                                        column_indices, scale, plan)


class TransposeLast2(torch.autograd.Function):
    """``x.transpose(-1, -2).contiguous()`` as one tiled kernel (ops.transpose_last2):
    the layout pass of modules/sparse_linear.py:89 and
    modules/sparse_attention.py:108-126, optionally widening half-precision
    storage to float32 on the way.  Its gradient is the same operation (narrowing
    back to the input's storage type inside the pass)."""

    @staticmethod
    def forward(ctx, x, dtype=None):
        ctx.in_dtype = x.dtype
        return ops.transpose_last2(x, dtype)

    @staticmethod
    def backward(ctx, grad_output):
        return ops.transpose_last2(grad_output, ctx.in_dtype), None


def transpose_last2(x, dtype=None):
    """Differentiable ``x.transpose(-1, -2).contiguous()`` (``.to(dtype)``)."""
    if torch.is_grad_enabled() and x.requires_grad:
        return TransposeLast2.apply(x, dtype)
    return ops.transpose_last2(x, dtype)


def _to_operand(x):
    """[B, S, in] -> the k-major float32 operand [B, in, S] of left_spmm.  The
    operators compute and return float32 whatever the storage type
    (src/spmm_cuda.cu:42); half-precision activations are widened inside this
    pass instead of in one of their own."""
    return transpose_last2(x, torch.float32 if x.dtype in (torch.float16, torch.bfloat16) else None)


class Spmm(torch.autograd.Function):
    """sparse(values, CSR topology) @ dense.  ``apply(m, k, values, row_indices,
    row_offsets, column_indices, dense[, transposed_out[, transposed_grads]])``; with
    ``transposed_out`` the product comes back transposed, [R, n, m] (written in that
    order by the kernel's store phase, ops.spmm_transposed_out: SparseAttention's
    head merge).  ``transposed_grads``: the gradient of `dense` is handed back as
    the transposed VIEW of a buffer the kernel wrote transposed (same shape and
    values; free for a receiver that wants that layout, see `_as_transposed`)."""

    @staticmethod
    def forward(ctx, m, k, values, row_indices, row_offsets, column_indices, dense,
                transposed_out=False, transposed_grads=False):
        ctx.shape = (m, k)
        ctx.topology = (row_indices, row_offsets, column_indices)
        ctx.transposed_out = bool(transposed_out)
        ctx.transposed_grads = bool(transposed_grads)
        ctx.save_for_backward(values, dense)
        if transposed_out:
            plan = None if _plans is None else _plans.spmm(m, k, dense.size(-1), row_indices,
                                                           row_offsets, column_indices)
            return ops.spmm_transposed_out(m, k, values, row_indices, row_offsets,
                                           column_indices, dense, block_rows=m, plan=plan)
        return _spmm(m, k, values, row_indices, row_offsets, column_indices, dense)

    @staticmethod
    def backward(ctx, grad_output):
        m, k = ctx.shape
        row_indices, row_offsets, column_indices = ctx.topology
        values, dense = ctx.saved_tensors
        if ctx.transposed_out:   # [R, n, m] -> the product's own layout
            grad_output = _as_transposed(grad_output)
            if dense.dim() == 2:
                grad_output = grad_output[0]
        else:
            grad_output = _contiguous(grad_output)
        grad_values = grad_dense = None
        if ctx.needs_input_grad[2]:
            # dL/dA sampled at the pattern: <dC[i,:], B[j,:]>
            grad_values = _sddmm(m, k, row_indices, row_offsets, column_indices, grad_output,
                                 dense.contiguous())
        if ctx.needs_input_grad[6]:
            # dL/dB = A^T @ dC
            if ctx.transposed_grads and dense.dim() == 3:
                grad_dense = _spmm_transposed(m, k, values, row_offsets, column_indices,
                                              grad_output, block_rows=k).transpose(1, 2)
            else:
                grad_dense = _spmm_transposed(m, k, values, row_offsets, column_indices,
                                              grad_output)
        return None, None, grad_values, None, None, None, grad_dense, None, None


class Sddmm(torch.autograd.Function):
    """(lhs @ rhs^T) sampled at a CSR mask.  ``apply(m, n, row_indices,
    row_offsets, column_indices, lhs_matrix, rhs_matrix[, transposed_grads])``;
    ``transposed_grads`` as in `Spmm` (both gradients, 3-D operands)."""

    @staticmethod
    def forward(ctx, m, n, row_indices, row_offsets, column_indices, lhs_matrix, rhs_matrix,
                transposed_grads=False):
        ctx.shape = (m, n)
        ctx.topology = (row_indices, row_offsets, column_indices)
        ctx.transposed_grads = bool(transposed_grads) and lhs_matrix.dim() == 3
        ctx.save_for_backward(lhs_matrix, rhs_matrix)
        return _sddmm(m, n, row_indices, row_offsets, column_indices, lhs_matrix, rhs_matrix)

    @staticmethod
    def backward(ctx, grad_output):
        m, n = ctx.shape
        row_indices, row_offsets, column_indices = ctx.topology
        lhs_matrix, rhs_matrix = ctx.saved_tensors
        grad_output = _contiguous(grad_output)
        grad_lhs = grad_rhs = None
        if ctx.needs_input_grad[5]:
            # dL/dlhs = dS @ rhs, dS sparse with the mask's pattern
            rhs = rhs_matrix.contiguous()
            if ctx.transposed_grads:
                plan = None if _plans is None else _plans.spmm(m, n, rhs.size(-1), row_indices,
                                                               row_offsets, column_indices)
                grad_lhs = ops.spmm_transposed_out(m, n, grad_output, row_indices, row_offsets,
                                                   column_indices, rhs, block_rows=m,
                                                   plan=plan).transpose(1, 2)
            else:
                grad_lhs = _spmm(m, n, grad_output, row_indices, row_offsets, column_indices, rhs)
        if ctx.needs_input_grad[6]:
            # dL/drhs = dS^T @ lhs
            if ctx.transposed_grads:
                grad_rhs = _spmm_transposed(m, n, grad_output, row_offsets, column_indices,
                                            lhs_matrix.contiguous(), block_rows=n).transpose(1, 2)
            else:
                grad_rhs = _spmm_transposed(m, n, grad_output, row_offsets, column_indices,
                                            lhs_matrix.contiguous())
        return None, None, None, None, None, grad_lhs, grad_rhs, None


class SparseLinearFunction(torch.autograd.Function):
    """One sparse weight x a batch of dense matrices (left_spmm).  ``apply(m, k,
    values, row_indices, row_offsets, column_indices, dense[, split_rows[,
    dense_blocks]])`` with dense [B,k,n] -> [B,m,n]; with ``split_rows = d`` the
    product comes back head split, [B * m/d, n, d] (every block of d output rows
    transposed, written by the kernel's store phase:
    modules/sparse_attention.py:38-45,108-126).  ``dense_blocks = d``: `dense` is
    given as [B * k/d, d, n] (the same memory: merged heads) and its gradient is
    handed back in that shape as the transposed view of a head-split buffer the
    kernel wrote (see `Spmm`, ``transposed_grads``)."""

    @staticmethod
    def forward(ctx, m, k, values, row_indices, row_offsets, column_indices, dense,
                split_rows=0, dense_blocks=0):
        ctx.shape = (m, k)
        ctx.topology = (row_indices, row_offsets, column_indices)
        ctx.split_rows = int(split_rows)
        ctx.dense_blocks = int(dense_blocks)
        if dense_blocks:
            dense = dense.reshape(-1, k, dense.size(-1))
        ctx.save_for_backward(values, dense)
        return _linear(m, k, values, row_indices, row_offsets, column_indices, dense, split_rows)

    @staticmethod
    def backward(ctx, grad_output):
        m, k = ctx.shape
        row_indices, row_offsets, column_indices = ctx.topology
        values, dense = ctx.saved_tensors
        if ctx.split_rows:   # [B * m/d, n, d] -> [B, m, n]
            grad_output = _as_transposed(grad_output).reshape(-1, m, grad_output.size(-2))
        else:
            grad_output = _contiguous(grad_output)
        grad_values = grad_dense = None
        if ctx.needs_input_grad[2]:
            # the [B,nnz] products summed over B (what autograd makes of the
            # reference's result for the shared `values`), inside the call
            grad_values = _sddmm(m, k, row_indices, row_offsets, column_indices, grad_output,
                                 dense.contiguous(), sum_replicas=True)
        if ctx.needs_input_grad[6]:
            if ctx.dense_blocks:
                grad_dense = _spmm_transposed(m, k, values, row_offsets, column_indices,
                                              grad_output, left=True,
                                              block_rows=ctx.dense_blocks).transpose(1, 2)
            else:
                grad_dense = _spmm_transposed(m, k, values, row_offsets, column_indices,
                                              grad_output, left=True)
                if dense.dim() == 2:
                    grad_dense = grad_dense[0]
        return None, None, grad_values, None, None, None, grad_dense, None, None


class HalfSparseLinearFunction(torch.autograd.Function):
    """``SparseLinear.forward`` for activations stored in float16 / bfloat16 (BASELINE
    config 5; the reference knows float32 only, src/left_replicated_spmm.cu:34-38):
    ``apply(m, k, values, row_indices, row_offsets, column_indices, x)`` with x [B, S, in]
    -> [B, out, S] float32.

    At a layer's density and size all three products run on the matrix cores and read
    their operands as the caller has them -- NO layout pass (csrc/sparse_linear_half.hip):
    the weight as one densified image (made in the forward pass, kept for the backward),
    x [B, S, in], dy [B, out, S]; float32 values and the float32 dy enter as half planes,
    not rounded.  Elsewhere the typed operators with the layout passes of
    modules/sparse_linear.py:89 inside the Function (the half operand is what is kept)."""

    @staticmethod
    def forward(ctx, m, k, values, row_indices, row_offsets, column_indices, x):
        ctx.shape = (m, k)
        ctx.topology = (row_indices, row_offsets, column_indices)
        ctx.in_dtype = x.dtype
        batch, seq = x.size(0), x.size(1)
        ctx.tiles = x.is_cuda and ops.half_linear_supported(m, k, seq, batch, column_indices.numel(),
                                                            values.dtype, x.dtype)
        if ctx.tiles:
            x = x.contiguous()
            image = ops.half_linear_image(m, k, values, row_offsets, column_indices, x.dtype)
            ctx.save_for_backward(values, x, image)
            return ops.half_linear_forward(m, image, values.dtype, x)
        dense = ops.transpose_last2(x)
        ctx.save_for_backward(values, dense)
        return _linear(m, k, values, row_indices, row_offsets, column_indices, dense)

    @staticmethod
    def backward(ctx, grad_output):
        m, k = ctx.shape
        row_indices, row_offsets, column_indices = ctx.topology
        grad_output = _contiguous(grad_output)
        grad_values = grad_x = None
        if ctx.tiles:
            values, x, image = ctx.saved_tensors
            batch, seq = x.size(0), x.size(1)
            # dy as planes, once for both gradients (float32 dy), or as it is (dy in x's type)
            planes = grad_output.dtype == torch.float32
            grad = ops.half_planes(grad_output, x.dtype) if planes else grad_output.to(x.dtype)
            if ctx.needs_input_grad[2]:
                plan = None if _plans is None else _plans.half_linear(m, k, row_offsets, column_indices)
                grad_values = ops.half_linear_weight_gradient(m, row_offsets, column_indices, grad, planes,
                                                              x, plan)
            if ctx.needs_input_grad[6]:
                grad_x = ops.half_linear_input_gradient(m, k, grad, planes, image, values.dtype, x, batch, seq)
                if grad_x is None:   # (bfloat16 tiles with float32 values AND a float32 dy)
                    grad_dense = _spmm_transposed(m, k, values, row_offsets, column_indices, grad_output,
                                                  left=True)
                    grad_x = ops.transpose_last2(grad_dense, ctx.in_dtype)
            return None, None, grad_values, None, None, None, grad_x
        values, dense = ctx.saved_tensors
        if ctx.needs_input_grad[2]:
            grad_values = _sddmm(m, k, row_indices, row_offsets, column_indices, grad_output, dense,
                                 sum_replicas=True)
        if ctx.needs_input_grad[6]:
            grad_dense = None
            if grad_output.dim() == 3:
                # W^T dy as a dense contraction on half tiles where that route serves the
                # shape (the float32 gradient enters as half planes, not rounded)
                values_t, _, row_offsets_t, column_indices_t = _transpose(
                    m, k, values, row_offsets, column_indices)
                grad_dense = ops.left_spmm_half_tiles(k, m, values_t, row_offsets_t, column_indices_t,
                                                      grad_output, ctx.in_dtype)
            if grad_dense is None:
                grad_dense = _spmm_transposed(m, k, values, row_offsets, column_indices, grad_output,
                                              left=True)
            grad_x = ops.transpose_last2(grad_dense, ctx.in_dtype)
        return None, None, grad_values, None, None, None, grad_x


class GroupProjectionFunction(torch.autograd.Function):
    """Several SparseLinear weights of one shape applied to ONE input in one launch
    (ops.left_spmm_group): the q, k and v projections of a self-attention block,
    which the reference runs one after the other (modules/sparse_attention.py:108-110).
    ``apply(m, k, split_rows, dense, values_0, row_indices_0, row_offsets_0,
    column_indices_0, values_1, ...)`` -> one product per weight ([B, m, n], or head
    split [B * m/d, n, d] with ``split_rows = d``).  The backward runs one summed
    SDDMM per weight and ONE launch for the input gradient  sum_w W_w^T dY_w
    (ops.left_spmm_group_sum: accumulated in registers, no partial results)."""

    @staticmethod
    def forward(ctx, m, k, split_rows, dense, *flat):
        values, ris, ros, cis = flat[0::4], flat[1::4], flat[2::4], flat[3::4]
        ctx.shape = (m, k, int(split_rows))
        ctx.topologies = list(zip(ris, ros, cis))
        ctx.save_for_backward(dense, *values)
        return tuple(ops.left_spmm_group(m, k, values, ris, ros, cis, dense, split_rows))

    @staticmethod
    def backward(ctx, *grads):
        m, k, split_rows = ctx.shape
        dense, *values = ctx.saved_tensors
        n = dense.size(-1)
        live, grad_ys = [], []
        for w, g in enumerate(grads):
            if g is None:
                continue
            if split_rows:   # [B * m/d, n, d] -> [B, m, n]
                g = _as_transposed(g).reshape(-1, m, n)
            else:
                g = _contiguous(g)
            live.append(w)
            grad_ys.append(g)
        grad_values = [None] * len(values)
        wanted = [(w, g) for w, g in zip(live, grad_ys) if ctx.needs_input_grad[4 + 4 * w]]
        grouped = (_plans is not None and 2 <= len(wanted) <= 4 and dense.dim() == 3 and dense.is_cuda
                   and dense.dtype == torch.float32 and all(g.dtype == torch.float32 for _, g in wanted))
        if grouped:   # one call: every weight's partial sums added by ONE launch
            topo = [ctx.topologies[w] for w, _ in wanted]
            plans = [_plans.sddmm(m, k, dense.size(-1), *t, summed=True) for t in topo]
            grouped = all(p is not None for p in plans)
        if grouped:
            outs = ops.sddmm_sum_group_planned(m, k, [t[0] for t in topo], [t[1] for t in topo],
                                               [t[2] for t in topo], [g for _, g in wanted], dense, plans)
            for (w, _), out in zip(wanted, outs):
                grad_values[w] = out
        else:
            for w, g in wanted:
                ri, ro, ci = ctx.topologies[w]
                grad_values[w] = _sddmm(m, k, ri, ro, ci, g, dense, sum_replicas=True)
        grad_dense = None
        if ctx.needs_input_grad[3] and live:
            if _cache is not None:
                vals, perms, ris_t, ros_t, cis_t = [], [], [], [], []
                for w in live:
                    _, ro, ci = ctx.topologies[w]
                    ri_t, ro_t, ci_t, perm = _cache.lookup(m, k, ro, ci, values[w])
                    vals.append(values[w]); perms.append(perm)
                    ris_t.append(ri_t); ros_t.append(ro_t); cis_t.append(ci_t)
                grad_dense = ops.left_spmm_group_sum(k, m, vals, perms, ris_t, ros_t, cis_t,
                                                     grad_ys)
            else:
                for w, g in zip(live, grad_ys):
                    _, ro, ci = ctx.topologies[w]
                    part = _spmm_transposed(m, k, values[w], ro, ci, g, left=True)
                    grad_dense = part if grad_dense is None else grad_dense + part
        flat = []
        for gv in grad_values:
            flat += [gv, None, None, None]
        return (None, None, None, grad_dense, *flat)


class SparseSoftmax(torch.autograd.Function):
    """sparse_softmax with a gradient: dX = scale * Y * (dY - rowsum(dY * Y)), the
    row sums taken over the stored entries (extension).
    ``apply(values, row_indices, row_offsets, column_indices[, scale])`` computes
    softmax(scale * values)."""

    @staticmethod
    def forward(ctx, values, row_indices, row_offsets, column_indices, scale=1.0):
        if scale == 1.0:
            out = ops.sparse_softmax(values, row_indices, row_offsets, column_indices)
        else:
            out = ops.sparse_softmax_scaled(values, row_indices, row_offsets, column_indices,
                                            scale)
        ctx.scale = float(scale)
        ctx.save_for_backward(out, row_offsets)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        out, row_offsets = ctx.saved_tensors
        grad_values = ops.sparse_softmax_backward(out, grad_output.contiguous(), row_offsets,
                                                  ctx.scale)
        return grad_values, None, None, None, None


class SparseAttentionFunction(torch.autograd.Function):
    """softmax(scale * sddmm(q, k)) @ v with the ONE-kernel forward
    (ops.sparse_attention) and a backward built from the separate operators.
    Nothing of size [R, nnz] is kept between the passes: the backward recomputes
    the weights (sddmm + scaled softmax) and then runs the standard chain

        dV = P^T dO          dP = sddmm(dO, v)
        dS = softmax'(P, dP) (sparse_softmax_backward, carries the scale)
        dQ = dS k            dK = dS^T q

    on the mask and its transpose (one csr_transpose with permutation per call,
    or the cached transposed topology)."""

    @staticmethod
    def forward(ctx, query, key, value, row_indices, row_offsets, column_indices, scale):
        out = _attention(query, key, value, row_indices, row_offsets, column_indices, scale)
        ctx.scale = float(scale)
        ctx.save_for_backward(query, key, value, row_indices, row_offsets, column_indices)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        query, key, value, row_indices, row_offsets, column_indices = ctx.saved_tensors
        topo = (row_indices, row_offsets, column_indices)
        m, n = query.size(-2), key.size(-2)
        grad_output = _contiguous(grad_output)
        scores = _sddmm(m, n, *topo, query, key)
        weights = ops.sparse_softmax_scaled(scores, *topo, ctx.scale)
        grad_weights = _sddmm(m, n, *topo, grad_output, value)
        grad_scores = ops.sparse_softmax_backward(weights, grad_weights, row_offsets, ctx.scale)
        grad_query = grad_key = grad_value = None
        if ctx.needs_input_grad[0]:
            grad_query = _spmm(m, n, grad_scores, *topo, key)
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            if _cache is not None:
                row_indices_t, row_offsets_t, column_indices_t, perm = _cache.lookup(
                    m, n, row_offsets, column_indices, grad_scores)
            else:
                _, row_offsets_t, column_indices_t, perm = ops.csr_transpose_with_permutation(
                    m, n, grad_scores.reshape(-1, grad_scores.shape[-1])[0].contiguous(),
                    row_offsets, column_indices, checked=False)
                row_indices_t = diffsort(row_offsets_t)
            def transposed_product(values, dense):
                if ops.spmm_permuted_fused(n, m, dense.size(-1), perm.numel()):
                    return ops.spmm_permuted(n, m, values, perm, row_indices_t, row_offsets_t,
                                             column_indices_t, dense)
                values_t = (_permute_cached(values, perm) if _cache is not None
                            else ops.permute_last(values, perm))
                return _spmm(n, m, values_t, row_indices_t, row_offsets_t, column_indices_t, dense)

            if ctx.needs_input_grad[1]:
                grad_key = transposed_product(grad_scores, query)
            if ctx.needs_input_grad[2]:
                grad_value = transposed_product(weights, grad_output)
        return grad_query, grad_key, grad_value, None, None, None, None


# ---------------------------------------------------------------------------
# many-mask family: one mask per batch element, shared by its heads.  Same
# ``apply`` signatures and gradient positions as the reference's sketches in
# tests/transformer/functions.py (Spmm :5-69, CsrSoftmax :70-120, Sddmm :122-188).
# ---------------------------------------------------------------------------
def diffsort_many_mask(row_offsets, masks):
    """Per-mask ``diffsort`` of stacked / concatenated offsets
    (tests/transformer/utils.py:51-62) -> flat [masks * rows]."""
    per_mask = row_offsets.reshape(masks, -1)
    return torch.cat([diffsort(per_mask[i]) for i in range(masks)])


class SpmmManyMask(torch.autograd.Function):
    """tests/transformer/functions.py:5-69."""

    @staticmethod
    def forward(ctx, b, m, k, nonzeros, values, row_indices, row_offsets, column_indices, dense):
        ctx.dims = (b, m, k)
        ctx.nonzeros = nonzeros
        ctx.save_for_backward(values, row_indices, row_offsets, column_indices, dense)
        return ops.spmm_many_mask(b, m, k, nonzeros, values, row_indices, row_offsets,
                                  column_indices, dense)

    @staticmethod
    def backward(ctx, grad_output):
        b, m, k = ctx.dims
        nonzeros = ctx.nonzeros
        values, row_indices, row_offsets, column_indices, dense = ctx.saved_tensors
        grad_output = _contiguous(grad_output)
        grad_values = grad_dense = None
        if ctx.needs_input_grad[4]:
            grad_values = ops.sddmm_many_mask(b, m, k, nonzeros, row_indices, row_offsets,
                                              column_indices, grad_output, dense)
            if grad_values.shape[-1] != values.shape[-1]:  # values rows were padded
                grad_values = torch.nn.functional.pad(
                    grad_values, (0, values.shape[-1] - grad_values.shape[-1]))
        if ctx.needs_input_grad[8]:
            values_t, row_offsets_t, column_indices_t = ops.csr_transpose_many_mask(
                b, m, k, nonzeros, values.detach(), row_offsets, column_indices)
            row_indices_t = diffsort_many_mask(row_offsets_t, b)
            grad_dense = ops.spmm_many_mask(b, k, m, nonzeros, values_t, row_indices_t,
                                            row_offsets_t, column_indices_t, grad_output)
        return None, None, None, None, grad_values, None, None, None, grad_dense


class SddmmManyMask(torch.autograd.Function):
    """tests/transformer/functions.py:122-188."""

    @staticmethod
    def forward(ctx, b, m, n, nonzeros, row_indices, row_offsets, column_indices, lhs_matrix,
                rhs_matrix):
        ctx.dims = (b, m, n)
        ctx.nonzeros = nonzeros
        ctx.save_for_backward(row_indices, row_offsets, column_indices, lhs_matrix, rhs_matrix)
        return ops.sddmm_many_mask(b, m, n, nonzeros, row_indices, row_offsets, column_indices,
                                   lhs_matrix, rhs_matrix)

    @staticmethod
    def backward(ctx, grad_output):
        b, m, n = ctx.dims
        nonzeros = ctx.nonzeros
        row_indices, row_offsets, column_indices, lhs_matrix, rhs_matrix = ctx.saved_tensors
        grad_output = _contiguous(grad_output)
        grad_lhs = grad_rhs = None
        if ctx.needs_input_grad[7]:
            grad_lhs = ops.spmm_many_mask(b, m, n, nonzeros, grad_output, row_indices,
                                          row_offsets, column_indices, rhs_matrix)
        if ctx.needs_input_grad[8]:
            grad_t, row_offsets_t, column_indices_t = ops.csr_transpose_many_mask(
                b, m, n, nonzeros, grad_output, row_offsets, column_indices)
            row_indices_t = diffsort_many_mask(row_offsets_t, b)
            grad_rhs = ops.spmm_many_mask(b, n, m, nonzeros, grad_t, row_indices_t,
                                          row_offsets_t, column_indices_t, lhs_matrix)
        return None, None, None, None, None, None, None, grad_lhs, grad_rhs


class CsrSoftmaxManyMask(torch.autograd.Function):
    """tests/transformer/functions.py:70-120 with the softmax Jacobian in the
    backward (the sketch there returns ``s * (1 - s)`` of a dense softmax of the
    incoming gradient).  Optional trailing ``scale``."""

    @staticmethod
    def forward(ctx, b, m, nonzeros, scores, row_indices, row_offsets, column_indices,
                scale=1.0):
        out = ops.sparse_softmax_many_mask(b, m, nonzeros, scores, row_indices, row_offsets,
                                           column_indices, None if scale == 1.0 else scale)
        ctx.dims = (b, m, float(scale))
        ctx.nonzeros = nonzeros
        ctx.save_for_backward(out, row_offsets)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        b, m, scale = ctx.dims
        out, row_offsets = ctx.saved_tensors
        grad_scores = ops.sparse_softmax_backward_many_mask(
            b, m, ctx.nonzeros, out, grad_output.contiguous(), row_offsets, scale)
        return None, None, None, grad_scores, None, None, None, None
