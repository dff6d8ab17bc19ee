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
    counter (bumped by every in-place write), size, device.  The caches below
    keep a reference to the keyed tensors, so an address cannot be recycled
    while its entry is alive."""
    return tuple((t.data_ptr(), t._version, t.numel(), t.dtype, str(t.device)) for t in tensors)


class _Lru:
    """Small bounded map (static topologies are few: a model's layers and masks)."""

    def __init__(self, capacity):
        self.capacity = capacity
        self._entries = collections.OrderedDict()

    def get(self, key):
        entry = self._entries.get(key)
        if entry is not None:
            self._entries.move_to_end(key)
        return entry

    def put(self, key, entry):
        self._entries[key] = entry
        while len(self._entries) > self.capacity:
            self._entries.popitem(last=False)
        return entry

    def clear(self):
        self._entries.clear()

    def __len__(self):
        return len(self._entries)


class TransposeCache:
    """Memoises the transposed topology (and the value permutation) of static
    CSR patterns, keyed by the identity and version of the index tensors.  The
    reference re-runs csr_transpose + diffsort in every backward although the
    topology never changes (modules/spmm.py:59-64, modules/sddmm.py:60-65,
    modules/sparse_linear.py:52-57); with the cache a backward is one gather of
    the values through the stored permutation."""

    def __init__(self, capacity=128):
        self._entries = _Lru(capacity)

    def lookup(self, m, n, row_offsets, column_indices, probe_values):
        key = (m, n) + _identity(row_offsets, column_indices)
        entry = self._entries.get(key)
        if entry is None:
            _, row_offsets_t, column_indices_t, permutation = ops.csr_transpose_with_permutation(
                m, n, probe_values.detach().reshape(-1, probe_values.shape[-1])[0].contiguous(),
                row_offsets, column_indices)
            entry = self._entries.put(key, (diffsort(row_offsets_t), row_offsets_t,
                                            column_indices_t, permutation.contiguous(),
                                            row_offsets, column_indices))
        return entry[:4]

    def clear(self):
        self._entries.clear()


class PlanCache:
    """Memoises the topology-only pre-pass of the LDS-tiled kernels
    (ops.spmm_plan / ops.sddmm_plan) per static topology and operand width, so
    that training steps launch kernels only.  Keyed like TransposeCache."""

    def __init__(self, capacity=256):
        self._entries = _Lru(capacity)

    def spmm(self, m, k, n, row_indices, row_offsets, column_indices):
        key = ("spmm", m, k, n) + _identity(row_indices, row_offsets, column_indices)
        entry = self._entries.get(key)
        if entry is None:
            plan = ops.spmm_plan(m, k, n, row_indices, row_offsets, column_indices)
            entry = self._entries.put(key, (plan, row_indices, row_offsets, column_indices))
        return entry[0]

    def sddmm(self, m, n, k, row_indices, row_offsets, column_indices, summed=False):
        key = ("sddmm_sum" if summed else "sddmm", m, n, k) + _identity(
            row_indices, row_offsets, column_indices)
        entry = self._entries.get(key)
        if entry is None:
            plan = (ops.sddmm_sum_plan if summed else ops.sddmm_plan)(
                m, n, k, row_indices, row_offsets, column_indices)
            entry = self._entries.put(key, (plan, row_indices, row_offsets, column_indices))
        return entry[0]

    def attention(self, m, n, d, row_indices, row_offsets, column_indices):
        key = ("attention", m, n, d) + _identity(row_indices, row_offsets, column_indices)
        entry = self._entries.get(key)
        if entry is None:
            plan = ops.sparse_attention_plan(m, n, d, row_indices, row_offsets, column_indices)
            entry = self._entries.put(key, (plan, row_indices, row_offsets, column_indices))
        return entry[0]

    def clear(self):
        self._entries.clear()


# Both caches are ON by default: the topologies of a sparse model are static,
# and the keys (address + version counter + size of every index tensor, with the
# tensors kept alive by the entry) change whenever a topology is replaced or
# written to.  `enable_transpose_cache(False)` / `enable_plan_cache(False)` give
# the reference's per-call behaviour.
TRANSPOSE_CACHE_DEFAULT = True
PLAN_CACHE_DEFAULT = True
_cache = TransposeCache() if TRANSPOSE_CACHE_DEFAULT else None
_plans = PlanCache() if PLAN_CACHE_DEFAULT else None


def enable_transpose_cache(enabled=True):
    """Cache transposed topologies across backward calls (default: on)."""
    global _cache
    _cache = TransposeCache() if enabled else None
    return _cache


def enable_plan_cache(enabled=True):
    """Cache the kernels' topology pre-pass across calls (default: on)."""
    global _plans
    _plans = PlanCache() if enabled else None
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
    if _cache is not None:
        row_indices_t, row_offsets_t, column_indices_t, perm = _cache.lookup(
            m, n, row_offsets, column_indices, values)
        return _permute_cached(values, perm), row_indices_t, row_offsets_t, column_indices_t
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
    if _cache is None:
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
    if _plans is None:
        return (ops.left_spmm if left else ops.spmm)(m, k, values, row_indices, row_offsets,
                                                     column_indices, dense)
    plan = _plans.spmm(m, k, dense.size(-1), row_indices, row_offsets, column_indices)
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
    if _plans is None:
        return (ops.sddmm_sum if sum_replicas else ops.sddmm)(
            m, n, row_indices, row_offsets, column_indices, lhs_matrix, rhs_matrix)
    plan = _plans.sddmm(m, n, lhs_matrix.size(-1), row_indices, row_offsets, column_indices,
                        summed=sum_replicas)
    return (ops.sddmm_sum_planned if sum_replicas else ops.sddmm_planned)(
        m, n, row_indices, row_offsets, column_indices, lhs_matrix, rhs_matrix, plan)


def _attention(query, key, value, row_indices, row_offsets, column_indices, scale):
    """Fused attention forward, through the cached plan of the mask when enabled."""
    if _plans is None:
        return ops.sparse_attention(query, key, value, row_indices, row_offsets, column_indices,
                                    scale)
    plan = _plans.attention(query.size(-2), key.size(-2), query.size(-1), row_indices,
                            row_offsets, column_indices)
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
        for w, g in zip(live, grad_ys):
            if ctx.needs_input_grad[4 + 4 * w]:
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
                    row_offsets, column_indices)
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
