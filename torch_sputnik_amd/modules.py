"""nn.Modules mirroring the reference's ``SparseLinear``
(modules/sparse_linear.py:69-89) and ``SparseAttention``
(modules/sparse_attention.py:38-128): same constructor arguments, attributes,
forward signatures and output layouts, so user code can switch imports.

Kept quirks (drop-in): ``SparseLinear.weight`` / ``bias`` are uninitialised
``torch.empty`` parameters, ``bias`` is never applied, the user must call
``setup_sparse_tensors()``; the topology tensors are plain attributes (not
buffers); ``SparseAttention.forward`` ignores its ``mask`` argument and uses
the fixed random mask drawn at construction.
Differences: the attention mask is created on the device given (default: the
current GPU) instead of a hard ``.cuda()``; ``differentiable_softmax=True``
opts in to a softmax with a gradient.
"""
import copy
import math

import torch
import torch.nn as nn

from . import functional, ops
from .functional import (GroupProjectionFunction, HalfSparseLinearFunction, SparseAttentionFunction,
                         Sddmm, SparseLinearFunction, SparseSoftmax, Spmm)
from .topology import dense_to_sparse, generate_mask


class SparseLinear(nn.Module):
    """y[B, out, seq] = W_csr[out, in] @ x[B, seq, in]^T  (note the layout)."""

    def __init__(self, input_features, output_features):
        super().__init__()
        self.input_features = input_features
        self.output_features = output_features
        self.weight = nn.Parameter(torch.empty(output_features, input_features))
        self.bias = nn.Parameter(torch.empty(output_features))

    def setup_sparse_tensors(self):
        values, row_indices, row_offsets, column_indices = dense_to_sparse(self.weight)
        self.values = nn.Parameter(values)
        self.row_indices = row_indices
        self.row_offsets = row_offsets
        self.column_indices = column_indices
        # The module owns this pattern and never writes to it: transposed topology
        # and kernel plans are cached for as long as these tensors live.  (Replace
        # the pattern by calling setup_sparse_tensors() again -- do not write into
        # the index tensors through .data or a raw pointer: functional.py.)
        functional.register_static_topology(row_indices, row_offsets, column_indices)

    def forward(self, x):
        # [B, S, in] -> the k-major operand [B, in, S] of left_spmm: the reference's
        # `x.transpose(1, 2).contiguous()` (modules/sparse_linear.py:89) as one tiled
        # kernel (same values, same layout)
        if x.dtype in (torch.float16, torch.bfloat16) and x.dim() == 3:
            # half-precision activations stay in half: the k-major operand is one tiled
            # pass without widening, the product reads it as it is (at layer densities on
            # the matrix cores, csrc/spmm_mfma.hip), and under autograd it is what the
            # backward pass keeps
            if torch.is_grad_enabled() and (x.requires_grad or self.values.requires_grad):
                return HalfSparseLinearFunction.apply(
                    self.output_features, self.input_features, self.values, self.row_indices,
                    self.row_offsets, self.column_indices, x)
            values = self.values.detach()
            if x.is_cuda and ops.half_linear_supported(self.output_features, self.input_features, x.size(1),
                                                       x.size(0), self.column_indices.numel(), values.dtype,
                                                       x.dtype):
                image = ops.half_linear_image(self.output_features, self.input_features, values,
                                              self.row_offsets, self.column_indices, x.dtype)
                return ops.half_linear_forward(self.output_features, image, values.dtype, x)
            return functional._linear(self.output_features, self.input_features, values,
                                      self.row_indices, self.row_offsets, self.column_indices,
                                      ops.transpose_last2(x))
        return self.project(functional._to_operand(x))

    def project(self, dense, split_rows=0, dense_blocks=0):
        """``W @ dense`` for an operand that is already k-major: [B, in, S] ->
        [B, out, S].  (SparseAttention chains its layout passes and calls this.)
        ``split_rows = d``: the product comes back head split, [B * out/d, S, d],
        written in that order by the kernel (no layout pass of its own).
        ``dense_blocks = d``: `dense` is given as [B * in/d, d, S] (merged heads, the
        same memory); see functional.SparseLinearFunction."""
        needs_grad = torch.is_grad_enabled() and (dense.requires_grad or self.values.requires_grad)
        if not needs_grad:
            # forward only: no autograd node.  (The weight's pattern is static: its
            # topology pre-pass comes from the plan cache, functional.PlanCache,
            # which is keyed on the identity and version of the index tensors.)
            if dense_blocks:
                dense = dense.reshape(-1, self.input_features, dense.size(-1))
            return functional._linear(self.output_features, self.input_features,
                                      self.values.detach(), self.row_indices, self.row_offsets,
                                      self.column_indices, dense, split_rows)
        return SparseLinearFunction.apply(
            self.output_features, self.input_features, self.values, self.row_indices,
            self.row_offsets, self.column_indices, dense, split_rows, dense_blocks)


def get_clones(module, num_of_deep_copies):
    return nn.ModuleList([copy.deepcopy(module) for _ in range(num_of_deep_copies)])


class SparseAttention(nn.Module):
    """Multi-head attention whose score matrix only exists at the nonzeros of a
    fixed random mask: SDDMM -> sparse softmax -> SpMM, with four SparseLinear
    projections."""

    def __init__(self, num_heads, embedding_size, max_sequence_length=512, device=None,
                 sparsity=0.9, mask_generator=None, differentiable_softmax=False,
                 fused_inference=True, low_memory_training=False, fused_training=None):
        super().__init__()
        assert embedding_size % num_heads == 0, \
            "Model dimension must be divisible by the number of heads."
        self.head_dim = embedding_size // num_heads
        self.num_heads = num_heads
        self.linears = get_clones(SparseLinear(embedding_size, embedding_size), 4)

        self.m = max_sequence_length
        self.n = max_sequence_length
        device = torch.device("cuda") if device is None else device
        self.mask2d = generate_mask(self.m, self.n, device, sparsity=sparsity,
                                    generator=mask_generator)
        _, self.row_indices, self.row_offsets, self.column_indices = dense_to_sparse(self.mask2d)
        functional.register_static_topology(self.row_indices, self.row_offsets, self.column_indices)

        self.sddmm = Sddmm.apply
        self.spmm = Spmm.apply
        self.differentiable_softmax = differentiable_softmax
        # forward-only calls (no gradient wanted) take the one-kernel attention
        self.fused_inference = fused_inference
        # Training that keeps NOTHING of size [B*H, nnz] between forward and backward: the
        # one-kernel forward, and a backward that recomputes scores and weights before it
        # runs the separate operators (and, unlike the reference's raw softmax call, the
        # gradient reaches Q and K).  A MEMORY option, not a speed one: measured at config 3
        # it is slower than the separate operators that keep their weights (0.75 against
        # 0.66 ms, round 4) -- hence the name; `fused_training` (rounds 1-3) is kept as an
        # alias of the same flag.
        self.low_memory_training = bool(low_memory_training if fused_training is None
                                        else fused_training)
        # The three input projections on side streams (their grids are small).  Off:
        # measured at config 3, the event traffic costs more than the overlap gives
        # (forward 0.253 ms on one stream, 0.28-0.30 ms on three; fwd+bwd 1.04 / 1.11).
        self.parallel_projections = False

    @property
    def fused_training(self):   # the flag's name in rounds 1-3
        return self.low_memory_training

    @fused_training.setter
    def fused_training(self, value):
        self.low_memory_training = bool(value)

    def attention(self, query, key, value, mask):
        """[B, H, S, D] operands, as modules/sparse_attention.py:66-82."""
        return self._attention3d(self.four_d_to_three_d(query), self.four_d_to_three_d(key),
                                 self.four_d_to_three_d(value))

    def _attention3d(self, q3d, k3d, v3d, merged=False):
        """-> [B*H, S, D]; ``merged``: transposed, [B*H, D, S] (= [B, E, S], the k-major
        operand of the output projection), written that way by the last kernel where
        it can (the SpMM of the separate-operator path), else by a layout pass."""
        scale = 1.0 / math.sqrt(self.head_dim)
        needs_grad = torch.is_grad_enabled() and (
            q3d.requires_grad or k3d.requires_grad or v3d.requires_grad)
        if self.fused_inference and not needs_grad:
            out = functional._attention(q3d, k3d, v3d, self.row_indices, self.row_offsets,
                                        self.column_indices, scale)
            return functional.transpose_last2(out) if merged else out
        if self.low_memory_training:
            out = SparseAttentionFunction.apply(q3d, k3d, v3d, self.row_indices,
                                                self.row_offsets, self.column_indices, scale)
            return functional.transpose_last2(out) if merged else out

        # [B*H, nnz]: scores only at the mask's nonzeros
        ours = getattr(self.sddmm, "__self__", None) is Sddmm and \
            getattr(self.spmm, "__self__", None) is Spmm
        if ours:   # gradients of q, k, v leave the kernels in the projections' layout
            scores = Sddmm.apply(self.m, self.n, self.row_indices, self.row_offsets,
                                 self.column_indices, q3d, k3d, merged)
        else:
            scores = self.sddmm(self.m, self.n, self.row_indices, self.row_offsets,
                                self.column_indices, q3d, k3d)
        # the reference divides the scores in a pass of its own
        # (modules/sparse_attention.py:72); here the softmax kernel applies it
        softmax = (SparseSoftmax.apply if self.differentiable_softmax
                   else ops.sparse_softmax_scaled)
        attention_weights = softmax(scores, self.row_indices, self.row_offsets,
                                    self.column_indices, scale)
        # [B*H, S, D] ([B*H, D, S] when merged)
        if merged and ours:
            return Spmm.apply(self.m, self.n, attention_weights, self.row_indices,
                              self.row_offsets, self.column_indices, v3d, True, True)
        out = self.spmm(self.m, self.n, attention_weights, self.row_indices, self.row_offsets,
                        self.column_indices, v3d)
        return functional.transpose_last2(out) if merged else out

    @staticmethod
    def four_d_to_three_d(tensor):
        n, c, h, w = tensor.size()
        return tensor.reshape(n * c, h, w)

    def forward(self, query, key, value, mask=None):
        """Same values and layouts as modules/sparse_attention.py:105-128 ([B, S, E] in,
        [B, S, E] out as a transposed view of [B, E, S]); the layout passes are chained:

          reference, per projection      here
          x^T copy (SparseLinear)        transpose_last2(x)            [B, E, S]
          y^T copy                       (none)
          head-split copy                transpose_last2(y as [B*H, D, S])  -> [B*H, S, D]
          context: head-merge copy       transpose_last2(ctx)          -> [B*H, D, S] = [B, E, S],
          + x^T copy of the last layer   which IS the last layer's k-major operand

        i.e. 7 passes (5 when query, key and value are one tensor) instead of 11, each
        one tiled kernel.  (`parallel_projections` runs the three input projections on
        side streams; measured slower at config 3, off by default.)"""
        batch_size, seq = query.size(0), query.size(1)
        heads, dim = self.num_heads, self.head_dim
        inputs = (query, key, value)
        if query is key and key is value:
            shared = functional._to_operand(query)
            operands = (shared, shared, shared)
        else:
            operands = tuple(functional._to_operand(x) for x in inputs)

        def head_split(projected):   # [B, H*D, S] -> [B*H, S, D]
            return functional.transpose_last2(projected.reshape(batch_size * heads, dim, seq))

        if query is key and key is value and not self.parallel_projections:
            # self-attention: the three projections of the one input in ONE launch
            # (one copy of the input's panel per workgroup), each stored head split
            q3d, k3d, v3d = self._project_group(self.linears[:3], operands[0], dim)
        elif not (query.is_cuda and self.parallel_projections):
            # (the projection kernel stores its product head split: [B*H, S, D])
            q3d, k3d, v3d = (net.project(d, split_rows=dim)
                             for net, d in zip(self.linears, operands))
        else:
            # (under autograd the backward of each projection runs on the stream its
            # forward ran on, so the three weight / input gradients overlap as well)
            main = torch.cuda.current_stream(query.device)
            results = [None, None, None]
            results[0] = head_split(self.linears[0].project(operands[0]))
            for i in (1, 2):
                side = self._side_stream(i, query.device)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    results[i] = head_split(self.linears[i].project(operands[i]))
                    results[i].record_stream(main)
            for i in (1, 2):
                main.wait_stream(self._side_stream(i, query.device))
            q3d, k3d, v3d = results

        context = self._attention3d(q3d, k3d, v3d, merged=True)         # [B*H, D, S]
        # = [B, E, S], the k-major operand of the output projection (handed over in the
        # per-head shape so that its gradient can come back in the layout the
        # attention's backward wants, without a pass of its own)
        return self.linears[-1].project(context, dense_blocks=dim).transpose(1, 2)

    @staticmethod
    def _project_group(nets, dense, split_rows):
        first = nets[0]
        m, k = first.output_features, first.input_features
        needs_grad = torch.is_grad_enabled() and (
            dense.requires_grad or any(net.values.requires_grad for net in nets))
        if not needs_grad:
            return ops.left_spmm_group(m, k, [net.values.detach() for net in nets],
                                       [net.row_indices for net in nets],
                                       [net.row_offsets for net in nets],
                                       [net.column_indices for net in nets], dense, split_rows)
        flat = []
        for net in nets:
            flat += [net.values, net.row_indices, net.row_offsets, net.column_indices]
        return GroupProjectionFunction.apply(m, k, split_rows, dense, *flat)

    def _side_stream(self, index, device):
        streams = self.__dict__.setdefault("_streams", {})
        key = (index, str(device))
        if key not in streams:
            streams[key] = torch.cuda.Stream(device=device)
        return streams[key]
