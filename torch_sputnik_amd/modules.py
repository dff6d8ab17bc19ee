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
from .functional import SparseAttentionFunction, Sddmm, SparseLinearFunction, SparseSoftmax, Spmm
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

    def forward(self, x):
        dense = x.transpose(1, 2).contiguous()
        needs_grad = torch.is_grad_enabled() and (x.requires_grad or self.values.requires_grad)
        if not needs_grad:
            # forward only: no autograd node.  (The weight's pattern is static: its
            # topology pre-pass comes from the plan cache, functional.PlanCache,
            # which is keyed on the identity and version of the index tensors.)
            return functional._spmm(self.output_features, self.input_features,
                                    self.values.detach(), self.row_indices, self.row_offsets,
                                    self.column_indices, dense, left=True)
        return SparseLinearFunction.apply(
            self.output_features, self.input_features, self.values, self.row_indices,
            self.row_offsets, self.column_indices, dense)


def get_clones(module, num_of_deep_copies):
    return nn.ModuleList([copy.deepcopy(module) for _ in range(num_of_deep_copies)])


class SparseAttention(nn.Module):
    """Multi-head attention whose score matrix only exists at the nonzeros of a
    fixed random mask: SDDMM -> sparse softmax -> SpMM, with four SparseLinear
    projections."""

    def __init__(self, num_heads, embedding_size, max_sequence_length=512, device=None,
                 sparsity=0.9, mask_generator=None, differentiable_softmax=False,
                 fused_inference=True, fused_training=False):
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

        self.sddmm = Sddmm.apply
        self.spmm = Spmm.apply
        self.differentiable_softmax = differentiable_softmax
        # forward-only calls (no gradient wanted) take the one-kernel attention
        self.fused_inference = fused_inference
        # training through the fused forward: the weights are recomputed in the
        # backward instead of being kept (and, unlike the reference's raw softmax
        # call, the gradient reaches Q and K)
        self.fused_training = fused_training

    def attention(self, query, key, value, mask):
        q3d = self.four_d_to_three_d(query)
        k3d = self.four_d_to_three_d(key)
        v3d = self.four_d_to_three_d(value)

        scale = 1.0 / math.sqrt(self.head_dim)
        needs_grad = torch.is_grad_enabled() and (
            q3d.requires_grad or k3d.requires_grad or v3d.requires_grad)
        if self.fused_inference and not needs_grad:
            return functional._attention(q3d, k3d, v3d, self.row_indices, self.row_offsets,
                                         self.column_indices, scale)
        if self.fused_training:
            return SparseAttentionFunction.apply(q3d, k3d, v3d, self.row_indices,
                                                 self.row_offsets, self.column_indices, scale)

        # [B*H, nnz]: scores only at the mask's nonzeros
        scores = self.sddmm(self.m, self.n, self.row_indices, self.row_offsets,
                            self.column_indices, q3d, k3d)
        # the reference divides the scores in a pass of its own
        # (modules/sparse_attention.py:72); here the softmax kernel applies it
        softmax = (SparseSoftmax.apply if self.differentiable_softmax
                   else ops.sparse_softmax_scaled)
        attention_weights = softmax(scores, self.row_indices, self.row_offsets,
                                    self.column_indices, scale)
        # [B*H, S, D]
        return self.spmm(self.m, self.n, attention_weights, self.row_indices, self.row_offsets,
                         self.column_indices, v3d)

    @staticmethod
    def four_d_to_three_d(tensor):
        n, c, h, w = tensor.size()
        return tensor.reshape(n * c, h, w)

    def forward(self, query, key, value, mask=None):
        batch_size = query.size(0)
        # SparseLinear returns [B, E, S]; bring it to [B, H, S, D]
        query, key, value = [
            net(x).transpose(1, 2).contiguous()
            .view(batch_size, -1, self.num_heads, self.head_dim).transpose(1, 2)
            for net, x in zip(self.linears, (query, key, value))]
        b, h, s, d = value.size()

        context = self.attention(query, key, value, mask).reshape(b, h, s, d)
        context = context.transpose(1, 2).contiguous().reshape(
            batch_size, -1, self.num_heads * self.head_dim)
        return self.linears[-1](context).transpose(1, 2)
