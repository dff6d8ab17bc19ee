#!/usr/bin/env python3
"""Generates tests/golden/*.npz -- run HERE (the reference is not on the GPU box).

    python oracle/make_golden.py            # needs /root/reference

What comes from the reference itself (imported, never copied):
  * the input distribution: tests/connectors.py ``Uniform`` and
    tests/initializers.py ``Uniform`` under ``np.random.seed`` (the reference's
    scripts seed nothing; the seeds are ours and stored in each file);
  * the expected outputs of the autograd layer: modules/spmm.py ``Spmm``,
    modules/sddmm.py ``Sddmm`` and modules/sparse_linear.py ``SparseLinear``
    run forward+backward on CPU, with this repo's ``torch_sputnik`` package
    answering their ``import torch_sputnik`` (CPU kernels = the numpy oracle,
    oracle/torch_cpu_backend.py).  That is also the drop-in check: the
    reference's modules run unchanged against this package's operator surface.
Expected op outputs are the DENSE definitions the reference's test scripts
compare against (tests/test_spmm.py:9-10, tests/test_sddmm.py:8-13,
tests/test_softmax.py:9-22), evaluated in float64, cross-checked here against
the oracle before anything is written.

The reference cannot produce CUDA outputs in this image (no nvcc, empty
sputnik submodule), so these fixtures pin the mathematical contract, not bit
patterns of the CUDA kernels.
"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = os.environ.get("SPUTNIK_REFERENCE", "/root/reference")
GOLDEN = os.path.join(REPO, "tests", "golden")

sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REFERENCE, "tests"))
sys.path.insert(0, REFERENCE)

import connectors  # noqa: E402  (reference: tests/connectors.py)
import initializers  # noqa: E402  (reference: tests/initializers.py)

from oracle import sputnik_oracle as O  # noqa: E402
from oracle import torch_cpu_backend  # noqa: E402

RTOL = 1e-9


def _close(a, b, what, rtol=RTOL):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    err = np.max(np.abs(a - b) / (1.0 + np.abs(b))) if a.size else 0.0
    assert err < rtol, f"{what}: oracle vs dense definition differ by {err}"


def _close32(a, b, what):
    """For results that passed through the reference modules' float32 tensors."""
    _close(a, b, what, rtol=2e-6)


def _save(name, **arrays):
    path = os.path.join(GOLDEN, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"wrote {path} ({os.path.getsize(path)} bytes)")


def _f32(x):
    return np.ascontiguousarray(x, dtype=np.float32)


def spmm_case(name, seed, m, k, n, sparsity, replicas=None):
    """tests/test_spmm.py:13-34 (2-D) and tests/test_spmm_3d.py:14-40 (3-D)."""
    np.random.seed(seed)
    connector = connectors.Uniform(sparsity, round_to=4)
    initializer = initializers.Uniform()
    if replicas is None:
        lhs = _f32(connector(initializer([m, k])))
        rhs = _f32(initializer([k, n]))
        values, row_indices, row_offsets, column_indices = O.dense_to_csr(lhs)
        expected = O.dense_spmm(lhs, rhs)
    else:
        mask = connector(initializer([m, k]))
        mask[mask != 0] = 1.0
        lhs = _f32(np.expand_dims(mask, 0) * initializer([replicas, m, k]))
        rhs = _f32(initializer([replicas, k, n]))
        _, row_indices, row_offsets, column_indices = O.dense_to_csr(mask)
        values = np.stack([lhs[r][mask != 0] for r in range(replicas)])
        expected = O.dense_spmm(lhs, rhs)
    got = O.spmm(m, k, values, row_indices, row_offsets, column_indices, rhs)
    _close(got, expected, name)
    _save(name, seed=seed, m=m, k=k, n=n, values=_f32(values), row_indices=row_indices,
          row_offsets=row_offsets, column_indices=column_indices, dense=rhs, expected=expected)


def sddmm_case(name, seed, m, k, n, sparsity, replicas=None):
    """tests/test_sddmm.py:16-32 (2-D, sparsity 0.0) and tests/test_sddmm_3d.py:17-38."""
    np.random.seed(seed)
    connector = connectors.Uniform(sparsity)
    initializer = initializers.Uniform()
    lead = [] if replicas is None else [replicas]
    lhs = _f32(initializer(lead + [m, k]))
    rhs = _f32(initializer(lead + [n, k]))
    mask = connector(np.ones([m, n]))
    _, row_indices, row_offsets, column_indices = O.dense_to_csr(mask)
    dense = O.dense_sddmm(mask, lhs, rhs)
    expected = dense[..., mask != 0]
    got = O.sddmm(m, n, row_indices, row_offsets, column_indices, lhs, rhs)
    _close(got, expected, name)
    _save(name, seed=seed, m=m, k=k, n=n, row_indices=row_indices, row_offsets=row_offsets,
          column_indices=column_indices, lhs=lhs, rhs=rhs, expected=expected)


def softmax_case(name, seed, m, n, sparsity):
    """tests/test_softmax.py:25-31 with the dense definition of :9-22."""
    np.random.seed(seed)
    matrix = _f32(connectors.Uniform(sparsity)(initializers.Uniform()([m, n])))
    values, row_indices, row_offsets, column_indices = O.dense_to_csr(matrix)
    expected = O.dense_softmax(matrix)[matrix != 0]
    got = O.sparse_softmax(values, row_indices, row_offsets, column_indices)
    # rows with no stored entry do not exist in the sparse result
    _close(got, expected, name)
    _save(name, seed=seed, m=m, n=n, values=values, row_indices=row_indices,
          row_offsets=row_offsets, column_indices=column_indices, expected=expected)


def transpose_case(name, seed, m, n, sparsity, zero_first_row=False):
    """tests/test_transpose.py:27-30 (4x4, row 0 zeroed) and a random case.
    Expected = CSR of the dense transpose (rows ascending inside each output row)."""
    np.random.seed(seed)
    matrix = _f32(initializers.Uniform()([m, n]))
    if sparsity > 0:
        matrix = _f32(connectors.Uniform(sparsity)(matrix))
    if zero_first_row:
        matrix[0, :] = 0
    values, _, row_offsets, column_indices = O.dense_to_csr(matrix)
    values_t, _, row_offsets_t, column_indices_t = O.dense_to_csr(matrix.T.copy())
    got = O.csr_transpose(m, n, values, row_offsets, column_indices)
    assert np.array_equal(got[0], values_t) and np.array_equal(got[1], row_offsets_t)
    assert np.array_equal(got[2], column_indices_t)
    _save(name, seed=seed, m=m, n=n, values=values, row_offsets=row_offsets,
          column_indices=column_indices, values_t=values_t, row_offsets_t=row_offsets_t,
          column_indices_t=column_indices_t)


def autograd_cases():
    """Forward + backward of the reference's own Python layer on CPU."""
    torch_cpu_backend.install()
    from modules.spmm import Spmm  # reference: modules/spmm.py
    from modules.sddmm import Sddmm  # reference: modules/sddmm.py
    from modules.sparse_linear import SparseLinear  # reference: modules/sparse_linear.py

    # --- Spmm.apply, 2-D -------------------------------------------------
    seed = 4101
    np.random.seed(seed)
    m, k, n = 24, 20, 12
    weight = _f32(connectors.Uniform(0.7, round_to=4)(initializers.Uniform()([m, k])))
    dense = _f32(initializers.Uniform()([k, n]))
    grad_out = _f32(initializers.Uniform(-1.0, 1.0)([m, n]))
    values, row_indices, row_offsets, column_indices = O.dense_to_csr(weight)
    v = torch.from_numpy(values).requires_grad_(True)
    d = torch.from_numpy(dense).requires_grad_(True)
    topo = [torch.from_numpy(x) for x in (row_indices, row_offsets, column_indices)]
    out = Spmm.apply(m, k, v, *topo, d)
    out.backward(torch.from_numpy(grad_out))
    wd = torch.from_numpy(weight).double().requires_grad_(True)
    dd = torch.from_numpy(dense).double().requires_grad_(True)
    (wd @ dd).backward(torch.from_numpy(grad_out).double())
    _close32(out.detach().numpy(), (wd @ dd).detach().numpy(), "Spmm fwd")
    _close32(d.grad.numpy(), dd.grad.numpy(), "Spmm grad_dense")
    _close32(v.grad.numpy(), wd.grad.numpy()[weight != 0], "Spmm grad_values")
    _save("autograd_spmm", seed=seed, m=m, k=k, n=n, values=values, row_indices=row_indices,
          row_offsets=row_offsets, column_indices=column_indices, dense=dense, grad_out=grad_out,
          out=(wd @ dd).detach().numpy(), grad_values=wd.grad.numpy()[weight != 0],
          grad_dense=dd.grad.numpy())

    # --- Sddmm.apply, 2-D ------------------------------------------------
    seed = 4102
    np.random.seed(seed)
    m, k, n = 20, 16, 28
    mask = connectors.Uniform(0.75)(np.ones([m, n]))
    lhs = _f32(initializers.Uniform()([m, k]))
    rhs = _f32(initializers.Uniform()([n, k]))
    _, row_indices, row_offsets, column_indices = O.dense_to_csr(mask)
    nnz = column_indices.shape[0]
    grad_out = _f32(initializers.Uniform(-1.0, 1.0)([nnz]))
    l = torch.from_numpy(lhs).requires_grad_(True)
    r = torch.from_numpy(rhs).requires_grad_(True)
    topo = [torch.from_numpy(x) for x in (row_indices, row_offsets, column_indices)]
    out = Sddmm.apply(m, n, *topo, l, r)
    out.backward(torch.from_numpy(grad_out))
    ld = torch.from_numpy(lhs).double().requires_grad_(True)
    rd = torch.from_numpy(rhs).double().requires_grad_(True)
    dense_out = (ld @ rd.t())[torch.from_numpy(mask != 0)]
    dense_out.backward(torch.from_numpy(grad_out).double())
    _close32(out.detach().numpy(), dense_out.detach().numpy(), "Sddmm fwd")
    _close32(l.grad.numpy(), ld.grad.numpy(), "Sddmm grad_lhs")
    _close32(r.grad.numpy(), rd.grad.numpy(), "Sddmm grad_rhs")
    _save("autograd_sddmm", seed=seed, m=m, k=k, n=n, row_indices=row_indices,
          row_offsets=row_offsets, column_indices=column_indices, lhs=lhs, rhs=rhs,
          grad_out=grad_out, out=dense_out.detach().numpy(), grad_lhs=ld.grad.numpy(),
          grad_rhs=rd.grad.numpy())

    # --- SparseLinear (left_spmm) fwd/bwd: tests/test_linear_3d.py:105-136 shapes
    seed = 4103
    np.random.seed(seed)
    torch.manual_seed(0)  # tests/test_linear.py:6
    batch, out_f, in_f, seq = 3, 256, 128, 72
    weight = _f32(connectors.Uniform(0.9, round_to=4)(initializers.Uniform(-1.0, 1.0)([out_f, in_f])))
    x = _f32(initializers.Uniform(-1.0, 1.0)([batch, seq, in_f]))
    grad_out = _f32(initializers.Uniform(-1.0, 1.0)([batch, out_f, seq]))
    layer = SparseLinear(in_f, out_f)
    with torch.no_grad():
        layer.weight.copy_(torch.from_numpy(weight))
    layer.setup_sparse_tensors()
    xt = torch.from_numpy(x).requires_grad_(True)
    y = layer(xt)  # [batch, out_f, seq]
    y.backward(torch.from_numpy(grad_out))
    wd = torch.from_numpy(weight).double().requires_grad_(True)
    xd = torch.from_numpy(x).double().requires_grad_(True)
    yd = torch.matmul(xd, wd.t()).transpose(1, 2)
    yd.backward(torch.from_numpy(grad_out).double())
    _close32(y.detach().numpy(), yd.detach().numpy(), "SparseLinear fwd")
    _close32(xt.grad.numpy(), xd.grad.numpy(), "SparseLinear grad_x")
    _close32(layer.values.grad.numpy(), wd.grad.numpy()[weight != 0], "SparseLinear grad_values")
    _save("autograd_sparse_linear", seed=seed, batch=batch, out_features=out_f, in_features=in_f,
          seq=seq, weight=weight, x=x, grad_out=grad_out, y=yd.detach().numpy(),
          grad_x=xd.grad.numpy(), grad_values=wd.grad.numpy()[weight != 0])


def spmm_bias_cases():
    """tests/test_spmm_bias_relu.py:19-45: m,k,n = 72,64,72, sparsity 0, bias = ones(m),
    expected ``dense_result + 1``; plus a signed case that makes the ReLU visible."""
    seed = 5101
    np.random.seed(seed)
    m, k, n = 72, 64, 72
    lhs = _f32(connectors.Uniform(0.0, round_to=4)(initializers.Uniform()([m, k])))
    rhs = _f32(initializers.Uniform()([k, n]))
    values, row_indices, row_offsets, column_indices = O.dense_to_csr(lhs)
    bias = np.ones(m, dtype=np.float32)
    expected = O.dense_spmm(lhs, rhs) + 1.0
    _close(O.spmm_bias(m, k, values, row_indices, row_offsets, column_indices, bias, rhs),
           expected, "spmm_bias")
    _save("spmm_bias_72x64x72", seed=seed, m=m, k=k, n=n, values=values, row_indices=row_indices,
          row_offsets=row_offsets, column_indices=column_indices, dense=rhs, bias=bias,
          relu=0, expected=expected)

    seed = 5102
    np.random.seed(seed)
    m, k, n = 40, 48, 36
    lhs = _f32(connectors.Uniform(0.8, round_to=4)(initializers.Uniform(-1.0, 1.0)([m, k])))
    rhs = _f32(initializers.Uniform(-1.0, 1.0)([k, n]))
    bias = _f32(initializers.Uniform(-0.5, 0.5)([m]))
    values, row_indices, row_offsets, column_indices = O.dense_to_csr(lhs)
    expected = np.maximum(O.dense_spmm(lhs, rhs) + bias.astype(np.float64)[:, None], 0.0)
    _close(O.spmm_bias(m, k, values, row_indices, row_offsets, column_indices, bias, rhs,
                       relu=True), expected, "spmm_bias_relu")
    _save("spmm_bias_relu_40x48x36", seed=seed, m=m, k=k, n=n, values=values,
          row_indices=row_indices, row_offsets=row_offsets, column_indices=column_indices,
          dense=rhs, bias=bias, relu=1, expected=expected)


def softmax_backward_case():
    """Gradient of the masked dense softmax of tests/test_softmax.py:9-22 (here with a
    scale), from dense float64 autograd."""
    seed = 5201
    np.random.seed(seed)
    m, n, scale = 48, 40, 0.125
    mask = connectors.Uniform(0.7)(np.ones([m, n])) != 0
    mask[5] = False  # an empty row
    x = _f32(initializers.Uniform(-4.0, 4.0)([m, n]))
    grad = _f32(initializers.Uniform(-1.0, 1.0)([m, n]))
    _, row_indices, row_offsets, column_indices = O.dense_to_csr(mask.astype(np.float32))
    xd = torch.from_numpy(x).double().requires_grad_(True)
    masked = (xd * scale).masked_fill(~torch.from_numpy(mask), float("-inf"))
    y = torch.softmax(masked, dim=-1)
    y = torch.where(torch.from_numpy(mask), y, torch.zeros_like(y))  # empty row: nan -> 0
    (y * torch.from_numpy(grad).double())[torch.from_numpy(mask)].sum().backward()
    y_sparse = y.detach().numpy()[mask]
    grad_sparse = grad[mask]
    expected = np.nan_to_num(xd.grad.numpy())[mask]
    _close(O.sparse_softmax_scaled(x[mask], row_indices, row_offsets, column_indices, scale),
           y_sparse, "softmax scaled")
    _close(O.sparse_softmax_backward(y_sparse, grad_sparse, row_offsets, scale), expected,
           "softmax backward")
    _save("softmax_backward_48x40", seed=seed, m=m, n=n, scale=scale, values=x[mask],
          row_indices=row_indices, row_offsets=row_offsets, column_indices=column_indices,
          softmax_out=y_sparse, grad_out=grad_sparse, grad_values=expected)


def many_mask_cases():
    """The many-mask family through the reference's own sketches:
    tests/transformer/functions.py ``Spmm`` / ``Sddmm`` (forward + backward) and
    the attention chain of tests/test_attention_many_masks.py:107-150, with the
    topology built by tests/transformer/utils.py ``dense_to_sparse_3d``.  Expected
    values are dense float64 autograd per batch element."""
    torch_cpu_backend.install()
    sys.path.insert(0, os.path.join(REFERENCE, "tests", "transformer"))
    import functions as ref_functions  # reference: tests/transformer/functions.py
    import utils as ref_utils  # reference: tests/transformer/utils.py

    seed = 6101
    np.random.seed(seed)
    b, heads, s, hn = 3, 2, 24, 8
    replicas = b * heads
    # different sparsity per batch element, as tests/test_attention_many_masks.py:26-36
    masks = np.stack([connectors.Uniform(sp)(np.ones([s, s])) for sp in (0.2, 0.5, 0.8)])
    masks = masks.astype(np.float32)
    mask_t = torch.from_numpy(masks)
    _, row_indices, row_offsets, column_indices, nnzs = ref_utils.dense_to_sparse_3d(mask_t)
    ri, ro, ci, nn = O.dense_to_csr_many_mask(masks)
    assert np.array_equal(row_offsets.numpy().reshape(-1), ro)
    assert np.array_equal(column_indices.numpy(), ci) and np.array_equal(nnzs.numpy(), nn)
    # (row_indices: torch.argsort is not stable, ties may order differently; any order is valid)
    width = int(nn.max())

    q = _f32(initializers.Uniform(-1.0, 1.0)([replicas, s, hn]))
    kk = _f32(initializers.Uniform(-1.0, 1.0)([replicas, s, hn]))
    v = _f32(initializers.Uniform(-1.0, 1.0)([replicas, s, hn]))
    grad_ctx = _f32(initializers.Uniform(-1.0, 1.0)([replicas, s, hn]))
    scale = 1.0 / np.sqrt(hn)

    # --- reference sketches: Sddmm and Spmm with many masks, forward + backward ------
    qt = torch.from_numpy(q).requires_grad_(True)
    kt = torch.from_numpy(kk).requires_grad_(True)
    scores = ref_functions.Sddmm.apply(b, s, s, nnzs, row_indices, row_offsets, column_indices,
                                       qt, kt)
    grad_scores = _f32(initializers.Uniform(-1.0, 1.0)(list(scores.shape)))
    for i in range(b):  # nothing flows through the padding
        grad_scores[i * heads:(i + 1) * heads, int(nn[i]):] = 0.0
    scores.backward(torch.from_numpy(grad_scores))

    rep_mask = np.repeat(masks != 0, heads, axis=0)  # [R, s, s]
    qd = torch.from_numpy(q).double().requires_grad_(True)
    kd = torch.from_numpy(kk).double().requires_grad_(True)
    dense_scores = torch.matmul(qd, kd.transpose(1, 2))
    expected_scores = np.zeros((replicas, width))
    dense_grad = np.zeros((replicas, s, s))
    for r in range(replicas):
        cnt = int(nn[r // heads])
        expected_scores[r, :cnt] = dense_scores[r].detach().numpy()[rep_mask[r]]
        dense_grad[r][rep_mask[r]] = grad_scores[r, :cnt]
    dense_scores.backward(torch.from_numpy(dense_grad))
    _close32(scores.detach().numpy(), expected_scores, "Sddmm many-mask fwd")
    _close32(qt.grad.numpy(), qd.grad.numpy(), "Sddmm many-mask grad_lhs")
    _close32(kt.grad.numpy(), kd.grad.numpy(), "Sddmm many-mask grad_rhs")

    weights_in = np.zeros((replicas, width), dtype=np.float32)
    for r in range(replicas):
        cnt = int(nn[r // heads])
        weights_in[r, :cnt] = initializers.Uniform(-1.0, 1.0)([cnt])
    wt = torch.from_numpy(weights_in).requires_grad_(True)
    vt = torch.from_numpy(v).requires_grad_(True)
    ctx = ref_functions.Spmm.apply(b, s, s, nnzs, wt, row_indices, row_offsets, column_indices, vt)
    ctx.backward(torch.from_numpy(grad_ctx))
    w_dense = np.zeros((replicas, s, s))
    for r in range(replicas):
        w_dense[r][rep_mask[r]] = weights_in[r, :int(nn[r // heads])]
    wd = torch.from_numpy(w_dense).requires_grad_(True)
    vd = torch.from_numpy(v).double().requires_grad_(True)
    ctx_d = torch.matmul(wd, vd)
    ctx_d.backward(torch.from_numpy(grad_ctx).double())
    expected_gw = np.zeros((replicas, width))
    for r in range(replicas):
        expected_gw[r, :int(nn[r // heads])] = wd.grad.numpy()[r][rep_mask[r]]
    _close32(ctx.detach().numpy(), ctx_d.detach().numpy(), "Spmm many-mask fwd")
    _close32(vt.grad.numpy(), vd.grad.numpy(), "Spmm many-mask grad_dense")
    _close32(wt.grad.numpy(), expected_gw, "Spmm many-mask grad_values")

    # --- attention chain, tests/test_attention_many_masks.py:107-150 ---------------
    import torch_sputnik
    sc = torch_sputnik.sddmm_many_mask(b, s, s, nnzs, row_indices, row_offsets, column_indices,
                                       torch.from_numpy(q), torch.from_numpy(kk)) / np.sqrt(hn)
    we = torch_sputnik.sparse_softmax_many_mask(b, s, nnzs, sc, row_indices, row_offsets,
                                                column_indices)
    rep = torch_sputnik.spmm_many_mask(b, s, s, nnzs, we, row_indices, row_offsets,
                                       column_indices, torch.from_numpy(v))
    logits = torch.matmul(torch.from_numpy(q).double(), torch.from_numpy(kk).double().transpose(1, 2))
    logits = (logits / np.sqrt(hn)).masked_fill(~torch.from_numpy(rep_mask), float("-inf"))
    dense_rep = torch.matmul(torch.softmax(logits, dim=-1), torch.from_numpy(v).double())
    _close32(rep.numpy(), dense_rep.numpy(), "many-mask attention")

    _save("many_mask_b3_h2_s24", seed=seed, b=b, heads=heads, s=s, hn=hn, masks=masks,
          row_indices=ri, row_offsets=ro, column_indices=ci, nnzs=nn, q=q, k=kk, v=v,
          scale=scale, scores=expected_scores, grad_scores=grad_scores, grad_q=qd.grad.numpy(),
          grad_k=kd.grad.numpy(), weights=weights_in, context=ctx_d.detach().numpy(),
          grad_context=grad_ctx, grad_weights=expected_gw, grad_v=vd.grad.numpy(),
          attention=dense_rep.numpy())


def main():
    os.makedirs(GOLDEN, exist_ok=True)
    # BASELINE.json config 1: 64^3, density 0.5
    spmm_case("spmm_c1_64_d050", 1101, 64, 64, 64, 0.5)
    spmm_case("spmm_2d_72x64x72", 1102, 72, 64, 72, 0.9)
    spmm_case("spmm_3d_r8_72x64x72", 1103, 72, 64, 72, 0.9, replicas=8)
    sddmm_case("sddmm_2d_dense_mask", 2101, 72, 64, 72, 0.0)
    sddmm_case("sddmm_3d_r8", 2102, 72, 64, 72, 0.9, replicas=8)
    softmax_case("softmax_72x72", 3101, 72, 72, 0.9)
    transpose_case("transpose_4x4_row0_zero", 3201, 4, 4, 0.0, zero_first_row=True)
    transpose_case("transpose_72x64", 3202, 72, 64, 0.8)
    autograd_cases()
    spmm_bias_cases()
    softmax_backward_case()
    many_mask_cases()


if __name__ == "__main__":
    main()
