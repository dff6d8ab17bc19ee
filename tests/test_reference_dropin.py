"""CPU, this container only: the reference's OWN Python layer (modules/spmm.py,
modules/sddmm.py, modules/sparse_linear.py), imported unchanged from
/root/reference, runs against this repo's ``torch_sputnik`` package.  That is
the drop-in claim: same five callables, same signatures, same return types.
Skipped where the reference checkout does not exist (the GPU box)."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch

from helpers import make_csr, rel_err

REFERENCE = os.environ.get("SPUTNIK_REFERENCE", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "modules")),
                                reason="reference checkout not present")


@pytest.fixture(scope="module")
def ref(cpu_ops):
    sys.path.insert(0, REFERENCE)
    try:
        mods = {name: importlib.import_module(f"modules.{name}")
                for name in ("spmm", "sddmm", "sparse_linear")}
    finally:
        sys.path.remove(REFERENCE)
    import torch_sputnik
    for mod in mods.values():
        assert mod.torch_sputnik is torch_sputnik       # their `import torch_sputnik` is ours
    return mods


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def test_reference_spmm_function(ref):
    dense_a, vals, ri, ro, ci = make_csr(18, 14, 0.6, seed=1, order="ascending")
    b = np.random.default_rng(2).uniform(-1, 1, (14, 9)).astype(np.float32)
    v, d = T(vals).requires_grad_(True), T(b).requires_grad_(True)
    out = ref["spmm"].Spmm.apply(18, 14, v, T(ri), T(ro), T(ci), d)
    out.square().sum().backward()
    ad = T(dense_a).double().requires_grad_(True)
    bd = T(b).double().requires_grad_(True)
    (ad @ bd).square().sum().backward()
    assert rel_err(out.detach().numpy(), (ad @ bd).detach().numpy()) < 2e-5
    assert rel_err(d.grad.numpy(), bd.grad.numpy()) < 2e-5
    assert rel_err(v.grad.numpy(), ad.grad.numpy()[dense_a != 0]) < 2e-5


def test_reference_sddmm_function(ref):
    mask, _, ri, ro, ci = make_csr(16, 20, 0.7, seed=3, round_to=1)
    rng = np.random.default_rng(4)
    lhs = rng.uniform(-1, 1, (16, 6)).astype(np.float32)
    rhs = rng.uniform(-1, 1, (20, 6)).astype(np.float32)
    l, r = T(lhs).requires_grad_(True), T(rhs).requires_grad_(True)
    out = ref["sddmm"].Sddmm.apply(16, 20, T(ri), T(ro), T(ci), l, r)
    out.square().sum().backward()
    ld, rd = T(lhs).double().requires_grad_(True), T(rhs).double().requires_grad_(True)
    dense = (ld @ rd.t())[T(mask != 0)]
    dense.square().sum().backward()
    assert rel_err(out.detach().numpy(), dense.detach().numpy()) < 2e-5
    assert rel_err(l.grad.numpy(), ld.grad.numpy()) < 2e-5
    assert rel_err(r.grad.numpy(), rd.grad.numpy()) < 2e-5


def test_reference_sparse_linear_module(ref):
    """Includes the backward that hands an int64 row_indices to left_spmm
    (modules/sparse_linear.py:57-65)."""
    torch.manual_seed(0)
    out_f, in_f, seq, batch = 24, 16, 7, 3
    w = (torch.randn(out_f, in_f) * (torch.rand(out_f, in_f) > 0.7)).float()
    layer = ref["sparse_linear"].SparseLinear(in_f, out_f)
    with torch.no_grad():
        layer.weight.copy_(w)
    layer.setup_sparse_tensors()
    x = torch.randn(batch, seq, in_f, requires_grad=True)
    y = layer(x)
    assert tuple(y.shape) == (batch, out_f, seq)
    y.square().sum().backward()
    wd = w.double().requires_grad_(True)
    xd = x.detach().double().requires_grad_(True)
    yd = torch.matmul(xd, wd.t()).transpose(1, 2)
    yd.square().sum().backward()
    assert rel_err(y.detach().numpy(), yd.detach().numpy()) < 2e-5
    assert rel_err(x.grad.numpy(), xd.grad.numpy()) < 2e-5
    assert rel_err(layer.values.grad.numpy(), wd.grad.numpy()[w.numpy() != 0]) < 2e-5


def test_own_functions_agree_with_reference_functions(ref, cpu_ops):
    dense_a, vals, ri, ro, ci = make_csr(20, 12, 0.5, seed=5)
    b = np.random.default_rng(6).uniform(-1, 1, (12, 5)).astype(np.float32)
    outs = []
    for fn in (ref["spmm"].Spmm, cpu_ops.Spmm):
        v, d = T(vals).requires_grad_(True), T(b).requires_grad_(True)
        fn.apply(20, 12, v, T(ri), T(ro), T(ci), d).square().sum().backward()
        outs.append((v.grad.clone(), d.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_reference_many_mask_sketches(cpu_ops, golden):
    """tests/transformer/functions.py (the reference's many-mask autograd
    sketches) imported unchanged: forward and backward through this package's
    ``*_many_mask`` ops, against the golden dense-autograd values."""
    path = os.path.join(REFERENCE, "tests", "transformer")
    if not os.path.isfile(os.path.join(path, "functions.py")):
        pytest.skip("reference sketches not present")
    saved = {k: sys.modules.pop(k, None) for k in ("functions", "utils")}
    sys.path.insert(0, path)
    try:
        functions = importlib.import_module("functions")
        utils = importlib.import_module("utils")
    finally:
        sys.path.remove(path)
        for k, v in saved.items():
            sys.modules.pop(k, None)
            if v is not None:
                sys.modules[k] = v
    g = golden("many_mask_b3_h2_s24")
    b, s = int(g["b"]), int(g["s"])
    # topology exactly as the reference builds it (stacked [b, s+1] offsets)
    _, ri, ro, ci, nnzs = utils.dense_to_sparse_3d(T(g["masks"]))
    assert ro.shape == (b, s + 1) and np.array_equal(nnzs.numpy(), g["nnzs"])
    q, k = T(g["q"]).requires_grad_(True), T(g["k"]).requires_grad_(True)
    scores = functions.Sddmm.apply(b, s, s, nnzs, ri, ro, ci, q, k)
    scores.backward(T(g["grad_scores"]))
    assert rel_err(scores.detach().numpy(), g["scores"]) < 2e-5
    assert rel_err(q.grad.numpy(), g["grad_q"]) < 2e-5
    assert rel_err(k.grad.numpy(), g["grad_k"]) < 2e-5
    w, v = T(g["weights"]).requires_grad_(True), T(g["v"]).requires_grad_(True)
    ctx = functions.Spmm.apply(b, s, s, nnzs, w, ri, ro, ci, v)
    ctx.backward(T(g["grad_context"]))
    assert rel_err(ctx.detach().numpy(), g["context"]) < 2e-5
    assert rel_err(w.grad.numpy(), g["grad_weights"]) < 2e-5
    assert rel_err(v.grad.numpy(), g["grad_v"]) < 2e-5
