"""GPU: the reference's Python callers running on the
HIP kernels -- ``Spmm`` (modules/spmm.py:41-74), ``Sddmm`` (modules/sddmm.py:42-74),
``SparseLinearFunction`` / ``SparseLinear`` (modules/sparse_linear.py:33-89) and
``SparseAttention`` (modules/sparse_attention.py:66-128) -- forward AND backward,
against

  * the committed fixtures the reference's own modules produced
    (tests/golden/autograd_*.npz, oracle/make_golden.py), and
  * dense float64 autograd restatements of the modules' definitions,

plus the two BASELINE configurations that only exist as compositions:
C4 (16 replicas of 4096^3 in ONE launch = one GPU's share of R=128) and C5
(SparseLinear 2048^2, density 0.2, batch 8 x seq 512, forward + backward, fp32
and fp16 inputs).  Tolerance: helpers.rel_err (per-row, 1e-4) for fp32.
"""
import math

import numpy as np
import pytest
import torch

from helpers import make_csr, rel_err, rel_err_torch

pytestmark = pytest.mark.gpu

TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def tsa():
    import torch_sputnik_amd
    return torch_sputnik_amd


@pytest.fixture(params=["per_call", "cached"])
def transpose_mode(request):
    """Both forms of the backward's transposed topology: recomputed per call (as
    the reference does, modules/spmm.py:59-64) and the cached permutation."""
    from torch_sputnik_amd import functional
    functional.enable_transpose_cache(request.param == "cached")
    yield request.param
    functional.enable_transpose_cache(functional.TRANSPOSE_CACHE_DEFAULT)


def T(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def scatter_dense(m, k, ro, ci, values):
    """CSR -> dense float64 [.., m, k] on the values' device (values [nnz] or [R, nnz])."""
    rows = torch.repeat_interleave(torch.arange(m, device=ro.device), (ro[1:] - ro[:-1]).long())
    a = torch.zeros(values.shape[:-1] + (m, k), dtype=torch.float64, device=values.device)
    a[..., rows, ci.long()] = values.double()
    return a, rows


# ----------------------------------------------------------------------------
# P1 / P2 / P3 on the committed fixtures (every kernel the dispatcher can choose)
# ----------------------------------------------------------------------------
def test_spmm_function_golden(tsa, dev, golden, spmm_kernel, sddmm_kernel, transpose_mode):
    g = golden("autograd_spmm")
    v = T(g["values"], dev).requires_grad_(True)
    d = T(g["dense"], dev).requires_grad_(True)
    out = tsa.Spmm.apply(int(g["m"]), int(g["k"]), v, T(g["row_indices"], dev),
                         T(g["row_offsets"], dev), T(g["column_indices"], dev), d)
    out.backward(T(g["grad_out"], dev))
    assert rel_err(out.detach().cpu().numpy(), g["out"]) < TOL
    assert rel_err(v.grad.cpu().numpy(), g["grad_values"], g["row_offsets"]) < TOL
    assert rel_err(d.grad.cpu().numpy(), g["grad_dense"]) < TOL


def test_sddmm_function_golden(tsa, dev, golden, spmm_kernel, sddmm_kernel, transpose_mode):
    g = golden("autograd_sddmm")
    lhs = T(g["lhs"], dev).requires_grad_(True)
    rhs = T(g["rhs"], dev).requires_grad_(True)
    out = tsa.Sddmm.apply(int(g["m"]), int(g["n"]), T(g["row_indices"], dev),
                          T(g["row_offsets"], dev), T(g["column_indices"], dev), lhs, rhs)
    out.backward(T(g["grad_out"], dev))
    assert rel_err(out.detach().cpu().numpy(), g["out"], g["row_offsets"]) < TOL
    assert rel_err(lhs.grad.cpu().numpy(), g["grad_lhs"]) < TOL
    assert rel_err(rhs.grad.cpu().numpy(), g["grad_rhs"]) < TOL


def test_sparse_linear_golden(tsa, dev, golden, spmm_kernel, sddmm_kernel, transpose_mode):
    """tests/test_linear_3d.py's shape (3 x 256 x 128 x 72) through SparseLinear."""
    g = golden("autograd_sparse_linear")
    layer = tsa.SparseLinear(int(g["in_features"]), int(g["out_features"])).to(dev)
    with torch.no_grad():
        layer.weight.copy_(T(g["weight"], dev))
    layer.setup_sparse_tensors()
    x = T(g["x"], dev).requires_grad_(True)
    y = layer(x)
    assert tuple(y.shape) == (int(g["batch"]), int(g["out_features"]), int(g["seq"]))
    y.backward(T(g["grad_out"], dev))
    assert rel_err(y.detach().cpu().numpy(), g["y"]) < TOL
    assert rel_err(x.grad.cpu().numpy(), g["grad_x"]) < TOL
    assert rel_err(layer.values.grad.cpu().numpy(), g["grad_values"],
                   layer.row_offsets.cpu().numpy()) < TOL
    # forward-only path (planned op) must agree bit for bit with the autograd one
    with torch.no_grad():
        assert torch.equal(layer(x.detach()), y.detach())


# ----------------------------------------------------------------------------
# P1 / P2: larger shapes (tiled kernels by themselves) and the batched backward
# ----------------------------------------------------------------------------
@pytest.mark.parametrize("m,k,n,replicas,sparsity", [
    (256, 192, 128, 1, 0.8),      # 64-column SpMM, k = 128 SDDMM panels
    (512, 512, 512, 1, 0.9),      # wide SpMM both ways, stationary SDDMM (k = 512)
    (300, 200, 72, 1, 0.7),       # ragged: row-gather SpMM, row-wave SDDMM
    (256, 320, 256, 3, 0.85),     # batched backward (extension: [R, nnz] transpose)
])
def test_spmm_function_vs_dense_autograd(tsa, dev, transpose_mode, m, k, n, replicas, sparsity):
    _, vals, ri, ro, ci = make_csr(m, k, sparsity, seed=m + k, order="ascending")
    rng = np.random.default_rng(n)
    shape_v = (len(vals),) if replicas == 1 else (replicas, len(vals))
    shape_b = (k, n) if replicas == 1 else (replicas, k, n)
    v = T(rng.uniform(-1, 1, shape_v).astype(np.float32), dev).requires_grad_(True)
    b = T(rng.uniform(-1, 1, shape_b).astype(np.float32), dev).requires_grad_(True)
    go = T(rng.uniform(-1, 1, shape_b[:-2] + (m, n)).astype(np.float32), dev)
    d_ri, d_ro, d_ci = T(ri, dev), T(ro, dev), T(ci, dev)
    out = tsa.Spmm.apply(m, k, v, d_ri, d_ro, d_ci, b)
    out.backward(go)

    vd = v.detach().double().requires_grad_(True)
    bd = b.detach().double().requires_grad_(True)
    rows = torch.repeat_interleave(torch.arange(m, device=dev), (d_ro[1:] - d_ro[:-1]).long())
    a = torch.zeros(shape_v[:-1] + (m, k), dtype=torch.float64, device=dev)
    a[..., rows, d_ci.long()] = vd          # in-place scatter, tracked by autograd
    want = torch.matmul(a, bd)
    want.backward(go.double())
    assert rel_err_torch(out.detach(), want.detach()) < TOL
    assert rel_err_torch(b.grad, bd.grad) < TOL
    assert rel_err(v.grad.cpu().numpy(), vd.grad.cpu().numpy(), ro) < TOL


@pytest.mark.parametrize("m,k,n,replicas,sparsity", [
    (256, 64, 256, 1, 0.9),       # attention head shape
    (512, 512, 384, 1, 0.8),
    (130, 70, 90, 1, 0.6),        # ragged
    (256, 128, 256, 4, 0.9),      # batched backward (extension)
])
def test_sddmm_function_vs_dense_autograd(tsa, dev, transpose_mode, m, k, n, replicas, sparsity):
    mask, _, ri, ro, ci = make_csr(m, n, sparsity, seed=m + n, order="ascending")
    rng = np.random.default_rng(k)
    lead = () if replicas == 1 else (replicas,)
    lhs = T(rng.uniform(-1, 1, lead + (m, k)).astype(np.float32), dev).requires_grad_(True)
    rhs = T(rng.uniform(-1, 1, lead + (n, k)).astype(np.float32), dev).requires_grad_(True)
    go = T(rng.uniform(-1, 1, lead + (len(ci),)).astype(np.float32), dev)
    d_ri, d_ro, d_ci = T(ri, dev), T(ro, dev), T(ci, dev)
    out = tsa.Sddmm.apply(m, n, d_ri, d_ro, d_ci, lhs, rhs)
    out.backward(go)

    ld = lhs.detach().double().requires_grad_(True)
    rd = rhs.detach().double().requires_grad_(True)
    rows = torch.repeat_interleave(torch.arange(m, device=dev), (d_ro[1:] - d_ro[:-1]).long())
    want = torch.matmul(ld, rd.transpose(-1, -2))[..., rows, d_ci.long()]
    want.backward(go.double())
    assert rel_err(out.detach().cpu().numpy(), want.detach().cpu().numpy(), ro) < TOL
    assert rel_err_torch(lhs.grad, ld.grad) < TOL
    assert rel_err_torch(rhs.grad, rd.grad) < TOL


# ----------------------------------------------------------------------------
# P4: the whole SparseAttention module against the dense definition
# ----------------------------------------------------------------------------
def dense_attention_module(module, query, key, value):
    """modules/sparse_attention.py:105-128 restated densely in float64: four
    masked-weight projections, heads split, masked softmax over the module's
    fixed mask, context, output projection.  Output [B, S, E] like the module."""
    heads, dim = module.num_heads, module.head_dim
    weights = []
    for layer in module.linears:
        w, _ = scatter_dense(layer.output_features, layer.input_features, layer.row_offsets,
                             layer.column_indices, layer.values.detach())
        weights.append(w)
    batch, seq, _ = query.shape

    def project(x, w):   # SparseLinear: [B, out, S]; the module brings it to [B, H, S, D]
        y = torch.matmul(x.double(), w.t())
        return y.view(batch, seq, heads, dim).transpose(1, 2)

    q, k, v = (project(x, w) for x, w in zip((query, key, value), weights))
    scores = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(dim)
    scores = scores.masked_fill(module.mask2d.to(scores.device) == 0, float("-inf"))
    probs = torch.nan_to_num(torch.softmax(scores, dim=-1))
    context = torch.matmul(probs, v).transpose(1, 2).reshape(batch, seq, heads * dim)
    return torch.matmul(context, weights[3].t())


def build_attention(tsa, dev, heads, embed, seq, seed, **flags):
    module = tsa.SparseAttention(heads, embed, max_sequence_length=seq, device=dev,
                                 mask_generator=np.random.default_rng(seed), **flags).to(dev)
    rng = np.random.default_rng(seed + 1)
    for layer in module.linears:
        w = rng.uniform(-1, 1, (embed, embed)) / math.sqrt(embed * 0.3)
        w = w * (rng.random((embed, embed)) < 0.3)
        with torch.no_grad():
            layer.weight.copy_(T(w.astype(np.float32), dev))
        layer.setup_sparse_tensors()
    return module


@pytest.mark.parametrize("heads,embed,seq,batch", [
    (8, 512, 1024, 8),     # config 3 at FULL size: 64 replicas (scores 512 MB in float64)
    (8, 512, 1024, 2),     # config 3's geometry (head_dim 64), batch reduced
    (4, 256, 256, 3),      # head_dim 64, small
    (2, 64, 128, 2),       # head_dim 32: composed from the three operators
])
def test_sparse_attention_module_forward_vs_dense(tsa, dev, heads, embed, seq, batch):
    module = build_attention(tsa, dev, heads, embed, seq, seed=heads + seq)
    rng = np.random.default_rng(7)
    q, k, v = (T(rng.uniform(-1, 1, (batch, seq, embed)).astype(np.float32), dev) for _ in range(3))
    want = dense_attention_module(module, q, k, v)
    with torch.no_grad():
        fused = module(q, k, v)                       # fused inference path
    assert tuple(fused.shape) == (batch, seq, embed)
    assert rel_err_torch(fused, want) < TOL
    module.fused_inference = False
    with torch.no_grad():
        composed = module(q, k, v)                    # SDDMM -> softmax -> SpMM
    assert rel_err_torch(composed, want) < TOL


@pytest.mark.parametrize("geometry", [(4, 256, 256, 2), (8, 512, 1024, 2)],
                         ids=["h4_e256_s256_b2", "c3_geometry_b2"])
@pytest.mark.parametrize("shared_input", [False, True])
@pytest.mark.parametrize("low_memory_training", [False, True])
def test_sparse_attention_module_backward_vs_dense(tsa, dev, low_memory_training, shared_input, geometry):
    """Gradients of the whole module w.r.t. the inputs and every projection's
    values, against dense float64 autograd.  (The reference's module calls the
    raw softmax op and so cuts the gradient to Q/K, modules/sparse_attention.py:76;
    `differentiable_softmax` / `low_memory_training` give the true gradient.)"""
    heads, embed, seq, batch = geometry   # (config 3's geometry: S 1024, 8 heads of 64, batch 2)
    module = build_attention(tsa, dev, heads, embed, seq, seed=11, differentiable_softmax=True,
                             low_memory_training=low_memory_training)
    rng = np.random.default_rng(8)
    q, k, v = (T(rng.uniform(-1, 1, (batch, seq, embed)).astype(np.float32), dev).requires_grad_(True)
               for _ in range(3))
    if shared_input:   # self-attention: the three projections run as one group launch
        k = v = q
    go = T(rng.uniform(-1, 1, (batch, seq, embed)).astype(np.float32), dev)
    out = module(q, k, v)
    out.backward(go)

    # dense float64 autograd with the masked weights as leaves
    leaves = []
    for layer in module.linears:
        w, rows = scatter_dense(layer.output_features, layer.input_features, layer.row_offsets,
                                layer.column_indices, layer.values.detach())
        leaves.append((w.requires_grad_(True), rows, layer))
    qd, kd, vd = (x.detach().double().requires_grad_(True) for x in (q, k, v))
    if shared_input:
        kd = vd = qd

    def project(x, w):
        return torch.matmul(x, w.t()).view(batch, seq, heads, embed // heads).transpose(1, 2)

    pq, pk, pv = (project(x, w[0]) for x, w in zip((qd, kd, vd), leaves))
    scores = torch.matmul(pq, pk.transpose(-1, -2)) / math.sqrt(embed // heads)
    scores = scores.masked_fill(module.mask2d.to(dev) == 0, float("-inf"))
    context = torch.matmul(torch.softmax(scores, dim=-1), pv).transpose(1, 2).reshape(batch, seq, embed)
    want = torch.matmul(context, leaves[3][0].t())
    want.backward(go.double())

    assert rel_err_torch(out.detach(), want.detach()) < TOL
    # gradients pass through three to five chained fp32 kernels: 5e-4
    for got, ref in (((q, qd),) if shared_input else ((q, qd), (k, kd), (v, vd))):
        assert rel_err_torch(got.grad, ref.grad) < 5 * TOL
    for w, rows, layer in leaves:
        want_grad = w.grad[rows, layer.column_indices.long()]
        assert rel_err(layer.values.grad.cpu().numpy(), want_grad.cpu().numpy(),
                       layer.row_offsets.cpu().numpy()) < 5 * TOL


@pytest.mark.parametrize("heads,embed,seq,batch,repeats", [
    (8, 512, 1024, 8, 6),     # config 3 at full size
    (4, 256, 512, 3, 25),     # many short steps
])
def test_sparse_attention_training_step_is_deterministic(tsa, dev, heads, embed, seq, batch, repeats):
    """Self-attention forward + backward over and over on a warm GPU: every output
    and every gradient bit-identical to the first step.  (No atomics anywhere; the
    group kernels, the phased transposing stores and the banded permutation all
    cross workgroup barriers -- a missing one shows up here as a flicker.)"""
    module = build_attention(tsa, dev, heads, embed, seq, seed=99, differentiable_softmax=True)
    rng = np.random.default_rng(5)
    x = T(rng.uniform(-1, 1, (batch, seq, embed)).astype(np.float32), dev).requires_grad_(True)
    go = T(rng.uniform(-1, 1, (batch, seq, embed)).astype(np.float32), dev)

    def step():
        x.grad = None
        for layer in module.linears:
            layer.values.grad = None
        out = module(x, x, x)
        out.backward(go)
        return [out.detach().clone(), x.grad.clone()] + [l.values.grad.clone() for l in module.linears]

    first = step()
    assert all(torch.isfinite(t).all() for t in first)
    for it in range(repeats):
        again = step()
        for a, b in zip(first, again):
            assert torch.equal(a, b), f"step {it}: a result changed between identical steps"
    with torch.no_grad():   # the inference path (fused attention kernel) as well
        ref = module(x, x, x)
        for _ in range(repeats):
            assert torch.equal(module(x, x, x), ref)


# ----------------------------------------------------------------------------
# C4: one GPU's share of the 128-replica product -- 16 x 4096^3 in ONE launch
# ----------------------------------------------------------------------------
def test_c4_share_sixteen_replicas_one_launch(dev):
    """values [16, nnz], dense [16, 4096, 4096] (1.07 GB), out 1.07 GB: replica
    strides exceed 2^31 bytes, so this is the test of the 64-bit stride arithmetic.
    Every element of every replica against the dense float64 product."""
    from torch_sputnik_amd import capi
    from torch_sputnik_amd.synthetic import random_csr, uniform
    m = k = n = 4096
    replicas = 16
    ri, ro, ci, nnz = random_csr(m, k, 0.1, dev, seed=404)
    vals = uniform((replicas, nnz), dev, 1) - 0.5
    b = uniform((replicas, k, n), dev, 2) - 0.5
    out = torch.full((replicas, m, n), float("nan"), device=dev)
    ws = torch.empty(capi.spmm_workspace_bytes(m, k, n, nnz) + 16, dtype=torch.uint8, device=dev)
    capi.spmm_batched(m, k, n, replicas, ri, vals, nnz, ro, ci, b, out, ws)
    torch.cuda.synchronize()
    assert not torch.isnan(out).any()
    rows = torch.repeat_interleave(torch.arange(m, device=dev), (ro[1:] - ro[:-1]).long())
    for r in range(replicas):
        a = torch.zeros(m, k, dtype=torch.float64, device=dev)
        a[rows, ci.long()] = vals[r].double()
        assert rel_err_torch(out[r], a @ b[r].double()) < TOL, f"replica {r}"
    # the torch op on the same operands (3-D rule of src/spmm_cuda.cu:46) is bit-identical
    import torch_sputnik
    assert torch.equal(torch_sputnik.spmm(m, k, vals, ri, ro, ci, b), out)


# ----------------------------------------------------------------------------
# C5: SparseLinear 2048^2, density 0.2, forward + backward: batch 8 x seq 512 (the
# shape rounds 1-3 measured) and batch 8 x seq 2048 -- BASELINE config 5 says
# M = N = K = 2048, and N of left_spmm is the sequence length
# (/root/reference/modules/sparse_linear.py:28,89), so THIS is the stated size
# ----------------------------------------------------------------------------
@pytest.mark.parametrize("seq", [512, 2048])
@pytest.mark.parametrize("dtype,weights", [(torch.float32, "float32"), (torch.float16, "half"),
                                           (torch.float16, "float32"), (torch.bfloat16, "half")])
def test_c5_sparse_linear_full_size(tsa, dev, transpose_mode, dtype, weights, seq):
    """fp32: 1e-4 against dense float64 autograd.  fp16 (BASELINE config 5; no
    reference semantics, src/spmm_cuda.cu:42,51): inputs and weights are stored
    in half precision, the arithmetic accumulates in fp32 -- the oracle is the
    float64 computation on the SAME fp16-rounded operands, so the fp32 bound
    applies to the forward; gradients are returned in the operand's storage
    type and are held to fp16 resolution (2e-3)."""
    from torch_sputnik_amd.synthetic import random_csr, uniform
    features, batch = 2048, 8
    ri, ro, ci, nnz = random_csr(features, features, 0.2, dev, seed=505)
    assert nnz == 838864
    rows = torch.repeat_interleave(torch.arange(features, device=dev), (ro[1:] - ro[:-1]).long())
    w = torch.zeros(features, features, device=dev)
    w[rows, ci.long()] = (uniform((nnz,), dev, 1) - 0.5) * 0.1 + 0.001
    layer = tsa.SparseLinear(features, features).to(dev)
    with torch.no_grad():
        layer.weight.copy_(w)
    layer.setup_sparse_tensors()
    assert layer.values.numel() == nnz
    # (weights "float32" under half activations: what bench.py's `fp16_storage` key runs --
    # only the activations are stored in half precision)
    if weights == "half":
        layer.values = torch.nn.Parameter(layer.values.detach().to(dtype))
    x = (uniform((batch, seq, features), dev, 2) - 0.5).to(dtype).requires_grad_(True)
    go = (uniform((batch, features, seq), dev, 3) - 0.5)
    y = layer(x)
    assert y.dtype == torch.float32 and tuple(y.shape) == (batch, features, seq)
    y.backward(go)

    wd = torch.zeros(features, features, dtype=torch.float64, device=dev)
    wd[rows, ci.long()] = layer.values.detach().double()
    wd.requires_grad_(True)
    xd = x.detach().double().requires_grad_(True)
    yd = torch.matmul(xd, wd.t()).transpose(1, 2)
    yd.backward(go.double())
    # (gradients come back in the operand's storage type: one unit in the last place of
    # float16 is 4.9e-4 of the value, of bfloat16 3.9e-3)
    grad_tol = {torch.float32: TOL, torch.float16: 2e-3, torch.bfloat16: 1.6e-2}[dtype]
    assert rel_err_torch(y.detach(), yd.detach()) < TOL
    assert x.grad.dtype == dtype and layer.values.grad.dtype == layer.values.dtype
    if layer.values.dtype == torch.float32:
        # float32 weights keep a float32 gradient: the float32 bound, whatever the
        # activations' storage type (the incoming gradient is not rounded on its way)
        assert rel_err(layer.values.grad.cpu().numpy(), wd.grad[rows, ci.long()].cpu().numpy(),
                       ro.cpu().numpy()) < TOL
    assert rel_err_torch(x.grad, xd.grad) < grad_tol
    want_dw = wd.grad[rows, ci.long()]
    assert rel_err(layer.values.grad.float().cpu().numpy(), want_dw.cpu().numpy(),
                   ro.cpu().numpy()) < grad_tol
