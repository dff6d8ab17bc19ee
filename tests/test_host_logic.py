"""CPU: the host side above the C ABI -- autograd.Functions, nn.Modules and
topology helpers -- with torch.ops.torch_sputnik.* answered by the oracle
(tests-only CPU backend).  Gradients are compared with dense autograd in
float64 and with the fixtures produced by the reference's own modules."""
import numpy as np
import pytest
import torch

from oracle import sputnik_oracle as O
from helpers import make_csr, rel_err

TOL = 2e-5  # float32 tensors between the ops


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def test_diffsort_and_dense_to_sparse(cpu_ops):
    tsa = cpu_ops
    ro = torch.tensor([0, 1, 4, 4, 6, 11, 12], dtype=torch.int32)
    assert tsa.diffsort(ro).tolist() == [2, 0, 5, 3, 1, 4]      # SURVEY.md quirk Q1 probe
    assert tsa.diffsort(ro).dtype == torch.int32
    dense, vals, _, ro_np, ci_np = make_csr(9, 7, 0.6, seed=1)
    v, ri, ro_t, ci_t = tsa.dense_to_sparse(T(dense))
    assert np.array_equal(v.numpy(), vals) and np.array_equal(ro_t.numpy(), ro_np)
    assert np.array_equal(ci_t.numpy(), ci_np)
    assert ro_t.dtype == ci_t.dtype == ri.dtype == torch.int32
    assert sorted(ri.tolist()) == list(range(9))


def test_generate_mask_counts(cpu_ops):
    mask = cpu_ops.generate_mask(1024, 1024, sparsity=0.9, generator=np.random.default_rng(0))
    assert int(mask.sum()) == 104860          # config 3's nnz
    assert mask.shape == (1024, 1024)


def test_spmm_function_matches_golden(cpu_ops, golden):
    g = golden("autograd_spmm")
    v = T(g["values"]).requires_grad_(True)
    d = T(g["dense"]).requires_grad_(True)
    out = cpu_ops.Spmm.apply(int(g["m"]), int(g["k"]), v, T(g["row_indices"]), T(g["row_offsets"]),
                             T(g["column_indices"]), d)
    out.backward(T(g["grad_out"]))
    assert rel_err(out.detach().numpy(), g["out"]) < TOL
    assert rel_err(v.grad.numpy(), g["grad_values"]) < TOL
    assert rel_err(d.grad.numpy(), g["grad_dense"]) < TOL


def test_sddmm_function_matches_golden(cpu_ops, golden):
    g = golden("autograd_sddmm")
    l = T(g["lhs"]).requires_grad_(True)
    r = T(g["rhs"]).requires_grad_(True)
    out = cpu_ops.Sddmm.apply(int(g["m"]), int(g["n"]), T(g["row_indices"]), T(g["row_offsets"]),
                              T(g["column_indices"]), l, r)
    out.backward(T(g["grad_out"]))
    assert rel_err(out.detach().numpy(), g["out"]) < TOL
    assert rel_err(l.grad.numpy(), g["grad_lhs"]) < TOL
    assert rel_err(r.grad.numpy(), g["grad_rhs"]) < TOL


def test_sparse_linear_matches_golden(cpu_ops, golden):
    g = golden("autograd_sparse_linear")
    layer = cpu_ops.SparseLinear(int(g["in_features"]), int(g["out_features"]))
    with torch.no_grad():
        layer.weight.copy_(T(g["weight"]))
    layer.setup_sparse_tensors()
    x = T(g["x"]).requires_grad_(True)
    y = layer(x)
    assert tuple(y.shape) == (int(g["batch"]), int(g["out_features"]), int(g["seq"]))
    y.backward(T(g["grad_out"]))
    assert rel_err(y.detach().numpy(), g["y"]) < TOL
    assert rel_err(x.grad.numpy(), g["grad_x"]) < TOL
    assert rel_err(layer.values.grad.numpy(), g["grad_values"]) < TOL
    assert "values" in layer.state_dict()       # nn.Parameter, as in the reference
    assert "row_offsets" not in layer.state_dict()


def test_batched_backward_extension(cpu_ops):
    """3-D Spmm backward (impossible in the reference: its csr_transpose is 1-D only)."""
    dense_a, vals, ri, ro, ci = make_csr(12, 10, 0.6, seed=2)
    rng = np.random.default_rng(3)
    r = 3
    v3 = rng.uniform(-1, 1, (r, len(vals))).astype(np.float32)
    b = rng.uniform(-1, 1, (r, 10, 6)).astype(np.float32)
    v = T(v3).requires_grad_(True)
    d = T(b).requires_grad_(True)
    out = cpu_ops.Spmm.apply(12, 10, v, T(ri), T(ro), T(ci), d)
    out.sum().backward()
    a = torch.zeros(r, 12, 10, dtype=torch.float64)
    rows = np.repeat(np.arange(12), np.diff(ro))
    a[:, rows, ci.astype(np.int64)] = T(v3).double()
    a.requires_grad_(True)
    dd = T(b).double().requires_grad_(True)
    torch.matmul(a, dd).sum().backward()
    assert rel_err(d.grad.numpy(), dd.grad.numpy()) < TOL
    assert rel_err(v.grad.numpy(), a.grad.numpy()[:, rows, ci.astype(np.int64)]) < TOL


def test_transpose_cache_gives_same_gradients(cpu_ops):
    from torch_sputnik_amd import functional
    dense_a, vals, ri, ro, ci = make_csr(15, 11, 0.7, seed=4)
    b = np.random.default_rng(5).uniform(-1, 1, (11, 8)).astype(np.float32)
    topo = (T(ri), T(ro), T(ci))

    def grads():
        v = T(vals).requires_grad_(True)
        d = T(b).requires_grad_(True)
        cpu_ops.Spmm.apply(15, 11, v, *topo, d).square().sum().backward()
        return v.grad.clone(), d.grad.clone()

    functional.enable_transpose_cache(False)     # the reference's per-call transpose
    try:
        base = grads()
        cache = functional.enable_transpose_cache(True)
        first, second = grads(), grads()
        assert len(cache._entries) == 1
    finally:
        functional.enable_transpose_cache(functional.TRANSPOSE_CACHE_DEFAULT)
    for got in (first, second):
        assert torch.equal(got[0], base[0]) and torch.equal(got[1], base[1])


def test_default_caches_serve_registered_topologies_only(cpu_ops):
    """by default nothing is remembered about arbitrary index tensors
    (the reference's per-call behaviour, modules/spmm.py:59-64); a pattern that
    its owner registered as static is cached, and its entries die with it."""
    import gc
    from torch_sputnik_amd import functional
    dense_a, vals, ri, ro, ci = make_csr(15, 11, 0.7, seed=4)
    b = np.random.default_rng(5).uniform(-1, 1, (11, 8)).astype(np.float32)

    def grads(topo):
        v = T(vals).requires_grad_(True)
        d = T(b).requires_grad_(True)
        cpu_ops.Spmm.apply(15, 11, v, *topo, d).square().sum().backward()
        return v.grad.clone(), d.grad.clone()

    functional.enable_transpose_cache(functional.TRANSPOSE_CACHE_DEFAULT)
    functional.enable_plan_cache(functional.PLAN_CACHE_DEFAULT)
    cache = functional._cache
    assert cache.scope == "static"
    topo = (T(ri), T(ro), T(ci))
    base = grads(topo)
    assert len(cache) == 0, "an unregistered pattern was cached"
    functional.register_static_topology(*topo)
    first, second = grads(topo), grads(topo)
    assert len(cache) == 1
    for got in (first, second):
        assert torch.equal(got[0], base[0]) and torch.equal(got[1], base[1])
    # an in-place write changes the key (version counter): a new entry, not a stale hit
    topo[2].add_(0)
    grads(topo)
    assert len(cache) == 2
    # the owner drops the pattern: registration and entries go with it
    del topo
    gc.collect()
    assert len(cache) == 0 and not functional._static


def test_unregistered_pattern_never_takes_the_synchronising_transpose(cpu_ops, monkeypatch):
    """the transpose that also returns the permutation waits for the
    stream when `checked`; that is for results a cache KEEPS.  A pattern no cache serves
    must get the reference's asynchronous per-call transpose (modules/spmm.py:59-62) --
    no host round trip in every backward, legal inside a stream capture."""
    from torch_sputnik_amd import functional, ops
    dense_a, vals, ri, ro, ci = make_csr(15, 11, 0.7, seed=4)
    b = np.random.default_rng(5).uniform(-1, 1, (11, 8)).astype(np.float32)
    calls = []
    real = ops.csr_transpose_with_permutation

    def spy(m, n, values, row_offsets, column_indices, checked=True):
        calls.append(bool(checked))
        return real(m, n, values, row_offsets, column_indices, checked)

    monkeypatch.setattr(ops, "csr_transpose_with_permutation", spy)
    functional.enable_transpose_cache(functional.TRANSPOSE_CACHE_DEFAULT)

    def backward(topo):
        v = T(vals).requires_grad_(True)
        d = T(b).requires_grad_(True)
        cpu_ops.Spmm.apply(15, 11, v, *topo, d).square().sum().backward()
        lhs = T(np.ones((15, 4), np.float32)).requires_grad_(True)
        rhs = T(np.ones((11, 4), np.float32)).requires_grad_(True)
        cpu_ops.Sddmm.apply(15, 11, *topo, lhs, rhs).sum().backward()

    topo = (T(ri), T(ro), T(ci))
    backward(topo)
    assert True not in calls, "an unregistered pattern took the checked (synchronising) transpose"
    functional.register_static_topology(*topo)
    backward(topo)
    backward(topo)
    assert calls.count(True) == 1, "a registered pattern is checked once, when it is cached"
    functional.unregister_static_topology(*topo)


def test_cache_is_bounded_by_bytes():
    from torch_sputnik_amd import functional
    lru = functional._Lru(capacity=100, max_bytes=4096)
    for i in range(10):
        lru.put(i, (torch.zeros(256, dtype=torch.float32),))   # 1 KiB each
    assert len(lru) == 4 and lru.bytes == 4096
    assert lru.get(0) is None and lru.get(9) is not None
    lru.forget("nothing")
    lru.clear()
    assert lru.bytes == 0


def test_sparse_softmax_backward(cpu_ops):
    mask, vals, ri, ro, ci = make_csr(10, 14, 0.6, seed=6, round_to=1, empty_rows=(3,))
    x = T((vals * 4 - 2).astype(np.float32)).requires_grad_(True)
    w = T(np.random.default_rng(7).uniform(-1, 1, len(vals)).astype(np.float32))
    y = cpu_ops.SparseSoftmax.apply(x, T(ri), T(ro), T(ci))
    (y * w).sum().backward()
    # dense reference: masked softmax in float64
    xd = torch.full((10, 14), float("-inf"), dtype=torch.float64)
    rows = np.repeat(np.arange(10), np.diff(ro))
    xs = T((vals * 4 - 2).astype(np.float32)).double().requires_grad_(True)
    xd = xd.index_put((T(rows), T(ci.astype(np.int64))), xs)
    keep = torch.tensor([r != 3 for r in range(10)])
    yd = torch.softmax(xd[keep], dim=-1)
    wd = torch.zeros(10, 14, dtype=torch.float64).index_put((T(rows), T(ci.astype(np.int64))), w.double())
    (yd * wd[keep]).sum().backward()
    assert rel_err(x.grad.numpy(), xs.grad.numpy()) < 1e-4


def test_sparse_attention_matches_dense(cpu_ops):
    torch.manual_seed(0)
    heads, emb, seq, batch = 2, 8, 12, 2
    attn = cpu_ops.SparseAttention(heads, emb, max_sequence_length=seq, device="cpu",
                                   sparsity=0.5, mask_generator=np.random.default_rng(1))
    ws = []
    for lin in attn.linears:
        w = torch.randn(emb, emb) * (torch.rand(emb, emb) > 0.4)
        with torch.no_grad():
            lin.weight.copy_(w)
        lin.setup_sparse_tensors()
        ws.append(w.double())
    q, k, v = (torch.randn(batch, seq, emb) for _ in range(3))
    out = attn(q, k, v, None)                       # [batch, seq, emb]
    assert tuple(out.shape) == (batch, seq, emb)
    # `fused_training` (rounds 1-3) is an alias of `low_memory_training`
    assert attn.low_memory_training is False and attn.fused_training is False
    attn.fused_training = True
    assert attn.low_memory_training is True
    attn.low_memory_training = False
    assert cpu_ops.SparseAttention(heads, emb, max_sequence_length=seq, device="cpu", sparsity=0.5,
                                   mask_generator=np.random.default_rng(1),
                                   fused_training=True).low_memory_training is True

    def proj(x, w):                                 # SparseLinear then the module's reshapes
        y = torch.matmul(x.double(), w.t())         # [b, s, e]
        return y.view(batch, seq, heads, emb // heads).transpose(1, 2)
    qd, kd, vd = proj(q, ws[0]), proj(k, ws[1]), proj(v, ws[2])
    scores = torch.matmul(qd, kd.transpose(-2, -1)) / (emb // heads) ** 0.5
    scores = scores.masked_fill(attn.mask2d == 0, float("-inf"))
    probs = torch.softmax(scores, dim=-1)
    probs = torch.nan_to_num(probs)                 # rows without any mask entry
    ctx = torch.matmul(probs, vd).transpose(1, 2).reshape(batch, seq, emb)
    want = torch.matmul(ctx, ws[3].t())
    assert rel_err(out.detach().numpy(), want.numpy()) < 1e-4
