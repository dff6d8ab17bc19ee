"""CPU: the extension surface -- fused bias/ReLU SpMM, scaled
softmax and its gradient, the many-mask family.  Oracle against the golden
fixtures (made by importing the reference's Python, oracle/make_golden.py) and
against dense definitions; the autograd wrappers against dense autograd with
the oracle answering the ops on CPU."""
import numpy as np
import pytest
import torch

from oracle import sputnik_oracle as O
from helpers import make_csr, rel_err


def _topo(g):
    return g["row_indices"], g["row_offsets"], g["column_indices"]


@pytest.mark.parametrize("name", ["spmm_bias_72x64x72", "spmm_bias_relu_40x48x36"])
def test_spmm_bias_golden(golden, name):
    g = golden(name)
    got = O.spmm_bias(int(g["m"]), int(g["k"]), g["values"], *_topo(g), g["bias"], g["dense"],
                      relu=bool(g["relu"]))
    assert rel_err(got, g["expected"]) < 1e-9
    if bool(g["relu"]):
        assert (g["expected"] == 0).any() and (got >= 0).all()  # the ReLU is exercised


def test_softmax_backward_golden(golden):
    g = golden("softmax_backward_48x40")
    y = O.sparse_softmax_scaled(g["values"], *_topo(g), float(g["scale"]))
    assert rel_err(y, g["softmax_out"]) < 1e-9
    dx = O.sparse_softmax_backward(g["softmax_out"], g["grad_out"], g["row_offsets"],
                                   float(g["scale"]))
    assert rel_err(dx, g["grad_values"]) < 1e-9
    # finite differences of the forward, a check that does not share the formula
    x = g["values"].astype(np.float64)
    eps = 1e-6
    for p in (0, 7, len(x) // 2, len(x) - 1):
        xp, xm = x.copy(), x.copy()
        xp[p] += eps
        xm[p] -= eps
        fd = ((O.sparse_softmax_scaled(xp, *_topo(g), float(g["scale"])) -
               O.sparse_softmax_scaled(xm, *_topo(g), float(g["scale"]))) / (2 * eps)
              * g["grad_out"]).sum()
        assert abs(fd - dx[p]) < 1e-6 * (1 + abs(dx[p]))


def test_many_mask_golden(golden):
    g = golden("many_mask_b3_h2_s24")
    b, s = int(g["b"]), int(g["s"])
    ri, ro, ci, nn = O.dense_to_csr_many_mask(g["masks"])
    assert np.array_equal(ro, g["row_offsets"]) and np.array_equal(ci, g["column_indices"])
    assert np.array_equal(nn, g["nnzs"]) and len(set(nn.tolist())) == 3  # ragged on purpose
    topo = (g["row_indices"], g["row_offsets"], g["column_indices"])
    scores = O.sddmm_many_mask(b, s, s, nn, *topo, g["q"], g["k"])
    assert rel_err(scores, g["scores"]) < 1e-9
    ctx = O.spmm_many_mask(b, s, s, nn, g["weights"], *topo, g["v"])
    assert rel_err(ctx, g["context"]) < 1e-9
    weights = O.sparse_softmax_many_mask(b, s, nn, scores, *topo, scale=float(g["scale"]))
    att = O.spmm_many_mask(b, s, s, nn, weights, *topo, g["v"])
    assert rel_err(att, g["attention"]) < 1e-9
    # transpose: per mask equal to the single-mask op, padding untouched (zero)
    vt, rot, cit = O.csr_transpose_many_mask(b, s, s, nn, g["weights"], g["row_offsets"],
                                             g["column_indices"])
    assert rot.shape == (b, s + 1)
    heads = int(g["heads"])
    first = np.concatenate(([0], np.cumsum(nn)))
    for i in range(b):
        v1, ro1, ci1 = O.csr_transpose(s, s, g["weights"][i * heads, :nn[i]],
                                       g["row_offsets"][i * (s + 1):(i + 1) * (s + 1)],
                                       g["column_indices"][first[i]:first[i + 1]])
        assert np.array_equal(vt[i * heads, :nn[i]], v1)
        assert np.array_equal(rot[i], ro1) and np.array_equal(cit[first[i]:first[i + 1]], ci1)
        assert not vt[i * heads, nn[i]:].any()


def test_many_mask_single_mask_is_the_plain_op():
    _, vals, ri, ro, ci = make_csr(20, 16, 0.6, seed=3)
    rng = np.random.default_rng(4)
    dense = rng.uniform(-1, 1, (4, 16, 9)).astype(np.float32)
    values = rng.uniform(-1, 1, (4, len(ci))).astype(np.float32)
    a = O.spmm_many_mask(1, 20, 16, [len(ci)], values, ri, ro, ci, dense)
    assert np.array_equal(a, O.spmm(20, 16, values, ri, ro, ci, dense))


# ---------------------------------------------------------------------------
# host logic: autograd wrappers with the oracle on the CPU dispatch key
# ---------------------------------------------------------------------------
def test_sparse_softmax_function_gradient(cpu_ops, golden):
    from torch_sputnik_amd.functional import SparseSoftmax
    g = golden("softmax_backward_48x40")
    topo = [torch.from_numpy(x) for x in _topo(g)]
    x = torch.from_numpy(g["values"]).requires_grad_(True)
    y = SparseSoftmax.apply(x, *topo, float(g["scale"]))
    y.backward(torch.from_numpy(g["grad_out"]))
    assert rel_err(y.detach().numpy(), g["softmax_out"]) < 1e-5
    assert rel_err(x.grad.numpy(), g["grad_values"]) < 1e-5
    # default scale keeps the four-argument form working
    y1 = SparseSoftmax.apply(x.detach(), *topo)
    assert rel_err(y1.numpy(), O.sparse_softmax(g["values"], *_topo(g))) < 1e-5


def test_many_mask_functions_match_dense_autograd(cpu_ops, golden):
    from torch_sputnik_amd.functional import CsrSoftmaxManyMask, SddmmManyMask, SpmmManyMask
    g = golden("many_mask_b3_h2_s24")
    b, s, heads = int(g["b"]), int(g["s"]), int(g["heads"])
    nn = torch.from_numpy(g["nnzs"])
    topo = [torch.from_numpy(g[x]) for x in ("row_indices", "row_offsets", "column_indices")]
    q = torch.from_numpy(g["q"]).requires_grad_(True)
    k = torch.from_numpy(g["k"]).requires_grad_(True)
    scores = SddmmManyMask.apply(b, s, s, nn, *topo, q, k)
    scores.backward(torch.from_numpy(g["grad_scores"]))
    assert rel_err(scores.detach().numpy(), g["scores"]) < 1e-5
    assert rel_err(q.grad.numpy(), g["grad_q"]) < 1e-5
    assert rel_err(k.grad.numpy(), g["grad_k"]) < 1e-5

    w = torch.from_numpy(g["weights"]).requires_grad_(True)
    v = torch.from_numpy(g["v"]).requires_grad_(True)
    ctx = SpmmManyMask.apply(b, s, s, nn, w, *topo, v)
    ctx.backward(torch.from_numpy(g["grad_context"]))
    assert rel_err(ctx.detach().numpy(), g["context"]) < 1e-5
    assert rel_err(w.grad.numpy(), g["grad_weights"]) < 1e-5
    assert rel_err(v.grad.numpy(), g["grad_v"]) < 1e-5

    # whole chain, differentiable end to end, against dense masked attention
    q2 = torch.from_numpy(g["q"]).requires_grad_(True)
    k2 = torch.from_numpy(g["k"]).requires_grad_(True)
    v2 = torch.from_numpy(g["v"]).requires_grad_(True)
    sc = SddmmManyMask.apply(b, s, s, nn, *topo, q2, k2)
    we = CsrSoftmaxManyMask.apply(b, s, nn, sc, *topo, float(g["scale"]))
    out = SpmmManyMask.apply(b, s, s, nn, we, *topo, v2)
    out.backward(torch.from_numpy(g["grad_context"]))
    assert rel_err(out.detach().numpy(), g["attention"]) < 1e-5

    mask = torch.from_numpy(np.repeat(g["masks"] != 0, heads, axis=0))
    qd = torch.from_numpy(g["q"]).double().requires_grad_(True)
    kd = torch.from_numpy(g["k"]).double().requires_grad_(True)
    vd = torch.from_numpy(g["v"]).double().requires_grad_(True)
    logits = (qd @ kd.transpose(1, 2) * float(g["scale"])).masked_fill(~mask, float("-inf"))
    dense = torch.softmax(logits, -1) @ vd
    dense.backward(torch.from_numpy(g["grad_context"]).double())
    for got, want in ((q2.grad, qd.grad), (k2.grad, kd.grad), (v2.grad, vd.grad)):
        assert rel_err(got.numpy(), want.numpy()) < 2e-5


def test_many_mask_argument_checks(cpu_ops):
    import torch_sputnik
    with pytest.raises(Exception):
        # CPU tensors never reach the HIP binding's checks; the oracle asserts instead
        torch_sputnik.spmm_many_mask(2, 4, 4, torch.tensor([1]), torch.zeros(2, 1),
                                     torch.zeros(8, dtype=torch.int32),
                                     torch.zeros(10, dtype=torch.int32),
                                     torch.zeros(1, dtype=torch.int32), torch.zeros(2, 4, 3))


def test_sparse_attention_oracle_matches_dense_masked_attention():
    rng = np.random.default_rng(11)
    r, s, d = 3, 40, 16
    mask = O.random_mask(s, s, 0.7, rng=rng) != 0
    mask[7] = False  # a query without keys
    _, ri, ro, ci = O.dense_to_csr(mask.astype(np.float32))
    q, k, v = (rng.uniform(-1, 1, (r, s, d)) for _ in range(3))
    scale = 1.0 / np.sqrt(d)
    got = O.sparse_attention(q, k, v, ri, ro, ci, scale)
    logits = np.where(mask[None], np.matmul(q, np.swapaxes(k, 1, 2)) * scale, -np.inf)
    with np.errstate(invalid="ignore"):
        w = np.exp(logits - logits.max(axis=-1, keepdims=True))
        w = np.nan_to_num(w / w.sum(axis=-1, keepdims=True))
    want = np.matmul(w, v)
    assert rel_err(got, want) < 1e-12
    assert not got[:, 7].any()


def test_attention_module_fused_and_composed_paths_agree(cpu_ops):
    from torch_sputnik_amd.modules import SparseAttention
    torch.manual_seed(0)
    layer = SparseAttention(num_heads=2, embedding_size=16, max_sequence_length=24,
                            device=torch.device("cpu"), sparsity=0.6,
                            mask_generator=np.random.default_rng(5))
    for lin in layer.linears:
        with torch.no_grad():
            lin.weight.copy_(torch.from_numpy(
                O.random_mask(16, 16, 0.5, rng=np.random.default_rng(1)) *
                np.random.default_rng(2).uniform(-1, 1, (16, 16)).astype(np.float32)))
        lin.setup_sparse_tensors()
    x = torch.rand(2, 24, 16)
    with torch.no_grad():
        fused = layer(x, x, x)
        layer.fused_inference = False
        composed = layer(x, x, x)
    assert rel_err(fused.numpy(), composed.numpy()) < 1e-5


def test_sparse_attention_function_matches_dense_autograd(cpu_ops):
    from torch_sputnik_amd.functional import SparseAttentionFunction
    rng = np.random.default_rng(21)
    r, s, d = 2, 28, 8
    mask = O.random_mask(s, s, 0.6, rng=rng) != 0
    mask[3] = False
    _, ri, ro, ci = O.dense_to_csr(mask.astype(np.float32))
    topo = [torch.from_numpy(x) for x in (ri, ro, ci)]
    q, k, v, go = (rng.uniform(-1, 1, (r, s, d)).astype(np.float32) for _ in range(4))
    qt, kt, vt = (torch.from_numpy(x).requires_grad_(True) for x in (q, k, v))
    out = SparseAttentionFunction.apply(qt, kt, vt, *topo, 0.3)
    out.backward(torch.from_numpy(go))
    qd, kd, vd = (torch.from_numpy(x).double().requires_grad_(True) for x in (q, k, v))
    logits = (qd @ kd.transpose(1, 2) * 0.3).masked_fill(~torch.from_numpy(mask), float("-inf"))
    w = torch.nan_to_num(torch.softmax(logits, -1))
    dense = w @ vd
    dense.backward(torch.from_numpy(go).double())
    assert rel_err(out.detach().numpy(), dense.detach().numpy()) < 1e-5
    for got, want in ((qt.grad, qd.grad), (kt.grad, kd.grad), (vt.grad, vd.grad)):
        assert rel_err(got.numpy(), torch.nan_to_num(want).numpy()) < 2e-5
