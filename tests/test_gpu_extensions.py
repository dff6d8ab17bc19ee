"""GPU: the extension kernels through the C ABI and through
torch.ops.torch_sputnik, against the oracle and the golden fixtures.  Same
tolerance as test_gpu_parity.py (1e-4, helpers.rel_err); index outputs bit-exact."""
import numpy as np
import pytest
import torch

from oracle import c_oracle
from oracle import sputnik_oracle as O
from helpers import make_csr, rel_err, rel_err_torch

pytestmark = pytest.mark.gpu

TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def capi():
    from torch_sputnik_amd import capi
    return capi


@pytest.fixture(scope="module")
def ts():
    import torch_sputnik
    return torch_sputnik


def T(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


# ----------------------------------------------------------------------------
# SpMM + bias (+ ReLU): every kernel family has its own epilogue
# ----------------------------------------------------------------------------
BIAS_SHAPES = [
    # m, k, n, sparsity, replicas
    (72, 64, 72, 0.0, 1),        # tests/test_spmm_bias_relu.py shape (row-gather kernel)
    (33, 47, 7, 0.6, 2),         # scalar path
    (512, 300, 128, 0.8, 3),     # 64-column tiled kernel
    (1024, 777, 1024, 0.9, 1),   # 256-column tiled kernel, short segments
    (512, 512, 512, 0.4, 2),     # 256-column tiled kernel, long segments
]


@pytest.mark.parametrize("m,k,n,sparsity,replicas", BIAS_SHAPES)
@pytest.mark.parametrize("relu", [0, 1])
def test_spmm_bias_capi_vs_oracle(capi, dev, spmm_kernel, m, k, n, sparsity, replicas, relu):
    _, vals, ri, ro, ci = make_csr(m, k, sparsity, seed=m + n + relu, order="ascending")
    rng = np.random.default_rng(k)
    vals = (vals * rng.choice([-1.0, 1.0], size=vals.shape)).astype(np.float32)
    b = rng.uniform(-1, 1, size=(replicas, k, n)).astype(np.float32)
    bias = rng.uniform(-2, 2, size=m).astype(np.float32)
    want = np.stack([c_oracle.spmm(m, k, vals, ro, ci, b[r]) for r in range(replicas)])
    want = want + bias.astype(np.float64)[None, :, None]
    if relu:
        want = np.maximum(want, 0.0)
    out = torch.full((replicas, m, n), float("nan"), device=dev)
    ws = torch.empty(capi.spmm_workspace_bytes(m, k, n, len(ci)) + 16, dtype=torch.uint8,
                     device=dev)
    capi.spmm_bias_batched(m, k, n, replicas, T(ri, dev), T(vals, dev), 0, T(ro, dev), T(ci, dev),
                           T(b, dev), T(bias, dev), relu, out, ws)
    got = out.cpu().numpy()
    assert not np.isnan(got).any()
    assert rel_err(got, want) < TOL
    if relu:
        assert (got >= 0).all() and (got == 0).any()


def test_spmm_null_bias_is_plain_spmm(capi, dev):
    m, k, n = 512, 300, 256
    _, vals, ri, ro, ci = make_csr(m, k, 0.8, seed=9)
    b = np.random.default_rng(1).uniform(-1, 1, size=(k, n)).astype(np.float32)
    args = (T(ri, dev), T(vals, dev), 0, T(ro, dev), T(ci, dev), T(b, dev))
    ws = torch.empty(capi.spmm_workspace_bytes(m, k, n, len(ci)), dtype=torch.uint8, device=dev)
    a = capi.spmm_batched(m, k, n, 1, *args, torch.empty(m, n, device=dev), ws)
    c = capi.spmm_bias_batched(m, k, n, 1, *args, None, 0, torch.empty(m, n, device=dev), ws)
    assert torch.equal(a, c)


@pytest.mark.parametrize("k", [512, 1600])   # 1600: four panels of the panel kernel
def test_spmm_bias_unsorted_columns_take_the_fallback_epilogue(capi, dev, spmm_kernel, k):
    m, n = 256, 256
    _, vals, ri, ro, ci = make_csr(m, k, 0.7, seed=21)
    rng = np.random.default_rng(3)
    ci = ci.copy()
    vals = vals.copy()
    for r in range(0, m, 3):  # shuffle the entries of every third row
        p = rng.permutation(ro[r + 1] - ro[r]) + ro[r]
        ci[ro[r]:ro[r + 1]] = ci[p]
        vals[ro[r]:ro[r + 1]] = vals[p]
    b = rng.uniform(-1, 1, size=(k, n)).astype(np.float32)
    bias = rng.uniform(-1, 1, size=m).astype(np.float32)
    want = np.maximum(O.spmm(m, k, vals, ri, ro, ci, b) + bias.astype(np.float64)[:, None], 0)
    ws = torch.empty(capi.spmm_workspace_bytes(m, k, n, len(ci)), dtype=torch.uint8, device=dev)
    out = capi.spmm_bias_batched(m, k, n, 1, T(ri, dev), T(vals, dev), 0, T(ro, dev), T(ci, dev),
                                 T(b, dev), T(bias, dev), 1, torch.empty(m, n, device=dev), ws)
    assert rel_err(out.cpu().numpy(), want) < TOL


@pytest.mark.parametrize("name", ["spmm_bias_72x64x72", "spmm_bias_relu_40x48x36"])
def test_spmm_bias_op_golden(ts, dev, golden, name):
    g = golden(name)
    fn = ts.spmm_bias_relu if int(g["relu"]) else ts.spmm_bias
    out = fn(int(g["m"]), int(g["k"]), T(g["values"], dev), T(g["row_indices"], dev),
             T(g["row_offsets"], dev), T(g["column_indices"], dev), T(g["bias"], dev),
             T(g["dense"], dev))
    assert out.shape == g["expected"].shape
    assert rel_err(out.cpu().numpy(), g["expected"]) < TOL
    with pytest.raises(RuntimeError):
        fn(int(g["m"]), int(g["k"]), T(g["values"], dev), T(g["row_indices"], dev),
           T(g["row_offsets"], dev), T(g["column_indices"], dev), T(g["bias"][:-1], dev),
           T(g["dense"], dev))


# ----------------------------------------------------------------------------
# scaled softmax and the softmax gradient
# ----------------------------------------------------------------------------
SOFTMAX_SHAPES = [
    # m, n, sparsity, replicas, scale
    (72, 72, 0.9, 1, 1.0),
    (1024, 1024, 0.9, 8, 0.125),      # C3 row lengths: 16 lanes per row
    (256, 2048, 0.85, 3, 0.5),        # 32 lanes per row
    (64, 4096, 0.5, 2, 0.25),         # long rows: streaming path
    (100, 37, 0.3, 5, 2.0),
]


@pytest.mark.parametrize("m,n,sparsity,replicas,scale", SOFTMAX_SHAPES)
def test_softmax_scaled_and_backward_capi(capi, dev, m, n, sparsity, replicas, scale):
    _, _, ri, ro, ci = make_csr(m, n, sparsity, seed=m + n, empty_rows=(1,))
    rng = np.random.default_rng(n)
    x = rng.uniform(-6, 6, size=(replicas, len(ci))).astype(np.float32)
    gy = rng.uniform(-1, 1, size=(replicas, len(ci))).astype(np.float32)
    y = capi.sparse_softmax_scaled_batched(m, replicas, T(x, dev), T(ri, dev), T(ro, dev),
                                           T(ci, dev), scale, torch.full_like(T(x, dev), np.nan))
    want_y = O.sparse_softmax_scaled(x, ri, ro, ci, scale)
    assert rel_err(y.cpu().numpy(), want_y) < TOL
    dx = capi.sparse_softmax_backward_batched(m, replicas, y, T(gy, dev), T(ro, dev), scale,
                                              torch.full_like(y, np.nan))
    want_dx = O.sparse_softmax_backward(y.cpu().numpy(), gy, ro, scale)
    got = dx.cpu().numpy()
    assert not np.isnan(got).any()
    assert rel_err(got, want_dx) < TOL


def test_softmax_backward_op_golden_and_autograd(ts, dev, golden):
    from torch_sputnik_amd.functional import SparseSoftmax
    g = golden("softmax_backward_48x40")
    topo = [T(g[x], dev) for x in ("row_indices", "row_offsets", "column_indices")]
    scale = float(g["scale"])
    y = ts.sparse_softmax_scaled(T(g["values"], dev), *topo, scale)
    assert rel_err(y.cpu().numpy(), g["softmax_out"]) < TOL
    dx = ts.sparse_softmax_backward(T(g["softmax_out"].astype(np.float32), dev),
                                    T(g["grad_out"], dev), topo[1], scale)
    with pytest.raises(RuntimeError):  # float64 is not a storage type of this library
        ts.sparse_softmax_backward(T(g["softmax_out"], dev), T(g["grad_out"], dev), topo[1], scale)
    assert rel_err(dx.cpu().numpy(), g["grad_values"]) < TOL
    x = T(g["values"], dev).requires_grad_(True)
    SparseSoftmax.apply(x, *topo, scale).backward(T(g["grad_out"], dev))
    assert rel_err(x.grad.cpu().numpy(), g["grad_values"]) < TOL


# ----------------------------------------------------------------------------
# many-mask family
# ----------------------------------------------------------------------------
def _many_mask_inputs(b, heads, s, hn, sparsities, seed):
    rng = np.random.default_rng(seed)
    masks = np.stack([O.random_mask(s, s, sparsities[i % len(sparsities)], round_to=4, rng=rng)
                      for i in range(b)])
    ri, ro, ci, nn = O.dense_to_csr_many_mask(masks)
    r = b * heads
    q = rng.uniform(-1, 1, (r, s, hn)).astype(np.float32)
    k = rng.uniform(-1, 1, (r, s, hn)).astype(np.float32)
    v = rng.uniform(-1, 1, (r, s, hn)).astype(np.float32)
    return masks, ri, ro, ci, nn, q, k, v


@pytest.mark.parametrize("b,heads,s,hn,sparsities", [
    (3, 2, 24, 8, (0.2, 0.5, 0.8)),
    (4, 8, 512, 64, (0.2, 0.5)),      # tests/test_attention_many_masks.py:26-36 sparsities
    (2, 4, 256, 64, (0.9,)),          # equal counts: no padding anywhere
    (8, 8, 1024, 64, (0.9, 0.8, 0.95, 0.5)),   # attention size, mixed sparsity: every op ONE launch
    # round 5: the SDDMM starts its masks largest first and deals a mask's heads over the XCDs
    # (15 replicas: no multiple of 8 -- the plain XCD order; two masks of one size; an empty one)
    (5, 3, 256, 64, (0.5, 0.9, 1.0, 0.7, 0.5)),
    (16, 4, 256, 64, (0.95, 0.5, 0.9, 0.9)),
    (70, 1, 64, 8, (0.5, 0.9, 0.2)),   # more masks than the softmax's class launches carry bits for
])
@pytest.mark.parametrize("plan_per_mask", [False, True],
                         ids=["single_mask_workspace", "many_mask_workspace"])
def test_many_mask_chain_capi_vs_oracle(capi, dev, b, heads, s, hn, sparsities, plan_per_mask):
    """plan_per_mask: the SDDMM workspace holds one plan per mask, so all masks run on
    the LDS-tiled kernel in one launch; with the single-mask workspace the (also
    single-launch) row-wave kernel takes the call."""
    masks, ri, ro, ci, nn, q, k, v = _many_mask_inputs(b, heads, s, hn, sparsities, seed=s + b)
    r, width = b * heads, int(nn.max())
    scale = 1.0 / np.sqrt(hn)
    d_ri, d_ro, d_ci = T(ri, dev), T(ro, dev), T(ci, dev)
    sddmm_ws = (capi.sddmm_many_mask_workspace_bytes(b, s, hn, s, width) if plan_per_mask
                else capi.sddmm_workspace_bytes(s, hn, s, width))
    ws = torch.empty(max(sddmm_ws, capi.spmm_workspace_bytes(s, s, hn, width),
                         capi.csr_transpose_workspace_bytes(s, s, width)) + 16,
                     dtype=torch.uint8, device=dev)
    scores = torch.zeros(r, width, device=dev)
    capi.sddmm_many_mask(b, s, hn, s, nn, r, d_ri, d_ro, d_ci, T(q, dev), T(k, dev), scores, ws)
    want_scores = O.sddmm_many_mask(b, s, s, nn, ri, ro, ci, q, k)
    assert rel_err(scores.cpu().numpy(), want_scores) < TOL

    weights = torch.zeros(r, width, device=dev)
    capi.sparse_softmax_many_mask(b, s, nn, r, scores, d_ri, d_ro, d_ci, scale, weights)
    want_weights = O.sparse_softmax_many_mask(b, s, nn, scores.cpu().numpy(), ri, ro, ci, scale)
    assert rel_err(weights.cpu().numpy(), want_weights) < TOL

    ctx = torch.full((r, s, hn), float("nan"), device=dev)
    capi.spmm_many_mask(b, s, s, hn, nn, r, d_ri, weights, d_ro, d_ci, T(v, dev), ctx, ws)
    want_ctx = O.spmm_many_mask(b, s, s, nn, weights.cpu().numpy(), ri, ro, ci, v)
    got = ctx.cpu().numpy()
    assert not np.isnan(got).any()
    assert rel_err(got, want_ctx) < TOL

    # transpose: bit-exact, padding untouched
    vt = torch.zeros(r, width, device=dev)
    rot = torch.empty(b, s + 1, dtype=torch.int32, device=dev)
    cit = torch.empty(len(ci), dtype=torch.int32, device=dev)
    capi.csr_transpose_many_mask(b, s, s, nn, r, weights, d_ro, d_ci, vt, rot, cit, None, ws)
    w_vt, w_rot, w_cit = O.csr_transpose_many_mask(b, s, s, nn, weights.cpu().numpy(), ro, ci)
    assert np.array_equal(rot.cpu().numpy(), w_rot)
    assert np.array_equal(cit.cpu().numpy(), w_cit)
    assert np.array_equal(vt.cpu().numpy(), w_vt)

    # softmax gradient, many masks
    gy = np.random.default_rng(1).uniform(-1, 1, (r, width)).astype(np.float32)
    dx = torch.zeros(r, width, device=dev)
    capi.sparse_softmax_backward_many_mask(b, s, nn, r, weights, T(gy, dev), d_ro, scale, dx)
    want_dx = O.sparse_softmax_backward_many_mask(b, s, nn, weights.cpu().numpy(), gy, ro, scale)
    assert rel_err(dx.cpu().numpy(), want_dx) < TOL


@pytest.mark.parametrize("b,heads,s,sparsities", [
    (3, 1, 96, (0.5, 0.8)),                    # one replica per mask: values staged with the entries
    (4, 2, 160, (0.7, 1.0, 0.9)),              # an EMPTY mask inside the batch
    (8, 8, 1024, (0.9, 0.8, 0.95, 0.5)),       # attention size
    (5, 2, 300, (0.6, 0.9)),                   # rows not a multiple of the 32-row chunks
])
@pytest.mark.parametrize("regions", [True, False], ids=["one_launch_per_phase", "mask_after_mask"])
def test_csr_transpose_many_mask_capi_vs_oracle(capi, dev, b, heads, s, sparsities, regions):
    """All masks in the same three launches (a region of tables per mask in the workspace)
    against the mask-after-mask form (the single-mask workspace) and the oracle: bit-exact
    values, offsets, indices and permutation; padding behind a mask's entries untouched."""
    rng = np.random.default_rng(s + b)
    masks = np.stack([O.random_mask(s, s, sparsities[i % len(sparsities)], round_to=4, rng=rng)
                      for i in range(b)])
    ri, ro, ci, nn = O.dense_to_csr_many_mask(masks)
    r, width = b * heads, int(nn.max())
    values = rng.uniform(-1, 1, (r, width)).astype(np.float32)
    ws_bytes = (capi.csr_transpose_many_mask_workspace_bytes(b, s, s, width) if regions
                else capi.csr_transpose_workspace_bytes(s, s, width))
    assert capi.csr_transpose_many_mask_workspace_bytes(b, s, s, width) > \
        capi.csr_transpose_workspace_bytes(s, s, width)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    d_ro, d_ci, d_values = T(ro, dev), T(ci, dev), T(values, dev)
    vt = torch.full((r, width), -7.0, device=dev)
    rot = torch.empty(b, s + 1, dtype=torch.int32, device=dev)
    cit = torch.full((len(ci),), -1, dtype=torch.int32, device=dev)
    perm = torch.full((len(ci),), -1, dtype=torch.int32, device=dev)
    capi.csr_transpose_many_mask(b, s, s, nn, r, d_values, d_ro, d_ci, vt, rot, cit, perm, ws)
    w_vt, w_rot, w_cit = O.csr_transpose_many_mask(b, s, s, nn, values, ro, ci)
    assert np.array_equal(rot.cpu().numpy(), w_rot)
    assert np.array_equal(cit.cpu().numpy(), w_cit)
    got = vt.cpu().numpy()
    first = 0
    for i in range(b):
        n_i = int(nn[i])
        rows = slice(i * heads, (i + 1) * heads)
        assert np.array_equal(got[rows, :n_i], w_vt[rows, :n_i])
        assert (got[rows, n_i:] == -7.0).all()
        # the permutation is relative to the mask's own entries
        p_i = perm[first:first + n_i].cpu().numpy()
        assert np.array_equal(values[rows][:, p_i], w_vt[rows, :n_i])
        first += n_i
    # topology only (no values), as the planning callers use it
    rot2 = torch.empty_like(rot)
    cit2 = torch.empty_like(cit)
    capi.csr_transpose_many_mask(b, s, s, nn, 0, None, d_ro, d_ci, None, rot2, cit2, None, ws)
    assert torch.equal(rot2, rot) and torch.equal(cit2, cit)


def test_many_mask_ops_golden_and_autograd(ts, dev, golden):
    from torch_sputnik_amd.functional import CsrSoftmaxManyMask, SddmmManyMask, SpmmManyMask
    g = golden("many_mask_b3_h2_s24")
    b, s = int(g["b"]), int(g["s"])
    nn = torch.from_numpy(g["nnzs"])  # host tensor, as tests/transformer/utils.py:36 builds it
    # stacked [b, s+1] / [b, s] index tensors are accepted like flat ones
    topo = [T(g["row_indices"].reshape(b, s), dev), T(g["row_offsets"].reshape(b, s + 1), dev),
            T(g["column_indices"], dev)]
    scores = ts.sddmm_many_mask(b, s, s, nn, *topo, T(g["q"], dev), T(g["k"], dev))
    assert scores.shape == g["scores"].shape
    assert rel_err(scores.cpu().numpy(), g["scores"]) < TOL
    weights = ts.sparse_softmax_many_mask(b, s, nn, scores * float(g["scale"]), *topo)
    att = ts.spmm_many_mask(b, s, s, nn, weights, *topo, T(g["v"], dev))
    assert rel_err(att.cpu().numpy(), g["attention"]) < TOL
    vt, rot, cit = ts.csr_transpose_many_mask(b, s, s, nn, T(g["weights"], dev), topo[1], topo[2])
    w_vt, w_rot, w_cit = O.csr_transpose_many_mask(b, s, s, g["nnzs"], g["weights"],
                                                   g["row_offsets"], g["column_indices"])
    assert rot.shape == (b, s + 1)
    assert np.array_equal(vt.cpu().numpy(), w_vt) and np.array_equal(rot.cpu().numpy(), w_rot)
    assert np.array_equal(cit.cpu().numpy(), w_cit)

    q = T(g["q"], dev).requires_grad_(True)
    k = T(g["k"], dev).requires_grad_(True)
    sc = SddmmManyMask.apply(b, s, s, nn, *topo, q, k)
    sc.backward(T(g["grad_scores"], dev))
    assert rel_err(q.grad.cpu().numpy(), g["grad_q"]) < TOL
    assert rel_err(k.grad.cpu().numpy(), g["grad_k"]) < TOL
    w = T(g["weights"], dev).requires_grad_(True)
    v = T(g["v"], dev).requires_grad_(True)
    ctx = SpmmManyMask.apply(b, s, s, nn, w, *topo, v)
    ctx.backward(T(g["grad_context"], dev))
    assert rel_err(ctx.detach().cpu().numpy(), g["context"]) < TOL
    assert rel_err(w.grad.cpu().numpy(), g["grad_weights"]) < TOL
    assert rel_err(v.grad.cpu().numpy(), g["grad_v"]) < TOL
    # chain with the fused scale
    q2 = T(g["q"], dev).requires_grad_(True)
    out = SpmmManyMask.apply(
        b, s, s, nn,
        CsrSoftmaxManyMask.apply(b, s, nn, SddmmManyMask.apply(b, s, s, nn, *topo, q2,
                                                                T(g["k"], dev)),
                                 *topo, float(g["scale"])),
        *topo, T(g["v"], dev))
    out.sum().backward()
    assert rel_err(out.detach().cpu().numpy(), g["attention"]) < TOL
    assert torch.isfinite(q2.grad).all()


def test_many_mask_errors(ts, dev, golden):
    g = golden("many_mask_b3_h2_s24")
    b, s = int(g["b"]), int(g["s"])
    nn = torch.from_numpy(g["nnzs"])
    topo = [T(g[x], dev) for x in ("row_indices", "row_offsets", "column_indices")]
    with pytest.raises(RuntimeError):  # 5 replicas for 3 masks
        ts.sddmm_many_mask(b, s, s, nn, *topo, T(g["q"][:5], dev), T(g["k"][:5], dev))
    with pytest.raises(RuntimeError):  # wrong number of counts
        ts.sddmm_many_mask(b, s, s, nn[:2], *topo, T(g["q"], dev), T(g["k"], dev))
    with pytest.raises(RuntimeError):  # value rows shorter than the longest mask
        ts.spmm_many_mask(b, s, s, nn, T(g["weights"][:, :10], dev), *topo, T(g["v"], dev))


# ----------------------------------------------------------------------------
# fused sparse attention forward
# ----------------------------------------------------------------------------
ATTENTION_SHAPES = [
    # m, n, sparsity, replicas, empty rows, row order
    (256, 256, 0.9, 4, (0, 255), "ascending"),
    (1024, 1024, 0.9, 8, (), "ascending"),        # BASELINE config 3 mask shape
    (200, 300, 0.8, 3, (17,), "random"),          # ragged: partial row block, partial last chunk
    (128, 512, 0.5, 2, (), "descending"),         # > 32 entries of a row inside one chunk
    (64, 128, 0.0, 2, (), "identity"),            # dense mask
    (130, 70, 0.97, 5, (1, 2, 3), "ascending"),   # mostly empty windows
]


@pytest.mark.parametrize("m,n,sparsity,replicas,empty,order", ATTENTION_SHAPES)
def test_sparse_attention_capi_vs_oracle(capi, dev, m, n, sparsity, replicas, empty, order):
    d = 64
    _, _, ri, ro, ci = make_csr(m, n, sparsity, seed=m + 3 * n, empty_rows=empty, order=order)
    rng = np.random.default_rng(m)
    q = rng.uniform(-2, 2, (replicas, m, d)).astype(np.float32)
    k = rng.uniform(-2, 2, (replicas, n, d)).astype(np.float32)
    v = rng.uniform(-1, 1, (replicas, n, d)).astype(np.float32)
    scale = 1.0 / np.sqrt(d)
    assert capi.sparse_attention_supported(m, n, d, len(ci))
    ws = torch.empty(capi.sparse_attention_workspace_bytes(m, n, d, len(ci)), dtype=torch.uint8,
                     device=dev)
    out = torch.full((replicas, m, d), float("nan"), device=dev)
    lse = torch.full((replicas, m), float("nan"), device=dev)
    capi.sparse_attention_forward(m, n, d, replicas, T(ri, dev), T(ro, dev), T(ci, dev), T(q, dev),
                                  T(k, dev), T(v, dev), scale, out, lse, ws)
    got = out.cpu().numpy()
    assert not np.isnan(got).any(), "some output elements were never written"
    want = O.sparse_attention(q, k, v, ri, ro, ci, scale)
    assert rel_err(got, want) < TOL
    for r in empty:
        assert not got[:, r].any()
    # log-sum-exp of the scaled scores, per row
    scores = O.sddmm(m, n, ri, ro, ci, q, k) * scale
    want_lse = np.full((replicas, m), -np.inf)
    for r in range(m):
        if ro[r + 1] > ro[r]:
            seg = scores[:, ro[r]:ro[r + 1]]
            mx = seg.max(axis=1)
            want_lse[:, r] = mx + np.log(np.exp(seg - mx[:, None]).sum(axis=1))
    got_lse = lse.cpu().numpy()
    finite = np.isfinite(want_lse)
    assert np.array_equal(np.isneginf(got_lse), ~finite)
    assert np.max(np.abs(got_lse[finite] - want_lse[finite])) < 1e-4 * (1 + np.abs(want_lse[finite]).max())


def test_sparse_attention_capi_c3_full_size(capi, dev):
    """BASELINE config 3 at FULL size through the C ABI: 64 replicas (batch 8 x 8
    heads), S = 1024, head_dim 64, mask density 0.1 -- every output element and
    every log-sum-exp against the dense float64 definition (scores [64, 1024, 1024]
    in float64 on the device, 512 MB).  The kernel's grid decode (XCD-local 64-bit
    work index) is only exercised at this replica count."""
    m = n = 1024
    d, replicas = 64, 64
    _, _, ri, ro, ci = make_csr(m, n, 0.9, seed=31, order="ascending")
    g = torch.Generator(device="cpu").manual_seed(32)
    q, k, v = (torch.empty(replicas, m, d).uniform_(-2, 2, generator=g).to(dev) for _ in range(3))
    scale = 1.0 / np.sqrt(d)
    ws = torch.empty(capi.sparse_attention_workspace_bytes(m, n, d, len(ci)), dtype=torch.uint8,
                     device=dev)
    out = torch.full((replicas, m, d), float("nan"), device=dev)
    lse = torch.full((replicas, m), float("nan"), device=dev)
    capi.sparse_attention_forward(m, n, d, replicas, T(ri, dev), T(ro, dev), T(ci, dev), q, k, v,
                                  scale, out, lse, ws)
    mask = torch.zeros(m, n, dtype=torch.bool, device=dev)
    rows = torch.repeat_interleave(torch.arange(m, device=dev), T(np.diff(ro).astype(np.int64), dev))
    mask[rows, T(ci.astype(np.int64), dev)] = True
    scores = torch.matmul(q.double(), k.double().transpose(1, 2)) * scale
    scores = scores.masked_fill(~mask, float("-inf"))
    want = torch.matmul(torch.nan_to_num(torch.softmax(scores, dim=-1)), v.double())
    assert not torch.isnan(out).any()
    assert rel_err_torch(out, want) < TOL
    want_lse = torch.logsumexp(scores, dim=-1)
    assert torch.max(torch.abs(lse.double() - want_lse)) < 1e-4 * (1 + want_lse.abs().max())


def test_sparse_attention_unsorted_columns(capi, dev):
    m, n, d, replicas = 256, 256, 64, 2
    _, _, ri, ro, ci = make_csr(m, n, 0.85, seed=77)
    rng = np.random.default_rng(5)
    ci = ci.copy()
    for r in range(0, m, 5):
        ci[ro[r]:ro[r + 1]] = ci[ro[r]:ro[r + 1]][rng.permutation(ro[r + 1] - ro[r])]
    q = rng.uniform(-1, 1, (replicas, m, d)).astype(np.float32)
    k = rng.uniform(-1, 1, (replicas, n, d)).astype(np.float32)
    v = rng.uniform(-1, 1, (replicas, n, d)).astype(np.float32)
    ws = torch.empty(capi.sparse_attention_workspace_bytes(m, n, d, len(ci)), dtype=torch.uint8,
                     device=dev)
    out = torch.empty(replicas, m, d, device=dev)
    capi.sparse_attention_forward(m, n, d, replicas, T(ri, dev), T(ro, dev), T(ci, dev), T(q, dev),
                                  T(k, dev), T(v, dev), 0.125, out, None, ws)
    assert rel_err(out.cpu().numpy(), O.sparse_attention(q, k, v, ri, ro, ci, 0.125)) < TOL


def test_sparse_attention_op_matches_three_op_chain(ts, dev):
    rng = np.random.default_rng(9)
    for d in (64, 32):  # 32: not served by the fused kernel, composed by the op
        m = n = 192
        _, _, ri, ro, ci = make_csr(m, n, 0.8, seed=d)
        topo = [T(x, dev) for x in (ri, ro, ci)]
        q, k, v = (T(rng.uniform(-1, 1, (6, m, d)).astype(np.float32), dev) for _ in range(3))
        scale = 1.0 / np.sqrt(d)
        fused = ts.sparse_attention(q, k, v, *topo, scale)
        chain = ts.spmm(m, n, ts.sparse_softmax(ts.sddmm(m, n, *topo, q, k) * scale, *topo),
                        *topo, v)
        assert fused.shape == chain.shape == (6, m, d)
        assert rel_err(fused.cpu().numpy(), chain.cpu().numpy()) < TOL
        want = O.sparse_attention(q.cpu().numpy(), k.cpu().numpy(), v.cpu().numpy(), ri, ro, ci,
                                  scale)
        assert rel_err(fused.cpu().numpy(), want) < TOL


def test_sparse_attention_large_scores_stay_finite(ts, dev):
    """Scores around +-90: exp of the raw values would overflow fp32."""
    m = n = 128
    _, _, ri, ro, ci = make_csr(m, n, 0.7, seed=3)
    rng = np.random.default_rng(4)
    q = (rng.uniform(-1, 1, (2, m, 64)) * 12).astype(np.float32)
    k = (rng.uniform(-1, 1, (2, n, 64)) * 12).astype(np.float32)
    v = rng.uniform(-1, 1, (2, n, 64)).astype(np.float32)
    out = ts.sparse_attention(T(q, dev), T(k, dev), T(v, dev), T(ri, dev), T(ro, dev), T(ci, dev),
                              0.125)
    got = out.cpu().numpy()
    assert np.isfinite(got).all()
    # conditioning: a score of magnitude ~100 carries an fp32 rounding error of
    # ~100 * 2^-24 * sqrt(64) ~ 5e-5, which the exponential turns into a relative
    # error of the weights of the same size -- 10x the bound of well-scaled inputs
    assert rel_err(got, O.sparse_attention(q, k, v, ri, ro, ci, 0.125)) < 10 * TOL


def test_many_mask_with_an_empty_mask(ts, dev):
    """One batch element masks everything out: its heads get zeros / nothing."""
    rng = np.random.default_rng(12)
    b, heads, s, hn = 3, 2, 32, 16
    masks = np.stack([O.random_mask(s, s, 0.5, rng=rng), np.zeros((s, s), np.float32),
                      O.random_mask(s, s, 0.8, rng=rng)])
    ri, ro, ci, nn = O.dense_to_csr_many_mask(masks)
    assert nn[1] == 0
    r = b * heads
    q = rng.uniform(-1, 1, (r, s, hn)).astype(np.float32)
    k = rng.uniform(-1, 1, (r, s, hn)).astype(np.float32)
    v = rng.uniform(-1, 1, (r, s, hn)).astype(np.float32)
    topo = [T(ri, dev), T(ro, dev), T(ci, dev)]
    nnt = torch.from_numpy(nn)
    scores = ts.sddmm_many_mask(b, s, s, nnt, *topo, T(q, dev), T(k, dev))
    assert rel_err(scores.cpu().numpy(), O.sddmm_many_mask(b, s, s, nn, ri, ro, ci, q, k)) < TOL
    weights = ts.sparse_softmax_many_mask(b, s, nnt, scores, *topo)
    out = ts.spmm_many_mask(b, s, s, nnt, weights, *topo, T(v, dev))
    want = O.spmm_many_mask(b, s, s, nn, O.sparse_softmax_many_mask(
        b, s, nn, scores.cpu().numpy(), ri, ro, ci), ri, ro, ci, v)
    got = out.cpu().numpy()
    assert rel_err(got, want) < TOL
    assert not got[heads:2 * heads].any()
    vt, rot, cit = ts.csr_transpose_many_mask(b, s, s, nnt, weights, topo[1], topo[2])
    w_vt, w_rot, w_cit = O.csr_transpose_many_mask(b, s, s, nn, weights.cpu().numpy(), ro, ci)
    assert np.array_equal(rot.cpu().numpy(), w_rot) and np.array_equal(cit.cpu().numpy(), w_cit)
    assert np.array_equal(vt.cpu().numpy(), w_vt)


def test_sparse_attention_function_gradients(ts, dev):
    """Fused forward + recomputing backward against dense float64 autograd."""
    from torch_sputnik_amd.functional import SparseAttentionFunction
    rng = np.random.default_rng(31)
    r, s, d = 4, 256, 64
    mask = O.random_mask(s, s, 0.9, rng=rng) != 0
    mask[9] = False
    _, ri, ro, ci = O.dense_to_csr(mask.astype(np.float32))
    topo = [T(x, dev) for x in (ri, ro, ci)]
    q, k, v, go = (rng.uniform(-1, 1, (r, s, d)).astype(np.float32) for _ in range(4))
    qt, kt, vt = (T(x, dev).requires_grad_(True) for x in (q, k, v))
    scale = 0.125
    out = SparseAttentionFunction.apply(qt, kt, vt, *topo, scale)
    out.backward(T(go, dev))
    qd, kd, vd = (torch.from_numpy(x).double().requires_grad_(True) for x in (q, k, v))
    logits = (qd @ kd.transpose(1, 2) * scale).masked_fill(~torch.from_numpy(mask), float("-inf"))
    dense = torch.nan_to_num(torch.softmax(logits, -1)) @ vd
    dense.backward(torch.from_numpy(go).double())
    assert rel_err(out.detach().cpu().numpy(), dense.detach().numpy()) < TOL
    for got, want in ((qt.grad, qd.grad), (kt.grad, kd.grad), (vt.grad, vd.grad)):
        assert rel_err(got.cpu().numpy(), torch.nan_to_num(want).numpy()) < TOL
    # the op that also returns the log-sum-exp agrees with the plain one
    o2, lse = ts.sparse_attention_with_lse(T(q, dev), T(k, dev), T(v, dev), *topo, scale)
    assert torch.equal(o2, out.detach()) and lse.shape == (r, s)
    assert torch.isneginf(lse[:, 9]).all() and torch.isfinite(lse[:, 10]).all()


# ----------------------------------------------------------------------------
# static topologies: plan once, run many times
# ----------------------------------------------------------------------------
@pytest.mark.parametrize("m,k,n,replicas", [(512, 512, 256, 8), (256, 300, 64, 4), (100, 64, 18, 2),
                                            (2048, 2048, 256, 4),
                                            (2048, 256, 512, 12)])  # 512-column kernel by replica count
def test_planned_ops_equal_the_per_call_ops(ts, dev, spmm_kernel, sddmm_kernel, m, k, n, replicas):
    rng = np.random.default_rng(m + n)
    _, vals, ri, ro, ci = make_csr(m, k, 0.8, seed=m, order="ascending")
    topo = [T(x, dev) for x in (ri, ro, ci)]
    b = T(rng.uniform(-1, 1, (replicas, k, n)).astype(np.float32), dev)
    v = T(rng.uniform(-1, 1, (replicas, len(ci))).astype(np.float32), dev)
    plan = ts.spmm_plan(m, k, n, *topo)
    for _ in range(2):  # a plan is reusable
        assert torch.equal(ts.spmm_planned(m, k, v, *topo, b, plan), ts.spmm(m, k, v, *topo, b))
    assert torch.equal(ts.left_spmm_planned(m, k, v[0].contiguous(), *topo, b, plan),
                       ts.left_spmm(m, k, v[0].contiguous(), *topo, b))
    for kk in (64, 40):
        lhs = T(rng.uniform(-1, 1, (replicas, m, kk)).astype(np.float32), dev)
        rhs = T(rng.uniform(-1, 1, (replicas, k, kk)).astype(np.float32), dev)
        splan = ts.sddmm_plan(m, k, kk, *topo)
        # (a planned product may take another kernel than the per-call one -- the pair-flat
        # kernel at kk = 64 -- whose partial sums are added in another order)
        torch.testing.assert_close(ts.sddmm_planned(m, k, *topo, lhs, rhs, splan),
                                   ts.sddmm(m, k, *topo, lhs, rhs), rtol=1e-5, atol=1e-5)
    from torch_sputnik_amd import capi
    if capi.spmm_workspace_bytes(m, k, n, len(ci)) > 4:
        with pytest.raises(RuntimeError):  # a plan made for other sizes is refused
            ts.spmm_planned(m, k, v, *topo, b, torch.zeros(4, dtype=torch.uint8, device=dev))


@pytest.mark.parametrize("m,n,width,replicas", [
    (1024, 1024, 64, 8),    # attention dV / dK: two panels, gathered values
    (512, 512, 1024, 3),    # projection dX: one panel
    (300, 200, 72, 2),      # ragged, partial column tile
    (256, 100, 18, 2),      # not served by the panel kernel: permuted copy + the usual product
    (128, 5000, 64, 1),     # transposed k = 5000 > 4096: permuted copy as well
])
def test_spmm_permuted_is_the_product_with_the_transpose(ts, dev, m, n, width, replicas):
    """A^T @ X through the transposed topology and the permutation, values kept in
    A's order, against the dense product in float64."""
    a, vals, ri, ro, ci = make_csr(m, n, 0.9, seed=m + width)
    rng = np.random.default_rng(width)
    v = rng.uniform(-1, 1, (replicas, len(ci))).astype(np.float32)
    x = rng.uniform(-1, 1, (replicas, m, width)).astype(np.float32)
    _, ro_t, ci_t, perm = ts.csr_transpose_with_permutation(m, n, T(v[0], dev), T(ro, dev), T(ci, dev))
    from torch_sputnik_amd.topology import diffsort
    ri_t = diffsort(ro_t)
    got = ts.spmm_permuted(n, m, T(v, dev), perm, ri_t, ro_t, ci_t, T(x, dev))
    dense = np.zeros((replicas, m, n))
    rows = np.repeat(np.arange(m), np.diff(ro))
    dense[:, rows, ci] = v
    want = np.einsum("rmn,rmw->rnw", dense, x.astype(np.float64))
    got = got.reshape(replicas, n, width)   # one replica comes back 2-D (src/spmm_cuda.cu:42)
    assert rel_err(got.cpu().numpy(), want.astype(np.float32)) < TOL
    # the shared-values (left) form
    left = ts.spmm_permuted(n, m, T(v[0], dev), perm, ri_t, ro_t, ci_t, T(x, dev), left=True)
    want_left = np.einsum("mn,rmw->rnw", dense[0], x.astype(np.float64))
    assert rel_err(left.cpu().numpy(), want_left.astype(np.float32)) < TOL
    # same numbers as the two-step form
    two_step = ts.spmm(n, m, ts.permute_last(T(v, dev), perm), ri_t, ro_t, ci_t, T(x, dev))
    assert rel_err(got.cpu().numpy(), two_step.reshape(replicas, n, width).cpu().numpy()) < TOL


@pytest.mark.parametrize("m,k,n,replicas,block,left", [
    (512, 512, 1024, 3, 64, True),     # projection with the head split (config 3 geometry)
    (512, 512, 1024, 2, 128, True),    # head_dim 128
    (1024, 1024, 64, 6, 1024, False),  # attention P.V stored as C^T: two panels, block = m
    (320, 200, 72, 2, 64, True),       # m not a multiple of 256, partial column tile
    (768, 96, 136, 2, 256, False),     # block = the workgroup's 256 rows; panel smaller than the tile
    (512, 512, 256, 2, 512, True),     # block = 2 workgroups of rows
    (192, 300, 20, 2, 64, True),       # n < 64: product + tiled transpose
    (256, 5000, 64, 1, 64, False),     # k too large for the panel kernel: product + transpose
    (128, 128, 64, 2, 32, True),       # block of 32 rows: product + transpose
])
def test_spmm_transposed_out(ts, dev, spmm_kernel, m, k, n, replicas, block, left):
    """The product stored as the transposes of its row blocks equals the plain
    product moved by a layout pass (bit for bit where both take the panel kernel:
    same summation order) and the dense float64 definition."""
    a, vals, ri, ro, ci = make_csr(m, k, 0.9, seed=m + n)
    rng = np.random.default_rng(n)
    v = rng.uniform(-1, 1, (len(ci),) if left else (replicas, len(ci))).astype(np.float32)
    b = rng.uniform(-1, 1, (replicas, k, n)).astype(np.float32)
    topo = [T(x, dev) for x in (ri, ro, ci)]
    got = ts.spmm_transposed_out(m, k, T(v, dev), *topo, T(b, dev), block, left=left)
    assert got.shape == (replicas * (m // block), n, block)
    plain = (ts.left_spmm if left else ts.spmm)(m, k, T(v, dev), *topo, T(b, dev))
    moved = plain.reshape(replicas * (m // block), block, n).transpose(1, 2)
    assert rel_err(got.cpu().numpy(), moved.cpu().numpy()) < TOL
    dense = np.zeros((replicas, m, k))
    rows = np.repeat(np.arange(m), np.diff(ro))
    dense[:, rows, ci] = v
    want = np.einsum("rmk,rkn->rmn", dense, b.astype(np.float64))
    want = want.reshape(replicas * (m // block), block, n).transpose(0, 2, 1)
    assert rel_err(got.cpu().numpy(), np.ascontiguousarray(want).astype(np.float32)) < TOL


@pytest.mark.parametrize("m,n,k,replicas,count", [
    (512, 512, 1024, 8, 3),    # config 3: the q, k, v weight gradients (4 panels x 8 replicas each)
    (256, 320, 256, 3, 4),     # one panel per replica, ragged mask width, four products
    (100, 60, 40, 5, 2),       # row-wave kernel ([R, nnz] partial vectors)
    (512, 512, 256, 1, 2),     # one replica of one panel: nothing left to sum
])
def test_sddmm_sum_group_equals_the_single_calls(ts, dev, m, n, k, replicas, count):
    """Round 5: the weight gradients of a group of projections in one call -- each product
    as sddmm_sum_planned runs it, the partial vectors of all of them added by ONE launch.
    Bit-identical to the single calls (same partial vectors, same order), masks of different
    sizes (one with an entry count that is no multiple of 4, one EMPTY)."""
    from torch_sputnik_amd import ops
    rng = np.random.default_rng(m + k)
    rhs = T(rng.uniform(-1, 1, (replicas, n, k)).astype(np.float32), dev)
    topo, lhs, plans, want = [], [], [], []
    for p in range(count):
        sparsity = (0.9, 0.8, 1.0, 0.95)[p] if count == 4 else (0.9, 0.85, 0.8)[p]
        _, _, ri, ro, ci = make_csr(m, n, sparsity, seed=p + n, round_to=1 if p == 1 else 4)
        t = [T(x, dev) for x in (ri, ro, ci)]
        left = T(rng.uniform(-1, 1, (replicas, m, k)).astype(np.float32), dev)
        plan = ops.sddmm_sum_plan(m, n, k, *t)
        topo.append(t)
        lhs.append(left)
        plans.append(plan)
        want.append(ops.sddmm_sum_planned(m, n, *t, left, rhs, plan))
    got = ops.sddmm_sum_group_planned(m, n, [t[0] for t in topo], [t[1] for t in topo],
                                      [t[2] for t in topo], lhs, rhs, plans)
    assert len(got) == count
    for g, w in zip(got, want):
        assert g.shape == w.shape and torch.equal(g, w)
    with pytest.raises(RuntimeError):   # five products: more than one launch adds
        ops.sddmm_sum_group_planned(m, n, [topo[0][0]] * 5, [topo[0][1]] * 5, [topo[0][2]] * 5,
                                    [lhs[0]] * 5, rhs, [plans[0]] * 5)


@pytest.mark.parametrize("m,n,width,replicas,block,left", [
    (512, 512, 1024, 2, 64, True),     # the input gradient of a projection, head split
    (1024, 1024, 64, 4, 64, False),    # round 5: attention dV / dK, TWO panels with cut rows
])
def test_spmm_transposed_out_with_permuted_values(ts, dev, m, n, width, replicas, block, left):
    """Transposed topology, values gathered through the permutation INSIDE the kernel,
    product stored in blocks: bit-identical to the permutation as a pass of its own
    followed by the product (round 5: also with two panels, whose rows the kernel cuts
    at the panel boundary so that every value is gathered once)."""
    a, vals, ri, ro, ci = make_csr(m, n, 0.9, seed=3)
    rng = np.random.default_rng(4)
    v = rng.uniform(-1, 1, len(ci) if left else (replicas, len(ci))).astype(np.float32)
    x = rng.uniform(-1, 1, (replicas, m, width)).astype(np.float32)
    _, ro_t, ci_t, perm = ts.csr_transpose_with_permutation(m, n, T(v if left else v[0], dev), T(ro, dev),
                                                            T(ci, dev))
    from torch_sputnik_amd import ops
    from torch_sputnik_amd.topology import diffsort
    assert ops.spmm_permuted_fused(n, m, width, len(ci))
    ri_t = diffsort(ro_t)
    got = ts.spmm_transposed_out(n, m, T(v, dev), ri_t, ro_t, ci_t, T(x, dev), block,
                                 permutation=perm, left=left)
    plain = ts.spmm_permuted(n, m, T(v, dev), perm, ri_t, ro_t, ci_t, T(x, dev), left=left)
    moved = plain.reshape(replicas * (n // block), block, width).transpose(1, 2)
    assert torch.equal(got, moved.contiguous())
    two_step = ts.spmm_transposed_out(n, m, ts.permute_last(T(v, dev), perm), ri_t, ro_t, ci_t, T(x, dev),
                                      block, left=left)
    assert torch.equal(got, two_step)


@pytest.mark.parametrize("m,k,n,replicas,count,block", [
    (512, 512, 1024, 3, 3, 64),    # q, k, v projections of config 3, head split
    (512, 512, 1024, 2, 4, 0),     # plain stores, rows dealt from row_indices
    (320, 200, 72, 2, 2, 64),      # ragged m, partial column tile
    (256, 96, 64, 1, 3, 256),      # panel smaller than a tile phase
    (128, 600, 64, 2, 2, 64),      # k > 512: one by one
    (192, 128, 20, 2, 3, 64),      # n < 64: one by one
])
def test_left_spmm_group_equals_the_single_products(ts, dev, m, k, n, replicas, count, block):
    rng = np.random.default_rng(m + k + count)
    mats = [make_csr(m, k, 0.9 if p else 0.7, seed=10 * m + p, order="descending")
            for p in range(count)]
    vals = [T(rng.uniform(-1, 1, len(mat[4])).astype(np.float32), dev) for mat in mats]
    topo = [[T(x, dev) for x in mat[2:5]] for mat in mats]
    b = T(rng.uniform(-1, 1, (replicas, k, n)).astype(np.float32), dev)
    outs = ts.left_spmm_group(m, k, vals, [t[0] for t in topo], [t[1] for t in topo],
                              [t[2] for t in topo], b, block)
    assert len(outs) == count
    for p in range(count):
        if block:
            want = ts.spmm_transposed_out(m, k, vals[p], *topo[p], b, block, left=True)
        else:
            want = ts.left_spmm(m, k, vals[p], *topo[p], b)
        assert outs[p].shape == want.shape
        assert rel_err(outs[p].cpu().numpy(), want.cpu().numpy()) < TOL, p
    # against the dense definition as well (first matrix)
    dense = np.zeros((m, k))
    dense[np.repeat(np.arange(m), np.diff(mats[0][3])), mats[0][4]] = vals[0].cpu().numpy()
    ref = np.einsum("mk,rkn->rmn", dense, b.cpu().numpy().astype(np.float64))
    if block:
        ref = ref.reshape(replicas * (m // block), block, n).transpose(0, 2, 1)
    assert rel_err(outs[0].cpu().numpy(), np.ascontiguousarray(ref).astype(np.float32)) < TOL


def test_spmm_group_capi_struct_array(capi, dev):
    """sputnik_hip_spmm_group_batched through ctypes (array of sputnik_hip_spmm_problem):
    two weights, one input, plain stores -- against the single C-ABI products -- and
    the refusals the header promises."""
    m, k, n, replicas = 256, 320, 128, 2
    rng = np.random.default_rng(12)
    b = T(rng.uniform(-1, 1, (replicas, k, n)).astype(np.float32), dev)
    problems, wants = [], []
    for p in range(2):
        _, _, ri, ro, ci = make_csr(m, k, 0.85, seed=40 + p)
        v = T(rng.uniform(-1, 1, len(ci)).astype(np.float32), dev)
        topo = [T(x, dev) for x in (ri, ro, ci)]
        out = torch.full((replicas, m, n), float("nan"), device=dev)
        problems.append({"row_indices": topo[0], "row_offsets": topo[1], "column_indices": topo[2],
                         "values": v, "dense": b, "out": out})
        ws = torch.empty(capi.spmm_workspace_bytes(m, k, n, len(ci)) + 16, dtype=torch.uint8, device=dev)
        wants.append(capi.spmm_batched(m, k, n, replicas, topo[0], v, 0, topo[1], topo[2], b,
                                       torch.empty(replicas, m, n, device=dev), ws))
    assert capi.spmm_group_batched(m, k, n, replicas, problems) == 0
    for p in range(2):
        assert rel_err(problems[p]["out"].cpu().numpy(), wants[p].cpu().numpy()) < TOL
    # summed into one output
    total = torch.full((replicas, m, n), float("nan"), device=dev)
    for p in problems:
        p["out"] = total
    assert capi.spmm_group_batched(m, k, n, replicas, problems, accumulate=True) == 0
    assert rel_err(total.cpu().numpy(), (wants[0] + wants[1]).cpu().numpy()) < TOL
    # not served: transposed stores together with accumulation; k beyond one panel
    assert capi.spmm_group_batched(m, k, n, replicas, problems, block_rows=64, accumulate=True) == -2
    assert capi.lib().sputnik_hip_spmm_group_supported(m, 600, n, 2, 0, 0) == 0
    assert capi.lib().sputnik_hip_spmm_group_supported(m, k, n, 5, 0, 0) == 0


@pytest.mark.parametrize("m,k,n,replicas,count,permuted", [
    (512, 512, 1024, 3, 3, True),    # input gradient of the q, k, v projections
    (512, 512, 1024, 2, 2, False),
    (320, 192, 72, 2, 4, True),
    (128, 640, 64, 2, 2, True),      # transposed k = 640 > 512: one by one
])
def test_left_spmm_group_sum(ts, dev, m, k, n, replicas, count, permuted):
    """sum_p A_p^T @ X_p (A_p given by its transposed topology + permutation) against
    the float64 definition and the sum of the single products."""
    from torch_sputnik_amd.topology import diffsort
    rng = np.random.default_rng(k + count)
    want = np.zeros((replicas, k, n))
    vals, perms, ris, ros, cis, xs = [], [], [], [], [], []
    for p in range(count):
        _, _, ri, ro, ci = make_csr(m, k, 0.9, seed=7 * m + p)
        v = rng.uniform(-1, 1, len(ci)).astype(np.float32)
        x = rng.uniform(-1, 1, (replicas, m, n)).astype(np.float32)
        dense = np.zeros((m, k))
        dense[np.repeat(np.arange(m), np.diff(ro)), ci] = v
        want += np.einsum("mk,rmn->rkn", dense, x.astype(np.float64))
        vt, ro_t, ci_t, perm = ts.csr_transpose_with_permutation(m, k, T(v, dev), T(ro, dev), T(ci, dev))
        vals.append(T(v, dev) if permuted else vt)
        perms.append(perm)
        ris.append(diffsort(ro_t)); ros.append(ro_t); cis.append(ci_t)
        xs.append(T(x, dev))
    got = ts.left_spmm_group_sum(k, m, vals, perms if permuted else [], ris, ros, cis, xs)
    assert got.shape == (replicas, k, n)
    assert rel_err(got.cpu().numpy(), want.astype(np.float32)) < TOL


@pytest.mark.parametrize("n,rows", [(104857, 64), (16384, 3), (16385, 2), (50001, 5), (100, 4), (40000, 1)])
def test_permute_last_banded_equals_the_plain_gather(dev, n, rows):
    from torch_sputnik_amd import ops
    g = torch.Generator(device="cpu").manual_seed(n)
    perm = torch.randperm(n, generator=g).to(torch.int32).to(dev)
    values = torch.rand(rows, n, generator=g).to(dev)
    lists = ops.banded_lists(perm)
    band = ops.permute_band_size()
    # the lists are what the header says they are
    t = torch.arange(n, device=dev)
    assert torch.equal(perm.long()[lists[0].long()], (t // band) * band + lists[1].long())
    got = ops.permute_last_banded(values, *lists)
    assert torch.equal(got, values[:, perm.long()])
    assert torch.equal(got, ops.permute_last(values, perm))
    # rows that are not 16-byte aligned take the scalar copy
    shifted = torch.rand(rows * n + 1, generator=g).to(dev)[1:].reshape(rows, n)
    assert torch.equal(ops.permute_last_banded(shifted, *lists), shifted[:, perm.long()])


def test_planned_attention_and_modules(ts, dev):
    from torch_sputnik_amd.modules import SparseAttention
    rng = np.random.default_rng(8)
    m = n = 256
    _, _, ri, ro, ci = make_csr(m, n, 0.9, seed=2, order="ascending")
    topo = [T(x, dev) for x in (ri, ro, ci)]
    for d in (64, 32):
        q, k, v = (T(rng.uniform(-1, 1, (6, m, d)).astype(np.float32), dev) for _ in range(3))
        plan = ts.sparse_attention_plan(m, n, d, *topo)
        for _ in range(2):
            assert torch.equal(ts.sparse_attention_planned(q, k, v, *topo, 0.2, plan),
                               ts.sparse_attention(q, k, v, *topo, 0.2))
    torch.manual_seed(0)
    layer = SparseAttention(num_heads=2, embedding_size=128, max_sequence_length=256, device=dev,
                            sparsity=0.9, mask_generator=np.random.default_rng(5))
    for lin in layer.linears:
        # unit-variance projections: with raw randn weights the scores reach +-40 and
        # the softmax amplifies fp32 rounding of two different kernel paths
        lin.weight = torch.nn.Parameter(torch.randn(128, 128, device=dev) / 6.0 *
                                        (torch.rand(128, 128, device=dev) < 0.3))
        lin.setup_sparse_tensors()
    x = torch.randn(3, 256, 128, device=dev)
    with torch.no_grad():
        from torch_sputnik_amd import functional
        functional.clear_caches()
        first = layer(x, x, x)     # builds the plans: output projection + attention mask
        plans = len(functional._plans._entries)   # (the grouped q/k/v projections need none)
        assert plans == 2
        second = layer(x, x, x)    # reuses them
        assert len(functional._plans._entries) == plans
    assert torch.equal(first, second)
    # the differentiable path (per-call pre-passes, separate operators) agrees
    y = layer(x.requires_grad_(True), x, x)
    assert rel_err(y.detach().cpu().numpy(), first.cpu().numpy()) < TOL


def test_hip_graph_replay_of_a_module_forward(dev):
    from torch_sputnik_amd.graphs import capture_forward
    from torch_sputnik_amd.modules import SparseAttention
    torch.manual_seed(1)
    layer = SparseAttention(num_heads=2, embedding_size=128, max_sequence_length=256, device=dev,
                            sparsity=0.9, mask_generator=np.random.default_rng(6))
    for lin in layer.linears:
        lin.weight = torch.nn.Parameter(torch.randn(128, 128, device=dev) *
                                        (torch.rand(128, 128, device=dev) < 0.3))
        lin.setup_sparse_tensors()
    x = torch.randn(2, 256, 128, device=dev)
    fast = capture_forward(layer, x, x, x)
    with torch.no_grad():
        want = layer(x, x, x)
    assert torch.equal(fast(x, x, x), want)
    x2 = torch.randn(2, 256, 128, device=dev)
    with torch.no_grad():
        want2 = layer(x2, x2, x2)
    assert torch.equal(fast(x2, x2, x2), want2)
    with pytest.raises(ValueError):
        fast(x2[:1], x2[:1], x2[:1])


def test_hip_graph_replay_of_a_training_step(dev):
    """Forward + backward of the whole SparseAttention module as ONE hipGraph: output,
    input gradient and every projection's value gradient bit-identical to the eager step
    (the library keeps no floating-point atomics), also for new inputs and a new incoming
    gradient, replay after replay."""
    from torch_sputnik_amd.graphs import capture_training_step
    from torch_sputnik_amd.modules import SparseAttention
    torch.manual_seed(2)
    layer = SparseAttention(num_heads=2, embedding_size=128, max_sequence_length=256, device=dev,
                            sparsity=0.9, mask_generator=np.random.default_rng(7),
                            differentiable_softmax=True)
    for lin in layer.linears:
        lin.weight = torch.nn.Parameter(torch.randn(128, 128, device=dev) *
                                        (torch.rand(128, 128, device=dev) < 0.3))
        lin.setup_sparse_tensors()

    def eager(x, g):
        xg = x.clone().requires_grad_(True)
        for lin in layer.linears:
            lin.values.grad = None
        out = layer(xg, xg, xg)
        out.backward(g)
        return out.detach().clone(), xg.grad.clone(), [lin.values.grad.clone() for lin in layer.linears]

    x, g = torch.randn(2, 256, 128, device=dev), torch.randn(2, 256, 128, device=dev)
    want = eager(x, g)
    step = capture_training_step(layer, x, x, x, grad_output=g)
    static = {id(p): gp for p, gp in zip(step.params, step.param_grads)}
    assert all(lin.values.grad is static[id(lin.values)] for lin in layer.linears)   # (for an optimizer)
    for xi, gi, wi in ((x, g, want), (x * 0.5 + 1.0, g * 2.0, None), (x, g, want)):
        wi = wi or eager(xi, gi)
        out = step(xi, xi, xi, grad_output=gi)
        torch.cuda.synchronize()
        assert torch.equal(out, wi[0])
        assert len(step.input_grads) == 1 and torch.equal(step.input_grads[0], wi[1])
        grads = {id(p): gp for p, gp in zip(step.params, step.param_grads)}
        for lin, gv in zip(layer.linears, wi[2]):
            assert torch.equal(grads[id(lin.values)], gv)
            assert grads[id(lin.weight)] is None      # (the reference's unused dense weight)
        wi = None
    with pytest.raises(ValueError):
        step(x[:1], x[:1], x[:1])


# ----------------------------------------------------------------------------
# layout pass: batched 2-D transpose (modules/sparse_linear.py:89,
# modules/sparse_attention.py:108-126)
# ----------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(3, 72, 128), (8, 512, 2048), (2, 4, 1024, 64), (5, 70, 33),
                                   (1, 1, 7), (64, 64, 1024), (100, 260)])
def test_transpose_last2_is_bit_exact(ts, dev, shape):
    from torch_sputnik_amd import ops
    x = torch.randn(*shape, device=dev)
    got = ops.transpose_last2(x)
    assert got.is_contiguous() and torch.equal(got, x.transpose(-1, -2).contiguous())
    # a non-contiguous view in, other dtypes through ATen
    view = x.transpose(-1, -2)
    assert torch.equal(ops.transpose_last2(view), x.contiguous())
    assert torch.equal(ops.transpose_last2(x.half()), x.half().transpose(-1, -2).contiguous())


def test_transpose_last2_gradient(dev):
    from torch_sputnik_amd import functional
    x = torch.randn(4, 96, 40, device=dev, requires_grad=True)
    g = torch.randn(4, 40, 96, device=dev)
    functional.transpose_last2(x).backward(g)
    assert torch.equal(x.grad, g.transpose(1, 2).contiguous())


@pytest.mark.parametrize("src,dst", [(torch.float16, torch.float32), (torch.bfloat16, torch.float32),
                                     (torch.float32, torch.float16), (torch.float32, torch.bfloat16),
                                     (torch.float16, torch.float16)])
@pytest.mark.parametrize("shape", [(8, 512, 2048), (3, 70, 33), (2, 4, 64, 128)])
def test_transpose_last2_with_storage_change(dev, src, dst, shape):
    """Half-precision activations are widened (and gradients narrowed) INSIDE the
    layout pass: bit-identical to transposing and then converting."""
    from torch_sputnik_amd import ops
    x = torch.randn(*shape, device=dev).to(src)
    got = ops.transpose_last2(x, dst)
    want = x.transpose(-1, -2).contiguous().to(dst)
    assert got.dtype == dst and got.is_contiguous() and torch.equal(got, want)


@pytest.mark.parametrize("rows,n", [(1, 1000), (64, 104860), (8, 838864), (3, 17)])
def test_permute_last_matches_index_select(dev, rows, n):
    from torch_sputnik_amd import ops
    g = torch.Generator(device=dev).manual_seed(n)
    values = torch.randn(rows, n, device=dev, generator=g)
    perm = torch.randperm(n, device=dev, generator=g).to(torch.int32)
    assert torch.equal(ops.permute_last(values, perm), values.index_select(-1, perm.long()))
    assert torch.equal(ops.permute_last(values[0].contiguous(), perm), values[0][perm.long()])
