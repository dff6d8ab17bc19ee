"""Native half-precision storage (BASELINE config 5): the
kernels read and write float16 / bfloat16 directly, all arithmetic is float32.

The reference has no half path (data_ptr<float>(), src/spmm_cuda.cu:42,51), so
the oracle is the float64 / C restatement evaluated ON THE ROUNDED INPUTS; what
differs from the float32 tests is only the rounding of a half OUTPUT: `half_err`
takes one unit in the last place of the storage type off the difference and holds
the rest to the north star's 1e-4.  Outputs that stay float32 are held to 1e-4
as they are.
"""
import numpy as np
import pytest
import torch

from oracle import c_oracle
from oracle import sputnik_oracle as O
from tests.helpers import make_csr, rel_err

pytestmark = pytest.mark.gpu

TOL = 1e-4                                                  # float32 outputs
HALF_TYPES = [torch.float16, torch.bfloat16]


def ulp(want, dtype):
    """Spacing of the storage type at |want| (float16 subnormals included)."""
    a = np.abs(np.asarray(want, np.float64))
    if dtype == torch.float16:
        return np.spacing(np.minimum(a, 65000.0).astype(np.float16)).astype(np.float64)
    if dtype == torch.bfloat16:
        return 2.0 ** (np.floor(np.log2(np.maximum(a, 2.0 ** -126))) - 7)
    return np.zeros_like(a)


def half_err(got, want, dtype, row_offsets=None):
    """rel_err of what is left of |got - want| after ONE unit in the last place of
    the output's storage type (its rounding) has been taken off: the test is
    ``half_err(...) < 1e-4``, the float32 bound on the arithmetic."""
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    diff = got - want
    rest = np.sign(diff) * np.maximum(np.abs(diff) - ulp(want, dtype), 0.0)
    return rel_err(want + rest, want, row_offsets)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def capi():
    from torch_sputnik_amd import capi
    assert "gfx950" in capi.version()
    return capi


@pytest.fixture(scope="module")
def ts():
    import torch_sputnik
    return torch_sputnik


def T(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def rounded(x, dtype, dev):
    """(device tensor in `dtype`, the same values as float32 numpy)."""
    t = T(np.asarray(x, np.float32), dev).to(dtype)
    return t, t.float().cpu().numpy()


# ----------------------------------------------------------------------------
# sparse softmax and its gradient
# ----------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", HALF_TYPES)
@pytest.mark.parametrize("m,n,sparsity,replicas", [
    (72, 72, 0.5, 1), (300, 700, 0.85, 3), (1024, 1024, 0.9, 4), (64, 3000, 0.5, 2),
    (100, 2000, 0.97, 2),   # short rows
    (40, 9000, 0.5, 1),     # rows longer than the largest window: strided passes
])
def test_softmax_half_capi_vs_oracle(capi, dev, dtype, m, n, sparsity, replicas):
    _, vals, ri, ro, ci = make_csr(m, n, sparsity, seed=m + n, round_to=1, empty_rows=(0, m - 1))
    rng = np.random.default_rng(m)
    v, v32 = rounded(rng.uniform(-8, 8, size=(replicas, len(vals))), dtype, dev)
    want = c_oracle.sparse_softmax(v32, ro, ci)
    out = torch.full((replicas, len(vals)), float("nan"), device=dev, dtype=dtype)
    capi.sparse_softmax_typed(m, replicas, v, T(ri, dev), T(ro, dev), T(ci, dev), 1.0, out)
    got = out.float().cpu().numpy()
    assert not np.isnan(got).any()
    assert half_err(got, want, dtype, ro) < TOL
    # scale folded in
    capi.sparse_softmax_typed(m, replicas, v, T(ri, dev), T(ro, dev), T(ci, dev), 0.125, out)
    want = c_oracle.sparse_softmax((v32.astype(np.float64) * 0.125).astype(np.float32), ro, ci)
    assert half_err(out.float().cpu().numpy(), want, dtype, ro) < 2 * TOL


@pytest.mark.parametrize("dtype", HALF_TYPES)
@pytest.mark.parametrize("in_phase,out_phase", [(0, 0), (1, 1), (3, 2), (2, 0), (5, 5), (7, 3)])
def test_softmax_half_unaligned_buffers(capi, dev, dtype, in_phase, out_phase):
    """Half rows move as aligned 8-byte pieces: any 2-byte aligned start, odd
    replica strides and outputs aligned differently from the inputs give the same
    answer, and nothing outside the output is written."""
    m, n, replicas = 300, 700, 3
    _, vals, ri, ro, ci = make_csr(m, n, 0.85, seed=m + n, round_to=1, empty_rows=(0, m // 2))
    nnz = len(vals)
    rng = np.random.default_rng(m)
    v, v32 = rounded(rng.uniform(-6, 6, size=(replicas, nnz)), dtype, dev)
    want = c_oracle.sparse_softmax(v32, ro, ci)
    src = torch.zeros(replicas * nnz + 16, device=dev, dtype=dtype)
    src[in_phase:in_phase + replicas * nnz] = v.reshape(-1)
    dst = torch.full((replicas * nnz + 16,), 7.0, device=dev, dtype=dtype)
    capi.sparse_softmax_typed(m, replicas, src[in_phase:in_phase + replicas * nnz].view(replicas, nnz),
                              T(ri, dev), T(ro, dev), T(ci, dev), 1.0,
                              dst[out_phase:out_phase + replicas * nnz].view(replicas, nnz))
    got = dst.float().cpu().numpy()
    assert np.all(got[:out_phase] == 7.0) and np.all(got[out_phase + replicas * nnz:] == 7.0)
    assert half_err(got[out_phase:out_phase + replicas * nnz].reshape(replicas, nnz), want, dtype,
                    ro) < TOL


@pytest.mark.parametrize("dtype", HALF_TYPES)
@pytest.mark.parametrize("m,n,sparsity,replicas", [(72, 72, 0.5, 1), (300, 700, 0.85, 3),
                                                   (1024, 1024, 0.9, 4), (40, 9000, 0.5, 1)])
def test_softmax_backward_half_capi_vs_oracle(capi, dev, dtype, m, n, sparsity, replicas):
    _, vals, ri, ro, ci = make_csr(m, n, sparsity, seed=m + n, round_to=1, empty_rows=(0,))
    rng = np.random.default_rng(m + 1)
    x = rng.uniform(-4, 4, size=(replicas, len(vals))).astype(np.float32)
    y, y32 = rounded(c_oracle.sparse_softmax(x, ro, ci), dtype, dev)
    g, g32 = rounded(rng.uniform(-1, 1, size=y32.shape), dtype, dev)
    scale = 0.25
    want = O.sparse_softmax_backward(y32, g32, ro, scale)
    out = torch.full(y.shape, float("nan"), device=dev, dtype=dtype)
    capi.sparse_softmax_backward_typed(m, replicas, y, g, T(ro, dev), scale, out)
    got = out.float().cpu().numpy()
    assert not np.isnan(got).any()
    assert half_err(got, want, dtype, ro) < TOL


@pytest.mark.parametrize("dtype", HALF_TYPES)
def test_softmax_half_op_keeps_the_storage_type(ts, dev, dtype):
    """The torch op hands half values to the kernel as they are (no widening
    copy) and returns the values' type; mixed operands of the gradient take the
    wider type."""
    import torch_sputnik_amd as tsa
    m, n = 128, 512
    _, vals, ri, ro, ci = make_csr(m, n, 0.8, seed=9, round_to=1)
    v, v32 = rounded(np.random.default_rng(2).uniform(-5, 5, size=(2, len(vals))), dtype, dev)
    y = ts.sparse_softmax(v, T(ri, dev), T(ro, dev), T(ci, dev))
    assert y.dtype == dtype and y.shape == v.shape
    want = c_oracle.sparse_softmax(v32, ro, ci)
    assert half_err(y.float().cpu().numpy(), want, dtype, ro) < TOL
    g = T(np.random.default_rng(3).uniform(-1, 1, size=tuple(y.shape)).astype(np.float32), dev).to(dtype)
    dx = tsa.ops.sparse_softmax_backward(y, g, T(ro, dev), 1.0)
    assert dx.dtype == dtype
    dx32 = tsa.ops.sparse_softmax_backward(y.float(), g, T(ro, dev), 1.0)
    assert dx32.dtype == torch.float32
    want_dx = O.sparse_softmax_backward(y.float().cpu().numpy(), g.float().cpu().numpy(), ro, 1.0)
    assert rel_err(dx32.cpu().numpy(), want_dx, ro) < TOL
    assert half_err(dx.float().cpu().numpy(), want_dx, dtype, ro) < TOL


# ----------------------------------------------------------------------------
# SDDMM: half operands read as they are (LDS slab in half, v_dot2 products)
# ----------------------------------------------------------------------------
SDDMM_SHAPES = [
    # m, k, n, sparsity, replicas
    (72, 64, 72, 0.0, 1),      # tests/test_sddmm.py: dense mask
    (72, 64, 72, 0.9, 4),
    (50, 7, 60, 0.7, 2),       # odd k: scalar loads (row-wave kernel)
    (50, 10, 60, 0.7, 1),
    (128, 32, 128, 0.9, 3),
    (1024, 64, 1024, 0.9, 2),  # attention block geometry (config 3)
    (200, 300, 150, 0.8, 1),
    (64, 1100, 96, 0.9, 2),    # k > 1024 on the row-wave kernel: several panels
    (300, 64, 500, 0.8, 3),    # quad kernel, ragged row / column blocks
    (256, 128, 256, 0.5, 2),   # k = 128, > 32 entries per row and slab
    (64, 64, 64, 0.0, 1),      # dense mask: full windows
    (300, 256, 200, 0.8, 2),   # k = 256 (128-row slabs)
    (512, 512, 512, 0.8, 2),   # k = 512: the plan's 64-row slabs, two panels of 256
    (130, 512, 70, 0.3, 1),
    (256, 1024, 96, 0.7, 2),   # four panels of 256 accumulate
    (200, 768, 130, 0.8, 2),
    (96, 320, 200, 0.6, 1),    # five panels of 64
]


@pytest.fixture(params=["auto", "tiled", "wave"])
def sddmm_kernel(request, monkeypatch):
    from torch_sputnik_amd import capi
    if request.param == "auto":
        monkeypatch.delenv("SPUTNIK_HIP_SDDMM_KERNEL", raising=False)
    else:
        monkeypatch.setenv("SPUTNIK_HIP_SDDMM_KERNEL", request.param)
    capi.reload_options()
    yield request.param
    monkeypatch.delenv("SPUTNIK_HIP_SDDMM_KERNEL", raising=False)
    capi.reload_options()


@pytest.mark.parametrize("dtype", HALF_TYPES)
@pytest.mark.parametrize("m,k,n,sparsity,replicas", SDDMM_SHAPES)
def test_sddmm_half_capi_vs_oracle(capi, dev, sddmm_kernel, dtype, m, k, n, sparsity, replicas):
    """float32 output of half operands: exact products, float32 sums -- the float32
    bound against the oracle on the rounded operands."""
    _, _, ri, ro, ci = make_csr(m, n, sparsity, seed=m + k + n, round_to=1, empty_rows=(m // 2,))
    rng = np.random.default_rng(k)
    lhs, lhs32 = rounded(rng.uniform(-1, 1, size=(replicas, m, k)), dtype, dev)
    rhs, rhs32 = rounded(rng.uniform(-1, 1, size=(replicas, n, k)), dtype, dev)
    want = c_oracle.sddmm(m, n, ro, ci, lhs32, rhs32)
    out = torch.full((replicas, len(ci)), float("nan"), device=dev)
    ws = torch.empty(capi.sddmm_workspace_bytes(m, k, n, len(ci)) + 16, dtype=torch.uint8, device=dev)
    topo = (T(ri, dev), T(ro, dev), T(ci, dev))
    capi.sddmm_typed(m, k, n, replicas, *topo, lhs, rhs, out, ws)
    got = out.cpu().numpy()
    assert not np.isnan(got).any()
    assert rel_err(got, want, ro) < TOL
    # planned form: same plan as the float operator's; bit-identical result when the same
    # kernel runs, the pair-flat kernel (round 4: planned products with 128 / 256-byte rows)
    # against the oracle -- which quad computes an entry decides the order of its partial sums
    capi.sddmm_plan(m, k, n, *topo, ws)
    out2 = torch.full_like(out, float("nan"))
    capi.sddmm_typed(m, k, n, replicas, *topo, lhs, rhs, out2, ws, planned=True)
    if capi.sddmm_kernel_name(m, k, n, len(ci), replicas, lhs.element_size(), planned=True) == \
            capi.sddmm_kernel_name(m, k, n, len(ci), replicas, lhs.element_size(), planned=False):
        assert torch.equal(out, out2)
    else:
        got2 = out2.cpu().numpy()
        assert not np.isnan(got2).any()
        assert rel_err(got2, want, ro) < TOL


@pytest.mark.parametrize("dtype", HALF_TYPES)
@pytest.mark.parametrize("m,k,n,sparsity,replicas", [
    (1024, 64, 1024, 0.9, 2), (300, 64, 500, 0.8, 3), (256, 128, 256, 0.5, 2),
    (300, 256, 200, 0.8, 2), (50, 10, 60, 0.7, 1), (128, 32, 128, 0.9, 3)])
def test_sddmm_half_output(capi, dev, sddmm_kernel, dtype, m, k, n, sparsity, replicas):
    """Half OUTPUT (single-pass shapes): the float32 sum rounded once."""
    _, _, ri, ro, ci = make_csr(m, n, sparsity, seed=m + k + n, round_to=1)
    rng = np.random.default_rng(k + 7)
    lhs, lhs32 = rounded(rng.uniform(-1, 1, size=(replicas, m, k)), dtype, dev)
    rhs, rhs32 = rounded(rng.uniform(-1, 1, size=(replicas, n, k)), dtype, dev)
    want = c_oracle.sddmm(m, n, ro, ci, lhs32, rhs32)
    out = torch.full((replicas, len(ci)), float("nan"), device=dev, dtype=dtype)
    ws = torch.empty(capi.sddmm_workspace_bytes(m, k, n, len(ci)) + 16, dtype=torch.uint8, device=dev)
    capi.sddmm_typed(m, k, n, replicas, T(ri, dev), T(ro, dev), T(ci, dev), lhs, rhs, out, ws)
    got = out.float().cpu().numpy()
    assert not np.isnan(got).any()
    assert half_err(got, want, dtype, ro) < TOL


def test_sddmm_half_output_needs_one_pass(capi, dev):
    m, k, n = 256, 2048, 96
    _, _, ri, ro, ci = make_csr(m, n, 0.7, seed=3, round_to=1)
    lhs = torch.zeros(1, m, k, device=dev, dtype=torch.float16)
    rhs = torch.zeros(1, n, k, device=dev, dtype=torch.float16)
    out = torch.zeros(1, len(ci), device=dev, dtype=torch.float16)
    ws = torch.empty(capi.sddmm_workspace_bytes(m, k, n, len(ci)) + 16, dtype=torch.uint8, device=dev)
    with pytest.raises(RuntimeError, match="status -2"):    # SPUTNIK_HIP_UNSUPPORTED
        capi.sddmm_typed(m, k, n, 1, T(ri, dev), T(ro, dev), T(ci, dev), lhs, rhs, out, ws)


@pytest.mark.parametrize("dtype", HALF_TYPES)
@pytest.mark.parametrize("planned", [False, True])
@pytest.mark.parametrize("m,k,n,sparsity,replicas", [
    (512, 1024, 512, 0.9, 8), (512, 512, 512, 0.8, 3), (256, 1024, 96, 0.7, 1),
    (200, 768, 130, 0.8, 2), (100, 40, 60, 0.5, 5), (128, 384, 200, 0.8, 3),
    (2048, 512, 2048, 0.8, 2)])
def test_sddmm_sum_half_capi_vs_oracle(capi, dev, sddmm_sum_slab, dtype, planned, m, k, n, sparsity,
                                       replicas):
    _, _, ri, ro, ci = make_csr(m, n, sparsity, seed=m + k + n, empty_rows=(m // 2,))
    rng = np.random.default_rng(k + 1)
    lhs, lhs32 = rounded(rng.uniform(-1, 1, size=(replicas, m, k)), dtype, dev)
    rhs, rhs32 = rounded(rng.uniform(-1, 1, size=(replicas, n, k)), dtype, dev)
    want = c_oracle.sddmm(m, n, ro, ci, lhs32, rhs32).astype(np.float64).sum(axis=0)
    out = torch.full((len(ci),), float("nan"), device=dev)
    ws = torch.empty(capi.sddmm_sum_workspace_bytes(m, k, n, len(ci)) + 16, dtype=torch.uint8, device=dev)
    scratch = torch.empty(capi.sddmm_sum_scratch_bytes(m, k, n, len(ci), replicas) + 16,
                          dtype=torch.uint8, device=dev)
    topo = (T(ri, dev), T(ro, dev), T(ci, dev))
    if planned:
        capi.sddmm_sum_plan(m, k, n, *topo, ws)
    capi.sddmm_sum_typed(m, k, n, replicas, *topo, lhs, rhs, out, ws, scratch, planned=planned)
    got = out.cpu().numpy()
    assert not np.isnan(got).any()
    assert rel_err(got[None, :], want[None, :].astype(np.float32), ro) < TOL


@pytest.fixture(params=["tile_rows_128", "tile_rows_256"])
def mfma_tile(request, monkeypatch):
    """SPUTNIK_HIP_MFMA_TILE: the matrix-core kernels' tile has 128 rows (four waves, two LDS
    stages, two workgroups per CU) or, where the output has 256 rows or more, 256 (eight
    waves, three stages); by itself the library picks by shape."""
    from torch_sputnik_amd import capi
    monkeypatch.setenv("SPUTNIK_HIP_MFMA_TILE", request.param.rsplit("_", 1)[1])
    capi.reload_options()
    yield request.param
    monkeypatch.delenv("SPUTNIK_HIP_MFMA_TILE", raising=False)
    capi.reload_options()


@pytest.fixture
def sddmm_mfma(monkeypatch):
    """SPUTNIK_HIP_SDDMM_KERNEL=mfma: the summed product of half operands takes the
    matrix-core kernel (csrc/sddmm_mfma.hip) for every shape it serves -- small test
    shapes included, which the dispatcher leaves on the vector kernels."""
    from torch_sputnik_amd import capi
    monkeypatch.setenv("SPUTNIK_HIP_SDDMM_KERNEL", "mfma")
    capi.reload_options()
    yield
    monkeypatch.delenv("SPUTNIK_HIP_SDDMM_KERNEL", raising=False)
    capi.reload_options()


def _sddmm_sum_typed(capi, dev, m, k, n, replicas, topo, lhs, rhs, with_scratch=True, planned=False,
                     with_workspace=True):
    nnz = topo[2].numel()
    out = torch.full((nnz,), float("nan"), device=dev)
    ws = torch.empty((capi.sddmm_sum_workspace_bytes(m, k, n, nnz) if with_workspace else 0) + 16,
                     dtype=torch.uint8, device=dev)
    scratch = torch.empty((capi.sddmm_sum_scratch_bytes(m, k, n, nnz, replicas) if with_scratch else 0)
                          + 16, dtype=torch.uint8, device=dev)
    if planned:
        capi.sddmm_sum_plan(m, k, n, *topo, ws)
    capi.sddmm_sum_typed(m, k, n, replicas, *topo, lhs, rhs, out, ws, scratch, planned=planned)
    return out


@pytest.mark.parametrize("dtype", HALF_TYPES)
@pytest.mark.parametrize("m,k,n,sparsity,replicas,round_to", [
    (256, 512, 256, 0.8, 2, 4),      # four tiles, two workgroups per tile
    (128, 64, 128, 0.5, 1, 4),       # one tile, ONE step
    (200, 192, 130, 0.8, 3, 1),      # ragged tiles (rows and columns clamped), nnz % 4 != 0
    (72, 128, 40, 0.5, 5, 1),        # a mask smaller than a tile
    (513, 64, 129, 0.9, 16, 1),      # one-row / one-column last tiles, 16 replicas
    (384, 1024, 640, 0.95, 2, 4),    # very sparse: most rows have no entry in a tile
    (300, 256, 300, 0.0, 2, 4),      # a dense mask: every tile element is sampled
])
def test_sddmm_sum_mfma_vs_oracle(capi, dev, sddmm_mfma, mfma_tile, dtype, m, k, n, sparsity, replicas, round_to):
    """The matrix-core route of the summed SDDMM (the weight gradient of
    modules/sparse_linear.py:44-49 on half storage) against the C oracle on the
    rounded operands: products exact, float32 sums."""
    _, _, ri, ro, ci = make_csr(m, n, sparsity, seed=m + k + n, empty_rows=(m // 2, m - 1),
                                round_to=round_to)
    rng = np.random.default_rng(k + 7)
    lhs, lhs32 = rounded(rng.uniform(-1, 1, size=(replicas, m, k)), dtype, dev)
    rhs, rhs32 = rounded(rng.uniform(-1, 1, size=(replicas, n, k)), dtype, dev)
    want = c_oracle.sddmm(m, n, ro, ci, lhs32, rhs32).astype(np.float64).sum(axis=0)
    topo = (T(ri, dev), T(ro, dev), T(ci, dev))
    out = _sddmm_sum_typed(capi, dev, m, k, n, replicas, topo, lhs, rhs)
    got = out.cpu().numpy()
    assert not np.isnan(got).any()
    assert rel_err(got[None, :], want[None, :].astype(np.float32), ro) < TOL
    # reproducible; the same bits on a kept plan and without any (the tile's entries are
    # found by a walk then: a different epilogue, the same sums); and the same bound without
    # the partial vectors' room (one workgroup per tile reduces everything: another order)
    assert torch.equal(_sddmm_sum_typed(capi, dev, m, k, n, replicas, topo, lhs, rhs), out)
    assert torch.equal(_sddmm_sum_typed(capi, dev, m, k, n, replicas, topo, lhs, rhs, planned=True), out)
    assert torch.equal(_sddmm_sum_typed(capi, dev, m, k, n, replicas, topo, lhs, rhs,
                                        with_workspace=False), out)
    alone = _sddmm_sum_typed(capi, dev, m, k, n, replicas, topo, lhs, rhs, with_scratch=False)
    assert rel_err(alone.cpu().numpy()[None, :], want[None, :].astype(np.float32), ro) < TOL


@pytest.mark.parametrize("dtype", HALF_TYPES)
@pytest.mark.parametrize("shuffled", ["every_row", "one_row"])
def test_sddmm_sum_mfma_unsorted_columns_and_views(capi, dev, sddmm_mfma, mfma_tile, dtype, shuffled):
    """Rows whose columns do not ascend (the plan marks them; a tile that holds one finds
    its entries by a flat walk over its rows' entries, the other tiles by the plan), an
    unaligned column_indices pointer (entry-wise index loads in the walk), and exact
    integer data (every element must be the exact integer)."""
    m, k, n, replicas = 260, 128, 200, 3
    _, _, ri, ro, ci = make_csr(m, n, 0.7, seed=11, round_to=1)
    rng = np.random.default_rng(5)
    ci = ci.copy()
    for r in (range(m) if shuffled == "every_row" else (140,)):
        rng.shuffle(ci[ro[r]:ro[r + 1]])
    assert not np.all(np.diff(ci[ro[140]:ro[141]]) > 0)
    lhs32 = rng.integers(-3, 4, size=(replicas, m, k)).astype(np.float32)
    rhs32 = rng.integers(-3, 4, size=(replicas, n, k)).astype(np.float32)
    lhs, rhs = T(lhs32, dev).to(dtype), T(rhs32, dev).to(dtype)
    want = c_oracle.sddmm(m, n, ro, ci, lhs32, rhs32).astype(np.float64).sum(axis=0)
    padded = torch.zeros(len(ci) + 1, dtype=torch.int32, device=dev)
    padded[1:] = T(ci, dev)
    topo = (T(ri, dev), T(ro, dev), padded[1:])            # 4 bytes off a 16-byte boundary
    for planned in (False, True):
        got = _sddmm_sum_typed(capi, dev, m, k, n, replicas, topo, lhs, rhs, planned=planned).cpu().numpy()
        assert np.array_equal(got, want.astype(np.float32))


def _sddmm_sum_mixed(capi, dev, m, k, n, replicas, topo, lhs, rhs, planned=False):
    nnz = topo[2].numel()
    out = torch.full((nnz,), float("nan"), device=dev)
    ws = torch.empty(capi.sddmm_sum_workspace_bytes(m, k, n, nnz) + 16, dtype=torch.uint8, device=dev)
    scratch = torch.empty(capi.sddmm_sum_mixed_scratch_bytes(m, k, n, nnz, replicas, lhs, rhs) + 16,
                          dtype=torch.uint8, device=dev)
    if planned:
        capi.sddmm_sum_plan(m, k, n, *topo, ws)
    capi.sddmm_sum_mixed(m, k, n, replicas, *topo, lhs, rhs, out, ws, scratch, planned=planned)
    return out


@pytest.mark.parametrize("dtype", HALF_TYPES)
@pytest.mark.parametrize("wide", ["lhs", "rhs"])
@pytest.mark.parametrize("m,k,n,sparsity,replicas,magnitude", [
    (256, 512, 256, 0.8, 2, 1.0),
    (200, 192, 130, 0.8, 3, 1.0),      # ragged tiles
    (128, 64, 384, 0.5, 1, 1e-3),      # small gradients: float16's low plane is kept out of
    (300, 256, 300, 0.7, 4, 3e-4),     # the subnormals by its 2^11 scale (values 3e-7 .. 3e-4)
    (256, 128, 256, 0.7, 2, 900.0),
])
def test_sddmm_sum_mixed_mfma_vs_oracle(capi, dev, sddmm_mfma, mfma_tile, dtype, wide, m, k, n, sparsity,
                                        replicas, magnitude):
    """A (float32, half) pair: the float32 operand -- the incoming gradient of
    modules/sparse_linear.py:44-49 when only the activations are stored in half precision --
    enters the matrix-core product as half planes whose sum is the value, NOT rounded to
    the storage type: held to the float32 bound against float64 on (float32 operand, rounded
    half operand)."""
    _, _, ri, ro, ci = make_csr(m, n, sparsity, seed=m + k + n + 1, empty_rows=(3,), round_to=1)
    rng = np.random.default_rng(k + 17)
    # full float32 mantissas, magnitudes spread over three decades
    wide32 = (rng.uniform(-1, 1, size=(replicas, m if wide == "lhs" else n, k)) *
              magnitude * 10.0 ** rng.uniform(-3, 0, size=(replicas, 1, k))).astype(np.float32)
    half_t, half32 = rounded(rng.uniform(-1, 1, size=(replicas, n if wide == "lhs" else m, k)), dtype, dev)
    lhs32, rhs32 = (wide32, half32) if wide == "lhs" else (half32, wide32)
    lhs, rhs = (T(wide32, dev), half_t) if wide == "lhs" else (half_t, T(wide32, dev))
    want = c_oracle.sddmm(m, n, ro, ci, lhs32, rhs32).astype(np.float64).sum(axis=0)
    topo = (T(ri, dev), T(ro, dev), T(ci, dev))
    got = _sddmm_sum_mixed(capi, dev, m, k, n, replicas, topo, lhs, rhs)
    assert rel_err(got.cpu().numpy()[None, :], want[None, :].astype(np.float32), ro) < TOL
    assert torch.equal(_sddmm_sum_mixed(capi, dev, m, k, n, replicas, topo, lhs, rhs, planned=True), got)


def test_sddmm_sum_mixed_op_takes_the_route_or_widens(ts, dev):
    """torch op: sddmm_sum of (float32, half) operands -- matrix cores where the shape is
    served, the half operand widened everywhere else; both against the oracle."""
    import torch_sputnik_amd as tsa
    for m, k, n, replicas in ((256, 512, 384, 4), (96, 40, 60, 3)):
        _, _, ri, ro, ci = make_csr(m, n, 0.8, seed=k, round_to=1)
        topo = (T(ri, dev), T(ro, dev), T(ci, dev))
        rng = np.random.default_rng(k)
        lhs32 = rng.uniform(-1, 1, size=(replicas, m, k)).astype(np.float32)
        rhs, rhs32 = rounded(rng.uniform(-1, 1, size=(replicas, n, k)), torch.float16, dev)
        want = c_oracle.sddmm(m, n, ro, ci, lhs32, rhs32).astype(np.float64).sum(0)
        total = tsa.ops.sddmm_sum(m, n, *topo, T(lhs32, dev), rhs)
        assert total.dtype == torch.float32
        assert rel_err(total.cpu().numpy()[None], want[None], ro) < TOL
        plan = ts.sddmm_sum_plan(m, n, k, *topo)
        assert torch.equal(ts.sddmm_sum_planned(m, n, *topo, T(lhs32, dev), rhs, plan), total)


@pytest.mark.parametrize("dtype,seq", [(torch.float16, 512), (torch.bfloat16, 512), (torch.float16, 2048)])
def test_sddmm_sum_mfma_config5(capi, dev, dtype, seq):
    """BASELINE config 5's weight gradient at full size -- 2048 x 2048 mask at density
    0.2, batch 8, reduction over seq 512 and over the stated 2048 -- on the route the
    dispatcher picks by itself (no knob), every entry against the C oracle."""
    m = n = 2048
    replicas = 8
    _, _, ri, ro, ci = make_csr(m, n, 0.8, seed=seq)
    rng = np.random.default_rng(seq)
    lhs, lhs32 = rounded(rng.uniform(-1, 1, size=(replicas, m, seq)), dtype, dev)
    rhs, rhs32 = rounded(rng.uniform(-1, 1, size=(replicas, n, seq)), dtype, dev)
    want = np.zeros(len(ci), np.float64)
    for r in range(replicas):
        want += c_oracle.sddmm(m, n, ro, ci, lhs32[r:r + 1], rhs32[r:r + 1])[0]
    topo = (T(ri, dev), T(ro, dev), T(ci, dev))
    got = _sddmm_sum_typed(capi, dev, m, seq, n, replicas, topo, lhs, rhs).cpu().numpy()
    assert rel_err(got[None, :], want[None, :].astype(np.float32), ro) < TOL


@pytest.mark.parametrize("dtype", HALF_TYPES)
def test_sddmm_half_ops(ts, dev, dtype):
    """torch ops: half operands go to the kernel as they are; `sddmm` returns float32
    (the reference's output type), `sddmm_narrow` the operands' type (also when the
    product takes several passes: float32 result rounded once)."""
    import torch_sputnik_amd as tsa
    for m, k, n in ((300, 64, 500), (256, 1024, 96)):
        _, _, ri, ro, ci = make_csr(m, n, 0.8, seed=k, round_to=1)
        topo = (T(ri, dev), T(ro, dev), T(ci, dev))
        rng = np.random.default_rng(k)
        lhs, lhs32 = rounded(rng.uniform(-1, 1, size=(2, m, k)), dtype, dev)
        rhs, rhs32 = rounded(rng.uniform(-1, 1, size=(2, n, k)), dtype, dev)
        want = c_oracle.sddmm(m, n, ro, ci, lhs32, rhs32)
        out = ts.sddmm(m, n, *topo, lhs, rhs)
        assert out.dtype == torch.float32
        assert rel_err(out.cpu().numpy(), want, ro) < TOL
        narrow = tsa.ops.sddmm_narrow(m, n, *topo, lhs, rhs)
        assert narrow.dtype == dtype
        assert half_err(narrow.float().cpu().numpy(), want, dtype, ro) < TOL
        # mixed operands take the wider type
        mixed = ts.sddmm(m, n, *topo, lhs, rhs.float())
        assert rel_err(mixed.cpu().numpy(), want, ro) < TOL
        total = tsa.ops.sddmm_sum(m, n, *topo, lhs, rhs)
        assert total.dtype == torch.float32
        assert rel_err(total.cpu().numpy()[None], want.astype(np.float64).sum(0)[None], ro) < TOL


# ----------------------------------------------------------------------------
# SpMM / left_spmm: half values and / or half dense operand, float32 product
# ----------------------------------------------------------------------------
SPMM_SHAPES = [
    # m, k, n, sparsity, replicas
    (72, 64, 72, 0.5, 1),        # tests/test_spmm.py size: row-gather kernel
    (128, 96, 64, 0.8, 2),
    (50, 60, 7, 0.7, 2),         # n = 7: scalar gather
    (50, 60, 10, 0.7, 1),
    (1024, 1024, 64, 0.9, 8),    # attention P.V geometry: two panels, widened on the way in
    (512, 512, 1024, 0.9, 4),    # projection geometry: one panel
    (300, 200, 132, 0.8, 3),     # ragged tiles (n = 132: partial last column tile)
    (256, 1500, 128, 0.9, 2),    # three panels
    (128, 3000, 64, 0.95, 1),    # k > 2048: row gather
]
SPMM_TYPES = [(torch.float16, torch.float16), (torch.bfloat16, torch.bfloat16),
              (torch.float16, torch.float32), (torch.float32, torch.bfloat16)]


@pytest.fixture(params=["auto", "panel", "gather"])
def half_spmm_kernel(request, monkeypatch):
    from torch_sputnik_amd import capi
    if request.param == "auto":
        monkeypatch.delenv("SPUTNIK_HIP_SPMM_KERNEL", raising=False)
    else:
        monkeypatch.setenv("SPUTNIK_HIP_SPMM_KERNEL", request.param)
    capi.reload_options()
    yield request.param
    monkeypatch.delenv("SPUTNIK_HIP_SPMM_KERNEL", raising=False)
    capi.reload_options()


@pytest.mark.parametrize("tv,tb", SPMM_TYPES)
@pytest.mark.parametrize("m,k,n,sparsity,replicas", SPMM_SHAPES)
def test_spmm_half_capi_vs_oracle(capi, dev, half_spmm_kernel, tv, tb, m, k, n, sparsity, replicas):
    _, vals, ri, ro, ci = make_csr(m, k, sparsity, seed=m + k + n, round_to=1, empty_rows=(m // 3,))
    rng = np.random.default_rng(n)
    v, v32 = rounded(rng.uniform(-1, 1, size=(replicas, len(vals))), tv, dev)
    b, b32 = rounded(rng.uniform(-1, 1, size=(replicas, k, n)), tb, dev)
    want = np.stack([O.spmm(m, k, v32[r], ri, ro, ci, b32[r]) for r in range(replicas)])
    out = torch.full((replicas, m, n), float("nan"), device=dev)
    topo = (T(ri, dev), T(ro, dev), T(ci, dev))
    capi.spmm_typed(m, k, n, replicas, topo[0], v, len(vals), topo[1], topo[2], b, out)
    got = out.cpu().numpy()
    assert not np.isnan(got).any()
    assert rel_err(got, want) < TOL
    # shared values (left_spmm): stride 0
    capi.spmm_typed(m, k, n, replicas, topo[0], v[0].contiguous(), 0, topo[1], topo[2], b, out)
    want0 = np.stack([O.spmm(m, k, v32[0], ri, ro, ci, b32[r]) for r in range(replicas)])
    assert rel_err(out.cpu().numpy(), want0) < TOL


@pytest.fixture
def spmm_mfma(monkeypatch):
    """SPUTNIK_HIP_SPMM_KERNEL=mfma: left_spmm with a half dense operand takes the
    matrix-core kernel (csrc/spmm_mfma.hip) for every shape it serves -- small test shapes
    included, which the dispatcher leaves on the vector kernels."""
    from torch_sputnik_amd import capi
    monkeypatch.setenv("SPUTNIK_HIP_SPMM_KERNEL", "mfma")
    capi.reload_options()
    yield
    monkeypatch.delenv("SPUTNIK_HIP_SPMM_KERNEL", raising=False)
    capi.reload_options()


def _operand(x, kind, tile, dev):
    """(device tensor, the values it holds as float32 numpy): stored in the tile type
    (rounded) or in float32 with full mantissas."""
    if kind == "half":
        return rounded(x, tile, dev)
    x32 = np.asarray(x, np.float32)
    return T(x32, dev), x32


@pytest.mark.parametrize("tile", HALF_TYPES)
@pytest.mark.parametrize("values_kind,dense_kind", [("half", "half"), ("float", "half"), ("half", "float"),
                                                    ("float", "float")])
@pytest.mark.parametrize("m,k,n,sparsity,replicas", [
    (256, 128, 256, 0.8, 2),     # four tiles per replica
    (128, 64, 128, 0.5, 1),      # one tile, ONE step
    (200, 192, 136, 0.8, 3),     # ragged: the last row tile and the last column tile are partial
    (130, 256, 72, 0.7, 2),      # the reference's test width (tests/test_spmm.py:13): one partial tile
    (384, 512, 264, 0.95, 2),    # very sparse
    (136, 2176, 128, 0.9, 1),    # k beyond one 2048-column segment of the image's rows
])
def test_left_spmm_half_tiles_vs_oracle(capi, dev, spmm_mfma, mfma_tile, tile, values_kind, dense_kind, m, k, n,
                                        sparsity, replicas):
    """left_spmm as a dense contraction on the matrix cores: the densified weight against
    [R, k, n], every pairing of (float32 | half) values and dense operand -- a float32
    operand enters as half planes, not rounded: the float32 bound against the oracle on
    the operands as stored."""
    _, vals, ri, ro, ci = make_csr(m, k, sparsity, seed=m + k + n + 3, round_to=1, empty_rows=(m // 3,))
    rng = np.random.default_rng(n + 5)
    # (full float32 mantissas; magnitudes over two decades so that the planes have work)
    v, v32 = _operand(vals * 10.0 ** rng.uniform(-2, 0, size=len(vals)) * rng.choice([-1, 1], size=len(vals)),
                      values_kind, tile, dev)
    b, b32 = _operand(rng.uniform(-1, 1, size=(replicas, k, n)) * 10.0 ** rng.uniform(-2, 0, size=(replicas, k, 1)),
                      dense_kind, tile, dev)
    want = O.left_spmm(m, k, v32, ri, ro, ci, b32)
    ws_bytes = capi.left_spmm_half_tiles_workspace_bytes(m, k, n, len(ci), replicas, v, b, tile)
    if tile == torch.bfloat16 and values_kind == dense_kind == "float":
        assert ws_bytes == 0      # six tile products per step: not served (the vector kernels win)
        return
    assert ws_bytes > 0
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    out = torch.full((replicas, m, n + 8), float("nan"), device=dev)[:, :, :n].contiguous()
    capi.left_spmm_half_tiles(m, k, n, replicas, T(ro, dev), T(ci, dev), v, b, tile, out, ws)
    got = out.cpu().numpy()
    assert not np.isnan(got).any()
    assert rel_err(got, want) < TOL
    # bias + ReLU epilogue, and reproducible
    bias = T(rng.uniform(-1, 1, size=(m,)).astype(np.float32), dev)
    fused = torch.empty_like(out)
    capi.left_spmm_half_tiles(m, k, n, replicas, T(ro, dev), T(ci, dev), v, b, tile, fused, ws,
                              bias=bias, relu=True)
    want_f = np.maximum(want.astype(np.float64) + bias.cpu().numpy()[None, :, None], 0.0)
    assert np.max(np.abs(fused.cpu().numpy() - want_f)) < 1e-4 * max(1.0, np.abs(want_f).max())
    again = torch.empty_like(out)
    capi.left_spmm_half_tiles(m, k, n, replicas, T(ro, dev), T(ci, dev), v, b, tile, again, ws)
    assert torch.equal(again, out)


def test_left_spmm_half_tiles_exact_integers_and_guard(capi, dev, spmm_mfma, mfma_tile):
    """Small integers (every product and sum exact): the tile kernel's fragment maps --
    rows against columns, the transposing reads of the [k][n] operand -- must give the
    exact matrix; nothing outside [R, m, n] is written (ragged tiles)."""
    m, k, n, replicas = 328, 128, 136, 2
    _, _, ri, ro, ci = make_csr(m, k, 0.6, seed=19, round_to=1)
    rng = np.random.default_rng(20)
    v32 = rng.integers(-4, 5, size=len(ci)).astype(np.float32)
    b32 = rng.integers(-4, 5, size=(replicas, k, n)).astype(np.float32)
    want = O.left_spmm(m, k, v32, ri, ro, ci, b32)
    v, b = T(v32, dev).half(), T(b32, dev).half()
    ws = torch.empty(capi.left_spmm_half_tiles_workspace_bytes(m, k, n, len(ci), replicas, v, b, torch.float16),
                     dtype=torch.uint8, device=dev)
    guard = torch.full((replicas * m * n + 4096,), 777.0, device=dev)
    out = guard[:replicas * m * n].view(replicas, m, n)
    capi.left_spmm_half_tiles(m, k, n, replicas, T(ro, dev), T(ci, dev), v, b, torch.float16, out, ws)
    assert np.array_equal(out.cpu().numpy(), want.astype(np.float32))
    assert bool((guard[replicas * m * n:] == 777.0).all())


@pytest.mark.parametrize("tile", HALF_TYPES)
@pytest.mark.parametrize("values_kind,grad_kind", [("half", "half"), ("float", "half"), ("half", "float"),
                                                   ("float", "float")])
@pytest.mark.parametrize("out_f,in_f,seq,batch,sparsity", [
    (256, 128, 128, 2, 0.8),
    (128, 64, 64, 1, 0.5),       # one tile, one step, in every product
    (192, 320, 192, 3, 0.8),     # ragged tiles in every product (192 = 128 + 64, 320 = 2 x 128 + 64)
    (384, 256, 448, 2, 0.95),    # very sparse
])
def test_half_linear_three_products_vs_oracle(spmm_mfma, mfma_tile, dev, tile, values_kind, grad_kind, out_f,
                                              in_f, seq, batch, sparsity):
    """A sparse layer on half-stored activations (csrc/sparse_linear_half.hip): forward,
    weight gradient and input gradient on the matrix cores, every operand read in the
    layout the caller has it (x [B, S, in], dy [B, out, S], one image of the weight) --
    against float64 on the operands as stored (float32 operands enter as half planes, not
    rounded: the float32 bound; dx is returned in x's type: one unit in its last place)."""
    from torch_sputnik_amd import ops
    _, vals, ri, ro, ci = make_csr(out_f, in_f, sparsity, seed=out_f + in_f + seq, round_to=1,
                                   empty_rows=(out_f // 3,))
    rng = np.random.default_rng(seq + 1)
    v, v32 = _operand(vals * 10.0 ** rng.uniform(-2, 0, size=len(vals)) * rng.choice([-1, 1], size=len(vals)),
                      values_kind, tile, dev)
    x, x32 = rounded(rng.uniform(-1, 1, size=(batch, seq, in_f)), tile, dev)
    g, g32 = _operand(rng.uniform(-1, 1, size=(batch, out_f, seq)) * 10.0 ** rng.uniform(-2, 0, size=(batch, out_f, 1)),
                      grad_kind, tile, dev)
    w = np.zeros((out_f, in_f), np.float64)
    rows = np.repeat(np.arange(out_f), np.diff(ro))
    w[rows, ci] = v32
    x64, g64 = x32.astype(np.float64), g32.astype(np.float64)
    want_y = np.einsum("oi,bsi->bos", w, x64)
    want_dw = np.einsum("bos,bsi->oi", g64, x64)[rows, ci]
    want_dx = np.einsum("bos,oi->bsi", g64, w)
    rod, cid = T(ro, dev), T(ci, dev)
    assert ops.half_linear_supported(out_f, in_f, seq, batch, len(ci), v.dtype, tile)   # (the knob)
    image = ops.half_linear_image(out_f, in_f, v, rod, cid, tile)
    y = ops.half_linear_forward(out_f, image, v.dtype, x)
    assert y.dtype == torch.float32 and tuple(y.shape) == (batch, out_f, seq)
    assert rel_err(y.cpu().numpy(), want_y) < TOL
    planes = grad_kind == "float"
    grad = ops.half_planes(g, tile) if planes else g
    for plan in (None, ops.half_linear_plan(out_f, in_f, rod, cid)):
        dw = ops.half_linear_weight_gradient(out_f, rod, cid, grad, planes, x, plan)
        assert rel_err(dw.cpu().numpy()[None, :], want_dw[None, :], ro) < TOL
    dx = ops.half_linear_input_gradient(out_f, in_f, grad, planes, image, v.dtype, x, batch, seq)
    if tile == torch.bfloat16 and values_kind == grad_kind == "float":
        assert dx is None       # nine plane pairs: not built, the caller takes the typed operators
        return
    assert dx.dtype == tile and tuple(dx.shape) == (batch, seq, in_f)
    assert half_err(dx.float().cpu().numpy(), want_dx, tile) < TOL


def test_half_linear_exact_integers(spmm_mfma, mfma_tile, dev):
    """Small integers (every product and sum exact): the four operand layouts' fragment
    maps must give the exact matrices."""
    from torch_sputnik_amd import ops
    out_f, in_f, seq, batch = 320, 128, 384, 2     # (256-row tiles: one whole, one ragged)
    _, _, ri, ro, ci = make_csr(out_f, in_f, 0.6, seed=23, round_to=1)
    rng = np.random.default_rng(24)
    v32 = rng.integers(-4, 5, size=len(ci)).astype(np.float32)
    x32 = rng.integers(-4, 5, size=(batch, seq, in_f)).astype(np.float32)
    g32 = rng.integers(-4, 5, size=(batch, out_f, seq)).astype(np.float32)
    rows = np.repeat(np.arange(out_f), np.diff(ro))
    w = np.zeros((out_f, in_f), np.float32)
    w[rows, ci] = v32
    v, x, g = T(v32, dev).half(), T(x32, dev).half(), T(g32, dev).half()
    rod, cid = T(ro, dev), T(ci, dev)
    image = ops.half_linear_image(out_f, in_f, v, rod, cid, torch.float16)
    assert np.array_equal(ops.half_linear_forward(out_f, image, v.dtype, x).cpu().numpy(),
                          np.einsum("oi,bsi->bos", w, x32))
    assert np.array_equal(ops.half_linear_weight_gradient(out_f, rod, cid, g, False, x).cpu().numpy(),
                          np.einsum("bos,bsi->oi", g32, x32)[rows, ci])
    dx = ops.half_linear_input_gradient(out_f, in_f, g, False, image, v.dtype, x, batch, seq)
    assert np.array_equal(dx.float().cpu().numpy(), np.einsum("bos,oi->bsi", g32, w))


@pytest.mark.parametrize("dtype", HALF_TYPES)
def test_left_spmm_typed_takes_the_tiles_at_layer_density(ts, capi, dev, dtype):
    """The op level: left_spmm with a half dense operand at a layer's density and size goes
    to the matrix cores by itself (no knob) -- float32 and half weights -- and agrees
    with the oracle; the float32 op never does."""
    m, k, n, r = 1024, 1024, 512, 8      # 256 tiles: a tile per CU
    _, vals, ri, ro, ci = make_csr(m, k, 0.8, seed=91)
    topo = (T(ri, dev), T(ro, dev), T(ci, dev))
    rng = np.random.default_rng(92)
    b, b32 = rounded(rng.uniform(-1, 1, size=(r, k, n)), dtype, dev)
    assert capi.left_spmm_half_tiles_workspace_bytes(m, k, n, len(ci), r, T(vals, dev), b, dtype) > 0
    for v, v32 in ((T(vals, dev), vals), rounded(vals, dtype, dev)):
        out = ts.left_spmm(m, k, v, *topo, b)
        assert out.dtype == torch.float32 and tuple(out.shape) == (r, m, n)
        assert rel_err(out.cpu().numpy(), O.left_spmm(m, k, v32, ri, ro, ci, b32)) < TOL


@pytest.mark.parametrize("dtype", HALF_TYPES)
def test_spmm_half_ops(ts, dev, dtype):
    """The torch ops hand half operands to the kernels as they are; the product is
    float32 (src/spmm_cuda.cu:42); bias / ReLU epilogue included."""
    import torch_sputnik_amd as tsa
    m, k, n, r = 512, 512, 256, 3
    _, vals, ri, ro, ci = make_csr(m, k, 0.85, seed=71)
    topo = (T(ri, dev), T(ro, dev), T(ci, dev))
    rng = np.random.default_rng(72)
    v, v32 = rounded(vals, dtype, dev)
    b, b32 = rounded(rng.uniform(-1, 1, size=(r, k, n)), dtype, dev)
    out = ts.left_spmm(m, k, v, *topo, b)
    assert out.dtype == torch.float32 and tuple(out.shape) == (r, m, n)
    want = O.left_spmm(m, k, v32, ri, ro, ci, b32)
    assert rel_err(out.cpu().numpy(), want) < TOL
    out2 = ts.spmm(m, k, v, *topo, b[0])
    assert tuple(out2.shape) == (m, n)
    assert rel_err(out2.cpu().numpy(), want[0]) < TOL
    bias = T(rng.uniform(-1, 1, size=(m,)).astype(np.float32), dev)
    fused = tsa.ops.spmm_bias_relu(m, k, v, *topo, bias, b[0])
    want_f = np.maximum(want[0].astype(np.float64) + bias.cpu().numpy()[:, None], 0.0)
    assert np.max(np.abs(fused.cpu().numpy() - want_f)) < 1e-3 * max(1.0, np.abs(want_f).max())


@pytest.mark.parametrize("tv,tb", SPMM_TYPES)
def test_spmm_half_large_k_is_widened_inside_the_call(capi, dev, tv, tb):
    """k = 2048 (four panels would lose by a factor of two): with the typed workspace
    the call widens the operands itself and runs the chunked float kernels; without
    it the row-gather kernel reads the half rows.  Same answer either way."""
    m, k, n, replicas = 2048, 2048, 512, 2
    _, vals, ri, ro, ci = make_csr(m, k, 0.8, seed=5)
    rng = np.random.default_rng(6)
    v, v32 = rounded(vals, tv, dev)
    b, b32 = rounded(rng.uniform(-1, 1, size=(replicas, k, n)), tb, dev)
    topo = (T(ri, dev), T(ro, dev), T(ci, dev))
    need = capi.spmm_typed_workspace_bytes(m, k, n, len(vals), replicas, v, 0, b)
    assert need > 0
    ws = torch.empty(need + 256, dtype=torch.uint8, device=dev)
    ws = ws[(256 - ws.data_ptr() % 256) % 256:]
    out = torch.full((replicas, m, n), float("nan"), device=dev)
    capi.spmm_typed(m, k, n, replicas, topo[0], v, 0, topo[1], topo[2], b, out, ws)
    want = torch.matmul(T(O.csr_to_dense(m, k, v32, ro, ci), dev).double(), T(b32, dev).double())
    from tests.helpers import rel_err_torch
    assert rel_err_torch(out, want) < TOL
    out2 = torch.full_like(out, float("nan"))
    capi.spmm_typed(m, k, n, replicas, topo[0], v, 0, topo[1], topo[2], b, out2, None)
    assert rel_err_torch(out2, want) < TOL


# ----------------------------------------------------------------------------
# Fuzz: seeded random shapes around the tile boundaries, every row order, empty
# rows, 1..4 replicas, both half types and the mixed pairings, each typed entry
# point against the oracle on the rounded inputs.
# ----------------------------------------------------------------------------
ORDERS = ("descending", "ascending", "random", "identity")


def _dims(rng, choices):
    return max(1, int(rng.choice(choices)) + int(rng.integers(-3, 4)) * int(rng.random() < 0.5))


def test_fuzz_typed_operators(capi, dev, sddmm_kernel):
    rng = np.random.default_rng(20261004)
    dummy_i = torch.zeros(1, dtype=torch.int32, device=dev)
    for it in range(36):
        m = _dims(rng, [16, 64, 128, 256, 300])
        n = _dims(rng, [32, 64, 128, 256, 520])
        sparsity = float(rng.choice([0.0, 0.5, 0.8, 0.9, 0.97]))
        empty = tuple(int(x) for x in rng.integers(0, m, size=int(rng.integers(0, 3))))
        order = ORDERS[int(rng.integers(0, len(ORDERS)))]
        replicas = int(rng.integers(1, 5))
        dtype = HALF_TYPES[it % 2]
        _, vals, ri, ro, ci = make_csr(m, n, sparsity, seed=1000 + it, round_to=1, empty_rows=empty,
                                       order=order)
        nnz = len(ci)
        topo = (T(ri, dev), T(ro, dev), T(ci, dev) if nnz else dummy_i)
        what = (it, m, n, sparsity, order, replicas, str(dtype))

        # SDDMM: [m, k] x [n, k] sampled at the mask
        k = int(rng.choice([7, 32, 64, 64, 128, 192, 256, 512]))
        lhs, lhs32 = rounded(rng.uniform(-1, 1, (replicas, m, k)), dtype, dev)
        rhs, rhs32 = rounded(rng.uniform(-1, 1, (replicas, n, k)), dtype, dev)
        want = c_oracle.sddmm(m, n, ro, ci, lhs32, rhs32)
        out = torch.full((replicas, max(nnz, 1)), float("nan"), device=dev)[:, :nnz].contiguous()
        ws = torch.empty(capi.sddmm_workspace_bytes(m, k, n, nnz) + 16, dtype=torch.uint8, device=dev)
        capi.sddmm_typed(m, k, n, replicas, *topo, lhs, rhs, out, ws)
        if nnz:
            got = out.cpu().numpy()
            assert not np.isnan(got).any(), what
            assert rel_err(got, want, ro) < TOL, what + (k,)

        # softmax pair on half scores
        if nnz:
            sc, sc32 = rounded(rng.uniform(-6, 6, (replicas, nnz)), dtype, dev)
            y = torch.full_like(sc, float("nan"))
            capi.sparse_softmax_typed(m, replicas, sc, *topo, 0.5, y)
            want_y = c_oracle.sparse_softmax((sc32.astype(np.float64) * 0.5).astype(np.float32), ro, ci)
            assert half_err(y.float().cpu().numpy(), want_y, dtype, ro) < 2 * TOL, what
            g, g32 = rounded(rng.uniform(-1, 1, (replicas, nnz)), dtype, dev)
            dx = torch.full_like(sc, float("nan"))
            capi.sparse_softmax_backward_typed(m, replicas, y, g, topo[1], 0.5, dx)
            want_dx = O.sparse_softmax_backward(y.float().cpu().numpy(), g32, ro, 0.5)
            assert half_err(dx.float().cpu().numpy(), want_dx, dtype, ro) < TOL, what

        # SpMM: the mask as A [m, n], B [n, cols]
        cols = int(rng.choice([1, 7, 18, 64, 128, 192, 256]))
        shared = bool(rng.random() < 0.5)
        tv, tb = [(dtype, dtype), (dtype, torch.float32), (torch.float32, dtype)][it % 3]
        v, v32 = rounded(rng.uniform(-1, 1, (nnz,) if shared else (replicas, nnz)), tv, dev)
        b, b32 = rounded(rng.uniform(-1, 1, (replicas, n, cols)), tb, dev)
        want_c = np.stack([O.spmm(m, n, v32 if shared else v32[r], ri, ro, ci, b32[r])
                           for r in range(replicas)])
        c = torch.full((replicas, m, cols), float("nan"), device=dev)
        capi.spmm_typed(m, n, cols, replicas, topo[0], v if nnz else torch.zeros(1, device=dev, dtype=tv),
                        0 if shared else nnz, topo[1], topo[2], b, c)
        got_c = c.cpu().numpy()
        assert not np.isnan(got_c).any(), what
        assert rel_err(got_c, want_c) < TOL, what + (cols, shared, str(tv), str(tb))


# ----------------------------------------------------------------------------
# csr_transpose of half values: topology bit-exact, values widened by the gather
# ----------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", HALF_TYPES)
@pytest.mark.parametrize("m,n,sparsity,replicas", [(72, 64, 0.8, 1), (257, 1000, 0.9, 1),
                                                   (2048, 2048, 0.8, 1), (1500, 300, 0.5, 3),
                                                   (64, 20000, 0.99, 1), (3000, 64, 0.3, 2)])
def test_transpose_half_values_bit_exact(capi, dev, dtype, m, n, sparsity, replicas):
    _, vals, _, ro, ci = make_csr(m, n, sparsity, seed=m + 3 * n, round_to=1)
    rng = np.random.default_rng(n)
    v, v32 = rounded(vals if replicas == 1 else rng.uniform(size=(replicas, len(vals))), dtype, dev)
    want = O.csr_transpose(m, n, v32, ro, ci)
    nnz = len(ci)
    out_v = torch.full(v32.shape, float("nan"), device=dev)
    out_ro = torch.full((n + 1,), -1, dtype=torch.int32, device=dev)
    out_ci = torch.full((nnz,), -1, dtype=torch.int32, device=dev)
    perm = torch.full((nnz,), -1, dtype=torch.int32, device=dev)
    ws = torch.empty(capi.csr_transpose_workspace_bytes(m, n, nnz) + 16, dtype=torch.uint8, device=dev)
    for checked in (False, True):
        capi.csr_transpose_typed(m, n, replicas, v, T(ro, dev), T(ci, dev), out_v, out_ro, out_ci,
                                 perm, ws, checked=checked)
        assert np.array_equal(out_v.cpu().numpy(), np.asarray(want[0], np.float32))   # exact: a move
        assert np.array_equal(out_ro.cpu().numpy(), want[1])
        assert np.array_equal(out_ci.cpu().numpy(), want[2])
        assert np.array_equal(v32.reshape(-1, nnz)[:, perm.cpu().numpy()].reshape(v32.shape),
                              out_v.cpu().numpy())


@pytest.mark.parametrize("dtype", HALF_TYPES)
def test_transpose_half_op(ts, dev, dtype):
    import torch_sputnik_amd as tsa
    m, n = 300, 500
    _, vals, _, ro, ci = make_csr(m, n, 0.8, seed=17, round_to=1)
    v, v32 = rounded(vals, dtype, dev)
    want = O.csr_transpose(m, n, v32, ro, ci)
    vt, rot, cit = ts.csr_transpose(m, n, v, T(ro, dev), T(ci, dev))
    assert vt.dtype == torch.float32
    assert np.array_equal(vt.cpu().numpy(), np.asarray(want[0], np.float32))
    assert np.array_equal(rot.cpu().numpy(), want[1]) and np.array_equal(cit.cpu().numpy(), want[2])
    vt2, _, _, perm = tsa.ops.csr_transpose_with_permutation(m, n, v, T(ro, dev), T(ci, dev))
    assert torch.equal(vt2, vt) and torch.equal(v.float()[perm.long()], vt)


# ----------------------------------------------------------------------------
# The autograd Functions on half tensors (user code that calls Spmm / Sddmm /
# SparseSoftmax with float16 / bfloat16 operands): forward float32 from the rounded
# operands at 1e-4, gradients in the operands' types at their resolution.
# ----------------------------------------------------------------------------
@pytest.fixture(scope="module")
def tsa():
    import torch_sputnik_amd
    return torch_sputnik_amd


def _grad_tol(dtype):   # a gradient returned in `dtype`: its rounding on top of the chain's 5e-4
    return 2e-3 if dtype == torch.float16 else 1.2e-2


@pytest.mark.parametrize("dtype", HALF_TYPES)
@pytest.mark.parametrize("m,k,n,replicas,sparsity", [(256, 192, 128, 1, 0.8), (512, 512, 512, 1, 0.9),
                                                     (256, 320, 256, 3, 0.85)])
def test_spmm_function_half_vs_dense_autograd(tsa, dev, dtype, m, k, n, replicas, sparsity):
    from tests.helpers import rel_err_torch
    _, vals, ri, ro, ci = make_csr(m, k, sparsity, seed=m + k, order="ascending")
    rng = np.random.default_rng(n)
    shape_v = (len(vals),) if replicas == 1 else (replicas, len(vals))
    shape_b = (k, n) if replicas == 1 else (replicas, k, n)
    v = T(rng.uniform(-1, 1, shape_v).astype(np.float32), dev).to(dtype).requires_grad_(True)
    b = T(rng.uniform(-1, 1, shape_b).astype(np.float32), dev).to(dtype).requires_grad_(True)
    go = T(rng.uniform(-1, 1, shape_b[:-2] + (m, n)).astype(np.float32), dev)
    d_ri, d_ro, d_ci = T(ri, dev), T(ro, dev), T(ci, dev)
    out = tsa.Spmm.apply(m, k, v, d_ri, d_ro, d_ci, b)
    assert out.dtype == torch.float32
    out.backward(go)
    vd = v.detach().double().requires_grad_(True)
    bd = b.detach().double().requires_grad_(True)
    rows = torch.repeat_interleave(torch.arange(m, device=dev), (d_ro[1:] - d_ro[:-1]).long())
    a = torch.zeros(shape_v[:-1] + (m, k), dtype=torch.float64, device=dev)
    a[..., rows, d_ci.long()] = vd
    want = torch.matmul(a, bd)
    want.backward(go.double())
    assert rel_err_torch(out.detach(), want.detach()) < TOL
    assert b.grad.dtype == dtype and v.grad.dtype == dtype
    assert rel_err_torch(b.grad.float(), bd.grad) < _grad_tol(dtype)
    assert rel_err(v.grad.float().cpu().numpy(), vd.grad.cpu().numpy(), ro) < _grad_tol(dtype)


@pytest.mark.parametrize("dtype", HALF_TYPES)
@pytest.mark.parametrize("m,k,n,replicas,sparsity", [(256, 64, 256, 1, 0.9), (512, 512, 384, 1, 0.8),
                                                     (256, 128, 256, 4, 0.9)])
def test_sddmm_function_half_vs_dense_autograd(tsa, dev, dtype, m, k, n, replicas, sparsity):
    from tests.helpers import rel_err_torch
    _, _, ri, ro, ci = make_csr(m, n, sparsity, seed=m + n, order="ascending")
    rng = np.random.default_rng(k)
    lead = () if replicas == 1 else (replicas,)
    lhs = T(rng.uniform(-1, 1, lead + (m, k)).astype(np.float32), dev).to(dtype).requires_grad_(True)
    rhs = T(rng.uniform(-1, 1, lead + (n, k)).astype(np.float32), dev).to(dtype).requires_grad_(True)
    go = T(rng.uniform(-1, 1, lead + (len(ci),)).astype(np.float32), dev)
    d_ri, d_ro, d_ci = T(ri, dev), T(ro, dev), T(ci, dev)
    out = tsa.Sddmm.apply(m, n, d_ri, d_ro, d_ci, lhs, rhs)
    assert out.dtype == torch.float32
    out.backward(go)
    ld = lhs.detach().double().requires_grad_(True)
    rd = rhs.detach().double().requires_grad_(True)
    rows = torch.repeat_interleave(torch.arange(m, device=dev), (d_ro[1:] - d_ro[:-1]).long())
    want = torch.matmul(ld, rd.transpose(-1, -2))[..., rows, d_ci.long()]
    want.backward(go.double())
    assert rel_err(out.detach().cpu().numpy(), want.detach().cpu().numpy(), ro) < TOL
    assert lhs.grad.dtype == dtype and rhs.grad.dtype == dtype
    assert rel_err_torch(lhs.grad.float(), ld.grad) < _grad_tol(dtype)
    assert rel_err_torch(rhs.grad.float(), rd.grad) < _grad_tol(dtype)


@pytest.mark.parametrize("dtype", HALF_TYPES)
def test_softmax_function_half_autograd(tsa, dev, dtype):
    m, n = 256, 512
    _, vals, ri, ro, ci = make_csr(m, n, 0.85, seed=3, round_to=1)
    rng = np.random.default_rng(4)
    x = T(rng.uniform(-4, 4, (3, len(vals))).astype(np.float32), dev).to(dtype).requires_grad_(True)
    go = T(rng.uniform(-1, 1, (3, len(vals))).astype(np.float32), dev).to(dtype)
    topo = (T(ri, dev), T(ro, dev), T(ci, dev))
    y = tsa.SparseSoftmax.apply(x, *topo, 0.5) if hasattr(tsa, "SparseSoftmax") else None
    assert y is not None and y.dtype == dtype
    y.backward(go)
    assert x.grad.dtype == dtype
    want_y = c_oracle.sparse_softmax((x.detach().float().cpu().numpy().astype(np.float64) * 0.5)
                                     .astype(np.float32), ro, ci)
    assert half_err(y.detach().float().cpu().numpy(), want_y, dtype, ro) < 2 * TOL
    want_dx = O.sparse_softmax_backward(y.detach().float().cpu().numpy(),
                                        go.float().cpu().numpy(), ro, 0.5)
    assert half_err(x.grad.float().cpu().numpy(), want_dx, dtype, ro) < TOL
