"""GPU: the HIP kernels, called through the C ABI (torch_sputnik_amd.capi) and
through torch.ops.torch_sputnik, against the oracle and the golden fixtures.

Tolerance: BASELINE.json's north star asks for 1e-4 on fp32 outputs.  The
oracle is float64; helpers.rel_err holds every element to 1e-4 * (|want| +
mean|want| of its own output row) and every element above 1 % of its row's
maximum to a pure relative 1e-4.
"""
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
    assert "gfx950" in capi.version()
    return capi


@pytest.fixture(scope="module")
def ts():
    import torch_sputnik
    return torch_sputnik


def T(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def topo_t(ri, ro, ci, dev):
    return T(ri, dev), T(ro, dev), T(ci, dev)


# ----------------------------------------------------------------------------
# SpMM
# ----------------------------------------------------------------------------
SPMM_SHAPES = [
    # m, k, n, sparsity, order, empty_rows
    (64, 64, 64, 0.5, "descending", ()),          # BASELINE config 1
    (72, 64, 72, 0.9, "descending", ()),          # tests/test_spmm.py
    (72, 64, 72, 0.9, "ascending", (0, 71)),
    (33, 47, 7, 0.6, "random", (5,)),             # n odd -> scalar path
    (33, 47, 18, 0.6, "identity", ()),            # n % 2 == 0 -> float2 path
    (257, 300, 260, 0.8, "descending", (256,)),   # two n-tiles, ragged
    (512, 1024, 512, 0.9, "ascending", ()),
    (1024, 777, 1024, 0.95, "descending", ()),
    (100, 4096, 256, 0.98, "random", (1, 2, 3)),
    (2048, 2048, 64, 0.9, "descending", ()),      # attention-like n: 64-column tiled kernel
    (512, 300, 128, 0.8, "ascending", (17,)),     # two 64-column tiles, partial last K chunk
    (100, 1000, 192, 0.9, "random", (0,)),        # padded row slots, three tiles
    (128, 256, 64, 0.3, "descending", ()),        # > 16 entries per row and chunk: extra windows
    (64, 128, 64, 0.0, "identity", ()),           # dense "sparse" matrix
]


@pytest.mark.parametrize("m,k,n,sparsity,order,empty", SPMM_SHAPES)
def test_spmm_capi_vs_oracle(capi, dev, spmm_kernel, m, k, n, sparsity, order, empty):
    _, vals, ri, ro, ci = make_csr(m, k, sparsity, seed=m * 7 + n, empty_rows=empty, order=order)
    b = np.random.default_rng(n).uniform(-1, 1, size=(k, n)).astype(np.float32)
    want = c_oracle.spmm(m, k, vals, ro, ci, b)
    out = torch.full((m, n), float("nan"), device=dev)
    ws = torch.empty(capi.spmm_workspace_bytes(m, k, n, len(ci)) + 16, dtype=torch.uint8, device=dev)
    capi.spmm_batched(m, k, n, 1, T(ri, dev), T(vals, dev), 0, T(ro, dev), T(ci, dev), T(b, dev),
                      out, ws)
    got = out.cpu().numpy()
    assert not np.isnan(got).any(), "some output elements were never written"
    assert rel_err(got, want) < TOL
    for r in empty:
        assert np.all(got[r] == 0)


STRESS = [
    # m, k, n, sparsity: corners of the tiled kernels' bookkeeping
    (256, 64, 256, 0.05),     # one chunk, nearly dense rows (up to 64 entries per row and chunk)
    (300, 200, 256, 0.1),     # dense rows, ragged m and k, entry windows clamped at the array end
    (64, 4096, 256, 0.995),   # most (row, chunk) segments empty
    (1000, 130, 512, 0.5),    # k % 64 = 2: last chunk almost empty
    (257, 65, 768, 0.7),      # three column tiles (n_tiles % 8 != 0), 1 padded chunk row
    (2048, 2048, 256, 0.8),   # medium-tile configuration
    (512, 96, 64, 0.2),       # 64-column kernel, > 32 entries per row and chunk
    (90, 1000, 320, 0.9),     # 64-column kernel, five tiles, padded row slots
    (300, 100, 1024, 0.1),    # 512-column kernel: dense rows (up to 32 entries per row and 32-row chunk)
    (128, 32, 512, 0.0),      # 512-column kernel: one chunk, fully dense
    (1000, 33, 1536, 0.5),    # 512-column kernel: three tiles, k % 32 = 1, row slots padded to 1024
    (384, 4096, 512, 0.97),   # 512-column kernel: 128 chunks, most segments empty, slots padded to 512
    (256, 40, 512, 0.2),      # 512-column kernel: one full chunk and a quarter, long segments
    (200, 41, 1024, 0.4),     # 512-column kernel: two tiles, last chunk 9 rows
    (512, 1000, 512, 0.05),   # 512-column kernel: segments of ~30 entries (two windows), k % 32 = 8
]


@pytest.mark.parametrize("m,k,n,sparsity", STRESS)
@pytest.mark.parametrize("order", ["descending", "random"])
def test_spmm_tiled_stress(capi, dev, spmm_kernel, m, k, n, sparsity, order):
    _, vals, ri, ro, ci = make_csr(m, k, sparsity, seed=3 * m + k + n, round_to=1, order=order,
                                   empty_rows=(m - 1,) if sparsity > 0.5 else ())
    b = np.random.default_rng(m + n).uniform(-1, 1, size=(k, n)).astype(np.float32)
    want = c_oracle.spmm(m, k, vals, ro, ci, b)
    out = torch.full((m, n), float("nan"), device=dev)
    ws = torch.empty(capi.spmm_workspace_bytes(m, k, n, len(ci)) + 16, dtype=torch.uint8, device=dev)
    capi.spmm_batched(m, k, n, 1, T(ri, dev), T(vals, dev), 0, T(ro, dev), T(ci, dev), T(b, dev),
                      out, ws)
    got = out.cpu().numpy()
    assert not np.isnan(got).any()
    assert rel_err(got, want) < TOL


TAIL_SHAPES = [
    # m, k, n, sparsity, replicas: dense widths that are multiples of 4 but of no
    # tile width -- the last column tile of the LDS-tiled kernels is partial
    (72, 64, 72, 0.5, 1),        # tests/test_spmm.py:13: 64-column kernel, tiles of 64 + 8
    (256, 300, 200, 0.8, 2),     # one 256-column tile, 200 used; 64-column kernel: 3 x 64 + 8
    (512, 512, 1000, 0.9, 1),    # two 512-column tiles (512 + 488); four 256-column tiles
    (300, 1024, 4000, 0.95, 1),  # eight 512-column tiles, the last with 416 columns
    (640, 96, 260, 0.3, 3),      # 256 + 4: a tile whose lanes are nearly all clamped
    (128, 2048, 68, 0.9, 2),     # 64 + 4
    (1000, 777, 516, 0.7, 1),    # 512 + 4
]


@pytest.mark.parametrize("m,k,n,sparsity,replicas", TAIL_SHAPES)
@pytest.mark.parametrize("order", ["ascending", "random"])
def test_spmm_partial_column_tiles(capi, dev, spmm_kernel, m, k, n, sparsity, replicas, order):
    """Any n that is a multiple of 4 stays on the LDS-tiled kernels (the reference
    accepts any n, src/spmm_cuda.cu:32); nothing may be written past a row's end
    or past the end of the output (the buffer carries a guard zone), and the last
    B row's clamped copies must stay inside B (B is the last allocation made)."""
    _, vals, ri, ro, ci = make_csr(m, k, sparsity, seed=m + n, round_to=1, order=order,
                                   empty_rows=(m // 3,))
    rng = np.random.default_rng(n)
    v = vals if replicas == 1 else rng.uniform(-1, 1, (replicas, len(vals))).astype(np.float32)
    b = rng.uniform(-1, 1, size=(replicas, k, n)).astype(np.float32)
    want = c_oracle.spmm(m, k, v, ro, ci, b if replicas > 1 else b[0])
    guard = 64
    flat = torch.full((replicas * m * n + guard,), float("nan"), device=dev)
    out = flat[:replicas * m * n].view(replicas, m, n)
    ws = torch.empty(capi.spmm_workspace_bytes(m, k, n, len(ci)) + 16, dtype=torch.uint8, device=dev)
    capi.spmm_batched(m, k, n, replicas, T(ri, dev), T(v, dev), 0 if replicas == 1 else len(vals),
                      T(ro, dev), T(ci, dev), T(b, dev), out, ws)
    got = out.cpu().numpy()
    assert not np.isnan(got).any(), "some output elements were never written"
    assert torch.isnan(flat[replicas * m * n:]).all(), "wrote past the end of the output"
    assert rel_err(got.reshape(want.shape), want) < TOL


def test_spmm_full_size_linearity(capi, dev):
    """BASELINE.json's headline size (4096^3, density 0.1), checked through
    size-independent properties: linearity in the values, agreement of the planned
    and unplanned entry points, and exact rows against the oracle on a sample."""
    from torch_sputnik_amd.synthetic import random_csr, uniform
    m = k = n = 4096
    ri, ro, ci, nnz = random_csr(m, k, 0.1, dev, seed=5234)
    v1, v2 = uniform((nnz,), dev, 1), uniform((nnz,), dev, 2)
    b = uniform((k, n), dev, 3) - 0.5
    ws = torch.empty(capi.spmm_workspace_bytes(m, k, n, nnz) + 16, dtype=torch.uint8, device=dev)

    def run(v):
        out = torch.empty(m, n, device=dev)
        capi.spmm_batched(m, k, n, 1, ri, v, 0, ro, ci, b, out, ws)
        return out

    c1, c2, c12 = run(v1), run(v2), run(v1 + 2 * v2)
    assert torch.allclose(c12, c1 + 2 * c2, rtol=1e-4, atol=1e-3)
    planned = torch.empty(m, n, device=dev)
    capi.spmm_plan(m, k, n, ri, ro, ci, ws)
    capi.spmm_batched_planned(m, k, n, 1, ri, v1, 0, ro, ci, b, planned, ws)
    assert torch.equal(planned, c1)
    # 16 sampled rows, float64 on the host
    rows = torch.randint(0, m, (16,), generator=torch.Generator().manual_seed(0)).tolist()
    ro_h, ci_h, v_h, b_h = ro.cpu().numpy(), ci.cpu().numpy(), v1.cpu().numpy(), b.cpu().double().numpy()
    for r in rows:
        p0, p1 = ro_h[r], ro_h[r + 1]
        want = v_h[p0:p1].astype(np.float64) @ b_h[ci_h[p0:p1]]
        assert rel_err(c1[r].cpu().numpy(), want) < TOL


def dense_fp64_product(m, k, ro, ci, vals, b):
    """The reference tests' dense definition (tests/test_spmm.py:9-10) in
    float64 on the device: scatter the CSR matrix, multiply."""
    rows = torch.repeat_interleave(torch.arange(m, device=ro.device), (ro[1:] - ro[:-1]).long())
    a = torch.zeros(m, k, device=ro.device, dtype=torch.float64)
    a[rows, ci.long()] = vals.double()
    return a @ b.double()


@pytest.mark.parametrize("density", [0.5, 0.25, 0.2, 0.15, 0.1, 0.05])
def test_spmm_full_size_whole_matrix(capi, dev, density):
    """BASELINE config 2, every density of the sweep: every element of the 4096^3
    product against the dense float64 product on the GPU (the CPU oracle would
    take minutes): the linearity test above cannot see a row that is wrong in the
    same way in every call."""
    from torch_sputnik_amd.synthetic import random_csr, uniform
    m = k = n = 4096
    ri, ro, ci, nnz = random_csr(m, k, density, dev, seed=int(density * 1000) + 11)
    vals = uniform((nnz,), dev, 1) - 0.5
    b = uniform((k, n), dev, 2) - 0.5
    want = dense_fp64_product(m, k, ro, ci, vals, b)
    ws = torch.empty(capi.spmm_workspace_bytes(m, k, n, nnz) + 16, dtype=torch.uint8, device=dev)
    out = torch.full((m, n), float("nan"), device=dev)
    capi.spmm_batched(m, k, n, 1, ri, vals, 0, ro, ci, b, out, ws)
    assert not torch.isnan(out).any()
    assert rel_err_torch(out, want) < TOL


def test_spmm_full_size_after_other_products(capi, dev):
    """The 4096^3 product after three products of OTHER topologies have gone
    through the same workspace, called three times back to back, every element
    checked.  A layout of the tiled kernel's main loop that was a few scalar
    instructions shorter per visit passed everything else and faulted in this
    sequence in most runs: a change of that loop has to
    pass this one, repeatedly (tools/repeat_primer.sh)."""
    from torch_sputnik_amd.synthetic import random_csr, uniform
    m = k = n = 4096
    density, seed = 0.1, 5234
    b = uniform((k, n), dev, 2) - 0.5
    ri, ro, ci, nnz = random_csr(m, k, density, dev, seed=seed)
    vals = uniform((nnz,), dev, 1) - 0.5
    ws = torch.empty(capi.spmm_workspace_bytes(m, k, n, nnz) + 16, dtype=torch.uint8, device=dev)
    for j in range(3):
        ri2, ro2, ci2, nnz2 = random_csr(m, k, density, dev, seed=seed + 1 + j)
        assert nnz2 == nnz
        scratch_out = torch.empty(m, n, device=dev)
        capi.spmm_batched(m, k, n, 1, ri2, vals, 0, ro2, ci2, b, scratch_out, ws)
        torch.cuda.synchronize()
    outs = []
    for _ in range(3):
        out = torch.full((m, n), float("nan"), device=dev)
        capi.spmm_batched(m, k, n, 1, ri, vals, 0, ro, ci, b, out, ws)
        outs.append(out)
    want = dense_fp64_product(m, k, ro, ci, vals, b)
    for out in outs:
        assert rel_err_torch(out, want) < TOL


def test_spmm_full_size_soak_on_a_warm_gpu(capi, dev):
    """tools/repeat_primer.sh as a test: the failure of DESIGN.md section 3.1 (a
    register of an in-flight load reused by the compiler) never showed on a cold
    GPU and in 40-75 % of the runs on a warm one.  One second of dense matmul
    first, then 32 products at the headline size that alternate between four
    topologies through ONE workspace, every one checked: the first of each
    topology against the dense float64 product, the repeats bit for bit against
    the first (the kernels are deterministic)."""
    from torch_sputnik_amd.synthetic import random_csr, uniform
    m = k = n = 4096
    x = torch.randn(4096, 4096, device=dev)
    t0 = torch.cuda.Event(enable_timing=True)
    t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    elapsed = 0.0
    while elapsed < 1000.0:    # ms of GPU time
        for _ in range(20):
            x = torch.tanh(x @ x) * 0.01
        t1.record()
        torch.cuda.synchronize()
        elapsed = t0.elapsed_time(t1)
    b = uniform((k, n), dev, 2) - 0.5
    topologies = []
    for j, density in enumerate((0.1, 0.1, 0.05, 0.25)):
        ri, ro, ci, nnz = random_csr(m, k, density, dev, seed=900 + j)
        topologies.append((ri, ro, ci, nnz, uniform((nnz,), dev, j) - 0.5))
    ws_bytes = max(capi.spmm_workspace_bytes(m, k, n, t[3]) for t in topologies) + 16
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    first = {}
    for rep in range(32):
        j = rep % len(topologies)
        ri, ro, ci, nnz, vals = topologies[j]
        out = torch.full((m, n), float("nan"), device=dev)
        capi.spmm_batched(m, k, n, 1, ri, vals, 0, ro, ci, b, out, ws)
        if j not in first:
            assert rel_err_torch(out, dense_fp64_product(m, k, ro, ci, vals, b)) < TOL
            first[j] = out
        else:
            assert torch.equal(out, first[j]), f"repeat {rep} of topology {j} differs"


@pytest.mark.parametrize("replicas,shared,m,k,n", [
    (3, False, 130, 96, 136), (5, True, 130, 96, 136), (1, False, 130, 96, 136),
    (4, False, 512, 512, 64), (3, True, 300, 256, 128),       # 64-column tiled kernel
    (2, False, 512, 512, 256), (3, True, 256, 1024, 512),     # 256-column tiled kernel
    (12, False, 2048, 256, 512), (13, True, 2000, 300, 1024),   # 512-column kernel by replica count
    (24, False, 512, 128, 1024), (25, True, 500, 128, 1000)])   # flat-stream kernel by replica count
def test_spmm_batched_capi(capi, dev, spmm_kernel, replicas, shared, m, k, n):
    _, vals, ri, ro, ci = make_csr(m, k, 0.85, seed=21)
    rng = np.random.default_rng(22)
    b = rng.uniform(-1, 1, size=(replicas, k, n)).astype(np.float32)
    v = vals if shared else rng.uniform(-1, 1, size=(replicas, len(vals))).astype(np.float32)
    want = c_oracle.spmm(m, k, v, ro, ci, b)
    out = torch.full((replicas, m, n), float("nan"), device=dev)
    ws = torch.empty(capi.spmm_workspace_bytes(m, k, n, len(ci)) + 16, dtype=torch.uint8, device=dev)
    capi.spmm_batched(m, k, n, replicas, T(ri, dev), T(v, dev), 0 if shared else len(vals),
                      T(ro, dev), T(ci, dev), T(b, dev), out, ws)
    got = out.cpu().numpy()
    assert not np.isnan(got).any()
    assert rel_err(got, want) < TOL


@pytest.mark.parametrize("m,k,n", [(96, 200, 128), (256, 512, 256), (300, 1000, 512), (200, 300, 64)])
def test_spmm_unsorted_columns(capi, dev, spmm_kernel, m, k, n):
    """Column indices need not ascend inside a row (the CUDA library does not
    require it either).  The larger shapes qualify for the LDS-tiled kernel,
    whose pre-pass must notice the order and hand over to the row-gather kernel."""
    _, vals, ri, ro, ci = make_csr(m, k, 0.8, seed=23)
    rng = np.random.default_rng(24)
    vals, ci = vals.copy(), ci.copy()
    for r in range(m):
        p = rng.permutation(ro[r + 1] - ro[r]) + ro[r]
        vals[ro[r]:ro[r + 1]], ci[ro[r]:ro[r + 1]] = vals[p], ci[p]
    b = rng.uniform(-1, 1, size=(k, n)).astype(np.float32)
    want = c_oracle.spmm(m, k, vals, ro, ci, b)
    out = torch.empty((m, n), device=dev)
    ws = torch.empty(capi.spmm_workspace_bytes(m, k, n, len(ci)) + 16, dtype=torch.uint8, device=dev)
    capi.spmm_batched(m, k, n, 1, T(ri, dev), T(vals, dev), 0, T(ro, dev), T(ci, dev), T(b, dev),
                      out, ws)
    assert rel_err(out.cpu().numpy(), want) < TOL


@pytest.fixture
def flat_loop(monkeypatch):
    """Forces the flat-stream SpMM kernel (csrc/spmm_flat.hip) and one of its
    three loops over the plan's stream: 2 entry, 3 group straight, 4 group diagonal."""
    from torch_sputnik_amd import capi as _capi

    def select(loop):
        monkeypatch.setenv("SPUTNIK_HIP_SPMM_KERNEL", "flat")
        monkeypatch.setenv("SPUTNIK_HIP_SPMM_SPARSE", str(loop))
        _capi.reload_options()
    yield select
    monkeypatch.delenv("SPUTNIK_HIP_SPMM_KERNEL", raising=False)
    monkeypatch.delenv("SPUTNIK_HIP_SPMM_SPARSE", raising=False)
    _capi.reload_options()


FLAT_SHAPES = [
    # m, k, n, sparsity, replicas, empty rows
    (128, 32, 512, 0.0, 1, ()),              # one chunk, every entry shares its column with 7 rows
    (200, 41, 1024, 0.4, 1, (0, 199)),       # partial last chunk (9 rows), padded row slots
    (300, 1000, 1000, 0.9, 2, (17,)),        # partial last column tile, two replicas
    (1000, 33, 1536, 0.5, 1, ()),            # three column tiles (not a multiple of 8), k % 32 = 1
    (384, 4096, 512, 0.97, 3, (1, 2, 3)),    # 128 chunks, most of them empty for a wave
    (2048, 2048, 512, 0.8, 1, ()),           # 16 row blocks
    (136, 70000, 512, 0.9995, 1, ()),        # k beyond the pre-pass's column limit: other kernels take it
]


@pytest.mark.parametrize("m,k,n,sparsity,replicas,empty", FLAT_SHAPES)
def test_spmm_flat_loops_vs_oracle(capi, dev, flat_loop, m, k, n, sparsity, replicas, empty):
    """Every loop of the flat-stream kernel against the oracle on one plan; the
    three sum every output element in the same (CSR) order, so they must also
    agree with each other bit for bit.  The plan is made once, then used with
    fresh values (it depends on the topology alone)."""
    _, vals, ri, ro, ci = make_csr(m, k, sparsity, seed=5 * m + k, round_to=1, empty_rows=empty,
                                   order="random")
    rng = np.random.default_rng(m + n)
    b = rng.uniform(-1, 1, size=(replicas, k, n)).astype(np.float32)
    v = rng.uniform(-1, 1, size=(replicas, len(vals))).astype(np.float32)
    want = c_oracle.spmm(m, k, v, ro, ci, b)
    tri, tro, tci, tb, tv = T(ri, dev), T(ro, dev), T(ci, dev), T(b, dev), T(v, dev)
    outs = []
    for loop in (2, 3, 4):
        flat_loop(loop)
        ws = torch.empty(capi.spmm_workspace_bytes(m, k, n, len(ci)) + 16, dtype=torch.uint8, device=dev)
        capi.spmm_plan(m, k, n, tri, tro, tci, ws)
        out = torch.full((replicas, m, n), float("nan"), device=dev)
        capi.spmm_batched_planned(m, k, n, replicas, tri, tv, len(vals), tro, tci, tb, out, ws)
        got = out.cpu().numpy()
        assert not np.isnan(got).any(), "some output elements were never written"
        assert rel_err(got, want) < TOL
        for r in empty:
            assert np.all(got[:, r] == 0)
        outs.append(out)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


@pytest.mark.parametrize("m,k,n,replicas", [
    (300, 1000, 1001, 2),      # n % 4 = 1, two column tiles, the second almost empty
    (2048, 512, 4095, 1),      # n = 4095 stays on the LDS kernels
    (256, 128, 1535, 1),       # n % 4 = 3, three tiles, the last one a column short
    (128, 96, 510, 1),         # one partial tile
])
def test_spmm_flat_any_n(capi, dev, flat_loop, m, k, n, replicas):
    """n need not be a multiple of 4 (the reference takes any n, src/spmm_cuda.cu:32):
    rows of B and C then start at odd dwords, and the lanes at the end of a row work on
    its last four columns.  Guard zones around the output must stay untouched."""
    _, vals, ri, ro, ci = make_csr(m, k, 0.8, seed=n, round_to=1, order="random")
    rng = np.random.default_rng(n + 1)
    b = rng.uniform(-1, 1, size=(replicas, k, n)).astype(np.float32)
    v = rng.uniform(-1, 1, size=(replicas, len(vals))).astype(np.float32)
    want = c_oracle.spmm(m, k, v, ro, ci, b)
    guard = 64
    for loop in (2, 4):
        flat_loop(loop)
        assert capi.spmm_kernel_name(m, k, n, len(ci), replicas).startswith("spmm_flat_kernel")
        buf = torch.full((guard + replicas * m * n + guard,), 12345.0, device=dev)
        out = buf[guard:guard + replicas * m * n].view(replicas, m, n)
        out.fill_(float("nan"))
        ws = torch.empty(capi.spmm_workspace_bytes(m, k, n, len(ci)) + 16, dtype=torch.uint8, device=dev)
        capi.spmm_batched(m, k, n, replicas, T(ri, dev), T(v, dev), len(vals), T(ro, dev), T(ci, dev),
                          T(b, dev), out, ws)
        got = out.cpu().numpy()
        assert not np.isnan(got).any(), "some output elements were never written"
        assert rel_err(got, want) < TOL
        assert bool((buf[:guard] == 12345.0).all()) and bool((buf[-guard:] == 12345.0).all())


@pytest.mark.parametrize("m,k,n,replicas,kernel", [
    # n in {66, 71, 4095} off the row-gather kernel (a lone product this narrow is
    # below the dispatcher's small-call threshold whatever n % 4 is: 8 replicas of it are not)
    (4096, 4096, 66, 8, "spmm_tiled64_kernel"),
    (4096, 4096, 71, 8, "spmm_tiled64_kernel"),
    (4096, 1024, 4095, 1, "spmm_flat_kernel"),
    (2048, 2048, 131, 8, "spmm_tiled64_kernel"),    # three 64-column tiles, the last 3 columns wide
])
def test_spmm_any_n_stays_on_the_lds_kernels(capi, dev, m, k, n, replicas, kernel):
    """The automatic dispatch for widths that are not a multiple of 4, against the
    dense float64 product on the device, with guard zones around the output."""
    from torch_sputnik_amd.synthetic import random_csr, uniform
    ri, ro, ci, nnz = random_csr(m, k, 0.1, dev, seed=n)
    assert capi.spmm_kernel_name(m, k, n, nnz, replicas).startswith(kernel)
    values = uniform((replicas, nnz), dev, 3)
    dense = uniform((replicas, k, n), dev, 4)
    guard = 64
    buf = torch.full((guard + replicas * m * n + guard,), 12345.0, device=dev)
    out = buf[guard:guard + replicas * m * n].view(replicas, m, n)
    out.fill_(float("nan"))
    ws = torch.empty(capi.spmm_workspace_bytes(m, k, n, nnz) + 16, dtype=torch.uint8, device=dev)
    capi.spmm_batched(m, k, n, replicas, ri, values, nnz, ro, ci, dense, out, ws)
    assert not torch.isnan(out).any()
    assert bool((buf[:guard] == 12345.0).all()) and bool((buf[-guard:] == 12345.0).all())
    rows = torch.repeat_interleave(torch.arange(m, device=dev), torch.diff(ro.long()))
    for r in range(replicas):
        a = torch.zeros(m, k, dtype=torch.float64, device=dev)
        a[rows, ci.long()] = values[r].double()
        assert rel_err_torch(out[r], a @ dense[r].double()) < TOL


def test_spmm_flat_mixed_sorted_and_unsorted_rows(capi, dev, flat_loop):
    """Row blocks with a row whose columns do not ascend take the order-independent
    path inside the same launch; their part of the stream is never read, but the
    prefetch of the block before runs into it (it must hold valid offsets)."""
    m, k, n = 512, 256, 512
    _, vals, ri, ro, ci = make_csr(m, k, 0.7, seed=77, order="identity")
    rng = np.random.default_rng(78)
    vals, ci = vals.copy(), ci.copy()
    for r in (130, 131, 400):            # rows of the second and fourth 128-row block
        p = rng.permutation(ro[r + 1] - ro[r]) + ro[r]
        vals[ro[r]:ro[r + 1]], ci[ro[r]:ro[r + 1]] = vals[p], ci[p]
    b = rng.uniform(-1, 1, size=(k, n)).astype(np.float32)
    want = c_oracle.spmm(m, k, vals, ro, ci, b)
    for loop in (2, 3, 4):
        flat_loop(loop)
        out = torch.full((m, n), float("nan"), device=dev)
        ws = torch.empty(capi.spmm_workspace_bytes(m, k, n, len(ci)) + 16, dtype=torch.uint8, device=dev)
        capi.spmm_batched(m, k, n, 1, T(ri, dev), T(vals, dev), 0, T(ro, dev), T(ci, dev), T(b, dev),
                          out, ws)
        got = out.cpu().numpy()
        assert not np.isnan(got).any()
        assert rel_err(got, want) < TOL


def test_spmm_one_plan_serves_any_replica_count(capi, dev):
    """A plan is made from the topology alone (sputnik_hip_spmm_plan knows no replica
    count), yet the kernel choice depends on it: 2 replicas of this shape run the
    256-column tiles, 12 the 512-column tiles' 64-row form, 24 the flat-stream kernel
    (round 4: taken when the tiles of ALL replicas fill the chip).  One planned
    workspace must serve all of them."""
    m, k, n = 512, 512, 1024   # (round 5: the flat kernel wants k >= 512)
    _, vals, ri, ro, ci = make_csr(m, k, 0.8, seed=31)
    nnz = len(ci)
    names = {r: capi.spmm_kernel_name(m, k, n, nnz, r) for r in (2, 12, 24)}
    assert names[24].startswith("spmm_flat_kernel") and not names[2].startswith("spmm_flat_kernel")
    ws = torch.empty(capi.spmm_workspace_bytes(m, k, n, nnz) + 16, dtype=torch.uint8, device=dev)
    topo = (T(ri, dev), T(ro, dev), T(ci, dev))
    capi.spmm_plan(m, k, n, topo[0], topo[1], topo[2], ws)
    rng = np.random.default_rng(32)
    for replicas in (24, 2, 12, 24):
        b = rng.uniform(-1, 1, size=(replicas, k, n)).astype(np.float32)
        v = rng.uniform(-1, 1, size=(replicas, nnz)).astype(np.float32)
        out = torch.full((replicas, m, n), float("nan"), device=dev)
        capi.spmm_batched_planned(m, k, n, replicas, topo[0], T(v, dev), nnz, topo[1], topo[2],
                                  T(b, dev), out, ws)
        got = out.cpu().numpy()
        assert not np.isnan(got).any()
        assert rel_err(got, c_oracle.spmm(m, k, v, ro, ci, b)) < TOL, f"{replicas} replicas"


@pytest.mark.parametrize("m,k,n,bias", [(1024, 4096, 72, False), (1000, 2100, 200, True),
                                        (4096, 4096, 72, False)])
def test_spmm_narrow_k_split(capi, dev, monkeypatch, m, k, n, bias):
    """Round 4: one product against a narrow dense operand (the reference's own test
    width n = 72, tests/test_spmm.py:13) gives the 64-column kernel few workgroups; the
    K chunks are dealt to several workgroups per tile and the partial tiles added in
    chunk order.  Against the oracle, with the epilogue behind the sum, with a row block
    that takes the order-independent path, twice (deterministic), and identical to the
    unsplit kernel up to the order of the partial sums."""
    monkeypatch.setenv("SPUTNIK_HIP_SPMM_KERNEL", "narrow")
    capi.reload_options()
    try:
        _, vals, ri, ro, ci = make_csr(m, k, 0.97, seed=m + n, round_to=1, empty_rows=(5,))
        ci = ci.copy()
        seg = ci[ro[9]:ro[10]]
        seg[:] = seg[::-1]                       # one row with descending columns
        rng = np.random.default_rng(n)
        b = rng.uniform(-1, 1, size=(k, n)).astype(np.float32)
        bias_v = rng.uniform(-1, 1, size=(m,)).astype(np.float32) if bias else None
        want = c_oracle.spmm(m, k, vals, ro, ci, b)
        if bias:
            want = np.maximum(want + bias_v[:, None], 0)
        topo = (T(ri, dev), T(ro, dev), T(ci, dev))
        ws = torch.empty(capi.spmm_workspace_bytes(m, k, n, len(ci)) + 16, dtype=torch.uint8, device=dev)
        assert ws.numel() > 4 * 2 * m * n, "no room for the partial tiles: the K split is not taken"
        outs = []
        for _ in range(2):
            out = torch.full((m, n + 8), float("nan"), device=dev)[:, :n].contiguous()
            if bias:
                capi.spmm_bias_batched(m, k, n, 1, topo[0], T(vals, dev), 0, topo[1], topo[2], T(b, dev),
                                       T(bias_v, dev), True, out, ws)
            else:
                capi.spmm_batched(m, k, n, 1, topo[0], T(vals, dev), 0, topo[1], topo[2], T(b, dev), out, ws)
            outs.append(out)
        got = outs[0].cpu().numpy()
        assert not np.isnan(got).any()
        assert rel_err(got, want) < TOL
        assert torch.equal(outs[0], outs[1])
        monkeypatch.setenv("SPUTNIK_HIP_SPMM_DEBUG", "64")   # the unsplit form
        capi.reload_options()
        if not bias:
            one = torch.full((m, n), float("nan"), device=dev)
            capi.spmm_batched(m, k, n, 1, topo[0], T(vals, dev), 0, topo[1], topo[2], T(b, dev), one, ws)
            assert rel_err(one.cpu().numpy(), want) < TOL
    finally:
        monkeypatch.delenv("SPUTNIK_HIP_SPMM_KERNEL", raising=False)
        monkeypatch.delenv("SPUTNIK_HIP_SPMM_DEBUG", raising=False)
        capi.reload_options()


@pytest.mark.parametrize("n,k", [(8, 40), (20, 100), (64, 64), (72, 300), (130, 512), (1000, 96)])
def test_workspace_free_kernels_on_every_row_length(capi, dev, monkeypatch, n, k):
    """Round 5: the SpMM row gather and the SDDMM row-wave kernel keep the next window of a
    row in flight and send their gathers out in batches, entries past a row's end gathering
    nothing.  Rows of EVERY length from 0 to past three windows (so every remainder of the
    window and of the batch occurs, next to empty rows), any lane-group width (n: 8 ... 64
    lanes; k: 4 ... 64 lanes and two k slices), NaN in the rows of B / rhs that no entry names
    (a padded gather must not reach the sum).  Bit-identical to the oracle's order: one row,
    ascending entries."""
    rng = np.random.default_rng(n + k)
    lengths = list(range(0, 3 * 64 + 6)) + [0, 0, 1]
    lengths = [min(l, k - 1) for l in lengths]
    m = len(lengths)
    ro = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int32)
    # (column 0 is never named: its row of B / rhs is poisoned below)
    ci = np.concatenate([np.sort(rng.choice(np.arange(1, k), size=l, replace=False)) for l in lengths]
                        ).astype(np.int32)
    ri = np.argsort(-np.asarray(lengths), kind="stable").astype(np.int32)
    vals = rng.uniform(-1, 1, size=len(ci)).astype(np.float32)
    topo = (T(ri, dev), T(ro, dev), T(ci, dev))
    monkeypatch.setenv("SPUTNIK_HIP_SPMM_KERNEL", "gather")
    monkeypatch.setenv("SPUTNIK_HIP_SDDMM_KERNEL", "wave")
    capi.reload_options()
    try:
        b = rng.uniform(-1, 1, size=(k, n)).astype(np.float32)
        want = c_oracle.spmm(m, k, vals, ro, ci, b)
        b[0, :] = np.nan
        out = torch.full((m, n), float("nan"), device=dev)
        capi.spmm_batched(m, k, n, 1, topo[0], T(vals, dev), 0, topo[1], topo[2], T(b, dev), out, None)
        got = out.cpu().numpy()
        assert not np.isnan(got).any()
        assert rel_err(got, want) < TOL
        # SDDMM: mask m x k (the same topology), operands of inner dimension n
        lhs = rng.uniform(-1, 1, size=(1, m, n)).astype(np.float32)
        rhs = rng.uniform(-1, 1, size=(1, k, n)).astype(np.float32)
        want_s = c_oracle.sddmm(m, k, ro, ci, lhs, rhs)
        rhs[0, 0, :] = np.nan
        out_s = torch.full((1, len(ci)), float("nan"), device=dev)
        capi.sddmm_batched(m, n, k, 1, topo[0], topo[1], topo[2], T(lhs, dev), T(rhs, dev), out_s, None)
        got_s = out_s.cpu().numpy()
        assert not np.isnan(got_s).any()
        assert rel_err(got_s, want_s.astype(np.float32), ro) < TOL
    finally:
        monkeypatch.delenv("SPUTNIK_HIP_SPMM_KERNEL", raising=False)
        monkeypatch.delenv("SPUTNIK_HIP_SDDMM_KERNEL", raising=False)
        capi.reload_options()


def test_spmm_kernel_name_reports_the_dispatch(capi):
    assert capi.spmm_kernel_name(4096, 4096, 4096, 1677724, 1).startswith("spmm_flat_kernel")
    # config 5 at its stated size: 64 tiles per replica, the flat kernel from 3 replicas on
    assert capi.spmm_kernel_name(2048, 2048, 2048, 838864, 8).startswith("spmm_flat_kernel")
    assert not capi.spmm_kernel_name(2048, 2048, 2048, 838864, 1).startswith("spmm_flat_kernel")
    assert capi.spmm_kernel_name(64, 64, 64, 2048, 1) == "spmm_rowgather_kernel"
    assert len(capi.build_id()) == 12


@pytest.mark.parametrize("name", ["spmm_c1_64_d050", "spmm_2d_72x64x72", "spmm_3d_r8_72x64x72"])
def test_spmm_op_golden(ts, dev, golden, name):
    g = golden(name)
    out = ts.spmm(int(g["m"]), int(g["k"]), T(g["values"], dev),
                  *topo_t(g["row_indices"], g["row_offsets"], g["column_indices"], dev),
                  T(g["dense"], dev))
    assert tuple(out.shape) == g["expected"].shape
    assert rel_err(out.cpu().numpy(), g["expected"]) < TOL


def test_left_spmm_op(ts, dev):
    m, k, n, r = 256, 128, 72, 3   # tests/test_linear_3d.py:105
    _, vals, ri, ro, ci = make_csr(m, k, 0.9, seed=25, order="ascending")
    b = np.random.default_rng(26).uniform(-1, 1, size=(r, k, n)).astype(np.float32)
    # int64 row_indices must be accepted
    out = ts.left_spmm(m, k, T(vals, dev), T(ri.astype(np.int64), dev), T(ro, dev), T(ci, dev),
                       T(b, dev))
    assert tuple(out.shape) == (r, m, n)
    assert rel_err(out.cpu().numpy(), O.left_spmm(m, k, vals, ri, ro, ci, b)) < TOL
    out2 = ts.left_spmm(m, k, T(vals, dev), T(ri, dev), T(ro, dev), T(ci, dev), T(b[0], dev))
    assert tuple(out2.shape) == (1, m, n)
    assert torch.equal(out2[0], out[0])


# ----------------------------------------------------------------------------
# SDDMM
# ----------------------------------------------------------------------------
SDDMM_SHAPES = [
    # m, k, n, sparsity, replicas
    (72, 64, 72, 0.0, 1),      # tests/test_sddmm.py: dense mask
    (72, 64, 72, 0.9, 4),      # tests/test_sddmm_3d.py (r reduced)
    (50, 7, 60, 0.7, 2),       # odd k -> scalar path
    (50, 10, 60, 0.7, 1),      # k % 2 == 0 -> float2
    (128, 32, 128, 0.9, 3),
    (1024, 64, 1024, 0.9, 2),  # attention block geometry (config 3)
    (200, 300, 150, 0.8, 1),   # k > 256: panels
    (64, 1100, 96, 0.9, 2),    # k > 1024: several panels, out accumulation
    (300, 64, 500, 0.8, 3),    # tiled kernel, ragged row / column blocks
    (256, 128, 256, 0.5, 2),   # tiled kernel, k = 128, > 32 entries per row and chunk
    (64, 64, 64, 0.0, 1),      # tiled kernel, dense mask
    (300, 256, 200, 0.8, 2),   # stationary kernel, k = 256 (128-row slabs), ragged
    (512, 512, 512, 0.8, 2),   # k = 512 (64-row slabs): SparseLinear weight gradient shape
    (130, 512, 70, 0.3, 1),    # k = 512, > 16 entries per row and slab, partial last slab
    (256, 1024, 96, 0.7, 2),   # k = 1024: two panels of 512, the second accumulates
    (64, 1024, 64, 0.0, 1),    # k = 1024, dense mask: full windows in every slab
    (200, 768, 130, 0.8, 2),   # three panels of 256
    (96, 320, 200, 0.6, 1),    # five panels of 64
    (128, 2048, 128, 0.9, 1),  # four panels of 512 (Spmm backward shapes)
    (64, 4160, 64, 0.5, 1),    # 65 x 64: too many panels, row-wave kernel
]


@pytest.mark.parametrize("m,k,n,sparsity,replicas", SDDMM_SHAPES)
def test_sddmm_capi_vs_oracle(capi, dev, sddmm_kernel, m, k, n, sparsity, replicas):
    _, _, ri, ro, ci = make_csr(m, n, sparsity, seed=m + k + n, round_to=1, empty_rows=(m // 2,))
    rng = np.random.default_rng(k)
    lhs = rng.uniform(-1, 1, size=(replicas, m, k)).astype(np.float32)
    rhs = rng.uniform(-1, 1, size=(replicas, n, k)).astype(np.float32)
    want = c_oracle.sddmm(m, n, ro, ci, lhs, rhs)
    out = torch.full((replicas, len(ci)), float("nan"), device=dev)
    ws = torch.empty(capi.sddmm_workspace_bytes(m, k, n, len(ci)) + 16, dtype=torch.uint8, device=dev)
    capi.sddmm_batched(m, k, n, replicas, T(ri, dev), T(ro, dev), T(ci, dev), T(lhs, dev),
                       T(rhs, dev), out, ws)
    got = out.cpu().numpy()
    assert not np.isnan(got).any()
    assert rel_err(got, want, ro) < TOL    # rows = the mask's CSR rows


# The pair-flat kernel (csrc/sddmm_flat.hip, round 4) serves PLANNED products whose rows are
# 128 / 256 bytes (k = 64 in float32): both operands in LDS, a flat list of entry pairs per
# (row block, slab) tile.  Shapes: ragged blocks and slabs, rows of every length incl.
# empty ones, one slab, many slabs, a dense mask, several replicas.
SDDMM_FLAT_SHAPES = [
    (1024, 1024, 0.9, 3),    # config 3's mask in small: 4 row blocks x 8 slabs
    (300, 500, 0.8, 2),      # ragged: 2 row blocks (the second partial), 4 slabs (the last partial)
    (128, 128, 0.5, 1),      # the smallest shape it takes: one partial block, one slab
    (256, 4096, 0.97, 2),    # 32 slabs, about one entry per row and slab: mostly singles
    (700, 130, 0.3, 1),      # two slabs, the second with two columns
    (512, 256, 0.0, 2),      # dense mask: every step full
]


@pytest.fixture
def sddmm_tiled_forced(monkeypatch):
    """Small products take the one-launch row-wave kernel on their own; the test knob
    puts them on the LDS kernels (tests/conftest.py, sddmm_kernel)."""
    from torch_sputnik_amd import capi as _capi
    monkeypatch.setenv("SPUTNIK_HIP_SDDMM_KERNEL", "tiled")
    _capi.reload_options()
    yield
    monkeypatch.delenv("SPUTNIK_HIP_SDDMM_KERNEL", raising=False)
    _capi.reload_options()


@pytest.mark.parametrize("order", ["sorted", "unsorted"])
@pytest.mark.parametrize("m,n,sparsity,replicas", SDDMM_FLAT_SHAPES)
def test_sddmm_flat_planned_vs_oracle(capi, dev, sddmm_tiled_forced, m, n, sparsity, replicas, order):
    k = 64
    _, _, ri, ro, ci = make_csr(m, n, sparsity, seed=m + n, round_to=1, empty_rows=(m // 2, m - 1))
    if order == "unsorted":   # the kernel pairs CSR neighbours wherever their columns lie
        rng = np.random.default_rng(3)
        ci = ci.copy()
        for r in range(0, m, 3):
            seg = ci[ro[r]:ro[r + 1]]
            rng.shuffle(seg)
    nnz = len(ci)
    assert capi.sddmm_kernel_name(m, k, n, nnz, replicas, planned=True) == "sddmm_flat_kernel"
    rng = np.random.default_rng(k + m)
    lhs = rng.uniform(-1, 1, size=(replicas, m, k)).astype(np.float32)
    rhs = rng.uniform(-1, 1, size=(replicas, n, k)).astype(np.float32)
    want = c_oracle.sddmm(m, n, ro, ci, lhs, rhs)
    topo = (T(ri, dev), T(ro, dev), T(ci, dev))
    ws = torch.empty(capi.sddmm_workspace_bytes(m, k, n, nnz) + 16, dtype=torch.uint8, device=dev)
    capi.sddmm_plan(m, k, n, *topo, ws)
    out = torch.full((replicas, nnz), float("nan"), device=dev)
    capi.sddmm_batched_planned(m, k, n, replicas, *topo, T(lhs, dev), T(rhs, dev), out, ws)
    got = out.cpu().numpy()
    assert not np.isnan(got).any(), "an entry was never written"
    assert rel_err(got, want, ro) < TOL
    # the per-call form plans its tables itself and takes the rhs-stationary kernel; it
    # leaves the lists behind the tables alone, so the planned workspace still serves.
    # (Not the same bits: which quad computes an entry decides the order of its partial
    # sums; both kernels are held to the oracle.)
    out2 = torch.full((replicas, nnz), float("nan"), device=dev)
    capi.sddmm_batched(m, k, n, replicas, *topo, T(lhs, dev), T(rhs, dev), out2, ws)
    assert rel_err(out2.cpu().numpy(), want, ro) < TOL
    out3 = torch.full((replicas, nnz), float("nan"), device=dev)
    capi.sddmm_batched_planned(m, k, n, replicas, *topo, T(lhs, dev), T(rhs, dev), out3, ws)
    assert torch.equal(out, out3), "the planned product is not reproducible"


# sum over the replicas inside the call (the gradient of values shared by a batch):
# against the oracle's per-replica products added in float64
SDDMM_SUM_SHAPES = [
    (512, 1024, 512, 0.9, 8),   # attention projection weight gradient: 2 panels x 8 replicas, one launch
    (512, 512, 512, 0.8, 3),    # one panel per replica
    (256, 1024, 96, 0.7, 1),    # one replica, two panels: still summed from partials
    (200, 768, 130, 0.8, 2),    # three panels of 256
    (100, 40, 60, 0.5, 5),      # row-wave kernel ([R, nnz] partials)
    (64, 64, 64, 0.0, 1),       # nothing to sum
    (33, 100, 47, 0.6, 4),      # odd sizes: scalar reduction
    (512, 512, 512, 0.9, 8),    # the summed form cuts k into 256-wide panels (16-wave workgroups)
    (300, 2048, 260, 0.95, 2),  # k = 2048: 8 panels per replica
    (256, 4096, 256, 0.95, 2),  # k = 4096: 16 panels per replica in one launch
    (128, 384, 200, 0.8, 3),    # k = 384: 128-wide panels (8-wave workgroups)
]


@pytest.mark.parametrize("planned", [False, True])
@pytest.mark.parametrize("m,k,n,sparsity,replicas", SDDMM_SUM_SHAPES)
def test_sddmm_sum_capi_vs_oracle(capi, dev, sddmm_kernel, sddmm_sum_slab, m, k, n, sparsity, replicas,
                                  planned):
    round_to = 1 if m == 33 else 4
    _, _, ri, ro, ci = make_csr(m, n, sparsity, seed=m + k + n, round_to=round_to,
                                empty_rows=(m // 2,))
    rng = np.random.default_rng(k + 1)
    lhs = rng.uniform(-1, 1, size=(replicas, m, k)).astype(np.float32)
    rhs = rng.uniform(-1, 1, size=(replicas, n, k)).astype(np.float32)
    want = c_oracle.sddmm(m, n, ro, ci, lhs, rhs).astype(np.float64).sum(axis=0)
    out = torch.full((len(ci),), float("nan"), device=dev)
    ws = torch.empty(capi.sddmm_sum_workspace_bytes(m, k, n, len(ci)) + 16, dtype=torch.uint8,
                     device=dev)
    scratch = torch.empty(capi.sddmm_sum_scratch_bytes(m, k, n, len(ci), replicas) + 16,
                          dtype=torch.uint8, device=dev)
    topo = (T(ri, dev), T(ro, dev), T(ci, dev))
    if planned:
        capi.sddmm_sum_plan(m, k, n, *topo, ws)
    capi.sddmm_sum_batched(m, k, n, replicas, *topo, T(lhs, dev), T(rhs, dev), out, ws, scratch,
                           planned=planned)
    got = out.cpu().numpy()
    assert not np.isnan(got).any()
    # one row of `replicas * k` products per entry: the tolerance scales as for one long row
    assert rel_err(got[None, :], want[None, :].astype(np.float32), ro) < TOL


def test_sddmm_sum_op_equals_the_summed_op(ts, dev):
    m, k, n, replicas = 512, 1024, 512, 8
    _, _, ri, ro, ci = make_csr(m, n, 0.9, seed=5)
    topo = [T(x, dev) for x in (ri, ro, ci)]
    rng = np.random.default_rng(1)
    lhs = T(rng.uniform(-1, 1, (replicas, m, k)).astype(np.float32), dev)
    rhs = T(rng.uniform(-1, 1, (replicas, n, k)).astype(np.float32), dev)
    each = ts.sddmm(m, n, *topo, lhs, rhs)
    total = ts.sddmm_sum(m, n, *topo, lhs, rhs)
    assert total.shape == (len(ci),)
    torch.testing.assert_close(total, each.double().sum(0).float(), rtol=2e-5, atol=2e-4)
    plan = ts.sddmm_sum_plan(m, n, k, *topo)
    assert torch.equal(ts.sddmm_sum_planned(m, n, *topo, lhs, rhs, plan), total)   # deterministic
    assert torch.equal(ts.sddmm_sum(m, n, *topo, lhs[0], rhs[0]).reshape(-1)[:8].isfinite(),
                       torch.ones(8, dtype=torch.bool, device=dev))


@pytest.mark.parametrize("name", ["sddmm_2d_dense_mask", "sddmm_3d_r8"])
def test_sddmm_op_golden(ts, dev, golden, name):
    g = golden(name)
    out = ts.sddmm(int(g["m"]), int(g["n"]),
                   *topo_t(g["row_indices"], g["row_offsets"], g["column_indices"], dev),
                   T(g["lhs"], dev), T(g["rhs"], dev))
    assert tuple(out.shape) == g["expected"].shape
    assert rel_err(out.cpu().numpy(), g["expected"], g["row_offsets"]) < TOL


# ----------------------------------------------------------------------------
# sparse softmax
# ----------------------------------------------------------------------------
@pytest.mark.parametrize("m,n,sparsity,replicas", [
    (72, 72, 0.9, 1), (72, 72, 0.0, 2), (1024, 1024, 0.9, 3), (64, 4096, 0.5, 2),
    (300, 700, 0.3, 1), (17, 5000, 0.1, 1)])
def test_softmax_capi_vs_oracle(capi, dev, m, n, sparsity, replicas):
    _, vals, ri, ro, ci = make_csr(m, n, sparsity, seed=m + n, round_to=1, empty_rows=(0, m - 1))
    rng = np.random.default_rng(m)
    v = rng.uniform(-8, 8, size=(replicas, len(vals))).astype(np.float32)
    want = c_oracle.sparse_softmax(v, ro, ci)
    out = torch.full((replicas, len(vals)), float("nan"), device=dev)
    capi.sparse_softmax_batched(m, replicas, T(v, dev), T(ri, dev), T(ro, dev), T(ci, dev), out)
    got = out.cpu().numpy()
    assert not np.isnan(got).any()
    assert rel_err(got, want, ro) < TOL


@pytest.mark.parametrize("m,n,sparsity,replicas", [(300, 700, 0.85, 3), (64, 64, 0.3, 2),
                                                   (100, 2000, 0.9, 2), (40, 3000, 0.5, 1)])
@pytest.mark.parametrize("in_phase,out_phase", [(0, 0), (1, 1), (3, 2), (2, 0)])
def test_softmax_unaligned_buffers(capi, dev, m, n, sparsity, replicas, in_phase, out_phase):
    """The kernel moves rows as aligned 16-byte pieces: buffers that start 1-3
    floats past a 16-byte boundary, replica strides that are not multiples of
    four (nnz is odd here) and outputs aligned differently from the inputs must
    give the same answer, and nothing outside the output may be written."""
    _, vals, ri, ro, ci = make_csr(m, n, sparsity, seed=m + n, round_to=1, empty_rows=(0, m // 2))
    nnz = len(vals)
    rng = np.random.default_rng(m)
    v = rng.uniform(-6, 6, size=(replicas, nnz)).astype(np.float32)
    want = c_oracle.sparse_softmax(v, ro, ci)
    src = torch.zeros(replicas * nnz + 8, device=dev)
    src[in_phase:in_phase + replicas * nnz] = T(v, dev).reshape(-1)
    dst = torch.full((replicas * nnz + 8,), 7.0, device=dev)
    capi.sparse_softmax_batched(m, replicas, src[in_phase:in_phase + replicas * nnz].view(replicas, nnz),
                                T(ri, dev), T(ro, dev), T(ci, dev),
                                dst[out_phase:out_phase + replicas * nnz].view(replicas, nnz))
    got = dst.cpu().numpy()
    assert np.all(got[:out_phase] == 7.0) and np.all(got[out_phase + replicas * nnz:] == 7.0)
    assert rel_err(got[out_phase:out_phase + replicas * nnz].reshape(replicas, nnz), want, ro) < TOL


def test_softmax_op_golden(ts, dev, golden):
    g = golden("softmax_72x72")
    out = ts.sparse_softmax(T(g["values"], dev),
                            *topo_t(g["row_indices"], g["row_offsets"], g["column_indices"], dev))
    assert rel_err(out.cpu().numpy(), g["expected"], g["row_offsets"]) < TOL


def test_softmax_large_magnitudes(ts, dev):
    """max-subtraction: huge scores must not overflow."""
    _, vals, ri, ro, ci = make_csr(40, 90, 0.5, seed=31, round_to=1)
    v = (vals * 2000 - 1000).astype(np.float32)
    out = ts.sparse_softmax(T(v, dev), T(ri, dev), T(ro, dev), T(ci, dev)).cpu().numpy()
    assert np.isfinite(out).all()
    assert np.max(np.abs(out - O.sparse_softmax(v, ri, ro, ci))) < TOL


# ----------------------------------------------------------------------------
# CSR transpose (bit-exact)
# ----------------------------------------------------------------------------
@pytest.mark.parametrize("m,n,sparsity,replicas,empty", [
    (4, 4, 0.0, 1, (0,)),                 # tests/test_transpose.py
    (72, 64, 0.8, 1, ()),
    (257, 1000, 0.9, 1, (0, 256)),
    (2048, 2048, 0.8, 1, ()),             # config 5 geometry
    (1500, 300, 0.5, 3, (7,)),            # batched values (extension)
    (64, 20000, 0.99, 1, ()),             # n > 16384: global-counter path
    (3000, 64, 0.3, 2, ()),               # rows_per_chunk > 1
])
def test_transpose_capi_bit_exact(capi, dev, m, n, sparsity, replicas, empty):
    _, vals, _, ro, ci = make_csr(m, n, sparsity, seed=m + 3 * n, round_to=1, empty_rows=empty)
    rng = np.random.default_rng(n)
    v = vals if replicas == 1 else rng.uniform(size=(replicas, len(vals))).astype(np.float32)
    want = O.csr_transpose(m, n, v, ro, ci)
    nnz = len(ci)
    out_v = torch.full(v.shape, float("nan"), device=dev)
    out_ro = torch.full((n + 1,), -1, dtype=torch.int32, device=dev)
    out_ci = torch.full((nnz,), -1, dtype=torch.int32, device=dev)
    perm = torch.full((nnz,), -1, dtype=torch.int32, device=dev)
    ws = torch.empty(capi.csr_transpose_workspace_bytes(m, n, nnz), dtype=torch.uint8, device=dev)
    capi.csr_transpose(m, n, replicas, T(v, dev), T(ro, dev), T(ci, dev), out_v, out_ro, out_ci,
                       perm, ws)
    assert np.array_equal(out_ro.cpu().numpy(), want[1])
    assert np.array_equal(out_ci.cpu().numpy(), want[2])
    assert np.array_equal(out_v.cpu().numpy(), want[0])
    assert np.array_equal(np.asarray(v)[..., perm.cpu().numpy()], want[0])


def _random_sparse_csr(m, n, nnz, seed):
    """nnz distinct (row, column) pairs of an m x n matrix, as CSR (numpy, no dense mask)."""
    rng = np.random.default_rng(seed)
    flat = np.unique(rng.integers(0, m * n, size=int(nnz * 1.05), dtype=np.int64))[:nnz]
    rows, cols = flat // n, (flat % n).astype(np.int32)
    ro = np.zeros(m + 1, dtype=np.int32)
    np.add.at(ro, rows + 1, 1)
    return np.cumsum(ro).astype(np.int32), cols


@pytest.mark.parametrize("m,n,nnz,replicas", [
    (4096, 4096, 8000, 1),          # tables would be 64 entries per nonzero: histogram path
    (3000, 70000, 50000, 2),        # wide, batched values
    (65536, 65536, 429497, 1),      # 65536^2 at density 1e-4, tables would be 1 GiB
])
def test_transpose_very_sparse_path_bit_exact(capi, dev, m, n, nnz, replicas):
    """The O(n + nnz) path (csrc/transpose.hip) against scipy's stable CSR -> CSC
    conversion, bit for bit, within a workspace that scales with the nonzeros."""
    import scipy.sparse as sp
    ro, ci = _random_sparse_csr(m, n, nnz, seed=m + n)
    nnz = len(ci)
    rng = np.random.default_rng(1)
    v = rng.uniform(size=(replicas, nnz)).astype(np.float32)
    ws_bytes = capi.csr_transpose_workspace_bytes(m, n, nnz)
    assert ws_bytes <= 4 * (2 * n + 3 * nnz) + 64 and ws_bytes < 64 * 2**20
    out_v = torch.full(v.shape, float("nan"), device=dev)
    out_ro = torch.full((n + 1,), -1, dtype=torch.int32, device=dev)
    out_ci = torch.full((nnz,), -1, dtype=torch.int32, device=dev)
    perm = torch.full((nnz,), -1, dtype=torch.int32, device=dev)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    capi.csr_transpose(m, n, replicas, T(v, dev), T(ro, dev), T(ci, dev), out_v, out_ro, out_ci,
                       perm, ws, checked=True)
    # scipy: entry ids as data -> the permutation; tocsc() keeps rows ascending per column
    a = sp.csr_matrix((np.arange(1, nnz + 1, dtype=np.int64), ci, ro), shape=(m, n)).tocsc()
    assert np.array_equal(out_ro.cpu().numpy(), a.indptr.astype(np.int32))
    assert np.array_equal(out_ci.cpu().numpy(), a.indices.astype(np.int32))
    want_perm = (a.data - 1).astype(np.int64)
    assert np.array_equal(perm.cpu().numpy(), want_perm)
    assert np.array_equal(out_v.cpu().numpy(), v[:, want_perm])
    # without a permutation output the path keeps its own
    out_v2 = torch.full(v.shape, float("nan"), device=dev)
    capi.csr_transpose(m, n, replicas, T(v, dev), T(ro, dev), T(ci, dev), out_v2, out_ro, out_ci,
                       None, ws)
    assert torch.equal(out_v2, out_v)


@pytest.mark.parametrize("m,n,nnz,passes", [(65536, 65536, 300000, 1), (300000, 40000, 200000, 2)])
def test_transpose_very_sparse_with_global_token_columns(capi, dev, m, n, nnz, passes):
    """a very sparse, very large mask in which a few columns are held by
    EVERY row (a global-attention token) -- the O(n + nnz) path must not rank such an
    output row by all pairs; bit-exact against scipy, in a bounded time."""
    import time
    import scipy.sparse as sp
    ro, ci = _random_sparse_csr(m, n, nnz, seed=3 * m + n)
    a = sp.csr_matrix((np.ones(len(ci), dtype=np.int8), ci, ro), shape=(m, n)).tolil()
    a[:, 5] = 1
    a[::2, n - 3] = 1          # every second row: a second long column
    a = a.tocsr()
    a.sort_indices()
    ro, ci = a.indptr.astype(np.int32), a.indices.astype(np.int32)
    nnz = len(ci)
    v = np.random.default_rng(2).uniform(size=(1, nnz)).astype(np.float32)
    ws_bytes = capi.csr_transpose_workspace_bytes(m, n, nnz)
    assert ws_bytes <= 4 * (2 * n + 3 * nnz) + 64, "not the O(n + nnz) path"
    out_v = torch.full(v.shape, float("nan"), device=dev)
    out_ro = torch.full((n + 1,), -1, dtype=torch.int32, device=dev)
    out_ci = torch.full((nnz,), -1, dtype=torch.int32, device=dev)
    perm = torch.full((nnz,), -1, dtype=torch.int32, device=dev)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    args = (m, n, 1, T(v, dev), T(ro, dev), T(ci, dev), out_v, out_ro, out_ci, perm, ws)
    capi.csr_transpose(*args, checked=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    capi.csr_transpose(*args, checked=True)
    torch.cuda.synchronize()
    assert time.perf_counter() - t0 < 0.25, "a long output row is ranked by all pairs again"   # (microseconds now; seconds before)
    want = sp.csr_matrix((np.arange(1, nnz + 1, dtype=np.int64), ci, ro), shape=(m, n)).tocsc()
    assert np.array_equal(out_ro.cpu().numpy(), want.indptr.astype(np.int32))
    assert np.array_equal(out_ci.cpu().numpy(), want.indices.astype(np.int32))
    assert np.array_equal(perm.cpu().numpy(), (want.data - 1).astype(np.int64))
    assert np.array_equal(out_v.cpu().numpy(), v[:, (want.data - 1).astype(np.int64)])
    # a row that holds the long column twice is detected there too
    bad = ci.copy()
    row = m // 2
    assert ro[row + 1] - ro[row] >= 2
    bad[ro[row]:ro[row + 1]] = 5
    with pytest.raises(RuntimeError):
        capi.csr_transpose(m, n, 1, T(v, dev), T(ro, dev), T(bad, dev), out_v, out_ro, out_ci, perm,
                           ws, checked=True)
    torch.cuda.synchronize()


@pytest.mark.parametrize("m,n,sparsity", [(300, 200, 0.8), (4096, 4096, 0.9995)],
                         ids=["table_path", "histogram_path"])
def test_transpose_detects_a_repeated_column(capi, ts, dev, m, n, sparsity):
    """A row that stores a column twice has no transpose slot for the second entry
    (VERDICT r2): both paths set the status word; the checked entry and the op that
    caches permutations refuse the pattern, a valid one passes."""
    _, vals, _, ro, ci = make_csr(m, n, sparsity, seed=9, round_to=1)
    nnz = len(ci)

    def run(cols):
        out_v = torch.empty(nnz, device=dev)
        out_ro = torch.empty(n + 1, dtype=torch.int32, device=dev)
        out_ci = torch.empty(nnz, dtype=torch.int32, device=dev)
        ws = torch.empty(capi.csr_transpose_workspace_bytes(m, n, nnz), dtype=torch.uint8, device=dev)
        capi.csr_transpose(m, n, 1, T(vals, dev), T(ro, dev), T(cols, dev), out_v, out_ro, out_ci,
                           None, ws, checked=True)

    run(ci)                                            # valid: no complaint
    row = int(np.argmax(np.diff(ro) >= 2))             # a row with at least two entries
    bad = ci.copy()
    bad[ro[row] + 1] = bad[ro[row]]                    # ... now holds its first column twice
    with pytest.raises(RuntimeError):
        run(bad)
    with pytest.raises(RuntimeError, match="valid CSR"):
        torch.ops.torch_sputnik.csr_transpose_with_permutation(m, n, T(vals, dev), T(ro, dev),
                                                               T(bad, dev))
    # the per-call form stays asynchronous and does not raise
    torch.ops.torch_sputnik.csr_transpose_with_permutation(m, n, T(vals, dev), T(ro, dev),
                                                           T(bad, dev), False)
    # a column outside [0, n) is reported on both paths too (the table path
    # used to skip it silently and hand back uninitialised slots)
    for wrong in (n, n + 7, -3):
        bad = ci.copy()
        bad[ro[row]] = wrong
        with pytest.raises(RuntimeError):
            run(bad)
    torch.cuda.synchronize()


@pytest.mark.parametrize("name", ["transpose_4x4_row0_zero", "transpose_72x64"])
def test_transpose_op_golden(ts, dev, golden, name):
    g = golden(name)
    vt, rot, cit = ts.csr_transpose(int(g["m"]), int(g["n"]), T(g["values"], dev),
                                    T(g["row_offsets"], dev), T(g["column_indices"], dev))
    assert np.array_equal(vt.cpu().numpy(), g["values_t"])
    assert np.array_equal(rot.cpu().numpy(), g["row_offsets_t"])
    assert np.array_equal(cit.cpu().numpy(), g["column_indices_t"])
    assert rot.dtype == torch.int32 and cit.dtype == torch.int32


def test_transpose_round_trip_full_size(ts, dev):
    """Size-independent property at config-5 size: transposing twice is the identity."""
    m = n = 2048
    _, vals, _, ro, ci = make_csr(m, n, 0.8, seed=41)
    v, r, c = T(vals, dev), T(ro, dev), T(ci, dev)
    vt, rot, cit = ts.csr_transpose(m, n, v, r, c)
    vtt, rott, citt = ts.csr_transpose(n, m, vt, rot, cit)
    assert torch.equal(vtt, v) and torch.equal(rott, r) and torch.equal(citt, c)


# ----------------------------------------------------------------------------
# determinism, empty inputs, errors
# ----------------------------------------------------------------------------
def test_bitwise_reproducible(ts, dev):
    m, k, n = 512, 512, 512
    _, vals, ri, ro, ci = make_csr(m, k, 0.9, seed=51)
    b = np.random.default_rng(52).uniform(-1, 1, size=(k, n)).astype(np.float32)
    args = (m, k, T(vals, dev), T(ri, dev), T(ro, dev), T(ci, dev), T(b, dev))
    first = ts.spmm(*args)
    for _ in range(3):
        assert torch.equal(ts.spmm(*args), first)


def test_empty_topology(ts, dev):
    m, k, n = 5, 4, 8
    ro = torch.zeros(m + 1, dtype=torch.int32, device=dev)
    ci = torch.zeros(0, dtype=torch.int32, device=dev)
    ri = torch.arange(m, dtype=torch.int32, device=dev)
    vals = torch.zeros(0, device=dev)
    out = ts.spmm(m, k, vals, ri, ro, ci, torch.ones(k, n, device=dev))
    assert tuple(out.shape) == (m, n) and torch.all(out == 0)
    assert ts.sddmm(m, k, ri, ro, ci, torch.ones(m, 3, device=dev),
                    torch.ones(k, 3, device=dev)).numel() == 0
    assert ts.sparse_softmax(vals, ri, ro, ci).numel() == 0
    vt, rot, cit = ts.csr_transpose(m, k, vals, ro, ci)
    assert rot.cpu().tolist() == [0] * (k + 1) and vt.numel() == 0 and cit.numel() == 0


def test_hip_graph_capture_of_attention_ops(ts, dev):
    """The ops neither synchronise nor allocate outside torch's allocator, so a
    launch-bound chain (SDDMM -> softmax -> SpMM) can be captured into a HIP graph
    and replayed on new data."""
    s_len, d, r = 256, 64, 4
    mask, _, ri, ro, ci = make_csr(s_len, s_len, 0.9, seed=81, round_to=1)
    topo = (T(ri, dev), T(ro, dev), T(ci, dev))
    q = torch.randn(r, s_len, d, device=dev)
    k = torch.randn(r, s_len, d, device=dev)
    v = torch.randn(r, s_len, d, device=dev)

    def block():
        scores = ts.sddmm(s_len, s_len, *topo, q, k) / 8.0
        probs = ts.sparse_softmax(scores, *topo)
        return ts.spmm(s_len, s_len, probs, *topo, v)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            block()  # warm-up outside capture
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        captured = block()
    q.copy_(torch.randn_like(q))
    v.copy_(torch.randn_like(v))
    graph.replay()
    torch.cuda.synchronize()
    eager = block()
    assert torch.equal(captured, eager)
    # and against the dense definition
    dense_scores = torch.matmul(q.double(), k.double().transpose(1, 2)) / 8.0
    dense_scores = dense_scores.masked_fill(T(mask, dev) == 0, float("-inf"))
    want = torch.matmul(torch.nan_to_num(torch.softmax(dense_scores, -1)), v.double())
    assert rel_err(captured.cpu().numpy(), want.cpu().numpy()) < TOL


def test_half_storage_is_widened(ts, dev):
    """fp16 storage (BASELINE config 5): operands widened once, fp32 math and output."""
    m, k, n, r = 128, 96, 64, 2
    _, vals, ri, ro, ci = make_csr(m, k, 0.8, seed=71)
    b = np.random.default_rng(72).uniform(-1, 1, size=(r, k, n)).astype(np.float32)
    v16, b16 = T(vals, dev).half(), T(b, dev).half()
    out = ts.left_spmm(m, k, v16, T(ri, dev), T(ro, dev), T(ci, dev), b16)
    assert out.dtype == torch.float32
    want = O.left_spmm(m, k, v16.float().cpu().numpy(), ri, ro, ci, b16.float().cpu().numpy())
    assert rel_err(out.cpu().numpy(), want) < TOL


def test_shape_errors_raise(ts, dev):
    _, vals, ri, ro, ci = make_csr(8, 8, 0.5, seed=61)
    v, r, o, c = T(vals, dev), T(ri, dev), T(ro, dev), T(ci, dev)
    with pytest.raises(RuntimeError):
        ts.spmm(8, 8, v, r, o, c, torch.ones(9, 4, device=dev))      # k mismatch
    with pytest.raises(RuntimeError):
        ts.spmm(8, 8, v[:-1], r, o, c, torch.ones(8, 4, device=dev))  # nnz mismatch
    with pytest.raises(RuntimeError):
        ts.spmm(8, 8, v.double(), r, o, c, torch.ones(8, 4, device=dev))  # dtype
    with pytest.raises(RuntimeError):
        ts.sddmm(8, 8, r, o, c, torch.ones(8, 4, device=dev), torch.ones(8, 5, device=dev))
    with pytest.raises((RuntimeError, NotImplementedError)):
        ts.spmm(8, 8, v.cpu(), r.cpu(), o.cpu(), c.cpu(), torch.ones(8, 4))  # no CPU path
