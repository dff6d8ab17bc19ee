"""GPU: seeded random shapes for every operator against the oracle -- ragged
sizes around the tile boundaries of the kernels (64 / 128 / 256 rows and
columns), empty rows, every row order, 1..5 replicas.  One process, a few
hundred small launches; sizes are kept where the numpy oracle takes
milliseconds."""
import numpy as np
import pytest
import torch

from oracle import sputnik_oracle as O
from helpers import make_csr, rel_err

pytestmark = pytest.mark.gpu

# SPUTNIK_FUZZ_SCALE=10 runs ten times the cases from a different seed (soak run)
import os
SCALE = int(os.environ.get("SPUTNIK_FUZZ_SCALE", "1"))
SEED_SHIFT = 0 if SCALE == 1 else 1000003

TOL = 1e-4
ORDERS = ("descending", "ascending", "random", "identity")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def capi():
    from torch_sputnik_amd import capi
    return capi


def T(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def _dims(rng, choices):
    return int(rng.choice(choices)) + int(rng.integers(-3, 4)) * int(rng.random() < 0.5)


def _case(rng, rows, cols):
    m = max(1, _dims(rng, rows))
    n = max(1, _dims(rng, cols))
    sparsity = float(rng.choice([0.0, 0.5, 0.8, 0.9, 0.97]))
    empty = tuple(int(x) for x in rng.integers(0, m, size=int(rng.integers(0, 3))))
    order = ORDERS[int(rng.integers(0, len(ORDERS)))]
    replicas = int(rng.integers(1, 6))
    return m, n, sparsity, empty, order, replicas


def test_fuzz_spmm(capi, dev, spmm_kernel):
    rng = np.random.default_rng(20261003 + SEED_SHIFT)
    for it in range(40 * SCALE):
        m, k, sparsity, empty, order, replicas = _case(rng, [16, 64, 128, 256, 300], [32, 64, 128, 256, 520])
        n = int(rng.choice([1, 7, 18, 64, 128, 192, 256, 512]))
        _, vals, ri, ro, ci = make_csr(m, k, sparsity, seed=it, round_to=1, empty_rows=empty,
                                       order=order)
        shared = bool(rng.random() < 0.5)
        values = vals if shared else rng.uniform(-1, 1, (replicas, len(ci))).astype(np.float32)
        b = rng.uniform(-1, 1, (replicas, k, n)).astype(np.float32)
        bias = rng.uniform(-1, 1, m).astype(np.float32) if rng.random() < 0.5 else None
        relu = bool(rng.random() < 0.5)
        want = np.stack([O.spmm_bias(m, k, values if shared else values[r], ri, ro, ci, bias, b[r],
                                     relu=relu) for r in range(replicas)])
        out = torch.full((replicas, m, n), float("nan"), device=dev)
        ws = torch.empty(capi.spmm_workspace_bytes(m, k, n, len(ci)) + 16, dtype=torch.uint8,
                         device=dev)
        dummy = torch.zeros(1, dtype=torch.int32, device=dev)
        capi.spmm_bias_batched(m, k, n, replicas, T(ri, dev), T(values, dev) if len(ci) else
                               torch.zeros(1, device=dev), 0 if shared else len(ci), T(ro, dev),
                               T(ci, dev) if len(ci) else dummy, T(b, dev),
                               None if bias is None else T(bias, dev), relu, out, ws)
        got = out.cpu().numpy()
        assert not np.isnan(got).any(), (it, m, k, n)
        assert rel_err(got, want) < TOL, (it, m, k, n, sparsity, order, replicas)


def test_fuzz_sddmm_planned_pair_flat(capi, dev, monkeypatch):
    """The pair-flat SDDMM (csrc/sddmm_flat.hip, round 4: planned products with rows of 128
    / 256 bytes) on random masks: any m, n >= 128, densities from one entry per row to
    dense, columns ascending or shuffled inside rows, empty rows, float32 / float16 /
    bfloat16 operands with k = 64 (and k = 128 in half), float32 output -- against the
    oracle on the rounded operands, and the same bits from a second planned call."""
    monkeypatch.setenv("SPUTNIK_HIP_SDDMM_KERNEL", "tiled")
    capi.reload_options()
    try:
        rng = np.random.default_rng(177 + SEED_SHIFT)
        for it in range(24 * SCALE):
            m = int(rng.choice([128, 200, 256, 300, 513, 777, 1024]))
            n = int(rng.choice([128, 130, 256, 500, 1024, 2048]))
            sparsity = float(rng.choice([0.0, 0.5, 0.9, 0.97, 0.995]))
            replicas = int(rng.integers(1, 4))
            dtype = [torch.float32, torch.float16, torch.bfloat16][int(rng.integers(0, 3))]
            k = 64 if dtype == torch.float32 else int(rng.choice([64, 128]))
            empty = tuple(int(x) for x in rng.choice(m, size=int(rng.integers(0, 4)), replace=False))
            _, _, ri, ro, ci = make_csr(m, n, sparsity, seed=300 + it, round_to=1, empty_rows=empty)
            nnz = len(ci)
            if nnz < 4 * m:
                continue
            if rng.random() < 0.5:   # shuffled columns: pairs are CSR neighbours wherever they lie
                ci = ci.copy()
                for r in range(0, m, 2):
                    rng.shuffle(ci[ro[r]:ro[r + 1]])
            assert capi.sddmm_kernel_name(m, k, n, nnz, replicas, 4 if dtype == torch.float32 else 2,
                                          planned=True) == "sddmm_flat_kernel"
            lhs = torch.from_numpy(rng.uniform(-1, 1, (replicas, m, k)).astype(np.float32)).to(dev).to(dtype)
            rhs = torch.from_numpy(rng.uniform(-1, 1, (replicas, n, k)).astype(np.float32)).to(dev).to(dtype)
            want = O.sddmm(m, n, ri, ro, ci, lhs.float().cpu().numpy(), rhs.float().cpu().numpy())
            topo = (T(ri, dev), T(ro, dev), T(ci, dev))
            ws = torch.empty(capi.sddmm_workspace_bytes(m, k, n, nnz) + 16, dtype=torch.uint8, device=dev)
            capi.sddmm_plan(m, k, n, *topo, ws)
            outs = []
            for _ in range(2):
                out = torch.full((replicas, nnz), float("nan"), device=dev)
                capi.sddmm_typed(m, k, n, replicas, *topo, lhs, rhs, out, ws, planned=True)
                outs.append(out)
            got = outs[0].cpu().numpy()
            assert not np.isnan(got).any(), (it, m, k, n, sparsity, dtype)
            assert rel_err(got, want, ro) < TOL, (it, m, k, n, sparsity, dtype)
            assert torch.equal(outs[0], outs[1]), (it, m, k, n)
    finally:
        monkeypatch.delenv("SPUTNIK_HIP_SDDMM_KERNEL", raising=False)
        capi.reload_options()


def test_fuzz_sddmm_softmax_transpose(capi, dev, sddmm_kernel):
    rng = np.random.default_rng(77 + SEED_SHIFT)
    for it in range(40 * SCALE):
        m, n, sparsity, empty, order, replicas = _case(rng, [16, 64, 128, 256, 300], [16, 64, 128, 256, 300])
        k = int(rng.choice([1, 5, 32, 64, 64, 128, 128, 200, 256, 320, 512, 768, 1024]))
        _, _, ri, ro, ci = make_csr(m, n, sparsity, seed=100 + it, round_to=1, empty_rows=empty,
                                    order=order)
        nnz = len(ci)
        if nnz == 0:
            continue
        lhs = rng.uniform(-1, 1, (replicas, m, k)).astype(np.float32)
        rhs = rng.uniform(-1, 1, (replicas, n, k)).astype(np.float32)
        d_ri, d_ro, d_ci = T(ri, dev), T(ro, dev), T(ci, dev)
        ws = torch.empty(capi.sddmm_workspace_bytes(m, k, n, nnz) + 16, dtype=torch.uint8, device=dev)
        scores = torch.full((replicas, nnz), float("nan"), device=dev)
        capi.sddmm_batched(m, k, n, replicas, d_ri, d_ro, d_ci, T(lhs, dev), T(rhs, dev), scores, ws)
        got = scores.cpu().numpy()
        assert not np.isnan(got).any(), (it, m, k, n)
        assert rel_err(got, O.sddmm(m, n, ri, ro, ci, lhs, rhs), ro) < TOL, (it, m, k, n, sparsity)

        scale = float(rng.choice([1.0, 0.125, 2.0]))
        probs = capi.sparse_softmax_scaled_batched(m, replicas, scores, d_ri, d_ro, d_ci, scale,
                                                   torch.empty_like(scores))
        want_p = O.sparse_softmax_scaled(got, ri, ro, ci, scale)
        assert rel_err(probs.cpu().numpy(), want_p, ro) < TOL, (it, m, n)

        tws = torch.empty(capi.csr_transpose_workspace_bytes(m, n, nnz), dtype=torch.uint8, device=dev)
        vt = torch.empty_like(probs)
        rot = torch.empty(n + 1, dtype=torch.int32, device=dev)
        cit = torch.empty(nnz, dtype=torch.int32, device=dev)
        perm = torch.empty(nnz, dtype=torch.int32, device=dev)
        capi.csr_transpose(m, n, replicas, probs, d_ro, d_ci, vt, rot, cit, perm, tws)
        w_vt, w_rot, w_cit = O.csr_transpose(m, n, probs.cpu().numpy(), ro, ci)
        assert np.array_equal(rot.cpu().numpy(), w_rot) and np.array_equal(cit.cpu().numpy(), w_cit)
        assert np.array_equal(vt.cpu().numpy(), w_vt)
        assert np.array_equal(probs.cpu().numpy()[:, perm.cpu().numpy()], w_vt)


def test_fuzz_csr_transpose_many_mask(capi, dev):
    """Batches of masks of mixed density (some empty, some dense) with 1..4 heads, on the
    region workspace (all masks in one launch per phase) and on the single-mask one: values,
    offsets, indices and permutation against the oracle, bit for bit."""
    rng = np.random.default_rng(20261005 + SEED_SHIFT)
    for it in range(24 * SCALE):
        m = max(2, _dims(rng, [16, 33, 64, 100, 256]))
        n = max(2, _dims(rng, [16, 40, 128, 300, 512]))
        b = int(rng.integers(2, 7))
        heads = int(rng.integers(1, 5))
        masks = np.stack([O.random_mask(m, n, float(rng.choice([0.0, 0.5, 0.9, 0.97, 1.0])),
                                        round_to=1, rng=rng) for _ in range(b)])
        ri, ro, ci, nn = O.dense_to_csr_many_mask(masks)
        width = int(nn.max())
        if width == 0:
            continue
        r = b * heads
        values = rng.uniform(-1, 1, (r, width + int(rng.integers(0, 5)))).astype(np.float32)
        regions = bool(rng.integers(0, 2))
        nbytes = (capi.csr_transpose_many_mask_workspace_bytes(b, m, n, width) if regions
                  else capi.csr_transpose_workspace_bytes(m, n, width))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        vt = torch.full(values.shape, -7.0, device=dev)
        rot = torch.empty(b, n + 1, dtype=torch.int32, device=dev)
        cit = torch.full((max(len(ci), 1),), -1, dtype=torch.int32, device=dev)
        perm = torch.full((max(len(ci), 1),), -1, dtype=torch.int32, device=dev)
        capi.csr_transpose_many_mask(b, m, n, nn, r, T(values, dev), T(ro, dev), T(ci, dev), vt, rot,
                                     cit, perm, ws)
        w_vt, w_rot, w_cit = O.csr_transpose_many_mask(b, m, n, nn, values[:, :width], ro, ci)
        tag = f"case {it}: b={b} heads={heads} m={m} n={n} nnz={list(nn)} regions={regions}"
        assert np.array_equal(rot.cpu().numpy(), w_rot), tag
        assert np.array_equal(cit.cpu().numpy()[:len(ci)], w_cit), tag
        got, first = vt.cpu().numpy(), 0
        for i in range(b):
            n_i = int(nn[i])
            rows = slice(i * heads, (i + 1) * heads)
            assert np.array_equal(got[rows, :n_i], w_vt[rows, :n_i]), tag
            assert (got[rows, n_i:] == -7.0).all(), tag
            p_i = perm[first:first + n_i].cpu().numpy()
            assert np.array_equal(values[rows][:, p_i], w_vt[rows, :n_i]), tag
            first += n_i


def test_fuzz_sparse_attention(capi, dev):
    rng = np.random.default_rng(5150 + SEED_SHIFT)
    for it in range(25 * SCALE):
        m, n, sparsity, empty, order, replicas = _case(rng, [16, 64, 128, 256, 300], [16, 64, 128, 256, 300])
        _, _, ri, ro, ci = make_csr(m, n, sparsity, seed=200 + it, round_to=1, empty_rows=empty,
                                    order=order)
        if len(ci) == 0:
            continue
        q = rng.uniform(-2, 2, (replicas, m, 64)).astype(np.float32)
        k = rng.uniform(-2, 2, (replicas, n, 64)).astype(np.float32)
        v = rng.uniform(-1, 1, (replicas, n, 64)).astype(np.float32)
        ws = torch.empty(capi.sparse_attention_workspace_bytes(m, n, 64, len(ci)), dtype=torch.uint8,
                         device=dev)
        out = torch.full((replicas, m, 64), float("nan"), device=dev)
        capi.sparse_attention_forward(m, n, 64, replicas, T(ri, dev), T(ro, dev), T(ci, dev),
                                      T(q, dev), T(k, dev), T(v, dev), 0.125, out, None, ws)
        got = out.cpu().numpy()
        assert not np.isnan(got).any(), (it, m, n)
        assert rel_err(got, O.sparse_attention(q, k, v, ri, ro, ci, 0.125)) < TOL, (it, m, n, sparsity)


def test_fuzz_round2_entry_points(dev):
    """Random shapes for the entry points added in round 2, through the ops (so
    that unsupported shapes exercise the fall-back compositions as well): the
    product stored transposed in row blocks, with and without values gathered
    through a permutation; the SDDMM summed over the replicas; the banded
    permutation."""
    import torch_sputnik as ts
    from torch_sputnik_amd import ops
    from torch_sputnik_amd.topology import diffsort
    rng = np.random.default_rng(777 + SEED_SHIFT)
    for it in range(30 * SCALE):
        block = int(rng.choice([32, 64, 128, 256]))
        m = block * int(rng.integers(1, 9))
        k = max(4, _dims(rng, [32, 64, 200, 512, 600, 1100]))
        n = int(rng.choice([8, 20, 64, 72, 128, 200, 256]))
        sparsity = float(rng.choice([0.0, 0.5, 0.9, 0.97]))
        replicas = int(rng.integers(1, 5))
        left = bool(rng.random() < 0.5)
        order = ORDERS[int(rng.integers(0, len(ORDERS)))]
        _, vals, ri, ro, ci = make_csr(m, k, sparsity, seed=500 + it, round_to=1, order=order,
                                       empty_rows=(int(rng.integers(0, m)),))
        if len(ci) == 0:
            continue
        values = vals if left else rng.uniform(-1, 1, (replicas, len(ci))).astype(np.float32)
        b = rng.uniform(-1, 1, (replicas, k, n)).astype(np.float32)
        topo = [T(x, dev) for x in (ri, ro, ci)]
        want = np.stack([O.spmm(m, k, values if left else values[r], ri, ro, ci, b[r])
                         for r in range(replicas)])
        got = ts.spmm_transposed_out(m, k, T(values, dev), *topo, T(b, dev), block, left=left)
        assert got.shape == (replicas * (m // block), n, block)
        # (compared in the product's own layout: the per-row criterion of rel_err is
        # about rows of C, not about columns of its blocks)
        back = got.transpose(1, 2).reshape(replicas, m, n)
        assert rel_err(back.cpu().numpy(), want) < TOL, (it, m, k, n, block, left)

        # the transposed product A^T @ X with the values kept in A's order, stored in blocks
        if k % 64 == 0 and k >= 64:
            x = rng.uniform(-1, 1, (replicas, m, n)).astype(np.float32)
            v1 = values if left else values[0]
            _, ro_t, ci_t, perm = ts.csr_transpose_with_permutation(m, k, T(v1, dev), topo[1], topo[2])
            ri_t = diffsort(ro_t)
            got_t = ts.spmm_transposed_out(k, m, T(v1, dev), ri_t, ro_t, ci_t, T(x, dev), 64,
                                           permutation=perm, left=True)
            dense = np.zeros((m, k))
            dense[np.repeat(np.arange(m), np.diff(ro)), ci] = v1
            want2 = np.einsum("mk,rmn->rkn", dense, x.astype(np.float64)).astype(np.float32)
            back_t = got_t.transpose(1, 2).reshape(replicas, k, n)
            assert rel_err(back_t.cpu().numpy(), want2) < TOL, (it, m, k, n, "permuted")

        # SDDMM summed over the replicas (mask m x k2 with inner dimension n2)
        n2 = int(rng.choice([20, 64, 128, 192, 512, 1024]))
        lhs = rng.uniform(-1, 1, (replicas, m, n2)).astype(np.float32)
        rhs = rng.uniform(-1, 1, (replicas, k, n2)).astype(np.float32)
        total = ts.sddmm_sum(m, k, *topo, T(lhs, dev), T(rhs, dev))
        want3 = O.sddmm(m, k, ri, ro, ci, lhs, rhs).astype(np.float64).reshape(replicas, -1).sum(0)
        assert rel_err(total.cpu().numpy()[None, :], want3[None, :].astype(np.float32), ro) < TOL, \
            (it, m, k, n2, replicas)

    for it in range(10 * SCALE):
        n = int(rng.integers(1, 70000))
        rows = int(rng.integers(1, 5))
        perm = torch.from_numpy(rng.permutation(n).astype(np.int32)).to(dev)
        values = torch.from_numpy(rng.uniform(-1, 1, (rows, n)).astype(np.float32)).to(dev)
        lists = ops.banded_lists(perm)
        assert torch.equal(ops.permute_last_banded(values, *lists), values[:, perm.long()]), (n, rows)
