"""CPU: the oracle against the committed golden vectors and the dense
definitions the reference's tests use; the C restatement against the numpy one."""
import numpy as np
import pytest

from oracle import c_oracle
from oracle import sputnik_oracle as O
from helpers import make_csr, rel_err


def _topo(g):
    return g["row_indices"], g["row_offsets"], g["column_indices"]


@pytest.mark.parametrize("name", ["spmm_c1_64_d050", "spmm_2d_72x64x72", "spmm_3d_r8_72x64x72"])
def test_spmm_golden(golden, name):
    g = golden(name)
    got = O.spmm(int(g["m"]), int(g["k"]), g["values"], *_topo(g), g["dense"])
    assert got.shape == g["expected"].shape
    assert rel_err(got, g["expected"]) < 1e-9


@pytest.mark.parametrize("name", ["sddmm_2d_dense_mask", "sddmm_3d_r8"])
def test_sddmm_golden(golden, name):
    g = golden(name)
    got = O.sddmm(int(g["m"]), int(g["n"]), *_topo(g), g["lhs"], g["rhs"])
    assert got.shape == g["expected"].shape
    assert rel_err(got, g["expected"]) < 1e-9


def test_softmax_golden(golden):
    g = golden("softmax_72x72")
    got = O.sparse_softmax(g["values"], *_topo(g))
    assert rel_err(got, g["expected"]) < 1e-9
    # every non-empty row sums to one
    sums = np.add.reduceat(got, g["row_offsets"][:-1][np.diff(g["row_offsets"]) > 0])
    assert np.allclose(sums, 1.0, atol=1e-12)


@pytest.mark.parametrize("name", ["transpose_4x4_row0_zero", "transpose_72x64"])
def test_transpose_golden(golden, name):
    g = golden(name)
    vt, rot, cit = O.csr_transpose(int(g["m"]), int(g["n"]), g["values"], g["row_offsets"],
                                   g["column_indices"])
    assert np.array_equal(vt, g["values_t"])
    assert np.array_equal(rot, g["row_offsets_t"])
    assert np.array_equal(cit, g["column_indices_t"])


def test_left_spmm_matches_spmm_with_shared_values():
    dense_a, vals, ri, ro, ci = make_csr(33, 29, 0.7, seed=5)
    rng = np.random.default_rng(6)
    b = rng.uniform(size=(4, 29, 17)).astype(np.float32)
    left = O.left_spmm(33, 29, vals, ri, ro, ci, b)
    tiled = O.spmm(33, 29, np.tile(vals, (4, 1)), ri, ro, ci, b)
    assert left.shape == (4, 33, 17)
    assert np.array_equal(left, tiled)
    assert rel_err(left, np.matmul(dense_a.astype(np.float64), b.astype(np.float64))) < 1e-12
    # 2-D dense still gives a 3-D result (src/left_replicated_spmm.cu:30)
    assert O.left_spmm(33, 29, vals, ri, ro, ci, b[0]).shape == (1, 33, 17)


def test_row_indices_order_is_irrelevant():
    _, vals, ri, ro, ci = make_csr(40, 24, 0.8, seed=7, empty_rows=(0, 13, 39))
    b = np.random.default_rng(8).uniform(size=(24, 10)).astype(np.float32)
    base = O.spmm(40, 24, vals, ri, ro, ci, b)
    for order in (O.diffsort(ro), np.arange(40, dtype=np.int32),
                  np.random.default_rng(9).permutation(40).astype(np.int32)):
        assert np.array_equal(O.spmm(40, 24, vals, order, ro, ci, b), base)
    assert np.all(base[[0, 13, 39]] == 0)


def test_diffsort_is_ascending_length():
    # SURVEY.md quirk Q1 probe: row lengths [1,3,0,2,5,1] -> [2,0,5,3,1,4]
    ro = np.concatenate(([0], np.cumsum([1, 3, 0, 2, 5, 1])))
    assert O.diffsort(ro).tolist() == [2, 0, 5, 3, 1, 4]


def test_transpose_twice_is_identity():
    _, vals, _, ro, ci = make_csr(37, 53, 0.85, seed=10, empty_rows=(5,))
    vt, rot, cit = O.csr_transpose(37, 53, vals, ro, ci)
    vtt, rott, citt = O.csr_transpose(53, 37, vt, rot, cit)
    assert np.array_equal(vtt, vals) and np.array_equal(rott, ro) and np.array_equal(citt, ci)


def test_empty_matrix():
    ro = np.zeros(6, np.int32)
    ci = np.zeros(0, np.int32)
    vals = np.zeros(0, np.float32)
    ri = np.arange(5, dtype=np.int32)
    assert np.all(O.spmm(5, 4, vals, ri, ro, ci, np.ones((4, 3), np.float32)) == 0)
    assert O.sddmm(5, 4, ri, ro, ci, np.ones((5, 2)), np.ones((4, 2))).shape == (0,)
    assert O.sparse_softmax(vals, ri, ro, ci).shape == (0,)
    vt, rot, cit = O.csr_transpose(5, 4, vals, ro, ci)
    assert rot.tolist() == [0] * 5 and vt.shape == (0,) and cit.shape == (0,)


@pytest.mark.skipif(not c_oracle.available(), reason="oracle/libsputnik_oracle.so not built")
class TestCOracle:
    def test_spmm(self):
        _, vals, ri, ro, ci = make_csr(129, 200, 0.9, seed=11, empty_rows=(3, 128))
        b = np.random.default_rng(12).uniform(-1, 1, size=(200, 77)).astype(np.float32)
        want = O.spmm(129, 200, vals, ri, ro, ci, b)
        assert rel_err(c_oracle.spmm(129, 200, vals, ro, ci, b), want) < 1e-6
        assert rel_err(c_oracle.spmm(129, 200, vals, ro, ci, b, f32_accumulate=True), want) < 1e-4

    def test_spmm_batched_and_shared_values(self):
        _, vals, ri, ro, ci = make_csr(31, 40, 0.6, seed=13)
        rng = np.random.default_rng(14)
        b = rng.uniform(size=(3, 40, 9)).astype(np.float32)
        v3 = rng.uniform(size=(3, vals.shape[0])).astype(np.float32)
        assert rel_err(c_oracle.spmm(31, 40, v3, ro, ci, b), O.spmm(31, 40, v3, ri, ro, ci, b)) < 1e-6
        assert rel_err(c_oracle.spmm(31, 40, vals, ro, ci, b), O.left_spmm(31, 40, vals, ri, ro, ci, b)) < 1e-6

    def test_sddmm(self):
        _, _, ri, ro, ci = make_csr(50, 60, 0.8, seed=15, empty_rows=(7,))
        rng = np.random.default_rng(16)
        lhs = rng.uniform(-1, 1, size=(2, 50, 33)).astype(np.float32)
        rhs = rng.uniform(-1, 1, size=(2, 60, 33)).astype(np.float32)
        assert rel_err(c_oracle.sddmm(50, 60, ro, ci, lhs, rhs), O.sddmm(50, 60, ri, ro, ci, lhs, rhs)) < 1e-6

    def test_softmax(self):
        _, vals, ri, ro, ci = make_csr(64, 300, 0.5, seed=17, empty_rows=(0, 63))
        vals = (vals * 20 - 10).astype(np.float32)
        assert rel_err(c_oracle.sparse_softmax(vals, ro, ci), O.sparse_softmax(vals, ri, ro, ci)) < 1e-6

    def test_transpose(self):
        _, vals, _, ro, ci = make_csr(45, 38, 0.7, seed=18, empty_rows=(44,))
        v2 = np.stack([vals, vals * 2])
        for v in (vals, v2):
            got = c_oracle.csr_transpose(45, 38, v, ro, ci)
            want = O.csr_transpose(45, 38, v, ro, ci)
            for a, b in zip(got, want):
                assert np.array_equal(a, b)
