"""Shared input builders for the parity tests (numpy only; seeds are explicit)."""
import numpy as np

from oracle import sputnik_oracle as O


def make_csr(m, n, sparsity, seed, round_to=4, empty_rows=(), order="descending"):
    """Random CSR pattern with the reference's input distribution
    (tests/connectors.py:34-59), optional forced-empty rows, and a choice of
    row_indices order (callers pass both)."""
    rng = np.random.default_rng(seed)
    mask = O.random_mask(m, n, sparsity, round_to=round_to, rng=rng)
    for r in empty_rows:
        mask[r, :] = 0
    values = (rng.uniform(0.0, 1.0, size=(m, n)).astype(np.float32) + np.float32(1e-3)) * mask
    vals, row_indices, row_offsets, column_indices = O.dense_to_csr(values)
    if order == "ascending":
        row_indices = O.diffsort(row_offsets)
    elif order == "random":
        row_indices = rng.permutation(m).astype(np.int32)
    elif order == "identity":
        row_indices = np.arange(m, dtype=np.int32)
    return values.astype(np.float32), vals, row_indices, row_offsets, column_indices


def rel_err(got, expected):
    """max |got - want| / (|want| + mean|want|).

    The north star's bound is 1e-4 on fp32 outputs.  A float32 sum of K
    products carries an absolute error of about 1e-7 * sum|a*b|, so outputs
    that cancel to nearly zero cannot be held to a purely relative bound;
    adding the mean magnitude of the expected tensor to the denominator is the
    usual rtol*|want| + atol test with atol = rtol * mean|want|.
    """
    got = np.asarray(got, np.float64)
    expected = np.asarray(expected, np.float64)
    if got.size == 0:
        return 0.0
    scale = np.abs(expected) + max(1e-30, float(np.mean(np.abs(expected))))
    return float(np.max(np.abs(got - expected) / scale))
