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


def _row_view(a, row_offsets):
    """(values as [rows, width] or flat, row id per element or None)."""
    a = np.asarray(a, np.float64)
    if row_offsets is None:
        if a.ndim >= 2 and a.shape[-1] < 16:
            return a.reshape(-1, a.shape[-2] * a.shape[-1]), None
        return (a.reshape(-1, a.shape[-1]) if a.ndim >= 2 else a.reshape(1, -1)), None
    lengths = np.diff(np.asarray(row_offsets, np.int64))
    ids = np.repeat(np.arange(len(lengths)), lengths)
    return a.reshape(-1, a.shape[-1]), ids


def rel_err(got, expected, row_offsets=None):
    """Worst violation of the fp32 parity bound, as a multiple of 1: the test
    is ``rel_err(...) < 1e-4`` (the north star's tolerance).  Two criteria, both
    PER ROW of the output (the unit one wavefront group accumulates):

      (1) |got - want| / (|want| + rowmean|want|)       every element
      (2) |got - want| / |want|                         elements with
                                                         |want| > 1e-2 * rowmax|want|

    (1) is rtol*|want| + atol with atol = rtol * mean magnitude of the SAME
    row: a float32 sum of K products carries an absolute error ~1e-7*sum|a*b|,
    so entries that cancel to nearly zero cannot be held to a purely relative
    bound, but the slack they get comes from their own row, never from a
    larger row elsewhere in the tensor.  (2) is a pure relative check on
    every entry that is not a cancellation.

    A "row" is the last dimension for dense outputs; for CSR-ordered outputs
    ([nnz] or [R, nnz]) pass ``row_offsets`` and it is the CSR row.  Dense
    outputs narrower than 16 columns (n = 1, 7 ...) have no row to speak of: a
    single cancelling element would be its own scale, so there the scale is
    taken over the whole [m, n] matrix of the replica; likewise CSR rows of fewer
    than 16 entries take the scale of their replica.
    """
    got = np.asarray(got, np.float64)
    expected = np.asarray(expected, np.float64)
    if got.size == 0:
        return 0.0
    assert got.shape == expected.shape, (got.shape, expected.shape)
    w, ids = _row_view(expected, row_offsets)
    g = got.reshape(w.shape)
    aw = np.abs(w)
    err = np.abs(g - w)
    if ids is None:
        rowmean = aw.mean(axis=1, keepdims=True)
        rowmax = aw.max(axis=1, keepdims=True)
    else:
        rows = int(ids.max()) + 1 if ids.size else 0
        count = np.maximum(np.bincount(ids, minlength=rows), 1)
        rowmean = np.stack([np.bincount(ids, weights=r, minlength=rows) / count for r in aw])
        rowmax = np.zeros((aw.shape[0], rows))
        for r in range(aw.shape[0]):
            np.maximum.at(rowmax[r], ids, aw[r])
        # CSR rows of fewer than 16 entries are no scale either (a row of ONE entry
        # that cancels to nearly zero would be held to a relative bound of itself):
        # such rows take the replica's mean / maximum, like narrow dense outputs
        short = count < 16
        rowmean[:, short] = aw.mean(axis=1, keepdims=True)
        rowmax[:, short] = aw.max(axis=1, keepdims=True)
        rowmean, rowmax = rowmean[:, ids], rowmax[:, ids]
    worst = float(np.max(err / (aw + np.maximum(rowmean, 1e-30))))
    significant = aw > 1e-2 * rowmax
    if significant.any():
        worst = max(worst, float(np.max(err[significant] / aw[significant])))
    return worst


def rel_err_torch(got, want):
    """rel_err on device tensors (full-size checks): rows = last dimension,
    ``want`` in float64."""
    import torch
    want = want.to(torch.float64)
    err = (got.to(torch.float64) - want).abs()
    aw = want.abs()
    rowmean = aw.mean(dim=-1, keepdim=True).clamp_min(1e-30)
    rowmax = aw.amax(dim=-1, keepdim=True)
    worst = float((err / (aw + rowmean)).max())
    significant = aw > 1e-2 * rowmax
    if bool(significant.any()):
        worst = max(worst, float((err[significant] / aw[significant]).max()))
    return worst
