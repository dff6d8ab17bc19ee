"""numpy restatement of the five torch_sputnik ops (TEST INFRASTRUCTURE ONLY).

Every function computes in float64 from the given (normally float32) inputs
and returns float64, so a float32 device kernel is compared against it with a
relative tolerance instead of against another float32 summation order.

Citations are to files under the reference checkout (``/root/reference``).
The arithmetic the reference runs lives in the un-vendored
``google-research/sputnik`` submodule (``.gitmodules:1-3``, pin unknown) and
in cuSPARSE; what is restated here is the contract visible at the reference's
call sites plus the dense definitions the reference's own tests compare with.
"""
import numpy as np


# --------------------------------------------------------------------------
# topology helpers
# --------------------------------------------------------------------------
def _check_topology(m, row_indices, row_offsets, column_indices):
    row_offsets = np.asarray(row_offsets).astype(np.int64)
    column_indices = np.asarray(column_indices).astype(np.int64)
    assert row_offsets.ndim == 1 and row_offsets.shape[0] == m + 1
    assert column_indices.ndim == 1
    assert row_offsets[0] == 0 and row_offsets[-1] == column_indices.shape[0]
    assert np.all(np.diff(row_offsets) >= 0)
    if row_indices is not None:
        row_indices = np.asarray(row_indices).astype(np.int64)
        assert row_indices.shape == (m,)
        # row_indices is only a processing order ("row swizzle"); results must
        # not depend on it.  It has to be a permutation of the rows.
        assert np.array_equal(np.sort(row_indices), np.arange(m))
    return row_offsets, column_indices


def _rows_of(row_offsets):
    """row id of every stored element, CSR order."""
    m = row_offsets.shape[0] - 1
    return np.repeat(np.arange(m, dtype=np.int64), np.diff(row_offsets))


def diffsort(row_offsets):
    """row_indices as the reference's modules build them.

    modules/spmm.py:4-6 -- ``argsort((offsets - roll(offsets, -1))[:-1],
    descending=True)``: the differences are minus the row lengths, so the
    order is by ASCENDING row length.
    """
    row_offsets = np.asarray(row_offsets).astype(np.int64)
    diffs = (row_offsets - np.roll(row_offsets, -1))[:-1]
    return np.argsort(-diffs, kind="stable").astype(np.int32)


def dense_to_csr(matrix):
    """Dense 2-D array -> (values, row_indices, row_offsets, column_indices).

    Restates tests/sparse_matrix.py:9-41 without the ``.cuda()`` calls:
    values in row-major order, ``row_indices = argsort(-row_length)``
    (descending lengths, :22), column indices ascending within a row.
    """
    matrix = np.asarray(matrix)
    assert matrix.ndim == 2
    mask = matrix != 0
    values = matrix[mask].astype(np.float32)
    row_offsets = np.concatenate(([0], np.cumsum(mask.sum(axis=1)))).astype(np.int32)
    row_indices = np.argsort(-np.diff(row_offsets), kind="stable").astype(np.int32)
    column_indices = np.nonzero(mask)[1].astype(np.int32)
    return values, row_indices, row_offsets, column_indices


def random_mask(m, n, sparsity, round_to=1, rng=None):
    """0/1 mask with the distribution of tests/connectors.py:34-59.

    ``num_dormant = round(sparsity * size)`` positions chosen uniformly
    without replacement are zero; with ``round_to > 1`` the number of
    nonzeros is rounded UP to a multiple of it (:49-52).
    """
    rng = np.random.default_rng(0) if rng is None else rng
    size = m * n
    if sparsity == 0.0:
        return np.ones((m, n), dtype=np.float32)
    num_dormant = int(round(sparsity * size))
    if round_to > 1:
        nnz = size - num_dormant
        nnz = (nnz + round_to - 1) // round_to * round_to
        num_dormant = size - nnz
    mask = np.ones(size, dtype=np.float32)
    mask[rng.choice(size, num_dormant, replace=False)] = 0.0
    return mask.reshape(m, n)


# --------------------------------------------------------------------------
# the five ops
# --------------------------------------------------------------------------
def _spmm_one(m, n, values, rows, column_indices, dense):
    out = np.zeros((m, n), dtype=np.float64)
    if values.shape[0]:
        # accumulate a_ij * B[j, :] into C[i, :] in CSR order
        np.add.at(out, rows, values[:, None] * dense[column_indices])
    return out


def spmm(m, k, values, row_indices, row_offsets, column_indices, dense):
    """C = A_csr @ B.  src/spmm_cuda.cu:9-60; definition tests/test_spmm.py:9-10.

    ``values`` [nnz] with ``dense`` [k,n] -> [m,n]; or ``values`` [R,nnz] with
    ``dense`` [R,k,n] -> [R,m,n] (topology shared, values per replica,
    src/spmm_cuda.cu:48-57).
    """
    row_offsets, column_indices = _check_topology(m, row_indices, row_offsets, column_indices)
    values = np.asarray(values, dtype=np.float64)
    dense = np.asarray(dense, dtype=np.float64)
    assert values.ndim in (1, 2) and dense.ndim == values.ndim + 1
    assert values.shape[-1] == column_indices.shape[0]
    assert dense.shape[-2] == k
    assert column_indices.size == 0 or (column_indices.min() >= 0 and column_indices.max() < k)
    n = dense.shape[-1]
    rows = _rows_of(row_offsets)
    if values.ndim == 1:
        return _spmm_one(m, n, values, rows, column_indices, dense)
    assert values.shape[0] == dense.shape[0]
    if values.shape[0] == 0:
        return np.zeros((0, m, n))
    return np.stack([_spmm_one(m, n, values[r], rows, column_indices, dense[r])
                     for r in range(values.shape[0])])


def left_spmm(m, k, values, row_indices, row_offsets, column_indices, dense):
    """C_r = A_csr @ B_r with ONE sparse matrix.  src/left_replicated_spmm.cu:8-44.

    ``values`` [nnz] is shared by all replicas (pointer not offset, :35);
    ``dense`` [R,k,n] or [k,n]; the output is ALWAYS 3-D [R,m,n] (:30).
    """
    row_offsets, column_indices = _check_topology(m, row_indices, row_offsets, column_indices)
    values = np.asarray(values, dtype=np.float64)
    dense = np.asarray(dense, dtype=np.float64)
    assert values.ndim == 1
    if dense.ndim == 2:
        dense = dense[None]
    assert dense.shape[-2] == k
    n = dense.shape[-1]
    rows = _rows_of(row_offsets)
    if dense.shape[0] == 0:
        return np.zeros((0, m, n))
    return np.stack([_spmm_one(m, n, values, rows, column_indices, dense[r])
                     for r in range(dense.shape[0])])


def sddmm(m, n, row_indices, row_offsets, column_indices, lhs, rhs):
    """out[p] = <lhs[i_p,:], rhs[j_p,:]> for every stored (i_p, j_p), CSR order.

    src/sddmm_cuda.cu:7-57; definition tests/test_sddmm.py:8-13 and
    tests/test_sddmm_3d.py:9-14 (``lhs @ rhs^T`` sampled at the mask).
    lhs [m,k] / [R,m,k], rhs [n,k] / [R,n,k] -> [nnz] / [R,nnz].
    """
    row_offsets, column_indices = _check_topology(m, row_indices, row_offsets, column_indices)
    lhs = np.asarray(lhs, dtype=np.float64)
    rhs = np.asarray(rhs, dtype=np.float64)
    assert lhs.ndim == rhs.ndim and lhs.ndim in (2, 3)
    assert lhs.shape[-1] == rhs.shape[-1]
    assert lhs.shape[-2] == m and rhs.shape[-2] == n
    assert column_indices.size == 0 or (column_indices.min() >= 0 and column_indices.max() < n)
    rows = _rows_of(row_offsets)

    def one(l, r):
        return np.einsum("pk,pk->p", l[rows], r[column_indices])

    if lhs.ndim == 2:
        return one(lhs, rhs)
    assert lhs.shape[0] == rhs.shape[0]
    if lhs.shape[0] == 0:
        return np.zeros((0, column_indices.shape[0]))
    return np.stack([one(lhs[r], rhs[r]) for r in range(lhs.shape[0])])


def sparse_softmax(values, row_indices, row_offsets, column_indices):
    """Row-wise softmax over the STORED entries only.

    src/softmax_cuda.cu:7-46; dense definition tests/test_softmax.py:9-22
    (zeros replaced by -1e9 before a dense softmax).  values [nnz] / [R,nnz].
    Empty rows contribute nothing.
    """
    m = np.asarray(row_offsets).shape[0] - 1
    row_offsets, column_indices = _check_topology(m, row_indices, row_offsets, column_indices)
    values = np.asarray(values, dtype=np.float64)
    assert values.ndim in (1, 2) and values.shape[-1] == column_indices.shape[0]
    rows = _rows_of(row_offsets)

    def one(x):
        row_max = np.full(m, -np.inf)
        np.maximum.at(row_max, rows, x)
        e = np.exp(x - row_max[rows])
        row_sum = np.zeros(m)
        np.add.at(row_sum, rows, e)
        return e / row_sum[rows]

    if values.ndim == 1:
        return one(values)
    if values.shape[0] == 0:
        return np.zeros((0, column_indices.shape[0]))
    return np.stack([one(values[r]) for r in range(values.shape[0])])


def csr_transpose(m, n, values, row_offsets, column_indices):
    """CSR(m x n) -> CSR of the transpose (n x m).

    src/transpose_cuda.cu:45-102 (cusparseCsr2cscEx2, CSR2CSC_ALG1, :90-99):
    returns (values_t [nnz], row_offsets_t [n+1], column_indices_t [nnz]).
    ALG1 is a stable counting sort by column, so within each output row the
    original row ids ascend.  ``values`` may also be [R,nnz] (extension,
    SURVEY.md section 8f rank 1): the same permutation is applied per replica.
    """
    row_offsets, column_indices = _check_topology(m, None, row_offsets, column_indices)
    values = np.asarray(values)
    assert values.shape[-1] == column_indices.shape[0]
    assert column_indices.size == 0 or (column_indices.min() >= 0 and column_indices.max() < n)
    rows = _rows_of(row_offsets)
    order = np.argsort(column_indices, kind="stable")
    values_t = values[..., order]
    column_indices_t = rows[order].astype(np.int32)
    counts = np.bincount(column_indices, minlength=n)
    row_offsets_t = np.concatenate(([0], np.cumsum(counts))).astype(np.int32)
    return values_t, row_offsets_t, column_indices_t


# --------------------------------------------------------------------------
# extensions: epilogues, softmax gradient, many-mask
# --------------------------------------------------------------------------
def spmm_bias(m, k, values, row_indices, row_offsets, column_indices, bias, dense, relu=False):
    """``spmm`` + per-output-row bias (+ ReLU).

    Call site tests/test_spmm_bias_relu.py:35-37 (the check at :43 compares
    with ``dense_result + 1`` for ``bias = ones(m)``: one bias per output ROW,
    broadcast over the n columns).
    """
    out = spmm(m, k, values, row_indices, row_offsets, column_indices, dense)
    if bias is not None:
        bias = np.asarray(bias, dtype=np.float64)
        assert bias.shape == (m,)
        out = out + bias[:, None]
    return np.maximum(out, 0.0) if relu else out


def sparse_softmax_scaled(values, row_indices, row_offsets, column_indices, scale):
    """softmax(scale * x): the ``/ math.sqrt(d)`` of modules/sparse_attention.py:72
    folded into the softmax."""
    return sparse_softmax(np.asarray(values, np.float64) * float(scale), row_indices,
                          row_offsets, column_indices)


def sparse_softmax_backward(softmax_out, grad_out, row_offsets, scale=1.0):
    """dX for Y = softmax(scale * X) over stored entries:
    ``scale * Y * (dY - rowsum(dY * Y))``.  (tests/transformer/functions.py:70-120
    sketches the wrapper; its formula is not the softmax Jacobian -- SURVEY.md 8f.)"""
    row_offsets = np.asarray(row_offsets).astype(np.int64)
    m = row_offsets.shape[0] - 1
    y = np.asarray(softmax_out, np.float64)
    g = np.asarray(grad_out, np.float64)
    assert y.shape == g.shape and y.shape[-1] == row_offsets[-1]
    rows = _rows_of(row_offsets)

    def one(y1, g1):
        dot = np.zeros(m)
        np.add.at(dot, rows, y1 * g1)
        return float(scale) * y1 * (g1 - dot[rows])

    if y.ndim == 1:
        return one(y, g)
    if y.shape[0] == 0:
        return np.zeros_like(y)
    return np.stack([one(y[r], g[r]) for r in range(y.shape[0])])


def sparse_attention(q, k, v, row_indices, row_offsets, column_indices, scale):
    """spmm(softmax(scale * sddmm(q, k)), v): the chain of
    modules/sparse_attention.py:66-82 (sddmm :68-71, ``/ math.sqrt(d)`` :72,
    sparse_softmax :76, spmm :79-82).  q [R,m,d], k and v [R,n,d] -> [R,m,d];
    rows without entries give zeros."""
    q = np.asarray(q, np.float64)
    k = np.asarray(k, np.float64)
    v = np.asarray(v, np.float64)
    m, n = q.shape[-2], k.shape[-2]
    scores = sddmm(m, n, row_indices, row_offsets, column_indices, q, k)
    weights = sparse_softmax_scaled(scores, row_indices, row_offsets, column_indices, scale)
    if q.ndim == 2:
        return spmm(m, n, weights, row_indices, row_offsets, column_indices, v)
    return spmm(m, n, weights, row_indices, row_offsets, column_indices, v).reshape(
        q.shape[0], m, v.shape[-1])


def dense_to_csr_many_mask(masks):
    """[b, m, n] 0/1 array -> (row_indices [b*m], row_offsets [b*(m+1)],
    column_indices [sum nnz], nnzs [b]).

    Restates tests/transformer/utils.py:17-38 / tests/test_attention_many_masks.py:54-73:
    per-mask ``to_sparse_csr`` topologies concatenated, every mask's offsets
    starting at 0, ``row_indices = diffsort(row_offsets)`` per mask.
    """
    masks = np.asarray(masks)
    assert masks.ndim == 3
    ri, ro, ci, nnzs = [], [], [], []
    for mask in masks:
        _, _, offsets, cols = dense_to_csr(mask)
        ri.append(diffsort(offsets))
        ro.append(offsets)
        ci.append(cols)
        nnzs.append(cols.shape[0])
    cat = lambda xs: np.concatenate(xs).astype(np.int32) if xs else np.zeros(0, np.int32)
    return cat(ri), cat(ro), cat(ci), np.asarray(nnzs, dtype=np.int64)


def _split_many_mask(b, m, nnzs, row_indices, row_offsets, column_indices):
    nnzs = [int(x) for x in np.asarray(nnzs).reshape(-1)]
    assert len(nnzs) == b
    row_indices = None if row_indices is None else np.asarray(row_indices).reshape(-1)
    row_offsets = np.asarray(row_offsets).reshape(-1)
    column_indices = np.asarray(column_indices).reshape(-1)
    assert row_offsets.shape[0] == b * (m + 1) and column_indices.shape[0] == sum(nnzs)
    first = np.concatenate(([0], np.cumsum(nnzs))).astype(np.int64)
    for i in range(b):
        yield (i, nnzs[i],
               None if row_indices is None else row_indices[i * m:(i + 1) * m],
               row_offsets[i * (m + 1):(i + 1) * (m + 1)],
               column_indices[first[i]:first[i + 1]])


def _heads(b, replicas):
    assert b > 0 and replicas % b == 0
    return replicas // b


def spmm_many_mask(b, m, k, nnzs, values, row_indices, row_offsets, column_indices, dense):
    """Replica r multiplies with mask ``r // heads`` (tests/transformer/functions.py:20;
    tests/test_attention_many_masks.py:143-150).  values [R, >= max nnz] (a
    replica's own ``nnzs[mask]`` leading entries count), dense [R,k,n] -> [R,m,n]."""
    values = np.asarray(values, np.float64)
    dense = np.asarray(dense, np.float64)
    heads = _heads(b, dense.shape[0])
    out = np.zeros((dense.shape[0], m, dense.shape[-1]))
    for i, nnz, ri, ro, ci in _split_many_mask(b, m, nnzs, row_indices, row_offsets, column_indices):
        sl = slice(i * heads, (i + 1) * heads)
        out[sl] = spmm(m, k, values[sl, :nnz], ri, ro, ci, dense[sl])
    return out


def sddmm_many_mask(b, m, n, nnzs, row_indices, row_offsets, column_indices, lhs, rhs):
    """tests/transformer/functions.py:135; tests/test_attention_many_masks.py:120-127.
    Returns [R, max nnz]; entries past a replica's own count are 0."""
    lhs = np.asarray(lhs, np.float64)
    rhs = np.asarray(rhs, np.float64)
    heads = _heads(b, lhs.shape[0])
    width = int(max([int(x) for x in np.asarray(nnzs).reshape(-1)] + [0]))
    out = np.zeros((lhs.shape[0], width))
    for i, nnz, ri, ro, ci in _split_many_mask(b, m, nnzs, row_indices, row_offsets, column_indices):
        sl = slice(i * heads, (i + 1) * heads)
        out[sl, :nnz] = sddmm(m, n, ri, ro, ci, lhs[sl], rhs[sl])
    return out


def sparse_softmax_many_mask(b, m, nnzs, values, row_indices, row_offsets, column_indices,
                             scale=1.0):
    """tests/transformer/functions.py:81; tests/test_attention_many_masks.py:132-138."""
    values = np.asarray(values, np.float64)
    heads = _heads(b, values.shape[0])
    out = np.zeros_like(values)
    for i, nnz, ri, ro, ci in _split_many_mask(b, m, nnzs, row_indices, row_offsets, column_indices):
        sl = slice(i * heads, (i + 1) * heads)
        out[sl, :nnz] = sparse_softmax_scaled(values[sl, :nnz], ri, ro, ci, scale)
    return out


def sparse_softmax_backward_many_mask(b, m, nnzs, softmax_out, grad_out, row_offsets, scale=1.0):
    y = np.asarray(softmax_out, np.float64)
    g = np.asarray(grad_out, np.float64)
    heads = _heads(b, y.shape[0])
    out = np.zeros_like(y)
    nnz_list = [int(x) for x in np.asarray(nnzs).reshape(-1)]
    row_offsets = np.asarray(row_offsets).reshape(-1)
    for i, nnz in enumerate(nnz_list):
        sl = slice(i * heads, (i + 1) * heads)
        out[sl, :nnz] = sparse_softmax_backward(y[sl, :nnz], g[sl, :nnz],
                                                row_offsets[i * (m + 1):(i + 1) * (m + 1)], scale)
    return out


def csr_transpose_many_mask(b, m, n, nnzs, values, row_offsets, column_indices):
    """tests/transformer/functions.py:50,165 -> (values_t [R, width],
    row_offsets_t [b, n+1] (indexed per mask by diffsort_many_mask,
    tests/transformer/utils.py:51-62), column_indices_t [sum nnz])."""
    values = np.asarray(values)
    heads = _heads(b, values.shape[0])
    values_t = np.zeros_like(values)
    ro_t, ci_t = [], []
    for i, nnz, _, ro, ci in _split_many_mask(b, m, nnzs, None, row_offsets, column_indices):
        sl = slice(i * heads, (i + 1) * heads)
        vt, rt, ct = csr_transpose(m, n, values[sl, :nnz], ro, ci)
        values_t[sl, :nnz] = vt
        ro_t.append(rt)
        ci_t.append(ct)
    return (values_t, np.stack(ro_t).astype(np.int32),
            np.concatenate(ci_t).astype(np.int32) if ci_t else np.zeros(0, np.int32))


# --------------------------------------------------------------------------
# dense definitions, exactly as the reference's tests state them
# --------------------------------------------------------------------------
def csr_to_dense(m, n, values, row_offsets, column_indices):
    row_offsets = np.asarray(row_offsets).astype(np.int64)
    out = np.zeros((m, n), dtype=np.float64)
    out[_rows_of(row_offsets), np.asarray(column_indices).astype(np.int64)] = values
    return out


def dense_spmm(sparse_dense, dense):
    """tests/test_spmm.py:9-10 -- ``torch.matmul(sparse, dense)``."""
    return np.matmul(np.asarray(sparse_dense, np.float64), np.asarray(dense, np.float64))


def dense_sddmm(mask, lhs, rhs):
    """tests/test_sddmm_3d.py:9-14 -- ``lhs @ rhs^T`` with masked_fill_(mask==0, 0)."""
    out = np.matmul(np.asarray(lhs, np.float64), np.swapaxes(np.asarray(rhs, np.float64), -2, -1))
    return np.where(np.asarray(mask) == 0, 0.0, out)


def dense_softmax(matrix):
    """tests/test_softmax.py:9-22 -- zeros -> -1e9, then softmax over the last dim."""
    x = np.asarray(matrix, np.float64).copy()
    x[x == 0] = -1e9
    x = x - x.max(axis=-1, keepdims=True)
    e = np.exp(x)
    return e / e.sum(axis=-1, keepdims=True)
