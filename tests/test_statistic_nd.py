"""SURVEY.md §8 row a12 — the legacy vaexfast.statisticNd_f8 pass.
CPU: the C restatement (oracle.statistic_nd) is pinned against the reference's own compiled vaexfast
(oracle/_ref) for every op / dimensionality / edges mode.  GPU: vaex_amd.vaexfast.statisticNd_f8 (HIP
kernels through the C-ABI) against the restatement: counts and min/max exact, fp64 sums within 1e-12."""
import numpy as np
import pytest

from oracle import oracle

OPS = {0: "add1", 1: "count", 2: "min_max", 3: "moments_01", 4: "moments_012"}


def make_case(seed, nd, n, op, use_edges, sizes=None):
    rng = np.random.default_rng(seed)
    sizes = sizes or [7, 5, 4, 6][:nd]
    blocks = []
    for d in range(nd):
        b = rng.normal(0.0, 1.5, n)
        b[rng.random(n) < 0.02] = np.nan
        blocks.append(b)
    w = rng.normal(1.0, 3.0, n)
    w[rng.random(n) < 0.05] = np.nan
    minima = [-2.0 - 0.25 * d for d in range(nd)]
    maxima = [2.5 + 0.5 * d for d in range(nd)]
    fields = oracle.STAT_FIELDS[op]
    grid = np.zeros(tuple(sizes) + (fields,), dtype=np.float64)
    if op == 2:
        grid[..., 0] = np.inf
        grid[..., 1] = -np.inf
    weights = None if op == 0 else [w]
    return blocks, weights, grid, minima, maxima


CASES = [(nd, op, e) for nd in (0, 1, 2, 3, 4) for op in OPS for e in (0, 1)]


@pytest.mark.parametrize("nd,op,use_edges", CASES)
def test_restatement_pinned_to_reference_vaexfast(nd, op, use_edges):
    vf = oracle.ref_module("vaexfast")
    if vf is None:
        pytest.skip("oracle/_ref/vaexfast not built (reference sources absent)")
    blocks, weights, grid, minima, maxima = make_case(100 + nd * 10 + op, nd, 5000, op, use_edges)
    want = grid.copy()
    got = grid.copy()
    # two chunks: the grid is accumulated into, not overwritten
    for i1, i2 in ((0, 2000), (2000, 5000)):
        bs = [b[i1:i2] for b in blocks]
        ws = None if weights is None else [w[i1:i2] for w in weights]
        vf.statisticNd_f8(bs, ws, want, [float(m) for m in minima], [float(m) for m in maxima], op, use_edges)
        oracle.statistic_nd(bs, ws, got, minima, maxima, op, use_edges)
    np.testing.assert_array_equal(got, want)  # same scalar loop, same order: bit-exact
    if op == 0 and not use_edges and nd:
        assert 0 < want.sum() < 5000


def compare(got, want, op):
    if op in (0, 1, 2):
        np.testing.assert_array_equal(got, want)
    else:
        np.testing.assert_array_equal(got[..., 0], want[..., 0])
        # fp64 sums: a different accumulation order on the GPU; 1e-12 relative to sum|w| (|w| < 20, <= 2e5 rows)
        for f in range(1, got.shape[-1]):
            scale = max(1.0, np.abs(want[..., f]).max())
            np.testing.assert_allclose(got[..., f], want[..., f], rtol=1e-12, atol=1e-12 * scale)


@pytest.mark.gpu
@pytest.mark.parametrize("nd,op,use_edges", CASES)
def test_statistic_nd_hip_vs_oracle(sa, gpu_ready, nd, op, use_edges):
    from vaex_amd import vaexfast
    blocks, weights, grid, minima, maxima = make_case(300 + nd * 10 + op, nd, 20000, op, use_edges)
    want, got = grid.copy(), grid.copy()
    for i1, i2 in ((0, 7000), (7000, 20000)):
        bs = [b[i1:i2] for b in blocks]
        ws = None if weights is None else [w[i1:i2] for w in weights]
        oracle.statistic_nd(bs, ws, want, minima, maxima, op, use_edges)
        assert vaexfast.statisticNd_f8(bs, ws, got, minima, maxima, op, use_edges) is None
    compare(got, want, op)


@pytest.mark.gpu
def test_statistic_nd_device_blocks_and_big_endian(sa, gpu_ready):
    import torch
    from vaex_amd import vaexfast
    blocks, weights, grid, minima, maxima = make_case(77, 2, 200000, 4, 0, sizes=[64, 32])
    want = grid.copy()
    oracle.statistic_nd(blocks, weights, want, minima, maxima, 4, 0)
    got = grid.copy()
    vaexfast.statisticNd_f8([torch.from_numpy(b).cuda() for b in blocks], [torch.from_numpy(weights[0]).cuda()], got, minima, maxima, 4)
    compare(got, want, 4)
    got = grid.copy()
    vaexfast.statisticNd_f8([b.astype(">f8") for b in blocks], [weights[0].astype(">f8")], got, minima, maxima, 4)
    compare(got, want, 4)


@pytest.mark.gpu
def test_statistic_nd_rejects(sa, gpu_ready):
    from vaex_amd import vaexfast
    g = np.zeros((4, 1))
    x = np.zeros(8)
    with pytest.raises(NotImplementedError):  # OP_COV multiplies columns on the device: native byte order only
        vaexfast.statisticNd_f8([x], [x.astype(">f8"), x], np.zeros((4, 12)), [0.0], [1.0], vaexfast.OP_COV)
    with pytest.raises(ValueError):  # 2 columns need 2*2 + 2*4 fields
        vaexfast.statisticNd_f8([x], [x, x], np.zeros((4, 8)), [0.0], [1.0], vaexfast.OP_COV)
    with pytest.raises(TypeError):
        vaexfast.statisticNd_f8([x.astype("f4")], None, g, [0.0], [1.0], 0)
    with pytest.raises(ValueError):   # (fewer dimensions than blocks + 1; MORE are the reference's "one run of values per cell", below)
        vaexfast.statisticNd_f8([x], None, np.zeros(4), [0.0], [1.0], 0)
    with pytest.raises(RuntimeError):
        vaexfast.statisticNd_f8([x], None, np.zeros((4, 2))[:, ::2], [0.0], [1.0], 0)
    with pytest.raises(ValueError):
        vaexfast.statisticNd_f8([x], None, g, [0.0], [1.0], 1)


@pytest.mark.gpu
def test_grids_of_higher_rank_are_one_run_of_values_per_cell_like_the_reference(sa, gpu_ready):
    """The reference never compares the grid's rank with the number of blocks (src/vaexfast.cpp:197-213, :1485-1493): the dimensions behind
    the binned ones are one contiguous run per cell, field f lands at offset f of it — its own unittest hands a (10, 2) grid to a call without
    blocks (packages/vaex-core/vaex/test/cmodule.py:26-35)."""
    vf = oracle.ref_module("vaexfast")
    if vf is None:
        pytest.skip("oracle/_ref/vaexfast not built (reference sources absent)")
    from vaex_amd import vaexfast
    rng = np.random.default_rng(77)
    x = rng.uniform(0, 10, 5000)
    w = rng.normal(0, 1, 5000)
    w[::17] = np.nan
    for blocks, weights, shape, lim, op in (([], [w, w], (10, 2), ([], []), 0), ([], w, (3, 2), ([], []), 1), ([], w, (2, 2), ([], []), 2),
                                            ([x], w, (6, 2, 3), ([0.0], [10.0]), 1), ([x], None, (6, 2, 2), ([0.0], [10.0]), 0), ([x, w], w, (4, 5, 2, 2), ([0.0, -3.0], [10.0, 3.0]), 1)):
        want, got = np.zeros(shape), np.zeros(shape)
        if op == 2:
            want[..., 0], got[..., 0] = np.inf, np.inf
            want.reshape(-1)[1], got.reshape(-1)[1] = -np.inf, -np.inf
        vf.statisticNd_f8(blocks, weights, want, lim[0], lim[1], op)
        assert vaexfast.statisticNd_f8(blocks, weights, got, lim[0], lim[1], op) is None
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-9, err_msg=str((shape, op)))
        assert want.any()


def _cov_case(seed, nd, n, ncol):
    rng = np.random.default_rng(seed)
    blocks = [rng.normal(0, 1.5, n) for _ in range(nd)]
    for b in blocks:
        b[rng.random(n) < 0.02] = np.nan
    ws = []
    for c in range(ncol):
        w = rng.normal(c, 2.0, n)
        w[rng.random(n) < 0.1] = np.nan
        ws.append(w)
    sizes = [6, 5, 4][:nd]
    return blocks, ws, sizes, [-2.0] * nd, [2.5] * nd


@pytest.mark.gpu
@pytest.mark.parametrize("nd,ncol,use_edges", [(0, 2, 0), (1, 1, 0), (1, 2, 1), (2, 3, 0), (2, 4, 1)])
def test_op_cov_vs_reference_vaexfast(sa, gpu_ready, nd, ncol, use_edges):
    """OP_COV (src/vaexfast.cpp:1117-1153; df.cov): per cell counts, sums, pair counts and pair sums of products —
    counts exact, sums within 1e-12 of the summed magnitude, against the reference's own compiled vaexfast."""
    vf = oracle.ref_module("vaexfast")
    if vf is None:
        pytest.skip("oracle/_ref/vaexfast not built (reference sources absent)")
    from vaex_amd import vaexfast
    blocks, ws, sizes, minima, maxima = _cov_case(500 + nd * 10 + ncol, nd, 30000, ncol)
    fields = 2 * ncol + 2 * ncol * ncol
    shape = tuple(s + (3 if use_edges else 0) for s in sizes) + (fields,)
    want, got = np.zeros(shape), np.zeros(shape)
    for i1, i2 in ((0, 9000), (9000, 30000)):
        bs, wc = [b[i1:i2] for b in blocks], [w[i1:i2] for w in ws]
        vf.statisticNd_f8(bs, wc, want, minima, maxima, 5, use_edges)
        assert vaexfast.statisticNd_f8(bs, wc, got, minima, maxima, vaexfast.OP_COV, use_edges) is None
    N = ncol
    count_fields = list(range(N)) + list(range(2 * N, 2 * N + N * N))
    for f in range(fields):
        if f in count_fields:
            np.testing.assert_array_equal(got[..., f], want[..., f], err_msg=f"field {f}")
        else:
            scale = max(1.0, np.abs(want[..., f]).max())
            np.testing.assert_allclose(got[..., f], want[..., f], rtol=1e-12, atol=1e-12 * scale, err_msg=f"field {f}")
    assert want[..., 2 * N:].any()


@pytest.mark.gpu
@pytest.mark.parametrize("nd,use_edges", [(0, 0), (1, 0), (2, 1)])
def test_op_first_vs_reference_vaexfast(sa, gpu_ready, nd, use_edges):
    """OP_FIRST (src/vaexfast.cpp:1155-1166): value of the row with the smallest order value, distinct orders so that the
    winner is unique — identical grids."""
    vf = oracle.ref_module("vaexfast")
    if vf is None:
        pytest.skip("oracle/_ref/vaexfast not built (reference sources absent)")
    from vaex_amd import vaexfast
    n = 20000
    blocks, ws, sizes, minima, maxima = _cov_case(900 + nd, nd, n, 1)
    rng = np.random.default_rng(3)
    order = rng.permutation(n).astype("f8")
    order[rng.random(n) < 0.05] = np.nan
    shape = tuple(s + (3 if use_edges else 0) for s in sizes) + (2,)
    want = np.zeros(shape)
    want[..., 1] = np.inf
    got = want.copy()
    for i1, i2 in ((0, 12000), (12000, n)):
        bs, wc = [b[i1:i2] for b in blocks], [ws[0][i1:i2], order[i1:i2]]
        vf.statisticNd_f8(bs, wc, want, minima, maxima, 6, use_edges)
        vaexfast.statisticNd_f8(bs, wc, got, minima, maxima, vaexfast.OP_FIRST, use_edges)
    # the reference lets a NaN VALUE win (only the order is compared, :1160): the value column goes through the AggFirst passes
    # as its bit pattern, so the same rows win here
    np.testing.assert_array_equal(got[..., 1], want[..., 1])
    np.testing.assert_array_equal(got[..., 0], want[..., 0])


@pytest.mark.gpu
@pytest.mark.parametrize("nd,op,use_edges", CASES)
def test_statistic_nd_f4_hip_vs_reference_vaexfast(sa, gpu_ready, nd, op, use_edges):
    """statisticNd_f4: float32 blocks / weights with the reference's float32 scaling arithmetic ((value - min) * scale in float32;
    the product with the bin count in float32 in the two-dimensional loop, in double elsewhere: src/vaexfast.cpp:1185-1262) —
    against the reference's own compiled entry.  Half of the rows sit exactly on a 1/64 lattice, i.e. on and next to bin edges,
    where a double-precision scaling puts rows into other cells than the reference does."""
    vf = oracle.ref_module("vaexfast")
    if vf is None:
        pytest.skip("oracle/_ref/vaexfast not built (reference sources absent)")
    from vaex_amd import vaexfast
    n = 40000
    rng = np.random.default_rng(500 + nd * 10 + op)
    sizes = [7, 5, 4, 6][:nd]
    blocks = []
    for d in range(nd):
        b = rng.normal(0.0, 1.5, n).astype("f4")
        b[: n // 2] = (rng.integers(-200, 200, n // 2) / 64.0).astype("f4")
        b[rng.random(n) < 0.02] = np.nan
        blocks.append(b)
    w = rng.normal(1.0, 3.0, n).astype("f4")
    w[rng.random(n) < 0.05] = np.nan
    minima = [-2.0 - 0.3 * d for d in range(nd)]     # (-2.3, -2.6: not float32 numbers — the reference rounds them first)
    maxima = [2.5 + 0.7 * d for d in range(nd)]
    grid = np.zeros(tuple(sizes) + (oracle.STAT_FIELDS[op],), dtype=np.float64)
    if op == 2:
        grid[..., 0] = np.inf
        grid[..., 1] = -np.inf
    weights = None if op == 0 else [w]
    want, got = grid.copy(), grid.copy()
    for i1, i2 in ((0, 15000), (15000, n)):
        bs = [b[i1:i2] for b in blocks]
        ws = None if weights is None else [x[i1:i2] for x in weights]
        vf.statisticNd_f4(bs, ws, want, minima, maxima, op, use_edges)
        assert vaexfast.statisticNd_f4(bs, ws, got, minima, maxima, op, use_edges) is None
    compare(got, want, op)
    if nd and op == 0:
        assert want.sum() > 0
    with pytest.raises(TypeError, match="float32"):
        vaexfast.statisticNd_f4([b.astype("f8") for b in blocks] or [np.zeros(3)], None, np.zeros(tuple(sizes or [4]) + (1,)), minima or [0.0], maxima or [1.0], 0, 0)


@pytest.mark.gpu
def test_statistic_nd_f4_first_and_float32_scaling_differs_from_float64(sa, gpu_ready):
    vf = oracle.ref_module("vaexfast")
    if vf is None:
        pytest.skip("oracle/_ref/vaexfast not built (reference sources absent)")
    from vaex_amd import vaexfast
    rng = np.random.default_rng(9)
    n = 30000
    x = (rng.integers(-300, 300, n) / 100.0).astype("f4")
    y = (rng.integers(-300, 300, n) / 100.0).astype("f4")
    v = rng.normal(0, 1, n).astype("f4"); v[::37] = np.nan
    order = rng.permutation(n).astype("f4")
    want = np.zeros((10, 2)); want[..., 1] = np.inf
    got = want.copy()
    vf.statisticNd_f4([x], [v, order], want, [-3.0], [3.0], 6, 0)
    vaexfast.statisticNd_f4([x], [v, order], got, [-3.0], [3.0], 6, 0)
    np.testing.assert_array_equal(got, want)
    # the float32 arithmetic matters: the same rows binned with the float64 entry land in other cells
    a, b = np.zeros((30, 30, 1)), np.zeros((30, 30, 1))
    vaexfast.statisticNd_f4([x, y], None, a, [-3.0, -3.0], [3.0, 3.0], 0, 0)
    vf.statisticNd_f4([x, y], None, b, [-3.0, -3.0], [3.0, 3.0], 0, 0)
    np.testing.assert_array_equal(a, b)
    c = np.zeros((30, 30, 1))
    vaexfast.statisticNd_f8([x.astype("f8"), y.astype("f8")], None, c, [-3.0, -3.0], [3.0, 3.0], 0, 0)
    assert (a != c).any()
